"""Seat-permuted evaluation matches between agents (boardlaw/arena/common.py:52-106): every env is assigned one
permutation of the agents over the seats; each round every agent moves in the envs where it is its turn
(`agent(worlds[mask], eval=True)` -- variable batch size, argmax actions), until every env has finished one game.
Returns per permutation: names, wins per seat, moves, games, wall time."""
import math
import time
from itertools import permutations

import numpy as np
import torch

from . import arrdict


def matchup_patterns(n_seats):
    return torch.as_tensor(list(permutations(range(n_seats))))


def matchup_indices(n_envs, n_seats):
    patterns = matchup_patterns(n_seats)
    return patterns.repeat((n_envs // len(patterns), 1))


def _tally(seat_wins, n_moves, seconds, assignment, names, boardsize):
    """One record per seat permutation: who sat where, wins per seat, moves, games, wall time."""
    names = np.array(names)
    records = []
    for perm in matchup_patterns(assignment.shape[1]):
        rows = (assignment == perm).all(-1)
        w = seat_wins[rows].sum(0)
        records.append(arrdict.dotdict(names=tuple(names[perm]), wins=tuple(float(x) for x in w), moves=float(n_moves[rows].sum()),
                                       games=float(w.sum()), times=float(seconds[rows].sum()), boardsize=boardsize))
    return records


def evaluate(worlds, agents):
    """Every env plays ONE game; env e seats the agents in the order `assignment[e]` (all seat permutations, tiled over
    the batch).  A round lets each agent act, with argmax actions, in exactly the unfinished envs where it is to move."""
    roster = list(agents.items()) if isinstance(agents, dict) else list(agents)
    n_seats, n_envs, dev = worlds.n_seats, worlds.n_envs, worlds.device
    assert n_seats == 2, 'Only support 2 seats for now'
    assert n_envs % math.factorial(n_seats) == 0, 'Number of envs needs to be divisible by the number of permutations of seats'
    assert len(roster) == n_seats, 'Need to pass one agent per seat'
    assignment = matchup_indices(n_envs, n_seats).to(dev)
    everyone = torch.arange(n_envs, device=dev)
    done = torch.zeros(n_envs, dtype=torch.bool, device=dev)
    seat_wins = torch.zeros((n_envs, n_seats), dtype=torch.int, device=dev)
    n_moves = torch.zeros(n_envs, dtype=torch.int, device=dev)
    seconds = torch.zeros(n_envs, dtype=torch.float, device=dev)
    while not bool(done.all()):
        for who, (_, agent) in enumerate(roster):
            to_move = (assignment[everyone, worlds.seats.long()] == who) & ~done
            if not bool(to_move.any()):
                continue
            t0 = time.time()
            picks = agent(worlds[to_move], eval=True).actions
            worlds[to_move], outcome = worlds[to_move].step(picks)
            done[to_move] = outcome.terminal
            elapsed = time.time() - t0
            seat_wins[to_move] += (outcome.rewards == 1).int()
            n_moves[to_move] += 1
            seconds[to_move] += elapsed / to_move.sum()
    return _tally(seat_wins.cpu(), n_moves.cpu(), seconds.cpu(), assignment.cpu(), [name for name, _ in roster],
                  getattr(worlds, 'boardsize', None))


# --------------------------------------------------------------------------------------------------------------------
# All-vs-all evaluation (boardlaw/arena/neural.py:46-200): thousands of envs at once, one env per still-needed game of an
# ordered pair (black agent, white agent); every step the agent with the most envs waiting for it moves in all of them.
# --------------------------------------------------------------------------------------------------------------------
def live_indices(residual):
    """(n,n) counts -> (sum(residual), 2) rows [i, j], residual[i, j] copies of each pair, pairs in row-major order."""
    residual = torch.as_tensor(residual).int().clamp(min=0)     # a pair with more games than asked for gets no env (neural.py:40-44)
    assert int(residual.sum()) < 100 * 1024 * 1024
    pairs = residual.nonzero(as_tuple=False)
    return pairs.repeat_interleave(residual[pairs[:, 0], pairs[:, 1]].long(), 0)


class Tracker:
    """Who plays whom in which env.  live[e] = (agent at seat 0, agent at seat 1), or (-1,-1) once env e's game is over."""

    def __init__(self, n_envs_per, games, names=None, max_dispatch=32 * 1024, device='cuda'):
        """games: (n,n) games already played per ordered pair -- a pandas DataFrame indexed by name on both axes (as the
        reference passes) or an array with `names`.  The diagonal is never played."""
        if hasattr(games, 'index'):
            assert list(games.index) == list(games.columns)
            names, games = list(games.index), games.values
        games = np.array(games, dtype=np.int64, copy=True)
        assert names is not None and games.shape == (len(names), len(names))
        games[np.diag_indices_from(games)] = n_envs_per
        self.names, self.n_envs_per, self.max_dispatch = list(names), n_envs_per, max_dispatch
        self.live = live_indices(n_envs_per - torch.as_tensor(games)).to(device)
        self.n_envs = len(self.live)

    def report(self):
        over = (self.live == -1).any(1)
        return int(over.sum()), int((~over).sum())

    def finished(self):
        return bool((self.live == -1).all())

    def suggest(self, seats):
        """The agent with the most unfinished envs waiting for its move -> (name, (n_envs,) mask of those envs (at most
        max_dispatch of them), their (black, white) pairs)."""
        waiting = self.live.gather(1, seats.long()[:, None]).squeeze(1)
        counts = torch.zeros(len(self.names), dtype=torch.long, device=self.live.device)
        alive = waiting[waiting > -1]
        counts.scatter_add_(0, alive, torch.ones_like(alive))
        pick = int(counts.argmax())
        mask = waiting == pick
        mask = mask & (mask.cumsum(0) < self.max_dispatch)
        return self.names[pick], mask, self.live[mask]

    def update(self, terminal, mask):
        """Retires the envs of `mask` whose step ended the game; returns their pairs."""
        ended = torch.zeros_like(mask)
        ended[mask] = terminal
        pairs = self.live[ended]
        self.live[ended] = -1
        return pairs


def _pair_add(totals, pairs, values=1):
    """totals[i, j] += values for every row (i, j) of pairs (duplicates accumulate)."""
    flat = pairs[:, 0].long() * totals.shape[1] + pairs[:, 1].long()
    if not torch.is_tensor(values):
        values = torch.full((len(flat),), values, dtype=totals.dtype, device=totals.device)
    totals.view(-1).scatter_add_(0, flat, values.to(totals.dtype))


class ChunkEvaluator:

    def __init__(self, worldfunc, agents, games=None, n_envs_per=1024, device='cuda'):
        self.agents = agents
        names = list(agents)
        if games is None:
            games = np.zeros((len(names), len(names)), np.int64)
        elif hasattr(games, 'index'):
            assert set(games.index) == set(names) and set(games.columns) == set(names)
            games = games.reindex(index=names, columns=names)
        self.tracker = Tracker(n_envs_per, games, names=names, device=device)
        self.worlds = worldfunc(self.tracker.n_envs).to(device)
        n = len(names)
        self.wins = torch.zeros((n, n, self.worlds.n_seats), dtype=torch.int, device=device)
        self.moves = torch.zeros((n, n), dtype=torch.int, device=device)
        self.times = torch.zeros((n, n), dtype=torch.float, device=device)
        self.steps = 0
        self.start = time.time()

    def finished(self):
        return self.tracker.finished()

    def record(self, transitions, pairs, seconds):
        won = (transitions.rewards == 1).int()
        for seat in range(self.wins.shape[-1]):
            _pair_add(self.wins[:, :, seat], pairs, won[:, seat])
        _pair_add(self.moves, pairs, 1)
        _pair_add(self.times, pairs, seconds / transitions.terminal.shape[0])
        # a pair is reported once, when its last game has ended
        complete = (self.wins.sum(-1) == self.tracker.n_envs_per).nonzero(as_tuple=False).cpu()
        results = []
        for i, j in complete.tolist():
            w = self.wins[i, j].cpu()
            results.append(arrdict.dotdict(names=(self.tracker.names[i], self.tracker.names[j]), wins=tuple(float(x) for x in w),
                                           moves=float(self.moves[i, j]), games=float(w.sum()), times=float(self.times[i, j]),
                                           boardsize=getattr(self.worlds, 'boardsize', None)))
            self.wins[i, j] = -1
        return results

    def step(self):
        name, mask, pairs = self.tracker.suggest(self.worlds.seats)
        self.steps += 1
        t0 = time.time()
        decisions = self.agents[name](self.worlds[mask])
        self.worlds[mask], transitions = self.worlds[mask].step(decisions.actions)
        seconds = time.time() - t0
        self.tracker.update(transitions.terminal, mask)
        return self.record(transitions, pairs, seconds)


def evaluate_chunk(worldfunc, agentfunc, subgames, n_envs_per):
    """arena/neural.py:193-200: play out every missing game of a block of the games matrix."""
    names = list(subgames.index) if hasattr(subgames, 'index') else list(subgames)
    evaluator = ChunkEvaluator(worldfunc, {n: agentfunc(n) for n in names}, subgames if hasattr(subgames, 'index') else None,
                               n_envs_per=n_envs_per)
    results = []
    while not evaluator.finished():
        results.extend(evaluator.step())
    return results




# --------------------------------------------------------------------------------------------------------------------
# Fan-out over GPUs (boardlaw/arena/neural.py:202-274 `evaluate_gen`, rebar/parallel.py:28-57 `CUDAPoolExecutor`): the games
# matrix is cut into diagonal and skew blocks of agents, every block is an `evaluate_chunk` job, and a pool of worker
# processes -- worker n pinned to GPU n % n_gpus, one process per GPU when n_workers == n_gpus -- plays them.  No collective:
# jobs are independent, results come back through a queue.
# --------------------------------------------------------------------------------------------------------------------
def chunk_jobs(games, names, n_envs_per, chunks):
    """neural.py:205-232: {(i, j): (names of the block, games played so far within it)} for every diagonal block i == j and every
    pair of blocks i < j that still has games to play.  Within a skew block the two diagonal sub-blocks count as played."""
    games = np.asarray(games)
    names = list(names)
    if isinstance(chunks, int):
        chunks = [list(range(i, min(i + chunks, len(names)))) for i in range(0, len(names), chunks)]
    else:
        chunks = [[names.index(n) for n in c] for c in chunks]
    jobs = {}
    for i, c in enumerate(chunks):
        sub = games[np.ix_(c, c)].copy()
        sub[np.diag_indices_from(sub)] = n_envs_per
        if (sub < n_envs_per).any():
            jobs[i, i] = ([names[k] for k in c], sub)
    for i in range(len(chunks)):
        for j in range(i + 1, len(chunks)):
            both = chunks[i] + chunks[j]
            sub = games[np.ix_(both, both)].copy()
            a = len(chunks[i])
            sub[:a, :a] = n_envs_per
            sub[a:, a:] = n_envs_per
            if (sub < n_envs_per).any():
                jobs[i, j] = ([names[k] for k in both], sub)
    return jobs


def _chunk_job(worldfunc, agentfunc, names, played, n_envs_per):
    agents = {n: agentfunc(n) for n in names}
    device = 'cuda' if torch.cuda.is_available() else 'cpu'
    evaluator = ChunkEvaluator(worldfunc, agents, played, n_envs_per=n_envs_per, device=device)     # games already on record are not replayed
    results = []
    while not evaluator.finished():
        results.extend(evaluator.step())
    return [dict(r) for r in results]


def _pool_worker(index, n_devices, inbox, outbox):
    """One worker process: pinned to GPU index % n_devices (rebar/parallel.py:31-35 does this with CUDA_VISIBLE_DEVICES before
    CUDA comes up; here the runtime is initialised lazily, so selecting the device is enough and the ids stay global)."""
    if n_devices > 0:
        torch.cuda.set_device(index % n_devices)
    while True:
        item = inbox.get()
        if item is None:
            return
        key, fn, args = item
        try:
            with torch.no_grad():
                outbox.put((key, fn(*args), None))
        except BaseException as e:           # the parent re-raises
            import traceback
            outbox.put((key, None, f'{type(e).__name__}: {e}\n{traceback.format_exc()}'))


def workers_per_gpu(n_envs):
    """How many matches to keep in flight on one GPU.  A match of up to 2048 envs leaves the chip mostly idle (both search kernels
    are bound by the latency of one env's chain): independent matches side by side, one PROCESS each (their
    own streams, generators and captured moves), reach 18 -> 34 -> 43 M sims/s at 1 / 2 / 3 per MI355X and fall back at 4
    (profiles/r03_arena_sweep.txt); from 4096 envs a second one still adds half, beyond that one fills the chip."""
    return 3 if n_envs <= 2048 else (2 if n_envs <= 4096 else 1)


def lpt_partition(costs, n):
    """Longest-processing-time-first: indices of `costs` dealt, largest first, to whichever of n bins is lightest -- the board sizes
    of a sweep over the workers of a pool.  Returns n lists of indices (each in dealing order: its largest job first)."""
    bins, loads = [[] for _ in range(n)], [0.] * n
    for i in sorted(range(len(costs)), key=lambda i: -costs[i]):
        k = loads.index(min(loads))
        bins[k].append(i); loads[k] += costs[i]
    return bins


def run_jobs(jobs, n_workers=None, context='spawn', poll_seconds=2.0, per_device=1):
    """jobs: {key: (picklable function, args)} -> yields (key, result) as they finish, from a pool of n_workers processes, by
    default `per_device` per GPU (worker n on GPU n % n_gpus; evaluate_gen passes workers_per_gpu(n_envs_per)).  n_workers == 0:
    everything in this process, in order (the reference's serial executor, rebar/parallel.py:15-27)."""
    n_devices = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_workers is None:
        n_workers = max(n_devices, 1) * max(int(per_device), 1)
    if n_workers == 0:
        for key, (fn, args) in jobs.items():
            yield key, fn(*args)
        return
    import torch.multiprocessing as mp
    ctx = mp.get_context(context)
    inbox, outbox = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_pool_worker, args=(i, n_devices, inbox, outbox), daemon=True) for i in range(min(n_workers, max(len(jobs), 1)))]
    for p in procs:
        p.start()
    try:
        for key, (fn, args) in jobs.items():
            inbox.put((key, fn, args))
        for _ in procs:
            inbox.put(None)
        import queue
        outstanding = len(jobs)
        while outstanding:
            try:
                key, result, err = outbox.get(timeout=poll_seconds)
            except queue.Empty:
                # a worker that died without posting (a crash inside the native library, an OOM kill) would leave this loop
                # waiting for ever: the reference's ProcessPoolExecutor raises BrokenProcessPool there (rebar/parallel.py:28-57)
                dead = [(i, p.exitcode) for i, p in enumerate(procs) if not p.is_alive() and p.exitcode not in (0, None)]
                if dead or not any(p.is_alive() for p in procs):
                    raise RuntimeError(f'arena worker(s) died with {outstanding} job(s) outstanding: (worker, exit code) {dead}')
                continue
            if err is not None:
                raise RuntimeError(f'arena job {key} failed in its worker:\n{err}')
            outstanding -= 1
            yield key, result
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()


def evaluate_gen(worldfunc, agentfunc, games, names=None, n_envs_per=512, chunks=64, n_workers=None, context='spawn'):
    """neural.py:202-274: plays every missing game of an all-vs-all `games` matrix ((n,n) games played per ordered pair; a pandas
    DataFrame indexed by name, or an array with `names`), block by block, on a pool of worker processes (one per GPU by default).
    Yields (list of result records of one finished block, running totals) as blocks finish.  worldfunc(n_envs) and
    agentfunc(name) must be picklable (module-level functions); they run in the workers.  n_workers None: workers_per_gpu(n_envs_per)
    processes per GPU -- blocks of up to 2048 games run three side by side on a GPU."""
    if hasattr(games, 'index'):
        assert list(games.index) == list(games.columns)
        names, games = list(games.index), games.values
    jobs = {k: (_chunk_job, (worldfunc, agentfunc, block_names, played, n_envs_per)) for k, (block_names, played) in chunk_jobs(games, names, n_envs_per, chunks).items()}
    stats = arrdict.dotdict(finished=0, total=len(jobs), moves=0., games=0., matchups=0, start=time.time())
    for key, records in run_jobs(jobs, n_workers, context, per_device=workers_per_gpu(n_envs_per)):
        results = [arrdict.dotdict(r) for r in records]
        stats['finished'] += 1
        stats['end'] = time.time()
        for r in results:
            stats['moves'] += r.moves; stats['games'] += r.games; stats['matchups'] += 1
        yield results, arrdict.dotdict(stats)


def _match_pool_worker(index, device, factory, factory_args, inbox, outbox):
    try:
        if device is not None and torch.cuda.is_available():
            torch.cuda.set_device(device)
        player = factory(*factory_args)
        outbox.put((index, 'ready', None))
    except BaseException as e:
        import traceback
        outbox.put((index, None, f'{type(e).__name__}: {e}\n{traceback.format_exc()}'))
        return
    while True:
        jobs = inbox.get()
        if jobs is None:
            return
        try:
            with torch.no_grad():
                outbox.put((index, [player(j) for j in jobs], None))
        except BaseException as e:
            import traceback
            outbox.put((index, None, f'{type(e).__name__}: {e}\n{traceback.format_exc()}'))


class MatchPool:
    """n_workers PERSISTENT worker processes on one device, each holding its own `player = factory(*factory_args)` -- its agents and
    the moves it has captured stay alive between calls, so a sweep that is played again and again (bench.py --config 5) pays process
    start-up and capture once.  play(assignments): assignments[k] = list of jobs for worker k (played in that order, player(job));
    returns the list of result lists.  The matches of different workers overlap on the GPU; nothing is exchanged between them."""

    def __init__(self, factory, factory_args=(), n_workers=1, device=None, context='spawn', start_timeout=600.):
        import torch.multiprocessing as mp
        ctx = mp.get_context(context)
        self.outbox = ctx.Queue()
        self.inboxes = [ctx.Queue() for _ in range(n_workers)]
        self.procs = [ctx.Process(target=_match_pool_worker, args=(i, device, factory, factory_args, self.inboxes[i], self.outbox), daemon=True)
                      for i in range(n_workers)]
        for p in self.procs:
            p.start()
        for _ in self.procs:
            self._get(start_timeout)

    def _get(self, timeout):
        import queue
        waited = 0.
        while True:
            try:
                index, result, err = self.outbox.get(timeout=2.0)
            except queue.Empty:
                waited += 2.0
                dead = [(i, p.exitcode) for i, p in enumerate(self.procs) if not p.is_alive()]
                if dead or waited > timeout:
                    self.close()
                    raise RuntimeError(f'arena match pool: worker(s) died or timed out: (worker, exit code) {dead}')
                continue
            if err is not None:
                self.close()
                raise RuntimeError(f'arena match pool: worker {index} failed:\n{err}')
            return index, result

    def play(self, assignments, timeout=3600.):
        assert len(assignments) == len(self.procs)
        for inbox, jobs in zip(self.inboxes, assignments):
            inbox.put(list(jobs))
        results = [None] * len(self.procs)
        for _ in self.procs:
            index, result = self._get(timeout)
            results[index] = result
        return results

    def close(self):
        for inbox in self.inboxes:
            try:
                inbox.put(None)
            except Exception:
                pass
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


from .analysis import rollout  # noqa: E402,F401  (kept here for callers that imported it from arena)
