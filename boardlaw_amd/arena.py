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
    residual = torch.as_tensor(residual).int()
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


from .analysis import rollout  # noqa: E402,F401  (kept here for callers that imported it from arena)
