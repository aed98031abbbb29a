"""bench.py -- MCTS simulations/sec on 9x9 Hex, 4096 envs x 64 sims per move (BASELINE.json config 2), per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,4,5}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

--config picks another BASELINE.json configuration (the metric is quoted on 2, the default; 3 is 2 on every GPU, i.e. --gpus 8):
  1  5x5, 64 envs x 16 sims, FCModel 16x4 (the reference's CPU-runnable plumbing case)
  4  13x13, 1024 envs per GPU x 256 sims, FCModel 1024x8, actor + learner: a step is one self-play move of the shard plus one
     learner step on a 64-move buffer (boardlaw/main.py:147-200 at steady state), gradients averaged over ranks by one RCCL
     all-reduce of a flat bucket (parallel.GradientBucket)
  5  the arena sweep: boards 3..11, 2048 games each between two 64-sim 512x4 search agents through arena.evaluate's masked calls
     (boardlaw/arena/common.py:75-106); a step is one sweep over the nine board sizes, four matches in flight per GPU (persistent
     worker processes, arena.MatchPool; --arena-workers)
each with the same one-line JSON (roofline of bl_sim_expand at that shape, cpu_baseline on a bounded sample of that shape).

A "step" is one self-play move of the whole batch: MCTSAgent(worlds) -- root evaluation + 63 simulations per env, i.e.
T = 64 network evaluations per env (SURVEY 8d counts the root as a simulation) -- followed by worlds.step(actions).
Inputs are synthetic and resident in HBM before the clock starts: Hex.initial pre-mixed with floor(81/3) random legal
moves (seeded), FCModel 512x4 with default init under manual_seed(0), fp16 autocast in simulate like the reference.
Multi-GPU: every rank owns an independent shard of 4096 envs and a network replica (self-play has no data-path
collective; each shard normalises q over its own envs -- "replicas" semantics, SURVEY 8e option 1) => weak scaling.

--envs N times another batch per GPU (the metric is quoted at 4096); the default line also carries `config.value_32k_envs`: the same
moves at 32768 envs per GPU, the reference's own actor shape (boardlaw/main.py:147), with its own roofline object.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     the dominant kernel (bl_sim_expand: descend+expand+step+observe), algorithmic bytes per launch (model
               of SURVEY 8d with d,k measured by the kernel's own counters) / its mean duration from HIP events
               recorded around every launch inside the timed region, vs the 8 TB/s HBM peak.
  cpu_baseline the C oracle (kind "port") driving the same search on this host: one process per physical core with
               4096/P envs each for ~11 s (sims/s summed; P and nproc stated), plus one process alone and the -O0 build
               (how the reference JIT-builds its sources) for ns/descent.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BOARD, ENVS, NODES, WIDTH, DEPTH = 9, 4096, 64, 512, 4
HBM_PEAK_GBS = 8000.0


def emit(line):
    """The run's ONE JSON line, as the LAST thing on stdout: RCCL prints its version banner through C stdio, which (not a tty) holds
    it back until exit -- after Python's line -- unless it is flushed first."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


def premix(worlds, moves, gen):
    """floor(S^2/3) uniformly random legal moves per env; the reference's learning.mix plays 2500."""
    for _ in range(moves):
        valid = worlds.valid
        r = torch.rand(valid.shape, device=valid.device, generator=gen) * valid
        worlds, _ = worlds.step(r.argmax(-1), check=False)
    return worlds


class TimedExpand:
    """Wraps one of the library's entry points (bl_sim_expand unless named) so that every launch inside the timed region is
    bracketed by HIP events on the stream it is launched on (torch's current stream)."""

    def __init__(self, lib, name='bl_sim_expand'):
        self.lib, self.orig = lib, getattr(lib, name)
        self.pairs, self.on = [], False

    def __call__(self, *args):
        if not self.on:
            return self.orig(*args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = self.orig(*args)
        b.record()
        self.pairs.append((a, b))
        return rc

    def mean_us(self):
        return 1e3 * float(np.mean([a.elapsed_time(b) for a, b in self.pairs]))


def tree_statistics(worlds, net, nodes):
    """d (policy evaluations / descent), k (expanded-child lookups / descent), Newton iterations / evaluation: measured
    by the counting variant of the kernel on one untimed search."""
    from boardlaw_amd.mcts import MCTS
    powf = os.environ.pop('BL_POWF_LIBM', None)      # the counting build exists for the default parity target only (same statistics)
    try:
        m = MCTS(worlds, n_nodes=nodes, count=True)
    finally:
        if powf is not None:
            os.environ['BL_POWF_LIBM'] = powf
    m.initialize(net)
    for _ in range(nodes - 1):
        m.simulate(net)
    c = m.counters.cpu().numpy().astype(np.float64).sum(0)
    descents = worlds.n_envs * (nodes - 1)
    return c[0] / descents, c[3] / descents, c[1] / max(c[0], 1)


def expand_bytes_per_env(A, S, d, k):
    """Algorithmic HBM bytes bl_sim_expand moves per env per launch (terms of SURVEY 8d that belong to this kernel):
    per visited node children row 2A + logits row 2A + seat 2(4 here) + terminal 1 + rand 2; per expanded child w 2 +
    n 2; expansion: parent board A + leaf board A + obs 8A + valid A + seats r/w 8 + children/parents/relation 6 +
    rewards 2S + terminal 1 + leaf id 2 + leaf seat 4."""
    return d * (4 * A + 5) + 4 * k + (11 * A + 8 + 6 + 2 * S + 1 + 2 + 4)


def total_bytes_per_sim(A, S, T, d, k):
    """SURVEY 8d's whole-path figure: bytes/sim = d(4A+5) + 4k + f(2S+2) + (d+1)(6S+7) + 13A + 6S + 19, f = (T+1)/2."""
    return d * (4 * A + 5) + 4 * k + (T + 1) / 2 * (2 * S + 2) + (d + 1) * (6 * S + 7) + 13 * A + 6 * S + 19


def cpu_baseline(seconds_budget=24.0, envs=ENVS, single_envs=256):
    """The same search on this box's host cores with the C oracle's kernels (tests/cpu_driver.py): every physical core,
    one process each, plus a single process and the -O0 build for ns/descent."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    try:
        from cpu_driver import run_cpu_baseline
        return run_cpu_baseline(BOARD, NODES, WIDTH, DEPTH, envs, seconds_budget, single_envs=single_envs)
    except Exception as e:  # pragma: no cover
        return {'value': None, 'unit': 'sims/s', 'cores': 0, 'kind': 'port', 'sample': f'unavailable: {type(e).__name__}: {e}'}


def measure_traffic(kernel='sim_expand', timeout=240):
    """HBM-side bytes per launch of the dominant kernel, observed in THIS run: two separate rocprofv3 --pmc passes (FETCH_SIZE,
    WRITE_SIZE; --kernel-trace only, no other trace domain) over `bench.py --steps 2 --warmup 1 --timed-only`, from /tmp.
    Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes and profiles/r02_calib_* confirmed on this
    device: both counters are in KB; on gfx950 FETCH_SIZE tallies the 128-B requests of coalesced reads at 64 B, i.e. reports
    half of their bytes (this kernel's reads are rows and lane-parallel slot statistics: coalesced), WRITE_SIZE is exact.
    traffic = 2 x FETCH_SIZE + WRITE_SIZE.  Returns (bytes or None, description)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not on PATH'
    means = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        out = tempfile.mkdtemp(prefix=f'bl_pmc_{counter}_', dir='/tmp')
        cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '-d', out, '--output-format', 'csv', '--', sys.executable,
               os.path.abspath(__file__), '--steps', '2', '--warmup', '1', '--timed-only']
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
        env['TMPDIR'] = '/tmp'
        try:
            subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, timeout=timeout, check=True)
            total, n = 0.0, 0
            for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if kernel in row.get('Kernel_Name', '') and row.get('Counter_Name') == counter:
                            total += float(row['Counter_Value']); n += 1
            if n == 0:
                return None, f'no {counter} rows for *{kernel}* in the rocprofv3 output'
            means[counter] = (total / n, n)
        except Exception as e:  # pragma: no cover
            return None, f'rocprofv3 --pmc {counter} failed: {type(e).__name__}: {str(e)[:200]}'
        finally:
            shutil.rmtree(out, ignore_errors=True)
    (f_kb, n_f), (w_kb, n_w) = means['FETCH_SIZE'], means['WRITE_SIZE']
    return (2 * f_kb + w_kb) * 1024, (f'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `bench.py --steps 2 '
                                       f'--warmup 1 --timed-only`, mean over {n_f} / {n_w} launches: FETCH_SIZE {f_kb:.1f} KB (x2: gfx950 counts coalesced 128-B requests '
                                       f'at 64 B), WRITE_SIZE {w_kb:.1f} KB')


def rate_of(agent, worlds, steps, warmup, barrier):
    """sims/s of `steps` self-play moves by `agent` from `worlds` (captured on the first call, then 1-2 warm-up moves)."""
    w = worlds
    for _ in range(1 + min(warmup, 2)):
        w = agent.play(w)[1]
    barrier()
    t = time.perf_counter()
    for _ in range(steps):
        w = agent.play(w)[1]
    barrier()
    return worlds.n_envs * agent.kwargs['n_nodes'] * steps / (time.perf_counter() - t)


def operating_point(envs, net, lib, steps, warmup, barrier, rank):
    """The same self-play moves at another batch size -- called with 32768 envs per GPU, the reference's own actor shape
    (boardlaw/main.py:147: `n_envs=32*1024`; BASELINE.md: "default actor shape"), where the search kernels are bound by throughput
    (bl_sim_expand by VALU issue, bl_sim_infer_finish by the L2's bandwidth: DESIGN.md section 5) instead of one env's chain.
    Returns sims/s of captured moves and bl_sim_expand's own roofline object at that batch (HIP events around every launch of an
    eager re-run of the moves; d, k counted by the kernel)."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent, MoveRng
    gen = torch.Generator(device='cuda'); gen.manual_seed(5000 + rank)
    worlds = premix(Hex.initial(envs, BOARD), BOARD * BOARD // 3, gen)
    inf = networks.Inference(net, fused=True)
    agent = MCTSAgent(inf, n_nodes=NODES, graph=True, rng=MoveRng())
    rate = rate_of(agent, worlds, steps, warmup, barrier)
    del agent
    t_exp, t_fin = TimedExpand(lib), TimedExpand(lib, 'bl_sim_infer_finish')
    lib.bl_sim_expand, lib.bl_sim_infer_finish = t_exp, t_fin
    try:
        probe = MCTSAgent(inf, n_nodes=NODES, graph=False, rng=MoveRng())
        w = probe.play(worlds)[1]
        t_exp.on = t_fin.on = True
        for _ in range(2):
            w = probe.play(w)[1]
        torch.cuda.synchronize()
    finally:
        lib.bl_sim_expand, lib.bl_sim_infer_finish = t_exp.orig, t_fin.orig
    d, k, its = tree_statistics(worlds, net, NODES)
    A = BOARD * BOARD
    per_launch = expand_bytes_per_env(A, 2, d, k) * envs
    achieved = per_launch / (t_exp.mean_us() * 1e-6) / 1e9
    whole = total_bytes_per_sim(A, 2, NODES, d, k)
    del probe, w, worlds
    torch.cuda.empty_cache()
    return {'sims_per_sec': rate, 'envs_per_gpu': envs, 'ms_per_step': 1e3 * envs * NODES / rate, 'steps': steps,
            'bl_sim_expand_us': t_exp.mean_us(), 'bl_sim_infer_finish_us': t_fin.mean_us() if t_fin.pairs else None,
            'gpu_ns_per_descent': 1e3 * t_exp.mean_us() / envs,
            'd_policy_evals_per_descent': round(d, 3), 'k_child_lookups_per_descent': round(k, 3),
            'bytes_per_sim_whole_path': round(whole, 1), 'hbm_frac_whole_path': whole * rate / (HBM_PEAK_GBS * 1e9),
            'roofline': {'bound': 'hbm', 'kernel': 'bl::sim_expand2_kernel (bl_sim_expand), one wave per env', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'kernel_us': t_exp.mean_us(), 'bytes_per_launch': per_launch,
                         'launches_timed': len(t_exp.pairs),
                         'binds': 'VALU issue: 101.5 M VALU instructions per launch on 1024 SIMDs = 165 of 186 us (profiles/r06_pmc32k_SQ1.csv); '
                                  'HBM-side traffic 177 MB per launch (profiles/r06_pmc32k_FETCH_SIZE / WRITE_SIZE.csv) = 0.95 TB/s'},
            'note': 'NOT the metric (BASELINE config 2 quotes it at 4096 envs): the reference\'s default actor shape, boardlaw/main.py:147'}


def hex_kernels(envs=(4096, 1 << 20), steps=1024, boardsize=11):
    """The reference's step / observe micro-benchmarks (boardlaw/hex/tests.py:186-215: 4096 envs x 1024 calls of `cuda.step` on one
    fixed action set resp. `cuda.observe`, after 1024 random moves) for the two board kernels of the path that really are HBM-bound,
    through the C ABI: bl_hex_step (the reference's in-place kernel), bl_hex_world_step (Hex.step as one launch: board in, board
    out, seats, rewards, terminal) and bl_hex_observe_valid (observe + valid mask).  samples/s as the reference prints them, and
    GB/s of algorithmic bytes against the 8 TB/s peak: step 2 S^2 + 16 B per env (board read + written in place, seat, action,
    rewards), world_step 2 S^2 + 25, observe_valid 10 S^2 + 4 (board + seat read, 8 S^2 of f32 planes + S^2 of mask written).
    4096 envs is the reference's size -- one 0.5 MB board batch, launch-latency-bound; 2^20 envs is where the kernels meet HBM
    (round 5: 64 consecutive envs per workgroup through LDS with 16-byte loads and stores, the flood as a bit-board fill --
    bl_hex_*_tiled, profiles/r05_hex_tiles.txt; the dispatchers pick them: step always, observe from 2^17 envs).
    Timed two ways: the reference's way (a host loop of launches, then a synchronise) and as one captured graph of the same
    launches under HIP events (device time only)."""
    from boardlaw_amd import _native
    from boardlaw_amd.hex import Hex
    L, S = _native.lib(), boardsize
    out = {'boardsize': S, 'calls': steps, 'harness': 'boardlaw/hex/tests.py:186-215', 'peak_GBs': HBM_PEAK_GBS, 'sizes': {}}
    gen = torch.Generator(device='cuda'); gen.manual_seed(7)
    for B in envs:
        worlds = Hex.initial(B, S)
        for _ in range(S * S // 3):                           # mid-game boards; the reference plays 1024 random moves
            valid = worlds.valid
            worlds, _ = worlds.step((torch.rand(valid.shape, device='cuda', generator=gen) * valid).argmax(-1), check=False)
        valid = worlds.valid
        actions = (torch.rand(valid.shape, device='cuda', generator=gen) * valid).argmax(-1).int().contiguous()
        board, seats = worlds.board.contiguous(), worlds.seats.int().contiguous()
        scratch = board.clone()
        board_out, seats_out = torch.empty_like(board), torch.empty_like(seats)
        rewards = torch.empty((B, 2), dtype=torch.float, device='cuda'); terminal = torch.empty((B,), dtype=torch.bool, device='cuda')
        obs = torch.empty((B, S, S, 2), dtype=torch.float, device='cuda'); vmask = torch.empty((B, S * S), dtype=torch.bool, device='cuda')
        st = lambda: _native.stream(board.device)
        calls = {
            'bl_hex_step': (lambda: L.bl_hex_step(scratch.data_ptr(), seats.data_ptr(), actions.data_ptr(), rewards.data_ptr(), B, S, st()), 2 * S * S + 16),
            'bl_hex_world_step': (lambda: L.bl_hex_world_step(board.data_ptr(), seats.data_ptr(), actions.data_ptr(), 0, board_out.data_ptr(), seats_out.data_ptr(),
                                                                rewards.data_ptr(), terminal.data_ptr(), B, S, st()), 2 * S * S + 25),
            'bl_hex_observe_valid': (lambda: L.bl_hex_observe_valid(board.data_ptr(), seats.data_ptr(), obs.data_ptr(), vmask.data_ptr(), B, S, st()), 10 * S * S + 4),
        }
        n = steps if B <= 65536 else max(16, steps // 16)
        res = {}
        for name, (fn, nbytes) in calls.items():
            for _ in range(3):
                _native.check(fn())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            host_s = (time.perf_counter() - t0) / n
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(n):
                        fn()
            torch.cuda.current_stream().wait_stream(side)
            graph.replay(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
            dev_s = 1e-3 * a.elapsed_time(b) / n
            res[name] = {'samples_per_sec_host_loop': B / host_s, 'samples_per_sec_graph': B / dev_s, 'us_per_call_graph': 1e6 * dev_s,
                         'bytes_per_env': nbytes, 'GBs': B * nbytes / dev_s / 1e9, 'frac_of_hbm_peak': B * nbytes / dev_s / 1e9 / HBM_PEAK_GBS}
        out['sizes'][str(B)] = res
    return out


class Learner:
    """Config 4's learner beside the actor (boardlaw/main.py:147-200 at steady state): a buffer of the last `buffer_len` moves,
    one AMP Adam step per move on one random timestep per env, the gradients in one flat bucket that is all-reduced over ranks
    in place (training.optimize + parallel.GradientBucket; RCCL on GPUs, a one-rank group included so the code path is the same)."""

    def __init__(self, net, n_envs, buffer_len, device):
        from boardlaw_amd import parallel
        if not torch.distributed.is_initialized():         # one rank: a one-rank group, so that the collective runs and is timed
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if 'MASTER_PORT' not in os.environ:             # a free port: two bench runs on one host must not meet at a fixed one
                import socket
                with socket.socket() as sock:
                    sock.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
            backend = os.environ.get('BENCH_BACKEND', 'nccl')
            kw = {'device_id': device} if backend == 'nccl' else {}
            torch.distributed.init_process_group(backend, rank=0, world_size=1, **kw)
        self.net, self.buffer, self.buffer_len = net, [], buffer_len
        self.opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        self.scaler = torch.amp.GradScaler('cuda')
        self.bucket = parallel.GradientBucket(net, always=True, timed=True)
        self.idxs = (torch.randint(buffer_len, (n_envs,), device=device), torch.arange(n_envs, device=device))
        self.steps, self.n_envs = 0, n_envs

    def push(self, worlds, decisions, transition):
        from boardlaw_amd import arrdict, learning, training
        self.buffer.append(arrdict.arrdict(worlds=worlds, decisions=decisions.half(), transitions=learning.half(transition)).detach())
        if len(self.buffer) >= self.buffer_len:
            chunk, self.buffer = training.as_chunk(self.buffer, self.n_envs)
            training.optimize(self.net, self.scaler, self.opt, chunk[self.idxs], bucket=self.bucket)
            self.bucket.events = self.bucket.events[-64:]
            self.steps += 1


def arena_pairs(sizes, width, depth, nodes, eager, rank):
    """Two search agents per board size (seeds 10 rank, 10 rank + 1), as config 5 plays them."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent, MoveRng
    pairs = {}
    for S in sizes:
        pair = {}
        for i, name in enumerate(('one', 'two')):
            torch.manual_seed(10 * rank + i)
            w0 = Hex.initial(1, S)
            pair[name] = MCTSAgent(networks.Inference(networks.FCModel(w0.obs_space, w0.action_space, width, depth).cuda(), fused=True),
                                   graph=not eager, n_nodes=nodes, rng=MoveRng())
        pairs[S] = pair
    return pairs


class ArenaPlayer:
    """What one worker process of config 5's match pool holds (arena.MatchPool builds it IN the worker): the agent pairs of the board
    sizes it is given, created on first use; player(S) plays one match of B games on an S x S board and returns its tallies."""

    def __init__(self, B, width, depth, nodes, eager, rank):
        sys.path.insert(0, ROOT)
        self.args, self.pairs = (B, width, depth, nodes, eager, rank), {}

    def __call__(self, S):
        from boardlaw_amd import arena
        from boardlaw_amd.hex import Hex
        B, width, depth, nodes, eager, rank = self.args
        if S not in self.pairs:
            self.pairs.update(arena_pairs([S], width, depth, nodes, eager, rank))
        res = arena.evaluate(Hex.initial(B, S), self.pairs[S])
        torch.cuda.synchronize()
        return {'S': S, 'moves': float(sum(r.moves for r in res)), 'games': float(sum(r.games for r in res)), 'wins': [[float(x) for x in r.wins] for r in res]}


def arena_config(args):
    """--config 5: the arena sweep on this GPU -- for every board size 3..11 one match of 2048 games between two 64-sim 512x4 search
    agents through arena.evaluate (seat-permuted, masked calls of a new batch size every round, argmax actions; captured moves per
    capacity bucket).  A step is one sweep over the nine sizes; sims = env-moves x 64.  Since round 6 the nine matches are played by
    arena.workers_per_gpu(envs) + 1 = 4 persistent worker processes per GPU, board sizes dealt largest first (arena.MatchPool,
    arena.lpt_partition): a match of 2048 games leaves most of the chip idle, three side by side do not (--arena-workers 0: the
    round-5 form, one match at a time in this process).  Prints the one-line JSON."""
    from boardlaw_amd import _native, arena, parallel
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent, MoveRng
    rank, world, local = parallel.env_rank()
    local = int(os.environ.get('BENCH_FORCE_DEVICE', local))
    torch.cuda.set_device(local)
    parallel.init(os.environ.get('BENCH_BACKEND', 'nccl'))
    lib = _native.lib()
    sizes, B, T = list(range(3, 12)), args.envs if args.envs != ENVS else 2048, args.nodes
    # workers_per_gpu (3 at 2048 games) is the optimum for EQUAL matches side by side; a sweep's matches are unequal (11x11 takes 18 x
    # as long as 3x3), and one more worker lets the longest match run alone: 19.0 / 33.4 / 46.8 / 55.6 M sims/s at 0 / 2 / 3 / 4
    # workers (profiles/r06_config5_workers.txt)
    W = arena.workers_per_gpu(B) + 1 if args.arena_workers is None else args.arena_workers
    pool, parts = None, None
    if W > 0:
        parts = [[sizes[i] for i in part] for part in arena.lpt_partition([S ** 2.2 for S in sizes], W)]      # a match's time grows like S^2.2 (profiles/r03_arena_sweep.txt)
        pool = arena.MatchPool(ArenaPlayer, (B, args.width, args.depth, T, args.eager, rank), n_workers=W, device=local)
        pairs = arena_pairs([9], args.width, args.depth, T, args.eager, rank)        # the roofline probe below runs in this process
    else:
        pairs = arena_pairs(sizes, args.width, args.depth, T, args.eager, rank)

    def sweep():
        if pool is not None:
            done = [r for rs in pool.play(parts) for r in rs]
            return sum(r['moves'] for r in done), sum(r['games'] for r in done)
        moves = games = 0
        for S in sizes:
            res = arena.evaluate(Hex.initial(B, S), pairs[S])
            moves += sum(r.moves for r in res); games += sum(r.games for r in res)
        return moves, games
    for _ in range(max(args.warmup, 1)):
        sweep()
    parallel.barrier()
    t0 = time.perf_counter()
    moves = games = 0
    for _ in range(args.steps):
        m_, g_ = sweep(); moves += m_; games += g_
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    per_rank_values, ranks_seen = parallel.gather_over_ranks(moves * T / elapsed)
    elapsed = parallel.max_over_ranks(elapsed)
    moves_all, games_all = parallel.sum_over_ranks(moves), parallel.sum_over_ranks(games)
    value = moves_all * T / elapsed
    if pool is not None:
        pool.close()
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    # roofline: every bl_sim_expand launch of one eager 9x9 match under HIP events; algorithmic bytes per launch = the kernel's
    # per-env figure at (A = 81, d, k measured on the match's opening position at full batch) x the envs of that masked call
    S = 9
    timer = TimedExpand(lib); lib.bl_sim_expand = timer
    live = []
    probe = {name: MCTSAgent(a.network, graph=False, n_nodes=T, rng=MoveRng()) for name, a in pairs[S].items()}

    class Counting:
        def __init__(self, agent): self.agent = agent
        def __call__(self, w, **kw):
            live.extend([w.n_envs] * (T - 1))
            return self.agent(w, **kw)
    timer.on = True
    arena.evaluate(Hex.initial(B, S), {k: Counting(v) for k, v in probe.items()})
    torch.cuda.synchronize(); timer.on = False
    lib.bl_sim_expand = timer.orig
    worlds9 = premix(Hex.initial(B, S), 4, torch.Generator(device='cuda'))
    global BOARD, NODES
    BOARD, NODES = S, T
    d, k, its = tree_statistics(worlds9, pairs[S]['one'].network.model, T)
    us = np.array([a.elapsed_time(b) * 1e3 for a, b in timer.pairs])
    per_env = expand_bytes_per_env(S * S, 2, d, k)
    achieved = per_env * float(np.sum(live[:len(us)])) / float(us.sum() * 1e-6) / 1e9
    out = {'metric': 'mcts_sims_per_sec', 'value': value, 'unit': 'sims/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
           'per_rank_values': per_rank_values, 'ranks_seen': ranks_seen,
           'config': {'workload': f'BASELINE config 5: arena sweep, boards 3..11, {B} games per board size between two {T}-sim FCModel {args.width}x{args.depth} search agents '
                                  '(arena.evaluate: seat-permuted, masked variable-size calls, argmax actions); step = one sweep over the nine sizes; sims = env-moves x sims/move',
                      'boards': sizes, 'envs_per_board': B, 'nodes': T, 'launch': 'eager' if args.eager else 'hip-graph per move and capacity bucket',
                      'matches_in_flight_per_gpu': max(W, 1), 'boards_per_worker': parts,
                      'games_per_sec': games_all / elapsed, 'env_moves_per_step': moves_all / args.steps,
                      'parallelism': f'replicas x{world} (every rank plays its own sweep' + (f' on {W} persistent worker processes on its GPU, board sizes dealt largest first' if W > 0 else '') + ')'},
           'roofline': {'bound': 'hbm', 'kernel': 'bl::sim_expand2_kernel (bl_sim_expand)', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'kernel_us': float(us.mean()), 'launches_timed': int(len(us)),
                        'bytes_per_env_per_launch': per_env, 'mean_envs_per_launch': float(np.mean(live[:len(us)])),
                        'timing': 'HIP events around every bl_sim_expand launch of one eager 9x9 match (the masked calls shrink from 1024 envs as games end); '
                                  'bytes per launch = per-env algorithmic bytes at 9x9 (d, k measured at the opening position) x the envs of that call'}}
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(envs=B)
        if isinstance(out['cpu_baseline'].get('sample'), str):
            out['cpu_baseline']['sample'] = 'config 5 sample: the 9x9 share of the sweep as plain self-play -- ' + out['cpu_baseline']['sample']
    emit(out)
    if world > 1:
        torch.distributed.destroy_process_group()


def respawn_per_gpu(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: re-executes itself as N ranks, one process per GPU
    (the reference's model: boardlaw/main.py:202-209, rebar/parallel.py:28-36), under torch.distributed.run on 127.0.0.1.
    Under a launcher (WORLD_SIZE set) the world size must equal --gpus."""
    world = os.environ.get('WORLD_SIZE')
    if world is not None:
        if int(world) != args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
        return
    if args.gpus > 1:
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)


def dry_run(args):
    """BENCH_DRY=1: the launch/rendezvous/reporting skeleton without any GPU work, so that the N > 1 path (respawn,
    process group, barrier, max-over-ranks, one JSON line from rank 0) is testable on a CPU box with gloo."""
    from boardlaw_amd import parallel
    rank, world, _ = parallel.env_rank()
    parallel.init(os.environ.get('BENCH_BACKEND', 'gloo'))
    parallel.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    parallel.barrier()
    own = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(own, device='cpu')
    per_rank, seen = parallel.gather_over_ranks(float(rank + 1), device='cpu')
    # the per-rank report of the real line (fold_fast, NUMA pinning) through the same collective; the dry run pins to node 0's cores
    # when the box has that list (it changes nothing on a one-node container) and pretends rank 1 failed bl_selftest()
    pinning = parallel.pin_to_numa_node(0 if os.path.exists('/sys/devices/system/node/node0/cpulist') else None)
    fold = [int(x) for x in parallel.gather_over_ranks(float(rank != 1), device='cpu')[0]]
    numa = [int(x) for x in parallel.gather_over_ranks(float(-1 if pinning['numa_node'] is None else pinning['numa_node']), device='cpu')[0]]
    cpus = [int(x) for x in parallel.gather_over_ranks(float(pinning['cpus']), device='cpu')[0]]
    if rank == 0:
        print(json.dumps({'metric': 'mcts_sims_per_sec', 'value': 0.0, 'unit': 'sims/s', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed, 'dry_run': True, 'per_rank_values': per_rank,
                          'ranks_seen': seen, 'baseline_config': 3 if (args.config == 2 and world > 1) else args.config,
                          'ranks': {'per_rank_values': per_rank, 'ranks_seen': seen, 'min': min(per_rank), 'max': max(per_rank), 'mean': float(np.mean(per_rank)),
                                    'fold_fast_per_rank': fold, 'numa_node_per_rank': numa, 'cpus_pinned_per_rank': cpus, 'pinned': pinning['pinned']}}))
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    global BOARD, NODES, WIDTH, DEPTH
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=2, choices=[1, 2, 3, 4, 5], help='BASELINE.json configuration (default 2: the metric; 3 = 2 on every GPU)')
    ap.add_argument('--envs', type=int, default=ENVS, help='envs per GPU (the metric is quoted at 4096)')
    ap.add_argument('--boardsize', type=int, default=BOARD, help='exploration only: the metric is quoted on 9x9')
    ap.add_argument('--nodes', type=int, default=NODES, help='exploration only: sims per move (metric: 64)')
    ap.add_argument('--width', type=int, default=WIDTH, help='exploration only: FCModel width (metric: 512)')
    ap.add_argument('--depth', type=int, default=DEPTH, help='exploration only: FCModel depth (metric: 4)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-traffic', action='store_true', help='skip the two rocprofv3 --pmc passes behind roofline.traffic')
    ap.add_argument('--plain-network', action='store_true', help='run the nn.Module under autocast instead of the fp16 inference plan')
    ap.add_argument('--torch-gemms', action='store_true', help='keep the Linears as torch (hipBLASLt) GEMMs instead of the fused MFMA kernel')
    ap.add_argument('--timed-only', action='store_true', help='for profilers: only capture, warm-up and the timed moves; prints a reduced line')
    ap.add_argument('--no-reference-rng', action='store_true', help='skip the second timed region (torch rand_like per simulation)')
    ap.add_argument('--no-two-actors', action='store_true', help='skip the two-actors-per-GPU region')
    ap.add_argument('--no-fold-safe', action='store_true', help='skip the timed region with the ISA-padded fold (value_fold_safe)')
    ap.add_argument('--no-soak', action='store_true', help='skip the 200 extra moves behind value_after_self_play_drift')
    ap.add_argument('--no-variants', action='store_true', help='skip the torch-GEMM and fp32-leaves timed regions (value_torch_gemms, value_fp32_leaves)')
    ap.add_argument('--no-hex-kernels', action='store_true', help='skip the step / observe micro-benchmark (hex_kernels)')
    ap.add_argument('--no-32k', action='store_true', help='skip the region at 32768 envs per GPU (value_32k_envs)')
    ap.add_argument('--no-learner', action='store_true', help='--config 4 without the learner step (self-play only)')
    ap.add_argument('--learner', action='store_true', help='any shape with the learner step beside the actor, like --config 4 (e.g. --envs 32768 --learner: boardlaw/main.py:147-200 at its own defaults)')
    ap.add_argument('--buffer', type=int, default=64, help='--config 4: moves in the learner\'s buffer (boardlaw/main.py:150 keeps 64)')
    ap.add_argument('--eager', action='store_true', help='launch kernel by kernel instead of replaying a HIP graph per move')
    ap.add_argument('--arena-workers', type=int, default=None, help='--config 5: worker processes per GPU (default: arena.workers_per_gpu(envs) + 1 = 4 for 2048 games per match; 0 = one match at a time in this process)')
    args = ap.parse_args()
    respawn_per_gpu(args)
    shapes = {1: (5, 64, 16, 16, 4), 4: (13, 1024, 256, 1024, 8)}      # (board, envs per GPU, sims/move, width, depth) -- BASELINE.json configs
    if args.config in shapes:
        flags = ('boardsize', 'envs', 'nodes', 'width', 'depth')
        for flag, preset in zip(flags, shapes[args.config]):
            if getattr(args, flag) == ap.get_default(flag):
                setattr(args, flag, preset)
    BOARD, NODES, WIDTH, DEPTH = args.boardsize, args.nodes, args.width, args.depth
    default_shape = (BOARD, NODES, WIDTH, DEPTH) == (9, 64, 512, 4) and args.config in (2, 3)
    if os.environ.get('BENCH_DRY') == '1':
        return dry_run(args)
    if args.config == 5:
        assert torch.cuda.is_available(), 'bench.py needs an MI355X; there is no CPU path'
        if args.steps == ap.get_default('steps'):
            args.steps, args.warmup = 3, 1                # a sweep is nine matches of 2048 games: seconds, not milliseconds
        return arena_config(args)

    assert torch.cuda.is_available(), 'bench.py needs an MI355X; there is no CPU path'
    from boardlaw_amd import parallel
    rank, world, local = parallel.env_rank()
    assert world == args.gpus, (world, args.gpus)
    # BENCH_FORCE_DEVICE / BENCH_BACKEND exist only to smoke-test the N>1 code path on a 1-GPU box (two ranks sharing
    # device 0 over gloo); the driver's multi-GPU runs use one device per rank and RCCL.
    local = int(os.environ.get('BENCH_FORCE_DEVICE', local))
    torch.cuda.set_device(local)
    # one host thread per GPU issues every launch and replay: keep it on the cores of its GPU's NUMA node (stated in the line)
    affinity_before = os.sched_getaffinity(0)
    pinning = parallel.pin_to_numa_node(parallel.gpu_numa_node(local))
    parallel.init(os.environ.get('BENCH_BACKEND', 'nccl'))   # used only for the barrier and the max-over-ranks of the elapsed time

    from boardlaw_amd import _native, networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent, MoveRng
    lib = _native.lib()

    gen = torch.Generator(device='cuda'); gen.manual_seed(1000 + rank)
    torch.manual_seed(0)
    worlds = Hex.initial(args.envs, BOARD)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=WIDTH, depth=DEPTH).cuda()
    worlds = premix(worlds, BOARD * BOARD // 3, gen)
    torch.manual_seed(1 + rank)
    agent = MCTSAgent(net if args.plain_network else networks.Inference(net, fused=not args.torch_gemms), n_nodes=NODES, graph=not args.eager, rng=MoveRng())

    timer = TimedExpand(lib)
    lib.bl_sim_expand = timer

    learner = None
    if (args.config == 4 or args.learner) and not args.no_learner:
        learner = Learner(net, args.envs, args.buffer, torch.device('cuda', local))
        args.warmup = max(args.warmup, args.buffer)          # the timed steps are steady state: buffer full, one learner step per move

    def move(w):
        # one actor step of the self-play loop (boardlaw/main.py:176-177): search + env step
        if learner is None:
            return agent.play(w)[1]
        d, new_w, transition = agent.play(w)
        learner.push(w, d, transition)                        # main.py:183-197: buffer, as_chunk, optimize (gradient all-reduce inside)
        return new_w

    launch = 'eager' if args.eager else 'hip-graph per move'
    if not args.eager:
        try:                                   # capture happens on the first call
            worlds = move(worlds)
        except Exception as e:                 # pragma: no cover - keep the bench alive if capture is refused
            print(f'[bench] graph capture failed ({type(e).__name__}: {e}); launching eagerly', file=sys.stderr)
            torch.cuda.synchronize()
            agent.graph, args.eager, launch = False, True, f'eager (graph capture failed: {type(e).__name__})'
    for _ in range(args.warmup):
        worlds = move(worlds)

    barrier = parallel.barrier
    barrier()
    timer.on = args.eager          # graph replays cannot carry per-launch events; see the probe below
    t0 = time.perf_counter()
    for _ in range(args.steps):
        worlds = move(worlds)
    barrier()
    elapsed = time.perf_counter() - t0
    timer.on = False
    # every rank's own rate and a count of the ranks, through the collective itself (RCCL on GPUs): the line proves N ranks ran
    per_rank_values, ranks_seen = parallel.gather_over_ranks(args.envs * NODES * args.steps / elapsed)
    elapsed = parallel.max_over_ranks(elapsed)
    # what could make ONE rank slow, per rank and through the collective too: a device that failed bl_selftest() runs the ISA-padded
    # fold (-10 %) and would set the max-over-ranks time silently; so would a rank left on the far socket
    fold_fast_per_rank = [int(x) for x in parallel.gather_over_ranks(float(_native.fold_fast(torch.device('cuda', local))))[0]]
    numa_per_rank = [int(x) for x in parallel.gather_over_ranks(float(-1 if pinning['numa_node'] is None else pinning['numa_node']))[0]]
    cpus_per_rank = [int(x) for x in parallel.gather_over_ranks(float(pinning['cpus']))[0]]
    rank_report = {'per_rank_values': per_rank_values, 'ranks_seen': ranks_seen, 'min': min(per_rank_values), 'max': max(per_rank_values),
                   'mean': float(np.mean(per_rank_values)), 'fold_fast_per_rank': fold_fast_per_rank,
                   'numa_node_per_rank': numa_per_rank, 'cpus_pinned_per_rank': cpus_per_rank, 'pinned': pinning['pinned']}

    sims_total = world * args.envs * NODES * args.steps
    value = sims_total / elapsed

    if args.timed_only:
        if rank == 0:
            emit({'metric': 'mcts_sims_per_sec', 'value': value, 'unit': 'sims/s', 'n_gpus': world, 'steps': args.steps,
                  'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'timed_only': True,
                  'per_rank_values': per_rank_values, 'ranks_seen': ranks_seen, 'ranks': rank_report})
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    value_torch_rng = None
    if world == 1 and not args.eager and not args.no_reference_rng and default_shape:
        # the same moves with the reference's RNG protocol issued call by call: one rand_like (B,T) f16 per simulation
        # (cuda.cu:191) instead of MoveRng's one launch per move -- the same numbers from the same seed, 62 more launches per move
        from boardlaw_amd.mcts import TorchRng
        ref_agent = MCTSAgent(agent.network, n_nodes=NODES, graph=True, rng=TorchRng())
        w2 = worlds
        for _ in range(1 + min(args.warmup, 2)):
            w2 = ref_agent.play(w2)[1]
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            w2 = ref_agent.play(w2)[1]
        barrier()
        value_torch_rng = args.envs * NODES * args.steps / (time.perf_counter() - t1)
        del ref_agent, w2

    value_torch_gemms = value_fp32_leaves = None
    if world == 1 and not args.eager and not args.no_variants and default_shape and not args.plain_network:
        # the network north_star names -- the Linears as PyTorch-ROCm (hipBLASLt) GEMMs, bit-identical to the module under fp16
        # autocast (tests/test_gpu_parity.py::test_inference_plan_matches_autocast) -- instead of the hand-written MFMA kernel
        value_torch_gemms = value if args.torch_gemms else rate_of(
            MCTSAgent(networks.Inference(net, fused=False), n_nodes=NODES, graph=True, rng=MoveRng()), worlds, args.steps, args.warmup, barrier)
        # the exact mode: leaves evaluated in fp32 like the reference's recorded (CPU) runs, f16 stores only -- a seeded run then IS
        # the reference's run in most envs (tests/test_fp32_leaves.py, profiles/r05_fp32_leaves.txt)
        value_fp32_leaves = rate_of(
            MCTSAgent(networks.Inference(net, fused=True, precision='fp32'), n_nodes=NODES, graph=True, rng=MoveRng()), worlds, args.steps, args.warmup, barrier)

    value_fold_safe = None
    if world == 1 and not args.eager and not args.no_fold_safe and default_shape:
        # the same moves with the ISA-padded fold (two wait states between a VALU write and the DPP read of it; the headline runs
        # the one-wait-state fold only because bl_selftest() verified it on THIS device -- a device that fails the self-test gets
        # this rate).  BL_FOLD_SAFE is read by the host layer when a search is built, i.e. at this agent's capture.
        os.environ['BL_FOLD_SAFE'] = '1'
        try:
            safe_agent = MCTSAgent(agent.network, n_nodes=NODES, graph=True, rng=MoveRng())
            w5 = worlds
            for _ in range(1 + min(args.warmup, 2)):
                w5 = safe_agent.play(w5)[1]
            barrier()
            t4 = time.perf_counter()
            for _ in range(args.steps):
                w5 = safe_agent.play(w5)[1]
            barrier()
            value_fold_safe = args.envs * NODES * args.steps / (time.perf_counter() - t4)
            del safe_agent, w5
        finally:
            del os.environ['BL_FOLD_SAFE']

    value_powf_libm = None
    if world == 1 and not args.eager and not args.no_fold_safe and default_shape and 'BL_POWF_LIBM' not in os.environ:
        # the second parity target -- the reference as its own JIT build computes (no -O flag: glibc's powf(bot, 2) under the Newton
        # derivative term instead of bot*bot; bl_tune_t.powf_libm, csrc/bl_powf.h): what the exact mode costs
        os.environ['BL_POWF_LIBM'] = '1'
        try:
            powf_agent = MCTSAgent(agent.network, n_nodes=NODES, graph=True, rng=MoveRng())
            w6 = worlds
            for _ in range(1 + min(args.warmup, 2)):
                w6 = powf_agent.play(w6)[1]
            barrier()
            t5 = time.perf_counter()
            for _ in range(args.steps):
                w6 = powf_agent.play(w6)[1]
            barrier()
            value_powf_libm = args.envs * NODES * args.steps / (time.perf_counter() - t5)
            del powf_agent, w6
        finally:
            del os.environ['BL_POWF_LIBM']

    two_actors = None
    if world == 1 and not args.eager and not args.no_two_actors and default_shape:
        # NOT the metric (its configuration is ONE 4096-env search per GPU): a second, independent config-2 search resident on
        # the same GPU, each actor replaying its captured moves on its own stream.  Both search kernels are latency-bound, so
        # a second actor fills cycles the first leaves idle -- what a deployment that wants sims/s per GPU would run.
        gen2 = torch.Generator(device='cuda'); gen2.manual_seed(2000 + rank)
        pair = [Hex(board=worlds.board.clone(), seats=worlds.seats.clone()), premix(Hex.initial(args.envs, BOARD), BOARD * BOARD // 3, gen2)]
        gens = [torch.Generator(device='cuda'), torch.Generator(device='cuda')]
        gens[0].manual_seed(3000 + rank); gens[1].manual_seed(4000 + rank)
        # a generator per actor: concurrently replayed graphs on one generator race for its Philox offset (MoveRng.__init__)
        actors = [MCTSAgent(agent.network, n_nodes=NODES, graph=True, rng=MoveRng(generator=g)) for g in gens]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream())

        def round_of_moves():
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    pair[i] = actors[i].play(pair[i])[1]
        for _ in range(3):
            round_of_moves()
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            round_of_moves()
        torch.cuda.synchronize()
        two_actors = {'sims_per_sec': 2 * args.envs * NODES * args.steps / (time.perf_counter() - t2), 'actors': 2, 'envs_per_actor': args.envs,
                      'note': 'two independent config-2 searches on one GPU, one stream each (not the metric: 8192 envs resident)'}
        for s_ in streams:
            torch.cuda.current_stream().wait_stream(s_)
        del actors, pair

    value_32k = None
    if world == 1 and not args.eager and default_shape and args.envs == ENVS and not args.no_32k and not args.plain_network and not args.torch_gemms:
        value_32k = operating_point(32768, net, lib, max(args.steps // 2, 5), args.warmup, barrier, rank)

    drifted = None
    if world == 1 and not args.eager and default_shape and not args.no_soak:
        # the timed region above starts from freshly pre-mixed boards; self-play drifts to its own mix of positions (shorter games
        # restart, trees get deeper): the same measurement after 150 further moves, reported beside the headline
        w4 = worlds
        for _ in range(150):
            w4 = move(w4)
        barrier()
        t3 = time.perf_counter()
        for _ in range(50):
            w4 = move(w4)
        barrier()
        drifted = {'sims_per_sec': args.envs * NODES * 50 / (time.perf_counter() - t3), 'moves_before': args.warmup + args.steps + 150, 'moves_timed': 50}
        del w4

    if rank == 0:
        A, S = BOARD * BOARD, 2
        if not args.eager:
            # the same moves launched kernel by kernel, only to bracket every bl_sim_expand launch with HIP events
            probe = MCTSAgent(agent.network, n_nodes=NODES, graph=False, rng=MoveRng())
            timer.on = True
            for _ in range(min(args.steps, 5)):
                worlds = probe.play(worlds)[1]
            torch.cuda.synchronize()
            timer.on = False
        lib.bl_sim_expand = timer.orig
        # SURVEY 8d "kernel-only sims/s (search kernels without the network)": the same moves with the finish step as its
        # own launch (bl_mlp_forward_f16 + bl_sim_finish instead of bl_sim_infer_finish), both search kernels under HIP events
        search_only = None
        if not args.plain_network and not args.torch_gemms and default_shape:
            t_exp, t_fin = TimedExpand(lib), TimedExpand(lib, 'bl_sim_finish')
            lib.bl_sim_expand, lib.bl_sim_finish = t_exp, t_fin
            split = MCTSAgent(agent.network, n_nodes=NODES, graph=False, rng=MoveRng(), fuse_finish=False)
            w3 = split.play(worlds)[1]
            t_exp.on = t_fin.on = True
            for _ in range(2):
                w3 = split.play(w3)[1]
            torch.cuda.synchronize()
            lib.bl_sim_expand, lib.bl_sim_finish = t_exp.orig, t_fin.orig
            if t_fin.pairs:
                search_only = {'sims_per_sec': args.envs / ((t_exp.mean_us() + t_fin.mean_us()) * 1e-6), 'bl_sim_expand_us': t_exp.mean_us(),
                               'bl_sim_finish_us': t_fin.mean_us(),
                               'note': 'envs / (bl_sim_expand + bl_sim_finish) per simulation, HIP events, network launched separately and not counted'}
            del split, w3
        d, k, its = tree_statistics(worlds, net, NODES)
        kernel_us = timer.mean_us()
        per_launch = expand_bytes_per_env(A, S, d, k) * args.envs
        achieved = per_launch / (kernel_us * 1e-6) / 1e9
        traffic, traffic_source = None, None
        if args.envs == ENVS and default_shape and not args.no_traffic and world == 1:      # a per-GPU figure: the N = 1 line carries it
            traffic, traffic_source = measure_traffic()
            tpath = os.path.join(ROOT, 'profiles', 'r02_traffic.json')
            if traffic is None and os.path.exists(tpath):
                # fallback, NOT measured in this run: the same two passes as committed with round 2
                traffic = json.load(open(tpath))['traffic_bytes_per_launch']
                traffic_source = f'static: profiles/r02_traffic.json ({traffic_source})'
        out = {
            'metric': 'mcts_sims_per_sec', 'value': value, 'unit': 'sims/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'per_rank_values': per_rank_values, 'ranks_seen': ranks_seen, 'ranks': rank_report,
            'config': {'workload': f'{BOARD}x{BOARD} Hex, {args.envs} envs/GPU x {NODES} sims/move, FCModel {WIDTH}x{DEPTH} fp16 autocast'
                                   + (' (BASELINE config 2)' if default_shape and args.envs == ENVS and world == 1 else
                                      f' (BASELINE config 3: config 2 on each of {world} GPUs, self-play only)' if default_shape and args.envs == ENVS else
                                      f' (BASELINE config {args.config}' + (', per-GPU shape' if args.config == 4 else '') + ')'
                                      if args.config in shapes and (BOARD, args.envs, NODES, WIDTH, DEPTH) == shapes[args.config] else ' (NOT one of BASELINE.json\'s configurations)')
                                   + ('; step = one self-play move of the batch' if learner is None else
                                      f'; step = one self-play move of the shard + one learner step (buffer of {args.buffer} moves, AMP Adam, one flat-bucket '
                                      'gradient all-reduce over the ranks: boardlaw/main.py:147-200 at steady state)'),
                       'baseline_config': 3 if (args.config == 2 and world > 1) else args.config,
                       'learner': None if learner is None else {
                           'steps_taken': learner.steps, 'buffer_moves': args.buffer, 'bucket_mb': round(learner.bucket.flat.numel() * 4 / 2**20, 2),
                           'allreduce_ms_mean': float(np.mean(learner.bucket.collective_ms()[-args.steps:])) if learner.bucket.events else None,
                           'backend': torch.distributed.get_backend() if torch.distributed.is_initialized() else None},
                       'envs_per_gpu': args.envs, 'nodes': NODES, 'boardsize': BOARD, 'parallelism': f'replicas x{world}', 'launch': launch,
                       'network': ('nn.Module under fp16 autocast' if args.plain_network else 'fp16 inference plan, torch GEMMs (bit-identical to autocast)'
                                   if args.torch_gemms
                                   else 'a launch per Linear, bl_mlp_layers_f16 (autocast rounding points), + bl_sim_finish' if not agent.network.prefers_fused(args.envs)
                                   else 'fused MFMA kernel bl_sim_infer_finish (autocast rounding points, another GEMM summation order: every pre-head output within '
                                        '3 f16 ulp of the activation scale + 1 % of autocast\'s, >= 99 % within 1 ulp -- tests/test_gpu_parity.py::test_fused_mlp_matches_autocast)')
                                   + '; root evaluation fp32',
                       'rng': 'MoveRng: stream-identical to the reference protocol (torch generator; Dirichlet and Categorical are torch\'s own calls; the T-1 rand_like (B,T) f16 draws of a move come from ONE launch, bl_rand_block, that evaluates the Philox counters those calls would use and advances the generator by what they would consume -- tests/test_rng_stream.py)',
                       'rng_stream_identical_to_reference_protocol': True,
                       'seeded_run_vs_reference': 'the RANDOM DRAWS are the reference protocol\'s bit for bit and the tree arithmetic is the reference CPU path\'s bit for bit GIVEN '
                                                  'the leaf evaluations; the leaf evaluations are f16 MFMA GEMMs here and f32 on the reference\'s CPU, so a seeded run is not the '
                                                  'reference\'s run: on tests/golden/search_9x9_w512.npz >= 95 % of the envs pick the reference\'s first action and >= 60 % have '
                                                  'identical root visit counts (tests/test_reference_fixtures.py)',
                       'parity_target': 'reference cpu.cpp as g++ -O1 and up compiles it (powf(bot, 2) folded to bot*bot); the reference\'s own JIT build passes no -O flag and calls '
                                        'libm powf, which differs from bot*bot on 0.036 % of floats: that target is bl_tune_t.powf_libm (BL_POWF_LIBM=1; glibc 2.35\'s powf restated in '
                                        'csrc/bl_powf.h, equal to the host libm on all 2^32 floats; checker oracle/liboracle_powf.so) -- value_powf_libm is its rate',
                       'value_torch_gemms': value_torch_gemms,   # the Linears as PyTorch-ROCm (hipBLASLt) GEMMs: bit-identical to the nn.Module under fp16 autocast -- the network north_star names
                       'value_fp32_leaves': value_fp32_leaves,   # the exact mode (networks.Inference(precision='fp32')): leaves in fp32 like the reference's recorded runs; tests/test_fp32_leaves.py
                       'value_powf_libm': value_powf_libm,   # the same moves in the reference's-own-build mode (bl_tune_t.powf_libm; ISA-padded fold)
                       'fold_fast': bool(_native.fold_fast(torch.device('cuda', local))),
                       'value_fold_safe': value_fold_safe,   # the ISA-padded fold (two wait states per dependent DPP step): what a device that fails bl_selftest() runs
                       'value_reference_rng_protocol': value,          # the headline IS on the reference's stream (MoveRng above)
                       'value_rand_like_call_by_call': value_torch_rng,  # TorchRng: the same stream drawn with T-1 separate launches
                       'value_after_self_play_drift': drifted,
                       'value_32k_envs': value_32k,        # 32768 envs per GPU (boardlaw/main.py:147), with its own roofline object
                       'search_kernels_only': search_only,
                       'two_actors_per_gpu': two_actors,
                       'network_mfma_bound_sims_per_sec': 2.5e15 / (2 * (2 * A * WIDTH + DEPTH * WIDTH * WIDTH + WIDTH * (A + 1))),
                       'd_policy_evals_per_descent': round(d, 3), 'k_child_lookups_per_descent': round(k, 3),
                       'newton_iters_per_eval': round(its, 3),
                       # the reference's own unit for its descent benchmark (boardlaw/mcts/tests.py:163-182: ns/descent of cuda.descend): one
                       # descent per env per bl_sim_expand launch -- the launch also expands, steps and observes the leaf, so an upper bound;
                       # beside cpu_baseline.ns_per_descent_*
                       'gpu_ns_per_descent': 1e3 * kernel_us / args.envs,
                       'bytes_per_sim_whole_path': round(total_bytes_per_sim(A, S, NODES, d, k), 1),
                       'hbm_frac_whole_path': total_bytes_per_sim(A, S, NODES, d, k) * value / world / (HBM_PEAK_GBS * 1e9)},
            'roofline': {'bound': 'hbm', 'kernel': 'bl::sim_expand2_kernel (bl_sim_expand)', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel_us': kernel_us, 'bytes_per_launch': per_launch, 'launches_timed': len(timer.pairs),
                         'timing': 'HIP events around every launch ' + ('inside the timed region' if args.eager else 'in an eager re-run of the same moves right after the timed (graph-replay) region')},
        }
        if world == 1 and not args.no_hex_kernels and default_shape:
            out['hex_kernels'] = hex_kernels()
        if world == 1 and not args.no_cpu_baseline and (default_shape or args.config in shapes):
            os.sched_setaffinity(0, affinity_before)          # the host baseline may use every core this container has
            # a bounded sample of THIS configuration's search on the host cores (config 4: 64 envs per process -- one 13x13 / 256-sim /
            # 1024x8 move of 64 envs is ~0.3 TFLOP of f32 numpy per process)
            out['cpu_baseline'] = cpu_baseline(envs=args.envs, single_envs=256 if default_shape else min(64, args.envs))
        emit(out)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
