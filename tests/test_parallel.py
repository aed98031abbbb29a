"""The N>1 path on CPU: two gloo ranks on 127.0.0.1 exercise sharding, the timing/throughput aggregation bench.py
uses, and the q-range all-reduce that makes sharded self-play reproduce the unsharded batch-global normalisation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _enc(f):
    b = np.float32(f).view(np.uint32)
    return np.uint32(~b) if b & 0x80000000 else np.uint32(b | 0x80000000)


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from boardlaw_amd import parallel, _native
    r, w = parallel.init('gloo')
    assert (r, w) == (rank, world)
    # sharding: contiguous, disjoint, covering
    sl = parallel.shard(4099, rank, world)
    # each rank's q range over its own envs, in the library's state layout (slot stride 64 words)
    rng = np.random.default_rng(7)
    q = rng.normal(size=(4099, 64, 2)).astype(np.float32) * 3
    mine = q[sl]
    # words in memory = the unsigned codes XOR 0x80000000 (order-preserving as signed int32); untouched words = the MAX's identity
    st = np.full(_native.QRANGE_WORDS, 0x80000000, np.uint32)
    st[64 * (rank * 3 % 64)] = ~_enc(mine.min()) ^ np.uint32(0x80000000); st[64 * (rank * 3 % 64) + 1] = _enc(mine.max()) ^ np.uint32(0x80000000)
    state = torch.from_numpy(st.view(np.int32).copy())
    parallel.allreduce_qrange(state)
    lo, hi = _native.qrange_decode(state).tolist()
    # timing aggregation as in bench.py: max over ranks, total work over that time
    elapsed = parallel.max_over_ranks(1.0 + rank)
    sims = parallel.sum_over_ranks(float((sl.stop - sl.start) * 64))
    parallel.barrier()
    out.put((rank, sl.start, sl.stop, lo, hi, float(q.min()), float(q.max()), elapsed, sims))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == 4099          # shards tile the env axis
    for r in res:
        assert r[3] == r[5] and r[4] == r[6]                                          # global q range on every rank
        assert r[7] == 2.0 and r[8] == 4099 * 64                                      # max time, total work


def test_shard_properties():
    from boardlaw_amd.parallel import shard
    for n in (1, 7, 4096, 32768, 4099):
        for world in (1, 2, 3, 8):
            sl = [shard(n, r, world) for r in range(world)]
            assert sl[0].start == 0 and sl[-1].stop == n
            assert all(a.stop == b.start for a, b in zip(sl, sl[1:]))
            sizes = [s.stop - s.start for s in sl]
            assert max(sizes) - min(sizes) <= 1


def _bench(args, env):
    import json, subprocess, sys
    e = {**os.environ, **env}
    e.pop('WORLD_SIZE', None); e.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` without a launcher re-executes itself as 2 ranks (dry run: rendezvous, barrier,
    max-over-ranks and the single JSON line, no GPU work); a launcher whose world size disagrees with --gpus is refused."""
    import subprocess, sys
    d = _bench(['--gpus', '2', '--steps', '1', '--warmup', '0'], {'BENCH_DRY': '1', 'BENCH_BACKEND': 'gloo'})
    assert d['n_gpus'] == 2 and d['ms_per_step'] >= 20.      # rank 1 sleeps 20 ms: the MAX over ranks is reported
    assert d['ranks_seen'] == 2 and d['per_rank_values'] == [1.0, 2.0]     # through the collective: both ranks took part
    assert _bench([], {'BENCH_DRY': '1'})['n_gpus'] == 1
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env={**os.environ, 'WORLD_SIZE': '3', 'BENCH_DRY': '1'},
                         capture_output=True, text=True)
    assert bad.returncode != 0 and 'WORLD_SIZE=3' in bad.stderr


def test_bench_eight_ranks_dry_run():
    """The shape of the driver's 8-GPU run -- `bench.py --gpus 8` re-executing itself as eight ranks on 127.0.0.1, rendezvous,
    barrier, max-over-ranks, every rank's figures through the collective, ONE JSON line -- on CPU ranks over gloo; under a launcher,
    too (the driver's own command line: torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8)."""
    import json, subprocess, sys
    d = _bench(['--gpus', '8', '--steps', '1', '--warmup', '0'], {'BENCH_DRY': '1', 'BENCH_BACKEND': 'gloo'})
    assert d['n_gpus'] == 8 and d['ranks_seen'] == 8 and d['per_rank_values'] == [float(r + 1) for r in range(8)]
    assert d['ms_per_step'] >= 80.                                   # rank 7 sleeps 80 ms: the MAX over ranks
    r = d['ranks']
    assert r['min'] == 1.0 and r['max'] == 8.0 and r['mean'] == 4.5 and r['fold_fast_per_rank'] == [1, 0, 1, 1, 1, 1, 1, 1]
    assert len(r['numa_node_per_rank']) == 8 and len(r['cpus_pinned_per_rank']) == 8 and all(c >= 1 for c in r['cpus_pinned_per_rank'])
    assert d['baseline_config'] == 3
    e = {**os.environ, 'BENCH_DRY': '1', 'BENCH_BACKEND': 'gloo'}
    e.pop('WORLD_SIZE', None); e.pop('RANK', None)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
                          '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1'],
                         env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['ranks_seen'] == 8


def test_numa_pinning_helpers(tmp_path):
    """parallel.pin_to_numa_node / gpu_numa_node against a fake sysfs: the cores of the GPU's node that this process may already
    use, and nothing at all when the node is unknown."""
    from boardlaw_amd import parallel
    before = os.sched_getaffinity(0)
    try:
        keep = sorted(before)[:max(1, len(before) // 2)]
        node = tmp_path / 'devices/system/node/node3'
        node.mkdir(parents=True)
        (node / 'cpulist').write_text(','.join(str(c) for c in keep) + ',100000-100003\n')      # cores outside the affinity mask are ignored
        assert parallel._cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
        assert parallel.pin_to_numa_node(None, sysfs=str(tmp_path)) == {'numa_node': None, 'cpus': len(before), 'pinned': False}
        assert parallel.pin_to_numa_node(5, sysfs=str(tmp_path))['pinned'] is False                 # no such node: left alone
        rep = parallel.pin_to_numa_node(3, sysfs=str(tmp_path))
        assert os.sched_getaffinity(0) == set(keep) and rep['cpus'] == len(keep) and rep['pinned'] == (set(keep) != before)
        assert parallel.gpu_numa_node(0, sysfs=str(tmp_path)) is None                               # no GPU / no PCI entry: unknown
    finally:
        os.sched_setaffinity(0, before)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """The real bench with --gpus 2: two ranks sharing device 0 over gloo (the driver's runs use one device per rank and RCCL)."""
    d = _bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--envs', '256', '--no-cpu-baseline'],
               {'BENCH_BACKEND': 'gloo', 'BENCH_FORCE_DEVICE': '0'})
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'replicas x2' and d['value'] > 0
    assert d['ranks_seen'] == 2 and len(d['per_rank_values']) == 2 and all(v > 0 for v in d['per_rank_values'])
    assert d['scaling'] == 'weak'


@pytest.mark.gpu
def test_rccl_collectives_on_the_device():
    """The collectives the sharded path uses -- barrier, MAX over ranks (float64), the q-range all-reduce (int64 MAX) -- through
    RCCL itself on the GPU, in a one-rank group (the pool's boxes have one GPU; N > 1 ranks are covered over gloo): the
    communicator comes up in this environment and the dtypes/ops exist in RCCL."""
    import subprocess, sys
    code = '''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from boardlaw_amd import parallel
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
parallel.barrier()
assert parallel.max_over_ranks(1.25) == 1.25 and parallel.sum_over_ranks(2.5) == 2.5
state = torch.tensor([[-1, 5, -2**31, 2**31 - 1]], dtype=torch.int32, device='cuda')
want = state.clone()
assert torch.equal(parallel.allreduce_qrange(state), want)
flat = torch.arange(8, dtype=torch.float32, device='cuda')
dist.all_reduce(flat); torch.cuda.synchronize()
assert dist.get_backend() == 'nccl'
dist.destroy_process_group()
print('rccl ok')
''' % ROOT
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'rccl ok' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.gpu
def test_qrange_sync_through_rccl_inside_a_search():
    """MCTS(qrange_sync=parallel.allreduce_qrange) -- the opt-in that makes N shards normalise q over ALL envs like one device
    (boardlaw/mcts/cpp/cuda.cu:101-105) -- with the collective going through RCCL itself, once per simulation, eagerly AND inside a
    captured move (a one-rank group: the pool's boxes have one GPU): every output equals the unsynchronised search's bit for bit
    (MAX over one rank is the identity), the generator ends at the same offset, and the cost per simulation is printed."""
    import subprocess, sys
    code = '''
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, %r)
from boardlaw_amd import networks, parallel
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTSAgent, MoveRng
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29543')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
torch.manual_seed(0)
worlds = Hex.initial(1024, 9)
net = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, 256, 2).cuda(), fused=True)
calls = [0]
def sync(state):
    calls[0] += 1
    return parallel.allreduce_qrange(state)
outs, times = {}, {}
for name, kw in (('plain', {}), ('synced', {'qrange_sync': sync})):
    for graph in (False, True):
        agent = MCTSAgent(net, n_nodes=32, graph=graph, rng=MoveRng(), **kw)
        torch.manual_seed(5)
        w = worlds
        if graph:
            agent.play(w); torch.manual_seed(5)          # capture consumes the generator: reseed
        res = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            d, w, t = agent.play(w)
            res.append(d)
        torch.cuda.synchronize(); times[name, graph] = (time.perf_counter() - t0) / 4
        outs[name, graph] = (res, torch.cuda.default_generators[0].get_offset())
assert calls[0] >= 31 * 4
for graph in (False, True):
    (a, oa), (b, ob) = outs['plain', graph], outs['synced', graph]
    assert oa == ob
    for da, db in zip(a, b):
        for k in ('logits', 'prior', 'v', 'actions', 'n_leaves', 'n_sims'):
            x, y = da[k], db[k]
            if x.dtype == torch.half: x, y = x.view(torch.int16), y.view(torch.int16)
            assert torch.equal(x, y), (graph, k)
print('qrange sync ok: us per simulation added by the collective: eager %%.1f, captured %%.1f' %% (
      1e6 * (times['synced', False] - times['plain', False]) / 31, 1e6 * (times['synced', True] - times['plain', True]) / 31))
dist.destroy_process_group()
''' % ROOT
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'qrange sync ok' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    print([l for l in r.stdout.splitlines() if 'qrange sync ok' in l][-1])


def _sharded_search_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    for p in (ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib
    from gpu_util import HashNetwork, ReplayRng, bits16, to_np
    from test_gpu_parity import oracle_search, premixed
    from boardlaw_amd import parallel
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS
    torch.cuda.set_device(0)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    orc = oracle_lib.load()
    S, B, T = 9, 384, 40
    board, seats = premixed(orc, B, S, 27, seed=91)
    rands = np.random.default_rng(17).random((T - 1, B, T)).astype(np.float16).view(np.uint16)
    want = oracle_search(orc, board, seats, T, rands)                  # ONE search over all B envs: the batch-global q range
    sl = parallel.shard(B, rank, world)
    calls = [0]

    def sync(row):
        # the real exchange between the two processes: the row travels through the gloo collective (staged through the host: the
        # ranks share one GPU here, and RCCL refuses two ranks on one device) as the int32 words it is made of -- ONE all-reduce(MAX)
        calls[0] += 1
        host = row.cpu()
        parallel.allreduce_qrange(host)
        row.copy_(host)
    world_ = Hex(board=torch.from_numpy(board[sl]).cuda(), seats=torch.from_numpy(seats[sl]).cuda())
    net = HashNetwork('cuda')
    m = MCTS(world_, n_nodes=T, rng=ReplayRng(np.ascontiguousarray(rands[:, sl]), 'cuda'), noise_eps=0., qrange_sync=sync)
    d = net(world_)
    m.plant_root(d.logits, d.v)
    for _ in range(T - 1):
        m.simulate(net)
    same = all(np.array_equal(to_np(mine), theirs[sl]) for mine, theirs in
               [(m.tree.children, want.children), (m.tree.parents, want.parents), (m.stats.n, want.n), (m.stats.w, want.w),
                (m.worlds.board, want.boards), (m.decisions.logits, want.logits)])
    same = same and np.array_equal(bits16(m.root_probs()), want.root_probs()[sl])
    # ... and WITHOUT the exchange the shard normalises over its own envs and parts from the unsharded search somewhere
    m2 = MCTS(world_, n_nodes=T, rng=ReplayRng(np.ascontiguousarray(rands[:, sl]), 'cuda'), noise_eps=0.)
    m2.plant_root(d.logits, d.v)
    for _ in range(T - 1):
        m2.simulate(net)
    differs = not np.array_equal(to_np(m2.stats.n), want.n[sl])
    out.put((rank, bool(same), calls[0], bool(differs)))
    torch.distributed.destroy_process_group()


@pytest.mark.gpu
def test_sharded_search_with_the_qrange_collective_equals_the_unsharded_search():
    """SURVEY 8e option 2 through a REAL collective between two processes: two ranks, each searching its shard
    of 384 envs with MCTS(qrange_sync=...) -- one int32 all-reduce(MAX) of the q-range row per simulation -- reproduce, each on its
    own envs, the ONE unsharded search of all 384 envs (the oracle's) bit for bit; without the exchange they do not."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_sharded_search_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True], res
    assert all(r[2] == 39 for r in res), res
    assert any(r[3] for r in res), 'the shards agree with the unsharded search even without the exchange: the test shows nothing'


@pytest.mark.gpu
def test_train_bench_one_rank_rccl():
    """tools/train_bench.py (config 4's launcher) at world size 1 over the nccl backend, on a reduced shape: self-play with captured
    moves and the fused inference plan, learner steps through the persistent gradient bucket, the all-reduce timed by device events."""
    import json, subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'train_bench.py'), '--steps', '2', '--envs', '256', '--boardsize', '9', '--nodes', '32',
                          '--width', '256', '--depth', '2', '--buffer', '3'], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and d['backend'] == 'nccl' and d['weights_identical_over_ranks']
    assert len(d['learner_step_ms']) == 2 and len(d['allreduce_ms']) == 2 and d['moves'] >= 4 and d['sims_per_sec_whole_job'] > 0


def test_arena_sweep_fans_out_over_worker_processes():
    """tools/arena_sweep.py (config 5's sweep, one job per board size on a pool of workers -- arena/neural.py:257-274's fan-out):
    the dry run plays real matches with deterministic agents on CPU worlds in two worker processes and reports one JSON line."""
    import json, subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'arena_sweep.py'), '--workers', '2', '--boards', '3,5', '--envs', '16', '--repeat', '2'],
                         env={**os.environ, 'ARENA_DRY': '1'}, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['matches'] == 4 and d['games'] == 64 and d['workers'] == 2 and d['dry_run'] and d['sims_per_sec'] > 0
    assert out.stdout.count('board 5x5') == 2 and out.stdout.count('board 3x3') == 2
