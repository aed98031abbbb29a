"""The N>1 path on CPU: two gloo ranks on 127.0.0.1 exercise sharding, the timing/throughput aggregation bench.py
uses, and the q-range all-reduce that makes sharded self-play reproduce the unsharded batch-global normalisation."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


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
    st = np.zeros(_native.QRANGE_WORDS, np.uint32)
    st[64 * (rank * 3 % 64)] = ~_enc(mine.min()); st[64 * (rank * 3 % 64) + 1] = _enc(mine.max())
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
