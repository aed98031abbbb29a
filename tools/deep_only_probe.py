"""How much of a bl_sim_expand launch is its deepest descents' own chain, and how much is everything else being resident?

At simulations 16 / 32 / 63 of a config-2 search the tree state is snapshotted and the SAME launch is repeated on subsets of
the envs (bl_search_t.order maps launch slot -> env; an entry >= B makes the slot's workgroup exit at once): all envs, only
the K deepest (by the counting build's levels), all but the K deepest, K random ones.  HIP events around every launch, the state
restored before each.  Usage: python tools/deep_only_probe.py [envs]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import _native, networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTS
from bench import premix

ENVS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gen = torch.Generator(device='cuda'); gen.manual_seed(0); torch.manual_seed(0)
worlds = premix(Hex.initial(ENVS, 9), 27, gen)
net = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda())
m = MCTS(worlds, n_nodes=64, count=True, obs_half=True)
m.initialize(net)
L = _native.lib()


def state_tensors():
    ts = [m.tree.children, m.tree.parents, m.tree.relation, m.worlds.board, m.worlds.seats, m.transitions.rewards,
          m.transitions.terminal, m.decisions.logits, m.decisions.v, m.stats.n, m.stats.w, m._qrange, m._leaves, m._obs, m._valid,
          m._leaf_seats, m._path, m._cpi, m._cca, m._nk, m._fav]
    return ts


def expand(rands, counted=False):
    s = ctypes.byref(m._search)
    st = _native.stream(m.device)
    if counted:
        _native.check(L.bl_sim_expand_counted(s, m.sim, rands.data_ptr(), m._leaves.data_ptr(), m._obs.data_ptr(), m._valid.data_ptr(),
                                              m._leaf_seats.data_ptr(), m.counters.data_ptr(), st))
    else:
        _native.check(L.bl_sim_expand(s, m.sim, rands.data_ptr(), m._leaves.data_ptr(), m._obs.data_ptr(), m._valid.data_ptr(),
                                      m._leaf_seats.data_ptr(), st))


def timed(order, rands, snap, reps=7):
    us = []
    for _ in range(reps):
        for t, c in zip(state_tensors(), snap):
            t.copy_(c)
        m._search.order = order.data_ptr() if order is not None else None
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); expand(rands); b.record()
        torch.cuda.synchronize()
        us.append(1e3 * a.elapsed_time(b))
    m._search.order = None
    return float(np.median(us))


for sim in range(1, 64):
    if sim in (16, 32, 63):
        rands = torch.rand((ENVS, 64), device='cuda', generator=gen).half()
        snap = [t.clone() for t in state_tensors()]
        m.counters.zero_()
        expand(rands, counted=True)
        torch.cuda.synchronize()
        lv = m.counters[:, 0].cpu().numpy()
        it = m.counters[:, 1].cpu().numpy()
        by_depth = np.argsort(-lv, kind='stable')
        print(f'sim {sim}: levels mean {lv.mean():.2f} p50 {np.percentile(lv, 50):.0f} p90 {np.percentile(lv, 90):.0f} p99 {np.percentile(lv, 99):.0f} max {lv.max():.0f}; '
              f'iterations mean {it.mean():.1f} max {it.max():.0f}; levels histogram {np.bincount(lv.astype(int)).tolist()}')
        full = timed(None, rands, snap)
        print(f'   all {ENVS} envs: {full:.1f} us')
        rng = np.random.default_rng(sim)
        for K in (1, 8, 64, 256, 1024):
            if K >= ENVS:
                continue
            def order_of(envs):
                o = np.full(ENVS, ENVS, dtype=np.int32)
                o[:len(envs)] = envs
                return torch.from_numpy(o).cuda()
            deep = timed(order_of(by_depth[:K]), rands, snap)
            rest = timed(order_of(by_depth[K:]), rands, snap)
            rnd = timed(order_of(rng.choice(ENVS, K, replace=False)), rands, snap)
            print(f'   K = {K:4d}: the K deepest alone {deep:6.1f} us (min depth {lv[by_depth[K - 1]]:.0f}) | all but them {rest:6.1f} us | K random envs {rnd:6.1f} us')
        for t, c in zip(state_tensors(), snap):
            t.copy_(c)
    m.simulate(net)
