"""Pricing of VERDICT r05 item 3: inside ONE simulation the q-range is fixed, so the network + finish step of the envs whose
descent is short could start while the deep descents are still going.  Before building the plumbing (a finish kernel that works on
a permuted subset of the envs, two captured branches per simulation) this measures its ceiling on one box, with the kernels as
they are:

  baseline   bl_sim_expand(all 4096 envs, two waves per env) -> bl_sim_infer_finish(all), one stream
  branch A   bl_sim_expand(all but the K deepest, via bl_search_t.order) -> bl_sim_infer_finish on 4096 - K rows (n_active)
  branch B   bl_sim_expand(the K deepest, four waves per env)            -> bl_sim_infer_finish on K rows (a second search of K envs kept
                                                                            in step: same kernel, same row count, other envs)
  forked     A and B on two streams, from one event to the join of both

at simulations 16 / 32 / 48 of a config-2 search (9x9, 64 sims, FCModel 512x4), K = 512 and 1024, the deepest envs taken from the
counting build's levels of the SAME launch (an oracle a real implementation would not have: it would sort by the previous
simulation's depth).  The tree state is restored before every timed repetition; results of the forked run are not meaningful (the
branches' finish steps write the same search), only its duration is.  Usage: python tools/tail_overlap_probe.py"""
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

ENVS, T = 4096, 64
gen = torch.Generator(device='cuda'); gen.manual_seed(0); torch.manual_seed(0)
worlds = premix(Hex.initial(ENVS, 9), 27, gen)
net = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda(), fused=True)
net.refresh()
L = _native.lib()
m = MCTS(worlds, n_nodes=T, count=True, obs_half=True)
m.initialize(net)
small = {}
for K in (512, 1024):
    small[K] = MCTS(Hex(board=worlds.board[:K].clone(), seats=worlds.seats[:K].clone()), n_nodes=T, obs_half=True)
    small[K].initialize(net)
fp = net.fused_params(ENVS)
nacts = {K: torch.full((1,), ENVS - K, dtype=torch.int32, device='cuda') for K in (512, 1024)}      # device scalars, set before any capture


def state_tensors(s):
    return [s.tree.children, s.tree.parents, s.tree.relation, s.worlds.board, s.worlds.seats, s.transitions.rewards, s.transitions.terminal,
            s.decisions.logits, s.decisions.v, s.stats.n, s.stats.w, s._qrange, s._leaves, s._obs, s._valid, s._leaf_seats, s._path, s._cpi,
            s._cca, s._nk, s._fav]


def expand(s, rands, order=None, waves=0, counted=False):
    s._search.order = order.data_ptr() if order is not None else None
    s._search.tune.expand_waves = waves
    st = _native.stream(s.device)
    if counted:
        _native.check(L.bl_sim_expand_counted(ctypes.byref(s._search), s.sim, rands.data_ptr(), s._leaves.data_ptr(), s._obs.data_ptr(), s._valid.data_ptr(),
                                              s._leaf_seats.data_ptr(), s.counters.data_ptr(), st))
    else:
        _native.check(L.bl_sim_expand(ctypes.byref(s._search), s.sim, rands.data_ptr(), s._leaves.data_ptr(), s._obs.data_ptr(), s._valid.data_ptr(),
                                      s._leaf_seats.data_ptr(), st))
    s._search.order = None
    s._search.tune.expand_waves = 0


def finish(s, rows=None):
    if rows is not None:
        s._search.n_active = nacts[ENVS - rows].data_ptr()
    _native.check(L.bl_sim_infer_finish(ctypes.byref(s._search), s.sim, s._leaves.data_ptr(), s._obs.data_ptr(), s._valid.data_ptr(), s._leaf_seats.data_ptr(),
                                        fp['w0'], fp['b0'], fp['wb'], fp['bb'], fp['al'], fp['wh'], fp['bh'], fp['W'], fp['D'], fp['K0pad'], fp['NHpad'],
                                        _native.stream(s.device)))
    s._search.n_active = None


def median_us(fn, restore, reps=7):
    """fn is captured into a HIP graph (eager launches cost the host ~10 us each: four launches and their events on two streams
    took longer to ISSUE than the kernels run); the graph is replayed `reps` times, the state restored before each."""
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    restore()
    with torch.cuda.stream(side):
        fn()                                   # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    restore()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    us = []
    for _ in range(reps):
        restore()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record()
        torch.cuda.synchronize()
        us.append(1e3 * a.elapsed_time(b))
    return float(np.median(us))


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print(f'# tail-overlap pricing, config 2 (9x9, {ENVS} envs, {T} sims, FCModel 512x4), one MI355X; microseconds, every variant a captured graph, median of 7 replays')
print('# (the tree state -- 300 MB -- is restored before every replay, so every variant starts with cold caches: the absolute figures are above the')
print('#  in-search ones, 48 + 36 us; the comparison is like for like)')
gains = []
for sim in range(1, T):
    if sim in (16, 32, 48):
        rands = torch.rand((ENVS, T), device='cuda', generator=gen).half()
        snap = [t.clone() for t in state_tensors(m)]
        snaps = {K: [t.clone() for t in state_tensors(small[K])] for K in small}

        def restore():
            for t, c in zip(state_tensors(m), snap):
                t.copy_(c)
            for K in small:
                for t, c in zip(state_tensors(small[K]), snaps[K]):
                    t.copy_(c)
        m.counters.zero_()
        expand(m, rands, counted=True)
        torch.cuda.synchronize()
        lv = m.counters[:, 0].cpu().numpy()
        by_depth = np.argsort(-lv, kind='stable')
        restore()
        # the small searches need leaves of their own for their finish step
        rsmall = {K: torch.rand((K, T), device='cuda', generator=gen).half() for K in small}

        def order_of(envs):
            o = np.full(ENVS, ENVS, dtype=np.int32); o[:len(envs)] = envs
            return torch.from_numpy(o).cuda()
        base_e = median_us(lambda: expand(m, rands), restore)
        base = median_us(lambda: (expand(m, rands), finish(m)), restore)
        print(f'sim {sim}: levels mean {lv.mean():.2f} p90 {np.percentile(lv, 90):.0f} max {lv.max():.0f} | baseline expand {base_e:.1f} + finish = {base:.1f}')
        for K in (512, 1024):
            shallow, deep = order_of(by_depth[K:]), order_of(by_depth[:K])
            sm = small[K]
            a_e = median_us(lambda: expand(m, rands, shallow), restore)
            a_f = median_us(lambda: finish(m, ENVS - K), lambda: (restore(), expand(m, rands, shallow)))
            b_e2 = median_us(lambda: expand(m, rands, deep, waves=2), restore)
            b_e4 = median_us(lambda: expand(m, rands, deep, waves=4), restore)
            b_f = median_us(lambda: finish(sm), lambda: (restore(), expand(sm, rsmall[K])))

            def forked():
                cur = torch.cuda.current_stream()
                s1.wait_stream(cur); s2.wait_stream(cur)
                with torch.cuda.stream(s1):
                    expand(m, rands, shallow); finish(m, ENVS - K)
                with torch.cuda.stream(s2):
                    expand(m, rands, deep, waves=4); finish(sm)
                cur.wait_stream(s1); cur.wait_stream(s2)
            fk = median_us(forked, lambda: (restore(), expand(sm, rsmall[K])))
            gains.append(base / fk - 1)
            print(f'   K = {K:4d} (deep envs: >= {lv[by_depth[K - 1]]:.0f} levels): A alone expand {a_e:.1f} + finish {a_f:.1f} = {a_e + a_f:.1f} | '
                  f'B alone expand {b_e2:.1f} (2 waves) / {b_e4:.1f} (4 waves) + finish {b_f:.1f} = {min(b_e2, b_e4) + b_f:.1f} | '
                  f'forked on two streams {fk:.1f} vs baseline {base:.1f}: {100 * (base / fk - 1):+.1f} %')
        restore()
    m.simulate(net)
    for K in small:
        small[K].simulate(net)
print(f'# forked / baseline over all cases: mean {100 * float(np.mean(gains)):+.1f} %, best {100 * float(np.max(gains)):+.1f} %  (kill line of the verdict: < +4 %)')
