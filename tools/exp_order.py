"""Experiment: does the dispatch order / wave priority of bl_sim_expand change its duration?

The launch ends with its deepest env.  Hypothesis: waves dispatched first are the oldest on their SIMD and win the issue
arbitration, so dispatching the envs with the deepest trees first (bl_search_t.order) lets them run at close to their
uncontended speed.  Modes, all on the bench workload (9x9, 4096 envs, 64 nodes, FCModel 512x4 fused), launched eagerly with
HIP events around every bl_sim_expand:
    base        slot == env
    deep_first  order = argsort(previous path length, descending)      (torch ops between launches: eager only)
    deep_last   ... ascending (control)
    prioN       s_setprio 3 for envs whose previous path length >= N
    deep_first+prioN
Prints mean expand us per mode and per third of the search."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import premix  # noqa: E402
from boardlaw_amd import _native, networks  # noqa: E402
from boardlaw_amd.hex import Hex  # noqa: E402
from boardlaw_amd.mcts import MCTS, MoveRng  # noqa: E402


def main(envs=4096, board=9, nodes=64, moves=4):
    lib = _native.lib()
    gen = torch.Generator(device='cuda'); gen.manual_seed(1000)
    torch.manual_seed(0)
    worlds = Hex.initial(envs, board)
    net = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, width=512, depth=4).cuda(), fused=True)
    worlds = premix(worlds, board * board // 3, gen)
    net.refresh_if_stale()
    orig = lib.bl_sim_expand

    modes = [('base', None, 0), ('deep_first', 'desc', 0), ('deep_last', 'asc', 0), ('prio6', None, 6), ('prio10', None, 10),
             ('prio16', None, 16), ('deep_first+prio8', 'desc', 8), ('base2', None, 0)]
    results = {}
    for name, order_mode, prio in modes:
        torch.manual_seed(7)
        pairs = []

        def timed(*args):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); rc = orig(*args); b.record()
            pairs.append((a, b))
            return rc
        lib.bl_sim_expand = timed
        w = worlds
        sim_idx = []
        totals = []
        for mv in range(moves):
            m = MCTS(w, n_nodes=nodes, rng=MoveRng(), obs_half=True)
            m.rng.start(nodes - 1, m.decisions.logits[:, :, 0])
            m._path.zero_()
            order = torch.arange(envs, dtype=torch.int32, device='cuda')
            m._search.order = order.data_ptr() if order_mode else None
            m._search.prio_thresh = prio
            m.initialize(net)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for s in range(nodes - 1):
                if order_mode and s > 0:
                    order.copy_(torch.argsort(m._path[:, 0].int(), descending=(order_mode == 'desc'), stable=True).int())
                sim_idx.append(s)
                m.simulate(net)
            t1.record()
            torch.cuda.synchronize()
            totals.append(t0.elapsed_time(t1))
            probs = m.root_probs().float()
            w, _ = w.step(torch.distributions.Categorical(probs=probs).sample(), check=False)
        us = np.array([1e3 * a.elapsed_time(b) for a, b in pairs])
        si = np.array(sim_idx)
        # skip the first move (warm-up)
        keep = np.arange(len(us)) >= (nodes - 1)
        us, si = us[keep], si[keep]
        thirds = [us[(si >= lo) & (si < hi)].mean() for lo, hi in [(0, 21), (21, 42), (42, 63)]]
        results[name] = (us.mean(), thirds, np.mean(totals[1:]))
        print(f'{name:18s} expand mean {us.mean():6.1f} us   by third {thirds[0]:6.1f} {thirds[1]:6.1f} {thirds[2]:6.1f}   '
              f'eager move {np.mean(totals[1:]):7.2f} ms', flush=True)
    lib.bl_sim_expand = orig


if __name__ == '__main__':
    main(*[int(a) for a in sys.argv[1:]])
