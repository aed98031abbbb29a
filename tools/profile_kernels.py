"""Micro-benchmark of the search kernels on a realistic mid-search tree (9x9, 4096 envs, 64 nodes): builds the tree
with the real network, then re-launches each kernel at several simulation depths and times it with HIP events.
Usage: python tools/profile_kernels.py [--envs 4096] [--reps 20]   (wrap in rocprofv3 for counters)"""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import _native, networks            # noqa: E402
from boardlaw_amd.hex import Hex                      # noqa: E402
from boardlaw_amd.mcts import MCTS                    # noqa: E402
from bench import premix                              # noqa: E402


def timeit(f, reps):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--board', type=int, default=9)
    ap.add_argument('--nodes', type=int, default=64)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--sims', type=str, default='2,8,16,32,48,63')
    args = ap.parse_args()
    L = _native.lib()
    gen = torch.Generator(device='cuda'); gen.manual_seed(0)
    torch.manual_seed(0)
    worlds = premix(Hex.initial(args.envs, args.board), args.board**2 // 3, gen)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda()
    m = MCTS(worlds, n_nodes=args.nodes)
    m.initialize(net)
    want = [int(s) for s in args.sims.split(',')]
    st = _native.stream()
    s = ctypes.byref(m._search)
    print(f'group override: {os.environ.get("BL_FORCE_GROUP", "-")}; envs {args.envs}, board {args.board}, nodes {args.nodes}')
    while m.sim < args.nodes:
        if m.sim in want:
            rands = torch.rand_like(m.decisions.logits[:, :, 0])
            sim = m.sim
            # bl_sim_expand mutates the tree (and re-running it on its own output is NOT a valid search state), so every
            # timed launch starts from a restored snapshot
            live = [m.tree.children, m.tree.parents, m.tree.relation, m.worlds.board, m.worlds.seats,
                    m.transitions.rewards, m.transitions.terminal]
            snap = [x.clone() for x in live]
            ts = []
            for _ in range(args.reps):
                for x, y in zip(live, snap):
                    x.copy_(y)
                ts.append(timeit(lambda: L.bl_sim_expand(s, sim, rands.data_ptr(), m._leaves.data_ptr(), m._obs.data_ptr(),
                                                         m._valid.data_ptr(), m._leaf_seats.data_ptr(), st), 1))
            for x, y in zip(live, snap):
                x.copy_(y)
            te = sum(ts[1:]) / max(len(ts) - 1, 1)
            probs = torch.empty((args.envs, args.board**2), dtype=torch.half, device='cuda')
            tr = timeit(lambda: L.bl_sim_root(s, sim, probs.data_ptr(), st), args.reps)
            with torch.no_grad(), torch.autocast('cuda'):
                from boardlaw_amd.mcts import LeafWorlds
                w = LeafWorlds(m, m._leaves, m._obs, m._valid, m._leaf_seats)
                tn = timeit(lambda: net(w), args.reps)
            print(f'sim {sim:3d}: expand {te:8.1f} us   root {tr:8.1f} us   network {tn:8.1f} us')
        m.simulate(net)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
