"""Distribution of per-env descent work (levels, Newton iterations) across a search: explains the kernel's tail."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTS
from bench import premix

gen = torch.Generator(device='cuda'); gen.manual_seed(0); torch.manual_seed(0)
worlds = premix(Hex.initial(4096, 9), 27, gen)
net = networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda()
m = MCTS(worlds, n_nodes=64, count=True)
m.initialize(net)
for sim in range(1, 64):
    m.counters.zero_()
    m.simulate(net)
    if sim in (2, 8, 16, 32, 48, 63):
        c = m.counters.cpu().numpy()
        per = c[3:].reshape(-1, 3)
        lv, it, mx = per[:, 0], per[:, 1], per[:, 2]
        print(f'sim {sim:2d}: levels mean {lv.mean():.2f} max {lv.max()} p99 {np.percentile(lv,99):.0f} | iters/env mean {it.mean():.1f} max {it.max()} p99 {np.percentile(it,99):.0f} | worst level iters max {mx.max()} p99 {np.percentile(mx,99):.0f} (#envs with a level >=20 its: {(mx>=20).sum()}, ==100: {(mx>=100).sum()})')
