"""bl_sim_expand's duration by simulation number, for several launch variants on ONE box: HIP events around every launch of
eagerly launched self-play moves (bench.py's workload).  Usage: python tools/expand_by_sim.py "ENVS:HELP ..." [moves]
e.g. "1:0 2:0 4:0"."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import TimedExpand, premix
from boardlaw_amd import _native, networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTSAgent, MoveRng

variants = [tuple(int(x) for x in v.split(':')) for v in (sys.argv[1] if len(sys.argv) > 1 else '1:0 2:0').split()]
moves = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lib = _native.lib()
gen = torch.Generator(device='cuda'); gen.manual_seed(1000); torch.manual_seed(0)
worlds0 = Hex.initial(4096, 9)
net = networks.Inference(networks.FCModel(worlds0.obs_space, worlds0.action_space, 512, 4).cuda(), fused=True)
worlds0 = premix(worlds0, 27, gen)
curves = {}
for envs, help_ in variants:
    os.environ['BL_EXPAND_ENVS'], os.environ['BL_EXPAND_HELP'] = str(envs), str(help_)
    timer = TimedExpand(lib); lib.bl_sim_expand = timer
    agent = MCTSAgent(net, n_nodes=64, graph=False, rng=MoveRng())
    torch.manual_seed(1)
    w = worlds0
    for _ in range(2):
        w = agent.play(w)[1]
    timer.on = True
    for _ in range(moves):
        w = agent.play(w)[1]
    torch.cuda.synchronize()
    lib.bl_sim_expand = timer.orig
    us = np.array([1e3 * a.elapsed_time(b) for a, b in timer.pairs]).reshape(moves, 63)
    curves[envs, help_] = us.mean(0)
    print(f'envs={envs} help={help_}: mean {us.mean():.1f} us per launch')
print('sim ' + ' '.join(f'{e}:{h:>5}' for e, h in variants))
for s in range(63):
    print(f'{s + 1:3d} ' + ' '.join(f'{curves[v][s]:7.1f}' for v in variants))
base = curves[variants[0]]
for v in variants[1:]:
    gain = base - curves[v]
    best_from = int(np.argmax(np.cumsum(gain[::-1])[::-1])) + 1
    print(f'variant {v} from simulation s on, else {variants[0]}: best s = {best_from}, mean launch {np.where(np.arange(63) + 1 >= best_from, curves[v], base).mean():.2f} us vs {base.mean():.2f}')
