"""How often does transition_q's batch-global (min, max) change from one simulation to the next?  (It is the only
cross-env dependency of the search: if it were stable, envs could run ahead of the batch's slowest descent.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import networks, _native
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTS
from bench import premix

gen = torch.Generator(device='cuda'); gen.manual_seed(0); torch.manual_seed(0)
worlds = premix(Hex.initial(4096, 9), 27, gen)
net = networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda()
m = MCTS(worlds, n_nodes=64)
m.initialize(net)
for _ in range(63):
    m.simulate(net)
rows = [tuple(_native.qrange_decode(m._qrange[s]).tolist()) for s in range(1, 65)]
same = [rows[i] == rows[i - 1] for i in range(1, len(rows))]
print('identical to the previous sim:', sum(same), 'of', len(same))
for s in (1, 2, 3, 4, 8, 16, 32, 48, 63, 64):
    print(f'  descend #{s}: lo {rows[s - 1][0]:+.6f} hi {rows[s - 1][1]:+.6f}')
print('changes at sims:', [i + 2 for i, x in enumerate(same) if not x])
