"""Distribution of per-env descent work and where a wave's cycles go (counting build of bl_sim_expand)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTS
from bench import premix

gen = torch.Generator(device='cuda'); gen.manual_seed(0); torch.manual_seed(0)
ENVS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
worlds = premix(Hex.initial(ENVS, 9), 27, gen)
net = networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda()
m = MCTS(worlds, n_nodes=64, count=True)
m.initialize(net)
for sim in range(1, 64):
    m.counters.zero_()
    m.simulate(net)
    if sim in (2, 16, 32, 63):
        c = m.counters.cpu().numpy().astype(np.float64)
        lv, it = c[:, 0], c[:, 1]
        worst = int(np.argmax(c[:, 8] + c[:, 9]))
        def row(x): return f'levels {x[0]:.1f} iters {x[1]:.1f} | cycles: loads {x[4]:.0f} (rows {x[10]:.0f}, +stats {x[11]:.0f}) terms {x[5]:.0f} folds {x[6]:.0f} update {x[7]:.0f} | descent {x[8]:.0f} expansion {x[9]:.0f}'
        print(f'sim {sim:2d}: levels max {lv.max():.0f} p99 {np.percentile(lv,99):.0f}; iters max {it.max():.0f}')
        print('   mean env :', row(c.mean(0)))
        print('   worst env:', row(c[worst]))
        tot = c[:, 8] + c[:, 9]
        order = np.argsort(tot)
        print('   total cycles per env (wave 0): p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f | slowest 5 envs: (levels, wave-0 evals, cycles) %s' % (
            *np.percentile(tot, [50, 90, 99, 99.9]), tot.max(), [(int(lv[i]), int(c[i, 10]), int(tot[i])) for i in order[-5:]]))
