"""Can the envs whose descent will be the launch's longest be told BEFORE the launch?  bl_sim_expand at 4096 envs is bound by its deepest
descent; four waves per env shorten a deep chain by ~20 % (profiles/r06_tail_overlap.txt, "B alone") but do not fit
the chip for every env (16384 waves on 8192 slots).  A launch that gives four waves only to a flagged minority needs a flag that
catches the critical envs.  This records, with the counting build, every env's descent length and descent cycles in every simulation of
config-2 searches and scores the predictors a kernel could afford:

  prev      the env's previous descent's length (path[0], what bl_search_t.prio_thresh already reads)
  runmax    the longest descent the env has had in this search (a lower bound of its tree's depth)

For a threshold that flags about K envs: in what share of the simulations is the slowest env (by cycles) flagged, and what is the
slowest UNFLAGGED env's time relative to the slowest overall (the launch's new critical chain if flagged envs became free).
Usage: python tools/deep_predict_probe.py [moves]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTS
from bench import premix

ENVS, T = 4096, 64
moves = int(sys.argv[1]) if len(sys.argv) > 1 else 3
gen = torch.Generator(device='cuda'); gen.manual_seed(0); torch.manual_seed(0)
print(f'# 9x9, {ENVS} envs, {T} sims, FCModel 512x4; counting build of bl_sim_expand (two waves per env)')
for move in range(moves):
    worlds = premix(Hex.initial(ENVS, 9), 27 + 4 * move, gen)
    net = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda(), fused=True)
    net.refresh()
    m = MCTS(worlds, n_nodes=T, count=True, obs_half=True)
    m.initialize(net)
    lv, cyc = [], []
    last = m.counters.clone()
    for sim in range(1, T):
        m.simulate(net)
        torch.cuda.synchronize()
        d = (m.counters - last).cpu().numpy(); last = m.counters.clone()
        lv.append(d[:, 0]); cyc.append(d[:, 8])
    lv, cyc = np.stack(lv), np.stack(cyc).astype(np.float64)          # (T-1, ENVS)
    print(f'move {move}: levels mean {lv.mean():.2f}, per-sim max mean {lv.max(1).mean():.1f}; descent cycles per-sim max mean {cyc.max(1).mean():.0f}, '
          f'mean {cyc.mean():.0f}; corr(levels, cycles) {np.corrcoef(lv.ravel(), cyc.ravel())[0, 1]:.3f}')
    runmax = np.maximum.accumulate(lv, 0)
    for name, pred in (('prev', lv), ('runmax', runmax), ('oracle', None)):
        for K in (256, 512, 1024):
            hit, rel, nflag = [], [], []
            for s in range(8, T - 1):
                score = lv[s] if pred is None else pred[s - 1]
                thr = np.sort(score)[-K]
                flag = score >= max(thr, 1)
                worst = cyc[s].argmax()
                hit.append(bool(flag[worst])); nflag.append(int(flag.sum()))
                rel.append(cyc[s][~flag].max() / cyc[s].max() if (~flag).any() else 0.)
            print(f'   {name:7s} K~{K:4d}: flagged {np.mean(nflag):6.0f} envs; slowest env flagged in {100 * np.mean(hit):5.1f} % of sims; '
                  f'slowest unflagged / slowest = {np.mean(rel):.3f}')
    for theta in (8, 10, 12, 14, 16):          # a fixed threshold on runmax, as a kernel would apply it
        hit, rel, nflag = [], [], []
        for s in range(8, T - 1):
            flag = runmax[s - 1] >= theta
            worst = cyc[s].argmax()
            hit.append(bool(flag[worst])); nflag.append(int(flag.sum()))
            rel.append(cyc[s][~flag].max() / cyc[s].max() if (~flag).any() else 0.)
        print(f'   runmax >= {theta:2d}: flagged mean {np.mean(nflag):6.0f} / at sim 32 {nflag[32 - 8]:5d} / last {nflag[-1]:5d} envs; slowest env flagged in '
              f'{100 * np.mean(hit):5.1f} % of sims; slowest unflagged / slowest = {np.mean(rel):.3f}')
    del m, net
