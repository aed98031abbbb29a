"""What would bl_sim_infer_finish gain if its first weights were already in the L2s when it starts?  Inside a search it follows bl_sim_expand,
whose traffic has pushed the network's weights out of the 4 MB L2s: every workgroup's first fragments (the intake Linear, the first blocks of
layer 1) come from further away, and the staging phase (7k of 73k cycles, profiles/r06_mlp_phases.txt) is about as long as that first fetch.
This times, at simulation 20 / 40 of a config-2 search (state restored before every repetition, HIP events, eager):

  cold     bl_sim_expand, then bl_sim_infer_finish            (what the search does)
  warm     bl_sim_expand, then a kernel that reads the whole packed network once per XCD, then bl_sim_infer_finish
  twice    bl_sim_expand, bl_sim_infer_finish, bl_sim_infer_finish again (the second one: weights AND activations warm; results meaningless)

Usage: python tools/warm_l2_probe.py"""
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

ENVS, T = int(os.environ.get('PROBE_ENVS', 4096)), 64
gen = torch.Generator(device='cuda'); gen.manual_seed(0); torch.manual_seed(0)
worlds = premix(Hex.initial(ENVS, 9), 27, gen)
net = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda(), fused=True)
net.refresh()
L = _native.lib()
m = MCTS(worlds, n_nodes=T, obs_half=True)
m.initialize(net)
fp = net.fused_params(ENVS)



def state_tensors(s):
    return [s.tree.children, s.tree.parents, s.tree.relation, s.worlds.board, s.worlds.seats, s.transitions.rewards, s.transitions.terminal,
            s.decisions.logits, s.decisions.v, s.stats.n, s.stats.w, s._qrange, s._leaves, s._obs, s._valid, s._leaf_seats, s._path, s._cpi,
            s._cca, s._nk, s._fav]


def expand(s, rands):
    _native.check(L.bl_sim_expand(ctypes.byref(s._search), s.sim, rands.data_ptr(), s._leaves.data_ptr(), s._obs.data_ptr(), s._valid.data_ptr(),
                                  s._leaf_seats.data_ptr(), _native.stream(s.device)))


def finish(s):
    _native.check(L.bl_sim_infer_finish(ctypes.byref(s._search), s.sim, s._leaves.data_ptr(), s._obs.data_ptr(), s._valid.data_ptr(), s._leaf_seats.data_ptr(),
                                        fp['w0'], fp['b0'], fp['wb'], fp['bb'], fp['al'], fp['wh'], fp['bh'], fp['W'], fp['D'], fp['K0pad'], fp['NHpad'],
                                        _native.stream(s.device)))


# the packed weights bl_sim_infer_finish streams (w0, wb, wh), read by a torch reduction: which XCD's L2 ends up holding what is the
# hardware's business -- this is the optimistic stand-in for a purpose-built warm-up
weights = [net._packed[k] for k in ('w0', 'wb', 'wh')]
print(f'# {ENVS} envs; warm-up reads {sum(t.numel() * t.element_size() for t in weights) / 1e6:.2f} MB in {len(weights)} tensors')


def warm():
    for t in weights:
        t.view(torch.int16 if t.element_size() == 2 else torch.int32).sum()


def timed(fn_before, fn_timed, restore, reps=9):
    us = []
    for _ in range(reps):
        restore(); torch.cuda.synchronize()
        fn_before()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn_timed(); b.record(); torch.cuda.synchronize()
        us.append(1e3 * a.elapsed_time(b))
    return float(np.median(us))


for sim in range(1, T):
    if sim in (20, 40):
        rands = torch.rand((ENVS, T), device='cuda', generator=gen).half()
        snap = [t.clone() for t in state_tensors(m)]

        def restore():
            for t, c in zip(state_tensors(m), snap):
                t.copy_(c)
        cold = timed(lambda: expand(m, rands), lambda: finish(m), restore)
        warmed = timed(lambda: (expand(m, rands), warm()), lambda: finish(m), restore)
        twice = timed(lambda: (expand(m, rands), finish(m)), lambda: finish(m), restore)
        print(f'sim {sim}: bl_sim_infer_finish after bl_sim_expand {cold:.1f} us | after expand + a pass over the weights {warmed:.1f} us | run again at once {twice:.1f} us')
        restore()
    m.simulate(net)
