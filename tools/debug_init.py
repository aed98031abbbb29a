import os, sys, ctypes, faulthandler
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boardlaw_amd import hex, _native
from boardlaw_amd.mcts import MCTS
worlds = hex.Hex.initial(256, 5, device='cuda')
worlds.board[:, 0, 0] = 1
m = MCTS(worlds, n_nodes=16)
torch.cuda.synchronize()
arrs = dict(children=m.tree.children, parents=m.tree.parents, relation=m.tree.relation, logits=m.decisions.logits, v=m.decisions.v,
            w=m.stats.w, n=m.stats.n, rewards=m.transitions.rewards, terminal=m.transitions.terminal, boards=m.worlds.board,
            seats=m.worlds.seats, qrange=m._qrange)
ref = {k: v.clone() for k, v in arrs.items()}
def dirty():
    for k, v in arrs.items():
        if v.dtype == torch.bool: v.fill_(True)
        elif v.dtype.is_floating_point: v.fill_(3.0)
        else: v.fill_(5)
def check(tag):
    torch.cuda.synchronize()
    bad = [k for k, v in arrs.items() if not torch.equal(v.view(torch.uint8), ref[k].view(torch.uint8))]
    print(tag, 'mismatching arrays:', bad, flush=True)
L = _native.lib()
def init():
    _native.check(L.bl_sim_init(ctypes.byref(m._search), worlds.board.data_ptr(), worlds.seats.data_ptr(), _native.stream()))
dirty(); init(); check('eager init')
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): init()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g): init()
dirty(); check('after dirty (expect all)')
g.replay(); check('replay 1')
dirty(); g.replay(); check('replay 2')
