import os, sys, faulthandler
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boardlaw_amd import hex, networks
from boardlaw_amd.mcts import MCTSAgent, MCTS
DEV='cuda'
torch.manual_seed(0)
B = int(sys.argv[1]); mode = sys.argv[2]
worlds = hex.Hex.initial(B, 5, device=DEV)
net = networks.FCModel(worlds.obs_space, worlds.action_space, width=32, depth=2).to(DEV)
eager, graphed = MCTSAgent(net, n_nodes=16), MCTSAgent(net, n_nodes=16, graph=True)
def sync(tag):
    torch.cuda.synchronize(); print('ok', tag, flush=True)
graphed(worlds); sync('capture+first replay')
for move in range(3):
    if mode == 'eager':
        a = eager(worlds); sync('eager move')
    elif mode == 'torchonly':
        x = torch.randn(4096, 4096, device=DEV); y = x @ x; z = torch.rand_like(y); del x, y, z; sync('torch work')
    elif mode == 'initonly':
        m = MCTS(worlds, n_nodes=16); sync('eager MCTS init'); del m
    elif mode == 'initnet':
        m = MCTS(worlds, n_nodes=16); m.initialize(net); sync('eager MCTS init+initialize'); del m
    elif mode == 'sim1':
        m = MCTS(worlds, n_nodes=16); m.initialize(net); m.simulate(net); sync('eager 1 sim'); del m
    b = graphed(worlds); sync('replay')
