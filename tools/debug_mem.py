"""Does memory grow over captured self-play moves?  (round-6 check at 32768 envs)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTSAgent, MoveRng
from bench import premix
ENVS = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
gen = torch.Generator(device='cuda'); gen.manual_seed(0); torch.manual_seed(0)
w = premix(Hex.initial(ENVS, 9), 27, gen)
net = networks.FCModel(w.obs_space, w.action_space, 512, 4).cuda()
agent = MCTSAgent(networks.Inference(net, fused=True), n_nodes=64, graph=True, rng=MoveRng())
for i in range(401):
    d, w, tr = agent.play(w)
    if i % 50 == 0:
        torch.cuda.synchronize(); print(i, 'allocated', torch.cuda.memory_allocated() >> 20, 'MiB reserved', torch.cuda.memory_reserved() >> 20, flush=True)
