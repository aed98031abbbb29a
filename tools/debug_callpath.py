"""Which of MCTSAgent's call paths replays at what rate, in which order (round-6 debugging of a slow `agent(worlds)` after other agents)."""
import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTSAgent, MoveRng
from bench import premix
gen = torch.Generator(device='cuda'); gen.manual_seed(0); torch.manual_seed(0)
worlds = premix(Hex.initial(4096, 9), 27, gen)
net = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda(), fused=True)
for mode in sys.argv[1:]:
    agent = MCTSAgent(net, n_nodes=64, graph=True, rng=MoveRng(), pad=(not mode.endswith('nopad')))
    f = (lambda: agent.play(worlds)) if mode.startswith('play') else (lambda: agent(worlds))
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): f()
    torch.cuda.synchronize(); print(mode, round((time.time()-t0)/10*1e3, 2), 'ms per move', flush=True)
    if mode.endswith('!'): del agent
