"""Upper bound for running the batch as G independent env groups out of step: G agents of 4096/G envs each (independent
searches -- NOT the batch-global q normalisation of one 4096-env search), each move captured as its own HIP graph and
replayed on its own stream, all groups concurrently.  Compared with one 4096-env agent."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTSAgent, MoveRng
from bench import premix

ENVS, NODES, STEPS = 4096, 64, 20
torch.manual_seed(0)
gen = torch.Generator(device='cuda'); gen.manual_seed(0)
base = Hex.initial(ENVS, 9)
net = networks.FCModel(base.obs_space, base.action_space, 512, 4).cuda()
base = premix(base, 27, gen)
inf = networks.Inference(net, fused=True)

for G in (1, 2, 4):
    n = ENVS // G
    worlds = [Hex(board=base.board[i * n:(i + 1) * n].clone(), seats=base.seats[i * n:(i + 1) * n].clone()) for i in range(G)]
    agents = [MCTSAgent(inf, n_nodes=NODES, graph=True, rng=MoveRng()) for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    for _ in range(3):                                   # capture + warm-up
        for i in range(G):
            with torch.cuda.stream(streams[i]):
                worlds[i] = agents[i].play(worlds[i])[1]
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        for i in range(G):
            with torch.cuda.stream(streams[i]):
                worlds[i] = agents[i].play(worlds[i])[1]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{G} group(s) of {n} envs: {ENVS * NODES * STEPS / dt / 1e6:.2f} M sims/s, {1e3 * dt / STEPS:.3f} ms per 4096-env move')
