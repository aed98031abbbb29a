"""G independent config-2 searches (4096 envs x 64 sims each, every one with the reference's batch-global q normalisation over
its own 4096 envs) resident on ONE GPU, each move a captured HIP graph replayed on the actor's own stream, all actors
concurrently: aggregate sims/s against one actor.  Both search kernels are latency-bound (bl_sim_expand ends with its deepest
descent, bl_sim_infer_finish is one workgroup's dependent chain on half of the CUs), so a second actor fills idle cycles."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTSAgent, MoveRng
from bench import premix

ENVS, NODES, STEPS = 4096, 64, 20
torch.manual_seed(0)
net = None
for G in (1, 2, 3, 4):
    gens = [torch.Generator(device='cuda') for _ in range(G)]
    worlds = []
    for i, g in enumerate(gens):
        g.manual_seed(i)
        w = Hex.initial(ENVS, 9)
        if net is None:
            net = networks.FCModel(w.obs_space, w.action_space, 512, 4).cuda()
            inf = networks.Inference(net, fused=True)
        worlds.append(premix(w, 27, g))
    agents = [MCTSAgent(inf, n_nodes=NODES, graph=True, rng=MoveRng(generator=gens[i])) for i in range(G)]     # a generator per actor
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
    print(f'{G} actor(s) x {ENVS} envs: {G * ENVS * NODES * STEPS / dt / 1e6:.2f} M sims/s in total, {1e3 * dt / STEPS:.3f} ms per round of moves')
    del agents, worlds
