"""How much of the GPU G env groups running OUT OF STEP can use, with the host taken out of the picture: G agents of 4096/G
envs each (independent searches -- NOT the batch-global q normalisation of one 4096-env search), each move a captured HIP
graph on the group's own stream.  tools/two_group_probe.py issues every group's replay from one host loop; this probe also
(a) times the host side of a replay (the loop without a device sync) and (b) issues each group's replays from its own host
thread, so that the rate it reports is the device's and not the launch path's."""
import os, sys, threading, time
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

for G in [int(g) for g in os.environ.get('GRP', '1,2,4,8').split(',')]:
    n = ENVS // G
    worlds = [Hex(board=base.board[i * n:(i + 1) * n].clone(), seats=base.seats[i * n:(i + 1) * n].clone()) for i in range(G)]
    gens = [torch.Generator(device='cuda') for _ in range(G)]
    agents = [MCTSAgent(inf, n_nodes=NODES, graph=True, rng=MoveRng(generator=gens[i])) for i in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    for _ in range(3):                                   # capture + warm-up
        for i in range(G):
            with torch.cuda.stream(streams[i]):
                worlds[i] = agents[i].play(worlds[i])[1]
        torch.cuda.synchronize()
    # (a) one host loop
    t0 = time.perf_counter()
    for _ in range(STEPS):
        for i in range(G):
            with torch.cuda.stream(streams[i]):
                worlds[i] = agents[i].play(worlds[i])[1]
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{G} group(s) of {n} envs, one host loop : {ENVS * NODES * STEPS / dt / 1e6:6.2f} M sims/s, {1e3 * dt / STEPS:.3f} ms per 4096-env move; '
          f'host issue time {1e3 * th / STEPS / G:.3f} ms per replay', flush=True)

    # (b) a host thread per group
    def work(i):
        with torch.cuda.stream(streams[i]):
            w = worlds[i]
            for _ in range(STEPS):
                w = agents[i].play(w)[1]
            worlds[i] = w
    ts = [threading.Thread(target=work, args=(i,)) for i in range(G)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{G} group(s) of {n} envs, thread per group: {ENVS * NODES * STEPS / dt / 1e6:6.2f} M sims/s, {1e3 * dt / STEPS:.3f} ms per 4096-env move', flush=True)
    del agents, worlds
