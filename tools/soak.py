"""600 self-play moves of bench.py's configuration on one captured graph: rate over a long run (positions drift from the bench's
pre-mixed boards to the stationary mix of a self-play run: 44.6 M sims/s), allocator growth, legality.  GPU box: python tools/soak.py [envs [moves]]
(round 6: `python tools/soak.py 32768 300` -- the reference's own actor shape)"""
import sys, time, torch
sys.path.insert(0, '.')
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTSAgent, MoveRng
from bench import premix
torch.manual_seed(0)
gen = torch.Generator(device='cuda'); gen.manual_seed(0)
ENVS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
MOVES = int(sys.argv[2]) if len(sys.argv) > 2 else 600
w = premix(Hex.initial(ENVS, 9), 27, gen)
net = networks.FCModel(w.obs_space, w.action_space, 512, 4).cuda()
agent = MCTSAgent(networks.Inference(net, fused=True), n_nodes=64, graph=True, rng=MoveRng())
for _ in range(3): w = agent.play(w)[1]
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved(); t0 = time.perf_counter()
term = 0
for i in range(MOVES):
    d, w, tr = agent.play(w)
    if i % 100 == 99:
        term += int(tr.terminal.sum())
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(ENVS, 'envs', MOVES, 'moves', round(dt, 2), 's', round(ENVS * 64 * MOVES / dt / 1e6, 2), 'M sims/s; allocated', m0 >> 20, '->', torch.cuda.memory_allocated() >> 20, 'MiB; reserved', r0 >> 20, '->', torch.cuda.memory_reserved() >> 20, 'MiB; legal', bool(w.valid.any(-1).all()), 'terminals seen', term)
