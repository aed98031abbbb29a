"""Where a learner step goes at the reference's own shape (boardlaw/main.py:147: 9x9, 32768 envs, buffer of 64 moves, 512x4)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import arrdict, learning, networks, training
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTSAgent, MoveRng
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
worlds = Hex.initial(B, 9)
net = networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda()
agent = MCTSAgent(networks.Inference(net, fused=True), n_nodes=64, graph=True, rng=MoveRng())
buf = []
for _ in range(64):
    d, w2, tr = agent.play(worlds)
    buf.append(arrdict.arrdict(worlds=worlds, decisions=d.half(), transitions=learning.half(tr)).detach())
    worlds = w2
opt = torch.optim.Adam(net.parameters(), lr=1e-3); scaler = torch.amp.GradScaler('cuda')
idxs = (torch.randint(64, (B,), device='cuda'), torch.arange(B, device='cuda'))
def t(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
ms_stack, chunk = t(lambda: arrdict.stack(buf))
ms_chunk, (chunk, _) = t(lambda: training.as_chunk(buf, B))
ms_index, batch = t(lambda: chunk[idxs])
ms_opt, _ = t(lambda: training.optimize(net, scaler, opt, batch))
print(f'{B} envs: arrdict.stack {ms_stack:.1f} ms, as_chunk (stack + reward_to_go) {ms_chunk:.1f} ms, chunk[idxs] {ms_index:.1f} ms, optimize {ms_opt:.1f} ms')
for k, v in arrdict.leaves_with_names(chunk) if hasattr(arrdict, 'leaves_with_names') else []:
    print(k, tuple(v.shape), v.dtype)
