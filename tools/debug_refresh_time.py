"""Cost of networks.Inference.refresh() (the re-cast + re-pack of the weights after every optimiser step)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import networks, heads
for S, W, D in ((9, 512, 4), (13, 1024, 8)):
    net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(S * S), width=W, depth=D).cuda()
    inf = networks.Inference(net, fused=True); inf.refresh()
    for _ in range(3): inf.refresh()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): inf.refresh()
    torch.cuda.synchronize(); print(f'{S}x{S} {W}x{D}: refresh {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms')
