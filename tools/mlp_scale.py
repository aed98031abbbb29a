"""Fused MLP forward time against the number of 32-row workgroups (is the L2 the limit when every CU streams the weights?)"""
import torch, sys
sys.path.insert(0, '.')
from boardlaw_amd import networks, heads
S, width, depth = 9, 512, 4
net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(S * S), width=width, depth=depth).cuda()
fused = networks.Inference(net, fused=True); fused.refresh()
class W_: pass
for B in (32, 1024, 2048, 4096, 6144, 8192, 16384):
    w = W_(); w.obs = (torch.rand(B, S, S, 2, device='cuda') < .3).half()
    with torch.no_grad(), torch.autocast('cuda'):
        for _ in range(5): fused.raw(w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): fused.raw(w)
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(B, 'rows', B // 32, 'workgroups', round(e0.elapsed_time(e1) * 1000 / 20, 2), 'us per forward')
