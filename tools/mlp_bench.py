"""Times networks.Inference(fused=True).raw vs the torch GEMM plan vs the plain autocast module for several batch sizes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boardlaw_amd import networks, heads

def timeit(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

S, width, depth = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (9, 512, 4)))
A = S * S
net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(S * S), width=width, depth=depth).cuda()
fused, plan = networks.Inference(net, fused=True), networks.Inference(net)
fused.refresh(); plan.refresh()
class W: pass
for B in (32, 256, 1024, 2048, 4096, 8192, 16384):
    w = W(); w.obs = (torch.rand(B, S, S, 2, device='cuda') < .3).half()
    with torch.no_grad(), torch.autocast('cuda'):
        tf = timeit(lambda: fused.raw(w)); tp = timeit(lambda: plan.raw(w)); tm = timeit(lambda: net.raw(w))
    flops = 2 * B * (2 * A * width + depth * width * width + width * (A + 1))
    print(f'B {B:6d}: fused {tf:7.1f} us ({flops / tf / 1e6:6.1f} TFLOP/s)   torch-gemm plan {tp:7.1f} us   autocast module {tm:7.1f} us')


import ctypes, numpy as np
from boardlaw_amd import _native
L = ctypes.CDLL(os.environ['BL_MLP_CLK_LIB']) if 'BL_MLP_CLK_LIB' in os.environ else None
if L is not None:
    pk = fused._packed; Wd, K0, K0pad, D, NH, NHpad = pk['dims']
    w = W(); w.obs = (torch.rand(4096, S, S, 2, device='cuda') < .3).half()
    x0 = w.obs.reshape(4096, -1).contiguous(); pol = torch.empty(4096, NH - 1, dtype=torch.half, device='cuda'); val = torch.empty(4096, dtype=torch.half, device='cuda')
    vp = ctypes.c_void_p
    for _ in range(3):
        L.bl_mlp_forward_f16(vp(x0.data_ptr()), 4096, K0, vp(pk['w0'].data_ptr()), vp(fused._static[0][1].data_ptr()), vp(pk['wb'].data_ptr()), vp(pk['bb'].data_ptr()),
                             vp(pk['al'].data_ptr()), vp(pk['wh'].data_ptr()), vp(pk['bh'].data_ptr()), Wd, D, K0pad, NH, NHpad, vp(pol.data_ptr()), vp(val.data_ptr()), vp(0))
        torch.cuda.synchronize()
    clk = np.zeros(64, np.int64); L.bl_mlp_debug_clk(vp(clk.ctypes.data))
    names = {1: 'staged'}
    for l in range(5): names.update({2 + 3 * l: f'layer{l} gemm', 3 + 3 * l: f'layer{l} epilogue', 4 + 3 * l: f'layer{l} barrier'})
    names[40] = 'heads'
    prev = clk[0]
    for i in sorted(names):
        print(f'{names[i]:18s} +{clk[i] - prev:7d}  (t={clk[i] - clk[0]})'); prev = clk[i]
