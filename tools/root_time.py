"""Root evaluation (fp32 Linears) time per call: bl_root_mlp_f32 against the hipBLASLt plan.  GPU box: python tools/root_time.py"""
import sys, torch
sys.path.insert(0, '.')
from boardlaw_amd import hex, networks
for S, B, width, depth in ((9, 4096, 512, 4), (13, 1024, 1024, 8), (9, 256, 512, 4)):
    worlds = hex.Hex.initial(B, S, device='cuda')
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=width, depth=depth).cuda()
    for fused in (False, True):
        inf = networks.Inference(net, fused=fused); inf.refresh()
        for _ in range(3): inf.root_raw(worlds)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10): inf.root_raw(worlds)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        flop = 2 * B * (2 * S * S * width + depth * width * width + width * (S * S + 1))
        print(f'{S}x{S} B={B} {width}x{depth} {"bl_root_mlp_f32" if fused else "hipBLASLt plan "}: {us:7.1f} us  {flop / us / 1e6:6.1f} TFLOP/s fp32')
