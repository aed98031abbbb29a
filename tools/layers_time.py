"""Per-call time of the leaf network's three plans at one shape: the one-kernel plan (bl_mlp_forward_f16), a launch per
Linear (bl_mlp_layers_f16), and the library GEMMs + bl_rezero_relu_f16.  usage: tools/layers_time.py [S width depth rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boardlaw_amd import networks, heads
S, width, depth, B = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (13, 1024, 8, 1024)))
net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(S * S), width=width, depth=depth).cuda()
class W_: pass
w = W_(); w.obs = (torch.rand(B, S, S, 2, device='cuda') < .3).half()
plan = networks.Inference(net, fused=True); plan.refresh()
def timed(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 10): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n // 10 * 10) * 1e3
res = {}
plan.FUSED_ALWAYS_BYTES = 1 << 40; res['one kernel'] = timed(lambda: plan.raw(w))
plan.FUSED_ALWAYS_BYTES = 0; plan.FUSED_MIN_TILES = 1 << 30
plan.PERSIST_PLAN = True; res['all Linears in one launch (row-tile sync)'] = timed(lambda: plan.raw(w))
plan.PERSIST_PLAN = 'xcd'; res['all Linears in one launch (hand-off inside the XCD L2)'] = timed(lambda: plan.raw(w))
plan.PERSIST_PLAN = False; res['launch per Linear'] = timed(lambda: plan.raw(w))
plan.LAYERS_PLAN = False; res['library GEMMs'] = timed(lambda: plan.raw(w))
assert not plan.persist_error()
print(f'{S}x{S} board, {width}x{depth}, {B} rows (us per forward, inside a captured graph): ' + ', '.join(f'{k} {v:.1f}' for k, v in res.items()))
