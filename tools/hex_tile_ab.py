"""The board kernels two ways at several batch sizes: a group of lanes per env (the launch-per-move kernels) against LDS-staged tiles of
64 consecutive envs with 16-byte loads/stores (bl_hex_*_tiled, round 5).  us per call inside a captured graph, GB/s of algorithmic
bytes.  usage: tools/hex_tile_ab.py [boardsize]   (the untiled kernels are reached by handing the dispatcher a board that is not
16-byte aligned: a view one byte into a larger buffer)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boardlaw_amd import _native
from boardlaw_amd.hex import Hex
S = int(sys.argv[1]) if len(sys.argv) > 1 else 11
L = _native.lib()
gen = torch.Generator(device='cuda'); gen.manual_seed(7)
def timed(fn, n):
    for _ in range(3): _native.check(fn())
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
for B in (1024, 4096, 16384, 65536, 262144, 1 << 20):
    worlds = Hex.initial(B, S)
    for _ in range(S * S // 3):
        valid = worlds.valid
        worlds, _ = worlds.step((torch.rand(valid.shape, device='cuda', generator=gen) * valid).argmax(-1), check=False)
    valid = worlds.valid
    actions = (torch.rand(valid.shape, device='cuda', generator=gen) * valid).argmax(-1).int().contiguous()
    A = S * S
    raw = torch.zeros(B * A + 32, dtype=torch.uint8, device='cuda')
    al, un = raw[:B * A].view(B, S, S), raw[1:B * A + 1].view(B, S, S)
    raw2 = torch.zeros(B * A + 32, dtype=torch.uint8, device='cuda')
    al2, un2 = raw2[:B * A].view(B, S, S), raw2[1:B * A + 1].view(B, S, S)
    seats = worlds.seats.int().contiguous(); seats_out = torch.empty_like(seats)
    rewards = torch.empty((B, 2), dtype=torch.float, device='cuda'); terminal = torch.empty((B,), dtype=torch.bool, device='cuda')
    obs = torch.empty((B, S, S, 2), dtype=torch.float, device='cuda'); vm = torch.empty((B, A), dtype=torch.bool, device='cuda')
    st = lambda: _native.stream(raw.device)          # evaluated inside the capture: the side stream
    n = 256 if B <= 65536 else 32
    row = []
    for name, nbytes, mk in (
        ('step', 2 * A + 16, lambda bd, bo: (lambda: L.bl_hex_step(bd.data_ptr(), seats.data_ptr(), actions.data_ptr(), rewards.data_ptr(), B, S, st()))),
        ('world_step', 2 * A + 25, lambda bd, bo: (lambda: L.bl_hex_world_step(bd.data_ptr(), seats.data_ptr(), actions.data_ptr(), 0, bo.data_ptr(), seats_out.data_ptr(), rewards.data_ptr(), terminal.data_ptr(), B, S, st()))),
        ('observe_valid', 10 * A + 4, lambda bd, bo: (lambda: L.bl_hex_observe_valid(bd.data_ptr(), seats.data_ptr(), obs.data_ptr(), vm.data_ptr(), B, S, st())))):
        res = {}
        for kind, bd, bo in (('lanes', un, un2), ('tiled', al, al2)):
            bd.copy_(worlds.board)
            res[kind] = timed(mk(bd, bo), n)
        row.append(f'{name}: lanes {res["lanes"]:8.2f} us ({B * nbytes / res["lanes"] / 1e3:7.1f} GB/s)  tiled {res["tiled"]:8.2f} us ({B * nbytes / res["tiled"] / 1e3:7.1f} GB/s)')
    print(f'{S}x{S} B={B:8d}  ' + ' | '.join(row), flush=True)
