import sys, numpy as np, torch
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
from test_network_golden import reference_network, f16bits, plan_decisions
from test_reference_fixtures import _gold, _ulp16
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
g = _gold('search_9x9_w512.npz')
S, B, T, width, depth, n_moves, seed = (int(x) for x in g['meta'])
net = reference_network(g, 'cuda')
fv = lambda bits: torch.from_numpy(bits.view(np.int16)).view(torch.half).float().numpy()
for fused in (True, False):
    inf = networks.Inference(net, fused=fused)
    created = g['m0_parents'][:, 1:] != -1
    dvs=[]; dls=[]; dabs=[]
    for sim in range(1, T):
        rows = np.nonzero(created[:, sim - 1])[0]
        if len(rows) == 0: continue
        world = Hex(board=torch.from_numpy(np.ascontiguousarray(g['m0_boards'][:, sim])).cuda(), seats=torch.from_numpy(np.ascontiguousarray(g['m0_seats'][:, sim])).cuda().int())
        logits, v = plan_decisions(inf, world)
        want_l, want_v = g['m0_tree_logits'][rows, sim], g['m0_tree_v'][rows, sim]
        got_l, got_v = f16bits(logits)[rows], f16bits(v)[rows]
        fin = want_l != 0xfc00
        dls.append(_ulp16(got_l[fin], want_l[fin]).ravel()); dvs.append(_ulp16(got_v, want_v).ravel()); dabs.append(np.abs(fv(got_v)-fv(want_v)).ravel())
    dl=np.concatenate(dls); dv=np.concatenate(dvs); da=np.concatenate(dabs)
    print('fused' if fused else 'torch-gemms', 'logits ulp: <=3', (dl<=3).mean(), 'max', dl.max(), '| v ulp <=4', (dv<=4).mean(), 'max', dv.max(), '| v abs: max', da.max(), 'q99', np.quantile(da,.99), 'q999', np.quantile(da,.999), 'mean |v|', np.abs(fv(g['m0_tree_v'][:,1:])).mean())
