"""The network and the root noise against the reference (8c fixture 3).

tests/golden/search_*.npz carry the reference FCModel's state_dict ('net_state::*'), what it was shown
(`net0_obs/valid/seats`, `net_board/net_seats`) and what it returned (`net0_logits/net0_v` in f32 for
MCTS.initialize, `net_logits_f32`, and `net_logits/net_v` after `.half()`), plus the Dirichlet draw of every search --
all produced by running the reference itself (tests/golden/make_golden.py).  On the CPU the restated modules must
reproduce them bit for bit (same torch, same operation order); the GPU cases state their tolerances."""
import numpy as np
import pytest
import torch

from test_oracle import SEARCHES, gold


def f32bits(t):
    return t.detach().contiguous().cpu().float().numpy().view(np.uint32)


def f16bits(t):
    return t.detach().contiguous().cpu().half().view(torch.int16).numpy().view(np.uint16)


def reference_network(g, device='cpu'):
    """boardlaw_amd.networks.FCModel carrying the reference network's parameters (strict key match)."""
    from boardlaw_amd import networks, heads
    S, B, T, width, depth, n_moves, seed = (int(x) for x in g['meta'])
    net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(S * S), width=width, depth=depth)
    sd = {k[len('net_state::'):]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith('net_state::')}
    assert set(sd) == set(net.state_dict()), 'parameter names differ from the reference (networks.py:10-40)'
    net.load_state_dict(sd, strict=True)
    return net.to(device)


class World:
    def __init__(self, obs, valid, seats):
        self.obs, self.valid, self.seats = obs, valid, seats


def plan_decisions(inf, world):
    """(logits, v) in f16 as MCTS.simulate stores them under the inference plan `inf`: the plan's pre-head outputs (`inf.raw`: the
    library GEMMs with fused=False, bl_mlp_forward_f16 / bl_mlp_layers_f16 with fused=True) through the heads in torch's own ops
    under autocast -- which bl_sim_finish reproduces bit for bit (tests/test_gpu_parity.py::test_finish_heads_match_torch).
    NOT `inf(world)`: calling the plan object runs the wrapped module unchanged (fp32)."""
    from boardlaw_amd import heads
    with torch.no_grad(), torch.autocast('cuda'):
        p, v = inf.raw(world)
        assert p.dtype == torch.half and v.dtype == torch.half
        logits = torch.nn.functional.log_softmax(p.masked_fill(~world.valid, -np.inf), -1).half()
        return logits, heads.scatter_values(torch.tanh(v), world.seats).half()


@pytest.mark.parametrize('name', SEARCHES)
def test_fcmodel_reproduces_reference_outputs(oracle, name):
    """networks.py:37-40, heads.py:47-52,101-104,128-142 in f32 on the CPU: identical bits."""
    g = gold(name)
    net = reference_network(g)
    S, B, T, width, depth, n_moves, seed = (int(x) for x in g['meta'])
    for move in range(n_moves):
        p = f'm{move}_'
        w = World(torch.from_numpy(g[p + 'net0_obs'].astype(np.float32)), torch.from_numpy(g[p + 'net0_valid']).bool(),
                  torch.from_numpy(g[p + 'net0_seats']))
        with torch.no_grad():
            d = net(w)
        assert np.array_equal(f32bits(d.logits), g[p + 'net0_logits'].view(np.uint32)), (name, move)
        assert np.array_equal(f32bits(d.v), g[p + 'net0_v'].view(np.uint32)), (name, move)
        # the leaf evaluations of the first simulations (CPU path: f32, then `.half()`, mcts/__init__.py:131-136)
        for i in range(g[p + 'net_logits_f32'].shape[0]):
            board, seats = np.ascontiguousarray(g[p + 'net_board'][i]), np.ascontiguousarray(g[p + 'net_seats'][i]).astype(np.int32)
            obs = oracle.hex_observe(board, seats)
            lw = World(torch.from_numpy(obs), torch.from_numpy((obs == 0).all(-1).reshape(B, -1)), torch.from_numpy(seats))
            with torch.no_grad():
                ld = net(lw)
            assert np.array_equal(f32bits(ld.logits), g[p + 'net_logits_f32'][i].view(np.uint32)), (name, move, i)
            assert np.array_equal(f32bits(ld.v), g[p + 'net_v_f32'][i].view(np.uint32)), (name, move, i)
            assert np.array_equal(f16bits(ld.logits), g[p + 'net_logits'][i]) and np.array_equal(f16bits(ld.v), g[p + 'net_v'][i])


class FixedDraw:
    def __init__(self, draw):
        self.draw = draw

    def dirichlet(self, alpha, shape):
        assert tuple(shape) == tuple(self.draw.shape[:-1]) and alpha.shape[-1] == self.draw.shape[-1]
        return self.draw.clone()


@pytest.mark.parametrize('name', SEARCHES)
def test_dirichlet_noise_reproduces_reference_root(name):
    """MCTS.initialize (mcts/__init__.py:72-80): dirichlet_noise(network logits, valid, eps) with the reference's own draw,
    stored as f16 -> the root row of the reference's tree, bit for bit (CPU)."""
    from boardlaw_amd.mcts import dirichlet_noise
    g = gold(name)
    n_moves = int(g['meta'][5])
    for move in range(n_moves):
        p = f'm{move}_'
        logits = torch.from_numpy(g[p + 'net0_logits'])
        valid = torch.from_numpy(g[p + 'net0_valid']).bool()
        noised = dirichlet_noise(logits, valid, eps=.25, alpha_scale=10, rng=FixedDraw(torch.from_numpy(g[p + 'dirichlet'])))
        assert np.array_equal(f16bits(noised), g[p + 'tree_logits'][:, 0]), (name, move)
        assert np.array_equal(f16bits(torch.from_numpy(g[p + 'net0_v'])), g[p + 'tree_v'][:, 0]), (name, move)


# ------------------------------------------------------------------------------------------------------------ GPU
def ulp16(a_bits, b_bits):
    """Distance in f16 representation steps between two uint16 arrays of finite same-sign-or-small values."""
    def key(x):
        x = x.astype(np.int32)
        return np.where(x & 0x8000, -(x & 0x7fff), x & 0x7fff)
    return np.abs(key(a_bits) - key(b_bits))


@pytest.mark.gpu
@pytest.mark.parametrize('name', SEARCHES)
def test_plant_root_kernel_against_reference_root(name):
    """The hot path's root evaluation -- fp32 Linears on the GPU (hipBLASLt, their own summation order) then
    bl_sim_plant_root (heads, the reference's Dirichlet draw, f16 store) -- against the reference's root row.  The device's
    GEMM order and expf/logf differ from the host's: tolerance 2 f16 ulp on finite logits (|logit| < 16), -inf pattern
    identical, values within 2 ulp."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS
    g = gold(name)
    S, B, T, width, depth, n_moves, seed = (int(x) for x in g['meta'])
    net = networks.Inference(reference_network(g, 'cuda'), fused=False)
    for move in range(1):      # the first move's root world is stored as a board (world0_board)
        p = f'm{move}_'
        world = Hex(board=torch.from_numpy(g['world0_board']).cuda(), seats=torch.from_numpy(g[p + 'net0_seats']).cuda())
        assert torch.equal(world.obs, torch.from_numpy(g[p + 'net0_obs'].astype(np.float32)).cuda())
        m = MCTS(world, n_nodes=T, rng=FixedDraw(torch.from_numpy(g[p + 'dirichlet']).cuda()))
        m.initialize(net)
        got, want = f16bits(m.decisions.logits[:, 0]), g[p + 'tree_logits'][:, 0]
        inf = want == 0xfc00
        assert np.array_equal(got == 0xfc00, inf)
        assert ulp16(got[~inf], want[~inf]).max() <= 2, ulp16(got[~inf], want[~inf]).max()
        assert ulp16(f16bits(m.decisions.v[:, 0]), g[p + 'tree_v'][:, 0]).max() <= 2
        # and the compacted row the descents read: exp_table of exactly the stored logits, kept actions in order
        from boardlaw_amd import _native
        nk = m._nk[:, 0].cpu().numpy()
        assert np.array_equal(nk, (~inf).sum(-1))
        tbl = _native.exp_table(torch.device('cuda')).cpu().numpy()
        for b in range(min(B, 8)):
            acts = np.nonzero(~inf[b])[0]
            assert np.array_equal(m._cca[b, 0, :nk[b]].cpu().numpy().view(np.uint32) & 0xffff, acts)
            assert np.array_equal(m._cca[b, 0, :nk[b]].cpu().numpy().view(np.uint32) >> 16, np.full(nk[b], 0xffff))
            assert np.array_equal(m._cpi[b, 0, :nk[b]].cpu().numpy(), tbl[got[b, acts]])


@pytest.mark.gpu
@pytest.mark.parametrize('name', SEARCHES)
def test_gpu_network_against_reference_outputs(name):
    """The reference network's parameters on the GPU against the outputs the reference recorded (f32 on its CPU path):
      * the module in fp32: |dlogit| <= 1e-4 (different GEMM summation order, device expf/logf), v within 1e-5;
      * the module under fp16 autocast, i.e. what the reference's MCTS.simulate stores on a GPU: within 3 f16 ulp of the f16
        values the reference stored (f32 on its CPU path, then `.half()`) on >= 99 % of the finite logits, none beyond 16 ulp
        -- fp16 autocast vs f32 is the reference's own GPU/CPU gap;
      * the torch-GEMM inference plan (networks.Inference(fused=False)): bit-identical to the module under autocast;
      * the fused plan (networks.Inference(fused=True)) where it applies: these fixtures' networks are 8..32 wide, below the
        MFMA kernels' 128, so the plan must fall back to the torch GEMMs and stay bit-identical.  The MFMA kernel itself is held
        against the reference's 512-wide recording in tests/test_reference_fixtures.py::test_fused_mfma_network_against_reference_outputs."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    g = gold(name)
    S, B, T, width, depth, n_moves, seed = (int(x) for x in g['meta'])
    net = reference_network(g, 'cuda')
    p = 'm0_'
    for i in range(g[p + 'net_logits_f32'].shape[0]):
        world = Hex(board=torch.from_numpy(np.ascontiguousarray(g[p + 'net_board'][i])).cuda(),
                    seats=torch.from_numpy(np.ascontiguousarray(g[p + 'net_seats'][i])).cuda().int())
        with torch.no_grad():
            d = net(world)
        want = torch.from_numpy(g[p + 'net_logits_f32'][i])
        fin = torch.isfinite(want)
        assert torch.equal(torch.isfinite(d.logits.cpu()), fin)
        assert (d.logits.cpu()[fin] - want[fin]).abs().max() <= 1e-4
        assert (d.v.cpu() - torch.from_numpy(g[p + 'net_v_f32'][i])).abs().max() <= 1e-5
        with torch.no_grad(), torch.autocast('cuda'):
            h = net(world)
        want16 = g[p + 'net_logits'][i]
        got16 = f16bits(h.logits)
        fin16 = want16 != 0xfc00
        assert np.array_equal(got16 != 0xfc00, fin16)
        d16 = ulp16(got16[fin16], want16[fin16])
        assert (d16 <= 3).mean() >= 0.99 and d16.max() <= 16, (d16.max(), (d16 <= 3).mean())
        for fused in (False, True):
            inf = networks.Inference(net, fused=fused)
            assert inf.fused_params(world.n_envs) is None, 'a network this narrow must stay on the torch GEMMs'
            pl, pv = plan_decisions(inf, world)
            assert np.array_equal(f16bits(pl), got16) and np.array_equal(f16bits(pv), f16bits(h.v)), (name, i, fused)


@pytest.mark.gpu
def test_plant_root_gamma_route_equals_dirichlet_route():
    """FastRng hands bl_sim_plant_root the UNNORMALISED Gamma(alpha) variates; TorchRng / MoveRng hand it torch's Dirichlet draw
    (the same variates divided by their sum and clamped).  plant_root renormalises over the valid actions either way
    (mcts/__init__.py:19-22), so the stored root rows may differ only by the rounding of the intermediate quotient: within 1 f16
    ulp, identical -inf pattern, identical kept-action lists."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS
    torch.manual_seed(0)
    B, S = 2048, 9
    gen = torch.Generator(device='cuda'); gen.manual_seed(5)
    world = Hex.initial(B, S)
    for _ in range(27):
        v = world.valid
        world, _ = world.step((torch.rand(v.shape, device='cuda', generator=gen) * v).argmax(-1), check=False)
    net = networks.Inference(networks.FCModel(world.obs_space, world.action_space, width=256, depth=2).cuda(), fused=True)
    gamma = torch._standard_gamma(torch.full((B, S * S), 10 / (S * S), device='cuda'))
    dirichlet = (gamma / gamma.sum(-1, keepdim=True)).clamp(torch.finfo(torch.float).tiny, 1 - torch.finfo(torch.float).eps)

    rows = []
    for draw in (gamma, dirichlet):
        m = MCTS(world, n_nodes=4, rng=FixedDraw(draw))
        m.initialize(net)
        rows.append((f16bits(m.decisions.logits[:, 0]), m._nk[:, 0].cpu().numpy(), (m._cca[:, 0].cpu().numpy().view(np.uint32) & 0xffff)))
    (la, nka, ca), (lb, nkb, cb) = rows
    inf = la == 0xfc00
    assert np.array_equal(inf, lb == 0xfc00) and np.array_equal(nka, nkb)
    assert ulp16(la[~inf], lb[~inf]).max() <= 1
    for b in range(0, B, 97):
        assert np.array_equal(ca[b, :nka[b]], cb[b, :nkb[b]])
