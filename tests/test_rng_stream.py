"""The reference's RNG protocol: one `at::rand_like` (B,T) f16 per descend
(boardlaw/mcts/cpp/cuda.cu:191), torch's Dirichlet at mcts/__init__.py:16-18, Categorical.sample at :221, all from torch's
generator.  MoveRng serves the T-1 descend draws of a move from ONE launch (bl_rand_block) and must be STREAM-IDENTICAL: the
same numbers as the T-1 calls, the same generator offset afterwards, eagerly and inside a captured HIP graph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _offset():
    return torch.cuda.default_generators[torch.cuda.current_device()].get_offset()


@pytest.mark.parametrize('shape,n_calls', [((4096, 64), 63), ((1024, 256), 7), ((64, 16), 15), ((333, 64), 5), ((1, 1), 3),
                                           ((32768, 128), 2), ((40000, 64), 3)])
def test_rand_block_equals_stacked_rand_like(shape, n_calls):
    """Same seed -> the block is the n_calls rand_like tensors, bit for bit, and the generator ends at the same offset.
    (32768,128) and (40000,64) exceed 4 * torch's thread count: several Philox blocks per thread and the float4's later
    components come into play."""
    from boardlaw_amd.mcts import MoveRng
    like = torch.empty(shape, dtype=torch.half, device=DEV)
    torch.manual_seed(1234)
    torch.rand(7, device=DEV)                               # a non-zero starting offset
    want = torch.stack([torch.rand_like(like) for _ in range(n_calls)])
    end = _offset()
    torch.manual_seed(1234)
    torch.rand(7, device=DEV)
    rng = MoveRng()
    rng.start(n_calls)
    got = torch.stack([rng.rand_like(like) for _ in range(n_calls)])
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert _offset() == end
    assert float(got.float().max()) < 1.0 and float(got.float().min()) >= 0.0
    # a further draw falls back to torch's own call and stays on the stream
    nxt = rng.rand_like(like)
    torch.manual_seed(1234)
    torch.rand(7, device=DEV)
    for _ in range(n_calls):
        torch.rand_like(like)
    assert torch.equal(nxt, torch.rand_like(like))


def test_rand_block_with_a_private_generator():
    from boardlaw_amd.mcts import MoveRng
    like = torch.empty((512, 64), dtype=torch.half, device=DEV)
    g1 = torch.Generator(device=DEV); g1.manual_seed(99)
    want = torch.stack([torch.rand(like.shape, dtype=torch.half, device=DEV, generator=g1) for _ in range(9)])
    g2 = torch.Generator(device=DEV); g2.manual_seed(99)
    before = _offset()
    rng = MoveRng(generator=g2)
    rng.start(9)
    got = torch.stack([rng.rand_like(like) for _ in range(9)])
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)) and g1.get_offset() == g2.get_offset()
    assert _offset() == before, 'the default generator must not be touched'


def test_rand_block_inside_a_captured_graph():
    """Captured once, replayed three times: every replay continues the generator's stream exactly like a graph of rand_like calls
    (torch refills the graph's seed/offset tensors before each replay; the kernel reads them like torch's kernels do)."""
    from boardlaw_amd.mcts import MoveRng
    like = torch.empty((2048, 64), dtype=torch.half, device=DEV)
    n_calls = 11

    def capture(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        return g, out

    def reference():
        a = torch._sample_dirichlet(torch.full((2048, 81), 10 / 81, device=DEV))
        return a, torch.stack([torch.rand_like(like) for _ in range(n_calls)])

    rng = MoveRng()

    def mine():
        rng.start(n_calls)
        a = rng.dirichlet(torch.full((81,), 10 / 81, device=DEV), (2048,))
        return a, torch.stack([rng.rand_like(like) for _ in range(n_calls)])

    g_ref, out_ref = capture(reference)
    g_mine, out_mine = capture(mine)
    torch.manual_seed(7)
    wants = []
    for _ in range(3):
        g_ref.replay()
        wants.append([t.clone() for t in out_ref])
    end = _offset()
    torch.manual_seed(7)
    for r in range(3):
        g_mine.replay()
        assert torch.equal(out_mine[0], wants[r][0]), r
        assert torch.equal(out_mine[1].view(torch.int16), wants[r][1].view(torch.int16)), r
    assert _offset() == end
    assert not torch.equal(wants[0][1], wants[1][1])


@pytest.mark.parametrize('graph', [False, True])
@pytest.mark.parametrize('S,B,T,width,depth', [(9, 512, 64, 256, 2), (5, 64, 16, 32, 2)])
def test_seeded_agent_under_move_rng_is_the_torch_rng_agent(S, B, T, width, depth, graph):
    """MCTSAgent(rng=MoveRng()) == MCTSAgent(rng=TorchRng()) under one seed, for three consecutive self-play moves: every
    decision output, the worlds, and the generator's offset -- i.e. a seeded run consumes the generator like the reference."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent, MoveRng, TorchRng
    torch.manual_seed(3)
    worlds0 = Hex.initial(B, S)
    net = networks.Inference(networks.FCModel(worlds0.obs_space, worlds0.action_space, width=width, depth=depth).cuda(), fused=True)
    runs = {}
    for name, rng in (('torch', TorchRng()), ('move', MoveRng())):
        agent = MCTSAgent(net, n_nodes=T, graph=graph, rng=rng)
        worlds = worlds0
        if graph:
            agent.play(worlds)                 # capture + warm-up consume the generator differently in the two agents: reseed after
        torch.manual_seed(11)
        outs = []
        for _ in range(3):
            d, worlds, t = agent.play(worlds)
            outs.append((d, worlds.board.clone(), worlds.seats.clone()))
        runs[name] = (outs, _offset())
    (a, off_a), (b, off_b) = runs['torch'], runs['move']
    assert off_a == off_b
    for (da, ba, sa), (db, bb, sb) in zip(a, b):
        for k in ('logits', 'prior', 'v', 'actions', 'n_leaves', 'n_sims'):
            x, y = da[k], db[k]
            if x.dtype == torch.half:
                x, y = x.view(torch.int16), y.view(torch.int16)
            assert torch.equal(x, y), k
        assert torch.equal(ba, bb) and torch.equal(sa, sb)


@pytest.mark.parametrize('B,T', [(777, 64), (130, 512), (70, 1024)])
def test_rand_block_for_descents_writes_the_slots_a_descent_can_read(B, T):
    """slots_upto_call (what mcts() asks for on the fused path): call c's tensor carries the reference's uniforms in the slots
    t <= c -- every node that exists at descend #c+1 -- and the generator still ends where T-1 rand_like calls end.
    T = 512 / 1024: the tile is 64 x (T + 2) f16 of LDS, above the 64 KiB a kernel gets without asking (round-3 advisor finding)."""
    from boardlaw_amd.mcts import MoveRng
    like = torch.empty((B, T), dtype=torch.half, device=DEV)
    torch.manual_seed(5)
    want = torch.stack([torch.rand_like(like) for _ in range(T - 1)])
    end = _offset()
    torch.manual_seed(5)
    rng = MoveRng()
    rng.start(T - 1, slots_upto_call=True)
    got = torch.stack([rng.rand_like(like) for _ in range(T - 1)])
    assert _offset() == end
    mask = torch.arange(T, device=DEV)[None, None, :] <= torch.arange(T - 1, device=DEV)[:, None, None]      # (T-1, 1, T): t <= c
    assert torch.equal(torch.where(mask, got, torch.zeros_like(got)).view(torch.int16), torch.where(mask, want, torch.zeros_like(want)).view(torch.int16))


@pytest.mark.parametrize('S', [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
@pytest.mark.parametrize('B', [64, 1000])
def test_fused_draws_equal_torchs(S, B):
    """The two fused draws of a move (round 5) against the torch launches they replace, bit for bit, at every board size they serve
    (A = S^2 <= 127: torch's reduce then sums a row in a fixed lane layout of width Wr = the largest power of two <= A, which the
    kernels reproduce -- Wr < 64 is where the first version went wrong):
      * MoveRng.categorical_f16 (torch's exponential_ draw + bl_categorical) == MoveRng.categorical (torch's softmax / div / argmax
        on the same draw), on logits with -inf entries, equal entries and one all-but-one-masked row; same generator offset after;
      * a root planted from the gamma variates (bl_sim_plant_root_gamma: normalise, clamp, mask, renormalise, mix inside the launch)
        == a root planted from torch's finished Dirichlet sample (MoveRng.dirichlet -> bl_sim_plant_root): logits row, compacted row."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS, MoveRng
    A = S * S
    gen = torch.Generator(device=DEV); gen.manual_seed(S * 1000 + B)
    logits = (torch.randn(B, A, device=DEV, generator=gen) * 2).half()
    mask = torch.rand(B, A, device=DEV, generator=gen) < 0.4
    mask[:, 0] = False if A == 1 else mask[:, 0]
    mask[torch.arange(B, device=DEV), torch.randint(A, (B,), device=DEV, generator=gen)] = False      # at least one finite entry per row
    logits[mask] = -float('inf')
    if A > 2:
        logits[1] = -float('inf'); logits[1, A - 1] = 0.5                                              # one live action
        logits[2, :] = 0.25                                                                             # ties
    mask = torch.isinf(logits)
    rng = MoveRng()
    torch.manual_seed(77); a = rng.categorical_f16(logits); off_a = _offset()
    torch.manual_seed(77); b = rng.categorical(logits.float()); off_b = _offset()
    assert torch.equal(a, b) and off_a == off_b
    assert not mask.gather(1, a[:, None]).any()

    torch.manual_seed(S)
    world = Hex.initial(B, S)
    for _ in range(A // 3):
        world, _ = world.step((torch.rand(world.valid.shape, device=DEV) * world.valid).argmax(-1), check=False)
    net = networks.Inference(networks.FCModel(world.obs_space, world.action_space, width=128, depth=1).cuda(), fused=True)
    rows = {}
    for route in ('gamma', 'dirichlet'):
        r = MoveRng()
        if route == 'dirichlet':
            r.FUSED_MAX_ACTIONS = 0                        # MCTS.initialize then takes torch's finished Dirichlet sample
        m = MCTS(world, n_nodes=4, rng=r)
        torch.manual_seed(5)
        m.initialize(net)
        nk = m._nk[:, 0].long()
        live = torch.arange(A, device=DEV)[None, :] < nk[:, None]            # the compacted row's kept entries (the rest is never read)
        rows[route] = (m.decisions.logits[:, 0].clone(), (torch.where(live, m._cpi[:, 0], 0.), torch.where(live, m._cca[:, 0], 0), nk), _offset())
    g, d = rows['gamma'], rows['dirichlet']
    assert torch.equal(g[0].view(torch.int16), d[0].view(torch.int16)) and g[2] == d[2]
    assert all(torch.equal(x, y) for x, y in zip(g[1], d[1]))
