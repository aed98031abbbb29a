"""Random-shape fuzz of the seeded-agent routes (the class of round 5's 8x8 bug: a kernel family that was only
ever run at the bench's shape drew action 0 for every board below 8x8).

40 cases drawn from a fixed seed: board size S in 1..19 with the lane-count boundaries forced in (A = S^2 in {1, 4, 25, 36 -- the
33 <= A < 64 band --, 49, 64 = one wave exactly, 81, 121 -- the last size torch's row-sum layout is reproduced for --, 144, 169,
361}), ragged batches B in 1..5000, T in 2..256 node slots, network widths on both sides of the fused kernels' 128 / 256
thresholds, `eval` True / False, eager and captured.  Per case:
  * MCTSAgent under MoveRng == MCTSAgent under TorchRng from one seed -- every decision output, the stepped worlds, the generator's
    offset: bl_rand_block, bl_sim_plant_root_gamma and bl_categorical against the torch launches they stand for
    (boardlaw/mcts/__init__.py:13-24, 216-229);
  * the search of the move replayed through the C oracle (handed the uniforms torch drew and the leaf evaluations the GPU stored)
    is bit-identical: every tree array, visit count, value sum, board and the root distribution."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'

FORCED_S = [1, 2, 5, 6, 7, 8, 9, 11, 12, 13, 19]         # A = 1, 4, 25, 36, 49, 64, 81, 121, 144, 169, 361


def _cases(n=40, seed=20260930):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        S = FORCED_S[i] if i < len(FORCED_S) else int(rng.integers(1, 20))
        A = S * S
        # bounded work for the oracle replay: B x T <= ~60k simulations per case, smaller for the big boards
        budget = 60000 if A <= 128 else 12000
        T = int(min(256, 2 + rng.integers(0, 255) ** 2 // 255)) if i % 3 else int(rng.integers(2, 17))     # mostly small, sometimes up to 256
        B = int(max(1, min(5000, rng.integers(1, max(2, budget // T)))))
        width = int(rng.choice([16, 128, 256, 512]))
        depth = int(rng.integers(0, 3))
        cases.append((S, B, T, width, depth, bool(i % 2), bool((i // 2) % 2)))
    return cases


def _offset():
    return torch.cuda.default_generators[torch.cuda.current_device()].get_offset()


@pytest.mark.parametrize('S,B,T,width,depth,graph,evaluate', _cases())
def test_seeded_agent_routes_at_random_shapes(oracle, S, B, T, width, depth, graph, evaluate):
    from gpu_util import bits16
    from test_gpu_parity import RecordingRng, assert_search_equals, premixed, replay_through_oracle
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent, MoveRng, TorchRng, mcts
    A = S * S
    board, seats = premixed(oracle, B, S, A // 3, seed=S * 7919 + B)
    worlds0 = Hex(board=torch.from_numpy(board).to(DEV), seats=torch.from_numpy(seats).to(DEV))
    torch.manual_seed(S * 100 + T)
    net = networks.Inference(networks.FCModel(worlds0.obs_space, worlds0.action_space, width=width, depth=depth).to(DEV), fused=True)
    with torch.no_grad():
        for p_ in net.model.parameters():
            if p_.ndim == 0:
                p_.fill_(0.3)                # ReZero gains start at 0: make the evaluation depend on the position

    # (1) MoveRng == TorchRng, two consecutive moves
    runs = {}
    for name, rng in (('torch', TorchRng()), ('move', MoveRng())):
        agent = MCTSAgent(net, n_nodes=T, graph=graph, rng=rng)
        worlds = worlds0
        if graph:
            agent.play(worlds, eval=evaluate)          # capture + warm-up consume the generator differently in the two agents: reseed after
        torch.manual_seed(11)
        outs = []
        for _ in range(2):
            d, worlds, t = agent.play(worlds, eval=evaluate)
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
    valid0 = worlds0.valid
    assert bool(valid0.gather(1, a[0][0].actions[:, None]).all()), 'an illegal action was drawn'

    # (2) the move's search through the oracle
    rng = RecordingRng(TorchRng())
    torch.manual_seed(12)
    m = mcts(worlds0, net, n_nodes=T, rng=rng)
    rands = np.stack([bits16(r) for r in rng.rands]) if T > 1 else np.zeros((0, B, T), np.uint16)
    want = replay_through_oracle(oracle, board, seats, T, rands, bits16(m.decisions.logits), bits16(m.decisions.v))
    assert_search_equals(m, want)
    assert (m.stats.n[:, 0] == 2 * (T - 1)).all()
