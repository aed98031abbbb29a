"""GPU parity: the HIP path (through the C ABI of libboardlaw_amd.so) against the CPU oracle and the reference's
golden vectors.  Bit-exact: every integer, byte, index and binary16 output must be identical (tolerance = 0).
Run on an MI355X with `pytest -m gpu`."""
import ctypes
import os

import numpy as np
import pytest
import torch

from gpu_util import HashNetwork, ReplayNetwork, ReplayRng, bits16, hash_network_np, t16, to_np
from oracle_lib import OracleSearch, f16_bits
from test_oracle import OPS, SEARCHES, gold, op_cases, random_tree, bits

pytestmark = pytest.mark.gpu
DEV = 'cuda'
SEED = int(os.environ.get('BL_TEST_SEED', '0'))      # tools/fuzz_parity.sh: the full-size oracle comparisons on other positions and draws


@pytest.fixture(scope='module', autouse=True)
def _native_loaded():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from boardlaw_amd import _native
    _native.lib()          # raises if the HIP extension is missing: there is no fallback to test


def dev(x):
    """numpy fixture array -> device tensor with the dtype the C ABI expects."""
    if x.dtype == np.uint16:
        return t16(x, DEV)
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t


def tree_tensors(g, c, tile=1):
    d = {k: np.ascontiguousarray(g[f'{c}_{k}']) for k in ('logits', 'w', 'n', 'c_puct', 'seats', 'terminal', 'children')}
    d = {k: np.concatenate([v] * tile, 0) for k, v in d.items()}
    t = {k: dev(v) for k, v in d.items()}
    t['terminal'] = t['terminal'].bool()
    return t


# ------------------------------------------------------------------------------------------------ Hex kernels
@pytest.mark.parametrize('S', [3, 4, 5, 7, 9, 11, 13])
def test_hex_step_observe_golden(S):
    from boardlaw_amd.hex import cuda as hcuda, Hex
    g = gold('hex_playouts.npz')
    boards, seats, actions = g[f'S{S}_board'], g[f'S{S}_seats'], g[f'S{S}_actions']
    n = boards.shape[0] * boards.shape[1]
    fb = torch.from_numpy(boards.reshape(n, S, S)).to(DEV)
    fs = torch.from_numpy(seats.reshape(n).astype(np.int32)).to(DEV)
    fa = torch.from_numpy(actions.reshape(n).astype(np.int32)).to(DEV)
    obs = hcuda.observe(fb, fs)
    assert np.array_equal(to_np(obs).astype(np.uint8), g[f'S{S}_obs'].reshape(n, S, S, 2))
    raw = fb.clone()
    rewards = hcuda.step(raw, fs, fa)
    assert np.array_equal(to_np(raw), g[f'S{S}_raw_board'].reshape(n, S, S))
    assert np.array_equal(to_np(rewards), g[f'S{S}_rewards'].reshape(n, 2))
    # the world object on top (hex/__init__.py:161-195)
    world = Hex(board=fb, seats=fs)
    assert np.array_equal(to_np(world.valid), g[f'S{S}_valid'].reshape(n, S * S))
    new, trans = world.step(fa.long())
    assert np.array_equal(to_np(new.board), g[f'S{S}_new_board'].reshape(n, S, S))
    assert np.array_equal(to_np(new.seats), g[f'S{S}_new_seats'].reshape(n))
    assert np.array_equal(to_np(trans.terminal), g[f'S{S}_terminal'].reshape(n))


@pytest.mark.parametrize('S,B', [(2, 7), (9, 4096), (13, 4096), (19, 1000), (32, 64)])
def test_hex_random_games_vs_oracle(oracle, S, B):
    """Random legal play from empty boards to the end of several games, both seats, ragged batch sizes."""
    from boardlaw_amd.hex import cuda as hcuda
    rng = np.random.default_rng(S)
    board = np.zeros((B, S, S), np.uint8); seats = np.zeros(B, np.int32)
    for step in range(min(3 * S * S, 400)):
        obs = oracle.hex_observe(board, seats)
        valid = (obs == 0).all(-1).reshape(B, -1)
        r = rng.random(valid.shape) * valid
        actions = r.argmax(-1).astype(np.int32)
        tb, ts, ta = torch.from_numpy(board).to(DEV), torch.from_numpy(seats).to(DEV), torch.from_numpy(actions).to(DEV)
        assert np.array_equal(to_np(hcuda.observe(tb, ts)), obs), step
        rewards = hcuda.step(tb, ts, ta)
        raw = board.copy(); want = oracle.hex_step(raw, seats, actions)
        assert np.array_equal(to_np(tb), raw), step
        assert np.array_equal(to_np(rewards), want), step
        board, seats, _, _ = oracle.hex_world_step(board, seats, actions)


@pytest.mark.parametrize('S,B', [(1, 5), (2, 70), (3, 1000), (5, 64), (9, 4096), (11, 4097), (12, 100), (13, 333), (15, 77), (16, 129)])
def test_hex_tiled_kernels_vs_oracle(oracle, S, B):
    """The board kernels as HBM streams (bl_hex_step_tiled / bl_hex_world_step_tiled / bl_hex_observe_valid_tiled, round 5: 64
    consecutive envs per workgroup through LDS, the flood as a bit-board fill, 16-byte loads and stores) against the C oracle along
    random games to the end, both seats, ragged batch sizes (the last workgroup's run of boards is not a multiple of 16 bytes),
    1x1 to 16x16 -- called directly, whatever sizes the dispatchers send to them; plus their refusals."""
    from boardlaw_amd import _native
    L = _native.lib()
    rng = np.random.default_rng(S + B)
    board = np.zeros((B, S, S), np.uint8); seats = np.zeros(B, np.int32)
    st = _native.stream(torch.device(DEV))
    for step in range(min(3 * S * S, 260)):
        obs = oracle.hex_observe(board, seats)
        valid = (obs == 0).all(-1).reshape(B, -1)
        actions = (rng.random(valid.shape) * valid).argmax(-1).astype(np.int32)
        tb, ts, ta = torch.from_numpy(board).to(DEV), torch.from_numpy(seats).to(DEV), torch.from_numpy(actions).to(DEV)
        o = torch.full((B, S, S, 2), 7., device=DEV); vm = torch.full((B, S * S), 7, dtype=torch.uint8, device=DEV)
        assert L.bl_hex_observe_valid_tiled(tb.data_ptr(), ts.data_ptr(), o.data_ptr(), vm.data_ptr(), B, S, st) == 0
        assert np.array_equal(to_np(o), obs) and np.array_equal(to_np(vm).astype(bool), valid), step
        o2 = torch.full((B, S, S, 2), 7., device=DEV)
        assert L.bl_hex_observe_valid_tiled(tb.data_ptr(), ts.data_ptr(), o2.data_ptr(), None, B, S, st) == 0 and torch.equal(o, o2)
        raw = board.copy(); want = oracle.hex_step(raw, seats, actions)
        nb, ns, nr, nt = oracle.hex_world_step(board, seats, actions)
        ob, os_ = torch.full_like(tb, 9), torch.full_like(ts, 9)
        r2 = torch.full((B, 2), 9., device=DEV); term = torch.full((B,), 9, dtype=torch.uint8, device=DEV)
        a64 = ta.long()
        assert L.bl_hex_world_step_tiled(tb.data_ptr(), ts.data_ptr(), (a64 if step % 2 else ta).data_ptr(), step % 2, ob.data_ptr(), os_.data_ptr(),
                                         r2.data_ptr(), term.data_ptr(), B, S, st) == 0
        assert np.array_equal(to_np(ob), nb) and np.array_equal(to_np(os_), ns) and np.array_equal(to_np(r2), nr) and np.array_equal(to_np(term).astype(bool), nt), step
        r = torch.full((B, 2), 9., device=DEV)
        assert L.bl_hex_step_tiled(tb.data_ptr(), ts.data_ptr(), ta.data_ptr(), r.data_ptr(), B, S, st) == 0
        assert np.array_equal(to_np(tb), raw) and np.array_equal(to_np(r), want), step
        board, seats = nb, ns
    big = torch.zeros((4, 17, 17), dtype=torch.uint8, device=DEV); z = torch.zeros(4 * 17 * 17 * 2, dtype=torch.int32, device=DEV)
    assert L.bl_hex_step_tiled(big.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 4, 17, st) == _native.BL_ETOOBIG
    assert L.bl_hex_observe_valid_tiled(big.data_ptr(), z.data_ptr(), z.data_ptr(), None, 4, 17, st) == _native.BL_ETOOBIG
    odd = torch.zeros(4 * 81 + 16, dtype=torch.uint8, device=DEV)[1:]
    assert L.bl_hex_step_tiled(odd.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 4, 9, st) == _native.BL_EINVAL


def test_hex_observe_tiled_through_the_dispatcher():
    """bl_hex_observe_valid / bl_hex_observe send batches of 2^17 envs and more to the tiled kernel: the same planes and mask as the
    lanes-per-env kernel computes for the same boards in two halves."""
    from boardlaw_amd.hex import cuda as hcuda, Hex
    torch.manual_seed(1)
    B, S = (1 << 17) + 77, 7
    w = Hex.initial(B, S, device=DEV)
    for _ in range(12):
        w, _ = w.step((torch.rand(w.valid.shape, device=DEV) * w.valid).argmax(-1), check=False)
    obs = hcuda.observe(w.board, w.seats)
    h = B // 2
    assert torch.equal(obs[:h], hcuda.observe(w.board[:h].contiguous(), w.seats[:h].contiguous()))
    assert torch.equal(obs[h:], hcuda.observe(w.board[h:].clone(), w.seats[h:].clone()))
    assert torch.equal(w.valid, (obs == 0).all(-1).reshape(B, S * S))


def test_hex_reference_known_answers():
    """boardlaw/hex/tests.py:58-91 and hex/__init__.py:274-297 through the product's Hex world."""
    from boardlaw_amd.hex import Hex, cuda as hcuda
    def tok(*moves):
        b = torch.zeros((1, 3, 3), dtype=torch.uint8, device=DEV)
        for ij, v in moves:
            b[0, ij // 3, ij % 3] = v
        return b
    def apply(seat, action, board):
        return hcuda.step(board, torch.tensor([seat], dtype=torch.int32, device=DEV), torch.tensor([action], dtype=torch.int32, device=DEV))
    b = tok(); r = apply(0, 2, b); assert torch.equal(b, tok((2, 3))) and (r == 0).all()          # black TR -> TOP
    b = tok(); r = apply(1, 2, b); assert torch.equal(b, tok((6, 5)))                              # white TR mirrored -> BL LEFT
    b = tok((3, 1), (4, 1)); apply(0, 7, b); assert torch.equal(b, tok((3, 4), (4, 4), (7, 4)))    # bottom flooding
    b = tok((1, 2), (4, 2)); apply(1, 1, b); assert torch.equal(b, tok((1, 5), (4, 5), (3, 5)))    # left flooding
    assert apply(0, 4, tok((1, 3), (7, 4))).tolist() == [[1., -1.]]
    assert apply(1, 4, tok((3, 5), (5, 6))).tolist() == [[-1., 1.]]
    w = Hex.initial(1, 3, device=DEV)
    for a in [5, 5, 6, 1]:
        w, _ = w.step(torch.tensor([a], device=DEV))
    assert w.board[0].tolist() == [[0, 0, 0], [5, 0, 1], [4, 2, 0]]
    w = Hex(board=torch.tensor([[[0, 6, 6], [1, 1, 1], [0, 2, 0]]], dtype=torch.uint8, device=DEV), seats=torch.zeros(1, dtype=torch.int, device=DEV))
    w, _ = w.step(torch.tensor([6], device=DEV))
    assert w.board[0].tolist() == [[0, 6, 6], [4, 4, 4], [4, 2, 0]]


# ------------------------------------------------------------------------------------------------ search kernels
@pytest.mark.parametrize('name', OPS)
@pytest.mark.parametrize('tile', [1, 37, 70, 260])
def test_descend_root_backup_golden(name, tile):
    """Reference trees from live searches; tiling the batch moves the launch through every lanes-per-env group width
    (64, 32, 16, 8) without changing the expected per-env answers (the q range is batch-global and tile-invariant)."""
    from boardlaw_amd.mcts import cuda as mcuda
    from boardlaw_amd import _native
    g = gold(name)
    if tile > 70 and g[op_cases(g, 'descend')[0] + '_logits'].shape[2] > 100:
        pytest.skip('memory-heavy duplicate of the smaller boards')
    for c in op_cases(g, 'descend'):
        t = tree_tensors(g, c, tile)
        m = mcuda.mcts(**t)
        rands = dev(np.concatenate([g[f'{c}_rands']] * tile, 0))
        d = mcuda.descend(m, rands)
        assert np.array_equal(to_np(d.parents), np.tile(g[f'{c}_parents'], tile)), c
        assert np.array_equal(to_np(d.actions), np.tile(g[f'{c}_actions'], tile)), c
        assert np.array_equal(bits16(mcuda.root(m)), np.tile(g[f'{c}_root_probs'], (tile, 1))), c
        mm = _native.qrange_decode(mcuda._qrange(m, torch.device(DEV, torch.cuda.current_device())))
        assert np.array_equal(mm.numpy(), g[f'{c}_qminmax']), c
    for c in op_cases(g, 'backup'):
        t = {k: dev(np.concatenate([g[f'{c}_{k}']] * tile, 0)) for k in ('v', 'w', 'n', 'rewards', 'parents', 'terminal', 'leaves')}
        t['terminal'] = t['terminal'].bool()
        leaves = t.pop('leaves')
        mcuda.backup(mcuda.Backup(**t), leaves)
        assert np.array_equal(bits16(t['w']), np.concatenate([g[f'{c}_w_after']] * tile, 0)), c
        assert np.array_equal(to_np(t['n']), np.concatenate([g[f'{c}_n_after']] * tile, 0)), c


@pytest.mark.parametrize('B,T,A,filled', [(5, 6, 4, 5), (16, 12, 9, 12), (33, 40, 25, 33), (8, 64, 81, 64), (6, 64, 169, 50),
                                          (3, 30, 361, 25), (2, 20, 1000, 12)])
def test_random_trees_vs_oracle(oracle, B, T, A, filled):
    """Structurally valid random trees incl. -inf logits, terminal nodes, ragged B, every supported lanes-per-action K."""
    from boardlaw_amd.mcts import cuda as mcuda
    rng = np.random.default_rng(A)
    d = random_tree(rng, B, T, A, filled)
    nd = {k: bits(v) for k, v in d.items()}
    m = mcuda.mcts(**{k: v.to(DEV) for k, v in d.items()})
    rands = torch.from_numpy(rng.random((B, T)).astype(np.float16).view(np.int16)).view(torch.half)
    got = mcuda.descend(m, rands.to(DEV))
    parents, actions = oracle.descend(**nd, rands=bits(rands))
    assert np.array_equal(to_np(got.parents), parents) and np.array_equal(to_np(got.actions), actions)
    assert np.array_equal(bits16(mcuda.root(m)), oracle.root(**nd))


def test_struct_checks_match_reference():
    """TensorProxy behaviour (boardlaw/cpp/common.h:33-37): dtype -> TypeError 'expected Half got Float', layout -> RuntimeError."""
    from boardlaw_amd.mcts import cuda as mcuda
    z = lambda *s, dtype: torch.zeros(s, dtype=dtype, device=DEV)
    with pytest.raises(TypeError, match='expected Half got Float'):
        mcuda.Backup(v=z(1, 2, 1, dtype=torch.float), w=z(1, 2, 1, dtype=torch.half), n=z(1, 2, dtype=torch.short),
                     rewards=z(1, 2, 1, dtype=torch.half), parents=z(1, 2, dtype=torch.short), terminal=z(1, 2, dtype=torch.bool))
    with pytest.raises(RuntimeError, match='contiguous'):
        mcuda.MCTS(z(2, 3, 4, dtype=torch.half).transpose(0, 1), z(2, 3, 1, dtype=torch.half), z(2, 3, dtype=torch.short),
                   z(2, dtype=torch.half), z(2, 3, dtype=torch.short), z(2, 3, dtype=torch.bool), z(2, 3, 4, dtype=torch.short))
    with pytest.raises(AssertionError, match='c_puct'):
        mcuda.mcts(z(2, 3, 4, dtype=torch.half), z(2, 3, 1, dtype=torch.half), z(2, 3, dtype=torch.short),
                   z(2, dtype=torch.half), z(2, 3, dtype=torch.short), z(2, 3, dtype=torch.bool), z(2, 3, 4, dtype=torch.short))


# ------------------------------------------------------------------------------------------------ whole searches
def compare_search(m, g, p, move):
    for mine, theirs in [(m.tree.children, 'children'), (m.tree.parents, 'parents'), (m.tree.relation, 'relation'),
                         (m.stats.n, 'n'), (m.stats.w, 'w'), (m.transitions.rewards, 'rewards'),
                         (m.transitions.terminal, 'terminal'), (m.worlds.board, 'boards'), (m.worlds.seats, 'seats'),
                         (m.decisions.v, 'tree_v'), (m.decisions.logits, 'tree_logits')]:
        assert np.array_equal(to_np(mine), g[p + theirs]), (move, theirs)
    assert np.array_equal(bits16(m.root_probs()), g[p + 'root_probs']), move
    assert np.array_equal(to_np(m.n_leaves()), g[p + 'dec_n_leaves']), move


@pytest.mark.parametrize('name', SEARCHES)
@pytest.mark.parametrize('fused', [True, False])
def test_whole_search_replay(name, fused):
    """Every recorded reference search (same uniforms, same network outputs) replayed through the product's MCTS on
    the GPU, both execution paths: identical trees, visit counts, values, boards and root distributions."""
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS
    g = gold(name)
    S, B, T, _, _, n_moves, _ = g['meta']
    world = Hex(board=dev(g['world0_board']), seats=dev(g['world0_seats']))
    for move in range(n_moves):
        p = f'm{move}_'
        net = ReplayNetwork(g[p + 'net_logits'], g[p + 'net_v'], DEV)
        m = MCTS(world, n_nodes=int(T), fused=fused, rng=ReplayRng(g[p + 'rands'], DEV))
        m.plant_root(t16(g[p + 'tree_logits'][:, 0], DEV), t16(g[p + 'tree_v'][:, 0], DEV))
        for i in range(T - 1):
            m.simulate(net)
            board, seats, _, _ = net.seen[-1]
            assert np.array_equal(board, g[p + 'net_board'][i]) and np.array_equal(seats, g[p + 'net_seats'][i]), (move, i)
        compare_search(m, g, p, move)
        # the root read-out of MCTSAgent (mcts/__init__.py:142-149,221): log of the f16 probabilities
        r = m.root()
        assert np.array_equal(bits16(r.logits), g[p + 'dec_logits']), move
        assert np.array_equal(bits16(r.prior), g[p + 'dec_prior']) and np.array_equal(bits16(r.v), g[p + 'dec_v']), move
        world, trans = world.step(dev(g[p + 'dec_actions']))
        assert np.array_equal(to_np(trans.rewards), g[p + 'step_rewards']) and np.array_equal(to_np(trans.terminal), g[p + 'step_terminal'])


def oracle_search(oracle, board, seats, T, rands, stats=None, c_puct=None):
    s = OracleSearch(oracle, board, seats, T)
    if c_puct is not None:
        s.c_puct = np.ascontiguousarray(c_puct.astype(np.float16).view(np.uint16))
    obs = oracle.hex_observe(board, seats)
    valid = (obs == 0).all(-1).reshape(board.shape[0], -1)
    l, v = hash_network_np(board, seats, valid)
    s.initialize(l, v)
    for i in range(T - 1):
        parents, actions = s.descend(rands[i], stats)
        leaves, nb, ns = s.expand(parents, actions)
        o = oracle.hex_observe(nb, ns)
        l, v = hash_network_np(nb, ns, (o == 0).all(-1).reshape(nb.shape[0], -1))
        s.finish(leaves, l, v)
    return s


def premixed(oracle, B, S, moves, seed):
    rng = np.random.default_rng(seed)
    board = np.zeros((B, S, S), np.uint8); seats = np.zeros(B, np.int32)
    for _ in range(moves):
        obs = oracle.hex_observe(board, seats)
        valid = (obs == 0).all(-1).reshape(B, -1)
        actions = (rng.random(valid.shape) * valid).argmax(-1).astype(np.int32)
        board, seats, _, _ = oracle.hex_world_step(board, seats, actions)
    return board, seats


@pytest.mark.parametrize('S,B,T', [(9, 4096, 64), (5, 64, 16), (13, 1024, 48), (3, 16384, 8), (11, 333, 32), (13, 96, 256), (2, 5, 3), (9, 1, 64), (19, 24, 40), (32, 3, 12),
                                   (4, 2048, 64), (6, 2048, 64), (7, 2048, 64), (8, 2048, 64), (10, 2048, 64),     # config 5's sizes between 3 and 11 at its batch size
                                   (9, 32768, 64)])        # the reference's own operating point: main.run's n_envs = 32*1024 (boardlaw/main.py:147)
def test_full_size_search_vs_oracle(oracle, S, B, T):
    """BASELINE config 2 at its full size (9x9, 4096 envs, 64 nodes) and neighbours, and the reference's default actor shape
    (9x9, 32768 envs, 64 nodes): a whole search on the GPU against the oracle-driven search on the host, with a
    device-independent integer network.  Everything must be identical."""
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS
    board, seats = premixed(oracle, B, S, (S * S) // 3, seed=S * 1000 + T + SEED)
    rng = np.random.default_rng(1 + SEED)
    rands = rng.random((T - 1, B, T)).astype(np.float16).view(np.uint16)
    want = oracle_search(oracle, board, seats, T, rands)

    world = Hex(board=torch.from_numpy(board).to(DEV), seats=torch.from_numpy(seats).to(DEV))
    net = HashNetwork(DEV)
    m = MCTS(world, n_nodes=T, rng=ReplayRng(rands, DEV), noise_eps=0.)
    d = net(world)
    m.plant_root(d.logits, d.v)
    for _ in range(T - 1):
        m.simulate(net)
    for mine, theirs in [(m.tree.children, want.children), (m.tree.parents, want.parents), (m.tree.relation, want.relation),
                         (m.stats.n, want.n), (m.stats.w, want.w), (m.transitions.rewards, want.rewards),
                         (m.transitions.terminal, want.terminal), (m.worlds.board, want.boards), (m.worlds.seats, want.seats),
                         (m.decisions.v, want.v), (m.decisions.logits, want.logits)]:
        assert np.array_equal(to_np(mine), theirs)
    assert np.array_equal(bits16(m.root_probs()), want.root_probs())
    # size-independent invariants of a finished search (every visit adds S=2 to n along its path; root saw them all)
    n = to_np(m.stats.n)
    assert (n[:, 0] == 2 * (T - 1)).all()
    assert (to_np(m.tree.parents)[:, 1:] < np.arange(1, T)[None]).all()


@pytest.mark.parametrize('S,B,T', [(9, 512, 64), (5, 200, 24)])
def test_search_with_per_env_c_puct(oracle, S, B, T):
    """c_puct is a per-env f16 array in the reference's struct (mcts/cpp/common.h:25-33): from 1/256 (value-driven: long
    Newton runs, deep narrow trees) to 8 (prior-driven).  Whole search against the oracle, everything identical."""
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS
    board, seats = premixed(oracle, B, S, (S * S) // 3, seed=77 + S)
    rng = np.random.default_rng(5)
    rands = rng.random((T - 1, B, T)).astype(np.float16).view(np.uint16)
    c = (2.0 ** rng.integers(-8, 4, B)).astype(np.float32) * (1 + rng.random(B).astype(np.float32) / 2)
    want = oracle_search(oracle, board, seats, T, rands, c_puct=c)

    world = Hex(board=torch.from_numpy(board).to(DEV), seats=torch.from_numpy(seats).to(DEV))
    net = HashNetwork(DEV)
    m = MCTS(world, n_nodes=T, rng=ReplayRng(rands, DEV), noise_eps=0.)
    m.c_puct.copy_(torch.from_numpy(c).to(DEV).half())
    d = net(world)
    m.plant_root(d.logits, d.v)
    for _ in range(T - 1):
        m.simulate(net)
    for mine, theirs in [(m.tree.children, want.children), (m.tree.parents, want.parents), (m.stats.n, want.n),
                         (m.stats.w, want.w), (m.worlds.board, want.boards), (m.decisions.logits, want.logits)]:
        assert np.array_equal(to_np(mine), theirs)
    assert np.array_equal(bits16(m.root_probs()), want.root_probs())


def test_toy_worlds_reference_goldens():
    """boardlaw/mcts/tests.py:242-279 (test_trivial, test_two_player, test_depth, test_multienv) on the generic path."""
    from boardlaw_amd import validation
    from boardlaw_amd.mcts import mcts
    g = gold('toy_envs.npz')
    agent = validation.ProxyAgent()
    torch.manual_seed(0)
    m = mcts(validation.Win.initial(device=DEV), agent, n_nodes=3)
    np.testing.assert_allclose(m.root().v.float().cpu().numpy(), g['win_v'])
    m = mcts(validation.WinnerLoser.initial(device=DEV), agent, n_nodes=3)
    np.testing.assert_allclose(m.root().v.float().cpu().numpy(), g['winnerloser_v'])
    m = mcts(validation.All.initial(length=3, device=DEV), agent, n_nodes=15, noise_eps=0.)
    np.testing.assert_allclose(m.root().v.float().cpu().numpy(), g['all_v'])
    m = mcts(validation.All.initial(n_envs=2, length=3, device=DEV), agent, n_nodes=15, noise_eps=0.)
    np.testing.assert_allclose(m.root().v.float().cpu().numpy(), g['all2_v'])


def test_planted_game_and_agent_surface():
    """boardlaw/mcts/tests.py:287-309 (planted 3x3 position) + the MCTSAgent output contract (mcts/__init__.py:223-229)."""
    from boardlaw_amd import hex, validation, networks
    from boardlaw_amd.mcts import mcts, MCTSAgent
    torch.manual_seed(3)
    world = hex.from_string("""
    wb.
    bw.
    wb.
    """, device=DEV)
    m = mcts(world, validation.RandomAgent(), n_nodes=63, c_puct=1., noise_eps=0.)
    probs = m.root().logits.exp()[0]
    assert (probs[2] > probs[8]) and (probs[5] > probs[7])

    worlds = hex.Hex.initial(64, 5, device=DEV)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=16, depth=4).to(DEV)
    agent = MCTSAgent(net, n_nodes=16)
    for _ in range(3):
        d = agent(worlds)
        assert d.logits.shape == (64, 25) and d.logits.dtype == torch.half and d.prior.dtype == torch.half
        assert d.v.shape == (64, 2) and d.v.dtype == torch.half
        assert d.n_sims.dtype == torch.long and (d.n_sims == 17).all() and d.n_leaves.dtype == torch.long
        assert d.actions.dtype == torch.long and worlds.valid.gather(1, d.actions[:, None]).all()
        worlds, _ = worlds.step(d.actions)
    sd = agent.state_dict()
    assert 'network.body.1.α' in sd and sd['kwargs.n_nodes'] == 16


def test_root_noise_matches_reference_within_f16():
    """dirichlet_noise + initialize (mcts/__init__.py:13-24,72-80) on the GPU vs the reference's recorded root prior:
    torch's device exp/log differ from the host's in the last bit, so this one is a tolerance test: 2 f16 ulps."""
    from boardlaw_amd.mcts import dirichlet_noise
    g = gold('search_5x5.npz')
    class Fixed:
        def __init__(self, d): self.d = d
        def dirichlet(self, alpha, shape): return self.d.clone()
    logits = torch.from_numpy(g['m0_net0_logits']).to(DEV)
    valid = torch.isfinite(logits)
    out = dirichlet_noise(logits, valid, .25, 10, Fixed(torch.from_numpy(g['m0_dirichlet']).to(DEV))).half()
    want = t16(g['m0_tree_logits'][:, 0], DEV)
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(out), fin)
    assert (out[fin].float() - want[fin].float()).abs().max() <= 2 * 2**-8


def test_graphed_move_equals_eager_move():
    """A HIP-graph replay of a whole move must produce exactly what the eager launch sequence produces for the same
    generator state (same draws, same kernels), move after move."""
    from boardlaw_amd import hex, networks
    from boardlaw_amd.mcts import MCTSAgent
    torch.manual_seed(0)
    worlds = hex.Hex.initial(256, 5, device=DEV)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=32, depth=2).to(DEV)
    eager, graphed = MCTSAgent(net, n_nodes=16), MCTSAgent(net, n_nodes=16, graph=True)
    graphed(worlds)                        # capture (consumes generator state during warm-up)
    for move in range(4):
        state = torch.cuda.get_rng_state()
        a = eager(worlds)
        torch.cuda.set_rng_state(state)
        b = graphed(worlds)
        for k in a:
            assert torch.equal(a[k], b[k]), (move, k)
        worlds, _ = worlds.step(a.actions)


def _expand_once(m, rands=None):
    """Runs bl_sim_expand for the search's current sim and returns the leaf view (helper for the head tests)."""
    import ctypes
    from boardlaw_amd import _native
    from boardlaw_amd.mcts import LeafWorlds
    rands = torch.rand_like(m.decisions.logits[:, :, 0]) if rands is None else rands
    _native.check(_native.lib().bl_sim_expand(ctypes.byref(m._search), m.sim, rands.data_ptr(), m._leaves.data_ptr(),
                                              m._obs.data_ptr(), m._valid.data_ptr(), m._leaf_seats.data_ptr(), _native.stream()))
    return LeafWorlds(m, m._leaves, m._obs, m._valid, m._leaf_seats)


@pytest.mark.parametrize('S,B', [(3, 65536), (4, 1000), (5, 4096), (9, 4096), (11, 512), (13, 512), (19, 64)])
def test_finish_heads_match_torch(S, B):
    """bl_sim_finish's heads (masked log-softmax -> f16, tanh -> seat scatter) against torch's own ops on this device,
    bit for bit: raw policy values over the whole f16 range of interest, every f16 value pattern for the value head
    (S=3 case), ragged action counts on both sides of the 64-lane width."""
    import ctypes
    from boardlaw_amd import _native, heads
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS
    torch.manual_seed(S)
    A = S * S
    worlds = Hex.initial(B, S, device=DEV)
    for _ in range(A // 3):
        v = worlds.valid
        worlds, _ = worlds.step((torch.rand(v.shape, device=DEV) * v).argmax(-1), check=False)
    m = MCTS(worlds, n_nodes=4, noise_eps=0.)
    m.plant_root(torch.log_softmax(torch.randn(B, A, device=DEV).masked_fill(~worlds.valid, -np.inf), -1), torch.zeros(B, 2, device=DEV))
    leaf = _expand_once(m)
    policy_raw = (torch.randn(B, A, device=DEV) * 4).half()
    policy_raw[::7] *= 8
    if B == 65536:
        value_raw = torch.arange(65536, dtype=torch.int32, device=DEV).to(torch.int16).view(torch.half)
    else:
        value_raw = (torch.randn(B, device=DEV) * 2).half()
    _native.check(_native.lib().bl_sim_finish(ctypes.byref(m._search), m.sim, m._leaves.data_ptr(), policy_raw.data_ptr(),
                                              value_raw.data_ptr(), m._valid.data_ptr(), m._leaf_seats.data_ptr(), _native.stream()))
    with torch.autocast('cuda'):
        want_logits = torch.nn.functional.log_softmax(policy_raw.masked_fill(~leaf.valid, -np.inf), -1).half()
        want_v = heads.scatter_values(torch.tanh(value_raw), leaf.seats).half()
    got_logits = m.decisions.logits[m.envs, m._leaves.long()]
    got_v = m.decisions.v[m.envs, m._leaves.long()]
    assert np.array_equal(bits16(got_logits), bits16(want_logits))
    nan = torch.isnan(want_v)
    assert torch.equal(torch.isnan(got_v), nan)
    assert np.array_equal(bits16(got_v)[~nan.cpu().numpy()], bits16(want_v)[~nan.cpu().numpy()])


def test_inference_plan_matches_autocast():
    """networks.Inference.raw (cached f16 weights + fused ReZero kernel) == FCModel.raw under fp16 autocast, bitwise."""
    from boardlaw_amd import networks, heads
    torch.manual_seed(5)
    for (S, width, depth, B) in [(9, 512, 4, 4096), (5, 16, 4, 64), (13, 256, 2, 1000)]:
        net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(S * S), width=width, depth=depth).to(DEV)
        with torch.no_grad():
            for blk in list(net.body)[1:]:
                getattr(blk, 'α').fill_(float(torch.randn(()) * 0.37))      # not f16-representable
        class W: pass
        w = W()
        w.obs = (torch.rand(B, S, S, 2, device=DEV) < .3).float()
        inf = networks.Inference(net)
        with torch.no_grad(), torch.autocast('cuda'):
            p0, v0 = net.raw(w)
            p1, v1 = inf.raw(w)
            w.obs = w.obs.half()
            p2, v2 = inf.raw(w)
        assert p0.dtype == torch.half and torch.equal(p0, p1) and torch.equal(v0, v1)
        assert torch.equal(p0, p2) and torch.equal(v0, v2)
        # a training step must be picked up by refresh()
        with torch.no_grad():
            net.policy.core.bias.add_(1.)
        inf.refresh()
        with torch.no_grad(), torch.autocast('cuda'):
            assert torch.equal(net.raw(w)[0], inf.raw(w)[0])


def test_raw_path_equals_torch_heads_path():
    """A whole search with the network's heads applied by bl_sim_finish equals the search with torch's heads feeding
    bl_sim_backup (same draws): trees, visit counts, root distribution."""
    from boardlaw_amd import hex, networks
    from boardlaw_amd.mcts import MCTS
    torch.manual_seed(1)
    worlds = hex.Hex.initial(512, 7, device=DEV)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=64, depth=3).to(DEV)
    with torch.no_grad():
        for blk in list(net.body)[1:]:
            getattr(blk, 'α').fill_(.3)
    class HeadsOnly:                      # hides .raw so that the search takes the torch-heads path
        def __init__(self, n): self.n = n
        def __call__(self, w): return self.n(w)
    results = []
    for network in (net, HeadsOnly(net), networks.Inference(net)):
        torch.manual_seed(2)
        m = MCTS(worlds, n_nodes=24, obs_half=bool(getattr(network, 'wants_half_obs', False)))
        m.initialize(network)
        for _ in range(23):
            m.simulate(network)
        results.append(m)
    for other in results[1:]:
        for a, b in [(results[0].tree.children, other.tree.children), (results[0].stats.n, other.stats.n),
                     (results[0].stats.w, other.stats.w), (results[0].decisions.logits, other.decisions.logits),
                     (results[0].decisions.v, other.decisions.v)]:
            assert np.array_equal(to_np(a), to_np(b))
        assert torch.equal(results[0].root_probs(), other.root_probs())


def test_sharded_search_with_merged_qrange_equals_unsharded(oracle):
    """SURVEY 8e option 2: two env shards that exchange their q-range state (element-wise MAX, what
    parallel.allreduce_qrange does over RCCL) after every backup reproduce the unsharded search bit for bit; without
    the exchange they do not (transition_q normalises over the whole batch, cuda.cu:101-105)."""
    from boardlaw_amd import parallel
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS
    S, B, T = 7, 512, 24
    board, seats = premixed(oracle, B, S, 16, seed=99)
    rands = np.random.default_rng(3).random((T - 1, B, T)).astype(np.float16).view(np.uint16)
    net = HashNetwork(DEV)

    def start(sl, r):
        w = Hex(board=torch.from_numpy(board[sl]).to(DEV), seats=torch.from_numpy(seats[sl]).to(DEV))
        m = MCTS(w, n_nodes=T, rng=ReplayRng(r, DEV), noise_eps=0.)
        d = net(w); m.plant_root(d.logits, d.v)
        return m

    full = start(slice(0, B), rands)
    for _ in range(T - 1):
        full.simulate(net)
    for merge in (True, False):
        a, b = start(slice(0, B // 2), rands[:, :B // 2]), start(slice(B // 2, B), rands[:, B // 2:])
        for _ in range(T - 1):
            a.simulate(net); b.simulate(net)
            if merge:
                row = parallel.merge_qrange(a._qrange[a.sim], b._qrange[b.sim])
                a._qrange[a.sim] = row; b._qrange[b.sim] = row
        same = all(np.array_equal(np.concatenate([to_np(x), to_np(y)]), to_np(z)) for x, y, z in
                   [(a.stats.n, b.stats.n, full.stats.n), (a.tree.children, b.tree.children, full.tree.children),
                    (a.stats.w, b.stats.w, full.stats.w)])
        same = same and np.array_equal(np.concatenate([bits16(a.root_probs()), bits16(b.root_probs())]), bits16(full.root_probs()))
        assert same == merge


@pytest.mark.parametrize('S,width,depth,B', [(9, 512, 4, 4096), (7, 128, 4, 1000), (8, 256, 4, 33), (6, 128, 1, 64), (13, 512, 2, 300), (3, 128, 4, 5000), (13, 1024, 8, 1024), (11, 768, 3, 200), (19, 1024, 2, 70), (9, 512, 5, 100),
                                              (9, 512, 4, 8300), (5, 256, 2, 9000)])      # more 32-row tiles than CUs: the 64-row instantiations
def test_fused_mlp_matches_autocast(S, width, depth, B):
    """bl_mlp_forward_f16 (one MFMA kernel for all Linears) vs the module under fp16 autocast: same rounding points,
    different GEMM summation order => a tolerance test.  Tolerance: 3 f16 ulps of the largest activation scale plus
    1% relative, on the pre-head outputs."""
    from boardlaw_amd import networks, heads
    torch.manual_seed(S)
    net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(S * S), width=width, depth=depth).to(DEV)
    with torch.no_grad():
        for blk in list(net.body)[1:]:
            getattr(blk, 'α').fill_(float(torch.randn(()) * 0.5))
    class W_: pass
    w = W_(); w.obs = (torch.rand(B, S, S, 2, device=DEV) < .3).half()
    fused = networks.Inference(net, fused=True)
    fused.FUSED_MIN_TILES = 0          # the kernel at every shape (the plan itself leaves wide networks on small batches to the GEMMs)
    fused.refresh()
    assert fused._packed is not None and fused.prefers_fused(B)
    with torch.no_grad(), torch.autocast('cuda'):
        p0, v0 = net.raw(w)
        p1, v1 = fused.raw(w)
    assert p1.shape == p0.shape and v1.shape == v0.shape and p1.dtype == torch.half
    for a, b in ((p0.float(), p1.float()), (v0.float(), v1.float())):
        tol = 3 * 2**-10 * a.abs().max().clamp(min=1.) + 0.01 * a.abs()
        assert ((a - b).abs() <= tol).all(), float((a - b).abs().max())
    # and the overwhelming majority of outputs should be bit-identical or 1 ulp off
    close = ((p0.float() - p1.float()).abs() <= 2**-9 * p0.float().abs().clamp(min=2**-5)).float().mean()
    assert close > 0.99, float(close)
    if B > 8192:
        # a row's outputs do not depend on the batch it is in: the first rows as a batch of their own (32-row tiles) against
        # the same rows inside the big batch (64-row tiles), bit for bit
        class W2_: pass
        w2 = W2_(); w2.obs = w.obs[:1000].contiguous()
        with torch.no_grad(), torch.autocast('cuda'):
            p2, v2 = fused.raw(w2)
        assert torch.equal(p2, p1[:1000]) and torch.equal(v2, v1[:1000])


@pytest.mark.parametrize('S,width,depth,B', [(13, 1024, 8, 1024), (9, 512, 4, 1000), (7, 128, 4, 33), (13, 768, 3, 200), (19, 1024, 2, 70), (6, 256, 0, 64), (11, 512, 5, 2049)])
def test_layers_mlp_matches_autocast_and_the_fused_kernel(S, width, depth, B):
    """bl_mlp_layers_f16 (a launch per Linear, every layer split over the chip: the plan for wide networks on small batches)
    against the module under fp16 autocast -- tolerance as for the one-kernel plan: 3 f16 ulps of the largest activation
    scale plus 1 % relative, >= 98 % of outputs within 1 ulp -- and against bl_mlp_forward_f16 itself (same rounding points,
    another order of the k blocks: >= 99 % of outputs bit-identical or 1 ulp apart)."""
    from boardlaw_amd import networks, heads
    torch.manual_seed(S + depth)
    net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(S * S), width=width, depth=depth).to(DEV)
    with torch.no_grad():
        for blk in list(net.body)[1:]:
            getattr(blk, 'α').fill_(float(torch.randn(()) * 0.5))
    class W_: pass
    w = W_(); w.obs = (torch.rand(B, S, S, 2, device=DEV) < .3).half()
    plan = networks.Inference(net, fused=True)
    plan.refresh()
    plan.FUSED_ALWAYS_BYTES = 0; plan.FUSED_MIN_TILES = 1 << 30          # never the one kernel
    assert plan._packed is not None and not plan.prefers_fused(B)
    with torch.no_grad(), torch.autocast('cuda'):
        p0, v0 = net.raw(w)
        p1, v1 = plan.raw(w)
        plan.FUSED_MIN_TILES = 0
        p2, v2 = plan.raw(w)
    assert p1.shape == p0.shape and v1.shape == v0.shape and p1.dtype == torch.half
    for a, b in ((p0.float(), p1.float()), (v0.float(), v1.float())):
        tol = 3 * 2**-10 * a.abs().max().clamp(min=1.) + 0.01 * a.abs()
        assert ((a - b).abs() <= tol).all(), float((a - b).abs().max())
    for ref, least in ((p0, 0.98), (p2, 0.99)):     # (the library's own summation order: 98.5 % at 1024x8, where both kernels agree bit for bit)
        close = ((ref.float() - p1.float()).abs() <= 2**-9 * ref.float().abs().clamp(min=2**-5)).float().mean()
        assert close > least, float(close)


@pytest.mark.parametrize('S,width,depth,B', [(13, 1024, 8, 1024), (13, 1024, 8, 1000), (9, 512, 4, 2048), (9, 512, 4, 77), (13, 768, 3, 200),
                                                 (6, 256, 0, 64), (11, 512, 5, 1500), (7, 256, 2, 4096)])
@pytest.mark.parametrize('mode', [True, 'xcd'])
def test_persistent_layers_kernel_equals_a_launch_per_linear(S, width, depth, B, mode):
    """bl_mlp_layers_persist_f16 (round 4: all Linears of the forward in ONE launch, workgroups synchronising per row tile through
    release/acquire counters) against bl_mlp_layers_f16 (a launch per Linear): the same arithmetic in the same order, so every output
    bit must be the same -- over repeated calls (the kernel re-zeroes its counters), inside a replayed capture, on ragged row counts,
    and for grids the kernel refuses (> 256 workgroups: the plan then launches per Linear by itself).  The error word stays clear.
    mode 'xcd' = bl_mlp_layers_xcd_f16 (round 5): the same launch with the hand-off kept inside one XCD's L2 (workgroups placed by
    HW_REG_XCC_ID tickets, plain stores, flag words, L1-bypassing loads)."""
    from boardlaw_amd import networks, heads
    torch.manual_seed(S + depth + B)
    net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(S * S), width=width, depth=depth).to(DEV)
    with torch.no_grad():
        for blk in list(net.body)[1:]:
            getattr(blk, 'α').fill_(float(torch.randn(()) * 0.5))
    class W_: pass
    plan = networks.Inference(net, fused=True)
    plan.refresh()
    plan.FUSED_ALWAYS_BYTES = 0; plan.FUSED_MIN_TILES = 1 << 30          # never the one kernel
    outs = {}
    for persist in (False, mode):
        plan.PERSIST_PLAN = persist
        res = []
        for rep in range(3):
            w = W_(); w.obs = (torch.rand(B, S, S, 2, device=DEV, generator=torch.Generator(device=DEV).manual_seed(rep)) < .3).half()
            with torch.no_grad():
                res.append(plan.raw(w))
        outs[persist] = res
    for (p0, v0), (p1, v1) in zip(outs[False], outs[mode]):
        assert torch.equal(p0.view(torch.int16), p1.view(torch.int16)) and torch.equal(v0.view(torch.int16), v1.view(torch.int16))
    # a captured forward replays (the counters are zero again after every launch)
    plan.PERSIST_PLAN = mode
    w = W_(); w.obs = (torch.rand(B, S, S, 2, device=DEV) < .3).half()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.raw(w)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        pg, vg = plan.raw(w)
    for rep in range(3):
        w.obs.copy_((torch.rand(B, S, S, 2, device=DEV) < .3).half())
        g.replay()
        plan.PERSIST_PLAN = False
        with torch.no_grad():
            pe, ve = plan.raw(w)
        plan.PERSIST_PLAN = mode
        assert torch.equal(pg.view(torch.int16), pe.view(torch.int16)) and torch.equal(vg.view(torch.int16), ve.view(torch.int16)), rep
    assert not plan.persist_error()


@pytest.mark.parametrize('mode', [True, 'xcd'])
def test_persistent_layers_hand_off_under_uneven_load(mode):
    """The in-launch hand-offs of bl_mlp_layers_persist_f16 / bl_mlp_layers_xcd_f16 while ANOTHER stream keeps part of the chip busy
    (a loop of GEMMs of changing size: workgroups of a row tile then start at very different times, and the L1s are warm with the
    previous forward's rows): 40 replays of a captured 1024x8 forward on 1024 rows of 13x13 with fresh inputs, every output word
    compared with the launch-per-Linear plan; the error word stays clear."""
    from boardlaw_amd import networks, heads
    S, width, depth, B = 13, 1024, 8, 1024
    torch.manual_seed(5)
    net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(S * S), width=width, depth=depth).to(DEV)
    with torch.no_grad():
        for blk in list(net.body)[1:]:
            getattr(blk, 'α').fill_(float(torch.randn(()) * 0.5))
    class W_: pass
    plan = networks.Inference(net, fused=True)
    plan.refresh()
    plan.FUSED_ALWAYS_BYTES = 0; plan.FUSED_MIN_TILES = 1 << 30
    plan.PERSIST_PLAN = mode
    w = W_(); w.obs = (torch.rand(B, S, S, 2, device=DEV) < .3).half()
    side, noise = torch.cuda.Stream(), torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.raw(w)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        pg, vg = plan.raw(w)
    mats = [torch.randn(n, n, device=DEV, dtype=torch.half) for n in (256, 1024, 3072)]
    bad = 0
    for rep in range(40):
        w.obs.copy_((torch.rand(B, S, S, 2, device=DEV) < .3).half())
        torch.cuda.synchronize()
        with torch.cuda.stream(noise):
            for k in range(6):
                m_ = mats[(rep + k) % 3]; m_ @ m_
        g.replay()
        torch.cuda.synchronize()
        plan.PERSIST_PLAN = False
        with torch.no_grad():
            pe, ve = plan.raw(w)
        plan.PERSIST_PLAN = mode
        bad += int((pg.view(torch.int16) != pe.view(torch.int16)).sum()) + int((vg.view(torch.int16) != ve.view(torch.int16)).sum())
    assert bad == 0 and not plan.persist_error(), bad


def test_move_rng_serves_one_block_per_move():
    from boardlaw_amd import hex, networks
    from boardlaw_amd.mcts import MCTSAgent, MoveRng, mcts
    torch.manual_seed(0)
    worlds = hex.Hex.initial(128, 5, device=DEV)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=32, depth=2).to(DEV)
    rng = MoveRng()
    m = mcts(worlds, net, n_nodes=16, rng=rng)
    assert rng.block.shape == (15, 128, 16) and rng.block.dtype == torch.half and rng.i == 15
    assert (to_np(m.stats.n)[:, 0] == 30).all()
    d = MCTSAgent(net, n_nodes=16, graph=True, rng=MoveRng())(worlds)
    assert worlds.valid.gather(1, d.actions[:, None]).all()


@pytest.mark.parametrize('group', [8, 16, 32])
def test_narrow_group_paths_in_subprocess(group):
    """The launch heuristic now always picks one wave per env; the 8/16/32-lanes-per-env kernels (LDS fold) stay in the
    library behind BL_FORCE_GROUP, which is read once per process -- so they are parity-tested in a child process."""
    import subprocess, sys
    env = dict(os.environ, BL_FORCE_GROUP=str(group))
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(here, 'test_gpu_parity.py'), '-q', '-x', '-m', 'gpu',
                        '-k', 'test_descend_root_backup_golden and (3x3 or 5x5 or 9x9) and (1- or 37-) or test_random_trees_vs_oracle or '
                              '(test_full_size_search_vs_oracle and 5-64-16) or (test_whole_search_replay and 5x5)'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout


def test_two_nodes_per_wave_expand_in_subprocess():
    """bl_expand.hip's sim_expand3_kernel (one wave per env evaluating the current node and its guessed continuation in its two
    halves; BL_EXPAND_WAVES=21, read once per process) is a measured alternative, not the default: the same full-size oracle
    comparisons must hold for it bit for bit."""
    import subprocess, sys
    env = dict(os.environ, BL_EXPAND_WAVES='21')
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(here, 'test_gpu_parity.py'), '-q', '-x', '-m', 'gpu',
                        '-k', '(test_whole_search_replay and fused) or (test_full_size_search_vs_oracle and 9-4096-64) or '
                              '(test_bench_launch_sequence_vs_oracle and 9-4096-64-512-4-graph) or test_search_with_per_env_c_puct'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout


def test_four_envs_per_wave_expand_in_subprocess():
    """bl_rows.hip (descend with four envs per wave, one per 16-lane DPP row, then the expansion as its own launch) is what the
    library picks from 16384 envs on -- the (9, 32768, 64) oracle comparisons above run it.  Forced at every batch size
    (BL_EXPAND_WAVES=16, read once per process) the other oracle comparisons must hold bit for bit too: boards 2x2 .. 9x9 (1 .. 6
    kept actions per lane), ragged and tiny batches (rows without an env, waves whose rows take several envs one after the
    other), per-env c_puct, recorded reference searches, masked searches, the bench's launch sequence eager and captured."""
    import subprocess, sys
    env = dict(os.environ, BL_EXPAND_WAVES='16')
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(here, 'test_gpu_parity.py'), '-q', '-x', '-m', 'gpu',
                        '-k', '(test_whole_search_replay and fused) or (test_full_size_search_vs_oracle and not 32768 and not 13- and not 19- and not 32-3 and not 11-333) or '
                              '(test_bench_launch_sequence_vs_oracle and (9-4096-64-512-4 or 5-64-16)) or test_search_with_per_env_c_puct or '
                              'test_ragged_and_tiny_batches or test_graphed_move_equals_eager_move or test_lazy_reset_equals_the_eager_reset'],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout


def test_removed_shared_workgroup_expand_is_refused():
    """Round 4's shared-workgroup bl_sim_expand (bl_tune_t.expand_envs = 2 / 4) was removed in round 5 (bit-exact, slower, and its
    protocol trapped on an exhausted poll budget): the field is reserved and any value but 0 / 1 is BL_EINVAL, not a silent default."""
    from boardlaw_amd import _native
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS
    m = MCTS(Hex.initial(8, 5, device=DEV), n_nodes=4)
    m._search.tune.expand_envs = 2
    z = torch.zeros(8 * 64, dtype=torch.int32, device=DEV)
    rc = _native.lib().bl_sim_expand(ctypes.byref(m._search), 1, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), _native.stream(m.device))
    assert rc == _native.BL_EINVAL


# ------------------------------------------------------------------------------------------------ widened rows
@pytest.mark.parametrize('inference,graph', [(None, False), ('fused', True)])
def test_actor_learner_loop_runs_and_learns_something(inference, graph):
    """a few actor/learner rounds on 5x5 Hex; the losses are finite and the weights move, also when the
    actor replays a captured graph with the fused inference plan (weights refreshed inside the graph)."""
    from boardlaw_amd import hex, networks, training
    torch.manual_seed(0)
    worlds = hex.Hex.initial(256, 5, device=DEV)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=128, depth=2).to(DEV)
    before = torch.cat([p.detach().flatten().clone() for p in net.parameters()])
    log = []
    training.run(worlds, net, n_steps=3, nodes=8, buffer_len=4, graph=graph, inference=inference,
                 on_step=lambda i, pl, vl: log.append((float(pl), float(vl))))
    after = torch.cat([p.detach().flatten() for p in net.parameters()])
    assert len(log) == 3 and all(np.isfinite(x).all() for x in log)
    assert (after - before).abs().max() > 0


def test_actor_learner_step_at_config_4_per_gpu_shape():
    """BASELINE config 4's per-GPU shape through the actor/learner loop: 13x13, 1024 envs, 256 sims/move, FCModel 1024x8, captured
    moves with the plan's own choice of network kernels (a launch per Linear), one learner step under AMP with the gradient
    all-reduce in the path (a no-op group of one here; two ranks at this network's size: tests/test_training.py).  Losses finite,
    every parameter tensor moved, the worlds still legal."""
    from boardlaw_amd import hex, networks, training
    torch.manual_seed(0)
    worlds = hex.Hex.initial(1024, 13, device=DEV)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=1024, depth=8).to(DEV)
    with torch.no_grad():
        for p_ in net.parameters():
            if p_.ndim == 0:
                p_.fill_(0.3)
    before = [p.detach().clone() for p in net.parameters()]
    log = []
    out = training.run(worlds, net, n_steps=1, nodes=256, buffer_len=3, graph=True, inference='fused',
                       on_step=lambda i, pl, vl: log.append((float(pl), float(vl))))
    torch.cuda.synchronize()
    assert len(log) == 1 and np.isfinite(log[0]).all() and log[0][0] > 0
    assert all((a.detach() - b).abs().max() > 0 for a, b in zip(net.parameters(), before))
    stones = (out.board != 0).flatten(1).sum(-1)
    assert out.n_envs == 1024 and int(stones.max()) <= 3 and int(stones.min()) >= 0 and bool(out.valid.any(-1).all())


def test_actor_learner_loop_with_two_actors_on_one_gpu():
    """training.run with a list of env batches: two actors searching concurrently (own stream, own generator, captured
    moves, fused plans), one learner step per actor chunk.  Same seed => same losses, bit for bit, whatever the overlap on
    the device; the actors play different games; the weights move."""
    from boardlaw_amd import hex, networks, training

    def once():
        torch.manual_seed(5)
        batches = [hex.Hex.initial(512, 5, device=DEV), hex.Hex.initial(512, 5, device=DEV)]
        net = networks.FCModel(batches[0].obs_space, batches[0].action_space, width=128, depth=2).to(DEV)
        before = torch.cat([p.detach().flatten().clone() for p in net.parameters()])
        log = []
        out = training.run(batches, net, n_steps=3, nodes=16, buffer_len=4, graph=True, inference='fused',
                           on_step=lambda i, pl, vl: log.append((float(pl), float(vl))))
        torch.cuda.synchronize()
        after = torch.cat([p.detach().flatten() for p in net.parameters()])
        return log, out, before, after

    log, out, before, after = once()
    assert isinstance(out, list) and len(out) == 2 and len(log) == 6 and all(np.isfinite(x).all() for x in log)
    assert (after - before).abs().max() > 0
    assert not torch.equal(out[0].board, out[1].board)
    log2, out2, _, after2 = once()
    assert log2 == log and torch.equal(after2, after) and all(torch.equal(a.board, b.board) for a, b in zip(out, out2))


def test_arena_evaluate_with_search_agents():
    """arena.common.evaluate with two MCTS agents on Hex -- masked, variable-size batches, argmax actions."""
    from boardlaw_amd import arena, hex, networks
    from boardlaw_amd.mcts import MCTSAgent
    torch.manual_seed(0)
    worlds = hex.Hex.initial(64, 4, device=DEV)
    nets = [networks.FCModel(worlds.obs_space, worlds.action_space, width=32, depth=1).to(DEV) for _ in range(2)]
    agents = {'a': MCTSAgent(nets[0], n_nodes=8), 'b': MCTSAgent(nets[1], n_nodes=8)}
    results = arena.evaluate(worlds, agents)
    assert len(results) == 2 and sum(r.games for r in results) == 64
    assert all(sum(r.wins) == r.games for r in results)


def test_ragged_and_tiny_batches():
    """Variable B as the arena produces it: 1, 3, 65 envs through the whole agent, eager and graphed."""
    from boardlaw_amd import hex, networks
    from boardlaw_amd.mcts import MCTSAgent
    torch.manual_seed(0)
    net = None
    for B in (1, 3, 65):
        worlds = hex.Hex.initial(B, 5, device=DEV)
        net = net or networks.FCModel(worlds.obs_space, worlds.action_space, width=16, depth=2).to(DEV)
        for graph in (False, True):
            d = MCTSAgent(net, n_nodes=8, graph=graph)(worlds, eval=True)
            assert d.actions.shape == (B,) and worlds.valid.gather(1, d.actions[:, None]).all()
            assert (d.n_sims == 9).all()


def test_graphed_play_equals_eager_play():
    """MCTSAgent.play (search + env step) replayed as one graph == the two eager calls, for the same generator state."""
    from boardlaw_amd import hex, networks
    from boardlaw_amd.mcts import MCTSAgent
    torch.manual_seed(0)
    worlds = hex.Hex.initial(128, 4, device=DEV)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=32, depth=2).to(DEV)
    eager, graphed = MCTSAgent(net, n_nodes=12), MCTSAgent(net, n_nodes=12, graph=True)
    graphed.play(worlds)
    for move in range(12):                  # long enough for 4x4 games to end and reset
        state = torch.cuda.get_rng_state()
        d0, w0, t0 = eager.play(worlds)
        torch.cuda.set_rng_state(state)
        d1, w1, t1 = graphed.play(worlds)
        assert torch.equal(d0.actions, d1.actions) and torch.equal(d0.logits, d1.logits)
        assert torch.equal(w0.board, w1.board) and torch.equal(w0.seats, w1.seats)
        assert torch.equal(t0.terminal, t1.terminal) and torch.equal(t0.rewards, t1.rewards)
        worlds = w1
    assert isinstance(worlds, hex.Hex)


def test_one_player_hex_variants():
    """hex.Solitaire/Lazy/Random (hex/__init__.py:224-274): the player is always seat 0, the scripted
    opponent answers inside step(), rewards are the player's column only."""
    from boardlaw_amd import hex, validation
    from boardlaw_amd.mcts import mcts
    with pytest.raises(ValueError):
        hex.Lazy.initial(2, 3, seat=1, device=DEV)
    w = hex.Lazy.initial(4, 3, device=DEV)
    assert w.n_seats == 1
    w2, tr = w.step(torch.full((4,), 4, device=DEV))
    assert type(w2) is hex.Lazy and (w2.seats == 0).all() and tr.rewards.shape == (4, 1) and not tr.terminal.any()
    # black in the centre; white answered at ITS first free cell = index 0 of the transposed frame = (0,0), a left-edge cell
    assert (w2.board[:, 1, 1] == 1).all() and (w2.board[:, 0, 0] == 5).all() and (w2.board != 0).sum() == 8
    # same position reached with plain two-player Hex
    h = hex.Hex.initial(4, 3, device=DEV)
    h, _ = h.step(torch.full((4,), 4, device=DEV))
    h, _ = h.step(torch.zeros(4, dtype=torch.long, device=DEV))
    assert torch.equal(h.board, w2.board)

    # random games to the end: the player only ever sees seat 0, every env finishes, rewards are -1/0/+1 and a reward
    # arrives exactly with `terminal` (a loss is delivered through the opponent's winning reply)
    torch.manual_seed(0)
    w = hex.Random.initial(64, 4, device=DEV)
    done = torch.zeros(64, dtype=torch.bool, device=DEV)
    for _ in range(40):
        a = torch.distributions.Categorical(probs=w.valid.float()).sample()
        w, tr = w.step(a)
        assert (w.seats == 0).all() and tr.rewards.shape == (64, 1)
        assert ((tr.rewards[:, 0] != 0) == tr.terminal).all() and tr.rewards.abs().max() <= 1
        assert (w.board[tr.terminal] == 0).all()
        done |= tr.terminal
    assert done.all()

    # a one-seat search over it runs on the generic path (S = 1)
    w = hex.Lazy.initial(8, 3, device=DEV)
    m = mcts(w, validation.RandomAgent(), n_nodes=8)
    assert not m.fused and m.stats.w.shape == (8, 8, 1) and (m.stats.n[:, 0] > 0).all()
    assert torch.allclose(m.root().logits.float().exp().sum(-1), torch.ones(8, device=DEV), atol=2e-2)


@pytest.mark.parametrize('S,B,T,width,depth', [(9, 1000, 64, 512, 4), (5, 100, 16, 256, 2), (11, 33, 64, 512, 1), (3, 70, 8, 256, 3),
                                                 (9, 4096, 64, 512, 4), (8, 5, 40, 1024, 2)])
def test_infer_finish_in_one_launch_equals_two_launches(S, B, T, width, depth):
    """bl_sim_infer_finish (network + heads + store + backup + next q range in one kernel) against bl_mlp_forward_f16
    followed by bl_sim_finish: every array of the finished search identical."""
    from boardlaw_amd import hex, networks
    from boardlaw_amd.mcts import mcts
    torch.manual_seed(S * 100 + T)
    worlds = hex.Hex.initial(B, S, device=DEV)
    for _ in range((S * S) // 3):
        r = torch.rand(worlds.valid.shape, device=DEV) * worlds.valid
        worlds, _ = worlds.step(r.argmax(-1), check=False)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=width, depth=depth).to(DEV)
    inf = networks.Inference(net, fused=True)
    assert inf.fused_params() is not None
    from boardlaw_amd.mcts import MCTS
    out = []
    for fuse in (True, False):
        m = MCTS(worlds, n_nodes=T, fuse_finish=fuse, obs_half=True)
        if fuse:
            torch.manual_seed(5)
            m.initialize(inf)
        else:
            # the same root for both runs (the two root routes agree to an f16 ulp, not bit for bit:
            # test_plant_root_in_one_launch_matches_the_composition); what is compared here is the simulations
            m.plant_root(out[0].decisions.logits[:, 0].clone(), out[0].decisions.v[:, 0].clone())
        torch.manual_seed(6)                 # same uniforms for both runs
        for _ in range(T - 1):
            m.simulate(inf)
        out.append(m)
    a, b = out
    for x, y in [(a.tree.children, b.tree.children), (a.tree.parents, b.tree.parents), (a.tree.relation, b.tree.relation),
                 (a.stats.n, b.stats.n), (a.stats.w, b.stats.w), (a.decisions.logits, b.decisions.logits),
                 (a.decisions.v, b.decisions.v), (a.worlds.board, b.worlds.board)]:
        assert np.array_equal(to_np(x), to_np(y))
    # the q-range state is a set of slots whose element-wise max is the range: slot assignment differs, the range must not
    from boardlaw_amd import _native
    for row in range(1, T + 1):
        assert torch.equal(_native.qrange_decode(a._qrange[row]), _native.qrange_decode(b._qrange[row]))
    assert np.array_equal(bits16(a.root_probs()), bits16(b.root_probs()))
    assert (to_np(a.stats.n)[:, 0] == 2 * (T - 1)).all()


@pytest.mark.parametrize('S,B,T,width,depth', [(9, 1000, 64, 512, 4), (9, 8193, 24, 512, 4), (5, 100, 16, 256, 2), (11, 33, 64, 512, 1), (3, 70, 8, 256, 3),
                                                 (7, 513, 20, 512, 0)])
def test_infer_finish_on_64_row_tiles_equals_32_row_tiles(S, B, T, width, depth):
    """bl_sim_infer_finish with 64 rows per workgroup (bl_tune_t.mlp_rows = 64; what the library picks by itself once the 32-row
    tiles outnumber the CUs: every weight fragment then feeds two MFMAs, and a wave finishes eight envs in two passes) against
    the 32-row kernel: the k order of every accumulator is the same, so every array of the finished search must be identical --
    ragged batches (rows beyond M in both row groups), one tile, many tiles; the (9, 8193) case is also the library's own choice."""
    from boardlaw_amd import hex, networks, _native
    from boardlaw_amd.mcts import MCTS
    torch.manual_seed(S * 100 + T)
    worlds = hex.Hex.initial(B, S, device=DEV)
    for _ in range((S * S) // 3):
        r = torch.rand(worlds.valid.shape, device=DEV) * worlds.valid
        worlds, _ = worlds.step(r.argmax(-1), check=False)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=width, depth=depth).to(DEV)
    with torch.no_grad():
        for p_ in net.parameters():
            if p_.ndim == 0:
                p_.fill_(0.3)
    inf = networks.Inference(net, fused=True)
    assert inf.fused_params(B) is not None
    out = []
    for rows in (32, 64, 0):
        m = MCTS(worlds, n_nodes=T, obs_half=True)
        m._search.tune.mlp_rows = rows
        torch.manual_seed(5)
        m.initialize(inf)
        torch.manual_seed(6)                 # same uniforms for all runs
        for _ in range(T - 1):
            m.simulate(inf)
        out.append(m)
    a = out[0]
    for b in out[1:]:
        for name in ('tree.children', 'tree.parents', 'tree.relation', 'stats.n', 'stats.w', 'decisions.logits', 'decisions.v', 'worlds.board', '_nk'):
            x, y = a, b
            for part in name.split('.'):
                x, y = getattr(x, part), getattr(y, part)
            assert np.array_equal(to_np(x), to_np(y)), (name, b._search.tune.mlp_rows)
        # the compacted rows' kept entries (what lies beyond a row's nk entries is never written)
        kept = (torch.arange(a._cpi.shape[-1], device=DEV)[None, None] < a._nk[..., None]) & (a.tree.parents != -1)[..., None]
        kept[:, 0] = torch.arange(a._cpi.shape[-1], device=DEV)[None] < a._nk[:, 0, None]
        assert torch.equal(a._cpi[kept].view(torch.int32), b._cpi[kept].view(torch.int32)) and torch.equal(a._cca[kept], b._cca[kept])
        for row in range(1, T + 1):
            assert torch.equal(_native.qrange_decode(a._qrange[row]), _native.qrange_decode(b._qrange[row]))
        assert np.array_equal(bits16(a.root_probs()), bits16(b.root_probs()))
    assert (to_np(a.stats.n)[:, 0] == 2 * (T - 1)).all()
    # widths whose two 64-row activation buffers do not fit the LDS refuse the explicit request
    if width == 512 and S == 9 and B == 1000:
        wide = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, width=1024, depth=1).to(DEV), fused=True)
        wide.FUSED_MIN_TILES = 0
        m = MCTS(worlds, n_nodes=4, obs_half=True)
        m._search.tune.mlp_rows = 64
        m.initialize(wide)
        with pytest.raises(_native.NativeError):
            m.simulate(wide)


@pytest.mark.parametrize('S,B', [(9, 4096), (3, 50), (13, 257), (32, 9), (1, 4)])
def test_world_step_in_one_launch_equals_the_composed_step(S, B):
    """bl_hex_world_step (Hex.step with reset=True as one kernel) against clone + bl_hex_step + the reference's mask
    arithmetic (hex/__init__.py:181-190), along random games with resets; i32 and i64 actions."""
    from boardlaw_amd import hex
    from boardlaw_amd.hex import cuda as hcuda
    torch.manual_seed(S)
    w = hex.Hex.initial(B, S, device=DEV)
    finished = 0
    for t in range(3 * S * S // 2 + 2):
        a = (torch.rand(w.valid.shape, device=DEV) * w.valid).argmax(-1)
        board = w.board.clone()
        rewards = hcuda.step(board, w.seats.int(), a.int())
        terminal = (rewards > 0).any(-1)
        board[terminal] = 0
        seats = 1 - w.seats
        seats[terminal] = 0
        nb, ns, nr, nt = hcuda.world_step(w.board, w.seats, a if t % 2 else a.int())
        assert torch.equal(nb, board) and torch.equal(ns, seats) and torch.equal(nr, rewards) and torch.equal(nt, terminal)
        w2, tr = w.step(a)                                   # the public entry point takes the same route
        assert torch.equal(w2.board, board) and torch.equal(tr.terminal, terminal) and w2.seats.dtype == torch.int32
        finished += int(terminal.sum())
        w = w2
    assert finished > 0


@pytest.mark.parametrize('S,B,width,depth,fused', [(9, 300, 256, 3, False), (9, 300, 256, 3, True), (9, 4096, 512, 4, True), (13, 1000, 1024, 8, True),
                                                   (3, 7, 128, 0, True), (11, 33, 768, 2, True), (5, 4097, 384, 1, True), (19, 20, 1024, 1, True),
                                                   (19, 20, 512, 2, True)])
def test_root_plan_matches_module(S, B, width, depth, fused):
    """Inference.root_raw -- the root evaluation's fp32 Linears -- against FCModel.raw in fp32 on the device.
      fused=False: torch's GEMMs + bl_rezero_relu_f32, both heads as one GEMM over the stacked weights;
      fused=True:  bl_root_mlp_f32, every Linear in one kernel on v_mfma_f32_16x16x4_f32 (19x19 at width 512: the
                   flattened board is wider than the body, so the plan falls back to the GEMMs).
    Same fp32 rounding points as the module; the K-summation order inside each Linear is the GEMM kernel's own.  Tolerance,
    on pre-head outputs of magnitude ~1: 1e-5 + 1e-5 |ref| per body layer -- two orders inside the f16 step (2^-11
    relative) at which plant_root stores them."""
    from boardlaw_amd import hex, networks
    torch.manual_seed(3)
    worlds = hex.Hex.initial(B, S, device=DEV)
    for _ in range(S):
        r = torch.rand(worlds.valid.shape, device=DEV) * worlds.valid
        worlds, _ = worlds.step(r.argmax(-1), check=False)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=width, depth=depth).to(DEV)
    with torch.no_grad():
        for blk in list(net.body)[1:]:
            getattr(blk, 'α').fill_(float(torch.randn(()) * 0.7))
        p0, v0 = net.raw(worlds)
    inf = networks.Inference(net, fused=fused)
    inf.ROOT_FUSED_MIN_ROWS = 0        # the kernel at every batch size (the plan itself switches to it from 2048 rows)
    p1, v1 = inf.root_raw(worlds)
    assert (inf._root_packed is not None) == (fused and -(-2 * S * S // 64) * 64 <= width)
    assert p1.dtype == torch.float and p1.shape == p0.shape and v1.shape == v0.shape
    def close(a, b):
        return bool(((a - b).abs() <= (depth + 1) * (1e-5 + 1e-5 * a.abs())).all())
    assert close(p0, p1) and close(v0, v1), (float((p0 - p1).abs().max()), float((v0 - v1).abs().max()))
    # the packed weights follow the module: after in-place updates the next call uses the new parameters
    with torch.no_grad():
        net.value.core.weight.mul_(2.); net.policy.core.bias.add_(1.); list(net.body)[0].weight.mul_(.5)
        p2, v2 = net.raw(worlds)
    p3, v3 = inf.root_raw(worlds)
    assert close(p2, p3) and close(v2, v3) and not close(v0, v3)


class _FixedDraw:
    """rng stub: hands the same Dirichlet draw to both code paths."""
    def __init__(self, draw): self.draw = draw
    def dirichlet(self, alpha, shape): return self.draw.clone()
    def rand_like(self, x): return torch.rand_like(x)


@pytest.mark.parametrize('S,B,eps', [(9, 500, .25), (5, 77, .25), (13, 40, .1), (3, 9, 0.)])
def test_plant_root_in_one_launch_matches_the_composition(S, B, eps):
    """bl_sim_plant_root (masked log-softmax + dirichlet mixing + tanh/scatter + f16 store in one kernel) against
    network(world) -> dirichlet_noise -> plant_root.  The kernel sums the masked draw in its own order, so: every stored
    value within 1 f16 ulp, and all but a handful identical."""
    from boardlaw_amd import hex, networks
    from boardlaw_amd.mcts import MCTS
    torch.manual_seed(S)
    worlds = hex.Hex.initial(B, S, device=DEV)
    for _ in range(S):
        worlds, _ = worlds.step((torch.rand(worlds.valid.shape, device=DEV) * worlds.valid).argmax(-1), check=False)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=128, depth=2).to(DEV)
    inf = networks.Inference(net, fused=True)
    draw = torch.distributions.Dirichlet(torch.full((S * S,), 10 / (S * S), device=DEV)).sample((B,))
    a = MCTS(worlds, n_nodes=4, noise_eps=eps, rng=_FixedDraw(draw), obs_half=True)
    a.initialize(inf)                                  # one launch
    b = MCTS(worlds, n_nodes=4, noise_eps=eps, rng=_FixedDraw(draw), obs_half=True, fuse_finish=False)
    b.initialize(inf)                                  # network(world) + dirichlet_noise + plant_root
    assert a.sim == b.sim == 1
    la, lb = a.decisions.logits[:, 0].float(), b.decisions.logits[:, 0].float()
    assert torch.equal(torch.isinf(la), torch.isinf(lb))
    fin = ~torch.isinf(la)
    assert ((la - lb)[fin].abs() <= 2**-10 * lb[fin].abs().clamp(min=2**-14)).all()
    assert (bits16_t(a.decisions.logits[:, 0]) == bits16_t(b.decisions.logits[:, 0])).float().mean() > 0.995
    assert torch.equal(a.decisions.v[:, 0], b.decisions.v[:, 0])


def bits16_t(t):
    return t.contiguous().view(torch.int16)


# ------------------------------------------------------------------------------------------------ the bench's own launch sequence
class RecordingRng:
    """Passes torch's draws through and keeps the descend uniforms, in order."""

    def __init__(self, inner):
        self.inner, self.rands = inner, []

    def dirichlet(self, alpha, shape):
        return self.inner.dirichlet(alpha, shape)

    def rand_like(self, x):
        r = self.inner.rand_like(x)
        self.rands.append(r)
        return r

    def categorical(self, logits):
        return self.inner.categorical(logits)


def replay_through_oracle(oracle, board, seats, T, rands_bits, logits_bits, v_bits):
    """The host search driven by the oracle, fed the GPU's own leaf evaluations: decisions.logits[b, leaf] / decisions.v[b, leaf]
    are what the network kernels stored for the node created (or re-visited) at that simulation."""
    s = OracleSearch(oracle, board, seats, T)
    s.initialize(logits_bits[:, 0], v_bits[:, 0])
    e = np.arange(board.shape[0])
    for i in range(T - 1):
        parents, actions = s.descend(rands_bits[i])
        leaves, _, _ = s.expand(parents, actions)
        s.finish(leaves, logits_bits[e, leaves], v_bits[e, leaves])
    return s


def assert_search_equals(m, want):
    for name, mine, theirs in [('children', m.tree.children, want.children), ('parents', m.tree.parents, want.parents),
                               ('relation', m.tree.relation, want.relation), ('n', m.stats.n, want.n), ('w', m.stats.w, want.w),
                               ('rewards', m.transitions.rewards, want.rewards), ('terminal', m.transitions.terminal, want.terminal),
                               ('boards', m.worlds.board, want.boards), ('seats', m.worlds.seats, want.seats),
                               ('v', m.decisions.v, want.v), ('logits', m.decisions.logits, want.logits)]:
        assert np.array_equal(to_np(mine), theirs), name
    assert np.array_equal(bits16(m.root_probs()), want.root_probs())
    assert np.array_equal(to_np(m.n_leaves()), want.n_leaves())


@pytest.mark.parametrize('S,B,T,width,depth,mode', [(9, 4096, 64, 512, 4, 'eager'), (9, 4096, 64, 512, 4, 'graph'),
                                                    (13, 1024, 256, 1024, 8, 'eager'), (5, 64, 16, 256, 2, 'graph'),
                                                    (9, 333, 64, 512, 4, 'eager-torch-gemms'),
                                                    (13, 1024, 256, 1024, 8, 'graph-plan'), (9, 512, 64, 1024, 4, 'eager-plan'),
                                                    (9, 32768, 64, 512, 4, 'graph')])     # `bench.py --envs 32768`: boardlaw/main.py:147's n_envs
def test_bench_launch_sequence_vs_oracle(oracle, S, B, T, width, depth, mode):
    """What bench.py times -- bl_sim_plant_root, then T-1 x (bl_sim_expand -> bl_sim_infer_finish) with the real network's
    fused fp16 plan (13x13/256 nodes: bl_mlp_forward_f16 + bl_sim_finish, the T > 64 route; '-plan': what the plan picks for
    config 4's 1024x8 network on 1024 rows, bl_mlp_layers_f16 + bl_sim_finish), eagerly with torch's
    per-simulation uniforms and as a captured HIP graph with MoveRng -- replayed on the host through the oracle, which is
    handed the uniforms and the leaf evaluations the GPU stored.  Every tree array, visit count, value sum, board and the
    root distribution must be identical: BASELINE config 2 and config 4's per-GPU shape at full size."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS, MoveRng, TorchRng, mcts
    board, seats = premixed(oracle, B, S, (S * S) // 3, seed=31 * S + T + SEED)
    world = Hex(board=torch.from_numpy(board).to(DEV), seats=torch.from_numpy(seats).to(DEV))
    torch.manual_seed(5 + SEED)
    net = networks.Inference(networks.FCModel(world.obs_space, world.action_space, width=width, depth=depth).to(DEV),
                             fused=(mode != 'eager-torch-gemms'))
    with torch.no_grad():
        for p_ in net.model.parameters():
            if p_.ndim == 0:
                p_.fill_(0.3)          # ReZero gains start at 0: make the evaluation depend on the position
    if mode.endswith('-plan'):         # the plan's own choice for a wide network on a small batch: bl_mlp_layers_f16 + bl_sim_finish
        net.refresh()
        assert not net.prefers_fused(B)
    else:
        net.FUSED_MIN_TILES = 0        # keep bl_mlp_forward_f16 on the 13x13 / 1024-row case (the plan would pick a launch per Linear there)
        net.refresh()
    if mode.startswith('graph'):
        rng = MoveRng()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            mcts(world, net, n_nodes=T, rng=rng)          # warm-up outside the capture
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            m = mcts(world, net, n_nodes=T, rng=rng)
        graph.replay(); graph.replay()                    # the second replay draws fresh uniforms and noise
        torch.cuda.synchronize()
        rands = bits16(rng.block)
    else:
        rng = RecordingRng(TorchRng())
        m = mcts(world, net, n_nodes=T, rng=rng)
        rands = np.stack([bits16(r) for r in rng.rands])
    assert rands.shape == (T - 1, B, T)
    want = replay_through_oracle(oracle, board, seats, T, rands, bits16(m.decisions.logits), bits16(m.decisions.v))
    assert_search_equals(m, want)
    assert (to_np(m.stats.n)[:, 0] == 2 * (T - 1)).all()


def test_fold_wait_states_under_load():
    """The hot path's serial fold pads each dependent DPP step with ONE wait state where the ISA asks for two (bl_expand.hip).
    The library only does so after bl_selftest() reproduced every prefix total of random chains with it on this device;
    here the same check runs > 1e8 fold steps with four waves per SIMD while an MFMA-heavy GEMM stream keeps the matrix
    pipes and the memory system busy next to it -- no wrong total -- and the ISA-padded variant is exact as well (a
    failure of that one would return a negative code)."""
    from boardlaw_amd import _native
    L = _native.lib()
    a = torch.randn(4096, 4096, device=DEV, dtype=torch.half)
    side = torch.cuda.Stream()
    steps = 0
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(40):
                a @ a
        wrong = L.bl_selftest(_native.stream())
        assert wrong == 0, f'{wrong} wrong prefix totals with the one-wait-state fold'
        steps += 6 * 4096 * sum((1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 54, 63, 64, 65, 80, 81, 96)) * 2   # both variants
    torch.cuda.synchronize()
    assert steps > 1e8
    assert _native.fold_fast(torch.device('cuda')) == (0 if os.environ.get('BL_FOLD_SAFE') else 1)


@pytest.mark.parametrize('A,B', [(81, 4096), (9, 1000), (169, 512), (700, 64)])
def test_draw_actions_is_an_inverse_cdf_sample(A, B):
    """bl_draw_actions (MoveRng's action draw): the picked action has positive probability and brackets u * total in the
    running sum (f64 on the host; 1e-5 of the total for the kernel's f32 summation order); the same row drawn 100k times
    reproduces its probabilities within 5 sigma."""
    from boardlaw_amd import _native
    rng = np.random.default_rng(A)
    p = rng.random((B, A)).astype(np.float32) ** 3 * (rng.random((B, A)) > .4)
    p[:, 0] = np.maximum(p[:, 0], 1e-3) * (np.arange(B) % 2)          # rows whose first action is impossible
    p[np.arange(B), rng.integers(0, A, B)] += .05
    ph = torch.from_numpy(p).to(DEV).half()
    u = torch.from_numpy(rng.random(B).astype(np.float32)).to(DEV)
    actions = torch.empty(B, dtype=torch.long, device=DEV)
    _native.check(_native.lib().bl_draw_actions(ph.data_ptr(), u.data_ptr(), actions.data_ptr(), B, A, _native.stream()))
    pv, a = ph.float().cpu().numpy().astype(np.float64), actions.cpu().numpy()
    cum = pv.cumsum(-1); total = cum[:, -1]; target = u.cpu().numpy().astype(np.float64) * total
    rows = np.arange(B)
    assert (pv[rows, a] > 0).all()
    assert (cum[rows, a] >= target - 1e-5 * total).all() and (cum[rows, a] - pv[rows, a] <= target + 1e-5 * total).all()
    n = 100_000
    row = ph[:1].expand(n, A).contiguous()
    un = torch.rand(n, device=DEV)
    out = torch.empty(n, dtype=torch.long, device=DEV)
    _native.check(_native.lib().bl_draw_actions(row.data_ptr(), un.data_ptr(), out.data_ptr(), n, A, _native.stream()))
    freq = np.bincount(out.cpu().numpy(), minlength=A) / n
    q = pv[0] / pv[0].sum()
    assert (np.abs(freq - q) <= 5 * np.sqrt(q * (1 - q) / n) + 5 / n).all()      # 5 sigma, plus a few counts for the rare ones


@pytest.mark.parametrize('n', [1, 5, 24, 31])
def test_copy_many_clones_every_layout(n):
    """bl_copy_many through _native.clone_many: contiguous tensors of every dtype and odd sizes, row-strided tree slices
    (decisions.logits[:, 0] and friends), misaligned views, empty tensors; more than BL_COPY_MAX items go in chunks."""
    from boardlaw_amd import _native
    g = torch.Generator(device='cuda'); g.manual_seed(n)
    dtypes = [torch.half, torch.float, torch.long, torch.uint8, torch.bool, torch.int16, torch.int32]
    srcs = []
    for k in range(n):
        dt = dtypes[k % len(dtypes)]
        shape = [(4096, 81), (4096,), (333, 7, 7), (1,), (0, 5), (17, 3)][k % 6]
        base = (torch.rand((*shape, 3) if k % 4 == 1 else shape, generator=g, device=DEV) * 100).to(dt)
        if k % 4 == 1:
            base = base[..., 1]                       # elementwise-strided: torch's copy
        if k % 4 == 2 and len(shape) >= 1 and shape[0] > 0:
            tree = (torch.rand((shape[0], 5, *shape[1:]), generator=g, device=DEV) * 100).to(dt)
            base = tree[:, 2]                         # row-strided slice of a (B,T,...) array
        if k % 4 == 3 and base.numel() > 3:
            base = base.reshape(-1)[1:]               # contiguous but not 16-byte aligned
        srcs.append(base)
    out = _native.clone_many(srcs)
    torch.cuda.synchronize()
    assert len(out) == len(srcs)
    for a, b in zip(srcs, out):
        assert b.shape == a.shape and b.dtype == a.dtype and b.is_contiguous() and b.data_ptr() != a.data_ptr() or a.numel() == 0
        assert torch.equal(a, b)
    dsts = [torch.empty_like(t) for t in out]
    _native.copy_many(dsts, out)
    assert all(torch.equal(a, b) for a, b in zip(dsts, out))
    with pytest.raises(_native.NativeError):
        _native.copy_many([torch.empty(4, device=DEV)], [torch.empty(5, device=DEV)])


def test_two_actors_on_two_streams_equal_the_same_calls_on_one():
    """Two independent searches resident on one GPU, each replaying its captured moves on its own stream (what
    bench.py's two_actors_per_gpu region and tools/multi_actor_probe.py time): the kernels of the two actors interleave
    on the device, the host issues the calls in the same order either way, so every decision and every world must be
    identical to the same calls made on a single stream."""
    from boardlaw_amd import hex, networks
    from boardlaw_amd.mcts import MCTSAgent, MoveRng

    def play(concurrent):
        torch.manual_seed(11)
        net = networks.FCModel(hex.Hex.initial(1, 9, device=DEV).obs_space, hex.Hex.initial(1, 9, device=DEV).action_space, width=256, depth=2).to(DEV)
        inf = networks.Inference(net, fused=True)
        worlds = []
        for i in range(2):
            w = hex.Hex.initial(1024, 9, device=DEV)
            for _ in range(10 + i):
                r = torch.rand(w.valid.shape, device=DEV) * w.valid
                w, _ = w.step(r.argmax(-1), check=False)
            worlds.append(w)
        gens = [torch.Generator(device=DEV) for _ in range(2)]
        for i, g in enumerate(gens):
            g.manual_seed(100 + i)
        # a generator per actor: two captured graphs on one generator race for its offset tensor (MoveRng.__init__)
        agents = [MCTSAgent(inf, n_nodes=32, graph=True, rng=MoveRng(generator=gens[i])) for i in range(2)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()] if concurrent else [torch.cuda.current_stream()] * 2
        torch.cuda.synchronize()
        out = []
        for move in range(4):
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    d, worlds[i], tr = agents[i].play(worlds[i])
                out.append((d, worlds[i], tr))
        torch.cuda.synchronize()
        return out

    a, b = play(True), play(False)
    for (da, wa, ta), (db, wb, tb) in zip(a, b):
        for k in ('logits', 'prior', 'v', 'actions', 'n_leaves', 'n_sims'):
            assert torch.equal(da[k], db[k]) or (da[k].dtype.is_floating_point and np.array_equal(bits16(da[k]), bits16(db[k]))), k
        assert torch.equal(wa.board, wb.board) and torch.equal(wa.seats, wb.seats)
        assert torch.equal(ta.terminal, tb.terminal) and torch.equal(ta.rewards, tb.rewards)


def test_inference_plan_selection():
    """Inference picks the one-kernel plan by what bounds it (one workgroup streams all the weights): always for 512x4,
    for 1024x8 only when the batch fills 96 workgroups -- below that a launch per Linear (bl_mlp_layers_f16); MCTS.simulate
    follows (fused finish or raw() + bl_sim_finish)."""
    from boardlaw_amd import hex, networks
    from boardlaw_amd.mcts import mcts
    w = hex.Hex.initial(8, 9, device=DEV)
    small = networks.Inference(networks.FCModel(w.obs_space, w.action_space, width=512, depth=4).to(DEV), fused=True); small.refresh()
    wide = networks.Inference(networks.FCModel(w.obs_space, w.action_space, width=1024, depth=8).to(DEV), fused=True); wide.refresh()
    assert small.prefers_fused(1) and small.prefers_fused(4096) and small.fused_params(8) is not None
    assert not wide.prefers_fused(1024) and not wide.prefers_fused(2048) and wide.prefers_fused(3072) and wide.prefers_fused(4096)
    assert wide.fused_params(1024) is None and wide.fused_params(4096) is not None and wide.fused_params() is not None
    assert not networks.Inference(small.model, fused=False).prefers_fused(4096)
    torch.manual_seed(0)
    a = mcts(w, wide, n_nodes=8)                       # 8 rows of 1024x8: a launch per Linear + bl_sim_finish
    wide.FUSED_MIN_TILES = 0
    torch.manual_seed(0)
    b = mcts(w, wide, n_nodes=8)                       # the same search through bl_sim_infer_finish
    assert (to_np(a.stats.n)[:, 0] == 14).all() and (to_np(b.stats.n)[:, 0] == 14).all()


@pytest.mark.parametrize('S,B,T', [(9, 512, 64), (5, 100, 16), (13, 64, 96)])
def test_lazy_reset_equals_the_eager_reset(S, B, T):
    """bl_tune_t.lazy_init (what mcts() uses): the (B,T,A) arrays get their reset values slot by slot from the simulations
    instead of from bl_sim_init's fills.  After a whole search every array -- including the slots of simulations that re-visited
    a terminal node and created nothing (children -1, logits NaN, the root world) -- equals the eagerly reset search's, from
    buffers that were poisoned beforehand."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS, TorchRng
    torch.manual_seed(S * 100 + T)
    world = Hex.initial(B, S)
    for k in range(S * S - 4):                                  # nearly full boards: many terminal re-visits
        v = world.valid
        world, _ = world.step((torch.rand(v.shape, device=DEV) * v).argmax(-1))
    net = networks.Inference(networks.FCModel(world.obs_space, world.action_space, width=256, depth=2).to(DEV), fused=True)
    runs = []
    for lazy in (False, True):
        torch.manual_seed(9)
        m = MCTS(world, n_nodes=T, rng=TorchRng(), lazy=lazy, obs_half=True)
        if lazy:      # the lazy reset leaves slots >= 1 of the big arrays untouched: poison them, nothing of it may survive the search
            m.tree.children[:, 1:] = 7; m.decisions.logits[:, 1:] = 7.; m.worlds.board[:, 1:] = 7
        m.initialize(net)
        for _ in range(T - 1):
            m.simulate(net)
        runs.append(m)
    a, b = runs
    unused = (to_np(a.tree.parents)[:, 1:] == -1).sum()
    for name, x, y in [('children', a.tree.children, b.tree.children), ('logits', a.decisions.logits, b.decisions.logits),
                       ('boards', a.worlds.board, b.worlds.board), ('seats', a.worlds.seats, b.worlds.seats), ('n', a.stats.n, b.stats.n),
                       ('w', a.stats.w, b.stats.w), ('parents', a.tree.parents, b.tree.parents), ('v', a.decisions.v, b.decisions.v)]:
        assert np.array_equal(to_np(x) if x.dtype != torch.half else bits16(x), to_np(y) if y.dtype != torch.half else bits16(y)), name
    assert unused > 0, 'the case must contain simulations that created no node'
