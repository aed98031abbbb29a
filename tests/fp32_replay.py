"""End-to-end seeded equivalence of the exact mode (networks.Inference(precision='fp32')) -- TEST INFRASTRUCTURE.

north_star asks for "visit counts and chosen actions exactly under a fixed RNG seed".  The tree arithmetic on the GPU is the
reference CPU path's bit for bit GIVEN the leaf evaluations (tests/test_gpu_parity.py); what is left is the leaf evaluations
themselves.  The reference's recorded runs evaluate leaves in f32 (`torch.cuda.amp.autocast` at boardlaw/mcts/__init__.py:131-134
is a no-op on the CPU) and round only the stores (`.half()`, :135-136); the exact mode does the same on the GPU, so what can still
differ is the f32 GEMM summation order (hipBLASLt / bl_root_mlp_f32 against the CPU's sgemm) where it flips a binary16 rounding.

Two sources of "what the reference computes":
  * the committed fixtures (tests/golden/search_*.npz: whole reference searches with their network's state_dict, Dirichlet draw and
    uniforms, made by running the reference -- tests/golden/make_golden.py);
  * `standin_search`: the reference's search re-enacted on the host for ANY seed -- the restated FCModel in f32 on the CPU (bit for
    bit the reference network on the recorded inputs: tests/test_network_golden.py), `dirichlet_noise` on the CPU (bit for bit the
    reference root row: same file), and every native call through the C oracle (bit for bit the reference's cpu.cpp:
    tests/test_oracle.py).  tests/test_fp32_standin.py pins it: on every fixture it reproduces the reference's recorded tree exactly.
"""
import numpy as np
import torch

import oracle_lib
from gpu_util import ReplayRng, bits16, to_np


class Recorded:
    """A reference search as the fixtures store it (m{move}_* arrays), with OracleSearch's attribute names."""

    def __init__(self, g, move):
        p = f'm{move}_'
        self.children, self.parents, self.relation = g[p + 'children'], g[p + 'parents'], g[p + 'relation']
        self.n, self.w = g[p + 'n'], g[p + 'w']
        self.logits, self.v = g[p + 'tree_logits'], g[p + 'tree_v']
        self.boards, self.seats = g[p + 'boards'], g[p + 'seats']
        self._root_probs = g[p + 'root_probs']

    def root_probs(self):
        return self._root_probs


class _World:
    def __init__(self, obs, valid, seats):
        self.obs, self.valid, self.seats = obs, valid, seats


def _host_world(orc, board, seats):
    obs = orc.hex_observe(np.ascontiguousarray(board), np.ascontiguousarray(seats.astype(np.int32)))
    return _World(torch.from_numpy(obs), torch.from_numpy((obs == 0).all(-1).reshape(board.shape[0], -1)), torch.from_numpy(seats.astype(np.int32)))


def _bits(t):
    return t.detach().contiguous().half().view(torch.int16).numpy().view(np.uint16)


def standin_search(orc, net_cpu, board, seats, T, dirichlet, rands_bits, noise_eps=.25, alpha_scale=10):
    """The reference's `mcts(worlds, network, n_nodes=T)` (boardlaw/mcts/__init__.py:154-160) on the host: f32 network on the CPU,
    the oracle's kernels, the caller's Dirichlet draw (B,A) f32 and uniforms (T-1,B,T) f16 bits.  Returns an OracleSearch."""
    from test_network_golden import FixedDraw
    from boardlaw_amd.mcts import dirichlet_noise
    s = oracle_lib.OracleSearch(orc, board, seats, T)
    with torch.no_grad():
        w = _host_world(orc, board, seats)
        d = net_cpu(w)                                                                      # mcts/__init__.py:72-76: f32
        noised = dirichlet_noise(d.logits, w.valid, noise_eps, alpha_scale, FixedDraw(torch.from_numpy(np.asarray(dirichlet, np.float32))))
        s.initialize(_bits(noised), _bits(d.v))                                             # :77-78: stores round to f16
        for i in range(T - 1):
            parents, actions = s.descend(np.ascontiguousarray(rands_bits[i]))
            leaves, nb, ns = s.expand(parents, actions)
            ld = net_cpu(_host_world(orc, nb, ns))                                          # :131-134: autocast is a no-op on the CPU
            s.finish(leaves, _bits(ld.logits), _bits(ld.v))                                 # :135-136: `.half()`
    return s


def gpu_search(net_cuda, board, seats, T, dirichlet, rands_bits, precision='fp32', fused=True, noise_eps=.25):
    """The product's fused search on the GPU with the same draw and uniforms; leaves evaluated in `precision`."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTS

    class Rng(ReplayRng):
        def dirichlet(self, alpha, shape):
            return torch.from_numpy(np.asarray(dirichlet, np.float32)).cuda()
    inf = networks.Inference(net_cuda, fused=fused, precision=precision)
    world = Hex(board=torch.from_numpy(np.ascontiguousarray(board)).cuda(), seats=torch.from_numpy(np.ascontiguousarray(seats.astype(np.int32))).cuda())
    m = MCTS(world, n_nodes=T, rng=Rng(rands_bits, 'cuda'), obs_half=inf.wants_half_obs, noise_eps=noise_eps)
    m.initialize(inf)
    for _ in range(T - 1):
        m.simulate(inf)
    return m


def _ulp16(a_bits, b_bits):
    def key(x):
        x = x.astype(np.int32)
        return np.where(x & 0x8000, -(x & 0x7fff), x & 0x7fff)
    return np.abs(key(a_bits) - key(b_bits))


def root_visits(children, n):
    """(B,A) visit counts of the root's children (0 where the action has no child)."""
    c = children[:, 0].astype(np.int64)
    return np.where(c >= 0, np.take_along_axis(n.astype(np.int64), np.maximum(c, 0), 1), 0)


def compare(m, want):
    """How far a GPU search `m` is from the reference search `want`: per-env fractions (1.0 = the reference's run)."""
    ch, pa, re_, n, w = (to_np(x) for x in (m.tree.children, m.tree.parents, m.tree.relation, m.stats.n, m.stats.w))
    lg, v = bits16(m.decisions.logits), bits16(m.decisions.v)
    B, T = n.shape
    same_shape = (ch == want.children).all((1, 2)) & (pa == want.parents).all(1) & (re_ == want.relation).all(1)     # the same tree
    same_stats = same_shape & (n == want.n).all(1) & (w == want.w).all((1, 2))                                    # ... visit counts, value sums
    same_all = same_stats & (lg == want.logits).all((1, 2)) & (v == want.v).all((1, 2))                            # ... and every stored evaluation
    mine_v, ref_v = root_visits(ch, n), root_visits(want.children, want.n)
    same_visits = (mine_v == ref_v).all(-1)
    tv = 0.5 * np.abs(mine_v / np.maximum(mine_v.sum(-1, keepdims=True), 1) - ref_v / np.maximum(ref_v.sum(-1, keepdims=True), 1)).sum(-1)
    # stored evaluations of the nodes both searches created at the same place (same parent chain is implied by same_shape; elsewhere
    # slot s may hold different positions, so the per-logit statistics are taken over envs whose trees agree)
    rows = same_shape[:, None] & (want.parents != -1)
    rows[:, 0] = True
    fin = rows[:, :, None] & (want.logits != 0xfc00) & ((want.logits & 0x7c00) != 0x7c00)
    du = _ulp16(lg[fin], want.logits[fin]) if fin.any() else np.zeros(1, np.int64)
    rp = bits16(m.root_probs())
    return dict(envs=B, first_action=float((re_[:, 1] == want.relation[:, 1]).mean()) if T > 1 else 1.0,
                root_visits_equal=float(same_visits.mean()), root_visits_tv=float(tv.mean()),
                tree_equal=float(same_shape.mean()), tree_and_stats_equal=float(same_stats.mean()), everything_equal=float(same_all.mean()),
                root_probs_equal=float((rp == want.root_probs()).all(-1).mean()),
                argmax_action_equal=float((_argmax_visits(mine_v) == _argmax_visits(ref_v)).mean()),
                logits_compared=int(fin.sum()), logits_equal=float((du == 0).mean()), logits_max_ulp=int(du.max()))


def _argmax_visits(v):
    return v.argmax(-1)


def fixture_case(g, move):
    """(board, seats, T, dirichlet, rands) of recorded move `move` of a search fixture."""
    p = f'm{move}_'
    return (np.ascontiguousarray(g[p + 'boards'][:, 0]), np.ascontiguousarray(g[p + 'seats'][:, 0]).astype(np.int32), int(g['meta'][2]),
            g[p + 'dirichlet'], g[p + 'rands'])
