"""The exact mode end to end: networks.Inference(precision='fp32') -- leaves evaluated like the reference's CPU
runs evaluate them (f32 network and heads, f16 stores; boardlaw/mcts/__init__.py:131-136) -- against whole searches the reference
recorded, every recorded move of every fixture that carries its network's state_dict.  See tests/fp32_replay.py.

CPU (`-m "not gpu"`): the host stand-in for the reference (restated FCModel on the CPU + dirichlet_noise + the C oracle) reproduces
every recorded search bit for bit -- which licenses it as the reference for seeds that have no fixture (tools/fp32_fuzz.py).
GPU: the product's fused search with fp32 leaves against the recorded searches; thresholds are what was observed on an MI355X
(profiles/r05_fp32_leaves.txt), stated per assertion."""
import numpy as np
import pytest
import torch

import fp32_replay
from test_oracle import SEARCHES, gold
from test_network_golden import reference_network

ALL = SEARCHES + ['search_9x9_w512.npz']


@pytest.mark.parametrize('name', ALL)
def test_host_standin_reproduces_the_recorded_reference_searches(oracle, name):
    """standin_search == the reference's own recorded search: every tree array, visit count, value sum, stored evaluation, board
    and the root distribution, for every recorded move (tolerance 0)."""
    g = gold(name)
    net = reference_network(g)
    for move in range(int(g['meta'][5])):
        board, seats, T, draw, rands = fp32_replay.fixture_case(g, move)
        s = fp32_replay.standin_search(oracle, net, board, seats, T, draw, rands)
        want = fp32_replay.Recorded(g, move)
        for k in ('children', 'parents', 'relation', 'n', 'w', 'logits', 'v', 'boards', 'seats'):
            assert np.array_equal(getattr(s, k), getattr(want, k)), (name, move, k)
        assert np.array_equal(s.root_probs(), want.root_probs()), (name, move)


# what an MI355X gave (profiles/r05_fp32_leaves.txt); the fp16 plan's figures on the same fixtures are in the same file
FLOORS = dict(first_action=0.99, root_visits_equal=0.95, tree_and_stats_equal=0.90, logits_equal=0.995)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ALL)
def test_fp32_leaves_replay_the_reference_search(name):
    """The product's fused search (bl_sim_plant_root, then bl_sim_expand -> fp32 Linears -> bl_sim_finish_f32 per simulation) with
    the reference's parameters, Dirichlet draw and uniforms, all recorded moves.  What must hold, over the envs of all moves:
      * the first simulation picks the reference's action in >= 99 % of the envs;
      * the root children's final visit counts are the reference's in >= 95 % of the envs;
      * the WHOLE tree with every visit count and value sum is the reference's in >= 90 % of the envs;
      * of the stored f16 logits in trees that agree, >= 99.5 % are the reference's bits, none further than 2 ulp
        (two f32 GEMM summation orders rounding to different binary16 neighbours)."""
    g = gold(name)
    net = reference_network(g, 'cuda')
    width = int(g['meta'][3])
    rows, weights = [], []
    for move in range(int(g['meta'][5])):
        board, seats, T, draw, rands = fp32_replay.fixture_case(g, move)
        m = fp32_replay.gpu_search(net, board, seats, T, draw, rands, precision='fp32', fused=(width % 128 == 0))
        r = fp32_replay.compare(m, fp32_replay.Recorded(g, move))
        rows.append(r); weights.append(r['envs'])
    mean = lambda k: float(np.average([r[k] for r in rows], weights=weights))
    for k, floor in FLOORS.items():
        assert mean(k) >= floor, (name, k, mean(k), rows)
    assert max(r['logits_max_ulp'] for r in rows) <= 2, (name, rows)
