"""Host-side containers and fixtures (CPU): arrdict/dotdict behaviours the worlds, agents and trainer rely on
(rebar/arrdict.py:33-48,123-148,199-222; rebar/dotdict.py:17-29), toy worlds, agent state dicts."""
import numpy as np
import pytest
import torch

from boardlaw_amd import arrdict, validation


def test_arrdict_setitem_fans_out():                       # rebar/arrdict.py:199-208
    d = arrdict.arrdict(a=np.array([0, 1]), b=arrdict.arrdict(c=np.array([10, 11])))
    d[0] = d[1]
    assert d.a.tolist() == [1, 1] and d.b.c.tolist() == [11, 11]
    with pytest.raises(ValueError):
        d[0] = 5


def test_namedarrtuple_is_closed():                        # rebar/arrdict.py:215-222
    D = arrdict.namedarrtuple('T', ('one', 'two'))
    d = D(one=1, two=2)
    d['one'] = -1
    with pytest.raises(KeyError):
        d['three'] = 3
    with pytest.raises(KeyError):
        del d['one']
    with pytest.raises(KeyError):
        D(one=1)


def test_attribute_delegation_indexing_and_arithmetic():
    d = arrdict.arrdict(x=torch.arange(6.).reshape(3, 2), y=arrdict.arrdict(z=torch.ones(3, dtype=torch.long)))
    assert d.shape.x == (3, 2) and d.y.z is d['y']['z']
    assert d[1:].x.shape == (2, 2) and d[torch.tensor([True, False, True])].y.z.shape == (2,)
    c = d.clone(); c.x += 1
    assert d.x[0, 0] == 0 and c.x[0, 0] == 1
    assert (d + 1).x[0, 0] == 1 and (d * d).x[2, 1] == 25 and (d == d).y.z.all()
    assert d.float().y.z.dtype == torch.float
    s = arrdict.stack([d, d], 1)
    assert s.x.shape == (3, 2, 2) and arrdict.cat([d, d]).y.z.shape == (6,)
    assert arrdict.leaves(d)[1] is d.y.z and 'x' in dir(d)
    with pytest.raises(AttributeError):
        d.nonexistent_attribute
    assert 'arrdict' in repr(d) and 'Tensor((3, 2)' in repr(d)
    assert arrdict.numpyify(d).x.dtype == np.float32 and arrdict.torchify(arrdict.arrdict(a=np.zeros(2))).a.dtype == torch.float


def test_mapping_and_starmapping():
    d = arrdict.dotdict(a=1, b=arrdict.dotdict(c=2))
    assert d.map(lambda v, k: v + k, 10).b.c == 12
    assert arrdict.starmapping(int.__add__)(d, d).a == 2


def test_toy_worlds_planted_values():
    """validation.All: the planted value is the probability of finishing with all ones under uniform play."""
    w = validation.All.initial(n_envs=3, n_seats=2, length=3, device='cpu')
    assert torch.allclose(w.v, torch.full((3, 2), 1 / 8))
    for _ in range(6):                                     # boardlaw/validation.py:202-210 (test_all_ones)
        w, t = w.step((w.seats % 2).long())
    assert t.terminal.all() and (t.rewards[:, 0] == 0).all() and (t.rewards[:, 1] == 1).all()
    wl = validation.WinnerLoser.initial(2, device='cpu')
    wl2, t = wl.step(torch.zeros(2, dtype=torch.long))
    assert (wl2.seats == 1).all() and not t.terminal.any()


def test_agent_state_dict_roundtrip_and_dummy_agent():
    from boardlaw_amd import networks, heads
    from boardlaw_amd.mcts import MCTSAgent, DummyAgent
    net = networks.FCModel(heads.Tensor((1,)), heads.Masked(2), width=8, depth=1)
    agent = MCTSAgent(net, n_nodes=12, c_puct=.5)
    sd = agent.state_dict()
    assert sd['kwargs.n_nodes'] == 12 and 'network.value.core.weight' in sd
    other = MCTSAgent(networks.FCModel(heads.Tensor((1,)), heads.Masked(2), width=8, depth=1))
    other.load_state_dict(sd)
    assert other.kwargs == {'n_nodes': 12, 'c_puct': .5}
    # this build's own options never enter the reference's checkpoint format (its load_state_dict feeds kwargs.* to MCTS())
    from boardlaw_amd.mcts import MoveRng
    mine = MCTSAgent(net, graph=True, n_nodes=12, rng=MoveRng(), fuse_finish=False, obs_half=True)
    assert sorted(k for k in mine.state_dict() if k.startswith('kwargs.')) == ['kwargs.n_nodes']
    # captured moves are keyed by the kwargs they were captured with and evicted least-recently-used by bytes
    class G:
        def __init__(self, n): self.nbytes = n
    mine.GRAPH_CACHE_BYTES = 100
    a = mine._graphed(('a',), lambda: G(60)); b = mine._graphed(('b',), lambda: G(30))
    assert mine._graphed(('a',), lambda: G(60)) is a                   # hit; 'a' is now the most recent
    c = mine._graphed(('c',), lambda: G(30))                             # 120 > 100: 'b', the least recent, goes
    assert [k[0] for k in mine._graphs] == ['a', 'c'] and mine._graphed(('b',), lambda: G(30)) is not b
    mine.kwargs['n_nodes'] = 24                                          # the reference mutates kwargs in place
    assert mine._graphed(('a',), lambda: G(60)) is not a
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), other.network.state_dict().values()))
    w = validation.All.initial(n_envs=4, length=3, device='cpu')
    d = DummyAgent(net)(w, eval=True)
    assert d.actions.shape == (4,) and (d.n_sims == 0).all() and (d.n_leaves == 1).all()


def test_dirichlet_noise_masks_and_renormalises():
    from boardlaw_amd.mcts import dirichlet_noise
    torch.manual_seed(0)
    logits = torch.log_softmax(torch.randn(5, 9), -1)
    valid = torch.rand(5, 9) > .3
    valid[:, 0] = True
    logits = logits.masked_fill(~valid, -np.inf)
    logits = torch.log_softmax(logits, -1)
    out = dirichlet_noise(logits, valid, .25)
    assert torch.isinf(out[~valid]).all() and torch.isfinite(out[valid]).all()
    assert torch.allclose(out.exp().sum(-1), torch.ones(5), atol=1e-5)
    assert torch.equal(dirichlet_noise(logits, valid, 0.).exp().argmax(-1), logits.argmax(-1))


def test_roctx_ranges_follow_the_reference_switch(monkeypatch):
    """rebar/profiling.py:15-28: markers only when the switch is set at decoration time; otherwise the function itself."""
    from boardlaw_amd import profiling
    def f(x):
        return x + 1
    monkeypatch.delenv('EMIT_ROCTX', raising=False); monkeypatch.delenv('EMIT_NVTX', raising=False)
    assert profiling.roctx(f) is f
    calls = []
    monkeypatch.setenv('EMIT_NVTX', '1')
    monkeypatch.setattr(profiling, 'push', lambda name: calls.append(('push', name)))
    monkeypatch.setattr(profiling, 'pop', lambda: calls.append(('pop',)))
    g = profiling.roctx(f)
    assert g is not f and g(1) == 2 and g.__name__ == 'f'
    assert calls[0][0] == 'push' and calls[0][1].endswith('f') and calls[1] == ('pop',)
    def boom():
        raise ValueError
    h = profiling.nvtx(boom)
    try:
        h()
    except ValueError:
        pass
    assert calls[-1] == ('pop',)          # the range is closed when the call raises
    monkeypatch.undo()
    lib = profiling._roctx()              # the ROCm image ships libroctx64: the real push/pop work without a GPU
    assert profiling.push('boardlaw_amd.test') >= 0 and profiling.pop() >= 0
