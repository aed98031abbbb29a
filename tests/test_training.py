"""Actor/learner pieces on CPU: targets against the reference's golden values and known answers, the loss,
and the 2-rank gradient all-reduce (gloo) against a single-process full-batch step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_reward_to_go_reference_known_answers():
    """boardlaw/learning.py:83-94"""
    from boardlaw_amd import learning
    r, v = torch.tensor([1., 2., 3.]), torch.tensor([4., 5., 6.])
    out = learning.reward_to_go(r.clone(), v.clone(), torch.tensor([False, False, False]), 1.)
    assert out.tolist() == [9., 8., 6.]
    out = learning.reward_to_go(r.clone(), v.clone(), torch.tensor([False, True, False]), 1.)
    assert out.tolist() == [3., 2., 6.]


def test_targets_match_reference_golden():
    from boardlaw_amd import learning
    g = np.load(os.path.join(GOLD, 'learning.npz'))
    reward, value, terminal = (torch.from_numpy(g[k]) for k in ('reward', 'value', 'terminal'))
    assert np.array_equal(learning.reward_to_go(reward.clone(), value.clone(), terminal, 1.).numpy(), g['reward_to_go'])
    assert np.array_equal(learning.present_value(reward[:-1], value.clone(), terminal, .9).numpy(), g['present_value_09'])


def _toy_batch(n, seed):
    from boardlaw_amd import arrdict, validation
    torch.manual_seed(seed)
    worlds = validation.All.initial(n_envs=n, n_seats=1, length=4, device='cpu')
    logits = torch.log_softmax(torch.randn(n, 2), -1)
    return arrdict.arrdict(worlds=worlds, decisions=arrdict.arrdict(logits=logits.half(), v=torch.randn(n, 1).half()),
                           reward_to_go=torch.randn(n, 1).half())


def _toy_net(seed):
    from boardlaw_amd import networks, heads
    torch.manual_seed(seed)
    net = networks.FCModel(heads.Tensor((1,)), heads.Masked(2), width=8, depth=2)
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim == 0:
                p.fill_(.4)
    return net


def test_losses_formula():
    """main.py:78-92: -sum(pi0 * log pi) mean + MSE(reward_to_go, v)."""
    from boardlaw_amd import training
    net, batch = _toy_net(0), _toy_batch(32, 1)
    pl, vl = training.losses(net, batch)
    d = net(batch.worlds)
    want_pl = -(batch.decisions.logits.float().exp() * d.logits).sum(-1).mean()
    want_vl = ((batch.reward_to_go - d.v) ** 2).mean()
    assert torch.allclose(pl, want_pl) and torch.allclose(vl, want_vl)


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out, bucketed=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from boardlaw_amd import parallel, training
    parallel.init('gloo')
    net, full = _toy_net(0), _toy_batch(64, 1)
    shard = full[parallel.shard(64, rank, world)]
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    scaler = torch.amp.GradScaler('cuda', enabled=False)
    bucket = parallel.GradientBucket(net) if bucketed else None
    training.optimize(net, scaler, opt, shard, bucket=bucket)
    if bucketed:
        # a second step through the same bucket: the views survive a step (bucket.zero(), not opt.zero_grad())
        assert bucket.intact()
        training.optimize(net, scaler, opt, shard, bucket=bucket)
        assert bucket.intact()
    out.put((rank, torch.cat([p.detach().flatten() for p in net.parameters()]).numpy()))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('bucketed', [False, True])
def test_two_rank_gradient_allreduce_equals_full_batch_step(bucketed):
    """bucketed: parallel.GradientBucket -- the gradients as views of one persistent flat buffer, all-reduced in place (what
    training.run uses with several ranks) -- against the same full-batch step(s) in one process."""
    from boardlaw_amd import training
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, bucketed)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60); assert p.exitcode == 0
    net, full = _toy_net(0), _toy_batch(64, 1)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for _ in range(2 if bucketed else 1):
        training.optimize(net, torch.amp.GradScaler('cuda', enabled=False), opt, full, sync_gradients=False)
    want = torch.cat([p.detach().flatten() for p in net.parameters()]).numpy()
    assert np.allclose(res[0], res[1], atol=0) and np.allclose(res[0], want, atol=2e-6)


def test_train_bench_two_ranks_dry_run():
    """tools/train_bench.py --gpus 2 (config 4's launcher: one process per GPU, the learner's gradients through ONE all-reduce of the
    persistent bucket) as a CPU dry run over gloo: the self-spawn, the rendezvous, the bucket all-reduce (checked against the known
    mean inside the script) and the single JSON line with both ranks seen."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'train_bench.py'), '--gpus', '2', '--steps', '3'], env={**env, 'TRAIN_DRY': '1'},
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['per_rank_values'] == [1.0, 2.0] and d['dry_run']


def test_as_chunk_shapes():
    from boardlaw_amd import arrdict, training, validation
    B, L = 6, 5
    buf = []
    worlds = validation.All.initial(n_envs=B, n_seats=1, length=3, device='cpu')
    for t in range(L):
        new, trans = worlds.step(torch.ones(B, dtype=torch.long))
        buf.append(arrdict.arrdict(worlds=worlds, decisions=arrdict.arrdict(v=torch.zeros(B, 1).half(), logits=torch.zeros(B, 2).half()),
                                   transitions=arrdict.arrdict(terminal=trans.terminal, rewards=trans.rewards.half())))
        worlds = new
    chunk, rest = training.as_chunk(buf, B)
    assert chunk.reward_to_go.shape == (L, B, 1) and chunk.reward_to_go.dtype == torch.half and len(rest) == L - 1
    # the game ends with +1 on step 3 (length 3, always playing 1): steps 0..2 see reward-to-go 1
    assert (chunk.reward_to_go[:3].float() == 1).all()


def test_arena_evaluate_reference_known_answer():
    """boardlaw/arena/common.py:108-115 (test_evaluate): on WinnerLoser the first seat always wins."""
    from boardlaw_amd import arena, validation
    worlds = validation.WinnerLoser.initial(4, device='cpu')
    results = arena.evaluate(worlds, {'one': validation.RandomAgent(), 'two': validation.RandomAgent()})
    assert results[0].wins == (2., 0.) and results[1].wins == (2., 0.)
    assert results[0].names == ('one', 'two') and results[1].names == ('two', 'one')


def test_rollout_on_toy_world():
    from boardlaw_amd import arena, validation
    worlds = validation.WinnerLoser.initial(6, device='cpu')
    trace = arena.rollout(worlds, [validation.RandomAgent(), validation.RandomAgent()], n_steps=4)
    assert trace.actions.shape == (4, 6) and trace.transitions.terminal.shape == (4, 6)
    assert trace.transitions.terminal[1::2].all() and not trace.transitions.terminal[0::2].any()
    # per-agent decisions widened to every env: agent 0 moves on even steps, agent 1 on odd ones (analysis.py:28-45)
    d0, d1 = trace.decisions['0'], trace.decisions['1']
    assert d0.mask.shape == (4, 6) and d0.mask[0::2].all() and not d0.mask[1::2].any()
    assert d1.mask[1::2].all() and not d1.mask[0::2].any()
    assert (d0.actions[0::2] == trace.actions[0::2]).all() and (d0.actions[1::2] == -1).all()
    assert torch.isnan(d1.logits[0::2]).all() and not torch.isnan(d1.logits[1::2]).any()
    assert d0.v.shape == (4, 6, 2)


def test_rollout_stop_conditions():
    from boardlaw_amd import analysis, validation
    agents = [validation.RandomAgent(), validation.RandomAgent()]
    worlds = validation.WinnerLoser.initial(3, device='cpu')
    assert analysis.rollout(worlds, agents, n_trajs=3).actions.shape[0] == 2        # 3 games end on the 2nd step
    assert analysis.rollout(worlds, agents, n_reps=2).actions.shape[0] == 4         # every env has finished twice
    with pytest.raises(AssertionError):
        analysis.rollout(worlds, agents)
    with pytest.raises(AssertionError):
        analysis.rollout(worlds, agents, n_steps=1, n_trajs=1)


def test_combine_decisions_missing_agent_steps():
    """An agent that does not act at some step still gets a frame there: all blank, mask False."""
    from boardlaw_amd import analysis, arrdict
    m0 = torch.tensor([True, False, True])
    d = arrdict.arrdict(actions=torch.tensor([4, 5]), logits=torch.zeros(2, 3))
    out = analysis.combine_decisions([{0: d}, {}], [{0: m0}, {}])
    assert out['0'].actions.tolist() == [[4, -1, 5], [-1, -1, -1]]
    assert out['0'].mask.tolist() == [[True, False, True], [False, False, False]]
    assert torch.isnan(out['0'].logits[1]).all()


# ------------------------------------------------------------------------------------------------ arena: all-vs-all, checkpoints
class _MockAgent:                                              # arena/neural.py:343-350
    def __init__(self, id):
        self.id = id

    def __call__(self, world):
        from boardlaw_amd import arrdict
        return arrdict.arrdict(actions=torch.full((world.n_envs,), self.id, dtype=torch.long, device=world.device))


def _mock_game():
    from boardlaw_amd import arrdict

    class MockGame(arrdict.namedarrtuple('MockGame', fields=('count', 'history'))):      # arena/neural.py:352-386
        @classmethod
        def initial(cls, n_envs=1, length=4, device='cpu'):
            return cls(history=torch.full((n_envs, length), -1, dtype=torch.long, device=device),
                       count=torch.zeros((n_envs,), dtype=torch.long, device=device))

        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            if isinstance(self['count'], torch.Tensor):
                self.n_envs, self.device, self.n_seats = self['count'].shape[0], self['count'].device, 2

        @property
        def seats(self):
            return self.count % 2

        def step(self, actions):
            history = self.history.clone()
            history.scatter_(1, self.count[:, None], actions[:, None])
            count = self.count + 1
            terminal = count == history.shape[1]
            done = [h for h in history[terminal]]
            count[terminal] = 0
            return type(self)(count=count, history=history), arrdict.arrdict(terminal=terminal), done
    return MockGame


def test_tracker_plays_every_ordered_pair():
    """arena/neural.py:388-417 (test_tracker): 16 agents, 4 games per ordered pair, every finished game was played by
    exactly the two agents of its pair, in seat order."""
    import pandas as pd
    from collections import Counter
    from boardlaw_amd import arena
    n_envs_per, length = 4, 8
    agents = {i: _MockAgent(i) for i in range(16)}
    tracker = arena.Tracker(n_envs_per, pd.DataFrame(0, list(agents), list(agents)), device='cpu')
    assert tracker.n_envs == 16 * 15 * n_envs_per
    worlds = _mock_game().initial(tracker.n_envs, length=length)
    games = []
    while not tracker.finished():
        name, mask, pairs = tracker.suggest(worlds.seats)
        assert (pairs.gather(1, worlds.seats[mask][:, None]).squeeze(1) == name).all()
        decisions = agents[name](worlds[mask])
        worlds[mask], transitions, done = worlds[mask].step(decisions.actions)
        games.extend(done)
        tracker.update(transitions.terminal, mask)
    counts = Counter(tuple(int(x) for x in g[:2]) for g in games)
    assert all(len(set(g.tolist())) <= 2 for g in games)
    assert len(counts) == 16 * 15 and set(counts.values()) == {n_envs_per}
    assert tracker.report() == (tracker.n_envs, 0)
    # games already on record are not replayed
    played = np.zeros((3, 3), int); played[0, 1] = 4; played[2, 0] = 1
    t2 = arena.Tracker(4, played, names=['a', 'b', 'c'], device='cpu')
    assert t2.n_envs == 6 * 4 - 5 and not ((t2.live[:, 0] == 0) & (t2.live[:, 1] == 1)).any()


def test_checkpoint_in_the_reference_wire_format(tmp_path):
    """tests/golden/snapshot_5x5.pt was written by the reference (its MCTSAgent.state_dict inside the trainer's torch.save,
    mcts/__init__.py:231-241, main.py:155-160): it loads into an agent here, the rebuilt network reproduces the outputs the
    reference's network gave, and a checkpoint written here has the same keys."""
    from boardlaw_amd import storage
    here = os.path.dirname(os.path.abspath(__file__))
    ck = storage.load(os.path.join(here, 'golden', 'snapshot_5x5.pt'))
    agent = storage.agent_from_checkpoint(ck)
    assert agent.kwargs == {'n_nodes': 16, 'c_puct': 1 / 8}
    want = np.load(os.path.join(here, 'golden', 'snapshot_5x5_outputs.npz'))

    class W:
        obs = torch.zeros(4, 5, 5, 2); valid = torch.ones(4, 25, dtype=torch.bool); seats = torch.zeros(4, dtype=torch.int)
    with torch.no_grad():
        d = agent.network(W)
    assert np.array_equal(d.logits.numpy().view(np.uint32), want['logits'].view(np.uint32))
    assert np.array_equal(d.v.numpy().view(np.uint32), want['v'].view(np.uint32))
    opt = torch.optim.Adam(agent.network.parameters(), lr=1e-3)
    storage.save(tmp_path / 'snap.pkl', agent=agent, opt=opt)
    mine = storage.load(tmp_path / 'snap.pkl')
    assert set(mine) == {'agent', 'opt'} and set(mine['agent']) == set(ck['agent'])
    assert all(torch.equal(mine['agent'][k], ck['agent'][k]) if torch.is_tensor(ck['agent'][k]) else mine['agent'][k] == ck['agent'][k]
               for k in ck['agent'])
    nested = storage.expand(ck['agent'], 1)
    assert set(nested) == {'network', 'kwargs'} and storage.collapse(nested, 1).keys() == ck['agent'].keys()


@pytest.mark.gpu
def test_rollout_traces_match_the_reference():
    """analysis.rollout (analysis.py:47-87) on the GPU against traces the reference produced from the same starting positions
    with the same deterministic agents (tests/golden/make_golden.py: gen_rollout), for all three stopping rules: actions,
    boards, seats, rewards, terminals and every agent's widened decision record incl. its NaN / -1 blanks and masks."""
    from boardlaw_amd import analysis, arrdict
    from boardlaw_amd.hex import Hex
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rollout_5x5.npz'))

    class EdgeAgent:
        def __init__(self, from_end, k=0):
            self.from_end, self.k = from_end, k

        def __call__(self, world, **kwargs):
            valid = world.valid
            order = valid.int().cumsum(-1) if not self.from_end else valid.int().flip(-1).cumsum(-1).flip(-1)
            want = torch.minimum(torch.full_like(order[:, :1], self.k + 1), valid.sum(-1, keepdim=True))
            hit = ((order == want) & valid).int()
            actions = hit.argmax(-1) if not self.from_end else valid.shape[-1] - 1 - hit.flip(-1).argmax(-1)
            return arrdict.arrdict(actions=actions, v=world.seats.float() + .5, count=valid.sum(-1).int())

    for tag, kw in (('steps', dict(n_steps=45)), ('trajs', dict(n_trajs=9)), ('reps', dict(n_reps=2))):
        worlds = Hex(board=torch.from_numpy(g['start_board']).cuda(), seats=torch.from_numpy(g['start_seats']).cuda())
        trace = analysis.rollout(worlds, [EdgeAgent(False, 1), EdgeAgent(True, 0)], **kw)
        eq = lambda t, name: np.array_equal(t.cpu().numpy(), g[f'{tag}_{name}'], equal_nan=True)
        assert eq(trace.actions, 'actions') and eq(trace.worlds.board, 'board') and eq(trace.worlds.seats, 'seats'), tag
        assert eq(trace.transitions.rewards, 'rewards') and eq(trace.transitions.terminal, 'terminal'), tag
        for a in ('0', '1'):
            for k in ('actions', 'v', 'count', 'mask'):
                assert eq(trace.decisions[a][k], f'dec{a}_{k}'), (tag, a, k)


@pytest.mark.gpu
def test_chunk_evaluator_with_search_agents():
    """arena.neural.ChunkEvaluator on the GPU: three search agents (the third one loaded from the reference-format checkpoint),
    every ordered pair plays n_envs_per games to the end through masked variable-size batches."""
    from boardlaw_amd import arena, networks, storage
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent
    here = os.path.dirname(os.path.abspath(__file__))
    agents = {}
    for i in range(2):
        torch.manual_seed(i)
        w = Hex.initial(1, 5)
        agents[f'net{i}'] = MCTSAgent(networks.FCModel(w.obs_space, w.action_space, width=32, depth=2).cuda(), n_nodes=8 + 8 * i)
    agents['snap'] = storage.agent_from_checkpoint(storage.load(os.path.join(here, 'golden', 'snapshot_5x5.pt')), device='cuda', inference='torch')
    ev = arena.ChunkEvaluator(lambda n: Hex.initial(n, 5), agents, n_envs_per=16)
    assert ev.tracker.n_envs == 6 * 16
    results = []
    while not ev.finished():
        results.extend(ev.step())
    assert sorted(r.names for r in results) == sorted((a, b) for a in agents for b in agents if a != b)
    for r in results:
        assert r.games == 16 and sum(r.wins) == 16 and r.moves >= 16 * 9 and r.boardsize == 5   # a 5x5 game takes >= 9 plies
    assert arena.evaluate_chunk(lambda n: Hex.initial(n, 5), lambda name: agents[name],
                                __import__('pandas').DataFrame(0, ['net0', 'net1'], ['net0', 'net1']), 4)[0].games == 4


@pytest.mark.gpu
@pytest.mark.parametrize('S,n,pool,T', [(9, 1000, 2048, 32), (9, 777, 2048, 32), (7, 64, 2048, 32), (9, 2048, 2048, 32), (5, 3, 2048, 32),
                                        (13, 2100, 4096, 16), (3, 17000, 32768, 100), (11, 2100, 4096, 16)])
def test_masked_captured_move_equals_the_eager_masked_call(S, n, pool, T):
    """One move captured for a capacity bucket (2048 / 1024 / 64 rows) serves a call of n envs with the other rows switched off
    on the device (bl_search_t.n_active): decisions identical, bit for bit, to the eager call on exactly those n envs under the
    same seed -- same Dirichlet rows, same uniforms (torch's kernels give row i the same numbers whatever the batch size), and a
    q-range over the n envs only.  Twice in a row: the second replay reuses the capture with another n.
    (13, 2100): from 128 actions on torch's own reduce kernel normalises the Dirichlet draw and picks its summation order by the
    tensor's shape -- a padded 13x13 move differed from the eager one in the last bit of some root logits (round 4 padded there;
    found here in round 5) -- so boards from 12x12 up are captured for exactly n rows.
    (3, 17000) with 100 nodes: the (rows, T) block of a 32768-row capture would advance the generator by two Philox rounds where the
    eager call advances it by one -- MCTSAgent._pad_keeps_the_stream refuses the padding and captures for exactly n rows."""
    from boardlaw_amd import networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent, MoveRng
    from test_reference_fixtures import EdgeAgent
    torch.manual_seed(S)
    worlds = Hex.initial(pool, S)
    for k in range(S * S // 3):                                   # mid-game positions, different per env
        v = worlds.valid
        worlds, _ = worlds.step((torch.rand(v.shape, device='cuda') * v).argmax(-1))
    net = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, width=256, depth=2).cuda(), fused=True)
    eager = MCTSAgent(net, n_nodes=T, rng=MoveRng())
    graphed = MCTSAgent(net, n_nodes=T, rng=MoveRng(), graph=True)
    if pool > 2048 and S != 11:
        padded = graphed._capacity(n, worlds[:n]) != n
        assert not padded, (S, n, graphed._capacity(n, worlds[:n]))
    for m in (n, max(1, n // 2 + 1)):
        sub = worlds[torch.randperm(pool, device='cuda')[:m]]
        for ev in (True, False):
            torch.manual_seed(100 + m); a = eager(sub, eval=ev)
            torch.manual_seed(100 + m); b = graphed(sub, eval=ev)
            for k in ('logits', 'prior', 'v', 'actions', 'n_leaves', 'n_sims'):
                x, y = a[k], b[k]
                assert x.shape == y.shape, k
                if x.dtype == torch.half:
                    x, y = x.view(torch.int16), y.view(torch.int16)
                assert torch.equal(x, y), (k, m, ev)
    assert len(graphed._graphs) <= 4          # (eval, not eval) x at most two capacity buckets (or two exact sizes)


@pytest.mark.gpu
@pytest.mark.parametrize('S', [7, 9, 11])
def test_arena_at_config_5_size_captured_equals_eager(S):
    """BASELINE config 5's shape: 2048 games per board size through arena.evaluate's masked calls, two 64-sim 512x4 search
    agents.  With captured moves (one per capacity bucket, rows beyond the call switched off) the whole match -- every move of
    every game -- is the eager match under the same seed: identical wins, moves and games per seat permutation; and the captures
    are few (the batches shrink through the buckets) and within the cache's byte bound."""
    from boardlaw_amd import arena, networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent, MoveRng
    nets = []
    for i in range(2):
        torch.manual_seed(i)
        w0 = Hex.initial(1, S)
        nets.append(networks.Inference(networks.FCModel(w0.obs_space, w0.action_space, width=512, depth=4).cuda(), fused=True))
    results = {}
    for mode in ('eager', 'graph'):
        pair = {name: MCTSAgent(net, graph=(mode == 'graph'), n_nodes=64, rng=MoveRng()) for name, net in zip(('one', 'two'), nets)}
        torch.manual_seed(7)
        results[mode] = arena.evaluate(Hex.initial(2048, S), pair)
        if mode == 'graph':
            for a in pair.values():
                assert 1 <= len(a._graphs) <= 8 and sum(g.nbytes for g in a._graphs.values()) <= a.GRAPH_CACHE_BYTES
                # measured from the allocator, not a formula: at least the capture's children array (rows x T x A i16)
                assert all(g.nbytes >= g.capacity * 64 * S * S * 2 for g in a._graphs.values()), [(g.capacity, g.nbytes) for g in a._graphs.values()]
    for e, g in zip(results['eager'], results['graph']):
        assert e.names == g.names and e.wins == g.wins and e.moves == g.moves and e.games == g.games, (e, g)
    assert sum(r.games for r in results['graph']) == 2048 and all(sum(r.wins) == r.games for r in results['graph'])
    assert all(r.moves >= r.games * (2 * S - 1) / 2 for r in results['graph'])


@pytest.mark.gpu
def test_learner_step_on_gpu_matches_cpu_fp32():
    """main.optimize (main.py:76-98) on the GPU under AMP against the same step in f32 on the CPU, same batch, same
    parameters: losses within 2e-3 relative, every updated parameter within 2e-3 absolute of the CPU's (lr 1e-3: Adam's first
    step moves each weight by ~lr, so agreement to 2e-3 means the gradient signs agree wherever the gradient is not tiny --
    f16 autocast forward/backward is the reference's own GPU/CPU gap)."""
    from boardlaw_amd import arrdict, networks, heads, training
    torch.manual_seed(0)
    B, S = 512, 5
    A = S * S
    obs = (torch.rand(B, S, S, 2) < .3).float(); obs[..., 1] *= 1 - obs[..., 0]
    valid = (obs == 0).all(-1).reshape(B, A)
    seats = torch.randint(0, 2, (B,), dtype=torch.int)
    target = torch.log_softmax(torch.randn(B, A).masked_fill(~valid, -np.inf), -1)
    rtg = torch.rand(B, 2) * 2 - 1

    def batch(dev):
        class W:
            pass
        w = W(); w.obs, w.valid, w.seats = obs.to(dev), valid.to(dev), seats.to(dev)
        return arrdict.arrdict(worlds=w, decisions=arrdict.arrdict(logits=target.to(dev).half()), reward_to_go=rtg.to(dev))

    results = {}
    for dev in ('cpu', 'cuda'):
        torch.manual_seed(1)
        net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(A), width=64, depth=3)
        with torch.no_grad():
            for p in net.parameters():
                if p.ndim == 0:
                    p.fill_(0.3)
        net = net.to(dev)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        scaler = torch.amp.GradScaler('cuda', enabled=(dev == 'cuda'))
        pl, vl = training.optimize(net, scaler, opt, batch(dev))
        results[dev] = (float(pl), float(vl), {k: v.detach().cpu().float() for k, v in net.state_dict().items()})
    (pc, vc, sc), (pg, vg, sg) = results['cpu'], results['cuda']
    assert abs(pc - pg) <= 2e-3 * abs(pc) and abs(vc - vg) <= 2e-3 * max(abs(vc), 1e-3), (pc, pg, vc, vg)
    for k in sc:
        assert (sc[k] - sg[k]).abs().max() <= 2e-3, k


# ------------------------------------------------------------------------------------------------ arena fan-out (neural.py:202-274)
def test_chunk_jobs_cover_every_missing_pair_once():
    """neural.py:205-232: diagonal blocks play their own pairs, skew blocks only the cross pairs; together every ordered pair
    with games missing is played in exactly one job, and complete pairs in none."""
    from boardlaw_amd import arena
    names = [f'a{i}' for i in range(7)]
    games = np.zeros((7, 7), int)
    games[0, 1] = 4; games[5, 6] = 4; games[6, 5] = 4          # already complete
    jobs = arena.chunk_jobs(games, names, n_envs_per=4, chunks=3)
    assert set(jobs) == {(0, 0), (1, 1), (0, 1), (0, 2), (1, 2)}        # block (2,2) = {a6} alone has nothing to play
    todo = np.zeros((7, 7), int)
    for block_names, played in jobs.values():
        idx = [names.index(n) for n in block_names]
        todo[np.ix_(idx, idx)] += (played < 4)
    want = (games < 4).astype(int); want[np.diag_indices(7)] = 0
    assert np.array_equal(todo, want)


def test_run_jobs_pool_and_serial():
    from boardlaw_amd import arena
    import pool_helpers
    jobs = {k: (pool_helpers.square, (k,)) for k in range(7)}
    assert dict(arena.run_jobs(jobs, n_workers=0)) == {k: k * k for k in range(7)}
    env = os.environ.get('PYTHONPATH', '')
    os.environ['PYTHONPATH'] = os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), env])      # the spawned workers import pool_helpers
    try:
        assert dict(arena.run_jobs(jobs, n_workers=2)) == {k: k * k for k in range(7)}
        with pytest.raises(RuntimeError, match='failed in its worker'):
            dict(arena.run_jobs({0: (pool_helpers.square, ('x', 'y'))}, n_workers=1))
        # a worker that dies without posting must not leave the parent waiting for ever (round-3 advisor finding)
        with pytest.raises(RuntimeError, match='died with 1 job'):
            dict(arena.run_jobs({0: (pool_helpers.die, (3,))}, n_workers=1, poll_seconds=0.2))
    finally:
        os.environ['PYTHONPATH'] = env


def test_match_pool_keeps_its_workers_and_lpt_partition():
    """arena.MatchPool (bench.py --config 5: several matches in flight per GPU): persistent workers -- the same processes and
    their players' state across plays --, results per worker in order; arena.lpt_partition deals the largest jobs first to the
    lightest worker; arena.workers_per_gpu is 3 for matches of up to 2048 games."""
    from boardlaw_amd import arena
    import pool_helpers
    assert [arena.workers_per_gpu(n) for n in (64, 2048, 2049, 4096, 8192)] == [3, 3, 2, 2, 1]
    sizes = list(range(3, 12))
    parts = arena.lpt_partition([S ** 2.2 for S in sizes], 3)
    assert sorted(i for p_ in parts for i in p_) == list(range(9)) and [p_[0] for p_ in parts] == [8, 7, 6]
    loads = [sum(sizes[i] ** 2.2 for i in p_) for p_ in parts]
    assert max(loads) / min(loads) < 1.25
    env = os.environ.get('PYTHONPATH', '')
    os.environ['PYTHONPATH'] = os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), env])
    try:
        with arena.MatchPool(pool_helpers.Accumulator, (100,), n_workers=2, device=None) as pool:
            a = pool.play([[1, 2, 3], [4]])
            b = pool.play([[5], [6, 7]])
        assert [[r[1] for r in rs] for rs in a] == [[101, 104, 109], [116]] and [[r[1] for r in rs] for rs in b] == [[125], [136, 149]]
        assert a[0][0][0] == b[0][0][0] and a[1][0][0] == b[1][0][0] and a[0][0][0] != a[1][0][0]        # the same two processes
        assert [r[2] for r in b[0]] == [4] and [r[2] for r in b[1]] == [2, 3]                            # which remember their earlier jobs
        with pytest.raises(RuntimeError, match='failed'):
            with arena.MatchPool(pool_helpers.Accumulator, (0,), n_workers=1, device=None) as pool:
                pool.play([['x']])
    finally:
        os.environ['PYTHONPATH'] = env


def test_evaluate_gen_two_workers_equals_one_chunk_evaluator(oracle):
    """The fan-out (two worker processes, blocks of two agents) plays exactly the games a single ChunkEvaluator over all four
    agents plays: same wins and moves per ordered pair (deterministic agents; each pair's games do not depend on the others)."""
    from boardlaw_amd import arena
    import pool_helpers
    names = ['e0', 'e1', 'e2', 'e3']
    agents = {n: pool_helpers.edge_agent(n) for n in names}
    ev = arena.ChunkEvaluator(pool_helpers.cpu_worlds, agents, None, n_envs_per=4, device='cpu')
    single = []
    while not ev.finished():
        single.extend(ev.step())
    env = os.environ.get('PYTHONPATH', '')
    os.environ['PYTHONPATH'] = os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), env])
    try:
        fanned, last = [], None
        for results, stats in arena.evaluate_gen(pool_helpers.cpu_worlds, pool_helpers.edge_agent, np.zeros((4, 4), int), names=names,
                                                 n_envs_per=4, chunks=2, n_workers=2):
            fanned.extend(results); last = stats
    finally:
        os.environ['PYTHONPATH'] = env
    key = lambda r: tuple(r.names)
    assert sorted(map(key, fanned)) == sorted(map(key, single)) and len(fanned) == 12
    a, b = {key(r): r for r in fanned}, {key(r): r for r in single}
    for k in a:
        assert tuple(a[k].wins) == tuple(b[k].wins) and a[k].moves == b[k].moves and a[k].games == 4
    assert last.finished == last.total == 3 and last.matchups == 12 and last.games == 48


def _big_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from boardlaw_amd import parallel, networks, heads
    torch.set_num_threads(2)
    parallel.init('gloo')
    torch.manual_seed(0)
    net = networks.FCModel(heads.Tensor((13, 13, 2)), heads.Masked(169), width=1024, depth=8)      # config 4's network: 8.9 M parameters
    gen = torch.Generator(); gen.manual_seed(100 + rank)
    for p in net.parameters():
        p.grad = torch.randn(p.shape, generator=gen)
    mine = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
    parallel.allreduce_gradients(net)
    flat = torch.cat([p.grad.flatten() for p in net.parameters()])
    out.put((rank, mine.double().sum().item(), flat.double().sum().item(), flat[::100003].clone().numpy(), flat.numel()))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_allreduce_at_config_4_network_size():
    """parallel.allreduce_gradients on the 1024x8 network of config 4 (one flat 35 MB fp32 bucket) over two gloo ranks: every rank
    ends with the mean of the two ranks' gradients, element for element."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_big_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (out.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(120); assert p.exitcode == 0
    assert res[0][3] == res[1][3] > 8_000_000
    assert np.array_equal(res[0][2], res[1][2])                                       # both ranks hold the same averaged gradient
    assert abs(res[0][1] - (res[0][0] + res[1][0]) / 2) <= 1e-3 * max(1., abs(res[0][1]))  # and it is the mean of the two
