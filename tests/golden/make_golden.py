"""Generates the committed golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and oracle/_ref/*.so from `make -C oracle ref`).
Nothing here travels to the GPU box except the .npz outputs; the reference's sources are imported in place and
never copied.

How the reference is run:
  * `boardlaw.mcts`, `boardlaw.hex`, `boardlaw.networks` are imported from /root/reference unmodified.
  * Its two native modules are taken from oracle/_ref (the reference's own CPU sources compiled at -O2 by
    oracle/Makefile) by pre-seeding the loader caches `boardlaw.mcts.cuda._cache` / `boardlaw.hex.cuda._cache`
    (boardlaw/mcts/cuda.py:5-11), so the reference's JIT loader (which needs ninja + minutes) is not involved.
  * `aljpy` (a logging helper the reference imports in rebar/profiling.py:9, absent from this image) is replaced
    by an in-memory module exposing `logger()`; it takes no part in any arithmetic.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import importlib.util
import logging
import os
import sys
import types
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('BOARDLAW_REFERENCE', '/root/reference')
VARIANT = os.environ.get('GOLDEN_REF_VARIANT', '')  # '' (=-O2) or '_O0'


def _load_ext(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    aljpy = types.ModuleType('aljpy')
    aljpy.logger = lambda *a, **k: logging.getLogger('aljpy')

    @contextlib.contextmanager
    def timer():
        yield None
    aljpy.timer = timer
    sys.modules['aljpy'] = aljpy
    pavlov = types.ModuleType('pavlov')
    pavlov.stats = types.ModuleType('pavlov.stats')
    pavlov.stats.mean = lambda *a, **k: None
    sys.modules['pavlov'] = pavlov
    sys.modules['pavlov.stats'] = pavlov.stats
    sys.path.insert(0, REF)
    os.chdir('/tmp')
    import matplotlib
    matplotlib.use('Agg')
    import boardlaw.mcts.cuda as mcuda
    import boardlaw.hex.cuda as hcuda
    mcuda._cache = _load_ext('mctscuda' + VARIANT, os.path.join(ROOT, 'oracle/_ref/mctscuda%s.so' % VARIANT))
    hcuda._cache = _load_ext('hexcuda', os.path.join(ROOT, 'oracle/_ref/hexcuda.so'))
    import boardlaw.mcts as mcts
    import boardlaw.hex as hex_
    import boardlaw.networks as networks
    import boardlaw.validation as validation
    return mcts, hex_, networks, validation, mcuda, hcuda


def np_(x):
    """tensor -> numpy, halves kept as raw uint16 bit patterns (npz-portable, bit-exact)."""
    x = x.detach().cpu().contiguous()
    if x.dtype == torch.half:
        return x.view(torch.int16).numpy().view(np.uint16).copy()
    if x.dtype == torch.bool:
        return x.numpy().astype(np.uint8)
    return x.numpy().copy()


# --------------------------------------------------------------------------------------------------------------
# 1. Hex board dynamics: random playouts through the reference's Hex world  (SURVEY 8c fixture 2)
# --------------------------------------------------------------------------------------------------------------
def gen_hex(hex_, hcuda, out):
    data = {}
    for S in (3, 4, 5, 7, 9, 11, 13):
        torch.manual_seed(100 + S)
        B = 24
        worlds = hex_.Hex.initial(B, S, device='cpu')
        rec = {k: [] for k in ('board', 'seats', 'actions', 'obs', 'valid', 'raw_board', 'rewards',
                               'new_board', 'new_seats', 'terminal')}
        for _ in range(3 * S * S):
            actions = torch.distributions.Categorical(probs=worlds.valid.float()).sample()
            raw = worlds.board.clone()
            raw_rewards = hcuda.step(raw, worlds.seats.int(), actions.int())
            new_worlds, trans = worlds.step(actions)
            assert torch.equal(raw_rewards, trans.rewards)
            rec['board'].append(np_(worlds.board)); rec['seats'].append(np_(worlds.seats))
            rec['actions'].append(np_(actions.int())); rec['obs'].append(np_(worlds.obs).astype(np.uint8))
            rec['valid'].append(np_(worlds.valid)); rec['raw_board'].append(np_(raw))
            rec['rewards'].append(np_(trans.rewards)); rec['new_board'].append(np_(new_worlds.board))
            rec['new_seats'].append(np_(new_worlds.seats)); rec['terminal'].append(np_(trans.terminal))
            worlds = new_worlds
        for k, v in rec.items():
            data[f'S{S}_{k}'] = np.stack(v)
    np.savez_compressed(os.path.join(out, 'hex_playouts.npz'), **data)
    print('hex_playouts', sum(v.nbytes for v in data.values()) // 1024, 'KiB raw')


# --------------------------------------------------------------------------------------------------------------
# 2./3. Search kernels on live trees + whole searches (SURVEY 8c fixtures 1 and 3)
# --------------------------------------------------------------------------------------------------------------
class Recorder:
    """Wraps the reference's native entry points (boardlaw/mcts/cuda.py:28-42) to log inputs/outputs."""

    def __init__(self, mcuda, keep_ops):
        self.mcuda, self.keep_ops = mcuda, keep_ops
        self.ops = []       # per-op records (only for sims listed in keep_ops)
        self.rands = []     # every descend's (B,T) f16 uniforms, in call order
        self.sim = 0
        self._descend, self._root, self._backup = mcuda.descend, mcuda.root, mcuda.backup

    def install(self):
        m = self.mcuda
        m.descend, m.root, m.backup = self.descend, self.root, self.backup

    def uninstall(self):
        m = self.mcuda
        m.descend, m.root, m.backup = self._descend, self._root, self._backup

    @staticmethod
    def tree(m):
        return dict(logits=np_(m.logits), w=np_(m.w), n=np_(m.n), c_puct=np_(m.c_puct), seats=np_(m.seats),
                    terminal=np_(m.terminal), children=np_(m.children))

    def descend(self, m):
        state = torch.get_rng_state()
        rands = torch.rand_like(m.logits[:, :, 0])          # what cpu.cpp:187 is about to draw
        torch.set_rng_state(state)
        self.sim += 1
        before = self.tree(m) if self.sim in self.keep_ops else None
        d = self._descend(m)
        self.rands.append(np_(rands))
        if before is not None:
            q = m.w.float() / (m.n.float().unsqueeze(-1) + 1e-4)
            before.update(rands=np_(rands), parents=np_(d.parents), actions=np_(d.actions),
                          root_probs=np_(self._root(m)), qminmax=np.array([q.min().item(), q.max().item()], np.float32))
            self.ops.append(('descend', self.sim, before))
        return d

    def root(self, m):
        return self._root(m)

    def backup(self, bk, leaves):
        # Backup exposes no properties (wrappers.cpp:67-69); the MCTS object that built it is patched to hand us
        # the tensors via self.pending.
        rec = None
        if self.sim in self.keep_ops:
            t = self.pending
            rec = dict(v=np_(t['v']), w=np_(t['w']), n=np_(t['n']), rewards=np_(t['rewards']),
                       parents=np_(t['parents']), terminal=np_(t['terminal']), leaves=np_(leaves))
        self._backup(bk, leaves)
        if rec is not None:
            rec.update(w_after=np_(self.pending['w']), n_after=np_(self.pending['n']))
            self.ops.append(('backup', self.sim, rec))


class RecordingNetwork:
    """Wraps an FCModel; logs what MCTS.initialize / MCTS.simulate (mcts/__init__.py:72-80,131-136) receive."""

    def __init__(self, net):
        self.net, self.calls = net, []

    def __call__(self, world):
        d = self.net(world)
        self.calls.append(dict(logits=d.logits.detach().clone(), v=d.v.detach().clone(),
                               board=world.board.clone(), seats=world.seats.clone()))
        return d


def run_search_fixture(mcts_mod, hex_, networks, mcuda, S, B, T, width, depth, n_moves, seed, keep_ops, mix_moves):
    torch.manual_seed(seed)
    worlds = hex_.Hex.initial(B, S, device='cpu')
    for _ in range(mix_moves):
        actions = torch.distributions.Categorical(probs=worlds.valid.float()).sample()
        worlds, _ = worlds.step(actions)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=width, depth=depth)
    # ReZero alphas start at 0 (networks.py:15) which would make the net ignore its body: perturb so value and
    # policy depend on the position. Recorded outputs are what the replay uses, so this only adds variety.
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim == 0:
                p.fill_(0.5)
    rnet = RecordingNetwork(net)

    out = dict(meta=np.array([S, B, T, width, depth, n_moves, seed], np.int64))
    out['world0_board'] = np_(worlds.board); out['world0_seats'] = np_(worlds.seats)
    # the network itself (SURVEY 8c fixture 3): its state_dict, so that a restated FCModel can be replayed against the
    # recorded outputs.  Keys as the reference names them (networks.py:10-40), '.' kept, prefixed 'net_state::'.
    for k, v in net.state_dict().items():
        out['net_state::' + k] = np_(v)
    ops_all = []

    # capture Backup's tensors: patch MCTS.backup to stash them (mcts/__init__.py:97-106)
    rec = Recorder(mcuda, keep_ops)
    orig_backup = mcts_mod.MCTS.backup

    def backup(self, leaves):
        rec.pending = dict(v=self.decisions.v, w=self.stats.w, n=self.stats.n, rewards=self.transitions.rewards,
                           parents=self.tree.parents, terminal=self.transitions.terminal)
        return orig_backup(self, leaves)
    mcts_mod.MCTS.backup = backup

    # capture the Dirichlet draw: dirichlet_noise (mcts/__init__.py:13-24) samples once per search
    draws = []
    orig_sample = torch.distributions.Dirichlet.sample

    def sample(self, shape=torch.Size()):
        d = orig_sample(self, shape)
        draws.append(d.clone())
        return d
    torch.distributions.Dirichlet.sample = sample

    rec.install()
    try:
        for move in range(n_moves):
            rec.sim, rec.rands, rec.ops = 0, [], []
            rnet.calls = []
            del draws[:]
            m = mcts_mod.mcts(worlds, rnet, n_nodes=T)
            r = m.root()
            # what MCTSAgent.__call__ does next (mcts/__init__.py:221)
            actions = torch.distributions.Categorical(logits=r.logits.float()).sample()
            agent_like = dict(logits=r.logits, prior=r.prior, v=r.v, n_leaves=m.n_leaves(), actions=actions)
            p = f'm{move}_'
            out[p + 'dirichlet'] = np_(draws[0])
            out[p + 'rands'] = np.stack(rec.rands)                                     # (T-1,B,T) f16 bits
            out[p + 'net0_logits'] = np_(rnet.calls[0]['logits'])                      # f32, initialize
            out[p + 'net0_v'] = np_(rnet.calls[0]['v'])
            out[p + 'net0_obs'] = np_(worlds.obs).astype(np.uint8); out[p + 'net0_valid'] = np_(worlds.valid)
            out[p + 'net0_seats'] = np_(worlds.seats)
            # the first simulations' network outputs before `.half()` (mcts/__init__.py:135-136): f32 on the CPU path
            out[p + 'net_logits_f32'] = np.stack([np_(c['logits']) for c in rnet.calls[1:4]])
            out[p + 'net_v_f32'] = np.stack([np_(c['v']) for c in rnet.calls[1:4]])
            out[p + 'net_logits'] = np.stack([np_(c['logits'].half()) for c in rnet.calls[1:]])  # (T-1,B,A)
            out[p + 'net_v'] = np.stack([np_(c['v'].half()) for c in rnet.calls[1:]])
            out[p + 'net_board'] = np.stack([np_(c['board']) for c in rnet.calls[1:]])  # leaf worlds seen by net
            out[p + 'net_seats'] = np.stack([np_(c['seats']) for c in rnet.calls[1:]])
            for k, v in agent_like.items():
                out[p + 'dec_' + k] = np_(v)
            out[p + 'root_probs'] = np_(mcuda.root(m._cuda()))                       # before mcts/__init__.py:147's log
            out[p + 'children'] = np_(m.tree.children); out[p + 'parents'] = np_(m.tree.parents)
            out[p + 'relation'] = np_(m.tree.relation); out[p + 'n'] = np_(m.stats.n); out[p + 'w'] = np_(m.stats.w)
            out[p + 'tree_logits'] = np_(m.decisions.logits); out[p + 'tree_v'] = np_(m.decisions.v)
            out[p + 'rewards'] = np_(m.transitions.rewards); out[p + 'terminal'] = np_(m.transitions.terminal)
            out[p + 'boards'] = np_(m.worlds.board); out[p + 'seats'] = np_(m.worlds.seats)
            for kind, sim, d in rec.ops:
                ops_all.append((move, kind, sim, d))
            worlds, trans = worlds.step(actions)
            out[p + 'step_rewards'] = np_(trans.rewards); out[p + 'step_terminal'] = np_(trans.terminal)
    finally:
        rec.uninstall()
        mcts_mod.MCTS.backup = orig_backup
        torch.distributions.Dirichlet.sample = orig_sample
    return out, ops_all


def gen_search(mcts_mod, hex_, networks, mcuda, out):
    # (name, S, B, T, width, depth, moves, seed, sims whose op-level tensors are kept, premix moves)
    configs = [
        ('search_3x3', 3, 32, 8, 8, 2, 6, 11, (1, 3, 7), 1),
        ('search_5x5', 5, 64, 16, 16, 4, 8, 12, (1, 5, 10, 15), 8),      # BASELINE config 1
        ('search_9x9', 9, 64, 64, 32, 2, 3, 13, (1, 20, 63), 27),        # BASELINE config 2's board/T at B=64
        ('search_13x13', 13, 16, 64, 32, 2, 2, 14, (1, 40, 63), 56),
    ]
    for name, S, B, T, width, depth, moves, seed, keep, mix in configs:
        res, ops = run_search_fixture(mcts_mod, hex_, networks, mcuda, S, B, T, width, depth, moves, seed, set(keep), mix)
        np.savez_compressed(os.path.join(out, name + '.npz'), **res)
        opd = {}
        for i, (move, kind, sim, d) in enumerate(ops):
            if move > 1:
                continue
            for k, v in d.items():
                opd[f'{kind}_m{move}_s{sim}_{k}'] = v
        np.savez_compressed(os.path.join(out, name.replace('search', 'ops') + '.npz'), **opd)
        print(name, os.path.getsize(os.path.join(out, name + '.npz')) // 1024, 'KiB;', 'ops',
              os.path.getsize(os.path.join(out, name.replace('search', 'ops') + '.npz')) // 1024, 'KiB')


# --------------------------------------------------------------------------------------------------------------
# 4. Toy-environment integration goldens (boardlaw/mcts/tests.py:242-279)
# --------------------------------------------------------------------------------------------------------------
def gen_toy(mcts_mod, validation, out):
    res = {}
    torch.manual_seed(21)
    agent = validation.ProxyAgent()
    m = mcts_mod.mcts(validation.Win.initial(device='cpu'), agent, n_nodes=3)
    res['win_v'] = np_(m.root().v.float())
    m = mcts_mod.mcts(validation.WinnerLoser.initial(device='cpu'), agent, n_nodes=3)
    res['winnerloser_v'] = np_(m.root().v.float())
    m = mcts_mod.mcts(validation.All.initial(length=3, device='cpu'), agent, n_nodes=15, noise_eps=0.)
    res['all_v'] = np_(m.root().v.float())
    m = mcts_mod.mcts(validation.All.initial(n_envs=2, length=3, device='cpu'), agent, n_nodes=15, noise_eps=0.)
    res['all2_v'] = np_(m.root().v.float())
    np.savez_compressed(os.path.join(out, 'toy_envs.npz'), **res)
    print('toy', {k: v.tolist() for k, v in res.items()})


# --------------------------------------------------------------------------------------------------------------
# 5. expf over every binary16 input, as this container's libm computes it (pins the GPU box's libm to ours)
# --------------------------------------------------------------------------------------------------------------
def gen_exp(out):
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle/liboracle.so'))
    table = np.zeros(65536, np.float32)
    lib.orc_exp_table(table.ctypes.data_as(ctypes.c_void_p))
    # cross-check against torch's own f16->f32->exp on CPU for finite inputs is NOT expected to be bit-equal
    # (torch uses SLEEF); the table is libm's, which is what the reference's cpu.cpp:86 calls.
    np.savez_compressed(os.path.join(out, 'expf_f16_table.npz'), table=table.view(np.uint32))
    print('expf table written')


# --------------------------------------------------------------------------------------------------------------
# 6. The rollout/trace API (boardlaw/analysis.py:47-87) with deterministic agents, and an agent checkpoint in the
#    reference's wire format (mcts/__init__.py:231-241 inside pavlov/storage.py:52-56's torch.save)
# --------------------------------------------------------------------------------------------------------------
class EdgeAgent:
    """Deterministic fixture agent: plays the k-th legal cell from the front (or back); also returns a float and an int
    field so that combine_decisions' NaN / -1 blanks show up in the trace."""

    def __init__(self, from_end, k=0):
        self.from_end, self.k = from_end, k

    def __call__(self, world, **kwargs):
        from rebar import arrdict
        valid = world.valid
        order = valid.int().cumsum(-1) if not self.from_end else valid.int().flip(-1).cumsum(-1).flip(-1)
        want = torch.minimum(torch.full_like(order[:, :1], self.k + 1), valid.sum(-1, keepdim=True))
        actions = ((order == want) & valid).int().argmax(-1) if not self.from_end else \
            (valid.shape[-1] - 1 - ((order == want) & valid).int().flip(-1).argmax(-1))
        return arrdict.arrdict(actions=actions, v=world.seats.float() + .5, count=valid.sum(-1).int())


def import_reference_analysis():
    class Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith('__'):
                raise AttributeError(k)
            return Stub(self.__name__ + '.' + k)
    for name in ('rebar.recording', 'pavlov.runs', 'pavlov.storage', 'boardlaw.arena'):     # imported, never used by rollout()
        sys.modules[name] = Stub(name)
    import pavlov, rebar, boardlaw
    pavlov.runs, pavlov.storage = sys.modules['pavlov.runs'], sys.modules['pavlov.storage']
    rebar.recording, boardlaw.arena = sys.modules['rebar.recording'], sys.modules['boardlaw.arena']
    import boardlaw.analysis as analysis
    return analysis


def gen_rollout(hex_, out):
    analysis = import_reference_analysis()
    res = {}
    # mixed starting positions and seats: env e has played e % 5 random legal moves
    torch.manual_seed(77)
    start = hex_.Hex.initial(12, 5, device='cpu')
    for k in range(4):
        actions = torch.distributions.Categorical(probs=start.valid.float()).sample()
        stepped, _ = start.step(actions)
        go = (torch.arange(12) % 5) > k
        start[go] = stepped[go]
    res['start_board'] = np_(start.board); res['start_seats'] = np_(start.seats)
    for tag, kw in (('steps', dict(n_steps=45)), ('trajs', dict(n_trajs=9)), ('reps', dict(n_reps=2))):
        worlds = start.clone()
        trace = analysis.rollout(worlds, [EdgeAgent(False, 1), EdgeAgent(True, 0)], **kw)
        res[tag + '_actions'] = np_(trace.actions); res[tag + '_board'] = np_(trace.worlds.board); res[tag + '_seats'] = np_(trace.worlds.seats)
        res[tag + '_rewards'] = np_(trace.transitions.rewards); res[tag + '_terminal'] = np_(trace.transitions.terminal)
        for a in ('0', '1'):
            d = trace.decisions[a]
            for k in ('actions', 'v', 'count', 'mask'):
                res[f'{tag}_dec{a}_{k}'] = np_(d[k])
    np.savez_compressed(os.path.join(out, 'rollout_5x5.npz'), **res)
    print('rollout', {k: v.shape for k, v in res.items() if k.endswith('_actions')})


def gen_snapshot(mcts_mod, hex_, networks, out):
    torch.manual_seed(31)
    worlds = hex_.Hex.initial(4, 5, device='cpu')
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=16, depth=3)
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim == 0:
                p.fill_(0.25)
    agent = mcts_mod.MCTSAgent(net, n_nodes=16, c_puct=1 / 8)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    # what main.run's storer writes (boardlaw/main.py:155-160 -> pavlov/storage.py:29-39,52-56)
    torch.save({'agent': agent.state_dict(), 'opt': opt.state_dict()}, os.path.join(out, 'snapshot_5x5.pt'))
    d = net(worlds)
    np.savez_compressed(os.path.join(out, 'snapshot_5x5_outputs.npz'), logits=np_(d.logits), v=np_(d.v))
    print('snapshot', sorted(agent.state_dict())[:3], '...')


# --------------------------------------------------------------------------------------------------------------
# 7. A search whose network is as wide as the bench's (BEST row of boardlaw/main.py:17-25: 9x9 -> 512x4), so that the
#    fused MFMA kernels (bl_mlp_forward_f16 / bl_sim_infer_finish, W >= 256) meet outputs the reference recorded
# --------------------------------------------------------------------------------------------------------------
def gen_search_wide(mcts_mod, hex_, networks, mcuda, out):
    res, _ = run_search_fixture(mcts_mod, hex_, networks, mcuda, 9, 64, 64, 512, 4, 1, 15, set(), 27)
    # the leaf worlds can be rebuilt from the tree's boards; keep the file small
    for k in [k for k in res if k.endswith('net_board')]:
        del res[k]
    np.savez_compressed(os.path.join(out, 'search_9x9_w512.npz'), **res)
    print('search_9x9_w512', os.path.getsize(os.path.join(out, 'search_9x9_w512.npz')) // 1024, 'KiB')


# --------------------------------------------------------------------------------------------------------------
# 8. Arena (boardlaw/arena/common.py:75-106 `evaluate`, arena/neural.py:46-200 `Tracker`/`ChunkEvaluator`/`evaluate_chunk`)
#    with deterministic agents from seeded mid-game positions.  The arena modules import the reference's run-directory,
#    SQL and Elo layers at module level (never used by these functions): those are replaced by empty stub modules.
#    numpy >= 2 dropped `np.math`, which common.py:80 uses: it is pointed at the stdlib module for the call.
# --------------------------------------------------------------------------------------------------------------
class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        return _Stub(self.__name__ + '.' + k)


def import_reference_arena():
    import math
    for name in ('pavlov.storage', 'pavlov.runs', 'boardlaw.sql', 'boardlaw.backup', 'boardlaw.elos', 'boardlaw.arena.live',
                 'boardlaw.arena.mohex', 'rebar.parallel'):
        sys.modules[name] = _Stub(name)
    import pavlov, boardlaw
    pavlov.storage, pavlov.runs = sys.modules['pavlov.storage'], sys.modules['pavlov.runs']
    boardlaw.sql, boardlaw.backup, boardlaw.elos = sys.modules['boardlaw.sql'], sys.modules['boardlaw.backup'], sys.modules['boardlaw.elos']
    sys.modules.pop('boardlaw.arena', None)         # gen_rollout may have put a stub there
    if not hasattr(np, 'math'):
        np.math = math
    import boardlaw.arena.common as common
    import boardlaw.arena.neural as neural
    return common, neural


def premixed_worlds(hex_, n_envs, S, moves, seed):
    """Seeded mid-game positions: env e has played between 0 and `moves` uniformly random legal moves."""
    torch.manual_seed(seed)
    worlds = hex_.Hex.initial(n_envs, S, device='cpu')
    upto = torch.randint(moves + 1, (n_envs,))
    for k in range(moves):
        actions = torch.distributions.Categorical(probs=worlds.valid.float()).sample()
        stepped, trans = worlds.step(actions)
        go = (upto > k) & ~trans.terminal
        worlds[go] = stepped[go]
    return worlds


def gen_arena(hex_, out):
    common, neural = import_reference_arena()
    import pandas as pd
    res = {}
    for S, n_envs in ((5, 64), (7, 2048), (9, 2048), (11, 2048), (3, 2048)):     # 3x3: the small end of config 5's sweep (round 4)
        start = premixed_worlds(hex_, n_envs, S, S * S // 3, 500 + S)
        res[f'S{S}_board'] = np_(start.board); res[f'S{S}_seats'] = np_(start.seats)
        agents = {'front': EdgeAgent(False, 1), 'back': EdgeAgent(True, 0)}
        results = common.evaluate(start.clone(), agents)
        for i, r in enumerate(results):
            res[f'S{S}_r{i}_names'] = np.array(r.names)
            res[f'S{S}_r{i}_wins'] = np.array(r.wins); res[f'S{S}_r{i}_moves'] = np.array(r.moves); res[f'S{S}_r{i}_games'] = np.array(r.games)
        print('arena', S, [(r.names, r.wins, r.moves) for r in results])
    # ChunkEvaluator: three deterministic agents, all ordered pairs, with some games already played
    names = ['a', 'b', 'c']
    agents = {'a': EdgeAgent(False, 0), 'b': EdgeAgent(True, 1), 'c': EdgeAgent(False, 2)}
    games = pd.DataFrame([[0, 3, 0], [1, 0, 5], [0, 2, 0]], names, names)
    for S in (5, 9):
        n_envs_per = 8

        def worldfunc(n_envs, S=S):
            return premixed_worlds(hex_, n_envs, S, S * S // 3, 600 + S)
        ev = neural.ChunkEvaluator(worldfunc, agents, games.copy(), n_envs_per=n_envs_per, device='cpu')
        res[f'chunk{S}_board'] = np_(ev.worlds.board); res[f'chunk{S}_seats'] = np_(ev.worlds.seats)
        res[f'chunk{S}_live'] = np_(ev.tracker.live)
        results, picks = [], []
        while not ev.finished():
            name, mask, live = ev.tracker.suggest(ev.worlds.seats)
            picks.append((names.index(name), int(mask.sum())))
            results.extend(ev.step())
        res[f'chunk{S}_picks'] = np.array(picks)
        res[f'chunk{S}_games'] = games.values
        res[f'chunk{S}_names'] = np.array([r.names for r in results])
        res[f'chunk{S}_wins'] = np.array([r.wins for r in results]); res[f'chunk{S}_moves'] = np.array([r.moves for r in results])
        # the evaluator's own tallies at the end (reported pairs are marked -1, arena/neural.py:160): covers the pairs that
        # started with games already played, which the reference never reports (their wins never sum to n_envs_per)
        res[f'chunk{S}_final_wins'] = np_(ev.stats.wins); res[f'chunk{S}_final_moves'] = np_(ev.stats.moves)
        print('chunk', S, len(picks), 'steps', [(r.names, r.wins, r.moves) for r in results])
    np.savez_compressed(os.path.join(out, 'arena.npz'), **res)


# --------------------------------------------------------------------------------------------------------------
# 10. One-player Hex (boardlaw/hex/__init__.py:224-271): the reference's Lazy and Random worlds stepped with seeded player
#     moves.  Every Hex.step the reference makes is recorded in call order -- the player's move, then the opponent's replies on
#     the sub-batch that still owes one -- so a replay needs no random stream of its own: Random's opponent draws are data.
# --------------------------------------------------------------------------------------------------------------
def gen_solitaire(hex_, out):
    res = {}
    for kind in ('Lazy', 'Random'):
        cls = getattr(hex_, kind)
        for S in (3, 5, 7):
            torch.manual_seed(700 + S)
            B = 32
            worlds = cls.initial(B, S, device='cpu')
            calls = []                      # (outer step, sub-batch size, actions) of every Hex.step
            orig = hex_.Hex.step
            step_no = [0]

            def recording(self, actions, _orig=orig):
                calls.append((step_no[0], int(actions.shape[0]), np_(actions.long())))
                return _orig(self, actions)
            rec = {k: [] for k in ('board', 'seats', 'actions', 'new_board', 'new_seats', 'rewards', 'terminal', 'obs', 'valid')}
            hex_.Hex.step = recording
            try:
                for t in range(2 * S * S):
                    step_no[0] = t
                    actions = torch.distributions.Categorical(probs=worlds.valid.float()).sample()
                    rec['board'].append(np_(worlds.board)); rec['seats'].append(np_(worlds.seats)); rec['actions'].append(np_(actions))
                    rec['obs'].append(np_(worlds.obs).astype(np.uint8)); rec['valid'].append(np_(worlds.valid))
                    worlds, trans = worlds.step(actions)
                    assert trans.rewards.shape == (B, 1) and worlds.n_seats == 1
                    rec['new_board'].append(np_(worlds.board)); rec['new_seats'].append(np_(worlds.seats))
                    rec['rewards'].append(np_(trans.rewards)); rec['terminal'].append(np_(trans.terminal))
            finally:
                hex_.Hex.step = orig
            for k, v in rec.items():
                res[f'{kind}{S}_{k}'] = np.stack(v)
            res[f'{kind}{S}_call_step'] = np.array([c[0] for c in calls]); res[f'{kind}{S}_call_size'] = np.array([c[1] for c in calls])
            res[f'{kind}{S}_call_actions'] = np.concatenate([c[2] for c in calls])
            print('solitaire', kind, S, len(calls), 'Hex.step calls in', 2 * S * S, 'steps; games ended', int(np.stack(rec['terminal']).sum()))
    np.savez_compressed(os.path.join(out, 'solitaire.npz'), **res)


# --------------------------------------------------------------------------------------------------------------
# 9. The learner (boardlaw/main.py:61-74 `as_chunk`, :76-98 `optimize`) on a buffer of real self-play (reference
#    MCTSAgent on its CPU path), CPU f32 (GradScaler and autocast disable themselves without a GPU).  main.py imports the
#    run-directory/stats layer (pavlov) at module level; it is replaced by a stub whose `stats` records what
#    `optimize` reports (`loss.value`, `loss.policy`).
# --------------------------------------------------------------------------------------------------------------
def import_reference_main(recorded):
    stats = types.ModuleType('pavlov.stats')

    @contextlib.contextmanager
    def defer():
        yield None
    stats.defer = defer

    def record(name, *args, **kwargs):
        recorded.setdefault(name, []).append([a.detach().clone() if torch.is_tensor(a) else a for a in args])
    for fn in ('mean', 'rate', 'cumsum', 'max', 'gpu'):
        setattr(stats, fn, record)
    import pavlov
    pavlov.stats = stats
    sys.modules['pavlov.stats'] = stats
    for name in ('pavlov.logs', 'pavlov.runs', 'pavlov.storage', 'pavlov.archive', 'boardlaw.arena', 'boardlaw.storage', 'boardlaw.noisescales'):
        sys.modules[name] = _Stub(name)
        parent, _, leaf = name.rpartition('.')
        setattr(sys.modules[parent], leaf, sys.modules[name])
    import boardlaw.main as main
    return main


def gen_learner(mcts_mod, hex_, networks, out):
    import warnings
    from rebar import arrdict
    import boardlaw.learning as learning
    recorded = {}
    main = import_reference_main(recorded)
    res = {}
    S, B, T_buf, nodes, width, depth = 5, 16, 8, 16, 16, 3
    torch.manual_seed(41)
    worlds = premixed_worlds(hex_, B, S, 6, 42)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=width, depth=depth)
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim == 0:
                p.fill_(0.3)
    agent = mcts_mod.MCTSAgent(net, n_nodes=nodes)
    buffer = []
    for _ in range(T_buf):                                   # main.py:172-186
        with torch.no_grad():
            decisions = agent(worlds, value=True)
        new_worlds, transition = worlds.step(decisions.actions)
        buffer.append(arrdict.arrdict(worlds=worlds, decisions=decisions.half(), transitions=learning.half(transition)).detach())
        worlds = new_worlds
    stacked = arrdict.stack(buffer)
    res['meta'] = np.array([S, B, T_buf, nodes, width, depth], np.int64)
    res['buf_board'] = np_(stacked.worlds.board); res['buf_seats'] = np_(stacked.worlds.seats)
    for k in ('logits', 'prior', 'v', 'n_sims', 'n_leaves', 'actions'):
        res['buf_dec_' + k] = np_(stacked.decisions[k])
    res['buf_rewards'] = np_(stacked.transitions.rewards); res['buf_terminal'] = np_(stacked.transitions.terminal)
    for k, v in net.state_dict().items():
        res['net_state::' + k] = np_(v)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        chunk, rest = main.as_chunk(buffer, B * 3)           # batch_size 3B: the three oldest steps are dropped
        res['reward_to_go'] = np_(chunk.reward_to_go); res['rest_len'] = np.array(len(rest))
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        scaler = torch.cuda.amp.GradScaler()
        torch.manual_seed(43)
        idxs = (torch.randint(T_buf, (B,)), torch.arange(B))  # main.py:168
        res['idx_t'] = np_(idxs[0])
        for step in range(3):
            main.optimize(net, scaler, opt, chunk[idxs])
            res[f'step{step}_policy_loss'] = np_(recorded['loss.policy'][-1][0]); res[f'step{step}_value_loss'] = np_(recorded['loss.value'][-1][0])
            for k, v in net.state_dict().items():
                res[f'step{step}_state::' + k] = np_(v)
    np.savez_compressed(os.path.join(out, 'learner_5x5.npz'), **res)
    print('learner', {k: float(res[k]) for k in res if k.endswith('_loss')})


# --------------------------------------------------------------------------------------------------------------
# 11. Config 5's EVEN board sizes (round 5): whole searches at 4x4, 6x6, 8x8, 10x10 (the block counts of the fused descent's
#     policy evaluation change at these sizes) and arena.evaluate at 2048 envs -- same recipes as gen_search / gen_arena, new files,
#     so that the files of earlier rounds stay byte for byte what they were.
# --------------------------------------------------------------------------------------------------------------
def gen_search_even(mcts_mod, hex_, networks, mcuda, out):
    # (name, S, B, T, width, depth, moves, seed, premix moves)
    configs = [
        ('search_4x4', 4, 48, 16, 16, 2, 4, 21, 3),
        ('search_6x6', 6, 32, 48, 32, 2, 2, 22, 12),
        ('search_8x8', 8, 32, 64, 32, 2, 2, 23, 21),
        ('search_10x10', 10, 24, 64, 32, 2, 2, 24, 33),
    ]
    for name, S, B, T, width, depth, moves, seed, mix in configs:
        res, _ = run_search_fixture(mcts_mod, hex_, networks, mcuda, S, B, T, width, depth, moves, seed, set(), mix)
        np.savez_compressed(os.path.join(out, name + '.npz'), **res)
        print(name, os.path.getsize(os.path.join(out, name + '.npz')) // 1024, 'KiB')


def gen_arena_even(hex_, out):
    common, neural = import_reference_arena()
    res = {}
    for S in (4, 6, 8, 10):
        start = premixed_worlds(hex_, 2048, S, S * S // 3, 500 + S)
        res[f'S{S}_board'] = np_(start.board); res[f'S{S}_seats'] = np_(start.seats)
        results = common.evaluate(start.clone(), {'front': EdgeAgent(False, 1), 'back': EdgeAgent(True, 0)})
        for i, r in enumerate(results):
            res[f'S{S}_r{i}_names'] = np.array(r.names)
            res[f'S{S}_r{i}_wins'] = np.array(r.wins); res[f'S{S}_r{i}_moves'] = np.array(r.moves); res[f'S{S}_r{i}_games'] = np.array(r.games)
        print('arena', S, [(r.names, r.wins, r.moves) for r in results])
    np.savez_compressed(os.path.join(out, 'arena_even.npz'), **res)


# --------------------------------------------------------------------------------------------------------------
# 12. validation.MonteCarloAgent and validation.SequentialMatrix (boardlaw/validation.py:32-77, 213-278; round 5).
#     SequentialMatrix: dilemma() and antisymmetric() stepped with seeded actions, every field of every step on record.
#     MonteCarloAgent: run on the reference's Hex (3x3, 4x4) and on SequentialMatrix with every Categorical draw it makes on
#     record IN CALL ORDER (the rollouts' uniform draws over the valid actions, then the final draw from the logits), so that a
#     replay needs no random stream of its own -- the GPU's generator is not the CPU's.
# --------------------------------------------------------------------------------------------------------------
def gen_validation(validation, hex_, out):
    res = {}
    for kind in ('dilemma', 'antisymmetric'):
        torch.manual_seed(800)
        world = getattr(validation.SequentialMatrix, kind)(n_envs=6, device='cpu')
        rec = {k: [] for k in ('seats', 'moves', 'obs', 'valid', 'actions', 'rewards', 'terminal')}
        res[f'{kind}_payoffs'] = np_(world.payoffs)
        for t in range(7):
            actions = torch.randint(2, (6,))
            rec['seats'].append(np_(world.seats)); rec['moves'].append(np_(world.moves)); rec['obs'].append(np_(world.obs)); rec['valid'].append(np_(world.valid))
            rec['actions'].append(np_(actions))
            world, trans = world.step(actions)
            rec['rewards'].append(np_(trans.rewards)); rec['terminal'].append(np_(trans.terminal))
        rec['seats'].append(np_(world.seats)); rec['moves'].append(np_(world.moves))
        for k, v in rec.items():
            res[f'{kind}_{k}'] = np.stack(v)

    draws = []
    orig_sample = torch.distributions.Categorical.sample

    def sample(self, shape=torch.Size()):
        d = orig_sample(self, shape)
        draws.append(d.clone())
        return d
    torch.distributions.Categorical.sample = sample
    try:
        cases = [('hex3_b1', lambda: premixed_worlds(hex_, 1, 3, 2, 810), 6, 1.), ('hex4_b1', lambda: premixed_worlds(hex_, 1, 4, 5, 811), 9, 2.),
                 ('hex3_b5', lambda: premixed_worlds(hex_, 5, 3, 3, 812), 4, 1.),
                 ('matrix_b1', lambda: validation.SequentialMatrix.dilemma(n_envs=1, device='cpu'), 7, 1.)]
        for name, make, n_rollouts, temperature in cases:
            world = make().clone()        # a fresh object: premixed_worlds assigns into its worlds in place, which leaves the reference's cached obs / valid stale
            del draws[:]
            torch.manual_seed(820)
            d = validation.MonteCarloAgent(n_rollouts, temperature)(world)
            if hasattr(world, 'board'):
                res[f'mc_{name}_board'] = np_(world.board); res[f'mc_{name}_seats'] = np_(world.seats)
            res[f'mc_{name}_meta'] = np.array([n_rollouts, temperature])
            res[f'mc_{name}_logits'] = np_(d.logits); res[f'mc_{name}_actions'] = np_(d.actions); res[f'mc_{name}_v'] = np_(d.v)
            res[f'mc_{name}_draw_sizes'] = np.array([len(x) for x in draws]); res[f'mc_{name}_draws'] = np.concatenate([np_(x) for x in draws])
            print('montecarlo', name, len(draws), 'draws', d.logits.shape, np_(d.v))
    finally:
        torch.distributions.Categorical.sample = orig_sample
    np.savez_compressed(os.path.join(out, 'validation.npz'), **res)


if __name__ == '__main__':
    only = set(sys.argv[1:])          # e.g. `make_golden.py arena learner wide`: only those; no arguments: everything
    want = lambda name: not only or name in only
    mcts_mod, hex_, networks, validation, mcuda, hcuda = import_reference()
    out = HERE if not VARIANT else os.path.join('/tmp', 'golden' + VARIANT)
    os.makedirs(out, exist_ok=True)
    if want('hex'): gen_hex(hex_, hcuda.module(), out)
    if want('search'): gen_search(mcts_mod, hex_, networks, mcuda, out)
    if want('wide'): gen_search_wide(mcts_mod, hex_, networks, mcuda, out)
    if want('toy'): gen_toy(mcts_mod, validation, out)
    if want('exp'): gen_exp(out)
    if want('snapshot'): gen_snapshot(mcts_mod, hex_, networks, out)
    if want('learner'): gen_learner(mcts_mod, hex_, networks, out)
    if want('arena'): gen_arena(hex_, out)
    if want('rollout'): gen_rollout(hex_, out)
    if want('solitaire'): gen_solitaire(hex_, out)
    if want('even'): gen_search_even(mcts_mod, hex_, networks, mcuda, out); gen_arena_even(hex_, out)
    if want('validation'): gen_validation(validation, hex_, out)


# tests/golden/learning.npz: produced by calling the reference's boardlaw.learning.reward_to_go / present_value on seeded
# random buffers (T=16, B=12, S=2); see the snippet in the commit that added it (same import stubs as above).
