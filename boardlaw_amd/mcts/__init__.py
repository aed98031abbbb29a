"""Vectorised MCTS over thousands of parallel games: `MCTS`, `mcts()`, `MCTSAgent`, `DummyAgent`.

Same surface, array names, layouts and random-draw order as boardlaw/mcts/__init__.py:13-257; the tree is a set of
contiguous SoA tensors over (B envs, T node slots):

    tree.children (B,T,A) i16 = -1     tree.parents, tree.relation (B,T) i16 = -1
    decisions.logits (B,T,A) f16 = NaN decisions.v (B,T,S) f16 = NaN
    stats.n (B,T) i16 = 0              stats.w (B,T,S) f16 = 0
    transitions.rewards (B,T,S) f16    transitions.terminal (B,T) bool
    worlds: the env's own arrays stacked to (B,T,...)

Two execution paths produce identical arrays:
  * generic  -- any world type (the toy envs of validation.py, or Hex with fused=False): the reference's own sequence
    of native calls (`cuda.descend`, `cuda.backup`, `cuda.root`) with torch indexing for the expansion glue;
  * fused    -- Hex worlds: one simulation is `bl_sim_expand` -> network -> `bl_sim_backup` on the arrays above, no
    host sync, `sim` kept on the host, capturable in a HIP graph.
"""
import ctypes

import numpy as np
import torch
import torch.distributions

from .. import _native, arrdict
from .. import profiling
from . import cuda


class TorchRng:
    """The reference's three random draws, in its shapes/dtypes/order, from torch's default generator."""

    # validate_args=False: the reference's default argument validation costs a device->host sync per draw and does not
    # touch the generator, so the draws are the same.
    def dirichlet(self, alpha, shape):
        return torch.distributions.Dirichlet(alpha, validate_args=False).sample(shape)     # mcts/__init__.py:16-18

    def rand_like(self, x):
        return torch.rand_like(x)                                           # mcts/cpp/cuda.cu:191

    def categorical(self, logits):
        return torch.distributions.Categorical(logits=logits, validate_args=False).sample()  # mcts/__init__.py:221


def torch_rand_geometry(numel, device):
    """(threads, loops) of torch's uniform kernel for `numel` elements on `device` (ATen/native/cuda/DistributionTemplates.h:
    calc_execution_policy with block 256, unroll 4): what decides which Philox counter feeds which element."""
    props = torch.cuda.get_device_properties(device)
    grid = min(props.multi_processor_count * (props.max_threads_per_multi_processor // 256), (numel + 255) // 256)
    threads = 256 * grid
    return threads, (numel - 1) // (4 * threads) + 1


class MoveRng(TorchRng):
    """The reference's random draws from the same torch generator, STREAM-IDENTICAL to TorchRng -- same seed, same numbers, same
    generator offset afterwards -- with the T-1 descend uniforms of a move produced by ONE launch (bl_rand_block) instead of T-1
    `rand_like` calls: the kernel evaluates the Philox counters those calls would have used (call c of the move sees the
    generator offset advanced by c calls) and the generator is advanced by what they would have consumed.  The Dirichlet and the
    Categorical draw stay torch's own.  Shapes the block cannot serve (non-f16, CPU) fall back to `rand_like` call by call.
    tests/test_rng_stream.py: the block equals the stacked rand_like tensors, and a seeded search is the TorchRng search."""

    def __init__(self, generator=None):
        """generator: a torch.Generator on the search's device, or None for torch's default one.  Searches that replay
        captured moves CONCURRENTLY (several actors on one GPU, each on its own stream) need a generator each: a captured
        graph reads its Philox offset from a tensor the generator owns and refills before every replay, so two graphs on
        one generator race for it (and, being the same graph, would draw the same numbers)."""
        self.block, self.i, self.generator, self.expected, self.slots_upto_call = None, 0, generator, 0, False

    def start(self, n_calls, like=None, slots_upto_call=False):
        """Announces that `n_calls` rand_like draws of one shape follow (a move's descents).  Nothing is drawn yet: the
        reference's Dirichlet comes first in the stream (MCTS.initialize), the block is cut at the first rand_like.
        slots_upto_call: the consumer is a search's descents on (B,T) tensors -- descend #c+1 can only read slots t <= c (the
        nodes that exist), so the block kernel leaves the others unwritten (half the Philox work; the written ones unchanged)."""
        self.block, self.i, self.expected, self.slots_upto_call = None, 0, n_calls, slots_upto_call

    def _draw_block(self, x):
        threads, loops = torch_rand_geometry(x.numel(), x.device)
        gen = self.generator if self.generator is not None else torch.cuda.default_generators[x.device.index if x.device.index is not None else torch.cuda.current_device()]
        with torch.cuda.device(x.device):
            seed, offset, intragraph, captured = _native.philox_state(gen, self.expected * 4 * loops)
            block = torch.empty((self.expected,) + tuple(x.shape), dtype=torch.half, device=x.device)
            tri = x.shape[-1] if (self.slots_upto_call and x.ndim == 2) else 0
            _native.check(_native.lib().bl_rand_block(block.data_ptr(), self.expected, x.numel(), threads, loops, seed, offset, intragraph,
                                                      captured, tri, _native.stream(x.device)))
        return block

    def rand_like(self, x):
        if self.block is None and self.expected > 0 and x.is_cuda and x.dtype == torch.half:
            self.block, self.i = self._draw_block(x), 0
        if self.block is None or self.i >= self.block.shape[0] or self.block.shape[1:] != x.shape:
            return torch.rand(x.shape, dtype=x.dtype, device=x.device, generator=self.generator)
        r = self.block[self.i]
        self.i += 1
        if self.i == self.block.shape[0]:
            self.expected = 0
        return r

    def dirichlet(self, alpha, shape):
        # torch.distributions.Dirichlet(alpha).sample(shape) with this rng's generator: the same three kernels (gamma
        # variates, their sum, the clamped quotient -- ATen's _sample_dirichlet), the same use of the generator
        return torch._sample_dirichlet(alpha.expand(*shape, alpha.shape[-1]), self.generator)

    # torch's reduce kernel sums rows of fewer than 128 elements in a fixed lane layout that bl_sim_plant_root_gamma / bl_categorical
    # reproduce; from 128 on it vectorises with an order that depends on each row's address alignment: torch's own launches stay
    FUSED_MAX_ACTIONS = 127

    _layout_checked = {}        # device index -> bool: torch's reduce order on this build is the one the fused draws reproduce

    @classmethod
    def torch_layout_ok(cls, device):
        """bl_categorical and bl_sim_plant_root_gamma restate the lane layout and tree order of torch's row reductions `as this torch
        build on ROCm orders it` (csrc/bl_device.h: torch_row_sum).  A torch upgrade could change that order silently: the draws
        would stay valid but stop being bit-identical to torch's and the reference's.  So, once per device and process (outside any
        capture; the generator is put back), the two fused draws are compared with the torch launches they stand for on a small
        case; on a mismatch the fused routes are switched off for the process (FUSED_MAX_ACTIONS = 0: torch's own launches run)."""
        device = torch.device(device)
        index = device.index if device.index is not None else torch.cuda.current_device()
        if index in cls._layout_checked:
            return cls._layout_checked[index]
        if torch.cuda.is_current_stream_capturing():
            return True                                                   # nothing cached: checked at the next eager use
        cls._layout_checked[index] = True                                 # (the check below goes through the routes it guards)
        gen = torch.cuda.default_generators[index]
        state = gen.get_state()
        ok = True
        try:
            with torch.cuda.device(index):
                dev = torch.device('cuda', index)
                for A in (25, 81):
                    gen.manual_seed(1234 + A)
                    logits = (torch.randn(64, A, device=dev) * 2).half()            # (device-side draws: the CPU generator is not touched)
                    logits[torch.rand(64, A, device=dev) < 0.3] = -float('inf')
                    logits[:, 0] = 0.5
                    rng = cls()
                    gen.manual_seed(99); a = rng.categorical_f16(logits); off_a = gen.get_offset()
                    gen.manual_seed(99); b = rng.categorical(logits.float()); off_b = gen.get_offset()
                    ok = ok and bool(torch.equal(a, b)) and off_a == off_b
                # the Dirichlet's normalisation: gamma variates normalised inside the root's launch against torch's finished sample
                from .. import hex as hexmod
                world = hexmod.Hex.initial(16, 5, device=dev)
                rows = []
                for fused_max in (cls.FUSED_MAX_ACTIONS, 0):
                    rng = cls(); rng.FUSED_MAX_ACTIONS = fused_max
                    m = MCTS(world, n_nodes=2, rng=rng)
                    gen.manual_seed(7)
                    alpha = torch.full((25,), 0.4, dtype=torch.float, device=dev)
                    pol, val = torch.linspace(-2, 2, 16 * 25, device=dev).reshape(16, 25).contiguous(), torch.zeros(16, device=dev)
                    draw = (rng.gamma_variates(alpha, (16,)) if fused_max else rng.dirichlet(alpha, (16,))).float().contiguous()
                    plant = _native.lib().bl_sim_plant_root_gamma if fused_max else _native.lib().bl_sim_plant_root
                    _native.check(plant(ctypes.byref(m._search), pol.data_ptr(), val.data_ptr(), world.valid.contiguous().data_ptr(),
                                        world.seats.int().contiguous().data_ptr(), draw.data_ptr(), 0.25, _native.stream(dev)))
                    rows.append(m.decisions.logits[:, 0].clone())
                ok = ok and bool(torch.equal(rows[0].view(torch.int16), rows[1].view(torch.int16)))
        finally:
            gen.set_state(state)
        if not ok:
            import warnings
            warnings.warn('boardlaw_amd: this torch build orders its row reductions differently from the layout bl_categorical / '
                          'bl_sim_plant_root_gamma reproduce; the fused draws are switched off (torch\'s own launches run, same stream)')
            cls.FUSED_MAX_ACTIONS = 0
        cls._layout_checked[index] = ok
        return ok

    def gamma_variates(self, alpha, shape):
        """The standard-gamma variates torch's Dirichlet sampler starts from (at::_sample_dirichlet = _standard_gamma, then sum and
        clamped quotient): the same kernel, the same use of the generator; bl_sim_plant_root_gamma does the rest inside the root's
        launch with torch's rounding and summation order -- same bits as dirichlet(), two launches fewer per move."""
        return torch._standard_gamma(alpha.expand(*shape, alpha.shape[-1]), self.generator)

    def categorical_f16(self, logits):
        """categorical(logits.float()) for (B,A) f16 device logits: torch's own exponential_ draw, then bl_categorical -- the
        normalisation, softmax, quotient and argmax of the launches below as one kernel with torch's rounding points and summation
        orders (tests/test_rng_stream.py::test_fused_draws_equal_torchs).  Same actions, same use of the generator."""
        if not (logits.is_cuda and logits.dtype == torch.half and logits.ndim == 2 and logits.shape[-1] <= self.FUSED_MAX_ACTIONS
                and (self.FUSED_MAX_ACTIONS == 0 or self.torch_layout_ok(logits.device))):
            return self.categorical(logits.float())
        B, A = logits.shape
        q = torch.empty((B, A), dtype=torch.float, device=logits.device).exponential_(1, generator=self.generator)
        actions = torch.empty((B,), dtype=torch.long, device=logits.device)
        with torch.cuda.device(logits.device):
            _native.check(_native.lib().bl_categorical(logits.contiguous().data_ptr(), q.data_ptr(), actions.data_ptr(), B, A, _native.stream(logits.device)))
        return actions

    def categorical(self, logits):
        """torch.distributions.Categorical(logits=logits).sample() with this rng's generator, minus argument checking: the
        normalisation and softmax of Categorical.__init__/probs, then what torch.multinomial does for ONE draw with replacement
        once it has validated its input (aten/src/ATen/native/Distributions.cpp: q = exponential_(1) noise, argmax(probs / q)).
        Same kernels on the same values and the same use of the generator -- the same actions as TorchRng.categorical -- without
        multinomial's eight validation launches (two of them device-side asserts)."""
        probs = torch.softmax(logits - logits.logsumexp(-1, keepdim=True), -1)
        flat = probs.reshape(-1, probs.shape[-1])
        q = torch.empty_like(flat).exponential_(1, generator=self.generator)
        torch.div(flat, q, out=q)
        return q.argmax(-1).reshape(probs.shape[:-1])


class FastRng(MoveRng):
    """MoveRng with the two per-move draws as ONE generator launch each plus the library -- the same distributions as the
    reference's calls, NOT its stream (a seeded run sees different numbers than TorchRng/MoveRng): opt-in, for throughput
    studies only."""

    def gamma(self, alpha, shape):
        """Unnormalised Dirichlet draw: the Gamma(alpha, 1) variates torch's Dirichlet sampler starts from
        (torch._sample_dirichlet = standard_gamma / sum).  bl_sim_plant_root normalises over the valid actions anyway
        (mcts/__init__.py:19-22), so the intermediate normalisation over all actions is dropped."""
        return torch._standard_gamma(alpha.expand(*shape, alpha.shape[-1]), self.generator)

    def draw_actions(self, probs):
        """Categorical(probs / probs.sum()) by inverse CDF from one torch.rand per env (bl_draw_actions)."""
        B, A = probs.shape
        u = torch.rand((B,), dtype=torch.float, device=probs.device, generator=self.generator)
        actions = torch.empty((B,), dtype=torch.long, device=probs.device)
        with torch.cuda.device(probs.device):
            _native.check(_native.lib().bl_draw_actions(probs.contiguous().data_ptr(), u.data_ptr(), actions.data_ptr(), B, A,
                                                        _native.stream(probs.device)))
        return actions


def dirichlet_noise(logits, valid, eps, alpha_scale=10, rng=None):
    """mcts/__init__.py:13-24: mix a Dirichlet(alpha_scale/A) draw over the valid actions into the root prior."""
    rng = rng or TorchRng()
    alpha = torch.full((valid.shape[-1],), alpha_scale / logits.size(-1), dtype=torch.float, device=logits.device)
    draw = rng.dirichlet(alpha, logits.shape[:-1])
    draw[~valid] = 0.
    draw = draw / draw.sum(-1, keepdims=True)
    return (logits.exp() * (1 - eps) + draw * eps).log()


class LeafWorlds:
    """What the network sees of the freshly expanded leaves on the fused path: obs/valid/seats were written by
    bl_sim_expand; the boards stay in the tree and are gathered only if someone asks."""

    def __init__(self, search, leaves, obs, valid, seats):
        self._search, self._leaves = search, leaves
        self.obs, self.valid, self.seats = obs, valid, seats
        self.n_envs, self.n_seats, self.device = obs.shape[0], 2, obs.device

    @property
    def board(self):
        return self._search.worlds.board[self._search.envs, self._leaves.long()]


class MCTS:

    def __init__(self, world, n_nodes=64, c_puct=1 / 16, noise_eps=.25, alpha_scale=10, fused=None, rng=None,
                 count=False, obs_half=False, qrange_sync=None, fuse_finish=True, n_active=None, lazy=False):
        """c_puct high: concentrates on prior; c_puct low: concentrates on value (mcts/__init__.py:29-33).
        n_active (fused path): a one-element int32 DEVICE tensor -- only the first n_active[0] envs of `world` are searched, the
        rest sit the simulations out and add nothing to the q-range (bl_search_t.n_active): what lets one captured move of B
        envs serve masked calls of any size <= B.
        lazy (fused path): the (B,T,A) arrays get their reset values slot by slot from the simulations instead of from one 105 MB
        fill per move (bl_tune_t.lazy_init); identical arrays once all n_nodes - 1 simulations have run -- what mcts() does."""
        from .. import hex as hexmod
        self.device = world.device
        self.n_envs = world.n_envs
        self.n_nodes = n_nodes
        self.n_seats = world.n_seats
        assert n_nodes > 0, 'MCTS requires at least one node'
        self.n_actions = int(np.prod(world.action_space))
        self.noise_eps, self.alpha_scale = noise_eps, alpha_scale
        self.rng = rng or TorchRng()
        # optional callable(state_row) run after every backup on the q-range row the next descent will read, e.g.
        # parallel.allreduce_qrange: env shards on several GPUs then normalise q over ALL envs like one big batch
        self.qrange_sync = qrange_sync
        # network forward + finish as one launch (bl_sim_infer_finish) when the network offers its packed weights
        self.fuse_finish = fuse_finish
        # the fused kernels hard-code two-seat Hex; its one-player variants (hex.Solitaire) take the generic path
        self.fused = (isinstance(world, hexmod.Hex) and world.n_seats == 2) if fused is None else fused
        if self.fused and not isinstance(world, hexmod.Hex):
            raise ValueError('The fused path is Hex-only')
        if n_active is not None and not self.fused:
            # the generic path searches every row it is given: padded rows would be searched and feed the batch-global q range
            raise ValueError('n_active (masked captured moves) needs the fused Hex path; call the agent with pad=False for other worlds')
        B, T, A, S, dev = self.n_envs, n_nodes, self.n_actions, self.n_seats, self.device
        self._envs = None
        self.sim = 0
        self.c_puct = torch.full((B,), c_puct, device=dev, dtype=torch.half)      # per-env and writable: not a shared constant
        self._root_world = world

        if self.fused:
            _native.require_device(world.board)
            e = lambda *shape, dtype: torch.empty(shape, device=dev, dtype=dtype)
            bs = world.boardsize
            self.tree = arrdict.arrdict(children=e(B, T, A, dtype=torch.short), parents=e(B, T, dtype=torch.short),
                                        relation=e(B, T, dtype=torch.short))
            self.worlds = type(world)(board=e(B, T, bs, bs, dtype=torch.uint8), seats=e(B, T, dtype=torch.int))
            self.transitions = arrdict.arrdict(rewards=e(B, T, S, dtype=torch.half), terminal=e(B, T, dtype=torch.bool))
            self.decisions = arrdict.arrdict(logits=e(B, T, A, dtype=torch.half), v=e(B, T, S, dtype=torch.half))
            self.stats = arrdict.arrdict(n=e(B, T, dtype=torch.short), w=e(B, T, S, dtype=torch.half))
            self._qrange = e(T + 1, _native.QRANGE_WORDS, dtype=torch.int32)
            self._exp = _native.exp_table(dev)
            self._leaves = e(B, dtype=torch.short)
            self._obs = e(B, bs, bs, 2, dtype=torch.half if obs_half else torch.float)
            self._valid = e(B, A, dtype=torch.bool)
            self._leaf_seats = e(B, dtype=torch.int)
            self._path = e(B, T + 2, dtype=torch.short)
            # compacted policy rows (include/boardlaw_amd.h: bl_search_t.cpi/cca/nk): what a descent level actually reads
            self._cpi = e(B, T, A, dtype=torch.float)
            self._cca = e(B, T, A, dtype=torch.int32)
            self._nk = e(B, T, dtype=torch.short)
            self._fav = e(B, T, dtype=torch.short)
            self.counters = torch.zeros((B, 12), dtype=torch.int64, device=dev) if count else None
            self._search = _native.Search(
                logits=self.decisions.logits.data_ptr(), v=self.decisions.v.data_ptr(), w=self.stats.w.data_ptr(),
                n=self.stats.n.data_ptr(), children=self.tree.children.data_ptr(), parents=self.tree.parents.data_ptr(),
                relation=self.tree.relation.data_ptr(), rewards=self.transitions.rewards.data_ptr(),
                terminal=self.transitions.terminal.data_ptr(), boards=self.worlds.board.data_ptr(),
                seats=self.worlds.seats.data_ptr(), c_puct=self.c_puct.data_ptr(), qrange=self._qrange.data_ptr(),
                exp_table=self._exp.data_ptr(), B=B, T=T, boardsize=bs, obs_f16=int(obs_half),
                path=self._path.data_ptr(), cpi=self._cpi.data_ptr(), cca=self._cca.data_ptr(), nk=self._nk.data_ptr(),
                fav=self._fav.data_ptr(), tune=_native.tune(dev))
            self._search.tune.lazy_init = int(bool(lazy))
            if n_active is not None:
                assert n_active.dtype == torch.int32 and n_active.numel() == 1 and n_active.device == world.board.device
                self._n_active = n_active                      # kept alive: the kernels read it through the pointer
                self._search.n_active = n_active.data_ptr()
            with torch.cuda.device(dev):
                _native.check(_native.lib().bl_sim_init(ctypes.byref(self._search), world.board.contiguous().data_ptr(),
                                                        world.seats.int().contiguous().data_ptr(), _native.stream(dev)))
        else:
            f = lambda shape, value, dtype: torch.full(shape, value, device=dev, dtype=dtype)
            self.tree = arrdict.arrdict(children=f((B, T, A), -1, torch.short), parents=f((B, T), -1, torch.short),
                                        relation=f((B, T), -1, torch.short))
            self.worlds = arrdict.stack([world for _ in range(T)], 1)
            self.transitions = arrdict.arrdict(rewards=f((B, T, S), 0., torch.half), terminal=f((B, T), False, torch.bool))
            self.decisions = arrdict.arrdict(logits=f((B, T, A), np.nan, torch.half), v=f((B, T, S), np.nan, torch.half))
            self.stats = arrdict.arrdict(n=f((B, T), 0, torch.short), w=f((B, T, S), 0., torch.half))
            self.worlds[:, 0] = world

    @property
    def envs(self):
        if self._envs is None:
            self._envs = torch.arange(self.n_envs, device=self.device)
        return self._envs

    # ------------------------------------------------------------------ mcts/__init__.py:72-80
    @profiling.roctx
    def initialize(self, network):
        world = self._root_world if self.fused else self.worlds[:, 0]
        if (self.fused and self.fuse_finish and hasattr(network, 'root_raw') and world.board.is_cuda
                and type(getattr(network.model, 'policy', None)).__name__ == 'MaskedOutput'):
            # the network's Linears in fp32, then heads + dirichlet noise + store as ONE launch (bl_sim_plant_root)
            assert self.sim == 0
            policy_raw, value_raw = network.root_raw(world)
            policy_raw, value_raw = policy_raw.float().contiguous(), value_raw.float().contiguous()
            valid = world.valid.contiguous()
            alpha = _constant(self.n_actions, self.alpha_scale / self.n_actions, self.device, torch.float)
            plant = _native.lib().bl_sim_plant_root
            if hasattr(self.rng, 'gamma'):
                draw = self.rng.gamma(alpha, (self.n_envs,)).float().contiguous()      # FastRng: not the reference's normalisation
            elif (hasattr(self.rng, 'gamma_variates') and self.n_actions <= getattr(self.rng, 'FUSED_MAX_ACTIONS', 0)
                  and (not hasattr(self.rng, 'torch_layout_ok') or self.rng.torch_layout_ok(self.device))
                  and self.n_actions <= getattr(self.rng, 'FUSED_MAX_ACTIONS', 0)):
                # the reference's draw (mcts/__init__.py:16-18) with its normalisation inside the root's launch: same bits
                draw = self.rng.gamma_variates(alpha, (self.n_envs,)).float().contiguous()
                plant = _native.lib().bl_sim_plant_root_gamma
            else:
                draw = self.rng.dirichlet(alpha, (self.n_envs,)).float().contiguous()  # mcts/__init__.py:16-18
            with torch.cuda.device(self.device):
                _native.check(plant(ctypes.byref(self._search), policy_raw.data_ptr(), value_raw.data_ptr(),
                                    valid.data_ptr(), world.seats.int().contiguous().data_ptr(),
                                    draw.data_ptr(), float(self.noise_eps), _native.stream(self.device)))
            self.sim = 1
            return
        with torch.no_grad():
            decisions = network(world)
        self.plant_root(dirichlet_noise(decisions.logits, world.valid, self.noise_eps, self.alpha_scale, self.rng), decisions.v)

    def plant_root(self, logits, v):
        """Stores the root evaluation (what initialize does after the network call); also the entry point for replaying
        a recorded search."""
        assert self.sim == 0
        self.decisions.logits[:, 0] = logits
        self.decisions.v[:, 0] = v
        if self.fused:
            with torch.cuda.device(self.device):
                _native.check(_native.lib().bl_sim_compact(ctypes.byref(self._search), None, _native.stream(self.device)))
        self.sim = 1

    # ------------------------------------------------------------------ generic path, mcts/__init__.py:82-140
    def _cuda(self):
        return cuda.mcts(self.decisions.logits, self.stats.w, self.stats.n, self.c_puct, self.worlds.seats,
                         self.transitions.terminal, self.tree.children)

    @profiling.roctx
    def descend(self):
        m = self._cuda()
        result = cuda.descend(m, self.rng.rand_like(m.logits[:, :, 0]))
        return result.parents.long(), result.actions.long()

    @profiling.roctx
    def backup(self, leaves):
        bk = cuda.Backup(v=self.decisions.v, w=self.stats.w, n=self.stats.n, rewards=self.transitions.rewards,
                         parents=self.tree.parents, terminal=self.transitions.terminal)
        cuda.backup(bk, leaves.short())

    def _simulate_generic(self, network):
        parents, actions = self.descend()
        # a descent that stopped on a terminal node re-visits it instead of creating a node
        leaves = self.tree.children[self.envs, parents, actions].long()
        leaves[leaves == -1] = self.sim
        self.tree.children[self.envs, parents, actions] = leaves.short()
        self.tree.parents[self.envs, leaves] = parents.short()
        self.tree.relation[self.envs, leaves] = actions.short()

        world, transition = self.worlds[self.envs, parents].step(actions)
        self.worlds[self.envs, leaves] = world
        self.transitions.rewards[self.envs, leaves] = transition.rewards.half()
        self.transitions.terminal[self.envs, leaves] = transition.terminal

        with torch.no_grad(), torch.autocast('cuda', enabled=(self.device.type == 'cuda')):
            decisions = network(world)
        self.decisions.logits[self.envs, leaves] = decisions.logits.half()
        self.decisions.v[self.envs, leaves] = decisions.v.half()
        self.backup(leaves)

    # ------------------------------------------------------------------ fused path
    def _simulate_fused(self, network):
        L, s, dev = _native.lib(), ctypes.byref(self._search), self.device
        rands = self.rng.rand_like(self.decisions.logits[:, :, 0])
        assert rands.is_contiguous() and rands.dtype == torch.half
        with torch.cuda.device(dev):
            st = _native.stream(dev)
            if self.counters is None:
                _native.check(L.bl_sim_expand(s, self.sim, rands.data_ptr(), self._leaves.data_ptr(), self._obs.data_ptr(),
                                              self._valid.data_ptr(), self._leaf_seats.data_ptr(), st))
            else:
                _native.check(L.bl_sim_expand_counted(s, self.sim, rands.data_ptr(), self._leaves.data_ptr(),
                                                      self._obs.data_ptr(), self._valid.data_ptr(),
                                                      self._leaf_seats.data_ptr(), self.counters.data_ptr(), st))
            world = LeafWorlds(self, self._leaves, self._obs, self._valid, self._leaf_seats)
            if getattr(network, 'leaf_fp32', False) and hasattr(network, 'root_raw'):
                # the exact mode (networks.Inference(precision='fp32')): the leaf is evaluated like the root -- fp32 Linears, fp32
                # heads, only the stores rounded to f16 -- which is what the reference's CPU runs do (mcts/__init__.py:131-136)
                policy_raw, value_raw = network.root_raw(world)
                policy_raw, value_raw = policy_raw.float().contiguous(), value_raw.float().contiguous()
                assert policy_raw.shape == (self.n_envs, self.n_actions) and value_raw.shape == (self.n_envs,)
                _native.check(L.bl_sim_finish_f32(s, self.sim, self._leaves.data_ptr(), policy_raw.data_ptr(), value_raw.data_ptr(),
                                                  self._valid.data_ptr(), self._leaf_seats.data_ptr(), st))
                return
            fp = network.fused_params(self.n_envs) if (self.fuse_finish and hasattr(network, 'fused_params')) else None
            if (fp is not None and self._obs.dtype == torch.half and self.n_nodes <= 64 and self.n_actions <= 128
                    and fp['W'] >= 256 and fp['K0'] == 2 * self.n_actions and fp['NH'] == self.n_actions + 1):
                # one launch: the network's Linears, its heads, the store, the backup and the next q range
                _native.check(L.bl_sim_infer_finish(s, self.sim, self._leaves.data_ptr(), self._obs.data_ptr(),
                                                    self._valid.data_ptr(), self._leaf_seats.data_ptr(), fp['w0'], fp['b0'],
                                                    fp['wb'], fp['bb'], fp['al'], fp['wh'], fp['bh'], fp['W'], fp['D'],
                                                    fp['K0pad'], fp['NHpad'], st))
                return
            if hasattr(network, 'raw'):
                # the network hands over its pre-head outputs; bl_sim_finish applies the heads, stores, backs up
                with torch.no_grad(), torch.autocast('cuda', enabled=True):
                    policy_raw, value_raw = network.raw(world)
                policy_raw, value_raw = policy_raw.contiguous(), value_raw.contiguous()
                assert policy_raw.dtype == torch.half and value_raw.dtype == torch.half
                assert policy_raw.shape == (self.n_envs, self.n_actions) and value_raw.shape == (self.n_envs,)
                _native.check(L.bl_sim_finish(s, self.sim, self._leaves.data_ptr(), policy_raw.data_ptr(), value_raw.data_ptr(),
                                              self._valid.data_ptr(), self._leaf_seats.data_ptr(), st))
                return
            with torch.no_grad(), torch.autocast('cuda', enabled=True):
                decisions = network(world)
            logits, v = decisions.logits.contiguous(), decisions.v.contiguous()
            kinds = {torch.float: 0, torch.half: 1}
            if logits.dtype not in kinds or v.dtype not in kinds:
                logits, v = logits.float(), v.float()
            assert logits.shape == (self.n_envs, self.n_actions) and v.shape == (self.n_envs, 2)
            _native.check(L.bl_sim_backup(s, self.sim, self._leaves.data_ptr(), logits.data_ptr(), kinds[logits.dtype],
                                          v.data_ptr(), kinds[v.dtype], st))

    @profiling.roctx
    def simulate(self, network):
        if self.sim >= self.n_nodes:
            raise ValueError('Called simulate more times than were declared in the constructor')
        if self.fused:
            self._simulate_fused(network)
            if self.qrange_sync is not None:
                self.qrange_sync(self._qrange[self.sim + 1])
        else:
            self._simulate_generic(network)
        self.sim += 1

    # ------------------------------------------------------------------ mcts/__init__.py:142-152
    def root_probs(self, with_logits=False):
        if self.fused:
            probs = torch.empty((self.n_envs, self.n_actions), dtype=torch.half, device=self.device)
            logits = torch.empty_like(probs) if with_logits else None
            table = _native.log_table(self.device) if with_logits else None
            with torch.cuda.device(self.device):
                _native.check(_native.lib().bl_sim_root(ctypes.byref(self._search), self.sim, probs.data_ptr(),
                                                        table.data_ptr() if with_logits else None,
                                                        logits.data_ptr() if with_logits else None, _native.stream(self.device)))
            return (probs, logits) if with_logits else probs
        probs = cuda.root(self._cuda())
        return (probs, None) if with_logits else probs

    @profiling.roctx
    def root(self):
        r, logits = self.root_probs(with_logits=True)
        self._root_probs = r
        # the reference takes r.log() on the device and r.float().log().half() on the host (mcts/__init__.py:147);
        # here the host's values are looked up per f16 bit pattern so both paths agree bit for bit (inside bl_sim_root on
        # the fused path)
        if logits is None:
            logits = _native.log_table(r.device)[r.view(torch.int16).long() & 0xffff]
        return arrdict.arrdict(
            logits=logits,
            prior=self.decisions.logits[:, 0],
            v=self.decisions.v[:, 0])

    def n_leaves(self):
        """Nodes that exist and have no child (mcts/__init__.py:151-152: `(children == -1).all(-1) & (parents != -1)`).
        A node has a child exactly when some node names it as its parent, so this reads the two (B,T) arrays instead of
        the (B,T,A) children array (42 MB at 9x9/4096/64)."""
        if self.fused:
            out = torch.empty((self.n_envs,), dtype=torch.long, device=self.device)
            with torch.cuda.device(self.device):
                _native.check(_native.lib().bl_sim_n_leaves(ctypes.byref(self._search), out.data_ptr(), _native.stream(self.device)))
            return out
        parents = self.tree.parents
        exists = parents != -1
        n_children = torch.zeros(parents.shape, dtype=torch.int32, device=parents.device)
        n_children.scatter_add_(1, parents.clamp(min=0).long(), exists.int())   # += 1 at parents[b,t] for every existing t
        return (exists & (n_children == 0)).sum(-1)


def mcts(worlds, network, **kwargs):
    kwargs.setdefault('obs_half', bool(getattr(network, 'wants_half_obs', False)))
    kwargs.setdefault('lazy', True)         # a whole search follows: every slot gets its reset values on the way
    m = MCTS(worlds, **kwargs)
    if hasattr(m.rng, 'start') and m.n_nodes > 1:
        # announces the move's T-1 descend draws; drawn after the root's Dirichlet
        m.rng.start(m.n_nodes - 1, slots_upto_call=m.fused) if isinstance(m.rng, MoveRng) else m.rng.start(m.n_nodes - 1)
    if hasattr(network, 'refresh_if_stale') and not (worlds.device.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
        network.refresh_if_stale()     # picks up optimiser steps; a captured move is refreshed by its replayer instead
    m.initialize(network)
    for _ in range(m.n_nodes - 1):
        m.simulate(network)
    return m


_constants = {}


def _constant(n, value, device, dtype=torch.long):
    """(n,) tensor filled with `value`, built once per (n, value, dtype, device) and shared: read-only by convention
    (callers clone what they hand out).  Saves a fill launch per use inside every move."""
    key = (n, value, dtype, device.type, device.index)
    if key not in _constants:
        if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return torch.full((n,), value, dtype=dtype, device=device)
        _constants[key] = torch.full((n,), value, dtype=dtype, device=device)
    return _constants[key]


class MCTSAgent:
    """mcts/__init__.py:209-241.  Output arrdict: logits (B,A) f16, prior (B,A) f16, n_sims (B) i64, n_leaves (B) i64,
    v (B,S) f16, actions (B) i64.

    graph=True (Hex on the GPU only) captures one whole move -- tree reset, root evaluation + noise, the T-1
    expand/network/backup rounds, root read-out and the action draw -- into a HIP graph per (batch size, board size,
    eval) and replays it: the search has no host-side decision in it (the fused path keeps `sim` on the host and never
    syncs), so replay removes every launch gap.  Random draws come from torch's generator exactly as in eager mode."""

    # kwargs of the reference's MCTS (mcts/__init__.py:29): the only ones a checkpoint carries
    REFERENCE_KWARGS = ('n_nodes', 'c_puct', 'noise_eps', 'alpha_scale')
    GRAPH_CACHE_BYTES = 8 << 30     # captured moves kept alive (each owns a whole tree): least recently used go first

    MIN_CAPACITY = 64

    def __init__(self, network, graph=False, pad=True, **kwargs):
        """pad (with graph=True): a call of n envs replays a move captured for the next power of two >= n with the extra rows
        switched off on the device (MCTS n_active), so the arena's masked calls -- a different n every round
        (arena/common.py:88-93) -- share a handful of captures instead of capturing every round."""
        self.network = network
        self.kwargs = kwargs
        self.graph = graph
        self.pad = pad
        self._graphs = {}           # insertion-ordered: oldest use first

    # A padded replay equals the eager call on the same envs because torch's random kernels give element i the same number
    # whatever the tensor's size: the uniform / exponential kernels hand element i to thread i % threads, Philox block
    # i // (4 threads), component (i // threads) % 4 -- with `threads` either covering the tensor (one element per thread) or
    # capped at the chip's CUs x 2048, the same cap for every size -- and the gamma kernel keeps one curand state per thread,
    # walked in grid-stride order.  What DOES depend on the size is how far a call advances the generator: 4 x `loops`
    # (torch_rand_geometry).  So a move may be padded from n to cap rows iff every random tensor of the move -- (rows, A) for the
    # Dirichlet and the action draw, (rows, T) for the descents -- has the same `loops` at both sizes.  (Round 4 tested
    # cap x A <= threads, which ignored the (rows, T) block: advisor finding.)
    # One more size-dependent step: the Dirichlet's normalisation.  Up to 127 actions it happens inside bl_sim_plant_root_gamma, row
    # by row; from 128 on torch's own reduce kernel sums the rows, and it picks its summation order by the tensor's SHAPE -- a
    # 13x13 move padded from 1051 to 2048 rows differed from the eager move in the last bit of some root logits (found in round 5
    # by the test below's 13x13 case; round 4 padded there).  Boards from 12x12 up are captured for exactly n rows.
    def _pad_keeps_the_stream(self, n, cap, world):
        A, T = int(np.prod(world.action_space)), int(self.kwargs.get('n_nodes', 64))
        if A > MoveRng.FUSED_MAX_ACTIONS:
            return False
        return all(torch_rand_geometry(n * x, world.device)[1] == torch_rand_geometry(cap * x, world.device)[1] for x in (A, T))

    def _capacity(self, n, world=None):
        if not self.pad:
            return n
        cap = self.MIN_CAPACITY
        while cap < n:
            cap *= 2
        if world is not None and not (cap == n or self._pad_keeps_the_stream(n, cap, world)):
            return n
        return cap

    def _kwargs_key(self):
        # the reference mutates agent.kwargs in place (arena: kwargs['n_nodes'] = ...): a captured move belongs to the
        # kwargs it was captured with
        return tuple(sorted((k, v if isinstance(v, (int, float, bool, str, type(None))) else id(v)) for k, v in self.kwargs.items()))

    def _graphed(self, key, build):
        key = key + (self._kwargs_key(),)
        g = self._graphs.pop(key, None)
        if g is None:
            g = build()
            used = sum(x.nbytes for x in self._graphs.values())
            while self._graphs and used + g.nbytes > self.GRAPH_CACHE_BYTES:
                used -= self._graphs.pop(next(iter(self._graphs))).nbytes
        self._graphs[key] = g       # most recently used last
        return g

    def _move(self, world, eval, kwargs, clone=True):
        m = mcts(world, self.network, **{**self.kwargs, **kwargs})
        r = m.root()
        if eval:
            actions = r.logits.argmax(-1)
        elif hasattr(m.rng, 'draw_actions') and m.fused:
            actions = m.rng.draw_actions(m._root_probs)
        elif hasattr(m.rng, 'categorical_f16') and m.fused:
            actions = m.rng.categorical_f16(r.logits)              # the same draw as below, in two launches instead of thirteen
        else:
            actions = m.rng.categorical(r.logits.float())
        d = arrdict.arrdict(
            logits=r.logits,
            prior=r.prior,
            n_sims=_constant(m.n_envs, m.sim + 1, m.device),     # the reference's off-by-one, kept
            n_leaves=m.n_leaves(),
            v=r.v,
            actions=actions)
        # prior and v are views of the tree: detach them from it, unless the caller (a graph replayer) clones anyway
        return d.clone() if clone else d

    @profiling.roctx
    def __call__(self, world, value=True, eval=False, **kwargs):
        if not self.graph or kwargs or world.device.type != 'cuda':
            return self._move(world, eval, kwargs)
        from .. import hex as hexmod
        fused = self.kwargs.get('fused')
        fused = (isinstance(world, hexmod.Hex) and world.n_seats == 2) if fused is None else fused
        cap = self._capacity(world.n_envs, world) if fused else world.n_envs     # n_active exists on the fused path only
        key = (type(world), cap, world.boardsize, bool(eval), world.device)
        return self._graphed(key, lambda: _GraphedMove(self, world, eval, capacity=cap if (self.pad and fused) else None))(world)

    @profiling.roctx
    def play(self, world, eval=False):
        """One actor step of the self-play loop (boardlaw/main.py:176-177): decisions = agent(world); new_world,
        transition = world.step(decisions.actions).  With graph=True both halves replay as ONE captured graph (the env
        step's validity asserts -- host syncs -- are skipped: the search only ever draws legal actions).
        Returns (decisions, new_world, transition)."""
        if not self.graph or world.device.type != 'cuda':
            d = self._move(world, eval, {})
            new_world, transition = world.step(d.actions)
            return d, new_world, transition
        key = ('play', type(world), world.n_envs, world.boardsize, bool(eval), world.device)
        return self._graphed(key, lambda: _GraphedMove(self, world, eval, step=True))(world)

    def load_state_dict(self, sd):
        self.network.load_state_dict({k[len('network.'):]: v for k, v in sd.items() if k.startswith('network.')})
        self.kwargs.update({k[len('kwargs.'):]: v for k, v in sd.items() if k.startswith('kwargs.')})
        self._graphs = {}

    def state_dict(self):
        # the reference's checkpoint format (mcts/__init__.py:236-241); its load_state_dict feeds every 'kwargs.*' entry to
        # MCTS(**kwargs), so this build's own options (rng, graph, fusion switches) stay out of the checkpoint
        return {**{f'network.{k}': v for k, v in self.network.state_dict().items()},
                **{f'kwargs.{k}': v for k, v in self.kwargs.items() if k in self.REFERENCE_KWARGS}}


class _GraphedMove:
    """One captured move for a fixed (world type, B, boardsize, eval).  Inputs are copied into static buffers, the
    graph is replayed, outputs are cloned out.  Network parameters are read in place, so training steps between
    replays are seen."""

    def __init__(self, agent, world, eval, step=False, capacity=None):
        dev = world.device
        n = world.n_envs
        self.capacity = capacity if capacity is not None else n
        assert self.capacity >= n and not (step and self.capacity != n)
        if self.capacity == n:
            self.board, self.seats = world.board.clone(), world.seats.clone()
        else:                                              # rows beyond the caller's envs: empty boards, never searched
            self.board = torch.zeros((self.capacity,) + tuple(world.board.shape[1:]), dtype=world.board.dtype, device=dev)
            self.seats = torch.zeros((self.capacity,), dtype=world.seats.dtype, device=dev)
            self.board[:n] = world.board; self.seats[:n] = world.seats
        # the number of live rows, read by the search kernels on the device: rewritten before every replay
        self.n_active = torch.full((1,), n, dtype=torch.int32, device=dev) if capacity is not None else None
        self.network = agent.network
        self.step = step
        kind = type(world)
        if hasattr(self.network, 'refresh_if_stale'):
            self.network.refresh_if_stale()
        extra = {} if self.n_active is None else {'n_active': self.n_active}

        def run():
            w = kind(board=self.board, seats=self.seats)
            d = agent._move(w, eval, extra, clone=False)   # __call__ clones the replay's outputs
            if not step:
                return d
            new_world, transition = w.step(d.actions, check=False)
            return d, arrdict.arrdict(board=new_world.board, seats=new_world.seats), transition

        with torch.cuda.device(dev):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            # the warm-up draws random numbers; put the generators back so that WHEN a move gets captured (the arena captures
            # one per capacity bucket as its batches shrink) does not show in the stream: a seeded run with captured moves
            # draws what the same run without them draws
            private = getattr(agent.kwargs.get('rng'), 'generator', None)
            gens = [torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]] + ([private] if private is not None else [])
            states = [g.get_state() for g in gens]
            with torch.cuda.stream(side):
                run()                                   # warm-up: builds lookup tables, lets hipBLASLt pick kernels
            torch.cuda.current_stream().wait_stream(side)
            for g, st in zip(gens, states):
                g.set_state(st)
            reserved = torch.cuda.memory_stats(dev)['reserved_bytes.all.allocated']     # cumulative: frees elsewhere cannot hide growth
            self.graph = torch.cuda.CUDAGraph()
            generator = getattr(agent.kwargs.get('rng'), 'generator', None)
            if generator is not None:
                self.graph.register_generator_state(generator)      # a private generator's offsets must be graph-managed too
            with torch.cuda.graph(self.graph):
                self.out = run()
            # what this capture keeps alive -- the tree, its scratch, the block of uniforms: a capture allocates from a pool of
            # its own, so the allocator's reserved bytes grow by exactly that pool (measured, not estimated from shapes)
            self.nbytes = max(torch.cuda.memory_stats(dev)['reserved_bytes.all.allocated'] - reserved, self.board.nbytes + self.seats.nbytes)
        self.kind = kind

    @staticmethod
    def _clone_all(*trees):
        """Fresh copies of every tensor of the given arrdicts with ONE launch (bl_copy_many; a .clone() per tensor is a
        launch each: ten per move)."""
        leaves = [l for t in trees for l in arrdict.leaves(t)]
        it = iter(_native.clone_many(leaves))
        return [t.map(lambda _: next(it)) for t in trees]

    def __call__(self, world):
        n = world.n_envs
        board, seats = (self.board, self.seats) if n == self.capacity else (self.board[:n], self.seats[:n])
        if (world.board.dtype == self.board.dtype and world.seats.dtype == self.seats.dtype and world.board.is_contiguous()
                and world.seats.is_contiguous()):
            _native.copy_many([board, seats], [world.board, world.seats])
        else:
            board.copy_(world.board); seats.copy_(world.seats)
        if self.n_active is not None:
            self.n_active.fill_(n)
        if hasattr(self.network, 'refresh_if_stale'):
            self.network.refresh_if_stale()    # in place, outside the graph: replays read the static f16 weight buffers
        self.graph.replay()
        if n != self.capacity:
            return self._clone_all(self.out.map(lambda t: t[:n]))[0]
        if not self.step:
            return self._clone_all(self.out)[0]
        d, w, t = self._clone_all(*self.out)
        return d, self.kind(board=w.board, seats=w.seats), t


class DummyAgent:
    """Acts straight from the network, no search (mcts/__init__.py:243-257)."""

    def __init__(self, network):
        self.network = network

    def __call__(self, world, eval=False):
        r = self.network(world)
        actions = r.logits.argmax(-1) if eval else torch.distributions.Categorical(logits=r.logits.float()).sample()
        return arrdict.arrdict(
            logits=r.logits, prior=r.logits,
            n_sims=torch.full((world.n_envs,), 0, device=world.device),
            n_leaves=torch.full((world.n_envs,), 1, device=world.device),
            v=r.v, actions=actions).clone()
