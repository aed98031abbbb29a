"""FCModel: the leaf-evaluation network (boardlaw/networks.py:10-40) -- Linear intake, `depth` ReZero
fully-connected residual blocks, masked log-softmax policy head and tanh value head.  It stays a torch.nn module
(state_dict keys identical to the reference's: body.N.weight/bias/α, policy.core.*, value.core.*)."""
import torch
from torch import nn
from torch.nn import functional as F

from . import arrdict, heads


class ReZeroResidual(nn.Linear):
    """x + α·Linear(relu(x)), α initialised to 0 (networks.py:10-18)."""

    def __init__(self, width):
        super().__init__(width, width)
        nn.init.orthogonal_(self.weight, gain=2**.5)
        self.register_parameter('α', nn.Parameter(torch.zeros(())))

    def forward(self, x, *args, **kwargs):
        return x + getattr(self, 'α') * super().forward(F.relu(x))


class FCModel(nn.Module):

    def __init__(self, obs_space, action_space, width=256, depth=64):
        super().__init__()
        self.obs_space, self.action_space = obs_space, action_space
        self.policy = heads.output(action_space, width)
        self.sampler = self.policy.sample
        self.body = nn.Sequential(heads.intake(obs_space, width), *[ReZeroResidual(width) for _ in range(depth)])
        self.value = heads.ValueOutput(width)

    def forward(self, worlds):
        neck = self.body(worlds.obs)
        return arrdict.arrdict(
            logits=self.policy(neck, worlds.valid),
            v=self.value(neck, worlds.valid, worlds.seats))

    def raw(self, worlds):
        """Pre-head outputs: (policy Linear output (B,A), value Linear output (B,)).  The heads themselves -- masked
        log-softmax, tanh + seat scatter -- are then applied by bl_sim_finish inside the search step."""
        neck = self.body(worlds.obs)
        return self.policy.raw(neck), self.value.core(neck).squeeze(-1)


def pack_fragment_major(m):
    """(N, K) f16 -> the layout bl_mlp_forward_f16 streams: [N/32][K/64][s=4][lane=64][8] with
    lane = 32*(k-half) + (n % 32); N % 32 == 0, K % 64 == 0."""
    N, K = m.shape
    return m.view(N // 32, 32, K // 64, 2, 4, 8).permute(0, 2, 4, 3, 1, 5).contiguous()


def pack_fragment_major_f32(m):
    """(N, K) f32 -> the layout bl_root_mlp_f32 streams: [N/16][K/16][lane=64][4] with lane = 16*(k-quarter) + (n % 16);
    N % 16 == 0, K % 16 == 0."""
    N, K = m.shape
    return m.view(N // 16, 16, K // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()


class Inference:
    """fp16 inference plan for an FCModel inside the search: the same arithmetic as the module under fp16 autocast
    (what the reference runs in MCTS.simulate), issued as 6 GEMMs + 5 elementwise launches instead of ~45:
      * weights are cast to f16 once per refresh() into static buffers (autocast re-casts all 13 tensors every call);
      * the ReZero tail x + alpha*y and the next block's relu are one fused kernel (bl_rezero_relu_f16) with torch's
        rounding points;
      * the heads are left to bl_sim_finish.
    Calling the object runs the wrapped module unchanged (fp32 root evaluation, training).  With fused=False results of
    raw() are bit-identical to FCModel.raw under autocast (tests/test_gpu_parity.py::test_inference_plan_matches_autocast)."""

    wants_half_obs = True
    # The root evaluation's plan must not depend on the batch size: a search's result for an env is a function of that env, the
    # random stream and the batch's q-range only -- so a captured move padded to a capacity bucket (MCTSAgent(pad=True), the
    # arena's masked calls) gives the bits of the eager call on exactly those envs.  bl_root_mlp_f32 computes every row in a
    # fixed order whatever M is; the library GEMMs it replaces pick their kernels (and summation orders) by M.  Below ~2048 rows
    # of 512x4 the kernel is no faster than the GEMMs (its 16-row workgroups do not fill the chip) and for 1024x8 on 1024 rows
    # it is ~0.3 ms slower per move (of 46 ms): the price of the invariance.  Raise to trade it back.
    ROOT_FUSED_MIN_ROWS = 0
    # bl_mlp_forward_f16's time is one workgroup's chain: every 32-row workgroup streams ALL the weights through its CU's L1
    # (64 B/clk).  Up to FUSED_ALWAYS_BYTES of weights (512x4: 2.4 MB, 28 us) that beats a launch per Linear at any batch size;
    # beyond it (1024x8: 17.9 MB, 170 us) only once the batch gives FUSED_MIN_TILES workgroups.  Below that every Linear is
    # its own launch, split over all CUs (bl_mlp_layers_f16; us per forward at 13x13, one kernel / launch per Linear / library
    # GEMMs: 1024x8 on 1024 rows 170 / 95 / 133, on 2048 rows 166 / 131 / 159, on 4096 rows 165 / 215 / 202; 768x6 on
    # 2048 rows 92 / 87 / 106, on 4096 rows 92 / 127 / 138).
    FUSED_ALWAYS_BYTES = 6 << 20
    FUSED_MIN_TILES = 96
    LAYERS_PLAN = True      # below that: bl_mlp_layers_f16 (False: torch's GEMMs + bl_rezero_relu_f16, bit-identical to autocast)
    # ... as ONE launch where the grid fits the chip (bl_mlp_layers_persist_f16): bit-identical, measured in round 4 and SLOWER at every
    # shape (1024x8 on 1024 rows: 91.9 against 80.8 us; profiles/r04_layers_persist.txt) -- a hand-off between workgroups through
    # memory costs what a launch boundary costs.  Off.
    PERSIST_PLAN = False

    def __init__(self, model, fused=False, precision='fp16'):
        """fused=True additionally runs all Linears as ONE MFMA kernel (bl_mlp_forward_f16) when the width is a multiple
        of 128: same rounding points, but the GEMMs' summation order is the kernel's own, so outputs equal the autocast
        module's to f16 rounding rather than bit for bit.
        precision='fp32': the EXACT mode -- leaves are evaluated like the root, in fp32 with only the stores rounded to f16
        (`decisions.logits.half()`, `decisions.v.half()`, boardlaw/mcts/__init__.py:131-136).  That is what the reference's
        recorded CPU runs do (`torch.cuda.amp.autocast` is a no-op there), so a seeded search stores the reference's own f16
        logits wherever the two f32 GEMM summation orders round to the same binary16 (tests/test_reference_fixtures.py:
        test_fp32_leaves_replay_the_reference_search).  Linears: bl_root_mlp_f32 (fused=True) or the library's f32 GEMMs;
        heads, store, backup and q-range: bl_sim_finish_f32.  Roughly 1.5x the fp16 plan's time per simulation; opt-in."""
        if precision not in ('fp16', 'fp32'):
            raise ValueError(f"precision must be 'fp16' or 'fp32', got {precision!r}")
        self.model = model
        self.fused = fused
        self.precision = precision
        self.leaf_fp32 = precision == 'fp32'
        self.wants_half_obs = not self.leaf_fp32       # fp32 leaves read the reference's f32 observation layout
        self._static = None
        self._packed = None
        self._persist = {}          # bl_mlp_layers_persist_f16's error word per device
        self._root_heads = None
        self._root_packed = None
        self._stamped = None

    def __call__(self, worlds):
        return self.model(worlds)

    def parameters(self):
        return self.model.parameters()

    def state_dict(self):
        return self.model.state_dict()

    def load_state_dict(self, sd):
        return self.model.load_state_dict(sd)

    def _sources(self):
        m = self.model
        blocks = list(m.body)
        srcs = [blocks[0].weight, blocks[0].bias]
        for blk in blocks[1:]:
            srcs += [blk.weight, blk.bias]
        srcs += [m.policy.core.weight, m.policy.core.bias, m.value.core.weight, m.value.core.bias]
        return srcs, [getattr(blk, 'α') for blk in blocks[1:]]

    def _fusable(self):
        m = self.model
        blocks = list(m.body)
        W, K0 = blocks[0].weight.shape
        A = m.policy.core.weight.shape[0]
        NHpad = -(-(A + 1) // 32) * 32
        buf, staging = 32 * (W + 8) * 2, (NHpad // 32) * 4096 + 32 * NHpad * 2     # bl_mlp_forward_f16's LDS budget
        return (W in (128, 256, 512, 768, 1024) and -(-K0 // 64) * 64 <= W and buf + max(buf, staging) <= 160 * 1024
                and type(m.policy).__name__ in ('MaskedOutput', 'DiscreteOutput') and blocks[0].weight.is_cuda)

    def _root_fusable(self):
        """bl_root_mlp_f32's limits: fp32 parameters on the GPU, width a multiple of 128 up to 1024, the flattened
        observation no wider than the body."""
        m = self.model
        blocks = list(m.body)
        W, K0 = blocks[0].weight.shape
        return (W % 128 == 0 and 128 <= W <= 1024 and -(-K0 // 64) * 64 <= W and blocks[0].weight.dtype == torch.float
                and type(m.policy).__name__ in ('MaskedOutput', 'DiscreteOutput') and m.value.core.weight.shape[0] == 1
                and blocks[0].weight.is_cuda)

    def _stamp(self):
        srcs, alphas = self._sources()
        return tuple((p.data_ptr(), p._version) for p in list(srcs) + list(alphas))

    def refresh_if_stale(self):
        """refresh() only if a parameter was modified in place (optimiser step, load_state_dict) or replaced since the
        last refresh.  Host-side check of tensor version counters: no device work when nothing changed.
        With PERSIST_PLAN (off by default) this is also where the one-launch kernel's error word is read -- once per move, before
        the next one is issued: a forward whose bounded waits ran out has written invalid logits, and the search must not go on."""
        if self.PERSIST_PLAN and self._persist and not torch.cuda.is_current_stream_capturing() and self.persist_error():
            raise RuntimeError('bl_mlp_layers_persist_f16: a bounded wait ran out (peers of a row tile were kept off the chip, e.g. by '
                               'another stream); the forwards since the last check are invalid -- use the launch-per-Linear plan')
        if self._static is None or self._stamp() != self._stamped:
            self.refresh()

    def refresh(self):
        """Re-cast the module's current parameters into the static f16 buffers (in place: safe to capture/replay)."""
        srcs, alphas = self._sources()
        with torch.no_grad():
            if self._static is None or self._static[0][0].device != srcs[0].device:
                self._static = ([torch.empty_like(p, dtype=torch.half) for p in srcs],
                                [torch.empty((), dtype=torch.float, device=a.device) for a in alphas])
                self._packed = None
            for dst, src in zip(self._static[0], srcs):
                dst.copy_(src)
            for dst, src in zip(self._static[1], alphas):
                dst.copy_(src)
            if srcs[0].is_cuda and not torch.cuda.is_current_stream_capturing():
                self._persist_error(srcs[0].device)          # exists before any capture
            if self.fused and self._fusable():
                # layout of bl_mlp_forward_f16: zero-padded intake, stacked blocks, policy+value head stacked
                w = self._static[0]
                W, K0 = w[0].shape
                D, A = len(alphas), w[-4].shape[0]
                K0pad, NHpad = -(-K0 // 64) * 64, -(-(A + 1) // 32) * 32
                if self._packed is None:
                    dev = w[0].device
                    z = lambda *s, dtype=torch.half: torch.zeros(s, dtype=dtype, device=dev)
                    self._packed = dict(w0=z(W, K0pad), wb=z(max(D, 1), W, W), bb=z(max(D, 1), W), al=z(max(D, 1), dtype=torch.float),
                                        wh=z(NHpad, W), bh=z(NHpad), dims=(W, K0, K0pad, D, A + 1, NHpad))
                pk = self._packed
                stage0 = torch.zeros((W, K0pad), dtype=torch.half, device=w[0].device); stage0[:, :K0] = w[0]
                pk['w0'].view(-1).copy_(pack_fragment_major(stage0).view(-1))
                for d in range(D):
                    pk['wb'][d].view(-1).copy_(pack_fragment_major(w[2 + 2 * d]).view(-1))
                    pk['bb'][d].copy_(w[3 + 2 * d]); pk['al'][d].copy_(self._static[1][d])
                stageh = torch.zeros((NHpad, W), dtype=torch.half, device=w[0].device)
                stageh[:A] = w[-4]; stageh[A] = w[-2][0]
                pk['wh'].view(-1).copy_(pack_fragment_major(stageh).view(-1))
                pk['bh'][:A].copy_(w[-3]); pk['bh'][A].copy_(w[-1][0])
            m = self.model
            if type(m.policy).__name__ in ('MaskedOutput', 'DiscreteOutput') and m.value.core.weight.shape[0] == 1:
                if self._root_heads is None or self._root_heads[0].device != m.policy.core.weight.device:
                    A, W = m.policy.core.weight.shape
                    dev = m.policy.core.weight.device
                    self._root_heads = (torch.empty((A + 1, W), dtype=torch.float, device=dev), torch.empty((A + 1,), dtype=torch.float, device=dev))
                wcat, bcat = self._root_heads
                A = wcat.shape[0] - 1
                wcat[:A].copy_(m.policy.core.weight); wcat[A:].copy_(m.value.core.weight)
                bcat[:A].copy_(m.policy.core.bias); bcat[A:].copy_(m.value.core.bias)
            if self.fused and self._root_fusable():
                # layout of bl_root_mlp_f32: fp32, zero-padded intake, stacked blocks, policy+value head stacked
                srcs32 = srcs
                W, K0 = srcs32[0].shape
                D, A = len(alphas), srcs32[-4].shape[0]
                K0pad, NHpad = -(-K0 // 64) * 64, -(-(A + 1) // 16) * 16
                dev = srcs32[0].device
                if self._root_packed is None or self._root_packed['w0'].device != dev:
                    z = lambda *s: torch.zeros(s, dtype=torch.float, device=dev)
                    self._root_packed = dict(w0=z(W, K0pad), b0=z(W), wb=z(max(D, 1), W, W), bb=z(max(D, 1), W), al=z(max(D, 1)),
                                             wh=z(NHpad, W), bh=z(NHpad), dims=(W, K0, K0pad, D, A + 1, NHpad))
                rp = self._root_packed
                stage0 = torch.zeros((W, K0pad), dtype=torch.float, device=dev); stage0[:, :K0] = srcs32[0]
                rp['w0'].view(-1).copy_(pack_fragment_major_f32(stage0).view(-1)); rp['b0'].copy_(srcs32[1])
                for d in range(D):
                    rp['wb'][d].view(-1).copy_(pack_fragment_major_f32(srcs32[2 + 2 * d].detach().float()).view(-1))
                    rp['bb'][d].copy_(srcs32[3 + 2 * d]); rp['al'][d].copy_(alphas[d])
                stageh = torch.zeros((NHpad, W), dtype=torch.float, device=dev)
                stageh[:A] = srcs32[-4]; stageh[A] = srcs32[-2][0]
                rp['wh'].view(-1).copy_(pack_fragment_major_f32(stageh).view(-1))
                rp['bh'][:A].copy_(srcs32[-3]); rp['bh'][A].copy_(srcs32[-1][0])
            else:
                self._root_packed = None
        self._stamped = self._stamp()

    def root_raw(self, worlds):
        """fp32 pre-head outputs for the root evaluation (MCTS.initialize runs the network outside autocast,
        mcts/__init__.py:72-76): the module's own fp32 parameters and torch's GEMMs, with each block's alpha*y, x + .
        and the next relu issued as one kernel (bl_rezero_relu_f32; same two roundings as torch's mul and add).
        The body and the policy head are bit-identical to FCModel.raw in fp32; the value head, computed by the stacked heads'
        GEMM instead of a one-column GEMM of its own, within 1e-6 (tests/test_gpu_parity.py::test_root_plan_matches_module)."""
        from . import _native
        self.refresh_if_stale()
        m = self.model
        blocks = list(m.body)
        obs = worlds.obs
        x0 = obs.reshape(obs.shape[0], -1).float().contiguous()
        L, st = _native.lib(), _native.stream(x0.device)
        # bl_root_mlp_f32 takes 16 rows per workgroup through all layers (102 vs 135 us for the library GEMMs at 4096 rows of
        # 512x4); used at every batch size, see ROOT_FUSED_MIN_ROWS
        if self.fused and self._root_packed is not None and x0.is_cuda and x0.shape[0] >= self.ROOT_FUSED_MIN_ROWS:
            rp = self._root_packed
            W, K0, K0pad, D, NH, NHpad = rp['dims']
            M = x0.shape[0]
            policy = torch.empty((M, NH - 1), dtype=torch.float, device=x0.device)
            value = torch.empty((M,), dtype=torch.float, device=x0.device)
            with torch.cuda.device(x0.device):
                _native.check(L.bl_root_mlp_f32(x0.data_ptr(), M, K0, rp['w0'].data_ptr(), rp['b0'].data_ptr(), rp['wb'].data_ptr(),
                                                rp['bb'].data_ptr(), rp['al'].data_ptr(), rp['wh'].data_ptr(), rp['bh'].data_ptr(),
                                                W, D, K0pad, NH, NHpad, policy.data_ptr(), value.data_ptr(), st))
            return policy, value
        with torch.no_grad():
            x = F.linear(x0, blocks[0].weight, blocks[0].bias)
            r = F.relu(x)
            for blk in blocks[1:]:
                y = F.linear(r, blk.weight, blk.bias)
                x_new, r = torch.empty_like(x), torch.empty_like(x)
                _native.check(L.bl_rezero_relu_f32(x.data_ptr(), y.data_ptr(), getattr(blk, 'α').data_ptr(), x_new.data_ptr(),
                                                   r.data_ptr(), x.numel(), st))
                x = x_new
            # both heads' Linears as one GEMM over the stacked (A+1, W) weights (refresh() keeps the stack current): the
            # value head alone, one output column, is a 17 us launch of its own plus torch's bias broadcast
            if self._root_heads is not None:
                out = F.linear(x, *self._root_heads)
                return out[:, :-1], out[:, -1]
            return F.linear(x, m.policy.core.weight, m.policy.core.bias), F.linear(x, m.value.core.weight, m.value.core.bias).squeeze(-1)

    def _persist_error(self, device):
        """The word bl_mlp_layers_persist_f16 raises when one of its bounded waits ran out (made at refresh(), outside any capture)."""
        key = device.index if device.index is not None else torch.cuda.current_device()
        if key not in self._persist:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('Inference: call refresh() (or one eager forward) before capturing')
            self._persist[key] = torch.zeros((1,), dtype=torch.int32, device=device)
        return self._persist[key]

    def persist_error(self):
        """True if a bounded wait of bl_mlp_layers_persist_f16 ever ran out (its workgroups' peers were kept off the chip): the forwards
        of that call were invalid.  Synchronises."""
        return any(int(c.item()) != 0 for c in self._persist.values())

    def prefers_fused(self, rows):
        """Whether the one-kernel plan is the faster one for a batch of `rows` (see FUSED_ALWAYS_BYTES)."""
        if not (self.fused and self._packed is not None):
            return False
        W, K0, K0pad, D, NH, NHpad = self._packed['dims']
        streamed = 2 * (W * K0pad + D * W * W + NHpad * W)
        return streamed <= self.FUSED_ALWAYS_BYTES or -(-rows // 32) >= self.FUSED_MIN_TILES

    def fused_params(self, rows=None):
        """Pointers and dims of the packed f16 weights for bl_sim_infer_finish, or None when the plan is not the fused
        kernel's (then the caller uses raw() + bl_sim_finish)."""
        if self._static is None:
            self.refresh()
        if not (self.fused and self._packed is not None) or (rows is not None and not self.prefers_fused(rows)):
            return None
        pk = self._packed
        W, K0, K0pad, D, NH, NHpad = pk['dims']
        return dict(w0=pk['w0'].data_ptr(), b0=self._static[0][1].data_ptr(), wb=pk['wb'].data_ptr(), bb=pk['bb'].data_ptr(),
                    al=pk['al'].data_ptr(), wh=pk['wh'].data_ptr(), bh=pk['bh'].data_ptr(), W=W, D=D, K0=K0, K0pad=K0pad,
                    NH=NH, NHpad=NHpad)

    def raw(self, worlds):
        from . import _native
        if self._static is None:
            self.refresh()
        w, alphas = self._static
        L = _native.lib()
        obs = worlds.obs
        x0 = obs.reshape(obs.shape[0], -1).half().contiguous()
        st = _native.stream(x0.device)
        if self.fused and self._packed is not None and self.prefers_fused(x0.shape[0]):
            pk = self._packed
            W, K0, K0pad, D, NH, NHpad = pk['dims']
            M = x0.shape[0]
            policy = torch.empty((M, NH - 1), dtype=torch.half, device=x0.device)
            value = torch.empty((M,), dtype=torch.half, device=x0.device)
            with torch.cuda.device(x0.device):
                _native.check(L.bl_mlp_forward_f16(x0.data_ptr(), M, K0, pk['w0'].data_ptr(), w[1].data_ptr(), pk['wb'].data_ptr(),
                                                   pk['bb'].data_ptr(), pk['al'].data_ptr(), pk['wh'].data_ptr(), pk['bh'].data_ptr(),
                                                   W, D, K0pad, NH, NHpad, policy.data_ptr(), value.data_ptr(), st))
            return policy, value
        if self.fused and self._packed is not None and self.LAYERS_PLAN and x0.is_cuda and self._packed['dims'][1] % 2 == 0:
            # wide network, small batch: a launch per Linear, each split over all CUs (bl_mlp_layers_f16), instead of the
            # library GEMMs + elementwise launches below -- 1024x8 on 1024 rows of 13x13: see DESIGN.md 4.4
            pk = self._packed
            W, K0, K0pad, D, NH, NHpad = pk['dims']
            M = x0.shape[0]
            policy = torch.empty((M, NH - 1), dtype=torch.half, device=x0.device)
            value = torch.empty((M,), dtype=torch.half, device=x0.device)
            scratch = torch.empty((2, M, W), dtype=torch.half, device=x0.device)
            with torch.cuda.device(x0.device):
                rc = _native.BL_ETOOBIG
                if self.PERSIST_PLAN:
                    # all Linears in one launch, workgroups synchronising per row tile (bl_mlp_layers_persist_f16, or with
                    # PERSIST_PLAN = 'xcd' bl_mlp_layers_xcd_f16: the hand-off inside one XCD's L2): grids up to 256
                    # workgroups.  The counter block is this call's own (like `scratch`: from the caching allocator, or from a capture's
                    # pool -- two actors sharing this plan on two streams never share one) and is zeroed by the call
                    local = self.PERSIST_PLAN == 'xcd'
                    counters = torch.empty((-(-M // 32) * (D + 2) * (8 if local else 1) + 9,), dtype=torch.int32, device=x0.device)
                    rc = (L.bl_mlp_layers_xcd_f16 if local else L.bl_mlp_layers_persist_f16)(x0.data_ptr(), M, K0, pk['w0'].data_ptr(), w[1].data_ptr(), pk['wb'].data_ptr(),
                                                     pk['bb'].data_ptr(), pk['al'].data_ptr(), pk['wh'].data_ptr(), pk['bh'].data_ptr(),
                                                     W, D, K0pad, NH, NHpad, scratch.data_ptr(), counters.data_ptr(), 1,
                                                     self._persist_error(x0.device).data_ptr(), policy.data_ptr(), value.data_ptr(), st)
                if rc == _native.BL_ETOOBIG:
                    rc = L.bl_mlp_layers_f16(x0.data_ptr(), M, K0, pk['w0'].data_ptr(), w[1].data_ptr(), pk['wb'].data_ptr(),
                                             pk['bb'].data_ptr(), pk['al'].data_ptr(), pk['wh'].data_ptr(), pk['bh'].data_ptr(),
                                             W, D, K0pad, NH, NHpad, scratch.data_ptr(), policy.data_ptr(), value.data_ptr(), st)
                _native.check(rc)
            return policy, value
        x = F.linear(x0, w[0], w[1])
        r = F.relu(x)
        for i, alpha in enumerate(alphas):
            y = F.linear(r, w[2 + 2 * i], w[3 + 2 * i])
            x_new, r = torch.empty_like(x), torch.empty_like(x)
            _native.check(L.bl_rezero_relu_f16(x.data_ptr(), y.data_ptr(), alpha.data_ptr(), x_new.data_ptr(), r.data_ptr(),
                                               x.numel(), st))
            x = x_new
        return F.linear(x, w[-4], w[-3]), F.linear(x, w[-2], w[-1]).squeeze(-1)
