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


class Inference:
    """fp16 inference plan for an FCModel inside the search: the same arithmetic as the module under fp16 autocast
    (what the reference runs in MCTS.simulate), issued as 6 GEMMs + 5 elementwise launches instead of ~45:
      * weights are cast to f16 once per refresh() into static buffers (autocast re-casts all 13 tensors every call);
      * the ReZero tail x + alpha*y and the next block's relu are one fused kernel (bl_rezero_relu_f16) with torch's
        rounding points;
      * the heads are left to bl_sim_finish.
    Calling the object runs the wrapped module unchanged (fp32 root evaluation, training).  Results of raw() are
    bit-identical to FCModel.raw under autocast (tests/test_gpu_parity.py::test_inference_plan_matches_autocast)."""

    wants_half_obs = True

    def __init__(self, model):
        self.model = model
        self._static = None

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

    def refresh(self):
        """Re-cast the module's current parameters into the static f16 buffers (in place: safe to capture/replay)."""
        srcs, alphas = self._sources()
        with torch.no_grad():
            if self._static is None or self._static[0][0].device != srcs[0].device:
                self._static = ([torch.empty_like(p, dtype=torch.half) for p in srcs],
                                [torch.empty((), dtype=torch.float, device=a.device) for a in alphas])
            for dst, src in zip(self._static[0], srcs):
                dst.copy_(src)
            for dst, src in zip(self._static[1], alphas):
                dst.copy_(src)

    def raw(self, worlds):
        from . import _native
        if self._static is None:
            self.refresh()
        w, alphas = self._static
        L = _native.lib()
        obs = worlds.obs
        x = F.linear(obs.reshape(obs.shape[0], -1).half(), w[0], w[1])
        r = F.relu(x)
        st = _native.stream(x.device)
        for i, alpha in enumerate(alphas):
            y = F.linear(r, w[2 + 2 * i], w[3 + 2 * i])
            x_new, r = torch.empty_like(x), torch.empty_like(x)
            _native.check(L.bl_rezero_relu_f16(x.data_ptr(), y.data_ptr(), alpha.data_ptr(), x_new.data_ptr(), r.data_ptr(),
                                               x.numel(), st))
            x = x_new
        return F.linear(x, w[-4], w[-3]), F.linear(x, w[-2], w[-1]).squeeze(-1)
