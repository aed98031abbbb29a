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
