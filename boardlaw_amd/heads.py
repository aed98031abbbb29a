"""Observation intakes and policy/value heads for the leaf-evaluation network (PyTorch-ROCm; the GEMMs are the only
MFMA work on the path).  Mirrors the parts of boardlaw/heads.py the Hex path reaches -- space tuples
(heads.py:7-11), the flattening Linear intake (heads.py:41-52), the masked log-softmax policy head (heads.py:92-104),
the seat-scattered tanh value head (heads.py:128-142) -- with the same parameter names, so reference checkpoints load."""
from collections import namedtuple

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

Empty = namedtuple('Empty', ())
Discrete = namedtuple('Discrete', ('dim',))
Masked = namedtuple('Masked', ('dim',))
Vector = namedtuple('Vector', ('dim',))
Tensor = namedtuple('Tensor', ('dim',))


class _FlatLinear(nn.Linear):
    """Linear over the trailing `event_ndim` dims, for inputs shaped (B, *event) or (T, B, *event)."""

    def __init__(self, event_shape, width):
        self._event_ndim = len(event_shape)
        super().__init__(int(np.prod(event_shape)), width)

    def forward(self, obs, *args, **kwargs):
        lead = obs.shape[:obs.ndim - self._event_ndim]
        return super().forward(obs.reshape(*lead, -1))


class TensorIntake(_FlatLinear):
    def __init__(self, space, width):
        super().__init__(tuple(space.dim), width)


class VectorIntake(_FlatLinear):
    def __init__(self, space, width):
        super().__init__(tuple(space.dim), width)


def intake(space, width):
    kinds = {'Tensor': TensorIntake, 'Vector': VectorIntake}
    name = type(space).__name__
    if name not in kinds:
        raise ValueError(f"Can't handle {space}")
    return kinds[name](space, width)


class _PolicyHead(nn.Module):
    def __init__(self, space, width):
        super().__init__()
        self.shape = (space.dim,) if np.isscalar(space.dim) else tuple(space.dim)
        self.core = nn.Linear(width, int(np.prod(self.shape)))

    def raw(self, x):
        return self.core(x).reshape(*x.shape[:-1], *self.shape)

    def sample(self, logits, test=False):
        return logits.argmax(-1) if test else torch.distributions.Categorical(logits=logits).sample()


class DiscreteOutput(_PolicyHead):
    def forward(self, x, *args, **kwargs):
        return F.log_softmax(self.raw(x), -1)


class MaskedOutput(_PolicyHead):
    """Invalid actions get logit -inf, hence probability exactly 0 in the search (the kernels see no mask)."""

    def forward(self, x, valid, *args, **kwargs):
        y = self.raw(x)
        return F.log_softmax(y.masked_fill(~valid, -np.inf), -1)


def output(space, width):
    kinds = {'Discrete': DiscreteOutput, 'Masked': MaskedOutput}
    name = type(space).__name__
    if name not in kinds:
        raise ValueError(f"Can't handle {space}")
    return kinds[name](space, width)


def scatter_values(v, seats):
    """Puts +v at the mover's seat and -v at the other one: (B,) -> (B,2).  heads.py:122-126"""
    idx = torch.stack([seats, 1 - seats], -1).long()
    return torch.full_like(torch.stack([v, -v], -1), np.nan).scatter(-1, idx, torch.stack([v, -v], -1))


class ValueOutput(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.core = nn.Linear(width, 1)

    def forward(self, x, valid, seats, *args, **kwargs):
        return scatter_values(torch.tanh(self.core(x).squeeze(-1)), seats)
