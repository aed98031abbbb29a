"""Training-target helpers of the actor/learner loop (boardlaw/learning.py:6-76): `mix`, `half`, `reward_to_go`,
`present_value`, `rel_entropy`, `gather`.  Pure torch; device-agnostic."""
import numpy as np
import torch

from . import arrdict


def mix(worlds, T=2500, check=False):
    """Plays T uniformly random legal moves so the batch starts de-correlated (learning.py:6-10)."""
    for _ in range(T):
        actions = torch.distributions.Categorical(probs=worlds.valid.float(), validate_args=False).sample()
        worlds, _ = worlds.step(actions, check=check) if _accepts_check(worlds) else worlds.step(actions)
    return worlds


def _accepts_check(worlds):
    import inspect
    return 'check' in inspect.signature(worlds.step).parameters


@arrdict.mapping
def half(x):
    return x.half() if isinstance(x, torch.Tensor) and x.dtype == torch.float else x


def rel_entropy(logits):
    valid = logits > -np.inf
    zeros = torch.zeros_like(logits)
    l = logits.where(valid, zeros)
    probs = l.exp().where(valid, zeros)
    return -(l * probs).sum(-1).mean(), torch.log(valid.sum(-1).float()).mean()


def gather(arr, indices):
    if isinstance(arr, dict):
        return type(arr)((k, gather(arr[k], indices[k])) for k in arr)
    return torch.gather(arr, -1, indices.long().unsqueeze(-1)).squeeze(-1)


def present_value(deltas, fallback, terminal, alpha):
    """Backward recursion over time: result[t] = fallback[t] where terminal[t], else deltas[t] + alpha*result[t+1]
    (learning.py:45-58).  Shapes: deltas (T-1, ...), fallback/terminal (T, ...)."""
    assert deltas.shape == fallback[:-1].shape == terminal[:-1].shape
    result = torch.full_like(fallback, np.nan)
    result[-1] = fallback[-1]
    for t in range(deltas.size(0) - 1, -1, -1):
        result[t] = torch.where(terminal[t], fallback[t], deltas[t] + alpha * result[t + 1])
    return result


def reward_to_go(reward, value, terminal, gamma=1.):
    """Monte-Carlo value targets over a buffer (learning.py:60-66): the final row bootstraps from the value estimates,
    terminal steps fall back to their reward.  NOTE: like the reference this writes into `value` in place."""
    fallback = value
    fallback[terminal] = reward[terminal]
    return present_value(reward[:-1], fallback, terminal, gamma).detach()
