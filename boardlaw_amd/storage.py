"""Checkpoints in the reference's wire format, so that runs trained with boardlaw load here and vice versa.

The reference's trainer hands `{'agent': agent, 'opt': opt, 'scaler': scaler}` to pavlov's storage, which turns every
object into its state_dict and `torch.save`s the nested dict (boardlaw/main.py:155-160, pavlov/storage.py:29-56,92-98);
the agent's state_dict is flat: 'network.<parameter>' tensors and 'kwargs.<name>' scalars (boardlaw/mcts/__init__.py:231-241).
The arena rebuilds an agent from such a file plus a pickled network object (boardlaw/arena/common.py:14-34); the pickle needs
the reference's classes, so here the network is rebuilt from the shapes in the state_dict instead (FCModel's shapes
determine it: board cells from the policy head, width from the intake, depth from the ReZero gains)."""
import io

import numpy as np
import torch

from . import heads, networks
from .mcts import MCTSAgent


def state_dicts(**objs):
    """Nested dict with every object replaced by its state_dict (pavlov/storage.py:29-39)."""
    out = {}
    for name, obj in objs.items():
        if isinstance(obj, dict):
            out[name] = state_dicts(**obj)
        else:
            out[name] = obj.state_dict() if hasattr(obj, 'state_dict') else obj
    return out


def expand(flat, depth=np.inf):
    """{'a.b': x} -> {'a': {'b': x}} to `depth` levels (pavlov/storage.py:25-37)."""
    if depth == 0 or not isinstance(flat, dict):
        return flat
    nested = {}
    for key, value in flat.items():
        head, _, tail = key.partition('.')
        nested.setdefault(head, {})[tail] = expand(value, depth - 1)
    return nested


def collapse(nested, depth=np.inf):
    """Inverse of expand for dict-valued entries (pavlov/storage.py:12-23)."""
    if depth == 0:
        return nested
    flat = {}
    for prefix, value in nested.items():
        if isinstance(value, dict):
            for k, v in value.items():
                flat[f'{prefix}.{k}'] = collapse(v, depth - 1)
        else:
            flat[prefix] = value
    return flat


def save(path, **objs):
    """torch.save of the state_dicts, written to a temporary name first (pavlov/storage.py:41-50)."""
    buf = io.BytesIO()
    torch.save(state_dicts(**objs), buf)
    tmp = str(path) + '.tmp'
    with open(tmp, 'wb') as f:
        f.write(buf.getvalue())
    import os
    os.replace(tmp, path)


def load(path, device='cpu'):
    return torch.load(path, map_location=device, weights_only=False)


def network_from_state(agent_state, device='cpu'):
    """FCModel with the shapes found under 'network.*' of an agent state_dict, parameters loaded."""
    sd = {k[len('network.'):]: v for k, v in agent_state.items() if k.startswith('network.')}
    width, n_inputs = sd['body.0.weight'].shape
    cells = sd['policy.core.weight'].shape[0]
    boardsize = int(round(cells ** .5))
    if boardsize * boardsize != cells or n_inputs != 2 * cells:
        raise ValueError(f'Not a Hex FCModel: {cells} actions, {n_inputs} inputs')
    depth = sum(1 for k in sd if k.startswith('body.') and k.endswith('.α'))
    net = networks.FCModel(heads.Tensor((boardsize, boardsize, 2)), heads.Masked(cells), width=width, depth=depth)
    net.load_state_dict(sd, strict=True)
    return net.to(device)


def agent_from_checkpoint(checkpoint, device='cpu', inference=None, **agent_kwargs):
    """arena.common.agent (arena/common.py:14-34) from a loaded snapshot/latest file: {'agent': state_dict, ...} or the
    agent state_dict itself.  inference: None | 'torch' | 'fused' wraps the network in a networks.Inference plan."""
    state = checkpoint['agent'] if 'agent' in checkpoint else checkpoint
    net = network_from_state(state, device)
    actor = net if inference is None else networks.Inference(net, fused=(inference == 'fused'))
    agent = MCTSAgent(actor, **agent_kwargs)
    agent.kwargs.update({k[len('kwargs.'):]: v for k, v in state.items() if k.startswith('kwargs.')})
    return agent
