"""Native search kernels behind the reference's `boardlaw.mcts.cuda` surface (boardlaw/mcts/cuda.py:13-42):
`mcts(...)` -> MCTS struct, `Backup(...)`, `descend(m)` -> Descent, `root(m)`, `backup(bk, leaves)`.

The structs do what the reference's TensorProxy constructors do (boardlaw/cpp/common.h:27-44): hold references,
demand contiguity (RuntimeError), exact dtypes (TypeError "expected Half got Float") and ndim.  The kernels run in
libboardlaw_amd.so on the current HIP stream, unsynchronised, like the reference's launches."""
import ctypes

import torch

from .. import _native

_NAMES = {torch.half: 'Half', torch.float: 'Float', torch.short: 'Short', torch.int: 'Int', torch.long: 'Long',
          torch.bool: 'Bool', torch.uint8: 'Byte', torch.double: 'Double'}


def _proxy(t, dtype, ndim, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a tensor')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')
    if t.dtype != dtype:
        raise TypeError(f'expected {_NAMES.get(dtype, dtype)} got {_NAMES.get(t.dtype, t.dtype)}')
    if t.ndim != ndim:
        raise RuntimeError(f'{name}: expected {ndim} dims got {t.ndim}')
    return t


def assert_shape(x, s):
    assert (x.ndim == len(s)) and x.shape == s, f'Expected {s}, got {x.shape}'


class MCTS:
    """mcts/cpp/common.h:25-55: logits f16 (B,T,A), w f16 (B,T,S), n i16 (B,T), c_puct f16 (B), seats i16 (B,T),
    terminal bool (B,T), children i16 (B,T,A)."""

    def __init__(self, logits, w, n, c_puct, seats, terminal, children):
        self.logits = _proxy(logits, torch.half, 3, 'logits')
        self.w = _proxy(w, torch.half, 3, 'w')
        self.n = _proxy(n, torch.short, 2, 'n')
        self.c_puct = _proxy(c_puct, torch.half, 1, 'c_puct')
        self.seats = _proxy(seats, torch.short, 2, 'seats')
        self.terminal = _proxy(terminal, torch.bool, 2, 'terminal')
        self.children = _proxy(children, torch.short, 3, 'children')

    def _tensors(self):
        return (self.logits, self.w, self.n, self.c_puct, self.seats, self.terminal, self.children)


class Backup:
    """mcts/cpp/common.h:103-110"""

    def __init__(self, v, w, n, rewards, parents, terminal):
        self.v = _proxy(v, torch.half, 3, 'v')
        self.w = _proxy(w, torch.half, 3, 'w')
        self.n = _proxy(n, torch.short, 2, 'n')
        self.rewards = _proxy(rewards, torch.half, 3, 'rewards')
        self.parents = _proxy(parents, torch.short, 2, 'parents')
        self.terminal = _proxy(terminal, torch.bool, 2, 'terminal')


class Descent:
    def __init__(self, parents, actions):
        self.parents, self.actions = parents, actions


def mcts(logits, w, n, c_puct, seats, terminal, children):
    """boardlaw/mcts/cuda.py:13-26, including its shape asserts, the c_puct > 0 check and the seats.short() cast."""
    B, T, A = logits.shape
    S = w.shape[-1]
    assert_shape(w, (B, T, S)); assert_shape(n, (B, T)); assert_shape(c_puct, (B,)); assert_shape(seats, (B, T))
    assert_shape(terminal, (B, T)); assert_shape(children, (B, T, A))
    assert (c_puct > 0.).all(), 'Zero c_puct not supported; will lead to an infinite loop in the kernel'
    assert len({t.device for t in (logits, w, n, c_puct, seats, terminal, children)}) == 1, 'Inputs span multiple devices'
    return MCTS(logits, w, n, c_puct, seats.short(), terminal, children)


def _qrange(m, dev):
    B, T, S = m.w.shape
    state = torch.empty(_native.QRANGE_WORDS, dtype=torch.int32, device=dev)
    _native.check(_native.lib().bl_mcts_qrange(m.w.data_ptr(), m.n.data_ptr(), B, T, S, state.data_ptr(), _native.stream(dev)))
    return state


def descend(m, rands=None):
    """mctscuda.descend (cuda.cu:184-203).  `rands`, if not given, is drawn exactly as the reference draws it:
    rand_like(logits[:, :, 0]) from torch's default generator for the device."""
    dev = _native.require_device(*m._tensors())
    B, T, A = m.logits.shape
    S = m.w.shape[-1]
    with torch.cuda.device(dev):
        state = _qrange(m, dev)
        if rands is None:
            rands = torch.rand_like(m.logits[:, :, 0])
        rands = _proxy(rands.contiguous(), torch.half, 2, 'rands')
        parents = m.seats.new_empty((B,)); actions = m.seats.new_empty((B,))
        _native.check(_native.lib().bl_mcts_descend_tuned(
            ctypes.byref(_native.tune()), m.logits.data_ptr(), m.w.data_ptr(), m.n.data_ptr(), m.c_puct.data_ptr(), m.seats.data_ptr(),
            m.terminal.data_ptr(), m.children.data_ptr(), rands.data_ptr(), state.data_ptr(),
            _native.exp_table(dev).data_ptr(), B, T, A, S, parents.data_ptr(), actions.data_ptr(), _native.stream(dev)))
    return Descent(parents, actions)


def root(m):
    """mctscuda.root (cuda.cu:120-136) -> (B,A) f16 probabilities."""
    dev = _native.require_device(*m._tensors())
    B, T, A = m.logits.shape
    S = m.w.shape[-1]
    with torch.cuda.device(dev):
        state = _qrange(m, dev)
        probs = torch.empty((B, A), dtype=torch.half, device=dev)
        _native.check(_native.lib().bl_mcts_root_tuned(
            ctypes.byref(_native.tune()), m.logits.data_ptr(), m.w.data_ptr(), m.n.data_ptr(), m.c_puct.data_ptr(), m.seats.data_ptr(),
            m.terminal.data_ptr(), m.children.data_ptr(), state.data_ptr(), _native.exp_table(dev).data_ptr(),
            B, T, A, S, probs.data_ptr(), _native.stream(dev)))
    return probs


def backup(bk, leaves):
    """mctscuda.backup (cuda.cu:238-248): mutates bk.w and bk.n in place."""
    leaves = _proxy(leaves, torch.short, 1, 'leaves')
    dev = _native.require_device(bk.v, bk.w, bk.n, bk.rewards, bk.parents, bk.terminal, leaves)
    B, T, S = bk.v.shape
    with torch.cuda.device(dev):
        _native.check(_native.lib().bl_mcts_backup(
            bk.v.data_ptr(), bk.w.data_ptr(), bk.n.data_ptr(), bk.rewards.data_ptr(), bk.parents.data_ptr(),
            bk.terminal.data_ptr(), leaves.data_ptr(), B, T, S, _native.stream(dev)))
