"""Native Hex kernels behind the reference's `boardlaw.hex.cuda` surface (boardlaw/hex/cuda.py:11-16): `step`, `observe`.

Same argument meaning, dtypes and in-place behaviour as hexcuda.step / hexcuda.observe
(boardlaw/hex/cpp/wrappers.cpp:20-34); the work is done by libboardlaw_amd.so on the current HIP stream."""
import torch

from .. import _native


def _check(t, dtype, ndim, name):
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')
    if t.dtype != dtype:
        raise TypeError(f'expected {dtype} got {t.dtype}')
    if ndim is not None and t.ndim != ndim:
        raise RuntimeError(f'{name}: expected {ndim} dims got {t.ndim}')


def step(board, seats, actions):
    """board (B,S,S) u8 is mutated in place; seats, actions (B,) i32; returns rewards (B,2) f32."""
    _check(board, torch.uint8, 3, 'board'); _check(seats, torch.int32, 1, 'seats'); _check(actions, torch.int32, 1, 'actions')
    dev = _native.require_device(board, seats, actions)
    B, S, _ = board.shape
    rewards = torch.empty((B, 2), dtype=torch.float32, device=dev)
    if B:
        with torch.cuda.device(dev):
            _native.check(_native.lib().bl_hex_step(board.data_ptr(), seats.data_ptr(), actions.data_ptr(),
                                                    rewards.data_ptr(), B, S, _native.stream(dev)))
    return rewards


def world_step(board, seats, actions):
    """Hex.step with reset=True in one launch: (new_board u8 (B,S,S), new_seats i32 (B,), rewards f32 (B,2), terminal bool (B,));
    the inputs are left untouched.  actions i32 or i64."""
    _check(board, torch.uint8, 3, 'board'); _check(seats, torch.int32, 1, 'seats')
    if actions.dtype not in (torch.int32, torch.int64) or not actions.is_contiguous():
        actions = actions.long().contiguous()
    dev = _native.require_device(board, seats, actions)
    B, S, _ = board.shape
    new_board, new_seats = torch.empty_like(board), torch.empty_like(seats)
    rewards = torch.empty((B, 2), dtype=torch.float32, device=dev)
    terminal = torch.empty((B,), dtype=torch.bool, device=dev)
    if B:
        with torch.cuda.device(dev):
            _native.check(_native.lib().bl_hex_world_step(board.data_ptr(), seats.data_ptr(), actions.data_ptr(),
                                                          int(actions.dtype == torch.int64), new_board.data_ptr(),
                                                          new_seats.data_ptr(), rewards.data_ptr(), terminal.data_ptr(), B, S,
                                                          _native.stream(dev)))
    return new_board, new_seats, rewards, terminal


def observe(board, seats):
    """board (..., S, S) u8, seats (...) any int -> obs (..., S, S, 2) f32."""
    S = board.shape[-1]
    flat = board.reshape(-1, S, S).contiguous()
    fseats = seats.reshape(-1).to(torch.int32).contiguous()
    _check(flat, torch.uint8, 3, 'board')
    dev = _native.require_device(flat, fseats)
    obs = torch.empty((flat.shape[0], S, S, 2), dtype=torch.float32, device=dev)
    if flat.shape[0]:
        with torch.cuda.device(dev):
            _native.check(_native.lib().bl_hex_observe(flat.data_ptr(), fseats.data_ptr(), obs.data_ptr(),
                                                       flat.shape[0], S, _native.stream(dev)))
    return obs.view(*board.shape, 2)


def observe_valid(board, seats):
    """board (B,S,S) u8, seats (B,) i32 -> (obs (B,S,S,2) f32, valid (B,S*S) bool) in one launch."""
    _check(board, torch.uint8, 3, 'board'); _check(seats, torch.int32, 1, 'seats')
    dev = _native.require_device(board, seats)
    B, S, _ = board.shape
    obs = torch.empty((B, S, S, 2), dtype=torch.float32, device=dev)
    valid = torch.empty((B, S * S), dtype=torch.bool, device=dev)
    if B:
        with torch.cuda.device(dev):
            _native.check(_native.lib().bl_hex_observe_valid(board.data_ptr(), seats.data_ptr(), obs.data_ptr(), valid.data_ptr(),
                                                             B, S, _native.stream(dev)))
    return obs, valid
