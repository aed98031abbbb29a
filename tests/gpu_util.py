"""Helpers shared by the GPU parity tests: numpy<->torch bit-exact conversions, replay hooks, a device-independent
integer 'hash network'."""
import numpy as np
import torch


def t16(bits, device):
    """uint16 bit patterns -> torch.half on device."""
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.half).to(device)


def bits16(t):
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


def to_np(t):
    t = t.detach().contiguous().cpu()
    if t.dtype == torch.half:
        return t.view(torch.int16).numpy().view(np.uint16)
    if t.dtype == torch.bool:
        return t.numpy().astype(np.uint8)
    return t.numpy()


class ReplayRng:
    """Serves recorded (B,T) f16 uniforms to descend, in order."""

    def __init__(self, rands_bits, device):
        self.rands = [t16(r, device) for r in rands_bits]
        self.i = 0

    def rand_like(self, x):
        r = self.rands[self.i]; self.i += 1
        assert r.shape == x.shape
        return r

    def dirichlet(self, alpha, shape):
        raise AssertionError('replays plant the root instead of drawing noise')

    def categorical(self, logits):
        raise AssertionError('replays use the recorded actions')


class ReplayNetwork:
    """Returns the recorded per-simulation network outputs and records what it was shown."""

    def __init__(self, logits_bits, v_bits, device):
        self.logits = [t16(l, device) for l in logits_bits]
        self.v = [t16(v, device) for v in v_bits]
        self.i = 0
        self.seen = []

    def __call__(self, world):
        from boardlaw_amd import arrdict
        self.seen.append((to_np(world.board), to_np(world.seats), to_np(world.obs), to_np(world.valid)))
        out = arrdict.arrdict(logits=self.logits[self.i], v=self.v[self.i])
        self.i += 1
        return out


# ---- a 'network' made of integer arithmetic only, so numpy on the host and torch on the GPU agree bit for bit ----
_P1, _P2, _M = 1000003, 7919, 4093


def _logit_table():
    # 4093 negative f16 logits in [-8, 0), fixed
    k = np.arange(_M, dtype=np.float64)
    return (-(k * 0.6180339887 % 1.0) * 8).astype(np.float16)


def _v_table():
    k = np.arange(_M, dtype=np.float64)
    return ((k * 0.7548776662 % 1.0) * 1.8 - 0.9).astype(np.float16)


def hash_network_np(board, seats, valid):
    """board (B,S,S) u8, seats (B,), valid (B,A) bool -> logits f16 bits (B,A), v f16 bits (B,2)."""
    B = board.shape[0]; A = valid.shape[1]
    cells = board.reshape(B, -1).astype(np.int64)
    h = (cells * (np.arange(A, dtype=np.int64) * 31 + 17)).sum(-1) * _P1 % 2147483647
    idx = (h[:, None] + np.arange(A, dtype=np.int64)[None] * _P2) % _M
    logits = _logit_table()[idx]
    logits = np.where(valid.astype(bool), logits, np.float16(-np.inf))
    vv = _v_table()[h % _M]
    v = np.stack([vv, -vv], -1)
    s = seats.astype(np.int64)
    v = np.where(s[:, None] == 0, v, v[:, ::-1])
    return np.ascontiguousarray(logits).view(np.uint16), np.ascontiguousarray(v.astype(np.float16)).view(np.uint16)


class HashNetwork:
    """The torch twin of hash_network_np (needs world.board)."""

    def __init__(self, device):
        self.lt = torch.from_numpy(_logit_table().view(np.int16)).view(torch.half).to(device)
        self.vt = torch.from_numpy(_v_table().view(np.int16)).view(torch.half).to(device)

    def __call__(self, world):
        from boardlaw_amd import arrdict
        board, seats, valid = world.board, world.seats, world.valid
        B, A = valid.shape
        dev = valid.device
        cells = board.reshape(B, -1).long()
        ar = torch.arange(A, device=dev, dtype=torch.long)
        h = (cells * (ar * 31 + 17)).sum(-1) * _P1 % 2147483647
        idx = (h[:, None] + ar[None] * _P2) % _M
        logits = self.lt[idx].masked_fill(~valid, float('-inf'))
        vv = self.vt[h % _M]
        v = torch.stack([vv, -vv], -1)
        v = torch.where(seats.long()[:, None] == 0, v, v.flip(-1))
        return arrdict.arrdict(logits=logits, v=v)
