"""ctypes binding of libboardlaw_amd.so (include/boardlaw_amd.h).

There is deliberately NO fallback: if the library is missing or a call fails this raises, so a GPU run can never
silently go through PyTorch or the CPU."""
import contextlib
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, 'libboardlaw_amd.so')
QRANGE_WORDS = 4096
BL_OK, BL_EINVAL, BL_ETOOBIG, BL_ELAUNCH = 0, -1, -2, -3

_vp, _i = ctypes.c_void_p, ctypes.c_int


class Tune(ctypes.Structure):
    """bl_tune_t: explicit tuning choices (zero = defaults); results never depend on them."""
    _fields_ = [(k, _i) for k in ('fold_fast', 'expand_waves', 'expand_deep', 'expand_legacy', 'group', 'mlp_no_xcd', 'lazy_init', 'expand_envs', 'mlp_rows', 'powf_libm')]


class Search(ctypes.Structure):
    """bl_search_t"""
    _fields_ = [(k, _vp) for k in ('logits', 'v', 'w', 'n', 'children', 'parents', 'relation', 'rewards', 'terminal',
                                   'boards', 'seats', 'c_puct', 'qrange', 'exp_table')] + \
               [('B', _i), ('T', _i), ('boardsize', _i), ('obs_f16', _i), ('path', _vp), ('order', _vp), ('prio_thresh', _i),
                ('cpi', _vp), ('cca', _vp), ('nk', _vp), ('fav', _vp), ('tune', Tune), ('n_active', _vp)]


SYMBOLS = {
    'bl_abi_version': (_i, []),
    'bl_strerror': (ctypes.c_char_p, [_i]),
    'bl_exp_table_host': (_i, [_vp]),
    'bl_mcts_qrange': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'bl_qrange_decode': (_i, [_vp, _vp]),
    'bl_mcts_descend': (_i, [_vp] * 10 + [_i] * 4 + [_vp, _vp, _vp]),
    'bl_mcts_descend_tuned': (_i, [ctypes.POINTER(Tune)] + [_vp] * 10 + [_i] * 4 + [_vp, _vp, _vp]),
    'bl_mcts_root': (_i, [_vp] * 9 + [_i] * 4 + [_vp, _vp]),
    'bl_mcts_root_tuned': (_i, [ctypes.POINTER(Tune)] + [_vp] * 9 + [_i] * 4 + [_vp, _vp]),
    'bl_mcts_backup': (_i, [_vp] * 7 + [_i] * 3 + [_vp]),
    'bl_hex_step': (_i, [_vp] * 4 + [_i, _i, _vp]),
    'bl_hex_observe': (_i, [_vp] * 3 + [_i, _i, _vp]),
    'bl_hex_world_step': (_i, [_vp] * 3 + [_i] + [_vp] * 4 + [_i, _i, _vp]),
    'bl_sim_expand': (_i, [ctypes.POINTER(Search), _i] + [_vp] * 5 + [_vp]),
    'bl_sim_expand_counted': (_i, [ctypes.POINTER(Search), _i] + [_vp] * 6 + [_vp]),
    'bl_sim_backup': (_i, [ctypes.POINTER(Search), _i, _vp, _vp, _i, _vp, _i, _vp]),
    'bl_sim_finish': (_i, [ctypes.POINTER(Search), _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bl_sim_finish_f32': (_i, [ctypes.POINTER(Search), _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bl_rezero_relu_f16': (_i, [_vp, _vp, _vp, _vp, _vp, ctypes.c_long, _vp]),
    'bl_mlp_forward_f16': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'bl_mlp_layers_f16': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'bl_mlp_layers_persist_f16': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'bl_mlp_layers_xcd_f16': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'bl_root_mlp_f32': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'bl_sim_infer_finish': (_i, [ctypes.POINTER(Search), _i] + [_vp] * 11 + [_i] * 4 + [_vp]),
    'bl_sim_root': (_i, [ctypes.POINTER(Search), _i, _vp, _vp, _vp, _vp]),
    'bl_hex_observe_valid': (_i, [_vp] * 4 + [_i, _i, _vp]),
    'bl_hex_step_tiled': (_i, [_vp] * 4 + [_i, _i, _vp]),
    'bl_hex_world_step_tiled': (_i, [_vp] * 3 + [_i] + [_vp] * 4 + [_i, _i, _vp]),
    'bl_hex_observe_valid_tiled': (_i, [_vp] * 4 + [_i, _i, _vp]),
    'bl_sim_n_leaves': (_i, [ctypes.POINTER(Search), _vp, _vp]),
    'bl_rezero_relu_f32': (_i, [_vp] * 5 + [ctypes.c_long, _vp]),
    'bl_sim_plant_root': (_i, [ctypes.POINTER(Search)] + [_vp] * 5 + [ctypes.c_float, _vp]),
    'bl_sim_plant_root_gamma': (_i, [ctypes.POINTER(Search)] + [_vp] * 5 + [ctypes.c_float, _vp]),
    'bl_categorical': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'bl_sim_init': (_i, [ctypes.POINTER(Search), _vp, _vp, _vp]),
    'bl_sim_compact': (_i, [ctypes.POINTER(Search), _vp, _vp]),
    'bl_draw_actions': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'bl_copy_many': (_i, [_vp, _i, _vp]),
    'bl_rand_block': (_i, [_vp, _i, ctypes.c_long, ctypes.c_long, _i, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_uint, _i, _i, _vp]),
    'bl_selftest': (_i, [_vp]),
    'bl_powf2': (_i, [_vp, _vp, ctypes.c_long, _vp]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Loads the library. torch is imported first on purpose: its bundled HIP runtime (soname libamdhip64.so.7) is then
    the one the library binds to, so stream handles and device pointers from torch are valid inside it."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise NativeError(f'{LIBPATH} is missing: run `python -m boardlaw_amd.build` (needs hipcc). '
                              'boardlaw_amd has no CPU or PyTorch fallback for its kernels.')
        L = ctypes.CDLL(LIBPATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


_fold_fast = {}


def fold_fast(device):
    """1 if bl_sim_expand may use its one-wait-state DPP fold on `device`: bl_selftest() reproduced every prefix total with it
    THERE (run once per device, outside any capture), and BL_FOLD_SAFE is not set.  Inside a capture on a device that has not
    been tested yet the ISA-padded fold is used (and nothing is cached)."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    if index not in _fold_fast:
        if torch.cuda.is_current_stream_capturing():
            return 0
        with torch.cuda.device(index):
            rc = lib().bl_selftest(torch.cuda.current_stream().cuda_stream)
        if rc < 0:
            raise NativeError(f'libboardlaw_amd: device self-test failed on cuda:{index} ({lib().bl_strerror(rc).decode()})')
        _fold_fast[index] = int(rc == 0)
    return 0 if _env_int('BL_FOLD_SAFE') else _fold_fast[index]


def _env_int(name, default=0):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


def tune(device=None):
    """bl_tune_t for searches on `device`.  The BL_* environment variables are experiment switches of THIS host layer (tools/,
    parity tests of the kernel variants); the library itself reads no environment."""
    return Tune(fold_fast=fold_fast(device) if device is not None else 0, expand_waves=_env_int('BL_EXPAND_WAVES'),
                expand_deep=_env_int('BL_EXPAND_DEEP'), expand_legacy=_env_int('BL_EXPAND_LEGACY'), group=_env_int('BL_FORCE_GROUP'),
                mlp_no_xcd=int(os.environ.get('BL_MLP_XCD', '1') == '0'), mlp_rows=_env_int('BL_MLP_ROWS'), powf_libm=_env_int('BL_POWF_LIBM'))


GENLIBPATH = os.path.join(HERE, 'libbl_torchgen.so')
_genlib = None


def philox_state(generator, increment):
    """What a torch random kernel gets from `gen->philox_cuda_state(increment)` (csrc/bl_torchgen.cpp): (seed, offset,
    offset_intragraph, captured) -- seed/offset are device pointers when `captured` (inside HIP-graph capture).  Advances the
    generator by `increment` like the torch kernel it stands in for."""
    global _genlib
    if _genlib is None:
        if not os.path.exists(GENLIBPATH):
            raise NativeError(f'{GENLIBPATH} is missing: run `python -m boardlaw_amd.build`')
        stamp = GENLIBPATH + '.torch'
        if os.path.exists(stamp) and open(stamp).read().strip() != torch.__version__:
            # the shim reads c10::GeneratorImpl / at::PhiloxCudaState by layout: never run it against another torch
            raise NativeError(f'{GENLIBPATH} was built for torch {open(stamp).read().strip()}, this is {torch.__version__}: '
                              'run `python -m boardlaw_amd.build`')
        L = ctypes.CDLL(GENLIBPATH)
        L.bl_torch_philox_state.restype, L.bl_torch_philox_state.argtypes = _i, [_vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int64)]
        _genlib = L
    out = (ctypes.c_int64 * 4)()
    if _genlib.bl_torch_philox_state(generator._cdata, increment, out) != 0:
        raise NativeError('bl_torch_philox_state: torch raised while reading the generator state')
    return int(out[0]) & (2 ** 64 - 1), int(out[1]) & (2 ** 64 - 1), int(out[2]), int(out[3])


def check(rc):
    if rc != 0:
        raise NativeError(f'libboardlaw_amd: {lib().bl_strerror(rc).decode()} (code {rc})')


def stream(device=None):
    return torch.cuda.current_stream(device).cuda_stream


class Copy(ctypes.Structure):
    """bl_copy_t"""
    _fields_ = [('src', _vp), ('dst', _vp)] + [(k, ctypes.c_ulonglong) for k in ('row_bytes', 'rows', 'src_pitch', 'dst_pitch')]


COPY_MAX = 24


def clone_many(tensors):
    """Fresh contiguous copies of device tensors with one bl_copy_many launch per COPY_MAX of them (a .clone() each is a
    launch each).  Handles contiguous tensors and views whose rows along dim 0 are contiguous (tree[:, 0] slices); anything
    else is copied by torch."""
    fresh = [torch.empty(t.shape, dtype=t.dtype, device=t.device) for t in tensors]
    items, dev = [], None
    for src, dst in zip(tensors, fresh):
        size = src.element_size()
        if src.numel() == 0:
            continue
        if not src.is_cuda or (dev is not None and src.device != dev):
            dst.copy_(src)
        elif src.is_contiguous():
            items.append(Copy(src.data_ptr(), dst.data_ptr(), src.numel() * size, 1, 0, 0)); dev = src.device
        elif src.dim() >= 1 and src[0].is_contiguous() and src.stride(0) >= src[0].numel():
            row = src[0].numel() * size
            items.append(Copy(src.data_ptr(), dst.data_ptr(), row, src.shape[0], src.stride(0) * size, row)); dev = src.device
        else:
            dst.copy_(src)
    with torch.cuda.device(dev) if dev is not None else contextlib.nullcontext():
        for i in range(0, len(items), COPY_MAX):
            chunk = items[i:i + COPY_MAX]
            check(lib().bl_copy_many((Copy * len(chunk))(*chunk), len(chunk), stream(dev)))
    return fresh


def copy_many(dsts, srcs):
    """dst.copy_(src) for same-shape, same-dtype contiguous device tensors, one launch."""
    items = []
    for dst, src in zip(dsts, srcs):
        if not (dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.shape == src.shape and dst.device == src.device and src.is_cuda):
            raise NativeError('copy_many wants contiguous device tensors of equal shape and dtype')
        items.append(Copy(src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size(), 1, 0, 0))
    if items:
        with torch.cuda.device(dsts[0].device):
            for i in range(0, len(items), COPY_MAX):
                chunk = items[i:i + COPY_MAX]
                check(lib().bl_copy_many((Copy * len(chunk))(*chunk), len(chunk), stream(dsts[0].device)))


def ptr(t):
    return t.data_ptr()


def require_device(*tensors):
    devs = {t.device for t in tensors}
    if len(devs) != 1:
        raise AssertionError('Inputs span multiple devices')
    dev = devs.pop()
    if dev.type != 'cuda':
        raise NativeError('boardlaw_amd kernels run on an MI355X only: tensors must live on a cuda (HIP) device, '
                          f'got {dev}. There is no CPU fallback.')
    return dev


_exp_tables = {}


def exp_table(device):
    """pi = expf(logit) for all 65536 binary16 patterns, from the host libm (cpu.cpp:86), uploaded once per device."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _exp_tables:
        host = torch.empty(65536, dtype=torch.float32)
        check(lib().bl_exp_table_host(host.data_ptr()))
        _exp_tables[key] = host.to(device)
    return _exp_tables[key]


_log_tables = {}


def log_table(device):
    """f16 -> f16 table of the reference's CPU root read-out `r.float().log().half()` (mcts/__init__.py:147) over all
    65536 binary16 inputs, so the root logits on the GPU carry the reference CPU path's bits, not the device log's."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _log_tables:
        bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
        _log_tables[key] = bits.view(torch.half).float().log().half().to(device)
    return _log_tables[key]


def qrange_decode(state):
    host = state.detach().cpu().contiguous()
    out = torch.empty(2, dtype=torch.float32)
    check(lib().bl_qrange_decode(host.data_ptr(), out.data_ptr()))
    return out
