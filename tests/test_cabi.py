"""CPU-side checks of the drop-in boundary: the library loads, exports exactly what include/boardlaw_amd.h declares,
validates arguments without a GPU, and the host-side mirror fails loudly (never falls back) without one."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'boardlaw_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bl_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from boardlaw_amd import _native
    L = _native.lib()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f'{s} declared in the header but not exported'
        assert s in _native.SYMBOLS, f'{s} has no ctypes signature in boardlaw_amd/_native.py'
    assert sorted(_native.SYMBOLS) == syms
    assert L.bl_abi_version() == 4


def test_argument_validation_without_gpu():
    from boardlaw_amd import _native
    L = _native.lib()
    assert L.bl_mcts_backup(None, None, None, None, None, None, None, 1, 1, 1, None) == -1
    assert L.bl_hex_step(None, None, None, None, 4, 3, None) == -1
    one = ctypes.c_void_p(1)
    assert L.bl_hex_observe(one, one, one, 4, 33, None) == -2
    assert L.bl_mcts_descend(*([one] * 10), 1, 1, 2000, 1, one, one, None) == -2
    assert b'limits' in L.bl_strerror(-2)


@pytest.mark.parametrize('ctype,mirror', [('bl_search_t', 'Search'), ('bl_copy_t', 'Copy'), ('bl_tune_t', 'Tune')])
def test_struct_layout_matches_the_header(tmp_path, ctype, mirror):
    """The header's structs as gcc lays them out from include/boardlaw_amd.h == their ctypes mirrors, field by field."""
    import subprocess
    from boardlaw_amd import _native
    cls = getattr(_native, mirror)
    names = [f[0] for f in cls._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "boardlaw_amd.h"\nint main(void) {\n'
                   + ''.join(f'printf("{n} %zu\\n", offsetof({ctype}, {n}));\n' for n in names)
                   + f'printf("sizeof %zu\\n", sizeof({ctype})); return 0; }}\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    assert int(got.pop('sizeof')) == ctypes.sizeof(cls)
    assert {k: int(v) for k, v in got.items()} == {n: getattr(cls, n).offset for n in names}
    if mirror == 'Copy':
        assert _native.COPY_MAX == int(re.search(r'#define BL_COPY_MAX (\d+)', open(os.path.join(ROOT, 'include', 'boardlaw_amd.h')).read()).group(1))


def test_library_reads_no_environment():
    """re-entrant, no global state -- every tuning choice is a bl_tune_t field set by the caller; the library
    imports neither getenv nor secure_getenv (the BL_* switches live in the Python host layer, boardlaw_amd/_native.py: tune)."""
    import subprocess
    from boardlaw_amd import _native
    undefined = subprocess.check_output(['nm', '-D', '--undefined-only', _native.LIBPATH]).decode()
    assert 'getenv' not in undefined, [l for l in undefined.splitlines() if 'getenv' in l]


def test_exp_table_is_host_libm(oracle):
    from boardlaw_amd import _native
    t = torch.empty(65536, dtype=torch.float32)
    _native.check(_native.lib().bl_exp_table_host(t.data_ptr()))
    assert np.array_equal(t.numpy().view(np.uint32), oracle.exp_table().view(np.uint32))


def test_qrange_decode_roundtrip():
    from boardlaw_amd import _native
    def enc(f):
        b = np.float32(f).view(np.uint32)
        return np.uint32(~b) if b & 0x80000000 else np.uint32(b | 0x80000000)
    # words in memory are the codes XOR 0x80000000: order-preserving as SIGNED int32, identity of the MAX = 0x80000000 (ABI 4)
    bias = np.uint32(0x80000000)
    st = np.full(4096, bias, np.uint32)
    st[64 * 5] = ~enc(-3.5) ^ bias; st[64 * 5 + 1] = enc(0.25) ^ bias; st[64 * 9] = ~enc(1.0) ^ bias; st[64 * 9 + 1] = enc(-7.0) ^ bias
    assert _native.qrange_decode(torch.from_numpy(st.view(np.int32))).tolist() == [-3.5, 0.25]
    # the signed order of the memory words is the order of the floats: a plain int32 max merges two states
    vals = np.array([-7.0, -3.5, -0.0, 0.0, 1e-30, 0.25, 1.0, 65504.0], np.float32)
    words = np.array([int(np.int32(enc(v) ^ bias)) for v in vals])
    assert (np.diff(words) > 0).all()
    from boardlaw_amd import parallel
    other = np.full(4096, bias, np.uint32); other[64 * 7] = ~enc(-9.0) ^ bias; other[64 * 7 + 1] = enc(0.125) ^ bias
    merged = parallel.merge_qrange(torch.from_numpy(st.view(np.int32)), torch.from_numpy(other.view(np.int32)))
    assert _native.qrange_decode(merged).tolist() == [-9.0, 0.25]


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_no_silent_cpu_fallback():
    from boardlaw_amd import _native
    from boardlaw_amd.hex import Hex
    w = Hex.initial(2, 3, device='cpu')
    with pytest.raises(_native.NativeError, match='no CPU fallback'):
        w.obs
    from boardlaw_amd.mcts import MCTS
    with pytest.raises(_native.NativeError):
        MCTS(w, n_nodes=4)


def test_host_mirror_surface():
    """Names the reference's callers use."""
    from boardlaw_amd.mcts import cuda as mcuda, MCTS, MCTSAgent, DummyAgent, mcts, dirichlet_noise  # noqa: F401
    from boardlaw_amd.hex import cuda as hcuda, Hex  # noqa: F401
    for f in ('mcts', 'Backup', 'descend', 'root', 'backup'):
        assert callable(getattr(mcuda, f))
    for f in ('step', 'observe'):
        assert callable(getattr(hcuda, f))
    with pytest.raises(TypeError, match='expected Half got Float'):
        mcuda.Backup(v=torch.zeros(1, 2, 1), w=torch.zeros(1, 2, 1).half(), n=torch.zeros(1, 2).short(),
                     rewards=torch.zeros(1, 2, 1).half(), parents=torch.zeros(1, 2).short(), terminal=torch.zeros(1, 2).bool())


def test_fcmodel_matches_reference_parameter_names_and_init_order():
    from boardlaw_amd import networks, heads
    torch.manual_seed(0)
    net = networks.FCModel(heads.Tensor((5, 5, 2)), heads.Masked(25), width=16, depth=2)
    keys = list(net.state_dict().keys())
    assert keys == ['policy.core.weight', 'policy.core.bias', 'body.0.weight', 'body.0.bias', 'body.1.weight',
                    'body.1.bias', 'body.1.α', 'body.2.weight', 'body.2.bias', 'body.2.α', 'value.core.weight',
                    'value.core.bias']
    class W: pass
    w = W(); w.obs = torch.zeros(3, 5, 5, 2); w.valid = torch.ones(3, 25, dtype=torch.bool); w.valid[:, 3] = False
    w.seats = torch.tensor([0, 1, 0])
    d = net(w)
    assert d.logits.shape == (3, 25) and torch.isinf(d.logits[:, 3]).all() and d.v.shape == (3, 2)
    assert torch.allclose(d.v[:, 0], -d.v[:, 1])
