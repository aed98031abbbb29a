"""bl_tune_t.powf_libm -- the second parity target: the reference AS IT BUILDS ITSELF.  Its JIT loader passes no -O flag
(boardlaw/cuda.py:29-45), so `powf(bot, 2)` at boardlaw/mcts/cpp/cpu.cpp:60 is a call into libm, not the `bot * bot` g++ makes of it from
-O1 on; glibc's powf is not correctly rounded and differs from bot * bot on 0.036 % of all floats.  boardlaw_amd/csrc/bl_powf.h restates
glibc 2.35's powf(x, 2) (x86-64 FMA variant) operation by operation; these tests pin the restatement to the HOST libm (the function the
reference's build calls), on the CPU through a g++-compiled harness over a stride of all 2^32 floats and on the GPU through bl_powf2."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = r'''
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include "%s/boardlaw_amd/csrc/bl_powf.h"
int main(int argc, char** argv) {
    const uint64_t stride = strtoull(argv[1], 0, 10), offset = strtoull(argv[2], 0, 10);
    uint64_t bad = 0, notsq = 0, n = 0;
    for (uint64_t u = offset; u < (1ull << 32); u += stride) {
        uint32_t ix = (uint32_t)u; float x; memcpy(&x, &ix, 4);
        volatile float xv = x;
        float want = powf(xv, 2.0f), got = bl_powf2_glibc(x, BLP_LOG2_TAB, BLP_EXP2_TAB);
        uint32_t a, c; memcpy(&a, &want, 4); memcpy(&c, &got, 4);
        if (a != c && !(want != want && got != got)) bad++;
        volatile float sq = xv * xv; float s = sq; uint32_t e; memcpy(&e, &s, 4);
        if (e != a && !(want != want)) notsq++;
        n++;
    }
    printf("%%llu %%llu %%llu\n", (unsigned long long)n, (unsigned long long)bad, (unsigned long long)notsq);
    return 0;
}
'''


def _glibc_with_fma():
    """The restatement is of ONE libm build: glibc's multiarch __powf_fma.  Other libms (or CPUs without FMA) compute other last bits."""
    try:
        import platform
        libc = platform.libc_ver()
        flags = open('/proc/cpuinfo').read()
        return libc[0] == 'glibc' and tuple(int(x) for x in libc[1].split('.')[:2]) >= (2, 28) and ' fma ' in flags
    except Exception:
        return False


@pytest.mark.skipif(not _glibc_with_fma(), reason='bl_powf.h restates glibc >= 2.28 powf on an x86-64 host with FMA')
def test_restatement_equals_host_libm_powf(tmp_path):
    """Every 211th float of all 2^32 (20 M values, every exponent, both signs, subnormals, infinities, NaNs): 0 differences from the
    host libm's powf(x, 2) (-fno-builtin: the compiler must not fold the call, which is the whole point), and libm itself differs
    from x * x on a few hundredths of a percent of them.  The exhaustive run (all 2^32: 0 differences, 1 548 806 = 0.0361 % unlike
    x * x) is recorded in profiles/r04_powf_exhaustive.txt."""
    src = tmp_path / 'h.cpp'
    src.write_text(HARNESS % ROOT)
    exe = tmp_path / 'h'
    subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-fno-builtin', str(src), '-o', str(exe), '-lm'])
    n, bad, notsq = (int(x) for x in subprocess.check_output([str(exe), '211', '7'], text=True).split())
    assert n > 20_000_000 and bad == 0
    assert 0.0002 < notsq / n < 0.0006


@pytest.mark.gpu
def test_device_powf2_equals_host_libm():
    """bl_powf2 (the device build of the same header: f64 FMAs on the GPU) against the host libm called through ctypes, on random bit
    patterns, a dense run of neighbouring floats and the special values."""
    import torch
    from boardlaw_amd import _native
    libm = ctypes.CDLL('libm.so.6'); libm.powf.restype = ctypes.c_float; libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(0)
    special = np.array([0, 0x80000000, 1, 0x007fffff, 0x00800000, 0x7f7fffff, 0x7f800000, 0xff800000, 0x7fc00000, 0x3f800000, 0xbf800000,
                        0x5f7fffff, 0x5f800000, 0x1e000000, 0x1f000000, 0x20000000], dtype=np.uint32)
    bits = np.concatenate([rng.integers(0, 2**32, 150000, dtype=np.uint64).astype(np.uint32), np.arange(0x3a000000, 0x3a000000 + 50000, dtype=np.uint32),
                           np.arange(0x3f7f0000, 0x3f7f0000 + 50000, dtype=np.uint32), special])
    x = bits.view(np.float32)
    want = np.array([libm.powf(float(v), 2.0) for v in x], dtype=np.float32)
    xd = torch.from_numpy(x.copy()).cuda(); out = torch.empty_like(xd)
    _native.check(_native.lib().bl_powf2(xd.data_ptr(), out.data_ptr(), xd.numel(), _native.stream(xd.device)))
    got = out.cpu().numpy()
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), (x[~same][:5], got[~same][:5], want[~same][:5])
    with np.errstate(all='ignore'):
        sq = x * x
    assert ((sq.view(np.uint32) != want.view(np.uint32)) & ~np.isnan(want)).sum() > 10      # the mode is not a no-op


@pytest.mark.gpu
def test_searches_in_powf_libm_mode_in_subprocess():
    """The searches that are compared with the oracle, again with BOTH sides in the reference's own build mode (BL_POWF_LIBM=1:
    bl_tune_t.powf_libm on the device, oracle/liboracle_powf.so -- which calls the host libm's powf -- as the checker; conftest.py): random
    trees through the general kernels, BASELINE config 2 at full size through the fused path with the integer network, the bench's
    own launch sequence with the real 512x4 network (eager and captured), per-env c_puct.  (Fixtures recorded from the -O2
    reference are not part of this run: they are the other mode's.)"""
    env = dict(os.environ, BL_POWF_LIBM='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_parity.py'), '-q', '-x', '-m', 'gpu',
                        '-k', 'test_random_trees_vs_oracle or (test_full_size_search_vs_oracle and (9-4096-64 or 5-64-16 or 13-1024-48 or 19-24-40)) or '
                              '(test_bench_launch_sequence_vs_oracle and 9-4096-64-512-4) or test_search_with_per_env_c_puct'],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout


def _cases():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'powf_case.npz'))
    return [{k: g[f'c{i}_{k}'] for k in ('logits', 'w', 'n', 'c_puct', 'seats', 'terminal', 'children', 'root_sq', 'root_powf')} for i in range(int(g['n_cases']))]


def test_the_two_parity_targets_differ_visibly_but_rarely():
    """tests/golden/powf_case.npz: trees (25, 49 and 81 actions) on which the two builds of the reference's CPU path give different root
    distributions (one probability, one f16 ulp) -- found by scanning random trees with the two oracle builds (liboracle.so: bot * bot;
    liboracle_powf.so: the host libm's powf): about ONE ROOT IN A MILLION (six processes, 3.1 M roots each, 3 finds each).  0.036 % of
    the derivative terms differ; the f16-rounded outputs absorb nearly all of it -- a whole 9x9 / 1024-env / 64-simulation search and
    10 240 random descents came out identical under both.  Env 0 of a case is the tree, env 1 carries the scanned batch's extreme
    slots so that the batch-global q range is the scanned batch's."""
    import oracle_lib
    cases = _cases()
    assert len(cases) >= 3
    for g in cases:
        d = {k: g[k] for k in ('logits', 'w', 'n', 'c_puct', 'seats', 'terminal', 'children')}
        a, b = oracle_lib.load('').root(**d), oracle_lib.load('_powf').root(**d)
        assert np.array_equal(a, g['root_sq']) and np.array_equal(b, g['root_powf'])
        assert (a[0] != b[0]).sum() >= 1 and np.array_equal(a[1], b[1])


@pytest.mark.gpu
def test_root_kernel_follows_the_mode_on_the_visible_cases():
    """The same trees through mctscuda.root's replacement (bl_mcts_root_tuned) with bl_tune_t.powf_libm off and on: each mode
    reproduces ITS oracle bit for bit -- so the switch reaches the kernel and changes what it should."""
    import torch
    from boardlaw_amd.mcts import cuda
    old = os.environ.pop('BL_POWF_LIBM', None)
    try:
        for g in _cases():
            half = lambda k: torch.from_numpy(g[k].view(np.int16)).view(torch.half).cuda()
            args = (half('logits'), half('w'), torch.from_numpy(g['n']).cuda(), half('c_puct'), torch.from_numpy(g['seats']).cuda(),
                    torch.from_numpy(g['terminal']).bool().cuda(), torch.from_numpy(g['children']).cuda())
            got = {}
            for mode in ('0', '1'):
                os.environ['BL_POWF_LIBM'] = mode
                got[mode] = cuda.root(cuda.mcts(*args)).view(torch.int16).cpu().numpy().view(np.uint16)
            assert np.array_equal(got['0'], g['root_sq']) and np.array_equal(got['1'], g['root_powf'])
            assert not np.array_equal(got['0'], got['1'])
    finally:
        os.environ.pop('BL_POWF_LIBM', None)
        if old is not None:
            os.environ['BL_POWF_LIBM'] = old


@pytest.mark.skipif(not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'), reason='needs the ROCm llvm-objdump')
def test_fused_descent_has_its_own_powf_instantiation(tmp_path):
    """bl_sim_expand's hot kernel must not carry the mode as a run-time branch (it cost 32 more spilled SGPRs and 7 VGPRs when it
    did): the mode is its own instantiation.  Checked on the built library's gfx950 code: the default instantiation of
    sim_expand2_kernel has no f64 arithmetic at all, the POWF one has glibc's f64 FMAs (and the launcher routes powf_libm searches
    to it: bl_expand.hip, bl_expand2_launch)."""
    import shutil
    lib = os.path.join(ROOT, 'boardlaw_amd', 'libboardlaw_amd.so')
    if not os.path.exists(lib):
        pytest.skip('library not built')
    work = tmp_path / 'lib.so'
    shutil.copy(lib, work)
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    subprocess.check_call([objdump, '--offloading', str(work)], stdout=subprocess.DEVNULL)
    found = {}
    for obj in sorted(tmp_path.glob('lib.so.*gfx950')):
        syms = subprocess.check_output([objdump, '-t', str(obj)], text=True)
        for line in syms.splitlines():
            name = line.split()[-1] if line.split() else ''
            for key, tag in (('default', 'sim_expand2_kernelILi3ELi1ELb1ELb0ELi2ELb0EE'), ('powf', 'sim_expand2_kernelILi3ELi1ELb0ELb0ELi2ELb1EE')):
                if tag in name and not name.endswith('.kd'):
                    dis = subprocess.check_output([objdump, '-d', f'--disassemble-symbols={name}', str(obj)], text=True)
                    found[key] = sum('_f64' in l for l in dis.splitlines())
    assert found.get('default') == 0, found
    assert found.get('powf', 0) >= 10, found
