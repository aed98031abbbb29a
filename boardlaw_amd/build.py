"""Builds libboardlaw_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so then travels with the tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = [os.path.join(HERE, 'csrc', f) for f in ('bl_kernels.hip', 'bl_expand.hip', 'bl_mlp.hip', 'bl_root.hip')]
HEADERS = [os.path.join(ROOT, 'include', 'boardlaw_amd.h'), os.path.join(HERE, 'csrc', 'bl_device.h')]
LIB = os.path.join(HERE, 'libboardlaw_amd.so')

# -ffp-contract=off: the search kernels must round like the reference's CPU path (no FMA); see DESIGN.md.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
         '-fno-gpu-flush-denormals-to-zero',
         '-I' + os.path.join(ROOT, 'include')]


def stale():
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > built for p in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libboardlaw_amd.so')
    cmd = [hipcc] + FLAGS + SOURCES + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
