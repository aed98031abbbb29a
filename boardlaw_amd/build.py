"""Builds the package's native pieces in-tree:

  libboardlaw_amd.so   HIP kernels + the C ABI (include/boardlaw_amd.h), hipcc --offload-arch=gfx950
  libbl_torchgen.so    host-only shim that reads torch's generator state (csrc/bl_torchgen.cpp), host compiler + torch headers

hipcc cross-compiles without a GPU, so this runs in the build container; the .so files then travel with the tree.  Every
translation unit is compiled to its own object (in parallel, only when stale) and the objects are linked: touching one
kernel file costs one file's compile time."""
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
UNITS = ('bl_search.hip', 'bl_sim.hip', 'bl_hex.hip', 'bl_abi.hip', 'bl_expand.hip', 'bl_rows.hip', 'bl_mlp.hip', 'bl_layers.hip', 'bl_root.hip', 'bl_rand.hip')
SOURCES = [os.path.join(HERE, 'csrc', f) for f in UNITS]
# every header under csrc/ (bl_device.h includes bl_powf.h, ...): editing any of them rebuilds every object
HEADERS = [os.path.join(ROOT, 'include', 'boardlaw_amd.h')] + sorted(glob.glob(os.path.join(HERE, 'csrc', '*.h')))
OBJDIR = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libboardlaw_amd.so')
GEN_SRC = os.path.join(HERE, 'csrc', 'bl_torchgen.cpp')
GEN_LIB = os.path.join(HERE, 'libbl_torchgen.so')
GEN_STAMP = GEN_LIB + '.torch'      # the torch version the shim was compiled against: it reads torch C++ objects by layout

# -ffp-contract=off: the search kernels must round like the reference's CPU path (no FMA); see DESIGN.md.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-fno-gpu-flush-denormals-to-zero',
         '-I' + os.path.join(ROOT, 'include')]
# measurement builds only (tools/ab_build.sh: variants of a kernel's compile-time switches, e.g. BL_HIPCC_EXTRA=-DBLM_RD64=2)
FLAGS += os.environ.get('BL_HIPCC_EXTRA', '').split()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    built = os.path.getmtime(target)
    return any(os.path.getmtime(p) > built for p in deps)


def _torchgen_stale():
    """The shim touches c10::GeneratorImpl and at::PhiloxCudaState by layout, so a libbl_torchgen.so built against another torch
    is stale whatever its mtime: the version it was built for is kept in a stamp file beside it."""
    if _newer(GEN_LIB, [GEN_SRC]) or not os.path.exists(GEN_STAMP):
        return True
    import torch
    return open(GEN_STAMP).read().strip() != torch.__version__


def stale():
    return _newer(LIB, SOURCES + HEADERS + [os.path.abspath(__file__)]) or _torchgen_stale()


def _hipcc():
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libboardlaw_amd.so')
    return hipcc


def build_torchgen(force=False, verbose=False):
    if not force and not _torchgen_stale():
        return GEN_LIB
    import torch
    t = os.path.dirname(torch.__file__)
    cxx = shutil.which('g++') or shutil.which('c++')
    cmd = [cxx, '-O2', '-std=c++17', '-fPIC', '-shared', '-w', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}', f'-I{t}/include', f'-I{t}/include/torch/csrc/api/include',
           '-I/opt/rocm/include', GEN_SRC, '-o', GEN_LIB + '.tmp', f'-L{t}/lib', '-ltorch_hip', '-lc10_hip', '-lc10', '-ltorch_cpu',
           f'-Wl,-rpath,{t}/lib']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(GEN_LIB + '.tmp', GEN_LIB)
    with open(GEN_STAMP, 'w') as f:
        f.write(torch.__version__)
    return GEN_LIB


def build(force=False, verbose=False):
    build_torchgen(force, verbose)
    if not force and not _newer(LIB, SOURCES + HEADERS + [os.path.abspath(__file__)]):
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, os.path.basename(src) + '.o')
        if force or _newer(obj, [src] + HEADERS + [os.path.abspath(__file__)]):
            jobs.append([hipcc] + FLAGS + ['-c', src, '-o', obj])
    if verbose:
        for j in jobs:
            print(' '.join(j))
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as pool:
        list(pool.map(subprocess.check_call, jobs))
    objs = [os.path.join(OBJDIR, os.path.basename(src) + '.o') for src in SOURCES]
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(link))
    subprocess.check_call(link)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True))
