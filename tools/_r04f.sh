#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ab() { env BL_EXPAND_ENVS=$1 BL_EXPAND_HELP=$2 timeout 300 python bench.py --envs $3 --no-cpu-baseline --no-reference-rng --no-two-actors --no-traffic --no-soak --no-fold-safe --steps 30 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$3 envs=$1 help=$2', 'Msims/s', round(d['value']/1e6,2), 'ms/move', round(d['ms_per_step'],3), 'expand_us', round(d['roofline']['kernel_us'],1))"; }
for r in 1 2; do
  ab 1 0 4096; ab 2 0 4096; ab 4 0 4096; ab 4 1030 4096; ab 1 0 1024; ab 2 0 1024; ab 4 0 1024
done > gpurun_out/r04f_ab.txt 2>&1
cat gpurun_out/r04f_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_shared_workgroup_expand_in_subprocess" 2>&1 | tail -3
python - <<'PY'
# bl_powf2 on the device against the host libm's powf(x, 2) (ctypes, so that no compiler folds it)
import ctypes, numpy as np, torch, sys
sys.path.insert(0, '.')
from boardlaw_amd import _native
libm = ctypes.CDLL('libm.so.6'); libm.powf.restype = ctypes.c_float; libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
rng = np.random.default_rng(0)
bits = np.concatenate([rng.integers(0, 2**32, 200000, dtype=np.uint64).astype(np.uint32), np.arange(0x3f000000, 0x3f000000 + 100000, dtype=np.uint32)])
x = bits.view(np.float32)
want = np.array([libm.powf(float(v), 2.0) for v in x], dtype=np.float32)
xd = torch.from_numpy(x.copy()).cuda(); out = torch.empty_like(xd)
_native.check(_native.lib().bl_powf2(xd.data_ptr(), out.data_ptr(), xd.numel(), _native.stream(xd.device)))
got = out.cpu().numpy()
same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
sq = (x * x).astype(np.float32)
print('bl_powf2 vs host libm powf(x,2):', int((~same).sum()), 'differences of', len(x), '| libm != x*x on', int(((sq.view(np.uint32) != want.view(np.uint32)) & ~np.isnan(want)).sum()))
PY
