#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ab() { env BL_EXPAND_ENVS=$1 BL_EXPAND_HELP=$2 timeout 300 python bench.py --no-cpu-baseline --no-reference-rng --no-two-actors --no-traffic --no-soak --no-fold-safe --steps 30 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('envs=$1 help=$2', 'Msims/s', round(d['value']/1e6,2), 'ms/move', round(d['ms_per_step'],3), 'expand_us', round(d['roofline']['kernel_us'],1))"; }
for r in 1 2; do
  ab 1 0; ab 2 0; ab 2 3; ab 2 6; ab 4 0; ab 4 3; ab 4 6; ab 4 10; ab 4 1030; ab 4 1539
done > gpurun_out/r04d_ab.txt 2>&1
cat gpurun_out/r04d_ab.txt
for e in "2 6" "4 6"; do set -- $e; echo "== BL_EXPAND_ENVS=$1 BL_EXPAND_HELP=$2"; BL_EXPAND_ENVS=$1 BL_EXPAND_HELP=$2 timeout 600 python tools/deep_only_probe.py 2>&1 | grep -v amdgpu.ids | grep "sim\|all \|K =    1\|K = 1024"; done > gpurun_out/r04d_deep_only.txt 2>&1
cat gpurun_out/r04d_deep_only.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_shared_workgroup_expand_in_subprocess" 2>&1 | tail -3
