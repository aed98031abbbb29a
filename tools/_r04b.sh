#!/bin/bash
# round 4, second GPU call: parity of the shared-workgroup kernel (envs 2 / 4) and its A/B against the round-2 kernel on one box
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_shared_workgroup_expand_in_subprocess" > gpurun_out/r04b_parity.txt 2>&1; echo "parity rc $?"
tail -5 gpurun_out/r04b_parity.txt
bash tools/ab_env.sh BL_EXPAND_ENVS "1 2 4" 2 > gpurun_out/r04b_ab.txt 2>&1
cat gpurun_out/r04b_ab.txt
for e in 1 2 4; do echo "== BL_EXPAND_ENVS=$e"; BL_EXPAND_ENVS=$e timeout 600 python tools/deep_only_probe.py 2>&1 | grep -v amdgpu.ids | grep "sim\|all \|K =    1\|K = 1024"; done > gpurun_out/r04b_deep_only.txt 2>&1
cat gpurun_out/r04b_deep_only.txt
