#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
{
echo "# tools/fuzz_parity.sh on the final build, other launch variants / parity target (5 tests per seed: full-size search, bench launch sequence eager + captured, per-env c_puct)"
echo "## BL_EXPAND_ENVS=2"; BL_EXPAND_ENVS=2 bash tools/fuzz_parity.sh 401 406
echo "## BL_EXPAND_ENVS=4"; BL_EXPAND_ENVS=4 bash tools/fuzz_parity.sh 407 412
echo "## BL_POWF_LIBM=1 (device: bl_tune_t.powf_libm; checker: liboracle_powf.so)"; BL_POWF_LIBM=1 bash tools/fuzz_parity.sh 413 420
} > gpurun_out/r04o_fuzz_variants.txt 2>&1
cat gpurun_out/r04o_fuzz_variants.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_shared_workgroup_expand_in_subprocess" 2>&1 | tail -2
