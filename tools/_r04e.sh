#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ab() { env BL_EXPAND_ENVS=$1 BL_EXPAND_HELP=$2 timeout 300 python bench.py --envs $3 --no-cpu-baseline --no-reference-rng --no-two-actors --no-traffic --no-soak --no-fold-safe --steps 30 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$3 envs=$1 help=$2', 'Msims/s', round(d['value']/1e6,2), 'ms/move', round(d['ms_per_step'],3), 'expand_us', round(d['roofline']['kernel_us'],1))"; }
for r in 1 2; do
  for B in 512 1024 2048; do ab 1 0 $B; ab 2 0 $B; ab 4 0 $B; ab 4 4 $B; done
done > gpurun_out/r04e_ab_small.txt 2>&1
cat gpurun_out/r04e_ab_small.txt
timeout 900 python -m pytest tests/test_parallel.py -x -q -m gpu -k "qrange_sync or train_bench" -s 2>&1 | tail -8
timeout 600 python -m pytest tests/test_reference_fixtures.py tests/test_rng_stream.py -x -q -m gpu 2>&1 | tail -3
