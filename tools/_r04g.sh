#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_powf.py -x -q -m gpu 2>&1 | tail -5
ab() { env BL_POWF_LIBM=$1 timeout 300 python bench.py --no-cpu-baseline --no-reference-rng --no-two-actors --no-traffic --no-soak --no-fold-safe --steps 30 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('powf_libm=$1', 'Msims/s', round(d['value']/1e6,2), 'ms/move', round(d['ms_per_step'],3), 'expand_us', round(d['roofline']['kernel_us'],1))"; }
for r in 1 2; do ab 0; ab 1; done > gpurun_out/r04g_powf_ab.txt 2>&1
cat gpurun_out/r04g_powf_ab.txt
