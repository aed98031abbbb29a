#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." [rounds] [bench args...] -- alternates bench.py over values of one BL_* host switch in ONE
# gpurun call (boxes differ by up to 25 % in kernel time; only same-box numbers compare)
var=$1; vals=$2; rounds=${3:-2}; shift 3
for r in $(seq $rounds); do for v in $vals; do
  env $var=$v timeout 300 python bench.py --no-cpu-baseline --no-reference-rng --no-two-actors --no-traffic --no-soak --steps 30 "$@" 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', 'Msims/s', round(d['value']/1e6,2), 'ms/move', round(d['ms_per_step'],3), 'expand_us', round(d['roofline']['kernel_us'],1))"
done; done
