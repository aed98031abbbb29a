#!/bin/bash
# usage: tools/sweep.sh "<groups>" "<envs>"   -- bench.py over BL_FORCE_GROUP x --envs, one short line each
for g in $1; do for e in $2; do
  BL_FORCE_GROUP=$g timeout 300 python bench.py --no-cpu-baseline --envs $e --steps 6 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('G=$g', d['config']['envs_per_gpu'], 'Msims/s', round(d['value']/1e6,2), 'ms/move', round(d['ms_per_step'],3), 'expand_us', round(d['roofline']['kernel_us'],1))"
done; done
