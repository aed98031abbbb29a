#!/bin/bash
# usage: tools/ab.sh [rounds] [bench args...]  -- alternates bench.py over every variants/*.so (copied over the in-tree
# library) in ONE gpurun call, because boxes differ by up to 25 % in kernel time and only same-box numbers compare.
rounds=${1:-2}; shift
cp boardlaw_amd/libboardlaw_amd.so /tmp/lib_orig.so
for r in $(seq $rounds); do for v in variants/*.so; do
  cp $v boardlaw_amd/libboardlaw_amd.so
  timeout 300 python bench.py --no-cpu-baseline --no-reference-rng --no-two-actors --no-traffic --no-soak --steps 30 "$@" 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'Msims/s', round(d['value']/1e6,2), 'ms/move', round(d['ms_per_step'],3), 'expand_us', round(d['roofline']['kernel_us'],1))"
done; done
cp /tmp/lib_orig.so boardlaw_amd/libboardlaw_amd.so
