#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ab() { env BL_EXPAND_WAVES=$1 timeout 600 python bench.py $2 --no-cpu-baseline --no-reference-rng --no-two-actors --no-traffic --no-soak --no-fold-safe --steps $3 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves=$1 [$2]', 'Msims/s', round(d['value']/1e6,2), 'ms/move', round(d['ms_per_step'],3), 'expand_us', round(d['roofline']['kernel_us'],1))"; }
for r in 1 2; do
  for w in 4 8; do ab $w "--envs 1024" 30; done
  for w in 2 4 8; do ab $w "--envs 2048" 30; done
  for w in 4 8; do ab $w "--envs 1024 --boardsize 13 --nodes 256 --width 1024 --depth 8" 6; done
done > gpurun_out/r04k_waves8.txt 2>&1
cat gpurun_out/r04k_waves8.txt
