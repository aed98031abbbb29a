#!/bin/bash
# round 4, first GPU call: the row-packed iteration's gate + the deep-only probe + a baseline bench line of HEAD on this box
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 tools/micro/fold_rows > gpurun_out/r04_fold_rows.txt 2>&1; echo "fold_rows rc $?"
timeout 900 python tools/deep_only_probe.py > gpurun_out/r04_deep_only.txt 2>&1; echo "deep_only rc $?"
timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-soak --no-two-actors --no-reference-rng > gpurun_out/r04_bench_head.json 2> gpurun_out/r04_bench_head.err; echo "bench rc $?"
tail -3 gpurun_out/r04_fold_rows.txt; tail -5 gpurun_out/r04_deep_only.txt
