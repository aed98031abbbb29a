#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>   -> gpurun_out/<tag>/{kernel_stats.csv, per_move.txt, gaps.txt}
# rocprofv3 --kernel-trace of a short bench.py run (graph replays), summarised with tools/rocpd_stats.py & friends.
tag=${1:-prof}; out=$PWD/gpurun_out/$tag; mkdir -p $out
repo=$PWD; cd /tmp; export TMPDIR=/tmp
steps=${STEPS:-8}; warm=${WARM:-2}
rm -rf /tmp/rp_$tag
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/rp_$tag -- python $repo/bench.py --steps $steps --warmup $warm --timed-only > $out/bench_under_rocprof.json 2> $out/rocprof.err
db=$(find /tmp/rp_$tag -name "*.db" | head -1)
echo "db: $db" >> $out/rocprof.err
python $repo/tools/rocpd_stats.py $db $out/kernel_stats.csv > /dev/null
# moves in the trace: the capture's eager warm-up run (1; the capture itself launches nothing) + warm-up replays + timed replays
python $repo/tools/per_move_kernels.py $db $steps > $out/per_move.txt 2>&1
python $repo/tools/gap_stats.py $db > $out/gaps.txt 2>&1
head -14 $out/kernel_stats.csv
cat $out/per_move.txt | tail -25
