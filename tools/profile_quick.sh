#!/bin/bash
# usage (GPU box, repo root): [BENCH_ARGS="--envs 32768"] tools/profile_quick.sh <tag>  -> gpurun_out/<tag>/{kernel_stats.csv,pmc_SQ1.csv}
# one rocprofv3 --kernel-trace --stats pass and one SQ counter pass of a short bench run (tools/profile_pmc.sh does all counter sets)
tag=${1:-quick}; out=$PWD/gpurun_out/$tag; mkdir -p $out
repo=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt_$$; rocprofv3 --kernel-trace --stats -d /tmp/kt_$$ --output-format csv -- python $repo/bench.py --steps 3 --warmup 1 --timed-only $BENCH_ARGS > /dev/null 2> $out/kt.err
f=$(find /tmp/kt_$$ -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats.csv; head -8 $out/kernel_stats.csv | cut -c1-220
rm -rf /tmp/pmc_$$; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/pmc_$$ --output-format csv -- python $repo/bench.py --steps 2 --warmup 1 --timed-only $BENCH_ARGS > /dev/null 2> $out/pmc.err
python $repo/tools/pmc_summary.py /tmp/pmc_$$ > $out/pmc_SQ1.csv; head -5 $out/pmc_SQ1.csv | cut -c1-260
