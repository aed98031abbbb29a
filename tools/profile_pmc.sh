#!/bin/bash
# usage (GPU box, repo root): [BENCH_ARGS="--envs 32768"] tools/profile_pmc.sh <tag>  -> gpurun_out/<tag>/pmc_*.csv
# rocprofv3 counter passes of a short bench run, each counter set in its own run with --kernel-trace only (no other trace
# domain), summarised per kernel with tools/pmc_summary.py; plus the FETCH/WRITE_SIZE calibration micro-benchmark.
tag=${1:-pmc}; out=$PWD/gpurun_out/$tag; mkdir -p $out
repo=$PWD; cd /tmp; export TMPDIR=/tmp
run() {   # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$name --output-format csv -- python $repo/bench.py --steps 2 --warmup 1 --timed-only $BENCH_ARGS > /dev/null 2> $out/pmc_$name.err
  python $repo/tools/pmc_summary.py /tmp/pmc_$name > $out/pmc_$name.csv
  head -4 $out/pmc_$name.csv | cut -c1-200
}
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
run TCC_HIT TCC_HIT_sum TCC_MISS_sum
run SQ1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run SQ2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/cal_$c --output-format csv -- $repo/tools/micro/fetch_calib > $out/calib_$c.txt 2> $out/calib_$c.err
  python $repo/tools/pmc_summary.py /tmp/cal_$c > $out/calib_$c.csv
  cat $out/calib_$c.txt $out/calib_$c.csv
done
