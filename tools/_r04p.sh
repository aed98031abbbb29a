#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/ab.sh 3 --steps 40 > gpurun_out/r04p_fin_prefetch_ab.txt 2>&1
cat gpurun_out/r04p_fin_prefetch_ab.txt
cp variants/b_fin_late.so boardlaw_amd/libboardlaw_amd.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "infer_finish or bench_launch_sequence or finish_heads" 2>&1 | tail -3
