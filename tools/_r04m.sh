#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent_layers or layers_mlp or (bench_launch_sequence and plan) or inference_plan" 2>&1 | tail -6
for shape in "13 1024 8 1024" "13 1024 8 256" "13 1024 8 512" "9 512 4 2048" "9 512 4 1024" "13 768 6 1024"; do timeout 300 python tools/layers_time.py $shape 2>&1 | tail -1; done > gpurun_out/r04m_layers.txt 2>&1
cat gpurun_out/r04m_layers.txt
timeout 600 python tools/config_rates.py 2>&1 | grep "config 4"
timeout 600 python -m pytest tests/test_parallel.py -q -m gpu -k "qrange_sync" -s 2>&1 | grep "qrange sync ok"
