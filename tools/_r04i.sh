#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cp boardlaw_amd/libboardlaw_amd.so /tmp/lib_orig.so
for r in 1 2 3; do for v in variants/a_ring2.so variants/b_ringfull.so; do
  cp $v boardlaw_amd/libboardlaw_amd.so
  timeout 300 python bench.py --no-cpu-baseline --no-reference-rng --no-two-actors --no-traffic --no-soak --no-fold-safe --steps 40 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'Msims/s', round(d['value']/1e6,2), 'ms/move', round(d['ms_per_step'],3), 'expand_us', round(d['roofline']['kernel_us'],1))"
done; done > gpurun_out/r04i_ring_ab.txt 2>&1
cp /tmp/lib_orig.so boardlaw_amd/libboardlaw_amd.so
cat gpurun_out/r04i_ring_ab.txt
for v in variants/a_ring2.so variants/b_ringfull.so; do cp $v boardlaw_amd/libboardlaw_amd.so; echo "== $v"; timeout 300 python tools/layers_time.py 2>&1 | tail -6; done > gpurun_out/r04i_layers.txt 2>&1
cp /tmp/lib_orig.so boardlaw_amd/libboardlaw_amd.so
cat gpurun_out/r04i_layers.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mlp or infer_finish or bench_launch_sequence or inference_plan" 2>&1 | tail -3
