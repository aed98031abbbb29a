#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_rng_stream.py tests/test_gpu_parity.py -x -q -m gpu -k "rng or graphed or two_actors or planted or plant_root or bench_launch_sequence or actor_learner or masked or ragged" 2>&1 | tail -4
ab() { env BL_FORK_DRAWS=$1 timeout 300 python bench.py --no-cpu-baseline --no-reference-rng --no-traffic --no-soak --no-fold-safe --steps 40 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fork_draws=$1', 'Msims/s', round(d['value']/1e6,2), 'ms/move', round(d['ms_per_step'],3), 'two actors', round(d['config']['two_actors_per_gpu']['sims_per_sec']/1e6, 2))"; }
for r in 1 2 3; do ab 0; ab 1; done > gpurun_out/r04n_fork_ab.txt 2>&1
cat gpurun_out/r04n_fork_ab.txt
