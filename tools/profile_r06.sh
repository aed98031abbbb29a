#!/bin/bash
# usage (GPU box, repo root): tools/profile_r06.sh  -- everything profiles/r06_* holds that is not an A/B of an experiment, in one gpurun call
out=gpurun_out/r06; mkdir -p $out
python -m pytest tests -m gpu -q > $out/gputests_full.txt 2>&1; tail -3 $out/gputests_full.txt > $out/gputests.txt
bash tools/profile_all.sh r06 > /dev/null 2>&1
python tools/envs_sweep.py > $out/envs_sweep.txt 2>&1
BENCH_ARGS="--envs 32768" bash tools/profile_quick.sh r06/prof32k > /dev/null 2>&1
(MLP_ENVS=32768 python tools/mlp_phases.py) 2>&1 | grep -v amdgpu.ids > $out/mlp_phases_32k.txt
python tools/config_rates.py 2>&1 | grep -v amdgpu.ids > $out/config_rates.txt
for c in 1 4 5; do python bench.py --config $c > $out/bench_config$c.json 2> $out/bench_config$c.err; done
python bench.py --envs 32768 --no-hex-kernels > $out/bench_envs32768.json 2> $out/bench_envs32768.err
(bash tools/fuzz_parity.sh 1 6; bash tools/fuzz_parity_32k.sh 1 3) 2>&1 | grep -v amdgpu.ids > $out/fuzz_parity.txt
python tools/soak.py 2>&1 | grep -v amdgpu.ids > $out/soak.txt
ls $out
