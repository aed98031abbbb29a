#!/bin/bash
# round 4, the measurement call: everything profiles/r04_* holds for the final build, in ONE gpurun call
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
python tools/mlp_phases.py > /dev/null 2>&1 || true
bash tools/profile_all.sh r04 > $out/all.log 2>&1
bash tools/profile_cfg4.sh > $out/cfg4.log 2>&1
timeout 600 python tools/config_rates.py > $out/config_rates.txt 2>&1
timeout 900 python tools/soak.py > $out/soak.txt 2>&1
timeout 900 python tools/train_bench.py --steps 2 --buffer 8 > $out/train_bench.txt 2> $out/train_bench.err
timeout 600 python -m pytest tests/test_parallel.py -q -m gpu -k "qrange_sync" -s 2>&1 | grep -v amdgpu > $out/qrange_sync.txt
timeout 600 python tools/expand_by_sim.py "1:0 2:0" 8 > $out/expand_by_sim.txt 2>&1
bash tools/fuzz_parity.sh 301 312 > $out/fuzz_parity.txt 2>&1
ls $out gpurun_out/cfg4
tail -3 $out/per_move.txt; head -6 $out/kernel_stats.csv | cut -c1-160; grep "config 4" $out/config_rates.txt; tail -2 $out/soak.txt; tail -1 $out/train_bench.txt | cut -c1-600; tail -3 $out/qrange_sync.txt; tail -3 $out/fuzz_parity.txt
python -c "import json; d=json.load(open('$out/bench.json')); print({k: d[k] for k in ('value','ms_per_step')}, d['roofline'], {k: d['config'].get(k) for k in ('value_fold_safe','value_powf_libm','value_after_self_play_drift','value_rand_like_call_by_call','two_actors_per_gpu','search_kernels_only')}, d.get('cpu_baseline'))"
