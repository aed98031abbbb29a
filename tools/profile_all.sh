#!/bin/bash
# usage (GPU box, repo root): tools/profile_all.sh <tag>  -- everything profiles/ holds for a round, into gpurun_out/<tag>/
tag=${1:-r03}; out=gpurun_out/$tag; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
STEPS=20 WARM=3 bash tools/profile_round.sh $tag > /dev/null 2>&1
bash tools/profile_pmc.sh $tag > /dev/null 2>&1
python tools/descent_stats.py 2>&1 | grep -v amdgpu.ids > $out/descent_phases.txt
[ -f tools/micro/libboardlaw_clk.so ] && python tools/mlp_phases.py 2>&1 | grep -v amdgpu.ids > $out/mlp_phases.txt
timeout 60 tools/micro/fold_variants > $out/fold_variants.txt 2>&1
timeout 60 tools/micro/arbitration > $out/arbitration.txt 2>&1
ls $out
