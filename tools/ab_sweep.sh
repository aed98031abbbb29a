#!/bin/bash
# usage: tools/ab_sweep.sh [rounds] [envs_sweep args...]  -- alternates tools/envs_sweep.py over every variants/*.so (copied over the
# in-tree library) in ONE gpurun call: per-kernel microseconds of bl_sim_expand / bl_sim_infer_finish per variant, same box.
rounds=${1:-2}; shift
cp boardlaw_amd/libboardlaw_amd.so /tmp/lib_orig.so
for r in $(seq $rounds); do for v in variants/*.so; do
  cp $v boardlaw_amd/libboardlaw_amd.so
  echo "== $v"; timeout 600 python tools/envs_sweep.py "$@" 2>&1 | grep -E "^ +[0-9]+ "
done; done
cp /tmp/lib_orig.so boardlaw_amd/libboardlaw_amd.so
