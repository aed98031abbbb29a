#!/bin/bash
# usage: tools/ab_layers.sh -- alternates tools/layers_time.py (config 4's network) over every variants/*.so in ONE gpurun call
cp boardlaw_amd/libboardlaw_amd.so /tmp/lib_orig.so
for r in 1 2; do for v in variants/*.so; do
  cp $v boardlaw_amd/libboardlaw_amd.so
  echo "$v: $(timeout 300 python tools/layers_time.py 13 1024 8 1024 2>&1 | tail -1)"
  echo "$v: $(timeout 300 python tools/layers_time.py 13 1024 8 256 2>&1 | tail -1)"
done; done
cp /tmp/lib_orig.so boardlaw_amd/libboardlaw_amd.so
