#!/bin/bash
# usage (GPU box, repo root): tools/fuzz_parity_cfg4.sh <first seed> <last seed>  -- tools/fuzz_parity.sh for config 4's per-GPU shape: the bench's launch
# sequence at 13x13 / 1024 envs / 256 nodes with the 1024x8 network (eager, captured, and the launch-per-Linear plan) replayed through the oracle
for seed in $(seq ${1:-1} ${2:-5}); do
  echo -n "seed $seed: "
  BL_TEST_SEED=$seed timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "(bench_launch and 13-1024) or (full_size and 13-)" 2>&1 | tail -1
done
