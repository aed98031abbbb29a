#!/bin/bash
# usage (GPU box, repo root): tools/fuzz_parity.sh <first seed> <last seed>  -- the full-size oracle comparisons (a whole 9x9/4096/64
# search with the integer network; the bench's launch sequence with the real network, eager and captured) on other pre-mixed
# positions, uniforms and network weights than the committed test run uses.  Prints one line per seed.
for seed in $(seq ${1:-1} ${2:-5}); do
  echo -n "seed $seed: "
  BL_TEST_SEED=$seed timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "(full_size and 9-4096) or (bench_launch and 9-4096) or c_puct" 2>&1 | tail -1
done
