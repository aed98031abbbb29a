#!/bin/bash
# usage (GPU box, repo root): tools/fuzz_parity_32k.sh <first seed> <last seed>  -- the oracle comparisons at the reference's own actor
# shape (9x9, 32768 envs, 64 sims: a whole search with the integer network; the bench's launch sequence with the real network as a
# captured graph) on other pre-mixed positions, uniforms and network weights than the committed test run uses.  One line per seed.
for seed in $(seq ${1:-1} ${2:-3}); do
  echo -n "seed $seed: "
  BL_TEST_SEED=$seed timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "(full_size and 9-32768) or (bench_launch and 9-32768)" 2>&1 | tail -1
done
