mkdir -p gpurun_out/r3fuzz
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config_4_per_gpu_shape or lazy_reset" 2>&1 | tail -3
bash tools/fuzz_parity.sh 201 230 > gpurun_out/r3fuzz/fuzz_9x9.txt 2>&1
for seed in $(seq 231 238); do echo -n "seed $seed (13x13, 1024 envs, 256 nodes, four waves per env; 5x5; 9x9 at 333/512 envs): "; BL_TEST_SEED=$seed timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bench_launch and not 9-4096" 2>&1 | tail -1; done > gpurun_out/r3fuzz/fuzz_other.txt 2>&1
tail -3 gpurun_out/r3fuzz/fuzz_9x9.txt; tail -3 gpurun_out/r3fuzz/fuzz_other.txt
