mkdir -p gpurun_out/r05e
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hex or world_step or one_player" > gpurun_out/r05e/pytest_hex.txt 2>&1
tail -5 gpurun_out/r05e/pytest_hex.txt
for S in 9 13; do timeout 600 python tools/hex_tile_ab.py $S 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05e/hex_tile_ab_$S.txt; done
