mkdir -p gpurun_out/r05d
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent_layers or removed_shared" > gpurun_out/r05d/pytest_persist.txt 2>&1
tail -5 gpurun_out/r05d/pytest_persist.txt
for shape in "13 1024 8 1024" "13 1024 8 256" "9 512 4 2048" "13 768 6 1024"; do timeout 300 python tools/layers_time.py $shape 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05d/layers_time.txt; done
cat gpurun_out/r05d/layers_time.txt
timeout 600 python -m pytest tests/test_training.py -x -q -m gpu -k "masked_captured" > gpurun_out/r05d/pytest_pad.txt 2>&1; tail -3 gpurun_out/r05d/pytest_pad.txt
