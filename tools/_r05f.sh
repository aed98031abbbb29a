mkdir -p gpurun_out/r05f
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05f/gputests.txt 2>&1
tail -4 gpurun_out/r05f/gputests.txt
python bench.py > gpurun_out/r05f/bench.json 2> gpurun_out/r05f/bench.err
tail -c 600 gpurun_out/r05f/bench.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05f/smoke.txt 2>&1; tail -2 gpurun_out/r05f/smoke.txt
