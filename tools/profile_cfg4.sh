repo=$PWD; out=$repo/gpurun_out/cfg4; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/rp_cfg4
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/rp_cfg4 -- python $repo/bench.py --boardsize 13 --nodes 256 --width 1024 --depth 8 --envs 1024 --steps 4 --warmup 1 --timed-only > $out/bench.json 2> $out/err.txt
db=$(find /tmp/rp_cfg4 -name "*.db" | head -1)
python $repo/tools/rocpd_stats.py $db $out/kernel_stats.csv > /dev/null
head -12 $out/kernel_stats.csv | cut -c1-200; tail -1 $out/bench.json | cut -c1-300
