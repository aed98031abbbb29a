bash tools/profile_all.sh r03 > gpurun_out/r03_all.log 2>&1
bash tools/profile_cfg4.sh > gpurun_out/r03_cfg4.log 2>&1
timeout 600 python tools/config_rates.py > gpurun_out/r03/config_rates.txt 2>&1
timeout 900 python tools/soak.py > gpurun_out/r03/soak.txt 2>&1
ls gpurun_out/r03 gpurun_out/cfg4
tail -3 gpurun_out/r03/per_move.txt; head -6 gpurun_out/r03/kernel_stats.csv | cut -c1-160; grep "config 4" gpurun_out/r03/config_rates.txt; tail -2 gpurun_out/r03/soak.txt
