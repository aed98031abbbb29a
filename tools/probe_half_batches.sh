set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p1
for W in 2 4; do
  for E in 4096 2048 1024; do
    BL_EXPAND_WAVES=$W timeout 300 python bench.py --envs $E --eager --no-cpu-baseline --no-reference-rng --no-two-actors --no-traffic --no-soak --steps 5 --warmup 2 > gpurun_out/p1/eager_w${W}_e${E}.json 2> gpurun_out/p1/eager_w${W}_e${E}.err
  done
  BL_EXPAND_WAVES=$W timeout 600 python tools/two_group_probe.py > gpurun_out/p1/groups_w${W}.txt 2>&1
done
