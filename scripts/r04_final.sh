cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
python3 bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --workload hrnet_x4 --batch 4 > gpurun_out/r04/bench_hrnet_x4.json 2>/dev/null
python3 bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --workload blurskip_x8 --lr-size 224 --batch 4 > gpurun_out/r04/bench_blurskip_x8.json 2>/dev/null
tail -c 400 gpurun_out/r04/bench.json
