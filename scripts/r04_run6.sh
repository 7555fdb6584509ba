cd $GRAFT_REPO_ROOT
python -m pytest tests/test_joint_gpu.py tests/test_determinism_gpu.py tests/test_data_parallel_gpu.py tests/test_bench_two_ranks_gpu.py tests/test_kbpn_gather_gpu.py tests/test_hrnet_gpu.py -q -x 2>&1 | tail -5 > gpurun_out/r04_t6.log
for v in 0 1; do
CSBSR_WGRAD_STREAM=$v python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg > gpurun_out/r04_ab_wgs$v.json 2>/dev/null
done
CSBSR_WGRAD_STREAM=1 python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --workload blurskip_x8 --lr-size 224 --batch 4 > gpurun_out/r04_ab_bs_wgs1.json 2>/dev/null
CSBSR_WGRAD_STREAM=1 python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --workload hrnet_x4 --batch 4 > gpurun_out/r04_ab_hr_wgs1.json 2>/dev/null
cat gpurun_out/r04_t6.log
