cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_kernels_gpu.py -q -x 2>&1 | tail -5 > gpurun_out/r04_t3.log
python -m pytest tests/test_hrnet_gpu.py tests/test_wc_parity_gpu.py -q -x -k "hrnet" 2>&1 | tail -5 >> gpurun_out/r04_t3.log
CSBSR_CONV_GLDS=2050 python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --workload hrnet_x4 --batch 4 > gpurun_out/r04_ab_hr_gk0.json 2>/dev/null
python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --workload hrnet_x4 --batch 4 > gpurun_out/r04_ab_hr_gk1.json 2>/dev/null
cat gpurun_out/r04_t3.log
