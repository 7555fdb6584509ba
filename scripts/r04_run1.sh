set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_kernels_gpu.py -q -k "split_precision" -x -s 2>&1 | grep -E "split conv|passed|failed|Error|error" | tail -30 > gpurun_out/r04_t1.log
python -m pytest tests/test_wc_parity_gpu.py tests/test_wc2_composed_gpu.py -q -s 2>&1 | grep -E "detector on the reference|gradients|composed|passed|failed|Error|assert|seg " | tail -60 >> gpurun_out/r04_t1.log
python -m pytest tests/test_full_size_gpu.py tests/test_joint_gpu.py -q -s -k "full_size or zero_pad" 2>&1 | tail -40 >> gpurun_out/r04_t1.log
for f in 0 1; do
CSBSR_SPLIT_FUSED=$f python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg > gpurun_out/r04_ab_fused$f.json 2> gpurun_out/r04_ab_fused$f.err
CSBSR_SPLIT_FUSED=$f python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --workload blurskip_x8 --lr-size 224 --batch 4 > gpurun_out/r04_ab_bs_fused$f.json 2>> gpurun_out/r04_ab_fused$f.err
done
tail -5 gpurun_out/r04_t1.log
