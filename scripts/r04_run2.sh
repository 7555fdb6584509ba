cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_kernels_gpu.py -q -x 2>&1 | tail -3 > gpurun_out/r04_t2.log
python scripts/study_split_plan.py --combos wc_pspnet_it40000 wc_blurskip_x8_it40000 2>&1 | grep -v amdgpu >> gpurun_out/r04_t2.log
for v in spread0 default; do
if [ $v = default ]; then unset CSBSR_LIB; else export CSBSR_LIB=$GRAFT_REPO_ROOT/csbsr_amd/libcsbsr_hip_$v.so; fi
python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg > gpurun_out/r04_ab_$v.json 2>/dev/null
python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --workload blurskip_x8 --lr-size 224 --batch 4 > gpurun_out/r04_ab_bs_$v.json 2>/dev/null
done
cat gpurun_out/r04_t2.log
