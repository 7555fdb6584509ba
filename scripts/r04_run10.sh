cd $GRAFT_REPO_ROOT
for v in 0 1; do
echo "== CSBSR_DC_COMP=$v"
CSBSR_DC_COMP=$v python scripts/study_split_plan.py --combos wc_pspnet_it40000 wc_blurskip_x8_it40000 2>&1 | grep -v amdgpu
done > gpurun_out/r04_t10.log 2>&1
cat gpurun_out/r04_t10.log
