cd $GRAFT_REPO_ROOT
for v in 0 1 2; do
echo "== CSBSR_DC_COMP=$v"
CSBSR_DC_COMP=$v python -m pytest tests/test_wc2_composed_gpu.py -q -s -k "split and (pixelshuffle or pspnet_it)" 2>&1 | grep -E "composed step|sr_preds|segment_preds|relative L2|passed|failed|Error" | cut -c1-110
done > gpurun_out/r04_t9.log 2>&1
cat gpurun_out/r04_t9.log
