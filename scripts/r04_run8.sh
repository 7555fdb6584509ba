cd $GRAFT_REPO_ROOT
python -m pytest tests/test_elementwise_gpu.py -q -x -k "subsampled" 2>&1 | tail -3 > gpurun_out/r04_t8.log
for v in 0 1; do
echo "== CSBSR_DC_COMP=$v" >> gpurun_out/r04_t8.log
CSBSR_DC_COMP=$v python -m pytest tests/test_wc2_composed_gpu.py -q -s -k "split" 2>&1 | grep -E "composed step|sr_preds|segment_preds|relative L2|gradients:|KBPN:|passed|failed|Error" >> gpurun_out/r04_t8.log
CSBSR_DC_COMP=$v python -m pytest tests/test_joint_gpu.py -q -s -k "forward_matches_golden" 2>&1 | grep -E "^e2e|passed|failed" | cut -c1-200 >> gpurun_out/r04_t8.log
done
CSBSR_DC_COMP=0 python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --no-other-precision-leg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('dc0', d['value'], d['ms_per_step'])" >> gpurun_out/r04_t8.log
CSBSR_DC_COMP=1 python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --no-other-precision-leg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('dc1', d['value'], d['ms_per_step'])" >> gpurun_out/r04_t8.log
cat gpurun_out/r04_t8.log
