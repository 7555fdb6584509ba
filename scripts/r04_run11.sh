cd $GRAFT_REPO_ROOT
python -m pytest tests/test_determinism_gpu.py -q -s -k "plan_hooks or wgrad_side" 2>&1 | grep -E "split forward|passed|failed|Error" > gpurun_out/r04_t11.log
python -m pytest tests/test_wc2_composed_gpu.py -q 2>&1 | tail -2 >> gpurun_out/r04_t11.log
for v in 0 1 0 1; do
CSBSR_DC_COMP=$v python bench.py --steps 3 --no-cpu-baseline --no-h2d-leg --no-other-precision-leg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('dc$v', d['value'], d['ms_per_step'])" >> gpurun_out/r04_t11.log
done
cat gpurun_out/r04_t11.log
