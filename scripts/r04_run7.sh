cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
CSBSR_WGRAD_STREAM=$v python bench.py --steps 3 --batch 4 --no-cpu-baseline --no-h2d-leg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('wgs$v batch4', d['value'], d['ms_per_step'], d['other_precision']['value'], d['peak_mem_gb'])
"
done
