#!/bin/bash
# Run on the GPU box (gpurun): bench line, rocprofv3 kernel stats of the same command, the two PMC passes for HBM traffic and the PMC
# pass for MFMA utilisation.  Outputs land in gpurun_out/$TAG/ (TAG = first argument, default r02); scripts/summarise_profiles.py turns
# them into the committed files under profiles/.  Counters are collected in their own runs (kernel trace only), one counter set per run.
set -u
TAG=${1:-r06}
EXTRA=${2:-}          # extra bench.py arguments for every pass, e.g. "--workload hrnet_x4 --batch 4" (configs 4 / 5)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py $EXTRA --dump-layers $O/layers.json > $O/bench.json 2> $O/bench.err
# the stats pass runs EXACTLY the steps the JSON reports (no H2D leg, no other-precision leg): per-step launch counts = calls / (steps + warmup)
timeout -s KILL 900 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py $EXTRA --no-cpu-baseline --no-h2d-leg --no-other-precision-leg --no-kernel-timing > $O/bench_under_rocprof.json 2> $O/stats.err
PM="$EXTRA --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-h2d-leg --no-other-precision-leg"
timeout -s KILL 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- python3 $R/bench.py $PM > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -s KILL 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- python3 $R/bench.py $PM > $O/pmc_write.json 2> $O/pmc_write.err
timeout -s KILL 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA -d $O/pmc_mfma -o mfma --output-format csv -- python3 $R/bench.py $PM > $O/pmc_mfma.json 2> $O/pmc_mfma.err
# L2 hit rate next to FETCH_SIZE (which also counts Infinity-Cache hits: MI355X_MICROARCH.md, HBM section)
timeout -s KILL 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/pmc_l2 -o l2 --output-format csv -- python3 $R/bench.py $PM > $O/pmc_l2.json 2> $O/pmc_l2.err
rm -f $O/*/*kernel_trace.csv $O/*/*agent_info.csv 2>/dev/null     # large; the stats / counter files are what is summarised
ls -la $O $O/* | head -40
tail -c 600 $O/bench.json
