# kernel-stats profile of one bench step set in each detector precision mode (gpurun): where the split mode's extra time goes
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
F="--steps 2 --warmup 1 --no-cpu-baseline --no-h2d-leg --no-other-precision-leg --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_prof_split -o b --output-format csv -- python3 $R/bench.py --detector-precision split $F > $R/gpurun_out/r03_prof_split.json 2> $R/gpurun_out/r03_prof_split.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_prof_fp16 -o b --output-format csv -- python3 $R/bench.py --detector-precision fp16 $F > $R/gpurun_out/r03_prof_fp16.json 2> $R/gpurun_out/r03_prof_fp16.err
rm -f $R/gpurun_out/r03_prof_*/*kernel_trace.csv $R/gpurun_out/r03_prof_*/*agent_info.csv
