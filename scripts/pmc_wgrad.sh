R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for cfg in "conv8s4 4" "deconv8s4 4"; do set -- $cfg
  python3 $R/scripts/bench_conv.py $1 10 2 wgrad $2 2>&1 | tail -1 | cut -c1-120
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmcw_$1_$2 -o p --output-format csv -- python3 $R/scripts/bench_conv.py $1 2 2 wgrad $2 > $R/gpurun_out/pmcw_$1_$2.log 2>&1
  rm -f $R/gpurun_out/pmcw_$1_$2/*trace.csv $R/gpurun_out/pmcw_$1_$2/*agent*
done
