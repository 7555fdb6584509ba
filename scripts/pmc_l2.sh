# L2 hit / miss counts + memory-side read bytes for a few conv micro-benchmarks (is a kernel fed from the XCD's L2 or over the fabric?)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
i=0
for spec in "deconv8s4 fwd" "conv8s4 fwd" "conv8s4 wgrad" "deconv8s4 wgrad" "sft825 wgrad" "c128 fwd" "gemm1x1 fwd"; do
  set -- $spec
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $R/gpurun_out/pmcl2_$i -o p --output-format csv -- python3 $R/scripts/bench_conv.py $1 3 3 $2 4 > $R/gpurun_out/pmcl2_$i.log 2>&1
  rm -f $R/gpurun_out/pmcl2_$i/*trace.csv $R/gpurun_out/pmcl2_$i/*agent*
done
ls $R/gpurun_out/pmcl2_1
