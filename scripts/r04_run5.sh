cd $GRAFT_REPO_ROOT
for rep in 0 1; do
for lib in libcsbsr_hip.so libcsbsr_hip_wgs0.so; do
  echo "== $lib pass $rep"
  for s in sft825_384 sft384_825 conv8s4 deconv8s4 res512 up1024 res256; do
    CSBSR_LIB=$GRAFT_REPO_ROOT/csbsr_amd/$lib python scripts/bench_conv.py $s 10 2 wgrad 4 2>/dev/null | tail -1
  done
done
done > gpurun_out/r04_wgs_ab.log 2>&1
python -m pytest tests/test_conv_kernels_gpu.py -q -x 2>&1 | tail -3 >> gpurun_out/r04_wgs_ab.log
cat gpurun_out/r04_wgs_ab.log
