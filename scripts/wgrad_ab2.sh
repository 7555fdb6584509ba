timeout 900 python -m pytest tests/test_conv_kernels_gpu.py -m gpu -x -q -k "test_conv_fwd_dgrad_wgrad" 2>&1 | tail -3
for v in 1; do echo DBG=$v; for sh in sft825_384 sft384_825 res512 res256 up1024 conv8s4 deconv8s4 c128; do CSBSR_WGRAD_DBG=$v python scripts/bench_conv.py $sh 10 3 wgrad 4 2>&1 | tail -1; done; done
python bench.py 2>&1 | tail -1 | cut -c1-200
