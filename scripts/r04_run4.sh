cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_kernels_gpu.py -q -x 2>&1 | tail -5 > gpurun_out/r04_t4.log
python -m pytest tests/test_determinism_gpu.py tests/test_wc_parity_gpu.py -q -x -k "pspnet or blurskip" 2>&1 | tail -5 >> gpurun_out/r04_t4.log
python scripts/glds_spread_ab.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_t4.log
cat gpurun_out/r04_t4.log
