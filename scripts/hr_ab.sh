# conv_hr A/B on the GPU box: its kernel tests, then forward / dgrad timings of the full-resolution 32 / 49-channel shapes (N = 4)
timeout 900 python -m pytest tests/test_conv_kernels_gpu.py -m gpu -x -q -k "direct or full_res or hr" 2>&1 | tail -3
for sh in hr32 hr49 hr32_49 hr49_32 hr1x1_32_49; do BENCH_NOBIAS=1 python scripts/bench_conv.py $sh 10 3 fwd,dgrad 4 2>&1 | tail -1; done
