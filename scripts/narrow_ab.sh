#!/bin/bash
# A/B of the 64-cout LDS-DMA tile (conv_igemm_glds_kernel<128,2,2,0>) against the register-staged kernel on the detector's 64-cout layers
cd "$(dirname "$0")/.."
for sh in up3_64 up3_128_64 up2_256_64 l1_64 bs_scale1; do
  for mode in 2 1026; do
    echo "--- $sh glds mode $mode (1026 = 64-cout tile off)"
    python scripts/bench_conv.py $sh 10 $mode fwd,dgrad 4 2>&1 | grep -v amdgpu.ids
  done
done
