#!/bin/bash
# Timing ablations of conv_igemm_glds_kernel<256,4,2,2> on the ResNet 512 -> 512 3x3 layer (variant libraries built with
# -DCSBSR_GLDS_ABLATE=1|2|3: 1 = every DMA source in one L2-resident window, 2 = no LDS fragment reads, 3 = both; results of the variants are garbage)
cd "$(dirname "$0")/.."
for v in "" _g1 _g2; do
  echo "--- lib$v"
  BENCH_NOX3=1 CSBSR_LIB=$PWD/csbsr_amd/libcsbsr_hip$v.so BENCH_NOBIAS=1 python scripts/bench_conv.py res512 10 2 fwd,dgrad 8 2>&1 | grep -v amdgpu.ids
  BENCH_NOX3=1 CSBSR_LIB=$PWD/csbsr_amd/libcsbsr_hip$v.so BENCH_NOBIAS=1 python scripts/bench_conv.py up1024 10 2 fwd 8 2>&1 | grep -v amdgpu.ids
done
