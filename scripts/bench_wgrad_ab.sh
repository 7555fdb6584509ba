#!/bin/bash
# A/B of the wgrad kernels on the layer shapes of config 2 (N = 4 = one KBPN micro-batch): register-staged (129) vs LDS-DMA (1)
for s in sft825 conv8s4 deconv8s4 res512 c128 up1024 gemm1x1 res256; do
  for m in 129 1 257 513; do
    echo -n "wgrad_dbg=$m  "; CSBSR_WGRAD_DBG=$m python scripts/bench_conv.py $s 10 2 wgrad 4 | tail -1
  done
done
