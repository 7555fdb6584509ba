"""Timing ablations of conv_x3_kernel<3> (library built with -DCSBSR_X3_ABLATE; results of the ablated variants are garbage):
which of halo DMA / weight stream / LDS fragment reads the K loop waits for.   python scripts/x3_ablate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from csbsr_amd import _lib as L
import bench_conv as BC
names = {0: "baseline", 1: "no halo DMA traffic", 2: "weights of step 0 every step", 4: "no LDS fragment reads", 8: "no weight loads",
         16: "halo DMA from an L2-resident region", 24: "L2-resident halo + no weight loads", 13: "MFMA + issue only"}
for abl, nm in names.items():
    L.load().csbsr_debug_set_conv_x3(1 | (abl << 4))
    print(f"--- ABL {abl}: {nm}")
    BC.run("sft825_384", 4, 448, 448, 825, 384, 3, 1, 1, iters=5, what=("fwd",))
L.load().csbsr_debug_set_conv_x3(1)
