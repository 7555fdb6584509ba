"""A/B: the wide-3x3 kernel (conv_x3<3>) below its 384-input-channel threshold against the LDS-ring kernel, SFT conv0 shapes of stages 1-2:
    python scripts/x3_threshold_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from csbsr_amd import _lib as L
import bench_conv as BC
for mode, nm in ((1, "x3 from 384 channels (default)"), (2, "x3 from 128 channels")):
    L.load().csbsr_debug_set_conv_x3(mode)
    print("---", nm)
    BC.run("sft256_697", 4, 448, 448, 256, 697, 3, 1, 1, iters=5, what=("fwd", "dgrad"))
    BC.run("sft128_569", 4, 448, 448, 128, 569, 3, 1, 1, iters=5, what=("fwd", "dgrad"))
L.load().csbsr_debug_set_conv_x3(1)
