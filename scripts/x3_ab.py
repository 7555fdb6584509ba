"""A/B of conv_x3_kernel<3>'s item order (cout tile slowest vs the cout tiles of a pixel tile side by side on one XCD) on the SFT shapes:
    python scripts/x3_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from csbsr_amd import _lib as L
import bench_conv as BC
for mode, nm in ((1 | 8, "cout tile slowest (r02)"), (1, "cout tiles of a pixel tile side by side")):
    L.load().csbsr_debug_set_conv_x3(mode)
    print("---", nm)
    BC.run("sft825_384", 4, 448, 448, 825, 384, 3, 1, 1, iters=5, what=("fwd", "dgrad"))
    BC.run("sft384_825", 4, 448, 448, 384, 825, 3, 1, 1, iters=5, what=("fwd",))
    BC.run("sft697_256", 4, 448, 448, 697, 256, 3, 1, 1, iters=5, what=("fwd", "dgrad"))
L.load().csbsr_debug_set_conv_x3(1)
