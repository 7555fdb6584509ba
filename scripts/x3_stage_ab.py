"""A/B of conv_x3's halo path: LDS-DMA pieces (buffer_load ... lds) vs register-staged pieces (buffer_load -> VGPRs -> ds_write_b128 a few
k-slices later), same process, same shapes.  Needs the library built with -DCSBSR_X3_ABLATE (make CXXFLAGS+=-DCSBSR_X3_ABLATE, or
CSBSR_LIB=<that build>):   python scripts/x3_stage_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from csbsr_amd import _lib as L
import bench_conv as BC
for rep in range(2):
    for mode, nm in ((1, "LDS-DMA halo"), (1 | (32 << 4), "register-staged halo")):
        L.load().csbsr_debug_set_conv_x3(mode)
        print("---", nm)
        BC.run("sft825_384", 4, 448, 448, 825, 384, 3, 1, 1, iters=5, what=("fwd", "dgrad"))
        BC.run("sft384_825", 4, 448, 448, 384, 825, 3, 1, 1, iters=5, what=("fwd",))
        BC.run("conv8s4", 4, 1792, 1792, 128, 128, 8, 4, 2, iters=5, what=("fwd",))
        BC.run("deconv8s4", 4, 448, 448, 128, 128, 8, 4, 2, 1, True, iters=5, what=("dgrad",))
L.load().csbsr_debug_set_conv_x3(1)
