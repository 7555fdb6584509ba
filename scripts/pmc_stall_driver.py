"""A few launches of each MFMA kernel family at the bench's shapes (N = 4), for the stall-counter passes of scripts/pmc_stall.sh:
conv_x3<3> (SFT 825 -> 384), conv_x3<2> (8x8 s4 conv), conv_tp (8x8 s4 deconv), conv_wgrad_glds<256,256> (SFT wgrad) and <128,256> (8x8 s4 wgrad),
conv_igemm_glds<256,4,2,2> (ResNet 512)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench_conv as BC

S = {"sft825_384": ((4, 448, 448, 825, 384, 3, 1, 1), ("fwd", "dgrad", "wgrad")),
     "conv8s4": ((4, 1792, 1792, 128, 128, 8, 4, 2), ("fwd", "wgrad")),
     "deconv8s4": ((4, 448, 448, 128, 128, 8, 4, 2, 1, True), ("fwd", "dgrad")),
     "res512": ((8, 224, 224, 512, 512, 3, 1, 1), ("fwd", "wgrad"))}
for name, (shape, what) in S.items():
    BC.run(name, *shape, iters=3, what=what)
