"""dgrad of the PSPNet stem (Conv2d 3 -> 64, 7x7 stride 2) at B = 8, HR 1792: conv_thin_tpd against the general kernel (debug bit 13).  (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM
eng = Engine()
params = {"l.weight": torch.randn(64, 3, 7, 7, device="cuda") / 12.0}
conv = Conv(eng, "l", params, 7, 2, 3, 1, bias=False)
dy = FM(torch.randn(8, 896, 896, 64, device="cuda", dtype=torch.float16), 64)
outs = []
for mode in (2 | 8192, 2):
    L.load().csbsr_debug_set_conv_glds(mode)
    dx = conv.bwd_input(dy, in_hw=(1792, 1792)); torch.cuda.synchronize()
    kern = int(L.load().csbsr_debug_last_conv_kernel())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): dx = conv.bwd_input(dy, in_hw=(1792, 1792))
    e1.record(); torch.cuda.synchronize()
    print(f"mode {mode}: kernel {kern}  {e0.elapsed_time(e1) / 5:.3f} ms")
    outs.append(dx.t[..., :3].float().clone())
print("max |diff| / max", float((outs[0] - outs[1]).abs().max() / outs[0].abs().max()))
