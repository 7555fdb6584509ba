"""A/B of the 128 -> 3 strided conv (dgrad of kb.up_conv1) on its streaming kernel vs the general 32-cout tile, N = 4, LR 448:
    python scripts/sc_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM
eng = Engine()
N = 4
conv = Conv(eng, "l", {"l.weight": torch.randn(3, 128, 8, 8, device="cuda") * 0.05}, 8, 4, 2, 1, transposed=True, bias=False)
dpk = FM(torch.randn(N, 1792, 1792, 128, device="cuda", dtype=torch.float16), 128)
derr = torch.empty(N, 3, 448, 448, device="cuda")
for mode, nm in ((2, "streaming kernel"), (2 | 4096, "general kernel")):
    L.load().csbsr_debug_set_conv_glds(mode)
    f = lambda: conv.bwd_input(dpk, out32=derr, in_hw=(448, 448))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{nm}: {ms:.3f} ms  ({dpk.t.numel() * 2 / ms / 1e9:.2f} TB/s of input)  kernel id {L.load().csbsr_debug_last_conv_kernel()}")
L.load().csbsr_debug_set_conv_glds(2)
