"""A/B of the streaming 3-channel-input kernel on the accumulating dgrads of kb.sr_reconst / output_conv (GPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM, pad8
eng = Engine()
lib = L.load()
N, H, W = 4, 1792, 1792
for cout, acc in ((128, True), (384, True), (512, False), (49, False)):
    w = torch.randn(3, cout, 3, 3, device="cuda") / (cout * 9) ** 0.5
    conv = Conv(eng, "l", {"l.weight": w}, 3, 1, 1, 1, bias=False, act=L.ACT_NONE)
    dpre = FM(torch.randn(N, H, W, 8, device="cuda", dtype=torch.float16), 3)
    dx = eng.new(N, H, W, cout)
    for mode in (2, 2 | 512):
        lib.csbsr_debug_set_conv_glds(mode)
        fn = lambda: conv.bwd_input(dpre, out=dx, accumulate=acc)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = N * H * W * pad8(cout) * 2 * (2 if acc else 1) / 1e9
        print(f"3->{cout} acc={acc} kernel {lib.csbsr_debug_last_conv_kernel()}: {ms:.3f} ms  {gb / ms:.2f} TB/s")
    lib.csbsr_debug_set_conv_glds(2)
