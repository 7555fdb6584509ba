"""conv_hr ablation: time the STAT variant with and without the output store (GPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM, pad8
eng = Engine()
for cin, cout, k in ((32, 32, 3), (49, 49, 3), (32, 49, 1)):
    N, H, W = 4, 1792, 1792
    params = {"l.weight": torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5}
    conv = Conv(eng, "l", params, k, 1, k // 2, 1, bias=False, act=L.ACT_LRELU, slope=0.1)
    x = FM(torch.randn(N, H, W, pad8(cin), device="cuda", dtype=torch.float16), cin)
    y = eng.new(N, H, W, cout)
    gap = eng.f32(N, pad8(cout))
    for name, fn in (("store", lambda: conv.fwd(x, out=y)), ("stat+store", lambda: conv.fwd(x, out=y, stat=gap, stat_mode=L.STAT_SAMPLE_SUM)),
                     ("stat only", lambda: conv.fwd(x, stat=gap, stat_mode=L.STAT_SAMPLE_SUM, store=False))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{cin}->{cout} k{k} {name:12s} {e0.elapsed_time(e1) / 10:.3f} ms")
