"""Micro-benchmark: conv forward with / without fused BatchNorm statistics (atomics contention check)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM, pad8
eng = Engine()
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for (N, H, W, cin, cout, k, s, p) in ((8, 1792, 1792, 3, 64, 7, 2, 3), (8, 448, 448, 64, 64, 3, 1, 1), (8, 224, 224, 256, 256, 3, 1, 1), (8, 224, 224, 512, 512, 3, 1, 1), (8, 448, 448, 1024, 256, 3, 1, 1)):
    params = {"l.weight": torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5}
    conv = Conv(eng, "l", params, k, s, p, 1, bias=False)
    x = FM(torch.randn(N, H, W, pad8(cin), device="cuda", dtype=torch.float16), cin)
    OH, OW = conv.out_size(H, W)
    y = eng.new(N, OH, OW, cout)
    stat = eng.f32(2, pad8(cout))
    a = t(lambda: conv.fwd(x, out=y, stat=stat, stat_mode=L.STAT_BN))
    b = t(lambda: conv.fwd(x, out=y))
    print(f"[{N},{H},{W}] {cin}->{cout} k{k}s{s}: with BN stats {a:8.1f} us   without {b:8.1f} us")
