import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, FM, _ptr
eng = Engine()
x = FM(torch.randn(1, 1792, 1792, 128, device="cuda", dtype=torch.float16), 128)
y = eng.new(1, 1792, 1792, 128)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nb = x.t.numel() * 2
ms = t(lambda: L.call("csbsr_axpby", x.npix, x.cp, _ptr(x.t), x.ld, 2.0, None, 0, 0.0, _ptr(y.t), y.ld, eng.stream))
print(f"axpby copy: {ms:.3f} ms  {2*nb/ms/1e6:.0f} GB/s (r+w)")
ms = t(lambda: L.call("csbsr_fill_f16", _ptr(y.t), y.npix, y.cp, y.ld, 1.0, eng.stream))
print(f"fill: {ms:.3f} ms  {nb/ms/1e6:.0f} GB/s (w)")
ms = t(lambda: y.t.copy_(x.t))
print(f"torch copy: {ms:.3f} ms  {2*nb/ms/1e6:.0f} GB/s (r+w)")
ms = t(lambda: eng.epilogue_bwd(x, out=y, act=L.ACT_LRELU, slope=0.1, dpre=x))
print(f"epilogue_bwd (r dout, r out, w dpre): {ms:.3f} ms  {3*nb/ms/1e6:.0f} GB/s")
