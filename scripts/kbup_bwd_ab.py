"""Backward of kb.up_conv1 at N = 4, LR 448 -> HR 1792: the fused pass (csrc/conv_kbup.hip) against the epilogue-backward pass + weight-gradient
launch it replaces.   python scripts/kbup_bwd_ab.py   (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM, grad_acc
eng = Engine()
eng._wg_on = False
N, h, w, C = 4, 448, 448, 128
params = {"l.weight": torch.randn(3, C, 8, 8, device="cuda") / 14.0, "l.a": torch.full((1,), 0.25, device="cuda")}
conv = Conv(eng, "l", params, 8, 4, 2, 1, transposed=True, bias=False, act=L.ACT_PRELU, prelu="l.a")
conv.frozen = False
x = FM(torch.randn(N, h, w, 8, device="cuda", dtype=torch.float16), 3); x.t[..., 3:] = 0
res = FM(torch.randn(N, 4 * h, 4 * w, C, device="cuda", dtype=torch.float16), C)
dout = FM(torch.randn(N, 4 * h, 4 * w, C, device="cuda", dtype=torch.float16) / 1024, C)
out = conv.fwd(x, res=res, res_mode=L.RES_ADD)
dpk = eng.new(N, 4 * h, 4 * w, C)

def old():
    eng.epilogue_bwd(dout, out=out, act=conv.act, slope=conv.slope, prelu=conv.prelu, res=res, res_mode=L.RES_ADD, dpre=dpk, dprelu=grad_acc(conv.prelu), creal=C)
    conv.bwd_weights(dpk, x)

def new():
    conv.bwd_thin_tp_fused(dout, x, dpk)

res_ = {}
for name, fn in (("pass + wgrad", old), ("fused", new)):
    grad_acc(conv.w).zero_(); grad_acc(conv.prelu).zero_()
    fn(); torch.cuda.synchronize()
    res_[name] = (dpk.t.clone(), conv.w.gacc.clone(), conv.prelu.gacc.clone())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:14s} {e0.elapsed_time(e1) / 5:.3f} ms")
a, b = res_["pass + wgrad"], res_["fused"]
rm = lambda u, v: float((u.float() - v.float()).abs().max() / v.float().abs().max())
print("dPre", rm(a[0], b[0]), "dW", rm(a[1], b[1]), "da", float(a[2]), float(b[2]))
