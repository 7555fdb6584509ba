"""Same-process A/B of the mirrored weight-gradient problem (engine.Conv._bwd_weights_impl) for layers with <= 64 output channels: ms per
launch and agreement of the two gradients.   python scripts/wgrad_mirror_ab.py   (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM, pad8, grad_acc

SHAPES = {      # N, H, W, cin, cout, k, pad, dil
    "blurskip 505->64 HR": (4, 1792, 1792, 505, 64, 3, 1, 1),
    "up_2 256->64 896^2": (8, 896, 896, 256, 64, 3, 1, 1),
    "output_conv 512->3 HR": (4, 1792, 1792, 512, 3, 3, 1, 1),
    "sr_reconst 128->3 HR": (4, 1792, 1792, 128, 3, 3, 1, 1),
    "fe_cat.0 1x1 128->32 HR": (4, 1792, 1792, 128, 32, 1, 0, 1),
    "aux 256->64 d2 224^2": (8, 224, 224, 256, 64, 3, 2, 2),
    "sr_reconst 384+128->3 HR": (4, 1792, 1792, (384, 128), 3, 3, 1, 1),
}
eng = Engine()
eng._wg_on = False
for name, (N, H, W, cin, cout, k, p, d) in list(SHAPES.items())[:int(os.environ.get("AB_SHAPES", "99"))]:
    segs = cin if isinstance(cin, tuple) else (cin,)
    cin = sum(segs)
    params = {"l.weight": torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5}
    conv = Conv(eng, "l", params, k, 1, p, d, bias=False, split=segs if len(segs) > 1 else None)
    x = tuple(FM(torch.randn(N, H, W, pad8(c), device="cuda", dtype=torch.float16), c) for c in segs)
    for f in x:
        f.t[..., f.c:] = 0
    x = x if len(x) > 1 else x[0]
    dy = FM(torch.randn(N, H, W, pad8(cout), device="cuda", dtype=torch.float16), cout)
    dy.t[..., cout:] = 0
    out = {}
    for mirror in (False, True):
        eng.wgrad_mirror = mirror
        g = grad_acc(conv.w); g.zero_()
        conv.bwd_weights(dy, x); torch.cuda.synchronize()
        ref = g.clone(); kern = int(L.load().csbsr_debug_last_wgrad_kernel())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): conv.bwd_weights(dy, x)
        e1.record(); torch.cuda.synchronize()
        out[mirror] = (e0.elapsed_time(e1) / 5, ref, kern)
    fl = 2.0 * N * H * W * cin * cout * k * k
    a, b = out[False], out[True]
    err = float((a[1] - b[1]).abs().max() / a[1].abs().max())
    print(f"{name:26s} plain {a[0]:7.3f} ms (kernel {a[2]}, {fl / a[0] / 1e9:6.0f} TF/s)   mirrored {b[0]:7.3f} ms (kernel {b[2]}, {fl / b[0] / 1e9:6.0f} TF/s)   max diff {err:.1e}", flush=True)
    del x, dy
    torch.cuda.empty_cache()
