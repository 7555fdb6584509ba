"""Which stored tensor carries KBPN's gradient error?  (CPU study; test infrastructure, not product: it drives the oracle.)

tests/test_wc_parity_gpu.py::test_kbpn_backward_on_reference_gradient feeds the HIP KBPN backward the reference's own dLoss/d(SR, kernel) of a
joint step (fixture wc_pspnet_it40000: the noise-like upstream gradient of a random-weight detector) and measures the KBPN parameter
gradients at median 3.6e-2 / p90 7.4e-2 / max 0.16 against the reference.  Round 4's forward study attributed the FORWARD error per
tensor kind; this is the same for the backward.  The fp32 oracle's autograd is run with the build's storage plan emulated one group at a
time:

  forward   W   conv weights rounded to fp16                         (changes the operands of dgrad, and gates through the activations)
            X   every tensor a conv reads rounded to fp16            (= one rounding per stored feature map; gates and wgrad operands)
  backward  dPre  the gradient wrt every conv's output (the operand the dgrad / wgrad kernels read) rounded to fp16 under the build's
                  power-of-two loss scale
            dX    the gradient wrt every conv's input (what a dgrad kernel stores) rounded to fp16, same scale
  each split by resolution (HR: the 4x / 8x maps; LR) -- LR maps are 1/16 of the bytes, so fp32 there would be nearly free.

Reported: relative L2 error of every KBPN parameter gradient against the exact fp32 run on the same upstream gradient (median / p90 / max
over the tensors), and the exact run against the reference's own stored gradients (the fixture's 32 samples per tensor).

    python tests/study_kbpn_backward.py [fixture] > profiles/r05_kbpn_backward_study.txt
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import torch.nn.functional as F

from golden_utils import load_golden, golden_cfg, det_params, fill_style
from oracle import csbsr_oracle as O


class plan:
    """fw / fx: round conv weights / conv inputs to fp16 in the forward (predicates on the layer name and 'HR' / 'LR');
    gpre / gx: round the gradient wrt conv outputs / conv inputs to fp16 under the loss scale ``gs``"""

    def __init__(self, names, lr_size, gs, fw=None, fx=None, gpre=None, gx=None):
        self.names, self.lr, self.gs = names, lr_size, gs
        no = lambda n, res: False
        self.fw, self.fx, self.gpre, self.gx = fw or no, fx or no, gpre or no, gx or no

    def __enter__(self):
        self.saved = (F.conv2d, F.conv_transpose2d)
        r16 = lambda t: t.half().float()
        gs = self.gs

        def rgrad(g):
            return (g * gs).half().float() / gs

        def wrap(fn):
            def f(x, w, b=None, *a, **k):
                n = self.names.get(id(w))
                if n is None or not n.startswith("sr_model") or k.get("groups", 1) != 1:
                    return fn(x, w, b, *a, **k)
                res_in = "HR" if x.shape[-1] > self.lr else "LR"
                xi = x
                if self.fx(n, res_in):
                    xi = x + (r16(x) - x).detach()                  # straight-through: the forward value is the rounded one
                if self.gx(n, res_in) and xi.requires_grad:
                    xi = xi * 1.0
                    xi.register_hook(rgrad)
                wi = w + (r16(w) - w).detach() if self.fw(n, res_in) else w
                y = fn(xi, wi, b, *a, **k)
                res_out = "HR" if y.shape[-1] > self.lr else "LR"
                if self.gpre(n, res_out) and y.requires_grad:
                    y.register_hook(rgrad)
                return y
            return f
        F.conv2d, F.conv_transpose2d = wrap(self.saved[0]), wrap(self.saved[1])
        return self

    def __exit__(self, *exc):
        F.conv2d, F.conv_transpose2d = self.saved
        return False


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "wc_pspnet_it40000"
    torch.set_num_threads(8)
    g = load_golden(case)
    cfg = golden_cfg(g)
    from test_wc_parity_gpu import _inputs
    x, hr, mask, k = _inputs(g)
    it = int(g["it"])
    dsr = torch.from_numpy(g["dsr16"].astype(np.float32)) / float(g["dsr_scale"])
    dkvec = torch.from_numpy(g["dkvec"])
    B, _, h, w = x.shape
    H = h * int(g["scale"])
    gs = float(2 ** round(math.log2(B * H * H)))          # the build's loss scale (modeling/build_model.py::_hip_backward)

    def grads(**kw):
        P = det_params(scale=int(g["scale"]), detector=str(g["detector"]), requires_grad=True, style=fill_style(g))
        names = {id(v): n for n, v in P.items()}
        with plan(names, w, gs, **kw):
            sr, kvec = O.kbpn_forward(P, x, it, k, cfg)
            ((sr * dsr).sum() + (kvec.reshape(dkvec.shape) * dkvec).sum()).backward()
        return {n: v.grad.detach().clone() for n, v in P.items() if n.startswith("sr_model") and v.grad is not None}

    def dist(ga, gb):
        v = np.array([float((ga[n] - gb[n]).norm() / (gb[n].norm() + 1e-30)) for n in gb if gb[n].numel() > 1 and float(gb[n].norm()) > 0])
        return np.median(v), np.percentile(v, 90), v.max()

    exact = grads()
    # the exact oracle against the reference's stored samples (what the fixture itself can resolve)
    import zlib
    refn = dict(zip((str(v) for v in g["grad_names"]), zip((float(v) for v in g["grad_norms"]), g["grad_samples32"])))
    es = []
    for n, gr in exact.items():
        if n not in refn or refn[n][0] <= 1e-9 or gr.numel() <= 1:
            continue
        flat = gr.reshape(-1)
        idx = [(zlib.crc32((n + str(j)).encode()) % flat.numel()) for j in range(32)]
        ref = torch.from_numpy(np.asarray(refn[n][1], dtype=np.float32))
        es.append(float((flat[idx] - ref).norm() / (ref.norm() + 1e-30)))
    print(f"{case}: KBPN parameter gradients from the reference's upstream gradient; loss scale 2^{int(math.log2(gs))}")
    print(f"exact fp32 oracle vs the reference's 32 stored samples per tensor: median {np.median(es):.2e} p90 {np.percentile(es, 90):.2e} max {np.max(es):.2e} ({len(es)} tensors)")
    yes = lambda n, res: True
    HRo, LRo = (lambda n, res: res == "HR"), (lambda n, res: res == "LR")
    rows = [("forward W (weights fp16)", dict(fw=yes)),
            ("forward X (stored activations fp16)", dict(fx=yes)),
            ("forward X, HR maps only", dict(fx=HRo)),
            ("forward X, LR maps only", dict(fx=LRo)),
            ("forward W + X", dict(fw=yes, fx=yes)),
            ("backward dPre (HR + LR)", dict(gpre=yes)),
            ("backward dPre, HR only", dict(gpre=HRo)),
            ("backward dPre, LR only", dict(gpre=LRo)),
            ("backward dX (HR + LR)", dict(gx=yes)),
            ("backward dX, HR only", dict(gx=HRo)),
            ("backward dX, LR only", dict(gx=LRo)),
            ("backward dPre + dX", dict(gpre=yes, gx=yes)),
            ("all four (the build's plan)", dict(fw=yes, fx=yes, gpre=yes, gx=yes)),
            ("all four, LR gradient maps fp32", dict(fw=yes, fx=yes, gpre=HRo, gx=HRo)),
            ("all four, LR maps fp32 forward and backward", dict(fw=yes, fx=HRo, gpre=HRo, gx=HRo))]
    print(f"{'rounded (everything else fp32)':48s} rel-L2 vs the exact run: median / p90 / max over the tensors")
    for name, kw in rows:
        m, p, mx = dist(grads(**kw), exact)
        print(f"{name:48s} {m:.2e} / {p:.2e} / {mx:.2e}", flush=True)


if __name__ == "__main__":
    main()
