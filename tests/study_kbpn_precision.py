"""Where does the fp16-storage KBPN lose its precision?  (CPU study; test infrastructure, not product: it drives the oracle.)

The oracle (fp32) is run on a contractive fixture with the build's storage plan emulated per layer GROUP: weights of a conv rounded
to fp16 ("W") and / or the tensor a conv reads rounded to fp16 ("X" -- every stored feature map is read by a conv, so this is one
rounding per stored tensor, fused sums included, exactly the build's epilogue stores).  The detector stays fp32 (~ the split mode).
Reported: SR image and segmentation map error against the reference fixture, max |a - b| / max |b|.

    python tests/study_kbpn_precision.py [fixture] > gpurun_out/kbpn_precision_study.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import torch.nn.functional as F

from golden_utils import load_golden, golden_cfg, det_params, max_rel_to_scale, fill_style, rel_err
from oracle import csbsr_oracle as O


class storage_sim:
    """comp: None | "bias" (per-sample channel means x tap sums of the rounding residual, non-transposed layers: what engine.Conv._dc_bias
    does) | "exact" (the residual filter's full response to the per-sample constant map, every layer: the upper bound of the idea)"""

    def __init__(self, names, round_w, round_x, comp=None):
        self.names, self.rw, self.rx, self.comp = names, round_w, round_x, comp

    def __enter__(self):
        self.saved = (F.conv2d, F.conv_transpose2d)
        oc, ot = self.saved
        r16 = lambda t: t.half().float()

        def wrap(fn, transposed):
            def f(x, w, b=None, *a, **k):
                n = self.names.get(id(w))
                if n is None or not n.startswith("sr_model"):
                    return fn(x, w, b, *a, **k)
                xi = r16(x) if self.rx(n) else x
                if not self.rw(n):
                    return fn(xi, w, b, *a, **k)
                w16 = r16(w)
                y = fn(xi, w16, b, *a, **k)
                if self.comp is None or ".kernel_predictor." in n:
                    return y
                m = (xi[:, :, ::8, ::8] if xi.shape[-1] >= 64 else xi).mean(dim=(2, 3))          # per sample, subsampled like the build
                if self.comp == "exact":
                    return y + fn(m[:, :, None, None].expand(-1, -1, xi.shape[2], xi.shape[3]), w - w16, None, *a, **k)
                if transposed:
                    return y
                return y + torch.einsum("nc,oc->no", m, (w - w16).sum((2, 3)))[:, :, None, None]
            return f
        F.conv2d, F.conv_transpose2d = wrap(oc, False), wrap(ot, True)
        return self

    def __exit__(self, *exc):
        F.conv2d, F.conv_transpose2d = self.saved
        return False


GROUPS = [("feat+predictor", lambda n: ".feat." in n or ".predictor." in n), ("up", lambda n: ".up." in n), ("down", lambda n: ".down." in n),
          ("sft", lambda n: ".sft." in n), ("kb.up_conv1", lambda n: ".kb.up_conv1" in n), ("kb.sr_reconst", lambda n: ".kb.sr_reconst" in n),
          ("kernel_predictor", lambda n: ".kernel_predictor." in n), ("output_conv", lambda n: "output_conv" in n)]


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "wc2_pspnet_it40000"
    torch.set_num_threads(8)
    g = load_golden(case)
    cfg = golden_cfg(g)
    P = det_params(scale=int(g["scale"]), detector=str(g["detector"]), requires_grad=False, style=fill_style(g),
                   pixel_shuffle=bool(g["pixel_shuffle"]) if "pixel_shuffle" in g else False)
    names = {id(v): k for k, v in P.items()}
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_wc_parity_gpu import _inputs
    x, hr, mask, k = _inputs(g)
    drop = {kk.split(".", 1)[1]: torch.from_numpy(v) for kk, v in g.items() if kk.startswith("dropmask.")}
    it = int(g["it"])

    def run(rw, rx, comp=None):
        with torch.no_grad(), storage_sim(names, rw, rx, comp):
            sr, kvec = O.kbpn_forward(P, x, it, k, cfg)
            bn = O.BNState(P, True)
            seg, aux = O.pspnet_forward(P, O.norm_sr(sr, cfg), bn, drop, kvec if cfg.detector == "PSPNet_BlurSkip" else None)
        return (max_rel_to_scale(sr, g["sr_preds"]), rel_err(sr, g["sr_preds"]), max_rel_to_scale(seg, g["segment_preds"]),
                rel_err(seg, g["segment_preds"]))

    no, yes = (lambda n: False), (lambda n: True)
    rows = [("fp32 (sanity)", no, no), ("W only", yes, no), ("X only", no, yes), ("W + X (the build's plan)", yes, yes)]
    for gname, pred in GROUPS:
        rows.append((f"W + X, {gname}: W exact", (lambda n, p=pred: not p(n)), yes))
        rows.append((f"W + X, {gname}: X exact", yes, (lambda n, p=pred: not p(n))))
        rows.append((f"W + X, {gname}: both exact", (lambda n, p=pred: not p(n)), (lambda n, p=pred: not p(n))))
    print(f"{case}: error against the reference fixture, max|a-b|/max|b| (rel-L2)")
    for name, rw, rx in rows:
        e = run(rw, rx)
        print(f"{name:44s} sr {e[0]:.2e} ({e[1]:.2e})   seg {e[2]:.2e} ({e[3]:.2e})", flush=True)
    print("-- the weight term's structure: its response to the input's per-sample channel means, given back (engine.Conv._dc_bias)")
    for name, rw, rx, comp in (("W only + bias compensation (the build)", yes, no, "bias"), ("W only + exact mean response (bound)", yes, no, "exact"),
                               ("W + X + bias compensation (the build)", yes, yes, "bias"), ("W + X + exact mean response (bound)", yes, yes, "exact")):
        e = run(rw, rx, comp)
        print(f"{name:44s} sr {e[0]:.2e} ({e[1]:.2e})   seg {e[2]:.2e} ({e[3]:.2e})", flush=True)


if __name__ == "__main__":
    main()
