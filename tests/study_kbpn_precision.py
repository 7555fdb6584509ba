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


def tapsum_round(w, transposed=False, stride=1):
    """fp16 rounding of a filter bank that keeps the TAP SUM of every (output, input) channel pair: round to nearest, then move the taps
    whose value sat closest to a rounding midpoint to their other fp16 neighbour until sum_taps (w - w16) is within half an ulp.  The
    rounding residual then has no response to any input that is constant over the filter's footprint (not only to the mean, and at no
    run-time cost).  Transposed layers: per output PHASE (the taps ky % stride, kx % stride an output pixel of that phase sees).
    Returns fp32 values that are exactly representable in fp16."""
    w16 = w.half().float()
    K0, K1 = w.shape[2], w.shape[3]
    if transposed and stride > 1:
        ph = (torch.arange(K0)[:, None] % stride) * stride + (torch.arange(K1)[None, :] % stride)
        groups = [(ph == g).reshape(-1) for g in range(stride * stride)]
    else:
        groups = [torch.ones(K0 * K1, dtype=torch.bool)]
    W = w.reshape(w.shape[0], w.shape[1], -1).clone()
    Q = w16.reshape(w.shape[0], w.shape[1], -1).clone()
    for gm in groups:
        wv, q = W[:, :, gm], Q[:, :, gm].clone()
        def spacing(q, sgn):        # distance to the fp16 neighbour in direction sgn: below a power of two the grid is twice as fine
            a = q.abs().clamp_min(2.0 ** -14)
            e = torch.floor(torch.log2(a))
            inward = (q * sgn < 0) & (a == torch.exp2(e)) & (a > 2.0 ** -14)
            return torch.exp2(e - 10 - inward.float())
        for _ in range(wv.shape[-1]):
            d = wv - q
            r = d.sum(-1, keepdim=True)
            sgn = torch.sign(r)
            ulp = spacing(q, sgn)
            gain = r.abs() - (r - sgn * ulp).abs()                    # how much |r| shrinks if this tap moves one ulp towards r
            cand = (torch.sign(d) == sgn) & (gain > 0)
            score = torch.where(cand, d.abs() / ulp, torch.full_like(d, -1.0))     # closest to its midpoint first: least extra error
            best = score.argmax(-1, keepdim=True)
            ok = score.gather(-1, best) > 0
            if not bool(ok.any()):
                break
            step = torch.zeros_like(q).scatter_(-1, best, (sgn * ulp.gather(-1, best)) * ok)
            q = (q + step).half().float()
        Q[:, :, gm] = q
    return Q.reshape(w.shape)


def r16_diffuse(x, block):
    """fp16 rounding of a stored map with the rounding error carried DOWN the rows of a ``block``-row group (noise shaping: the group's
    column sums are rounded once, not ``block`` times -- the error moves to vertical frequencies a low-pass consumer damps); block 0 = the
    whole column"""
    N, C, H, W = x.shape
    q = torch.empty_like(x)
    carry = torch.zeros(N, C, W, dtype=x.dtype)
    for r in range(H):
        if block and r % block == 0:
            carry.zero_()
        t = x[:, :, r, :] + carry
        q[:, :, r, :] = t.half().float()
        carry = t - q[:, :, r, :]
    return q


class storage_sim:
    """comp: None | "bias" (per-sample channel means x tap sums of the rounding residual, non-transposed layers: what engine.Conv._dc_bias
    does) | "exact" (the residual filter's full response to the per-sample constant map, every layer: the upper bound of the idea)"""

    def __init__(self, names, round_w, round_x, comp=None, diffuse=None):
        self.names, self.rw, self.rx, self.comp, self.diffuse = names, round_w, round_x, comp, diffuse
        self.cache = {}

    def __enter__(self):
        self.saved = (F.conv2d, F.conv_transpose2d)
        oc, ot = self.saved
        r16 = lambda t: t.half().float()

        def wrap(fn, transposed):
            def f(x, w, b=None, *a, **k):
                n = self.names.get(id(w))
                if n is None or not n.startswith("sr_model"):
                    return fn(x, w, b, *a, **k)
                if self.rx(n):
                    if self.diffuse is None or x.shape[2] < 8:
                        xi = r16(x)
                    else:
                        key = ("x", id(x))
                        if key not in self.cache:
                            self.cache[key] = (x, r16_diffuse(x, self.diffuse))      # (the tensor is kept alive with its rounded copy: ids stay unique)
                        xi = self.cache[key][1]
                else:
                    xi = x
                if not self.rw(n):
                    return fn(xi, w, b, *a, **k)
                if self.comp in ("tapsum", "tapsum+bias") and ".kernel_predictor." not in n:
                    key = (id(w), transposed)
                    if key not in self.cache:
                        st = (a[0] if a else k.get("stride", 1))
                        st = st[0] if isinstance(st, (tuple, list)) else st
                        self.cache[key] = tapsum_round(w, transposed, int(st))
                    w16 = self.cache[key]
                else:
                    w16 = r16(w)
                y = fn(xi, w16, b, *a, **k)
                if self.comp in (None, "tapsum") or ".kernel_predictor." in n:
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

    def run(rw, rx, comp=None, diffuse=None):
        with torch.no_grad(), storage_sim(names, rw, rx, comp, diffuse):
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
    if os.environ.get("STUDY_QUICK"):
        rows = rows[:4]
    for name, rw, rx in rows:
        e = run(rw, rx)
        print(f"{name:44s} sr {e[0]:.2e} ({e[1]:.2e})   seg {e[2]:.2e} ({e[3]:.2e})", flush=True)
    print("-- the weight term's structure: its response to the input's per-sample channel means, given back (engine.Conv._dc_bias)")
    for name, rw, rx, comp in (("W only + bias compensation (the build)", yes, no, "bias"), ("W only + exact mean response (bound)", yes, no, "exact"),
                               ("W + X + bias compensation (the build)", yes, yes, "bias"), ("W + X + exact mean response (bound)", yes, yes, "exact"),
                               ("W only, tap-sum-preserving rounding", yes, no, "tapsum"), ("W only, tap-sum rounding + bias comp.", yes, no, "tapsum+bias"),
                               ("W + X, tap-sum-preserving rounding", yes, yes, "tapsum"), ("W + X, tap-sum rounding + bias comp.", yes, yes, "tapsum+bias")):
        e = run(rw, rx, comp)
        print(f"{name:44s} sr {e[0]:.2e} ({e[1]:.2e})   seg {e[2]:.2e} ({e[3]:.2e})", flush=True)


    if os.environ.get("STUDY_DIFFUSE"):
        print("-- activation storage with error diffusion down groups of rows (r16_diffuse)")
        for blk in (4, 8, 0):
            e = run(no, yes, None, blk)
            print(f"{'X only, rows diffused in groups of ' + (str(blk) if blk else 'H'):44s} sr {e[0]:.2e} ({e[1]:.2e})   seg {e[2]:.2e} ({e[3]:.2e})", flush=True)
        for blk in (4, 0):
            e = run(yes, yes, "tapsum+bias", blk)
            print(f"{'W (tap-sum + bias) + X diffused, groups of ' + (str(blk) if blk else 'H'):44s} sr {e[0]:.2e} ({e[1]:.2e})   seg {e[2]:.2e} ({e[3]:.2e})", flush=True)


if __name__ == "__main__":
    main()
