"""Would Winograd F(2x2, 3x3) on the SFT 3x3 stride-1 layers keep KBPN inside its forward tolerance?  (CPU study; test infrastructure,
not product: it drives the oracle.  Round-5 review item 3: "study first, on the CPU; stop if SR > 9e-4".)

The SFT layers (/root/reference/model/modeling/kbpn.py:493-518) are the largest block of MFMA work in the step; Winograd F(2x2,3x3)
would run their forward / dgrad products with 16 instead of 36 multiplies per 2x2 output tile.  On the matrix pipe the transformed
operands must be fp16: here the oracle (fp32) runs with the build's storage plan (every conv's weights rounded to fp16
tap-sum-preservingly, every stored map rounded to fp16 -- tests/study_kbpn_precision.py) and, on the chosen SFT layers, the product
computed the Winograd way: input transform B^T d B and weight transform G g G^T in fp32 FROM the fp16-stored operands, both results
ROUNDED TO fp16, the 16 per-position channel contractions accumulated in fp32, output transform A^T m A in fp32.  The 441 kernel-code
channels (spatially constant; the build folds them into an exact fp32 per-border-class bias) stay a direct fp32 product.

    python tests/study_winograd.py [fixture] > profiles/r06_winograd_study.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F

from golden_utils import load_golden, golden_cfg, det_params, max_rel_to_scale, fill_style, rel_err
from oracle import csbsr_oracle as O
from study_kbpn_precision import tapsum_round

BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
r16 = lambda t: t.half().float()


def winograd_conv3x3(x, w, round_v=True, round_u=True, v_scale=1.0):
    """3x3 stride-1 pad-1 cross-correlation of x [N,C,H,W] (H, W even) with w [O,C,3,3] as F(2x2,3x3); fp32 everywhere except the two
    transformed operands, rounded to fp16 when asked.  ``v_scale``: the input transform's results divided by this before rounding (and
    the product multiplied back) -- |B^T d B| reaches 4 max|d|, the scale only matters for overflow, not precision, in floating point."""
    N, C, H, W = x.shape
    O_ = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    d = F.unfold(xp, kernel_size=4, stride=2).reshape(N, C, 4, 4, -1)              # [N, C, 4, 4, T]
    V = torch.einsum("ai,ncijt,bj->ncabt", BT, d, BT)
    U = torch.einsum("ai,ocij,bj->ocab", G, w, G)
    if round_v:
        V = r16(V / v_scale) * v_scale
    if round_u:
        U = r16(U)
    M = torch.einsum("ocab,ncabt->noabt", U, V)
    Y = torch.einsum("ia,noabt,jb->noijt", AT, M, AT)                             # [N, O, 2, 2, T]
    return F.fold(Y.reshape(N, O_ * 4, -1), output_size=(H, W), kernel_size=2, stride=2)


# 1-D forms along x (the rows stay a direct 3-tap sum in the transformed domain): F(2,3) and F(4,3)
W1D = {
    "F(2,3)": (BT, G, AT),
    "F(4,3)": (torch.tensor([[4., 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]),
               torch.tensor([[1 / 4., 0, 0], [-1 / 6., -1 / 6., -1 / 6.], [-1 / 6., 1 / 6., -1 / 6.], [1 / 24., 1 / 12., 1 / 6.], [1 / 24., -1 / 12., 1 / 6.], [0, 0, 1]]),
               torch.tensor([[1., 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]])),
}


def winograd_conv3x3_1d(x, w, form="F(2,3)", round_v=True, round_u=True):
    """the same product with the Winograd transform along x only: out[y, m t + i] = A^T sum_ky (U[ky] . V[y + ky - 1]) -- m outputs per
    m + 2 inputs, the row taps accumulate in the transformed domain (what an MFMA kernel with one accumulator set per position does)."""
    bt, g, at = W1D[form]
    m, a = at.shape[0], bt.shape[0]
    N, C, H, W = x.shape
    Wp = (W + m - 1) // m * m
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1))
    d = xp.unfold(3, a, m)                                              # [N, C, H + 2, T, a]
    V = torch.einsum("pj,nchtj->nchtp", bt, d)
    U = torch.einsum("pj,ockj->ockp", g, w)                            # [O, C, ky, a]
    if round_v:
        V = r16(V)
    if round_u:
        U = r16(U)
    M = sum(torch.einsum("ocp,nchtp->nohtp", U[:, :, ky], V[:, :, ky:ky + H]) for ky in range(3))
    Y = torch.einsum("ip,nohtp->nohti", at, M)
    return Y.reshape(N, w.shape[0], H, Wp)[..., :W]


class sim:
    """the build's KBPN storage plan on every conv + Winograd on the layers ``wino`` selects"""

    def __init__(self, names, wino, code_ch=441, **kw):
        self.names, self.wino, self.code, self.kw, self.cache = names, wino, code_ch, kw, {}

    def __enter__(self):
        self.saved = (F.conv2d, F.conv_transpose2d)
        oc, ot = self.saved

        def wrap(fn, transposed):
            def f(x, w, b=None, *a, **k):
                n = self.names.get(id(w))
                if n is None or not n.startswith("sr_model"):
                    return fn(x, w, b, *a, **k)
                xi = r16(x)
                if ".kernel_predictor." in n:
                    w16 = r16(w)
                else:
                    key = (id(w), transposed)
                    if key not in self.cache:
                        st = (a[0] if a else k.get("stride", 1))
                        st = st[0] if isinstance(st, (tuple, list)) else st
                        self.cache[key] = tapsum_round(w, transposed, int(st))
                    w16 = self.cache[key]
                if transposed or not self.wino(n) or w.shape[2] != 3:
                    return fn(xi, w16, b, *a, **k)
                cf = w.shape[1] - self.code if "conv0" in n else w.shape[1]        # conv0 reads cat(features, 441 code channels)
                kw = dict(self.kw)
                form = kw.pop("form", None)
                y = winograd_conv3x3_1d(xi[:, :cf], w16[:, :cf], form, **kw) if form else winograd_conv3x3(xi[:, :cf], w16[:, :cf], **kw)
                if cf < w.shape[1]:
                    y = y + oc(x[:, cf:], w[:, cf:], None, 1, 1)                    # the folded code channels: exact fp32 in the build
                return y + (b.view(1, -1, 1, 1) if b is not None else 0)
            return f
        F.conv2d, F.conv_transpose2d = wrap(oc, False), wrap(ot, True)
        return self

    def __exit__(self, *exc):
        F.conv2d, F.conv_transpose2d = self.saved
        return False


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "wc2_pspnet_it40000"
    torch.set_num_threads(8)
    g = load_golden(case)
    cfg = golden_cfg(g)
    P = det_params(scale=int(g["scale"]), detector=str(g["detector"]), requires_grad=False, style=fill_style(g),
                   pixel_shuffle=bool(g["pixel_shuffle"]) if "pixel_shuffle" in g else False)
    names = {id(v): k for k, v in P.items()}
    from test_wc_parity_gpu import _inputs
    x, hr, mask, k = _inputs(g)
    drop = {kk.split(".", 1)[1]: torch.from_numpy(v) for kk, v in g.items() if kk.startswith("dropmask.")}
    it = int(g["it"])

    # self-check of the emulation: without the two roundings it IS the convolution
    xx, ww = torch.randn(2, 5, 8, 12), torch.randn(7, 5, 3, 3)
    chk = float((winograd_conv3x3(xx, ww, False, False) - F.conv2d(xx, ww, None, 1, 1)).abs().max())
    assert chk < 1e-4, chk
    for form in W1D:
        chk = float((winograd_conv3x3_1d(xx, ww, form, False, False) - F.conv2d(xx, ww, None, 1, 1)).abs().max())
        assert chk < 2e-4, (form, chk)

    def run(wino, **kw):
        with torch.no_grad(), sim(names, wino, **kw):
            sr, kvec = O.kbpn_forward(P, x, it, k, cfg)
            bn = O.BNState(P, True)
            seg, aux = O.pspnet_forward(P, O.norm_sr(sr, cfg), bn, drop, kvec if cfg.detector == "PSPNet_BlurSkip" else None)
        return (max_rel_to_scale(sr, g["sr_preds"]), rel_err(sr, g["sr_preds"]), max_rel_to_scale(seg, g["segment_preds"]),
                rel_err(seg, g["segment_preds"]))

    no = lambda n: False
    c1 = lambda n: ".sft." in n and "conv1" in n
    c01 = lambda n: ".sft." in n
    rows = [("direct products (the build's plan today)", no, {}),
            ("Winograd SFT conv1, fp32 transforms (sanity)", c1, dict(round_v=False, round_u=False)),
            ("Winograd SFT conv1, fp16 V only", c1, dict(round_u=False)),
            ("Winograd SFT conv1, fp16 U only", c1, dict(round_v=False)),
            ("Winograd SFT conv1, fp16 U and V", c1, {}),
            ("Winograd SFT conv1 + conv0, fp16 U and V", c01, {}),
            ("1-D F(2,3) along x, SFT conv1 + conv0, fp16 U and V", c01, dict(form="F(2,3)")),
            ("1-D F(4,3) along x, SFT conv1 + conv0, fp16 U and V", c01, dict(form="F(4,3)"))]
    print(f"{case}: error against the reference fixture, max|a-b|/max|b| (rel-L2); stop criterion: SR > 9e-4")
    for name, wino, kw in rows:
        e = run(wino, **kw)
        print(f"{name:52s} sr {e[0]:.2e} ({e[1]:.2e})   seg {e[2]:.2e} ({e[3]:.2e})", flush=True)


if __name__ == "__main__":
    main()
