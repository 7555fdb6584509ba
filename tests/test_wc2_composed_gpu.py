"""The COMPOSED hot path (HIP KBPN -> HIP detector -> losses -> HIP backward, one JointModelWithLoss.forward + backward) against
REFERENCE goldens whose detector carries the CONTRACTIVE deterministic fill (csbsr_amd/utils/detfill.py, style "contractive": smooth
low-pass filters, small residual-branch gains -- the stack damps perturbations like a trained network; tests/golden/wc2_*.npz, made
by tests/golden/make_golden.py --wc2 from /root/reference).  On these the whole step is held to FIXED bounds, in both detector
precision modes, for all three detectors and the PixelShuffle KBPN variant:

    SR image, blur kernel, SR loss                     <= 1e-3 of the tensor's maximum            (north_star's tolerance)
    segmentation map (max |a-b| / max|b|), its loss, BatchNorm running buffers, 1 - IoU of the thresholded maps,
    every gradient tensor (joint phase) in relative L2 as (median, p90, max) of the distribution          <= B[fixture][mode]

The gradient bounds are NOT 1e-3 and cannot be for any implementation whose forward is not bit-identical: a forward deviation delta
(in units of the BatchNorm'd activations' std) flips the ReLU gate of ~0.4 delta of the elements, each flip changes that element's
gradient by 100 %, so a gradient tensor behind L such layers carries a relative L2 error of ~sqrt(0.4 delta L) -- 2e-2 for
delta = 1e-3 (fp16 storage), 2e-3 for delta = 1e-5 (split mode).  The fixtures record the REFERENCE's own response to a 1e-3
perturbation of its SR image (cond_*), printed next to every measurement.  Each measured value is printed with its bound; the bounds
leave >= 2x margin over the values measured on MI355X (the path is bit-reproducible, so the values do not move run to run)."""
import numpy as np
import pytest
import torch

from golden_utils import load_golden, max_rel_to_scale
from test_wc_parity_gpu import _inputs, _model, _grad_errors, _zero_by_construction

pytestmark = pytest.mark.gpu

# Fixed bounds per (fixture, detector precision mode): segmentation map (max |a-b| / max|b|), segmentation loss, BatchNorm running
# buffers, 1 - IoU of the thresholded maps, and (median, 90th percentile, max) of the per-tensor gradient relative-L2 errors.  Measured on
# MI355X (printed by the test; the path is bit-reproducible so the values do not move run to run) with ~2x margin each; the split-mode
# bounds on the map were tightened in r04 (5e-3 / 5e-3 / 1e-2 -> 4e-3 / 3e-3 / 5e-3) with the compensation of KBPN's weight rounding
# (engine.Conv._dc_bias: measured 2.18e-3 -> 1.66e-3, 1.44e-3 -> 1.19e-3, 3.08e-3 -> 1.98e-3) and again in r05 with the tap-sum-preserving
# rounding of KBPN's weights (engine.Conv._wq; same-run A/B, nearest -> tap-sum: PSPNet 1.76e-3 -> 1.08e-3, BlurSkip 1.20e-3 -> 1.02e-3,
# PixelShuffle 3.67e-3 -> 3.07e-3, HRNet-OCR 1.84e-3 -> 1.92e-3 in max norm; in relative L2 -- asserted since r05 as ``segl2`` -- 6.3 -> 5.0e-4,
# 3.0 -> 2.5e-4, 1.36 -> 1.10e-3, 1.38 -> 1.22e-3): 4e-3 / 3e-3 / 5e-3 / 1.5e-2 -> 2.2e-3 / 2e-3 / 4e-3 / 6e-3.  What the
# numbers say: in split mode (the default, and the mode bench.py's headline is quoted in) the step sits INSIDE the reference's own response
# to a 1e-3 perturbation of its SR image (cond_* in the fixtures: seg 4.8e-3 / 9.8e-3 / 3.7e-3 / 9.2e-3, IoU 0.995 / 0.997 / 0.965 / 0.988,
# gradient median 1.3e-2 / 5.8e-3 / 5.9e-2 / 2.0e-2 for PSPNet / BlurSkip / HRNet-OCR / PixelShuffle); plain fp16 detector storage costs
# 2-3x on the map and, through ~300 BatchNorm'd ReLU layers of HRNet-OCR, 6x on the gradients (0.22 median): it is the throughput mode,
# not the parity mode.  HRNet-OCR's IoU: its thresholded map has 5.6 % positives, so the ~75 pixels within 3e-3 of the threshold are 3.7 %.
B = {
    "wc2_pspnet_it40000": {"fp16": dict(seg=7e-3, segl2=3e-3, segl=5e-4, bn=1e-3, iou1=1e-2, grad=(2.5e-2, 7e-2, 0.15)),
                           "split": dict(seg=2.2e-3, segl2=1e-3, segl=2e-4, bn=5e-4, iou1=5e-3, grad=(2e-2, 5e-2, 0.1))},
    "wc2_blurskip_x8_it40000": {"fp16": dict(seg=1.3e-2, segl2=3.5e-3, segl=5e-4, bn=1e-3, iou1=1e-2, grad=(1.5e-2, 2e-2, 3e-2)),
                                "split": dict(seg=2e-3, segl2=5e-4, segl=2e-4, bn=5e-4, iou1=5e-3, grad=(5e-3, 1e-2, 1.5e-2))},
    "wc2_hrnet_ocr_it40000": {"fp16": dict(seg=1e-2, segl2=8e-3, segl=2e-3, bn=5e-3, iou1=0.15, grad=(0.5, 0.7, 2.0)),
                              "split": dict(seg=4e-3, segl2=2.5e-3, segl=1.5e-3, bn=3e-3, iou1=0.1, grad=(8e-2, 0.12, 0.3))},
    "wc2_pspnet_pixelshuffle_it40000": {"fp16": dict(seg=1e-2, segl2=3.5e-3, segl=1e-3, bn=2e-3, iou1=3e-2, grad=(5e-2, 0.12, 0.3)),
                                        "split": dict(seg=6e-3, segl2=2.2e-3, segl=5e-4, bn=1e-3, iou1=2e-2, grad=(4e-2, 0.1, 0.2))},
}
CASES = ["wc2_pspnet_it40000", "wc2_blurskip_x8_it40000", "wc2_hrnet_ocr_it40000", "wc2_pspnet_pixelshuffle_it40000"]


@pytest.mark.parametrize("precision", ["fp16", "split"])
@pytest.mark.parametrize("case", CASES)
def test_composed_step_matches_the_reference(case, precision):
    from oracle import csbsr_oracle as O
    g = load_golden(case)
    assert str(g["fill"]) == "contractive"
    x, hr, mask, k = _inputs(g)
    m = _model(g, precision)
    it, beta, det = int(g["it"]), float(g["beta"]), str(g["detector"])
    seg_l, sr_l, seg, sr, kp = m(it, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
    loss = (1 - beta) * sr_l.mean() + beta * seg_l.mean()
    loss.backward()
    torch.cuda.synchronize()
    e = {kk: max_rel_to_scale(v.detach().cpu(), g[kk]) for kk, v in
         (("sr_preds", sr), ("kernel_preds", kp), ("sr_loss", sr_l), ("segment_preds", seg), ("segment_loss", seg_l))}
    seg_ref = torch.from_numpy(g["segment_preds"])
    iou = float(O.iou(seg.cpu(), seg_ref).min())
    l2 = {kk: float((v.detach().cpu().double() - torch.from_numpy(g[kk]).double()).norm() / torch.from_numpy(g[kk]).double().norm())
          for kk, v in (("sr_preds", sr), ("segment_preds", seg))}
    sd = m.state_dict()
    e_bn = max(max_rel_to_scale(sd[kk[4:]].cpu(), v) for kk, v in g.items() if kk.startswith("buf."))
    e_loss = abs(float(loss.detach()) - float(g["loss"])) / abs(float(g["loss"]))
    bb = B[case][precision]
    pos = float((seg_ref > 0.5).float().mean())
    rows = [("sr_preds", e["sr_preds"], 1e-3), ("kernel_preds", e["kernel_preds"], 1e-3), ("sr_loss", e["sr_loss"], 1e-3),
            ("segment_preds", e["segment_preds"], bb["seg"]), ("segment_loss", e["segment_loss"], bb["segl"]), ("bn_buffers", e_bn, bb["bn"]),
            ("loss", e_loss, 1e-3), ("1-IoU", 1 - iou, bb["iou1"])]
    print(f"\n{case} [{precision}] composed step vs reference (positives {pos:.1%}); reference's own response to a 1e-3 SR perturbation: "
          f"IoU {float(g['cond_iou']):.4f} seg {float(g['cond_seg_max']):.1e} segloss {float(g['cond_segloss']):.1e} bn {float(g['cond_bn']):.1e} "
          f"grads median {float(g['cond_grad_median']):.1e} p90 {float(g['cond_grad_p90']):.1e}")
    for name, val, bound in rows:
        print(f"   {name:14s} {val:.2e}  bound {bound:.0e}  margin {bound / max(val, 1e-30):.1f}x")
    # (the rows above are max |a - b| / max |b|; in relative L2 the same maps sit at:)
    print(f"   relative L2: sr_preds {l2['sr_preds']:.2e}  segment_preds {l2['segment_preds']:.2e}")
    print(f"   relative L2 bound on segment_preds {bb['segl2']:.1e}  margin {bb['segl2'] / l2['segment_preds']:.1f}x")
    assert l2["sr_preds"] < 1e-3 and l2["segment_preds"] < bb["segl2"], (case, precision, l2)
    grads = {kk: v.grad for kk, v in m._named_full() if isinstance(v, torch.nn.Parameter)}
    errs = [er for er in _grad_errors(g, grads, "") if not _zero_by_construction(er[0])]
    v = np.array([max(en, es) for n, numel, en, es in errs if numel > 1])
    worst = max((er for er in errs if er[1] > 1), key=lambda er: max(er[2], er[3]))
    gb = bb["grad"]
    gm = (float(np.median(v)), float(np.percentile(v, 90)), float(v.max()))
    print(f"   gradients: {len(v)} tensors, rel-L2 median {gm[0]:.2e} (bound {gb[0]:.0e}, {gb[0] / gm[0]:.1f}x)  p90 {gm[1]:.2e} (bound {gb[1]:.0e}, "
          f"{gb[1] / gm[1]:.1f}x)  max {gm[2]:.2e} (bound {gb[2]:.0e}, {gb[2] / gm[2]:.1f}x; {worst[0]})")
    for split_at, tag in ((("segmentation_model",), "detector"), (("sr_model",), "KBPN")):
        vv = np.array([max(en, es) for n, numel, en, es in errs if numel > 1 and n.startswith(split_at)])
        if len(vv):
            print(f"      {tag}: {len(vv)} tensors, median {np.median(vv):.2e} p90 {np.percentile(vv, 90):.2e} max {vv.max():.2e}")
    for name, val, bound in rows:
        assert val < bound, (case, precision, name, val, bound)
    assert gm[0] < gb[0] and gm[1] < gb[1] and gm[2] < gb[2], (case, precision, gm, gb)
    # PReLU slopes (40 scalars in KBPN, 3 in PSPNet's up-blocks): each is a signed sum over every negative-input element of dy * x with
    # heavy cancellation, so its error is judged against the typical magnitude of the slope gradients, not against its own (possibly
    # near-zero) value, with the slack tied to the fixture's tensor-level bound: |hip - ref| <= 0.25 |ref| + max(0.1, 10 x median bound)
    # x median|ref|.  A sign or scale bug is O(1) of |ref| on all of them.
    sc = []
    for n, ref, smp in zip((str(v_) for v_ in g["grad_names"]), g["grad_norms"], g["grad_samples32"]):
        if ref > 0 and grads.get(n) is not None and grads[n].numel() == 1:
            sc.append((n, float(grads[n]), float(smp[0])))          # (the 32 "samples" of a scalar are all its signed value)
    if sc:
        med = float(np.median([abs(r) for _, _, r in sc]))
        worst_s = max(sc, key=lambda t: abs(t[1] - t[2]) - 0.25 * abs(t[2]))
        print(f"      {len(sc)} scalar (PReLU slope) gradients: median |ref| {med:.2e}; worst hip {worst_s[1]:+.3e} vs ref {worst_s[2]:+.3e} ({worst_s[0]})")
        slack = max(0.1, 10 * gb[0])
        bad = [(n, h, r) for n, h, r in sc if abs(h - r) > 0.25 * abs(r) + slack * med]
        assert not bad, bad[:6]
    assert 0.02 < pos < 0.98, "degenerate fixture: the thresholded reference map is (almost) constant"
