"""GPU parity of the full hot path (JointModelWithLoss forward + backward through the C ABI) against
(a) the golden vectors produced by the reference and (b) the oracle run on the same inputs, at the small fixture size (HR 64).

Every bound below is a fixed number (north_star: 1e-3 relative, fp16 storage / fp32 accumulate):
  SR image, kernel, SR loss: max |a-b| / max|b| <= 1e-3 per tensor.
  Segmentation map / loss / BatchNorm buffers of the COMPOSED path: the random-weight detector amplifies the (tolerated) 1e-3 error of
  its input ~100x -- the reference's own response is recorded in the tests/golden/wc_* fixtures -- so they are held to the constants
  SEG_BOUND here, and to 1e-3 where the claim can be made: on the reference's own SR image (tests/test_wc_parity_gpu.py).
  READ THIS BEFORE QUOTING A NUMBER FROM THIS FILE: SEG_BOUND and the joint-phase JOINT_MEDIAN / JOINT_P90 gradient bounds are SANITY
  bounds only (0.12 .. 0.45 on the map, 0.5 .. 1.2 on the gradient median: "the detector half ran and is not garbage").  They carry no
  parity claim.  The parity claims for the segmentation side live in tests/test_wc_parity_gpu.py (detector on the reference's SR image,
  1e-3) and tests/test_wc2_composed_gpu.py (composed step on the contractive fixtures, tight fixed bounds); what THIS file pins at
  north_star's tolerance is the SR side of every phase and cfg variant (SR image / kernel / SR loss <= 1e-3, SR-phase gradients <= 3e-2).
  Gradients: relative L2 per parameter tensor vs the fp32 oracle, 3e-2 in the SR-only phases (the oracle's own fp32
  evaluation-order noise reaches 1e-2 on a few tensors and 60 % on near-zero PReLU-slope sums, tests/test_oracle_golden.py);
  joint-phase gradients are pinned half by half in tests/test_wc_parity_gpu.py and by test_sr_loss_gradients_match_oracle here.
"""
import numpy as np
import pytest
import torch

from golden_utils import load_golden, det_params, rel_err, max_rel_to_scale, golden_cfg
from oracle import csbsr_oracle as O

pytestmark = pytest.mark.gpu

# composed-path bounds at HR 64 (max |a-b| / max|b|): (segmentation map, segmentation loss, BatchNorm running buffers, min IoU)
SEG_BOUND = {"PSPNet": (0.12, 2e-2, 5e-2, 0.95), "PSPNet_BlurSkip": (0.12, 2e-2, 5e-2, 0.95), "HRNet_OCR": (0.45, 0.1, 0.3, 0.70)}


def gate_flip_prone(name):
    """fe_kernel.0 of a stage's kernel predictor is a conv3x3 + LeakyReLU of a SPATIALLY CONSTANT map (kbpn.py:565-567): each of its 49
    pre-activations is one number per (sample, border class), so when one of them changes sign the activation derivative of the whole
    channel -- every pixel at once -- jumps, and with it 1/49 of that weight's gradient.  The e2e_pspnet_* fixtures (seed 1121) hold such a
    value: stage index 1, sample 0, channel 24 is 3.6e-6 against an rms of 1.1e-2, i.e. 3e-4 of the typical magnitude, while the kernel
    vector the HIP path feeds it is only good to ~1e-3 (north_star's tolerance).  Its sign differs from the reference's and that ONE
    tensor sits 4.7e-2 away (every other kernel-predictor tensor: <= 4e-3; measured with scripts/debug_grads.py).  A discontinuity of
    the function, not an error of the gradient: these tensors get a bound of 0.1."""
    return name.endswith("kernel_predictor.fe_kernel.0.layer.weight")


def build_model(g, micro_batch=8):
    from csbsr_amd.config import cfg as base_cfg
    from csbsr_amd.modeling.build_model import JointModelWithLoss
    from csbsr_amd.utils.detfill import deterministic_fill
    cfg = base_cfg.clone()
    cfg.MODEL.SCALE_FACTOR = int(g["scale"])
    if "beta" in g:
        cfg.SOLVER.TASK_LOSS_WEIGHT = float(g["beta"])
    if "detector" in g:
        cfg.MODEL.DETECTOR_TYPE = str(g["detector"])
        cfg.SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP = float(g["sfo_sr_amp"])
        cfg.SOLVER.ORIENTED_WEIGHT_ITER = int(g["oriented_w_iter"])
    if "pixel_shuffle" in g:
        cfg.MODEL.SR_PIXEL_SHUFFLE = bool(g["pixel_shuffle"])
    if "residual_learning" in g:
        cfg.MODEL.SR_RESIDUAL_LEARNING, cfg.SOLVER.ONLY_KERNEL_LOSS_FOR_PRETRAIN = bool(g["residual_learning"]), bool(g["only_kernel_loss"])
    if "kernel_sft" in g:
        cfg.MODEL.KBPN_KERNEL_SFT, cfg.MODEL.SUM_LR_ERROR_POS = bool(g["kernel_sft"]), ("LR" if bool(g["lr_error"]) else "HR")
    if "zero_pad_kernel" in g:
        cfg.MODEL.ZERO_PAD_KERNEL = bool(g["zero_pad_kernel"])
    m = JointModelWithLoss(cfg, 1000, 0, None, antialias=bool(g["antialias"]))
    deterministic_fill(m.state_dict())
    m.ss_loss_fn.alpha = float(g["alpha"])
    m.micro_batch = micro_batch
    m.max_resident = 0 if micro_batch == 1 else 8        # micro_batch=1 exercises the recompute-in-backward path
    m.train()
    drop = {k.split(".", 1)[1]: torch.from_numpy(v) for k, v in g.items() if k.startswith("dropmask.")}
    m.dropout_masks = drop        # {} -> every Dropout2d is the identity
    return m, cfg


def run_hip(g, micro_batch=8):
    m, cfg = build_model(g, micro_batch)
    t = lambda k: torch.from_numpy(g[k])
    it = int(g["it"])
    seg_l, sr_l, seg, sr, kp = m(it, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
    pc = m.pc
    loss = (1 - pc.beta) * sr_l.mean() + pc.beta * seg_l.mean()
    if pc.joint_pretrain[0] <= it < pc.joint_pretrain[1]:
        loss = sr_l.mean()
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: (None if v.grad is None else v.grad.detach().cpu()) for k, v in m._named_full() if isinstance(v, torch.nn.Parameter)}
    outs = dict(segment_loss=seg_l.detach().cpu(), sr_loss=sr_l.detach().cpu(), segment_preds=seg.cpu(), sr_preds=sr.cpu(),
                kernel_preds=kp.cpu(), loss=float(loss))
    bufs = {k: v.detach().cpu() for k, v in m.state_dict().items() if "running" in k}
    return outs, grads, bufs


def run_oracle(g):
    cfg = golden_cfg(g)
    P = det_params(scale=cfg.scale, detector=cfg.detector, pixel_shuffle=cfg.pixel_shuffle, kernel_sft=cfg.kernel_sft, lr_error=cfg.lr_error,
                   zero_pad_kernel=cfg.zero_pad_kernel)
    t = lambda k: torch.from_numpy(g[k])
    drop = {k.split(".", 1)[1]: torch.from_numpy(v) for k, v in g.items() if k.startswith("dropmask.")}
    out = O.joint_forward(P, cfg, int(g["it"]), t("x"), t("hr"), t("mask"), t("kernel"), alpha=float(g["alpha"]), drop=drop or None)
    loss = O.calc_loss(out["segment_loss"], out["sr_loss"], int(g["it"]), cfg)
    loss.backward()
    return P, out, loss


@pytest.mark.parametrize("case", ["e2e_pspnet_it40000", "e2e_pspnet_it40000_dropout", "e2e_pspnet_it1", "e2e_pspnet_it10001",
                                  "e2e_pspnet_it20001", "e2e_pspnet_it40000_noaa", "e2e_pspnet_it40000_lr24", "e2e_blurskip_x8_it40000",
                                  "e2e_pspnet_wf_it40000", "e2e_hrnet_ocr_it40000", "e2e_pspnet_pixelshuffle_it20001",
                                  "e2e_pspnet_noresidual_it40000", "e2e_pspnet_konly_it10001", "e2e_pspnet_nosft_it40000",
                                  "e2e_pspnet_lrerr_it40000", "e2e_pspnet_lrerr_it1", "e2e_pspnet_zeropad_it40000"])
def test_forward_matches_golden(case):
    g = load_golden(case)
    det = str(g["detector"]) if "detector" in g else "PSPNet"
    b_seg, b_segl, b_bn, b_iou = SEG_BOUND[det]
    outs, grads, bufs = run_hip(g)
    worst = {k: max_rel_to_scale(outs[k], g[k]) for k in ("segment_preds", "sr_preds", "kernel_preds", "segment_loss", "sr_loss")}
    e_bn = max([max_rel_to_scale(bufs[k[4:]], v) for k, v in g.items() if k.startswith("buf.")] or [0.0])
    iou = float(O.iou(outs["segment_preds"], torch.from_numpy(g["segment_preds"])).min())
    print(case, {k: f"{v:.1e}" for k, v in worst.items()}, f"BN buffers {e_bn:.1e} IoU vs ref {iou:.4f}")
    # MODEL.SR_PIXEL_SHUFFLE with the deterministic kaiming-like fill: the 3x3 conv + PixelShuffle blocks have 2.25x the fan-in gain
    # of the 8x8 stride-4 deconvs they replace, activations grow ~30x through the four stages (SR loss 9.5 instead of 0.3) and
    # the fp16 rounding of the larger intermediate sums shows as 2.2e-3 of the output's maximum: an ill-conditioned FIXTURE (random-fill
    # gain), not a property of the variant -- the contractive-fill fixture wc2_pspnet_pixelshuffle_it40000 (joint phase) holds the same
    # code to 1e-3 (tests/test_wc2_composed_gpu.py, measured 7.6e-4).  Its own bound here: 4e-3
    tol_sr = 4e-3 if bool(g.get("pixel_shuffle", False)) else 1e-3
    if not bool(g.get("residual_learning", True)):
        # MODEL.SR_RESIDUAL_LEARNING = False: the SR image is the stack's output alone, max |sr| 0.48 instead of 1.19 with the (exactly
        # computed) bicubic term, so the same absolute error -- 6.8e-4 of the image range here, 8e-4 for the default variant -- is 2.5x larger
        # relative to this map's own maximum
        tol_sr = 2.5e-3
    if bool(g.get("lr_error", False)):
        # MODEL.SUM_LR_ERROR_POS = 'LR': the back-projection error reaches the next stage through a 27-term 3x3 conv of the fp16 LR error
        # image instead of a 192-term 8x8 deconv (less averaging of its storage rounding) and the stage's LR features are rounded once more
        # (down(...) and down(...) + conv(error) are both stored): 1.06e-3 / 1.09e-3 of the SR image's maximum on the two fixtures, where the
        # default variant has 6.2e-4 with the same weights; gradients agree with the oracle like the default variant's (2.7e-3 median at iter 1)
        tol_sr = 1.5e-3
    for k in ("sr_preds", "kernel_preds", "sr_loss"):
        # (with the w^F weight on, the SR loss is weighted by exp(|seg - mask|): it inherits the segmentation map's conditioning)
        assert worst[k] < (5e-3 if k == "sr_loss" and float(g.get("sfo_sr_amp", 0.0)) != 0 else tol_sr), (k, worst[k])
    assert worst["segment_preds"] < b_seg and worst["segment_loss"] < b_segl and e_bn < b_bn
    if det != "HRNet_OCR":             # (random-weight HRNet-OCR: most probabilities sit within 1e-2 of the 0.5 threshold, IoU is noise)
        assert iou > b_iou
    assert abs(outs["loss"] - float(g["loss"])) < b_segl * abs(float(g["loss"]))


@pytest.mark.parametrize("case", ["e2e_pspnet_it1", "e2e_pspnet_it10001", "e2e_pspnet_it20001", "e2e_pspnet_it40000",
                                  "e2e_pspnet_it40000_dropout", "e2e_blurskip_x8_it40000", "e2e_pspnet_wf_it40000", "e2e_hrnet_ocr_it40000",
                                  "e2e_pspnet_pixelshuffle_it20001", "e2e_pspnet_noresidual_it40000", "e2e_pspnet_konly_it10001",
                                  "e2e_pspnet_nosft_it40000", "e2e_pspnet_lrerr_it40000", "e2e_pspnet_lrerr_it1", "e2e_pspnet_zeropad_it40000"])
def test_gradients_match_oracle(case):
    """Per-parameter relative L2 error of the HIP gradients vs the fp32 oracle.

    SR-only phases (iter < 30001): 3e-2 per tensor, median < 5e-3.  Joint phase: the gradient of the COMPOSED path goes through the
    random-weight detector evaluated on an SR image that differs by the tolerated ~1e-3, which moves the reference's own gradients by
    a median 0.15 .. 1.4 in relative L2 (fixture keys cond_grad_*, tests/golden/wc_*): there the composed gradient only gets the
    fixed sanity bounds JOINT_MEDIAN / JOINT_P90, and the two halves are held to 3e-2 separately on the reference's own
    intermediate tensors (tests/test_wc_parity_gpu.py) plus test_sr_loss_gradients_match_oracle below."""
    JOINT_MEDIAN, JOINT_P90 = {"PSPNet": 0.5, "PSPNet_BlurSkip": 0.3, "HRNet_OCR": 1.2}, {"PSPNet": 0.7, "PSPNet_BlurSkip": 0.5, "HRNet_OCR": 1.6}
    g = load_golden(case)
    outs, grads, _ = run_hip(g)
    P, out, loss = run_oracle(g)
    it = int(g["it"])
    joint = it >= 30001
    hrnet = "detector" in g and str(g["detector"]) == "HRNet_OCR"
    det = str(g["detector"]) if "detector" in g else "PSPNet"
    errs, bad = [], []
    names = [str(n) for n in g["grad_names"]]
    for n, ref_norm in zip(names, g["grad_norms"]):
        hip = grads[n]
        if ref_norm < 0:                       # reference: grad None (frozen / unused in this phase)
            assert hip is None or float(hip.abs().max()) == 0.0, n
            continue
        og = P[n].grad
        if ref_norm < 1e-7:
            assert hip is None or float(hip.norm()) < 1e-5, n
            continue
        assert hip is not None, n
        if og.numel() == 1 and hrnet:
            continue        # chaotic at this size (see tests/test_hrnet_gpu.py for the backward check of this detector)
        if og.numel() == 1:
            # PReLU slope = signed sum over ~1e6 products; fp32 orders already differ by 10 % (test_oracle_golden.py)
            tol = (0.3 if bool(g.get("pixel_shuffle", False)) else 0.1) * abs(float(og)) + 2e-3
            if not joint:
                assert abs(float(hip) - float(og)) <= tol, (n, float(hip), float(og))
            continue
        e = rel_err(hip, og)
        errs.append(e)
        # (PixelShuffle variant: run to run the HIP gradients of the stage-1 kernel predictor themselves move by 1-3e-2, see
        # test_forward_matches_golden; observed against the oracle: typical maximum 1.3-4e-2, once 0.22 in ~15 runs -- so per tensor only a
        # sanity bound, the distribution carries the check)
        if not joint and e > (0.3 if bool(g.get("pixel_shuffle", False)) else 0.1 if gate_flip_prone(n) else 3e-2):
            bad.append((n, e))
    errs = np.array(errs)
    print(case, "grad rel-L2 vs fp32 oracle: median %.2e  p90 %.2e  max %.2e  (n=%d)" % (np.median(errs), np.percentile(errs, 90), errs.max(), len(errs)))
    assert not bad, (len(bad), bad[:10])
    assert np.isfinite(errs).all()
    if joint:
        assert np.median(errs) < JOINT_MEDIAN[det] and np.percentile(errs, 90) < JOINT_P90[det]
    else:
        if bool(g.get("pixel_shuffle", False)):       # measured median 7.2-8.8e-3, p90 1.2-1.4e-2
            assert np.median(errs) < 1.5e-2 and np.percentile(errs, 90) < 3e-2
        else:
            assert np.median(errs) < 5e-3


@pytest.mark.parametrize("case", ["e2e_pspnet_it40000", "e2e_pspnet_wf_it40000"])
def test_sr_loss_gradients_match_oracle(case):
    """d(mean sr_loss)/d(parameters) in the joint phase, with and without the w^F weight (a14): no BatchNorm stack in this path, so the
    bound is direct.  The w^F map exp(|seg - mask|) is a detached function of the fp16 segmentation output (1.5e-2 off fp32 at this size),
    which moves the weighted gradients by about that much: measured median 2.9e-3, max 1.5e-2."""
    g = load_golden(case)
    m, cfg = build_model(g)
    t = lambda k: torch.from_numpy(g[k])
    it = int(g["it"])
    seg_l, sr_l, seg, sr, kp = m(it, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
    sr_l.mean().backward()
    grads = {k: v.grad for k, v in m._named_full() if isinstance(v, torch.nn.Parameter)}
    oc = golden_cfg(g)
    P = det_params(scale=oc.scale, detector=oc.detector)
    out = O.joint_forward(P, oc, it, t("x"), t("hr"), t("mask"), t("kernel"), alpha=float(g["alpha"]))
    out["sr_loss"].mean().backward()
    errs = []
    for n, p in P.items():
        og = getattr(p, "grad", None)
        if n.startswith("segmentation_model"):
            assert grads.get(n) is None or float(grads[n].abs().max()) == 0.0, n
            continue
        if og is None or og.numel() == 1 or float(og.norm()) < 1e-9:
            continue
        assert grads[n] is not None, n
        e = rel_err(grads[n].cpu(), og)
        assert e < 0.1, (n, e)
        if not gate_flip_prone(n):
            errs.append(e)
    errs = np.array(errs)
    print(case, "SR-loss grads: median %.2e max %.2e (n=%d)" % (np.median(errs), errs.max(), len(errs)))
    assert len(errs) > 100 and np.median(errs) < 5e-3 and errs.max() < 3e-2


def test_micro_batching_is_exact():
    """KBPN has no batch-coupled op: micro-batch 1 (forward recomputed inside backward) == whole batch resident.
    The SR output is identical up to the fp32 summation order of the GAP partial sums; everything downstream of the fp16 BatchNorm
    stack moves within the fp16 noise floor (see above)."""
    g = load_golden("e2e_pspnet_it1")
    o1, g1, _ = run_hip(g, micro_batch=1)
    o2, g2, _ = run_hip(g, micro_batch=8)
    assert max_rel_to_scale(o1["sr_preds"], o2["sr_preds"]) < 1e-4
    assert max_rel_to_scale(o1["sr_loss"], o2["sr_loss"]) < 1e-4
    worst = max(rel_err(g1[k], g2[k]) for k in g1 if g1[k] is not None and g1[k].numel() > 1)
    assert worst < 5e-3, worst


def test_loss_scale_overflow_backoff():
    """fp16 gradient overflow follows GradScaler semantics: that step is skipped -- every gradient is None, so optimizer.step() leaves
    weights AND Adam moments untouched (never inf/nan, never a momentum-only update) --, the automatic loss scale drops by 2^4 for
    the next step, which then succeeds."""
    import warnings
    g = load_golden("e2e_pspnet_it40000")
    m, cfg = build_model(g)
    t = lambda k: torch.from_numpy(g[k])
    m.scale_backoff = -40                      # 2^40 x the automatic scale: certain overflow
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        seg_l, sr_l, *_ = m(40000, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
        (0.7 * sr_l.mean() + 0.3 * seg_l.mean()).backward()
    assert any("overflow" in str(x.message) for x in w)
    assert m.overflow_steps == 1 and m.scale_backoff == -36 and m.last_step_overflowed
    assert all(p.grad is None for p in m.parameters())
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in m.parameters()]
    opt.step()                                 # nothing to do: no parameter has a gradient
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, m.parameters())) and len(opt.state) == 0
    m.scale_backoff = 0
    m.zero_grad()
    seg_l, sr_l, *_ = m(40000, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
    (0.7 * sr_l.mean() + 0.3 * seg_l.mean()).backward()
    assert m.overflow_steps == 1 and not m.last_step_overflowed
    assert all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None)
    assert sum(float(p.grad.abs().sum()) for p in m.parameters() if p.grad is not None) > 0


def test_three_step_training_trajectory_matches_oracle():
    """The harness around the path: calc_loss phase gating (trainer.py:406-438), KBPN phase freezing (kbpn.py:118-142),
    Adam with the reference's hyper-parameters (train.py:91) -- three optimiser steps in the SR-pretrain phase (iters 1..3; no
    BatchNorm stack in the gradient, so the comparison is direct) from the same deterministic weights, HIP vs the oracle with
    torch.optim.Adam on the CPU.  The loss moves by ~1e-3 per step; its change after 3 steps must agree to 5 %."""
    g = load_golden("e2e_pspnet_it1")
    t = lambda k: torch.from_numpy(g[k])
    lr = 2e-4
    # oracle
    cfg_o = golden_cfg(g)
    P = det_params(scale=cfg_o.scale, detector=cfg_o.detector)
    frozen = lambda n: "kernel_predictor" in n or ".predictor." in n or n.startswith("segmentation_model")
    leaves = [v for k, v in P.items() if v.requires_grad and not frozen(k)]
    opt_o = torch.optim.Adam(leaves, lr=lr, betas=(0.9, 0.999), eps=1e-8)
    lo = []
    for it in (1, 2, 3, 4):
        out = O.joint_forward(P, cfg_o, it, t("x"), t("hr"), t("mask"), t("kernel"), alpha=float(g["alpha"]))
        for k, v in out["bn_buffers"].items():
            P[k] = v.detach()
        loss = O.calc_loss(out["segment_loss"], out["sr_loss"], it, cfg_o)
        lo.append(float(loss))
        if it == 4:
            break
        opt_o.zero_grad()
        loss.backward()
        for k, v in P.items():       # the segmentation net gets gradient None in this phase in the reference (unused by the loss)
            if v.requires_grad and frozen(k):
                v.grad = None
        opt_o.step()
    # HIP
    m, cfg = build_model(g)
    opt = torch.optim.Adam(m.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8)
    lh = []
    for it in (1, 2, 3, 4):
        seg_l, sr_l, *_ = m(it, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
        loss = sr_l.mean()           # calc_loss inside SR_PRETRAIN_ITER
        lh.append(float(loss))
        if it == 4:
            break
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    print("loss trajectory oracle", lo, "hip", lh)
    assert abs(lh[0] - lo[0]) < 1e-3 * lo[0]
    d_o, d_h = lo[3] - lo[0], lh[3] - lh[0]
    assert abs(d_o) > 1e-4 and abs(d_h - d_o) < 0.05 * abs(d_o) + 1e-5, (d_o, d_h)
    for a, b in zip(lh, lo):          # Adam's first steps are sign-like (+-lr per weight): near-zero gradient entries may move the other way
        assert abs(a - b) < 4e-3 * b


@pytest.mark.parametrize("case", ["e2e_pspnet_it40000", "e2e_blurskip_x8_it40000", "e2e_hrnet_ocr_it40000"])
def test_joint_model_inference_matches_oracle(case):
    """JointModel.forward (build_model.py:466-496) in eval mode: iter = -1 (kernel predictor on), SR clipped to [0,1] before the
    detector, running-statistics BatchNorm, no dropout, kernel normalised to sum 1 -- vs the oracle and its fp16-storage emulation."""
    from csbsr_amd.config import cfg as base_cfg
    from csbsr_amd.modeling.build_model import JointModel
    from csbsr_amd.utils.detfill import deterministic_fill
    g = load_golden(case)
    oc = golden_cfg(g)
    cfg = base_cfg.clone()
    cfg.MODEL.SCALE_FACTOR, cfg.MODEL.DETECTOR_TYPE = oc.scale, oc.detector
    m = JointModel(cfg, antialias=bool(g["antialias"]))
    deterministic_fill(m.state_dict())
    m.eval()
    x, k = torch.from_numpy(g["x"]), torch.from_numpy(g["kernel"])
    # the deterministic fill's running statistics are arbitrary numbers: calibrate them to this input with one train-mode pass of the
    # oracle at momentum 1 (running := batch statistics), as a trained checkpoint's would be, and load them into both sides
    P0 = det_params(scale=oc.scale, detector=oc.detector, requires_grad=False)
    with torch.no_grad():
        sr_c, kv_c = O.kbpn_forward(P0, x, -1, k, oc)
        bn_c = O.BNState(P0, True, momentum=1.0)
        xin_c = O.norm_sr(sr_c.clamp(0, 1), oc)
        if oc.detector == "HRNet_OCR":
            O.hrnet_ocr_forward(P0, xin_c, bn_c, None)
        else:
            O.pspnet_forward(P0, xin_c, bn_c, None, kv_c if oc.detector == "PSPNet_BlurSkip" else None)
    calib = {k_: v for k_, v in bn_c.new.items() if "running" in k_}
    sd = m.state_dict()
    for k_, v in calib.items():
        sd[k_].copy_(v)
    sr, seg, kp = m(x, k)
    torch.cuda.synchronize()

    def oracle():
        P = det_params(scale=oc.scale, detector=oc.detector, requires_grad=False)
        P.update(calib)
        with torch.no_grad():
            sr_o, kvec = O.kbpn_forward(P, x, -1, k, oc)
            sr_o = sr_o.clamp(0, 1)
            bn = O.BNState(P, False)
            xin = O.norm_sr(sr_o, oc)
            if oc.detector == "HRNet_OCR":
                seg_o, _ = O.hrnet_ocr_forward(P, xin, bn, None)
            else:
                seg_o, _ = O.pspnet_forward(P, xin, bn, None, kvec if oc.detector == "PSPNet_BlurSkip" else None)
            kv = kvec / kvec.sum(1, keepdim=True)
        return sr_o, seg_o, kv.reshape(kp.shape)
    sr_o, seg_o, kp_o = oracle()
    assert max_rel_to_scale(sr.cpu(), sr_o) < 1e-3
    assert max_rel_to_scale(kp.cpu(), kp_o) < 1e-3
    assert abs(float(kp.sum(dim=(1, 2, 3)).mean()) - 1.0) < 1e-4
    e = max_rel_to_scale(seg.cpu(), seg_o)
    iou = float(O.iou(seg.cpu(), seg_o).min())
    print(case, "inference seg: hip %.2e  IoU vs oracle %.4f" % (e, iou))
    assert e < SEG_BOUND[oc.detector][0]
    if oc.detector != "HRNet_OCR":             # (random-weight HRNet-OCR: most probabilities sit within 1e-2 of the 0.5 threshold)
        assert iou > SEG_BOUND[oc.detector][3]


@pytest.mark.parametrize("it", [1, 20001])
def test_specialised_kernel_paths_agree_with_the_general_kernels(it):
    """csrc/conv_tp.hip only takes large launches by default: here the phase-decomposed transposed convs, incl. the dgrads that take
    over up_conv1's / down.conv's epilogue-backward pass, are forced onto a whole training step at LR 32x32 (whole 8x32 tiles),
    against the same step through the general implicit-GEMM kernels + stand-alone epilogue-backward passes: same model, same
    batch, every output and every KBPN gradient tensor, in the two SR-loss-driven phases (in the joint phase the random-weight
    detector at this size turns run-to-run summation-order noise into O(1) gradient differences, whichever kernels run)."""
    from csbsr_amd import _lib as L
    from csbsr_amd.data.synthetic import make_batch
    g = load_golden("e2e_pspnet_it40000")
    lib = L.load()
    x, hr, mask, k = make_batch(2, 32, seed=5)
    res = []
    for mode in (2, 0):
        lib.csbsr_debug_set_conv_tp(mode)
        try:
            m, cfg = build_model(g)
            seg_l, sr_l, seg, sr, kp = m(it, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
            sr_l.mean().backward()
            torch.cuda.synchronize()
            fused = [s_.up2.last_fused for s_ in m._runtime()["kbpn"].stages]
        finally:
            lib.csbsr_debug_set_conv_tp(1)
        assert all(fused) == (mode == 2)
        grads = {n: v.grad.detach().cpu() for n, v in m._named_full() if isinstance(v, torch.nn.Parameter) and v.grad is not None}
        res.append((sr.cpu(), kp.cpu(), sr_l.detach().cpu(), grads))
    (sr_a, kp_a, l_a, g_a), (sr_b, kp_b, l_b, g_b) = res
    assert max_rel_to_scale(sr_a, sr_b) < 5e-4 and max_rel_to_scale(kp_a, kp_b) < 5e-4 and max_rel_to_scale(l_a, l_b) < 5e-4
    assert g_a.keys() == g_b.keys()
    errs = {n: float((g_a[n] - g_b[n]).norm() / (g_b[n].norm() + 1e-30)) for n in g_a if n.startswith("sr_model") and g_b[n].numel() > 1}
    worst = max(errs, key=errs.get)
    print(f"it {it}: {len(errs)} KBPN gradient tensors, fused vs general kernels: median {np.median(list(errs.values())):.2e} worst {errs[worst]:.2e} ({worst})")
    # the two paths differ by the fp16 rounding of one intermediate per fused pass (measured: median 6.7e-4, worst 1.1e-3)
    assert errs[worst] < 5e-3 and np.median(list(errs.values())) < 2e-3
    # PReLU slopes (scalars; those of up_conv1 / down.conv now come out of the dgrad epilogue): signed sums with heavy cancellation --
    # the reference's own fp32 evaluation orders differ by ~10 % on them (tests/test_oracle_golden.py) -- so 15 % like everywhere else
    for n in g_a:
        if n.startswith("sr_model") and g_b[n].numel() == 1:
            assert abs(float(g_a[n]) - float(g_b[n])) < 0.15 * abs(float(g_b[n])) + 1e-6 * float(l_b.abs().max()), n


def test_zero_pad_kernel_recompute_replays_the_pad_decisions():
    """MODEL.ZERO_PAD_KERNEL with its nn.Dropout layers ON (kbpn.py:543-554: the pad discriminator's hard per-sample choice between the
    bicubic and the zero-padding update map is random per call in training).  A KBPN forward that is recomputed inside the backward
    (non-resident micro-batches) must be the SAME function as the one that produced the losses: it replays the stored decisions
    instead of drawing new dropout masks (KBPN.forward(pad_replay=...))."""
    g = load_golden("e2e_pspnet_zeropad_it40000")
    t = lambda k: torch.from_numpy(g[k])
    m, cfg = build_model(g, micro_batch=1)          # two micro-batches, max_resident = 0: both forwards are recomputed in the backward
    assert m.max_resident == 0
    m.dropout_masks, m.dropout_enabled = None, True
    kb = m._runtime()["kbpn"]
    assert kb.zero_pad
    calls = []
    orig = kb.forward

    def spy(*a, **kw):
        out = orig(*a, **kw)
        calls.append((kw.get("pad_replay"), [x.clone() for x in kb.pad_taken], kb.pad_dropout and kb.training_mode))
        return out
    kb.forward = spy
    torch.manual_seed(11)
    seg_l, sr_l, seg, sr, kp = m(40000, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
    first = [c[1] for c in calls]
    assert len(first) == 2 and all(len(f) == kb.S for f in first) and all(c[2] for c in calls), "dropout must be live in the first forwards"
    (0.7 * sr_l.mean() + 0.3 * seg_l.mean()).backward()
    torch.cuda.synchronize()
    redo = calls[2:]
    assert len(redo) == 2 and all(c[0] is not None for c in redo), "the recomputed forwards must be handed the stored decisions"
    for replay, taken, _ in redo:
        i = [j for j, f in enumerate(first) if all(torch.equal(a, b) for a, b in zip(f, replay))]
        assert i, "replayed decisions are not those of a first forward"
        assert all(torch.equal(a, b) for a, b in zip(taken, replay))
    assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in m.parameters())
