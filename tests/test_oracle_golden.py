"""Pin the oracle (oracle/csbsr_oracle.py) against golden vectors produced by the REFERENCE itself
(tests/golden/make_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from golden_utils import load_golden, det_params, rel_err, sample_idx, golden_cfg, fill_style
from oracle import csbsr_oracle as O

CASES = ["e2e_pspnet_it40000", "e2e_pspnet_it40000_dropout", "e2e_pspnet_it1", "e2e_pspnet_it10001",
         "e2e_pspnet_it20001", "e2e_pspnet_it40000_noaa", "e2e_pspnet_it40000_lr24",
         "e2e_blurskip_x8_it40000", "e2e_pspnet_wf_it40000", "e2e_hrnet_ocr_it40000", "e2e_pspnet_pixelshuffle_it20001",
         "e2e_pspnet_noresidual_it40000", "e2e_pspnet_konly_it10001", "e2e_pspnet_nosft_it40000", "e2e_pspnet_lrerr_it40000", "e2e_pspnet_lrerr_it1",
         "e2e_pspnet_zeropad_it40000"]

# fp32 CPU vs fp32 CPU, same torch build: differences come only from op ordering (grouped conv vs the
# reference's per-sample loop, vector kernel vs expanded map)
TOL_OUT = 2e-5
TOL_GRAD = 3e-2   # fp32 evaluation-order noise: up to 1.1e-2 on kb.sr_reconst weights (7e-3 vs an fp64 run of the oracle), <3e-4 elsewhere


def run_oracle(g, grads=True):
    cfg = golden_cfg(g)
    P = det_params(scale=cfg.scale, detector=cfg.detector, requires_grad=grads, pixel_shuffle=cfg.pixel_shuffle, kernel_sft=cfg.kernel_sft,
                   lr_error=cfg.lr_error, zero_pad_kernel=cfg.zero_pad_kernel)
    t = lambda k: torch.from_numpy(g[k])
    drop = {k.split(".", 1)[1]: torch.from_numpy(v) for k, v in g.items() if k.startswith("dropmask.")}
    taps = {}
    out = O.joint_forward(P, cfg, int(g["it"]), t("x"), t("hr"), t("mask"), t("kernel"),
                          alpha=float(g["alpha"]), drop=drop or None, taps=taps)
    loss = O.calc_loss(out["segment_loss"], out["sr_loss"], int(g["it"]), cfg)
    if grads:
        loss.backward()
    return P, out, loss, taps


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_outputs_and_grads(case):
    g = load_golden(case)
    P, out, loss, taps = run_oracle(g)
    hrnet = "detector" in g and str(g["detector"]) == "HRNet_OCR"
    # HRNet-OCR: ~300 BatchNorm layers deep, two of them over only B=4 object-context vectors: the same fp32 math in another op order
    # (functional vs module calls) lands 2.1e-5 away on the segmentation loss
    tol_out = 1e-4 if hrnet else TOL_OUT
    for k in ("segment_loss", "sr_loss", "segment_preds", "sr_preds", "kernel_preds"):
        assert rel_err(out[k].detach(), g[k]) < tol_out, k
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    for k, v in g.items():
        if k.startswith("tap."):
            mine = taps[k[4:]].detach()
            if tuple(mine.shape) != v.shape:
                mine = mine[:, :8]
            assert rel_err(mine, v) < TOL_OUT, k
        if k.startswith("buf."):
            assert rel_err(out["bn_buffers"][k[4:]], v) < tol_out, k
    # gradients: per-parameter L2 norm + 4 sampled elements, and the frozen/unused set must agree
    names = [str(n) for n in g["grad_names"]]
    n_checked = 0
    for n, ref_norm, ref_s in zip(names, g["grad_norms"], g["grad_samples"]):
        gr = P[n].grad
        mine_norm = 0.0 if gr is None else float(gr.double().norm())
        if ref_norm < 0:          # reference: grad is None (parameter unused or frozen in this phase)
            assert mine_norm == 0.0 or _frozen_in_phase(n, int(g["it"]), str(g["detector"]) if "detector" in g else "PSPNet"), n
            continue
        if ref_norm < 1e-7:       # exact zero, or the mathematically-zero grad of a conv bias feeding train-mode BN
            assert mine_norm < 1e-6, n
            continue
        # PReLU slopes are one scalar = a signed sum over ~1e6 products with heavy cancellation: two fp32
        # evaluation orders of the same math differ by 7e-3 typically and by up to 60% of a near-zero value vs fp64 (measured), so they get
        # their own bound (w^F case: reference 2.4e-4, fp64 oracle 3.4e-4, fp32 oracle 1.3e-3 on stage-0 down.conv's slope)
        tol, atol = (0.1, 2e-3) if gr.numel() == 1 else (TOL_GRAD, 1e-12)
        if hrnet and n.startswith("sr_model"):
            # beta = 0.9 pushes the segmentation gradient, back-propagated through ~300 BatchNorm layers, into KBPN: there the
            # reference's own fp32 result is noise-limited (measured against an fp64 run of the oracle: PReLU slopes 0.0083 (ref) /
            # 0.027 (fp32 oracle) / 0.021 (fp64); kb.sr_reconst weights 7 %; median over all tensors 5e-4)
            tol, atol = (1.0, 3e-2) if gr.numel() == 1 else (0.1, 1e-12)
        if bool(g.get("lr_error", False)) and "kb.sr_reconst" in n:
            # SUM_LR_ERROR_POS = 'LR': the gradient of the stages' 3-channel reconstruction weights is a heavily cancelling sum there; on
            # stage 2's the reference's own fp32 value (0.09876) is 4.0 % from an fp64 run of the oracle (0.10288), the fp32 oracle (0.10201)
            # 0.9 % -- every other tensor of the fixture agrees to <= 0.2 %
            tol = 6e-2
        if bool(g.get("zero_pad_kernel", False)) and ".kb.kernel_predictor.fe_SR.0." in n:
            # MODEL.ZERO_PAD_KERNEL: in the stages whose update is zero-padded the predictor's gradient is tiny (norm 3e-4) and noise-limited:
            # stage 2's fe_SR.0 weight is 2.4 % from an fp64 run of the oracle in the reference's own fp32 result, 0.9 % in the fp32 oracle's
            tol = 5e-2
        assert abs(mine_norm - ref_norm) <= tol * ref_norm + atol, (n, mine_norm, ref_norm)
        if gr.numel() == 1:
            n_checked += 1
            continue
        idx = sample_idx(n, gr.numel())
        assert np.allclose(gr.reshape(-1)[idx].numpy(), ref_s, rtol=1e-2, atol=(0.5 if hrnet else 0.25) * ref_norm / np.sqrt(gr.numel()) + 1e-9), n
        n_checked += 1
    blurskip = "detector" in g and str(g["detector"]) == "PSPNet_BlurSkip"
    assert n_checked >= (26 if blurskip else 1000 if hrnet else 150 if int(g["it"]) >= 30001 else 20)


def test_oracle_matches_reference_with_the_contractive_fill():
    """wc2_pspnet_it40000 (HR 256, B 2, detfill style "contractive"): pins the second fill rule -- the oracle regenerates the weights from
    names + shapes, the fixture was produced by the reference's own modules filled through the same rule -- and the oracle at a size where
    every BatchNorm sees >= 1e3 values per channel."""
    import zlib
    from csbsr_amd.data.synthetic import make_batch
    g = load_golden("wc2_pspnet_it40000")
    assert fill_style(g) == "contractive"
    cfg = golden_cfg(g)
    x, hr, mask, k = make_batch(int(g["B"]), int(g["lr"]), scale=cfg.scale, ksize=21, seed=int(g["seed"]))
    assert np.allclose(x.numpy(), g["x"], atol=1e-5)
    P = det_params(scale=cfg.scale, detector=cfg.detector, style="contractive")
    out = O.joint_forward(P, cfg, int(g["it"]), torch.from_numpy(g["x"]), hr, mask, torch.from_numpy(g["kernel"]), alpha=float(g["alpha"]))
    loss = O.calc_loss(out["segment_loss"], out["sr_loss"], int(g["it"]), cfg)
    loss.backward()
    for kk in ("segment_loss", "sr_loss", "segment_preds", "sr_preds", "kernel_preds"):
        assert rel_err(out[kk].detach(), g[kk]) < TOL_OUT, kk
    errs = []
    for n, ref_norm, smp in zip((str(v) for v in g["grad_names"]), g["grad_norms"], g["grad_samples32"]):
        gr = P[n].grad
        if ref_norm < 1e-7 or gr is None or gr.numel() == 1:
            continue
        flat = gr.reshape(-1)
        idx = [(zlib.crc32((n + str(j)).encode()) % flat.numel()) for j in range(32)]
        errs.append(max(abs(float(flat.double().norm()) - ref_norm) / ref_norm,
                        float(np.sqrt(np.mean((flat[idx].double().numpy() - smp.astype(np.float64)) ** 2)) / (ref_norm / np.sqrt(flat.numel())))))
    errs = np.array(errs)
    assert len(errs) > 200 and np.median(errs) < 2e-3 and errs.max() < TOL_GRAD, (np.median(errs), errs.max())


def _frozen_in_phase(name, it, detector="PSPNet"):
    """requires_grad=False phases of the reference (kbpn.py:118-142, 414-447): the oracle computes the
    gradient anyway; the harness (csbsr_amd) masks it.  Only used to excuse a non-zero oracle grad."""
    if detector == "PSPNet_BlurSkip":     # build_model.py:321-330: only blur_skip.* trains
        return "blur_skip" not in name
    if 1 <= it < 10001:
        return "kernel_predictor" in name or ".predictor." in name
    if 10001 <= it < 20001:
        return "kernel_predictor" not in name and ".predictor." not in name and name.startswith("sr_model")
    return False


def test_sdf_matches_reference_and_own_edt():
    g = load_golden("sdf_handdrawn")
    s1 = O.compute_sdf(g["mask"], use_scipy=True)
    s2 = O.compute_sdf(g["mask"], use_scipy=False)
    assert np.abs(s1 - g["sdf"]).max() < 1e-12
    assert np.abs(s2 - g["sdf"]).max() < 1e-9


def test_iou_identity():
    a = torch.rand(2, 1, 8, 8)
    assert torch.allclose(O.iou(a, a), torch.ones(2, 1, dtype=torch.double))
