"""Generate golden vectors from the REFERENCE itself (imported from /root/reference with the shims in
ref_shims.py), run on CPU in the build container.  The reference has no tests/fixtures of its own for
this path, so these files are what pins the oracle (oracle/csbsr_oracle.py).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Weights: csbsr_amd.utils.detfill (closed form, keyed on state_dict name+shape); inputs:
csbsr_amd.data.synthetic.make_batch.  Only data is written: inputs, outputs, per-parameter gradient
L2 norms and a few sampled gradient elements.  torch version recorded in each file.
"""
import os
import sys
import zlib

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_shims  # noqa: E402

ref_shims.install()
from csbsr_amd.utils.detfill import deterministic_fill  # noqa: E402
from csbsr_amd.data.synthetic import make_batch  # noqa: E402


def grad_digest(model):
    names, norms, samples = [], [], []
    for n, p in model.named_parameters():
        names.append(n)
        if p.grad is None:
            norms.append(-1.0)
            samples.append(np.zeros(4, np.float32))
            continue
        g = p.grad.detach().reshape(-1)
        norms.append(float(g.double().norm()))
        idx = [(zlib.crc32((n + str(j)).encode()) % g.numel()) for j in range(4)]
        samples.append(g[idx].numpy().astype(np.float32))
    return names, np.array(norms), np.stack(samples)


class DropCapture:
    """Replace nn.Dropout2d.forward by a recorded/seeded channel mask (order of calls = keys)."""
    KEYS = ("drop_1", "drop_2a", "drop_2b", "drop_2c", "aux_drop")

    def __init__(self, enabled, seed=7, keys=None):
        self.KEYS = keys or self.KEYS
        self.enabled, self.gen, self.masks, self.i = enabled, torch.Generator().manual_seed(seed), {}, 0

    def __call__(self, mod, x):
        key = self.KEYS[self.i]
        self.i += 1
        if not self.enabled or not mod.training:
            self.masks[key] = None
            return x
        keep = (torch.rand(x.shape[0], x.shape[1], generator=self.gen) >= mod.p).float() / (1 - mod.p)
        self.masks[key] = keep
        return x * keep[:, :, None, None]


def run_case(name, it, B=2, lr=16, scale=4, dropout=False, antialias=True, alpha=None, taps=False, overrides=(), seed=1121,
             detector="PSPNet"):
    ref_shims.ANTIALIAS = antialias
    cfg, JM, J, FR = ref_shims.build_reference(detector=detector, scale=scale, overrides=overrides)
    model = JM(cfg, 1000, 0, FR(scale, "bicubic"))
    deterministic_fill(model)
    model.train()
    if alpha is not None:
        model.ss_loss_fn.alpha = alpha
    cap = DropCapture(dropout, keys=("ocr_drop",) if detector == "HRNet_OCR" else None)      # spatial_ocr_block.py:283
    orig = nn.Dropout2d.forward
    # PSPNet calls: drop_1, drop_2 x3 (main path) then aux dropout (pspnet.py:105-120)
    nn.Dropout2d.forward = lambda self, x: cap(self, x)
    tapd = {}
    hooks = []
    if taps:
        sm = model.sr_model

        def tap(key, pick=lambda o: o):
            def hook(mod, inp, out):
                tapd[key] = pick(out).detach()
            return hook
        hooks.append(sm.feat.register_forward_hook(tap("init_f")))
        for s, st in enumerate(sm.back_projection_stages, 1):
            hooks.append(st.kb.sr_reconst.register_forward_hook(tap(f"s{s}.sr_t")))
            hooks.append(st.kb.register_forward_hook(tap(f"s{s}.h", lambda o: o[0])))
            hooks.append(st.kb.register_forward_hook(tap(f"s{s}.kvec", lambda o: o[1][:, :, 0, 0])))
            if hasattr(st, "sft"):
                hooks.append(st.sft.register_forward_hook(tap(f"s{s}.low")))
    x, hr, mask, k = make_batch(B, lr, scale=scale, ksize=cfg.BLUR.KERNEL_SIZE_OUTPUT, seed=seed, antialias=antialias)
    try:
        seg_loss, sr_loss, seg, sr, kpred = model(it, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
        from model.engine.trainer import calc_loss
        import argparse
        loss, _, _ = calc_loss(seg_loss, 0.0, sr_loss, 0.0, it, cfg, argparse.Namespace())
        model.zero_grad()
        loss.backward()
    finally:
        nn.Dropout2d.forward = orig
        for h in hooks:
            h.remove()
    names, norms, samples = grad_digest(model)
    bufs = {n: b.detach().numpy() for n, b in model.named_buffers()
            if n in ("segmentation_model.feats.bn1.running_mean", "segmentation_model.feats.bn1.running_var",
                     "segmentation_model.up_3.conv.1.running_var", "segmentation_model.aux.1.running_mean",
                     "segmentation_model.blur_skip.3.norm.running_var",
                     "segmentation_model.backbone.bn1.running_mean", "segmentation_model.backbone.stage4.2.fuse_layers.3.2.0.1.running_var",
                     "segmentation_model.ocr_distri_head.object_context_block.f_object.3.0.running_var",
                     "segmentation_model.ocr_distri_head.object_context_block.f_pixel.3.0.running_mean",
                     "segmentation_model.ocr_distri_head.object_context_block.f_up.1.0.running_var",
                     "segmentation_model.aux_head.1.0.running_mean")}
    out = dict(x=x.numpy(), hr=hr.numpy(), mask=mask.numpy(), kernel=k.numpy(), it=np.int64(it),
               # (ONLY_KERNEL_LOSS_FOR_PRETRAIN returns the unreduced [B,1,K,K] kernel MSE map as "sr_loss" in the kernel pretraining phase,
               # sr_loss_functions.py:50-51; calc_loss takes its mean: stored as the per-sample means, whose mean is the same)
               segment_loss=seg_loss.detach().numpy(), sr_loss=sr_loss.detach().reshape(sr_loss.shape[0], -1).mean(1).numpy(), loss=np.float64(loss.item()),
               segment_preds=seg.detach().numpy(), sr_preds=sr.detach().numpy(), kernel_preds=kpred.detach().numpy(),
               grad_names=np.array(names), grad_norms=norms, grad_samples=samples,
               alpha=np.float64(model.ss_loss_fn.alpha), antialias=np.bool_(antialias), scale=np.int64(scale),
               detector=np.array(detector), sfo_sr_amp=np.float64(cfg.SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP),
               oriented_w_iter=np.int64(cfg.SOLVER.ORIENTED_WEIGHT_ITER), beta=np.float64(cfg.SOLVER.TASK_LOSS_WEIGHT),
               pixel_shuffle=np.bool_(cfg.MODEL.SR_PIXEL_SHUFFLE), torch_version=np.array(torch.__version__),
               residual_learning=np.bool_(cfg.MODEL.SR_RESIDUAL_LEARNING), only_kernel_loss=np.bool_(cfg.SOLVER.ONLY_KERNEL_LOSS_FOR_PRETRAIN),
               kernel_sft=np.bool_(cfg.MODEL.KBPN_KERNEL_SFT), lr_error=np.bool_(cfg.MODEL.SUM_LR_ERROR_POS == "LR"),
               zero_pad_kernel=np.bool_(cfg.MODEL.ZERO_PAD_KERNEL))
    for kname, v in cap.masks.items():
        if v is not None:
            out["dropmask." + kname] = v.numpy()
    for kname, v in bufs.items():
        out["buf." + kname] = v
    for kname, v in tapd.items():
        v = v.numpy()
        out["tap." + kname] = v if v.size <= 70000 else v[:, :8]          # first 8 channels of big maps
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={loss.item():.6f} seg={seg_loss.tolist()} sr={out['sr_loss'].tolist()} -> {os.path.getsize(path)/1e3:.0f} kB")


def _iou_min(a, b, th=0.5, smooth=1e-5):
    """estimate_metrics.IoU (thr 0.5, smooth 1e-5) between two probability maps, minimum over the batch"""
    pa, pb = (a > th).float(), (b > th).float()
    inter = (pa * pb).sum((1, 2, 3))
    union = pa.sum((1, 2, 3)) + pb.sum((1, 2, 3)) - inter
    return float(((inter + smooth) / (union + smooth)).min())


def run_case_wc(name, it, B=2, lr=64, scale=4, alpha=0.7, overrides=(), seed=1121, detector="PSPNet", dropout=False, eps=1e-3,
                fill="random"):
    """Fixtures at a well-conditioned size (HR >= 192: BatchNorm over >= 1e3 values per channel even at 1/32 resolution) that let
    the two halves of the path be checked against the reference SEPARATELY, with fixed bounds:
      * sr_preds (full fp32) + segment_preds + BN buffers + detector gradients  -> the detector fed the reference's own SR image;
      * dsr / dkvec = dLoss/d(sr_model outputs) of the reference + KBPN gradients -> the KBPN backward fed the reference's own
        upstream gradient;
    and the reference's own CONDITIONING: the same step re-run with the SR image moved by eps * max|sr| of seeded uniform noise
    (eps = 1e-3 = north_star's tolerance on the SR image), recording how far segment_preds / losses / BN buffers / gradients move.
    Inputs are regenerated from ``seed`` by csbsr_amd.data.synthetic.make_batch (checksums stored).
    ``fill``: csbsr_amd.utils.detfill style.  "contractive" (the wc2_* fixtures) gives the detector smooth low-gain filters, so it damps
    perturbations like a trained network instead of amplifying them ~100x: on those the COMPOSED path is held to tight fixed bounds."""
    ref_shims.ANTIALIAS = True
    cfg, JM, J, FR = ref_shims.build_reference(detector=detector, scale=scale, overrides=overrides)
    x, hr, mask, k = make_batch(B, lr, scale=scale, ksize=cfg.BLUR.KERNEL_SIZE_OUTPUT, seed=seed)
    from model.engine.trainer import calc_loss
    import argparse

    def one_pass(perturb_eps):
        model = JM(cfg, 1000, 0, FR(scale, "bicubic"))
        deterministic_fill(model, fill)
        model.train()
        model.ss_loss_fn.alpha = alpha
        cap_d = DropCapture(dropout, keys=("ocr_drop",) if detector == "HRNet_OCR" else None)
        orig = nn.Dropout2d.forward
        nn.Dropout2d.forward = lambda self, x_: cap_d(self, x_)
        held = {}
        sr_fwd = model.sr_model.forward

        def fwd(x_, it_, k_):
            sr, kp = sr_fwd(x_, it_, k_)
            if perturb_eps:
                g = torch.Generator().manual_seed(4242)
                sr = sr + (torch.rand(sr.shape, generator=g) * 2 - 1) * (perturb_eps * float(sr.detach().abs().max()))
            if sr.requires_grad:
                sr.retain_grad()
            if kp.requires_grad:
                kp.retain_grad()
            held["sr"], held["kp"] = sr, kp
            return sr, kp
        model.sr_model.forward = fwd
        try:
            seg_loss, sr_loss, seg, sr, kpred = model(it, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
            loss, _, _ = calc_loss(seg_loss, 0.0, sr_loss, 0.0, it, cfg, argparse.Namespace())
            model.zero_grad()
            loss.backward()
        finally:
            nn.Dropout2d.forward = orig
        grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in model.named_parameters()}
        bufs = {n: b.detach().clone() for n, b in model.named_buffers() if "running" in n and n.startswith("segmentation_model")}
        return dict(seg_loss=seg_loss.detach(), sr_loss=sr_loss.detach(), seg=seg.detach(), sr=sr.detach(), kpred=kpred.detach(),
                    loss=float(loss), grads=grads, bufs=bufs, masks=cap_d.masks, alpha=float(model.ss_loss_fn.alpha),
                    dsr=None if held["sr"].grad is None else held["sr"].grad.detach(),
                    dkp=None if held["kp"].grad is None else held["kp"].grad.detach())

    a = one_pass(0.0)
    b = one_pass(eps)
    mx = lambda u, v: float((u - v).abs().max() / (v.abs().max() + 1e-30))
    rl2 = lambda u, v: float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30))
    names, norms, samples = [], [], []
    cond_g = []
    for n, g in a["grads"].items():
        names.append(n)
        if g is None:
            norms.append(-1.0)
            samples.append(np.zeros(32, np.float32))
            continue
        flat = g.reshape(-1)
        norms.append(float(flat.double().norm()))
        idx = [(zlib.crc32((n + str(j)).encode()) % flat.numel()) for j in range(32)]
        samples.append(flat[idx].numpy().astype(np.float32))
        if flat.numel() > 1 and norms[-1] > 1e-12 and b["grads"][n] is not None:
            cond_g.append(rl2(b["grads"][n], g))
    cond_g = np.array(cond_g) if cond_g else np.zeros(1)
    # fp16 storage scale of dsr: the largest power of two that keeps max|dsr| * gs below 2^14 (HRNet-OCR's gradient at the SR image is
    # ~1e4x PSPNet's)
    gs = 1.0
    if a["dsr"] is not None:
        gs = float(2 ** int(np.floor(np.log2(16384.0 / max(float(a["dsr"].abs().max()), 1e-30)))))
    out = dict(x=x.numpy(), seed=np.int64(seed), B=np.int64(B), lr=np.int64(lr), it=np.int64(it),
               hr_sum=np.float64(hr.double().sum()), mask_sum=np.float64(mask.double().sum()), kernel=k.numpy(),
               segment_loss=a["seg_loss"].numpy(), sr_loss=a["sr_loss"].numpy(), loss=np.float64(a["loss"]),
               segment_preds=a["seg"].numpy(), sr_preds=a["sr"].numpy(), kernel_preds=a["kpred"].numpy(),
               grad_names=np.array(names), grad_norms=np.array(norms), grad_samples32=np.stack(samples),
               alpha=np.float64(a["alpha"]), antialias=np.bool_(True), scale=np.int64(scale),
               detector=np.array(detector), sfo_sr_amp=np.float64(cfg.SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP),
               oriented_w_iter=np.int64(cfg.SOLVER.ORIENTED_WEIGHT_ITER), beta=np.float64(cfg.SOLVER.TASK_LOSS_WEIGHT),
               torch_version=np.array(torch.__version__), fill=np.array(fill), pixel_shuffle=np.bool_(cfg.MODEL.SR_PIXEL_SHUFFLE),
               # the reference's own response to an SR image moved by eps * max|sr| (uniform noise)
               cond_eps=np.float64(eps), cond_seg_max=np.float64(mx(b["seg"], a["seg"])), cond_seg_l2=np.float64(rl2(b["seg"], a["seg"])),
               cond_segloss=np.float64(mx(b["seg_loss"], a["seg_loss"])), cond_iou=np.float64(_iou_min(b["seg"], a["seg"])),
               cond_bn=np.float64(max([mx(b["bufs"][n], a["bufs"][n]) for n in a["bufs"]] or [0.0])),
               cond_grad_median=np.float64(np.median(cond_g)), cond_grad_p90=np.float64(np.percentile(cond_g, 90)))
    if a["dsr"] is not None:      # upstream gradient of the KBPN backward, stored as fp16 of (grad * 2^20): the backward is linear in it
        out["dsr16"] = (a["dsr"] * gs).half().numpy()
        out["dsr_scale"] = np.float64(gs)
        out["dkvec"] = a["dkp"].sum(dim=(2, 3)).numpy() if a["dkp"] is not None else np.zeros((B, a["kpred"].shape[-1] ** 2), np.float32)
    for kname, v in a["masks"].items():
        if v is not None:
            out["dropmask." + kname] = v.numpy()
    for kname, v in a["bufs"].items():
        out["buf." + kname] = v.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={a['loss']:.6f} cond(eps={eps:g}): IoU {out['cond_iou']:.4f} seg max {out['cond_seg_max']:.2e} l2 {out['cond_seg_l2']:.2e} segloss "
          f"{out['cond_segloss']:.2e} bn {out['cond_bn']:.2e} grads median {out['cond_grad_median']:.2e} p90 {out['cond_grad_p90']:.2e}"
          f" -> {os.path.getsize(path)/1e6:.2f} MB")



def run_trajectory(name, steps=12, it0=40000, B=2, lr=64, scale=4, detector="PSPNet", overrides=(), seed0=2100, fill="contractive",
                   lr_rate=None):
    """``steps`` consecutive optimiser steps of the reference's training loop in the JOINT phase (trainer.py:57-72: zero_grad, forward,
    calc_loss, backward, Adam step -- train.py:91 hyper-parameters -- and the alpha schedule of trainer.py:495-508), a fresh synthetic batch
    per step (seed0 + step), Dropout2d masks seeded and recorded.  Stored per step: the two per-sample loss vectors, the scalar loss, alpha,
    the gradient L2 norm per gradient bucket (segmentation net, KBPN stage 1..S, KBPN head) and the L2 distance the parameters have
    moved from the start; plus a checksum of the final weights.  The HIP path must reproduce the CURVE (tests/test_trajectory_gpu.py)."""
    ref_shims.ANTIALIAS = True
    cfg, JM, J, FR = ref_shims.build_reference(detector=detector, scale=scale, overrides=overrides)
    from model.engine.trainer import calc_loss
    import argparse
    model = JM(cfg, 1000, 0, FR(scale, "bicubic"))
    deterministic_fill(model, fill)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=float(cfg.SOLVER.LR) if lr_rate is None else lr_rate, betas=(0.9, 0.999), eps=1e-8)
    start = {n: p.detach().clone() for n, p in model.named_parameters()}
    S = cfg.MODEL.NUM_STAGES

    def bucket(n):
        if n.startswith("segmentation_model"):
            return 0
        if n.startswith("sr_model.back_projection_stages."):
            return int(n.split(".")[2]) + 1
        if n.startswith("sr_model.output_conv"):
            return S
        return S + 1
    rec = dict(seg_loss=[], sr_loss=[], loss=[], alpha=[], gnorm=[], moved=[])
    masks_all = {}
    orig = nn.Dropout2d.forward
    for step in range(steps):
        it = it0 + step
        model.ss_loss_fn.fix_alpha = False            # trainer.py:501-508 (not the SR-pretrain phase)
        model.ss_loss_fn.update_alpha()
        x, hr, mask, k = make_batch(B, lr, scale=scale, ksize=cfg.BLUR.KERNEL_SIZE_OUTPUT, seed=seed0 + step)
        cap = DropCapture(True, seed=7 + step, keys=("ocr_drop",) if detector == "HRNet_OCR" else None)
        nn.Dropout2d.forward = lambda self, x_: cap(self, x_)
        try:
            opt.zero_grad()
            seg_loss, sr_loss, seg, sr, kpred = model(it, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
            loss, _, _ = calc_loss(seg_loss, 0.0, sr_loss, 0.0, it, cfg, argparse.Namespace())
            loss.backward()
        finally:
            nn.Dropout2d.forward = orig
        gn = np.zeros(S + 2)
        for n, p in model.named_parameters():
            if p.grad is not None:
                gn[bucket(n)] += float(p.grad.double().pow(2).sum())
        opt.step()
        mv = np.sqrt(sum(float((p.detach() - start[n]).double().pow(2).sum()) for n, p in model.named_parameters()))
        rec["seg_loss"].append(seg_loss.detach().numpy().copy()); rec["sr_loss"].append(sr_loss.detach().numpy().copy())
        rec["loss"].append(float(loss)); rec["alpha"].append(float(model.ss_loss_fn.alpha)); rec["gnorm"].append(np.sqrt(gn)); rec["moved"].append(mv)
        for kname, v in cap.masks.items():
            if v is not None:
                masks_all[f"dropmask.{step}.{kname}"] = v.numpy()
        print(f"  {name} step {step}: loss {float(loss):.6f} seg {seg_loss.mean():.6f} sr {sr_loss.mean():.6f} alpha {model.ss_loss_fn.alpha:.3f} "
              f"|g| {np.sqrt(gn).round(5).tolist()} moved {mv:.5f}", flush=True)
    final = {n: p.detach() for n, p in model.named_parameters()}
    names = list(final)
    out = dict(steps=np.int64(steps), it0=np.int64(it0), B=np.int64(B), lr=np.int64(lr), scale=np.int64(scale), seed0=np.int64(seed0),
               detector=np.array(detector), fill=np.array(fill), lr_rate=np.float64(opt.param_groups[0]["lr"]), beta=np.float64(cfg.SOLVER.TASK_LOSS_WEIGHT),
               antialias=np.bool_(True), sfo_sr_amp=np.float64(cfg.SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP),
               oriented_w_iter=np.int64(cfg.SOLVER.ORIENTED_WEIGHT_ITER),
               seg_loss=np.stack(rec["seg_loss"]), sr_loss=np.stack(rec["sr_loss"]), loss=np.array(rec["loss"]), alpha=np.array(rec["alpha"]),
               gnorm=np.stack(rec["gnorm"]), moved=np.array(rec["moved"]), torch_version=np.array(torch.__version__),
               final_names=np.array(names), final_delta_norm=np.array([float((final[n] - start[n]).double().norm()) for n in names]),
               final_delta_dot=np.array([float(((final[n] - start[n]).double() * torch.sign(final[n] - start[n]).double()).sum()) for n in names]))
    out.update(masks_all)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {steps} steps -> {os.path.getsize(path)/1e3:.0f} kB")


def aux_cases():
    """Fixtures of the rows either side of the hot path, every value produced by the reference's own classes:
    degradation (GaussianBlur.make + conv_kernel2d + FactorResize), SplitPatch / JointPatch, PSNR / SSIM / the 99-threshold IoU
    sweep, and the key / shape / checksum digest of a checkpoint saved from the DataParallel-wrapped reference model."""
    import model.data.blur.blur as RB
    from model.data.transforms.transforms import FactorResize
    from model.data.samplers.patch_sampler import SplitPatch, JointPatch
    from model.utils.estimate_metrics import PSNR, SSIM, IoU
    out = {}
    # ---- f1: degradation with scripted draws (theta from torch.rand, sigmas from np.random.rand: blur.py:128,160-167)
    ref_shims.ANTIALIAS = True
    g = torch.Generator().manual_seed(31)
    hr = torch.rand(2, 3, 48, 64, generator=g)
    draws = [(0.37, 0.81, 0.12), (0.05, 0.93, 0.66)]          # (theta_u, sigma_x_u, sigma_y_u) in [0,1)
    kernels, lrs, params = [], [], []
    t_rand, n_rand = torch.rand, np.random.rand
    try:
        for b, (tu, su, sv) in enumerate(draws):
            seq = iter([su, sv])
            torch.rand = lambda *a, **k: torch.tensor([tu])
            np.random.rand = lambda *a: next(seq)
            gb = RB.GaussianBlur(size=21, isotropic=False, device="cpu", range_deterioration_ratio=(0.2, 4))
            kern = gb.make()
            torch.rand, np.random.rand = t_rand, n_rand
            img = RB.conv_kernel2d(hr[b], kern, device="cpu")
            lrs.append(FactorResize(4, "bicubic")(img).numpy())
            kernels.append(kern.numpy())
            params.append([0.2 + 3.8 * su, 0.2 + 3.8 * sv, (180.0 * tu) * np.pi / 180])
    finally:
        torch.rand, np.random.rand = t_rand, n_rand
    out.update(deg_hr=hr.numpy(), deg_params=np.array(params, np.float32), deg_kernels=np.stack(kernels), deg_lr=np.stack(lrs))
    # ---- f2: patch sampler (index permutation) and the threshold sweep
    x = torch.arange(3 * 12 * 20, dtype=torch.float32).reshape(3, 12, 20)
    patches, shape = SplitPatch(1, 3, 6, 5)(x)
    out.update(patch_in=x.numpy(), patch_out=patches.numpy(), patch_shape=np.array(shape))
    two = torch.stack([x, x + 1000.0])                       # JointPatch on a batch of two images' patches
    pp = torch.cat([SplitPatch(1, 3, 6, 5)(two[i])[0] for i in range(2)])
    out.update(joint_out=JointPatch()(pp, shape).numpy())
    pred = torch.rand(2, 1, 40, 52, generator=g)
    mask = (torch.rand(2, 1, 40, 52, generator=g) > 0.7).float()
    thresholds = [i * 0.01 for i in range(1, 100)]
    tmap = torch.Tensor(thresholds).view(len(thresholds), 1, 1)
    bi = (pred - tmap > torch.Tensor([0])).float()
    out.update(iou_pred=pred.numpy(), iou_mask=mask.numpy(), iou_sweep=np.reshape(IoU()(bi, mask), (2, -1)))
    # ---- f4: PSNR / SSIM
    a = torch.rand(2, 3, 40, 52, generator=g)
    b_ = (a + 0.05 * torch.randn(2, 3, 40, 52, generator=g)).clamp(0, 1)
    out.update(met_a=a.numpy(), met_b=b_.numpy(), met_psnr=PSNR()(a, b_), met_ssim=SSIM()(a, b_))
    # ---- f3: checkpoint digest, as trainer.py:117-131 saves it from the DataParallel-wrapped model
    cfg, JM, J, FR = ref_shims.build_reference(detector="PSPNet", scale=4)
    model = JM(cfg, 1000, 0, FR(4, "bicubic"))
    deterministic_fill(model)
    sd = nn.DataParallel(model).state_dict()
    keys = list(sd.keys())
    out.update(ckpt_keys=np.array(keys), ckpt_shapes=np.array([",".join(str(d) for d in v.shape) for v in sd.values()]),
               ckpt_sum=np.array([float(v.double().sum()) for v in sd.values()]),
               ckpt_abs=np.array([float(v.double().abs().sum()) for v in sd.values()]))
    np.savez_compressed(os.path.join(HERE, "aux_reference.npz"), **out)
    print("aux_reference written:", {k: getattr(v, "shape", None) for k, v in out.items() if not k.startswith("ckpt")}, len(keys), "checkpoint keys")


def sdf_case():
    from model.utils.boundary_loss import compute_sdf1_1
    m = np.zeros((3, 1, 40, 56), np.uint8)
    m[0, 0, 5:9, 3:50] = 1
    m[0, 0, 9:30, 20:23] = 1
    m[0, 0, 0:3, 53:56] = 1                      # touches the image corner
    m[1, 0, 17, 11] = 1                          # single pixel
    yy, xx = np.mgrid[:40, :56]
    m[2, 0] = ((yy - 20) ** 2 + (xx - 30) ** 2 < 90).astype(np.uint8)
    m[2, 0, 18:22, 28:32] = 0                    # hole
    sdf = compute_sdf1_1(m.astype(np.float32), m.shape)
    np.savez_compressed(os.path.join(HERE, "sdf_handdrawn.npz"), mask=m, sdf=sdf)
    print("sdf_handdrawn written")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "--only-new" in sys.argv:        # keep the committed fixtures, add the missing ones
        _rc = run_case

        def run_case(name, *a, **k):     # noqa: F811
            if not os.path.exists(os.path.join(HERE, name + ".npz")):
                _rc(name, *a, **k)
    else:
        sdf_case()
    if "--aux" in sys.argv:
        aux_cases()
        sys.exit(0)
    if "--wc2" in sys.argv:         # well-conditioned size AND a contractive detector fill: the composed path under tight fixed bounds
        only = [a for a in sys.argv[1:] if not a.startswith("--")]
        todo = {
            "wc2_pspnet_it40000": lambda n: run_case_wc(n, 40000, B=2, lr=64, fill="contractive"),
            "wc2_blurskip_x8_it40000": lambda n: run_case_wc(n, 40000, B=2, lr=32, scale=8, alpha=0.8, detector="PSPNet_BlurSkip", seed=9, fill="contractive",
                                                              overrides=("SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP", 1.0, "SOLVER.ORIENTED_WEIGHT_ITER", 0)),
            "wc2_hrnet_ocr_it40000": lambda n: run_case_wc(n, 40000, B=4, lr=48, alpha=0.8, detector="HRNet_OCR", seed=13, dropout=True, fill="contractive",
                                                            overrides=("SOLVER.TASK_LOSS_WEIGHT", 0.9)),
            "wc2_pspnet_pixelshuffle_it40000": lambda n: run_case_wc(n, 40000, B=2, lr=64, seed=17, fill="contractive",
                                                                      overrides=("MODEL.SR_PIXEL_SHUFFLE", True)),
            "traj_pspnet_it40000": lambda n: run_trajectory(n, steps=12, lr=64, B=2),
        }
        for n, fn in todo.items():
            if not only or n in only:
                fn(n)
        sys.exit(0)
    if "--wc" in sys.argv:          # the well-conditioned-size fixtures only (minutes of CPU each)
        run_case_wc("wc_pspnet_it40000", 40000, B=2, lr=64)
        run_case_wc("wc_blurskip_x8_it40000", 40000, B=2, lr=32, scale=8, alpha=0.8, detector="PSPNet_BlurSkip", seed=9,
                    overrides=("SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP", 1.0, "SOLVER.ORIENTED_WEIGHT_ITER", 0))
        run_case_wc("wc_hrnet_ocr_it40000", 40000, B=4, lr=48, alpha=0.8, detector="HRNet_OCR", seed=13, dropout=True,
                    overrides=("SOLVER.TASK_LOSS_WEIGHT", 0.9))
        sys.exit(0)
    if "--variants" in sys.argv:      # (only) the KBPN structure variants: no SFT layer between the stages (kbpn.py:169-171,190); the
        # back-projection error added to the next stage's LR features instead of this stage's HR features (kbpn.py:174-187,369-374,404-409)
        run_case("e2e_pspnet_nosft_it40000", 40000, alpha=0.7, seed=29, overrides=("MODEL.KBPN_KERNEL_SFT", False))
        run_case("e2e_pspnet_lrerr_it40000", 40000, alpha=0.7, seed=31, overrides=("MODEL.SUM_LR_ERROR_POS", "LR"))
        run_case("e2e_pspnet_lrerr_it1", 1, seed=37, overrides=("MODEL.SUM_LR_ERROR_POS", "LR"))
        sys.exit(0)
    if "--zeropad" in sys.argv:       # MODEL.ZERO_PAD_KERNEL (kbpn.py:543-554,583-596).  The pad discriminator's two nn.Dropout(0.2) layers are
        # disabled while the fixture is made: in the reference's train mode they make the HARD per-sample decision (p.item() >= 0.5)
        # itself random, so no implementation could reproduce a fixture made with them on
        orig_do = nn.Dropout.forward
        nn.Dropout.forward = lambda self, x: x
        try:
            # (with the deterministic fill the discriminators of stages 1-2 pick the bicubic map and those of stages 3-4 the zero padding)
            run_case("e2e_pspnet_zeropad_it40000", 40000, alpha=0.7, seed=41, overrides=("MODEL.ZERO_PAD_KERNEL", True))
        finally:
            nn.Dropout.forward = orig_do
        sys.exit(0)
    run_case("e2e_pspnet_it40000", 40000, taps=True, alpha=0.7)
    run_case("e2e_pspnet_it40000_dropout", 40000, dropout=True, alpha=0.7)
    run_case("e2e_pspnet_it1", 1)
    run_case("e2e_pspnet_it10001", 10001)
    run_case("e2e_pspnet_it20001", 20001)
    run_case("e2e_pspnet_it40000_noaa", 40000, antialias=False, alpha=0.7)
    run_case("e2e_pspnet_it40000_lr24", 40000, lr=24, B=1, alpha=0.9, seed=5)
    # BASELINE config 5: x8, PSPNet_BlurSkip, w^F (m^F = 1) on the SR loss, only blur_skip trainable
    run_case("e2e_blurskip_x8_it40000", 40000, lr=8, scale=8, alpha=0.8, detector="PSPNet_BlurSkip", seed=9,
             overrides=("SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP", 1.0, "SOLVER.ORIENTED_WEIGHT_ITER", 0))
    # BASELINE config 4: HRNet-W48 + OCR detector, beta = 0.9, batch 4 (BatchNorm over the 4 object-context vectors)
    run_case("e2e_hrnet_ocr_it40000", 40000, B=4, alpha=0.8, detector="HRNet_OCR", seed=13, dropout=True,
             overrides=("SOLVER.TASK_LOSS_WEIGHT", 0.9))
    # MODEL.SR_PIXEL_SHUFFLE = True (north_star's "PixelShuffle x4 upsample"): conv3x3 + PixelShuffle in place of the four deconvs per stage
    run_case("e2e_pspnet_pixelshuffle_it20001", 20001, seed=17, overrides=("MODEL.SR_PIXEL_SHUFFLE", True))
    # config 2 with the w^F weight on (README row 'CSBSR w/ PSPNet + w^F')
    run_case("e2e_pspnet_wf_it40000", 40000, alpha=0.7, seed=11,
             overrides=("SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP", 1.0, "SOLVER.ORIENTED_WEIGHT_ITER", 0))
    # cfg variants of the KBPN path: no bicubic residual (kbpn.py:112-116); kernel-MSE-only loss in the kernel pretraining phase
    # (sr_loss_functions.py:50-51, with a non-zero kernel weight in the other phases)
    run_case("e2e_pspnet_noresidual_it40000", 40000, alpha=0.7, seed=19, overrides=("MODEL.SR_RESIDUAL_LEARNING", False))
    run_case("e2e_pspnet_konly_it10001", 10001, seed=23, overrides=("SOLVER.ONLY_KERNEL_LOSS_FOR_PRETRAIN", True))
