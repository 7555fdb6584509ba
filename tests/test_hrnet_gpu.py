"""HRNet-W48 + OCR detector alone (BASELINE config 4's segmentation half) through the C ABI vs the oracle, at a size where
BatchNorm is well conditioned (HR 192x192, B=4: branch maps 48/24/12/6, >= 144 values per channel; the OCR head's BatchNorm runs
over the B region vectors, so B=2 would make it a sign function).

A fixed random linear functional of the two probability maps is back-propagated and every parameter gradient, plus the gradient
wrt the input image, is compared with autograd on the oracle, with FIXED bounds:
  * detector_precision = "split" (hi + lo fp16 forward): outputs and BatchNorm buffers 1e-3, gradients 3e-2 per tensor -- the
    ~300-layer random-weight BatchNorm stack amplifies perturbations ~500x (tests/golden/wc_hrnet_ocr: cond_seg_max 0.55 for a 1e-3
    input perturbation), so this is only reachable with ~22-bit forward operands;
  * plain fp16 storage: the same code path under loose fixed sanity bounds (its error is that amplification times 2^-11)."""
import numpy as np
import pytest
import torch

from golden_utils import det_params, rel_err, max_rel_to_scale
from oracle import csbsr_oracle as O

pytestmark = pytest.mark.gpu


def _oracle(x, r1, r2, drop):
    P = det_params(detector="HRNet_OCR")
    xr = x.clone().requires_grad_(True)
    bn = O.BNState(P, True)
    seg, aux = O.hrnet_ocr_forward(P, xr, bn, drop)
    ((seg * r1).sum() + (aux * r2).sum()).backward()
    return P, seg.detach(), aux.detach(), xr.grad, bn.new


@pytest.mark.parametrize("precision", ["split", "fp16"])
def test_hrnet_ocr_forward_backward_vs_oracle(precision):
    from csbsr_amd import _lib as L
    from csbsr_amd.config import cfg as base_cfg
    from csbsr_amd.engine import FM, pad8
    from csbsr_amd.modeling.build_model import JointModelWithLoss
    from csbsr_amd.utils.detfill import deterministic_fill
    torch.manual_seed(5)
    B, H = 4, 192
    x = torch.randn(B, 3, H, H)
    r1, r2 = torch.randn(B, 1, H, H) / (H * H), 0.4 * torch.randn(B, 1, H, H) / (H * H)
    keep = (torch.rand(B, 512) >= 0.05).float() / 0.95
    drop = {"ocr_drop": keep}
    P, seg_o, aux_o, dx_o, bufs_o = _oracle(x, r1, r2, drop)
    split = precision == "split"

    cfg = base_cfg.clone()
    cfg.MODEL.DETECTOR_TYPE = "HRNet_OCR"
    m = JointModelWithLoss(cfg, 1000, 0, None)
    deterministic_fill(m.state_dict())
    m.train()
    m.detector_precision = precision
    rt = m._runtime()
    eng, net = rt["eng"], rt["psp"]
    gs = float(2 ** round(np.log2(B * H * H)))
    eng.grad_scale = gs
    xin = eng.nchw32_to_fm(x.cuda().contiguous(), split=split)
    seg, aux = net.forward(xin, {"ocr_drop": keep.cuda().contiguous()}, True)
    dxin = net.backward((r1 * gs).cuda().contiguous(), (r2 * gs).cuda().contiguous())
    torch.cuda.synchronize()
    # forward
    e_out = {name: max_rel_to_scale(mine.cpu(), ref) for name, mine, ref in (("seg", seg, seg_o), ("aux", aux, aux_o))}
    e_bn = max(max_rel_to_scale(rt["P"][k].cpu(), v) for k, v in bufs_o.items() if "running" in k)
    print(f"[{precision}] seg {e_out['seg']:.2e} aux {e_out['aux']:.2e} BN buffers {e_bn:.2e}")
    if split:
        assert e_out["seg"] < 1e-3 and e_out["aux"] < 1e-3 and e_bn < 1e-3
    else:
        assert e_out["seg"] < 0.5 and e_out["aux"] < 0.5 and e_bn < 0.3
    # backward
    errs, bad = [], []
    dx = dxin.t[..., :3].float().cpu().permute(0, 3, 1, 2) / gs
    items = [("d/dx", dx, dx_o)]
    for k, p in P.items():
        if not k.startswith("segmentation_model") or getattr(p, "grad", None) is None:
            continue
        t = rt["P"][k]
        assert getattr(t, "gacc_touched", False), k
        items.append((k, t.gacc.cpu() / gs, p.grad))
    zero = 0
    gmax = max(float(ref.norm()) for _, _, ref in items)
    for k, mine, ref in items:
        if float(ref.norm()) < 1e-7 * max(gmax, 1.0):          # conv biases feeding BatchNorm, f_pixel / f_object (dead for a single object region)
            assert float(mine.norm()) < 1e-5 * max(gmax, 1.0), k
            zero += 1
            continue
        e = rel_err(mine.reshape(ref.shape), ref)
        errs.append(e)
        if ref.numel() > 1 and e > 3e-2:
            bad.append((k, e))
    errs = np.array(errs)
    print("[%s] HRNet-OCR grads vs fp32 oracle: median %.2e p90 %.2e max %.2e (n=%d, %d exact zeros)"
          % (precision, np.median(errs), np.percentile(errs, 90), errs.max(), len(errs), zero))
    assert len(errs) > 900 and np.isfinite(errs).all()
    if split:
        assert not bad, (len(bad), bad[:10])
    else:
        assert np.median(errs) < 1.5


def _fm(x, split=False):
    from csbsr_amd.engine import FM, pad8
    N, Cc, H, W = x.shape
    cp = pad8(Cc)
    t = torch.zeros(N, H, W, 2 * cp if split else cp, dtype=torch.float16)
    hi = x.permute(0, 2, 3, 1).half()
    t[..., :Cc] = hi
    if split:
        t[..., cp:cp + Cc] = (x.permute(0, 2, 3, 1) - hi.float()).half()
        return FM(t.cuda()[..., :cp], Cc, lo=cp)
    return FM(t.cuda(), Cc)


def _from_fm(fm):
    return fm.t[..., :fm.c].float().cpu().permute(0, 3, 1, 2)


def _from_fm_full(fm):
    """value of a (possibly split hi + lo) map"""
    v = fm.t[..., :fm.c].float()
    if fm.lo:
        v = v + fm.t.as_strided(fm.t.shape, fm.t.stride(), fm.t.storage_offset() + fm.lo)[..., :fm.c].float()
    return v.cpu().permute(0, 3, 1, 2)


def _model():
    from csbsr_amd.config import cfg as base_cfg
    from csbsr_amd.modeling.build_model import JointModelWithLoss
    from csbsr_amd.utils.detfill import deterministic_fill
    cfg = base_cfg.clone()
    cfg.MODEL.DETECTOR_TYPE = "HRNet_OCR"
    m = JointModelWithLoss(cfg, 1000, 0, None)
    deterministic_fill(m.state_dict())
    m.train()
    return m, m._runtime()


def _cmp_grads(rt, P, prefix, gs, tol):
    """per-tensor relative L2 error of the HIP gradients vs the fp32 oracle (P), bounded by ``tol``.  Mathematically-zero gradients
    (conv biases feeding BatchNorm, the dead f_pixel / f_object transforms) come out of fp32 autograd as ~1e-8 noise: they are
    checked to be ~0 instead."""
    names = [k for k, p in P.items() if k.startswith(prefix) and getattr(p, "grad", None) is not None]
    gmax = max(float(P[k].grad.norm()) for k in names)
    errs, zero = [], 0
    for k in names:
        t = rt["P"][k]
        assert getattr(t, "gacc_touched", False), k
        mine = t.gacc.cpu() / gs
        ref = P[k].grad
        if float(ref.norm()) < 1e-5 * gmax:
            assert float(mine.norm()) < 1e-4 * gmax, k
            zero += 1
            continue
        e = rel_err(mine.reshape(ref.shape), ref)
        assert e < tol or ref.numel() == 1, (k, e)
        errs.append(e)
    return np.array(errs), zero


@pytest.mark.parametrize("split", [True, False])
def test_hr_module_vs_oracle(split):
    """One HighResolutionModule (stage 4: four branches x four BasicBlocks + the 4x4 fuse) forward and backward against the fp32
    oracle: split-fp16 forward 1e-3 on the outputs and 3e-2 on every gradient; plain fp16 storage 1e-2 / 0.2 (12 BatchNorm layers)."""
    from csbsr_amd.modeling.hrnet_ocr import _HRModule
    torch.manual_seed(6)
    chans, sizes, B = (48, 96, 192, 384), (48, 24, 12, 6), 2
    pre = "segmentation_model.backbone.stage4.1"
    xs = [torch.relu(torch.randn(B, c, s, s)).half().float() for c, s in zip(chans, sizes)]
    rs = [torch.randn(B, c, s, s).half().float() for c, s in zip(chans, sizes)]
    def run():
        P = det_params(detector="HRNet_OCR")
        xr = [x.clone().requires_grad_(True) for x in xs]
        ys_o = O.hr_module(P, O.BNState(P, True), pre, xr)
        sum((y * r).sum() for y, r in zip(ys_o, rs)).backward()
        return P, xr, ys_o
    P, xr, ys_o = run()
    m, rt = _model()
    gs = 1024.0
    mod = _HRModule(rt["eng"], rt["P"], pre, chans)
    ys = mod.fwd([_fm(x, split) for x in xs], True)
    dxs = mod.bwd([_fm(r * gs) for r in rs])
    torch.cuda.synchronize()
    tol_y, tol_g = (1e-3, 3e-2) if split else (1e-2, 0.25)
    for i in range(4):
        e_y = max_rel_to_scale(_from_fm_full(ys[i]), ys_o[i].detach())
        e = rel_err(_from_fm(dxs[i]) / gs, xr[i].grad)
        print("branch %d [%s]: y %.2e d/dx %.2e" % (i, "split" if split else "fp16", e_y, e))
        assert e_y < tol_y and e < tol_g, (i, e_y, e)
    errs, zero = _cmp_grads(rt, P, pre + ".", gs, tol_g)
    print("hr_module grads: median %.2e max %.2e (n=%d)" % (np.median(errs), errs.max(), len(errs)))
    # 16 BasicBlocks x 6 tensors + 16 fuse conv/BN units x 3
    assert len(errs) == 144 and np.median(errs) < (5e-3 if split else 6e-2)


@pytest.mark.parametrize("split", [True, False])
def test_ocr_head_vs_oracle(split):
    """Everything after the 720-channel concat (aux head, 3x3 reduction, soft-region pooling, the region-vector chain with its
    BatchNorm over B vectors, fuse conv + dropout, class head, bilinear up + sigmoid), forward and backward, fixed bounds."""
    torch.manual_seed(7)
    B, h, H = 4, 24, 96
    feats = torch.relu(torch.randn(B, 720, h, h)).half().float()
    r1, r2 = torch.randn(B, 1, H, H) / (H * H), 0.4 * torch.randn(B, 1, H, H) / (H * H)
    keep = (torch.rand(B, 512) >= 0.05).float() / 0.95
    def run():
        P = det_params(detector="HRNet_OCR")
        fr = feats.clone().requires_grad_(True)
        bn = O.BNState(P, True)
        seg_o, aux_o = O.hrnet_ocr_head(P, fr, bn, {"ocr_drop": keep}, (H, H))
        ((seg_o * r1).sum() + (aux_o * r2).sum()).backward()
        return P, fr, bn, seg_o, aux_o
    P, fr, bn, seg_o, aux_o = run()
    m, rt = _model()
    net = rt["psp"]
    gs = float(2 ** round(np.log2(B * H * H)))
    seg, aux = net._head_fwd(_fm(feats, split), {"ocr_drop": keep.cuda().contiguous()}, True, H, H)
    dcat = net._head_bwd((r1 * gs).cuda().contiguous(), (r2 * gs).cuda().contiguous())
    torch.cuda.synchronize()
    print("head: seg %.2e aux %.2e dcat %.2e" % (max_rel_to_scale(seg.cpu(), seg_o.detach()), max_rel_to_scale(aux.cpu(), aux_o.detach()),
                                                rel_err(_from_fm(dcat) / gs, fr.grad)))
    tol = 1e-3 if split else 5e-3
    assert max_rel_to_scale(seg.cpu(), seg_o.detach()) < tol
    assert max_rel_to_scale(aux.cpu(), aux_o.detach()) < tol
    assert rel_err(_from_fm(dcat) / gs, fr.grad) < 3e-2
    for k, v in bn.new.items():
        if "running" in k:
            assert max_rel_to_scale(rt["P"][k].cpu(), v) < tol, k
    errs, zero = 0, 0
    for pre in ("segmentation_model.aux_head", "segmentation_model.conv3x3", "segmentation_model.ocr_distri_head", "segmentation_model.cls_head"):
        e, z = _cmp_grads(rt, P, pre + ".", gs, 3e-2 if split else 0.1)
        errs, zero = np.concatenate([np.atleast_1d(errs), e]) if isinstance(errs, np.ndarray) else e, zero + z
    print("head grads: median %.2e max %.2e (n=%d, %d exact zeros)" % (np.median(errs), errs.max(), len(errs), zero))
    assert np.median(errs) < 3e-2 and zero >= 16 + 4
