"""HRNet-W48 + OCR detector alone (BASELINE config 4's segmentation half) through the C ABI vs the oracle, at a size where
BatchNorm is well conditioned (HR 192x192, B=4: branch maps 48/24/12/6, >= 144 values per channel; the OCR head's BatchNorm runs
over the B region vectors, so B=2 would make it a sign function).

The end-to-end fixture (tests/golden/e2e_hrnet_ocr_it40000, HR 64: stage-4 maps are 2x2) pins the forward; its gradients are
chaotic under ANY fp16-storage evaluation (the CPU emulation is 90 % away from fp32 there), so the backward is pinned here:
a fixed random linear functional of the two probability maps is back-propagated and every parameter gradient, plus the
gradient wrt the input image, is compared with autograd on the oracle.  Bounds are relative to what the fp16-storage
emulation of the oracle loses on the same input."""
import numpy as np
import pytest
import torch

from golden_utils import det_params, rel_err, max_rel_to_scale, fp16_storage_sim
from oracle import csbsr_oracle as O

pytestmark = pytest.mark.gpu


def _oracle(x, r1, r2, drop):
    P = det_params(detector="HRNet_OCR")
    xr = x.clone().requires_grad_(True)
    bn = O.BNState(P, True)
    seg, aux = O.hrnet_ocr_forward(P, xr, bn, drop)
    ((seg * r1).sum() + (aux * r2).sum()).backward()
    return P, seg.detach(), aux.detach(), xr.grad, bn.new


def test_hrnet_ocr_forward_backward_vs_oracle():
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
    with fp16_storage_sim():
        Ps, seg_s, aux_s, dx_s, bufs_s = _oracle(x, r1, r2, drop)

    cfg = base_cfg.clone()
    cfg.MODEL.DETECTOR_TYPE = "HRNet_OCR"
    m = JointModelWithLoss(cfg, 1000, 0, None)
    deterministic_fill(m.state_dict())
    m.train()
    rt = m._runtime()
    eng, net = rt["eng"], rt["psp"]
    gs = float(2 ** round(np.log2(B * H * H)))
    eng.grad_scale = gs
    xin = eng.nchw32_to_fm(x.cuda().contiguous())
    seg, aux = net.forward(xin, {"ocr_drop": keep.cuda().contiguous()}, True)
    dxin = net.backward((r1 * gs).cuda().contiguous(), (r2 * gs).cuda().contiguous())
    torch.cuda.synchronize()
    # forward
    for name, mine, ref, sim in (("seg", seg, seg_o, seg_s), ("aux", aux, aux_o, aux_s)):
        e, es = max_rel_to_scale(mine.cpu(), ref), max_rel_to_scale(sim, ref)
        print(f"{name}: hip {e:.2e}  emulation {es:.2e}")
        assert e < 2e-3 + 2.0 * es, (name, e, es)
    for k, v in bufs_o.items():
        if "running" in k and ("stage4.2" in k or "f_up" in k or "f_object" in k or "f_pixel" in k or "conv_bn_dropout" in k):
            assert max_rel_to_scale(rt["P"][k].cpu(), v) < 2e-2 + 3.0 * max_rel_to_scale(bufs_s[k], v), k
    # backward
    errs, sims, bad = [], [], []
    dx = dxin.t[..., :3].float().cpu().permute(0, 3, 1, 2) / gs
    items = [("d/dx", dx, dx_o, dx_s)]
    for k, p in P.items():
        if not k.startswith("segmentation_model") or getattr(p, "grad", None) is None:
            continue
        t = rt["P"][k]
        assert getattr(t, "gacc_touched", False), k
        items.append((k, t.gacc.cpu() / gs, p.grad, Ps[k].grad))
    zero = 0
    for k, mine, ref, sim in items:
        if float(ref.norm()) < 1e-7:          # conv biases feeding BatchNorm, f_pixel / f_object (dead for a single object region)
            assert float(mine.norm()) < 1e-5, k
            zero += 1
            continue
        e, es = rel_err(mine.reshape(ref.shape), ref), rel_err(sim, ref)
        errs.append(e)
        sims.append(es)
        bad.append((k, e, es))
    errs, sims = np.array(errs), np.array(sims)
    # independent realisations of the same chaotic noise (see tests/test_joint_gpu.py): loose per-tensor bound, distributions compared
    bad = [b for b in bad if b[1] > max(2.5 * b[2], 1.5 * sims.max()) + 3e-2]
    print("HRNet-OCR grads vs fp32 oracle: median %.2e p90 %.2e max %.2e (n=%d, %d exact zeros); emulation median %.2e p90 %.2e max %.2e"
          % (np.median(errs), np.percentile(errs, 90), errs.max(), len(errs), zero, np.median(sims), np.percentile(sims, 90), sims.max()))
    assert len(errs) > 900 and not bad, bad[:10]
    assert np.median(errs) < 1.5 * np.median(sims) + 5e-3
    assert np.percentile(errs, 90) < 1.5 * np.percentile(sims, 90) + 3e-2


def _fm(x):
    from csbsr_amd.engine import FM, pad8
    N, Cc, H, W = x.shape
    t = torch.zeros(N, H, W, pad8(Cc), dtype=torch.float16)
    t[..., :Cc] = x.permute(0, 2, 3, 1).half()
    return FM(t.cuda(), Cc)


def _from_fm(fm):
    return fm.t[..., :fm.c].float().cpu().permute(0, 3, 1, 2)


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


def _cmp_grads(rt, P, Ps, prefix, gs, tol):
    """per-tensor relative L2 error of the HIP gradients vs the fp32 oracle (P); bounded by ``tol`` plus twice what the fp16-storage
    emulation of the oracle (Ps) loses on the same tensor.  Mathematically-zero gradients (conv biases feeding BatchNorm, the dead
    f_pixel / f_object transforms) come out of fp32 autograd as ~1e-8 noise: they are checked to be ~0 instead."""
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
        e, es = rel_err(mine.reshape(ref.shape), ref), rel_err(Ps[k].grad, ref)
        assert e < tol + 2.0 * es, (k, e, es)
        errs.append(e)
    return np.array(errs), zero


def test_hr_module_vs_oracle():
    """One HighResolutionModule (stage 4: four branches x four BasicBlocks + the 4x4 fuse) forward and backward: shallow enough
    (12 BatchNorm layers deep) for a direct bound against the fp32 oracle."""
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
    with fp16_storage_sim():
        Ps, xs_s, _ = run()
    m, rt = _model()
    gs = 1024.0
    mod = _HRModule(rt["eng"], rt["P"], pre, chans)
    ys = mod.fwd([_fm(x) for x in xs], True)
    dxs = mod.bwd([_fm(r * gs) for r in rs])
    torch.cuda.synchronize()
    for i in range(4):
        assert max_rel_to_scale(_from_fm(ys[i]), ys_o[i].detach()) < 1e-2, i
        e, es = rel_err(_from_fm(dxs[i]) / gs, xr[i].grad), rel_err(xs_s[i].grad, xr[i].grad)
        print("d/dx%d: hip %.2e emulation %.2e" % (i, e, es))
        assert e < 1e-2 + 2.0 * es, (i, e, es)
    errs, zero = _cmp_grads(rt, P, Ps, pre + ".", gs, 1e-2)
    print("hr_module grads: median %.2e max %.2e (n=%d)" % (np.median(errs), errs.max(), len(errs)))
    # 16 BasicBlocks x 6 tensors + 16 fuse conv/BN units x 3; the 4-5 % is what fp16 storage costs here (emulation: same to 2 digits)
    assert len(errs) == 144 and np.median(errs) < 6e-2


def test_ocr_head_vs_oracle():
    """Everything after the 720-channel concat (aux head, 3x3 reduction, soft-region pooling, the region-vector chain with its
    BatchNorm over B vectors, fuse conv + dropout, class head, bilinear up + sigmoid), forward and backward."""
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
    with fp16_storage_sim():
        Ps = run()[0]
    m, rt = _model()
    net = rt["psp"]
    gs = float(2 ** round(np.log2(B * H * H)))
    seg, aux = net._head_fwd(_fm(feats), {"ocr_drop": keep.cuda().contiguous()}, True, H, H)
    dcat = net._head_bwd((r1 * gs).cuda().contiguous(), (r2 * gs).cuda().contiguous())
    torch.cuda.synchronize()
    print("head: seg %.2e aux %.2e dcat %.2e" % (max_rel_to_scale(seg.cpu(), seg_o.detach()), max_rel_to_scale(aux.cpu(), aux_o.detach()),
                                                rel_err(_from_fm(dcat) / gs, fr.grad)))
    assert max_rel_to_scale(seg.cpu(), seg_o.detach()) < 5e-3
    assert max_rel_to_scale(aux.cpu(), aux_o.detach()) < 5e-3
    assert rel_err(_from_fm(dcat) / gs, fr.grad) < 3e-2
    for k, v in bn.new.items():
        if "running" in k:
            assert max_rel_to_scale(rt["P"][k].cpu(), v) < 5e-3, k
    errs, zero = 0, 0
    for pre in ("segmentation_model.aux_head", "segmentation_model.conv3x3", "segmentation_model.ocr_distri_head", "segmentation_model.cls_head"):
        e, z = _cmp_grads(rt, P, Ps, pre + ".", gs, 1e-2)
        errs, zero = np.concatenate([np.atleast_1d(errs), e]) if isinstance(errs, np.ndarray) else e, zero + z
    print("head grads: median %.2e max %.2e (n=%d, %d exact zeros)" % (np.median(errs), errs.max(), len(errs), zero))
    assert np.median(errs) < 3e-2 and zero >= 16 + 4
