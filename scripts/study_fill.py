"""CPU study (no GPU): conditioning of the detector under the two deterministic fills (csbsr_amd/utils/detfill.py).
    python scripts/study_fill.py [lr] [B] [detector]
For each fill: (a) fp16-storage emulation of the whole path vs the fp32 oracle, (b) the fp32 oracle's response to a 1e-3 uniform
perturbation of its SR image.  Prints max-rel error of the segmentation map, IoU and the gradient-error distribution."""
import sys, time, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from golden_utils import fp16_storage_sim, rel_err, max_rel_to_scale
from oracle import csbsr_oracle as O
from csbsr_amd.data.synthetic import make_batch
from csbsr_amd.utils.detfill import det_state_dict
from csbsr_amd.modeling.shapes import joint_state_shapes

lr = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
det = sys.argv[3] if len(sys.argv) > 3 else "PSPNet"
scale = 8 if det == "PSPNet_BlurSkip" else 4
torch.set_num_threads(8)
cfg = O.PathCfg(scale=scale, detector=det, beta=0.9 if det == "HRNet_OCR" else 0.3)
x, hr, mask, k = make_batch(B, lr, scale=scale, ksize=21, seed=1121)
it = 40000
GS = float(2 ** 20)


def params(style):
    sd = det_state_dict(joint_state_shapes(scale=scale, detector=det), style)
    for kk, v in sd.items():
        if v.is_floating_point() and not kk.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    return sd


def run(style, plan, perturb=0.0):
    P = params(style)
    t0 = time.time()
    if plan == "fp16":
        with fp16_storage_sim():
            out = O.joint_forward(P, cfg, it, x, hr, mask, k, alpha=0.7)
            loss = O.calc_loss(out["segment_loss"], out["sr_loss"], it, cfg)
            (loss * GS).backward()
        sc = 1.0 / GS
    else:
        if perturb:
            kf = O.kbpn_forward
            def kfp(*a, **kw):
                sr, kv = kf(*a, **kw)
                g = torch.Generator().manual_seed(4242)
                return sr + (torch.rand(sr.shape, generator=g) * 2 - 1) * (perturb * float(sr.detach().abs().max())), kv
            O.kbpn_forward = kfp
        try:
            out = O.joint_forward(P, cfg, it, x, hr, mask, k, alpha=0.7)
        finally:
            if perturb:
                O.kbpn_forward = kf
        loss = O.calc_loss(out["segment_loss"], out["sr_loss"], it, cfg)
        loss.backward()
        sc = 1.0
    grads = {n: p.grad.detach() * sc for n, p in P.items() if p.requires_grad and p.grad is not None}
    print(f"  [{style}/{plan}{'/pert' if perturb else ''}] {time.time() - t0:.0f}s loss {float(loss):.6f}", flush=True)
    return out, grads


def cmp(tag, a, b):
    (oa, ga), (ob, gb) = a, b
    e = [rel_err(ga[n], gb[n]) for n in gb if n in ga and float(gb[n].norm()) > 1e-12]
    es = [rel_err(ga[n], gb[n]) for n in gb if n in ga and n.startswith("sr_model") and float(gb[n].norm()) > 1e-12]
    print(f"{tag}: seg max-rel {max_rel_to_scale(oa['segment_preds'].detach(), ob['segment_preds'].detach()):.2e}  "
          f"sr {max_rel_to_scale(oa['sr_preds'].detach(), ob['sr_preds'].detach()):.2e}  IoU {float(O.iou(oa['segment_preds'].detach(), ob['segment_preds'].detach()).min()):.4f}  "
          f"segloss {max_rel_to_scale(oa['segment_loss'].detach(), ob['segment_loss'].detach()):.2e}  grads median {np.median(e):.2e} p90 {np.percentile(e, 90):.2e} "
          f"max {np.max(e):.2e} | kbpn grads median {np.median(es):.2e} max {np.max(es):.2e}", flush=True)


for style in sys.argv[4].split(",") if len(sys.argv) > 4 else ("random", "contractive"):
    ref = run(style, "fp32")
    print(f"  seg range [{float(ref[0]['segment_preds'].min()):.3f}, {float(ref[0]['segment_preds'].max()):.3f}] mean {float(ref[0]['segment_preds'].mean()):.3f}")
    cmp(f"{style}: fp16 storage vs fp32", run(style, "fp16"), ref)
    cmp(f"{style}: fp32 with SR + 1e-3 noise vs fp32", run(style, "fp32", 1e-3), ref)
