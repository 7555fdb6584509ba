"""CPU study (no GPU): what each precision plan costs at a well-conditioned size.
    python scripts/study_precision.py [lr] [B] [detector]
Plans: fp32 oracle | fp16 storage everywhere | fp16 storage in KBPN only (detector fp32 = the hi+lo split mode's ideal)."""
import sys, time, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from golden_utils import det_params, fp16_storage_sim, rel_err, max_rel_to_scale
from oracle import csbsr_oracle as O
from csbsr_amd.data.synthetic import make_batch

lr = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
det = sys.argv[3] if len(sys.argv) > 3 else "PSPNet"
scale = 8 if det == "PSPNet_BlurSkip" else 4
torch.set_num_threads(8)
cfg = O.PathCfg(scale=scale, detector=det, beta=0.9 if det == "HRNet_OCR" else 0.3)
x, hr, mask, k = make_batch(B, lr, scale=scale, ksize=21, seed=1121)
it = 40000
GS = float(2 ** 20)


def run(plan):
    P = det_params(scale=scale, detector=det)
    t0 = time.time()
    if plan == "fp32":
        out = O.joint_forward(P, cfg, it, x, hr, mask, k, alpha=0.7)
    elif plan == "fp16":
        with fp16_storage_sim():
            out = O.joint_forward(P, cfg, it, x, hr, mask, k, alpha=0.7)
    else:   # kbpn16: KBPN under the emulation, detector in fp32
        with fp16_storage_sim():
            sr, kvec = O.kbpn_forward(P, x, it, k, cfg)
        bn = O.BNState(P, True)
        xin = O.norm_sr(sr, cfg)
        if det == "HRNet_OCR":
            seg, aux = O.hrnet_ocr_forward(P, xin, bn, None)
        else:
            seg, aux = O.pspnet_forward(P, xin, bn, None, kvec if det == "PSPNet_BlurSkip" else None)
        sr_loss, kpred = O.kbpn_loss(sr, hr, x, kvec, k, cfg, seg, mask, it)
        sdf = torch.from_numpy(O.compute_sdf(mask.numpy())).float()
        seg_loss = cfg.main_w * O.boundary_combo_loss(seg, mask, 0.7, cfg, sdf) + cfg.aux_w * O.boundary_combo_loss(aux, mask, 0.7, cfg, sdf)
        out = {"segment_loss": seg_loss, "sr_loss": sr_loss, "segment_preds": seg, "sr_preds": sr, "kernel_preds": kpred, "aux_preds": aux,
               "bn_buffers": bn.new}
    loss = O.calc_loss(out["segment_loss"], out["sr_loss"], it, cfg)
    if plan == "fp32":
        loss.backward()
    elif plan == "fp16":
        with fp16_storage_sim():
            (loss * GS).backward()
    else:
        with fp16_storage_sim():      # backward rounding everywhere (the split mode keeps a plain fp16 backward)
            (loss * GS).backward()
    sc = 1.0 if plan == "fp32" else 1.0 / GS
    grads = {n: (None if p.grad is None else p.grad.detach() * sc) for n, p in P.items() if p.requires_grad}
    print(plan, "time %.1fs loss %.6f" % (time.time() - t0, float(loss)))
    return out, grads


ref, gref = run("fp32")
for plan in ("fp16", "kbpn16"):
    o, g = run(plan)
    for kk in ("sr_preds", "kernel_preds", "segment_preds", "aux_preds", "segment_loss", "sr_loss"):
        print("  %-14s max|d|/max|ref| = %.2e" % (kk, max_rel_to_scale(o[kk].detach(), ref[kk].detach())))
    iou = O.iou(o["segment_preds"].detach(), ref["segment_preds"].detach())
    print("  IoU vs fp32", iou.flatten().tolist())
    bn_e = [max_rel_to_scale(o["bn_buffers"][n], ref["bn_buffers"][n]) for n in ref["bn_buffers"] if "running" in n]
    print("  bn buffers worst %.2e median %.2e" % (max(bn_e), float(np.median(bn_e))))
    es_seg, es_sr = [], []
    for n, gr in gref.items():
        if gr is None or gr.numel() == 1 or float(gr.norm()) < 1e-12 or g[n] is None:
            continue
        (es_seg if n.startswith("segmentation") else es_sr).append((rel_err(g[n], gr), n))
    for nm, es in (("seg", es_seg), ("sr", es_sr)):
        if not es:
            continue
        v = np.array([e for e, _ in es])
        print("  grads %s: median %.2e p90 %.2e max %.2e (%s)" % (nm, np.median(v), np.percentile(v, 90), v.max(), max(es)[1]))
