"""CPU study: the detector in isolation on IDENTICAL input (the fp32 oracle's SR image): how far fp16 storage, and an exact forward
with fp16-rounded backward (the hi+lo split plan), sit from the fp32 detector.   python scripts/study_detector.py [lr] [B] [detector]"""
import sys, time, os
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from golden_utils import det_params, fp16_storage_sim, rel_err, max_rel_to_scale
from oracle import csbsr_oracle as O
from csbsr_amd.data.synthetic import make_batch

lr = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
det = sys.argv[3] if len(sys.argv) > 3 else "PSPNet"
scale = 4
torch.set_num_threads(8)
cfg = O.PathCfg(scale=scale, detector=det)
x, hr, mask, k = make_batch(B, lr, scale=scale, ksize=21, seed=1121)
GS = float(2 ** 14)


class RB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.half().float()


class bwd_round_sim:
    """exact forward; gradients rounded to fp16 wherever the build stores an activation gradient"""
    def __enter__(self):
        self.saved = (F.conv2d, F.conv_transpose2d, F.batch_norm)
        oc, ot, ob = self.saved
        F.conv2d = lambda x, w, b=None, *a, **k: RB.apply(oc(RB.apply(x), w.half().float() if False else w, b, *a, **k))
        F.batch_norm = lambda x, *a, **k: RB.apply(ob(x, *a, **k))
        return self

    def __exit__(self, *e):
        F.conv2d, F.conv_transpose2d, F.batch_norm = self.saved


with torch.no_grad():
    P0 = det_params(scale=scale, detector=det, requires_grad=False)
    sr, kvec = O.kbpn_forward(P0, x, 40000, k, cfg)
sdf = torch.from_numpy(O.compute_sdf(mask.numpy())).float()


def run(plan):
    P = det_params(scale=scale, detector=det)
    srl = sr.clone().requires_grad_(True)
    ctx = fp16_storage_sim() if plan == "fp16" else (bwd_round_sim() if plan == "split" else None)
    if ctx:
        ctx.__enter__()
    try:
        bn = O.BNState(P, True)
        xin = O.norm_sr(srl, cfg)
        if det == "HRNet_OCR":
            seg, aux = O.hrnet_ocr_forward(P, xin, bn, None)
        else:
            seg, aux = O.pspnet_forward(P, xin, bn, None, None)
        loss = (cfg.main_w * O.boundary_combo_loss(seg, mask, 0.7, cfg, sdf) + cfg.aux_w * O.boundary_combo_loss(aux, mask, 0.7, cfg, sdf)).mean()
        (loss * (GS if plan != "fp32" else 1.0)).backward()
    finally:
        if ctx:
            ctx.__exit__()
    sc = 1.0 if plan == "fp32" else 1.0 / GS
    g = {n: p.grad * sc for n, p in P.items() if p.requires_grad and p.grad is not None}
    return seg.detach(), aux.detach(), bn.new, g, srl.grad * sc


ref = run("fp32")
for plan in ("fp16", "split"):
    t0 = time.time()
    o = run(plan)
    print(plan, "seg %.2e aux %.2e  dsr rel-L2 %.2e" % (max_rel_to_scale(o[0], ref[0]), max_rel_to_scale(o[1], ref[1]), rel_err(o[4], ref[4])))
    bn_e = [max_rel_to_scale(o[2][n], ref[2][n]) for n in ref[2] if "running" in n]
    print("  bn buffers worst %.2e" % max(bn_e))
    es = [(rel_err(o[3][n], ref[3][n]), n) for n in ref[3] if ref[3][n].numel() > 1 and float(ref[3][n].norm()) > 1e-12]
    v = np.array([e for e, _ in es])
    print("  grads: median %.2e p90 %.2e max %.2e (%s)" % (np.median(v), np.percentile(v, 90), v.max(), max(es)[1]))
