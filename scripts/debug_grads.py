"""Debug helper (GPU): HIP gradients vs the oracle in fp64 and fp32."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from golden_utils import load_golden, det_params, rel_err
from test_joint_gpu import run_hip, run_oracle
from oracle import csbsr_oracle as O

case = sys.argv[1] if len(sys.argv) > 1 else "e2e_pspnet_it40000"
g = load_golden(case)
outs, grads, _ = run_hip(g)
P32, out32, _ = run_oracle(g)
# fp64 oracle
P64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v) for k, v in det_params().items()}
cfg = O.PathCfg(antialias=bool(g["antialias"]), scale=int(g["scale"]))
t = lambda k: torch.from_numpy(g[k]).double()
drop = {k.split(".", 1)[1]: torch.from_numpy(v).double() for k, v in g.items() if k.startswith("dropmask.")}
orig = O.boundary_combo_loss
O.boundary_combo_loss = lambda pred, target, alpha, cfg, sdf=None: orig(pred, target, alpha, cfg, sdf.double())
out64 = O.joint_forward(P64, cfg, int(g["it"]), t("x"), t("hr"), t("mask"), t("kernel"), alpha=float(g["alpha"]), drop=drop or None)
O.calc_loss(out64["segment_loss"], out64["sr_loss"], int(g["it"]), cfg).backward()
rows = []
for n in grads:
    g64 = P64[n].grad
    if g64 is None or grads[n] is None or float(g64.norm()) < 1e-9: continue
    eh = float((grads[n].double() - g64).norm() / g64.norm())
    e32 = float((P32[n].grad.double() - g64).norm() / g64.norm())
    rows.append((eh, e32, n, g64.numel(), float(g64.norm())))
rows.sort(reverse=True)
print("hip_vs_fp64  fp32oracle_vs_fp64  name  numel  norm")
for r in rows[:45]: print("%.3e  %.3e  %s  %d  %.3e" % r)
eh = np.array([r[0] for r in rows if r[3] > 1]); e32 = np.array([r[1] for r in rows if r[3] > 1])
print("tensors: hip median %.2e p90 %.2e max %.2e | fp32 median %.2e p90 %.2e max %.2e" % (np.median(eh), np.percentile(eh, 90), eh.max(), np.median(e32), np.percentile(e32, 90), e32.max()))
print("---- in network order (segmentation_model)")
d = {r[2]: r for r in rows}
for n in grads:
    if n in d and (n.startswith("segmentation_model") and ("layer" not in n or "layer1.0" in n or "layer4.2" in n or "layer3.5" in n)):
        r = d[n]; print("%.3e  %.3e  %s  %d" % (r[0], r[1], n, r[3]))
