"""SR-loss-only gradient of the w^F case: HIP vs fp32 oracle (debug aid; the seg path is chaotic in fp16, this one is not)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from golden_utils import load_golden, rel_err, det_params, golden_cfg
from oracle import csbsr_oracle as O
import test_joint_gpu as T

for case in sys.argv[1:]:
    g = load_golden(case)
    m, cfg = T.build_model(g)
    t = lambda k: torch.from_numpy(g[k])
    it = int(g["it"])
    seg_l, sr_l, seg, sr, kp = m(it, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
    sr_l.mean().backward()
    grads = {k: (None if v.grad is None else v.grad.detach().cpu()) for k, v in m._named_full() if isinstance(v, torch.nn.Parameter)}
    oc = golden_cfg(g)
    P = det_params(scale=oc.scale, detector=oc.detector)
    out = O.joint_forward(P, oc, it, t("x"), t("hr"), t("mask"), t("kernel"), alpha=float(g["alpha"]))
    out["sr_loss"].mean().backward()
    errs = []
    for n in P:
        if grads.get(n) is None or getattr(P[n], "grad", None) is None or P[n].grad.numel() == 1 or float(P[n].grad.norm()) < 1e-9:
            continue
        errs.append((rel_err(grads[n], P[n].grad), n))
    errs.sort(reverse=True)
    print(case, "SR-only grads: median %.2e max %.2e n=%d" % (np.median([e for e, _ in errs]), errs[0][0], len(errs)), errs[:5])
