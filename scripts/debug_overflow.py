import sys, os, warnings
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from golden_utils import load_golden
import test_joint_gpu as T
g = load_golden("e2e_pspnet_it40000")
m, cfg = T.build_model(g)
t = lambda k: torch.from_numpy(g[k])
for sb in (-40, 0, 0):
    m.scale_backoff = sb
    m.zero_grad()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        seg_l, sr_l, *_ = m(40000, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
        (0.7 * sr_l.mean() + 0.3 * seg_l.mean()).backward()
    P = m._rt["P"]
    bad = [(k, float(v.gacc.abs().max())) for k, v in P.items() if getattr(v, "gacc", None) is not None and not torch.isfinite(v.gacc).all()]
    norms = torch._foreach_norm([v.gacc for v in P.values() if getattr(v, "gacc", None) is not None])
    print("backoff", sb, "overflow_steps", m.overflow_steps, "nonfinite gacc:", len(bad), bad[:3], "max norm", float(torch.stack(norms).max()), "losses", float(seg_l.mean()), float(sr_l.mean()))
