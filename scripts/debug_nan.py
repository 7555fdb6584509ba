"""Find where non-finite values appear in a full-size HRNet-OCR step (debug aid)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from csbsr_amd.config import cfg as base_cfg
from csbsr_amd.modeling.build_model import JointModelWithLoss
from csbsr_amd.data.synthetic import make_batch

B, lr = int(sys.argv[1]), int(sys.argv[2])
cfg = base_cfg.clone()
cfg.MODEL.DETECTOR_TYPE, cfg.SOLVER.TASK_LOSS_WEIGHT = "HRNet_OCR", 0.9
m = JointModelWithLoss(cfg, 9000, 40000, None)
m.train()
if len(sys.argv) > 3:
    m.grad_scale = float(sys.argv[3])
x, hr, mask, k = make_batch(B, 112, seed=1)
rep = lr // 112
x, hr, mask = x.repeat(1, 1, rep, rep), hr.repeat(1, 1, rep, rep), mask.repeat(1, 1, rep, rep)
x, hr, mask, k = (t.cuda().contiguous() for t in (x, hr, mask, k))
seg_l, sr_l, seg, sr, kp = m(40000, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
print("seg_l", seg_l.tolist(), "sr_l", sr_l.tolist(), "seg finite", bool(torch.isfinite(seg).all()), "sr finite", bool(torch.isfinite(sr).all()))
(0.1 * sr_l.mean() + 0.9 * seg_l.mean()).backward()
bad = [(n, float(p.grad.abs().max())) for n, p in m._named_full() if isinstance(p, torch.nn.Parameter) and p.grad is not None and not torch.isfinite(p.grad).all()]
print("non-finite grads:", len(bad), bad[:8])
big = sorted(((float(p.grad.abs().max()), n) for n, p in m._named_full() if isinstance(p, torch.nn.Parameter) and p.grad is not None and torch.isfinite(p.grad).all()), reverse=True)[:6]
print("largest finite grads:", big)
