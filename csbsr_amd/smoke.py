"""One tiny invocation of the hot path on cuda:0, checked against the oracle (driver smoke test)."""
import os
import sys

import torch


def run_smoke():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from csbsr_amd.config import cfg as base_cfg
    from csbsr_amd.modeling.build_model import JointModelWithLoss
    from csbsr_amd.utils.detfill import deterministic_fill, det_state_dict
    from csbsr_amd.modeling.shapes import joint_state_shapes
    from csbsr_amd.data.synthetic import make_batch
    from oracle import csbsr_oracle as O          # checker only

    assert torch.cuda.is_available(), "smoke() needs an MI355X"
    cfg = base_cfg.clone()
    m = JointModelWithLoss(cfg, 1000, 0, None)
    deterministic_fill(m.state_dict())
    m.train()
    m.dropout_enabled = False
    m.ss_loss_fn.alpha = 0.8
    x, hr, mask, k = make_batch(2, 16, seed=3)
    seg_l, sr_l, seg, sr, kp = m(40000, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
    loss = 0.7 * sr_l.mean() + 0.3 * seg_l.mean()
    loss.backward()
    torch.cuda.synchronize()
    P = det_state_dict(joint_state_shapes())
    with torch.no_grad():
        ref = O.joint_forward(P, O.PathCfg(), 40000, x, hr, mask, k, alpha=0.8)
    e_sr = float((sr.cpu() - ref["sr_preds"]).abs().max() / ref["sr_preds"].abs().max())
    e_k = float((kp.cpu() - ref["kernel_preds"]).abs().max() / ref["kernel_preds"].abs().max())
    e_seg = float((seg.cpu() - ref["segment_preds"]).abs().max() / ref["segment_preds"].abs().max())
    iou = float(O.iou(seg.cpu(), ref["segment_preds"]).min())
    ngrad = sum(1 for p in m.parameters() if p.grad is not None and torch.isfinite(p.grad).all())
    print(f"smoke: sr max-rel {e_sr:.2e}  kernel {e_k:.2e}  seg {e_seg:.2e}  IoU-vs-oracle {iou:.4f}  "
          f"loss {float(loss):.5f} (oracle {float(0.7 * ref['sr_loss'].mean() + 0.3 * ref['segment_loss'].mean()):.5f})  grads {ngrad}/290")
    assert e_sr < 2e-3 and e_k < 2e-3, "SR path deviates from the oracle"
    assert e_seg < 5e-2 and iou > 0.95, "segmentation path deviates from the oracle"
    assert ngrad == 290
