"""Run-to-run reproducibility at BASELINE's full size (LR 448 -> HR 1792; the persistent tile kernels, the 256-row LDS-DMA tiles and the
split-K grids only run at this size): one joint-phase step twice in one process -- which tensors differ bit for bit -- and a checksum
line to compare between two processes.      python scripts/check_determinism_full.py [B] [fp16|split] [pspnet|hrnet]      (GPU)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from csbsr_amd.config import cfg as base_cfg
from csbsr_amd.modeling.build_model import JointModelWithLoss
from csbsr_amd.utils.detfill import deterministic_fill
from csbsr_amd.data.synthetic import make_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = sys.argv[2] if len(sys.argv) > 2 else "split"
det = sys.argv[3] if len(sys.argv) > 3 else "pspnet"
cfg = base_cfg.clone()
if det == "hrnet":
    cfg.MODEL.DETECTOR_TYPE = "HRNet_OCR"
m = JointModelWithLoss(cfg, 1000, 0, None)
deterministic_fill(m.state_dict(), "contractive")
m.micro_batch, m.max_resident, m.detector_precision = 4, 8, prec
m.dropout_masks = {}
m.ss_loss_fn.alpha = 0.8
m.train()
x, hr, mask, k = make_batch(B, 112, seed=77)
x, hr, mask = (t.repeat(1, 1, 4, 4).contiguous() for t in (x, hr, mask))
sd0 = {kk: v.detach().clone() for kk, v in m.state_dict().items()}


def step():
    m.load_state_dict(sd0)
    for p in m.parameters():
        p.grad = None
    seg_l, sr_l, seg, sr, kp = m(40000, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
    (0.7 * sr_l.mean() + 0.3 * seg_l.mean()).backward()
    torch.cuda.synchronize()
    out = {"seg_l": seg_l, "sr_l": sr_l, "seg": seg, "sr": sr, "kp": kp}
    out = {kk: v.detach().clone() for kk, v in out.items()}
    for n, v in m._named_full():
        if isinstance(v, torch.nn.Parameter) and v.grad is not None:
            out["grad." + n] = v.grad.detach().clone()
    for n, v in m.state_dict().items():
        if "running" in n:
            out["buf." + n] = v.detach().clone()
    return out


a = step()
b = step()
bad = [(kk, float((a[kk].float() - b[kk].float()).abs().max() / (a[kk].float().abs().max() + 1e-30))) for kk in a if not torch.equal(a[kk], b[kk])]
print(f"{len(a)} tensors, {len(bad)} differ between two runs in one process")
for kk, e in sorted(bad, key=lambda t: -t[1])[:40]:
    print(f"   {e:.2e}  {kk}")
h = hashlib.sha256()
for kk in sorted(a):
    h.update(a[kk].contiguous().cpu().numpy().tobytes())
print("checksum of run 0:", h.hexdigest()[:16], " loss parts", float(a["sr_l"].mean()), float(a["seg_l"].mean()))
for kk in ("sr", "seg", "kp", "grad.sr_model.feat.0.weight", "grad.segmentation_model.final.0.weight"):
    if kk in a:
        print("  ", kk, hashlib.sha256(a[kk].contiguous().cpu().numpy().tobytes()).hexdigest()[:12])
