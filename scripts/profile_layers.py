"""Per-layer achieved TFLOP/s and GB/s of the conv / wgrad launches in one training step (GPU)."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd.config import cfg as base_cfg
from csbsr_amd.modeling.build_model import JointModelWithLoss
from csbsr_amd.data.synthetic import make_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lr = int(sys.argv[2]) if len(sys.argv) > 2 else 448
m = JointModelWithLoss(base_cfg.clone(), 9000, 40000, None)
m.micro_batch = 8
m.train()
rt = m._runtime()
x, hr, mask, k = make_batch(B, 112, seed=1)
rep = lr // 112
x, hr, mask = x.repeat(1, 1, rep, rep), hr.repeat(1, 1, rep, rep), mask.repeat(1, 1, rep, rep)
x, hr, mask, k = (t.cuda().contiguous() for t in (x, hr, mask, k))
def step():
    m.zero_grad()
    seg_l, sr_l, *_ = m(40000, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
    (0.7 * sr_l.mean() + 0.3 * seg_l.mean()).backward()
step()
rt["eng"].timing = []
torch.cuda.synchronize()
import time
t0 = time.time(); step(); torch.cuda.synchronize(); dt = time.time() - t0
agg = collections.OrderedDict()
for kind, fl, by, e0, e1, name, shp, *rest in rt["eng"].timing:
    key = (kind + str(rest[0] if rest else ''), shp)
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0, name])
    a[0] += 1; a[1] += fl; a[2] += by; a[3] += e0.elapsed_time(e1)
rows = sorted(agg.items(), key=lambda kv: -kv[1][3])
tot = sum(v[3] for v in agg.values())
print(f"step {dt*1e3:.0f} ms; conv+wgrad {tot:.0f} ms")
print("kind   (N,H,W,Cin,Cout,k,s,T)                       n    ms     %    TF/s   GB/s  example")
for (kind, shp), (n, fl, by, ms, name) in rows[:70]:
    print(f"{kind:6s} {str(shp):44s} {n:3d} {ms:7.1f} {100*ms/tot:5.1f} {fl/ms/1e9:7.1f} {by/ms/1e6:6.0f}  {name[-50:]}")
