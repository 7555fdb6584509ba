"""Same-run A/B of the KBPN backward's concat-gradient order (gathered per stage vs accumulated by every producer), config 2 at one
micro-batch of 4:   python scripts/gather_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd.config import cfg as base_cfg
from csbsr_amd.modeling.build_model import JointModelWithLoss
from csbsr_amd.data.synthetic import make_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = JointModelWithLoss(base_cfg.clone(), 9000, 40000, None)
m.micro_batch = 4
m.train()
rt = m._runtime()
x, hr, mask, k = make_batch(B, 112, seed=1)
x, hr, mask = (t.repeat(1, 1, 4, 4).cuda().contiguous() for t in (x, hr, mask))
k = k.cuda()
def step():
    m.zero_grad()
    seg_l, sr_l, *_ = m(40000, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
    (0.7 * sr_l.mean() + 0.3 * seg_l.mean()).backward()
step()
for rep in range(3):
    for gather in (True, False):
        rt["kbpn"].gather = gather
        step(); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        print(f"gather={gather}: {(time.time() - t0) / 3 * 1e3:.1f} ms/step (B={B}), peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GB")
