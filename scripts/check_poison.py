import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.chdir('/root/repo/tests')
import torch, numpy as np
import test_joint_gpu as T
from golden_utils import load_golden, rel_err
val = float(sys.argv[1]) if len(sys.argv) > 1 else 7.0
case = sys.argv[2] if len(sys.argv) > 2 else "e2e_pspnet_pixelshuffle_it20001"
g = load_golden(case)
P, out, loss = T.run_oracle(g)
for rep in range(3):
    t = torch.full(((30 << 30) // 2,), val, dtype=torch.float16, device="cuda"); del t      # poison the caching allocator's pool
    outs, grads, _ = T.run_hip(g)
    errs = {}
    for n, ref_norm in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        if ref_norm < 1e-7 or grads[n] is None or P[n].grad.numel() == 1: continue
        errs[n] = rel_err(grads[n], P[n].grad)
    v = np.array(list(errs.values())); w = max(errs, key=errs.get)
    print(f"poison {val}: median {np.median(v):.2e} max {v.max():.2e} ({w}) nan={int(np.isnan(v).sum())}  sr_err {float((outs['sr_preds']-torch.from_numpy(g['sr_preds'])).abs().max()):.2e}")
