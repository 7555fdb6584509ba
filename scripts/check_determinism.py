"""Run the same training step of a fixture several times on the HIP path and report which gradient tensors differ run to run (GPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(os.path.join(ROOT, "tests"))
import torch, numpy as np
import test_joint_gpu as T
from golden_utils import load_golden, rel_err
case = sys.argv[1] if len(sys.argv) > 1 else "e2e_pspnet_pixelshuffle_it20001"
g = load_golden(case)
runs = [T.run_hip(g) for _ in range(5)]
o0, g0, _ = runs[0]
for i, (o, gr, _) in enumerate(runs[1:], 1):
    d = {n: rel_err(gr[n], g0[n]) for n in g0 if g0[n] is not None and g0[n].numel() > 1 and float(g0[n].norm()) > 0}
    top = sorted(d.items(), key=lambda kv: -kv[1])[:6]
    print(f"run {i} vs 0: sr {float((o['sr_preds'] - o0['sr_preds']).abs().max() / o0['sr_preds'].abs().max()):.2e} kernel "
          f"{float((o['kernel_preds'] - o0['kernel_preds']).abs().max() / o0['kernel_preds'].abs().max()):.2e}; median {np.median(list(d.values())):.2e}; top:",
          [(n.replace('sr_model.back_projection_stages.', 's'), f"{e:.1e}") for n, e in top])
