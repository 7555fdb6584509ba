"""Per-tensor gradient error of the HIP path vs the fp32 oracle and vs the fp16-storage emulation (debug aid)."""
import sys, os, re
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from golden_utils import load_golden, rel_err
import test_joint_gpu as T

case, pat = sys.argv[1], sys.argv[2]
g = load_golden(case)
outs, grads, _ = T.run_hip(g)
P, out, loss = T.run_oracle(g)
Ps = T.run_oracle_fp16_sim(g)[0]
for n in P:
    if grads.get(n) is None or getattr(P[n], "grad", None) is None:
        continue
    if re.search(pat, n):
        og = P[n].grad
        print(n[19:] if n.startswith("seg") else n[9:], tuple(og.shape), "|g| %.3g  hip %.3f sim %.3f" % (float(og.norm()), rel_err(grads[n], og), rel_err(Ps[n].grad, og)))
