"""What the KBPN forward keeps for its backward, per image at LR 448 (debug aid)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from csbsr_amd.config import cfg as base_cfg
from csbsr_amd.modeling.build_model import JointModelWithLoss
from csbsr_amd.engine import FM
m = JointModelWithLoss(base_cfg.clone(), 9000, 40000, None); m.train()
rt = m._runtime()
x = torch.rand(1, 3, 448, 448, device="cuda"); k = torch.rand(1, 1, 21, 21, device="cuda")
rt["kbpn"].forward(x, 40000, k, save=True)
sv = rt["kbpn"].saved
seen = {}
def walk(o, path):
    if isinstance(o, FM): o = o.t
    if isinstance(o, torch.Tensor):
        base = o.untyped_storage().data_ptr()
        if base not in seen:
            seen[base] = (o.untyped_storage().nbytes(), path)
        return
    if isinstance(o, dict):
        for kk, v in o.items(): walk(v, path + "." + str(kk))
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o): walk(v, path + f"[{i}]")
    elif hasattr(o, "__dict__"):
        for kk, v in vars(o).items(): walk(v, path + "." + kk)
walk(sv, "sv")
tot = sum(v[0] for v in seen.values())
print("total GB", tot / 2**30)
for n, p in sorted(seen.values(), reverse=True)[:45]:
    print(f"{n/2**20:9.1f} MB  {p}")
