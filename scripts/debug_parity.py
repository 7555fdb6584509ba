"""Debug helper (GPU): per-tensor forward errors of the HIP path vs the golden taps / oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from golden_utils import load_golden, det_params, rel_err, max_rel_to_scale
from test_joint_gpu import build_model
from oracle import csbsr_oracle as O

case = sys.argv[1] if len(sys.argv) > 1 else "e2e_pspnet_it40000"
g = load_golden(case)
m, cfg = build_model(g)
t = lambda k: torch.from_numpy(g[k])
it = int(g["it"])
rt = m._runtime()
kb = rt["kbpn"]
x = t("x").cuda()
sr32, kvec = kb.forward(x, it, t("kernel").cuda(), save=True)
torch.cuda.synchronize()
sv = kb.saved
def fm2nchw(fm): return fm.t[..., :fm.c].float().cpu().permute(0, 3, 1, 2)
def show(name, mine, ref):
    ref = torch.as_tensor(ref)
    if tuple(mine.shape) != tuple(ref.shape): mine = mine[:, :ref.shape[1]]
    print(f"{name:28s} maxrel {max_rel_to_scale(mine, ref):.2e}  relL2 {rel_err(mine, ref):.2e}")
show("init_f", fm2nchw(sv["feats"][-1]), g["tap.init_f"])
for s in range(1, 5):
    q = sv["stages"][s-1]
    show(f"s{s}.sr_t", q["sr_t32"].cpu(), g[f"tap.s{s}.sr_t"])
    show(f"s{s}.kvec", q["vec"].cpu(), g[f"tap.s{s}.kvec"])
    hs = sv["concat_h"].slice(128*(s-1), 128*s)
    show(f"s{s}.h", fm2nchw(hs), g[f"tap.s{s}.h"])
    if "lowp" in q: show(f"s{s}.low", fm2nchw(q["lowp"]), g[f"tap.s{s}.low"])
show("sr_preds", sr32.cpu(), g["sr_preds"])
# segmentation net on the reference's sr_preds (isolates PSPNet error)
sr_ref = t("sr_preds").cuda().contiguous()
xin, mean, invstd = m._norm_sr(sr_ref)
drop = {k: None for k in ("drop_1", "drop_2a", "drop_2b", "drop_2c", "aux_drop")}
seg32, aux32 = rt["psp"].forward(xin, drop, True)
torch.cuda.synchronize()
show("seg | ref sr", seg32.cpu(), g["segment_preds"])
# oracle intermediates of PSPNet
P = det_params(requires_grad=False)
pc = O.PathCfg()
with torch.no_grad():
    bn = O.BNState(P, True)
    xn = O.norm_sr(t("sr_preds"), pc)
    show("xin", fm2nchw(xin), xn)
    psp = rt["psp"].saved
    pre = "segmentation_model.feats"
    import torch.nn.functional as F
    c1 = F.conv2d(xn, P[pre + ".conv1.weight"], None, 2, 3)
    show("stem raw", fm2nchw(psp["stem"][0]), c1)
    a = F.relu(bn(pre + ".bn1", c1)); show("stem act", fm2nchw(psp["stem"][3]), a)
    p = F.max_pool2d(a, 3, 2, 1); show("pool", fm2nchw(psp["stem"][4]), p)
    f, x3 = O.resnet34_dilated(P, O.BNState(P, True), xn)
    show("layer4 out", fm2nchw(psp["blocks"][-1][-1]), f)
    show("layer3 out", fm2nchw(psp["aux"][0]), x3)
    for bi in range(len(psp["blocks"])):
        pass
    pm = O.psp_module(P, f)
    show("psp bott", fm2nchw(psp["psp"][2]), pm)
    u1 = O.psp_upsample(P, O.BNState(P, True), "segmentation_model.up_1", pm)
    show("up1", fm2nchw(psp["ups"][0][6]), u1)
    u2 = O.psp_upsample(P, O.BNState(P, True), "segmentation_model.up_2", u1)
    show("up2", fm2nchw(psp["ups"][1][6]), u2)
    u3 = O.psp_upsample(P, O.BNState(P, True), "segmentation_model.up_3", u2)
    show("up3", fm2nchw(psp["ups"][2][6]), u3)
    main, aux = O.pspnet_forward(P, xn, O.BNState(P, True))
    show("seg (oracle same in)", seg32.cpu(), main)
    show("aux", aux32.cpu(), aux)
