"""Per-layer precision plan of the split detector mode: which layer groups need all three K blocks?

For each reference fixture (the detector run on the REFERENCE's own SR image, tests/test_wc_parity_gpu.py) and each plan variant:
segmentation-map / loss / BatchNorm-buffer error against the reference and the distribution of the detector gradient errors.
    python scripts/study_split_plan.py [case ...]            (GPU)
Writes gpurun_out/split_plan_study.json.
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch

import test_wc_parity_gpu as T
from golden_utils import load_golden, max_rel_to_scale

GROUPS = {
    "PSPNet": [("stem", r"feats\.conv1"), ("layer1", r"feats\.layer1\."), ("layer2", r"feats\.layer2\."), ("layer3", r"feats\.layer3\."),
               ("layer4", r"feats\.layer4\."), ("psp", r"\.psp\."), ("up_1", r"\.up_1\."), ("up_2", r"\.up_2\."), ("up_3", r"\.up_3\."),
               ("final", r"\.final\."), ("aux", r"\.aux\.")],
    "PSPNet_BlurSkip": [("trunk", r"feats\.|\.psp\.|\.up_[123]\.|\.aux\."), ("bs_conv0", r"blur_skip\.[02]\.conv_(scale|shift)\.0"),
                        ("bs_conv1", r"blur_skip\.[02]\.conv_(scale|shift)\.1"), ("bs_cb", r"blur_skip\.[13]\.layer"), ("final", r"\.final\.")],
    "HRNet_OCR": [("stem", r"backbone\.(conv1|conv2)"), ("layer1", r"backbone\.layer1\."), ("trans", r"backbone\.transition"),
                  ("stage2", r"backbone\.stage2\."), ("stage3", r"backbone\.stage3\."), ("stage4", r"backbone\.stage4\."),
                  ("head", r"^(?!.*backbone)")],
}


def run(case, plan, hp):
    g = load_golden(case)
    x, hr, mask, k = T._inputs(g)
    m = T._model(g, "split")
    m.detector_plan, m.detector_hp_dgrad = plan, hp
    B = x.shape[0]
    seg_l, sr_l, seg, sr, kp = m.forward_from_sr(int(g["it"]), torch.from_numpy(g["sr_preds"]), torch.from_numpy(g["kernel_preds"]).reshape(B, -1),
                                                 x, hr, mask, k)
    beta = float(g["beta"])
    ((1 - beta) * sr_l.mean() + beta * seg_l.mean()).backward()
    torch.cuda.synchronize()
    sd = m.state_dict()
    grads = {kk: v.grad for kk, v in m._named_full() if isinstance(v, torch.nn.Parameter)}
    errs = [e for e in T._grad_errors(g, grads, "segmentation_model") if not T._zero_by_construction(e[0])]
    v = np.array([max(en, es) for n, numel, en, es in errs if numel > 1])
    out = dict(seg=max_rel_to_scale(seg.cpu(), g["segment_preds"]), segl=max_rel_to_scale(seg_l.detach().cpu(), g["segment_loss"]),
               bn=max(max_rel_to_scale(sd[kk[4:]].cpu(), vv) for kk, vv in g.items() if kk.startswith("buf.")),
               g_med=float(np.median(v)), g_p90=float(np.percentile(v, 90)), g_max=float(v.max()))
    if "dsr16" in g:
        dsr_ref = torch.from_numpy(g["dsr16"].astype(np.float32)) / float(g["dsr_scale"])
        out["dsr"] = float((m.last_dsr.cpu() - dsr_ref).norm() / dsr_ref.norm())
    del m
    torch.cuda.empty_cache()
    return out


def main():
    combos = "--combos" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cases = args or ["wc_pspnet_it40000", "wc_blurskip_x8_it40000", "wc_hrnet_ocr_it40000"]
    res = {}
    for case in cases:
        det = str(load_golden(case)["detector"])
        groups = GROUPS[det]
        variants = [("base", None, True), ("no_hp_dgrad", None, False), ("all2", [(".", 2)], True), ("all1", [(".", 1)], True)]
        for name, pat in groups:
            variants.append((f"{name}=1", [(pat, 1)], True))
            variants.append((f"{name}=2", [(pat, 2)], True))
        if combos:      # candidate PLANS (several groups at once), hp dgrads off as in the default
            gp = dict(groups)
            variants = [("base", None, False)]
            if det == "PSPNet_BlurSkip":
                variants += [("bs_conv0+1=2", [(gp["bs_conv0"], 2), (gp["bs_conv1"], 2)], False),
                             ("bs_conv0+1+cb=2", [(gp["bs_conv0"], 2), (gp["bs_conv1"], 2), (gp["bs_cb"], 2)], False),
                             ("bs_all=2 final=2", [(gp["bs_conv0"], 2), (gp["bs_conv1"], 2), (gp["bs_cb"], 2), (gp["final"], 2)], False),
                             ("bs_conv0+1=2 decoder=2", [(gp["bs_conv0"], 2), (gp["bs_conv1"], 2), (r"\.(up_[123]|final)\.", 2)], False)]
            elif det == "PSPNet":
                variants += [("up_1+2+3=2", [(gp["up_1"], 2), (gp["up_2"], 2), (gp["up_3"], 2)], False),
                             # round 6: the decoder groups one at a time and together on tap-sum-rounded weights (review item 6)
                             ("psp=2", [(gp["psp"], 2)], False), ("up_2=2", [(gp["up_2"], 2)], False), ("up_3=2", [(gp["up_3"], 2)], False),
                             ("final=2", [(gp["final"], 2)], False),
                             ("up_2+3+final=2", [(gp["up_2"], 2), (gp["up_3"], 2), (gp["final"], 2)], False),
                             ("up_1+2+3+final=2", [(gp["up_1"], 2), (gp["up_2"], 2), (gp["up_3"], 2), (gp["final"], 2)], False),
                             ("psp+up_1+2+3+final=2", [(gp["psp"], 2), (gp["up_1"], 2), (gp["up_2"], 2), (gp["up_3"], 2), (gp["final"], 2)], False),
                             ("up_1=2", [(gp["up_1"], 2)], False), ("aux=1", [(gp["aux"], 1)], False),
                             ("up_1=2 aux=1", [(gp["up_1"], 2), (gp["aux"], 1)], False)]
        res[case] = {}
        for vname, plan, hp in variants:
            try:
                r = run(case, plan, hp)
            except Exception as ex:       # a variant some kernel refuses: recorded, the sweep goes on
                r = {"error": repr(ex)[:300]}
            res[case][vname] = r
            print(case, vname, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/split_plan_%s.json" % ("combos" if combos else "study"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
