"""How much of the KBPN gradient error comes from the BACKWARD's own recovery of PReLU gates?  (CPU study; test infrastructure: it drives
the oracle.)

The build stores out = prelu(pre) +- res in fp16 and its epilogue-backward rebuilds the activation as fp16(out) -+ fp16(res) to decide
the PReLU gate (1 or the slope).  Where |pre| is below half an ulp of the residual the sum rounds to the residual and the sign is lost.
Here the oracle's fp32 autograd is run twice on a contractive reference fixture -- exact, and with the gates of the residual-PReLU layers
of the up / down blocks (up_conv2 - x, up_conv3 + h0, down_conv2 - x, down_conv3 + l0; kb.up_conv1 + h optional: the build rebuilds that one
exactly since r04) decided the build's way, everything else exact -- and the KBPN parameter gradients are compared.

    python tests/study_gate_recovery.py [fixture] > profiles/r04_gate_recovery_study.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import torch.nn.functional as F

from golden_utils import load_golden, golden_cfg, det_params, fill_style
from oracle import csbsr_oracle as O


class PreluRes(torch.autograd.Function):
    """out = prelu(pre, a) + sign * res; backward with the gate taken from fp16(out) - sign * fp16(res) > 0 (the build's storage)"""

    @staticmethod
    def forward(ctx, pre, a, res, sign):
        out = F.prelu(pre, a) + sign * res
        y16 = out.half().float() - sign * res.half().float()
        ctx.save_for_backward(pre, a, y16)
        ctx.sign = sign
        return out

    @staticmethod
    def backward(ctx, g):
        pre, a, y16 = ctx.saved_tensors
        pos = y16 > 0
        dpre = torch.where(pos, g, g * a)
        da = (g * torch.where(pos, torch.zeros_like(pre), y16 / a)).sum().reshape(a.shape)
        return dpre, da, ctx.sign * g, None


STATS = []


def _conv_pre(P, pre, x, s, p, transposed):
    w, b = P[pre + ".layer.weight"], P.get(pre + ".layer.bias")
    return F.conv_transpose2d(x, w, b, s, p) if transposed else F.conv2d(x, w, b, s, p)


def patched_blocks(include_kb, exact=()):
    def res_prelu(P, pre, y, res, sign):
        if pre.rsplit(".", 1)[1] in exact:      # this layer keeps its exact gate (what a saved act(pre) / sign map would give)
            return F.prelu(y, P[pre + ".act.weight"]) + sign * res
        with torch.no_grad():
            out = F.prelu(y, P[pre + ".act.weight"]) + sign * res
            lost = ((out.half().float() - sign * res.half().float()) > 0) != (y > 0)
            STATS.append((pre, float(lost.float().mean())))
        return PreluRes.apply(y, P[pre + ".act.weight"], res, sign)

    def up_block(P, pre, x, cfg):
        k, s, p = cfg.conv_kspd
        x = O.conv_block(P, pre + ".conv", x, 1, 0, act="prelu")
        h0 = O.deconv_block(P, pre + ".up_conv1", x, s, p, act="prelu")
        d = res_prelu(P, pre + ".up_conv2", _conv_pre(P, pre + ".up_conv2", h0, s, p, False), x, -1.0)
        return res_prelu(P, pre + ".up_conv3", _conv_pre(P, pre + ".up_conv3", d, s, p, True), h0, 1.0)

    def down_block(P, pre, x, cfg):
        k, s, p = cfg.conv_kspd
        x = O.conv_block(P, pre + ".conv", x, 1, 0, act="prelu")
        l0 = O.conv_block(P, pre + ".down_conv1", x, s, p, act="prelu")
        d = res_prelu(P, pre + ".down_conv2", _conv_pre(P, pre + ".down_conv2", l0, s, p, True), x, -1.0)
        return res_prelu(P, pre + ".down_conv3", _conv_pre(P, pre + ".down_conv3", d, s, p, False), l0, 1.0)
    return up_block, down_block


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "wc2_pspnet_it40000"
    torch.set_num_threads(8)
    g = load_golden(case)
    cfg = golden_cfg(g)
    from test_wc_parity_gpu import _inputs
    x, hr, mask, k = _inputs(g)
    it = int(g["it"])
    torch.manual_seed(0)
    probe = None

    def grads(patch, exact=()):
        nonlocal probe
        P = det_params(scale=int(g["scale"]), detector=str(g["detector"]), requires_grad=True, style=fill_style(g))
        saved = (O.up_block, O.down_block)
        if patch:
            O.up_block, O.down_block = patched_blocks(False, exact)
        try:
            sr, kvec = O.kbpn_forward(P, x, it, k, cfg)
        finally:
            O.up_block, O.down_block = saved
        if probe is None:
            probe = torch.randn_like(sr)
        (sr * probe).sum().backward()
        return {n: v.grad.clone() for n, v in P.items() if n.startswith("sr_model") and v.grad is not None}

    ga = grads(False)
    gb = grads(True)
    errs = {n: float((ga[n] - gb[n]).norm() / (ga[n].norm() + 1e-30)) for n in ga if ga[n].numel() > 1 and float(ga[n].norm()) > 0}
    v = np.array(list(errs.values()))
    print(f"{case}: KBPN parameter gradients, exact gates vs gates recovered from fp16(out) -+ fp16(res) (everything else fp32)")
    print(f"  {len(v)} tensors: rel-L2 median {np.median(v):.2e}  p90 {np.percentile(v, 90):.2e}  max {v.max():.2e}")
    for n, e in sorted(errs.items(), key=lambda kv: -kv[1])[:6]:
        print(f"     {e:.2e}  {n}")
    print("  fraction of elements whose gate is lost, per layer:")
    for pre, fr in STATS:
        print(f"     {fr:.2e}  {pre}")
    for label, ex in (("the LR layers exact (up_conv2, down_conv3: what the build does since r04 -- act(pre) saved, residual applied by a second LR kernel)", ("up_conv2", "down_conv3")),
                      ("the HR layers exact (up_conv3, down_conv2)", ("up_conv3", "down_conv2"))):
        gc = grads(True, ex)
        e2 = np.array([float((ga[n] - gc[n]).norm() / (ga[n].norm() + 1e-30)) for n in errs])
        print(f"  {label}: median {np.median(e2):.2e}  p90 {np.percentile(e2, 90):.2e}  max {e2.max():.2e}")


if __name__ == "__main__":
    main()
