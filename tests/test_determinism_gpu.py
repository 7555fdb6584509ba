"""Run-to-run reproducibility of the HIP path: the reference's CPU path is bit-reproducible (two runs with one
seed give identical loss, sums and gradient norms), so is this one -- the library holds no floating-point atomics, every reduction is
a fixed tree over partial rows (csrc/common.h).  Two forward + backward passes of ONE model object on the same inputs must agree
BIT FOR BIT on every output, every BatchNorm buffer update and every parameter gradient, for each detector and for both detector
precision modes, with and without the micro-batched / recompute-in-backward schedule."""
import pytest
import torch

from golden_utils import load_golden
from test_joint_gpu import build_model

pytestmark = pytest.mark.gpu


def _step(m, g):
    t = lambda k: torch.from_numpy(g[k])
    it = int(g["it"])
    for p in m.parameters():
        p.grad = None
    seg_l, sr_l, seg, sr, kp = m(it, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
    pc = m.pc
    loss = (1 - pc.beta) * sr_l.mean() + pc.beta * seg_l.mean()
    loss.backward()
    torch.cuda.synchronize()
    out = {"segment_loss": seg_l.detach().clone(), "sr_loss": sr_l.detach().clone(), "segment_preds": seg.detach().clone(),
           "sr_preds": sr.detach().clone(), "kernel_preds": kp.detach().clone()}
    for k, v in m._named_full():
        if isinstance(v, torch.nn.Parameter) and v.grad is not None:
            out["grad." + k] = v.grad.detach().clone()
    return out


@pytest.mark.parametrize("case,precision,micro_batch", [
    ("e2e_pspnet_it40000", "fp16", 8),
    ("e2e_pspnet_it40000", "split", 1),
    ("e2e_pspnet_pixelshuffle_it20001", "fp16", 8),
    ("e2e_blurskip_x8_it40000", "fp16", 8),
    ("e2e_hrnet_ocr_it40000", "fp16", 2),
    ("e2e_hrnet_ocr_it40000", "split", 8),
])
def test_two_runs_are_bit_identical(case, precision, micro_batch):
    g = load_golden(case)
    m, _ = build_model(g, micro_batch)
    m.detector_precision = precision
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    a = _step(m, g)
    bufs_a = {k: v.detach().clone() for k, v in m.state_dict().items() if "running" in k}
    m.load_state_dict(sd0)          # BatchNorm buffers back to their values before the first step
    b = _step(m, g)
    bufs_b = {k: v.detach().clone() for k, v in m.state_dict().items() if "running" in k}
    assert set(a) == set(b)
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    bad += [k for k in bufs_a if not torch.equal(bufs_a[k], bufs_b[k])]
    assert not bad, f"{len(bad)} tensors differ between two runs, e.g. {bad[:6]}"
    assert sum(1 for k in a if k.startswith("grad.")) >= 20


def test_lean_saves_rebuild_the_same_bits():
    """``lean_saves`` drops the kernel predictors' fe_SR chains from the saved set and rebuilds them inside the backward (KBPN.forward): the
    rebuilt maps are the same kernels on the same inputs in the same order, so every gradient must be bit-identical to the full-save run."""
    g = load_golden("e2e_pspnet_it40000")
    m, _ = build_model(g, 8)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.lean_saves = False
    a = _step(m, g)
    m.load_state_dict(sd0)
    m.lean_saves = True
    b = _step(m, g)
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    assert set(a) == set(b) and not bad, bad[:6]
