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


@pytest.mark.parametrize("case,micro_batch", [("e2e_pspnet_it40000", 1), ("e2e_hrnet_ocr_it40000", 8)])
def test_wgrad_side_stream_gives_the_same_bits(case, micro_batch):
    """``CSBSR_WGRAD_STREAM=1`` (Engine.wgrad_stream, opt-in) runs every weight-gradient kernel on a second HIP stream: same kernels, same
    per-parameter accumulation order, operands kept alive for the side stream -- every gradient must equal the one-stream run bit for
    bit (a missing event wait or a buffer handed out too early shows as a difference)."""
    g = load_golden(case)
    m, _ = build_model(g, micro_batch)
    m.max_resident = 8
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    eng = m._runtime()["eng"]
    assert eng.wgrad_stream() is None, "the side stream is opt-in"
    a = _step(m, g)
    m.load_state_dict(sd0)
    eng._wg_on = True
    try:
        b = _step(m, g)
        assert eng.wg_stream is not None
    finally:
        eng._wg_on = False
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    assert not bad, f"{len(bad)} tensors differ with the wgrad stream on, e.g. {bad[:6]}"


def test_precision_plan_hooks():
    """``model.detector_plan`` / ``Conv.fwd_blocks`` (the per-layer refinement of the split mode): the default plan runs all
    three hi/lo products everywhere in the fused form; a two-product plan for the decoder changes the segmentation map by less than
    1e-3 of its maximum and nothing else of the forward; the unfused three-block form agrees with the fused one to accumulation noise."""
    g = load_golden("wc2_pspnet_it40000")
    from test_wc_parity_gpu import _inputs, _model
    x, hr, mask, k = _inputs(g)
    outs = {}
    for name, plan, fused in (("fused", None, True), ("blocks", None, False), ("decoder2", [(r"\.up_[123]\.|\.final\.", 2)], True)):
        m = _model(g, "split")
        m.detector_plan = plan
        m._runtime()["eng"].split_fused = fused
        with torch.no_grad():
            seg_l, sr_l, seg, sr, kp = m(int(g["it"]), x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
        torch.cuda.synchronize()
        convs = m._runtime()["psp"].all_convs()
        assert {c.fwd_blocks for c in convs} == ({3} if plan is None else {2, 3})
        outs[name] = (seg.clone(), sr.clone())
        del m
    d = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert torch.equal(outs["fused"][1], outs["blocks"][1]) and torch.equal(outs["fused"][1], outs["decoder2"][1])      # KBPN is untouched
    e_form, e_plan = d(outs["blocks"][0], outs["fused"][0]), d(outs["decoder2"][0], outs["fused"][0])
    print(f"split forward: three-block vs fused form {e_form:.2e}; two products in the decoder vs three {e_plan:.2e}")
    assert e_form < 5e-4 and 0.0 < e_plan < 2e-3, (e_form, e_plan)


def test_thin_dact_launch_matches_the_separate_pass():
    """``CSBSR_THIN_DACT=1`` (Engine.thin_dact, opt-in): kb.sr_reconst's accumulating dgrad also runs up_conv3's epilogue-backward pass
    (csrc/conv_thin.hip, DACT) -- dPre, the residual's gradient and the PReLU-slope gradient from one launch.  Same arithmetic except that
    the completed gradient is not rounded to fp16 between the two steps: every gradient agrees to fp16 storage noise."""
    g = load_golden("e2e_pspnet_it40000")
    res = {}
    for on in (False, True):
        m, _ = build_model(g, 8)
        m._runtime()["eng"].thin_dact = on
        res[on] = _step(m, g)
        if on:
            st = m._runtime()["kbpn"].stages[0]
            assert st.sr_reconst.last_fused
        del m
    bad = []
    for k in res[False]:
        a, b = res[False][k].float(), res[True][k].float()
        if a.numel() == 1:       # PReLU slopes: sums of signed terms that nearly cancel (their own check: tests/test_conv_kernels_gpu.py)
            continue
        e = float((a - b).abs().max() / (a.abs().max() + 1e-30))
        if e > 2e-3:
            bad.append((k, e))
    assert not bad, bad[:6]
