"""Joint-phase TRAINING TRAJECTORY against the reference (last row): tests/golden/traj_pspnet_it40000.npz holds 12
consecutive optimiser steps of the reference's own loop (trainer.py:57-72 + the alpha schedule of :495-508, Adam of train.py:91) at
iteration 40000.. (joint phase: all 290 tensors train), HR 256, B 2, a fresh synthetic batch per step, Dropout2d masks recorded,
detector weights from the contractive fill.  The HIP path runs the same loop -- same batches, same masks, torch.optim.Adam on its
parameters -- and must stay ON the reference's curve, in both detector precision modes:

    scalar loss, steps 0-3 / every step                                 <= 2e-3 / LOSS_BAND relative (measured 7.7e-4 / 1.5e-3 .. 2.7e-3)
    per-sample segmentation / SR loss at every step                     <= SEG_BAND / SR_BAND        (1.7e-3 .. 6.0e-3, 1.1e-2 .. 2.6e-2)
    gradient L2 norm of each gradient bucket at every step             <= GNORM_BAND relative       (r03: 0.08 .. 0.13; r04: 0.030 split, 0.073 fp16)
    L2 distance the parameters have moved from the start, per step     <= MOVED_BAND relative       (3e-4 .. 2.6e-3)
    alpha schedule                                                      exact
(measured on MI355X, r03, over both precision modes and four builds of the library that differ only in fp32 summation order / which
kernel takes a 64-channel layer / the order in which the concat-feature gradient is summed; bands >= 2x the largest value seen.)  The
trajectories separate slowly -- the per-sample SR loss from 2e-4 at step 0 to 2e-2 at step 11 -- because Adam's normalised update moves a
parameter whose gradient is noise by the full learning rate in a noise-determined direction, so ANY fp16-level change of the arithmetic
(including between two correct builds of this library: 1.7e-3 .. 6.0e-3 on the per-sample segmentation loss) grows the same way; the
first four steps, before that growth, are held to 2e-3, and the scalar loss stays within 0.3 % of the reference's throughout while it
falls by 35 % -- the statement that matters for training.

Round 4 (the weight-rounding compensation of KBPN's forward, engine.Conv._dc_bias, and the fp32 folded constants): the per-bucket
gradient norms start 9e-4 from the reference's at step 0 (r03: 2e-2) and reach 3.0e-2 (split) / 7.3e-2 (fp16 detector) by step 11 -- the same
Adam-driven separation as the losses', not a per-bucket bias: which bucket is worst changes from step to step (kbpn.1 at step 4, kbpn.4 at
step 8, kbpn.0 at step 9).  The band went from 0.3 to 0.15.

This is the evidence that gradient errors of the size the single-step tests report (median 1e-2 .. 3e-2 per tensor in relative L2) do
not bend training: Adam's normalised step turns a gradient with the right sign pattern into the right update, and the loss curve of
step t+1 checks the update of step t."""
import numpy as np
import pytest
import torch

from golden_utils import load_golden

pytestmark = pytest.mark.gpu

LOSS_EARLY = 2e-3
LOSS_BAND = {"fp16": 8e-3, "split": 8e-3}
SEG_BAND = {"fp16": 1.5e-2, "split": 1.5e-2}
SR_BAND = {"fp16": 6e-2, "split": 6e-2}
GNORM_BAND = {"fp16": 0.15, "split": 0.15}
MOVED_BAND = {"fp16": 1e-2, "split": 1e-2}


@pytest.mark.parametrize("precision,optim", [("fp16", "torch"), ("split", "torch"), ("split", "hip")])
def test_joint_phase_trajectory_follows_the_reference(precision, optim):
    from csbsr_amd.config import cfg as base_cfg
    from csbsr_amd.modeling.build_model import JointModelWithLoss
    from csbsr_amd.utils.detfill import deterministic_fill
    from csbsr_amd.data.synthetic import make_batch
    g = load_golden("traj_pspnet_it40000")
    steps, it0, B, lr, scale, seed0 = (int(g[k]) for k in ("steps", "it0", "B", "lr", "scale", "seed0"))
    cfg = base_cfg.clone()
    cfg.MODEL.SCALE_FACTOR, cfg.MODEL.DETECTOR_TYPE = scale, str(g["detector"])
    cfg.SOLVER.TASK_LOSS_WEIGHT, cfg.SOLVER.BATCH_SIZE = float(g["beta"]), 6        # config_csbsr_pspnet.yaml: BATCH_SIZE 6 -> per_epoch 167
    m = JointModelWithLoss(cfg, 1000, 0, None)
    deterministic_fill(m.state_dict(), str(g["fill"]))
    m.detector_precision = precision
    m.micro_batch, m.max_resident = 8, 8
    m.train()
    params = [p for p in m.parameters() if p.requires_grad]
    m._runtime()                                         # parameters move to the device here
    if optim == "hip":        # the hand-written multi-tensor Adam (csbsr_amd/optim.py, what bench.py steps with) on the same curve
        from csbsr_amd.optim import Adam as HipAdam
        opt = HipAdam(params, lr=float(g["lr_rate"]), betas=(0.9, 0.999), eps=1e-8)
    else:
        opt = torch.optim.Adam(params, lr=float(g["lr_rate"]), betas=(0.9, 0.999), eps=1e-8)
    named = [(k, v) for k, v in m._named_full() if isinstance(v, torch.nn.Parameter)]
    start = {k: v.detach().clone() for k, v in named}
    S = m.pc.num_stages
    bucket = {"seg": 0, **{f"kbpn.{s}": s for s in range(1, S + 1)}, "kbpn.0": S + 1}
    beta = float(g["beta"])
    worst = {"loss": 0.0, "seg_loss": 0.0, "sr_loss": 0.0, "gnorm": 0.0, "moved": 0.0}
    for step in range(steps):
        it = it0 + step
        m.ss_loss_fn.fix_alpha = False
        m.ss_loss_fn.update_alpha()
        assert abs(m.ss_loss_fn.alpha - float(g["alpha"][step])) < 1e-12
        x, hr, mask, k = make_batch(B, lr, scale=scale, ksize=21, seed=seed0 + step)
        m.dropout_masks = {kk.split(".", 2)[2]: torch.from_numpy(v) for kk, v in g.items() if kk.startswith(f"dropmask.{step}.")}
        assert len(m.dropout_masks) == 5
        opt.zero_grad()
        seg_l, sr_l, seg, sr, kp = m(it, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
        loss = (1 - beta) * sr_l.mean() + beta * seg_l.mean()
        loss.backward()
        assert not m.last_step_overflowed
        gn = np.zeros(S + 2)
        for kk, v in named:
            if v.grad is not None:
                gn[bucket[m._bucket_of(kk)]] += float(v.grad.double().pow(2).sum())
        gn = np.sqrt(gn)
        opt.step()
        mv = float(np.sqrt(sum(float((v.detach() - start[kk]).double().pow(2).sum()) for kk, v in named)))
        e_loss = abs(float(loss.detach()) - float(g["loss"][step])) / abs(float(g["loss"][step]))
        e_seg = float(np.abs(seg_l.detach().cpu().numpy() - g["seg_loss"][step]).max() / np.abs(g["seg_loss"][step]).max())
        e_sr = float(np.abs(sr_l.detach().cpu().numpy() - g["sr_loss"][step]).max() / np.abs(g["sr_loss"][step]).max())
        e_gn = float((np.abs(gn - g["gnorm"][step]) / g["gnorm"][step]).max())
        e_mv = abs(mv - float(g["moved"][step])) / float(g["moved"][step])
        print(f"   |g| per bucket (seg, kbpn.1..{S}, kbpn.0): hip {np.round(gn, 4).tolist()} ref {np.round(g['gnorm'][step], 4).tolist()}")
        print(f"step {step}: loss {float(loss.detach()):.6f} (ref {float(g['loss'][step]):.6f}, rel {e_loss:.1e})  seg {e_seg:.1e}  sr {e_sr:.1e}  "
              f"|g| per bucket rel {e_gn:.1e}  moved {mv:.5f} (ref {float(g['moved'][step]):.5f}, rel {e_mv:.1e})")
        for name, val in (("loss", e_loss), ("seg_loss", e_seg), ("sr_loss", e_sr), ("gnorm", e_gn), ("moved", e_mv)):
            worst[name] = max(worst[name], val)
        if step < 4:
            assert e_loss < LOSS_EARLY, (step, e_loss)
    print(f"[{precision}] worst over {steps} steps:", {k: f"{v:.2e}" for k, v in worst.items()},
          "bands:", LOSS_BAND[precision], SEG_BAND[precision], SR_BAND[precision], GNORM_BAND[precision], MOVED_BAND[precision])
    assert worst["loss"] < LOSS_BAND[precision] and worst["seg_loss"] < SEG_BAND[precision] and worst["sr_loss"] < SR_BAND[precision], worst
    assert worst["gnorm"] < GNORM_BAND[precision], worst
    assert worst["moved"] < MOVED_BAND[precision], worst
    # the weights actually moved (otherwise the curve would say nothing about the updates): 12 Adam steps of 2e-5 on 89 M parameters
    assert float(g["moved"][-1]) > 0.5
