"""GPU checks of the rows either side of the hot path against values produced by the
reference's own classes (tests/golden/aux_reference.npz)."""
import numpy as np
import pytest
import torch

from golden_utils import load_golden, max_rel_to_scale

pytestmark = pytest.mark.gpu


def test_device_degradation_matches_crack_dataset():
    """f1: GaussianBlur.make + conv_kernel2d + FactorResize(4, bicubic) for fixed (sigma_x, sigma_y, theta), plus the batch's SDF."""
    from csbsr_amd.data.degrade import DeviceDegradation
    g = load_golden("aux_reference")
    deg = DeviceDegradation(scale=4, ksize=21)
    hr = torch.from_numpy(g["deg_hr"])
    mask = torch.zeros(2, 1, 48, 64)
    mask[0, 0, 10:14, 5:50] = 1
    mask[1, 0, 20:40, 30:33] = 1
    x, hr_d, mask_d, k, sdf = deg(hr, mask, params=torch.from_numpy(g["deg_params"]))
    torch.cuda.synchronize()
    assert tuple(k.shape) == (2, 1, 21, 21) and tuple(x.shape) == (2, 3, 12, 16)
    assert max_rel_to_scale(k[:, 0].cpu(), g["deg_kernels"]) < 1e-6
    assert abs(float(k.sum()) - 2.0) < 1e-5
    assert max_rel_to_scale(x.cpu(), g["deg_lr"]) < 1e-5
    from oracle import csbsr_oracle as O
    assert max_rel_to_scale(sdf.cpu(), torch.from_numpy(O.compute_sdf(mask.numpy())).float()) < 2e-6
    # random draws follow the reference's ranges and feed the model's own argument order
    p = deg.draw_params(64)
    assert float(p[:, :2].min()) >= 0.2 and float(p[:, :2].max()) <= 4.0 and float(p[:, 2].min()) >= 0 and float(p[:, 2].max()) <= np.pi


def test_threshold_sweep_and_metrics_match_the_reference():
    """f2 / f4: IoU at the 99 thresholds (inference.py:111-119), PSNR and SSIM (estimate_metrics.py) on the device."""
    from csbsr_amd.utils.estimate_metrics import iou_sweep, PSNR, SSIM, IoU
    from csbsr_amd.inference import THRESHOLDS
    g = load_golden("aux_reference")
    iou = iou_sweep(torch.from_numpy(g["iou_pred"]), torch.from_numpy(g["iou_mask"]), THRESHOLDS).cpu().numpy()
    assert iou.shape == (2, 99) and np.abs(iou - g["iou_sweep"]).max() < 1e-6
    a, b = torch.from_numpy(g["met_a"]), torch.from_numpy(g["met_b"])
    assert np.abs(PSNR()(a, b) - g["met_psnr"]).max() < 1e-4
    assert np.abs(SSIM()(a, b) - g["met_ssim"]).max() < 1e-5
    single = IoU()((torch.from_numpy(g["iou_pred"]) > 0.5).float(), torch.from_numpy(g["iou_mask"]))
    assert np.abs(single[:, 0] - g["iou_sweep"][:, 49]).max() < 1e-6          # threshold 0.50 is index 49


def test_checkpoint_saved_by_the_reference_drives_the_kernels():
    """f3: weights arriving through fix_model_state_dict + load_state_dict (a DataParallel-prefixed state_dict) reproduce the golden
    outputs -- the load path re-packs the fp32 OIHW / IOHW masters into the fp16 MFMA operands."""
    from csbsr_amd.config import cfg
    from csbsr_amd.modeling.build_model import JointModelWithLoss
    from csbsr_amd.utils.misc import fix_model_state_dict
    from csbsr_amd.utils.detfill import det_value
    g = load_golden("aux_reference")
    e = load_golden("e2e_pspnet_it1")
    keys = [str(k) for k in g["ckpt_keys"]]
    shapes = [tuple(int(d) for d in str(s).split(",") if d) for s in g["ckpt_shapes"]]
    bare = set(k[7:] for k in keys)
    sd = {k: det_value(k[7:], shp, bare) for k, shp in zip(keys, shapes)}
    m = JointModelWithLoss(cfg.clone(), 1000, 0, None)
    m.train()
    m.dropout_masks = {}
    t = lambda k: torch.from_numpy(e[k])
    seg_l, sr_l, seg, sr, kp = m(1, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))      # random init: not the golden
    assert max_rel_to_scale(sr.cpu(), e["sr_preds"]) > 1e-2
    missing, unexpected = m.load_state_dict(fix_model_state_dict(sd), strict=False)
    assert not missing and not unexpected
    seg_l, sr_l, seg, sr, kp = m(1, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
    torch.cuda.synchronize()
    assert max_rel_to_scale(sr.cpu(), e["sr_preds"]) < 1e-3 and max_rel_to_scale(sr_l.detach().cpu(), e["sr_loss"]) < 1e-3


def test_patch_tiled_evaluation_step():
    """f2: one evaluation batch as inference_for_ss runs it: 2 x 2 LR patches per image through JointModel, stitched; the stitched maps
    equal the per-patch outputs placed by hand, the metrics are finite and the threshold sweep is monotone where it must be."""
    from csbsr_amd.config import cfg as base_cfg
    from csbsr_amd.modeling.build_model import JointModel
    from csbsr_amd.utils.detfill import deterministic_fill
    from csbsr_amd.data.patch_sampler import SplitPatch
    from csbsr_amd.data.synthetic import make_batch
    from csbsr_amd.inference import evaluate_batch
    cfg = base_cfg.clone()
    m = JointModel(cfg)
    deterministic_fill(m.state_dict())
    m.eval()
    x, hr, mask, k = make_batch(2, 32, seed=4)                     # LR 32 -> patches of 16
    sp = SplitPatch(2, 3, 16, 16)
    patches, shapes = zip(*(sp(x[i]) for i in range(2)))
    imgs = torch.stack(patches)                                    # [B, 4, 3, 16, 16]
    img_shape = np.array(shapes[0])
    img_shape[5:] *= 4                                             # the SR patches are 4x larger (CrackDataSetTest hands both shapes over)
    seg_shape = img_shape.copy()
    seg_shape[1], seg_shape[4] = 1, 1
    kt = k.repeat(1, 4, 1, 1)                                      # [B, nPatch, K, K]
    out = evaluate_batch(m, imgs, img_shape, seg_shape, hr, mask, kt, ksize=21)
    assert tuple(out["sr_preds"].shape) == (2, 3, 128, 128) and tuple(out["segment_preds"].shape) == (2, 1, 128, 128)
    assert out["iou"].shape == (2, 99) and np.isfinite(out["iou"]).all() and np.isfinite(out["psnr"]).all() and np.isfinite(out["ssim"]).all()
    # stitching: patch (iy, ix) of image b is patch index b*4 + iy*2 + ix of the model's batch
    sr_p, seg_p, _ = m(imgs.view(-1, 3, 16, 16), torch.zeros(8, 1, 21, 21))
    for b in range(2):
        for iy in range(2):
            for ix in range(2):
                tile = out["sr_preds"][b, :, iy * 64:(iy + 1) * 64, ix * 64:(ix + 1) * 64]
                # (the patch batch of 8 and the whole-image run pick different kernels / summation orders for some layers: fp16-rounding level)
                assert float((tile - sr_p[b * 4 + iy * 2 + ix].clamp(0, 1)).abs().max()) < 2e-3
