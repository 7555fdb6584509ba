"""Patch-tiled evaluation step on the device: the body of ``inference_for_ss``
(model/engine/inference.py:76-119) for one batch of test images -- patches through JointModel, stitch, clip, PSNR / SSIM of the SR
image, PSNR of the kernel, and the IoU of the segmentation map at every threshold 0.01 .. 0.99 -- without the [B,99,H,W] broadcast
tensor and the host numpy reductions."""
import numpy as np
import torch

from .data.patch_sampler import JointPatch
from .utils.estimate_metrics import psnr_ssim, iou_sweep

THRESHOLDS = [i * 0.01 for i in range(1, 100)]           # inference.py:50


@torch.no_grad()
def evaluate_batch(model, imgs, img_unfold_shape, seg_unfold_shape, sr_targets, masks, kernel_targets, ksize, thresholds=THRESHOLDS):
    """imgs [B, nPatch, 3, h, w] LR patches (as CrackDataSetTest delivers them), kernel_targets [B, nPatch, K, K];
    returns dict(sr_preds, segment_preds, psnr [B], ssim [B], kernel_psnr [B*nPatch], iou [B, T]) -- numpy arrays for the metrics."""
    joint = JointPatch()
    imgs = imgs.view(-1, *imgs.shape[2:])
    kernel_targets = kernel_targets.view(-1, 1, *kernel_targets.shape[2:])
    dummy = torch.zeros((imgs.shape[0], 1, ksize, ksize))
    sr_preds, segment_preds, kernel_preds = model(imgs, dummy, sr_targets=sr_targets)
    sr_preds = joint(sr_preds, img_unfold_shape)
    segment_preds = joint(segment_preds, seg_unfold_shape)
    sr_preds = sr_preds.clamp(0, 1)
    kernel_preds = kernel_preds.clamp(0, 1)
    ps, ss = psnr_ssim(sr_preds, sr_targets)
    kps, _ = psnr_ssim(kernel_preds, kernel_targets)
    iou = iou_sweep(segment_preds, masks, thresholds)
    return dict(sr_preds=sr_preds, segment_preds=segment_preds, kernel_preds=kernel_preds, psnr=ps.cpu().numpy(), ssim=ss.cpu().numpy(),
                kernel_psnr=kps.cpu().numpy(), iou=iou.cpu().numpy())
