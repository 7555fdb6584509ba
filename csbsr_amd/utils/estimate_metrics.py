"""Evaluation metrics on the device: PSNR, SSIM and the IoU threshold sweep of
model/utils/estimate_metrics.py:64-191 and model/engine/inference.py:50-53,111-119, as single-pass HIP kernels
(csbsr_psnr_ssim, csbsr_iou_sweep) instead of five depthwise convolutions per SSIM and a [B,99,H,W] broadcast + a host numpy
reduction per batch.  Same class names / call conventions as the reference; results come back as numpy arrays like there."""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L
from ..engine import _ptr, _reduction_scratch


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dev32(t, like=None):
    dev = like.device if like is not None and like.is_cuda else (t.device if t.is_cuda else torch.device("cuda:0"))
    return t.to(dev, torch.float32).contiguous()


def psnr_ssim(img1, img2):
    """per-sample (PSNR [B], SSIM [B]) fp32 device tensors of two [B,C,H,W] batches in [0,1]."""
    L.load()
    a = _dev32(img1)
    b = _dev32(img2, a)
    B, Cc, H, W = a.shape
    _reduction_scratch(a.device)         # the sums are an order-fixed fold of per-workgroup partial rows (csrc/common.h)
    sums = torch.zeros(B, 2, dtype=torch.float32, device=a.device)
    ps, ss = torch.empty(B, dtype=torch.float32, device=a.device), torch.empty(B, dtype=torch.float32, device=a.device)
    L.call("csbsr_psnr_ssim", _ptr(a), _ptr(b), B, Cc, H, W, _ptr(sums), _ptr(ps), _ptr(ss), _stream(a))
    return ps, ss


class PSNR:
    """estimate_metrics.py:89-101: 10 log10(1 / mse), batch dim retained, images in [0,1]."""
    name = "PSNR"

    def __call__(self, img1, img2):
        return psnr_ssim(img1, img2)[0].cpu().numpy().copy()


class SSIM:
    """estimate_metrics.py:164-191 (window 11, sigma 1.5, size_average=False: one value per sample)."""
    name = "SSIM"

    def __init__(self, window_size=11, size_average=False):
        if window_size != 11 or size_average:
            raise NotImplementedError("the device SSIM is built for the evaluation loop's setting: window 11, per-sample values")

    def __call__(self, img1, img2):
        return psnr_ssim(img1, img2)[1].cpu().numpy().copy()


def iou_sweep(segment_preds, masks, thresholds, smooth=1e-5):
    """IoU of (segment_preds - t > 0) vs (masks > 0.5) for every threshold: [B, len(thresholds)] fp32 device tensor
    (inference.py:111-119 with estimate_metrics.IoU)."""
    L.load()
    p = _dev32(segment_preds)
    m = _dev32(masks, p)
    B = p.shape[0]
    hw = p[0].numel()
    th = torch.tensor([float(t) for t in thresholds], dtype=torch.float32)          # == torch.Tensor(thresholds): fp32 roundings
    assert bool((th[1:] > th[:-1]).all()), "thresholds must ascend"
    th = th.to(p.device)
    T = th.numel()
    hist = torch.zeros(B, 2, T + 1, dtype=torch.int32, device=p.device)
    out = torch.empty(B, T, dtype=torch.float32, device=p.device)
    L.call("csbsr_iou_sweep", _ptr(p), _ptr(m), _ptr(th), B, hw, T, float(smooth), _ptr(hist), _ptr(out), None, None, _stream(p))
    return out


class IoU:
    """estimate_metrics.py:64-84 for binary maps (threshold 0.5)."""
    name = "IoU"

    def __init__(self, th=0.5):
        self.th = 0.5

    def __call__(self, output, target):
        if output.dim() == 4 and output.shape[1] > 1:       # already a stack of binarised maps [B,T,H,W]
            B, T = output.shape[:2]
            outs = [iou_sweep(output[:, t:t + 1], target, [self.th]) for t in range(T)]
            return torch.cat(outs, 1).cpu().numpy()
        return iou_sweep(output, target, [self.th]).cpu().numpy()
