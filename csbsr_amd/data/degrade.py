"""Device-side synthetic-degradation batch generator.

The reference degrades every sample inside ``CrackDataSet.__getitem__`` (model/data/crack_dataset.py:40-64): a random anisotropic
Gaussian blur kernel (``set_blur(mode="gaus")`` -> ``GaussianBlur.make``, model/data/blur/blur.py:121-167, sigma_x, sigma_y ~
U(0.2, 4), theta ~ U(0, 180 deg)), a depthwise zero-padded blur of the HR crop (``conv_kernel2d``, :169-186) and a bicubic
down-scaling by the SR factor (``FactorResize``, transforms/transforms.py:505-531) -- per sample, on "cuda", from loader workers.
Here the whole minibatch is degraded in three launches on the training stream, and the signed distance map the boundary loss needs
(boundary_loss.py:40-67, identical for the main and the auxiliary head) is computed once per batch next to it, so the loader only
has to deliver HR crops + masks.

Kernels: csbsr_gaussian_kernels (new), csbsr_blur_fwd (stride 1), csbsr_aa_bicubic_down_fwd, csbsr_sdf.  No CPU / torch fallback.
"""
import math

import torch

from .. import _lib as L
from ..engine import _ptr


class DeviceDegradation:
    def __init__(self, scale, ksize=21, sigma_range=(0.2, 4.0), theta_range=(0.0, 180.0), isotropic=False, antialias=True, device="cuda:0",
                 seed=None):
        L.load()
        self.scale, self.K, self.sigma_range, self.theta_range = int(scale), int(ksize), tuple(sigma_range), tuple(theta_range)
        self.isotropic, self.antialias = bool(isotropic), bool(antialias)
        self.device = torch.device(device)
        self.gen = torch.Generator(device="cpu")
        if seed is not None:
            self.gen.manual_seed(seed)

    @property
    def _stream(self):
        import ctypes as C
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def draw_params(self, B):
        """(sigma_x, sigma_y, theta [rad]) per sample with the reference's distributions (blur.py:128,160-167)."""
        u = torch.rand(B, 3, generator=self.gen)
        lo, hi = self.sigma_range
        sx = lo + (hi - lo) * u[:, 0]
        sy = sx.clone() if self.isotropic else lo + (hi - lo) * u[:, 1]
        th = (self.theta_range[0] + (self.theta_range[1] - self.theta_range[0]) * u[:, 2]) * math.pi / 180.0
        return torch.stack([sx, sy, th], 1)

    def kernels(self, params):
        """[B,3] (sigma_x, sigma_y, theta) -> [B,1,K,K] fp32 on the device."""
        p = params.to(self.device, torch.float32).contiguous()
        out = torch.empty(p.shape[0], 1, self.K, self.K, dtype=torch.float32, device=self.device)
        L.call("csbsr_gaussian_kernels", _ptr(p), _ptr(out), p.shape[0], self.K, self._stream)
        return out

    def __call__(self, hr, mask=None, params=None, with_sdf=True):
        """hr [B,3,H,W] fp32 in [0,1] (host or device), mask [B,1,H,W] {0,1}.
        Returns (x_lr [B,3,H/s,W/s], hr, mask, kernels [B,1,K,K], sdf [B,1,H,W] or None) -- all on the device, the argument order of
        JointModelWithLoss.forward(iter, x, sr_targets, segment_targets, kernel_targets)."""
        hr = hr.to(self.device, torch.float32).contiguous()
        B, C, H, W = hr.shape
        assert H % self.scale == 0 and W % self.scale == 0
        if params is None:
            params = self.draw_params(B)
        k = self.kernels(params)
        blurred = torch.empty_like(hr)
        L.call("csbsr_blur_fwd", _ptr(hr), _ptr(k), B, C, H, W, self.K, 1, None, _ptr(blurred), None, 0, self._stream)
        h, w = H // self.scale, W // self.scale
        x = torch.empty(B, C, h, w, dtype=torch.float32, device=self.device)
        L.call("csbsr_aa_bicubic_down_fwd", _ptr(blurred), _ptr(x), B * C, H, W, self.scale, int(self.antialias), self._stream)
        sdf = None
        if mask is not None:
            mask = mask.to(self.device, torch.float32).contiguous()
            if with_sdf:
                sdf = self.sdf(mask)
        return x, hr, mask, k, sdf

    def sdf(self, mask):
        B, _, H, W = mask.shape
        out = torch.empty(B, 1, H, W, dtype=torch.float32, device=self.device)
        scratch = torch.empty(3 * B * H * W + 2 * B, dtype=torch.float32, device=self.device)
        L.call("csbsr_sdf", _ptr(mask), _ptr(out), _ptr(scratch), B, H, W, self._stream)
        return out
