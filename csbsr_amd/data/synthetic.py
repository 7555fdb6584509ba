"""Synthetic crack minibatches shaped like ``CrackDataSet.__getitem__`` output
(reference: model/data/crack_dataset.py:40-64, model/data/blur.py:121-179): HR texture in [0,1],
binary crack mask (random-walk polylines), anisotropic Gaussian blur kernel (sigma ~ U(0.2,4), theta ~
U(0,pi)), LR = bicubic-down(depthwise-blur(HR)).  There is no dataset on the bench box, so this is the
input generator for bench.py, smoke() and the tests.  Host-side plumbing: plain
torch ops, not part of the timed hot path.
"""
import math

import torch
import torch.nn.functional as F

MEAN = (0.4741, 0.4937, 0.5048)


def gaussian_kernels(B, ksize, gen, device="cpu"):
    sx = torch.empty(B).uniform_(0.2, 4.0, generator=gen)
    sy = torch.empty(B).uniform_(0.2, 4.0, generator=gen)
    th = torch.empty(B).uniform_(0.0, math.pi, generator=gen)
    r = torch.arange(ksize, dtype=torch.float32) - (ksize - 1) / 2
    yy, xx = torch.meshgrid(r, r, indexing="ij")
    c, s = torch.cos(th)[:, None, None], torch.sin(th)[:, None, None]
    xr = c * xx + s * yy
    yr = -s * xx + c * yy
    k = torch.exp(-0.5 * ((xr / sx[:, None, None]) ** 2 + (yr / sy[:, None, None]) ** 2))
    k = k / k.sum(dim=(1, 2), keepdim=True)
    return k[:, None].to(device)


def crack_masks(B, H, W, gen):
    m = torch.zeros(B, 1, H, W)
    for b in range(B):
        n = int(torch.randint(1, 4, (1,), generator=gen))
        for _ in range(n):
            y = float(torch.rand(1, generator=gen)) * H
            x = float(torch.rand(1, generator=gen)) * W
            ang = float(torch.rand(1, generator=gen)) * 2 * math.pi
            half = int(torch.randint(1, 5, (1,), generator=gen))
            steps = max(H, W)
            turn = (torch.rand(steps, generator=gen) - 0.5) * 0.6
            for t in range(steps):
                ang += float(turn[t])
                y += math.sin(ang)
                x += math.cos(ang)
                yi, xi = int(y), int(x)
                if not (0 <= yi < H and 0 <= xi < W):
                    break
                m[b, 0, max(0, yi - half):yi + half + 1, max(0, xi - half):xi + half + 1] = 1.0
        if m[b].sum() == 0:
            m[b, 0, H // 2 - 1:H // 2 + 2, :] = 1.0
    return m


def make_hr_mask(B, H, gen):
    """HR texture [B,3,H,H] in [0,1] and its crack mask [B,1,H,H] (the part of make_batch that precedes the degradation: bench.py
    generates these at full size and degrades them on the device, csbsr_amd.data.degrade.DeviceDegradation)."""
    g = max(4, H // 32)
    base = torch.rand(B, 3, g, g, generator=gen)
    hr = F.interpolate(base, size=(H, H), mode="bilinear", align_corners=False)
    hr = hr * 0.5 + torch.tensor(MEAN).view(1, 3, 1, 1) - 0.25
    hr = (hr + 0.02 * torch.randn(B, 3, H, H, generator=gen)).clamp(0, 1)
    mask = crack_masks(B, H, H, gen)
    hr = (hr * (1 - 0.5 * mask)).clamp(0, 1)
    return hr, mask


def make_batch(B, lr_size, scale=4, ksize=21, seed=1121, device="cpu", antialias=True):
    """Returns (x_lr [B,3,lr,lr], hr [B,3,H,H], mask [B,1,H,H], kernel [B,1,k,k]) fp32 NCHW."""
    gen = torch.Generator().manual_seed(seed)
    H = lr_size * scale
    hr, mask = make_hr_mask(B, H, gen)
    k = gaussian_kernels(B, ksize, gen)
    w = k.repeat_interleave(3, dim=0)
    blurred = F.conv2d(hr.reshape(1, B * 3, H, H), w, padding=(ksize - 1) // 2, groups=B * 3).reshape(B, 3, H, H)
    x = F.interpolate(blurred, size=(lr_size, lr_size), mode="bicubic", align_corners=False, antialias=antialias)
    x = x.clamp(0, 1)
    return x.to(device), hr.to(device), mask.to(device), k.to(device)
