"""Closed-form deterministic parameter fill keyed on state_dict name + shape.

The joint network has 89 M parameters -- too many to commit as fixtures -- so golden vectors are
generated with every parameter/buffer set by this rule ("golden-vector plan").
The rule only needs the ``state_dict`` names and shapes, which this build shares with the reference
(``sr_model.*`` / ``segmentation_model.*``), so the reference, the oracle and the HIP path can each
regenerate identical weights independently.
"""
import math
import re
import zlib

import torch


def _wave(name: str, n: int) -> torch.Tensor:
    """Deterministic pseudo-noise, float64, length n, roughly uniform in [-1, 1] (std ~0.53): a
    seeded CPU generator keyed on the entry's name (bit-reproducible for a given torch build)."""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    return (torch.rand(n, generator=g, dtype=torch.float64) * 2.0 - 1.0) * 0.918


def _smooth_profile(k: int) -> torch.Tensor:
    """1-D low-pass tap profile of length k, sum 1, deliberately NOT symmetric (a flipped or shifted tap order must still show up):
    binomial weights tilted by +-12 % end to end."""
    if k == 1:
        return torch.ones(1, dtype=torch.float64)
    b = torch.tensor([math.comb(k - 1, j) for j in range(k)], dtype=torch.float64)
    tilt = 1.0 + 0.24 * (torch.arange(k, dtype=torch.float64) / (k - 1) - 0.5)
    p = b * tilt
    return p / p.sum()


STYLES = ("random", "contractive")


def det_value(name: str, shape, sd_keys, style: str = "random") -> torch.Tensor:
    """``style``:
      "random"       every conv filter is (pseudo-)white noise at kaiming scale.  Such a BatchNorm'd ReLU stack is a ~100-500x amplifier
                     of any perturbation of its input (white filters pass white noise in full and suppress the smooth signal), which is
                     what the e2e_* / wc_* fixtures carry: good for pinning every tap and channel index, bad for end-to-end bounds.
      "contractive"  the DETECTOR's (``segmentation_model.*``) spatial filters are smooth low-pass profiles (separable tilted binomials,
                     sum 1) times a random channel-mixing matrix at kaiming scale plus 15 % of the white part, and the last BatchNorm of
                     every residual branch has a small gain (gamma ~ 0.25), the 1-class heads carry a bias of -1: like a trained network, the stack damps white perturbations
                     (x ~0.4 per 3x3 layer) instead of amplifying them, so the COMPOSED path -- HIP SR image into HIP detector -- can be
                     held to tight fixed bounds (tests/golden/wc2_*).  KBPN (``sr_model.*``) keeps the random fill."""
    assert style in STYLES, style
    n = 1
    for d in shape:
        n *= int(d)
    shape = tuple(int(d) for d in shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    w = _wave(name, max(n, 1))
    prefix = name.rsplit(".", 1)[0]
    is_bn = (prefix + ".running_mean") in sd_keys
    smooth = style == "contractive" and name.startswith("segmentation_model.")
    if name.endswith("running_mean"):
        v = 0.1 * w
    elif name.endswith("running_var"):
        v = 1.0 + 0.3 * w
    elif len(shape) == 4:
        fan = shape[1] * shape[2] * shape[3]
        # _wave has std ~0.53; aim at kaiming-like std sqrt(2/fan)
        v = w * (math.sqrt(2.0 / fan) / 0.53)
        if smooth and shape[2] * shape[3] > 1:
            mix = _wave(name + "#mix", shape[0] * shape[1]).reshape(shape[0], shape[1], 1, 1) * (math.sqrt(2.0 / shape[1]) / 0.53)
            prof = torch.outer(_smooth_profile(shape[2]), _smooth_profile(shape[3])).reshape(1, 1, shape[2], shape[3])
            v = (mix * prof).reshape(-1) + 0.15 * v
        if (style == "contractive" and name.startswith("sr_model.") and shape[2] == 3 and shape[0] > shape[1] and shape[0] % shape[1] == 0
                and shape[0] // shape[1] in (4, 16, 64)):
            # ConvAndPixelShuffleBlock (MODEL.SR_PIXEL_SHUFFLE): every output pixel of conv3x3 + PixelShuffle(s) sums 9 C kaiming-scaled
            # terms where the ConvTranspose2d(k = 2 s) it replaces sums 4 C terms of weights scaled for a fan of C k^2 -- 4x the output
            # std per up-projection, which compounds to activations of several hundred and an "SR image" of magnitude ~170 by stage 4.
            # An eighth of the kaiming scale keeps the stack's SR residual below the magnitude of the image itself (max |sr| 3.7 instead of
            # 168), as it is in a trained network -- and in the deconvolution variant with plain kaiming weights.
            v = v * 0.125
        if "kernel_predictor.fe_cat.2" in name:
            v = v * 0.02       # kernel refinement delta << kernel, as in a trained net: keeps k/sum(k) well conditioned
    elif is_bn and name.endswith(".weight"):
        v = 1.0 + 0.2 * w
        if smooth and _is_branch_end(name, sd_keys):
            v = 0.25 * v
    elif is_bn and name.endswith(".bias"):
        v = 0.1 * w
    elif name.endswith(".weight") and n == 1:      # PReLU slope, kept positive
        v = 0.05 + 0.04 * w
    elif smooth and shape == (1,) and name.endswith(".bias"):
        v = -1.0 + 0.05 * w                         # the 1-class heads lean towards background, like a trained crack detector
    else:                                           # conv bias
        v = 0.05 * w
    return v.reshape(shape).to(torch.float32)


def _is_branch_end(name: str, sd_keys) -> bool:
    """last BatchNorm of a residual branch: ``...bn2`` of a BasicBlock (ResNet-34 trunk of PSPNet, HRNet branches), ``...bn3`` of an HRNet
    Bottleneck -- i.e. the bn with the highest index among its siblings ``<block>.bn<k>``, in a block that has a ``conv1``."""
    prefix = name.rsplit(".", 1)[0]                 # ...<block>.bn2
    block, leaf = prefix.rsplit(".", 1) if "." in prefix else ("", prefix)
    if not (leaf.startswith("bn") and leaf[2:].isdigit()) or (block + ".conv1.weight") not in sd_keys:
        return False
    if not re.search(r"\.(layer\d+|branches\.\d+)\.\d+$", block):      # (the stems' conv1/bn1/conv2/bn2 are plain chains, not branches)
        return False
    k = int(leaf[2:])
    return k >= 2 and (block + f".bn{k + 1}.weight") not in sd_keys


@torch.no_grad()
def deterministic_fill(module_or_sd, style: str = "random"):
    """Overwrite every entry of a state_dict (or a module's) in place; returns the state_dict."""
    sd = module_or_sd if isinstance(module_or_sd, dict) else module_or_sd.state_dict()
    keys = set(sd.keys())
    for k, t in sd.items():
        t.copy_(det_value(k, t.shape, keys, style).to(t.dtype))
    return sd


def det_state_dict(shapes: dict, style: str = "random") -> dict:
    """Build a fresh ``{name: tensor}`` from ``{name: shape}``."""
    keys = set(shapes.keys())
    return {k: det_value(k, s, keys, style) for k, s in shapes.items()}
