"""Closed-form deterministic parameter fill keyed on state_dict name + shape.

The joint network has 89 M parameters -- too many to commit as fixtures -- so golden vectors are
generated with every parameter/buffer set by this rule ("golden-vector plan").
The rule only needs the ``state_dict`` names and shapes, which this build shares with the reference
(``sr_model.*`` / ``segmentation_model.*``), so the reference, the oracle and the HIP path can each
regenerate identical weights independently.
"""
import math
import zlib

import torch


def _wave(name: str, n: int) -> torch.Tensor:
    """Deterministic pseudo-noise, float64, length n, roughly uniform in [-1, 1] (std ~0.53): a
    seeded CPU generator keyed on the entry's name (bit-reproducible for a given torch build)."""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    return (torch.rand(n, generator=g, dtype=torch.float64) * 2.0 - 1.0) * 0.918


def det_value(name: str, shape, sd_keys) -> torch.Tensor:
    n = 1
    for d in shape:
        n *= int(d)
    shape = tuple(int(d) for d in shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    w = _wave(name, max(n, 1))
    prefix = name.rsplit(".", 1)[0]
    is_bn = (prefix + ".running_mean") in sd_keys
    if name.endswith("running_mean"):
        v = 0.1 * w
    elif name.endswith("running_var"):
        v = 1.0 + 0.3 * w
    elif len(shape) == 4:
        fan = shape[1] * shape[2] * shape[3]
        # _wave has std ~0.53; aim at kaiming-like std sqrt(2/fan)
        v = w * (math.sqrt(2.0 / fan) / 0.53)
        if "kernel_predictor.fe_cat.2" in name:
            v = v * 0.02       # kernel refinement delta << kernel, as in a trained net: keeps k/sum(k) well conditioned
    elif is_bn and name.endswith(".weight"):
        v = 1.0 + 0.2 * w
    elif is_bn and name.endswith(".bias"):
        v = 0.1 * w
    elif name.endswith(".weight") and n == 1:      # PReLU slope, kept positive
        v = 0.05 + 0.04 * w
    else:                                           # conv bias
        v = 0.05 * w
    return v.reshape(shape).to(torch.float32)


@torch.no_grad()
def deterministic_fill(module_or_sd):
    """Overwrite every entry of a state_dict (or a module's) in place; returns the state_dict."""
    sd = module_or_sd if isinstance(module_or_sd, dict) else module_or_sd.state_dict()
    keys = set(sd.keys())
    for k, t in sd.items():
        t.copy_(det_value(k, t.shape, keys).to(t.dtype))
    return sd


def det_state_dict(shapes: dict) -> dict:
    """Build a fresh ``{name: tensor}`` from ``{name: shape}``."""
    keys = set(shapes.keys())
    return {k: det_value(k, s, keys) for k, s in shapes.items()}
