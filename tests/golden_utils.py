"""Helpers shared by the parity tests: load a golden fixture, rebuild the deterministic weights, run
the oracle with autograd and digest gradients the same way tests/golden/make_golden.py did."""
import os
import sys
import zlib

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")

from csbsr_amd.utils.detfill import det_state_dict  # noqa: E402
from csbsr_amd.modeling.shapes import joint_state_shapes  # noqa: E402


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def golden_cfg(g):
    """oracle PathCfg of a fixture (fixtures made before config 5 existed carry no detector / w^F keys)."""
    from oracle import csbsr_oracle as O
    kw = dict(antialias=bool(g["antialias"]), scale=int(g["scale"]))
    if "detector" in g:
        kw.update(detector=str(g["detector"]), sfo_sr_amp=float(g["sfo_sr_amp"]), oriented_w_iter=int(g["oriented_w_iter"]))
    if "beta" in g:
        kw.update(beta=float(g["beta"]))
    if "pixel_shuffle" in g:
        kw.update(pixel_shuffle=bool(g["pixel_shuffle"]))
    if "residual_learning" in g:
        kw.update(residual_learning=bool(g["residual_learning"]), only_kernel_loss=bool(g["only_kernel_loss"]))
    if "kernel_sft" in g:
        kw.update(kernel_sft=bool(g["kernel_sft"]), lr_error=bool(g["lr_error"]))
    if "zero_pad_kernel" in g:
        kw.update(zero_pad_kernel=bool(g["zero_pad_kernel"]))
    return O.PathCfg(**kw)


def fill_style(g):
    """detfill style a fixture was generated with (fixtures older than the contractive fill carry no key)."""
    return str(g["fill"]) if "fill" in g else "random"


def det_params(scale=4, num_stages=4, detector="PSPNet", requires_grad=True, pixel_shuffle=False, style="random", kernel_sft=True, lr_error=False,
               zero_pad_kernel=False):
    shapes = joint_state_shapes(scale=scale, num_stages=num_stages, detector=detector, pixel_shuffle=pixel_shuffle, kernel_sft=kernel_sft,
                                lr_error=lr_error, zero_pad_kernel=zero_pad_kernel)
    sd = det_state_dict(shapes, style)
    if requires_grad:
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
    return sd


def sample_idx(name, numel):
    return [(zlib.crc32((name + str(j)).encode()) % numel) for j in range(4)]


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a)).double().reshape(-1)
    b = torch.as_tensor(np.asarray(b)).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel_to_scale(a, b):
    a = torch.as_tensor(np.asarray(a)).double().reshape(-1)
    b = torch.as_tensor(np.asarray(b)).double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


class fp16_storage_sim:
    """Context manager: run the oracle with the build's precision plan emulated on CPU -- operands of every
    conv / deconv and the outputs of conv, deconv and batch-norm rounded to fp16, fp32 accumulation.  Used to
    derive how far ANY fp16-storage implementation of the reference arithmetic sits from the fp32 reference on
    a given input, so the GPU tolerances for the 40-layer BN segmentation net are measured, not guessed."""

    def __enter__(self):
        import torch.nn.functional as F
        self.F = F
        self.saved = (F.conv2d, F.conv_transpose2d, F.batch_norm)
        oc, ot, ob = self.saved
        r16 = lambda t: t if t is None else t.half().float()

        def conv2d(x, w, b=None, *a, **k):
            if k.get("groups", 1) != 1 or (len(a) >= 4 and a[3] != 1):      # depthwise blur stays fp32 (as in the build)
                return oc(x, w, b, *a, **k)
            return r16(oc(r16(x), r16(w), b, *a, **k))

        def conv_t(x, w, b=None, *a, **k):
            return r16(ot(r16(x), r16(w), b, *a, **k))

        def bnorm(x, *a, **k):
            return r16(ob(x, *a, **k))
        F.conv2d, F.conv_transpose2d, F.batch_norm = conv2d, conv_t, bnorm
        return self

    def __exit__(self, *exc):
        self.F.conv2d, self.F.conv_transpose2d, self.F.batch_norm = self.saved
        return False
