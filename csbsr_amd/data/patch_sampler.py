"""SplitPatch / JointPatch of the evaluation path (model/data/samplers/patch_sampler.py:15-51): a test image is cut into
non-overlapping patches that run through the model as one batch and are stitched back.  Pure index permutations -- tensor views on
whatever device the tensor lives on (device memory plumbing, no arithmetic)."""
import numpy as np
import torch


class SplitPatch:
    def __init__(self, batch_size, ch, patch_sizeh, patch_sizew):
        self.kc, self.kh, self.kw = ch, patch_sizeh, patch_sizew
        self.batch_size = batch_size

    def __call__(self, x):
        """x [C,H,W] -> (patches [nC*nH*nW, kc, kh, kw], unfold_shape [batch, nC, nH, nW, kc, kh, kw])"""
        C, H, W = x.shape
        nC, nH, nW = C // self.kc, H // self.kh, W // self.kw
        v = x[:nC * self.kc, :nH * self.kh, :nW * self.kw].reshape(nC, self.kc, nH, self.kh, nW, self.kw)
        patches = v.permute(0, 2, 4, 1, 3, 5).contiguous()
        shape = np.append(self.batch_size, np.array(patches.shape))
        return patches.view(-1, self.kc, self.kh, self.kw), shape


class JointPatch:
    def __call__(self, patches, unfold_shape, batch_size=-1):
        s = [int(v) for v in unfold_shape]
        s[0] = -1
        p = patches.view(*s)
        out_c, out_h, out_w = s[1] * s[4], s[2] * s[5], s[3] * s[6]
        return p.permute(0, 1, 4, 2, 5, 3, 6).contiguous().view(-1, out_c, out_h, out_w)
