"""PSPNet (dilated ResNet-34 + pyramid pooling + 3 upsample stages + aux head) on the HIP engine.

Counterpart of /root/reference/model/modeling/pspnet_pytorch/pspnet.py:23-123 and extractors.py:41-70,
112-161 with the reference's parameter names; train-mode BatchNorm statistics come from the conv
epilogue, Dropout2d channel masks are fused into the following resize / BN-apply, the PSP concat buffer is
written in place by its producers.  Explicit backward.
"""
import ctypes as C
import os

import torch

from .. import _lib as L
from ..engine import FM, Conv, BatchNorm, grad_acc, pad8, _ptr
from .shapes import RESNET34

DROP_P = {"drop_1": 0.3, "drop_2a": 0.15, "drop_2b": 0.15, "drop_2c": 0.15, "aux_drop": 0.1}   # pspnet.py:67,73,83
DROP_C = {"drop_1": 1024, "drop_2a": 256, "drop_2b": 64, "drop_2c": 64, "aux_drop": 256}


class ConvBN:
    def __init__(self, eng, P, conv_name, bn_name, cin, cout, k, stride, pad, dil=1, bias=False):
        self.conv = Conv(eng, conv_name, P, k, stride, pad, dil, bias=bias, act=L.ACT_NONE)
        self.bn = BatchNorm(eng, bn_name, P, cout)
        self.eng = eng

    def fwd(self, x, training=True):
        if not training:                # eval: running statistics (F.batch_norm(training=False))
            raw = self.conv.fwd(x)
            return raw, self.bn.rmean, torch.rsqrt(self.bn.rvar + 1e-5)
        stat = self.bn.new_stat()
        raw = self.conv.fwd(x, stat=stat, stat_mode=L.STAT_BN)
        mean, invstd = self.bn.finalize(stat, raw.npix, update_running=training)
        return raw, mean, invstd


class _SFTLike:
    """SFTLikeBlock (blocks.py:86-120): features * sigmoid(conv(prelu(conv(cat)))) + conv(prelu(conv(cat))), cat = features ++ the
    spatially constant kernel code -- the constant 441 channels are folded into border-class biases (Conv.fwd_folded)."""

    def __init__(self, eng, P, pre, cf, cconst):
        def mk(name, act, prelu):
            return Conv(eng, f"{pre}.{name}.layer", P, 3, 1, 1, bias=True, act=act, prelu=f"{pre}.{name}.act.weight" if prelu else False,
                        split=(cf, cconst) if prelu else None)
        self.sc0, self.sc1 = mk("conv_scale.0", L.ACT_PRELU, True), mk("conv_scale.1", L.ACT_SIGMOID, False)
        self.sh0, self.sh1 = mk("conv_shift.0", L.ACT_PRELU, True), mk("conv_shift.1", L.ACT_NONE, False)

    def convs(self):
        return [self.sc0, self.sc1, self.sh0, self.sh1]


class PSPNet:
    drop_keys = tuple(DROP_P)

    def __init__(self, eng, params, prefix="segmentation_model", blur_dim=None):
        """blur_dim: build PSPNet_BlurSkip (pspnet.py:127-207) whose forward also takes the kernel code [B, blur_dim]; in that
        variant only blur_skip.* is trainable (build_model.py:352-368), so the backward stops there."""
        self.eng, self.P, self.prefix = eng, params, prefix
        self.blur_dim = blur_dim
        self.blur_skip = []
        if blur_dim is not None:
            for i in range(2):
                self.blur_skip.append((_SFTLike(eng, params, f"{prefix}.blur_skip.{2 * i}", 64, blur_dim),
                                       ConvBN(eng, params, f"{prefix}.blur_skip.{2 * i + 1}.layer", f"{prefix}.blur_skip.{2 * i + 1}.norm",
                                              64, 64, 3, 1, 1)))
            m = torch.ones(4, 3)
            m[2, 0] = m[3, 0] = 0.0          # first row/col: tap 0 reads outside
            m[1, 2] = m[3, 2] = 0.0          # last row/col: tap 2 reads outside
            self.Mtap = m.to(eng.device)
        e, P, f = eng, params, prefix + ".feats"
        self.stem = ConvBN(e, P, f + ".conv1", f + ".bn1", 3, 64, 7, 2, 3)
        self.blocks = []
        inpl = 64
        for li, (planes, nblk, stride, dil) in enumerate(RESNET34, 1):
            for b in range(nblk):
                bp = f"{f}.layer{li}.{b}"
                first = b == 0
                s_, d_ = (stride, 1) if first else (1, dil)
                blk = {"c1": ConvBN(e, P, bp + ".conv1", bp + ".bn1", inpl if first else planes, planes, 3, s_, d_, d_),
                       "c2": ConvBN(e, P, bp + ".conv2", bp + ".bn2", planes, planes, 3, 1, d_, d_),
                       "down": None, "layer": li}
                if first and (stride != 1 or inpl != planes):
                    blk["down"] = ConvBN(e, P, bp + ".downsample.0", bp + ".downsample.1", inpl, planes, 1, stride, 0)
                self.blocks.append(blk)
            inpl = planes
        self.psp_convs = [Conv(e, f"{prefix}.psp.stages.{i}.1", P, 1, bias=False) for i in range(4)]
        self.bottleneck = Conv(e, prefix + ".psp.bottleneck", P, 1, bias=True, act=L.ACT_RELU)
        self.ups = [ConvBN(e, P, f"{prefix}.{n}.conv.0", f"{prefix}.{n}.conv.1", ci, co, 3, 1, 1, bias=True)
                    for n, ci, co in (("up_1", 1024, 256), ("up_2", 256, 64), ("up_3", 64, 64))]
        self.up_prelu = [P[f"{prefix}.{n}.conv.2.weight"] for n in ("up_1", "up_2", "up_3")]
        self.final = Conv(e, prefix + ".final.0", P, 1, bias=True, act=L.ACT_SIGMOID)
        self.aux0 = ConvBN(e, P, prefix + ".aux.0", prefix + ".aux.1", 256, 256, 3, 1, 1)
        self.aux4 = Conv(e, prefix + ".aux.4", P, 1, bias=True, act=L.ACT_SIGMOID)
        self.saved = None
        # "split": every forward activation of the detector is an fp16 hi + lo pair (~22 mantissa bits) and every forward conv runs
        # three MFMA passes (x_hi w_hi + x_lo w_hi + x_hi w_lo) in one fp32 accumulator; the backward reads the hi planes (BatchNorm /
        # max-pool backward also the lo planes, so their masks are the forward's).  Set by the model (detector_precision).
        self.split = False
        # split mode: how many of the three upsample stages, counted from the output, run plain fp16 again (their input is the hi plane of
        # the stage before).  Rounding introduced there only passes the remaining 1-2 conv + BatchNorm layers and the 1x1 head, so it is
        # not amplified like the trunk's, while those stages hold the largest maps of the detector (64 channels at HR / HR/2).
        self.split_tail_plain = int(os.environ.get("CSBSR_SPLIT_TAIL_PLAIN", "0"))

    def all_convs(self):
        cs = [self.stem.conv] + [b[k].conv for b in self.blocks for k in ("c1", "c2", "down") if b[k] is not None]
        cs += self.psp_convs + [self.bottleneck] + [u.conv for u in self.ups] + [self.final, self.aux0.conv, self.aux4]
        for sft, cb in self.blur_skip:
            cs += sft.convs() + [cb.conv]
        return cs

    def invalidate(self):
        for c in self.all_convs():
            c.invalidate()

    def make_dropout(self, B, training, enabled=True):
        if not training or not enabled:
            return {k: None for k in DROP_P}
        out = {}
        for k, p in DROP_P.items():
            keep = (torch.rand(B, DROP_C[k], device=self.eng.device) >= p).to(torch.float32) / (1.0 - p)
            out[k] = keep.contiguous()
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, xin, drop, training=True, kvec=None):
        """xin: FM [B,H,W,8] (3 real channels, already normalised); kvec [B, blur_dim] fp32 for the BlurSkip variant.
        Returns (seg32, aux32) fp32 [B,1,H,W]."""
        e = self.eng
        B, H, W = xin.N, xin.H, xin.W
        sv = {"xin": xin, "drop": drop}
        keep_trunk = self.blur_dim is None      # BlurSkip: the trunk is frozen, nothing of it is needed by the backward
        raw, m, iv = self.stem.fwd(xin, training)
        a = self.stem.bn.apply(raw, m, iv, act=L.ACT_RELU)
        PH, PW = (a.H + 2 - 3) // 2 + 1, (a.W + 2 - 3) // 2 + 1
        p = e.new(B, PH, PW, 64, split=bool(a.lo))
        L.call("csbsr_maxpool3x3s2_fwd_split", _ptr(a.t), a.ld, a.lo, _ptr(p.t), p.ld, p.lo, B, a.H, a.W, 64, e.stream)
        sv["stem"] = (raw, m, iv, a, p)
        x = p
        bsv = []
        x3 = None
        nb = len(self.blocks)
        for bi, blk in enumerate(self.blocks):
            r1, m1, i1 = blk["c1"].fwd(x, training)
            a1 = blk["c1"].bn.apply(r1, m1, i1, act=L.ACT_RELU)
            r2, m2, i2 = blk["c2"].fwd(a1, training)
            if blk["down"] is not None:
                rd, md, idv = blk["down"].fwd(x, training)
                res = blk["down"].bn.apply(rd, md, idv, act=L.ACT_NONE)
                dsv = (rd, md, idv)
            else:
                res, dsv = x, None
            out = None
            if bi == nb - 1:    # last block writes straight into the PSP concat buffer (channels 2048:2560)
                cat = e.new(B, r2.H, r2.W, 2560, split=bool(r2.lo))
                out = cat.slice(2048, 2560)
            y = blk["c2"].bn.apply(r2, m2, i2, act=L.ACT_RELU, res=res, out=out)
            bsv.append((x, r1, m1, i1, a1, r2, m2, i2, res, dsv, y) if keep_trunk else None)
            x = y
            if blk["layer"] == 3 and (bi + 1 == nb or self.blocks[bi + 1]["layer"] == 4):
                x3 = y
        sv["blocks"] = bsv
        fH, fW = x.H, x.W
        # pyramid pooling
        psp = []
        for i, size in enumerate((1, 2, 3, 6)):
            pooled = e.new(B, size, size, 512, split=bool(x.lo))
            L.call("csbsr_adaptive_avgpool_fwd_split", _ptr(x.t), x.ld, x.lo, _ptr(pooled.t), pooled.ld, pooled.lo, B, fH, fW, 512, size, size,
                   e.stream)
            pc = self.psp_convs[i].fwd(pooled)
            e.bilinear(pc, fH, fW, False, out=cat.slice(512 * i, 512 * (i + 1)))
            psp.append((pooled, pc))
        bott = self.bottleneck.fwd(cat)
        sv["psp"] = (cat, psp, bott)
        # upsample stages; the dropout that follows stage j is fused into stage j+1's resize (or the last BN apply)
        cur, cur_drop = bott, drop["drop_1"]
        usv = []
        names = ("drop_2a", "drop_2b", "drop_2c")
        for j, up in enumerate(self.ups):
            if cur.lo and j >= len(self.ups) - self.split_tail_plain:
                cur = FM(cur.t, cur.c, H=cur.H, W=cur.W)          # the hi plane alone: this stage and everything after it is plain fp16
            u = e.bilinear(cur, cur.H * 2, cur.W * 2, False, drop=cur_drop)
            raw, m, iv = up.fwd(u, training)
            last = j == 2
            y = up.bn.apply(raw, m, iv, act=L.ACT_PRELU, prelu=self.up_prelu[j], drop=drop[names[j]] if last else None)
            usv.append((cur, cur_drop, u, raw, m, iv, y) if keep_trunk else None)
            cur, cur_drop = y, drop[names[j]]
        sv["ups"] = usv
        if self.blur_dim is not None:
            cur = self._blur_skip_fwd(cur, kvec, training, sv)
        seg32 = e.f32(B, 1, H, W, zero=False)
        self.final.fwd(cur, out32=seg32)
        # aux head on layer3 output
        ra, ma, ia = self.aux0.fwd(x3, training)
        aa = self.aux0.bn.apply(ra, ma, ia, act=L.ACT_RELU, drop=drop["aux_drop"])
        aux_lo = e.f32(B, 1, x3.H, x3.W, zero=False)
        self.aux4.fwd(aa, out32=aux_lo)
        aux32 = e.f32(B, 1, H, W, zero=False)
        L.call("csbsr_bilinear32_fwd", _ptr(aux_lo), _ptr(aux32), B, x3.H, x3.W, H, W, 1, e.stream)
        sv["aux"] = (x3, ra, ma, ia, aa, aux_lo)
        sv["seg32"], sv["p3"] = seg32, cur
        if not keep_trunk:
            sv["stem"] = sv["psp"] = sv["aux"] = None
        self.saved = sv
        return seg32, aux32

    # ------------------------------------------------------------------ BlurSkip branch (pspnet.py:191-198)
    def _blur_skip_fwd(self, p, kvec, training, sv):
        e = self.eng
        q, bsv = p, []
        for sft, cb in self.blur_skip:
            t1, f1 = sft.sc0.fwd_folded(q, kvec, self.Mtap)
            sc = sft.sc1.fwd(t1)
            t2, f2 = sft.sh0.fwd_folded(q, kvec, self.Mtap)
            y = sft.sh1.fwd(t2, res=q, res2=sc, res_mode=L.RES_FMA)              # q * scale + shift
            raw, m, iv = cb.fwd(y, training)
            z = cb.bn.apply(raw, m, iv, act=L.ACT_RELU)
            bsv.append((q, t1, f1, sc, t2, f2, y, raw, m, iv, z))
            q = z
        out = e.new(p.N, p.H, p.W, p.c, split=bool(p.lo))
        L.call("csbsr_axpby_split", p.npix, p.cp, _ptr(p.t), p.ld, p.lo, 1.0, _ptr(q.t), q.ld, q.lo, 1.0, _ptr(out.t), out.ld, out.lo,
               e.stream)    # p + _p
        sv["blur_skip"] = bsv
        return out

    def _blur_skip_bwd(self, d):
        """d: gradient wrt (p + _p).  Accumulates the blur_skip.* gradients; nothing upstream of it is trainable."""
        e = self.eng
        bsv = self.saved["blur_skip"]

        def act_bwd(conv, dout, out):
            e.epilogue_bwd(dout, out=out, act=conv.act, slope=conv.slope, prelu=conv.prelu, dpre=dout, dbias=grad_acc(conv.b),
                           dprelu=None if conv.prelu is None else grad_acc(conv.prelu), creal=conv.cout)
            return dout

        for i in (1, 0):
            sft, cb = self.blur_skip[i]
            q, t1, f1, sc, t2, f2, y, raw, m, iv, z = bsv[i]
            bsv[i] = None
            draw = cb.bn.backward(d, raw, m, iv, act=L.ACT_RELU)
            cb.conv.bwd_weights(draw, y)
            dy = cb.conv.bwd_input(draw)
            del draw, d
            need_dq = i > 0                   # block 0's input is the frozen trunk's output
            dq = e.new(q.N, q.H, q.W, q.c) if need_dq else None
            dsc = e.new(q.N, q.H, q.W, q.c)
            e.epilogue_bwd(dy, out=y, res=q, res2=sc, res_mode=L.RES_FMA, dpre=dy, dres=dq, dres2=dsc, dbias=grad_acc(sft.sh1.b),
                           creal=sft.sh1.cout)
            for c1, c0, t, dz, fold in ((sft.sh1, sft.sh0, t2, dy, f2), (sft.sc1, sft.sc0, t1, None, f1)):
                if dz is None:
                    dz = act_bwd(c1, dsc, sc)
                c1.bwd_weights(dz, t)
                if t.H >= 3 and t.W >= 3 and t.H * t.W >= 1024 and e.fold_prelu and e.prelu_fold_ok(c0.prelu):      # (csbsr_border_class_sums_prelu's own precondition; a slope safely > 0)
                    # conv0's PReLU derivative rides on conv1's dgrad (mask = conv0's saved output, slope read on the device); its bias
                    # gradient is the sum of the border-class sums the folded weight gradient takes anyway, its slope gradient comes out
                    # of the same pass over (dPre, t): no epilogue-backward pass over the 505-channel HR map (8.5 ms each at B = 4)
                    dt = c1.bwd_input(dz, mask=(t, c0.prelu))
                    c0.bwd_weights_folded(dt, q, fold, self.Mtap, bias_grad=True, prelu_out=t)
                else:
                    dt = c1.bwd_input(dz)
                    act_bwd(c0, dt, t)
                    c0.bwd_weights_folded(dt, q, fold, self.Mtap)
                if need_dq:
                    c0.bwd_input(dt, seg=0, out=dq, accumulate=True)
                del dt
            del dy, dsc
            d = dq

    # ------------------------------------------------------------------ backward
    def _sigmoid_head_bwd(self, conv, dprob32, prob32, x, frozen=False):
        """1-channel sigmoid head: returns dgrad wrt x; accumulates weight / bias grads."""
        e = self.eng
        B, _, H, W = prob32.shape
        dpre = e.new(B, H, W, 1)
        L.call("csbsr_sigmoid_bwd_to_nhwc8", _ptr(dprob32), _ptr(prob32), _ptr(dpre.t), B * H * W, 1.0, e.stream)
        if not frozen:
            e.epilogue_bwd(dpre, dbias=grad_acc(conv.b), creal=1)
            conv.bwd_weights(dpre, x)
        return conv.bwd_input(dpre)

    def backward(self, dseg32, daux32):
        """Gradients (scaled) wrt the two probability maps -> returns FM gradient wrt xin [B,H,W,8]."""
        e, sv = self.eng, self.saved
        drop = sv["drop"]
        xin = sv["xin"]
        B, H, W = xin.N, xin.H, xin.W
        if self.blur_dim is not None:
            self._blur_skip_bwd(self._sigmoid_head_bwd(self.final, dseg32, sv["seg32"], sv["p3"], frozen=True))
            self.saved = None
            return None
        # aux head
        x3, ra, ma, ia, aa, aux_lo = sv["aux"]
        daux_lo = e.f32(B, 1, x3.H, x3.W, zero=False)
        L.call("csbsr_bilinear32_bwd", _ptr(daux32), _ptr(daux_lo), B, x3.H, x3.W, H, W, 1, e.stream)
        daa = self._sigmoid_head_bwd(self.aux4, daux_lo, aux_lo, aa)
        dra = self.aux0.bn.backward(daa, ra, ma, ia, act=L.ACT_RELU, drop=drop["aux_drop"])
        self.aux0.conv.bwd_weights(dra, x3)
        dx3_aux = self.aux0.conv.bwd_input(dra)
        del daa, dra
        # main head
        d = self._sigmoid_head_bwd(self.final, dseg32, sv["seg32"], sv["p3"])
        names = ("drop_2a", "drop_2b", "drop_2c")
        for j in (2, 1, 0):
            cur, cur_drop, u, raw, m, iv, y = sv["ups"][j]
            up = self.ups[j]
            draw = up.bn.backward(d, raw, m, iv, act=L.ACT_PRELU, prelu=self.up_prelu[j], drop=drop[names[j]] if j == 2 else None,
                                  dprelu=grad_acc(self.up_prelu[j]))
            up.conv.bwd_weights(draw, u)
            grad_acc(up.conv.b)          # a conv bias feeding train-mode BN has an identically zero gradient (not None)
            du = up.conv.bwd_input(draw)
            d = e.new(B, cur.H, cur.W, cur.c)
            e.bilinear_bwd(du, d, False, False, drop=cur_drop)
            del draw, du
        cat, psp, bott = sv["psp"]
        self.eng.epilogue_bwd(d, out=bott, act=L.ACT_RELU, dpre=d, dbias=grad_acc(self.bottleneck.b), creal=1024)
        self.bottleneck.bwd_weights(d, cat)
        dcat = self.bottleneck.bwd_input(d)
        del d
        fH, fW = cat.H, cat.W
        df = dcat.slice(2048, 2560)
        for i, size in enumerate((1, 2, 3, 6)):
            pooled, pc = psp[i]
            dpc = e.new(B, size, size, 512)
            e.bilinear_bwd(dcat.slice(512 * i, 512 * (i + 1)), dpc, False, False)
            self.psp_convs[i].bwd_weights(dpc, pooled)
            dpool = self.psp_convs[i].bwd_input(dpc)
            L.call("csbsr_adaptive_avgpool_bwd", _ptr(dpool.t), _ptr(df.t), df.ld, 1, B, fH, fW, 512, size, size, e.stream)
        dy = df
        nb = len(self.blocks)
        for bi in range(nb - 1, -1, -1):
            blk = self.blocks[bi]
            x, r1, m1, i1, a1, r2, m2, i2, res, dsv, y = sv["blocks"][bi]
            if blk["layer"] == 3 and (bi + 1 == nb or self.blocks[bi + 1]["layer"] == 4):
                L.call("csbsr_axpby", dy.npix, dy.cp, _ptr(dy.t), dy.ld, 1.0, _ptr(dx3_aux.t), dx3_aux.ld, 1.0, _ptr(dy.t), dy.ld, e.stream)
            dres = e.new(B, res.H, res.W, res.c)
            dr2 = blk["c2"].bn.backward(dy, r2, m2, i2, act=L.ACT_RELU, res=res, dres=dres)
            blk["c2"].conv.bwd_weights(dr2, a1)
            da1 = blk["c2"].conv.bwd_input(dr2)
            dr1 = blk["c1"].bn.backward(da1, r1, m1, i1, act=L.ACT_RELU)
            blk["c1"].conv.bwd_weights(dr1, x)
            del dr2, da1
            if blk["down"] is not None:
                rd, md, idv = dsv
                drd = blk["down"].bn.backward(dres, rd, md, idv, act=L.ACT_NONE)
                blk["down"].conv.bwd_weights(drd, x)
                dx = blk["down"].conv.bwd_input(drd, in_hw=(x.H, x.W))
                blk["c1"].conv.bwd_input(dr1, out=dx, accumulate=True, in_hw=(x.H, x.W))
            else:
                dx = dres
                blk["c1"].conv.bwd_input(dr1, out=dx, accumulate=True, in_hw=(x.H, x.W))
            dy = dx
            sv["blocks"][bi] = None
        raw, m, iv, a, p = sv["stem"]
        da = e.new(B, a.H, a.W, 64)
        assert dy.flat_ok() and dy.ld == dy.cp
        L.call("csbsr_maxpool3x3s2_bwd_split", _ptr(a.t), a.ld, a.lo, _ptr(p.t), p.ld, p.lo, _ptr(dy.t), _ptr(da.t), B, a.H, a.W, 64, e.stream)
        draw = self.stem.bn.backward(da, raw, m, iv, act=L.ACT_RELU)
        self.stem.conv.bwd_weights(draw, xin)
        dxin = self.stem.conv.bwd_input(draw, in_hw=(H, W))
        self.saved = None
        return dxin
