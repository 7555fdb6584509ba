"""HRNet-W48 + OCR detector (BASELINE config 4) on the HIP engine, explicit backward.

Counterpart of /root/reference/model/modeling/hrnet_ocr/nets/hrnet.py:101-158 (HRNet_W48_OCR),
backbones/hrnet/hrnet_backbone.py:36-106, 108-297, 295-545 (BasicBlock, Bottleneck, HighResolutionModule,
HighResolutionNet hrnet48) and modules/spatial_ocr_block.py:37-66, 114-305 (SpatialGather_Module, _ObjectAttentionBlock,
SpatialOCR_Module) with the reference's parameter names.

With the path's single class the OCR attention collapses exactly: the object-region representation is one vector per
sample (softmax-over-pixels pooling of the 512-channel map, a HIP kernel), ``softmax`` over the single region is 1, so the
"context" map is the per-sample vector f_up(f_down(ctx)) broadcast over the grid.  That vector chain ([B, 512] fp32, BatchNorm
over the B region vectors) is a handful of tiny fp32 matrix products and is evaluated with torch ops on the device; it enters
the 1024 -> 512 fuse conv as a stride-0 input segment.  f_pixel / f_object cannot influence the output (their gradients are
identically zero in the reference too) but their BatchNorm running statistics are updated like the reference's.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from .. import _lib as L
from ..engine import FM, Conv, BatchNorm, grad_acc, pad8, _ptr
from .pspnet import ConvBN
from .shapes import HRNET_W48

DROP_P = {"ocr_drop": 0.05}        # spatial_ocr_block.py:283 / nets/hrnet.py:125
DROP_C = {"ocr_drop": 512}


class CBN:
    """conv (+bias) -> train-mode BatchNorm -> optional residual add -> optional ReLU, with its backward."""

    def __init__(self, eng, P, conv, norm, k, stride=1, bias=False, relu=True):
        self.u = ConvBN(eng, P, conv, norm, None, P[conv + ".weight"].shape[0], k, stride, (k - 1) // 2, bias=bias)
        self.eng, self.act = eng, L.ACT_RELU if relu else L.ACT_NONE
        self.sv = None

    @property
    def conv(self):
        return self.u.conv

    def fwd(self, x, training, res=None, out=None, drop=None, keep=True):
        raw, m, iv = self.u.fwd(x, training)
        y = self.u.bn.apply(raw, m, iv, act=self.act, res=res, drop=drop, out=out)
        self.sv = (x, raw, m, iv, res, drop) if keep else None
        return y

    def bwd(self, dy, dres=None, dres_acc=False, dx_out=None, dx_acc=False, need_dx=True):
        x, raw, m, iv, res, drop = self.sv
        self.sv = None
        draw = self.u.bn.backward(dy, raw, m, iv, act=self.act, res=res, drop=drop, dres=dres, dres_acc=dres_acc)
        self.conv.bwd_weights(draw, x)
        if self.conv.b is not None:
            grad_acc(self.conv.b)          # bias feeding train-mode BN: identically zero gradient (not None)
        if not need_dx:
            return None
        xs = x if isinstance(x, (tuple, list)) else (x,)
        return self.conv.bwd_input(draw, out=dx_out, accumulate=dx_acc, in_hw=(xs[0].H, xs[0].W))


class _Basic:
    def __init__(self, eng, P, pre):
        self.c1 = CBN(eng, P, pre + ".conv1", pre + ".bn1", 3)
        self.c2 = CBN(eng, P, pre + ".conv2", pre + ".bn2", 3)
        self.eng = eng

    def cbns(self):
        return [self.c1, self.c2]

    def fwd(self, x, training):
        return self.c2.fwd(self.c1.fwd(x, training), training, res=x)

    def bwd(self, dy):
        dx = self.eng.new(dy.N, dy.H, dy.W, dy.c)
        da = self.c2.bwd(dy, dres=dx)
        self.c1.bwd(da, dx_out=dx, dx_acc=True)
        return dx


class _Bottleneck:
    def __init__(self, eng, P, pre, has_down):
        self.c1 = CBN(eng, P, pre + ".conv1", pre + ".bn1", 1)
        self.c2 = CBN(eng, P, pre + ".conv2", pre + ".bn2", 3)
        self.c3 = CBN(eng, P, pre + ".conv3", pre + ".bn3", 1)
        self.down = CBN(eng, P, pre + ".downsample.0", pre + ".downsample.1", 1, relu=False) if has_down else None
        self.eng = eng

    def cbns(self):
        return [self.c1, self.c2, self.c3] + ([self.down] if self.down else [])

    def fwd(self, x, training):
        res = self.down.fwd(x, training) if self.down else x
        return self.c3.fwd(self.c2.fwd(self.c1.fwd(x, training), training), training, res=res)

    def bwd(self, dy):
        dres = self.eng.new(dy.N, dy.H, dy.W, dy.c)
        d1 = self.c2.bwd(self.c3.bwd(dy, dres=dres))
        dx = self.down.bwd(dres) if self.down else dres
        self.c1.bwd(d1, dx_out=dx, dx_acc=True)
        return dx


class _HRModule:
    """4 BasicBlocks per branch, then the all-to-all fuse (hrnet_backbone.py:271-297)."""

    def __init__(self, eng, P, pre, chans):
        self.eng, self.nb = eng, len(chans)
        self.branches = [[_Basic(eng, P, f"{pre}.branches.{i}.{b}") for b in range(4)] for i in range(self.nb)]
        self.fuse = {}
        for i in range(self.nb):
            for j in range(self.nb):
                fp = f"{pre}.fuse_layers.{i}.{j}"
                if j > i:
                    self.fuse[i, j] = [CBN(eng, P, fp + ".0", fp + ".1", 1, relu=False)]
                elif j < i:
                    self.fuse[i, j] = [CBN(eng, P, f"{fp}.{k}.0", f"{fp}.{k}.1", 3, stride=2, relu=k != i - j - 1) for k in range(i - j)]
        self.sv = None

    def cbns(self):
        out = [c for br in self.branches for blk in br for c in blk.cbns()]
        for v in self.fuse.values():
            out += v
        return out

    def fwd(self, xs, training, outs=None):
        e = self.eng
        xs = list(xs)
        for i in range(self.nb):
            for blk in self.branches[i]:
                xs[i] = blk.fwd(xs[i], training)
        ys = []
        for i in range(self.nb):
            terms = []
            for j in range(self.nb):
                if j == i:
                    terms.append(xs[j])
                elif j > i:
                    t = self.fuse[i, j][0].fwd(xs[j], training)
                    terms.append(e.bilinear(t, xs[i].H, xs[i].W, True))
                else:
                    t = xs[j]
                    for c in self.fuse[i, j]:
                        t = c.fwd(t, training)
                    terms.append(t)
            ys.append(sum_act(e, terms, relu=True, out=None if outs is None else outs[i]))
        self.sv = (ys, [(x.H, x.W, x.c) for x in xs])
        return ys

    def bwd(self, dys):
        """dys[i]: gradient wrt output i (consumed: the ReLU mask is applied in place)."""
        e = self.eng
        ys, shp = self.sv
        self.sv = None
        dxs = [None] * self.nb
        for i in range(self.nb):
            e.epilogue_bwd(dys[i], out=ys[i], act=L.ACT_RELU, dpre=dys[i])
        for i in range(self.nb):                 # identity terms first: they create the accumulators
            dxs[i] = e.new(dys[i].N, dys[i].H, dys[i].W, dys[i].c)
            L.call("csbsr_axpby", dys[i].npix, dys[i].cp, _ptr(dys[i].t), dys[i].ld, 1.0, None, 0, 0.0, _ptr(dxs[i].t), dxs[i].ld, e.stream)
        for i in range(self.nb):
            for j in range(self.nb):
                if j > i:
                    H, W, c = shp[j]
                    dt = e.new(dys[i].N, H, W, dys[i].c)
                    e.bilinear_bwd(dys[i], dt, False, True)
                    self.fuse[i, j][0].bwd(dt, dx_out=dxs[j], dx_acc=True)
                elif j < i:
                    d = dys[i]
                    chain = self.fuse[i, j]
                    for k in range(len(chain) - 1, -1, -1):
                        d = chain[k].bwd(d, dx_out=dxs[j] if k == 0 else None, dx_acc=k == 0)
        for i in range(self.nb - 1, -1, -1):
            d = dxs[i]
            for blk in reversed(self.branches[i]):
                d = blk.bwd(d)
            dxs[i] = d
        return dxs


def sum_act(eng, terms, relu, out=None):
    t0 = terms[0]
    if out is None:
        out = eng.new(t0.N, t0.H, t0.W, t0.c, split=bool(t0.lo))
    n = len(terms)
    ptrs = (C.c_void_p * 4)(*[_ptr(t.t) for t in terms], *([None] * (4 - n)))
    lds = (C.c_int64 * 4)(*[t.ld for t in terms], *([0] * (4 - n)))
    los = (C.c_int64 * 4)(*[t.lo for t in terms], *([0] * (4 - n)))
    L.call("csbsr_sum_act_split", t0.npix, t0.cp, n, ptrs, lds, los, _ptr(out.t), out.ld, out.lo, int(relu), eng.stream)
    return out


class _VecBN:
    """BatchNorm2d applied to a map that is constant over space: [B, C] region vectors standing for [B, C, h, w] with n_sp
    identical positions (n_sp = 1 for the 1x1 proxy).  Batch statistics over B; the running-variance update uses the
    unbiased factor of the real element count B * n_sp, as F.batch_norm would."""

    def __init__(self, P, name):
        self.P, self.name = P, name

    def __call__(self, x, leaves, training, n_sp):
        P, nm = self.P, self.name
        g = leaves.setdefault(nm + ".weight", P[nm + ".weight"].detach().clone().requires_grad_(True))
        b = leaves.setdefault(nm + ".bias", P[nm + ".bias"].detach().clone().requires_grad_(True))
        if not training:
            return (x - P[nm + ".running_mean"]) * torch.rsqrt(P[nm + ".running_var"] + 1e-5) * g + b
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        with torch.no_grad():
            n = x.shape[0] * n_sp
            P[nm + ".running_mean"].mul_(0.9).add_(0.1 * mean)
            P[nm + ".running_var"].mul_(0.9).add_(0.1 * var * (n / (n - 1.0)))
            P[nm + ".num_batches_tracked"].add_(1)
        return (x - mean) * torch.rsqrt(var + 1e-5) * g + b


class HRNetOCR:
    drop_keys = ("ocr_drop",)

    def __init__(self, eng, params, prefix="segmentation_model"):
        self.eng, self.P, self.prefix = eng, params, prefix
        e, P, b = eng, params, prefix + ".backbone"
        self.stem = [CBN(e, P, b + ".conv1", b + ".bn1", 3, stride=2), CBN(e, P, b + ".conv2", b + ".bn2", 3, stride=2)]
        self.layer1 = [_Bottleneck(e, P, f"{b}.layer1.{i}", i == 0) for i in range(4)]
        self.trans, self.stages = [], []
        prev = (256,)
        for si, (stage, nmod, chans) in enumerate(HRNET_W48, 1):
            tp = f"{b}.transition{si}"
            tr = []
            for i, c in enumerate(chans):
                if i < len(prev):
                    tr.append(CBN(e, P, f"{tp}.{i}.0", f"{tp}.{i}.1", 3) if prev[i] != c else None)
                else:
                    tr.append(CBN(e, P, f"{tp}.{i}.0.0", f"{tp}.{i}.0.1", 3, stride=2))
            self.trans.append(tr)
            self.stages.append([_HRModule(e, P, f"{b}.{stage}.{m}", chans) for m in range(nmod)])
            prev = chans
        self.chans = prev
        self.aux0 = CBN(e, P, prefix + ".aux_head.0", prefix + ".aux_head.1.0", 3, bias=True)
        self.aux2 = Conv(e, prefix + ".aux_head.2", P, 1, bias=True)
        self.conv3 = CBN(e, P, prefix + ".conv3x3.0", prefix + ".conv3x3.1.0", 3, bias=True)
        ob = prefix + ".ocr_distri_head.object_context_block"
        self.ob = ob
        self.f_pixel = [CBN(e, P, ob + ".f_pixel.0", ob + ".f_pixel.1.0", 1, bias=True),
                        CBN(e, P, ob + ".f_pixel.2", ob + ".f_pixel.3.0", 1, bias=True)]
        cd = prefix + ".ocr_distri_head.conv_bn_dropout"
        self.fuse = ConvBN(e, P, cd + ".0", cd + ".1.0", 1024, 512, 1, 1, 0, bias=True)
        self.fuse.conv.split = (512, 512)
        self.cls = Conv(e, prefix + ".cls_head", P, 1, bias=True)
        self.saved = None
        self.split = False          # detector_precision == "split" (set by the model; the maps carry it: FM.lo)

    # ------------------------------------------------------------------ bookkeeping
    def _cbns(self):
        out = list(self.stem)
        for blk in self.layer1:
            out += blk.cbns()
        for tr in self.trans:
            out += [t for t in tr if t is not None]
        for st in self.stages:
            for m in st:
                out += m.cbns()
        return out + [self.aux0, self.conv3] + self.f_pixel

    def all_convs(self):
        return [c.conv for c in self._cbns()] + [self.aux2, self.fuse.conv, self.cls]

    def invalidate(self):
        for c in self.all_convs():
            c.invalidate()

    def make_dropout(self, B, training, enabled=True):
        if not training or not enabled:
            return {k: None for k in DROP_P}
        return {k: ((torch.rand(B, DROP_C[k], device=self.eng.device) >= p).to(torch.float32) / (1.0 - p)).contiguous()
                for k, p in DROP_P.items()}

    # ------------------------------------------------------------------ the region-vector chain (tiny, fp32 torch ops on device)
    def _vec_chain(self, ctx, training, hw):
        """ctx [B, 512] fp32 (requires grad) -> context vector [B, 512]; f_object is evaluated for its BatchNorm statistics only."""
        P, ob = self.P, self.ob
        leaves = {}

        def lin(name, x):
            w = leaves.setdefault(name + ".weight", P[name + ".weight"].detach().clone().requires_grad_(True))
            b = leaves.setdefault(name + ".bias", P[name + ".bias"].detach().clone().requires_grad_(True))
            return F.linear(x, w.flatten(1), b)

        def unit(seq, idx, x, n_sp):
            return torch.relu(_VecBN(P, f"{ob}.{seq}.{idx + 1}.0")(lin(f"{ob}.{seq}.{idx}", x), leaves, training, n_sp))
        with torch.no_grad():
            unit("f_object", 2, unit("f_object", 0, ctx.detach(), 1), 1)
        for k in [k for k in leaves if ".f_object." in k]:
            del leaves[k]
        value = unit("f_down", 0, ctx, 1)
        cvec = unit("f_up", 0, value, hw)
        return cvec, leaves

    # ------------------------------------------------------------------ forward
    def forward(self, xin, drop, training=True, kvec=None):
        """xin: FM [B,H,W,8] (normalised SR image).  Returns (seg32, aux32) fp32 [B,1,H,W] probability maps."""
        cat, ysz = self._backbone_fwd(xin, training)
        seg32, aux32 = self._head_fwd(cat, drop, training, xin.H, xin.W)
        self.saved.update(xin=xin, ysz=ysz)
        return seg32, aux32

    def _backbone_fwd(self, xin, training):
        """HighResolutionNet.forward + the bilinear concat of nets/hrnet.py:143-149 -> 720-channel map at 1/4 resolution."""
        e = self.eng
        B = xin.N
        x = self.stem[1].fwd(self.stem[0].fwd(xin, training), training)
        for blk in self.layer1:
            x = blk.fwd(x, training)
        ys = [x]
        cat = None
        for si, (tr, mods) in enumerate(zip(self.trans, self.stages)):
            xs = []
            for i, t in enumerate(tr):
                src = ys[i] if i < len(ys) else ys[-1]
                xs.append(t.fwd(src, training) if t is not None else src)
            for mi, m in enumerate(mods):
                outs = None
                if si == len(self.stages) - 1 and mi == len(mods) - 1:      # branch 0 of the last module lands in the 720-ch concat
                    cat = e.new(B, xs[0].H, xs[0].W, sum(self.chans), split=bool(xs[0].lo))
                    outs = [cat.slice(0, self.chans[0])] + [None] * (len(xs) - 1)
                xs = m.fwd(xs, training, outs)
            ys = xs
        h, w = ys[0].H, ys[0].W
        off = self.chans[0]
        for j in range(1, 4):
            e.bilinear(ys[j], h, w, True, out=cat.slice(off, off + self.chans[j]))
            off += self.chans[j]
        return cat, [(y.H, y.W) for y in ys]

    def _head_fwd(self, cat, drop, training, H, W):
        """aux head, 3x3 reduction, OCR (gather + distribute), class head, bilinear up + sigmoid (nets/hrnet.py:150-158)."""
        e = self.eng
        B, h, w = cat.N, cat.H, cat.W
        a = self.aux0.fwd(cat, training)
        aux_lo = e.f32(B, 1, h, w, zero=False)
        self.aux2.fwd(a, out32=aux_lo)
        f = self.conv3.fwd(cat, training)
        # soft object region + pooled region vector
        probs = torch.softmax(aux_lo.reshape(B, -1), dim=1).contiguous()
        ctx = e.f32(B, 512)
        L.call("csbsr_weighted_pool_fwd_split", _ptr(f.t), f.ld, f.lo, _ptr(probs), _ptr(ctx), B, h * w, 512, e.stream)
        keep = training and torch.is_grad_enabled()
        ctx_leaf = ctx.detach().requires_grad_(keep)
        with torch.enable_grad() if keep else torch.no_grad():
            cvec, leaves = self._vec_chain(ctx_leaf, training, h * w)
        # f_pixel (query transform): dead for one region, evaluated for its running statistics
        self.f_pixel[1].fwd(self.f_pixel[0].fwd(f, training, keep=False), training, keep=False)
        ct = torch.zeros(B, 1, 1, 512, dtype=torch.float16, device=e.device)
        ct[:, 0, 0] = cvec.detach().to(torch.float16)
        cfm = FM(ct, 512, bcast=True, H=h, W=w)
        if f.lo:     # split-fp16 features: the constant context segment of the 1x1 fuse conv is folded into a per-sample fp32 bias
            if training:
                stat = self.fuse.bn.new_stat()
                raw = self.fuse.conv.fwd_const_1x1(cvec.detach(), f, stat=stat, stat_mode=L.STAT_BN)
                m, iv = self.fuse.bn.finalize(stat, raw.npix, update_running=True)
            else:
                raw = self.fuse.conv.fwd_const_1x1(cvec.detach(), f)
                m, iv = self.fuse.bn.rmean, torch.rsqrt(self.fuse.bn.rvar + 1e-5)
        else:
            raw, m, iv = self.fuse.fwd((cfm, f), training)
        o = self.fuse.bn.apply(raw, m, iv, act=L.ACT_RELU, drop=drop["ocr_drop"])
        cls_lo = e.f32(B, 1, h, w, zero=False)
        self.cls.fwd(o, out32=cls_lo)
        seg32, aux32 = e.f32(B, 1, H, W, zero=False), e.f32(B, 1, H, W, zero=False)
        for lo, hi in ((cls_lo, seg32), (aux_lo, aux32)):
            L.call("csbsr_bilinear32_fwd", _ptr(lo), _ptr(hi), B, h, w, H, W, 1, e.stream)
            hi.sigmoid_()
        self.saved = dict(cat=cat, a=a, f=f, probs=probs, ctx=ctx_leaf, cvec=cvec, leaves=leaves, cfm=cfm, fuse=(raw, m, iv), o=o,
                          drop=drop["ocr_drop"], seg32=seg32, aux32=aux32, hw=(h, w))
        return seg32, aux32

    # ------------------------------------------------------------------ backward
    def _head_dlogit(self, dprob32, prob32, h, w):
        """d(prob) at HR -> d(logit) at the head's resolution, fp32 [B,1,h,w]."""
        e = self.eng
        B, _, H, W = prob32.shape
        g = (dprob32 * prob32 * (1.0 - prob32)).contiguous()
        lo = e.f32(B, 1, h, w, zero=False)
        L.call("csbsr_bilinear32_bwd", _ptr(g), _ptr(lo), B, h, w, H, W, 1, e.stream)
        return lo

    def _logit_head_bwd(self, conv, dlogit32, x):
        e = self.eng
        dpre = e.nchw32_to_fm(dlogit32)
        e.epilogue_bwd(dpre, dbias=grad_acc(conv.b), creal=1)
        conv.bwd_weights(dpre, x)
        return conv.bwd_input(dpre)

    def backward(self, dseg32, daux32):
        sv = self.saved
        dcat = self._head_bwd(dseg32, daux32)
        self.saved = None
        return self._backbone_bwd(dcat, sv["ysz"])

    def _head_bwd(self, dseg32, daux32):
        e, sv, P = self.eng, self.saved, self.P
        B = sv["cat"].N
        h, w = sv["hw"]
        f, cat = sv["f"], sv["cat"]
        # main head
        do = self._logit_head_bwd(self.cls, self._head_dlogit(dseg32, sv["seg32"], h, w), sv["o"])
        raw, m, iv = sv["fuse"]
        draw = self.fuse.bn.backward(do, raw, m, iv, act=L.ACT_RELU, drop=sv["drop"])
        self.fuse.conv.bwd_weights(draw, (sv["cfm"], f))
        grad_acc(self.fuse.conv.b)
        df = self.fuse.conv.bwd_input(draw, seg=1)
        dcvec = e.f32(B, 512)
        self.fuse.conv.bwd_input(draw, seg=0, stat=dcvec)
        del draw, do
        # region-vector chain (scaled gradients flow through unchanged: everything here is linear in them)
        leaves = sv["leaves"]
        names = list(leaves)
        grads = torch.autograd.grad(sv["cvec"], [sv["ctx"]] + [leaves[k] for k in names], grad_outputs=dcvec[:, :512].contiguous(),
                                    allow_unused=True)
        dctx = grads[0].contiguous()
        for k, g in zip(names, grads[1:]):
            acc = grad_acc(P[k])
            if g is not None:
                acc.add_(g.reshape(acc.shape))
        for c in self.f_pixel:                       # zero gradients (softmax over a single region), as in the reference
            grad_acc(c.conv.w), grad_acc(c.conv.b), grad_acc(c.u.bn.gamma), grad_acc(c.u.bn.beta)
        for seq, idx in (("f_object", 0), ("f_object", 2)):
            for suffix in (f"{idx}.weight", f"{idx}.bias", f"{idx + 1}.0.weight", f"{idx + 1}.0.bias"):
                grad_acc(P[f"{self.ob}.{seq}.{suffix}"])
        # pooling adjoint: df += probs (x) dctx ; dprobs = f . dctx ; softmax backward on the aux logits
        dprobs = e.f32(B, h * w, zero=False)
        L.call("csbsr_weighted_pool_bwd", _ptr(f.t), f.ld, _ptr(sv["probs"]), _ptr(dctx), _ptr(df.t), df.ld, _ptr(dprobs), B, h * w, 512, e.stream)
        probs = sv["probs"]
        dlog_gather = probs * (dprobs - (probs * dprobs).sum(1, keepdim=True))
        dcat = self.conv3.bwd(df)
        del df
        # aux head (its logits also feed the region softmax)
        dlog_aux = self._head_dlogit(daux32, sv["aux32"], h, w)
        dlog_aux += dlog_gather.reshape(B, 1, h, w)
        da = self._logit_head_bwd(self.aux2, dlog_aux, sv["a"])
        self.aux0.bwd(da, dx_out=dcat, dx_acc=True)
        return dcat

    def _backbone_bwd(self, dcat, ysz):
        e = self.eng
        B = dcat.N
        # concat -> branch gradients
        dys = [dcat.slice(0, self.chans[0])]
        off = self.chans[0]
        for j in range(1, 4):
            Hj, Wj = ysz[j]
            d = e.new(B, Hj, Wj, self.chans[j])
            e.bilinear_bwd(dcat.slice(off, off + self.chans[j]), d, False, True)
            dys.append(d)
            off += self.chans[j]
        # stages, transitions
        for si in range(len(self.stages) - 1, -1, -1):
            for m in reversed(self.stages[si]):
                dys = m.bwd(dys)
            tr = self.trans[si]
            n_prev = len(tr) - 1 if si > 0 else 1
            dprev = [None] * n_prev
            for i, t in enumerate(tr):
                src = i if i < n_prev else n_prev - 1
                if t is None:
                    assert dprev[src] is None
                    dprev[src] = dys[i]
            for i, t in enumerate(tr):
                if t is None:
                    continue
                src = i if i < n_prev else n_prev - 1
                if dprev[src] is None:
                    dprev[src] = t.bwd(dys[i])
                else:
                    t.bwd(dys[i], dx_out=dprev[src], dx_acc=True)
            dys = dprev
        d = dys[0]
        for blk in reversed(self.layer1):
            d = blk.bwd(d)
        d = self.stem[1].bwd(d)
        return self.stem[0].bwd(d)
