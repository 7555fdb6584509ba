"""KBPN blind-SR network on the HIP engine: explicit forward and backward.

Mirrors the reference module tree (parameter names, phase behaviour) of
/root/reference/model/modeling/kbpn.py -- KBPN.forward :84-116, KernelBackProjectionStageWithSFT.forward
:172-189, UpBlock :450-469, DownBlock :472-489, KBlock :380-409, KernelPredictorLikeIKC :562-578,
SFTlayer :509-518, predictor_withGAP :322-341 -- but none of its structure: concat buffers are written in
place, the kernel-code maps are stride-0 convolution segments, every conv epilogue is fused, and the
backward pass is hand-scheduled (gradient fan-in by accumulate flags, no autograd tape).
"""
import ctypes as C

import os

import torch
import torch.nn.functional as F

from .. import _lib as L
from ..engine import FM, Conv, ShuffleConv, grad_acc, pad8, _ptr
from .shapes import CONV_SETTING

A_NONE, A_RELU, A_LRELU, A_PRELU, A_SIG = L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU, L.ACT_PRELU, L.ACT_SIGMOID


class _Stage:
    pass


def border_tap_mask():
    """Mtap[a, k] = 1 when tap k of a zero-padded 3-tap window lands inside the image for an output coordinate of border class a
    (a = first * 2 + last: 0 interior, 1 last, 2 first, 3 first and last)."""
    m = torch.ones(4, 3)
    m[2, 0] = m[3, 0] = 0.0          # first row/col: tap 0 reads outside
    m[1, 2] = m[3, 2] = 0.0          # last row/col: tap 2 reads outside
    return m


def ring_tap_classes():
    """Rtap[t, k, a] = 1 when tap k of an output coordinate of two-ring type t (0, 1, 2, 3, 4 = coordinate 0, 1, interior, size-2,
    size-1; size >= 5) reads an input coordinate of border class a (0 interior, 1 last, 2 first); out-of-image taps have no entry."""
    r = torch.zeros(5, 3, 4)
    r[0, 1, 2] = r[0, 2, 0] = 1.0                       # coordinate 0: tap 0 is outside, tap 1 reads the first row, tap 2 an interior row
    r[1, 0, 2] = r[1, 1, 0] = r[1, 2, 0] = 1.0
    r[2, :, 0] = 1.0
    r[3, 0, 0] = r[3, 1, 0] = r[3, 2, 1] = 1.0
    r[4, 0, 0] = r[4, 1, 1] = 1.0                       # last coordinate: tap 2 is outside
    return r


def _act_fn(conv):
    if conv.act == A_LRELU:
        return lambda t: F.leaky_relu(t, conv.slope)
    return F.relu if conv.act == A_RELU else (lambda t: t)


def kernel_branch_table(kv, w0, w1, w_tail, mtap, rtap, act0, act1):
    """[B, 5, 5, cout] table of  W_tail . act1(conv3x3(act0(conv3x3(kv expanded over the image, w0)), w1))  per two-ring class of the pixel
    (both convs zero-padded, stride 1): plain differentiable torch ops on a handful of small tensors.
    kv [B, cin] the spatially constant input, w0 [c0, cin, 3, 3], w1 [c1, c0, 3, 3], w_tail [cout, c1]."""
    T = torch.einsum("ocyx,nc->noyx", w0, kv)                               # per-tap response to the constant code
    V1 = act0(torch.einsum("noyx,ay,bx->nabo", T, mtap, mtap))               # [B, 4, 4, c0] border-class values of the first conv
    A1 = torch.einsum("tya,nabc->ntybc", rtap, V1)
    A2 = torch.einsum("uxb,ntybc->ntuyxc", rtap, A1)                         # [B, 5, 5, 3, 3, c0]: what each tap of the second conv reads
    V2 = act1(torch.einsum("ocyx,ntuyxc->ntuo", w1, A2))                     # [B, 5, 5, c1] two-ring-class values of the second conv
    return torch.einsum("oc,ntuc->ntuo", w_tail, V2)


class KBPN:
    def __init__(self, eng, params, cfg, prefix="sr_model"):
        self.eng, self.P, self.cfg, self.prefix = eng, params, cfg, prefix
        self.scale = cfg.scale
        self.S = cfg.num_stages
        self.K = cfg.ksize_out
        self.kk = self.K * self.K
        k, s, p = CONV_SETTING[self.scale]
        self.sft = bool(getattr(cfg, "kernel_sft", True))
        self.lr_err = bool(getattr(cfg, "lr_error", False))
        if self.lr_err and getattr(cfg, "pixel_shuffle", False):
            raise NotImplementedError("MODEL.SUM_LR_ERROR_POS='LR' together with MODEL.SR_PIXEL_SHUFFLE is not built")
        md = 128
        kc = cfg.ksize * cfg.ksize
        e, P = eng, params
        self.layers = []

        def mk(name, *a, **kw):
            c = Conv(e, name, P, *a, **kw)
            self.layers.append(c)
            return c

        def block(name, kk_, st, pd, act, transposed=False, bias=False, slope=0.01, split=None):
            if transposed and getattr(cfg, "pixel_shuffle", False):      # MODEL.SR_PIXEL_SHUFFLE: conv3x3 + PixelShuffle(scale)
                c = ShuffleConv(e, name + ".layer", P, self.scale, bias=bias, act=act, slope=slope,
                                prelu=(name + ".act.weight") if act == A_PRELU else False)
                self.layers.append(c)
                return c
            return mk(name + ".layer", kk_, st, pd, 1, transposed=transposed, bias=bias, act=act, slope=slope,
                      prelu=(name + ".act.weight") if act == A_PRELU else False, split=split)

        self.feat = [mk(f"{prefix}.feat.{i}", 3, 1, 1, act=A_RELU) for i in (0, 2, 4, 6)]
        self.pred = [block(f"{prefix}.predictor.feat_ext.{i}", 3, 1, 1, A_PRELU) for i in range(3)]
        self.stages = []
        for st in range(1, self.S + 1):
            sp = f"{prefix}.back_projection_stages.{st - 1}"
            o = _Stage()
            o.up_conv = block(sp + ".up.conv", 1, 1, 0, A_PRELU, bias=True)
            o.up1 = block(sp + ".up.up_conv1", k, s, p, A_PRELU, transposed=True)
            o.up2 = block(sp + ".up.up_conv2", k, s, p, A_PRELU)
            o.up3 = block(sp + ".up.up_conv3", k, s, p, A_PRELU, transposed=True)
            o.sr_reconst = block(sp + ".kb.sr_reconst", 3, 1, 1, A_NONE, split=(md * (st - 1), md) if st > 1 else None)
            kp = sp + ".kb.kernel_predictor"
            o.fe_sr = [block(kp + ".fe_SR.0", 3, 1, 1, A_RELU), block(kp + ".fe_SR.1", 1, 1, 0, A_LRELU),
                       block(kp + ".fe_SR.2", 3, 1, 1, A_LRELU), block(kp + ".fe_SR.3", 3, 1, 1, A_LRELU),
                       block(kp + ".fe_SR.4", 3, 1, 1, A_LRELU)]
            o.fe_k = [block(kp + ".fe_kernel.0", 3, 1, 1, A_LRELU), block(kp + ".fe_kernel.1", 3, 1, 1, A_LRELU)]
            o.fe_cat = [block(kp + ".fe_cat.0", 1, 1, 0, A_LRELU, split=(kc, kc)), block(kp + ".fe_cat.1", 3, 1, 1, A_LRELU),
                        block(kp + ".fe_cat.2", 3, 1, 1, A_NONE)]
            if self.lr_err:      # MODEL.SUM_LR_ERROR_POS = 'LR' (kbpn.py:369-374): a 3x3 conv of the LR error, added to the next stage's LR features
                o.kb_up, o.kb_conv = None, block(sp + ".kb.conv", 3, 1, 1, A_NONE)
            else:
                o.kb_up, o.kb_conv = block(sp + ".kb.up_conv1", k, s, p, A_PRELU, transposed=True), None
            if st < self.S:
                o.down_conv = block(sp + ".down.conv", 1, 1, 0, A_PRELU, bias=True)
                o.down1 = block(sp + ".down.down_conv1", k, s, p, A_PRELU)
                o.down2 = block(sp + ".down.down_conv2", k, s, p, A_PRELU, transposed=True)
                o.down3 = block(sp + ".down.down_conv3", k, s, p, A_PRELU)
                sf = sp + ".sft"
                if not self.sft:     # MODEL.KBPN_KERNEL_SFT = False (kbpn.py:169-171,190): the concatenated LR features go to the next stage as they are
                    self.stages.append(o)
                    continue
                o.sc0 = mk(sf + ".SFT_scale_conv0", 3, 1, 1, act=A_LRELU, slope=0.1, split=(md * st, self.kk))
                o.sc1 = mk(sf + ".SFT_scale_conv1", 3, 1, 1, act=A_SIG)
                o.sh0 = mk(sf + ".SFT_shift_conv0", 3, 1, 1, act=A_LRELU, slope=0.1, split=(md * st, self.kk))
                o.sh1 = mk(sf + ".SFT_shift_conv1", 3, 1, 1, act=A_NONE)
                for c_ in (o.sc0, o.sc1, o.sh0, o.sh1):
                    c_.winograd = True          # the layers the Winograd study covers (tests/study_winograd.py): engine.Conv._launch, csrc/conv_x3w.hip
            self.stages.append(o)
        self.output_conv = block(f"{prefix}.output_conv", 3, 1, 1, A_NONE)
        # compensation of the forward weights' fp16 rounding (Conv._dc_bias) on the (non-transposed) layers whose kernels take a
        # per-channel bias at no cost: the VGG head, the initial predictor, the strided and 1x1 convolutions of the Up / Down blocks, SFT
        # layers and the 3-channel image heads.  Not the per-stage kernel
        # predictors (full-resolution 32 / 49-channel layers on the bias-free direct kernels): exempting their weights from rounding
        # altogether moves nothing.
        for c in self.layers:
            c.dc_comp = ".kernel_predictor." not in c.name
        # bicubic 7x7 -> 21x21 as a fixed linear map U [kk, kc] (nn.Upsample(size, 'bicubic'), kbpn.py:317,558)
        eye = torch.eye(kc).reshape(kc, 1, cfg.ksize, cfg.ksize)
        up = F.interpolate(eye, size=(self.K, self.K), mode="bicubic", align_corners=False) if cfg.ksize != self.K else eye
        self.U = up.reshape(kc, self.kk).t().contiguous().to(eng.device)       # [kk, kc]
        # MODEL.ZERO_PAD_KERNEL (kbpn.py:543-554,583-596): per sample, a three-layer MLP on the stage predictor's 7x7 update picks centred zero
        # padding instead of the bicubic map (hard threshold on .item(): no gradient reaches the MLP); Z is that padding as a [kk, kc] matrix
        self.zero_pad = bool(getattr(cfg, "zero_pad_kernel", False))
        self.pad_dropout = True        # the MLP's two nn.Dropout(0.2) in training (the owner turns it off with its own dropout switch)
        self._pad_replay, self.pad_taken = None, []
        pz = (self.K - cfg.ksize) // 2
        self.Z = F.pad(eye, (pz, pz, pz, pz)).reshape(kc, self.kk).t().contiguous().to(eng.device)
        if self.zero_pad:
            for st_i, o in enumerate(self.stages):
                pd = f"{prefix}.back_projection_stages.{st_i}.kb.kernel_predictor.pad_descriminator"
                o.pad_disc = [(P[f"{pd}.{i}.weight"], P[f"{pd}.{i}.bias"]) for i in (0, 3, 6)]
        self.kc = kc
        self.saved = None
        self.training_mode = True      # set by the owner (nn.Module.training of the model)
        self._gather = {}
        self.gather = os.environ.get("CSBSR_GATHER_DCH", "1") != "0"      # backward: gather each stage's slice of d(concat_h) (see _gather_conv)
        self.Mtap = border_tap_mask().to(eng.device)       # folded constant-operand convs: see _kernel_branch_fwd / Conv.fwd_folded
        self.Rtap = ring_tap_classes().to(eng.device)

    # ------------------------------------------------------------------ phase logic (kbpn.py:118-142, 414-447)
    def set_phase(self, it):
        c = self.cfg
        sr_pre = c.sr_pretrain[0] <= it < c.sr_pretrain[1]
        k_pre = c.kernel_pretrain[0] <= it < c.kernel_pretrain[1]
        self.use_predictor = not sr_pre
        for l in self.layers:
            n = l.name
            is_kernel = ("kernel_predictor" in n) or (".predictor." in n)
            if sr_pre:
                l.frozen = is_kernel
            elif k_pre:
                l.frozen = not is_kernel
            else:
                l.frozen = False

    def invalidate(self):
        for l in self.layers:
            l.invalidate()
        self._gather = {}

    def _gather_conv(self, kind, s):
        """The gradient of stage ``s``'s 128-channel slice of the concatenated HR features (``concat_h``, kbpn.py:355-372) is a sum of
        convolutions of small maps: the 3-channel dPre of ``output_conv`` and of every later stage's ``sr_reconst`` through their 3x3
        weights ("img"), and the 128-channel dPre of this and every later stage's ``down.conv`` through their 1x1 weights ("down").
        Autograd adds them one producer at a time -- each a read-modify-write of a multi-GB slice of the gradient buffer.  Here the
        producers' dPre maps are kept side by side as channel slots of one buffer each, the matching weight slices are stacked into
        ONE gradient-free conv per kind and stage (rebuilt when the master weights change), and the slice is produced when its stage
        needs it: one write and one read-modify-write instead of up to seven."""
        c = self._gather.get((kind, s))
        if c is None:
            sl = slice(128 * (s - 1), 128 * s)
            if kind == "img":        # slot 0: output_conv, slot t: sr_reconst of stage S - t + 1 (t = 1 .. S - s)
                ws = [self.output_conv.w[:, sl]] + [self.stages[sp - 1].sr_reconst.w[:, sl] for sp in range(self.S, s, -1)]
                if len(ws) == 1:
                    w = ws[0].detach().contiguous()
                else:                # 3 real + 5 zero rows per slot: the slots are 8-channel aligned (FM channel granularity); always the
                    # whole (zero-filled) slot buffer, so every stage's launch has the 32-channel shape the direct HR kernel takes
                    w = torch.zeros(8 * self.S, 128, 3, 3, dtype=torch.float32, device=self.eng.device)
                    for t, wt in enumerate(ws):
                        w[8 * t:8 * t + 3] = wt.detach()
                c = Conv(self.eng, "gather", {"gather.weight": w}, 3, 1, 1, bias=False)
            else:                    # slot u: down.conv of stage S - 1 - u (u = 0 .. S - 1 - s)
                w = torch.cat([self.stages[sp - 1].down_conv.w[:, sl].detach() for sp in range(self.S - 1, s - 1, -1)], 0).contiguous()
                c = Conv(self.eng, "gather", {"gather.weight": w}, 1, 1, 0, bias=False)
            self._gather[(kind, s)] = c
        return c

    # ------------------------------------------------------------------ helpers
    def _kfm(self, vec, H, W):
        t = torch.zeros(vec.shape[0], 1, 1, pad8(self.kk), dtype=torch.float16, device=vec.device)
        t[:, 0, 0, :self.kk] = vec.to(torch.float16)
        return FM(t, self.kk, bcast=True, H=H, W=W)

    def _bcast_grad(self, g, H, W):
        """[B, c] fp32 (already scaled) -> stride-0 FM used as dPre of a GAP'ed conv output."""
        t = torch.zeros(g.shape[0], 1, 1, pad8(g.shape[1]), dtype=torch.float16, device=g.device)
        t[:, 0, 0, :g.shape[1]] = g.to(torch.float16)
        return FM(t, g.shape[1], bcast=True, H=H, W=W)

    # ------------------------------------------------------------------ forward
    def forward(self, x32, it, kernel_gt, save=True, lean=False, pad_replay=None):
        """x32: fp32 NCHW LR batch on device.  Returns (sr32 [B,3,H,W] fp32, kvec [B,kk] fp32 normalised).
        ``pad_replay``: MODEL.ZERO_PAD_KERNEL only -- the per-stage pad decisions (``self.pad_taken`` of an earlier forward of the same
        micro-batch) to replay instead of evaluating the pad discriminator again: its nn.Dropout layers make the hard choice random in
        training, and a forward recomputed inside the backward must be the SAME function as the one that produced the losses.
        ``lean``: do not keep the fe_SR chain of the per-stage kernel predictors for the backward (208 of the ~1040 HR channel planes a
        stage saves: 5.3 of 26.5 GB per image at HR 1792^2); the backward rebuilds it from the saved 3-channel SR estimate with five thin
        convolutions per stage (~2 % of a step, bit-identical values: the path is order-fixed)."""
        e = self.eng
        self.lean = bool(lean)
        self._pad_replay, self.pad_taken = pad_replay, []
        self.set_phase(it)
        B, _, h, w = x32.shape
        H, W = h * self.scale, w * self.scale
        sv = {"x32": x32, "B": B, "h": h, "w": w}
        x16 = e.nchw32_to_fm(x32)
        f = x16
        feats = [x16]
        for c in self.feat:
            f = c.fwd(f)
            feats.append(f)
        sv["feats"] = feats
        init_f = f
        if self.use_predictor:
            z = init_f
            zs = [z]
            for c in self.pred[:2]:
                z = c.fwd(z)
                zs.append(z)
            gap = e.f32(B, pad8(self.kc))
            z3 = self.pred[2].fwd(z, stat=gap, stat_mode=L.STAT_SAMPLE_SUM)
            zs.append(z3)
            v49 = gap[:, :self.kc] / float(h * w)
            k441 = v49 @ self.U.t()
            ksum = k441.sum(1, keepdim=True)
            kvec = k441 / ksum
            sv["pred"] = (zs, k441, ksum)
        else:
            kvec = kernel_gt.reshape(B, -1).to(torch.float32)
        concat_h = e.new(B, H, W, 128 * self.S)
        concat_l = e.new(B, h, w, 128 * (self.S - 1)) if self.S > 1 else None
        sv["concat_h"], sv["concat_l"] = concat_h, concat_l
        low = init_f
        stg = []
        for s in range(1, self.S + 1):
            st = self.stages[s - 1]
            q = {}
            q["low_in"] = low
            xu = st.up_conv.fwd(low)
            h0 = st.up1.fwd(xu)
            d = st.up2.fwd(h0, res=xu, res_mode=L.RES_SUB)
            hs = concat_h.slice(128 * (s - 1), 128 * s)
            # (LR-error variant: h leaves the KBlock unchanged, kbpn.py:407-409 -- the UpBlock writes the stage's slice itself)
            hh = st.up3.fwd(d, out=hs if self.lr_err else None, res=h0, res_mode=L.RES_ADD)
            q.update(xu=xu, h0=h0, d=d, h=hh)
            segs = (concat_h.slice(0, 128 * (s - 1)), hh) if s > 1 else (hh,)
            sr_t32 = e.f32(B, 3, H, W, zero=False)
            sr_t16 = e.new(B, H, W, 3)
            st.sr_reconst.fwd(segs, out=sr_t16, out32=sr_t32)
            q.update(sr_t32=sr_t32, sr_t16=sr_t16, kvec_in=kvec)
            if self.use_predictor:
                kvec2 = self._kernel_predictor_fwd(st, q, sr_t16, kvec, H, W)
            else:
                kvec2 = kvec
            ksum = kvec2.sum(1, keepdim=True)
            vec = kvec2 / ksum
            q.update(kvec2=kvec2, ksum=ksum, vec=vec)
            err16 = e.new(B, h, w, 3, zero=True)
            L.call("csbsr_blur_fwd", _ptr(sr_t32), _ptr(vec.contiguous()), B, 3, H, W, self.K, self.scale, _ptr(x32), None,
                   _ptr(err16.t), err16.ld, e.stream)
            q["err16"] = err16
            if not self.lr_err:
                st.kb_up.fwd(err16, out=hs, res=hh, res_mode=L.RES_ADD)
            kvec = vec
            if s < self.S:
                chp = concat_h.slice(0, 128 * s)
                xd = st.down_conv.fwd(chp)
                l0 = st.down1.fwd(xd)
                dd = st.down2.fwd(l0, res=xd, res_mode=L.RES_SUB)
                lows = concat_l.slice(128 * (s - 1), 128 * s)
                if self.lr_err:          # low = down(concat_h) + conv(error)   (kbpn.py:183-185)
                    q["lowd"] = st.down3.fwd(dd, res=l0, res_mode=L.RES_ADD)
                    st.kb_conv.fwd(err16, out=lows, res=q["lowd"], res_mode=L.RES_ADD)
                else:
                    st.down3.fwd(dd, out=lows, res=l0, res_mode=L.RES_ADD)
                fpre = concat_l.slice(0, 128 * s)
                if not self.sft:
                    q.update(xd=xd, l0=l0, dd=dd)
                    low = fpre
                    stg.append(q if save else None)
                    continue
                # the 441 kernel-code channels of the SFT input are spatially constant: folded exactly into a class bias
                t1, fold1 = st.sc0.fwd_folded(fpre, vec, self.Mtap)
                sc = st.sc1.fwd(t1)
                t2, fold2 = st.sh0.fwd_folded(fpre, vec, self.Mtap)
                lowp = st.sh1.fwd(t2, res=fpre, res2=sc, res_mode=L.RES_FMA)
                q.update(xd=xd, l0=l0, dd=dd, fold1=fold1, fold2=fold2, t1=t1, sc=sc, t2=t2, lowp=lowp)
                low = lowp
            stg.append(q if save else None)
            if not save:
                del q
        sr32 = e.f32(B, 3, H, W, zero=False)
        self.output_conv.fwd(concat_h, out32=sr32)
        if getattr(self.cfg, "residual_learning", True):      # MODEL.SR_RESIDUAL_LEARNING (kbpn.py:112-116): sr += bicubic_up(x)
            L.call("csbsr_bicubic_up_add", _ptr(x32), _ptr(sr32), B * 3, h, w, self.scale, e.stream)
        sv["stages"] = stg
        self.saved = sv if save else None
        return sr32, kvec

    def _kernel_predictor_fwd(self, st, q, sr_t16, kvec, H, W):
        e = self.eng
        B = sr_t16.N
        a = [sr_t16]
        x = sr_t16
        for c in st.fe_sr:
            x = c.fwd(x)
            a.append(x)
        cb, kctx = self._kernel_branch_fwd(st, kvec, H, W)
        c1 = st.fe_cat[0].fwd_classbias(a[-1], cb, 1)
        c2 = st.fe_cat[1].fwd(c1)
        gap = e.f32(B, pad8(self.kc))
        st.fe_cat[2].fwd(c2, stat=gap, stat_mode=L.STAT_SAMPLE_SUM, store=False)
        d49 = gap[:, :self.kc] / float(H * W)
        q["kp"] = (a[:1] if getattr(self, "lean", False) else a, kctx, c1, c2)
        q["kp_map"] = self._update_map(st, d49)
        return kvec + torch.einsum("bc,bkc->bk", d49, q["kp_map"])

    def _update_map(self, st, d49):
        """[B, kk, kc] linear map from the predictor's 7x7 update to the 21x21 kernel update: bicubic (kbpn.py:558,598-599), or per sample
        the zero padding the pad discriminator picks (kbpn.py:583-596)."""
        B = d49.shape[0]
        if not self.zero_pad:
            return self.U.unsqueeze(0).expand(B, -1, -1)
        if self._pad_replay is not None:      # recomputed forward: the decision of the forward that produced the losses
            take_up = self._pad_replay[len(self.pad_taken)]
            self.pad_taken.append(take_up)
            return torch.where(take_up.reshape(B, 1, 1), self.U.unsqueeze(0), self.Z.unsqueeze(0))
        with torch.no_grad():
            drop = self.pad_dropout and self.training_mode
            hdn = d49
            for i, (w, b) in enumerate(st.pad_disc):
                hdn = F.linear(hdn, w.detach().to(torch.float32), b.detach().to(torch.float32))
                if i < 2:
                    hdn = F.dropout(F.relu(hdn), 0.2, training=drop)
            take_up = torch.sigmoid(hdn).reshape(B) >= 0.5
        self.pad_taken.append(take_up)
        return torch.where(take_up.reshape(B, 1, 1), self.U.unsqueeze(0), self.Z.unsqueeze(0))

    # ------------------------------------------------------------------ backward
    def _act_bwd(self, conv, dout, out, res=None, res2=None, res_mode=L.RES_NONE, dres=None, dres_acc=False, dres2=None,
                 dres2_acc=False, dpre=None):
        """dOut -> dPre for a fused conv epilogue (in place unless ``dpre`` is given); accumulates bias / PReLU-slope grads."""
        fz = conv.frozen
        self.eng.epilogue_bwd(dout, out=out, act=conv.act, slope=conv.slope, prelu=conv.prelu, res=res, res2=res2, res_mode=res_mode,
                              dpre=dout if dpre is None else dpre, dres=dres, dres_acc=dres_acc, dres2=dres2, dres2_acc=dres2_acc,
                              dbias=None if (conv.b is None or fz) else grad_acc(conv.b),
                              dprelu=None if (conv.prelu is None or fz) else grad_acc(conv.prelu), creal=conv.cout)
        return dout if dpre is None else dpre

    def _wg(self, conv, dpre, x):
        if not conv.frozen:
            conv.bwd_weights(dpre, x)

    def backward(self, dsr32, dkvec_final, stage_done=None):
        """dsr32: fp32 [B,3,H,W] gradient wrt sr (scaled by eng.grad_scale); dkvec_final: [B,kk] fp32 gradient wrt
        the returned normalised kernel vector (scaled).  Accumulates parameter gradients (scaled).
        ``stage_done(s)``: called when no later kernel of this backward writes the parameter gradients of back-projection stage s
        any more (s = S .. 1; output_conv counts with stage S) and with s = 0 after the predictor / VGG head: the data-parallel
        reducer launches that stage's bucket there."""
        e, sv = self.eng, self.saved
        B, h, w = sv["B"], sv["h"], sv["w"]
        H, W = h * self.scale, w * self.scale
        concat_h, concat_l = sv["concat_h"], sv["concat_l"]
        S = self.S
        # The gradient of the concatenated HR features is never held as a whole (13 GB at micro-batch 4, read-modify-written by six
        # dgrads): its producers' small dPre maps wait in the slots of these two buffers and each stage's 128-channel slice is
        # gathered when the stage's backward starts (_gather_conv).
        # (self.gather = False, CSBSR_GATHER_DCH=0: autograd's order -- the whole gradient buffer, every producer accumulating into it
        # as it runs; kept for same-run A/B timing and as a cross-check of the gathered path in tests/test_kbpn_gather_gpu.py)
        gather = self.gather
        dimg = e.new(B, H, W, 8 * S, zero=True) if gather else None                # slot 0: output_conv, slot S - s + 1: sr_reconst(s)
        dxdc = e.new(B, H, W, 128 * (S - 1), zero=False) if (gather and S > 1) else None      # slot S - 1 - s: down.conv(s)
        dch = None if gather else e.new(B, H, W, 128 * S, zero=False)
        img_slot = lambda t: dimg.slice(8 * t, 8 * t + 8, creal=3)
        dcl = e.new(B, h, w, 128 * (self.S - 1), zero=True) if self.S > 1 else None
        # output conv
        # (a slot's pixels are 16 S bytes apart: the weight gradients and the single-producer launches read a compact copy instead --
        # through the slots the thin wgrads ran 40 % slower)
        dpre_out = e.nchw32_to_fm(dsr32)
        if gather and S > 1:
            e.nchw32_to_fm(dsr32, out=img_slot(0))
        self._wg(self.output_conv, dpre_out, concat_h)
        if not gather:
            self.output_conv.bwd_input(dpre_out, out=dch, accumulate=False)
        dvec_next = dkvec_final.clone()     # gradient wrt the normalised kernel vector leaving stage s
        dlowp = None
        for s in range(self.S, 0, -1):
            st, q = self.stages[s - 1], sv["stages"][s - 1]
            if s < self.S and self.sft:
                # ---- SFT backward
                fpre = concat_l.slice(0, 128 * s)
                dfpre = dcl.slice(0, 128 * s)
                dsc = e.new(B, h, w, 128 * s)
                e.epilogue_bwd(dlowp, out=q["lowp"], res=fpre, res2=q["sc"], res_mode=L.RES_FMA, dpre=dlowp, dres=dfpre, dres_acc=True,
                               dres2=dsc, dbias=None if st.sh1.frozen else grad_acc(st.sh1.b), creal=st.sh1.cout)
                for c1, c0, t, dz, fold in ((st.sh1, st.sh0, q["t2"], dlowp, q["fold2"]), (st.sc1, st.sc0, q["t1"], None, q["fold1"])):
                    if dz is None:
                        dz = self._act_bwd(c1, dsc, q["sc"])
                    self._wg(c1, dz, t)
                    # conv0's LeakyReLU derivative rides on conv1's dgrad (mask = conv0's saved output) and its bias gradient comes out of
                    # the border-class sums the folded weight gradient takes anyway: no epilogue-backward pass over the C-channel map
                    dt = c1.bwd_input(dz, mask=(t, c0.slope))
                    dvec_next = dvec_next + c0.bwd_weights_folded(dt, fpre, fold, self.Mtap, frozen=c0.frozen, bias_grad=True)
                    c0.bwd_input(dt, seg=0, out=dfpre, accumulate=True)
                    del dt
                del dsc, dlowp
            derr = None
            if s < self.S:
                # ---- DownBlock backward
                dlow_s = dcl.slice(128 * (s - 1), 128 * s)
                lows = concat_l.slice(128 * (s - 1), 128 * s)
                if self.lr_err:      # low = down(...) + kb.conv(error): the error branch takes dlow_s before the in-place pass below turns it into dPre
                    self._wg(st.kb_conv, dlow_s, q["err16"])
                    derr = e.f32(B, 3, h, w, zero=False)
                    st.kb_conv.bwd_input(dlow_s, out32=derr, in_hw=(h, w))
                    lows = q["lowd"]
                dl0 = e.new(B, h, w, 128)
                self._act_bwd(st.down3, dlow_s, lows, res=q["l0"], res_mode=L.RES_ADD, dres=dl0)
                self._wg(st.down3, dlow_s, q["dd"])
                # (down_conv2's epilogue-backward pass -- PReLU derivative from dd + xd, bias / slope sums, d(xd) = -dOut -- rides on
                # down_conv3's dgrad where the phase-decomposed kernel takes the launch)
                dxd = dxdc.slice(128 * (S - 1 - s), 128 * (S - s)) if gather else e.new(B, H, W, 128)
                ddd = st.down3.bwd_input(dlow_s, in_hw=(H, W), dact=(st.down2, q["dd"]), dres=(q["xd"], dxd, L.RES_SUB))
                if not st.down3.last_fused:
                    self._act_bwd(st.down2, ddd, q["dd"], res=q["xd"], res_mode=L.RES_SUB, dres=dxd)
                self._wg(st.down2, ddd, q["l0"])
                st.down2.bwd_input(ddd, out=dl0, accumulate=True, in_hw=(h, w))
                del ddd
                self._act_bwd(st.down1, dl0, q["l0"])
                self._wg(st.down1, dl0, q["xd"])
                st.down1.bwd_input(dl0, out=dxd, accumulate=True, in_hw=(H, W), dact=(st.down_conv, q["xd"]))
                del dl0
                if not st.down1.last_fused:
                    self._act_bwd(st.down_conv, dxd, q["xd"])
                chp = concat_h.slice(0, 128 * s)
                self._wg(st.down_conv, dxd, chp)
                if not gather:
                    st.down_conv.bwd_input(dxd, out=dch.slice(0, 128 * s), accumulate=True)
                del dxd
            # ---- KBlock backward
            hs = concat_h.slice(128 * (s - 1), 128 * s)
            # this stage's slice of the concatenated gradient: every 3-channel producer so far in one write-only launch, then every
            # down.conv producer in one accumulating launch
            if gather:
                dhs = e.new(B, H, W, 128)
                nimg = S - s + 1
                self._gather_conv("img", s).bwd_input(dpre_out if nimg == 1 else dimg, out=dhs, accumulate=False)
                if s < S:
                    self._gather_conv("down", s).bwd_input(dxdc.slice(0, 128 * (S - s)), out=dhs, accumulate=True)
            else:
                dhs = dch.slice(128 * (s - 1), 128 * s)
            # out = act(pre) + h: the gradient wrt h IS dOut, so dPre goes to a fresh buffer and dOut's own storage (dead after this
            # block) carries on as dh -- one HR write stream less than copying it out (same below for up3 / h0)
            dh = dhs
            if not self.lr_err:
                dpk = e.new(B, H, W, 128)
                if st.kb_up.thin_tp_fused_ok(q["err16"]):
                    # one pass over dOut: the pre-activation is rebuilt from the 3-channel error image (csrc/conv_kbup.hip) instead of
                    # read back as hs - h; dPre, the weight gradient and the slope gradient leave together
                    st.kb_up.bwd_thin_tp_fused(dhs, q["err16"], dpk, frozen=st.kb_up.frozen)
                else:
                    self._act_bwd(st.kb_up, dhs, hs, res=q["h"], res_mode=L.RES_ADD, dpre=dpk)
                    self._wg(st.kb_up, dpk, q["err16"])
                derr = e.f32(B, 3, h, w, zero=False)
                st.kb_up.bwd_input(dpk, out32=derr, in_hw=(h, w))
                del dpk
            vec = q["vec"].contiguous()
            dvec = e.f32(B, self.kk)
            if derr is not None:
                dsr_t = e.f32(B, 3, H, W, zero=False)
                L.call("csbsr_blur_bwd_input", _ptr(derr), _ptr(vec), _ptr(dsr_t), 0, B, 3, H, W, self.K, self.scale, e.stream)
                L.call("csbsr_blur_bwd_kernel", _ptr(derr), _ptr(q["sr_t32"]), _ptr(dvec), B, 3, H, W, self.K, self.scale, e.stream)
            else:                    # LR-error variant, last stage: its error map feeds nothing (kbpn.py:179-181); only the predictor reads sr_t
                dsr_t = e.f32(B, 3, H, W, zero=True)
            dvec = dvec + dvec_next
            # vec = kvec2 / sum(kvec2)
            dk2 = (dvec - (dvec * q["vec"]).sum(1, keepdim=True)) / q["ksum"]
            if self.use_predictor:
                dkin = self._kernel_predictor_bwd(st, q, dk2, dsr_t, H, W)
            else:
                dkin = dk2
            dvec_next = dkin
            # sr_t = sr_reconst(cat(concat_h[:128(s-1)], h))
            dpre = e.nchw32_to_fm(dsr_t)
            if s > 1 and gather:
                e.nchw32_to_fm(dsr_t, out=img_slot(S - s + 1))
            if s > 1 and not gather:
                st.sr_reconst.bwd_input(dpre, seg=0, out=dch.slice(0, 128 * (s - 1)), accumulate=True)
            segs = (concat_h.slice(0, 128 * (s - 1)), q["h"]) if s > 1 else (q["h"],)
            self._wg(st.sr_reconst, dpre, segs)
            # (its gradient wrt the earlier stages' slices waits in the slot; the part wrt this stage's own h is added now)
            # ... and, where the thin-input kernel takes the launch, up_conv3's whole epilogue-backward pass rides on it (h = prelu(pre) + h0:
            # the launch that completes dh writes dPre = dh x prelu'(h - h0) in its place, the unmasked total to a second buffer as dh0
            # and the slope-gradient partials): one pass over the HR map instead of two (csrc/conv_thin.hip, DACT)
            spare = e.new(B, H, W, 128)
            st.sr_reconst.bwd_input(dpre, seg=1 if s > 1 else 0, out=dh, accumulate=True, dact=(st.up3, q["h"]),
                                    dres=(q["h0"], spare, L.RES_ADD))
            del dpre, dsr_t, derr
            # ---- UpBlock backward
            if st.sr_reconst.last_fused:
                dpu, dh0 = dh, spare
            else:
                dpu = spare
                self._act_bwd(st.up3, dh, q["h"], res=q["h0"], res_mode=L.RES_ADD, dpre=dpu)
                dh0 = dh
            del spare
            self._wg(st.up3, dpu, q["d"])
            dd_ = st.up3.bwd_input(dpu, in_hw=(h, w))
            del dh, dpu
            dxu = e.new(B, h, w, 128)
            self._act_bwd(st.up2, dd_, q["d"], res=q["xu"], res_mode=L.RES_SUB, dres=dxu)
            self._wg(st.up2, dd_, q["h0"])
            # (the dgrad that completes dh0 also applies up1's PReLU derivative and sums its bias / slope gradients where the
            # phase-decomposed kernel takes the launch: no epilogue-backward pass over the HR map)
            st.up2.bwd_input(dd_, out=dh0, accumulate=True, in_hw=(H, W), dact=(st.up1, q["h0"]))
            del dd_
            if not st.up2.last_fused:
                self._act_bwd(st.up1, dh0, q["h0"])
            self._wg(st.up1, dh0, q["xu"])
            st.up1.bwd_input(dh0, out=dxu, accumulate=True, in_hw=(h, w))
            del dh0
            self._act_bwd(st.up_conv, dxu, q["xu"])
            self._wg(st.up_conv, dxu, q["low_in"])
            if self.sft or s == 1:
                dlowp = st.up_conv.bwd_input(dxu)        # gradient wrt this stage's `low` input
            else:                    # no SFT layer: `low` IS the first 128 (s - 1) channels of the concatenated LR features
                st.up_conv.bwd_input(dxu, out=dcl.slice(0, 128 * (s - 1)), accumulate=True)
                dlowp = None
            del dxu
            sv["stages"][s - 1] = None
            if stage_done is not None:
                e.join_wgrad()           # the stage's weight gradients are complete before its bucket is exchanged
                stage_done(s)
        # ---- initial kernel predictor + VGG head
        dinit = dlowp
        feats = sv["feats"]
        if self.use_predictor:
            zs, k441, ksum = sv["pred"]
            dk = dvec_next
            kv = k441 / ksum
            dk441 = (dk - (dk * kv).sum(1, keepdim=True)) / ksum
            d49 = (dk441 @ self.U) / float(h * w)
            z3 = zs[3]
            dz = e.new(B, h, w, self.kc)
            for b in range(B):      # GAP backward: every pixel of sample b receives d49[b]
                gb = self._bcast_grad(d49[b:b + 1], h, w)
                zb = FM(z3.t[b:b + 1], z3.c)
                db = FM(dz.t[b:b + 1], dz.c)
                c = self.pred[2]
                e.epilogue_bwd(FM(gb.t.expand(1, h, w, gb.cp).contiguous(), gb.c), out=zb, act=c.act, prelu=c.prelu, dpre=db,
                               dprelu=None if c.frozen else grad_acc(c.prelu), creal=c.cout)
            for i in (2, 1, 0):
                c = self.pred[i]
                if i < 2:
                    self._act_bwd(c, dz, zs[i + 1])
                self._wg(c, dz, zs[i])
                if i > 0:
                    dz = c.bwd_input(dz)
                else:
                    c.bwd_input(dz, out=dinit, accumulate=True)
        d = dinit
        for i in (3, 2, 1, 0):
            c = self.feat[i]
            self._act_bwd(c, d, feats[i + 1])
            self._wg(c, d, feats[i])
            if i > 0:
                d = c.bwd_input(d)
        self.saved = None
        e.join_wgrad()
        if stage_done is not None:
            stage_done(0)

    def _kernel_predictor_bwd(self, st, q, dk2, dsr_t, H, W):
        """backward of kvec2 = kvec_in + U @ GAP(fe_cat(...)); adds the fe_SR path into dsr_t (fp32 planar)."""
        e = self.eng
        a, kctx, c1, c2 = q["kp"]
        if len(a) == 1:              # lean save: rebuild the fe_SR chain from the SR estimate (same kernels, same order: bit-identical)
            a = list(a)
            for c in st.fe_sr:
                a.append(c.fwd(a[-1]))
        B = dk2.shape[0]
        d49 = torch.einsum("bk,bkc->bc", dk2, q["kp_map"]) / float(H * W)
        g = self._bcast_grad(d49, H, W)
        cat2, cat1, cat0 = st.fe_cat[2], st.fe_cat[1], st.fe_cat[0]
        # Every layer of this branch is a bias-free conv + ReLU / LeakyReLU with a single consumer, so each dgrad applies the activation
        # derivative of the layer below in its own epilogue (mask = that layer's saved output): what a dgrad writes IS the next
        # dPre, and the ten stand-alone HR epilogue-backward passes per stage (read dOut, read out, write dPre) are gone.
        msk = lambda conv, out: (out, 0.0 if conv.act == A_RELU else conv.slope)
        self._wg(cat2, g, c2)
        dc2 = self._fold_const_dgrad(cat2, d49, msk(cat1, c2), H, W)
        self._wg(cat1, dc2, c1)
        dc1 = cat1.bwd_input(dc2, mask=msk(cat0, c1))
        del dc2
        if not cat0.frozen:
            cat0.bwd_weights(dc1, a[-1], split_override=(cat0.split[0], 0))
        da = cat0.bwd_input(dc1, seg=0, mask=msk(st.fe_sr[4], a[5]))
        dkin = self._kernel_branch_bwd(st, dc1, kctx, H, W)      # the fe_kernel branch: 25 class sums of dPre + a tiny autograd graph
        del dc1
        # fe_SR chain
        for i in (4, 3, 2, 1, 0):
            c = st.fe_sr[i]
            self._wg(c, da, a[i])
            if i > 0:
                da = c.bwd_input(da, mask=msk(st.fe_sr[i - 1], a[i]))
            else:
                c.bwd_input(da, out32=dsr_t, accumulate=True)
        return dk2 + dkin

    # ------------------------------------------------------------------ constant-operand folding (exact)
    def _kernel_branch_fwd(self, st, kvec, H, W):
        """The fe_kernel branch of KernelPredictorLikeIKC (kbpn.py:565-569: GAP'ed kernel code expanded over the image -> conv3x3 + LReLU
        -> conv3x3 + LReLU -> second half of fe_cat.0's input) never exists as a map.  A zero-padded 3x3 conv of a spatially constant
        map takes one value per BORDER class (16, by which image edges the pixel touches); a second one on top takes one value per
        TWO-RING class (25 = 5 row types x 5 column types: coordinate 0, 1, interior, size-2, size-1), and fe_cat.0 is 1x1, so the whole
        branch enters fe_cat.0 as a bias table [B, 25, 32] (conv desc cbias_mode 1) -- exact, instead of a 441 -> 49 and a 49 -> 49 3x3
        convolution and a 49-channel concat half at HR resolution.  The table is a few small fp32 contractions; they are recorded as a
        torch autograd graph (leaves: the kernel vector and the three weight tensors), which IS the backward of the branch."""
        e = self.eng
        assert H >= 5 and W >= 5, "two-ring classes need a map of at least 5 x 5"
        k0, k1, cat0 = st.fe_k[0], st.fe_k[1], st.fe_cat[0]
        with torch.enable_grad():
            kv = kvec.detach().to(torch.float32).requires_grad_(True)
            w0, w1, wc = (c.w.detach().requires_grad_(not c.frozen) for c in (k0, k1, cat0))
            cbv = kernel_branch_table(kv, w0, w1, wc[:, cat0.split[0]:, 0, 0], self.Mtap, self.Rtap, _act_fn(k0), _act_fn(k1))
        cb = e.f32(kvec.shape[0], 25, pad8(cat0.cout))
        cb[:, :, :cat0.cout] = cbv.detach().reshape(-1, 25, cat0.cout)
        return cb, (cbv, kv, (w0, w1, wc))

    def _kernel_branch_bwd(self, st, dc1, kctx, H, W):
        """adjoint of the above: 25 two-ring class sums of fe_cat.0's dPre, then the recorded graph; returns dL/d(kernel vector)."""
        e = self.eng
        cbv, kv, ws = kctx
        cat0 = st.fe_cat[0]
        B = dc1.N
        sums = e.f32(B, 25, dc1.cp)
        L.call("csbsr_ring_class_sums", _ptr(dc1.t), dc1.ld, _ptr(sums), B, H, W, dc1.cp, e.stream)
        leaves = [kv] + [w for w in ws if w.requires_grad]
        grads = torch.autograd.grad(cbv, leaves, sums[:, :, :cat0.cout].reshape(B, 5, 5, cat0.cout))
        gi = iter(grads[1:])
        for conv, w in zip((st.fe_k[0], st.fe_k[1], cat0), ws):
            if w.requires_grad:
                g = next(gi)
                if conv is cat0:      # only the folded half of fe_cat.0's input channels: the other half's wgrad kernel (possibly on the
                    c0 = cat0.split[0]          # wgrad stream) owns those elements of the accumulator
                    grad_acc(conv.w)[:, c0:].add_(g[:, c0:])
                else:
                    grad_acc(conv.w).add_(g)
        return grads[0]

    def _fold_const_dgrad(self, conv, gvec, mask, H, W):
        """dgrad of a zero-padded 3x3 conv whose dOut is spatially constant per sample (``gvec`` [B, cout]: the backward of the global
        average pool behind fe_cat.2, kbpn.py:573-578): dIn takes one value per border class -- T = g . W per tap, the 16 class
        sums with the taps flipped -- times the activation derivative of the layer below (``mask`` = (its saved output, slope)).
        Replaces a 2*H*W*cin*cout*9 FLOP convolution of a constant map by a masked fill."""
        e = self.eng
        B = gvec.shape[0]
        w16 = conv.w.to(torch.float16).float()                      # same operand rounding as the MFMA path
        g16 = gvec.to(torch.float16).float()
        T = torch.einsum("no,ocyx->ncyx", g16, w16)                  # [B, cin, 3, 3]
        mf = self.Mtap.flip(1)                                       # dIn(y) = sum_ky dOut(y - ky + 1) W(ky): tap 2 falls off the first row
        V = torch.einsum("ncyx,ay,bx->nabc", T, mf, mf)              # [B, 4, 4, cin]
        cp = pad8(conv.cin)
        Vp = e.f32(B, 16, cp)
        Vp[:, :, :conv.cin] = V.reshape(B, 16, conv.cin)
        out = e.new(B, H, W, conv.cin)
        mfm, mslope = mask
        assert mfm.cp == cp and (mfm.H, mfm.W) == (H, W) and not mfm.bcast
        L.call("csbsr_border_class_fill_masked", _ptr(Vp), _ptr(out.t), out.ld, _ptr(mfm.t), mfm.ld, float(mslope), B, H, W, cp, e.stream)
        return out
