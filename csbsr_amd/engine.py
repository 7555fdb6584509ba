"""Host-side runtime for the HIP hot path: feature-map views, layer objects that own packed weights
and launch the C-ABI kernels (include/csbsr_hip.h), and explicit backward passes.

Design (MI355X-first):
  * feature maps are fp16 NHWC with channels padded to 8 and an explicit pixel stride, so ``torch.cat``
    of the reference (kbpn.py:174,179,187; pspnet.py:40) becomes writing into channel slices of one buffer
    and two-segment convolutions -- no concat copies;
  * spatially constant operands (blur-kernel code maps, kbpn.py:405,513,565) are never materialised: they
    enter the convolution as a stride-0 segment;
  * backward is explicit (no autograd tape): every dgrad kernel either overwrites or accumulates into the
    gradient buffer of its input, activation gradients carry a power-of-two loss scale (fp16 range), weight
    gradients accumulate in fp32.
PyTorch is used for device memory and the current HIP stream only.
"""
import ctypes as C
import math
import os

import torch

from . import _lib as L


def pad8(c):
    return (c + 7) // 8 * 8


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class FM:
    """fp16 channels-last feature map view: tensor [N,H,W,Cp] (last stride 1), ``c`` real channels.
    ``bcast``: tensor is [N,1,1,Cp] and stands for a spatially constant map of logical size (H, W).
    ``lo``: split-fp16 map of the detector precision mode -- ``t`` is the hi plane and the lo plane (value = hi + lo, ~22 mantissa
    bits) lives ``lo`` elements further on in the same allocation with the same strides; 0 = plain fp16.  Backward kernels read
    ``t`` (the hi plane) like any other map."""
    __slots__ = ("t", "c", "bcast", "H", "W", "lo")

    def __init__(self, t, c, bcast=False, H=None, W=None, lo=0):
        assert t.dtype == torch.float16 and t.dim() == 4 and t.stride(3) == 1 and t.shape[3] % 8 == 0
        self.t, self.c, self.bcast, self.lo = t, c, bcast, lo
        self.H = t.shape[1] if H is None else H
        self.W = t.shape[2] if W is None else W

    @property
    def N(self):
        return self.t.shape[0]

    @property
    def cp(self):
        return self.t.shape[3]

    @property
    def ld(self):
        return self.t.stride(2)

    def strides(self):
        if self.bcast:
            return self.t.stride(0), 0, 0
        return self.t.stride(0), self.t.stride(1), self.t.stride(2)

    def seg(self):
        sn, sy, sx = self.strides()
        return L.Seg(_ptr(self.t), sn, sy, sx, self.cp, self.c)

    def split_segs(self):
        """the two conv input segments of a split map: [hi | lo] as one 2*cp-channel segment (the planes must be adjacent, i.e.
        an unsliced map) and the hi plane again (weights from csbsr_pack_weights_split)."""
        assert self.lo == self.cp and not self.bcast, "split conv inputs must be whole [hi | lo] buffers"
        sn, sy, sx = self.strides()
        return L.Seg(_ptr(self.t), sn, sy, sx, 2 * self.cp, 0), L.Seg(_ptr(self.t), sn, sy, sx, self.cp, self.c)

    def slice(self, c0, c1, creal=None):
        return FM(self.t[..., c0:c1], (c1 - c0) if creal is None else creal, self.bcast, self.H, self.W, self.lo)

    @property
    def npix(self):
        return self.N * self.H * self.W

    def flat_ok(self):
        """True when pixels are laid out with one constant stride (needed by the elementwise kernels)."""
        t = self.t
        return (not self.bcast) and t.stride(1) == t.shape[2] * t.stride(2) and t.stride(0) == t.shape[1] * t.stride(1)


# Partial-row scratch of the order-fixed two-stage reductions (csbsr_set_reduction_scratch; the library has no floating-point atomics,
# every sum is a fixed tree over partial rows, so a step is bit-reproducible): ONE buffer per device for the life of the process,
# shared by every Engine on that device and registered with the library under that device's index -- so a second model (an
# evaluator next to the trainer, a replica on another GPU) neither steals the registration nor leaves a dangling pointer behind when
# it is garbage-collected.  Launches on one device are stream-ordered per Engine; two Engines on one device must not run reducing
# kernels concurrently on different streams (they do not: one Python thread drives a replica).
_RED_SCRATCH = {}


def _reduction_scratch(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    buf = _RED_SCRATCH.get(idx)
    if buf is None:
        # poisoned with NaN once: a fold that reads a slot its producer did not write shows up immediately instead of adding garbage
        # 256 MB: the largest user is the BatchNorm-statistics rows of a 64-channel conv at HR 1792^2, B = 8 (one [sum | sumsq] row per
        # 128-pixel tile: 26 M floats); the last 4 M floats are the second level of csbsr_sum_partials (csrc/common.h)
        buf = torch.full((64 << 20,), float("nan"), dtype=torch.float32, device=torch.device("cuda", idx))
        _RED_SCRATCH[idx] = buf
        with torch.cuda.device(idx):
            L.call("csbsr_set_reduction_scratch", _ptr(buf), buf.numel())
    return buf


class Engine:
    def __init__(self, device="cuda:0", grad_scale=1.0):
        L.load()
        self.device = torch.device(device)
        self.grad_scale = float(grad_scale)
        self._ws = None
        self._red = _reduction_scratch(self.device) if torch.cuda.is_available() else None
        self.training = True
        self.use_hr = os.environ.get("CSBSR_CONV_HR", "1") != "0"       # A/B hook: 0 routes the HR small-channel layers through the implicit-GEMM kernels
        # Winograd F(2,3)-along-x kernel for the wide 3x3 layers (csrc/conv_x3w.hip): built, parity-tested, 3-9 % faster per launch than the
        # direct kernels (−9 ms per config-2 step on the SFT convs) for a composed map 13 % further from the reference --
        # opt-in: CSBSR_CONV_X3W=1
        self.use_x3w = {"0": 0, "all": 2}.get(os.environ.get("CSBSR_CONV_X3W", "0"), 1)      # 0 off, 1 the SFT convs (Conv.winograd), "all": every eligible 3x3 layer
        self.use_x3n = os.environ.get("CSBSR_CONV_X3N", "1") != "0"     # A/B hook: 0 keeps the many-channels -> <= 64-cout 3x3 layers on the LDS-DMA tiles (csrc/conv_x3n.hip)
        self.use_head1 = os.environ.get("CSBSR_HEAD1", "1") != "0"       # A/B hook: 0 runs the 1-channel heads on the general conv kernels
        self.use_x3 = os.environ.get("CSBSR_CONV_X3", "1") != "0"       # A/B hook: 0 routes the wide 3x3 layers through the implicit-GEMM kernels
        self.use_tp = os.environ.get("CSBSR_CONV_TP", "1") != "0"       # A/B hook: 0 routes the 2x2-tap transposed layers through the implicit-GEMM kernels
        self.split_fused = os.environ.get("CSBSR_SPLIT_FUSED", "1") != "0"   # A/B hook: 0 = the three-block split forward (x_hi staged twice)
        self.dc_comp = os.environ.get("CSBSR_DC_COMP", "1") != "0"           # A/B hook: 0 = no compensation of the forward weights' fp16 rounding (Conv._dc_bias)
        self.tapsum = os.environ.get("CSBSR_TAPSUM", "1") != "0"             # A/B hook: 0 = KBPN weights rounded to nearest instead of tap-sum-preserving (Conv._wq)
        # Weight gradients on a second HIP stream (opt-in, CSBSR_WGRAD_STREAM=1).  In the backward a layer's wgrad is a side branch (it
        # only feeds the parameter's gradient accumulator) while the dgrad chain is the critical path; the wgrads are MFMA-bound on
        # L2-resident tiles, much of what the chain runs between two of its convolutions is HBM-bound, so two streams could let the
        # dispatcher fill CUs an HBM-bound kernel leaves idle.  Same kernels, same per-parameter accumulation order (all wgrads stay
        # in program order on the side stream): bit-identical results (the GPU suite passes with it on).  Measured, round 4, same-run A/B: config 4
        # (HRNet-OCR, hundreds of small launches) 5.57 vs 5.57 img/s; config 2 at B = 4 6.92 -> 6.88 (the big kernels each fill the
        # chip: nothing to overlap); config 2 at B = 8 collapses to 2.4 img/s -- at 239 of 288 GB the operands the lagging side
        # stream still holds (record_stream) leave the caching allocator without free blocks and it falls back to synchronising
        # hipFree / hipMalloc cycles.  Hence off by default.
        self.wgrad_mirror = os.environ.get("CSBSR_WGRAD_MIRROR", "1") == "1"      # see Conv._bwd_weights_impl
        self.thin_tp_fused = os.environ.get("CSBSR_KBUP_FUSED", "1") == "1"       # see Conv.bwd_thin_tp_fused (A/B timing: 0)
        # up_conv3's epilogue-backward pass on kb.sr_reconst's dgrad (csrc/conv_thin.hip, DACT): 26 GB per step less fabric traffic, but the
        # fused launch is no faster than the two it replaces (1065 vs 1067 ms per step, same run) -- opt-in
        self.thin_dact = os.environ.get("CSBSR_THIN_DACT", "0") == "1"
        self.fold_prelu = os.environ.get("CSBSR_FOLD_PRELU", "1") == "1"        # pspnet.py _blur_skip_bwd (A/B timing: 0)
        self.wg_stream = None
        self._wg_on = os.environ.get("CSBSR_WGRAD_STREAM", "0") == "1"
        self._ws_by_stream = {}
        self._zero_blk, self._zero_off, self._zero_key, self._zero_by_stream = None, 0, None, {}
        self._probe_epoch = 0
        self.timing = None              # list of (kind, flops, bytes, start_event, end_event) when profiling is on

    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def new(self, N, H, W, c, zero=False, split=False):
        f = torch.zeros if zero else torch.empty
        cp = pad8(c)
        if split:        # [hi | lo] planes side by side: a conv reads them as one 2*cp-channel segment
            return FM(f((N, H, W, 2 * cp), dtype=torch.float16, device=self.device)[..., :cp], c, lo=cp)
        return FM(f((N, H, W, cp), dtype=torch.float16, device=self.device), c)

    ZERO_ARENA = 1 << 20          # floats per arena block (4 MB)

    def f32(self, *shape, zero=True):
        if not zero:
            return torch.empty(shape, dtype=torch.float32, device=self.device)
        # small zero-initialised buffers (BatchNorm statistic rows, reduction targets, class-bias tables: hundreds per pass, ~2000 per
        # HRNet-OCR step) are cut from a pre-zeroed block instead of costing a fill launch each; a block is never re-zeroed or reused --
        # when it is used up the next one is allocated (one fill launch per 2^20 floats handed out) and the old one dies with its views
        n = 1
        for d in shape:
            n *= int(d)
        if n == 0 or n > 65536:
            return torch.zeros(shape, dtype=torch.float32, device=self.device)
        n_al = (n + 63) // 64 * 64
        # one arena per STREAM (as workspace()): the block's zero fill runs on the stream that creates it, so a cut handed to a consumer on
        # another stream (the reducer's side stream, the opt-in wgrad stream, the CU-mask measurement streams) would have no ordering against it
        skey = torch.cuda.current_stream(self.device).cuda_stream
        if skey != self._zero_key:
            self._zero_by_stream[self._zero_key] = (self._zero_blk, self._zero_off)
            self._zero_blk, self._zero_off = self._zero_by_stream.get(skey, (None, 0))
            self._zero_key = skey
        if self._zero_blk is None or self._zero_off + n_al > self.ZERO_ARENA:
            self._zero_blk, self._zero_off = torch.zeros(self.ZERO_ARENA, dtype=torch.float32, device=self.device), 0
        # (a fresh tensor on the block's storage, NOT a view of it: views share one autograd version counter, and an in-place update of any
        # buffer cut from the block would invalidate every other one that autograd has saved -- the OCR region-vector chain does save one)
        t = torch.empty(0, dtype=torch.float32, device=self.device).set_(self._zero_blk.untyped_storage(), self._zero_off, tuple(int(d) for d in shape))
        self._zero_off += n_al
        return t

    def workspace(self, nfloat):
        """wgrad slab workspace of the CURRENT stream (allocated under it, so the caching allocator orders its reuse on that stream)"""
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws_by_stream.get(key)
        if ws is None or ws.numel() < nfloat:
            ws = torch.empty(int(nfloat * 1.25) + 1024, dtype=torch.float32, device=self.device)
            self._ws_by_stream[key] = ws
        return ws

    def wgrad_stream(self):
        """the side stream weight gradients run on (None: the caller's stream)"""
        if not self._wg_on:
            return None
        if self.wg_stream is None:
            self.wg_stream = torch.cuda.Stream(self.device)
        return self.wg_stream

    def join_wgrad(self):
        """the caller's stream waits for every weight gradient issued so far (before anything reads or exchanges the accumulators)"""
        if self.wg_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.wg_stream)

    def prelu_fold_ok(self, p):
        """May a learned PReLU slope's layer take the FOLDED backward (Conv.bwd_weights_folded(prelu_out=), the dgrad mask above it)?  That
        form takes the gate from the sign of the saved output and divides the slope-gradient sum by slope^2: right for a slope safely above
        zero, inf / the wrong gate at slope <= 0.  The slope is read WITHOUT stalling the host: at most once per optimiser step a call
        starts an asynchronous copy into pinned host memory; a later call takes the value once the copy's event has completed (``query``,
        never ``synchronize``) and until then decides from the last completed value (one or two optimiser steps old: the threshold, 1e-3
        -- the reference's SFTLikeBlock starts these slopes at 0.01 --, is ten Adam steps of lr 1e-4 above zero); only the very first
        call of a parameter reads synchronously.  ``new_step`` drops the probes of reloaded parameters."""
        st = getattr(p, "_slope_probe", None)
        if st is None:
            host = torch.empty(p.numel(), dtype=torch.float32, pin_memory=True)
            host.copy_(p.detach().reshape(-1).to(torch.float32))
            ok = bool(float(host.min()) > 1e-3)
            p._slope_probe = [host, None, ok, self._probe_epoch]
            return ok
        host, ev, ok, epoch = st
        if ev is not None and ev.query():          # the copy started by an earlier call has landed: take its value (never wait for it)
            ok, ev = bool(float(host.min()) > 1e-3), None
        if ev is None and epoch != self._probe_epoch:      # at most one probe per optimiser step (Engine.new_step)
            host.copy_(p.detach().reshape(-1).to(torch.float32), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            epoch = self._probe_epoch
        p._slope_probe = [host, ev, ok, epoch]
        return ok

    def new_step(self, params=()):
        """called when the master weights may have changed (optimiser step, load_state_dict): slope probes of the previous weights are
        dropped for re-assigned / reloaded parameters (``params``) and every live probe may issue one new asynchronous read"""
        self._probe_epoch += 1
        for p in params:
            if hasattr(p, "_slope_probe"):
                del p._slope_probe

    # ------------------------------------------------------------------ elementwise wrappers
    def epilogue_bwd(self, dout, out=None, act=L.ACT_NONE, slope=0.0, prelu=None, res=None, res2=None, res_mode=L.RES_NONE,
                     dpre=None, dres=None, dres_acc=False, dres2=None, dres2_acc=False, dbias=None, dprelu=None, creal=None):
        assert dout.flat_ok()
        for f in (out, res, res2, dpre, dres, dres2):
            assert f is None or f.flat_ok()
        d = L.EpiBwdDesc()
        d.npix, d.c, d.creal = dout.npix, dout.cp, dout.c if creal is None else creal
        d.dout, d.dout_ld = _ptr(dout.t), dout.ld
        if out is not None:
            d.out, d.out_ld = _ptr(out.t), out.ld
        if res is not None:
            d.res, d.res_ld = _ptr(res.t), res.ld
        if res2 is not None:
            d.res2, d.res2_ld = _ptr(res2.t), res2.ld
        d.act, d.act_slope, d.prelu, d.res_mode = act, slope, _ptr(prelu), res_mode
        if dpre is not None:
            d.dpre, d.dpre_ld = _ptr(dpre.t), dpre.ld
        if dres is not None:
            d.dres, d.dres_ld, d.dres_accumulate = _ptr(dres.t), dres.ld, int(dres_acc)
        if dres2 is not None:
            d.dres2, d.dres2_ld, d.dres2_accumulate = _ptr(dres2.t), dres2.ld, int(dres2_acc)
        d.dbias, d.dprelu = _ptr(dbias), _ptr(dprelu)
        L.call("csbsr_epilogue_backward", C.byref(d), self.stream)

    def nchw32_to_fm(self, src, cp=None, mean=None, invstd=None, out=None, split=False):
        N, Cc, H, W = src.shape
        assert src.is_contiguous() and src.dtype == torch.float32
        if out is None:
            out = self.new(N, H, W, Cc if cp is None else cp, split=split)
            out.c = Cc
        L.call("csbsr_nchw32_to_nhwc16_split", _ptr(src), _ptr(out.t), N, Cc, H, W, out.cp, out.ld, out.lo, _ptr(mean), _ptr(invstd),
               self.stream)
        return out

    def fm_to_nchw32(self, fm, dst, C_, alpha=1.0, beta=0.0):
        assert fm.flat_ok() and dst.is_contiguous()
        L.call("csbsr_nhwc16_to_nchw32", _ptr(fm.t), fm.ld, _ptr(dst), fm.N, C_, fm.H, fm.W, alpha, beta, self.stream)

    def bilinear(self, x, OH, OW, align, out=None, drop=None):
        if out is None:
            out = self.new(x.N, OH, OW, x.c, split=bool(x.lo))
        assert x.flat_ok() and out.flat_ok()
        L.call("csbsr_bilinear_fwd_split", _ptr(x.t), x.ld, x.lo, _ptr(out.t), out.ld, out.lo, x.N, x.H, x.W, x.cp, OH, OW, int(align),
               _ptr(drop), self.stream)
        return out

    def bilinear_bwd(self, dy, dx, acc, align, drop=None):
        L.call("csbsr_bilinear_bwd", _ptr(dy.t), dy.ld, _ptr(dx.t), dx.ld, int(acc), dx.N, dx.H, dx.W, dx.cp, dy.H, dy.W, int(align),
               _ptr(drop), self.stream)


# ---------------------------------------------------------------------------------------------- conv layer

class Conv:
    """One Conv2d / ConvTranspose2d of the reference bound to its fp32 master parameters.

    ``w``: OIHW (conv) or IOHW (transposed) fp32 parameter; gradients accumulate (scaled by the engine's
    grad_scale) into ``w.gacc`` fp32 tensors created on first use and read out by the model.
    ``split``: real channel counts of the (up to two) input segments.
    """

    def __init__(self, eng, name, params, k, stride=1, pad=0, dil=1, transposed=False, bias=True, act=L.ACT_NONE, slope=0.0,
                 prelu=False, split=None):
        self.eng, self.name = eng, name
        self.w = params[name + ".weight"]
        self.b = params.get(name + ".bias") if bias else None
        self.k, self.stride, self.pad, self.dil, self.transposed = k, stride, pad, dil, transposed
        if transposed:
            self.cin, self.cout = self.w.shape[0], self.w.shape[1]
        else:
            self.cout, self.cin = self.w.shape[0], self.w.shape[1]
        self.split = (self.cin, 0) if split is None else tuple(split)
        assert sum(self.split) == self.cin
        self.act, self.slope = act, slope
        self.prelu = params[prelu] if prelu else None
        self._packed = {}
        self.hp_dgrad = False        # detector precision mode: dgrad against [w_hi | w_lo] (two MFMA passes), see bwd_input
        # detector precision mode, per layer (the model's precision plan): K blocks a forward conv of a split (hi + lo) input runs --
        # 3: x_hi w_hi + x_lo w_hi + x_hi w_lo;  2: [x_hi | x_lo] w_hi (the weight's rounding error stays);  1: x_hi w_hi, plain fp16
        # operands.  The output is stored as a hi + lo pair in every case, so the layers around it need not know.
        self.fwd_blocks = 3
        # KBPN (plain fp16 operands): first-order compensation of the forward weights' fp16 rounding, see _dc_bias
        self.dc_comp = False
        self.winograd = False        # True: the layer may run as Winograd F(2,3) along x (KBPN's SFT convs: the set tests/study_winograd.py validates)

    # -- packed operand cache (invalidated by the model at every optimiser step)
    def invalidate(self):
        self._packed.clear()

    def _pack(self, key, kind, seg0, seg1, row_off, nrows, stride, pad, k_off=0):
        if key in self._packed:
            return self._packed[key]
        D0, D1 = self.w.shape[0], self.w.shape[1]
        n = L.load().csbsr_packed_weight_elems(kind, D0, D1, self.k, self.k, stride, seg0, seg1, nrows)
        dst = torch.empty(n, dtype=torch.float16, device=self.eng.device)
        L.call("csbsr_pack_weights", _ptr(self._wq()), _ptr(dst), kind, D0, D1, self.k, self.k, stride, pad, seg0, seg1, row_off, nrows,
               k_off, self.eng.stream)
        self._packed[key] = dst
        return dst

    def _wq(self):
        """the fp32 tensor the plain-fp16 operand packs read: the master weights, or -- on the layers that carry the rounding compensation
        (``dc_comp``: KBPN but its per-stage kernel predictors) -- their TAP-SUM-PRESERVING fp16 rounding (csbsr_round_weights, held as fp32
        so every pack kernel converts it exactly; once per optimiser step).  Nearest rounding leaves each filter a residual whose tap sum
        is a random walk of ~0.3 sqrt(taps) ulp: a spatially coherent per-(o, c) gain error that a low-pass detector passes in full.
        Moving the one or two taps nearest to a rounding midpoint to their other neighbour brings the tap sum under half an ulp (3x3:
        rms 3.6e-5 -> 6e-6 of kaiming-scaled weights, per-tap error +9 %) -- per output phase for transposed layers, which the bias form
        of the compensation (_dc_bias) cannot reach.  tests/study_kbpn_precision.py, the build's W + X storage plan on the contractive PSPNet
        fixture: segmentation map 1.61e-3 (nearest) / 1.38e-3 (nearest + bias, round 4) / 1.21e-3 (tap-sum) / 1.15e-3 (tap-sum + bias, this);
        relative L2 7.8 / 5.6 / 5.4 / 4.5e-4.  Forward, dgrad and wgrad operands all come from the same rounded tensor."""
        if not (self.dc_comp and self.eng.tapsum):
            return self.w
        q = self._packed.get("wq")
        if q is None:
            q = torch.empty_like(self.w)
            S = torch.empty(self.w.shape[0], self.w.shape[1], dtype=torch.float32, device=self.eng.device)
            mode = self.stride if (self.transposed and self.stride > 1) else 1
            L.call("csbsr_round_weights", _ptr(self.w), _ptr(q), _ptr(S), self.w.shape[0], self.w.shape[1], self.k, self.k, mode, self.eng.stream)
            self._packed["wq"], self._packed["dc_table"] = q, S
        return q

    # split-fp16 forward operand (detector precision mode): [w_hi | w_hi | w_lo] per tap, weights pre-scaled by WSCALE (undone by
    # the epilogue's out_scale) so the lo halves of kaiming-sized weights stay in fp16's normal range
    WSCALE = 256.0

    def _pack_split(self, key, kind, creal, nrows, stride, pad, k_off=0, layout=0, row_off=0, tapsum=False):
        if key in self._packed:
            return self._packed[key]
        D0, D1 = self.w.shape[0], self.w.shape[1]
        n = L.load().csbsr_packed_weight_elems_split(kind, D0, D1, self.k, self.k, stride, creal, nrows, layout)
        dst = torch.empty(n, dtype=torch.float16, device=self.eng.device)
        # (layout 2 = [w_hi | w_hi], a layer whose plan drops the x_hi w_lo product: its one rounding of the weights is the tap-sum-preserving one)
        src = self._wq() if (layout == 2 or tapsum) else self.w
        L.call("csbsr_pack_weights_split", _ptr(src), _ptr(dst), kind, D0, D1, self.k, self.k, stride, pad, creal, row_off, nrows, k_off,
               self.WSCALE, layout, self.eng.stream)
        self._packed[key] = dst
        return dst

    def _split_operand(self, x, key, kind, creal, nrows, stride, pad, k_off=0):
        """(input FMs, packed weights, out_scale, K blocks) of a forward conv whose input is a split map, per this layer's plan."""
        nb = self.fwd_blocks
        if nb == 1:      # the hi plane alone against plain fp16 weights
            return (FM(x.t, x.c, H=x.H, W=x.W),), self._pack((key, 1), kind, creal, 0, 0, nrows, stride, pad, k_off), 1.0, 1
        if nb == 3 and self.eng.split_fused and x.cp >= 32 and pad8(nrows) > 32 and kind == 0:
            # fused form (csbsr_conv_desc_t::split_fused): one staged K slice = 32 channels of [x_hi | x_lo] against [w_hi | w_lo], all three
            # products from it -- the LDS-DMA kernels' launch time follows the staged bytes, 2/3 of the three-block form's
            # (should the library refuse the fused form for this launch -- csbsr_conv_split_fused_eligible: the LDS-DMA kernels switched off by a
            # debug mode, a tile they do not take -- _launch falls back to the three-block operand)
            self._fs_fallback = lambda: self._pack_split((key, 3), kind, creal, nrows, stride, pad, k_off=k_off, layout=0)
            return (x,), self._pack_split((key, "fs"), kind, creal, nrows, stride, pad, k_off=k_off, layout=3), 1.0 / self.WSCALE, 4
        if nb == 2 and self.eng.split_fused and x.cp >= 32 and pad8(nrows) > 32 and kind == 0:
            # the two-product plan in the fused stage (csbsr_conv_desc_t::split_fused = 2, round 6): [x_hi | x_lo] staged once against
            # [w_hi | -] -- the two-block form below stages w_hi twice and measured no faster than the fused THREE products.  The hi halves
            # are the tap-sum-preserving rounding (a layer that keeps its weights' rounding), so the packed lo halves are zero and unread
            self._fs_fallback = lambda: self._pack_split((key, 2), kind, creal, nrows, stride, pad, k_off=k_off, layout=2)
            return (x,), self._pack_split((key, "fs2"), kind, creal, nrows, stride, pad, k_off=k_off, layout=3, tapsum=True), 1.0 / self.WSCALE, 5
        wt = self._pack_split((key, nb), kind, creal, nrows, stride, pad, k_off=k_off, layout=0 if nb == 3 else 2)
        return (x,), wt, 1.0 / self.WSCALE, nb

    def _dc_bias(self, xs):
        """bias + sum_c mean_c(x) * sum_taps (w - fp16(w))[o, c]: what the rounding of the weights to fp16 takes away from the layer's
        response to the MEAN of its input, given back through the bias.  The rounding residual of a filter is a fixed, spatially coherent
        perturbation: its response to the (large, positive after ReLU / PReLU) channel means is a per-output-channel offset of relative
        size 2^-12 sqrt(C) that no later layer averages out, and a low-pass detector passes it in full -- on the contractive reference
        fixtures it is the larger half of the segmentation map's deviation.  Measured on the composed HIP step
        (tests/test_wc2_composed_gpu.py, split mode, without / with): segmentation map 2.18e-3 -> 1.66e-3 of its maximum (relative L2
        8.1e-4 -> 6.1e-4) with PSPNet, 3.08e-3 -> 1.98e-3 (2.26e-3 -> 1.54e-3) with HRNet-OCR, 1.44e-3 -> 1.19e-3 (3.6e-4 -> 3.1e-4) with
        BlurSkip; SR image 6.95e-4 -> 6.2e-4; kernel vector at HR 64 6.6e-4 -> 4.5e-4; 0.35 % of a step (one mean kernel per input
        segment and one small contraction per layer and forward, one tap-sum kernel per layer and optimiser step).  The means are PER
        SAMPLE (a per-sample bias row, csbsr_conv_desc_t::bias_sn: KBPN keeps no batch-coupled operation, micro-batching and
        data-parallel sharding stay exact -- a batch mean measured 1.24-1.42e-3 on the PSPNet fixture but broke both), from every 8th
        row and column (csbsr_channel_mean_sub: 1/64 of a pass over the input, order-fixed).  Stride-1 / strided convolutions only: a
        transposed layer's residual response differs per output phase, and its average over the phases -- all a per-channel bias can
        carry -- measured no gain (the oracle simulation agrees: nothing for the average, 7 % for the exact per-phase form).  A constant
        of the backward (its gradient would be 2^-12 of the layer's)."""
        assert not self.transposed and 1 <= len(xs) <= 2
        S = self._packed.get("dc_table")
        if S is None:
            if self.eng.tapsum:
                self._wq()                      # tap sums of what the tap-sum-preserving rounding left
                S = self._packed["dc_table"]
            else:                               # (A/B: round to nearest)
                S = torch.empty(self.w.shape[0], self.w.shape[1], dtype=torch.float32, device=self.eng.device)
                scratch = torch.empty_like(self.w)
                L.call("csbsr_round_weights", _ptr(self.w), _ptr(scratch), _ptr(S), self.w.shape[0], self.w.shape[1], self.k, self.k, 0, self.eng.stream)
                self._packed["dc_table"] = S
        ms = []
        for f in xs:
            if f.bcast:
                ms.append(f.t[:, 0, 0, :].to(torch.float32).contiguous())
                continue
            sn, sy, sx = f.strides()
            m = torch.empty(f.N, f.cp, dtype=torch.float32, device=self.eng.device)
            step = 8 if f.H * f.W >= (1 << 18) else (4 if f.H * f.W >= (1 << 14) else 1)
            L.call("csbsr_channel_mean_sub", _ptr(f.t), sn, sy, sx, f.N, f.H, f.W, f.cp, step, _ptr(m), self.eng.stream)
            ms.append(m)
        N = xs[0].N
        out = torch.empty(N, self.cout, dtype=torch.float32, device=self.eng.device)
        m1 = ms[1] if len(ms) > 1 else None
        L.call("csbsr_dc_bias", _ptr(S), self.cout, self.w.shape[1], _ptr(ms[0]), ms[0].shape[1], xs[0].c, _ptr(m1), 0 if m1 is None else m1.shape[1],
               0 if m1 is None else xs[1].c, _ptr(self.b), N, _ptr(out), self.eng.stream)
        return out                                        # [N, cout] per-sample bias rows, csbsr_conv_desc_t::bias_sn = cout

    def _head1_ok(self, xs, out, out32, res, stat):
        f = xs[0]
        return (self.eng.use_head1 and self.cout == 1 and self.k == 1 and self.stride == 1 and not self.transposed and len(xs) == 1
                and out is None and out32 is not None and res is None and stat is None and self.prelu is None
                and self.act in (L.ACT_NONE, L.ACT_SIGMOID) and f.cp in (64, 128, 256) and f.c == f.cp and not f.bcast and f.flat_ok()
                and (f.lo in (0, f.cp)))

    def _head1_w(self):
        """the head's weight row as a contiguous fp32 vector (a view of the master weights: [1, C, 1, 1])"""
        return self.w.reshape(-1)

    def out_size(self, H, W):
        k, s, p, d = self.k, self.stride, self.pad, self.dil
        if self.transposed:
            return (H - 1) * s - 2 * p + k, (W - 1) * s - 2 * p + k
        return (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1

    def _launch(self, xs, wt, transposed, k, stride, pad, dil, H, W, OH, OW, cout, out, out32, bias, act, slope, prelu, res, res2,
                res_mode, accumulate, stat, stat_mode, out_scale, cbias=None, mask=None, hr=None, tp=None, dact=None, x3=None, dres=None,
                cb_mode=0, split_blocks=3, x3n=None):
        d = L.ConvDesc()
        x0 = xs[0]
        if x0.lo:                       # split-fp16 input: [hi | lo] (+ hi again for the x_hi w_lo block), weights from _pack_split
            assert len(xs) == 1 and not transposed and split_blocks in (2, 3, 4, 5)      # 4: the fused three-product form, 5: the fused two-product form
            s0, s1 = x0.split_segs()
            d.inp[0] = s0
            if split_blocks == 3:
                d.inp[1] = s1
            d.split_fused = 1 if split_blocks == 4 else (2 if split_blocks == 5 else 0)
        else:
            d.inp[0] = x0.seg()
            if len(xs) > 1:
                d.inp[1] = xs[1].seg()
        d.N, d.H, d.W, d.OH, d.OW = x0.N, H, W, OH, OW
        d.transposed, d.KH, d.KW, d.stride, d.pad, d.dil = int(transposed), k, k, stride, pad, dil
        d.wt, d.cout = _ptr(wt), cout
        d.coutp = pad8(cout)
        if out is not None:
            assert out.cp == d.coutp and (out.H, out.W) == (OH, OW), (out.cp, d.coutp, out.H, OH)
            sn, sy, sx = out.strides()
            d.out16, d.o_sn, d.o_sy, d.o_sx, d.o_lo = _ptr(out.t), sn, sy, sx, out.lo
        if out32 is not None:           # fp32 NCHW planar [N, cout, OH, OW]
            assert out32.is_contiguous() and tuple(out32.shape) == (x0.N, cout, OH, OW)
            d.out32, d.o32_sn, d.o32_sy, d.o32_sx, d.o32_sc = _ptr(out32), cout * OH * OW, OW, 1, OH * OW
        d.bias, d.cbias, d.act, d.act_slope, d.prelu = _ptr(bias), _ptr(cbias), act, slope, _ptr(prelu)
        if bias is not None and bias.dim() == 2:          # per-sample bias rows (Conv._dc_bias)
            d.bias_sn = bias.shape[1]
        d.cbias_mode = cb_mode
        d.res_mode = res_mode
        if res is not None:
            sn, sy, sx = res.strides()
            d.res, d.r_sn, d.r_sy, d.r_sx, d.r_lo = _ptr(res.t), sn, sy, sx, res.lo
        if res2 is not None:
            sn, sy, sx = res2.strides()
            d.res2, d.r2_sn, d.r2_sy, d.r2_sx, d.r2_lo = _ptr(res2.t), sn, sy, sx, res2.lo
        d.accumulate, d.stat_mode, d.stat, d.out_scale = int(accumulate), stat_mode, _ptr(stat), out_scale
        if mask is not None:            # (saved forward output of the consumer layer, its negative slope): see csbsr_conv_desc_t.mask
            mfm, mslope = mask
            assert out is not None and mfm.cp == d.coutp and (mfm.H, mfm.W) == (OH, OW) and not mfm.bcast
            sn, sy, sx = mfm.strides()
            if torch.is_tensor(mslope):      # a learned PReLU slope: read on the device (csbsr_conv_desc_t::mask_prelu)
                d.mask, d.m_sn, d.m_sy, d.m_sx, d.mask_slope, d.mask_prelu = _ptr(mfm.t), sn, sy, sx, 0.0, _ptr(mslope)
            else:
                d.mask, d.m_sn, d.m_sy, d.m_sx, d.mask_slope = _ptr(mfm.t), sn, sy, sx, float(mslope)
        if x0.lo and split_blocks in (4, 5) and not L.load().csbsr_conv_split_fused_eligible(C.byref(d)):
            if split_blocks == 4:
                d.split_fused, d.inp[1], d.wt, split_blocks = 0, x0.split_segs()[1], _ptr(self._fs_fallback()), 3
            else:
                d.split_fused, d.wt, split_blocks = 0, _ptr(self._fs_fallback()), 2
        self.last_fused = False
        use_tp = tp is not None and self.eng.use_tp
        if dact is not None:
            # the layer below (a Conv with PReLU / bias) whose activation derivative AND bias / slope gradient sums this launch would
            # take over from the stand-alone epilogue-backward pass: only the phase-decomposed transposed kernel can (csrc/conv_tp.hip)
            below, saved = dact
            fz = getattr(below, "frozen", False)
            sn, sy, sx = saved.strides()
            d.mask, d.m_sn, d.m_sy, d.m_sx, d.mask_slope = _ptr(saved.t), sn, sy, sx, float(below.slope)
            d.mask_prelu = _ptr(below.prelu)
            d.dact_bias = None if (below.b is None or fz) else _ptr(grad_acc(below.b))
            d.dact_prelu = None if (below.prelu is None or fz) else _ptr(grad_acc(below.prelu))
            if dres is not None:        # the layer below was out = act(pre) +- res: (res FM, FM receiving d(res), its res_mode)
                rfm, dfm, rmode = dres
                sn, sy, sx = rfm.strides()
                d.res_mode, d.res, d.r_sn, d.r_sy, d.r_sx = rmode, _ptr(rfm.t), sn, sy, sx
                sn, sy, sx = dfm.strides()
                d.dres, d.dr_sn, d.dr_sy, d.dr_sx = _ptr(dfm.t), sn, sy, sx
            if use_tp and L.load().csbsr_conv_tp_eligible(C.byref(d)):
                self.last_fused = True
            elif (dres is not None and self.eng.thin_dact and L.load().csbsr_conv_thin_dact_eligible(C.byref(d))
                  and (below.prelu is None or self.eng.prelu_fold_ok(below.prelu))):      # (the kernel divides its slope sum by the slope)
                self.last_fused = True      # the thin-input accumulating dgrad (csrc/conv_thin.hip, DACT): csbsr_conv_forward dispatches it
                use_tp = False
            else:
                d.mask, d.mask_prelu, d.dact_bias, d.dact_prelu = None, None, None, None
                if dres is not None:
                    d.res_mode, d.res, d.dres = L.RES_NONE, None, None
        tm = self.eng.timing
        if tm is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        # full-resolution 32 / 49-channel 3x3 layers: the direct kernel (csrc/conv_hr.hip) with its own fragment-ordered weights
        # (the wide form of conv_x3n goes first: the one launch both take -- the 32 -> 128-channel gather of a stage's thin dgrads at full
        # resolution -- measured 4.8 -> 2.6 ms on it, csrc/conv_x3n.hip)
        xn = (L.load().csbsr_conv_x3n_eligible(C.byref(d)) if (x3n is not None and self.eng.use_x3n and (not x0.lo or split_blocks == 5)) else 0)
        if xn != 2 and hr is not None and self.eng.use_hr and L.load().csbsr_conv_hr_eligible(C.byref(d)):
            kind, c_real, rows_real, row_off = hr
            key = ("hr", kind, row_off)
            if key not in self._packed:
                n = L.load().csbsr_packed_weight_elems_hr(k, c_real, rows_real)
                dst = torch.empty(n, dtype=torch.float16, device=self.eng.device)
                L.call("csbsr_pack_weights_hr", _ptr(self._wq()), _ptr(dst), kind, k, self.w.shape[0], self.w.shape[1], c_real, rows_real, row_off, 0,
                       self.eng.stream)
                self._packed[key] = dst
            d.wt = _ptr(self._packed[key])
            L.call("csbsr_conv_hr_forward", C.byref(d), self.eng.stream)
        elif xn:
            # many input channels -> <= 64 output channels at full resolution (PSPNet_BlurSkip's conv1's, the dgrads of its conv0's): resident
            # pixel tile + streamed fragment-ordered weights (csrc/conv_x3n.hip); a split input runs its two-product plan as 2 x Cp plain
            # channels against [w | w] (the tap-sum-rounded, pre-scaled weights repeated for the lo plane)
            kind, c_real, rows_real, row_off, k_off = x3n
            plane = x0.cp if x0.lo else 0
            in_ch = 2 * x0.cp if x0.lo else x0.cp
            key = ("x3n", kind, row_off, k_off, c_real, plane)
            if key not in self._packed:
                n = L.load().csbsr_packed_weight_elems_x3n(in_ch, rows_real)
                dst = torch.empty(n, dtype=torch.float16, device=self.eng.device)
                L.call("csbsr_pack_weights_x3n", _ptr(self._wq()), _ptr(dst), kind, self.w.shape[0], self.w.shape[1], c_real, rows_real, row_off,
                       k_off, in_ch, plane, self.WSCALE if x0.lo else 1.0, self.eng.stream)
                self._packed[key] = dst
            d.wt = _ptr(self._packed[key])
            L.call("csbsr_conv_x3n_forward", C.byref(d), self.eng.stream)
        elif (x3 is not None and x3[0] in (0, 1) and self.eng.use_x3w and (self.winograd or self.eng.use_x3w == 2)
              and L.load().csbsr_conv_x3w_eligible(C.byref(d))):
            # wide low-resolution 3x3 stride-1 layers (SFT convs and their dgrads) as Winograd F(2, 3) along x: 2/3 of the direct kernel's
            # MFMA work, transformed fp16 weights packed once per optimiser step (csrc/conv_x3w.hip)
            kind, c_real, rows_real, row_off, k_off = x3
            key = ("x3w", kind, row_off, k_off, c_real)
            if key not in self._packed:
                n = L.load().csbsr_packed_weight_elems_x3w(c_real, rows_real)
                dst = torch.empty(n, dtype=torch.float16, device=self.eng.device)
                L.call("csbsr_pack_weights_x3w", _ptr(self._wq()), _ptr(dst), kind, self.w.shape[0], self.w.shape[1], c_real, rows_real, row_off,
                       k_off, self.eng.stream)
                self._packed[key] = dst
            d.wt = _ptr(self._packed[key])
            L.call("csbsr_conv_x3w_forward", C.byref(d), self.eng.stream)
        elif x3 is not None and self.eng.use_x3 and L.load().csbsr_conv_x3_eligible(C.byref(d)):
            # wide low-resolution 3x3 layers (SFT convs and their dgrads): per-chunk halo tile + fragment-ordered weights from L2 (csrc/conv_x3.hip)
            # ... and the k = 2 x stride strided layers (kind 2: 8x8 stride-4 convs, dgrads of the 8x8 stride-4 deconvs): chunk = input phase
            kind, c_real, rows_real, row_off, k_off = x3
            key = ("x3", kind, row_off, k_off, c_real)
            if key not in self._packed:
                if kind == 2:
                    n = L.load().csbsr_packed_weight_elems_x3_strided(stride, c_real, rows_real)
                    dst = torch.empty(n, dtype=torch.float16, device=self.eng.device)
                    L.call("csbsr_pack_weights_x3_strided", _ptr(self._wq()), _ptr(dst), self.w.shape[0], self.w.shape[1], k, stride, c_real,
                           rows_real, row_off, k_off, self.eng.stream)
                else:
                    n = L.load().csbsr_packed_weight_elems_x3(c_real, rows_real)
                    dst = torch.empty(n, dtype=torch.float16, device=self.eng.device)
                    L.call("csbsr_pack_weights_x3", _ptr(self._wq()), _ptr(dst), kind, self.w.shape[0], self.w.shape[1], c_real, rows_real, row_off,
                           k_off, self.eng.stream)
                self._packed[key] = dst
            d.wt = _ptr(self._packed[key])
            L.call("csbsr_conv_x3_forward", C.byref(d), self.eng.stream)
        elif use_tp and L.load().csbsr_conv_tp_eligible(C.byref(d)):
            # 2x2-tap transposed layers (8x8 stride 4 / 12x12 stride 8): resident halo tile + streamed fragment-ordered weights (csrc/conv_tp.hip)
            c_real, rows_real, row_off, k_off = tp
            key = ("tp", row_off, k_off)
            if key not in self._packed:
                n = L.load().csbsr_packed_weight_elems_tp(stride, c_real)
                dst = torch.empty(n, dtype=torch.float16, device=self.eng.device)
                L.call("csbsr_pack_weights_tp", _ptr(self._wq()), _ptr(dst), self.w.shape[0], self.w.shape[1], k, k, stride, pad, c_real, rows_real,
                       row_off, k_off, self.eng.stream)
                self._packed[key] = dst
            d.wt = _ptr(self._packed[key])
            L.call("csbsr_conv_tp_forward", C.byref(d), self.eng.stream)
        else:
            L.call("csbsr_conv_forward", C.byref(d), self.eng.stream)
        if tm is not None:
            ev1.record()
            # algorithmic work of the layer: ONE product per (pixel, cout, cin, tap).  The split-precision launches execute more --
            # three K blocks for a hi+lo input against hi+lo weights, two for a gradient against hi+lo weights (the same tensor passed
            # twice) -- which goes into the last field, not into the FLOPs
            twice = len(xs) == 2 and xs[0] is xs[1]
            ctot = xs[0].c if twice else sum(f.c for f in xs)
            executed = (2 if split_blocks in (2, 5) else 3) if x0.lo else (2 if twice else 1)
            npx = x0.N * OH * OW
            taps = k * k if not transposed else ((k + stride - 1) // stride) ** 2
            flops = 2.0 * npx * cout * ctot * taps
            # algorithmic bytes: the input(s) once, the output once, and every map the fused epilogue reads (residuals, the accumulated-into
            # output, the activation-derivative mask) or writes besides (d(res)): what a fused launch cannot avoid moving
            nepi = sum(1 for f in (res, res2) if f is not None and not f.bcast) + int(bool(accumulate)) + int(mask is not None or dact is not None) + int(dres is not None)
            nbytes = 2.0 * (sum(0 if f.bcast else f.N * f.H * f.W * f.c for f in (xs[:1] if twice else xs)) + (npx * cout if out is not None else 0)
                            + nepi * npx * cout)
            tm.append(("conv", flops, nbytes, ev0, ev1, self.name, (x0.N, H, W, ctot, cout, k, stride, int(transposed)),
                       int(L.load().csbsr_debug_last_conv_kernel()), executed))

    def fwd(self, x, out=None, out32=None, res=None, res2=None, res_mode=L.RES_NONE, stat=None, stat_mode=L.STAT_NONE, store=True):
        xs = x if isinstance(x, (tuple, list)) else (x,)
        assert tuple(f.c for f in xs) == tuple(c for c in self.split if c > 0), (self.name, [f.c for f in xs], self.split)
        H, W = xs[0].H, xs[0].W
        OH, OW = self.out_size(H, W)
        sp = bool(xs[0].lo)
        if self._head1_ok(xs, out, out32, res, stat):
            # 1-channel head (PSPNet's ``final`` / aux classifier): fp32 VALU dot product over the pixel's channels, one streaming pass
            # (csbsr_head1_fwd) instead of a 32-row MFMA tile for one output channel; the split planes are summed, the weights not rounded
            f = xs[0]
            tm = self.eng.timing
            if tm is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            L.call("csbsr_head1_fwd", _ptr(f.t), f.ld, f.lo, f.cp, _ptr(self._head1_w()), _ptr(self.b), int(self.act == L.ACT_SIGMOID),
                   _ptr(out32), f.npix, self.eng.stream)
            if tm is not None:
                ev1.record()
                tm.append(("conv", 2.0 * f.npix * f.c, 2.0 * f.npix * f.c * (2 if sp else 1) + 4.0 * f.npix, ev0, ev1, self.name,
                           (f.N, H, W, f.c, 1, 1, 1, 0), 21, 1))
            return None
        if out is None and store and out32 is None:
            out = self.eng.new(xs[0].N, OH, OW, self.cout, split=sp)
        osc, nb = 1.0, 3
        x_in = xs
        if sp:
            assert not self.transposed and self.split[1] == 0
            xs, wt, osc, nb = self._split_operand(xs[0], "fwd_split", 0, self.cin, self.cout, self.stride, self.pad)
        elif self.transposed:
            wt = self._pack("fwd", 2, self.split[0], self.split[1], 0, self.cout, self.stride, self.pad)
        else:
            wt = self._pack("fwd", 0, self.split[0], self.split[1], 0, self.cout, self.stride, self.pad)
        hr = (0, self.cin, self.cout, 0) if (not sp and not self.transposed and self.k in (1, 3) and len(xs) == 1 and self.prelu is None) else None
        # (split inputs: only where the layer's plan keeps the weight's rounding, i.e. runs fewer than three products)
        bias = self._dc_bias(x_in) if (self.dc_comp and self.eng.dc_comp and not self.transposed and (not sp or self.fwd_blocks < 3)
                                       and (OH * OW) % 256 == 0 and self.cin > 8) else self.b
        self._launch(xs, wt, self.transposed, self.k, self.stride, self.pad, self.dil, H, W, OH, OW, self.cout, out, out32, bias,
                     self.act, self.slope, self.prelu, res, res2, res_mode, False, stat, stat_mode, osc, hr=hr,
                     tp=(self.cin, self.cout, 0, 0) if (self.transposed and len(xs) == 1 and not sp) else None,
                     x3=((0 if self.k == 3 else 2, self.cin, self.cout, 0, 0)
                         if (not sp and not self.transposed and len(xs) == 1 and (self.k == 3 or self.k == 2 * self.stride)) else None),
                     split_blocks=nb,
                     x3n=(0, self.cin, self.cout, 0, 0) if (not self.transposed and len(xs) == 1 and self.k == 3 and self.stride == 1
                                                           and (stat is None or stat_mode == L.STAT_BN)) else None)
        return out

    def bwd_input(self, dpre, seg=0, out=None, accumulate=False, out32=None, stat=None, in_hw=None, mask=None, dact=None, dres=None):
        """dgrad wrt input segment ``seg``; dpre: gradient wrt the pre-activation output (FM, may be bcast).
        ``mask`` = (saved output FM of the layer that produced this input, its negative slope): the activation derivative of that layer
        is applied in the epilogue, so what leaves is its dPre (only on the launch that completes the gradient).
        ``dact`` = (Conv of the layer below, its saved output FM): like ``mask`` but also that layer's bias / PReLU-slope gradient sums,
        i.e. its whole epilogue-backward pass; taken only where the fused kernel exists -- ``self.last_fused`` tells the caller
        whether it still has to run that pass.  ``dres`` = (res FM, FM for d(res), res_mode) when that layer was out = act(pre) +- res."""
        c_seg = self.split[seg]
        row_off = 0 if seg == 0 else self.split[0]
        k, s, p, d = self.k, self.stride, self.pad, self.dil
        H, W = dpre.H, dpre.W
        if (self.eng.use_head1 and self.cout == 1 and k == 1 and s == 1 and not self.transposed and seg == 0 and self.split[1] == 0
                and c_seg % 8 == 0 and not accumulate and out32 is None and stat is None and mask is None and dact is None and dres is None
                and not dpre.bcast and dpre.flat_ok()):
            # dgrad of a 1-channel head: dX[pixel][c] = dPre[pixel] w[c] as one streaming store pass (csbsr_head1_bwd_input)
            if out is None:
                out = self.eng.new(dpre.N, H, W, c_seg)
            if out.flat_ok() and not out.lo:
                tm = self.eng.timing
                if tm is not None:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                L.call("csbsr_head1_bwd_input", _ptr(dpre.t), dpre.ld, _ptr(self._head1_w()), c_seg, _ptr(out.t), out.ld, dpre.npix, self.eng.stream)
                if tm is not None:
                    ev1.record()
                    tm.append(("conv", 2.0 * dpre.npix * c_seg, 2.0 * dpre.npix * (c_seg + 8), ev0, ev1, self.name,
                               (dpre.N, H, W, 1, c_seg, 1, 1, 0), 22, 1))
                return out
        hp = self.hp_dgrad and not dpre.bcast and not self.transposed
        if hp:       # weights as fp16 hi + lo pairs against the (plain fp16) gradient passed twice
            kind = 1 if s == 1 else 2
            wt = self._pack_split(("dg_hp", seg), kind, self.cout, c_seg, s if kind == 2 else 1, p, layout=1, row_off=row_off)
            if s == 1:
                OH, OW = in_hw if in_hw else (H + 2 * (d * (k - 1) - p) - d * (k - 1), W + 2 * (d * (k - 1) - p) - d * (k - 1))
                tr, ps, pp, dd = False, 1, d * (k - 1) - p, d
            else:
                assert d == 1 and in_hw is not None
                OH, OW = in_hw
                tr, ps, pp, dd = True, s, p, 1
        elif self.transposed:                      # dgrad of a transposed conv = strided conv on dOut
            wt = self._pack(("dg", seg), 0, self.cout, 0, row_off, c_seg, s, p)
            OH, OW = in_hw if in_hw else ((H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1)
            tr, ps, pp, dd = False, s, p, 1
        elif s == 1:                             # flipped kernel, pad' = d(k-1) - p
            wt = self._pack(("dg", seg), 1, self.cout, 0, row_off, c_seg, 1, p)
            OH, OW = in_hw if in_hw else (H + 2 * (d * (k - 1) - p) - d * (k - 1), W + 2 * (d * (k - 1) - p) - d * (k - 1))
            tr, ps, pp, dd = False, 1, d * (k - 1) - p, d
        else:                                    # strided conv: gather-form transposed conv on dOut
            assert d == 1 and in_hw is not None
            wt = self._pack(("dg", seg), 2, self.cout, 0, row_off, c_seg, s, p)
            OH, OW = in_hw
            tr, ps, pp, dd = True, s, p, 1
        if out is None and out32 is None and stat is None:
            out = self.eng.new(dpre.N, OH, OW, c_seg)
        hr = (1, self.cout, c_seg, row_off) if (not hp and not self.transposed and s == 1 and k in (1, 3) and stat is None) else None
        self._launch((dpre, dpre) if hp else (dpre,), wt, tr, k, ps, pp, dd, H, W, OH, OW, c_seg, out, out32, None, L.ACT_NONE, 0.0, None,
                     None, None, L.RES_NONE, accumulate, stat, L.STAT_SAMPLE_SUM if stat is not None else L.STAT_NONE,
                     1.0 / self.WSCALE if hp else 1.0, mask=mask, hr=hr,
                     tp=(self.cout, c_seg, row_off, 0) if (tr and not hp and stat is None) else None, dact=dact, dres=dres,
                     x3=((2, self.cout, c_seg, row_off, 0) if (self.transposed and not hp and stat is None and k == 2 * s) else
                         (1, self.cout, c_seg, row_off, 0) if (not hp and not self.transposed and s == 1 and k == 3 and stat is None) else None),
                     x3n=(1, self.cout, c_seg, row_off, 0) if (not hp and not self.transposed and s == 1 and k == 3 and stat is None and dact is None) else None)
        return out

    # -- exact folding of a spatially constant second input segment (SFT conv0: cat(features, kernel code), kbpn.py:513)
    def fwd_folded(self, x, kvec, mtap, out=None):
        """conv over the feature segment only; the constant segment enters as a per-(sample, border class) bias
        T = W_const . k, 16 masked tap sums (see csbsr_border_class_fill)."""
        assert not self.transposed and self.stride == 1 and self.k == 3 and self.split[1] > 0
        cf = self.split[0]
        B = x.N
        sp = bool(x.lo)
        xs, osc, nb = (x,), 1.0, 3
        if sp:
            xs, wt, osc, nb = self._split_operand(x, "fwd_feat_split", 0, cf, self.cout, 1, self.pad)
        else:
            wt = self._pack("fwd_feat", 0, cf, 0, 0, self.cout, 1, self.pad, 0)
        # the constant part is a tiny fp32 mat-vec on the fp32 master weights and the fp32 kernel code (rounds 1-3 rounded both to fp16 in the
        # plain mode "to mirror the MFMA path": a precision loss the reference does not have and the fold does not need)
        w16c = self.w[:, cf:]
        k16 = kvec.to(torch.float32)
        T = torch.einsum("ocyx,nc->noyx", w16c, k16)
        V = torch.einsum("noyx,ay,bx->nabo", T, mtap, mtap).reshape(B, 16, self.cout)
        cb = self.eng.f32(B, 16, pad8(self.cout))
        cb[:, :, :self.cout] = V
        H, W = x.H, x.W
        if out is None:
            out = self.eng.new(B, H, W, self.cout, split=sp)
        bias = self._dc_bias((x,)) if (self.dc_comp and self.eng.dc_comp and (not sp or self.fwd_blocks < 3) and (x.H * x.W) % 256 == 0) else self.b      # (the feature segment; the constant one is an fp32 mat-vec)
        self._launch(xs, wt, False, 3, 1, self.pad, self.dil, H, W, H, W, self.cout, out, None, bias, self.act, self.slope, self.prelu,
                     None, None, L.RES_NONE, False, None, L.STAT_NONE, osc, cbias=cb,
                     x3=None if sp else (0, cf, self.cout, 0, 0), split_blocks=nb, x3n=(0, cf, self.cout, 0, 0))
        return out, (w16c, k16)

    def fwd_classbias(self, x, cb, cb_mode, out=None):
        """conv over input segment 0 only; segment 1 -- a map that is constant within each position class (``cb_mode`` 0: the 16 border
        classes, 1: the 25 two-ring classes, include/csbsr_hip.h) -- enters through ``cb`` [B, classes, pad8(cout)] fp32, its exact
        contribution per (sample, class) computed by the caller.  Plain fp16 inputs."""
        assert not self.transposed and self.stride == 1 and self.split[1] > 0 and not x.lo
        cf = self.split[0]
        wt = self._pack("fwd_feat", 0, cf, 0, 0, self.cout, 1, self.pad, 0)
        if out is None:
            out = self.eng.new(x.N, x.H, x.W, self.cout)
        self._launch((x,), wt, False, self.k, 1, self.pad, self.dil, x.H, x.W, x.H, x.W, self.cout, out, None, self.b, self.act, self.slope,
                     self.prelu, None, None, L.RES_NONE, False, None, L.STAT_NONE, 1.0, cbias=cb, cb_mode=cb_mode,
                     hr=(0, cf, self.cout, 0) if (self.k == 1 and cb_mode == 1 and self.prelu is None) else None)
        return out

    def fwd_const_1x1(self, cvec, x, out=None, stat=None, stat_mode=L.STAT_NONE):
        """1x1 conv over cat(spatially constant vector [B, c0], x): the constant segment is a per-sample bias W[:, :c0] . cvec (exact; a
        1x1 conv has no border classes).  Used for split-fp16 inputs, which cannot be paired with a plain fp16 broadcast segment."""
        assert self.k == 1 and not self.transposed and self.stride == 1
        c0 = self.split[0]
        B = x.N
        sp = bool(x.lo)
        xs, osc, nb = (x,), 1.0, 3
        if sp:
            xs, wt, osc, nb = self._split_operand(x, "fwd_x_split", 0, self.split[1], self.cout, 1, 0, k_off=c0)
        else:
            wt = self._pack("fwd_x", 0, self.split[1], 0, 0, self.cout, 1, 0, c0)
        T = cvec.to(torch.float32) @ self.w[:, :c0, 0, 0].t()              # [B, cout]
        cb = self.eng.f32(B, 16, pad8(self.cout))
        cb[:, :, :self.cout] = T[:, None, :]
        if out is None:
            out = self.eng.new(B, x.H, x.W, self.cout, split=sp)
        self._launch(xs, wt, False, 1, 1, 0, 1, x.H, x.W, x.H, x.W, self.cout, out, None, self.b, self.act, self.slope, self.prelu,
                     None, None, L.RES_NONE, False, stat, stat_mode, osc, cbias=cb, split_blocks=nb)
        return out

    def bwd_weights_folded(self, dpre, x, saved, mtap, frozen=False, bias_grad=False, prelu_out=None):
        """wgrad of the feature part on the MFMA, of the constant part from 16 border-class sums of dPre; returns dL/dk [B, c_const].
        ``bias_grad``: the layer's bias gradient -- the sum of dPre over samples and pixels -- is the sum of the 16 class sums: callers
        that fused this layer's activation derivative into the dgrad above it (``bwd_input(mask=)``) get it here instead of from an
        epilogue-backward pass over the map.  ``prelu_out``: this (PReLU) layer's saved output -- its slope gradient is taken in the same pass."""
        w16c, k16 = saved
        cf = self.split[0]
        if not frozen:
            self.bwd_weights(dpre, x, split_override=(cf, 0))
        B = dpre.N
        sums = self.eng.f32(B, 16, dpre.cp)
        if prelu_out is not None and self.prelu is not None and not frozen:
            # a PReLU layer: the slope gradient sum_{pre <= 0} dOut * pre = sum_{out <= 0} dPre * out / slope^2 comes out of the same pass
            negdot = self.eng.f32(B, dpre.cp)
            L.call("csbsr_border_class_sums_prelu", _ptr(dpre.t), dpre.ld, _ptr(prelu_out.t), prelu_out.ld, _ptr(sums), _ptr(negdot), B, dpre.H,
                   dpre.W, dpre.cp, self.eng.stream)
            grad_acc(self.prelu).add_((negdot[:, :self.cout].sum() / (self.prelu.detach() * self.prelu.detach())).reshape(self.prelu.shape))
        else:
            L.call("csbsr_border_class_sums", _ptr(dpre.t), dpre.ld, _ptr(sums), B, dpre.H, dpre.W, dpre.cp, self.eng.stream)
        if bias_grad and self.b is not None and not frozen:
            grad_acc(self.b).add_(sums.sum((0, 1))[:self.cout])
        S = torch.einsum("nabo,ay,bx->noyx", sums[:, :, :self.cout].reshape(B, 4, 4, self.cout), mtap, mtap)
        if not frozen:
            grad_acc(self.w)[:, cf:].add_(torch.einsum("noyx,nc->ocyx", S, k16))
        return torch.einsum("ocyx,noyx->nc", w16c, S)

    def bwd_weights(self, dpre, x, split_override=None):
        """wgrad; accumulates (scaled) into self.w.gacc [same shape as w] fp32.  Runs on the engine's wgrad stream when there is one:
        the operands are final when this is called (every call site), so the side stream only has to start after the caller's
        stream got here, and the operands' memory must not be handed out again before the side stream is done with it."""
        side = self.eng.wgrad_stream()
        if side is None:
            return self._bwd_weights_impl(dpre, x, split_override)
        side.wait_stream(torch.cuda.current_stream(self.eng.device))
        for f in ((dpre,) + (tuple(x) if isinstance(x, (tuple, list)) else (x,))):
            f.t.record_stream(side)
        with torch.cuda.stream(side):
            self._bwd_weights_impl(dpre, x, split_override)

    def thin_tp_fused_ok(self, x):
        """kb.up_conv1's shape: ConvTranspose2d(3 -> C, 8x8, stride 4) + PReLU without bias -- csbsr_thin_tp_backward takes its backward"""
        return (self.transposed and self.k == 8 and self.stride == 4 and self.cin == 3 and self.b is None and self.act == L.ACT_PRELU
                and self.prelu is not None and self.cout % 8 == 0 and 8 <= self.cout <= 128 and 64 % (self.cout // 8) == 0 and x.W <= 1024
                and self.eng.thin_tp_fused)

    def bwd_thin_tp_fused(self, dout, x, dpre, frozen=False):
        """dOut -> dPre, weight gradient and PReLU-slope gradient of the layer in ONE pass over dOut: the pre-activation is rebuilt from
        the 3-channel input (csrc/conv_kbup.hip), so the epilogue-backward pass over (dOut, saved output, residual) and the separate
        weight-gradient launch are gone.  ``dout`` is left untouched (it lives on as the residual's gradient)."""
        wt = self._pack("fwd", 2, self.split[0], self.split[1], 0, self.cout, self.stride, self.pad)
        N, h, w = x.N, x.H, x.W
        ns = int(L.load().csbsr_thin_tp_backward_slabs(N, h))
        slabs = None if frozen else self.eng.workspace(ns * 8 * 64 * self.cout)
        part = None if frozen else self.eng.f32(4 * ns, zero=False)
        L.call("csbsr_thin_tp_backward", _ptr(dout.t), *dout.strides(), _ptr(x.t), *x.strides(), _ptr(wt), 3, self.cout, self.stride, self.pad,
               _ptr(self.prelu), N, h, w, _ptr(dpre.t), *dpre.strides(), _ptr(slabs), _ptr(part), self.eng.stream)
        if not frozen:
            L.call("csbsr_unpack_wgrad", _ptr(slabs), _ptr(grad_acc(self.w)), 3, 8, 8, self.cout, 0, self.w.shape[0], self.w.shape[1], 0, 0,
                   1.0, ns, 8, self.eng.stream)
            grad_acc(self.prelu).add_(part.sum())
        return dpre

    def _bwd_weights_impl(self, dpre, x, split_override=None):
        xs = x if isinstance(x, (tuple, list)) else (x,)
        # A layer with <= 64 output channels fills half (or a tenth) of the kernels' 128-row tiles.  For a stride-1 "same" conv the MIRRORED
        # problem  G'[ci][tap'][co] = sum_pix X[pix][ci] dPre[pix - off(tap')][co]  is the same sum with the roles swapped -- rows = input
        # channels (one launch per input segment), columns = taps x output channels, taps mirrored -- and csbsr_unpack_wgrad writes it
        # back transposed and flipped.  Same-process A/B (scripts/wgrad_mirror_ab.py): config 5's 505 -> 64 HR layers 18.7 -> 10.9 ms per
        # launch, the decoder's 256 -> 64 4.35 -> 2.49, and the 3-channel image heads 512 -> 3 3.86 -> 2.37 / 128 -> 3 1.07 -> 0.68
        # (against the taps-in-rows kernel written for them, csrc/conv_wgrad.hip).
        mirror = (self.eng.wgrad_mirror and not self.transposed and self.stride == 1 and self.cout <= 64
                  and 2 * self.pad == self.dil * (self.k - 1) and all(f.cp >= 128 and not f.bcast for f in xs))
        if mirror:
            off = 0
            for i, f in enumerate(xs):
                creal = f.c if split_override is None else split_override[i]
                self._wgrad_launch(f, (dpre,), self.cout, 0, creal, 3, off)
                off += creal
        elif self.transposed:                    # A = input (LR), B = dOut (HR)
            assert len(xs) == 1
            self._wgrad_launch(xs[0], (dpre,), self.cout, 0, self.w.shape[0], 0, 0)
        else:
            seg0, seg1 = split_override if split_override is not None else self.split
            self._wgrad_launch(dpre, xs, seg0, seg1, self.w.shape[0], 0, 0)

    def _wgrad_launch(self, a, bs, seg0, seg1, A_real, unpack_mode, a_off):
        """G[a][tap][b] = sum_pix A[pix][a] B[pix @ tap][b] into fp32 slabs, then the slabs into the master parameter's accumulator
        (unpack_mode 3 = the mirrored problem: rows index the weight's dim 1 from ``a_off``, taps flipped)."""
        d = L.WgradDesc()
        sn, sy, sx = a.strides()
        d.a, d.a_sn, d.a_sy, d.a_sx, d.ca, d.ca_real = _ptr(a.t), sn, sy, sx, a.cp, a.c
        d.b[0] = bs[0].seg()
        if len(bs) > 1:
            d.b[1] = bs[1].seg()
        d.N, d.AH, d.AW, d.BH, d.BW = a.N, a.H, a.W, bs[0].H, bs[0].W
        d.KH, d.KW, d.stride, d.pad, d.dil = self.k, self.k, self.stride, self.pad, self.dil
        cbtot = sum(f.cp for f in bs)
        ktot = self.k * self.k * cbtot
        splits = L.load().csbsr_wgrad_splits_desc(C.byref(d))
        g = self.eng.workspace(splits * a.cp * ktot)
        d.g, d.splits = _ptr(g), splits
        tm = self.eng.timing
        if tm is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        L.call("csbsr_conv_wgrad", C.byref(d), self.eng.stream)
        if tm is not None:
            ev1.record()
            flops = 2.0 * a.N * a.H * a.W * a.c * sum(f.c for f in bs) * self.k * self.k
            nbytes = 2.0 * (a.N * a.H * a.W * a.c + sum(0 if f.bcast else f.N * f.H * f.W * f.c for f in bs))
            tm.append(("wgrad", flops, nbytes, ev0, ev1, self.name, (a.N, a.H, a.W, a.c, sum(f.c for f in bs), self.k, self.stride, int(self.transposed)),
                       int(L.load().csbsr_debug_last_wgrad_kernel())))
        gacc = grad_acc(self.w)
        L.call("csbsr_unpack_wgrad", _ptr(g), C.c_void_p(gacc.data_ptr() + 4 * a_off * self.k * self.k), A_real, self.k, self.k, seg0, seg1,
               self.w.shape[0], self.w.shape[1], unpack_mode, 0, 1.0, splits, a.cp, self.eng.stream)


class ShuffleConv(Conv):
    """ConvAndPixelShuffleBlock (MODEL.SR_PIXEL_SHUFFLE, /root/reference/model/modeling/kbpn.py:280-289): conv3x3 to cout * s^2
    channels + activation + nn.PixelShuffle(s).

    Output pixel (y s + py, x s + px), channel c is channel c s^2 + py s + px of the conv at (y, x): exactly a transposed convolution
    with a (3s x 3s) kernel, stride s, padding s whose weight is an index permutation of the conv weight,
        Wt[cin][c][(2 - ky) s + py][(2 - kx) s + px] = W[c s^2 + py s + px][cin][ky][kx],
    so the layer runs on the transposed-convolution kernels (forward: s^2 output phases of 3x3 taps; dgrad: a strided conv; wgrad)
    with that derived operand, and the weight gradient is permuted back into the master parameter's accumulator."""

    def __init__(self, eng, name, params, factor, bias=False, act=L.ACT_NONE, slope=0.0, prelu=False):
        self.master = params[name + ".weight"]
        self.factor = s = int(factor)
        derived = {name + ".weight": self._remap(self.master)}
        if bias:
            raise NotImplementedError("ConvAndPixelShuffleBlock carries no bias on this path (kbpn.py: bias=False)")
        if prelu:
            derived[prelu] = params[prelu]
        super().__init__(eng, name, derived, 3 * s, s, s, 1, transposed=True, bias=False, act=act, slope=slope, prelu=prelu)

    def _remap(self, w):
        s = self.factor
        C, cin = w.shape[0] // (s * s), w.shape[1]
        return w.reshape(C, s, s, cin, 3, 3).flip(4, 5).permute(3, 0, 4, 1, 5, 2).reshape(cin, C, 3 * s, 3 * s).contiguous()

    def _unmap(self, g):
        s = self.factor
        cin, C = g.shape[0], g.shape[1]
        return g.reshape(cin, C, 3, s, 3, s).permute(1, 3, 5, 0, 2, 4).flip(4, 5).reshape(C * s * s, cin, 3, 3)

    def invalidate(self):
        super().invalidate()
        self.w.copy_(self._remap(self.master))       # the optimiser steps the master; same tensor object keeps its accumulator

    def _bwd_weights_impl(self, dpre, x, split_override=None):      # (whole body on the wgrad stream: the remap reads what the kernel wrote)
        super()._bwd_weights_impl(dpre, x, split_override)
        grad_acc(self.master).add_(self._unmap(self.w.gacc))
        self.w.gacc.zero_()


def grad_acc(p):
    """fp32 gradient accumulator attached to a master parameter tensor."""
    g = getattr(p, "gacc", None)
    if g is None:
        g = torch.zeros_like(p, dtype=torch.float32)
        p.gacc = g
    p.gacc_touched = True
    return g


# ---------------------------------------------------------------------------------------------- batch norm

class BatchNorm:
    """Train-mode BatchNorm2d fed by the conv epilogue's per-channel sum / sumsq."""

    def __init__(self, eng, name, params, c):
        self.eng, self.name, self.c = eng, name, c
        self.gamma, self.beta = params[name + ".weight"], params[name + ".bias"]
        self.rmean, self.rvar = params[name + ".running_mean"], params[name + ".running_var"]
        self.nbt = params[name + ".num_batches_tracked"]

    def new_stat(self):
        return self.eng.f32(2, pad8(self.c))

    def finalize(self, stat, count, update_running=True):
        cp = pad8(self.c)
        mean, invstd = self.eng.f32(cp, zero=False), self.eng.f32(cp, zero=False)      # (cp == c for BatchNorm layers: csbsr_bn_finalize writes every element)
        L.call("csbsr_bn_finalize", _ptr(stat), count, self.c, cp, 1e-5, 0.1, _ptr(mean), _ptr(invstd),
               _ptr(self.rmean) if update_running else None, _ptr(self.rvar) if update_running else None, self.eng.stream)
        if update_running:
            self.nbt += 1
        return mean, invstd

    def _desc(self, x, mean, invstd, res, act, prelu, drop):
        d = L.BnDesc()
        d.npix, d.hw, d.c, d.creal = x.npix, x.H * x.W, x.cp, self.c
        d.x, d.x_ld, d.x_lo = _ptr(x.t), x.ld, x.lo
        cp = x.cp
        if self.gamma.numel() != cp:            # pad per-channel params once (channels here are always multiples of 8)
            raise L.CsbsrHipError("BatchNorm channel count must be a multiple of 8")
        d.mean, d.invstd, d.gamma, d.beta = _ptr(mean), _ptr(invstd), _ptr(self.gamma), _ptr(self.beta)
        if res is not None:
            d.res, d.res_ld, d.res_lo = _ptr(res.t), res.ld, res.lo
        d.act, d.prelu, d.drop = act, _ptr(prelu), _ptr(drop)
        return d

    def apply(self, x, mean, invstd, act=L.ACT_NONE, prelu=None, res=None, drop=None, out=None):
        if out is None:
            out = self.eng.new(x.N, x.H, x.W, x.c, split=bool(x.lo))
        assert x.flat_ok() and out.flat_ok() and (res is None or res.flat_ok())
        d = self._desc(x, mean, invstd, res, act, prelu, drop)
        d.y, d.y_ld, d.y_lo = _ptr(out.t), out.ld, out.lo
        L.call("csbsr_bn_apply", C.byref(d), self.eng.stream)
        return out

    def backward(self, dy, x, mean, invstd, act=L.ACT_NONE, prelu=None, res=None, drop=None, dres=None, dres_acc=False, dprelu=None):
        """returns dx (gradient wrt the conv output); accumulates gamma/beta grads."""
        dx = self.eng.new(x.N, x.H, x.W, x.c)
        d = self._desc(x, mean, invstd, res, act, prelu, drop)
        d.dy, d.dy_ld = _ptr(dy.t), dy.ld
        red = self.eng.f32(2, x.cp)
        d.red, d.dprelu = _ptr(red), _ptr(dprelu)
        d.dx, d.dx_ld = _ptr(dx.t), dx.ld
        if dres is not None:
            d.dres, d.dres_ld, d.dres_accumulate = _ptr(dres.t), dres.ld, int(dres_acc)
        d.dgamma, d.dbeta = _ptr(grad_acc(self.gamma)), _ptr(grad_acc(self.beta))
        L.call("csbsr_bn_backward", C.byref(d), self.eng.stream)
        return dx
