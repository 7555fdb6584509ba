"""Drop-in counterparts of the reference's ``model.modeling.build_model`` classes for the KBPN + PSPNet path:

    JointModelWithLoss(cfg, num_train_ds, resume_iter, sr_transforms)
        .forward(iter, x, sr_targets=None, segment_targets=None, kernel_targets=None)
            -> (segment_loss[B], sr_loss[B], segment_preds, sr_preds, kernel_preds[B,1,K,K])
    JointModel(cfg).forward(x, damy_kernel, sr_targets=None) -> (sr_preds, segment_preds, kernel_preds)

(/root/reference/model/modeling/build_model.py:323-416, 441-500).  Same constructor / forward signatures,
same state_dict keys, same attributes the trainer touches (``sr_model``, ``segmentation_model``,
``ss_loss_fn.{alpha,fix_alpha,iter,update_alpha()}``, ``iter_cnt``), same per-sample unreduced losses, so
``loss = calc_loss(...); loss.backward(); optimizer.step()`` from trainer.py:67-71 runs unchanged.

Underneath nothing is torch.nn: the forward hands raw device pointers to libcsbsr_hip.so (csbsr_amd/engine.py)
and the two loss vectors are outputs of one torch.autograd.Function whose backward runs the hand-written HIP
backward pass; PyTorch only owns memory, streams and the optimizer.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from .. import _lib as L
from ..config import path_config
from ..engine import Engine, FM, grad_acc, _ptr
from .kbpn import KBPN
from .pspnet import PSPNet
from .hrnet_ocr import HRNetOCR
from .shapes import joint_state_shapes


class BoundaryComboState:
    """alpha schedule of BoundaryComboLoss (loss_functions.py:27-41, 76-81); the loss itself is fused in HIP."""

    def __init__(self, per_epoch, resume_iter=0, alpha_min=0.01, decrease_ratio=1.0):
        self.per_epoch, self.alpha_min, self.decrease_ratio = per_epoch, alpha_min, decrease_ratio
        self.fix_alpha = False
        self.iter = resume_iter % per_epoch
        self.alpha = 1.0 - (resume_iter // per_epoch) * 0.01 * decrease_ratio
        self.alpha = alpha_min if self.alpha <= alpha_min else self.alpha

    def update_alpha(self):
        if self.iter % self.per_epoch == 0 and self.alpha > self.alpha_min and not self.fix_alpha:
            self.alpha -= 0.01 * self.decrease_ratio
            self.iter = 1
        else:
            self.iter += 1


class _ParamGroup(nn.Module):
    """Holds parameters / buffers under the reference's dotted names (``sr_model.feat.0.weight`` ...)."""

    def __init__(self, shapes, prefix):
        super().__init__()
        self._names = {}
        for full, shp in shapes.items():
            if not full.startswith(prefix + "."):
                continue
            local = full[len(prefix) + 1:]
            key = local.replace(".", "__")
            self._names[local] = key
            if full.endswith(("running_mean", "running_var")):
                self.register_buffer(key, torch.zeros(shp) if full.endswith("mean") else torch.ones(shp))
            elif full.endswith("num_batches_tracked"):
                self.register_buffer(key, torch.zeros((), dtype=torch.long))
            else:
                self.register_parameter(key, nn.Parameter(torch.zeros(shp)))

    def named_local(self):
        for local, key in self._names.items():
            yield local, getattr(self, key)


def _init_reference_style(name, t, gen):
    """Random init with the reference's distributions (kbpn.py:73-82 kaiming-normal convs / zero biases, PReLU 0.01;
    extractors.py:124-130 normal(0, sqrt(2/n)) convs, BN gamma=1 beta=0; pspnet PReLU 0.25, default Conv2d init elsewhere)."""
    with torch.no_grad():
        if t.dim() == 4:
            if name.startswith("sr_model") or ".blur_skip." in name:      # blocks.py:53-61 (kaiming / xavier normal)
                fan_in = t.shape[1] * t.shape[2] * t.shape[3]
                t.normal_(0, math.sqrt(2.0 / fan_in), generator=gen)
            elif ".feats." in name:
                n = t.shape[2] * t.shape[3] * t.shape[0]
                t.normal_(0, math.sqrt(2.0 / n), generator=gen)
            else:
                fan_in = t.shape[1] * t.shape[2] * t.shape[3]
                bound = 1.0 / math.sqrt(fan_in)
                t.uniform_(-bound, bound, generator=gen)
        elif name.endswith("act.weight"):
            t.fill_(0.01)
        elif name.endswith("conv.2.weight"):
            t.fill_(0.25)
        elif name.endswith(".weight"):
            t.fill_(1.0)
        else:
            t.zero_()


class _JointBase(nn.Module):
    def __init__(self, cfg, antialias=True, device="cuda:0", seed=None):
        super().__init__()
        if cfg.MODEL.SR != "KBPN" or cfg.MODEL.DETECTOR_TYPE not in ("PSPNet", "PSPNet_BlurSkip", "HRNet_OCR"):
            raise NotImplementedError(f"csbsr_amd builds KBPN + PSPNet / PSPNet_BlurSkip / HRNet_OCR; got SR={cfg.MODEL.SR} "
                                      f"DETECTOR_TYPE={cfg.MODEL.DETECTOR_TYPE}")
        if cfg.MODEL.SUM_LR_ERROR_POS not in ("HR", "LR"):      # kbpn.py:174-187: the reference's forward handles exactly these two
            raise NotImplementedError(f"MODEL.SUM_LR_ERROR_POS={cfg.MODEL.SUM_LR_ERROR_POS!r}: 'HR' or 'LR'")
        if cfg.MODEL.NUM_CLASSES != 1:      # build_model.py:209: every kernel of this path assumes the 1-class crack map
            raise NotImplementedError(f"MODEL.NUM_CLASSES={cfg.MODEL.NUM_CLASSES}: only the 1-class detectors are built")
        if cfg.MODEL.SR_SEG_INV or not cfg.MODEL.JOINT_LEARNING:
            raise NotImplementedError("only MODEL.JOINT_LEARNING=True with SR_SEG_INV=False (SR feeds the detector, one joint loss) is built")
        self.cfg = cfg
        self.pc = path_config(cfg, antialias)
        self.scale_factor = cfg.MODEL.SCALE_FACTOR
        self.norm_method = cfg.SOLVER.NORM_SR_OUTPUT
        self.seg_model_name = cfg.MODEL.DETECTOR_TYPE
        self.blur_skip = self.seg_model_name == "PSPNet_BlurSkip"
        self._device = torch.device(device)
        shapes = joint_state_shapes(self.pc.scale, self.pc.num_stages, self.pc.ksize, self.pc.ksize_out, self.seg_model_name,
                                    pixel_shuffle=self.pc.pixel_shuffle, kernel_sft=self.pc.kernel_sft, lr_error=self.pc.lr_error,
                                    zero_pad_kernel=self.pc.zero_pad_kernel)
        # registration order = reference state_dict order: segmentation_model.* then sr_model.*
        self.segmentation_model = _ParamGroup(shapes, "segmentation_model")
        self.sr_model = _ParamGroup(shapes, "sr_model")
        gen = torch.Generator().manual_seed(cfg.SEED if seed is None else seed)
        for full, t in self._named_full():
            if t.is_floating_point() and not full.endswith(("running_mean", "running_var")):
                _init_reference_style(full, t.data, gen)
        if self.blur_skip:        # build_model.py:352-368: everything but blur_skip.* is fixed
            for full, t in self._named_full():
                if isinstance(t, nn.Parameter):
                    t.requires_grad = ".blur_skip." in full
        self._rt = None
        # KBPN runs in micro-batches (exact: no batch-coupled op).  Larger ones amortise per-launch costs (B=8: 4.10 img/s at 1,
        # 4.24 at 2, 4.30 at 4, same peak memory); ``max_resident`` = how many micro-batches keep their activations for the backward
        # (26.5 GB per image at HR 1792^2), the others are recomputed there.  None = as many as the free HBM allows.
        self.micro_batch = 4
        self.max_resident = None
        self.lean_saves = None        # None: lean KBPN saves (KBPN.forward) only when that keeps more micro-batches resident; True / False force it
        self.dropout_enabled = True
        self.dropout_masks = None     # tests may inject {name: [B,C] fp32} keep-masks
        # Precision plan of the detector's FORWARD pass (PSPNet / PSPNet_BlurSkip / HRNet-OCR):
        #   "split"  (default: the mode that meets north_star's 1e-3 on the segmentation output) activations and weights as fp16 hi + lo
        #            pairs (~22 mantissa bits), three MFMA passes per conv into one fp32 accumulator -- the detector matches the fp32
        #            reference to ~1e-4 on identical inputs where plain fp16 storage gives 3e-3 (contractive weights) .. 4e-2 (random
        #            weights, a ~100x amplifier of every layer's rounding: DESIGN.md section 2).  Costs ~3x the detector's forward MFMA
        #            time (~11 % of a config-2 step) and 2x its activation memory; the backward is unchanged (fp16 hi planes).
        #   "fp16"   fp16 activation storage, one MFMA pass: the throughput configuration, reported beside the headline by bench.py.
        self.detector_precision = "split"
        # Per-layer refinement of the split mode (the precision PLAN): an ordered list of (regex on the conv's parameter name, K blocks)
        # -- 3 = full hi + lo product, 2 = hi + lo activation against the fp16 weight, 1 = plain fp16 operands (Conv.fwd_blocks); the
        # first match wins, unmatched layers run 3 blocks.  Swept in round 4 on the reference's own SR images (scripts/study_split_plan.py,
        # profiles/r04_split_plan_study.json): every group of the trunk needs all three blocks (layer 3 alone at two blocks: 2.2e-3 on the
        # map); only the tail (up_1 .. final) and the PSP module stay under 1e-3 with two, each eating a third to a half of the margin for
        # < 2 % of the step, so the default plan is None -- EXCEPT, since round 5, the four blur_skip conv0 / conv1 pairs of PSPNet_BlurSkip
        # (64 -> 505 -> 64 at full HR resolution, 48 % of a config-5 step): with their fp16 weights rounded tap-sum-preservingly and the mean
        # compensation on (engine.Conv._wq / _dc_bias: what a layer with fewer than three blocks gets) the two-block plan measures 2.36e-4 on
        # the map against 2.10e-4 with three (nearest rounding: 4.99e-4; gradients median 1.4e-3 vs 9e-4, bound 3e-2) on the reference's own
        # SR image -- inside 1e-3 with 4x margin -- for a third of those layers' forward MFMA time.  The same rounding does NOT rescue the
        # BatchNorm'd trunk layers, which stay at three.  Round 6 (review item 6, scripts/study_split_plan.py --combos on tap-sum-rounded weights,
        # profiles/r06_split_plan_combos.json): the decoder TAIL of PSPNet -- up_1, up_2, up_3, final: 35 ms of split forward convolutions per
        # config-2 step -- at two blocks together measures 4.15e-4 on the reference's own SR image against 7.9e-5 with three (up_2 1.7e-4, up_3
        # 1.1e-4, final 3.0e-4, up_1 3.6e-4 alone; gradients median 1.14e-2 vs 1.03e-2, max 2.24e-2, bound 3e-2): inside the review's "sum
        # <= 6e-4" criterion, and adopted; the PSP module on top of it (6.7e-4) is not.
        # ``detector_hp_dgrad``: dgrads against [w_hi | w_lo] (two K blocks).  The same sweep shows it buys nothing -- every detector
        # gradient tensor and dLoss/dSR agree with the reference equally well without it (PSPNet median 1.15e-2 vs 1.12e-2, HRNet-OCR
        # 1.65e-2 both, BlurSkip 9.4e-4 vs 9.1e-4: the error is the ReLU-gate flips of the forward, not the weights' rounding) -- so it is off.
        # BlurSkip's two conv blocks between the SFT layers (blur_skip.1 / .3: 64 -> 64 at full resolution, BatchNorm'd) join the plan in round 6:
        # 2.36e-4 -> 3.79e-4 on the map, gradients unchanged (median 1.5e-3, max 3.6e-3; scripts/study_split_plan.py --combos,
        # profiles/r06_split_plan_combos.json); the decoder tail on top of that measures 8.8e-4 on this detector and stays at three.
        self.detector_plan = [(r"blur_skip\.[02]\.conv_(scale|shift)\.[01]\.|blur_skip\.[13]\.layer", 2)] if self.blur_skip else None
        if self.seg_model_name == "PSPNet" and __import__("os").environ.get("CSBSR_DEC_PLAN", "1") != "0":
            self.detector_plan = [(r"\.(up_[123]|final)\.", 2)]
        if __import__("os").environ.get("CSBSR_BS_PLAN") == "0":        # (A/B hook: three blocks everywhere)
            self.detector_plan = None
        self.detector_hp_dgrad = __import__("os").environ.get("CSBSR_HP_DGRAD") == "1"      # (A/B hook; default off)

    # ---- naming: state_dict keys are the reference's dotted names
    def _named_full(self):
        for grp in ("segmentation_model", "sr_model"):
            for local, t in getattr(self, grp).named_local():
                yield f"{grp}.{local}", t

    def state_dict(self, *a, **kw):
        from collections import OrderedDict
        return OrderedDict((k, v.detach() if isinstance(v, nn.Parameter) else v) for k, v in self._named_full())

    def load_state_dict(self, sd, strict=True):
        own = dict(self._named_full())
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        with torch.no_grad():
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(v)
        if self._rt is not None:
            self._invalidate()
            self._rt["eng"].new_step([own[k] for k in sd if k in own])      # slope probes of the weights just replaced are stale
        return missing, unexpected

    # ---- runtime (engine + layer objects) is built lazily on the device
    def _runtime(self):
        if self._rt is None:
            if not torch.cuda.is_available():
                raise L.CsbsrHipError("csbsr_amd needs an MI355X (no CPU fallback)")
            self.to(self._device)
            eng = Engine(self._device)
            P = {k: (v.data if isinstance(v, nn.Parameter) else v) for k, v in self._named_full()}
            self._rt = {"eng": eng, "P": P, "kbpn": KBPN(eng, P, self.pc), "psp": HRNetOCR(eng, P) if self.seg_model_name == "HRNet_OCR"
                        else PSPNet(eng, P, blur_dim=self.pc.ksize_out ** 2 if self.blur_skip else None)}
            self._make_grad_buckets()
        if self.detector_precision not in ("fp16", "split"):
            raise ValueError(f"detector_precision must be 'fp16' or 'split', got {self.detector_precision!r}")
        split = self.detector_precision == "split"
        self._rt["psp"].split = split
        import re
        plan = [(re.compile(pat), int(nb)) for pat, nb in (self.detector_plan or ())]
        for c in self._rt["psp"].all_convs():      # the split mode's dgrads run against fp16 hi + lo weight pairs (Conv.bwd_input)
            c.hp_dgrad = split and bool(self.detector_hp_dgrad)
            c.fwd_blocks = next((nb for pat, nb in plan if pat.search(c.name)), 3)
            c.dc_comp = c.fwd_blocks < 3          # a layer that keeps its weights' fp16 rounding gets the mean compensation (Conv._dc_bias)
        return self._rt

    def _bucket_of(self, name):
        """gradient bucket of a parameter: the order buckets complete in the explicit backward (segmentation net, then KBPN stages
        S .. 1 -- output_conv goes with stage S --, then the predictor + VGG head)."""
        if name.startswith("segmentation_model"):
            return "seg"
        S = self.pc.num_stages
        if name.startswith("sr_model.back_projection_stages."):
            return f"kbpn.{int(name.split('.')[2]) + 1}"
        if name.startswith("sr_model.output_conv"):
            return f"kbpn.{S}"
        return "kbpn.0"

    def _make_grad_buckets(self):
        """One flat fp32 buffer per bucket; every trainable parameter's accumulator (``gacc``, written by the kernels) is a view of
        it, so a bucket is zeroed / all-reduced / scaled as ONE tensor."""
        rt = self._rt
        sizes = {}
        for k, v in self._named_full():
            if isinstance(v, nn.Parameter) and v.requires_grad:
                sizes[self._bucket_of(k)] = sizes.get(self._bucket_of(k), 0) + v.numel()
        flats = {b: torch.zeros(n, dtype=torch.float32, device=self._device) for b, n in sizes.items()}
        off = {b: 0 for b in sizes}
        for k, v in self._named_full():
            if isinstance(v, nn.Parameter) and v.requires_grad:
                b, t = self._bucket_of(k), rt["P"][k]
                t.gacc = flats[b][off[b]:off[b] + v.numel()].view(t.shape)
                t.gacc_touched = False
                off[b] += v.numel()
        rt["flat"] = flats

    def _invalidate(self):
        rt = self._rt
        rt["kbpn"].invalidate()
        rt["psp"].invalidate()
        rt["eng"].new_step()            # (master weights may have been stepped: every PReLU slope probe may issue one new asynchronous read)

    # ---- shared forward pieces
    def _mount(self, t):
        return None if t is None else t.to(self._device, torch.float32).contiguous()

    def _instnorm_stats(self, sr32):
        eng = self._rt["eng"]
        B, Cc, H, W = sr32.shape
        red = eng.f32(B * Cc, 2)
        L.call("csbsr_plane_reduce", _ptr(sr32), None, B * Cc, H * W, _ptr(red), eng.stream)
        mean = red[:, 0] / (H * W)
        var = (red[:, 1] / (H * W) - mean * mean).clamp_min(0)
        return mean.contiguous(), torch.rsqrt(var + 1e-5).contiguous()

    def _norm_sr(self, sr32):
        """build_model.py:125-141 -> FM fp16 NHWC8 for the segmentation net, plus what the backward needs."""
        eng = self._rt["eng"]
        B = sr32.shape[0]
        if self.norm_method == "instance":
            mean, invstd = self._instnorm_stats(sr32)
        elif self.norm_method == "all":
            mean = torch.tensor(self.pc.mean, device=self._device).repeat(B).contiguous()
            invstd = (1.0 / torch.tensor(self.pc.std, device=self._device)).repeat(B).contiguous()
        else:
            mean = invstd = None
        return eng.nchw32_to_fm(sr32, mean=mean, invstd=invstd, split=self.detector_precision == "split"), mean, invstd


class _JointFn(torch.autograd.Function):
    """(segment_loss[B], sr_loss[B]) = f(parameters); backward = the HIP backward pass."""

    @staticmethod
    def forward(ctx, model, st, seg_loss, sr_loss, *params):
        ctx.model, ctx.st = model, st          # the saved state travels with THIS graph: a later forward cannot clobber it
        ctx.set_materialize_grads(False)       # an unused loss vector arrives as None, not as a zero tensor to be scanned
        return seg_loss.clone(), sr_loss.clone()

    @staticmethod
    def backward(ctx, dseg, dsr):
        st, ctx.st = ctx.st, None
        if st is None:
            raise RuntimeError("csbsr_amd: backward called twice on the same forward (activations are freed by the first backward)")
        grads = ctx.model._hip_backward(st, dseg, dsr)
        return (None, None, None, None) + tuple(grads)


class JointModelWithLoss(_JointBase):
    def __init__(self, cfg, num_train_ds, resume_iter, sr_transforms=None, antialias=True, device="cuda:0", seed=None):
        super().__init__(cfg, antialias, device, seed)
        if cfg.SOLVER.SEG_LOSS_FUNC != "BoundaryCombo" or cfg.SOLVER.SR_LOSS_FUNC != "KBPN":
            raise NotImplementedError("csbsr_amd builds SEG_LOSS_FUNC=BoundaryCombo with SR_LOSS_FUNC=KBPN")
        if cfg.SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SS_AMP != 0 or cfg.SOLVER.CRACK_ORIENTED_WEIGHT4SR_AMP != 0 or cfg.SOLVER.INTERM_SSLOSSWEGHT4SR:
            raise NotImplementedError("oriented loss weights other than SEG_FAIL_ORIENTED_WEIGHT4SR are not built")
        seg_rsm = resume_iter - (cfg.SOLVER.SR_PRETRAIN_ITER[1] - 1) if resume_iter > (cfg.SOLVER.SR_PRETRAIN_ITER[1] - 1) else 0
        per_epoch = num_train_ds // cfg.SOLVER.BATCH_SIZE + 1
        self.ss_loss_fn = BoundaryComboState(per_epoch, seg_rsm, decrease_ratio=cfg.SOLVER.BOUNDARY_DEC_RATIO)
        self.sr_loss_fn = "KBPNLoss"
        self.aux_weight, self.main_weight = cfg.SOLVER.SEG_AUX_LOSS_WEIGHT, cfg.SOLVER.SEG_MAIN_LOSS_WEIGHT
        self.iter_cnt = True
        self.grad_scale = None          # None: chosen per call as 2^round(log2(B*H*W)) (see _hip_backward)
        self.scale_backoff = 0          # log2 reduction of the automatic scale after overflowed steps
        self.overflow_steps = 0
        self.last_step_overflowed = False
        self.last_dsr = self.last_dkvec = None      # set by the backward of forward_from_sr (validation)
        self.reducer = None             # csbsr_amd.parallel.GradBucketReducer when data-parallel

    # ------------------------------------------------------------------ forward
    def forward(self, iter, x, sr_targets=None, segment_targets=None, kernel_targets=None, segment_sdf=None):
        """``segment_sdf`` (optional, not in the reference's signature): the signed distance map of ``segment_targets`` already on the
        device -- csbsr_amd.data.degrade.DeviceDegradation computes it with the batch -- so the loss does not recompute it."""
        rt = self._runtime()
        eng, kbpn, psp, pc = rt["eng"], rt["kbpn"], rt["psp"], self.pc
        kbpn.training_mode, kbpn.pad_dropout = self.training, self.dropout_enabled and self.dropout_masks is None
        self._invalidate()              # master weights may have been stepped by the optimiser
        x, hr, mask, kgt = self._mount(x), self._mount(sr_targets), self._mount(segment_targets), self._mount(kernel_targets)
        B, _, h, w = x.shape
        H, W = h * pc.scale, w * pc.scale
        mb = max(1, min(self.micro_batch, B))
        training = self.training
        keep = training and torch.is_grad_enabled()
        # (the rank agreement inside _auto_resident is a collective: only a forward that will be followed by a backward takes part in it, so
        # a rank-0-only validation pass, an evaluator sharing the model or a no_grad call can never leave the other ranks waiting)
        n_res, lean = (self.max_resident, bool(self.lean_saves)) if self.max_resident is not None else self._auto_resident(B, mb, H, W, agree=keep)
        if self.max_resident is None and keep and n_res == 0 and mb >= B and B >= 2 and not self.blur_skip:
            # the whole batch as ONE micro-batch is all-or-nothing: when it does not fit (RCCL buffers, another tenant, fragmentation)
            # halve the micro-batch so that part of the batch stays resident instead of recomputing every KBPN forward in the backward.
            # n_res is the agreed minimum over the ranks, so every rank takes this branch (and its second agreement) together
            mb2 = (B + 1) // 2
            n2, lean2 = self._auto_resident(B, mb2, H, W, agree=keep)
            if n2 > 0:
                mb, n_res, lean = mb2, n2, lean2
        self._n_res, self._lean, self._mb_used = n_res, lean, mb
        single = mb >= B
        sr32 = eng.f32(B, 3, H, W, zero=False)
        kvec = eng.f32(B, pc.ksize_out ** 2, zero=False)
        saves, self._pad_takes = [], []
        for i, b0 in enumerate(range(0, B, mb)):
            resident = keep and i < n_res and not self.blur_skip     # BlurSkip: KBPN is frozen, no backward through it
            s_, k_ = kbpn.forward(x[b0:b0 + mb], iter, kgt[b0:b0 + mb], save=resident, lean=lean)
            saves.append(kbpn.saved if resident else None)
            self._pad_takes.append(kbpn.pad_taken)        # ZERO_PAD_KERNEL: a recomputed forward replays these (KBPN.forward)
            kbpn.saved = None
            sr32[b0:b0 + mb] = s_
            kvec[b0:b0 + mb] = k_
        return self._detector_and_losses(iter, x, hr, mask, kgt, sr32, kvec, saves, mb, sdf=self._mount(segment_sdf))

    def forward_from_sr(self, iter, sr_preds, kernel_vec, x, sr_targets, segment_targets, kernel_targets):
        """Validation entry point: the detector + loss half of ``forward`` fed a GIVEN SR image [B,3,H,W] and (un-normalised) kernel
        vector [B,kk] instead of KBPN's -- e.g. the reference's own sr_preds from a golden fixture -- so the detector, the losses
        and their backward can be compared with the reference on identical inputs.  ``backward()`` on the returned losses stops at
        the SR image: its gradient (true scale) is left in ``self.last_dsr`` / ``self.last_dkvec``; KBPN parameters get no gradient."""
        rt = self._runtime()
        self._invalidate()
        x, hr, mask, kgt = self._mount(x), self._mount(sr_targets), self._mount(segment_targets), self._mount(kernel_targets)
        sr32, kvec = self._mount(sr_preds), self._mount(kernel_vec).reshape(x.shape[0], -1)
        self._n_res = 0
        return self._detector_and_losses(iter, x, hr, mask, kgt, sr32, kvec, None, x.shape[0])

    @torch.no_grad()
    def kbpn_backward_from(self, iter, x, kernel_targets, dsr, dkvec):
        """Validation entry point: KBPN forward + backward with a GIVEN upstream gradient (dLoss/d sr_preds [B,3,H,W] and
        dLoss/d kernel vector [B,kk], true scale) -- e.g. the reference's own, from a golden fixture.  Returns {state_dict name: grad}."""
        rt = self._runtime()
        eng, kbpn, pc = rt["eng"], rt["kbpn"], self.pc
        kbpn.training_mode, kbpn.pad_dropout = self.training, self.dropout_enabled and self.dropout_masks is None
        self._invalidate()
        x, kgt, dsr, dkvec = self._mount(x), self._mount(kernel_targets), self._mount(dsr), self._mount(dkvec)
        B, _, h, w = x.shape
        gs = float(2 ** round(math.log2(B * h * w * pc.scale * pc.scale)))
        eng.grad_scale = gs
        names = [k for k, v in self._named_full() if isinstance(v, nn.Parameter) and k.startswith("sr_model")]
        for k in names:
            t = rt["P"][k]
            if getattr(t, "gacc", None) is not None:
                t.gacc.zero_()
            t.gacc_touched = False
        mb = max(1, min(self.micro_batch, B))
        for b0 in range(0, B, mb):
            kbpn.forward(x[b0:b0 + mb], iter, kgt[b0:b0 + mb], save=True)
            kbpn.backward((dsr[b0:b0 + mb] * gs).contiguous(), (dkvec[b0:b0 + mb] * gs).contiguous())
        eng.join_wgrad()
        return {k: (rt["P"][k].gacc / gs if getattr(rt["P"][k], "gacc_touched", False) else None) for k in names}

    def _detector_and_losses(self, iter, x, hr, mask, kgt, sr32, kvec, saves, mb, sdf=None):
        rt, pc = self._rt, self.pc
        eng, psp = rt["eng"], rt["psp"]
        B, _, h, w = x.shape
        H, W = h * pc.scale, w * pc.scale
        training = self.training
        xin, mean, invstd = self._norm_sr(sr32)
        drop = psp.make_dropout(B, training, self.dropout_enabled) if self.dropout_masks is None else \
            {k: self.dropout_masks.get(k) for k in psp.drop_keys}
        drop = {k: (None if v is None else v.to(self._device, torch.float32).contiguous()) for k, v in drop.items()}
        seg32, aux32 = psp.forward(xin, drop, training, kvec=kvec if self.blur_skip else None)
        keep = training and torch.is_grad_enabled()
        psp_saved, psp.saved = (psp.saved if keep else None), None      # carried by the autograd node, not by the (shared) layer object
        # ---- losses (forward sums only; gradients are produced in _hip_backward)
        hw = H * W
        if sdf is None:
            sdf = eng.f32(B, 1, H, W, zero=False)
            scratch = eng.f32(3 * B * hw + 2 * B, zero=False)
            L.call("csbsr_sdf", _ptr(mask), _ptr(sdf), _ptr(scratch), B, H, W, eng.stream)
            del scratch
        alpha = float(self.ss_loss_fn.alpha)
        seg_loss = eng.f32(B)
        sums_m, sums_a = eng.f32(B, 8), eng.f32(B, 8)
        pw, lw = pc.bce_w, pc.wbd_w
        for p_, sums, wgt in ((seg32, sums_m, self.main_weight), (aux32, sums_a, self.aux_weight)):
            L.call("csbsr_segloss_reduce", _ptr(p_), _ptr(mask), _ptr(sdf), B, hw, _ptr(sums), pw[0], pw[1], eng.stream)
            L.call("csbsr_segloss_finish", _ptr(p_), _ptr(mask), _ptr(sdf), B, hw, _ptr(sums), alpha, pw[0], pw[1], lw[0], lw[1], wgt,
                   None, _ptr(seg_loss), None, 0, eng.stream)
        # KBPNLoss: L1(sr, hr), L1(down(blur(sr)), x), MSE(kernel) * 0
        ksum = kvec.sum(1, keepdim=True)
        vec = (kvec / ksum).contiguous()
        K = pc.ksize_out
        wmap = None
        if iter > pc.oriented_w_iter and pc.oriented_w_iter != -1 and pc.sfo_sr_amp != 0:
            wmap = torch.exp(pc.sfo_sr_amp * (seg32 - mask).abs()).contiguous()      # oriented_weight.py:73-83 (detached)
        wmap_lr = None
        if wmap is not None:
            wmap_lr = eng.f32(B, 1, h, w, zero=False)
            L.call("csbsr_bilinear32_fwd", _ptr(wmap), _ptr(wmap_lr), B, H, W, h, w, 0, eng.stream)
        blurred = eng.f32(B, 3, H, W, zero=False)
        L.call("csbsr_blur_fwd", _ptr(sr32), _ptr(vec), B, 3, H, W, K, 1, None, _ptr(blurred), None, 0, eng.stream)
        lr_pred = eng.f32(B, 3, h, w, zero=False)
        L.call("csbsr_aa_bicubic_down_fwd", _ptr(blurred), _ptr(lr_pred), B * 3, H, W, pc.scale, int(pc.antialias), eng.stream)
        s_hr, s_lr = eng.f32(B), eng.f32(B)
        L.call("csbsr_l1_fwd_bwd", _ptr(sr32), _ptr(hr), _ptr(wmap), B, 3, hw, _ptr(s_hr), 0.0, None, None, 0, eng.stream)
        L.call("csbsr_l1_fwd_bwd", _ptr(lr_pred), _ptr(x), _ptr(wmap_lr), B, 3, h * w, _ptr(s_lr), 0.0, None, None, 0, eng.stream)
        kpred = vec.reshape(B, 1, K, K)
        k_l = ((kpred - kgt) ** 2).mean((1, 2, 3))
        # SOLVER.ONLY_KERNEL_LOSS_FOR_PRETRAIN (sr_loss_functions.py:50-51): during the kernel-module pretraining phase the SR loss IS the
        # kernel MSE (the reference returns the unreduced [B,1,K,K] map there and calc_loss takes its mean: the per-sample mean has the same mean)
        sr_w = (0.0, 0.0, 1.0) if (pc.only_kernel_loss and pc.kernel_pretrain[0] <= iter < pc.kernel_pretrain[1]) else pc.sr_w
        sr_loss = sr_w[0] * s_hr / (3 * hw) + sr_w[1] * s_lr / (3 * h * w) + sr_w[2] * k_l
        del blurred
        if keep:
            st = dict(iter=iter, x=x, hr=hr, mask=mask, kgt=kgt, sr32=sr32, kvec=kvec, ksum=ksum, vec=vec, mean=mean, invstd=invstd,
                      seg32=seg32, aux32=aux32, sdf=sdf, sums_m=sums_m, sums_a=sums_a, alpha=alpha, lr_pred=lr_pred,
                      wmap=wmap, wmap_lr=wmap_lr, saves=saves, mb=mb, B=B, h=h, w=w, psp_saved=psp_saved, n_res=self._n_res, sr_w=sr_w,
                      pad_takes=getattr(self, "_pad_takes", None) if saves is not None else None)
            params = [p for p in self.parameters()]
            seg_loss, sr_loss = _JointFn.apply(self, st, seg_loss, sr_loss, *params)
        return seg_loss, sr_loss, seg32, sr32, kpred

    def _auto_resident(self, B, mb, H, W, agree=True):
        """micro-batches whose KBPN activations fit next to the detector's working set (measured at HR 1792^2: 26.5 GB per image
        of KBPN activations, 6.3 / 9.5 GB per image for PSPNet / HRNet-OCR incl. their backward workspaces -- the split-precision
        detector holds two planes per activation) with 18 GB to spare.  The budget is what is FREE now (driver-reported free memory
        plus what torch's caching allocator holds but does not use), so another tenant of the GPU or a second model in the process
        lowers the residency instead of running the step out of memory; the rest is recomputed in the backward."""
        r = (H * W) / float(1792 * 1792)
        free, _ = torch.cuda.mem_get_info(self._device)
        free += torch.cuda.memory_reserved(self._device) - torch.cuda.memory_allocated(self._device)
        det = (9.5e9 if self.seg_model_name == "HRNet_OCR" else 6.3e9) * r * B
        if self.detector_precision == "split":
            det += 4.9e9 * r * B
        n_mb = (B + mb - 1) // mb

        def fit(per_img):
            imgs = int((free - 18e9 - det) // (per_img * r)) if r > 0 else B
            return max(0, min(n_mb, imgs // mb))
        full, lean = fit(26.5e9), fit(21.2e9)      # lean saves: the kernel predictors' fe_SR chains are rebuilt in the backward (KBPN.forward)
        if agree and self.reducer is not None and self.reducer.active:
            # data-parallel: the ranks must take the SAME schedule (at ~240 of 288 GB the collective library's buffers can tip one rank into
            # recomputing a KBPN forward, and every other rank would wait for it at the all-reduce): the minimum over the ranks, agreed
            # in one tiny collective per problem shape.  EVERY rank must present each training shape (B, micro-batch, H, W) -- the
            # data-parallel contract anyway: equal shards, train.py:105-112 of the reference -- and only training forwards with autograd
            # on take part (``agree``); eval / no_grad forwards keep nothing resident and use the local figures.  The agreement is kept
            # for the life of the model: renewing it when ONE rank's free memory drops later would be a collective only that rank enters
            key = (B, mb, H, W, self.detector_precision)
            agreed = self.__dict__.setdefault("_sched_agreed", {})
            if key not in agreed:
                agreed[key] = self.reducer.agree_min([full, lean])
            full, lean = agreed[key]
        if self.lean_saves is not None:
            return (lean, True) if self.lean_saves else (full, False)
        return (lean, True) if lean > full else (full, False)

    # ------------------------------------------------------------------ backward
    def _hip_backward(self, st, dseg_loss, dsr_loss):
        rt, pc = self._rt, self.pc
        eng, kbpn, psp = rt["eng"], rt["kbpn"], rt["psp"]
        B, h, w = st["B"], st["h"], st["w"]
        H, W = h * pc.scale, w * pc.scale
        hw = H * W
        # loss scale of the fp16 activation gradients: the per-pixel loss gradient is O(1/(B*H*W)); PSPNet keeps that magnitude down
        # to the input, HRNet-OCR grows it ~1e5x towards the stem (measured, reference-style init), so it starts 2^8 lower.
        # ``scale_backoff`` is the dynamic part (GradScaler semantics): a backward that overflowed returns zero gradients for that
        # step and lowers the scale for the following ones.
        gs = self.grad_scale or float(2 ** (round(math.log2(B * hw)) - (8 if self.seg_model_name == "HRNet_OCR" else 0) - self.scale_backoff))
        eng.grad_scale = gs
        pnames = [k for k, v in self._named_full() if isinstance(v, nn.Parameter)]     # == self.parameters() order
        for k in pnames:                    # fresh fp32 accumulators for this backward
            rt["P"][k].gacc_touched = False
        torch._foreach_zero_(list(rt["flat"].values()))      # the accumulators are views of a handful of flat buckets
        dsr32 = eng.f32(B, 3, H, W)
        # which halves of the backward run follows from which loss vector the caller's scalar loss used (autograd hands None for an
        # unused output): a function of the training phase, identical on every rank, and no device read-back
        seg_active = dseg_loss is not None
        psp.saved = st["psp_saved"]
        if seg_active:
            gsc = (dseg_loss.to(torch.float32) * gs).contiguous()
            dseg32, daux32 = eng.f32(B, 1, H, W, zero=False), eng.f32(B, 1, H, W, zero=False)
            pw, lw = pc.bce_w, pc.wbd_w
            for p_, sums, wgt, dp in ((st["seg32"], st["sums_m"], self.main_weight, dseg32), (st["aux32"], st["sums_a"], self.aux_weight, daux32)):
                L.call("csbsr_segloss_finish", _ptr(p_), _ptr(st["mask"]), _ptr(st["sdf"]), B, hw, _ptr(sums), st["alpha"], pw[0], pw[1],
                       lw[0], lw[1], wgt, _ptr(gsc), None, _ptr(dp), 0, eng.stream)
            dxin = psp.backward(dseg32, daux32)
            eng.join_wgrad()                    # the detector's weight gradients (side stream) are complete
            if self.blur_skip:                  # only blur_skip.* trains: no gradient leaves the segmentation net
                return self._finish_backward(pnames, gs, ("segmentation_model",), st)
            if self.reducer is not None:        # segmentation gradients are final: exchange them under the KBPN backward
                self.reducer.launch_flat(rt["flat"].get("seg"))
                st["seg_launched"] = True
            if st["mean"] is not None and self.norm_method == "instance":
                red = eng.f32(B * 3, 2)
                L.call("csbsr_instnorm_bwd", _ptr(dxin.t), dxin.ld, _ptr(st["sr32"]), _ptr(st["mean"]), _ptr(st["invstd"]), _ptr(dsr32), 0,
                       B, 3, hw, _ptr(red), eng.stream)
            else:
                eng.fm_to_nchw32(dxin, dsr32, 3)
                if st["invstd"] is not None:
                    dsr32 *= st["invstd"].reshape(B, 3, 1, 1)
            del dxin, dseg32, daux32
        else:
            psp.saved = None
            if self.blur_skip:
                return self._finish_backward(pnames, gs, (), st)
        # ---- SR loss gradients
        dkvec = eng.f32(B, pc.ksize_out ** 2)
        if dsr_loss is not None:
            g = (dsr_loss.to(torch.float32) * gs)
            K = pc.ksize_out
            sr_w = st["sr_w"]
            g_hr = (g * sr_w[0] / (3 * hw)).contiguous()
            g_lr = (g * sr_w[1] / (3 * h * w)).contiguous()
            L.call("csbsr_l1_fwd_bwd", _ptr(st["sr32"]), _ptr(st["hr"]), _ptr(st["wmap"]), B, 3, hw, None, 1.0, _ptr(g_hr), _ptr(dsr32), 1,
                   eng.stream)
            dlr = eng.f32(B, 3, h, w, zero=False)
            L.call("csbsr_l1_fwd_bwd", _ptr(st["lr_pred"]), _ptr(st["x"]), _ptr(st["wmap_lr"]), B, 3, h * w, None, 1.0, _ptr(g_lr), _ptr(dlr), 0,
                   eng.stream)
            dbl = eng.f32(B, 3, H, W, zero=False)
            L.call("csbsr_aa_bicubic_down_bwd", _ptr(dlr), _ptr(dbl), 0, B * 3, H, W, pc.scale, int(pc.antialias), eng.stream)
            L.call("csbsr_blur_bwd_input", _ptr(dbl), _ptr(st["vec"]), _ptr(dsr32), 1, B, 3, H, W, K, 1, eng.stream)
            dvec = eng.f32(B, K * K)
            L.call("csbsr_blur_bwd_kernel", _ptr(dbl), _ptr(st["sr32"]), _ptr(dvec), B, 3, H, W, K, 1, eng.stream)
            if sr_w[2] != 0:
                dvec += (g * sr_w[2] / (K * K)).reshape(B, 1) * 2 * (st["vec"] - st["kgt"].reshape(B, -1))
            dkvec = (dvec - (dvec * st["vec"]).sum(1, keepdim=True)) / st["ksum"]
            del dbl, dlr
        # ---- KBPN backward (per micro-batch; recompute the forward when it was not kept)
        mb = st["mb"]
        saves = st["saves"]
        if saves is None:        # forward_from_sr: the graph ends at the given SR image
            self.last_dsr, self.last_dkvec = dsr32 / gs, dkvec / gs
            return self._finish_backward(pnames, gs, ("segmentation_model",) if seg_active else (), st)
        order = list(enumerate(range(0, B, mb)))
        # resident micro-batches last-in first-out (frees HBM before the recomputed ones run), then the rest with their forward
        # recomputed here (KBPN has no batch-coupled op: exact)
        sched = [(i, b0, False) for i, b0 in reversed(order) if saves[i] is not None] + [(i, b0, True) for i, b0 in order if i >= st["n_res"]]
        launched = set()

        def stage_done(s):      # last micro-batch only: stage s's parameter gradients are final -> exchange them under the rest
            if self.reducer is not None:
                self.reducer.launch_flat(rt["flat"].get(f"kbpn.{s}"))
                launched.add(f"kbpn.{s}")
        for j, (i, b0, recompute) in enumerate(sched):
            if recompute:
                kbpn.forward(st["x"][b0:b0 + mb], st["iter"], st["kgt"][b0:b0 + mb], save=True,
                             pad_replay=st["pad_takes"][i] if kbpn.zero_pad else None)
            else:
                kbpn.saved, saves[i] = saves[i], None
            kbpn.backward(dsr32[b0:b0 + mb].contiguous(), dkvec[b0:b0 + mb].contiguous(),
                          stage_done=stage_done if j == len(sched) - 1 else None)
        st["kbpn_launched"] = launched
        return self._finish_backward(pnames, gs, ("sr_model",), st)

    def _finish_backward(self, pnames, gs, reduce_groups, st):
        rt = self._rt
        rt["eng"].join_wgrad()
        if self.reducer is not None:
            # whatever this phase's backward wrote and has not been launched yet (every rank takes the same branch: the set of
            # buckets depends on the training phase only)
            for b, flat in rt["flat"].items():
                grp = "segmentation_model" if b == "seg" else "sr_model"
                if grp in reduce_groups and not (b == "seg" and st.get("seg_launched")) and b not in st.get("kbpn_launched", ()):
                    self.reducer.launch_flat(flat)
            self.reducer.finish()
        inv = 1.0 / gs
        # overflow check on the (already all-reduced, so rank-consistent) accumulators: one scalar read back per step
        finite = bool(torch.isfinite(torch.stack(torch._foreach_norm(list(rt["flat"].values()))).sum()))
        if not finite:
            self.overflow_steps += 1
            if self.grad_scale is None:
                self.scale_backoff += 4
            import warnings
            warnings.warn(f"csbsr_amd: fp16 gradient overflow at loss scale {gs:g}; this step is skipped (every gradient is None)"
                          + ("" if self.grad_scale is not None else f", next scale {gs / 16:g}"))
            self.last_step_overflowed = True
            # GradScaler semantics: optimizer.step() must be a no-op for this step.  Gradients of None make torch optimisers skip the
            # parameter entirely (no moment decay, no step count, no weight move); zeros would still move Adam's weights by its momentum.
            return [None] * len(pnames)
        self.last_step_overflowed = False
        # parameters no kernel touched (frozen phase / unused) keep grad None; the others leave as accumulator x 1 / scale -- one multi-tensor
        # launch set instead of one launch per parameter (290 with PSPNet, 1109 with HRNet-OCR; same fp32 products)
        out = [None] * len(pnames)
        idx = [i for i, k in enumerate(pnames) if getattr(rt["P"][k], "gacc_touched", False)]
        if idx:
            for i, g in zip(idx, torch._foreach_mul([rt["P"][pnames[i]].gacc for i in idx], inv)):
                out[i] = g
        return out


class JointModel(_JointBase):
    """Inference counterpart (build_model.py:441-500): SR clipped to [0,1] before segmentation, kernel normalised."""

    def __init__(self, cfg, antialias=True, device="cuda:0", seed=None):
        super().__init__(cfg, antialias, device, seed)
        self.ksize = cfg.BLUR.KERNEL_SIZE_OUTPUT

    @torch.no_grad()
    def forward(self, x, damy_kernel, sr_targets=None):
        rt = self._runtime()
        eng, kbpn, psp = rt["eng"], rt["kbpn"], rt["psp"]
        kbpn.training_mode = self.training
        self._invalidate()
        x, kgt = self._mount(x), self._mount(damy_kernel)
        B = x.shape[0]
        sr32, kvec = kbpn.forward(x, -1, kgt, save=False)
        sr32.clamp_(0, 1)
        xin, _, _ = self._norm_sr(sr32)
        seg32, _ = psp.forward(xin, {k: None for k in psp.drop_keys}, training=self.training,
                               kvec=kvec if self.blur_skip else None)
        psp.saved = None
        kvec = kvec / kvec.sum(1, keepdim=True)
        return sr32, seg32, kvec.reshape(B, 1, self.ksize, self.ksize)
