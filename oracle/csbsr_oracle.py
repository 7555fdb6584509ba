"""ORACLE -- test infrastructure only.  Never imported by the product (csbsr_amd/), only by tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.

A CPU restatement, in plain fp32 torch tensor ops, of the reference's joint blind-SR + segmentation
training hot path ``JointModelWithLoss.forward`` (+ autograd backward):

    /root/reference/model/modeling/build_model.py:370-416      orchestration
    /root/reference/model/modeling/kbpn.py:84-116, 172-189     KBPN + back-projection stages
    /root/reference/model/modeling/pspnet_pytorch/pspnet.py:95-123, extractors.py:150-161
    /root/reference/model/utils/{sr_loss_functions,loss_functions,boundary_loss}.py

The arithmetic of the reference lives in third-party torch ops (F.conv2d, F.conv_transpose2d,
F.batch_norm, F.interpolate, ... -- torch is unpinned in requirement.txt; this container's
torch 2.10 is the arbiter) plus scipy.ndimage.distance_transform_edt and
skimage.segmentation.find_boundaries (absent here; its published rule is restated below).

Pinning: the reference holds NO tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, imported in the build container by
tests/golden/make_golden.py (fixtures committed under tests/golden/*.npz) and checked by
tests/test_oracle_golden.py.

Parameters are passed as a flat ``{state_dict name: tensor}`` dict using the reference's key names.
Nothing from the reference is imported or copied.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- config


class PathCfg:
    """The cfg keys the path consumes (reference: model/config/defaults.py:17-97 merged with
    config/config_csbsr_pspnet.yaml)."""

    def __init__(self, **kw):
        self.scale = 4                       # MODEL.SCALE_FACTOR
        self.num_stages = 4                  # MODEL.NUM_STAGES
        self.ksize = 7                       # BLUR.KERNEL_SIZE (estimated kernel side)
        self.ksize_out = 21                  # BLUR.KERNEL_SIZE_OUTPUT
        self.sr_pretrain = (1, 10001)        # SOLVER.SR_SR_MODULE_PRETRAIN_ITER
        self.kernel_pretrain = (10001, 20001)  # SOLVER.SR_KERNEL_MODULE_PRETRAIN_ITER
        self.joint_pretrain = (1, 30001)     # SOLVER.SR_PRETRAIN_ITER
        self.norm_sr = "instance"            # SOLVER.NORM_SR_OUTPUT
        self.mean = (0.4741, 0.4937, 0.5048)
        self.std = (0.1621, 0.1532, 0.1523)
        self.sr_w = (0.4, 0.4, 0.0)          # SOLVER.SR_LOSS_FUNC_SR_WEIGHT[:3]  (defaults.py:72)
        self.bce_w = (1.0, 1.0)              # SOLVER.BCELOSS_WEIGHT (yaml)
        self.wbd_w = (1.0, 1.0)              # SOLVER.WB_AND_D_WEIGHT
        self.aux_w = 0.4                     # SOLVER.SEG_AUX_LOSS_WEIGHT
        self.main_w = 1.0                    # SOLVER.SEG_MAIN_LOSS_WEIGHT
        self.beta = 0.3                      # SOLVER.TASK_LOSS_WEIGHT
        self.antialias = True                # torchvision>=0.17 Resize default on tensors
        self.oriented_w_iter = -1            # SOLVER.ORIENTED_WEIGHT_ITER
        self.sfo_sr_amp = 0.0                # SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP
        self.detector = "PSPNet"             # MODEL.DETECTOR_TYPE: "PSPNet" | "PSPNet_BlurSkip" | "HRNet_OCR"
        self.pixel_shuffle = False           # MODEL.SR_PIXEL_SHUFFLE
        self.residual_learning = True        # MODEL.SR_RESIDUAL_LEARNING (kbpn.py:32,112-116)
        self.only_kernel_loss = False        # SOLVER.ONLY_KERNEL_LOSS_FOR_PRETRAIN (sr_loss_functions.py:32,50-51)
        self.kernel_sft = True               # MODEL.KBPN_KERNEL_SFT (kbpn.py:165,169-171,190)
        self.lr_error = False                # MODEL.SUM_LR_ERROR_POS == 'LR' (kbpn.py:166,174-187,369-374,404-409)
        self.zero_pad_kernel = False         # MODEL.ZERO_PAD_KERNEL (kbpn.py:543-554,583-596)
        self.__dict__.update(kw)

    @property
    def conv_kspd(self):                     # kbpn.py:22-25
        return {2: (6, 2, 2), 4: (8, 4, 2), 8: (12, 8, 2)}[self.scale]


# ----------------------------------------------------------------------------- small ops

def prelu(x, a):
    return F.prelu(x, a)


def conv_block(P, pre, x, stride=1, padding=1, dilation=1, act=None, slope=0.01):
    """ConvBlock(norm=None): conv (+bias if present) + activation.  kbpn.py:196-270."""
    y = F.conv2d(x, P[pre + ".layer.weight"], P.get(pre + ".layer.bias"), stride, padding, dilation)
    return _act(P, pre, y, act, slope)


def deconv_block(P, pre, x, stride, padding, act=None, pixel_shuffle=False):
    """DeconvBlock, kbpn.py:273-277 -- or, with MODEL.SR_PIXEL_SHUFFLE, ConvAndPixelShuffleBlock, kbpn.py:280-289: conv3x3 to
    cout * s^2 channels, activation, nn.PixelShuffle(s)."""
    if pixel_shuffle:
        y = F.conv2d(x, P[pre + ".layer.weight"], P.get(pre + ".layer.bias"), 1, 1)
        return F.pixel_shuffle(_act(P, pre, y, act, 0.01), stride)
    y = F.conv_transpose2d(x, P[pre + ".layer.weight"], P.get(pre + ".layer.bias"), stride, padding)
    return _act(P, pre, y, act, 0.01)


def _act(P, pre, y, act, slope):
    if act == "prelu":
        return F.prelu(y, P[pre + ".act.weight"])
    if act == "relu":
        return F.relu(y)
    if act == "lrelu":
        return F.leaky_relu(y, slope)
    assert act is None
    return y


def bicubic_up(x, size=None, scale=None):
    """nn.Upsample(mode='bicubic'), align_corners=False, A=-0.75."""
    return F.interpolate(x, size=size, scale_factor=scale, mode="bicubic", align_corners=False)


def kernel_up_normalise(vec, ksize_out):
    """upscale_and_reshape + sum-normalisation.  kbpn.py:335-341 (predictor), 580-602 (IKC: no
    normalisation there)."""
    B, C = vec.shape[:2]
    ks = int(round(math.sqrt(C)))
    k = vec.reshape(B, 1, ks, ks)
    if ks != ksize_out:
        k = bicubic_up(k, size=(ksize_out, ksize_out))
    return k


def blur_down(sr, vec, ks, stride):
    """Per-sample depthwise cross-correlation with the (B, ks*ks) kernel.  kbpn.py:394-402 (stride =
    scale) and sr_loss_functions.py:89-94 (stride 1).  Grouped over batch*channels == the loop."""
    B, C, H, W = sr.shape
    w = vec.reshape(B, 1, ks, ks).repeat_interleave(C, dim=0)            # (B*C,1,ks,ks)
    y = F.conv2d(sr.reshape(1, B * C, H, W), w, stride=stride, padding=(ks - 1) // 2, groups=B * C)
    return y.reshape(B, C, y.shape[-2], y.shape[-1])


# ----------------------------------------------------------------------------- KBPN blocks

def kbpn_feat(P, x):
    """VGG16 head convs 0,2,5,7 (state_dict indices 0,2,4,6) + ReLU.  kbpn.py:42-44."""
    f = x
    for i in (0, 2, 4, 6):
        f = F.relu(F.conv2d(f, P[f"sr_model.feat.{i}.weight"], P[f"sr_model.feat.{i}.bias"], 1, 1))
    return f


def predictor_with_gap(P, f, cfg):
    """kbpn.py:292-341 -> normalised kernel vector (B, ksize_out**2)."""
    pre = "sr_model.predictor.feat_ext"
    z = f
    for i in range(3):
        z = conv_block(P, f"{pre}.{i}", z, act="prelu")
    vec = z.mean(dim=(2, 3), keepdim=True)                               # GAP
    ks = int(round(math.sqrt(vec.shape[1])))
    if ks != cfg.ksize_out:
        k = kernel_up_normalise(vec, cfg.ksize_out)
        k = k / k.sum(dim=(2, 3), keepdim=True)
        return k.reshape(k.shape[0], -1)
    v = vec.reshape(vec.shape[0], -1)
    return v / v.sum(dim=1, keepdim=True)


def up_block(P, pre, x, cfg):
    """kbpn.py:450-469."""
    k, s, p = cfg.conv_kspd
    x = conv_block(P, pre + ".conv", x, 1, 0, act="prelu")
    h0 = deconv_block(P, pre + ".up_conv1", x, s, p, act="prelu", pixel_shuffle=cfg.pixel_shuffle)
    l0 = conv_block(P, pre + ".up_conv2", h0, s, p, act="prelu")
    h1 = deconv_block(P, pre + ".up_conv3", l0 - x, s, p, act="prelu", pixel_shuffle=cfg.pixel_shuffle)
    return h1 + h0


def down_block(P, pre, x, cfg):
    """kbpn.py:472-489."""
    k, s, p = cfg.conv_kspd
    x = conv_block(P, pre + ".conv", x, 1, 0, act="prelu")
    l0 = conv_block(P, pre + ".down_conv1", x, s, p, act="prelu")
    h0 = deconv_block(P, pre + ".down_conv2", l0, s, p, act="prelu", pixel_shuffle=cfg.pixel_shuffle)
    l1 = conv_block(P, pre + ".down_conv3", h0 - x, s, p, act="prelu")
    return l1 + l0


def sft_layer(P, pre, feats, kvec):
    """kbpn.py:493-518; ``conditions`` is the kernel vector expanded over the LR grid."""
    B, _, H, W = feats.shape
    cond = kvec.reshape(B, -1, 1, 1).expand(B, kvec.shape[1], H, W)
    cat = torch.cat((feats, cond), 1)

    def c(n, x):
        return F.conv2d(x, P[f"{pre}.{n}.weight"], P[f"{pre}.{n}.bias"], 1, 1)
    scale = torch.sigmoid(c("SFT_scale_conv1", F.leaky_relu(c("SFT_scale_conv0", cat), 0.1)))
    shift = c("SFT_shift_conv1", F.leaky_relu(c("SFT_shift_conv0", cat), 0.1))
    return feats * scale + shift


def kernel_predictor_ikc(P, pre, sr_t, kvec, cfg):
    """KernelPredictorLikeIKC.forward, kbpn.py:562-578.  Returns the updated kernel vector
    (pre_kernel + delta); the reference carries it as an expanded (B,441,h,w) map."""
    B, _, H, W = sr_t.shape
    x = conv_block(P, pre + ".fe_SR.0", sr_t, act="relu")
    x = conv_block(P, pre + ".fe_SR.1", x, 1, 0, act="lrelu")
    for i in (2, 3, 4):
        x = conv_block(P, f"{pre}.fe_SR.{i}", x, act="lrelu")
    fh = kvec.reshape(B, -1, 1, 1).expand(B, kvec.shape[1], H, W)
    fh = conv_block(P, pre + ".fe_kernel.0", fh, act="lrelu")
    fh = conv_block(P, pre + ".fe_kernel.1", fh, act="lrelu")
    y = torch.cat((x, fh), 1)
    y = conv_block(P, pre + ".fe_cat.0", y, 1, 0, act="lrelu")
    y = conv_block(P, pre + ".fe_cat.1", y, act="lrelu")
    y = conv_block(P, pre + ".fe_cat.2", y, act=None)
    delta = y.mean(dim=(2, 3), keepdim=True)
    ks = int(round(math.sqrt(delta.shape[1])))
    if ks != cfg.ksize_out:
        up = kernel_up_normalise(delta, cfg.ksize_out)
        if cfg.zero_pad_kernel:
            # kbpn.py:583-596: a small MLP on the 7x7 update decides per sample (hard threshold on .item(): no gradient reaches it) between
            # bicubic upsampling and centred zero padding.  Its nn.Dropout(0.2) layers are taken as identity here (their expectation; the
            # fixtures are generated with them disabled) -- in the reference's train mode they make the decision itself random.
            v = delta.reshape(B, -1)
            pd = pre + ".pad_descriminator"
            h = F.relu(F.linear(v, P[pd + ".0.weight"], P[pd + ".0.bias"]))
            h = F.relu(F.linear(h, P[pd + ".3.weight"], P[pd + ".3.bias"]))
            pr = torch.sigmoid(F.linear(h, P[pd + ".6.weight"], P[pd + ".6.bias"])).detach().reshape(B)
            padn = (cfg.ksize_out - ks) // 2
            zp = F.pad(delta.reshape(B, 1, ks, ks), (padn, padn, padn, padn))
            up = torch.where((pr >= 0.5).reshape(B, 1, 1, 1), up, zp)
        delta = up
    return kvec + delta.reshape(B, -1)


def k_block(P, pre, concat_h, h, x_lr, kvec, it, cfg):
    """KBlock.forward (SUM_LR_ERROR_POS='HR'), kbpn.py:380-409."""
    ksz, s, p = cfg.conv_kspd
    sr_t = conv_block(P, pre + ".sr_reconst", concat_h, act=None)
    if not (cfg.sr_pretrain[0] <= it < cfg.sr_pretrain[1]):
        kvec = kernel_predictor_ikc(P, pre + ".kernel_predictor", sr_t, kvec, cfg)
    vec = kvec / kvec.sum(dim=1, keepdim=True)
    pseudo_lr = blur_down(sr_t, vec, cfg.ksize_out, cfg.scale)
    err = pseudo_lr - x_lr
    if cfg.lr_error:             # kbpn.py:407-409: h leaves unchanged, the error enters the next stage's LR features
        return h, vec, sr_t, conv_block(P, pre + ".conv", err, act=None)
    e_h = deconv_block(P, pre + ".up_conv1", err, s, p, act="prelu", pixel_shuffle=cfg.pixel_shuffle)
    return h + e_h, vec, sr_t, None


def kbpn_forward(P, x, it, kernel_gt, cfg, taps=None):
    """KBPN.forward, kbpn.py:84-116.  Returns (sr, kernel vector (B, ksize_out**2))."""
    f0 = kbpn_feat(P, x)
    if cfg.sr_pretrain[0] <= it < cfg.sr_pretrain[1]:
        kvec = kernel_gt.reshape(kernel_gt.shape[0], -1)
    else:
        kvec = predictor_with_gap(P, f0, cfg)
    if taps is not None:
        taps["init_f"], taps["init_kernel"] = f0, kvec
    low, concat_h, concat_l = f0, None, None
    S = cfg.num_stages
    for s in range(1, S + 1):
        pre = f"sr_model.back_projection_stages.{s - 1}"
        h = up_block(P, pre + ".up", low, cfg)
        pre_cat = h if concat_h is None else torch.cat((concat_h, h), 1)
        h, kvec, sr_t, err_feat = k_block(P, pre + ".kb", pre_cat, h, x, kvec, it, cfg)
        concat_h = h if concat_h is None else torch.cat((concat_h, h), 1)
        if taps is not None:
            taps[f"s{s}.h"], taps[f"s{s}.kvec"], taps[f"s{s}.sr_t"] = h, kvec, sr_t
        if s < S:
            low = down_block(P, pre + ".down", concat_h, cfg)
            if cfg.lr_error:
                low = low + err_feat                                       # kbpn.py:184-185
            concat_l = low if concat_l is None else torch.cat((concat_l, low), 1)
            low = sft_layer(P, pre + ".sft", concat_l, kvec) if cfg.kernel_sft else concat_l      # kbpn.py:190
            if taps is not None:
                taps[f"s{s}.low"] = low
    sr = conv_block(P, "sr_model.output_conv", concat_h, act=None)
    if cfg.residual_learning:                                             # kbpn.py:112-116
        sr = sr + bicubic_up(x, scale=cfg.scale)
    return sr, kvec


# ----------------------------------------------------------------------------- PSPNet

class BNState:
    """Train-mode BatchNorm bookkeeping: collects updated running stats."""

    def __init__(self, P, training=True, momentum=0.1):
        self.P, self.training, self.new, self.momentum = P, training, {}, momentum

    def __call__(self, pre, x):
        P = self.P
        rm, rv = P[pre + ".running_mean"].clone(), P[pre + ".running_var"].clone()
        y = F.batch_norm(x, rm, rv, P[pre + ".weight"], P[pre + ".bias"], self.training, self.momentum, 1e-5)
        if self.training:
            self.new[pre + ".running_mean"], self.new[pre + ".running_var"] = rm, rv
            self.new[pre + ".num_batches_tracked"] = P[pre + ".num_batches_tracked"] + 1
        return y


def _drop(x, masks, name):
    """Dropout2d with an externally supplied (B,C) keep-mask already scaled by 1/(1-p)."""
    if masks is None or masks.get(name) is None:
        return x
    return x * masks[name].reshape(x.shape[0], x.shape[1], 1, 1)


def basic_block(P, bn, pre, x, stride, dilation, has_down):
    """extractors.py:41-70."""
    out = F.conv2d(x, P[pre + ".conv1.weight"], None, stride, dilation, dilation)
    out = F.relu(bn(pre + ".bn1", out))
    out = F.conv2d(out, P[pre + ".conv2.weight"], None, 1, dilation, dilation)
    out = bn(pre + ".bn2", out)
    res = x
    if has_down:
        res = bn(pre + ".downsample.1", F.conv2d(x, P[pre + ".downsample.0.weight"], None, stride))
    return F.relu(out + res)


RESNET34_LAYERS = ((64, 3, 1, 1), (128, 4, 2, 1), (256, 6, 1, 2), (512, 3, 1, 4))  # planes, blocks, stride, dil


def resnet34_dilated(P, bn, x):
    """extractors.py:112-161: block 0 of every layer gets dilation 1 (``_make_layer`` does not pass
    it), later blocks get the layer dilation."""
    pre = "segmentation_model.feats"
    x = F.relu(bn(pre + ".bn1", F.conv2d(x, P[pre + ".conv1.weight"], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    inpl, x3 = 64, None
    for li, (planes, blocks, stride, dil) in enumerate(RESNET34_LAYERS, 1):
        for b in range(blocks):
            first = b == 0
            has_down = first and (stride != 1 or inpl != planes)
            x = basic_block(P, bn, f"{pre}.layer{li}.{b}", x, stride if first else 1, 1 if first else dil, has_down)
        inpl = planes
        if li == 3:
            x3 = x
    return x, x3


def psp_module(P, f):
    """pspnet.py:23-41."""
    h, w = f.shape[2:]
    pri = []
    for i, size in enumerate((1, 2, 3, 6)):
        p = F.adaptive_avg_pool2d(f, (size, size))
        p = F.conv2d(p, P[f"segmentation_model.psp.stages.{i}.1.weight"])
        pri.append(F.interpolate(p, size=(h, w), mode="bilinear", align_corners=False))
    pri.append(f)
    y = F.conv2d(torch.cat(pri, 1), P["segmentation_model.psp.bottleneck.weight"], P["segmentation_model.psp.bottleneck.bias"])
    return F.relu(y)


def psp_upsample(P, bn, pre, x):
    """pspnet.py:44-57."""
    h, w = 2 * x.shape[2], 2 * x.shape[3]
    p = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)
    p = F.conv2d(p, P[pre + ".conv.0.weight"], P[pre + ".conv.0.bias"], 1, 1)
    return F.prelu(bn(pre + ".conv.1", p), P[pre + ".conv.2.weight"])


def sft_like_block(P, pre, feats, kvec):
    """SFTLikeBlock.forward, blocks.py:105-120: cat(features, expanded kernel code) -> scale / shift branches."""
    B, _, H, W = feats.shape
    cond = kvec.reshape(B, -1, 1, 1).expand(B, kvec.shape[1], H, W)
    cat = torch.cat((feats, cond), 1)

    def cb(name, x, act):
        y = F.conv2d(x, P[f"{pre}.{name}.layer.weight"], P[f"{pre}.{name}.layer.bias"], 1, 1)
        if act == "prelu":
            return F.prelu(y, P[f"{pre}.{name}.act.weight"])
        return torch.sigmoid(y) if act == "sigmoid" else y
    scale = cb("conv_scale.1", cb("conv_scale.0", cat, "prelu"), "sigmoid")
    shift = cb("conv_shift.1", cb("conv_shift.0", cat, "prelu"), None)
    return feats * scale + shift


def pspnet_forward(P, x, bn, drop=None, kvec=None):
    """PSPNet.forward, pspnet.py:95-123, and PSPNet_BlurSkip.forward, pspnet.py:173-207 when ``kvec`` (the GAP of the kernel
    prediction, [B, K*K]) is given -> (main prob map, aux prob map)."""
    h, w = x.shape[2:]
    f, x3 = resnet34_dilated(P, bn, x)
    p = _drop(psp_module(P, f), drop, "drop_1")
    p = _drop(psp_upsample(P, bn, "segmentation_model.up_1", p), drop, "drop_2a")
    p = _drop(psp_upsample(P, bn, "segmentation_model.up_2", p), drop, "drop_2b")
    p = _drop(psp_upsample(P, bn, "segmentation_model.up_3", p), drop, "drop_2c")
    if kvec is not None:
        q = p
        for i in range(4):
            pre = f"segmentation_model.blur_skip.{i}"
            if i % 2 == 0:
                q = sft_like_block(P, pre, q, kvec)
            else:       # blocks.ConvBlock(64, 64): conv (no bias) + BatchNorm2d + ReLU
                q = F.relu(bn(pre + ".norm", F.conv2d(q, P[pre + ".layer.weight"], None, 1, 1)))
        p = p + q
    a = F.conv2d(x3, P["segmentation_model.aux.0.weight"], None, 1, 1)
    a = _drop(F.relu(bn("segmentation_model.aux.1", a)), drop, "aux_drop")
    a = torch.sigmoid(F.conv2d(a, P["segmentation_model.aux.4.weight"], P["segmentation_model.aux.4.bias"]))
    a = F.interpolate(a, size=(h, w), mode="bilinear", align_corners=True)
    main = torch.sigmoid(F.conv2d(p, P["segmentation_model.final.0.weight"], P["segmentation_model.final.0.bias"]))
    return main, a


# ----------------------------------------------------------------------------- losses

# ----------------------------------------------------------------------------- HRNet-W48 + OCR (BASELINE config 4)

HRNET_W48 = (("stage2", 1, (48, 96)), ("stage3", 4, (48, 96, 192)), ("stage4", 3, (48, 96, 192, 384)))   # hrnet_config.py:46-73


def _conv_bn(P, bn, conv, norm, x, stride=1, pad=1, relu=True):
    y = bn(norm, F.conv2d(x, P[conv + ".weight"], P.get(conv + ".bias"), stride, pad))
    return F.relu(y) if relu else y


def hr_basic_block(P, bn, pre, x):
    """hrnet_backbone.py:36-63 (no downsample inside the HRNet branches: in == out channels)."""
    out = _conv_bn(P, bn, pre + ".conv1", pre + ".bn1", x)
    out = _conv_bn(P, bn, pre + ".conv2", pre + ".bn2", out, relu=False)
    return F.relu(out + x)


def hr_bottleneck(P, bn, pre, x):
    """hrnet_backbone.py:66-106."""
    out = _conv_bn(P, bn, pre + ".conv1", pre + ".bn1", x, pad=0)
    out = _conv_bn(P, bn, pre + ".conv2", pre + ".bn2", out)
    out = _conv_bn(P, bn, pre + ".conv3", pre + ".bn3", out, pad=0, relu=False)
    res = x
    if (pre + ".downsample.0.weight") in P:
        res = _conv_bn(P, bn, pre + ".downsample.0", pre + ".downsample.1", x, pad=0, relu=False)
    return F.relu(out + res)


def hr_module(P, bn, pre, xs):
    """HighResolutionModule.forward, hrnet_backbone.py:271-297: 4 BasicBlocks per branch, then every output resolution i sums all
    branches j (1x1 conv + BN + bilinear(align_corners=True) up for j > i, a chain of i-j stride-2 3x3 conv + BN (+ReLU between) for j < i)."""
    nb = len(xs)
    xs = list(xs)
    for i in range(nb):
        for b in range(4):
            xs[i] = hr_basic_block(P, bn, f"{pre}.branches.{i}.{b}", xs[i])
    outs = []
    for i in range(nb):
        y = None
        for j in range(nb):
            fp = f"{pre}.fuse_layers.{i}.{j}"
            if j == i:
                t = xs[j]
            elif j > i:
                t = _conv_bn(P, bn, fp + ".0", fp + ".1", xs[j], pad=0, relu=False)
                t = F.interpolate(t, size=xs[i].shape[-2:], mode="bilinear", align_corners=True)
            else:
                t = xs[j]
                for k in range(i - j):
                    t = _conv_bn(P, bn, f"{fp}.{k}.0", f"{fp}.{k}.1", t, stride=2, relu=k != i - j - 1)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hrnet_backbone(P, bn, pre, x):
    """HighResolutionNet.forward, hrnet_backbone.py:505-545 (hrnet48; stem = two stride-2 3x3 convs)."""
    x = _conv_bn(P, bn, pre + ".conv1", pre + ".bn1", x, stride=2)
    x = _conv_bn(P, bn, pre + ".conv2", pre + ".bn2", x, stride=2)
    for b in range(4):
        x = hr_bottleneck(P, bn, f"{pre}.layer1.{b}", x)
    ys = [x]
    for si, (stage, nmod, chans) in enumerate(HRNET_W48, 1):
        tp = f"{pre}.transition{si}"
        xs = []
        for i in range(len(chans)):
            if i < len(ys):
                if (f"{tp}.{i}.0.weight") in P:      # only transition1.0 (256 -> 48): channel counts match afterwards
                    xs.append(_conv_bn(P, bn, f"{tp}.{i}.0", f"{tp}.{i}.1", ys[i]))
                else:
                    xs.append(ys[i])
            else:                                   # new lowest-resolution branch from the previous lowest one
                xs.append(_conv_bn(P, bn, f"{tp}.{i}.0.0", f"{tp}.{i}.0.1", ys[-1], stride=2))
        for m in range(nmod):
            xs = hr_module(P, bn, f"{pre}.{stage}.{m}", xs)
        ys = xs
    return ys


def hrnet_ocr_forward(P, x, bn, drop=None):
    """HRNet_W48_OCR.forward, nets/hrnet.py:139-158, with SpatialGather_Module (spatial_ocr_block.py:37-66) and
    SpatialOCR_Module / _ObjectAttentionBlock (:114-305) for num_classes = 1 -> (main prob map, aux prob map)."""
    ys = hrnet_backbone(P, bn, "segmentation_model.backbone", x)
    h, w = ys[0].shape[-2:]
    feats = torch.cat([ys[0]] + [F.interpolate(t, size=(h, w), mode="bilinear", align_corners=True) for t in ys[1:]], 1)
    return hrnet_ocr_head(P, feats, bn, drop, x.shape[-2:])


def hrnet_ocr_head(P, feats, bn, drop, out_hw):
    """nets/hrnet.py:150-158: everything after the 720-channel concat."""
    pre = "segmentation_model"
    H, W = out_hw
    h, w = feats.shape[-2:]
    a = _conv_bn(P, bn, pre + ".aux_head.0", pre + ".aux_head.1.0", feats)
    out_aux = F.conv2d(a, P[pre + ".aux_head.2.weight"], P[pre + ".aux_head.2.bias"])
    f = _conv_bn(P, bn, pre + ".conv3x3.0", pre + ".conv3x3.1.0", feats)
    B, C = f.shape[:2]
    # soft object region = softmax over pixels of the (single-class) aux logits; context = region-weighted mean feature
    probs = F.softmax(out_aux.reshape(B, 1, -1), dim=2)
    ctx = torch.matmul(probs, f.reshape(B, C, -1).permute(0, 2, 1)).permute(0, 2, 1).unsqueeze(3)          # [B, C, 1, 1]
    ob = pre + ".ocr_distri_head.object_context_block"

    def seq(name, t, n):
        for i in range(n):
            t = _conv_bn(P, bn, f"{ob}.{name}.{2 * i}", f"{ob}.{name}.{2 * i + 1}.0", t, pad=0)
        return t
    query = seq("f_pixel", f, 2).reshape(B, 256, -1).permute(0, 2, 1)
    key = seq("f_object", ctx, 2).reshape(B, 256, -1)
    value = seq("f_down", ctx, 1).reshape(B, 256, -1).permute(0, 2, 1)
    sim = F.softmax((256 ** -0.5) * torch.matmul(query, key), dim=-1)            # [B, hw, 1]: one object region -> all ones
    context = torch.matmul(sim, value).permute(0, 2, 1).reshape(B, 256, h, w)
    context = seq("f_up", context, 1)
    o = _conv_bn(P, bn, pre + ".ocr_distri_head.conv_bn_dropout.0", pre + ".ocr_distri_head.conv_bn_dropout.1.0",
                 torch.cat([context, f], 1), pad=0)
    o = _drop(o, drop, "ocr_drop")
    out = F.conv2d(o, P[pre + ".cls_head.weight"], P[pre + ".cls_head.bias"])
    up = lambda t: torch.sigmoid(F.interpolate(t, size=(H, W), mode="bilinear", align_corners=True))
    return up(out), up(out_aux)


def norm_sr(sr, cfg):
    """build_model.py:125-141."""
    if cfg.norm_sr == "instance":
        return F.instance_norm(sr, eps=1e-5)
    if cfg.norm_sr == "all":
        m = torch.tensor(cfg.mean, dtype=sr.dtype).reshape(1, 3, 1, 1)
        s = torch.tensor(cfg.std, dtype=sr.dtype).reshape(1, 3, 1, 1)
        return (sr - m) / s
    return sr


def factor_resize_down(x, factor, antialias=True):
    """FactorResize('bicubic') -> torchvision Resize on a float tensor, transforms.py:505-531."""
    H, W = x.shape[-2:]
    return F.interpolate(x, size=(int(H / factor), int(W / factor)), mode="bicubic", align_corners=False,
                         antialias=antialias)


def kbpn_loss(sr, hr, x_lr, kvec, kernel_gt, cfg, seg=None, seg_t=None, it=0):
    """KBPNLoss.forward + Get_pseudo_lr, sr_loss_functions.py:39-102 -> (loss (B,), kernel (B,1,K,K))."""
    B = sr.shape[0]
    hr_l = (sr - hr).abs()
    vec = kvec / kvec.sum(dim=1, keepdim=True)
    blurred = blur_down(sr, vec, cfg.ksize_out, 1)
    lr_pred = factor_resize_down(blurred, cfg.scale, cfg.antialias)
    lr_l = (lr_pred - x_lr).abs()
    kpred = vec.reshape(B, 1, cfg.ksize_out, cfg.ksize_out)
    k_l = (kpred - kernel_gt) ** 2
    if it > cfg.oriented_w_iter and cfg.oriented_w_iter != -1 and cfg.sfo_sr_amp != 0:
        # SegmentFailerOrientedExpWeight, oriented_weight.py:73-83; sr_loss_functions.py:58-71
        w = torch.exp(cfg.sfo_sr_amp * (seg.detach() - seg_t).abs())
        hr_l = w * hr_l
        lr_l = F.interpolate(w, scale_factor=1 / cfg.scale, mode="bilinear") * lr_l
    if cfg.only_kernel_loss and cfg.kernel_pretrain[0] <= it < cfg.kernel_pretrain[1]:
        # sr_loss_functions.py:50-51 returns the UNREDUCED kernel MSE map there; trainer.calc_loss takes its mean, which equals the mean of
        # the per-sample means returned here (one value per sample keeps the (B,) contract of every other phase)
        return k_l.mean((1, 2, 3)), kpred
    loss = cfg.sr_w[0] * hr_l.mean((1, 2, 3)) + cfg.sr_w[1] * lr_l.mean((1, 2, 3)) + cfg.sr_w[2] * k_l.mean((1, 2, 3))
    return loss, kpred


def edt_exact(mask: np.ndarray) -> np.ndarray:
    """Exact Euclidean distance of every non-zero pixel to the nearest zero pixel (0 for zero pixels):
    what scipy.ndimage.distance_transform_edt computes.  Two-pass lower envelope (Felzenszwalb &
    Huttenlocher) in float64; tests cross-check it against scipy on the golden masks."""
    H, W = mask.shape
    INF = 1e20
    f = np.where(mask != 0, INF, 0.0)

    def dt1d(f):
        n = f.shape[0]
        d = np.empty(n)
        v = np.zeros(n, dtype=np.int64)
        z = np.empty(n + 1)
        k = 0
        z[0], z[1] = -INF, INF
        for q in range(1, n):
            while True:
                s = ((f[q] + q * q) - (f[v[k]] + v[k] * v[k])) / (2.0 * q - 2.0 * v[k])
                if s <= z[k]:
                    k -= 1
                else:
                    break
            k += 1
            v[k] = q
            z[k], z[k + 1] = s, INF
        k = 0
        for q in range(n):
            while z[k + 1] < q:
                k += 1
            d[q] = (q - v[k]) ** 2 + f[v[k]]
        return d
    g = np.empty_like(f)
    for x in range(W):
        g[:, x] = dt1d(f[:, x])
    for y in range(H):
        g[y, :] = dt1d(g[y, :])
    g = np.where(g >= INF / 2, 0.0, g)  # no zero pixel anywhere: scipy returns... (never hit: guarded by any())
    return np.sqrt(g)


def inner_boundary(pos: np.ndarray) -> np.ndarray:
    """skimage.segmentation.find_boundaries(mode='inner', connectivity=1) on a bool image: foreground
    pixels whose 4-neighbourhood (reflecting nothing: grey dilation/erosion use 'reflect' borders, so
    image edges never create a boundary) contains background.  boundary_loss.py:62."""
    p = pos.astype(bool)
    pad = np.pad(p, 1, mode="edge")
    nb_all = pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:]
    return p & ~nb_all


def compute_sdf(mask: np.ndarray, use_scipy=True) -> np.ndarray:
    """compute_sdf1_1, boundary_loss.py:40-67, for a (B,1,H,W) {0,1} array -> float64 (B,1,H,W)."""
    B = mask.shape[0]
    out = np.zeros(mask.shape, dtype=np.float64)
    if use_scipy:
        from scipy.ndimage import distance_transform_edt as edt
    else:
        edt = edt_exact
    for b in range(B):
        pos = mask[b, 0].astype(np.uint8).astype(bool)
        if pos.any():
            neg = ~pos
            posdis, negdis = edt(pos), edt(neg)
            bd = inner_boundary(pos)
            sdf = (negdis - negdis.min()) / (negdis.max() - negdis.min()) - (posdis - posdis.min()) / (posdis.max() - posdis.min())
            sdf[bd] = 0
            out[b, 0] = sdf
    return out


def boundary_combo_loss(pred, target, alpha, cfg, sdf=None):
    """BoundaryComboLoss.forward, loss_functions.py:49-75 with BCE_DiceLoss (:317-345),
    WeightedBCELoss (:189-210), BinaryDiceLoss (:258-314) and BoundaryLoss (boundary_loss.py:26-38)."""
    sm = 1e-8
    p = pred.clamp(min=sm)
    pw = cfg.bce_w
    # WeightedBCELoss clamps again (no-op) and adds smooth inside the logs
    bce = -(pw[0] * target * torch.log(p + sm) + pw[1] * (1 - target) * torch.log(1 - p + sm)) / sum(pw)
    bce = bce.mean(dim=(1, 2, 3))
    pf, tf = p.reshape(p.shape[0], -1), target.reshape(target.shape[0], -1)
    num = 2 * (pf * tf).sum(1) + 1e-6
    den = (pf.pow(2) + tf.pow(2)).sum(1) + 1e-6
    dice = 1 - num / den
    lw = cfg.wbd_w
    wbd = (lw[0] * bce + lw[1] * dice) / sum(lw)
    if sdf is None:
        sdf = torch.from_numpy(compute_sdf(target.detach().cpu().numpy())).float()
    bd = (p * sdf).mean(dim=(1, 2, 3))
    return alpha * wbd + (1 - alpha) * bd


# ----------------------------------------------------------------------------- joint forward

def joint_forward(P, cfg, it, x, hr, mask, kernel_gt, alpha=1.0, drop=None, training=True, taps=None):
    """JointModelWithLoss.forward (KBPN + PSPNet branch), build_model.py:402-416."""
    sr, kvec = kbpn_forward(P, x, it, kernel_gt, cfg, taps)
    bn = BNState(P, training)
    if cfg.detector == "HRNet_OCR":
        seg, aux = hrnet_ocr_forward(P, norm_sr(sr, cfg), bn, drop)
    else:
        seg, aux = pspnet_forward(P, norm_sr(sr, cfg), bn, drop, kvec if cfg.detector == "PSPNet_BlurSkip" else None)
    sr_loss, kpred = kbpn_loss(sr, hr, x, kvec, kernel_gt, cfg, seg, mask, it)
    sdf = torch.from_numpy(compute_sdf(mask.cpu().numpy())).float()
    seg_loss = cfg.main_w * boundary_combo_loss(seg, mask, alpha, cfg, sdf) + \
        cfg.aux_w * boundary_combo_loss(aux, mask, alpha, cfg, sdf)
    return {"segment_loss": seg_loss, "sr_loss": sr_loss, "segment_preds": seg, "sr_preds": sr,
            "kernel_preds": kpred, "aux_preds": aux, "bn_buffers": bn.new}


def calc_loss(seg_loss, sr_loss, it, cfg):
    """trainer.py:406-438 (JOINT_LEARNING, fixed TASK_LOSS_WEIGHT)."""
    seg_l, sr_l = seg_loss.mean(), sr_loss.mean()
    loss = (1 - cfg.beta) * sr_l + cfg.beta * seg_l
    if cfg.joint_pretrain[0] <= it < cfg.joint_pretrain[1]:
        loss = sr_l
    return loss


def iou(output, target, th=0.5, smooth=1e-5):
    """estimate_metrics.py:64-84."""
    o, t = output > th, target > th
    inter = (o & t).sum(dim=(2, 3)).double()
    union = (o | t).sum(dim=(2, 3)).double()
    return (inter + smooth) / (union + smooth)
