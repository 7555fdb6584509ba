"""Parameter / buffer registry of the joint network: ``{state_dict key: shape}`` with the reference's
key names so released ``.pth`` files and the deterministic test fill apply unchanged.

Key layout follows the reference modules (nothing imported from them):
  sr_model.*            KBPN, /root/reference/model/modeling/kbpn.py:17-116, 145-189, 292-602
  segmentation_model.*  PSPNet on dilated ResNet-34, pspnet_pytorch/pspnet.py:59-93, extractors.py:112-147
"""
from collections import OrderedDict

CONV_SETTING = {2: (6, 2, 2), 4: (8, 4, 2), 8: (12, 8, 2)}   # kbpn.py:22-25 (k, stride, pad)
RESNET34 = ((64, 3, 1, 1), (128, 4, 2, 1), (256, 6, 1, 2), (512, 3, 1, 4))  # planes, blocks, stride, dilation


def _bn(d, pre, c):
    d[pre + ".weight"] = (c,)
    d[pre + ".bias"] = (c,)
    d[pre + ".running_mean"] = (c,)
    d[pre + ".running_var"] = (c,)
    d[pre + ".num_batches_tracked"] = ()


def _cb(d, pre, cin, cout, k, bias=False, prelu=False, deconv=False, shuffle=0):
    """``shuffle`` = s: ConvAndPixelShuffleBlock (kbpn.py:280-289) in place of the DeconvBlock: a 3x3 conv to cout * s^2 channels"""
    if shuffle:
        d[pre + ".layer.weight"] = (cout * shuffle * shuffle, cin, 3, 3)
    else:
        d[pre + ".layer.weight"] = (cin, cout, k, k) if deconv else (cout, cin, k, k)
    if bias:
        d[pre + ".layer.bias"] = (cout,)
    if prelu:
        d[pre + ".act.weight"] = (1,)


def kbpn_shapes(scale=4, num_stages=4, ksize=7, ksize_out=21, md=128, prefix="sr_model", pixel_shuffle=False, kernel_sft=True, lr_error=False,
                zero_pad_kernel=False):
    """pixel_shuffle: MODEL.SR_PIXEL_SHUFFLE -- the four up-sampling layers of a stage are conv3x3 + PixelShuffle(scale)
    (kbpn.py:372-373,457-459,479-480) instead of transposed convolutions; same key names, different weight shapes.
    kernel_sft: MODEL.KBPN_KERNEL_SFT -- False: the stages have no ``sft`` module (kbpn.py:169-171).
    zero_pad_kernel: MODEL.ZERO_PAD_KERNEL -- every stage's kernel predictor carries the three Linear layers of its ``pad_descriminator``
    (kbpn.py:543-554).
    lr_error: MODEL.SUM_LR_ERROR_POS == 'LR' -- ``kb.conv`` (3x3, 3 -> md at LR, no activation) in place of ``kb.up_conv1`` (kbpn.py:369-374)."""
    d = OrderedDict()
    k, s, p = CONV_SETTING[scale]
    ps = scale if pixel_shuffle else 0
    for i, (ci, co) in zip((0, 2, 4, 6), ((3, 64), (64, 64), (64, 128), (128, 128))):
        d[f"{prefix}.feat.{i}.weight"] = (co, ci, 3, 3)
        d[f"{prefix}.feat.{i}.bias"] = (co,)
    kc = ksize * ksize
    cond = ksize_out * ksize_out
    for i, co in enumerate((md, md, kc)):
        _cb(d, f"{prefix}.predictor.feat_ext.{i}", md, co, 3, prelu=True)
    for st in range(1, num_stages + 1):
        sp = f"{prefix}.back_projection_stages.{st - 1}"
        up_st = max(st - 1, 1)
        _cb(d, sp + ".up.conv", md * up_st, md, 1, bias=True, prelu=True)
        _cb(d, sp + ".up.up_conv2", md, md, k, prelu=True)
        _cb(d, sp + ".up.up_conv1", md, md, k, prelu=True, deconv=True, shuffle=ps)
        _cb(d, sp + ".up.up_conv3", md, md, k, prelu=True, deconv=True, shuffle=ps)
        _cb(d, sp + ".kb.sr_reconst", md * st, 3, 3)
        kp = sp + ".kb.kernel_predictor"
        _cb(d, kp + ".fe_SR.0", 3, kc, 3)
        _cb(d, kp + ".fe_SR.1", kc, 32, 1)
        _cb(d, kp + ".fe_SR.2", 32, 32, 3)
        _cb(d, kp + ".fe_SR.3", 32, 32, 3)
        _cb(d, kp + ".fe_SR.4", 32, kc, 3)
        _cb(d, kp + ".fe_kernel.0", cond, kc, 3)
        _cb(d, kp + ".fe_kernel.1", kc, kc, 3)
        _cb(d, kp + ".fe_cat.0", 2 * kc, 32, 1)
        _cb(d, kp + ".fe_cat.1", 32, 32, 3)
        _cb(d, kp + ".fe_cat.2", 32, kc, 3)
        if zero_pad_kernel:
            for i, (ci, co) in zip((0, 3, 6), ((kc, 8), (8, 8), (8, 1))):
                d[f"{kp}.pad_descriminator.{i}.weight"] = (co, ci)
                d[f"{kp}.pad_descriminator.{i}.bias"] = (co,)
        if lr_error:
            _cb(d, sp + ".kb.conv", 3, md, 3)
        else:
            _cb(d, sp + ".kb.up_conv1", 3, md, k, prelu=True, deconv=True, shuffle=ps)
        if st < num_stages:
            _cb(d, sp + ".down.conv", md * st, md, 1, bias=True, prelu=True)
            _cb(d, sp + ".down.down_conv1", md, md, k, prelu=True)
            _cb(d, sp + ".down.down_conv3", md, md, k, prelu=True)
            _cb(d, sp + ".down.down_conv2", md, md, k, prelu=True, deconv=True, shuffle=ps)
            cc = md * st + cond
            for nm, co in () if not kernel_sft else (("SFT_scale_conv0", cc), ("SFT_scale_conv1", md * st), ("SFT_shift_conv0", cc), ("SFT_shift_conv1", md * st)):
                d[f"{sp}.sft.{nm}.weight"] = (co, cc, 3, 3)
                d[f"{sp}.sft.{nm}.bias"] = (co,)
    _cb(d, f"{prefix}.output_conv", md * num_stages, 3, 3)
    return d


def pspnet_shapes(prefix="segmentation_model", n_classes=1, blur_dim=None, n_layer_blurskip=2):
    """blur_dim: PSPNet_BlurSkip (pspnet.py:127-159): SFTLikeBlock(blur_dim+64 -> 64) + ConvBlock(64, 64, BN, ReLU), x n_layer."""
    d = OrderedDict()
    f = prefix + ".feats"
    d[f + ".conv1.weight"] = (64, 3, 7, 7)
    _bn(d, f + ".bn1", 64)
    inpl = 64
    for li, (planes, blocks, stride, dil) in enumerate(RESNET34, 1):
        for b in range(blocks):
            bp = f"{f}.layer{li}.{b}"
            d[bp + ".conv1.weight"] = (planes, inpl if b == 0 else planes, 3, 3)
            _bn(d, bp + ".bn1", planes)
            d[bp + ".conv2.weight"] = (planes, planes, 3, 3)
            _bn(d, bp + ".bn2", planes)
            if b == 0 and (stride != 1 or inpl != planes):
                d[bp + ".downsample.0.weight"] = (planes, inpl, 1, 1)
                _bn(d, bp + ".downsample.1", planes)
        inpl = planes
    for i in range(4):
        d[f"{prefix}.psp.stages.{i}.1.weight"] = (512, 512, 1, 1)
    d[prefix + ".psp.bottleneck.weight"] = (1024, 2560, 1, 1)
    d[prefix + ".psp.bottleneck.bias"] = (1024,)
    for nm, (ci, co) in (("up_1", (1024, 256)), ("up_2", (256, 64)), ("up_3", (64, 64))):
        d[f"{prefix}.{nm}.conv.0.weight"] = (co, ci, 3, 3)
        d[f"{prefix}.{nm}.conv.0.bias"] = (co,)
        _bn(d, f"{prefix}.{nm}.conv.1", co)
        d[f"{prefix}.{nm}.conv.2.weight"] = (1,)
    if blur_dim is not None:
        cc = blur_dim + 64
        for i in range(n_layer_blurskip):
            sp = f"{prefix}.blur_skip.{2 * i}"
            for br, last_act in (("conv_scale", None), ("conv_shift", None)):
                d[f"{sp}.{br}.0.layer.weight"] = (cc, cc, 3, 3)
                d[f"{sp}.{br}.0.layer.bias"] = (cc,)
                d[f"{sp}.{br}.0.act.weight"] = (1,)
                d[f"{sp}.{br}.1.layer.weight"] = (64, cc, 3, 3)
                d[f"{sp}.{br}.1.layer.bias"] = (64,)
            cp = f"{prefix}.blur_skip.{2 * i + 1}"
            d[cp + ".layer.weight"] = (64, 64, 3, 3)
            _bn(d, cp + ".norm", 64)
    d[prefix + ".final.0.weight"] = (n_classes, 64, 1, 1)
    d[prefix + ".final.0.bias"] = (n_classes,)
    d[prefix + ".aux.0.weight"] = (256, 256, 3, 3)
    _bn(d, prefix + ".aux.1", 256)
    d[prefix + ".aux.4.weight"] = (n_classes, 256, 1, 1)
    d[prefix + ".aux.4.bias"] = (n_classes,)
    return d


HRNET_W48 = (("stage2", 1, (48, 96)), ("stage3", 4, (48, 96, 192)), ("stage4", 3, (48, 96, 192, 384)))   # hrnet_config.py:46-73


def hrnet_ocr_shapes(prefix="segmentation_model", n_classes=1):
    """HRNet_W48_OCR (nets/hrnet.py:101-137) over HighResolutionNet hrnet48 (hrnet_backbone.py:295-503), registration order."""
    d = OrderedDict()
    b = prefix + ".backbone"

    def cb(conv, norm, cin, cout, k, bias=False):
        d[conv + ".weight"] = (cout, cin, k, k)
        if bias:
            d[conv + ".bias"] = (cout,)
        _bn(d, norm, cout)
    cb(b + ".conv1", b + ".bn1", 3, 64, 3)
    cb(b + ".conv2", b + ".bn2", 64, 64, 3)
    for i in range(4):
        bp = f"{b}.layer1.{i}"
        cb(bp + ".conv1", bp + ".bn1", 64 if i == 0 else 256, 64, 1)
        cb(bp + ".conv2", bp + ".bn2", 64, 64, 3)
        cb(bp + ".conv3", bp + ".bn3", 64, 256, 1)
        if i == 0:
            cb(bp + ".downsample.0", bp + ".downsample.1", 64, 256, 1)
    prev = (256,)
    for si, (stage, nmod, chans) in enumerate(HRNET_W48, 1):
        tp = f"{b}.transition{si}"
        for i, c in enumerate(chans):
            if i < len(prev):
                if prev[i] != c:
                    cb(f"{tp}.{i}.0", f"{tp}.{i}.1", prev[i], c, 3)
            else:
                cb(f"{tp}.{i}.0.0", f"{tp}.{i}.0.1", prev[-1], c, 3)
        for m in range(nmod):
            mp = f"{b}.{stage}.{m}"
            for i, c in enumerate(chans):
                for blk in range(4):
                    bp = f"{mp}.branches.{i}.{blk}"
                    cb(bp + ".conv1", bp + ".bn1", c, c, 3)
                    cb(bp + ".conv2", bp + ".bn2", c, c, 3)
            for i, ci in enumerate(chans):
                for j, cj in enumerate(chans):
                    fp = f"{mp}.fuse_layers.{i}.{j}"
                    if j > i:
                        cb(fp + ".0", fp + ".1", cj, ci, 1)
                    elif j < i:
                        for k in range(i - j):
                            cb(f"{fp}.{k}.0", f"{fp}.{k}.1", cj, ci if k == i - j - 1 else cj, 3)
        prev = chans
    cb(prefix + ".conv3x3.0", prefix + ".conv3x3.1.0", 720, 512, 3, bias=True)
    ob = prefix + ".ocr_distri_head.object_context_block"
    cb(ob + ".f_pixel.0", ob + ".f_pixel.1.0", 512, 256, 1, bias=True)
    cb(ob + ".f_pixel.2", ob + ".f_pixel.3.0", 256, 256, 1, bias=True)
    cb(ob + ".f_object.0", ob + ".f_object.1.0", 512, 256, 1, bias=True)
    cb(ob + ".f_object.2", ob + ".f_object.3.0", 256, 256, 1, bias=True)
    cb(ob + ".f_down.0", ob + ".f_down.1.0", 512, 256, 1, bias=True)
    cb(ob + ".f_up.0", ob + ".f_up.1.0", 256, 512, 1, bias=True)
    cb(prefix + ".ocr_distri_head.conv_bn_dropout.0", prefix + ".ocr_distri_head.conv_bn_dropout.1.0", 1024, 512, 1, bias=True)
    d[prefix + ".cls_head.weight"] = (n_classes, 512, 1, 1)
    d[prefix + ".cls_head.bias"] = (n_classes,)
    cb(prefix + ".aux_head.0", prefix + ".aux_head.1.0", 720, 720, 3, bias=True)
    d[prefix + ".aux_head.2.weight"] = (n_classes, 720, 1, 1)
    d[prefix + ".aux_head.2.bias"] = (n_classes,)
    return d


def joint_state_shapes(scale=4, num_stages=4, ksize=7, ksize_out=21, detector="PSPNet", pixel_shuffle=False, kernel_sft=True, lr_error=False,
                       zero_pad_kernel=False):
    """state_dict order of JointModelWithLoss: segmentation_model.* first, then sr_model.*
    (MetaSSModel.__init__ runs before MetaSRModel's body, build_model.py:52-60,191-197)."""
    d = OrderedDict()
    if detector == "PSPNet":
        d.update(pspnet_shapes())
    elif detector == "PSPNet_BlurSkip":
        d.update(pspnet_shapes(blur_dim=ksize_out * ksize_out))
    elif detector == "HRNet_OCR":
        d.update(hrnet_ocr_shapes())
    else:
        raise NotImplementedError(detector)
    d.update(kbpn_shapes(scale, num_stages, ksize, ksize_out, pixel_shuffle=pixel_shuffle, kernel_sft=kernel_sft, lr_error=lr_error,
                         zero_pad_kernel=zero_pad_kernel))
    return d
