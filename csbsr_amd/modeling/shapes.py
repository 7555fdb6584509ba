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


def _cb(d, pre, cin, cout, k, bias=False, prelu=False, deconv=False):
    d[pre + ".layer.weight"] = (cin, cout, k, k) if deconv else (cout, cin, k, k)
    if bias:
        d[pre + ".layer.bias"] = (cout,)
    if prelu:
        d[pre + ".act.weight"] = (1,)


def kbpn_shapes(scale=4, num_stages=4, ksize=7, ksize_out=21, md=128, prefix="sr_model"):
    d = OrderedDict()
    k, s, p = CONV_SETTING[scale]
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
        _cb(d, sp + ".up.up_conv1", md, md, k, prelu=True, deconv=True)
        _cb(d, sp + ".up.up_conv3", md, md, k, prelu=True, deconv=True)
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
        _cb(d, sp + ".kb.up_conv1", 3, md, k, prelu=True, deconv=True)
        if st < num_stages:
            _cb(d, sp + ".down.conv", md * st, md, 1, bias=True, prelu=True)
            _cb(d, sp + ".down.down_conv1", md, md, k, prelu=True)
            _cb(d, sp + ".down.down_conv3", md, md, k, prelu=True)
            _cb(d, sp + ".down.down_conv2", md, md, k, prelu=True, deconv=True)
            cc = md * st + cond
            for nm, co in (("SFT_scale_conv0", cc), ("SFT_scale_conv1", md * st), ("SFT_shift_conv0", cc), ("SFT_shift_conv1", md * st)):
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


def joint_state_shapes(scale=4, num_stages=4, ksize=7, ksize_out=21, detector="PSPNet"):
    """state_dict order of JointModelWithLoss: segmentation_model.* first, then sr_model.*
    (MetaSSModel.__init__ runs before MetaSRModel's body, build_model.py:52-60,191-197)."""
    d = OrderedDict()
    if detector == "PSPNet":
        d.update(pspnet_shapes())
    elif detector == "PSPNet_BlurSkip":
        d.update(pspnet_shapes(blur_dim=ksize_out * ksize_out))
    else:
        raise NotImplementedError(detector)
    d.update(kbpn_shapes(scale, num_stages, ksize, ksize_out))
    return d
