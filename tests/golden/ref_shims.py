"""Import shims that let the *reference* (``/root/reference``, Python, read-only) run on CPU in
this container so golden vectors can be generated from it.  Used ONLY by ``make_golden.py``.

Nothing here is reference code: these are stand-ins for third-party packages the container lacks
(yacs, torchvision, skimage, cv2, timm, wandb) plus a ``"cuda" -> "cpu"`` device remap, following the
recipe recorded in SURVEY.md Appendix B.  The reference never travels to the GPU box; only the
``.npz`` fixtures this produces do.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import yaml

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class CfgNode(dict):
    """Minimal yacs.config.CfgNode: attribute dict + merge_from_file + freeze."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def _merge(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                if k not in self:
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                self[k] = v

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v

    def freeze(self):
        pass

    def defrost(self):
        pass

    def clone(self):
        import copy
        return copy.deepcopy(self)


def _vgg16(pretrained=False, **kw):
    cfgs = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    layers, cin = [], 3
    for v in cfgs:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    m = nn.Module()
    m.features = nn.Sequential(*layers)
    return m


class _InterpolationMode:
    BICUBIC = "bicubic"
    BILINEAR = "bilinear"
    NEAREST = "nearest"


ANTIALIAS = True  # torchvision>=0.17 default for tensors; recorded in fixture metadata


class _Resize:
    def __init__(self, size, interpolation="bilinear", **kw):
        self.size, self.mode = size, interpolation

    def __call__(self, img):
        squeeze = img.dim() == 3
        if squeeze:
            img = img.unsqueeze(0)
        out = F.interpolate(img, size=tuple(self.size), mode=self.mode, align_corners=False,
                            antialias=ANTIALIAS)
        return out.squeeze(0) if squeeze else out


def find_boundaries(label_img, connectivity=1, mode="thick", background=0):
    """skimage.segmentation.find_boundaries restated for bool arrays of any ndim, mode='inner'
    (published algorithm: dilation != erosion with the connectivity-1 footprint, AND foreground)."""
    from scipy import ndimage as ndi
    if label_img.dtype == bool:
        label_img = label_img.astype(np.uint8)
    ndim = label_img.ndim
    footprint = ndi.generate_binary_structure(ndim, connectivity)
    boundaries = ndi.grey_dilation(label_img, footprint=footprint) != ndi.grey_erosion(label_img, footprint=footprint)
    if mode == "inner":
        foreground = label_img != background
        boundaries &= foreground
    elif mode != "thick":
        raise NotImplementedError(mode)
    return boundaries


def install():
    os.environ.setdefault("HOME", "/tmp")
    _mod("yacs")
    _mod("yacs.config", CfgNode=CfgNode)
    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models", vgg16=_vgg16)
    _mod("torchvision.models.densenet", densenet121=None, densenet161=None)
    _mod("torchvision.models.squeezenet", squeezenet1_1=None)
    _mod("torchvision.models.resnet")
    tf = _mod("torchvision.transforms", InterpolationMode=_InterpolationMode, Resize=_Resize,
              RandomCrop=object, RandomResizedCrop=object, ToPILImage=object, __all__=[])
    tv.transforms = tf
    tf.functional = _mod("torchvision.transforms.functional", InterpolationMode=_InterpolationMode)
    sk = _mod("skimage")
    sk.segmentation = _mod("skimage.segmentation", find_boundaries=find_boundaries)
    sk.draw = _mod("skimage.draw", disk=None)

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):          # dunder probes (inspect.getmodule walks sys.modules reading __file__): behave like a module
                raise AttributeError(k)
            return 0
    sys.modules["cv2"] = _Any("cv2")
    _mod("wandb")
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=nn.Identity, to_2tuple=lambda x: (x, x), trunc_normal_=lambda *a, **k: None)
    _mod("timm.models.registry", register_model=lambda f: f)
    _mod("timm.models.vision_transformer", _cfg=lambda **k: {})

    # "cuda" -> cpu
    _t_to, _m_to = torch.Tensor.to, nn.Module.to

    def _fix(a):
        return "cpu" if (isinstance(a, str) and a.startswith("cuda")) else a

    def t_to(self, *a, **k):
        return _t_to(self, *[_fix(x) for x in a], **{kk: _fix(v) for kk, v in k.items()})

    def m_to(self, *a, **k):
        return _m_to(self, *[_fix(x) for x in a], **{kk: _fix(v) for kk, v in k.items()})

    torch.Tensor.to = t_to
    nn.Module.to = m_to
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self

    sys.path.insert(0, REF)
    os.chdir(REF)


def build_reference(detector="PSPNet", scale=4, overrides=()):
    """JointModelWithLoss from the reference, yaml config, no pretrained downloads."""
    from model.config import cfg as _cfg
    import copy
    cfg = copy.deepcopy(_cfg)
    cfg.merge_from_file(os.path.join(REF, "config/config_csbsr_pspnet.yaml"))
    cfg.MODEL.SR_SCRATCH = True
    cfg.MODEL.DETECTOR_TYPE = detector
    cfg.MODEL.SCALE_FACTOR = scale
    cfg.merge_from_list(list(overrides))
    from model.modeling.pspnet_pytorch import extractors
    extractors.resnet34 = lambda pretrained=True: extractors.ResNet(extractors.BasicBlock, [3, 4, 6, 3])
    # HRNet backbone: the ImageNet checkpoint is not in the container (no network); keep the random init, weights are
    # overwritten by the deterministic fill anyway
    from model.modeling.hrnet_ocr.tools.module_helper import ModuleHelper
    ModuleHelper.load_model = staticmethod(lambda model, pretrained=None, all_match=True, network=None: model)
    from model.modeling.build_model import JointModelWithLoss, JointModel
    from model.data.transforms.transforms import FactorResize
    return cfg, JointModelWithLoss, JointModel, FactorResize
