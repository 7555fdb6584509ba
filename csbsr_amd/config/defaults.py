"""Config tree with the reference's key names (/root/reference/model/config/defaults.py:14-121) for the
keys the hot path consumes, so ``cfg.merge_from_file('config/config_csbsr_pspnet.yaml')`` written for the
reference works unchanged.  yacs is not required: CfgNode here is a small attribute dict with
merge_from_file / merge_from_list / freeze."""
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v
        object.__setattr__(self, "_frozen", False)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen"):
            raise AttributeError(f"cfg is frozen; cannot set {k}")
        self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

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
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def clone(self):
        c = copy.deepcopy(self)
        c.defrost()
        return c


_C = CfgNode({
    "DEVICE": "cuda",
    "MODEL": {"SCALE_FACTOR": 4, "DETECTOR_TYPE": "PSPNet", "SR": "KBPN", "NUM_CLASSES": 1, "NUM_STAGES": 4,
              "SR_SEG_INV": False, "JOINT_LEARNING": True, "SR_RESIDUAL_LEARNING": True, "KBPN_KERNEL_SFT": True,
              "SR_PIXEL_SHUFFLE": False, "SR_SCRATCH": True, "SUM_LR_ERROR_POS": "HR", "ZERO_PAD_KERNEL": False,
              "OPTIMIZER": "Adam", "UP_SAMPLE_METHOD": "deconv"},
    "SOLVER": {"MAX_ITER": 300000, "SR_PRETRAIN_ITER": [1, 30001], "SR_SR_MODULE_PRETRAIN_ITER": [1, 10001],
               "SR_KERNEL_MODULE_PRETRAIN_ITER": [10001, 20001], "ONLY_KERNEL_LOSS_FOR_PRETRAIN": False,
               "SEG_PRETRAIN_ITER": [0, 0], "BATCH_SIZE": 8, "TASK_LOSS_WEIGHT": 0.3, "SEG_LOSS_FUNC": "BoundaryCombo",
               "BOUNDARY_DEC_RATIO": 1.0, "WB_AND_D_WEIGHT": [1, 1], "BCELOSS_WEIGHT": [1, 1], "SEG_AUX_LOSS_WEIGHT": 0.4,
               "SEG_MAIN_LOSS_WEIGHT": 1.0, "ORIENTED_WEIGHT_ITER": -1, "SEG_FAIL_ORIENTED_WEIGHT4SR_AMP": 0.0,
               "SEG_FAIL_ORIENTED_WEIGHT4SS_AMP": 0.0, "CRACK_ORIENTED_WEIGHT4SR_AMP": 0.0, "INTERM_SSLOSSWEGHT4SR": False,
               "SR_LOSS_FUNC": "KBPN", "SR_LOSS_FUNC_SR_WEIGHT": [0.4, 0.4, 0, 2], "NORM_SR_OUTPUT": "instance", "LR": 2e-5,
               "SCHEDULER": False, "DOWNSCALE_INTERPOLATION": "bicubic"},
    "BLUR": {"FLAG": True, "KERNEL_SIZE": 7, "KERNEL_SIZE_OUTPUT": 21, "ISOTROPIC": False},
    "INPUT": {"IMAGE_SIZE": [448, 448], "MEAN": [0.4741, 0.4937, 0.5048], "STD": [0.1621, 0.1532, 0.1523]},
    "OUTPUT_DIR": "output/CSBSR", "SEED": 1121,
})
cfg = _C


class PathConfig:
    """Flat view of the keys the kernels need."""
    pass


def path_config(c, antialias=True):
    p = PathConfig()
    p.scale = c.MODEL.SCALE_FACTOR
    p.num_stages = c.MODEL.NUM_STAGES
    p.ksize = c.BLUR.KERNEL_SIZE
    p.ksize_out = c.BLUR.KERNEL_SIZE_OUTPUT
    p.sr_pretrain = tuple(c.SOLVER.SR_SR_MODULE_PRETRAIN_ITER)
    p.kernel_pretrain = tuple(c.SOLVER.SR_KERNEL_MODULE_PRETRAIN_ITER)
    p.joint_pretrain = tuple(c.SOLVER.SR_PRETRAIN_ITER)
    p.norm_sr = c.SOLVER.NORM_SR_OUTPUT
    p.mean, p.std = tuple(c.INPUT.MEAN), tuple(c.INPUT.STD)
    p.sr_w = tuple(float(v) for v in c.SOLVER.SR_LOSS_FUNC_SR_WEIGHT[:3])
    p.bce_w = tuple(float(v) for v in c.SOLVER.BCELOSS_WEIGHT)
    p.wbd_w = tuple(float(v) for v in c.SOLVER.WB_AND_D_WEIGHT)
    p.aux_w, p.main_w = float(c.SOLVER.SEG_AUX_LOSS_WEIGHT), float(c.SOLVER.SEG_MAIN_LOSS_WEIGHT)
    p.beta = float(c.SOLVER.TASK_LOSS_WEIGHT)
    p.antialias = antialias
    p.oriented_w_iter = c.SOLVER.ORIENTED_WEIGHT_ITER
    p.sfo_sr_amp = float(c.SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP)
    p.pixel_shuffle = bool(c.MODEL.SR_PIXEL_SHUFFLE)
    p.residual_learning = bool(c.MODEL.SR_RESIDUAL_LEARNING)             # kbpn.py:32,69-70,112-116
    p.only_kernel_loss = bool(c.SOLVER.ONLY_KERNEL_LOSS_FOR_PRETRAIN)    # sr_loss_functions.py:32,50-51
    p.kernel_sft = bool(c.MODEL.KBPN_KERNEL_SFT)                         # kbpn.py:165,169-171,190: False = no SFT layer between the stages
    p.zero_pad_kernel = bool(c.MODEL.ZERO_PAD_KERNEL)                    # kbpn.py:543-554,583-596: per-sample choice zero-pad / bicubic for the 7x7 -> 21x21 kernel update
    p.lr_error = str(c.MODEL.SUM_LR_ERROR_POS) == "LR"                   # kbpn.py:166,174-187,369-374,404-409: back-projection error added at LR
    return p
