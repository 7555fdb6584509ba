"""ctypes binding of libcsbsr_hip.so (the C ABI declared in include/csbsr_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel reports an error this
module raises.  PyTorch-ROCm is used only for device memory (tensors) and streams; every pointer that
crosses this boundary is a raw ``data_ptr()``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSBSR_LIB") or os.path.join(_HERE, "libcsbsr_hip.so")       # CSBSR_LIB: a variant build (kernel A/B experiments)

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_PRELU, ACT_SIGMOID = 0, 1, 2, 3, 4
RES_NONE, RES_ADD, RES_SUB, RES_MUL, RES_FMA = 0, 1, 2, 3, 4
STAT_NONE, STAT_BN, STAT_SAMPLE_SUM = 0, 1, 2

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class Seg(C.Structure):
    _fields_ = [("ptr", vp), ("sn", i64), ("sy", i64), ("sx", i64), ("c", i32), ("creal", i32)]


class ConvDesc(C.Structure):
    _fields_ = [("inp", Seg * 2), ("N", i32), ("H", i32), ("W", i32), ("OH", i32), ("OW", i32),
                ("transposed", i32), ("KH", i32), ("KW", i32), ("stride", i32), ("pad", i32), ("dil", i32),
                ("wt", vp), ("cout", i32), ("coutp", i32),
                ("out16", vp), ("o_sn", i64), ("o_sy", i64), ("o_sx", i64),
                ("out32", vp), ("o32_sn", i64), ("o32_sy", i64), ("o32_sx", i64), ("o32_sc", i64),
                ("bias", vp), ("cbias", vp), ("act", i32), ("act_slope", f32), ("prelu", vp),
                ("res_mode", i32), ("res", vp), ("r_sn", i64), ("r_sy", i64), ("r_sx", i64),
                ("res2", vp), ("r2_sn", i64), ("r2_sy", i64), ("r2_sx", i64),
                ("accumulate", i32), ("stat_mode", i32), ("stat", vp), ("out_scale", f32),
                ("o_lo", i64), ("r_lo", i64), ("r2_lo", i64),
                ("mask", vp), ("m_sn", i64), ("m_sy", i64), ("m_sx", i64), ("mask_slope", f32), ("cbias_mode", i32),
                ("mask_prelu", vp), ("dact_bias", vp), ("dact_prelu", vp),
                ("dres", vp), ("dr_sn", i64), ("dr_sy", i64), ("dr_sx", i64),
                ("split_fused", i32), ("_pad_sf", i32), ("bias_sn", i64)]


class WgradDesc(C.Structure):
    _fields_ = [("a", vp), ("a_sn", i64), ("a_sy", i64), ("a_sx", i64), ("ca", i32), ("ca_real", i32),
                ("b", Seg * 2), ("N", i32), ("AH", i32), ("AW", i32), ("BH", i32), ("BW", i32),
                ("KH", i32), ("KW", i32), ("stride", i32), ("pad", i32), ("dil", i32),
                ("g", vp), ("splits", i32), ("_pad1", i32)]


class EpiBwdDesc(C.Structure):
    _fields_ = [("npix", i64), ("c", i32), ("creal", i32),
                ("dout", vp), ("dout_ld", i64), ("out", vp), ("out_ld", i64),
                ("res", vp), ("res_ld", i64), ("res2", vp), ("res2_ld", i64),
                ("act", i32), ("act_slope", f32), ("prelu", vp), ("res_mode", i32), ("_pad", i32),
                ("dpre", vp), ("dpre_ld", i64),
                ("dres", vp), ("dres_ld", i64), ("dres_accumulate", i32), ("_pad1", i32),
                ("dres2", vp), ("dres2_ld", i64), ("dres2_accumulate", i32), ("_pad2", i32),
                ("dbias", vp), ("dprelu", vp)]


class BnDesc(C.Structure):
    _fields_ = [("npix", i64), ("hw", i64), ("c", i32), ("creal", i32),
                ("x", vp), ("x_ld", i64), ("mean", vp), ("invstd", vp), ("gamma", vp), ("beta", vp),
                ("res", vp), ("res_ld", i64), ("act", i32), ("_pad", i32), ("prelu", vp), ("drop", vp),
                ("y", vp), ("y_ld", i64), ("dy", vp), ("dy_ld", i64), ("red", vp), ("dprelu", vp),
                ("dx", vp), ("dx_ld", i64), ("dres", vp), ("dres_ld", i64), ("dres_accumulate", i32), ("_pad1", i32),
                ("dgamma", vp), ("dbeta", vp), ("x_lo", i64), ("res_lo", i64), ("y_lo", i64)]


# name -> (restype, argtypes): every symbol include/csbsr_hip.h declares
SIGNATURES = {
    "csbsr_version": (i32, []),
    "csbsr_last_error": (C.c_char_p, []),
    "csbsr_conv_forward": (i32, [C.POINTER(ConvDesc), vp]),
    "csbsr_conv_wgrad": (i32, [C.POINTER(WgradDesc), vp]),
    "csbsr_conv_hr_eligible": (i32, [C.POINTER(ConvDesc)]),
    "csbsr_conv_hr_forward": (i32, [C.POINTER(ConvDesc), vp]),
    "csbsr_conv_x3_eligible": (i32, [C.POINTER(ConvDesc)]),
    "csbsr_conv_x3_forward": (i32, [C.POINTER(ConvDesc), vp]),
    "csbsr_packed_weight_elems_x3": (i64, [i32, i32]),
    "csbsr_pack_weights_x3": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_conv_x3n_eligible": (i32, [C.POINTER(ConvDesc)]),
    "csbsr_conv_x3n_forward": (i32, [C.POINTER(ConvDesc), vp]),
    "csbsr_packed_weight_elems_x3n": (i64, [i32, i32]),
    "csbsr_pack_weights_x3n": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.c_float, vp]),
    "csbsr_conv_x3w_eligible": (i32, [C.POINTER(ConvDesc)]),
    "csbsr_conv_x3w_forward": (i32, [C.POINTER(ConvDesc), vp]),
    "csbsr_packed_weight_elems_x3w": (i64, [i32, i32]),
    "csbsr_pack_weights_x3w": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_packed_weight_elems_x3_strided": (i64, [i32, i32, i32]),
    "csbsr_pack_weights_x3_strided": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_conv_tp_eligible": (i32, [C.POINTER(ConvDesc)]),
    "csbsr_conv_tp_forward": (i32, [C.POINTER(ConvDesc), vp]),
    "csbsr_packed_weight_elems_tp": (i64, [i32, i32]),
    "csbsr_pack_weights_tp": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_packed_weight_elems_hr": (i64, [i32, i32, i32]),
    "csbsr_pack_weights_hr": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_packed_weight_elems": (i64, [i32] * 9),
    "csbsr_pack_weights": (i32, [vp, vp] + [i32] * 12 + [vp]),
    "csbsr_packed_weight_elems_split": (i64, [i32] * 9),
    "csbsr_pack_weights_split": (i32, [vp, vp] + [i32] * 11 + [f32, i32, vp]),
    "csbsr_axpby_split": (i32, [i64, i32, vp, i64, i64, f32, vp, i64, i64, f32, vp, i64, i64, vp]),
    "csbsr_nchw32_to_nhwc16_split": (i32, [vp, vp, i32, i32, i32, i32, i32, i64, i64, vp, vp, vp]),
    "csbsr_maxpool3x3s2_fwd_split": (i32, [vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, vp]),
    "csbsr_maxpool3x3s2_bwd_split": (i32, [vp, i64, i64, vp, i64, i64, vp, vp, i32, i32, i32, i32, vp]),
    "csbsr_adaptive_avgpool_fwd_split": (i32, [vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_sum_act_split": (i32, [i64, i32, i32, vp, vp, vp, vp, i64, i64, i32, vp]),
    "csbsr_weighted_pool_fwd_split": (i32, [vp, i64, i64, vp, vp, i32, i64, i32, vp]),
    "csbsr_bilinear_fwd_split": (i32, [vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "csbsr_unpack_wgrad": (i32, [vp, vp] + [i32] * 9 + [f32, i32, i32, vp]),
    "csbsr_wgrad_splits": (i32, [i32, i32, i64]),
    "csbsr_wgrad_splits_desc": (i32, [vp]),
    "csbsr_epilogue_backward": (i32, [C.POINTER(EpiBwdDesc), vp]),
    "csbsr_set_reduction_scratch": (i32, [vp, i64]),
    "csbsr_axpby": (i32, [i64, i32, vp, i64, f32, vp, i64, f32, vp, i64, vp]),
    "csbsr_fill_f16": (i32, [vp, i64, i32, i64, f32, vp]),
    "csbsr_sum_act": (i32, [i64, i32, i32, vp, vp, vp, i64, i32, vp]),
    "csbsr_weighted_pool_fwd": (i32, [vp, i64, vp, vp, i32, i64, i32, vp]),
    "csbsr_weighted_pool_bwd": (i32, [vp, i64, vp, vp, vp, i64, vp, i32, i64, i32, vp]),
    "csbsr_nchw32_to_nhwc16": (i32, [vp, vp, i32, i32, i32, i32, i32, i64, vp, vp, vp]),
    "csbsr_nhwc16_to_nchw32": (i32, [vp, i64, vp, i32, i32, i32, i32, f32, f32, vp]),
    "csbsr_plane_reduce": (i32, [vp, vp, i32, i64, vp, vp]),
    "csbsr_channel_mean_sub": (i32, [vp, i64, i64, i64, i32, i32, i32, i32, i32, vp, vp]),
    "csbsr_thin_tp_backward_slabs": (i32, [i32, i32]),
    "csbsr_conv_thin_dact_eligible": (i32, [vp]),
    "csbsr_conv_split_fused_eligible": (i32, [C.POINTER(ConvDesc)]),
    "csbsr_thin_tp_backward": (i32, [vp, i64, i64, i64, vp, i64, i64, i64, vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, i64, i64, i64, vp, vp, vp]),
    "csbsr_round_weights": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "csbsr_dc_bias": (i32, [vp, i32, i32, vp, i64, i32, vp, i64, i32, vp, i32, vp, vp]),
    "csbsr_instnorm_bwd": (i32, [vp, i64, vp, vp, vp, vp, i32, i32, i32, i64, vp, vp]),
    "csbsr_border_class_fill": (i32, [vp, vp, i64, i32, i32, i32, i32, vp]),
    "csbsr_border_class_fill_masked": (i32, [vp, vp, i64, vp, i64, C.c_float, i32, i32, i32, i32, vp]),
    "csbsr_border_class_sums": (i32, [vp, i64, vp, i32, i32, i32, i32, vp]),
    "csbsr_border_class_sums_prelu": (i32, [vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, vp]),
    "csbsr_ring_class_sums": (i32, [vp, i64, vp, i32, i32, i32, i32, vp]),
    "csbsr_bn_finalize": (i32, [vp, i64, i32, i32, f32, f32, vp, vp, vp, vp, vp]),
    "csbsr_bn_apply": (i32, [C.POINTER(BnDesc), vp]),
    "csbsr_bn_backward": (i32, [C.POINTER(BnDesc), vp]),
    "csbsr_maxpool3x3s2_fwd": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "csbsr_maxpool3x3s2_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "csbsr_adaptive_avgpool_fwd": (i32, [vp, i64, vp, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_adaptive_avgpool_bwd": (i32, [vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_bilinear_fwd": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "csbsr_bilinear_bwd": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "csbsr_bilinear32_fwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_bilinear32_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_bicubic_up_add": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "csbsr_aa_bicubic_down_fwd": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "csbsr_aa_bicubic_down_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_blur_fwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i64, vp]),
    "csbsr_blur_bwd_input": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_blur_bwd_kernel": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "csbsr_sdf": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "csbsr_segloss_reduce": (i32, [vp, vp, vp, i32, i64, vp, f32, f32, vp]),
    "csbsr_segloss_finish": (i32, [vp, vp, vp, i32, i64, vp, f32, f32, f32, f32, f32, f32, vp, vp, vp, i32, vp]),
    "csbsr_l1_fwd_bwd": (i32, [vp, vp, vp, i32, i32, i64, vp, f32, vp, vp, i32, vp]),
    "csbsr_sigmoid_bwd_to_nhwc8": (i32, [vp, vp, vp, i64, f32, vp]),
    "csbsr_gaussian_kernels": (i32, [vp, vp, i32, i32, vp]),
    "csbsr_iou_sweep": (i32, [vp, vp, vp, i32, i64, i32, f32, vp, vp, vp, vp, vp]),
    "csbsr_psnr_ssim": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "csbsr_head1_fwd": (i32, [vp, i64, i64, i32, vp, vp, i32, vp, i64, vp]),
    "csbsr_head1_bwd_input": (i32, [vp, i64, vp, i32, vp, i64, i64, vp]),
    "csbsr_adam_step": (i32, [vp, vp, vp, i32, C.c_double, C.c_double, C.c_float, vp]),
}

# private hooks (csbsr_amd/csrc/csbsr_debug.h): kernel selection for A/B timing and kernel attribution for bench.py
DEBUG_SIGNATURES = {
    "csbsr_debug_set_wgrad_tr": (None, [i32]),
    "csbsr_debug_set_conv_glds": (None, [i32]),
    "csbsr_debug_set_conv_tp": (None, [i32]),
    "csbsr_debug_set_conv_x3": (None, [i32]),
    "csbsr_debug_set_conv_x3w": (None, [i32]),
    "csbsr_debug_set_conv_x3n": (None, [i32]),
    "csbsr_debug_set_wgrad_hr": (None, [i32]),
    "csbsr_debug_last_conv_kernel": (i32, []),
    "csbsr_debug_last_wgrad_kernel": (i32, []),
    "csbsr_debug_cu_trace": (i32, [vp, i32, i32, vp]),
    "csbsr_debug_stream_create_cu_mask": (i32, [C.POINTER(vp), C.POINTER(C.c_uint32), i32]),
    "csbsr_debug_stream_destroy": (i32, [vp]),
    "csbsr_debug_stream_set_cu_budget": (i32, [vp, i32]),
    "csbsr_debug_stream_cu_budget": (i32, [vp]),
}

_lib = None


class CsbsrHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it is not built -- there is no CPU / torch fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CsbsrHipError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            f"(or `make -C csbsr_amd/csrc`).  csbsr_amd has no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for table in (SIGNATURES, DEBUG_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
    _lib = lib
    if os.environ.get("CSBSR_WGRAD_DBG"):          # A/B hook: bit0 transpose reads, 2 no thin, 4 no tap order, 8 no flat grid, 16 flat everywhere
        lib.csbsr_debug_set_wgrad_tr(int(os.environ["CSBSR_WGRAD_DBG"]))
    if os.environ.get("CSBSR_CONV_X3"):            # A/B hook: 0 off, 1 default, 2 every eligible launch
        lib.csbsr_debug_set_conv_x3(int(os.environ["CSBSR_CONV_X3"]))
    if os.environ.get("CSBSR_CONV_X3W"):           # A/B hook: 0 off, 1 default, 2 every eligible launch (+ 8 x min input channels / 32)
        lib.csbsr_debug_set_conv_x3w(1 if os.environ["CSBSR_CONV_X3W"] == "all" else int(os.environ["CSBSR_CONV_X3W"]))
    if os.environ.get("CSBSR_CONV_X3N"):           # A/B hook: 0 off, 1 default, 2 every eligible launch
        lib.csbsr_debug_set_conv_x3n(int(os.environ["CSBSR_CONV_X3N"]))
    if os.environ.get("CSBSR_WGRAD_HR"):           # A/B hook: 0 off, 1 default, 2 every eligible launch
        lib.csbsr_debug_set_wgrad_hr(int(os.environ["CSBSR_WGRAD_HR"]))
    if os.environ.get("CSBSR_CONV_TP"):            # A/B hook: 0 off, 1 default, 2 every eligible launch
        lib.csbsr_debug_set_conv_tp(int(os.environ["CSBSR_CONV_TP"]))
    if os.environ.get("CSBSR_CONV_GLDS"):          # A/B hook for kernel selection experiments
        lib.csbsr_debug_set_conv_glds(int(os.environ["CSBSR_CONV_GLDS"]))
    return lib


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise CsbsrHipError(f"{name} failed ({rc}): {lib.csbsr_last_error().decode()}")
