/*
 * csbsr_hip.h -- C ABI of libcsbsr_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the CSBSR joint
 * blind-SR + segmentation training hot path.
 *
 * The reference (Yuki-11/CSBSR) has no FFI / plugin registry: its "operator API" for this path is the set
 * of torch.nn.functional calls made by JointModelWithLoss.forward + autograd
 * (/root/reference/model/modeling/build_model.py:370-416).  Each entry point below replaces one family of
 * those calls; the cited lines are the reference call sites it stands in for.  The Python host
 * (csbsr_amd/) binds these with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - raw device pointers + explicit sizes/strides (in ELEMENTS), a hipStream_t passed as void*;
 *  - feature maps are fp16, channels-last (N,H,W,C) with C padded to a multiple of 8 and an explicit pixel
 *    stride ("ld") so channel slices of a wider buffer can be read / written in place (concat-free);
 *  - 3-channel images, probability maps, losses and everything that feeds a log / division / small difference
 *    are fp32 planar NCHW -- the layout the reference's loader delivers and its callers read back;
 *  - every function is asynchronous on the given stream, never allocates, never synchronises, never throws;
 *  - return 0 on success, non-zero on a bad argument / launch failure (text via csbsr_last_error()).
 */
#ifndef CSBSR_HIP_H
#define CSBSR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* csbsr_stream_t; /* hipStream_t */

int csbsr_version(void);
const char* csbsr_last_error(void);

/* ------------------------------------------------------------------------------------------- convolution */
enum { CSBSR_ACT_NONE = 0, CSBSR_ACT_RELU = 1, CSBSR_ACT_LRELU = 2, CSBSR_ACT_PRELU = 3, CSBSR_ACT_SIGMOID = 4 };
/* epilogue combine:  out = act(conv+bias)  (+ res | - res | * res | + res*res2) */
enum { CSBSR_RES_NONE = 0, CSBSR_RES_ADD = 1, CSBSR_RES_SUB = 2, CSBSR_RES_MUL = 3, CSBSR_RES_FMA = 4 };
/* per-channel reductions fused into the conv epilogue (computed on act(conv+bias), before res) */
enum { CSBSR_STAT_NONE = 0, CSBSR_STAT_BN = 1 /* sum, sumsq per channel: fp32[2][coutp] */,
       CSBSR_STAT_SAMPLE_SUM = 2 /* sum per (sample, channel): fp32[N][coutp]  (global average pool) */ };

/* One input segment of a convolution: NHWC fp16 view with element strides.  A spatially constant operand
 * (the blur-kernel code expanded over the image: kbpn.py:405,513,565-567) is passed with sy = sx = 0. */
typedef struct {
  const void* ptr;
  int64_t sn, sy, sx; /* element strides of sample / row / pixel */
  int32_t c;          /* channels in this segment, multiple of 8 (zero-padded) */
  int32_t creal;      /* real (unpadded) channel count, 0 = unknown: lets 3-channel images take the dense-K kernel */
} csbsr_seg_t;

/* Implicit-GEMM convolution / transposed convolution, MFMA fp16 -> fp32 accumulate.
 * Replaces F.conv2d (kbpn.py:241 via ConvBlock, kbpn.py:513-517, extractors.py:36-38, pspnet.py:30-41,47,72-86)
 * and F.conv_transpose2d (kbpn.py:273-277) and, with re-packed weights, their autograd dgrads. */
typedef struct {
  csbsr_seg_t in[2];       /* input = channel-concat of up to two segments (in[1].c == 0: unused) */
  int32_t N, H, W;         /* input spatial size */
  int32_t OH, OW;          /* output spatial size */
  int32_t transposed;      /* 0: conv   1: transposed conv (gather form, one phase per output residue) */
  int32_t KH, KW;          /* full kernel size */
  int32_t stride, pad, dil;
  const void* wt;          /* packed fp16 weights from csbsr_pack_weights */
  int32_t cout;            /* real output channels */
  int32_t coutp;           /* padded (multiple of 8) channel count of out16 / res / res2 */
  void* out16;             /* fp16 NHWC output or NULL */
  int64_t o_sn, o_sy, o_sx;
  float* out32;            /* optional fp32 output with arbitrary strides (planar NCHW: sc = H*W) or NULL */
  int64_t o32_sn, o32_sy, o32_sx, o32_sc;
  const float* bias;       /* fp32[cout] or NULL */
  const float* cbias;      /* fp32 [N][16 | 25][coutp] or NULL: extra bias per (sample, position class of the output pixel) = the exact
                              contribution of a folded input segment that is constant within each class (cbias_mode below;
                              see csbsr_border_class_fill / csbsr_ring_class_sums) */
  int32_t act;             /* CSBSR_ACT_* */
  float act_slope;         /* LRELU slope */
  const float* prelu;      /* device scalar for CSBSR_ACT_PRELU */
  int32_t res_mode;        /* CSBSR_RES_* */
  const void* res;         /* fp16 NHWC */
  int64_t r_sn, r_sy, r_sx;
  const void* res2;        /* second operand of CSBSR_RES_FMA */
  int64_t r2_sn, r2_sy, r2_sx;
  int32_t accumulate;      /* 1: out += result (gradient fan-in) */
  int32_t stat_mode;       /* CSBSR_STAT_* */
  float* stat;
  float out_scale;         /* accumulator multiplied by this first (1.0 default) */
  /* split-fp16 operands (detector precision mode): element offset from the hi plane of out16 / res / res2 to a lo plane of the
   * same strides, value = hi + lo (~22 mantissa bits); 0 = plain fp16.  A split INPUT needs no field: it is passed as
   * in[0] = the [hi | lo] channel pair (2c channels), in[1] = the hi plane again, with weights from csbsr_pack_weights_split. */
  int64_t o_lo, r_lo, r2_lo;
  /* dgrad launches: activation-derivative mask fused into the epilogue.  The result (after residual / accumulate) is multiplied by
   * (mask > 0 ? 1 : mask_slope) where mask = the saved forward OUTPUT of the layer whose input gradient this launch produces
   * (ReLU: slope 0; LeakyReLU / PReLU: the slope): autograd of F.relu / F.leaky_relu (kbpn.py:236-247) without a pass of its own.
   * Only valid on the launch that completes the gradient (the last accumulating contribution).  NULL = off. */
  const void* mask; int64_t m_sn, m_sy, m_sx; float mask_slope;
  int32_t cbias_mode;      /* classes of cbias: 0 = the 16 border classes (y==0)*8 + (y==OH-1)*4 + (x==0)*2 + (x==OW-1);
                              1 = the 25 two-ring classes ty*5 + tx, t = 0,1,2,3,4 for coordinate 0, 1, interior, size-2, size-1
                              (needs OH, OW >= 5): what a map takes after TWO zero-padded 3x3 convs of a constant */
  /* mask_prelu: the masking layer's PReLU slope on the device (overrides mask_slope; every kernel that takes a mask except csbsr_conv_hr_forward).
   * csbsr_conv_tp_forward only (NULL elsewhere): where the masking layer's
   * bias / PReLU-slope gradients go -- dact_bias[c] += sum over pixels of the masked result, dact_prelu[0] += sum over the pixels
   * with mask <= 0 of (unmasked result) x mask / slope, i.e. exactly what csbsr_epilogue_backward adds for that layer
   * (kbpn.py:230-262: the PReLU of a DeconvBlock / ConvBlock whose input gradient this launch completes) */
  const float* mask_prelu; float* dact_bias; float* dact_prelu;
  /* csbsr_conv_tp_forward only: the masking layer had a residual (out = act(pre) +- res: res / res_mode describe IT, not this launch):
   * its activation is rebuilt as mask -+ res, and dres (fp16 NHWC, same geometry as out16) receives d(res) = +- the unmasked result
   * (kbpn.py:254-256, DownBlock: l1 = down_conv3(h0 - x)) */
  void* dres; int64_t dr_sn, dr_sy, dr_sx;
  /* (mask_prelu / dact_prelu / dres are also taken by csbsr_conv_forward for ONE launch shape, csbsr_conv_thin_dact_eligible: the
   * accumulating 3x3 dgrad from a <= 3-channel image gradient that completes the output gradient of a PReLU + residual layer --
   * kb.sr_reconst's dgrad completing up_conv3's dOut, kbpn.py:385-392,460-468: out16 (accumulate) receives dPre, dres the unmasked total) */
  /* csbsr_conv_forward only.  1: FUSED split-fp16 input -- in[0] = the [hi | lo] channel pair (2c channels, c >= 32), in[1]
   * unused, weights from csbsr_pack_weights_split layout 3 ([w_hi | w_lo] per 32-channel slice).  One staged K slice then holds 32
   * channels of x_hi and x_lo against the same 32 of w_hi and w_lo and feeds all three products (x_hi w_hi + x_lo w_hi + x_hi w_lo)
   * from it: 2/3 of the operand traffic of the three-block form above for the same arithmetic.  LDS-DMA kernels only (> 32 padded
   * output channels); the call fails otherwise.  2: the same fused stage WITHOUT the x_hi w_lo product ([x_hi | x_lo] w_hi, the two-product
   * precision plan of a layer that keeps its weights' fp16 rounding; the w_lo halves of the layout-3 operand are not read). */
  int32_t split_fused; int32_t _pad_sf;
  /* csbsr_conv_forward / csbsr_conv_x3_forward: element stride between the samples' rows of ``bias`` (0 = one fp32[cout] row for the whole
   * batch, the reference's bias).  Non-zero = a per-sample bias [N][bias_sn]: the host's compensation of the forward weights' fp16
   * rounding (csbsr_amd/engine.py Conv._dc_bias).  Needs a non-transposed layer whose samples are whole pixel tiles (OH * OW a multiple
   * of 256); the call fails otherwise. */
  int64_t bias_sn;
} csbsr_conv_desc_t;

int csbsr_conv_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s);

/* Direct 3x3 convolution for the 32 / 49-channel full-resolution layers of the kernel predictor (kbpn.py:521-578) and their dgrads:
 * halo tile in LDS, weights in registers (packed in MFMA-fragment order by csbsr_pack_weights_hr), see csrc/conv_hr.hip.  Same
 * descriptor as csbsr_conv_forward; csbsr_conv_hr_eligible says whether a launch qualifies (3x3 / pad 1 or 1x1 / pad 0, stride 1, one plain fp16
 * segment of 32 or 56 padded channels, <= 64 couts, ReLU / LeakyReLU / none, optional mask and per-sample sums, nothing else fused).
 * pack: kind 0 = forward conv (W OIHW, rows = D0), kind 1 = dgrad of a stride-1 conv (rows = D1, taps flipped). */
int32_t csbsr_conv_hr_eligible(const csbsr_conv_desc_t* d);
int csbsr_conv_hr_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s);
/* ksize = 3 (pad 1) or 1 (pad 0); kind 0 forward (w = OIHW), 1 dgrad of a stride-1 conv (rows = the conv's input channels, taps flipped) */
int64_t csbsr_packed_weight_elems_hr(int32_t ksize, int32_t c_real, int32_t rows_real);
int csbsr_pack_weights_hr(const float* w, void* dst, int32_t kind, int32_t ksize, int32_t D0, int32_t D1, int32_t c_real, int32_t rows_real,
                          int32_t row_off, int32_t k_off, csbsr_stream_t s);

/* Phase-decomposed transposed convolution (ConvTranspose2d with stride < kernel <= 2 stride: the 8x8 stride-4 / 12x12 stride-8
 * up-projections of /root/reference/model/modeling/kbpn.py:230-262 via DeconvBlock, and the dgrads of the matching strided Conv2d
 * layers that autograd runs for kbpn.py:230-262): the low-resolution halo tile stays in LDS for all phases and taps, only the
 * fragment-ordered weights (csbsr_pack_weights_tp) stream through a 4-stage LDS ring, register epilogue -- see csrc/conv_tp.hip.
 * Same descriptor as csbsr_conv_forward except d->wt, which must come from csbsr_pack_weights_tp; csbsr_conv_tp_eligible says
 * whether a launch qualifies (128 padded input channels in one plain fp16 segment, 72..128 padded output channels, output exactly
 * stride x input, bias / ReLU / leaky / PReLU / residual add or subtract / accumulate / activation mask). */
int32_t csbsr_conv_tp_eligible(const csbsr_conv_desc_t* d);
int csbsr_conv_tp_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s);
/* w indexed [contracted channel][row][kh][kw] (IOHW of ConvTranspose2d, or OIHW of a Conv2d seen from its dgrad); packs rows
 * [row_off, row_off + rows_real) against contracted channels [k_off, k_off + c_real). */
int64_t csbsr_packed_weight_elems_tp(int32_t stride, int32_t c_real);
int csbsr_pack_weights_tp(const float* w, void* dst, int32_t D0, int32_t D1, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                          int32_t c_real, int32_t rows_real, int32_t row_off, int32_t k_off, csbsr_stream_t s);

/* 3x3 stride-1 convolution of the wide low-resolution layers (the SFT scale / shift convolutions F.conv2d(cat(fea, k), w, b, 1, 1) of
 * /root/reference/model/modeling/kbpn.py:505-520 and their dgrads): per-chunk halo tile in LDS shared by the nine taps, fragment-ordered
 * weights (csbsr_pack_weights_x3) straight from L2 into registers, one barrier per nine K steps -- see csrc/conv_x3.hip.  Same
 * descriptor and fused epilogue as csbsr_conv_forward except d->wt; csbsr_conv_x3_eligible says whether a launch qualifies (one plain
 * fp16 input segment of >= 128 channels padded to a multiple of 64, >= 72 padded output channels, no statistics / fp32 output / split). */
int32_t csbsr_conv_x3_eligible(const csbsr_conv_desc_t* d);
int csbsr_conv_x3_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s);
/* kind 0 forward (w = OIHW), 1 dgrad of the stride-1 conv (rows = its input channels, taps flipped) */
int64_t csbsr_packed_weight_elems_x3(int32_t c_real, int32_t rows_real);
int csbsr_pack_weights_x3(const float* w, void* dst, int32_t kind, int32_t D0, int32_t D1, int32_t c_real, int32_t rows_real,
                          int32_t row_off, int32_t k_off, csbsr_stream_t s);
/* The same kernel family also takes the k = 2 x stride strided convolutions -- the 8x8 stride-4 Conv2d's of DownBlock / UpBlock
 * (kbpn.py:215-262) and the dgrads of the 8x8 stride-4 ConvTranspose2d's: a chunk is then (input phase, 64 channels), i.e. the
 * sub-sampled grid every kernel offset with the same residue modulo the stride reads, shared by its 2 x 2 taps (csbsr_conv_x3_eligible
 * / csbsr_conv_x3_forward accept both shapes; stride <= 4, pad < stride, 64..512 padded input channels in whole chunks).  Weights:
 * w is [row][contracted channel][kh][kw] with dims D0 x D1 -- a Conv2d's OIHW parameter, or a ConvTranspose2d's IOHW parameter seen
 * from its dgrad (rows = its input channels). */
int64_t csbsr_packed_weight_elems_x3_strided(int32_t stride, int32_t c_real, int32_t rows_real);
int csbsr_pack_weights_x3_strided(const float* w, void* dst, int32_t D0, int32_t D1, int32_t ksize, int32_t stride, int32_t c_real,
                                  int32_t rows_real, int32_t row_off, int32_t k_off, csbsr_stream_t s);

/* 3x3 stride-1 convolution from many input channels into <= 64 output channels at full resolution: the second convolutions of
 * PSPNet_BlurSkip's SFT-like blocks, F.conv2d(505 -> 64) at 1792^2 (/root/reference/model/modeling/blocks.py:105-120, pspnet.py:176-190), and
 * the dgrads of their 64 -> 505 first convolutions: 8-row x 64-pixel x 64-cout tile, per-32-channel halo tile in LDS shared by the nine
 * taps, fragment-ordered weights from L2 -- see csrc/conv_x3n.hip.  Same descriptor and fused epilogue as csbsr_conv_forward except d->wt;
 * a split [hi | lo] input is presented as ONE 2 x Cp-channel segment with split_fused = 2 (the two-product plan [x_hi | x_lo] w_hi; the
 * pack repeats the weights for the lo plane), the output may be a hi + lo pair.
 * The WIDE form of the same kernel (8 x 32 pixels x 128 couts, two workgroups per CU) takes the layers with MORE than 64 padded output
 * channels from 32 input channels up -- the 64 -> 505 first convolutions of those blocks, KBPN's SFT convolutions
 * (kbpn.py:505-520) with the constant segment folded into a class bias, the 32 -> 128 gather of a stage's thin dgrads -- same entry
 * points, same pack (its layout follows the padded row count).
 * csbsr_conv_x3n_eligible returns 0 (not taken), 1 (narrow form: one input segment of >= 64 channels in whole 32-channel chunks, 33 .. 64
 * padded output channels; BatchNorm sums allowed when no per-pixel epilogue operand is fused) or 2 (wide form: > 64 padded output
 * channels, >= 32 input channels -- above 384 only launches without a sigmoid / multiply epilogue --, no statistics); neither takes the fp32
 * side output or the fused epilogue-backward sums. */
int32_t csbsr_conv_x3n_eligible(const csbsr_conv_desc_t* d);
int csbsr_conv_x3n_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s);
/* kind 0 forward (w = OIHW), 1 dgrad of the stride-1 conv; in_ch = padded input channels the kernel walks (2 x plane for a split input),
 * plane = padded channels of one plane of a split input (0: plain input), scale = factor folded into the fp16 weights (the split path's
 * 256, undone by csbsr_conv_desc_t::out_scale) */
int64_t csbsr_packed_weight_elems_x3n(int32_t in_ch, int32_t rows_real);
int csbsr_pack_weights_x3n(const float* w, void* dst, int32_t kind, int32_t D0, int32_t D1, int32_t c_real, int32_t rows_real,
                           int32_t row_off, int32_t k_off, int32_t in_ch, int32_t plane, float scale, csbsr_stream_t s);

/* Winograd F(2, 3) along x for the same wide 3x3 stride-1 layers (the SFT convolutions F.conv2d(cat(fea, k), w, b, 1, 1) of
 * /root/reference/model/modeling/kbpn.py:505-520 and their dgrads): two neighbouring output pixels from four products per (channel, row tap)
 * instead of six -- 2/3 of the MFMA work of csbsr_conv_x3_forward; input transform on the fly (registers -> LDS), weights transformed by the
 * pack (U = G w, rounded to fp16 once), output transform on the accumulators, same descriptor and fused epilogue -- see csrc/conv_x3w.hip.
 * The transformed operands are fp16: results differ from the direct product at the fp16 rounding level of the layer's inputs
 * (tests/study_winograd.py; tests/test_conv_kernels_gpu.py holds both to the same bound).  Measured at parity with csbsr_conv_x3_forward
 * (the CU's vector-memory path bounds it, csrc/conv_x3w.hip): off unless enabled (csbsr_conv_x3w_eligible then returns 0).  Eligible: one plain fp16 input segment of >= 128
 * channels padded to a multiple of 32, >= 72 padded output channels, no statistics / fp32 output / split operands. */
int32_t csbsr_conv_x3w_eligible(const csbsr_conv_desc_t* d);
int csbsr_conv_x3w_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s);
/* kind 0 forward (w = OIHW), 1 dgrad of the stride-1 conv (rows = its input channels, taps flipped) */
int64_t csbsr_packed_weight_elems_x3w(int32_t c_real, int32_t rows_real);
int csbsr_pack_weights_x3w(const float* w, void* dst, int32_t kind, int32_t D0, int32_t D1, int32_t c_real, int32_t rows_real,
                           int32_t row_off, int32_t k_off, csbsr_stream_t s);

/* Weight-gradient GEMM: G[split][a][tap][b] = sum over the split's pixels of A[pix][a] * B[pix @ tap][b]  (fp32; the pixel
 * range is cut into csbsr_wgrad_splits() slabs, each written once -- no atomics, no zero-fill; csbsr_unpack_wgrad sums them).
 * Conv: A = dPre (output side), B = input.  Transposed conv: A = its input (LR side), B = dOut (HR side).
 * Autograd wgrad of the calls above. */
typedef struct {
  const void* a;           /* fp16 NHWC [N, AH, AW, ca] ungathered side */
  int64_t a_sn, a_sy, a_sx;
  int32_t ca;              /* channels (multiple of 8) */
  int32_t ca_real;         /* real (unpadded) channels of A, 0 = ca: lets 3-channel heads take the taps-in-rows kernel */
  csbsr_seg_t b[2];        /* gathered side, up to two segments */
  int32_t N, AH, AW;       /* grid the reduction runs over */
  int32_t BH, BW;          /* spatial size of the gathered side */
  int32_t KH, KW, stride, pad, dil;
  float* g;                /* fp32 [splits][ca][KH*KW*(cb0+cb1)] */
  int32_t splits;          /* = csbsr_wgrad_splits(ca, KH*KW*(cb0+cb1), N*AH*AW) */
  int32_t _pad1;
} csbsr_wgrad_desc_t;

int32_t csbsr_wgrad_splits(int32_t ca, int32_t ktot, int64_t npix);
int32_t csbsr_wgrad_splits_desc(const csbsr_wgrad_desc_t* d);   /* same, for a filled descriptor (d->splits ignored): use this one */
int csbsr_conv_wgrad(const csbsr_wgrad_desc_t* d, csbsr_stream_t s);

/* fp32 master weights W[D0][D1][KH][KW] (the reference's OIHW conv / IOHW deconv parameters, whose state_dict
 * layout is part of the drop-in boundary) -> packed fp16 operand [phase][rows_p][Kp] of csbsr_conv_forward.
 *  kind 0  rows = D0, contracted channels = D1:            forward conv (W is OIHW) and dgrad of a transposed conv (W is IOHW)
 *  kind 1  rows = D1, contracted = D0, taps flipped:       dgrad of a stride-1 conv
 *  kind 2  rows = D1, contracted = D0, phase decomposed:   forward transposed conv (IOHW) and dgrad of a strided conv (OIHW)
 * seg0_real + seg1_real = size of the contracted dim (two-segment inputs, each zero-padded to a multiple of 8);
 * rows [row_off, row_off+nrows) of the row dim are emitted (dgrad wrt one input segment). */
int64_t csbsr_packed_weight_elems(int32_t kind, int32_t D0, int32_t D1, int32_t KH, int32_t KW, int32_t stride,
                                  int32_t seg0_real, int32_t seg1_real, int32_t nrows);
int csbsr_pack_weights(const float* w, void* dst, int32_t kind, int32_t D0, int32_t D1, int32_t KH, int32_t KW,
                       int32_t stride, int32_t pad, int32_t seg0_real, int32_t seg1_real, int32_t row_off,
                       int32_t nrows, int32_t k_off, csbsr_stream_t s);   /* k_off: first contracted channel (sub-range packing) */
/* Split-fp16 ("hi + lo") operand for the detector precision mode: per tap the K axis holds three blocks of pad8(creal) channels,
 * [w_hi | w_hi | w_lo] with w_hi = fp16(w * wscale), w_lo = fp16(w * wscale - w_hi), matching an input passed to
 * csbsr_conv_forward as in[0] = [x_hi | x_lo], in[1] = x_hi:  x_hi w_hi + x_lo w_hi + x_hi w_lo in ONE fp32 accumulator (three
 * MFMA passes; the dropped x_lo w_lo term is ~2^-22 relative).  wscale (a power of two, undone by the conv's out_scale) keeps w_lo
 * out of fp16's subnormal range.  Single-segment layers only. */
/* layout 0: the three-block forward operand above.  layout 1: two blocks [w_hi | w_lo] for the dgrads of that mode, whose input (a
 * plain fp16 activation gradient) is passed twice, in[0] = in[1] = dY.  layout 2: two blocks [w_hi | w_hi] against in[0] = [x_hi | x_lo]
 * alone (a layer whose precision plan keeps the activation's ~22 bits but not the weight's).  layout 3: the fused form of layout 0
 * (csbsr_conv_desc_t::split_fused): the flat (tap, channel) K index cut into 32-wide slices, each stored as [w_hi (32) | w_lo (32)]. */
int64_t csbsr_packed_weight_elems_split(int32_t kind, int32_t D0, int32_t D1, int32_t KH, int32_t KW, int32_t stride,
                                        int32_t creal, int32_t nrows, int32_t layout);
int csbsr_pack_weights_split(const float* w, void* dst, int32_t kind, int32_t D0, int32_t D1, int32_t KH, int32_t KW,
                             int32_t stride, int32_t pad, int32_t creal, int32_t row_off, int32_t nrows, int32_t k_off,
                             float wscale, int32_t layout, csbsr_stream_t s);
/* packed fp32 wgrad slabs G[split][ca_padded][tap][b(padded segments)] -> grad[a][b_off + b][kh][kw] += scale * sum_split G
 * (grad is [D0][D1][KH][KW]; transpose_ab bit 0: a indexes D1 and b indexes D0; bit 1: taps mirrored, (kh, kw) -> (KH-1-kh, KW-1-kw) --
 * the two together unpack the slab of a stride-1 conv's mirrored wgrad problem, A = input, B = dOut) */
int csbsr_unpack_wgrad(const float* g, float* grad, int32_t A, int32_t KH, int32_t KW, int32_t seg0_real,
                       int32_t seg1_real, int32_t D0, int32_t D1, int32_t transpose_ab, int32_t b_off, float scale,
                       int32_t splits, int32_t ca_padded, csbsr_stream_t s);

/* ------------------------------------------------------------------------------------------- elementwise */
/* Backward of the conv epilogue out = act(pre) (+|-|*|fma) res: writes dPre, optionally dRes / dRes2, the bias
 * gradient (column sums of dPre) and the PReLU slope gradient.  Autograd of F.prelu / relu / leaky_relu /
 * sigmoid / add / sub / mul (kbpn.py:236-247, 459-469, 480-489, 513-518). */
typedef struct {
  int64_t npix;
  int32_t c, creal;
  const void* dout; int64_t dout_ld;
  const void* out;  int64_t out_ld;    /* saved epilogue output (post res) */
  const void* res;  int64_t res_ld;
  const void* res2; int64_t res2_ld;
  int32_t act; float act_slope; const float* prelu;
  int32_t res_mode; int32_t _pad;
  void* dpre; int64_t dpre_ld;         /* may alias dout */
  void* dres; int64_t dres_ld; int32_t dres_accumulate; int32_t _pad1;
  void* dres2; int64_t dres2_ld; int32_t dres2_accumulate; int32_t _pad2;
  float* dbias;                        /* fp32[c] += or NULL */
  float* dprelu;                       /* fp32 scalar += or NULL */
} csbsr_epi_bwd_desc_t;
int csbsr_epilogue_backward(const csbsr_epi_bwd_desc_t* d, csbsr_stream_t s);
/* Caller-owned fp32 scratch for the library's reductions.  EVERY floating-point sum of the path (conv-fused BatchNorm / per-sample
 * statistics, bias / PReLU / BatchNorm gradient sums, loss sums, border-class sums, blur-kernel gradients, pooling) is order-fixed:
 * the producing kernel writes one partial row per workgroup (or pixel tile) here and a fold kernel adds the rows in a fixed tree --
 * no fp32 atomics anywhere, so two runs on the same inputs are bit-identical (the reference's CPU path is too).  Registered PER
 * DEVICE: the call binds ``buf`` to the calling thread's current HIP device and the reducing entry points look their device's buffer
 * up at launch, so several models / replicas in one process do not overwrite each other's registration.  Used by whatever stream
 * the calls are issued on: one stream at a time per device.  64 Mi floats cover every shape of the path at B = 8, HR 1792^2 (the
 * last 4 Mi are the second level of the fold).  Without a registered (or with too small a) buffer the reducing entry points FAIL
 * (status 1, csbsr_last_error): there is no atomics fallback. */
int csbsr_set_reduction_scratch(float* buf, int64_t elems);

/* Adam step (torch.optim.Adam as /root/reference/train.py:91 builds it: lr, betas = (0.9, 0.999), eps = 1e-8, no weight decay, no amsgrad) for a
 * LIST of fp32 tensors in one launch: m = lerp(m, g, 1 - beta1); v = beta2 v + (1 - beta2) g g; p -= step_size m / (sqrt(v) / bc2_sqrt + eps), with
 * step_size = lr / (1 - beta1^t) and bc2_sqrt = sqrt(1 - beta2^t) per tensor (a parameter whose gradient was None in some step has a smaller t,
 * exactly like torch's per-parameter state["step"]).  ``tensors`` (device memory): the table; ``block_tensor`` / ``block_chunk`` (device, int32
 * [nblocks]): workgroup b updates elements [8192 block_chunk[b], 8192 (block_chunk[b] + 1)) of tensor block_tensor[b]. */
typedef struct {
  float* p; const float* g; float* m; float* v;
  int64_t n;
  float step_size, bc2_sqrt;
} csbsr_adam_tensor_t;
int csbsr_adam_step(const csbsr_adam_tensor_t* tensors, const int32_t* block_tensor, const int32_t* block_chunk, int32_t nblocks,
                    double beta1, double beta2, float eps, csbsr_stream_t s);

int csbsr_axpby(int64_t npix, int32_t c, const void* x, int64_t x_ld, float a, const void* z, int64_t z_ld,
                float b, void* y, int64_t y_ld, csbsr_stream_t s);
int csbsr_fill_f16(void* p, int64_t npix, int32_t c, int64_t ld, float v, csbsr_stream_t s);

/* Split-fp16 ("hi + lo") variants for the detector precision mode (forward only: the backward kernels read the hi planes).
 * A split map is two fp16 planes of equal strides, value = hi + lo; ``*_lo`` is the element offset from the hi pointer to the lo
 * plane, 0 = that operand is plain fp16.  Arithmetic is fp32 on the combined value; outputs are re-split.  Same reference call sites
 * as the plain functions they extend. */
int csbsr_axpby_split(int64_t npix, int32_t c, const void* x, int64_t x_ld, int64_t x_lo, float a, const void* z, int64_t z_ld,
                      int64_t z_lo, float b, void* y, int64_t y_ld, int64_t y_lo, csbsr_stream_t s);
int csbsr_nchw32_to_nhwc16_split(const float* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t cp,
                                 int64_t dst_ld, int64_t dst_lo, const float* mean, const float* invstd, csbsr_stream_t s);
int csbsr_maxpool3x3s2_fwd_split(const void* x, int64_t x_ld, int64_t x_lo, void* y, int64_t y_ld, int64_t y_lo, int32_t N,
                                 int32_t H, int32_t W, int32_t c, csbsr_stream_t s);
/* (x, y: the saved forward input / output, possibly split; dy / dx plain fp16 dense) */
int csbsr_maxpool3x3s2_bwd_split(const void* x, int64_t x_ld, int64_t x_lo, const void* y, int64_t y_ld, int64_t y_lo,
                                 const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t c, csbsr_stream_t s);
int csbsr_adaptive_avgpool_fwd_split(const void* x, int64_t x_ld, int64_t x_lo, void* y, int64_t y_ld, int64_t y_lo, int32_t N,
                                     int32_t H, int32_t W, int32_t c, int32_t OH, int32_t OW, csbsr_stream_t s);
int csbsr_sum_act_split(int64_t npix, int32_t c, int32_t n, const void* const* xs, const int64_t* x_lds, const int64_t* x_los, void* y,
                        int64_t y_ld, int64_t y_lo, int32_t relu, csbsr_stream_t s);
int csbsr_weighted_pool_fwd_split(const void* x, int64_t x_ld, int64_t x_lo, const float* w, float* out, int32_t N, int64_t hw,
                                  int32_t c, csbsr_stream_t s);
int csbsr_bilinear_fwd_split(const void* x, int64_t x_ld, int64_t x_lo, void* y, int64_t y_ld, int64_t y_lo, int32_t N, int32_t H,
                             int32_t W, int32_t c, int32_t OH, int32_t OW, int32_t align_corners, const float* drop,
                             csbsr_stream_t s);

/* HRNet-W48 + OCR detector (BASELINE config 4).
 * csbsr_sum_act: y = relu?(x_0 + ... + x_{n-1}), n <= 4 fp16 NHWC maps -- the fuse sum of HighResolutionModule.forward
 * (model/modeling/hrnet_ocr/backbones/hrnet/hrnet_backbone.py:276-296).
 * csbsr_weighted_pool_{fwd,bwd}: out[n][c] = sum_p w[n][p] * x[n][p][c] and its adjoint (dx +=, dw =) -- the soft object-region
 * pooling torch.matmul(softmax(probs), feats) of SpatialGather_Module.forward (modules/spatial_ocr_block.py:58-66).  out is
 * accumulated into (zero it first); c <= 2048. */
int csbsr_sum_act(int64_t npix, int32_t c, int32_t n, const void* const* xs, const int64_t* x_lds, void* y, int64_t y_ld,
                  int32_t relu, csbsr_stream_t s);
int csbsr_weighted_pool_fwd(const void* x, int64_t x_ld, const float* w, float* out, int32_t N, int64_t hw, int32_t c,
                            csbsr_stream_t s);
int csbsr_weighted_pool_bwd(const void* x, int64_t x_ld, const float* w, const float* dout, void* dx, int64_t dx_ld, float* dw,
                            int32_t N, int64_t hw, int32_t c, csbsr_stream_t s);

/* boundary layout converters: fp32 NCHW (what the reference's loader hands over, crack_dataset.py:40-64;
 * optionally normalised per (n,c) -- InstanceNorm2d(3) apply, build_model.py:136) -> fp16 NHWC, and back */
int csbsr_nchw32_to_nhwc16(const float* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t cp,
                           int64_t dst_ld, const float* mean, const float* invstd, csbsr_stream_t s);
int csbsr_nhwc16_to_nchw32(const void* src, int64_t src_ld, float* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                           float alpha, float beta, csbsr_stream_t s);
/* out[plane] = {sum a, sum a*a (b NULL) | sum a*b} over fp32 planes (instance-norm statistics) */
int csbsr_plane_reduce(const float* a, const float* b, int32_t planes, int64_t hw, float* out, csbsr_stream_t s);
/* out[n][c] = mean over the pixels (step i, step j) of x[n, y, x, c]  (fp16 NHWC view with element strides sn / sy / sx, cp channels, a
 * multiple of 8; out [N][cp] fp32): the per-sample input statistic of the host's compensation of the forward weights' fp16 rounding
 * (csbsr_amd/engine.py Conv._dc_bias; nothing in the reference corresponds to it -- the reference multiplies fp32 weights).  Order-fixed. */
int csbsr_channel_mean_sub(const void* x, int64_t sn, int64_t sy, int64_t sx, int32_t N, int32_t H, int32_t W, int32_t cp,
                           int32_t step, float* out, csbsr_stream_t s);
/* How the plain-fp16 (KBPN) layers round their fp32 master weights [D0][D1][KH][KW] (F.conv2d / F.conv_transpose2d of the reference multiply
 * fp32 weights, kbpn.py:196-289; this build multiplies fp16 ones): wq = the fp16 value of every weight, held as fp32 (the csbsr_pack_weights*
 * calls then convert exactly), S (optional) [D0][D1] = sum over taps of (w - wq).  mode 0 = round to nearest; mode >= 1 = tap-sum-preserving:
 * per (d0, d1) and tap group -- all taps (mode 1: Conv2d of any stride) or the taps (ky % mode, kx % mode) that one output phase of a
 * stride-``mode`` ConvTranspose2d sees -- taps move to their other fp16 neighbour until the group's summed rounding residual is under half an
 * ulp: the residual filter has no response to locally constant input.  csbsr_dc_bias: out[n][o] = (bias ? bias[o] : 0) + sum_c S[o][c] *
 * mean[n][c] with the means of up to two input segments (c0 / c1 real channels, row strides m0_ld / m1_ld) -- the per-sample bias rows
 * handed to csbsr_conv_forward with bias_sn = cout, giving back what is left of the residual's response to the input's mean */
int csbsr_round_weights(const float* w, float* wq, float* S, int32_t D0, int32_t D1, int32_t KH, int32_t KW, int32_t mode, csbsr_stream_t s);
int csbsr_dc_bias(const float* S, int32_t cout, int32_t cin, const float* m0, int64_t m0_ld, int32_t c0, const float* m1, int64_t m1_ld,
                  int32_t c1, const float* bias, int32_t N, float* out, csbsr_stream_t s);
int csbsr_instnorm_bwd(const void* dy, int64_t dy_ld, const float* x, const float* mean, const float* invstd,
                       float* dx, int32_t accumulate, int32_t N, int32_t C, int64_t hw, float* red,
                       csbsr_stream_t s);

/* Constant-operand folding (exact): a zero-padded 3x3 conv of a spatially constant map (fe_kernel.0 applied to the
 * expanded kernel code, kbpn.py:565-567) takes one value per border class ((y==0)*8 + (y==H-1)*4 + (x==0)*2 + (x==W-1)).
 * fill: out[n,y,x,:] = V[n][class][:] (fp32 [N][16][c] -> fp16 NHWC);  sums: its adjoint (sums zeroed by the caller). */
int csbsr_border_class_fill(const float* V, void* out, int64_t ld, int32_t N, int32_t H, int32_t W, int32_t c,
                            csbsr_stream_t s);
/* The same fill for the adjoint direction: the dgrad of a 3x3 conv whose dOut is spatially constant (the backward of a global average
 * pool) also takes one value per border class -- V from the taps flipped -- and the activation derivative of the layer below is applied
 * on the way out: out *= mask > 0 ? 1 : mask_slope (mask = that layer's saved output, [N,H,W,c] with pixel pitch mask_ld). */
int csbsr_border_class_fill_masked(const float* V, void* out, int64_t ld, const void* mask, int64_t mask_ld, float mask_slope, int32_t N,
                                   int32_t H, int32_t W, int32_t c, csbsr_stream_t s);
int csbsr_border_class_sums(const void* x, int64_t ld, float* sums, int32_t N, int32_t H, int32_t W, int32_t c,
                            csbsr_stream_t s);
/* the same plus negdot[n][ch] += sum over the pixels with t <= 0 of x * t (t: a second fp16 map of the same geometry): with x = dPre and
 * t = the saved output of a PReLU layer, sum(negdot) / slope^2 is that layer's slope gradient (blocks.py:105-120, the SFTLikeBlock's
 * conv_scale.0 / conv_shift.0 whose activation derivative rode on the dgrad above as csbsr_conv_desc_t::mask + mask_prelu).  Maps of >= 1024 pixels. */
int csbsr_border_class_sums_prelu(const void* x, int64_t ld, const void* t, int64_t t_ld, float* sums, float* negdot, int32_t N, int32_t H,
                                  int32_t W, int32_t c, csbsr_stream_t s);
/* Two 3x3 convs deep (fe_kernel.0 -> fe_kernel.1 on the expanded kernel code, kbpn.py:565-569) the map takes one value per
 * two-ring class ty*5 + tx (conv desc cbias_mode 1): the whole branch is a [N][25][c] table and never exists as a map.  Adjoint of
 * adding such a table: sums[n][class][:] += sum over the pixels of that class of x[n,y,x,:]  (fp32 [N][25][c], H, W >= 5). */
int csbsr_ring_class_sums(const void* x, int64_t ld, float* sums, int32_t N, int32_t H, int32_t W, int32_t c,
                          csbsr_stream_t s);

/* ------------------------------------------------------------------------------------------- batch norm */
/* Train-mode BatchNorm2d (extractors.py:47-66, pspnet.py:48-49,83); sum / sumsq come from the conv epilogue. */
int csbsr_bn_finalize(const float* stat, int64_t count, int32_t c, int32_t cstride, float eps, float momentum,
                      float* mean, float* invstd, float* running_mean, float* running_var, csbsr_stream_t s);
/* y = drop[n][c] * act( (x-mean)*invstd*gamma + beta + res ) ; backward = reduce + apply */
typedef struct {
  int64_t npix, hw;
  int32_t c, creal;
  const void* x; int64_t x_ld;         /* conv output (pre-BN) */
  const float *mean, *invstd, *gamma, *beta;
  const void* res; int64_t res_ld;
  int32_t act; int32_t _pad;
  const float* prelu;
  const float* drop;                   /* fp32 [N][c] keep-mask / (1-p), or NULL */
  void* y; int64_t y_ld;
  const void* dy; int64_t dy_ld;       /* backward only from here */
  float* red;                          /* fp32 [2][c], zeroed by caller */
  float* dprelu;
  void* dx; int64_t dx_ld;
  void* dres; int64_t dres_ld; int32_t dres_accumulate; int32_t _pad1;
  float *dgamma, *dbeta;
  int64_t x_lo, res_lo, y_lo;          /* split-fp16 planes of x / res / y (element offset from the hi plane; 0 = plain fp16) */
} csbsr_bn_desc_t;
int csbsr_bn_apply(const csbsr_bn_desc_t* d, csbsr_stream_t s);
int csbsr_bn_backward(const csbsr_bn_desc_t* d, csbsr_stream_t s);

/* ------------------------------------------------------------------------------------------- pooling / resize */
int csbsr_maxpool3x3s2_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t c, csbsr_stream_t s);
int csbsr_maxpool3x3s2_bwd(const void* x, const void* y, const void* dy, void* dx, int32_t N, int32_t H, int32_t W,
                           int32_t c, csbsr_stream_t s);
/* F.adaptive_avg_pool2d (pspnet.py:32), bins floor/ceil */
int csbsr_adaptive_avgpool_fwd(const void* x, int64_t x_ld, void* y, int32_t N, int32_t H, int32_t W, int32_t c,
                               int32_t OH, int32_t OW, csbsr_stream_t s);
int csbsr_adaptive_avgpool_bwd(const void* dy, void* dx, int64_t dx_ld, int32_t accumulate, int32_t N, int32_t H,
                               int32_t W, int32_t c, int32_t OH, int32_t OW, csbsr_stream_t s);
/* F.interpolate(mode='bilinear'), both align_corners modes (pspnet.py:39,56,122); optional Dropout2d channel
 * scale applied to the input side (pspnet.py:105-114) */
int csbsr_bilinear_fwd(const void* x, int64_t x_ld, void* y, int64_t y_ld, int32_t N, int32_t H, int32_t W,
                       int32_t c, int32_t OH, int32_t OW, int32_t align_corners, const float* drop, csbsr_stream_t s);
int csbsr_bilinear_bwd(const void* dy, int64_t dy_ld, void* dx, int64_t dx_ld, int32_t accumulate, int32_t N,
                       int32_t H, int32_t W, int32_t c, int32_t OH, int32_t OW, int32_t align_corners,
                       const float* drop, csbsr_stream_t s);
int csbsr_bilinear32_fwd(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW,
                         int32_t align_corners, csbsr_stream_t s);
int csbsr_bilinear32_bwd(const float* dy, float* dx, int32_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW,
                         int32_t align_corners, csbsr_stream_t s);
/* nn.Upsample(scale_factor, 'bicubic') of the LR input added to the SR residual (kbpn.py:110-114) */
int csbsr_bicubic_up_add(const float* x, float* out, int32_t planes, int32_t H, int32_t W, int32_t scale,
                         csbsr_stream_t s);
/* FactorResize('bicubic') = F.interpolate(bicubic, antialias) by an integer factor (transforms.py:505-531) */
int csbsr_aa_bicubic_down_fwd(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t f,
                              int32_t antialias, csbsr_stream_t s);
int csbsr_aa_bicubic_down_bwd(const float* dy, float* dx, int32_t accumulate, int32_t planes, int32_t H, int32_t W,
                              int32_t f, int32_t antialias, csbsr_stream_t s);

/* ------------------------------------------------------------------------------------------- blur kernels */
/* Per-sample depthwise KxK cross-correlation (kbpn.py:394-402 stride = scale; sr_loss_functions.py:89-94
 * stride 1) of fp32 NCHW images with kvec fp32 [N][K*K]:  y = blur(x) - sub.  y16: optional fp16 NHWC copy. */
int csbsr_blur_fwd(const float* x, const float* kvec, int32_t N, int32_t C, int32_t H, int32_t W, int32_t K,
                   int32_t stride, const float* sub, float* y32, void* y16, int64_t y16_ld, csbsr_stream_t s);
int csbsr_blur_bwd_input(const float* dy, const float* kvec, float* dx, int32_t accumulate, int32_t N, int32_t C,
                         int32_t H, int32_t W, int32_t K, int32_t stride, csbsr_stream_t s);
int csbsr_blur_bwd_kernel(const float* dy, const float* x, float* dk, int32_t N, int32_t C, int32_t H, int32_t W,
                          int32_t K, int32_t stride, csbsr_stream_t s);

/* ------------------------------------------------------------------------------------------- losses */
/* Exact Euclidean distance transform + normalised signed distance map of compute_sdf1_1
 * (boundary_loss.py:40-67): mask fp32 [N][H][W] in {0,1} -> sdf.  scratch: 3*N*H*W + 2*N floats. */
int csbsr_sdf(const float* mask, float* sdf, float* scratch, int32_t N, int32_t H, int32_t W, csbsr_stream_t s);
/* BoundaryComboLoss = alpha*(lw0*WBCE + lw1*Dice)/(lw0+lw1) + (1-alpha)*mean(p*sdf)
 * (loss_functions.py:49-75,189-210,258-345; boundary_loss.py:26-38) on probability maps fp32 [N][HW]:
 * reduce -> sums[N][8];  finish -> loss[n] += weight*L, dp (+)= gscale[n]*weight*dL/dp */
int csbsr_segloss_reduce(const float* p, const float* t, const float* sdf, int32_t N, int64_t hw, float* sums,
                         float pw0, float pw1, csbsr_stream_t s);
int csbsr_segloss_finish(const float* p, const float* t, const float* sdf, int32_t N, int64_t hw, const float* sums,
                         float alpha, float pw0, float pw1, float lw0, float lw1, float weight, const float* gscale,
                         float* loss, float* dp, int32_t dp_accumulate, csbsr_stream_t s);
/* L1 terms of KBPNLoss (sr_loss_functions.py:41-54): sums[n] += sum w|a-b| ; da (+)= gscale*gs_n[n]*w*sign(a-b) */
int csbsr_l1_fwd_bwd(const float* a, const float* b, const float* wmap, int32_t N, int32_t C, int64_t hw,
                     float* sums, float gscale, const float* gs_n, float* da, int32_t da_accumulate, csbsr_stream_t s);
/* d(pre-sigmoid) of a 1-channel head as channel 0 of an fp16 NHWC8 tensor */
int csbsr_sigmoid_bwd_to_nhwc8(const float* dp, const float* p, void* out, int64_t npix, float scale,
                               csbsr_stream_t s);

/* 1-channel heads: F.conv2d(x, w[1, C, 1, 1], b) (+ sigmoid) of PSPNet's ``final`` and aux classifiers (/root/reference/model/modeling/pspnet.py:
 * 117-121, 100-106).  x: fp16 NHWC, pixel stride ld, C in {64, 128, 256}; lo != 0: a split map, value = x[.] + x[. + lo]; w fp32 [C] (NOT rounded);
 * out fp32 [npix].  csbsr_head1_bwd_input: dx[pixel][c] = dpre[pixel * dpre_ld] * w[c] (fp16 NHWC, the dgrad of the same conv). */
int csbsr_head1_fwd(const void* x, int64_t ld, int64_t lo, int32_t c, const float* w, const float* bias, int32_t sigmoid, float* out,
                    int64_t npix, csbsr_stream_t s);
int csbsr_head1_bwd_input(const void* dpre, int64_t dpre_ld, const float* w, int32_t c, void* dx, int64_t ld, int64_t npix, csbsr_stream_t s);

/* ------------------------------------------------------------------------------------------- data path either side of the hot path */
/* Anisotropic Gaussian blur kernels, out[n] = exp(-(a x^2 + 2 b x y + c y^2)) / sum on linspace(-K/2, K/2, K)^2 with (a, b, c) from
 * params[n] = (sigma_x, sigma_y, theta in radians): GaussianBlur.make(), model/data/blur/blur.py:121-167 (the degradation batch
 * generator, crack_dataset.py:40-64; blur + antialiased bicubic down-scaling reuse csbsr_blur_fwd / csbsr_aa_bicubic_down_fwd). */
int csbsr_gaussian_kernels(const float* params, float* out, int32_t N, int32_t K, csbsr_stream_t s);
/* IoU of (pred - t_i > 0) against (mask > 0.5) for T ascending thresholds in one pass: inference.py:50-53,111-119 with
 * estimate_metrics.IoU (:64-84).  hist = caller-zeroed uint32 [N][2][T+1] workspace; iou / inter / uni are fp32 [N][T] (inter, uni
 * optional). */
int csbsr_iou_sweep(const float* pred, const float* mask, const float* thresholds, int32_t N, int64_t hw, int32_t T, float smooth,
                    uint32_t* hist, float* iou, float* inter, float* uni, csbsr_stream_t s);
/* PSNR = 10 log10(1 / mse) and SSIM (11x11 Gaussian window, sigma 1.5, zero padding) per sample of fp32 NCHW batches in [0,1]:
 * estimate_metrics.py:89-101 (PSNR), :135-191 (SSIM).  sums = caller-zeroed fp32 [N][2] workspace. */
int csbsr_psnr_ssim(const float* a, const float* b, int32_t N, int32_t C, int32_t H, int32_t W, float* sums, float* psnr, float* ssim,
                    csbsr_stream_t s);

/* Backward of kb.up_conv1 -- ConvTranspose2d(3 -> cout, 8x8, stride 4, pad 2, no bias) + PReLU (+ residual) -- in one streaming pass over the
 * output gradient (replaces the autograd backward of /root/reference/model/modeling/kbpn.py:372-374,405-409 for that layer: PReLU backward,
 * conv_transpose2d weight gradient, slope gradient).  The pre-activation is rebuilt from the 3-channel input x [N, h, w, 8] and the forward's
 * phase-packed weights (csbsr_pack_weights kind 2, the operand of csbsr_conv_forward), so neither the saved output nor the residual is read:
 *   dpre [N, 4h, 4w, cout] fp16 = dout * (pre > 0 ? 1 : *prelu);
 *   slabs (optional, with dprelu_part): csbsr_thin_tp_backward_slabs(N, h) fp32 slabs [8][64 taps][cout] of the weight gradient in
 *   csbsr_conv_wgrad's layout (fold with csbsr_unpack_wgrad(.., A = 3, KH = KW = 8, seg0 = cout, .., splits = that count, ca_padded = 8));
 *   dprelu_part [4 * slabs] partial sums of dout * min(pre, 0) (one per workgroup: add them in index order).  Strides in elements. */
int32_t csbsr_thin_tp_backward_slabs(int32_t N, int32_t h);
int csbsr_thin_tp_backward(const void* dout, int64_t d_sn, int64_t d_sy, int64_t d_sx, const void* x, int64_t x_sn, int64_t x_sy,
                           int64_t x_sx, const void* wt_packed, int32_t cin, int32_t cout, int32_t stride, int32_t pad,
                           const float* prelu, int32_t N, int32_t h, int32_t w, void* dpre, int64_t p_sn, int64_t p_sy, int64_t p_sx,
                           float* slabs, float* dprelu_part, csbsr_stream_t s);
/* 1 if csbsr_conv_forward takes this descriptor with split_fused = 1 (the fused three-product stage of a split-fp16 input exists only in the
 * LDS-DMA kernels); 0: pack the three-block operand (csbsr_pack_weights_split layout 0) and clear split_fused */
int32_t csbsr_conv_split_fused_eligible(const csbsr_conv_desc_t* d);
/* 1 if csbsr_conv_forward takes this launch WITH its dact / dres fields (see csbsr_conv_desc_t::dres) */
int32_t csbsr_conv_thin_dact_eligible(const csbsr_conv_desc_t* d);
#ifdef __cplusplus
}
#endif
#endif
