// Kernel-argument block and the fused epilogue shared by the two implicit-GEMM convolution kernels
// (conv_igemm.hip: register-staged, any channel count; conv_igemm_glds.hip: LDS-DMA, MFMA-bound shapes).
#pragma once
#include "common.h"

struct ConvK {
  csbsr_seg_t in[2];
  int N, H, W, OH, OW;
  int transposed, KHt, KWt;  // taps per phase
  int stride, pad, dil;
  int ctot, c0;              // padded channels: total, segment 0
  int Kp;                    // padded K of the packed weights
  int rows_p;                // padded weight rows per phase
  const half_t* wt;
  int cout, coutp;
  half_t* out16; long o_sn, o_sy, o_sx;
  float* out32; long o32_sn, o32_sy, o32_sx, o32_sc;
  const float* bias;
  const float* cbias;        // optional [N][16 | 25][coutp]: bias per (sample, position class of the output pixel) -- folded constant segment
  int cb_mode;               // 0: 16 border classes, 1: 25 two-ring classes (csbsr_conv_desc_t::cbias_mode)
  int act; float act_slope; const float* prelu;
  int res_mode; const half_t* res; long r_sn, r_sy, r_sx;
  const half_t* res2; long r2_sn, r2_sy, r2_sx;
  int accumulate;
  int stat_mode; float* stat;
  float out_scale;
  unsigned tiles_m, tiles_n;
  int tap_group;             // 1: strided layer, taps issued residue class by residue class (glds kernel, see issue())
  int tile2d;                // 1: 16-wide 2-D pixel tiles (glds kernel, see there)
  int nphase_flat;           // > 1: 1-D grid with the transposed conv's output phase as the fastest index (glds kernel)
  // split-fp16 ("hi + lo") operands of the detector precision mode: element offset from a tensor's hi plane to its lo plane
  // (value = hi + lo, both fp16, same strides); 0 = plain fp16
  long o_lo, r_lo, r2_lo;
  // activation-derivative mask fused into a dgrad epilogue: the result (after residual / accumulate) is multiplied by
  // (mask > 0 ? 1 : mask_slope), mask = the saved forward output of the layer whose input gradient this launch produces --
  // i.e. dPre of that layer leaves this kernel directly and the stand-alone epilogue-backward pass is skipped
  const half_t* mask; long m_sn, m_sy, m_sx; float mask_slope;
  const float* mask_prelu;   // the masking layer's PReLU slope on the device (overrides mask_slope): csbsr_conv_desc_t::mask_prelu
  // fused statistics, order-fixed: every workgroup writes ITS sums as one partial row stat_part[(phase * tiles_m + tile_m) * stat_ld + ...]
  // ([sum | sumsq] for CSBSR_STAT_BN, [sum] for CSBSR_STAT_SAMPLE_SUM); the launcher folds the rows into ``stat`` with
  // csbsr_sum_partials* (no atomics: two runs are bit-identical).  hw_pad > 0: the linear pixel index is laid out per sample,
  // each sample padded to hw_pad (a multiple of the pixel tile) positions, so a tile never straddles two samples.
  float* stat_part; long stat_ld; int hw_pad;
  long bias_sn;              // per-sample bias: element stride between the samples' bias rows (0: one row), csbsr_conv_desc_t::bias_sn
  int fs;                    // 1: fused split-fp16 input (csbsr_conv_desc_t::split_fused): in[0] = [hi | lo], c0 = ctot = 2 x plane channels; 2: the same stage
                             // without the x_hi w_lo product ([x_hi | x_lo] w_hi: a layer whose precision plan keeps its weights' fp16 rounding)
};

// value -> (hi, lo) fp16 pair with hi + lo == value to ~2^-22 relative (lo is exact down to fp16's subnormal spacing, 6e-8)
__device__ __forceinline__ void split_store(half_t* o, long lo_off, const float (&v)[8]) {
  h8 hv, lv;
#pragma unroll
  for (int e = 0; e < 8; ++e) { hv[e] = (half_t)v[e]; lv[e] = (half_t)(v[e] - (float)hv[e]); }
  *reinterpret_cast<h8*>(o) = hv;
  if (lo_off) *reinterpret_cast<h8*>(o + lo_off) = lv;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0): between the epilogue passes that
// means sitting out the write acknowledgement of the previous pass's global stores (~2 us per workgroup on a 16 us tile).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// weight rows are padded so that a 128-row tile never reads past the buffer when the wide tile is used
static inline int conv_rows_padded(int nrows) { return round_up(nrows, nrows > 64 ? 128 : 32); }

// Operands the epilogue reads from HBM for one (pixel, 8 channels): residual(s) and, when accumulating, the old output.  Issued for
// all of a thread's rows up front so their latencies overlap instead of serialising row by row (a 128x128 tile has only 8 MFMA
// k-steps per 8 epilogue rows on the short-K layers: one exposed HBM round trip per row was most of the epilogue).
struct EpiPre { h8 r, old; };
__device__ __forceinline__ void conv_epilogue_prefetch(const ConvK& p, int co, int n, int oy, int ox, EpiPre& q) {
  if (p.res_mode != CSBSR_RES_NONE) {
    q.r = *reinterpret_cast<const h8*>(p.res + n * p.r_sn + oy * p.r_sy + ox * p.r_sx + co);
  }
  if (p.out16 && p.accumulate) q.old = *reinterpret_cast<const h8*>(p.out16 + n * p.o_sn + oy * p.o_sy + ox * p.o_sx + co);
}

// one output pixel x 8 consecutive channels: scale + bias + activation, fused statistics, residual combine, stores
__device__ __forceinline__ void conv_epilogue_row(const ConvK& p, float (&v)[8], const float (&bias)[8], float slope, int co, int n, int oy,
                                                  int ox, float (&ssum)[8], float (&ssq)[8], const EpiPre* pre = nullptr) {
  const float* cb = nullptr;
  if (p.cbias) {
    if (p.cb_mode == 0) {
      const int cls = (oy == 0) * 8 + (oy == p.OH - 1) * 4 + (ox == 0) * 2 + (ox == p.OW - 1);
      cb = p.cbias + ((size_t)n * 16 + cls) * p.coutp + co;
    } else {
      const int ty = oy < 2 ? oy : (oy >= p.OH - 2 ? oy - p.OH + 5 : 2), tx = ox < 2 ? ox : (ox >= p.OW - 2 ? ox - p.OW + 5 : 2);
      cb = p.cbias + ((size_t)n * 25 + ty * 5 + tx) * p.coutp + co;
    }
  }
  // (the class row as two 16-byte loads -- rows are coutp floats, co a multiple of 8: as eight indexed reads of a pointer of unknown
  // alignment it was eight dependent 4-byte loads per row piece, 10 % of a launch of the SFT conv0 forwards)
  float cbv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (cb) {
    const f4 c0 = *reinterpret_cast<const f4*>(cb), c1 = *reinterpret_cast<const f4*>(cb + 4);
    cbv[0] = c0[0]; cbv[1] = c0[1]; cbv[2] = c0[2]; cbv[3] = c0[3]; cbv[4] = c1[0]; cbv[5] = c1[1]; cbv[6] = c1[2]; cbv[7] = c1[3];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float t = v[e] * p.out_scale + bias[e];
    if (cb) t += cbv[e];
    t = apply_act(t, p.act, slope);
    v[e] = (co + e < p.cout) ? t : 0.f;
  }
  if (p.stat_mode == CSBSR_STAT_BN) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
  } else if (p.stat_mode == CSBSR_STAT_SAMPLE_SUM) {      // a tile holds pixels of ONE sample (ConvK::hw_pad / 2-D tiles)
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] += v[e];
  }
  if (p.res_mode != CSBSR_RES_NONE) {
    const half_t* rp = p.res + n * p.r_sn + oy * p.r_sy + ox * p.r_sx + co;
    const h8 r = pre ? pre->r : *reinterpret_cast<const h8*>(rp);
    h8 r2 = {0, 0, 0, 0, 0, 0, 0, 0}, rl = {0, 0, 0, 0, 0, 0, 0, 0}, r2l = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p.r_lo) rl = *reinterpret_cast<const h8*>(rp + p.r_lo);
    if (p.res_mode == CSBSR_RES_FMA) {
      const half_t* r2p = p.res2 + n * p.r2_sn + oy * p.r2_sy + ox * p.r2_sx + co;
      r2 = *reinterpret_cast<const h8*>(r2p);
      if (p.r2_lo) r2l = *reinterpret_cast<const h8*>(r2p + p.r2_lo);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float rv = (float)r[e] + (float)rl[e];
      switch (p.res_mode) {
        case CSBSR_RES_ADD: v[e] += rv; break;
        case CSBSR_RES_SUB: v[e] -= rv; break;
        case CSBSR_RES_MUL: v[e] *= rv; break;
        default: v[e] += rv * ((float)r2[e] + (float)r2l[e]); break;
      }
    }
  }
  if (p.out16) {
    half_t* o = p.out16 + n * p.o_sn + oy * p.o_sy + ox * p.o_sx + co;
    if (p.accumulate) {
      const h8 old = pre ? pre->old : *reinterpret_cast<const h8*>(o);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)old[e];
    }
    if (p.mask) {
      const h8 mk = *reinterpret_cast<const h8*>(p.mask + n * p.m_sn + oy * p.m_sy + ox * p.m_sx + co);
      const float ms_ = p.mask_prelu ? *p.mask_prelu : p.mask_slope;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= ((float)mk[e] > 0.f ? 1.f : ms_);
    }
    split_store(o, p.o_lo, v);
  }
  if (p.out32) {
    float* o = p.out32 + n * p.o32_sn + oy * p.o32_sy + ox * p.o32_sx;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (co + e < p.cout) {
        float* q = o + (co + e) * p.o32_sc;
        *q = (p.accumulate && !p.out16) ? *q + v[e] : v[e];
      }
  }
}

// ---- straight-line rows for the common fused epilogues ---------------------------------------------------------------------------
// No sample statistics / constant-segment bias / fp32 output, whole channel octets, identity / ReLU / leaky activation (as
// the select t > 0 ? t : t * sneg), residual add or subtract, optional accumulate, optional BatchNorm sums: the mode switches of
// conv_epilogue_row become two multipliers picked once per kernel.  The general row re-tests every mode per element (~100 scalar
// branches per row): its 8 rows took 7.7 us of an 18 us transposed-conv workgroup whose 8 K slices take 5.9 us (CSBSR_TS build).
struct EpiFast { bool ok, has_res, has_old, has_mask, bn, masked, has_cb, amax; float sneg, rsign, osc, mslope; int cout; long o_lo; };
// which launches the straight-line rows cover (host and device: conv_x3 picks its kernel instance by it).  The activation is
// the select t > 0 ? t : t * sneg with a kernel-uniform sneg: exact for any slope (a learned PReLU slope may exceed 1) and NaN-
// preserving (round 4's max(t, 0) + sneg * min(t, 0) zeroed a NaN accumulator: maxnum / minnum return the non-NaN operand).
__host__ __device__ __forceinline__ bool conv_epilogue_fast_ok(const ConvK& p) {
  return (p.stat_mode == CSBSR_STAT_NONE || p.stat_mode == CSBSR_STAT_BN) && !p.out32 && p.out16 && p.act != CSBSR_ACT_SIGMOID &&
         (p.res_mode == CSBSR_RES_NONE || p.res_mode == CSBSR_RES_ADD || p.res_mode == CSBSR_RES_SUB) && !p.r_lo;
}
__device__ __forceinline__ EpiFast conv_epilogue_fast_setup(const ConvK& p, float slope) {
  EpiFast f;
  f.has_cb = p.cbias != nullptr;      // a position-class bias: the caller adds the pixel's class row to ``bias`` (conv_class_bias_row)
  f.ok = conv_epilogue_fast_ok(p);
  f.o_lo = p.o_lo;
  f.sneg = p.act == CSBSR_ACT_NONE ? 1.f : (p.act == CSBSR_ACT_RELU ? 0.f : slope);
  f.amax = f.sneg >= 0.f && f.sneg < 1.f;      // the activation as max(t, t * sneg): two operations per element instead of compare + select + multiply, NaN in = NaN out
  f.rsign = p.res_mode == CSBSR_RES_ADD ? 1.f : (p.res_mode == CSBSR_RES_SUB ? -1.f : 0.f);
  f.has_res = p.res_mode != CSBSR_RES_NONE; f.has_old = p.accumulate != 0; f.bn = p.stat_mode == CSBSR_STAT_BN;
  f.has_mask = p.mask != nullptr; f.mslope = (p.mask && p.mask_prelu) ? *p.mask_prelu : p.mask_slope;
  f.masked = (p.cout & 7) != 0;      // the last channel octet is partly padding: those lanes are forced to zero (one branch per row)
  f.osc = p.out_scale; f.cout = p.cout;
  return f;
}
// bias + the class row of output pixel (oy, ox) of sample n (ConvK::cbias: 16 border classes or 25 two-ring classes), 8 couts from co
__device__ __forceinline__ void conv_class_bias_row(const ConvK& p, const float (&bias)[8], int co, int n, int oy, int ox, float (&out)[8]) {
  int cls, ncls;
  if (p.cb_mode == 0) { cls = (oy == 0) * 8 + (oy == p.OH - 1) * 4 + (ox == 0) * 2 + (ox == p.OW - 1); ncls = 16; }
  else {
    const int ty = oy < 2 ? oy : (oy >= p.OH - 2 ? oy - p.OH + 5 : 2), tx = ox < 2 ? ox : (ox >= p.OW - 2 ? ox - p.OW + 5 : 2);
    cls = ty * 5 + tx; ncls = 25;
  }
  const float* cb = p.cbias + ((size_t)n * ncls + cls) * p.coutp + co;
  const f4 c0 = *reinterpret_cast<const f4*>(cb), c1 = *reinterpret_cast<const f4*>(cb + 4);
  out[0] = bias[0] + c0[0]; out[1] = bias[1] + c0[1]; out[2] = bias[2] + c0[2]; out[3] = bias[3] + c0[3];
  out[4] = bias[4] + c1[0]; out[5] = bias[5] + c1[1]; out[6] = bias[6] + c1[2]; out[7] = bias[7] + c1[3];
}
// one pixel x 8 channels co..co+7; o = &out16[pixel][co]; rr / oo / mm = residual / old output / activation mask (zeros when absent;
// only read when EXTRA)
template <bool EXTRA, bool BNSTAT>
__device__ __forceinline__ void conv_epilogue_fast_values(const EpiFast& f, const float (&v)[8], const float (&bias)[8], int co, const h8& rr,
                                                          const h8& oo, float (&ssum)[8], float (&ssq)[8], const h8& mm, float (&t)[8],
                                                          bool valid = true) {
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = v[e] * f.osc + bias[e];
  // the activation t > 0 ? t : t * sneg (never max(t, 0): fmaxf returns the non-NaN operand and would turn an overflowed -- NaN -- accumulator
  // into 0, hiding it from the optimiser's overflow check).  For 0 <= sneg < 1 that select equals max(t, t * sneg) for every finite t, and a
  // NaN t makes both operands NaN; the identity (sneg = 1) does nothing; a learned PReLU slope outside [0, 1) keeps the select.
  if (f.amax) {
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = __builtin_fmaxf(t[e], t[e] * f.sneg);
  } else if (f.sneg != 1.f) {
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = t[e] > 0.f ? t[e] : t[e] * f.sneg;
  }
  if (f.masked && co + 8 > f.cout) {      // (only the last channel octet is partly padding)
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (co + e < f.cout) ? t[e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if constexpr (BNSTAT) { const float ts = valid ? t[e] : 0.f; ssum[e] += ts; ssq[e] += ts * ts; }      // (valid: the lane's pixel / octet exists)
    if constexpr (EXTRA) {
      t[e] += f.rsign * (float)rr[e];
      t[e] += (float)oo[e];
      if (f.has_mask) t[e] *= ((float)mm[e] > 0.f ? 1.f : f.mslope);
    }
  }
}
template <bool EXTRA, bool BNSTAT>
__device__ __forceinline__ void conv_epilogue_fast_row(const EpiFast& f, const float (&v)[8], const float (&bias)[8], int co, half_t* o,
                                                       const h8& rr, const h8& oo, float (&ssum)[8], float (&ssq)[8],
                                                       const h8& mm = h8{1, 1, 1, 1, 1, 1, 1, 1}) {
  float t[8];
  conv_epilogue_fast_values<EXTRA, BNSTAT>(f, v, bias, co, rr, oo, ssum, ssq, mm, t);
  split_store(o, f.o_lo, t);
}

#if defined(__HIP_DEVICE_COMPILE__)
// Stores that are UNCONDITIONAL in control flow: a buffer store whose lanes outside the image / past the last channel octet carry an
// out-of-range offset (dropped by the bounds check) instead of sitting behind an exec-mask branch.  The compiler's waitcnt pass merges the
// paths of a branch to the FEWEST operations in flight, so one skippable store makes every later counted wait a full drain (vmcnt(0) =
// the write acknowledgement of everything before it); with these, a tile's sixteen stores are sixteen on every path.
typedef unsigned conv_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_make_rs(const void* base) {
  const unsigned long a = reinterpret_cast<unsigned long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi_ << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
// voff: byte offset of the 8 channels from the resource base, bit 31 set = masked lane
__device__ __forceinline__ void split_store_rs(__amdgpu_buffer_rsrc_t rs, int voff, long lo_off, const float (&v)[8]) {
  h8 hv, lv;
#pragma unroll
  for (int e = 0; e < 8; ++e) { hv[e] = (half_t)v[e]; lv[e] = (half_t)(v[e] - (float)hv[e]); }
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(conv_u4, hv), rs, voff, 0, 0);
  if (lo_off) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(conv_u4, lv), rs, voff + 2 * (int)lo_off, 0, 0);
}
#endif

// The straight-line epilogue of a RESIDENT-PIXEL tile (csrc/conv_x3.hip, csrc/conv_x3n.hip): a lane holds, for each of its 4 tile rows nt, 4
// pieces mp = 2 mt + pair of 8 consecutive couts co0 + 16 mp .. of ONE pixel (row oy0 + nt, column ox): acc[mt][nt][8 pair + e].
//
// What a tile's epilogue costs is not its arithmetic but the ORDER of its memory operations: gfx9's vmcnt counts loads and stores in one
// in-order queue, so a load issued after a store cannot be waited for without sitting out that store's write acknowledgement (~2 us, more
// while HBM absorbs a write stream).  As sixteen rows each loading its bias / operands and then storing, a tile paid sixteen of those in a
// row (measured on the 64 -> 505 layer of config 5, N = 4: 9.8 ms per launch of which the K loop is 5.8, profiles/r06_x3n_ablation.txt).
// Here: the bias -- plus the class-0 row of a position-class bias when the whole tile is interior -- is loaded for all four pieces BEFORE
// the first store; launches with per-pixel operands (residual, old output, activation mask) load batch k + 1's operands before batch k's
// stores (a batch = NB rows of one piece column), so every wait the compiler inserts is a counted one that leaves the stores in flight;
// and the stores are unconditional in control flow (split_store_rs), so that the K loop that follows can start under them: the caller's
// wait for the next tile's first halo chunk may leave CONV_TILE_STORES operations outstanding.
// Loads of lanes outside the image / past the last channel octet go to clamped (valid) addresses and are discarded.
// ``interior``: every pixel of the lane's four rows has position class 0 (the caller's wave-uniform test); border tiles of a class-bias
// launch take the row-by-row path (class rows per pixel).
#define CONV_TILE_STORES 16      // vector-memory stores every path of conv_epilogue_fast_tile issues per wave, at least
#if defined(__HIP_DEVICE_COMPILE__)
template <bool BNSTAT, int NB = 2>
__device__ __forceinline__ void conv_epilogue_fast_tile(const ConvK& p, const EpiFast& fe, const f16v (&acc)[2][4], int co0, int n, int oy0,
                                                        int ox, bool interior, float (&bsum)[4][8], float (&bsq)[4][8]) {
  const int oxc = ox < p.OW ? ox : p.OW - 1;
  const bool extra = fe.has_res || fe.has_old || fe.has_mask;
  // one buffer resource per tile (wave-uniform base: the tile's first row), the lane's row / pixel / channel offset in the vector offset
  const __amdgpu_buffer_rsrc_t rs = conv_make_rs(p.out16 + n * p.o_sn + (long)oy0 * p.o_sy);      // (the tile's first row is inside the image)
  int oyc[4], rowoff[4];
  bool valid[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int oy = oy0 + nt;
    oyc[nt] = oy < p.OH ? oy : p.OH - 1;
    valid[nt] = oy < p.OH && ox < p.OW;
    rowoff[nt] = (2 * nt * (int)p.o_sy) | (valid[nt] ? 0 : (int)0x80000000);
  }
  const int pixoff = 2 * (int)(oxc * p.o_sx);
  if (fe.has_cb && !interior) {
    // border tile of a class-bias launch: row by row (1-2 % of the tiles)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int mp = 0; mp < 4; ++mp) {
        const int co = co0 + 16 * mp;
        const int coc = co < p.coutp ? co : p.coutp - 8;
        float v[8], bias[8], brow[8], t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = acc[mp >> 1][nt][8 * (mp & 1) + e];
          bias[e] = 0.f;
        }
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int i = coc + e;
            const float tb = p.bias[n * p.bias_sn + (i < p.cout ? i : p.cout - 1)];
            bias[e] = i < p.cout ? tb : 0.f;
          }
        }
        conv_class_bias_row(p, bias, coc, n, oyc[nt], oxc, brow);
        h8 rr = {0, 0, 0, 0, 0, 0, 0, 0}, oo = {0, 0, 0, 0, 0, 0, 0, 0}, mm = {1, 1, 1, 1, 1, 1, 1, 1};
        if (fe.has_res) rr = *reinterpret_cast<const h8*>(p.res + n * p.r_sn + oyc[nt] * p.r_sy + oxc * p.r_sx + coc);
        if (fe.has_old) oo = *reinterpret_cast<const h8*>(p.out16 + n * p.o_sn + oyc[nt] * p.o_sy + oxc * p.o_sx + coc);
        if (fe.has_mask) mm = *reinterpret_cast<const h8*>(p.mask + n * p.m_sn + oyc[nt] * p.m_sy + oxc * p.m_sx + coc);
        if (extra && !BNSTAT) conv_epilogue_fast_values<true, BNSTAT>(fe, v, brow, co, rr, oo, bsum[mp], bsq[mp], mm, t, valid[nt] && co < p.coutp);
        else conv_epilogue_fast_values<false, BNSTAT>(fe, v, brow, co, rr, oo, bsum[mp], bsq[mp], mm, t, valid[nt] && co < p.coutp);
        split_store_rs(rs, (rowoff[nt] + pixoff + 2 * coc) | (co >= p.coutp ? (int)0x80000000 : 0), fe.o_lo, t);
        __builtin_amdgcn_sched_barrier(0);      // (keeps a piece's temporaries from being hoisted across the straight-line pieces: registers)
      }
    }
    return;
  }
  // ---- bias (+ the interior class row) of the four pieces, before any store
  float bias[4][8];
#pragma unroll
  for (int mp = 0; mp < 4; ++mp) {
    const int co = co0 + 16 * mp;
    const int coc = co < p.coutp ? co : p.coutp - 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[mp][e] = 0.f;
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = coc + e;
        const float t = p.bias[n * p.bias_sn + (i < p.cout ? i : p.cout - 1)];
        bias[mp][e] = i < p.cout ? t : 0.f;
      }
    }
    if (fe.has_cb) {
      const float* cb = p.cbias + (size_t)n * (p.cb_mode == 0 ? 16 : 25) * p.coutp + (p.cb_mode == 0 ? 0 : 12 * (size_t)p.coutp) + coc;      // class 0 / the centre class (2, 2)
      const f4 c0 = *reinterpret_cast<const f4*>(cb), c1 = *reinterpret_cast<const f4*>(cb + 4);
      bias[mp][0] += c0[0]; bias[mp][1] += c0[1]; bias[mp][2] += c0[2]; bias[mp][3] += c0[3];
      bias[mp][4] += c1[0]; bias[mp][5] += c1[1]; bias[mp][6] += c1[2]; bias[mp][7] += c1[3];
    }
  }
  if (!extra) {
    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int mp = 0; mp < 4; ++mp) {
        const int co = co0 + 16 * mp;
        const int coc = co < p.coutp ? co : p.coutp - 8;
        float v[8], t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[mp >> 1][nt][8 * (mp & 1) + e];
        conv_epilogue_fast_values<false, BNSTAT>(fe, v, bias[mp], co, z, z, bsum[mp], bsq[mp], z, t, valid[nt] && co < p.coutp);
        split_store_rs(rs, (rowoff[nt] + pixoff + 2 * coc) | (co >= p.coutp ? (int)0x80000000 : 0), fe.o_lo, t);
        __builtin_amdgcn_sched_barrier(0);      // (keeps a piece's temporaries from being hoisted across the straight-line pieces: registers)
      }
    }
    return;
  }
  // ---- per-pixel operands: batches b = (piece column mp, rows nb .. nb + NB - 1), batch b + 1's loads ahead of batch b's stores
  // (not together with fused BatchNorm sums: no such launch in the model, and that instance has no registers for both -- the launcher refuses)
  if constexpr (BNSTAT) return;
  constexpr int NBATCH = 4 * (4 / NB);
  h8 rr[2][NB], oo[2][NB], mm[2][NB];
  auto load_batch = [&](int b, int st) __attribute__((always_inline)) {
    const int mp = b / (4 / NB), nb = (b % (4 / NB)) * NB;
    const int co = co0 + 16 * mp;
    const int coc = co < p.coutp ? co : p.coutp - 8;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const int oyr = oyc[nb + r];
      rr[st][r] = h8{0, 0, 0, 0, 0, 0, 0, 0}; oo[st][r] = h8{0, 0, 0, 0, 0, 0, 0, 0}; mm[st][r] = h8{1, 1, 1, 1, 1, 1, 1, 1};
      if (fe.has_res) rr[st][r] = *reinterpret_cast<const h8*>(p.res + n * p.r_sn + oyr * p.r_sy + oxc * p.r_sx + coc);
      if (fe.has_old) oo[st][r] = *reinterpret_cast<const h8*>(p.out16 + n * p.o_sn + oyr * p.o_sy + oxc * p.o_sx + coc);
      if (fe.has_mask) mm[st][r] = *reinterpret_cast<const h8*>(p.mask + n * p.m_sn + oyr * p.m_sy + oxc * p.m_sx + coc);
    }
  };
  load_batch(0, 0);
#pragma unroll
  for (int b = 0; b < NBATCH; ++b) {
    if (b + 1 < NBATCH) load_batch(b + 1, (b + 1) & 1);
    const int mp = b / (4 / NB), nb = (b % (4 / NB)) * NB;
    const int co = co0 + 16 * mp;
    const int coc = co < p.coutp ? co : p.coutp - 8;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      float v[8], t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[mp >> 1][nb + r][8 * (mp & 1) + e];
      conv_epilogue_fast_values<true, BNSTAT>(fe, v, bias[mp], co, rr[b & 1][r], oo[b & 1][r], bsum[mp], bsq[mp], mm[b & 1][r], t, valid[nb + r] && co < p.coutp);
      split_store_rs(rs, (rowoff[nb + r] + pixoff + 2 * coc) | (co >= p.coutp ? (int)0x80000000 : 0), fe.o_lo, t);
        __builtin_amdgcn_sched_barrier(0);      // (keeps a piece's temporaries from being hoisted across the straight-line pieces: registers)
    }
  }
}
#endif

// per-thread partial statistics -> the workgroup's LDS bins sStat[2][BN], in a FIXED order.
// CPR = channel chunks per staged row: lanes l, l+CPR, l+2CPR .. of a wave own the same 8 channels, so they are folded with
// xor-shuffles first; then the NW waves add their sums to the bins one wave after the other (wave order = summation order; LDS
// atomics would add them in arrival order).  Called by every thread of the workgroup (barriers inside).
template <int CPR, int NW>
__device__ __forceinline__ void conv_epilogue_flush_stats(const ConvK& p, float* sStat, int BN, int local_col, int co, float (&ssum)[8], float (&ssq)[8]) {
  if (p.stat_mode == CSBSR_STAT_NONE) return;
  const bool sq = p.stat_mode == CSBSR_STAT_BN;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float a = co < p.coutp ? ssum[e] : 0.f, b = (sq && co < p.coutp) ? ssq[e] : 0.f;
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); if (sq) b += __shfl_xor(b, o, 64); }
    ssum[e] = a; ssq[e] = b;
  }
  const int wid = threadIdx.x >> 6;
  const bool mine = (threadIdx.x & 63) < CPR && co < p.coutp;
#pragma unroll 1
  for (int w = 0; w < NW; ++w) {
    if (wid == w && mine) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sStat[local_col + e] += ssum[e];
        if (sq) sStat[BN + local_col + e] += ssq[e];
      }
    }
    lds_barrier();
  }
}
// the workgroup's bins -> its partial row (one thread per cout of the tile)
__device__ __forceinline__ void conv_epilogue_store_stats(const ConvK& p, const float* sStat, int BN, int cout0, size_t row) {
  const int tid = threadIdx.x;
  if (p.stat_mode == CSBSR_STAT_NONE || tid >= BN || cout0 + tid >= p.coutp) return;
  float* r = p.stat_part + row * p.stat_ld + cout0 + tid;
  r[0] = sStat[tid];
  if (p.stat_mode == CSBSR_STAT_BN) r[p.coutp] = sStat[BN + tid];
}
// host side of the above: partial-row geometry before the launch, the fixed-order fold after it
struct ConvStatPlan { long rows; int tps; };
static inline int conv_stat_prepare(ConvK& p, int BM, int nphase, ConvStatPlan& pl, hipStream_t st) {
  pl.rows = 0; pl.tps = 0;
  p.stat_part = nullptr; p.stat_ld = 0; p.hw_pad = 0;
  if (p.stat_mode == CSBSR_STAT_NONE) return 0;
  if (p.stat_mode == CSBSR_STAT_SAMPLE_SUM) {
    CSBSR_CHECK(!p.transposed, "conv: per-sample sums of a transposed convolution are not built");
    if (p.tile2d) pl.tps = (p.OW / 16) * (p.OH / (BM / 16));
    else {
      p.hw_pad = round_up(p.OH * p.OW, BM);
      pl.tps = p.hw_pad / BM;
      p.tiles_m = (unsigned)(p.N * pl.tps);
    }
  }
  pl.rows = (long)p.tiles_m * nphase;
  p.stat_ld = (p.stat_mode == CSBSR_STAT_BN ? 2 : 1) * (long)p.coutp;
  p.stat_part = csbsr_red_scratch(pl.rows * p.stat_ld);
  CSBSR_NEED_SCRATCH(p.stat_part, "conv (fused statistics)");
  if (p.transposed && hipMemsetAsync(p.stat_part, 0, (size_t)pl.rows * p.stat_ld * 4, st) != hipSuccess) {      // phases of unequal size leave tiles unvisited
    csbsr_set_error("conv: memset of the statistics rows failed");
    return 2;
  }
  return 0;
}
static inline int conv_stat_finish(const ConvK& p, const ConvStatPlan& pl, hipStream_t st) {
  if (p.stat_mode == CSBSR_STAT_BN) return csbsr_sum_partials(p.stat_part, (int)pl.rows, p.stat_ld, 2 * p.coutp, p.stat, st);
  if (p.stat_mode == CSBSR_STAT_SAMPLE_SUM) return csbsr_sum_partials_batched(p.stat_part, pl.tps, p.stat_ld, p.coutp, p.stat, p.N, p.coutp, st);
  return 0;
}

// Register-direct epilogue for one 32x32 accumulator tile (no statistics requested): v_permlane32_swap pairs turn the MFMA
// layout (lane l / l+32 hold couts 8q..8q+3 / 8q+4..8q+7 of pixel l%32) into 8 consecutive couts per lane, so every lane issues
// 16-byte stores / residual loads for its own pixel -- no LDS round trip, no barriers, one address computation per pixel.
// ``bias_tab``: per-cout bias already staged by the caller (LDS, indexed by absolute cout, zeros past p.cout), or nullptr to read p.bias
// from global memory here -- 8 dependent L2 round trips per call, which is most of a short tile's epilogue.
__device__ __forceinline__ void conv_epilogue_direct_tile(const ConvK& p, const f16v& acc, int cbase /*first cout of the 32-wide tile*/,
                                                          float slope, int n, int oy, int ox, const float* bias_tab = nullptr,
                                                          const EpiFast* fe = nullptr) {
  if (n < 0) return;
  const int hi = (threadIdx.x & 63) >> 5;
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned a = __float_as_uint(acc[8 * pair + j]), b = __float_as_uint(acc[8 * pair + 4 + j]);
      auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
      // lower lanes: r[0] = own group 2*pair, r[1] = partner's group 2*pair ; upper lanes: r[0] = partner's group 2*pair+1, r[1] = own
      v[j] = __uint_as_float(r[0]);
      v[4 + j] = __uint_as_float(r[1]);
    }
    const int co = cbase + 16 * pair + 8 * hi;
    if (co >= p.coutp) continue;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = bias_tab ? bias_tab[co + e] : ((p.bias && co + e < p.cout) ? p.bias[co + e] : 0.f);
    float s0[8], s1[8];
    if (fe && fe->ok && !fe->bn) {          // straight-line row (the caller set fe up once per kernel)
      half_t* o = p.out16 + n * p.o_sn + oy * p.o_sy + ox * p.o_sx + co;
      h8 rr = {0, 0, 0, 0, 0, 0, 0, 0}, oo = {0, 0, 0, 0, 0, 0, 0, 0};
      if (fe->has_res) rr = *reinterpret_cast<const h8*>(p.res + n * p.r_sn + oy * p.r_sy + ox * p.r_sx + co);
      if (fe->has_old) oo = *reinterpret_cast<const h8*>(o);
      float brow[8];
      if (fe->has_cb) conv_class_bias_row(p, bias, co, n, oy, ox, brow);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) brow[e] = bias[e];
      }
      if (fe->has_res || fe->has_old) conv_epilogue_fast_row<true, false>(*fe, v, brow, co, o, rr, oo, s0, s1);
      else conv_epilogue_fast_row<false, false>(*fe, v, brow, co, o, rr, oo, s0, s1);
      continue;
    }
    conv_epilogue_row(p, v, bias, slope, co, n, oy, ox, s0, s1);
  }
}

// which kernel the last csbsr_conv_forward call of this thread dispatched to (csbsr_debug_last_conv_kernel): lets the host-side
// timing attribute each launch to a kernel name, so bench.py's roofline block is about ONE kernel, the one rocprof lists
enum { CONVK_IGEMM32 = 0, CONVK_IGEMM64, CONVK_IGEMM128, CONVK_GLDS128, CONVK_GLDS256, CONVK_THIN_COUT, CONVK_THIN_CIN, CONVK_GLDS256W, CONVK_HR, CONVK_TP, CONVK_X3, CONVK_THIN_TP, CONVK_X3S, CONVK_THIN_CIN2, CONVK_GLDS64, CONVK_THIN_SC, CONVK_THIN_TPD, CONVK_X3F, CONVK_X3SF, CONVK_X3W, CONVK_X3N };
extern thread_local int g_last_conv_kernel;
int conv_desc_to_k(const csbsr_conv_desc_t* d, ConvK& k);   // conv_igemm.hip: validation + argument block of csbsr_conv_forward
bool conv_thin_eligible(const ConvK& k);                  // conv_thin.hip: 3-channel image heads
int conv_thin_launch(const ConvK& k, hipStream_t st);
void conv_thin_enable(int on);
bool conv_thin_cin_eligible(const ConvK& k, int creal);   // 3-channel image in
int conv_thin_cin_launch(const ConvK& k, int creal, hipStream_t st);
bool conv_thin_cin2_eligible(const ConvK& k, int creal);  // ... with a plain (optionally accumulating) epilogue: the streaming kernel
int conv_thin_cin2_launch(const ConvK& k, int creal, hipStream_t st);
void conv_thin_cin2_enable(int on);
int conv_thin_cin2_dact_launch(const ConvK& k, const csbsr_conv_desc_t* d, hipStream_t st);   // + the epilogue-backward pass of the PReLU + residual layer below
extern "C" int32_t csbsr_conv_thin_dact_eligible(const csbsr_conv_desc_t* d);
bool conv_thin_tp_eligible(const ConvK& k, int creal, bool second_seg);   // 3-channel image into a 2x2-tap transposed conv
int conv_thin_tp_launch(const ConvK& k, hipStream_t st);
bool conv_thin_sc_eligible(const ConvK& k);        // 8x8 s4 / 12x12 s8 strided conv from 128 into <= 3 channels (dgrad of kb.up_conv1)
int conv_thin_sc_launch(const ConvK& k, hipStream_t st);
void conv_thin_sc_enable(int on);
bool conv_thin_tpd_eligible(const ConvK& k, int KH);   // stride-2 transposed conv from 64 channels into <= 3 (dgrad of the detectors' stems)
int conv_thin_tpd_launch(const ConvK& k, int KH, hipStream_t st);
void conv_thin_tpd_enable(int on);
bool conv_glds_eligible(const ConvK& k);
int conv_glds_launch(const ConvK& k, int nphase, long maxM, hipStream_t st);
