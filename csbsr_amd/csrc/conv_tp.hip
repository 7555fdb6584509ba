// Transposed convolution with a 2 x 2-tap phase decomposition (kernel <= 2 x stride: DBPN's 8x8 stride-4 / 12x12 stride-8 up-projections,
// kbpn.py:230-262 UpBlock / DownBlock deconvs, and -- same geometry -- the dgrads of their strided convolutions), 64 or 128 input
// channels -> 128 output channels.  Per output phase (oy % s, ox % s) the layer is a 2x2 convolution of the LOW-resolution input with
// that phase's weights, K = 4 taps x cin: short-K GEMMs that the general LDS-DMA kernel ran at ~520 TF/s because every
// (tile, phase, tap) re-fetched its pixel operand through L2 (80 % hit rate, 20 % at fabric latency behind a 2-3 stage ring) and paid a
// 3.3 us prologue + 3.3 us LDS-staged epilogue around 6.7 us of K loop.  Here:
//
//  * one persistent workgroup (8 waves) per CU owns an 8 x 32 tile of input positions; its (8+2) x (32+2) halo goes HBM -> LDS ONCE
//    (global_load_lds_dwordx4, zero page outside the image) and serves all taps of up to 8 phases: the pixel operand crosses the
//    fabric ~1.3 times instead of 64;
//  * the only operand streamed in the K loop is the phase's weights, 16 KB per 64-channel K step, L2-resident (<= 2 MB per layer),
//    packed in MFMA-fragment order (csbsr_pack_weights_tp) so the 4-stage LDS ring is filled by straight 1 KB wave copies and read
//    with conflict-free lane-linear ds_read_b128; the ring runs 3 steps ahead and straight through phase and tile boundaries;
//  * pixel fragments are ds_read_b128 at (per-phase lane base + compile-time offset): odd 16-byte pixel pitch, no swizzle;
//  * D = W x X^T orientation: a lane ends up with 4+4 consecutive output channels of one pixel, v_permlane32_swap makes that 8 -- the
//    epilogue (bias, PReLU, residual add / subtract, accumulate, activation-derivative mask) runs in registers with 16-byte loads and
//    stores, no LDS staging.
//
// vmcnt discipline: loads complete in order, stores do not order against them, so every counted wait below only ever has stores that
// are OLDER than the load it waits for (safe, at worst it also sits out their acknowledgement), and the stages the next phase starts
// on are confirmed (vmcnt(0) + barrier) before the epilogue issues its stores.
#include "common.h"
#include "conv_common.h"
#include "csbsr_debug.h"
#include <cstdlib>

#define TP_TH 8
#define TP_TW 32
#define TP_HW (TP_TW + 2)
#define TP_HH (TP_TH + 2)
#define TP_NPIX (TP_HH * TP_HW)          // 340 halo pixels
#define TP_NST 4                         // weight ring stages
#define TP_STAGE 16384                   // bytes per stage: 128 couts x 64 channels
#define TP_MAXPG 8                       // phases per work item

struct ConvTpK {
  const half_t* in; long i_sn, i_sy, i_sx;
  int N, H, W;                      // input (low-resolution) size; output = stride x that
  int stride, pad;
  const half_t* wt;                 // [phase][4 taps x NKC][16 KB fragment image]
  int cout, coutp;
  half_t* out16; long o_sn, o_sy, o_sx;
  const float* bias; int act; float slope; const float* prelu; float out_scale;
  int res_mode; const half_t* res; long r_sn, r_sy, r_sx;
  int accumulate;
  const half_t* mask; long m_sn, m_sy, m_sx; float mask_slope;
  unsigned tiles_x, tiles_y, pg, pgroups;     // phases per work item, items per tile
  int dbg;                          // ablation bits (CSBSR_TP_DBG): 1 no stores, 2 no K-loop math, 4 no weight DMA, 8 / 16 no A / B fragment reads
};

__device__ __forceinline__ void tp_wait_barrier_all() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int NKC>
__global__ __launch_bounds__(512) void conv_tp_kernel(const ConvTpK p, const half_t* __restrict__ zero_page) {
  constexpr int SLOTS = NKC * 8 + 1;                    // 16-byte slots per pixel: odd -> consecutive pixels walk all banks
  constexpr int PITCH = SLOTS * 16;
  constexpr int NG = TP_NPIX * SLOTS;
  constexpr int NINST = (NG + 63) / 64;                 // wave instructions that fill the halo tile
  constexpr int XBYTES = NINST * 1024;
  constexpr int WOFF = XBYTES;
  constexpr int BOFF = WOFF + TP_NST * TP_STAGE;        // 128 biases
  constexpr int NKS = 4 * NKC;                          // K steps per phase (64 channels of one tap each)
  constexpr int NFI = (NINST + 7) / 8;
  constexpr int DQ = 512 / SLOTS, DC = 512 % SLOTS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sBias = reinterpret_cast<float*>(smem + BOFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pix = lane & 31, hi = lane >> 5;
  const int mh = wid & 1, pr = wid >> 1;                // cout half, row pair of the tile
  const half_t* zp = zero_page + (lane & 7) * 8;
  const int s = p.stride;
  const unsigned per_img = p.tiles_x * p.tiles_y, ntiles = per_img * (unsigned)p.N, items = ntiles * p.pgroups;
  unsigned it = blockIdx.x;
  if (it >= items) return;
  if (tid < 128) sBias[tid] = (p.bias && tid < p.cout) ? p.bias[tid] : 0.f;
  const float slope = p.act == CSBSR_ACT_PRELU ? *p.prelu : (p.act == CSBSR_ACT_RELU ? 0.f : (p.act == CSBSR_ACT_NONE ? 1.f : p.slope));
  const float rsign = p.res_mode == CSBSR_RES_ADD ? 1.f : (p.res_mode == CSBSR_RES_SUB ? -1.f : 0.f);

  // halo-tile DMA roles (same for every tile): chunk g = (wid + 8 i) * 64 + lane = (halo pixel q, slot c)
  int f_ty0, f_tx0, f_c0;
  {
    const int g = wid * 64 + lane;
    const int q = g / SLOTS;
    f_c0 = g - q * SLOTS;
    f_ty0 = q / TP_HW;
    f_tx0 = q - f_ty0 * TP_HW;
  }
  const int isy = (int)p.i_sy, isx = (int)p.i_sx;
  const long stage_elems = TP_STAGE / 2;
  // one weight stage: 16 wave instructions, this wave's two
  auto issue_w = [&](const half_t* src, int slot) {
    if (p.dbg & 4) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int blk = 2 * wid + j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + blk * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(smem + WOFF + slot * TP_STAGE + blk * 1024), 16, 0, 0);
    }
  };
  const char* wl = smem + WOFF + mh * 8192 + lane * 16;   // A fragments: + slot * 16384 + mt * 4096 + kk * 1024
  const char* xl = smem + pix * PITCH + hi * 16;          // B fragments: + per-phase pixel base + compile-time (tap, row, channel) offset

  {                                                       // ring prologue: the first phase's stages 0..2
    const half_t* w0 = p.wt + (size_t)((it / ntiles) * p.pg) * NKS * stage_elems;
    issue_w(w0, 0); issue_w(w0 + stage_elems, 1); issue_w(w0 + 2 * stage_elems, 2);
  }

  for (; it < items; it += gridDim.x) {
    const unsigned pgi = it / ntiles, tile = it - pgi * ntiles;
    const int n = tile / per_img;
    const unsigned r_ = tile - n * per_img;
    const int Y0 = (r_ / p.tiles_x) * TP_TH, X0 = (r_ % p.tiles_x) * TP_TW;
    const unsigned itn = it + gridDim.x;
    // ---- halo tile -> LDS (every wave is past the previous item's last K step: end-of-phase barrier)
    {
      const half_t* tbase = p.in + n * p.i_sn + (long)(Y0 - 1) * p.i_sy + (long)(X0 - 1) * p.i_sx;
      int ty = f_ty0, tx = f_tx0, c = f_c0;
#pragma unroll 2
      for (int i = 0; i < NFI; ++i) {
        const int inst = wid + 8 * i;
        if (inst < NINST) {
          const int iy = Y0 - 1 + ty, ix = X0 - 1 + tx;
          const bool ok = ty < TP_HH && c < NKC * 8 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          const half_t* src = ok ? tbase + (ty * isy + tx * isx + c * 8) : zp;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(smem + inst * 1024), 16, 0, 0);
        }
        c += DC; tx += DQ;
        if (c >= SLOTS) { c -= SLOTS; ++tx; }
        if (tx >= TP_HW) { tx -= TP_HW; ++ty; }
        if (tx >= TP_HW) { tx -= TP_HW; ++ty; }
      }
    }

    for (unsigned j = 0; j < p.pg; ++j) {
      const unsigned ph = pgi * p.pg + j;
      const int py = ph / s, px = ph - py * s;
      const int by = (py + p.pad) / s, bx = (px + p.pad) / s;
      const unsigned phn = (j + 1 < p.pg) ? ph + 1 : (itn < items ? (itn / ntiles) * p.pg : ph);   // (last phase of all: harmless refetch)
      const half_t* wcur = p.wt + (size_t)ph * NKS * stage_elems;
      const half_t* wnext = p.wt + (size_t)phn * NKS * stage_elems;
      const char* xph = xl + ((2 * pr + by) * TP_HW + bx) * PITCH;

      f16v acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        // stage ks landed everywhere + every wave is done with step ks-1 (whose slot is refilled below)
        if (ks == 0) { if (j == 0) tp_wait_barrier_all(); }
        else if (ks < 3) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        else { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        issue_w(ks + 3 < NKS ? wcur + (ks + 3) * stage_elems : wnext + (ks + 3 - NKS) * stage_elems, (ks + 3) % TP_NST);
        const int tap = ks / NKC, kc = ks % NKC;
        const int jy = tap >> 1, jx = tap & 1;
        const int xo = ((1 - jy) * TP_HW + (1 - jx)) * PITCH + kc * 128;
        const int wo = (ks % TP_NST) * TP_STAGE;
        if (!(p.dbg & 2)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          h8 a0, a1, b0, b1;
          if (!(p.dbg & 8)) {
            a0 = *reinterpret_cast<const h8*>(wl + wo + kk * 1024);
            a1 = *reinterpret_cast<const h8*>(wl + wo + 4096 + kk * 1024);
          } else { a0 = a1 = h8{1, 1, 1, 1, 1, 1, 1, 1} * (half_t)slope; }
          if (!(p.dbg & 16)) {
            b0 = *reinterpret_cast<const h8*>(xph + xo + kk * 32);
            b1 = *reinterpret_cast<const h8*>(xph + xo + TP_HW * PITCH + kk * 32);
          } else { b0 = b1 = h8{1, 1, 1, 1, 1, 1, 1, 1} * (half_t)rsign; }
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
        }
        }
      }
      // the next phase's first three stages are in flight: confirm them (and that every wave has left the K loop) before any store
      tp_wait_barrier_all();

      // ---- epilogue: acc[mt][nt][4q + jj] = cout 64 mh + 32 mt + 8 q + 4 hi + jj of pixel (row 2 pr + nt, column pix)
      const int qx = X0 + pix;
      const long ox = (long)s * qx + px;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int qy = Y0 + 2 * pr + nt;
        const bool live = qy < p.H && qx < p.W;
        const long oy = (long)s * qy + py;
        const long ooff = n * p.o_sn + oy * p.o_sy + ox * p.o_sx + 64 * mh + 8 * hi;
        const long roff = n * p.r_sn + oy * p.r_sy + ox * p.r_sx + 64 * mh + 8 * hi;
        const long moff = n * p.m_sn + oy * p.m_sy + ox * p.m_sx + 64 * mh + 8 * hi;
        h8 rr[4], oo[4], mm[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {          // piece q = (mt, pair): couts 64 mh + 16 q + 8 hi ..
          const bool lq = live && 64 * mh + 16 * q + 8 * hi < p.coutp;
          if (rsign != 0.f && lq) rr[q] = *reinterpret_cast<const h8*>(p.res + roff + 16 * q);
          if (p.accumulate && lq) oo[q] = *reinterpret_cast<const h8*>(p.out16 + ooff + 16 * q);
          if (p.mask && lq) mm[q] = *reinterpret_cast<const h8*>(p.mask + moff + 16 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int mt = q >> 1, pair = q & 1;
          float v[8];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const unsigned a = __float_as_uint(acc[mt][nt][8 * pair + jj]), b = __float_as_uint(acc[mt][nt][8 * pair + 4 + jj]);
            auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
            v[jj] = __uint_as_float(r[0]);
            v[4 + jj] = __uint_as_float(r[1]);
          }
          const int co = 64 * mh + 16 * q + 8 * hi;
          const f4 b0 = *reinterpret_cast<const f4*>(sBias + co), b1 = *reinterpret_cast<const f4*>(sBias + co + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = v[e] * p.out_scale + (e < 4 ? b0[e & 3] : b1[e & 3]);
            t = t > 0.f ? t : t * slope;                   // identity 1, ReLU 0, leaky / PReLU slope
            v[e] = (co + e < p.cout) ? t : 0.f;
          }
          if (live && co < p.coutp) {
            if (rsign != 0.f) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rsign * (float)rr[q][e];
            }
            if (p.accumulate) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += (float)oo[q][e];
            }
            if (p.mask) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= ((float)mm[q][e] > 0.f ? 1.f : p.mask_slope);
            }
            h8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
            if (!(p.dbg & 1) || v[0] == 12345.678f) *reinterpret_cast<h8*>(p.out16 + ooff + 16 * q) = hv;
          }
        }
      }
    }      // phases
  }        // items
}

// ---- weights in ring-stage order: dst[phase][ks = tap * NKC + kc][mt][kk][lane][e] =
//        W(cout 32 mt + lane % 32, channel 64 kc + 16 kk + 8 (lane / 32) + e, kh = (py + pad) % s + s jy, kw likewise), tap = 2 jy + jx
// W is indexed [contracted channel][row][kh][kw]: ConvTranspose2d's IOHW parameter, or a Conv2d's OIHW parameter seen from its dgrad
// (contracted = the conv's output channels, rows = its input channels) -- csbsr_pack_weights kind 2 in this kernel's order.
struct PackTpK { const float* w; half_t* dst; int D1, KH, KW, stride, pad, nkc, c_real, rows_real, row_off, k_off; };
__global__ void pack_weights_tp_kernel(const PackTpK p, long total) {
  const int nks = 4 * p.nkc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), blk = (int)((i >> 9) & 15);
    const long stage = i >> 13;
    const int ks = (int)(stage % nks), ph = (int)(stage / nks);
    const int mt = blk >> 2, kk = blk & 3;
    const int row = 32 * mt + (lane & 31);
    const int tap = ks / p.nkc, kc = ks % p.nkc;
    const int c = 64 * kc + 16 * kk + 8 * (lane >> 5) + e;
    const int jy = tap >> 1, jx = tap & 1;
    const int py = ph / p.stride, px = ph % p.stride;
    const int kh = (py + p.pad) % p.stride + p.stride * jy, kw = (px + p.pad) % p.stride + p.stride * jx;
    float v = 0.f;
    if (row < p.rows_real && c < p.c_real && kh < p.KH && kw < p.KW)
      v = p.w[(((long)(p.k_off + c) * p.D1 + p.row_off + row) * p.KH + kh) * p.KW + kw];
    p.dst[i] = (half_t)v;
  }
}

static int tp_nkc(int cin_padded) { return cin_padded == 64 ? 1 : (cin_padded == 128 ? 2 : 0); }

extern "C" int64_t csbsr_packed_weight_elems_tp(int32_t stride, int32_t c_real) {
  const int nkc = tp_nkc(round_up(c_real, 8) <= 64 ? 64 : 128);
  return (int64_t)stride * stride * 4 * nkc * (TP_STAGE / 2);
}

extern "C" int csbsr_pack_weights_tp(const float* w, void* dst, int32_t D0, int32_t D1, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                                     int32_t c_real, int32_t rows_real, int32_t row_off, int32_t k_off, csbsr_stream_t s) {
  CSBSR_CHECK(w && dst, "pack_tp: null pointer");
  CSBSR_CHECK(stride >= 2 && KH == KW && KH > stride && KH <= 2 * stride && pad >= 0 && pad < stride, "pack_tp: needs stride < kernel <= 2 stride");
  CSBSR_CHECK(c_real >= 1 && c_real <= 128 && rows_real >= 1 && rows_real <= 128, "pack_tp: at most 128 channels either side");
  CSBSR_CHECK(k_off >= 0 && k_off + c_real <= D0 && row_off >= 0 && row_off + rows_real <= D1, "pack_tp: range out of bounds");
  PackTpK p;
  p.w = w; p.dst = reinterpret_cast<half_t*>(dst); p.D1 = D1; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.nkc = tp_nkc(round_up(c_real, 8) <= 64 ? 64 : 128);
  p.c_real = c_real; p.rows_real = rows_real; p.row_off = row_off; p.k_off = k_off;
  const long total = csbsr_packed_weight_elems_tp(stride, c_real);
  hipLaunchKernelGGL(pack_weights_tp_kernel, dim3((int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(s), p, total);
  CSBSR_LAUNCH_CHECK("csbsr_pack_weights_tp");
  return 0;
}

static int g_conv_tp_mode = 1;      // 0 off, 1 problems that fill the chip, 2 every eligible launch (tests)
extern "C" void csbsr_debug_set_conv_tp(int mode) { g_conv_tp_mode = mode; }

// Which launches take this kernel: transposed, stride < kernel <= 2 stride (2 x 2 taps per phase), output exactly stride x input,
// one plain-fp16 input segment of 64 or 128 padded channels, 65..128 (padded: 72..128) output channels, bias / ReLU / leaky / PReLU /
// residual add or subtract / accumulate / activation mask; no fp32 or split outputs, statistics or constant-segment bias.
extern "C" int32_t csbsr_conv_tp_eligible(const csbsr_conv_desc_t* d) {
  if (!d || !g_conv_tp_mode || !d->transposed || d->KH != d->KW || d->stride < 2 || d->KH <= d->stride || d->KH > 2 * d->stride) return 0;
  if (d->pad < 0 || d->pad >= d->stride || d->dil != 1 || d->OH != d->H * d->stride || d->OW != d->W * d->stride) return 0;
  if (d->in[1].c != 0 || d->in[0].sx == 0 || (d->in[0].c != 64 && d->in[0].c != 128)) return 0;
  if (d->coutp <= 64 || d->coutp > 128 || !d->out16 || d->out32 || d->cbias || d->o_lo || d->r_lo || d->r2_lo) return 0;
  if (d->stat_mode != CSBSR_STAT_NONE) return 0;
  if (d->res_mode != CSBSR_RES_NONE && d->res_mode != CSBSR_RES_ADD && d->res_mode != CSBSR_RES_SUB) return 0;
  if (d->act == CSBSR_ACT_SIGMOID) return 0;
  if (d->in[0].sn >= (1ll << 31) || d->o_sn >= (1ll << 40)) return 0;
  if (g_conv_tp_mode == 1 && (long)d->N * d->H * d->W < 128L * TP_TH * TP_TW) return 0;
  return 1;
}

static half_t* g_tp_zero_page[CSBSR_MAX_DEVICES] = {};

template <int NKC>
static int launch_tp(const ConvTpK& k, hipStream_t st, const half_t* zp) {
  constexpr int SLOTS = NKC * 8 + 1;
  constexpr int NINST = (TP_NPIX * SLOTS + 63) / 64;
  constexpr int SM_BYTES = NINST * 1024 + TP_NST * TP_STAGE + 128 * 4;
  static_assert(SM_BYTES <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tp_kernel<NKC>), hipFuncAttributeMaxDynamicSharedMemorySize, SM_BYTES);
    attr_set = true;
  }
  const unsigned items = k.tiles_x * k.tiles_y * k.N * k.pgroups;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  const unsigned g = items < (unsigned)ncu ? items : (unsigned)ncu;
  hipLaunchKernelGGL((conv_tp_kernel<NKC>), dim3(g), dim3(512), SM_BYTES, st, k, zp);
  CSBSR_LAUNCH_CHECK("csbsr_conv_tp_forward");
  return 0;
}

extern "C" int csbsr_conv_tp_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s) {
  CSBSR_CHECK(csbsr_conv_tp_eligible(d), "conv_tp: launch not eligible (see csbsr_conv_tp_eligible)");
  CSBSR_CHECK(d->in[0].ptr && d->wt, "conv_tp: null pointer");
  CSBSR_CHECK(d->act != CSBSR_ACT_PRELU || d->prelu, "conv_tp: PReLU without a slope");
  ConvTpK k;
  k.in = reinterpret_cast<const half_t*>(d->in[0].ptr); k.i_sn = d->in[0].sn; k.i_sy = d->in[0].sy; k.i_sx = d->in[0].sx;
  k.N = d->N; k.H = d->H; k.W = d->W; k.stride = d->stride; k.pad = d->pad;
  k.wt = reinterpret_cast<const half_t*>(d->wt);
  k.cout = d->cout; k.coutp = d->coutp;
  k.out16 = reinterpret_cast<half_t*>(d->out16); k.o_sn = d->o_sn; k.o_sy = d->o_sy; k.o_sx = d->o_sx;
  k.bias = d->bias; k.act = d->act; k.slope = d->act_slope; k.prelu = d->prelu; k.out_scale = d->out_scale;
  k.res_mode = d->res_mode; k.res = reinterpret_cast<const half_t*>(d->res); k.r_sn = d->r_sn; k.r_sy = d->r_sy; k.r_sx = d->r_sx;
  k.accumulate = d->accumulate;
  k.mask = reinterpret_cast<const half_t*>(d->mask); k.m_sn = d->m_sn; k.m_sy = d->m_sy; k.m_sx = d->m_sx; k.mask_slope = d->mask_slope;
  k.tiles_x = (unsigned)((d->W + TP_TW - 1) / TP_TW); k.tiles_y = (unsigned)((d->H + TP_TH - 1) / TP_TH);
  const unsigned nphase = (unsigned)(d->stride * d->stride);
  k.pg = nphase < TP_MAXPG ? nphase : TP_MAXPG;
  while (nphase % k.pg) --k.pg;
  k.pgroups = nphase / k.pg;
  { const char* e = getenv("CSBSR_TP_DBG"); k.dbg = e ? atoi(e) : 0; }
  int dev = 0;
  CSBSR_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < CSBSR_MAX_DEVICES, "conv_tp: no current device");
  if (!g_tp_zero_page[dev]) {
    CSBSR_CHECK(hipMalloc(reinterpret_cast<void**>(&g_tp_zero_page[dev]), 256) == hipSuccess, "conv_tp: zero page alloc failed");
    (void)hipMemset(g_tp_zero_page[dev], 0, 256);
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  g_last_conv_kernel = CONVK_TP;
  if (d->in[0].c == 64) return launch_tp<1>(k, st, g_tp_zero_page[dev]);
  return launch_tp<2>(k, st, g_tp_zero_page[dev]);
}
