// 3x3 stride-1 convolution from MANY input channels into <= 64 output channels at full resolution -- PSPNet_BlurSkip's conv_scale.1 /
// conv_shift.1 (505 -> 64 at 1792^2, /root/reference/model/modeling/blocks.py:105-120; 20 % of a config-5 step) and the dgrads of its
// 64 -> 505 conv0's -- built like csrc/conv_x3.hip (resident pixel tile, weights streamed from L2 in fragment order) with the tile turned
// around for a NARROW output: the LDS-DMA implicit-GEMM kernel runs these layers on a 128-pixel x 64-cout tile, stages every input pixel
// once per tap and measured 550-590 TF/s (43 flop per staged byte; DESIGN.md section 7.9 of round 5).  Here
//
//  * one persistent workgroup per CU (4 waves) computes an 8-row x 64-pixel x 64-cout tile; a wave owns ALL 64 couts x 4 rows x 32 pixels
//    (2 x 4 MFMA tiles, 128 accumulator registers): the four waves are 2 row quads x 2 column halves and stream the same 4 KB of
//    weights per K step (the second to fourth request hit L1);
//  * the pixel operand is staged per 32-channel chunk: the chunk's (8+2) x (64+2) halo goes HBM / L2 -> LDS with buffer-load LDS-DMA
//    (zeros outside the image from the bounds check) into one of two 52 KB buffers, one DMA piece per k-slice inside the previous chunk's
//    K loop; every tap reads the same tile at a different offset, so an input pixel crosses the fabric 1.29 times instead of 9;
//    pixel pitch 80 bytes (4 sixteen-byte channel slots + 1 pad: conflict-free ds_read_b128 rows);
//  * split-precision inputs (the detector's hi + lo activation pairs) run the two-product plan  [x_hi | x_lo] w_hi  as a plain
//    convolution over 2 x Cp input channels: the pack repeats the (tap-sum-rounded, pre-scaled) weights for the lo plane (``dup``), the
//    epilogue stores hi + lo pairs (conv_common.h split_store) -- no kernel code knows about the planes;
//  * synchronisation: one barrier and one counted s_waitcnt per chunk (nine K steps = 144 MFMAs per wave); the general fused epilogue
//    rows of conv_common.h.
#include "common.h"
#include "conv_common.h"
#include "csbsr_debug.h"

#ifndef XN_ABL
#define XN_ABL 0                          // timing ablations (variant builds only): 1 no epilogue, 2 no MFMAs, 4 one wide workgroup per CU, 16 lean stores dropped
#endif
#define XN_TH 8
#define XN_HH (XN_TH + 2)
#define XN_SLOTS 5                        // 32 channels = 4 sixteen-byte slots + 1 pad slot
#define XN_PITCH (XN_SLOTS * 16)
#define XN_TPITCH 144                     // pitch of a pixel in the wide form's output transposition tile: 64 couts x 2 bytes + 16
#define XN_NT 9
#define XN_RING 3
#define XN_DIST 2
// The tile geometry of the two forms.  NARROW (<= 64 couts): 8 x 64 pixels x 64 couts, one workgroup per CU.  WIDE (> 64 couts from 64 .. 128
// input channels -- BlurSkip's 64 -> 505 conv0's and the dgrads of its conv1's, one or two chunks of K per tile, where the fused epilogue of
// 128 accumulator registers per lane is a third of a tile's time): 8 x 32 pixels x 128 couts, the four waves 2 cout halves x 2 row quads as
// in csrc/conv_x3.hip, 56 KB of LDS and <= 256 registers so that TWO workgroups share a CU -- one's epilogue (VALU + stores) runs under the
// other's K loop (MFMA), which a single workgroup's program order cannot do.
template <bool WIDE> struct XNGeo {
  static constexpr int TW = WIDE ? 32 : 64;               // tile width in pixels
  static constexpr int HW = TW + 2;
  static constexpr int NINST = WIDE ? 28 : 52;            // wave instructions that fill the halo tile: ceil(10 * HW * 5 / 64) rounded to 4
  static constexpr int BUF = NINST * 1024;
  static constexpr int CT = WIDE ? 128 : 64;              // couts per tile
  static constexpr int WSTEP = WIDE ? 8192 : 4096;        // bytes of one K step's weights: [mh 2 (wide)][mt 2][kk 2][lane][8]
  static_assert(XN_HH * HW * XN_SLOTS <= NINST * 64, "halo tile fits its DMA instructions");
};

struct XNExtra {
  unsigned tiles_x, tiles_y, nct, nch;   // pixel tiles, cout tiles (64 or 128 channels), 32-channel chunks
};

#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t xn_make_rs(const half_t* base) {
  const unsigned long a = reinterpret_cast<unsigned long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi_ << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
#endif

// FAST: straight-line epilogue rows (conv_epilogue_fast_ok), else the general fused row.  BNS (with FAST, one cout tile): fused BatchNorm
// statistics -- every lane keeps the sum and the sum of squares of its 32 output channels over all its tiles in registers, the lanes of a
// wave fold by xor-shuffles, the four waves through LDS in wave order, and the workgroup writes ONE partial row; the launcher folds the rows
// in a fixed tree (csbsr_sum_partials): order-fixed like every reduction of the library.
// LEAN (the wide form with FAST): the launch has no per-pixel epilogue operand (residual, accumulated-into output, activation mask); its epilogue
// issues no vector-memory load between its stores (below).
template <bool FAST, bool BNS = false, bool WIDE = false, bool LEAN = false>
__global__ __launch_bounds__(256, WIDE ? 2 : 1) void conv_x3n_kernel(const ConvK p, const XNExtra q) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(!(BNS && WIDE), "fused BatchNorm sums: the narrow form");
  static_assert(!LEAN || (WIDE && FAST), "LEAN: an instance of the wide form's straight-line rows");
  using G = XNGeo<WIDE>;
  constexpr int XN_TW = G::TW, XN_HW = G::HW, XN_NINST = G::NINST, XN_BUF = G::BUF, XN_WSTEP = G::WSTEP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pix = lane & 31, hi = lane >> 5;
  const int chf = WIDE ? 0 : (wid & 1), rq = wid >> 1;     // column half (32 pixels; narrow), row quad
  const int mh = WIDE ? (wid & 1) : 0;                     // cout half of the 128-cout tile (wide)
  const unsigned per_img = q.tiles_x * q.tiles_y, ntiles = per_img * (unsigned)p.N, items = ntiles * q.nct;
  unsigned it = blockIdx.x;
  if (it >= items) return;
  const float slope = (p.act == CSBSR_ACT_PRELU) ? *p.prelu : p.act_slope;
  const EpiFast fe = conv_epilogue_fast_setup(p, slope);
  const half_t* in0 = reinterpret_cast<const half_t*>(p.in[0].ptr);
  const int isy = (int)p.in[0].sy, isx = (int)p.in[0].sx;

  // halo DMA roles (the same for every tile and chunk): instruction i of this wave fills LDS bytes (wid + 4 i) * 1024 + lane * 16 =
  // chunk-slot g = (wid + 4 i) * 64 + lane = (halo pixel g / 5, slot g % 5); slot 4 is the pad
  constexpr int NFI = XN_NINST / 4;                  // 13 instructions per wave
  constexpr int DQ = 256 / XN_SLOTS, DC = 256 % XN_SLOTS;
  int voff[NFI], iy0[NFI], ix0[NFI];
  {
    const int g = wid * 64 + lane;
    const int hq = g / XN_SLOTS;
    int c = g - hq * XN_SLOTS, ty = hq / XN_HW, tx = hq - ty * XN_HW;
#pragma unroll
    for (int i = 0; i < NFI; ++i) {
      const bool in = ty < XN_HH && c < 4;
      voff[i] = 2 * (ty * isy + tx * isx + c * 8);
      iy0[i] = in ? ty : 0x40000000;
      ix0[i] = tx;
      c += DC; tx += DQ;
      if (c >= XN_SLOTS) { c -= XN_SLOTS; ++tx; }
      if (tx >= XN_HW) { tx -= XN_HW; ++ty; }
      if (tx >= XN_HW) { tx -= XN_HW; ++ty; }
    }
  }
  // (uniform) address of halo pixel (0, 0), channel 32 chunk, of tile (n, Y0, X0): input pixel (Y0 - 1, X0 - 1)
  auto chunk_src = [&](int n, int Y0, int X0, int chunk) -> const half_t* {
    return in0 + n * p.in[0].sn + (long)(Y0 - 1) * p.in[0].sy + (long)(X0 - 1) * p.in[0].sx + chunk * 32;
  };
  auto issue_one = [&](__amdgpu_buffer_rsrc_t rs, int by, int bx, int i, int buf) __attribute__((always_inline)) {
    const bool ok = (unsigned)(iy0[i] + by) < (unsigned)p.H && (unsigned)(ix0[i] + bx) < (unsigned)p.W;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + buf * XN_BUF + (wid + 4 * i) * 1024), 16,
                                             ok ? voff[i] : -1, 0, 0, 0);
  };
  auto decode = [&](unsigned item, int& ct, int& n, int& Y0, int& X0) {
    item = xcd_remap(item, items);           // the cout tiles of one pixel tile side by side on one XCD (csrc/conv_x3.hip)
    const unsigned tile = item / q.nct;
    ct = item - tile * q.nct;
    n = tile / per_img;
    const unsigned r_ = tile - n * per_img;
    Y0 = (r_ / q.tiles_x) * XN_TH; X0 = (r_ % q.tiles_x) * XN_TW;
  };
  // the weights of K step (ct, chunk, tap): [mt 2][kk 2] fragments, one 16-byte load per lane each (narrow: the same 4 KB for all four
  // waves; wide: the wave's cout half of 8 KB)
  const unsigned wlane = (unsigned)(lane * 16 + mh * 4096);
  auto load_w = [&](int ct, int step, h8 (&w)[2][2]) __attribute__((always_inline)) {
    const char* b = reinterpret_cast<const char*>(p.wt) + ((size_t)ct * q.nch * XN_NT + step) * XN_WSTEP;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) w[mt][kk] = *reinterpret_cast<const h8*>(b + (wlane + (mt * 2 + kk) * 1024));
  };
  const char* xl = smem + ((4 * rq) * XN_HW + 32 * chf + pix) * XN_PITCH + hi * 16;      // the wave's first row / column of the halo tile

  float bsum[4][8], bsq[4][8];          // BNS: [mt * 2 + pair][e] = channel 32 mt + 16 pair + 8 hi + e
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[a][e] = bsq[a][e] = 0.f;

  int ct, n, Y0, X0;
  decode(it, ct, n, Y0, X0);
  {
    const __amdgpu_buffer_rsrc_t rs = xn_make_rs(chunk_src(n, Y0, X0, 0));
#pragma unroll
    for (int i = 0; i < NFI; ++i) issue_one(rs, Y0 - 1, X0 - 1, i, 0);
  }
  h8 wreg[XN_RING][2][2];                                 // K step g = chunk * 9 + tap lives in wreg[tap % 3]
#pragma unroll
  for (int g = 0; g < XN_DIST; ++g) load_w(ct, g, wreg[g]);
  const int nsteps = (int)q.nch * XN_NT;

  for (; it < items; it += gridDim.x) {
    const unsigned itn = it + gridDim.x;
    int ctn = ct, nn = n, Y0n = Y0, X0n = X0;
    if (itn < items) decode(itn, ctn, nn, Y0n, X0n);
    const unsigned par = ((it - blockIdx.x) / gridDim.x) * q.nch;     // chunk c of this tile lives in buffer (par + c) & 1
    f16v acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if constexpr (WIDE) __builtin_amdgcn_s_setprio(1);      // the K loop's MFMA issue ahead of the co-resident workgroup's epilogue
    // WIDE: the tile's bias (plus the interior class row of a position-class bias) for the wave's 64 couts, one value per lane, loaded here
    // and handed to the epilogue through LDS: no register of the 256 is free for 32 bias values, and a load issued between the
    // epilogue's stores would have to sit out their write acknowledgement (conv_common.h, conv_epilogue_fast_tile)
    float bv = 0.f, bcv = 0.f;
    const bool interior = Y0 + 4 * rq >= 2 && Y0 + 4 * rq + 3 <= p.OH - 3 && X0 + 32 * chf >= 2 && X0 + 32 * chf + 31 <= p.OW - 3;      // (wave-uniform: position class 0 everywhere)
    if constexpr (LEAN) {
      const int cl = G::CT * ct + 64 * mh + lane;
      const int clc = cl < p.cout ? cl : p.cout - 1;
      if (p.bias) bv = p.bias[n * p.bias_sn + clc];
      if (fe.has_cb && interior) bcv = p.cbias[((size_t)n * (p.cb_mode == 0 ? 16 : 25) + (p.cb_mode == 0 ? 0 : 12)) * p.coutp + (cl < p.coutp ? cl : p.coutp - 1)];
      // (combined and written to the wave's LDS slot at the end of the first chunk: a counted wait, nine taps later)
    }
    for (int c = 0; c < (int)q.nch; ++c) {
      // chunk c's halo has landed everywhere, and every wave is done with the other buffer (chunk c - 1): refill that one.  The only
      // vector-memory instructions issued after this chunk's last DMA piece (k-slice NFI - 1 = tap 6 of the previous chunk) are the
      // weight loads of the taps that followed it (4 each) -- except right after an epilogue, where the count is simply drained.
      // The first chunk of a tile that follows an epilogue of the straight-line rows: that epilogue's CONV_TILE_STORES stores (at least:
      // unconditional buffer stores, conv_common.h) were issued after the DMA pieces too and may stay in flight -- draining them here
      // (vmcnt(0)) meant sitting out their write acknowledgement once per tile.
      if (c == 0) {
        if (((FAST && !WIDE) || LEAN) && it != blockIdx.x) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (XN_NT - 1 - (NFI - 1) / 2) + CONV_TILE_STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (XN_NT - 1 - (NFI - 1) / 2)) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int bnext = (int)((par + c + 1) & 1);
      const bool same = c + 1 < (int)q.nch;
      const int sn_ = same ? n : (itn < items ? nn : n), sY = same ? Y0 : (itn < items ? Y0n : Y0), sX = same ? X0 : (itn < items ? X0n : X0);
      const __amdgpu_buffer_rsrc_t nrs = xn_make_rs(chunk_src(sn_, sY, sX, same ? c + 1 : 0));
      asm volatile("" ::: "memory");
      const char* xb = xl + ((par + c) & 1) * XN_BUF;
      h8 bfr[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bfr[nt] = *reinterpret_cast<const h8*>(xb + (nt * XN_HW) * XN_PITCH);      // tap (0, 0), k-slice 0
#pragma unroll
      for (int tap = 0; tap < XN_NT; ++tap) {
        {   // weights XN_DIST K steps ahead (past the tile's last step: the next tile's first ones)
          const int g = c * XN_NT + tap + XN_DIST;
          if (g < nsteps) load_w(ct, g, wreg[(tap + XN_DIST) % XN_RING]);
          else load_w(ctn, g - nsteps, wreg[(tap + XN_DIST) % XN_RING]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#if XN_ABL & 2
            acc[0][i][0] += (float)wreg[tap % XN_RING][0][kk][0] * (float)bfr[i][0];
            acc[1][i][0] += (float)wreg[tap % XN_RING][1][kk][0] * (float)bfr[i][0];
#else
            acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap % XN_RING][0][kk], bfr[i], acc[0][i], 0, 0, 0);
            acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap % XN_RING][1][kk], bfr[i], acc[1][i], 0, 0, 0);
#endif
            // the same row's fragment of the next k-slice / next tap (the next chunk starts over after its barrier)
            if (kk < 1) bfr[i] = *reinterpret_cast<const h8*>(xb + ((i + ky) * XN_HW + kx) * XN_PITCH + (kk + 1) * 32);
            else if (tap < XN_NT - 1) bfr[i] = *reinterpret_cast<const h8*>(xb + ((i + (tap + 1) / 3) * XN_HW + (tap + 1) % 3) * XN_PITCH);
            if (i == 1 && tap * 2 + kk < NFI) issue_one(nrs, sY - 1, sX - 1, tap * 2 + kk, bnext);       // one DMA piece per k-slice
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if constexpr (LEAN) {
        if (c == 0) reinterpret_cast<float*>(smem + 2 * XN_BUF)[wid * 64 + lane] = (G::CT * ct + 64 * mh + lane < p.cout) ? bv + bcv : 0.f;
      }
    }

    if constexpr (WIDE) __builtin_amdgcn_s_setprio(0);
    // ---- epilogue: acc[mt][nt][8 pair + e] = cout CT ct + 64 mh + 32 mt + 16 pair + 8 hi + e of pixel (row 4 rq + nt, column 32 chf + pix)
    const int ox = X0 + 32 * chf + pix;
#if XN_ABL & 1
    if (p.N == 12345)
#endif
    if constexpr (FAST && !WIDE) {
      // straight-line rows with every load ahead of the stores (conv_common.h)
      conv_epilogue_fast_tile<BNS>(p, fe, acc, G::CT * ct + 64 * mh + 8 * hi, n, Y0 + 4 * rq, ox, interior, bsum, bsq);
    } else if constexpr (LEAN) {
      // the wide form's launches without per-pixel operands: bias through LDS (no vector-memory load in the epilogue; border tiles of a
      // class-bias launch load the pixel's class row per piece), and the output TRANSPOSED through a wave-private LDS tile so that a store
      // instruction writes whole lines: in the accumulator layout a lane owns 16 bytes of a pixel and its neighbour lane the next PIXEL
      // (coutp x 2 bytes away) -- 64 separate 16-byte write requests per instruction, and the 64 -> 505 layer (13 GB of output) ran at the
      // L2's request rate: 9.1 ms per launch against 6.8 with coalesced stores and 5.6 with none (profiles/r06_x3n_ablation.txt).  Row by
      // row: the four pieces of a row go to LDS as [pixel][64 couts] (pitch 144 bytes: conflict-free 16-byte writes), come back as 8 lanes per
      // pixel, and leave as four stores of 8 pixels x 128 bytes.  Unconditional buffer stores (conv_common.h).
      const __amdgpu_buffer_rsrc_t rs = conv_make_rs(p.out16 + n * p.o_sn + (long)(Y0 + 4 * rq) * p.o_sy);
      const int oxc = ox < p.OW ? ox : p.OW - 1;
      const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      float s0[8], s1[8];
      const float* bl = reinterpret_cast<const float*>(smem + 2 * XN_BUF) + wid * 64;
      char* tt = smem + 2 * XN_BUF + 1024 + wid * (32 * XN_TPITCH);
      const bool border_cb = fe.has_cb && !interior;
      const int rpix = lane >> 3, rchunk = lane & 7;                       // read-back role: pixel rpix + 8 k, 16-byte chunk rchunk
      const int rco = G::CT * ct + 64 * mh + 8 * rchunk;
      const int rcobad = rco >= p.coutp ? (int)0x80000000 : 0;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int oy = Y0 + 4 * rq + nt;
        const int rowbad = oy >= p.OH ? (int)0x80000000 : 0;
        // plane 0: the fp16 values; plane 1 (hi + lo outputs): the remainders, recomputed rather than kept (registers)
#pragma unroll
        for (int plane = 0; plane < 2; ++plane) {
          if (plane == 1 && !fe.o_lo) break;
#pragma unroll
          for (int mp = 0; mp < 4; ++mp) {
            const int co = G::CT * ct + 64 * mh + 16 * mp + 8 * hi;
            const int coc = co < p.coutp ? co : p.coutp - 8;
            const f4 b0 = *reinterpret_cast<const f4*>(bl + 16 * mp + 8 * hi), b1 = *reinterpret_cast<const f4*>(bl + 16 * mp + 8 * hi + 4);
            float v[8], t[8], brow[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[mp >> 1][nt][8 * (mp & 1) + e];
            if (border_cb) {
              const float bias[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
              conv_class_bias_row(p, bias, coc, n, oy < p.OH ? oy : p.OH - 1, oxc, brow);
            }
            conv_epilogue_fast_values<false, false>(fe, v, brow, co, z, z, s0, s1, z, t);
            h8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              hv[e] = (half_t)t[e];
              if (plane == 1) hv[e] = (half_t)(t[e] - (float)hv[e]);
            }
            *reinterpret_cast<h8*>(tt + pix * XN_TPITCH + (16 * mp + 8 * hi) * 2) = hv;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int px = rpix + 8 * k;
            const h8 o = *reinterpret_cast<const h8*>(tt + px * XN_TPITCH + rchunk * 16);
            const int voff = (2 * nt * (int)p.o_sy + 2 * (int)((X0 + px) * p.o_sx) + 2 * rco) | rowbad | rcobad | (X0 + px >= p.OW ? (int)0x80000000 : 0);
#if XN_ABL & 16
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(conv_u4, o), rs, voff | (int)0x80000000, 0, 0);
#else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(conv_u4, o), rs, voff + (plane ? 2 * (int)fe.o_lo : 0), 0, 0);
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (WIDE && FAST && !fe.o_lo) {
      // the wide form with per-pixel operands (the masked / accumulating dgrads), plain fp16 output: the operands are loaded row piece by
      // row piece as before (a piece's 16 bytes per lane share their 128-byte line with the neighbouring pieces: L1 hits), but the finished
      // row leaves through the wave's LDS tile as whole lines like the LEAN instance's -- stores are what the L2 sees one request per
      // 16 bytes of
      const __amdgpu_buffer_rsrc_t rs = conv_make_rs(p.out16 + n * p.o_sn + (long)(Y0 + 4 * rq) * p.o_sy);
      char* tt = smem + 2 * XN_BUF + 1024 + wid * (32 * XN_TPITCH);
      const int rpix = lane >> 3, rchunk = lane & 7;
      const int rco = G::CT * ct + 64 * mh + 8 * rchunk;
      const int rcobad = rco >= p.coutp ? (int)0x80000000 : 0;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int oy = Y0 + 4 * rq + nt;
        const int rowbad = oy >= p.OH ? (int)0x80000000 : 0;
#pragma unroll
        for (int mp = 0; mp < 4; ++mp) {
          const int mt = mp >> 1, pair = mp & 1;
          const int co = G::CT * ct + 64 * mh + 32 * mt + 16 * pair + 8 * hi;
          if (oy >= p.OH || ox >= p.OW || co >= p.coutp) continue;
          float v[8], bias[8], s0[8], s1[8], brow[8], t[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = acc[mt][nt][8 * pair + e];
            bias[e] = (p.bias && co + e < p.cout) ? p.bias[n * p.bias_sn + co + e] : 0.f;
          }
          h8 rr = {0, 0, 0, 0, 0, 0, 0, 0}, oo = {0, 0, 0, 0, 0, 0, 0, 0}, mm = {1, 1, 1, 1, 1, 1, 1, 1};
          if (fe.has_res) rr = *reinterpret_cast<const h8*>(p.res + n * p.r_sn + oy * p.r_sy + ox * p.r_sx + co);
          if (fe.has_old) oo = *reinterpret_cast<const h8*>(p.out16 + n * p.o_sn + oy * p.o_sy + ox * p.o_sx + co);
          if (fe.has_mask) mm = *reinterpret_cast<const h8*>(p.mask + n * p.m_sn + oy * p.m_sy + ox * p.m_sx + co);
          if (fe.has_cb) conv_class_bias_row(p, bias, co, n, oy, ox, brow);
          else {
#pragma unroll
            for (int e = 0; e < 8; ++e) brow[e] = bias[e];
          }
          conv_epilogue_fast_values<true, false>(fe, v, brow, co, rr, oo, s0, s1, mm, t);
          h8 hv;
#pragma unroll
          for (int e = 0; e < 8; ++e) hv[e] = (half_t)t[e];
          *reinterpret_cast<h8*>(tt + pix * XN_TPITCH + (16 * mp + 8 * hi) * 2) = hv;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int px = rpix + 8 * k;
          const h8 o = *reinterpret_cast<const h8*>(tt + px * XN_TPITCH + rchunk * 16);
          const int voff = (2 * nt * (int)p.o_sy + 2 * (int)((X0 + px) * p.o_sx) + 2 * rco) | rowbad | rcobad | (X0 + px >= p.OW ? (int)0x80000000 : 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(conv_u4, o), rs, voff, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int oy = Y0 + 4 * rq + nt;
#pragma unroll
        for (int mp = 0; mp < 4; ++mp) {
          const int mt = mp >> 1, pair = mp & 1;
          const int co = G::CT * ct + 64 * mh + 32 * mt + 16 * pair + 8 * hi;
          if (oy >= p.OH || ox >= p.OW || co >= p.coutp) continue;
          float v[8], bias[8], s0[8], s1[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = acc[mt][nt][8 * pair + e];
            bias[e] = (p.bias && co + e < p.cout) ? p.bias[n * p.bias_sn + co + e] : 0.f;
          }
          if constexpr (FAST) {      // (the wide form with per-pixel operands: row by row)
            half_t* o = p.out16 + n * p.o_sn + oy * p.o_sy + ox * p.o_sx + co;
            h8 rr = {0, 0, 0, 0, 0, 0, 0, 0}, oo = {0, 0, 0, 0, 0, 0, 0, 0}, mm = {1, 1, 1, 1, 1, 1, 1, 1};
            if (fe.has_res) rr = *reinterpret_cast<const h8*>(p.res + n * p.r_sn + oy * p.r_sy + ox * p.r_sx + co);
            if (fe.has_old) oo = *reinterpret_cast<const h8*>(o);
            if (fe.has_mask) mm = *reinterpret_cast<const h8*>(p.mask + n * p.m_sn + oy * p.m_sy + ox * p.m_sx + co);
            float brow[8];
            if (fe.has_cb) conv_class_bias_row(p, bias, co, n, oy, ox, brow);
            else {
#pragma unroll
              for (int e = 0; e < 8; ++e) brow[e] = bias[e];
            }
            if (fe.has_res || fe.has_old || fe.has_mask) conv_epilogue_fast_row<true, false>(fe, v, brow, co, o, rr, oo, s0, s1, mm);
            else conv_epilogue_fast_row<false, false>(fe, v, brow, co, o, rr, oo, s0, s1, mm);
          } else {
            conv_epilogue_row(p, v, bias, slope, co, n, oy, ox, s0, s1);
          }
        }
      }
    }
    ct = ctn; n = nn; Y0 = Y0n; X0 = X0n;
  }
  if constexpr (BNS) {
    // lanes of a half-wave hold different pixels of the same channels: fold over the 32 pixel lanes, then the waves in wave order
    float* sS = reinterpret_cast<float*>(smem);          // [wave 4][128]: sums of channels 0..63, then sums of squares
    __syncthreads();                                     // (everybody is done with the halo buffers)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float u = bsum[a][e], w_ = bsq[a][e];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { u += __shfl_xor(u, o, 64); w_ += __shfl_xor(w_, o, 64); }
        if (pix == 0) {
          const int ch = 32 * (a >> 1) + 16 * (a & 1) + 8 * hi + e;
          sS[wid * 128 + ch] = u; sS[wid * 128 + 64 + ch] = w_;
        }
      }
    __syncthreads();
    if (tid < 128) {
      const float t = ((sS[tid] + sS[128 + tid]) + sS[256 + tid]) + sS[384 + tid];
      const int ch = tid & 63;
      if (ch < p.coutp) p.stat_part[(size_t)blockIdx.x * p.stat_ld + (tid < 64 ? ch : p.coutp + ch)] = t;
    }
  }
#endif
}

// ---- weights in K-step order: dst[ct][chunk][tap]([mh] wide)[mt][kk][lane][e] = scale x W(row CT ct + 64 mh + 32 mt + perm(lane % 32), channel,
// tap) with input channel j = 32 chunk + 16 kk + 8 (lane / 32) + e and perm as in csbsr_pack_weights_x3 (a lane's accumulator registers
// 8 pair .. 8 pair + 7 are consecutive channels); the WIDE layout (CT = 128) whenever the rows pad to more than 64.  ``plane`` > 0 (a split
// input run as 2 x plane plain channels): the weight of input channel j is that of channel j mod plane -- [w | w].  kind 0: forward, W is
// OIHW; kind 1: dgrad of the stride-1 conv (rows = the conv's input channels, contracted channels its outputs, taps flipped).
struct PackXNK { const float* w; half_t* dst; int kind, D1, nch, nct, c_real, rows_real, row_off, k_off, plane, wide; float scale; };
__global__ void pack_weights_x3n_kernel(const PackXNK p, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), kk = (int)((i >> 9) & 1), mt = (int)((i >> 10) & 1);
    const int mh = p.wide ? (int)((i >> 11) & 1) : 0;
    const long step = i >> (p.wide ? 12 : 11);
    const int tap = (int)(step % XN_NT);
    const long t2 = step / XN_NT;
    const int chunk = (int)(t2 % p.nch), ct = (int)(t2 / p.nch);
    const int m = lane & 31, q_ = m >> 3, h_ = (m >> 2) & 1;
    const int row = (p.wide ? 128 : 64) * ct + 64 * mh + 32 * mt + 16 * (q_ >> 1) + 8 * h_ + 4 * (q_ & 1) + (m & 3);
    int c = 32 * chunk + 16 * kk + 8 * (lane >> 5) + e;
    if (p.plane > 0) c = c >= p.plane ? c - p.plane : c;
    const int ky = tap / 3, kx = tap % 3;
    float v = 0.f;
    if (row < p.rows_real && c < p.c_real) {
      const int rr = p.row_off + row, cc = p.k_off + c;
      if (p.kind == 0) v = p.w[(((long)rr * p.D1 + cc) * 3 + ky) * 3 + kx];
      else v = p.w[(((long)cc * p.D1 + rr) * 3 + (2 - ky)) * 3 + (2 - kx)];
    }
    p.dst[i] = (half_t)(v * p.scale);
  }
}

static inline bool xn_wide_rows(int rows_real) { return round_up(rows_real, 8) > 64; }

// in_ch = padded input channels the kernel walks (a multiple of 32; for a split input: 2 x the plane's padded channels)
extern "C" int64_t csbsr_packed_weight_elems_x3n(int32_t in_ch, int32_t rows_real) {
  const bool wide = xn_wide_rows(rows_real);
  const int nch = (in_ch + 31) / 32, ct = wide ? 128 : 64, nct = (round_up(rows_real, 8) + ct - 1) / ct;
  return (int64_t)nct * nch * XN_NT * ((wide ? XNGeo<true>::WSTEP : XNGeo<false>::WSTEP) / 2);
}

extern "C" int csbsr_pack_weights_x3n(const float* w, void* dst, int32_t kind, int32_t D0, int32_t D1, int32_t c_real, int32_t rows_real,
                                      int32_t row_off, int32_t k_off, int32_t in_ch, int32_t plane, float scale, csbsr_stream_t s) {
  CSBSR_CHECK(w && dst && (kind == 0 || kind == 1), "pack_x3n: bad args");
  const int kdim = kind == 0 ? D1 : D0, rdim = kind == 0 ? D0 : D1;
  CSBSR_CHECK(c_real >= 1 && rows_real >= 1 && k_off >= 0 && k_off + c_real <= kdim && row_off >= 0 && row_off + rows_real <= rdim,
              "pack_x3n: range out of bounds");
  CSBSR_CHECK(in_ch % 32 == 0 && (plane == 0 ? in_ch >= c_real : (in_ch == 2 * plane && plane >= c_real)), "pack_x3n: bad channel geometry");
  PackXNK p;
  p.w = w; p.dst = reinterpret_cast<half_t*>(dst); p.kind = kind; p.D1 = D1;
  p.wide = xn_wide_rows(rows_real) ? 1 : 0;
  p.nch = in_ch / 32; p.nct = (round_up(rows_real, 8) + (p.wide ? 127 : 63)) / (p.wide ? 128 : 64);
  p.c_real = c_real; p.rows_real = rows_real; p.row_off = row_off; p.k_off = k_off; p.plane = plane; p.scale = scale;
  const long total = csbsr_packed_weight_elems_x3n(in_ch, rows_real);
  const long nb = (total + 255) / 256;
  hipLaunchKernelGGL(pack_weights_x3n_kernel, dim3((int)(nb > 8192 ? 8192 : nb)), dim3(256), 0, reinterpret_cast<hipStream_t>(s), p, total);
  CSBSR_LAUNCH_CHECK("csbsr_pack_weights_x3n");
  return 0;
}

static int g_conv_x3n_mode = 1;      // 0 off, 1 launches that fill the chip (default), 2 every eligible launch (tests), 3 narrow form only; +4: the wide form only up to 384 input channels
extern "C" void csbsr_debug_set_conv_x3n(int mode) { g_conv_x3n_mode = mode & 7; }

// Which launches take this kernel (return value 1: the narrow form, 2: the wide form): 3x3, stride 1, pad 1, dilation 1, ONE input segment in
// whole 32-channel chunks -- plain fp16, or a split [hi | lo] pair presented as one 2 x Cp-channel segment with split_fused = 2 (the
// two-product plan) --, fp16 output (hi + lo pairs allowed), any fused epilogue of the general kernels (split residual operands included:
// BlurSkip's conv_shift.1 combines x * scale + shift on hi + lo pairs) except sample statistics, the fp32 side output and the fused
// epilogue-backward sums.  Narrow: 33 .. 64 padded output channels from >= 64 (mode 2: >= 32) input channels, BatchNorm sums allowed.  Wide:
// more than 64 padded output channels from 32 .. 384 input channels (measured against csrc/conv_x3.hip's whole-K-resident tile, one launch
// of the batch of 8 at 448^2: 128 -> 569 2.94 -> 2.42 ms, 256 -> 697 5.39 -> 4.94, 384 -> 825 8.47 -> 8.25, 825 -> 825 16.7 -> 17.2: the
// overlapped epilogue is worth less and the twice-staged weights more as K grows).  With the epilogue of the later commits (loads ahead of
// stores, whole-line stores) the wide form wins at every width -- 825 -> 825 16.74 -> 16.36 ms forward, 16.46 -> 15.89 dgrad, 825 -> 384
// 7.38 -> 7.20, 384 <- 825 dgrad 7.20 -> 6.93 -- and takes them all.
extern "C" int32_t csbsr_conv_x3n_eligible(const csbsr_conv_desc_t* d) {
  const int mode = g_conv_x3n_mode & 3;
  if (!d || !mode || d->transposed || d->KH != 3 || d->KW != 3 || d->dil != 1) return 0;
  if (d->stride != 1 || d->pad != 1 || d->OH != d->H || d->OW != d->W) return 0;
  if (d->in[0].c % 32 != 0 || d->in[0].c < ((mode == 2 || d->coutp > 64) ? 32 : 64)) return 0;
  if (d->in[1].c != 0 || d->in[0].sx == 0 || (d->split_fused != 0 && d->split_fused != 2)) return 0;
  if (d->coutp <= 32 || !d->out16 || d->out32) return 0;
  if (d->dact_bias || d->dact_prelu || d->dres) return 0;
  if (d->in[0].sy >= (1l << 31) / 2 / (XN_HH + 1)) return 0;
  const bool wide = d->coutp > 64;
  if (wide) {
    if (mode == 3 || d->stat_mode != CSBSR_STAT_NONE) return 0;
    // above 384 input channels only the launches of the straight-line rows (the SFT conv1 epilogues -- sigmoid, f x scale + shift -- have
    // their own instance of conv_x3; +4 in the debug mode: the old limit, everything above 384 channels stays on conv_x3)
    const bool fast_like = d->act != CSBSR_ACT_SIGMOID && !d->r_lo &&
                           (d->res_mode == CSBSR_RES_NONE || d->res_mode == CSBSR_RES_ADD || d->res_mode == CSBSR_RES_SUB);
    if (d->in[0].c > 384 && (!fast_like || (g_conv_x3n_mode & 4))) return 0;
  } else {
    if (d->stat_mode != CSBSR_STAT_NONE && d->stat_mode != CSBSR_STAT_BN) return 0;    // (BatchNorm sums: the straight-line rows only, see the launcher)
    if (d->stat_mode == CSBSR_STAT_BN && (d->r_lo || d->r2_lo || d->res_mode != CSBSR_RES_NONE || d->accumulate || d->mask)) return 0;      // (BatchNorm sums: no per-pixel operand)
  }
  if (mode != 2 && (long)d->N * d->OH * d->OW < 512L * XN_TH * 64) return 0;
  return wide ? 2 : 1;
}

template <bool FAST, bool BNS = false, bool WIDE = false, bool LEAN = false>
static int launch_x3n(const ConvK& k, const XNExtra& q, unsigned g, hipStream_t st) {
  constexpr int SM_BYTES = 2 * XNGeo<WIDE>::BUF + (WIDE ? 1024 + 4 * 32 * XN_TPITCH : 0);      // (wide: + the four waves' bias slots and output transposition tiles)
  static LdsAttrOnce attr;
  if (int e = csbsr_lds_attr(attr, reinterpret_cast<const void*>(conv_x3n_kernel<FAST, BNS, WIDE, LEAN>), SM_BYTES, "conv_x3n")) return e;
  hipLaunchKernelGGL((conv_x3n_kernel<FAST, BNS, WIDE, LEAN>), dim3(g), dim3(256), SM_BYTES, st, k, q);
  CSBSR_LAUNCH_CHECK("csbsr_conv_x3n_forward");
  return 0;
}

extern "C" int csbsr_conv_x3n_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s) {
  const int form = csbsr_conv_x3n_eligible(d);
  CSBSR_CHECK(form, "conv_x3n: launch not eligible (see csbsr_conv_x3n_eligible)");
  const bool wide = form == 2;
  csbsr_conv_desc_t dd = *d;
  dd.split_fused = 0;                  // the planes are plain channels to this kernel (and to the argument-block validation)
  ConvK k;
  if (int rc = conv_desc_to_k(&dd, k)) return rc;
  const int tw = wide ? XNGeo<true>::TW : XNGeo<false>::TW, ctile = wide ? XNGeo<true>::CT : XNGeo<false>::CT;
  XNExtra q;
  q.tiles_x = (unsigned)((d->OW + tw - 1) / tw); q.tiles_y = (unsigned)((d->OH + XN_TH - 1) / XN_TH);
  q.nct = (unsigned)((d->coutp + ctile - 1) / ctile); q.nch = (unsigned)(d->in[0].c / 32);
  const int slots = csbsr_cu_budget(reinterpret_cast<hipStream_t>(s)) * ((wide && !(XN_ABL & 4)) ? 2 : 1);      // (wide: two workgroups per CU)
  const unsigned items = q.tiles_x * q.tiles_y * (unsigned)d->N * q.nct;
  const unsigned g = items < (unsigned)slots ? items : (unsigned)slots;
  const bool fast_rows = conv_epilogue_fast_ok(k);
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  if (wide) {
    const bool lean = fast_rows && k.res_mode == CSBSR_RES_NONE && !k.accumulate && !k.mask;
    g_last_conv_kernel = CONVK_X3N | (fast_rows ? 1 : 0) << 8 | 4 << 8 | (lean ? 8 : 0) << 8;
    if (lean) return launch_x3n<true, false, true, true>(k, q, g, st);
    return fast_rows ? launch_x3n<true, false, true>(k, q, g, st) : launch_x3n<false, false, true>(k, q, g, st);
  }
  if (k.stat_mode == CSBSR_STAT_BN) {
    CSBSR_CHECK(fast_rows && q.nct == 1 && k.stat && k.res_mode == CSBSR_RES_NONE && !k.accumulate && !k.mask,
                "conv_x3n: fused BatchNorm sums need the straight-line epilogue rows without per-pixel operands and one cout tile");
    k.stat_ld = 2 * (long)k.coutp;
    k.stat_part = csbsr_red_scratch((long)g * k.stat_ld);
    CSBSR_NEED_SCRATCH(k.stat_part, "conv_x3n (fused statistics)");
    g_last_conv_kernel = CONVK_X3N | 3 << 8;
    if (int rc = launch_x3n<true, true>(k, q, g, st)) return rc;
    return csbsr_sum_partials(k.stat_part, (int)g, k.stat_ld, 2 * k.coutp, k.stat, st);
  }
  g_last_conv_kernel = CONVK_X3N | (fast_rows ? 1 : 0) << 8;
  return fast_rows ? launch_x3n<true>(k, q, g, st) : launch_x3n<false>(k, q, g, st);
}
