// 3x3 stride-1 convolution of the wide LOW-resolution layers (the SFT scale / shift convolutions, kbpn.py:493-518, and their dgrads) with
// the Winograd transform F(2, 3) ALONG X: two neighbouring output pixels of a row come from four products per (input channel, row tap)
// instead of six, so the layer costs 2/3 of the MFMA work of csrc/conv_x3.hip -- the one lever left on layers whose direct kernels
// already run at the vendor library's GEMM rate.  For one row tap ky, with d0..d3 the four input pixels
// x = 2t - 1 .. 2t + 2 of row y + ky - 1 and w0..w2 the tap's three weights:
//
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3            (input transform, exact fp16 differences rounded once;
//                                                                           the kernel holds -V2 and -U2: same product)
//     U0 = w0        U1 = (w0 + w1 + w2) / 2       U2 = (w0 - w1 + w2) / 2       U3 = w2      (weight transform, packed once per step)
//     M_p = sum over (ky, channel) of U_p V_p  (four fp32 accumulator sets: the MFMAs)
//     out[2t] = M0 + M1 + M2        out[2t + 1] = M1 - M2 - M3                (output transform, lane-local on the accumulators)
//
// The row taps accumulate in the transformed domain, so only x is transformed: 4 / 3 of the weight bytes, twice the accumulators per
// output (which is what the 256 AGPRs of a one-wave-per-SIMD kernel can hold), no 2-D transform's 16-position accumulator blow-up.
// Precision: tests/study_winograd.py (profiles/r06_winograd_study.txt) -- the SR image moves by < 3e-5 of its maximum.
//
// MEASURED (round 6, N = 8, 448^2, scripts/bench_conv.py; profiles/r06_winograd_kernel.txt) AND NOT THE DEFAULT: 825 -> 384 6.85 ms against
// 7.49 ms for conv_x3<3>, 384 -> 825 8.47 / 8.73, 256 -> 697 5.47 / 5.70, 569 -> 128 1.87 / 2.05, dgrads alike: 3-9 % faster, not the 1.5x of
// the MFMA count -- 9 ms of a 1007 ms config-2 step on the SFT convs (7.94 -> 8.02 img/s), for a composed segmentation map that moves from
// 1.11e-3 to 1.25e-3 of the reference's: opt-in (csbsr_debug_set_conv_x3w(1), CSBSR_CONV_X3W=1), with its tests.  The ablation builds (CSBSR_X3W_ABLATE) say where the time goes:
// MFMAs + LDS fragment reads alone 4.41 ms (2070 TF/s of direct-equivalent work), + the weight stream 4.80, + the halo stream 6.12, both
// 6.85.  F(2, 3) needs a fresh weight fragment per two MFMAs per position where the direct kernel reuses one over eight pixel blocks, and
// a transformed pixel fragment per MFMA pair: twice the operand traffic per MFMA.  The first build split the waves by ROWS, so every weight
// fragment crossed the CU's vector-memory path twice (64 B/clk for the weights alone, 7.55 ms: parity); this one splits them by POSITION
// pair (each fragment loaded by one wave, the two halves of the output transform meet through LDS in the epilogue).  What is left is the
// halo stream: 64-byte requests at a two-pixel stride, 12 per lane and chunk -- neither a deeper lead (loads two chunks ahead, registers
// carried over the back edge), nor a weight ring of 6 instead of 3, nor four-tile runs per lane (10 loads for 4 tiles instead of 16: 7.32 ms,
// the three staging waves fall behind the fourth) moved it.  The 256 accumulators of the positions are the whole AGPR file, so the tile
// cannot grow to raise the reuse, and LDS (at ~80 of 128 B/clk with the V fragments) has no room to share more.
//
//  * one persistent workgroup per CU (4 waves) computes a 4-row x 64-pixel x 128-cout tile; a wave owns 64 couts x 4 rows x 32 x-tiles x TWO
//    of the four positions = acc[mt 2][row 4][position 2] 32x32 MFMA tiles, 256 accumulator registers; the two position-pair waves of a
//    cout half meet in the epilogue (one accumulator set each way through LDS);
//  * the pixel operand is staged per 32-channel chunk: every lane loads the four pixels (16 bytes = 8 channels each) of three
//    (halo row, x-tile, channel octet) items from L2 / HBM into registers -- buffer loads, zeros outside the image from the bounds
//    check -- transforms them with 16 packed fp16 adds and writes the four V vectors to one of two 60 KB LDS buffers, all of it issued
//    INSIDE the previous chunk's K loop (the loads in its first K step, the transforms + ds_writes in its last three);
//  * V layout [halo row 6][position 4][x-tile 32][4 channel octets + 1 pad]: the odd 80-byte tile pitch keeps the ds_read_b128 fragment
//    reads of the MFMA B operand conflict-free;
//  * weights as in conv_x3: packed in MFMA-fragment order per K step (row tap, 16-channel slice) x [position][mt], 8 KB per wave and
//    K step straight from L2 into registers two steps ahead; cout tiles of one pixel tile run side by side on one XCD;
//  * one barrier per chunk (six K steps, 96 MFMAs per wave); the general fused epilogue rows of conv_common.h.
#include "common.h"
#include "conv_common.h"
#include "csbsr_debug.h"

#define XW_TH 4
#define XW_TW 64
#define XW_NT 32                          // x-tiles (two outputs each) per tile row
#define XW_HH (XW_TH + 2)
#define XW_TPITCH 80                      // bytes per (halo row, position, x-tile): 32 channels = 4 sixteen-byte slots + 1 pad slot
#define XW_BUF (XW_HH * 4 * XW_NT * XW_TPITCH)      // 61440
#define XW_WSTEP 16384                    // bytes of one K step's weights for the 128-cout tile: [cout half mh][position][mt][lane][8]
#define XW_STEPS 6                        // K steps per chunk: 3 row taps x 2 sixteen-channel slices
#ifndef XW_RING
#define XW_RING 3                         // (6 measured the same: 6.85 vs 6.81 ms)
#endif
#define XW_DIST (XW_RING - 1)

typedef unsigned xw_u4 __attribute__((ext_vector_type(4)));

struct XWExtra {
  unsigned tiles_x, tiles_y, nct, nch;   // pixel tiles, 128-cout tiles, 32-channel chunks
};

#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t xw_make_rs(const half_t* base) {
  const unsigned long a = reinterpret_cast<unsigned long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi_ << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
static __device__ __forceinline__ h8 xw_as_h8(const xw_u4& v) {
  union { xw_u4 u; h8 h; } c;
  c.u = v;
  return c.h;
}
#endif

// EPI: 0 general fused row, 1 straight-line rows (conv_epilogue_fast_ok), 2 the SFT conv1 rows (sigmoid / FMA residual)
// ABL (timing experiments through csbsr_debug_set_conv_x3w bits 8.., results are garbage): 1 no halo loads for the next chunk, 2 every K step
// loads the weights of step 0, 4 no LDS fragment reads in the K loop, 8 no weight loads in the K loop, 16 no transform + V writes
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(256) void conv_x3w_kernel(const ConvK p, const XWExtra q) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tcol = lane & 31, hi = lane >> 5;
  const int mh = wid & 1, pp = wid >> 1;      // cout half (64 couts), position pair (Winograd positions 2 pp, 2 pp + 1)
  const unsigned per_img = q.tiles_x * q.tiles_y, ntiles = per_img * (unsigned)p.N, items = ntiles * q.nct;
  unsigned it = blockIdx.x;
  if (it >= items) return;
  const float slope = (p.act == CSBSR_ACT_PRELU) ? *p.prelu : p.act_slope;
  const EpiFast fe = conv_epilogue_fast_setup(p, slope);
  const half_t* in0 = reinterpret_cast<const half_t*>(p.in[0].ptr);
  const int isy = (int)p.in[0].sy, isx = (int)p.in[0].sx;

  // staging roles (the same for every tile and chunk): item i of this lane = (halo row 2 i + tid / 128, x-tile (tid / 4) % 32, octet tid % 4)
  const int s_slot = tid & 3, s_t = (tid >> 2) & 31, s_r0 = tid >> 7;
  int voff[3];
  unsigned soff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int hr = 2 * i + s_r0;
    voff[i] = 2 * (hr * isy + 2 * s_t * isx + s_slot * 8);
    soff[i] = (unsigned)(((hr * 4) * XW_NT + s_t) * XW_TPITCH + s_slot * 16);
  }
  const int step_x = 2 * isx;             // bytes between neighbouring pixels

  auto decode = [&](unsigned item, int& ct, int& n, int& Y0, int& X0) {
    // the cout tiles of ONE pixel tile are consecutive items and an XCD owns a contiguous run of the item order (csrc/conv_x3.hip)
    item = xcd_remap(item, items);
    const unsigned tile = item / q.nct;
    ct = item - tile * q.nct;
    n = tile / per_img;
    const unsigned r_ = tile - n * per_img;
    Y0 = (r_ / q.tiles_x) * XW_TH; X0 = (r_ % q.tiles_x) * XW_TW;
  };
  // (uniform) address of halo pixel (0, 0), channel 32 chunk, of tile (n, Y0, X0): input pixel (Y0 - 1, X0 - 1)
  auto chunk_src = [&](int n, int Y0, int X0, int chunk) -> const half_t* {
    if (ABL & 32) { n = 0; Y0 = 4 * (int)(blockIdx.x & 31) + 4; X0 = 64; chunk = 0; }      // every halo from one small (L2-resident) region
    return in0 + n * p.in[0].sn + (long)(Y0 - 1) * p.in[0].sy + (long)(X0 - 1) * p.in[0].sx + chunk * 32;
  };
  // the four pixels of item i: rows / columns outside the image read as zeros (out-of-range offset -> hardware bounds check).  The
  // validity mask is ARITHMETIC (sign bits), not a select: the compiler turned `ok ? off : -1` in front of a buffer load into divergent
  // branches with one load per arm and an s_waitcnt vmcnt(0) between them -- inside the K loop
  auto load_item = [&](__amdgpu_buffer_rsrc_t rs, int Y0, int X0, int i, xw_u4 (&d)[4]) __attribute__((always_inline)) {
    const int y = Y0 - 1 + 2 * i + s_r0, x = X0 - 1 + 2 * s_t;
    const int my = ~(y >> 31) & ((y - p.H) >> 31);            // all ones iff 0 <= y < H
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = x + j;
      const int m = my & ~(xx >> 31) & ((xx - p.W) >> 31);
      d[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((voff[i] + j * step_x) & m) | ~m, 0, 0);
    }
  };
  // V0 = d0 - d2, V1 = d1 + d2, V2' = d1 - d2 (= -V2: the pack stores -U2, the product is the same), V3 = d1 - d3: 16 v_pk_add_f16 (the
  // vector subtraction compiled to per-half v_sub_f16 + v_pack_b32_f16)
  auto pk_sub = [&](const xw_u4& a, const xw_u4& b) __attribute__((always_inline)) -> xw_u4 {
    xw_u4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned t;
      asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a[k]), "v"(b[k]));
      r[k] = t;
    }
    return r;
  };
  auto pk_add = [&](const xw_u4& a, const xw_u4& b) __attribute__((always_inline)) -> xw_u4 {
    xw_u4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned t;
      asm("v_pk_add_f16 %0, %1, %2" : "=v"(t) : "v"(a[k]), "v"(b[k]));
      r[k] = t;
    }
    return r;
  };
  auto store_item = [&](const xw_u4 (&d)[4], int i, int buf) __attribute__((always_inline)) {
    char* o = smem + buf * XW_BUF + soff[i];
    *reinterpret_cast<xw_u4*>(o) = pk_sub(d[0], d[2]);
    *reinterpret_cast<xw_u4*>(o + XW_NT * XW_TPITCH) = pk_add(d[1], d[2]);
    *reinterpret_cast<xw_u4*>(o + 2 * XW_NT * XW_TPITCH) = pk_sub(d[1], d[2]);
    *reinterpret_cast<xw_u4*>(o + 3 * XW_NT * XW_TPITCH) = pk_sub(d[1], d[3]);
  };
  // a wave's weights of K step (ct, chunk, ky, kk): its two positions x [mt 2] fragments, 4 KB, one 16-byte load per lane each -- every
  // fragment of the step's 16 KB is loaded by exactly ONE wave of the workgroup (the first build split the waves by rows: each fragment
  // crossed the CU's vector-memory path twice, 64 B/clk for the weights alone)
  const unsigned wlane = (unsigned)(mh * 8192 + pp * 4096 + lane * 16);
  auto load_w = [&](int ct, int step, h8 (&w)[2][2]) __attribute__((always_inline)) {
    const char* b = reinterpret_cast<const char*>(p.wt) + ((size_t)ct * q.nch * XW_STEPS + step) * XW_WSTEP;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) w[pl][mt] = *reinterpret_cast<const h8*>(b + (wlane + (pl * 2 + mt) * 1024));
  };
  // this lane's B-operand fragments: V[halo row r + ky][position 2 pp + pl][x-tile tcol], channels 16 kk + 8 hi ..
  const char* vl = smem + ((2 * pp) * XW_NT + tcol) * XW_TPITCH + hi * 16;

  int ct, n, Y0, X0;
  decode(it, ct, n, Y0, X0);
  {   // chunk 0 of the first tile, synchronously
    const __amdgpu_buffer_rsrc_t rs = xw_make_rs(chunk_src(n, Y0, X0, 0));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      xw_u4 d[4];
      load_item(rs, Y0, X0, i, d);
      store_item(d, i, 0);
    }
  }
  h8 wreg[XW_RING][2][2];                  // K step g = chunk * 6 + s lives in wreg[s % XW_RING] (XW_RING divides 6)
#pragma unroll
  for (int g = 0; g < XW_DIST; ++g) load_w(ct, g, wreg[g]);
  const int nsteps = (int)q.nch * XW_STEPS;

  for (; it < items; it += gridDim.x) {
    const unsigned itn = it + gridDim.x;
    int ctn = ct, nn = n, Y0n = Y0, X0n = X0;
    if (itn < items) decode(itn, ctn, nn, Y0n, X0n);
    const unsigned par = ((it - blockIdx.x) / gridDim.x) * q.nch;     // chunk c of this tile lives in buffer (par + c) & 1
    f16v acc[2][4][2];                       // [mt][row][local position]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c_ = 0; c_ < 2; ++c_)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][c_][r] = 0.f;

    for (int c = 0; c < (int)q.nch; ++c) {
      // this wave's V pieces of chunk c are in LDS; after the barrier everybody's are, and every wave is done reading the other buffer
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int bnext = (int)((par + c + 1) & 1);
      // the next chunk (past the tile's last one: the next tile's first; past the last tile: a harmless refetch that keeps the instruction
      // stream uniform): all twelve loads in this chunk's FIRST K step, the transforms + writes in its last three -- three to five K steps
      // (1500-2500 cycles) between a load and its use.  (Carrying the loaded registers over the loop's back edge for a six-step lead makes the
      // compiler's s_waitcnt insertion give up on their age: it drained the whole weight ring at the top of every chunk.)
      const bool same = c + 1 < (int)q.nch;
      const int sn_ = same ? n : (itn < items ? nn : n), sY = same ? Y0 : (itn < items ? Y0n : Y0), sX = same ? X0 : (itn < items ? X0n : X0);
      const __amdgpu_buffer_rsrc_t nrs = xw_make_rs(chunk_src(sn_, sY, sX, same ? c + 1 : 0));
      asm volatile("" ::: "memory");
      xw_u4 stg[3][4];
      const char* vb = vl + ((par + c) & 1) * XW_BUF;
      h8 bfr[2][4];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int r = 0; r < 4; ++r) bfr[pl][r] = *reinterpret_cast<const h8*>(vb + ((r * 4 + pl) * XW_NT) * XW_TPITCH);      // row tap 0, slice 0
#pragma unroll
      for (int s = 0; s < XW_STEPS; ++s) {
        {   // weights XW_DIST K steps ahead (past the tile's last step: the next tile's first ones)
          const int g = c * XW_STEPS + s + XW_DIST;
          if (ABL & 8) {}
          else if (ABL & 2) load_w(ct, 0, wreg[(s + XW_DIST) % XW_RING]);
          else if (g < nsteps) load_w(ct, g, wreg[(s + XW_DIST) % XW_RING]);
          else load_w(ctn, g - nsteps, wreg[(s + XW_DIST) % XW_RING]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int ky1 = (s + 1) >> 1, kk1 = (s + 1) & 1;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[0][r][pl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s % XW_RING][pl][0], bfr[pl][r], acc[0][r][pl], 0, 0, 0);
            acc[1][r][pl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s % XW_RING][pl][1], bfr[pl][r], acc[1][r][pl], 0, 0, 0);
            // the same (position, row) fragment of the next K step (the next chunk starts over after its barrier)
            if (s + 1 < XW_STEPS && !(ABL & 4))
              bfr[pl][r] = *reinterpret_cast<const h8*>(vb + (((r + ky1) * 4 + pl) * XW_NT) * XW_TPITCH + kk1 * 32);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0 && pl == 0 && r < 3 && !(ABL & 1)) load_item(nrs, sY, sX, r, stg[r]);
            if (s >= 3 && pl == 1 && r == 1 && !(ABL & 16)) store_item(stg[s - 3], s - 3, bnext);      // transform + four ds_write_b128 in the MFMAs' shadow
          }
        }
      }
    }

    // ---- epilogue.  out[2t] = M0 + M1 + M2, out[2t + 1] = M1 - M2' ... with the kernel's sign of position 2 (V2' = -V2 against -U2: M2 is the
    // plain product): the position-pair waves exchange ONE accumulator set each through LDS -- wave pp = 0 holds M0, M1: it keeps M0 + M1,
    // sends M1, receives M2 and finishes the EVEN pixels; wave pp = 1 holds M2, M3: it keeps -(M2 + M3), sends M2, receives M1 and finishes
    // the ODD pixels -- one (mt, row) block of 16 registers per round through one of two 16 KB regions of the V buffer the last chunk has
    // just left (the other buffer already holds the next tile's first chunk).  acc[mt][row][pl][8 pair + e] = cout 128 ct + 64 mh + 32 mt +
    // 16 pair + 8 hi + e of x-tile tcol in row Y0 + row.
    {
      char* xbase = smem + ((par + q.nch - 1) & 1) * XW_BUF;
      char* xmine = xbase + wid * 4096 + lane * 16;
      const char* xpart = xbase + (wid ^ 2) * 4096 + lane * 16;
      const int ox = X0 + 2 * tcol + pp;
      __builtin_amdgcn_s_barrier();              // every wave is out of the last chunk's K loop: its V buffer is free
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int mt = k >> 2, r = k & 3;
        char* wr = xmine + (k & 1) * 16384;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f4 t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = pp ? acc[mt][r][0][4 * j + e] : acc[mt][r][1][4 * j + e];      // pp = 0 sends M1 (pl 1), pp = 1 sends M2 (pl 0); a select: an index computed from pp would put the accumulators in scratch
          *reinterpret_cast<f4*>(wr + j * 1024) = t;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* rd = xpart + (k & 1) * 16384;
        float got[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f4 t = *reinterpret_cast<const f4*>(rd + j * 1024);
#pragma unroll
          for (int e = 0; e < 4; ++e) got[4 * j + e] = t[e];
        }
        const int oy = Y0 + r;
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
          const int co = 128 * ct + 64 * mh + 32 * mt + 16 * pair + 8 * hi;
          if (oy >= p.OH || ox >= p.OW || co >= p.coutp) continue;
          float v[8], bias[8], s0[8], s1[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float a0 = acc[mt][r][0][8 * pair + e], a1 = acc[mt][r][1][8 * pair + e], g_ = got[8 * pair + e];
            v[e] = pp ? g_ - (a0 + a1) : (a0 + a1) + g_;            // even: (M0 + M1) + M2;  odd: M1 - (M2 + M3)
            bias[e] = (p.bias && co + e < p.cout) ? p.bias[n * p.bias_sn + co + e] : 0.f;
          }
          if constexpr (EPI == 1) {
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
          } else if constexpr (EPI == 2) {
            // the SFT conv1's (kbpn.py:505-516): bias + sigmoid (scale branch) or bias + res x res2 (shift branch: f x scale + shift)
            half_t* o = p.out16 + n * p.o_sn + oy * p.o_sy + ox * p.o_sx + co;
            h8 hv;
            if (p.act == CSBSR_ACT_SIGMOID) {
#pragma unroll
              for (int e = 0; e < 8; ++e) hv[e] = (half_t)(co + e < p.cout ? 1.f / (1.f + __expf(-(v[e] * p.out_scale + bias[e]))) : 0.f);
            } else {
              const h8 r1 = *reinterpret_cast<const h8*>(p.res + n * p.r_sn + oy * p.r_sy + ox * p.r_sx + co);
              const h8 r2 = *reinterpret_cast<const h8*>(p.res2 + n * p.r2_sn + oy * p.r2_sy + ox * p.r2_sx + co);
#pragma unroll
              for (int e = 0; e < 8; ++e) hv[e] = (half_t)((co + e < p.cout ? v[e] * p.out_scale + bias[e] : 0.f) + (float)r1[e] * (float)r2[e]);
            }
            *reinterpret_cast<h8*>(o) = hv;
          } else {
            conv_epilogue_row(p, v, bias, slope, co, n, oy, ox, s0, s1);
          }
        }
      }
    }
    // The epilogue's global STORES share vmcnt with the loads and complete out of order with them: left pending into the next tile's K loop
    // they make the compiler's s_waitcnt insertion treat every loop-carried load there as of unknown age (it drained the weight ring to
    // vmcnt(6) at the top of EVERY chunk with a ring of six).  Draining once per tile, through the builtin the pass understands, is cheap.
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
    ct = ctn; n = nn; Y0 = Y0n; X0 = X0n;
  }
#endif
}

// ---- transformed weights in K-step order: dst[ct][chunk][ky][kk][mh][pos][mt][lane][e] = U_pos of (row 128 ct + 64 mh + 32 mt + perm(lane % 32),
// channel 32 chunk + 16 kk + 8 (lane / 32) + e, row tap ky), perm as in csbsr_pack_weights_x3 (a lane's accumulator registers 8 pair .. 8 pair + 7
// are consecutive channels).  kind 0: forward, W is OIHW [row][channel][ky][kx]; kind 1: dgrad of the stride-1 conv (rows = the conv's input
// channels, contracted channels its outputs, taps flipped).  The transform runs in fp32 on the values the direct kernels would multiply
// with (the caller passes the tap-sum-rounded tensor where the layer has one) and is rounded to fp16 once.
struct PackXWK { const float* w; half_t* dst; int kind, D1, nch, nct, c_real, rows_real, row_off, k_off; };
__global__ void pack_weights_x3w_kernel(const PackXWK p, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), mt = (int)((i >> 9) & 1), pos = (int)((i >> 10) & 3), mh = (int)((i >> 12) & 1);
    const long step = i >> 13;
    const int kk = (int)(step & 1);
    const long t1 = step >> 1;
    const int ky = (int)(t1 % 3);
    const long t2 = t1 / 3;
    const int chunk = (int)(t2 % p.nch), ct = (int)(t2 / p.nch);
    const int m = lane & 31, q_ = m >> 3, h_ = (m >> 2) & 1;
    const int row = 128 * ct + 64 * mh + 32 * mt + 16 * (q_ >> 1) + 8 * h_ + 4 * (q_ & 1) + (m & 3);
    const int c = 32 * chunk + 16 * kk + 8 * (lane >> 5) + e;
    float v = 0.f;
    if (row < p.rows_real && c < p.c_real) {
      const int rr = p.row_off + row, cc = p.k_off + c;
      float w0, w1, w2;
      if (p.kind == 0) {
        const float* b = p.w + (((long)rr * p.D1 + cc) * 3 + ky) * 3;
        w0 = b[0]; w1 = b[1]; w2 = b[2];
      } else {
        const float* b = p.w + (((long)cc * p.D1 + rr) * 3 + (2 - ky)) * 3;
        w0 = b[2]; w1 = b[1]; w2 = b[0];
      }
      v = pos == 0 ? w0 : pos == 1 ? 0.5f * (w0 + w1 + w2) : pos == 2 ? -0.5f * (w0 - w1 + w2) : w2;      // (position 2 against V2' = d1 - d2 = -V2)
    }
    p.dst[i] = (half_t)v;
  }
}

extern "C" int64_t csbsr_packed_weight_elems_x3w(int32_t c_real, int32_t rows_real) {
  const int nch = (round_up(c_real, 8) + 31) / 32, nct = (round_up(rows_real, 8) + 127) / 128;
  return (int64_t)nct * nch * XW_STEPS * (XW_WSTEP / 2);
}

extern "C" int csbsr_pack_weights_x3w(const float* w, void* dst, int32_t kind, int32_t D0, int32_t D1, int32_t c_real, int32_t rows_real,
                                      int32_t row_off, int32_t k_off, csbsr_stream_t s) {
  CSBSR_CHECK(w && dst && (kind == 0 || kind == 1), "pack_x3w: bad args");
  const int kdim = kind == 0 ? D1 : D0, rdim = kind == 0 ? D0 : D1;
  CSBSR_CHECK(c_real >= 1 && rows_real >= 1 && k_off >= 0 && k_off + c_real <= kdim && row_off >= 0 && row_off + rows_real <= rdim,
              "pack_x3w: range out of bounds");
  PackXWK p;
  p.w = w; p.dst = reinterpret_cast<half_t*>(dst); p.kind = kind; p.D1 = D1;
  p.nch = (round_up(c_real, 8) + 31) / 32; p.nct = (round_up(rows_real, 8) + 127) / 128;
  p.c_real = c_real; p.rows_real = rows_real; p.row_off = row_off; p.k_off = k_off;
  const long total = csbsr_packed_weight_elems_x3w(c_real, rows_real);
  const long nb = (total + 255) / 256;
  hipLaunchKernelGGL(pack_weights_x3w_kernel, dim3((int)(nb > 8192 ? 8192 : nb)), dim3(256), 0, reinterpret_cast<hipStream_t>(s), p, total);
  CSBSR_LAUNCH_CHECK("csbsr_pack_weights_x3w");
  return 0;
}

static int g_conv_x3w_mode = 0;      // 0 off (DEFAULT: measured at parity with the direct kernels, see the header), 1 launches that fill the chip, 2 every eligible launch (tests)
static int g_conv_x3w_min_c = 128;   // smallest padded input-channel count taken in mode 1
static int g_conv_x3w_abl = 0;
extern "C" void csbsr_debug_set_conv_x3w(int mode) {
  g_conv_x3w_mode = mode & 7;
  if ((mode >> 3) & 31) g_conv_x3w_min_c = ((mode >> 3) & 31) * 32;
  g_conv_x3w_abl = mode >> 8;
}

// Which launches take this kernel: 3x3, stride 1, pad 1, dilation 1, ONE plain-fp16 input segment whose padded channels are a multiple
// of 32, >= 72 padded output channels, fp16 output, any fused epilogue of the general kernels except statistics, the fp32 side output,
// split (hi + lo) operands and the fused epilogue-backward sums -- i.e. what csbsr_conv_x3_eligible takes, from 128 input channels up.
extern "C" int32_t csbsr_conv_x3w_eligible(const csbsr_conv_desc_t* d) {
  if (!d || !g_conv_x3w_mode || d->transposed || d->KH != 3 || d->KW != 3 || d->dil != 1) return 0;
  if (d->stride != 1 || d->pad != 1 || d->OH != d->H || d->OW != d->W) return 0;
  if (d->in[0].c % 32 != 0 || d->in[0].c < (g_conv_x3w_mode == 2 ? 32 : g_conv_x3w_min_c)) return 0;
  if (d->in[1].c != 0 || d->in[0].sx == 0 || d->split_fused) return 0;
  if (d->coutp < 72 || !d->out16 || d->out32 || d->o_lo || d->r_lo || d->r2_lo) return 0;
  if (d->stat_mode != CSBSR_STAT_NONE) return 0;
  if (d->dact_bias || d->dact_prelu || d->dres) return 0;
  if (d->in[0].sy >= (1l << 31) / 2 / (XW_HH + 1)) return 0;
  if (g_conv_x3w_mode == 1 && (long)d->N * d->OH * d->OW * ((d->coutp + 127) / 128) < 512L * XW_TH * XW_TW) return 0;
  return 1;
}

template <int EPI, int ABL = 0>
static int launch_x3w(const ConvK& k, const XWExtra& q, unsigned g, hipStream_t st) {
  constexpr int SM_BYTES = 2 * XW_BUF;
  static LdsAttrOnce attr;
  if (int e = csbsr_lds_attr(attr, reinterpret_cast<const void*>(conv_x3w_kernel<EPI, ABL>), SM_BYTES, "conv_x3w")) return e;
  hipLaunchKernelGGL((conv_x3w_kernel<EPI, ABL>), dim3(g), dim3(256), SM_BYTES, st, k, q);
  CSBSR_LAUNCH_CHECK("csbsr_conv_x3w_forward");
  return 0;
}
static bool x3w_sft_rows_ok(const ConvK& k) {
  const bool sig = k.act == CSBSR_ACT_SIGMOID && k.res_mode == CSBSR_RES_NONE;
  const bool fma = k.act == CSBSR_ACT_NONE && k.res_mode == CSBSR_RES_FMA && k.res && k.res2 && !k.r_lo && !k.r2_lo;
  return (sig || fma) && k.out16 && !k.out32 && !k.o_lo && !k.cbias && !k.mask && !k.accumulate && k.stat_mode == CSBSR_STAT_NONE;
}

extern "C" int csbsr_conv_x3w_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s) {
  CSBSR_CHECK(csbsr_conv_x3w_eligible(d), "conv_x3w: launch not eligible (see csbsr_conv_x3w_eligible)");
  ConvK k;
  if (int rc = conv_desc_to_k(d, k)) return rc;
  XWExtra q;
  q.tiles_x = (unsigned)((d->OW + XW_TW - 1) / XW_TW); q.tiles_y = (unsigned)((d->OH + XW_TH - 1) / XW_TH);
  q.nct = (unsigned)((d->coutp + 127) / 128); q.nch = (unsigned)(d->in[0].c / 32);
  const int ncu = csbsr_cu_budget(reinterpret_cast<hipStream_t>(s));
  const unsigned items = q.tiles_x * q.tiles_y * (unsigned)d->N * q.nct;
  const unsigned g = items < (unsigned)ncu ? items : (unsigned)ncu;
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  const bool fast_rows = conv_epilogue_fast_ok(k);
  const bool sft_rows = !fast_rows && x3w_sft_rows_ok(k);
  g_last_conv_kernel = CONVK_X3W | (fast_rows ? 1 : sft_rows ? 2 : 0) << 8;
#ifdef CSBSR_X3W_ABLATE
  switch (g_conv_x3w_abl) {
    case 1: return launch_x3w<1, 1>(k, q, g, st);
    case 2: return launch_x3w<1, 2>(k, q, g, st);
    case 4: return launch_x3w<1, 4>(k, q, g, st);
    case 8: return launch_x3w<1, 8>(k, q, g, st);
    case 16: return launch_x3w<1, 16>(k, q, g, st);
    case 17: return launch_x3w<1, 17>(k, q, g, st);
    case 25: return launch_x3w<1, 25>(k, q, g, st);
    case 29: return launch_x3w<1, 29>(k, q, g, st);
    case 32: return launch_x3w<1, 32>(k, q, g, st);
    case 34: return launch_x3w<1, 34>(k, q, g, st);
    default: break;
  }
#endif
  if (fast_rows) return launch_x3w<1>(k, q, g, st);
  if (sft_rows) return launch_x3w<2>(k, q, g, st);
  return launch_x3w<0>(k, q, g, st);
}
