// 3x3 stride-1 convolution for the wide LOW-resolution layers -- the SFT scale / shift convolutions (kbpn.py:505-520: 128 s -> 128 s + 441
// and back, i.e. 384 -> 825 -> 384 at stage 3) and their dgrads, 22 % of the training step -- built like csrc/conv_tp.hip instead of
// as an implicit GEMM through an LDS ring:
//
//  * one persistent workgroup per CU (4 waves, one per SIMD) computes an 8 x 32 pixel x 128 cout tile; a wave owns 64 couts x 4 rows x 32
//    pixels (2 x 4 MFMA tiles, 128 accumulator registers);
//  * the pixel operand is staged per 64-channel chunk: the chunk's (8+2) x (32+2) halo goes HBM/L2 -> LDS with
//    global_load_lds_dwordx4 (zero page outside the image) into one of two 48 KB buffers while the previous chunk's nine taps are
//    being multiplied -- every tap reads the same LDS tile at a different offset, so the input crosses the fabric 1.3 times per cout
//    tile instead of 9;
//  * the weight operand never touches LDS: packed in MFMA-fragment order (csbsr_pack_weights_x3, couts permuted so that a lane's
//    accumulators are consecutive channels), 8 KB per wave and K step straight from L2 into registers, two steps ahead; work is
//    ordered cout-tile-major so the 128-cout weight slice everybody streams (0.9 MB for 384 input channels) stays L2-resident;
//    (ablation: all K steps loading the same, L1-resident weights changed 908 -> 926 TF/s: the weight stream is not the stall);
//  * synchronisation: one barrier per CHUNK (nine K steps), none per step; one counted s_waitcnt per chunk for the halo DMA;
//  * epilogue: the general fused row of conv_common.h (bias, per-border-class bias of the folded constant segment, activations,
//    residual add / sub / mul / fma, accumulate, activation mask) straight from registers, once per 54-117 K steps.
#include "common.h"
#include "conv_common.h"
#include "csbsr_debug.h"

//
// KS = 2: the same machine for the 8x8 stride-4 convolutions (k = 2 x stride: DownBlock / UpBlock strided convs and the dgrads of the
// 8x8 stride-4 deconvolutions, kbpn.py:215-262 -- the step's dominant MFMA launches).  Output pixel (y, x) reads input
// (s (y + jy) + py - pad, s (x + jx) + px - pad) for kernel offset (s jy + py, s jx + px): for one input PHASE (py, px) that is a 2 x 2-tap
// stride-1 convolution of the phase's sub-sampled grid, so a chunk is (phase, 64 channels), its halo the (8+1) x (32+1) sub-grid
// pixels of the tile (every input pixel crosses the fabric ONCE per tile instead of once per tap it belongs to), and the accumulators
// simply run over all s^2 x cin/64 chunks: K = 64 s^2 cin/64 x 4 taps = 8192 for 128 channels.
#ifndef X3_RING3
#define X3_RING3 3
#endif
#define X3_RING_OVERRIDE(NT) ((NT) % 3 == 0 ? X3_RING3 : 4)
#define X3_TH 8
#define X3_TW 32
#define X3_SLOTS 9                        // 64 channels = 8 sixteen-byte slots + 1 pad slot: odd pitch, conflict-free rows of pixels
#define X3_PITCH (X3_SLOTS * 16)
#define X3_WSTEP 16384                    // bytes of one K step's weights for the 128-cout tile: [wave mt][k-slice kk][lane][8]
// KS x KS taps per chunk: halo (8 + KS - 1) x (32 + KS - 1) pixels (340 / 297), filled by 48 / 44 wave instructions (a multiple of 4)
constexpr int x3_ninst(int KS) { return (((X3_TH + KS - 1) * (X3_TW + KS - 1) * X3_SLOTS + 63) / 64 + 3) / 4 * 4; }
constexpr int x3_buf(int KS) { return x3_ninst(KS) * 1024; }

typedef unsigned u4v __attribute__((ext_vector_type(4)));

struct X3Extra {
  unsigned tiles_x, tiles_y, nct, nch;   // pixel tiles, 128-cout tiles, chunks (64 channels [x input phase])
  unsigned ncc;                          // 64-channel chunks per input phase (KS = 2)
  unsigned ct_major;                     // 1: cout tile as the slowest index of the item order (A/B timing; default 0, see decode())
};

// (the buffer-descriptor type and builtins exist in the device pass only: the host pass sees an empty kernel body and emits the stub)
#if defined(__HIP_DEVICE_COMPILE__)
// buffer descriptor over [base, base + 2 GB) built from provably wave-uniform halves (offsets >= 0x7fffffff read as zeros)
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t x3_make_rs(const half_t* base) {
  const unsigned long a = reinterpret_cast<unsigned long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi_ << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
#endif

// ABL (builds with -DCSBSR_X3_ABLATE only; timing experiments, results are garbage): 1 halo DMA pieces fetch nothing (out-of-range
// offsets), 2 every K step loads the weights of step 0, 4 no LDS fragment reads in the K loop, 8 no weight loads in the K loop,
// 16 every halo comes from the same few (L2-resident) rows of the input, 32 halo pieces through registers + ds_write instead of LDS-DMA
template <int KS, int ABL = 0>
__global__ __launch_bounds__(256) void conv_x3_kernel(const ConvK p, const X3Extra q, const half_t* __restrict__ zero_page) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int X3_HW = X3_TW + KS - 1, X3_HH = X3_TH + KS - 1;
  constexpr int X3_NINST = x3_ninst(KS), X3_BUF = x3_buf(KS);
  constexpr int NT = KS * KS;                           // K steps (taps) per chunk
  constexpr int RING = X3_RING_OVERRIDE(NT), DIST = RING - 1;      // weight register ring; loads run DIST K steps ahead (NT % RING == 0)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pix = lane & 31, hi = lane >> 5;
  const half_t* zp = zero_page + (lane & 7) * 8;
  const unsigned per_img = q.tiles_x * q.tiles_y, ntiles = per_img * (unsigned)p.N, items = ntiles * q.nct;
  unsigned it = blockIdx.x;
  if (it >= items) return;
  const float slope = (p.act == CSBSR_ACT_PRELU) ? *p.prelu : p.act_slope;
  const EpiFast fe = conv_epilogue_fast_setup(p, slope);
  const half_t* in0 = reinterpret_cast<const half_t*>(p.in[0].ptr);
  const int isy = (int)p.in[0].sy, isx = (int)p.in[0].sx;

  // halo DMA roles (same for every tile and chunk): chunk-slot g = (wid + 4 i) * 64 + lane = (halo pixel, slot)
  int f_ty0, f_tx0, f_c0;
  {
    const int g = wid * 64 + lane;
    const int hq = g / X3_SLOTS;
    f_c0 = g - hq * X3_SLOTS;
    f_ty0 = hq / X3_HW;
    f_tx0 = hq - f_ty0 * X3_HW;
  }
  constexpr int NFI = X3_NINST / 4;                  // 12 / 11 instructions per wave
  constexpr int DQ = 256 / X3_SLOTS, DC = 256 % X3_SLOTS;
  // The DMA of a chunk is issued one instruction per k-slice INSIDE the previous chunk's K loop (a burst of 11-12 address computations
  // + LDS-DMA issues at the chunk boundary idles the MFMA pipe for ~1.5k cycles, a third of a 2 x 2-tap chunk), as buffer loads: the
  // per-lane byte offset and halo coordinates of each instruction are kernel constants, the chunk enters through the (uniform) base
  // of the buffer descriptor, and lanes outside the image get an out-of-range offset, i.e. zeros, from the hardware bounds check.
  const int st = KS == 3 ? 1 : p.stride;               // input pixels per halo pixel
  int voff[NFI], iy0[NFI], ix0[NFI];
  {
    int ty = f_ty0, tx = f_tx0, c = f_c0;
#pragma unroll
    for (int i = 0; i < NFI; ++i) {
      const bool in = ty < X3_HH && c < 8;
      voff[i] = 2 * (ty * st * isy + tx * st * isx + c * 8);
      iy0[i] = in ? st * ty : 0x40000000;
      ix0[i] = st * tx;
      c += DC; tx += DQ;
      if (c >= X3_SLOTS) { c -= X3_SLOTS; ++tx; }
      if (tx >= X3_HW) { tx -= X3_HW; ++ty; }
      if (tx >= X3_HW) { tx -= X3_HW; ++ty; }
    }
  }
  // (uniform) source of chunk `chunk` of tile (n, Y0, X0): the input coordinates of halo pixel (0, 0) and its address
  auto chunk_src = [&](int n, int Y0, int X0, int chunk, int& by, int& bx) -> const half_t* {
    int coff;
    if (ABL & 16) { n = 0; Y0 = 8 * (int)(blockIdx.x & 31); X0 = 0; chunk = 0; }      // every halo from one small (L2-resident) region
    if (KS == 3) { by = Y0 - 1; bx = X0 - 1; coff = chunk * 64; }
    else {
      const int ph = chunk / (int)q.ncc, py = ph / p.stride;
      by = Y0 * p.stride + py - p.pad; bx = X0 * p.stride + (ph - py * p.stride) - p.pad; coff = (chunk - ph * (int)q.ncc) * 64;
    }
    return in0 + n * p.in[0].sn + (long)by * p.in[0].sy + (long)bx * p.in[0].sx + coff;
  };
  auto issue_one = [&](__amdgpu_buffer_rsrc_t rs, int by, int bx, int i, int buf) __attribute__((always_inline)) {
    const bool ok = !(ABL & 1) && (unsigned)(iy0[i] + by) < (unsigned)p.H && (unsigned)(ix0[i] + bx) < (unsigned)p.W;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + buf * X3_BUF + (wid + 4 * i) * 1024), 16,
                                             ok ? voff[i] : -1, 0, 0, 0);
  };
  // ABL & 32: the same pieces through registers (buffer load -> VGPRs, ds_write_b128 LAG k-slices later) instead of LDS-DMA
  constexpr bool STG = (ABL & 32) != 0;
  constexpr int LAG = KS == 3 ? 8 : 4;
  static_assert(!STG || NFI + LAG <= NT * 4, "staged pieces must be written inside the chunk that loads them");
  auto load_piece = [&](__amdgpu_buffer_rsrc_t rs, int by, int bx, int i) __attribute__((always_inline)) -> u4v {
    const bool ok = (unsigned)(iy0[i] + by) < (unsigned)p.H && (unsigned)(ix0[i] + bx) < (unsigned)p.W;
    return __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? voff[i] : -1, 0, 0);
  };
  auto store_piece = [&](const u4v& v, int i, int buf) __attribute__((always_inline)) {
    *reinterpret_cast<u4v*>(smem + buf * X3_BUF + (wid + 4 * i) * 1024 + lane * 16) = v;
  };
  auto issue_x = [&](int n, int Y0, int X0, int chunk, int buf) {
    int by, bx;
    const __amdgpu_buffer_rsrc_t rs = x3_make_rs(chunk_src(n, Y0, X0, chunk, by, bx));
    if constexpr (STG) {
      u4v t[NFI];
#pragma unroll
      for (int i = 0; i < NFI; ++i) t[i] = load_piece(rs, by, bx, i);
#pragma unroll
      for (int i = 0; i < NFI; ++i) store_piece(t[i], i, buf);
    } else {
#pragma unroll
      for (int i = 0; i < NFI; ++i) issue_one(rs, by, bx, i, buf);
    }
  };
  auto decode = [&](unsigned item, int& ct, int& n, int& Y0, int& X0) {
    // The cout tiles of ONE pixel tile are consecutive items, and an XCD owns a contiguous run of the item order (xcd_remap: the
    // launcher keeps the grid a multiple of 8, so item % 8 == blockIdx % 8 == the XCD): the nct workgroups that need the same input
    // halo run side by side on one XCD at the same time, the halo comes from HBM once and the other nct - 1 requests meet it in that
    // XCD's L2 (merged misses or hits).  With the cout tile as the SLOWEST index every pass over the image re-fetched the whole input
    // from HBM (nct x the input bytes: FETCH_SIZE 4.8 GB per launch for 1.6 GB algorithmic at 825 -> 384).  The nct weight slices an
    // XCD now streams at once (nct x 128 couts x 64 ch x 9 taps x 2 B = 0.44 MB per chunk) stay L2-resident because its 32 workgroups
    // walk the chunks in loose lock step.
    unsigned tile;
    if (q.ct_major) { ct = item / ntiles; tile = item - ct * ntiles; }
    else {
      item = xcd_remap(item, items);
      tile = item / q.nct;
      ct = item - tile * q.nct;
    }
    n = tile / per_img;
    const unsigned r_ = tile - n * per_img;
    Y0 = (r_ / q.tiles_x) * X3_TH; X0 = (r_ % q.tiles_x) * X3_TW;
  };
  // a wave's weights of K step (ct, chunk, tap): 4 fragment-ordered KB, one 16-byte load per lane each
  // (a wave = 64 couts x 4 rows x 32 pixels: two A fragments per k-slice from L2, four B fragments from LDS for eight MFMAs -- half the
  // LDS reads per MFMA of a 32-cout x 256-pixel wave, which ran LDS-bound: one ds_read_b128 per MFMA per wave is 128 B/clk per CU)
  const int mh = wid & 1, rq = wid >> 1;
  const unsigned wlane = (unsigned)(mh * 8192 + lane * 16);
  auto load_w = [&](int ct, int step, h8 (&w)[2][4]) __attribute__((always_inline)) {
    if (ABL & 2) step = 0;
    const char* b = reinterpret_cast<const char*>(p.wt) + ((size_t)ct * q.nch * NT + step) * X3_WSTEP;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) w[mt][kk] = *reinterpret_cast<const h8*>(b + (wlane + mt * 4096 + kk * 1024));
  };
  const char* xl = smem + ((4 * rq) * X3_HW + pix) * X3_PITCH + hi * 16;      // the wave's first row of the halo tile

  int ct, n, Y0, X0;
  decode(it, ct, n, Y0, X0);
  issue_x(n, Y0, X0, 0, 0);
  h8 wreg[RING][2][4];                                    // K step g = chunk * NT + tap lives in wreg[tap % RING] (NT % RING == 0)
#pragma unroll
  for (int g = 0; g < DIST; ++g) load_w(ct, g, wreg[g]);
  const int nsteps = (int)q.nch * NT;

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

    for (int c = 0; c < (int)q.nch; ++c) {
      // chunk c's halo has landed everywhere, and every wave is done with the other buffer (chunk c - 1): refill that one.  The only
      // vector-memory instructions issued after this chunk's last DMA piece (k-slice NFI - 1 of the previous chunk) are the weight
      // loads of the taps that followed it (8 each) -- except right after an epilogue, where the count is simply drained.
      if constexpr (STG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's staged pieces are in LDS
      else if (c == 0) {
        // after an epilogue of the straight-line rows its CONV_TILE_STORES (unconditional buffer) stores may stay in flight (conv_common.h)
        if ((ABL & 1024) != 0 && it != blockIdx.x) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((8 * (NT - 1 - (NFI - 1) / 4) + CONV_TILE_STORES) > 63 ? 63 : (8 * (NT - 1 - (NFI - 1) / 4) + CONV_TILE_STORES)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NT - 1 - (NFI - 1) / 4)) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int bnext = (int)((par + c + 1) & 1);
      // the next chunk (past the tile's last one: the next tile's first; past the last tile: a harmless refetch that keeps the
      // instruction count uniform), issued piece by piece below
      int nby, nbx;
      const half_t* nsrc = c + 1 < (int)q.nch ? chunk_src(n, Y0, X0, c + 1, nby, nbx)
                                              : (itn < items ? chunk_src(nn, Y0n, X0n, 0, nby, nbx) : chunk_src(n, Y0, X0, 0, nby, nbx));
      const __amdgpu_buffer_rsrc_t nrs = x3_make_rs(nsrc);
      asm volatile("" ::: "memory");
      const char* xb = xl + ((par + c) & 1) * X3_BUF;
      h8 bfr[4];
      u4v stg[LAG];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bfr[nt] = *reinterpret_cast<const h8*>(xb + (nt * X3_HW) * X3_PITCH);      // tap (0,0), k-slice 0
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        // weights DIST K steps ahead (past the tile's last step: the next tile's first ones)
        {
          const int g = c * NT + tap + DIST;
          if (ABL & 8) {}
          else if (g < nsteps) load_w(ct, g, wreg[(tap + DIST) % RING]);
          else load_w(ctn, g - nsteps, wreg[(tap + DIST) % RING]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int ky = tap / KS, kx = tap % KS;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap % RING][0][kk], bfr[i], acc[0][i], 0, 0, 0);
            acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap % RING][1][kk], bfr[i], acc[1][i], 0, 0, 0);
            // the same row's fragment of the next k-slice / next tap (the next chunk starts over after its barrier)
            if (ABL & 4) {}
            else if (kk < 3) bfr[i] = *reinterpret_cast<const h8*>(xb + ((i + ky) * X3_HW + kx) * X3_PITCH + (kk + 1) * 32);
            else if (tap < NT - 1) bfr[i] = *reinterpret_cast<const h8*>(xb + ((i + (tap + 1) / KS) * X3_HW + (tap + 1) % KS) * X3_PITCH);
            if constexpr (STG) {
              if (i == 1) {
                const int sl = tap * 4 + kk;
                if (sl >= LAG && sl - LAG < NFI) store_piece(stg[(sl - LAG) % LAG], sl - LAG, bnext);
                if (sl < NFI) stg[sl % LAG] = load_piece(nrs, nby, nbx, sl);
              }
            } else if (i == 1 && tap * 4 + kk < NFI) issue_one(nrs, nby, nbx, tap * 4 + kk, bnext);       // one DMA piece per k-slice
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }

    // ---- epilogue: acc[mt][nt][8 pair + e] = cout 128 ct + 64 mh + 32 mt + 16 pair + 8 hi + e of pixel (row 4 rq + nt, column pix)
    const int ox = X0 + pix;
    if constexpr ((ABL & 1024) != 0) {
      // straight-line rows (conv_common.h; the host picks the instance by conv_epilogue_fast_ok), every load ahead of the stores
      float d0[4][8], d1[4][8];
      const bool interior = Y0 + 4 * rq >= 2 && Y0 + 4 * rq + 3 <= p.OH - 3 && X0 >= 2 && X0 + 31 <= p.OW - 3;
      conv_epilogue_fast_tile<false>(p, fe, acc, 128 * ct + 64 * mh + 8 * hi, n, Y0 + 4 * rq, ox, interior, d0, d1);
    } else if constexpr ((ABL & 2048) != 0) {
      // the SFT conv1's (kbpn.py:505-516): bias + sigmoid (the scale branch) or bias + res x res2 (the shift branch, whose epilogue
      // finishes the layer: f x scale + shift) -- plain fp16 output, nothing else (x3_sft_rows_ok on the host); the same batching
      const int oxc = ox < p.OW ? ox : p.OW - 1;
#pragma unroll
      for (int mp = 0; mp < 4; ++mp) {
        const int co = 128 * ct + 64 * mh + 16 * mp + 8 * hi;
        const int coc = co < p.coutp ? co : p.coutp - 8;
        float bias[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[e] = 0.f;
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int i_ = coc + e;
            const float t = p.bias[n * p.bias_sn + (i_ < p.cout ? i_ : p.cout - 1)];
            bias[e] = i_ < p.cout ? t : 0.f;
          }
        }
        h8 r1[4], r2[4];
        int oyc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int oy = Y0 + 4 * rq + nt;
          oyc[nt] = oy < p.OH ? oy : p.OH - 1;
        }
        if (p.act != CSBSR_ACT_SIGMOID) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            r1[nt] = *reinterpret_cast<const h8*>(p.res + n * p.r_sn + oyc[nt] * p.r_sy + oxc * p.r_sx + coc);
            r2[nt] = *reinterpret_cast<const h8*>(p.res2 + n * p.r2_sn + oyc[nt] * p.r2_sy + oxc * p.r2_sx + coc);
          }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int oy = Y0 + 4 * rq + nt;
          if (oy >= p.OH || ox >= p.OW || co >= p.coutp) continue;
          half_t* o = p.out16 + n * p.o_sn + oy * p.o_sy + ox * p.o_sx + co;
          h8 hv;
          if (p.act == CSBSR_ACT_SIGMOID) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (half_t)(co + e < p.cout ? 1.f / (1.f + __expf(-(acc[mp >> 1][nt][8 * (mp & 1) + e] * p.out_scale + bias[e]))) : 0.f);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              hv[e] = (half_t)((co + e < p.cout ? acc[mp >> 1][nt][8 * (mp & 1) + e] * p.out_scale + bias[e] : 0.f) + (float)r1[nt][e] * (float)r2[nt][e]);
          }
          *reinterpret_cast<h8*>(o) = hv;
        }
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int oy = Y0 + 4 * rq + nt;
#pragma unroll
        for (int mp = 0; mp < 4; ++mp) {
          const int mt = mp >> 1, pair = mp & 1;
          const int co = 128 * ct + 64 * mh + 32 * mt + 16 * pair + 8 * hi;
          if (oy >= p.OH || ox >= p.OW || co >= p.coutp) continue;
          float v[8], bias[8], s0[8], s1[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = acc[mt][nt][8 * pair + e];
            bias[e] = (p.bias && co + e < p.cout) ? p.bias[n * p.bias_sn + co + e] : 0.f;
          }
          conv_epilogue_row(p, v, bias, slope, co, n, oy, ox, s0, s1);
        }
      }
    }
    ct = ctn; n = nn; Y0 = Y0n; X0 = X0n;
  }
#endif
}

// ---- weights in K-step order: dst[ct][chunk][tap][mt][kk][lane][e] = W(row 128 ct + 32 mt + perm(lane % 32), channel 64 chunk + 16 kk
// + 8 (lane / 32) + e, tap), perm(8 q + 4 h + j) = 16 (q / 2) + 8 h + 4 (q % 2) + j (a lane's accumulator registers 8 pair .. 8 pair + 7
// are then consecutive channels).  kind 0: forward, W is OIHW [row][channel][ky][kx]; kind 1: dgrad of the stride-1 conv: rows are the
// conv's input channels, contracted channels its outputs, taps flipped (W[channel][row][2 - ky][2 - kx]).
struct PackX3K { const float* w; half_t* dst; int kind, D1, nch, nct, c_real, rows_real, row_off, k_off, ksize, stride, ncc; };
__global__ void pack_weights_x3_kernel(const PackX3K p, long total) {
  const int NT = p.kind == 2 ? 4 : 9;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), blk = (int)((i >> 9) & 15);
    const long stage = i >> 13;
    const int tap = (int)(stage % NT);
    const long t2 = stage / NT;
    const int chunk = (int)(t2 % p.nch), ct = (int)(t2 / p.nch);
    const int mt = blk >> 2, kk = blk & 3;
    const int m = lane & 31, q_ = m >> 3, h_ = (m >> 2) & 1;
    const int row = 128 * ct + 32 * mt + 16 * (q_ >> 1) + 8 * h_ + 4 * (q_ & 1) + (m & 3);
    float v = 0.f;
    if (p.kind == 2) {
      // chunk = (input phase, 64 channels); tap (jy, jx) is kernel offset (s jy + py, s jx + px)
      const int ph = chunk / p.ncc, cc = chunk - ph * p.ncc;
      const int c = 64 * cc + 16 * kk + 8 * (lane >> 5) + e;
      const int kh = p.stride * (tap >> 1) + ph / p.stride, kw = p.stride * (tap & 1) + ph % p.stride;
      if (row < p.rows_real && c < p.c_real)
        v = p.w[(((long)(p.row_off + row) * p.D1 + p.k_off + c) * p.ksize + kh) * p.ksize + kw];
    } else {
      const int c = 64 * chunk + 16 * kk + 8 * (lane >> 5) + e;
      const int ky = tap / 3, kx = tap % 3;
      if (row < p.rows_real && c < p.c_real) {
        const int rr = p.row_off + row, cc = p.k_off + c;
        if (p.kind == 0) v = p.w[(((long)rr * p.D1 + cc) * 3 + ky) * 3 + kx];
        else v = p.w[(((long)cc * p.D1 + rr) * 3 + (2 - ky)) * 3 + (2 - kx)];
      }
    }
    p.dst[i] = (half_t)v;
  }
}

extern "C" int64_t csbsr_packed_weight_elems_x3(int32_t c_real, int32_t rows_real) {
  const int nch = (round_up(c_real, 8) + 63) / 64, nct = (round_up(rows_real, 8) + 127) / 128;
  return (int64_t)nct * nch * 9 * (X3_WSTEP / 2);
}

extern "C" int64_t csbsr_packed_weight_elems_x3_strided(int32_t stride, int32_t c_real, int32_t rows_real) {
  const int ncc = (round_up(c_real, 8) + 63) / 64, nct = (round_up(rows_real, 8) + 127) / 128;
  return (int64_t)nct * stride * stride * ncc * 4 * (X3_WSTEP / 2);
}

static int pack_x3_launch(PackX3K& p, long total, csbsr_stream_t s, const char* what) {
  const long nb = (total + 255) / 256;
  hipLaunchKernelGGL(pack_weights_x3_kernel, dim3((int)(nb > 8192 ? 8192 : nb)), dim3(256), 0, reinterpret_cast<hipStream_t>(s), p, total);
  CSBSR_LAUNCH_CHECK(what);
  return 0;
}

extern "C" int csbsr_pack_weights_x3(const float* w, void* dst, int32_t kind, int32_t D0, int32_t D1, int32_t c_real, int32_t rows_real,
                                     int32_t row_off, int32_t k_off, csbsr_stream_t s) {
  CSBSR_CHECK(w && dst && (kind == 0 || kind == 1), "pack_x3: bad args");
  const int kdim = kind == 0 ? D1 : D0, rdim = kind == 0 ? D0 : D1;
  CSBSR_CHECK(c_real >= 1 && rows_real >= 1 && k_off >= 0 && k_off + c_real <= kdim && row_off >= 0 && row_off + rows_real <= rdim,
              "pack_x3: range out of bounds");
  PackX3K p;
  p.w = w; p.dst = reinterpret_cast<half_t*>(dst); p.kind = kind; p.D1 = D1;
  p.nch = (round_up(c_real, 8) + 63) / 64; p.nct = (round_up(rows_real, 8) + 127) / 128;
  p.c_real = c_real; p.rows_real = rows_real; p.row_off = row_off; p.k_off = k_off;
  p.ksize = 3; p.stride = 1; p.ncc = p.nch;
  return pack_x3_launch(p, csbsr_packed_weight_elems_x3(c_real, rows_real), s, "csbsr_pack_weights_x3");
}

// k = 2 x stride convolution (or the dgrad of such a ConvTranspose2d): W is [row][contracted channel][kh][kw] -- a Conv2d's OIHW
// parameter, or a ConvTranspose2d's IOHW parameter seen from its dgrad (rows = its input channels, contracted = its outputs)
extern "C" int csbsr_pack_weights_x3_strided(const float* w, void* dst, int32_t D0, int32_t D1, int32_t ksize, int32_t stride,
                                             int32_t c_real, int32_t rows_real, int32_t row_off, int32_t k_off, csbsr_stream_t s) {
  CSBSR_CHECK(w && dst && stride >= 2 && ksize == 2 * stride, "pack_x3_strided: bad args (kernel size must be 2 x stride)");
  CSBSR_CHECK(c_real >= 1 && rows_real >= 1 && k_off >= 0 && k_off + c_real <= D1 && row_off >= 0 && row_off + rows_real <= D0,
              "pack_x3_strided: range out of bounds");
  PackX3K p;
  p.w = w; p.dst = reinterpret_cast<half_t*>(dst); p.kind = 2; p.D1 = D1;
  p.ncc = (round_up(c_real, 8) + 63) / 64; p.nch = stride * stride * p.ncc; p.nct = (round_up(rows_real, 8) + 127) / 128;
  p.c_real = c_real; p.rows_real = rows_real; p.row_off = row_off; p.k_off = k_off;
  p.ksize = ksize; p.stride = stride;
  return pack_x3_launch(p, csbsr_packed_weight_elems_x3_strided(stride, c_real, rows_real), s, "csbsr_pack_weights_x3_strided");
}

static int g_conv_x3_mode = 1;      // 0 off, 1 launches that fill the chip, 2 every eligible launch (tests)
static int g_conv_x3_abl = 0;       // bits 4.. of the debug mode: ablation variant of the 3x3 kernel (CSBSR_X3_ABLATE builds)
static int g_conv_x3_ct_major = 0;  // bit 3 of the debug mode: the old cout-tile-major item order (A/B timing)
extern "C" void csbsr_debug_set_conv_x3(int mode) { g_conv_x3_mode = mode & 7; g_conv_x3_ct_major = (mode >> 3) & 1; g_conv_x3_abl = mode >> 4; }

// Which launches take this kernel.  3x3, stride 1, pad 1, dilation 1: ONE plain-fp16 input segment whose padded channels are a multiple of
// 64, >= 128 (measured at N = 4, 448^2, with the halo DMA issued inside the K loop: 825 -> 384 1041 TF/s against 899 for the
// LDS-ring implicit GEMM, its dgrad-shaped twin 1069, 384 -> 825 809 against 800, but 256 -> 697 641 against 817: with few chunks the
// per-tile epilogue and pipeline restart outweigh the K loop's gain; >= 128 when forced by csbsr_debug_set_conv_x3(2)).  k = 2 x stride
// (8x8 stride 4): see below.  Both: >= 72 padded output channels, fp16 output; any fused epilogue of the general kernels except
// statistics, the fp32 side output and split (hi + lo) operands.
static bool x3_is_strided(const csbsr_conv_desc_t* d) { return d->stride >= 2 && d->KH == 2 * d->stride; }

extern "C" int32_t csbsr_conv_x3_eligible(const csbsr_conv_desc_t* d) {
  if (!d || !g_conv_x3_mode || d->transposed || d->KH != d->KW || d->dil != 1) return 0;
  const bool strided = x3_is_strided(d);
  if (strided) {
    // k = 2 x stride (8x8 stride 4): 64 .. 512 padded input channels in whole chunks, pad < stride, the exact strided output size
    if (d->pad < 0 || d->pad >= d->stride || d->stride > 4) return 0;
    if (d->OH != (d->H + 2 * d->pad - d->KH) / d->stride + 1 || d->OW != (d->W + 2 * d->pad - d->KW) / d->stride + 1) return 0;
    if (d->in[0].c < 64 || d->in[0].c > 512 || d->in[0].c % 64 != 0) return 0;
  } else {
    if (d->KH != 3 || d->stride != 1 || d->pad != 1 || d->OH != d->H || d->OW != d->W) return 0;
    // (>= 128 input channels since round 6 -- 128 -> 569 at 448^2: 2.89 ms against 3.20 on the LDS-ring kernel at N = 8, 1.53 / 1.69 at N = 4;
    // 256 -> 697: 5.35 / 5.60 and 2.84 / 2.88 -- ; the >= 384 of rounds 2-5 was measured on round 2's kernel: 256 -> 697 641 against 817 TF/s)
    const int min_c = 128;
    if (d->in[0].c < min_c || d->in[0].c % 64 != 0) return 0;
  }
  if (d->in[1].c != 0 || d->in[0].sx == 0) return 0;
  if (d->coutp < 72 || !d->out16 || d->out32 || d->o_lo || d->r_lo || d->r2_lo) return 0;
  if (d->stat_mode != CSBSR_STAT_NONE) return 0;
  if (d->dact_bias || d->dact_prelu || d->dres) return 0;
  if (d->in[0].sy >= (1l << 31) / 2 / (strided ? 4 * (X3_TH + 1) : 1)) return 0;
  if (g_conv_x3_mode == 1 && (long)d->N * d->OH * d->OW * ((d->coutp + 127) / 128) < 512L * X3_TH * X3_TW) return 0;
  return 1;
}

static half_t* g_x3_zero_page[CSBSR_MAX_DEVICES] = {};

template <int KS, int ABL = 0>
static int launch_x3(const ConvK& k, const X3Extra& q, unsigned g, const half_t* zp, hipStream_t st) {
  constexpr int SM_BYTES = 2 * x3_buf(KS);
  static LdsAttrOnce attr;
  if (int e = csbsr_lds_attr(attr, reinterpret_cast<const void*>(conv_x3_kernel<KS, ABL>), SM_BYTES, "conv_x3")) return e;
  hipLaunchKernelGGL((conv_x3_kernel<KS, ABL>), dim3(g), dim3(256), SM_BYTES, st, k, q, zp);
  CSBSR_LAUNCH_CHECK("csbsr_conv_x3_forward");
  return 0;
}
// the instance with the straight-line epilogue rows where they cover the launch (template bit 1024), the general rows elsewhere
// the SFT conv1 epilogues: sigmoid, or no activation with the FMA residual; nothing else fused
static bool x3_sft_rows_ok(const ConvK& k) {
  const bool sig = k.act == CSBSR_ACT_SIGMOID && k.res_mode == CSBSR_RES_NONE;
  const bool fma = k.act == CSBSR_ACT_NONE && k.res_mode == CSBSR_RES_FMA && k.res && k.res2 && !k.r_lo && !k.r2_lo;
  return (sig || fma) && k.out16 && !k.out32 && !k.o_lo && !k.cbias && !k.mask && !k.accumulate && k.stat_mode == CSBSR_STAT_NONE;
}
template <int KS>
static int launch_x3_epi(const ConvK& k, const X3Extra& q, unsigned g, const half_t* zp, hipStream_t st) {
  if (conv_epilogue_fast_ok(k)) return launch_x3<KS, 1024>(k, q, g, zp, st);
  if (KS == 3 && x3_sft_rows_ok(k)) return launch_x3<3, 2048>(k, q, g, zp, st);
  return launch_x3<KS, 0>(k, q, g, zp, st);
}

extern "C" int csbsr_conv_x3_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s) {
  CSBSR_CHECK(csbsr_conv_x3_eligible(d), "conv_x3: launch not eligible (see csbsr_conv_x3_eligible)");
  ConvK k;
  if (int rc = conv_desc_to_k(d, k)) return rc;
  const bool strided = x3_is_strided(d);
  X3Extra q;
  q.tiles_x = (unsigned)((d->OW + X3_TW - 1) / X3_TW); q.tiles_y = (unsigned)((d->OH + X3_TH - 1) / X3_TH);
  q.nct = (unsigned)((d->coutp + 127) / 128); q.ncc = (unsigned)(d->in[0].c / 64);
  q.nch = strided ? q.ncc * (unsigned)(d->stride * d->stride) : q.ncc;
  q.ct_major = (unsigned)g_conv_x3_ct_major;
  int dev = 0, ncu = 256;
  CSBSR_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < CSBSR_MAX_DEVICES, "conv_x3: no current device");
  ncu = csbsr_cu_budget(reinterpret_cast<hipStream_t>(s));      // the stream's CU partition (csrc/streams.hip), else the whole device
  if (!g_x3_zero_page[dev]) {
    CSBSR_CHECK(hipMalloc(reinterpret_cast<void**>(&g_x3_zero_page[dev]), 256) == hipSuccess, "conv_x3: zero page alloc failed");
    (void)hipMemset(g_x3_zero_page[dev], 0, 256);
  }
  const unsigned items = q.tiles_x * q.tiles_y * (unsigned)d->N * q.nct;
  const unsigned g = items < (unsigned)ncu ? items : (unsigned)ncu;
  const bool fast_rows = conv_epilogue_fast_ok(k);      // (launch_x3_epi picks the instance by the same test)
  g_last_conv_kernel = (strided ? (fast_rows ? CONVK_X3SF : CONVK_X3S) : (fast_rows ? CONVK_X3F : CONVK_X3)) |
                       ((!strided && !fast_rows && x3_sft_rows_ok(k)) ? 1 : 0) << 8;      // bit 8: the <3, 2048> instance (sigmoid / FMA rows)
#ifdef CSBSR_X3_ABLATE
  if (!strided) {
    hipStream_t st_ = reinterpret_cast<hipStream_t>(s);
    switch (g_conv_x3_abl) {
      case 1: return launch_x3<3, 1>(k, q, g, g_x3_zero_page[dev], st_);
      case 2: return launch_x3<3, 2>(k, q, g, g_x3_zero_page[dev], st_);
      case 4: return launch_x3<3, 4>(k, q, g, g_x3_zero_page[dev], st_);
      case 8: return launch_x3<3, 8>(k, q, g, g_x3_zero_page[dev], st_);
      case 5: return launch_x3<3, 5>(k, q, g, g_x3_zero_page[dev], st_);
      case 9: return launch_x3<3, 9>(k, q, g, g_x3_zero_page[dev], st_);
      case 12: return launch_x3<3, 12>(k, q, g, g_x3_zero_page[dev], st_);
      case 13: return launch_x3<3, 13>(k, q, g, g_x3_zero_page[dev], st_);
      case 16: return launch_x3<3, 16>(k, q, g, g_x3_zero_page[dev], st_);
      case 24: return launch_x3<3, 24>(k, q, g, g_x3_zero_page[dev], st_);
      default: break;
    }
  }
  // 32: register-staged halo pieces (results are correct with this one).  Measured against the LDS-DMA pieces in one process
  // (scripts/x3_stage_ab.py, N = 4): 825 -> 384 1104 vs 1105 TF/s, its dgrad 842 vs 820, 384 -> 825 893 vs 868, the 8x8 stride-4 conv 945 vs
  // 970 -- so the LDS-DMA instruction itself is not what the halo stream costs
  if (g_conv_x3_abl == 32)
    return strided ? launch_x3<2, 32>(k, q, g, g_x3_zero_page[dev], reinterpret_cast<hipStream_t>(s))
                   : launch_x3<3, 32>(k, q, g, g_x3_zero_page[dev], reinterpret_cast<hipStream_t>(s));
#endif
  return strided ? launch_x3_epi<2>(k, q, g, g_x3_zero_page[dev], reinterpret_cast<hipStream_t>(s))
                 : launch_x3_epi<3>(k, q, g, g_x3_zero_page[dev], reinterpret_cast<hipStream_t>(s));
}
