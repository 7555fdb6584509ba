// Direct 3x3 convolution for the 32 / 49-channel layers that run at FULL (HR) resolution -- the kernel predictor's fe_SR.2-4,
// fe_kernel.1, fe_cat.1-2 and their dgrads (kbpn.py:521-578): 140-175 FLOP per byte of input + output, i.e. HBM-bound layers that the
// implicit-GEMM kernels ran ~4x off their byte roofline (each of the 9 taps re-gathers the pixel operand through the L1/LDS path and
// every fragment pair is read from LDS: a 32-cout K slice is 4 MFMAs per wave against ~60 address instructions).
//
//  * ONE persistent workgroup per CU walks 16 x 32 output tiles; a tile's (16+2) x (32+2) input halo goes HBM -> LDS ONCE (16-byte
//    channel chunks, out-of-image pixels zero-filled by the buffer descriptor's bounds check) -- 1.2x the tile's own bytes instead of
//    9x -- into a ring of 2-3 tile buffers: the pieces of the tile two ahead are issued one at a time between the MFMAs of the current
//    one, so ~100 KB per CU are always under way (see the kernel's comment for what the first version -- one tile per workgroup, 2-3
//    workgroups per CU -- lost);
//  * the WEIGHTS never touch LDS: they are packed in MFMA-fragment order (csbsr_pack_weights_hr) and each lane keeps its A fragments
//    of all K steps of one 32-cout tile in registers (18 steps x 4 VGPRs for 32 channels, 32 x 4 for 49 -> 56), so the K loop is one
//    ds_read_b128 of the pixel operand per MFMA and nothing else;
//  * K is flattened over (tap, 8-channel chunk): an MFMA K step = two chunks, one per half-wave, each half-wave addressing its own
//    (tap, chunk) -- 56-channel maps need no padding to 64 (63 chunks -> 32 steps);
//  * bank conflicts: a 64-byte pixel pitch (32 channels) would put pixels p and p + 4 on the same 16-byte slots, so a 32-channel
//    pixel is laid out on FIVE slots (80 bytes, the fifth slot zero-filled): an odd pitch in 16-byte slots is conflict-free
//    for the 16 consecutive-pixel lanes of a ds_read_b128 group, as the 112-byte pitch of 56 channels already is -- and with no XOR
//    swizzle every fragment address is one per-lane base register plus a compile-time offset (no address arithmetic in the K loop);
//  * epilogue in registers: v_permlane32_swap turns the MFMA layout into 8 consecutive couts per lane, branch-free activation, optional
//    fused activation-derivative mask (dgrads; requested one tile ahead), optional global-average-pool sums (fe_cat.2), 16-byte stores;
//  * a 49-cout layer's two 32-cout tiles are computed from the same halo tile (the waves split by cout tile).
//
// Replaces F.conv2d at kbpn.py:536-547 (via ConvBlock) and its autograd dgrad for the eligible layers.
#include "common.h"
#include "conv_common.h"
#include <cstdlib>

#define HR_TH 16
#define HR_TW 32
// waves per workgroup: 8 (two per SIMD: one wave's epilogue runs beside the other's K loop) except for the 56-channel 3x3 layers, whose
// 32 weight fragments per lane do not fit 256 registers next to the rest (4 waves, 512 registers each)
constexpr int hr_nw(int CH8, int TAPS) { return (CH8 == 7 && TAPS == 9) ? 4 : 8; }

struct ConvHrK {
  const half_t* in; long i_sn, i_sy, i_sx;
  int N, H, W;
  const half_t* wt;                 // [cout tiles][NKS][64 lanes][8] fragment order
  int cout, coutp, ntile_c;         // real / padded couts, 32-cout tiles
  half_t* out16; long o_sn, o_sy, o_sx;
  int act; float slope;
  const half_t* mask; long m_sn, m_sy, m_sx; float mask_slope;
  float* stat;                      // optional [N][coutp] per-sample channel sums of act(conv) (global average pool)
  float* stat_part;                 // ... as order-fixed partial rows [N][workgroups][waves][coutp] (launcher: zeroed before, folded after)
  unsigned tiles_x, tiles_y;
  const float* cbias;               // optional [N][25][coutp] two-ring position-class bias (csbsr_conv_desc_t.cbias, cbias_mode 1), added before the activation
};

// Ring depth: as many halo tiles as fit the 160 KB of LDS, at most 3 (the tiles NBUF - 1 ahead are in flight while one is multiplied)
constexpr int hr_ninst(int CH8, int TAPS) {
  const int halo = TAPS == 9 ? 1 : 0, slots = (CH8 % 2) ? CH8 : CH8 + 1, HR_NW = hr_nw(CH8, TAPS);
  return (((HR_TH + 2 * halo) * (HR_TW + 2 * halo) * slots + 63) / 64 + HR_NW - 1) / HR_NW * HR_NW;
}
constexpr int hr_nbuf(int CH8, int TAPS) { return 3 * hr_ninst(CH8, TAPS) * 1024 + 1024 <= 160 * 1024 ? 3 : 2; }
constexpr int hr_smem(int CH8, int TAPS) { return hr_nbuf(CH8, TAPS) * hr_ninst(CH8, TAPS) * 1024 + 16 + 64 * 4; }

// TAPS = 9 (3x3, one-pixel halo) or 1 (the 1x1 layers of the same chains -- fe_SR.1, fe_cat.0 and their dgrads: no halo, two or four
// MFMA K steps per 32 pixels, a pure HBM stream).  NCT = 32-cout tiles a workgroup computes from ONE halo tile (2: waves 0-1 take the
// first, waves 2-3 the second, eight rows each -- a 49-cout layer reads its input once instead of once per cout tile).
// One persistent workgroup per CU walks its tiles through a ring of NBUF halo buffers: the DMA of the tile NBUF - 1 ahead is issued one
// piece at a time between the MFMAs (buffer loads with kernel-constant per-lane offsets, out-of-image pixels zero-filled by the
// descriptor's bounds check), so ~100 KB per CU are under way at any time -- with one tile per workgroup and 2-3 workgroups per CU the
// loads of a workgroup were never in flight while it multiplied, and the layers ran at 2-3.6 TB/s.
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t hr_make_rs(const half_t* base) {
  const unsigned long a = reinterpret_cast<unsigned long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi_ << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
#endif

template <int CH8, bool STAT, int TAPS, int NCT, bool MASK, bool CB = false>
__global__ __launch_bounds__(64 * hr_nw(CH8, TAPS)) void conv_hr_kernel(const ConvHrK p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int HR_NW = hr_nw(CH8, TAPS);
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int HR_HW = HR_TW + 2 * HALO, HR_HH = HR_TH + 2 * HALO, HR_NPIX = HR_HH * HR_HW;      // 18 x 34 = 612 halo pixels (3x3)
  constexpr int NCHUNK = TAPS * CH8;                    // K in 8-channel chunks
  constexpr int NKS = (NCHUNK + 1) / 2;                 // MFMA K steps (16 channels = two chunks)
  constexpr int SLOTS = (CH8 % 2) ? CH8 : CH8 + 1;      // 16-byte slots per pixel in LDS: odd, so consecutive pixels walk all banks
  constexpr int PIXB = SLOTS * 16;                      // bytes per pixel in LDS
  constexpr int NINST = hr_ninst(CH8, TAPS);            // wave instructions (pieces) per tile, a multiple of 4
  constexpr int NFI = NINST / HR_NW;                    // per wave
  constexpr int TILE_BYTES = NINST * 1024;
  constexpr int NBUF = hr_nbuf(CH8, TAPS);
  constexpr int ZERO_OFF = NBUF * TILE_BYTES;           // one zero chunk for the padded half K step
  constexpr int R = HR_TH / (HR_NW / NCT);              // tile rows per wave
  constexpr int NMF = R * NKS;                          // MFMAs per wave and tile: the pieces are spread over them
  constexpr int SP = NMF >= NFI ? NMF / NFI : 1, PP = NMF >= NFI ? 1 : (NFI + NMF - 1) / NMF;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pix = lane & 31, hi = lane >> 5;
  const float aslope = p.act == CSBSR_ACT_RELU ? 0.f : (p.act == CSBSR_ACT_LRELU ? p.slope : 1.f);
  const bool has_out = p.out16 != nullptr;
  const int ctl = NCT == 2 ? wid / (HR_NW / 2) : 0, row0 = NCT == 2 ? (wid % (HR_NW / 2)) * R : wid * R;
  // ---- persistent workgroup: its cout tiles' weights stay in registers (NKS fragments per lane, straight from the fragment-ordered
  // pack).  Workgroups whose blockIdx / 8 agree modulo the number of cout-tile groups share a group; a workgroup's virtual block ids
  // vb = j0, j0 + G', ... keep vb % 8 == blockIdx % 8, so xcd_remap still hands every XCD one contiguous run of the (row-major) tile
  // order and neighbouring tiles' halos meet in its L2.
  const int groups = (p.ntile_c + NCT - 1) / NCT;
  const int ct = ((blockIdx.x >> 3) % groups) * NCT + ctl;
  const int ct_ld = ct < p.ntile_c ? ct : p.ntile_c - 1;
  const unsigned gsub = gridDim.x / groups;                             // workgroups per group (launcher: a multiple of 8)
  const unsigned j0 = ((blockIdx.x >> 3) / groups) * 8 + (blockIdx.x & 7);
  h8 wf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) wf[ks] = *reinterpret_cast<const h8*>(p.wt + ((size_t)(ct_ld * NKS + ks) * 64 + lane) * 8);
  if (tid < 4) reinterpret_cast<float*>(smem + ZERO_OFF)[tid] = 0.f;
  const unsigned per_img = p.tiles_x * p.tiles_y, total = per_img * (unsigned)p.N;
  const char* lbase = smem + pix * PIXB;               // per-lane base: every fragment address below is lbase + buffer + a compile-time constant

  // ---- DMA roles, the same for every tile: piece i of a wave is wave instruction wid + HR_NW i, its lane fills 16-byte chunk
  // g = (wid + HR_NW i) * 64 + lane of the halo tile = (halo row ty, halo column tx, slot c)
  int voff[NFI], iy0[NFI], ix0[NFI];
#pragma unroll
  for (int i = 0; i < NFI; ++i) {
    const int g = (wid + HR_NW * i) * 64 + lane, q = g / SLOTS, c = g - q * SLOTS;
    const int ty = q / HR_HW, tx = q - ty * HR_HW;
    voff[i] = 2 * (int)(ty * p.i_sy + tx * p.i_sx + c * 8);
    iy0[i] = (ty < HR_HH && c < CH8) ? ty : 0x40000000;
    ix0[i] = tx;
  }
  auto tile_at = [&](unsigned vb, int& n, int& y0, int& x0) {
    if (vb >= total) vb -= ((vb - total) / gsub + 1) * gsub;      // past the last tile: refetch the last one (uniform instruction counts)
    const unsigned lt = xcd_remap(vb, total);
    n = lt / per_img;
    const unsigned r_ = lt - n * per_img;
    y0 = (r_ / p.tiles_x) * HR_TH; x0 = (r_ % p.tiles_x) * HR_TW;
  };
  auto tile_rs = [&](int n, int y0, int x0) {
    return hr_make_rs(p.in + n * p.i_sn + (long)(y0 - HALO) * p.i_sy + (long)(x0 - HALO) * p.i_sx);
  };
  auto issue_piece = [&](__amdgpu_buffer_rsrc_t rs, int y0, int x0, int i, int buf) __attribute__((always_inline)) {
    const bool ok = (unsigned)(iy0[i] + y0 - HALO) < (unsigned)p.H && (unsigned)(ix0[i] + x0 - HALO) < (unsigned)p.W;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + buf * TILE_BYTES + (wid + HR_NW * i) * 1024), 16,
                                             ok ? voff[i] : -1, 0, 0, 0);
  };

  if (j0 >= total) return;
  // (MASK is a template parameter and the loads are unconditional -- dead lanes re-read an in-range element: a branch around them
  // made hipcc drain vmcnt to 0 at the join, i.e. wait for every DMA piece in flight at the top of each tile)
  h8 mk[MASK ? R : 1][2], mkn[MASK ? R : 1][2];
  auto load_masks = [&](h8 (&m)[MASK ? R : 1][2], int n, int y0, int x0) __attribute__((always_inline)) {
#pragma unroll
    for (int rr = 0; rr < (MASK ? R : 1); ++rr)
#pragma unroll
      for (int pair = 0; pair < 2; ++pair) {
        int oy = y0 + row0 + rr, ox = x0 + pix, co = ct * 32 + 16 * pair + 8 * hi;
        oy = oy < p.H ? oy : p.H - 1; ox = ox < p.W ? ox : p.W - 1; co = co < p.coutp ? co : 0;
        m[rr][pair] = *reinterpret_cast<const h8*>(p.mask + n * p.m_sn + oy * p.m_sy + ox * p.m_sx + co);
      }
  };
  if (MASK) {
    int n, y0, x0;
    tile_at(j0, n, y0, x0);
    load_masks(mkn, n, y0, x0);
  }
#pragma unroll
  for (int k = 0; k < NBUF - 1; ++k) {
    int n, y0, x0;
    tile_at(j0 + k * gsub, n, y0, x0);
    const __amdgpu_buffer_rsrc_t rs = tile_rs(n, y0, x0);
#pragma unroll
    for (int i = 0; i < NFI; ++i) issue_piece(rs, y0, x0, i, k);
  }
  int buf = 0;
  // STAT: per-sample sums of the outputs (the global average pool behind fe_cat.2).  A wave keeps its 16 per-lane sums in registers
  // across ALL its tiles of a sample and adds them -- folded over the 32 pixels -- to its own partial row
  // stat_part[(n * gridDim.x + blockIdx.x) * HR_NW + wid][coutp] when the sample changes: one writer per row, a fixed tile order
  // (the launcher zeroes the rows and folds them per sample with csbsr_sum_partials_batched), so two runs are bit-identical.
  float gsum[STAT ? 16 : 1];
#pragma unroll
  for (int e = 0; e < (STAT ? 16 : 1); ++e) gsum[e] = 0.f;
  int cur_n = -1;
  auto flush_stat = [&](int n_) {
#pragma unroll
    for (int e = 0; e < (STAT ? 16 : 1); ++e) {
      float a = gsum[e];
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) a += __shfl_xor(a, o, 64);
      gsum[e] = a;
    }
    if (pix == 0) {
      float* row = p.stat_part + (((size_t)n_ * gridDim.x + blockIdx.x) * HR_NW + wid) * p.coutp;
#pragma unroll
      for (int pair = 0; pair < 2; ++pair)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int co = ct * 32 + 16 * pair + 8 * hi + e;
          if (STAT && co < p.coutp) row[co] += gsum[8 * pair + e];
        }
    }
#pragma unroll
    for (int e = 0; e < (STAT ? 16 : 1); ++e) gsum[e] = 0.f;
  };
  for (unsigned vb = j0; vb < total; vb += gsub) {
    int n, y0, x0, nn, y0n, x0n;
    tile_at(vb, n, y0, x0);
    tile_at(vb + (NBUF - 1) * gsub, nn, y0n, x0n);
    const __amdgpu_buffer_rsrc_t rsn = tile_rs(nn, y0n, x0n);
    const int bfill = buf == 0 ? NBUF - 1 : buf - 1;
    if (MASK) {
#pragma unroll
      for (int rr = 0; rr < R; ++rr) { mk[rr][0] = mkn[rr][0]; mk[rr][1] = mkn[rr][1]; }
    }
    // the activation-derivative masks are requested ONE TILE AHEAD (those of the next tile here, before any of this tile's DMA pieces):
    // the wait in front of their first use then only drains requests older than they are -- the previous tile's pieces, due anyway --
    // and the pieces issued since stay in flight.  (Requested at the top of their own tile they forced the tile one ahead to land
    // first: half the prefetch depth, 0.87 instead of 0.45 ms on the masked 1x1 layers.)
    if (MASK) {
      int n1, y01, x01;
      tile_at(vb + gsub, n1, y01, x01);
      load_masks(mkn, n1, y01, x01);
    }
    // this tile has landed everywhere (the NBUF - 2 tiles after it may still be in flight) and every wave is done with the buffer of
    // the previous tile: the tile NBUF - 1 ahead goes there, piece by piece, below
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NFI) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    if constexpr (STAT) {
      if (n != cur_n) {        // workgroup-uniform
        if (cur_n >= 0) flush_stat(cur_n);
        cur_n = n;
      }
    }
    const char* tb = lbase + buf * TILE_BYTES;
    // four rows at a time: four independent accumulator chains, each row's next fragment requested right after the MFMA that consumed
    // the current one (four MFMAs = 128 cycles of cover for the LDS latency -- with ONE wave per SIMD nothing else hides it; left to
    // the compiler the read sat directly in front of its MFMA and the K loop ran at LDS-latency pace)
    constexpr int RG = R < 4 ? R : 4;
#pragma unroll
    for (int rp = 0; rp < R / RG; ++rp) {
      f16v acc[RG];
#pragma unroll
      for (int q = 0; q < RG; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      auto frag = [&](int q, int ks) __attribute__((always_inline)) {
        // chunk kc = 2 ks + hi of the flattened (tap, chunk) axis: compile-time for each half-wave
        const int kc0 = 2 * ks, kc1 = 2 * ks + 1;
        const int t0 = kc0 / CH8, c0 = kc0 % CH8, t1 = kc1 / CH8, c1 = kc1 % CH8;
        const int a0 = ((t0 / 3) * HR_HW + (t0 % 3)) * PIXB + (c0 << 4);
        const int a1 = ((t1 / 3) * HR_HW + (t1 % 3)) * PIXB + (c1 << 4);
        const char* rbase = tb + (row0 + RG * rp + q) * HR_HW * PIXB;
        // odd chunk count: the last step's upper half reads the zero chunk
        const char* addr = (kc1 >= NCHUNK) ? (hi ? smem + ZERO_OFF : rbase + a0) : rbase + (hi ? a1 : a0);
        return *reinterpret_cast<const h8*>(addr);
      };
      h8 bfr[RG];
#pragma unroll
      for (int q = 0; q < RG; ++q) bfr[q] = frag(q, 0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int q = 0; q < RG; ++q) {
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], bfr[q], acc[q], 0, 0, 0);
          if (ks + 1 < NKS) bfr[q] = frag(q, ks + 1);
          const int idx = (rp * NKS + ks) * RG + q;      // (the pieces are spread over the tile's MFMAs in this order)
          if (PP == 1) {
            if (idx % SP == 0 && idx / SP < NFI) issue_piece(rsn, y0n, x0n, idx / SP, bfill);
          } else {
#pragma unroll
            for (int u = 0; u < PP; ++u)
              if (idx * PP + u < NFI) issue_piece(rsn, y0n, x0n, idx * PP + u, bfill);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- epilogue: acc[4q + j] = cout 8q + 4 hi + j of pixel `pix`; swap pairs -> 8 consecutive couts per lane
#pragma unroll
      for (int q = 0; q < RG; ++q) {
        const int rr = RG * rp + q, row = row0 + rr;
        const int oy = y0 + row, ox = x0 + pix;
        const bool live = oy < p.H && ox < p.W;
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned a = __float_as_uint(acc[q][8 * pair + j]), b = __float_as_uint(acc[q][8 * pair + 4 + j]);
            auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
            v[j] = __uint_as_float(r[0]);
            v[4 + j] = __uint_as_float(r[1]);
          }
          // branch-free: none / ReLU / LeakyReLU are max(t, t * aslope) with aslope = 1 / 0 / slope; output channels past cout need no
          // zeroing (their packed weight rows are zero); dead lanes (past the image edge, past coutp) only skip the store and the sums.
          // (With the activation chosen by uniform branches per element this epilogue was ~400 scalar branches per tile and wave --
          // 2/3 of the tile time once a single wave per SIMD had to run it.)
          const int co = ct * 32 + 16 * pair + 8 * hi;
          const bool st_ok = live && co < p.coutp;
          if constexpr (CB) {      // the folded constant segment: one of 25 values per (sample, cout), by the pixel's two-ring class
            if (st_ok) {
              const int ty = oy < 2 ? oy : (oy >= p.H - 2 ? oy - p.H + 5 : 2), tx = ox < 2 ? ox : (ox >= p.W - 2 ? ox - p.W + 5 : 2);
              const float* cbp = p.cbias + ((size_t)n * 25 + ty * 5 + tx) * p.coutp + co;
              const f4 c0 = *reinterpret_cast<const f4*>(cbp), c1 = *reinterpret_cast<const f4*>(cbp + 4);
              v[0] += c0[0]; v[1] += c0[1]; v[2] += c0[2]; v[3] += c0[3]; v[4] += c1[0]; v[5] += c1[1]; v[6] += c1[2]; v[7] += c1[3];
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * aslope);
          if constexpr (STAT) {
            const float lf = st_ok ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) gsum[8 * pair + e] += v[e] * lf;
          }
          if (MASK) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= ((float)mk[MASK ? rr : 0][pair][e] > 0.f ? 1.f : p.mask_slope);
          }
          if (has_out && st_ok) {
            h8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
            *reinterpret_cast<h8*>(p.out16 + n * p.o_sn + oy * p.o_sy + ox * p.o_sx + co) = hv;
          }
        }
      }
    }
    buf = buf + 1 == NBUF ? 0 : buf + 1;
  }      // tiles
  if constexpr (STAT) {
    if (cur_n >= 0) flush_stat(cur_n);
  }
#endif
}

// ---- weights in fragment order:  dst[ct][ks][lane][e] = W(cout = 32 ct + lane%32, chunk kc = 2 ks + lane/32, channel 8 (kc % CH8) + e)
// with tap = kc / CH8 -> (ky, kx).  kind 0: forward conv, W is OIHW [cout][cin];  kind 1: dgrad of a stride-1 conv: rows are the conv's
// INPUT channels, contracted channels its output channels, taps flipped (W[contracted][row][2-ky][2-kx]).
struct PackHrK { const float* w; half_t* dst; int kind, D0, D1, ch8, nks, rows_real, c_real, ntile_c, row_off, k_off, ks; };
__global__ void pack_weights_hr_kernel(const PackHrK p) {
  const long total = (long)p.ntile_c * p.nks * 64 * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    const int lane = (int)((i >> 3) & 63);
    const long t = i >> 9;
    const int ks = (int)(t % p.nks), ct = (int)(t / p.nks);
    const int row = ct * 32 + (lane & 31);
    const int kc = 2 * ks + (lane >> 5);
    const int tap = kc / p.ch8, c = (kc % p.ch8) * 8 + e;
    float v = 0.f;
    if (row < p.rows_real && tap < p.ks * p.ks && c < p.c_real) {
      const int ky = tap / p.ks, kx = tap % p.ks;
      const int rr = p.row_off + row, cc = p.k_off + c;
      if (p.kind == 0) v = p.w[(((long)rr * p.D1 + cc) * p.ks + ky) * p.ks + kx];
      else v = p.w[(((long)cc * p.D1 + rr) * p.ks + (p.ks - 1 - ky)) * p.ks + (p.ks - 1 - kx)];
    }
    p.dst[i] = (half_t)v;
  }
}

static bool hr_geometry(int ksize, int c_real, int rows_real, int& ch8, int& nks, int& ntile_c) {
  if (ksize != 1 && ksize != 3) return false;
  const int cp = round_up(c_real, 8);
  if (cp == 32) ch8 = 4;
  else if (cp == 56) ch8 = 7;
  else return false;
  nks = (ksize * ksize * ch8 + 1) / 2;
  ntile_c = (round_up(rows_real, 8) + 31) / 32;
  return rows_real >= 1 && (ntile_c <= 2 || ntile_c == 4);      // 4: two groups of workgroups, two 32-cout tiles each (128 couts)
}

extern "C" int64_t csbsr_packed_weight_elems_hr(int32_t ksize, int32_t c_real, int32_t rows_real) {
  int ch8, nks, nt;
  if (!hr_geometry(ksize, c_real, rows_real, ch8, nks, nt)) return 0;
  return (int64_t)nt * nks * 64 * 8;
}

extern "C" int csbsr_pack_weights_hr(const float* w, void* dst, int32_t kind, int32_t ksize, int32_t D0, int32_t D1, int32_t c_real, int32_t rows_real,
                                     int32_t row_off, int32_t k_off, csbsr_stream_t s) {
  CSBSR_CHECK(w && dst && (kind == 0 || kind == 1), "pack_hr: bad args");
  PackHrK p;
  CSBSR_CHECK(hr_geometry(ksize, c_real, rows_real, p.ch8, p.nks, p.ntile_c), "pack_hr: 1x1 or 3x3, channels must pad to 32 or 56, rows to <= 64 or to 128");
  p.ks = ksize;
  const int kdim = kind == 0 ? D1 : D0, rdim = kind == 0 ? D0 : D1;
  CSBSR_CHECK(k_off >= 0 && k_off + c_real <= kdim && row_off >= 0 && row_off + rows_real <= rdim, "pack_hr: range out of bounds");
  p.w = w; p.dst = reinterpret_cast<half_t*>(dst); p.kind = kind; p.D0 = D0; p.D1 = D1;
  p.rows_real = rows_real; p.c_real = c_real; p.row_off = row_off; p.k_off = k_off;
  const long total = (long)p.ntile_c * p.nks * 512;
  hipLaunchKernelGGL(pack_weights_hr_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(s), p);
  CSBSR_LAUNCH_CHECK("csbsr_pack_weights_hr");
  return 0;
}

// Which launches take this kernel: 3x3 / pad 1 or 1x1 / pad 0, stride 1, dilation 1, one plain-fp16 input segment of 32 or 56 (padded) channels,
// <= 64 (or exactly 128: two workgroup groups, each re-reading the halo) output channels, ReLU / LeakyReLU / no activation, no bias / residual / accumulate / fp32 side output / BatchNorm sums; large
// maps only (the tile grid must fill the chip).
extern "C" int32_t csbsr_conv_hr_eligible(const csbsr_conv_desc_t* d) {
  if (!d || d->transposed || d->KH != d->KW || d->stride != 1 || d->dil != 1) return 0;
  if (!((d->KH == 3 && d->pad == 1) || (d->KH == 1 && d->pad == 0))) return 0;
  if (d->in[1].c != 0 || d->in[0].sx == 0 || (d->in[0].c != 32 && d->in[0].c != 56)) return 0;
  if ((d->coutp > 64 && d->coutp != 128) || d->OH != d->H || d->OW != d->W) return 0;
  if (d->coutp == 128 && (d->stat_mode != CSBSR_STAT_NONE || d->mask)) return 0;      // 128 couts: the plain variant only
  if (d->bias || d->res_mode != CSBSR_RES_NONE || d->accumulate || d->out32 || d->o_lo || d->mask_prelu) return 0;
  // a position-class bias: the two-ring table on the plain 1x1 variant with one cout tile (fe_cat.0 with its folded kernel branch)
  if (d->cbias && !(d->cbias_mode == 1 && d->KH == 1 && d->coutp <= 32 && d->stat_mode == CSBSR_STAT_NONE && !d->mask && d->H >= 5 && d->W >= 5)) return 0;
  if (d->act != CSBSR_ACT_NONE && d->act != CSBSR_ACT_RELU && d->act != CSBSR_ACT_LRELU) return 0;
  if (d->stat_mode == CSBSR_STAT_BN || d->out_scale != 1.0f) return 0;
  if (d->mask && d->stat_mode != CSBSR_STAT_NONE) return 0;
  if ((long)d->N * d->H * d->W < 256L * 1024) return 0;
  return 1;
}

template <int CH8, int TAPS, int NCT>
static int launch_hr(const ConvHrK& k, hipStream_t st) {
  constexpr int SM_BYTES = hr_smem(CH8, TAPS);
  static_assert(SM_BYTES <= 160 * 1024, "LDS budget");
  static LdsAttrOnce a0, a1, a2;
  if (int e = csbsr_lds_attr(a0, reinterpret_cast<const void*>(conv_hr_kernel<CH8, true, TAPS, NCT, false>), SM_BYTES, "conv_hr")) return e;
  if (int e = csbsr_lds_attr(a1, reinterpret_cast<const void*>(conv_hr_kernel<CH8, false, TAPS, NCT, false>), SM_BYTES, "conv_hr")) return e;
  if (int e = csbsr_lds_attr(a2, reinterpret_cast<const void*>(conv_hr_kernel<CH8, false, TAPS, NCT, true>), SM_BYTES, "conv_hr")) return e;
  // persistent: one workgroup per CU (the ring of halo tiles takes the LDS), a multiple of 8 x (cout-tile groups); never more than there is work
  const unsigned total = k.tiles_x * k.tiles_y * k.N;
  const unsigned groups = (unsigned)((k.ntile_c + NCT - 1) / NCT), unit = 8u * groups;
  unsigned g = (unsigned)csbsr_cu_budget(st);      // the stream's CU partition (csrc/streams.hip), else the whole device
  if (g > total * groups) g = total * groups;
  g = (g + unit - 1) / unit * unit;
  dim3 grid(g);
  if (k.stat) {
    const long rows_n = (long)g * hr_nw(CH8, TAPS), elems = (long)k.N * rows_n * k.coutp;
    ConvHrK ks = k;
    ks.stat_part = csbsr_red_scratch(elems);
    CSBSR_NEED_SCRATCH(ks.stat_part, "conv_hr (per-sample sums)");
    if (hipMemsetAsync(ks.stat_part, 0, (size_t)elems * 4, st) != hipSuccess) { csbsr_set_error("conv_hr: memset of the partial rows failed"); return 2; }
    hipLaunchKernelGGL((conv_hr_kernel<CH8, true, TAPS, NCT, false>), grid, dim3(64 * hr_nw(CH8, TAPS)), SM_BYTES, st, ks);
    if (int e = csbsr_sum_partials_batched(ks.stat_part, (int)rows_n, k.coutp, k.coutp, k.stat, k.N, k.coutp, st)) return e;
  } else if (k.cbias) {
    if constexpr (TAPS == 1 && NCT == 1) {
      static LdsAttrOnce a3;
      if (int e = csbsr_lds_attr(a3, reinterpret_cast<const void*>(conv_hr_kernel<CH8, false, TAPS, NCT, false, true>), SM_BYTES, "conv_hr")) return e;
      hipLaunchKernelGGL((conv_hr_kernel<CH8, false, TAPS, NCT, false, true>), grid, dim3(64 * hr_nw(CH8, TAPS)), SM_BYTES, st, k);
    } else {
      csbsr_set_error("conv_hr: class bias on a variant that is not built");
      return 1;
    }
  } else if (k.mask) hipLaunchKernelGGL((conv_hr_kernel<CH8, false, TAPS, NCT, true>), grid, dim3(64 * hr_nw(CH8, TAPS)), SM_BYTES, st, k);
  else hipLaunchKernelGGL((conv_hr_kernel<CH8, false, TAPS, NCT, false>), grid, dim3(64 * hr_nw(CH8, TAPS)), SM_BYTES, st, k);
  CSBSR_LAUNCH_CHECK("csbsr_conv_hr_forward");
  return 0;
}

extern "C" int csbsr_conv_hr_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s) {
  CSBSR_CHECK(csbsr_conv_hr_eligible(d), "conv_hr: launch not eligible (see csbsr_conv_hr_eligible)");
  CSBSR_CHECK(d->in[0].ptr && d->wt && (d->out16 || d->stat), "conv_hr: null pointer");
  ConvHrK k;
  k.in = reinterpret_cast<const half_t*>(d->in[0].ptr); k.i_sn = d->in[0].sn; k.i_sy = d->in[0].sy; k.i_sx = d->in[0].sx;
  k.N = d->N; k.H = d->H; k.W = d->W;
  k.wt = reinterpret_cast<const half_t*>(d->wt);
  k.cout = d->cout; k.coutp = d->coutp; k.ntile_c = (d->coutp + 31) / 32;
  k.out16 = reinterpret_cast<half_t*>(d->out16); k.o_sn = d->o_sn; k.o_sy = d->o_sy; k.o_sx = d->o_sx;
  k.act = d->act; k.slope = d->act_slope;
  k.mask = reinterpret_cast<const half_t*>(d->mask); k.m_sn = d->m_sn; k.m_sy = d->m_sy; k.m_sx = d->m_sx; k.mask_slope = d->mask_slope;
  k.stat = d->stat_mode == CSBSR_STAT_SAMPLE_SUM ? d->stat : nullptr; k.stat_part = nullptr;
  k.cbias = d->cbias;
  k.tiles_x = (unsigned)((d->W + HR_TW - 1) / HR_TW); k.tiles_y = (unsigned)((d->H + HR_TH - 1) / HR_TH);
  CSBSR_CHECK(d->in[0].sy < (1l << 31) / 64, "conv_hr: row stride too large for 32-bit piece offsets");
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  const bool two = k.ntile_c >= 2;
  g_last_conv_kernel = CONVK_HR | ((d->in[0].c == 32 ? 0 : 1) | (d->KH == 3 ? 0 : 2) | (two ? 4 : 0) | (k.mask ? 8 : 0) | (k.stat ? 16 : 0) | (k.cbias ? 32 : 0)) << 8;      // bits 8..: <CH8 7, 1x1, two cout tiles, mask, stat, class bias>
  if (d->KH == 3) {
    if (d->in[0].c == 32) return two ? launch_hr<4, 9, 2>(k, st) : launch_hr<4, 9, 1>(k, st);
    return two ? launch_hr<7, 9, 2>(k, st) : launch_hr<7, 9, 1>(k, st);
  }
  if (d->in[0].c == 32) return two ? launch_hr<4, 1, 2>(k, st) : launch_hr<4, 1, 1>(k, st);
  return two ? launch_hr<7, 1, 2>(k, st) : launch_hr<7, 1, 1>(k, st);
}
