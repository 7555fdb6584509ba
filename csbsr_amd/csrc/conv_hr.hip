// Direct 3x3 convolution for the 32 / 49-channel layers that run at FULL (HR) resolution -- the kernel predictor's fe_SR.2-4,
// fe_kernel.1, fe_cat.1-2 and their dgrads (kbpn.py:521-578): 140-175 FLOP per byte of input + output, i.e. HBM-bound layers that the
// implicit-GEMM kernels ran ~4x off their byte roofline (each of the 9 taps re-gathers the pixel operand through the L1/LDS path and
// every fragment pair is read from LDS: a 32-cout K slice is 4 MFMAs per wave against ~60 address instructions).
//
//  * one workgroup (4 waves) = 16 x 32 output tiles; a tile's (16+2) x (32+2) input halo goes HBM -> LDS ONCE with
//    global_load_lds_dwordx4 (16-byte channel chunks, zero page for out-of-image pixels) -- 1.2x the tile's own bytes instead of 9x;
//  * the WEIGHTS never touch LDS: they are packed in MFMA-fragment order (csbsr_pack_weights_hr) and each lane keeps its A fragments
//    of all K steps of one 32-cout tile in registers (18 steps x 4 VGPRs for 32 channels, 32 x 4 for 49 -> 56), so the K loop is one
//    ds_read_b128 of the pixel operand per MFMA and nothing else;
//  * K is flattened over (tap, 8-channel chunk): an MFMA K step = two chunks, one per half-wave, each half-wave addressing its own
//    (tap, chunk) -- 56-channel maps need no padding to 64 (63 chunks -> 32 steps);
//  * bank conflicts: a 64-byte pixel pitch (32 channels) would put pixels p and p + 4 on the same 16-byte slots, so a 32-channel
//    pixel is laid out on FIVE slots (80 bytes, the fifth fetched from the zero page): an odd pitch in 16-byte slots is conflict-free
//    for the 16 consecutive-pixel lanes of a ds_read_b128 group, as the 112-byte pitch of 56 channels already is -- and with no XOR
//    swizzle every fragment address is one per-lane base register plus a compile-time offset (no address arithmetic in the K loop);
//  * epilogue in registers: v_permlane32_swap turns the MFMA layout into 8 consecutive couts per lane, activation, optional fused
//    activation-derivative mask (dgrads), optional global-average-pool sums (fe_cat.2), 16-byte stores.
//
// Replaces F.conv2d at kbpn.py:536-547 (via ConvBlock) and its autograd dgrad for the eligible layers.
#include "common.h"
#include "conv_common.h"
#include <cstdlib>

#define HR_TH 16
#define HR_TW 32

struct ConvHrK {
  const half_t* in; long i_sn, i_sy, i_sx;
  int N, H, W;
  const half_t* wt;                 // [cout tiles][NKS][64 lanes][8] fragment order
  int cout, coutp, ntile_c;         // real / padded couts, 32-cout tiles
  half_t* out16; long o_sn, o_sy, o_sx;
  int act; float slope;
  const half_t* mask; long m_sn, m_sy, m_sx; float mask_slope;
  float* stat;                      // optional [N][coutp] per-sample channel sums of act(conv) (global average pool)
  unsigned tiles_x, tiles_y;
  int dbg;                          // ablation bits (CSBSR_HR_DBG): 1 no stores, 2 no K loop, 4 no tile DMA
};

// TAPS = 9 (3x3, one-pixel halo) or 1 (the 1x1 layers of the same chains -- fe_SR.1, fe_cat.0 and their dgrads: no halo, two or four
// MFMA K steps per 32 pixels, a pure HBM stream)
template <int CH8, bool STAT, int TAPS>
__global__ __launch_bounds__(256, (CH8 == 4 ? 3 : 2)) void conv_hr_kernel(const ConvHrK p, const half_t* __restrict__ zero_page) {
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int HR_HW = HR_TW + 2 * HALO, HR_NPIX = (HR_TH + 2 * HALO) * HR_HW;      // 18 x 34 = 612 halo pixels (3x3)
  constexpr int NCHUNK = TAPS * CH8;                    // K in 8-channel chunks
  constexpr int NKS = (NCHUNK + 1) / 2;                 // MFMA K steps (16 channels = two chunks)
  constexpr int SLOTS = (CH8 % 2) ? CH8 : CH8 + 1;      // 16-byte slots per pixel in LDS: odd, so consecutive pixels walk all banks
  constexpr int PIXB = SLOTS * 16;                      // bytes per pixel in LDS
  constexpr int NG = HR_NPIX * SLOTS;                   // 16-byte chunks of the halo tile (incl. the pad slots)
  constexpr int NINST = (NG + 63) / 64;                 // wave instructions to fill it
  constexpr int TILE_BYTES = NINST * 1024;              // (rounded up: overhang lanes fetch the zero page)
  constexpr int ZERO_OFF = TILE_BYTES;                  // one zero chunk for the padded half K step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sStat = reinterpret_cast<float*>(smem + ZERO_OFF + 16);      // [32] per-cout sums of the current tile

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pix = lane & 31, hi = lane >> 5;
  const float slope = p.slope;
  const half_t* zp = zero_page + (lane & 7) * 8;
  // ---- persistent workgroup: ONE 32-cout tile's weights stay in registers (NKS fragments per lane, straight from the
  // fragment-ordered pack) while the workgroup walks its share of the pixel tiles -- reloading them per tile cost as many bytes
  // through the CU's load path as the tile itself.  Workgroups whose blockIdx / 8 agree modulo ntile_c share a cout tile; a
  // workgroup's virtual block ids vb = j0, j0 + G', ... keep vb % 8 == blockIdx % 8, so xcd_remap still hands every XCD one
  // contiguous run of the (row-major) tile order and neighbouring tiles' halos meet in its L2.
  const int ct = (blockIdx.x >> 3) % p.ntile_c;
  const unsigned gsub = gridDim.x / p.ntile_c;                          // workgroups per cout tile (launcher: a multiple of 8)
  const unsigned j0 = ((blockIdx.x >> 3) / p.ntile_c) * 8 + (blockIdx.x & 7);
  h8 wf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) wf[ks] = *reinterpret_cast<const h8*>(p.wt + ((size_t)(ct * NKS + ks) * 64 + lane) * 8);
  if (tid < 4) reinterpret_cast<float*>(smem + ZERO_OFF)[tid] = 0.f;
  const unsigned per_img = p.tiles_x * p.tiles_y, total = per_img * (unsigned)p.N;
  const char* lbase = smem + pix * PIXB;               // per-lane base: every fragment address below is lbase + a compile-time constant
  // DMA roles are the same for every tile: chunk g = (wid + 4 i) * 64 + lane of the halo tile = (halo row ty, halo column tx, slot
  // c).  The first instruction's role is computed once; consecutive instructions of a lane are 256 chunks apart, i.e. a fixed
  // (DQ pixels, DC slots) step with carries -- a handful of compares instead of two integer divisions per instruction.
  constexpr int NFI = (NINST + 3) / 4;
  constexpr int DQ = 256 / SLOTS, DC = 256 % SLOTS;
  int f_ty0, f_tx0, f_c0;
  {
    const int g = wid * 64 + lane;
    const int q = g / SLOTS;
    f_c0 = g - q * SLOTS;
    f_ty0 = q / HR_HW;
    f_tx0 = q - f_ty0 * HR_HW;
  }
  const int isy = (int)p.i_sy, isx = (int)p.i_sx;        // (within one image: < 2^31 elements, launcher checks)

  for (unsigned vb = j0; vb < total; vb += gsub) {
    const unsigned lt = xcd_remap(vb, total);
    const int n = lt / per_img;
    const unsigned r_ = lt - n * per_img;
    const int y0 = (r_ / p.tiles_x) * HR_TH, x0 = (r_ % p.tiles_x) * HR_TW;
    if (vb != j0) __syncthreads();                     // every wave is done reading the previous tile (and its sums are flushed)
    // ---- halo tile -> LDS
    const half_t* tbase = p.in + n * p.i_sn + (long)(y0 - HALO) * p.i_sy + (long)(x0 - HALO) * p.i_sx;      // wave-uniform
    int ty = f_ty0, tx = f_tx0, c = f_c0;
#pragma unroll 2      // (fully unrolled the scheduler computes every 64-bit source address up front: 2 x NFI registers)
    for (int i = 0; i < NFI; ++i) {
      const int inst = wid + 4 * i;
      if (inst < NINST) {
        const int iy = y0 - HALO + ty, ix = x0 - HALO + tx;
        const bool ok = ty < HR_TH + 2 * HALO && c < CH8 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && !(p.dbg & 4);
        const half_t* src = ok ? tbase + (ty * isy + tx * isx + c * 8) : zp;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + inst * 1024), 16, 0, 0);
      }
      c += DC; tx += DQ;
      if (c >= SLOTS) { c -= SLOTS; ++tx; }
      if (tx >= HR_HW) { tx -= HR_HW; ++ty; }
      if (tx >= HR_HW) { tx -= HR_HW; ++ty; }
    }
    if (STAT && tid < 32) sStat[tid] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float gsum[STAT ? 16 : 1];
#pragma unroll
    for (int e = 0; e < (STAT ? 16 : 1); ++e) gsum[e] = 0.f;
#pragma unroll 1      // (rows unrolled: the scheduler hoists every row's NKS fragment reads and spills the weights)
    for (int rr = 0; rr < HR_TH / 4; ++rr) {
      const int row = (HR_TH / 4) * wid + rr;            // output row of the tile
      const char* rbase = lbase + row * HR_HW * PIXB;
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      if (!(p.dbg & 2))
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        // chunk kc = 2 ks + hi of the flattened (tap, chunk) axis: compile-time for each half-wave
        const int kc0 = 2 * ks, kc1 = 2 * ks + 1;
        const int t0 = kc0 / CH8, c0 = kc0 % CH8, t1 = kc1 / CH8, c1 = kc1 % CH8;
        const int a0 = ((t0 / 3) * HR_HW + (t0 % 3)) * PIXB + (c0 << 4);
        const int a1 = ((t1 / 3) * HR_HW + (t1 % 3)) * PIXB + (c1 << 4);
        // odd chunk count: the last step's upper half reads the zero chunk
        const char* addr = (kc1 >= NCHUNK) ? (hi ? smem + ZERO_OFF : rbase + a0) : rbase + (hi ? a1 : a0);
        const h8 bf = *reinterpret_cast<const h8*>(addr);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], bf, acc, 0, 0, 0);
      }
      // ---- epilogue: acc[4q + j] = cout 8q + 4 hi + j of pixel `pix`; swap pairs -> 8 consecutive couts per lane
      const int oy = y0 + row, ox = x0 + pix;
      const bool live = oy < p.H && ox < p.W;
#pragma unroll
      for (int pair = 0; pair < 2; ++pair) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned a = __float_as_uint(acc[8 * pair + j]), b = __float_as_uint(acc[8 * pair + 4 + j]);
          auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
          v[j] = __uint_as_float(r[0]);
          v[4 + j] = __uint_as_float(r[1]);
        }
        const int co = ct * 32 + 16 * pair + 8 * hi;
        if (co >= p.coutp || !live) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = v[e];
          if (p.act == CSBSR_ACT_RELU) t = fmaxf(t, 0.f);
          else if (p.act == CSBSR_ACT_LRELU) t = fmaxf(t, t * slope);
          v[e] = (co + e < p.cout) ? t : 0.f;
        }
        if constexpr (STAT) {
#pragma unroll
          for (int e = 0; e < 8; ++e) gsum[8 * pair + e] += v[e];
        }
        if (p.mask) {
          const h8 mk = *reinterpret_cast<const h8*>(p.mask + n * p.m_sn + oy * p.m_sy + ox * p.m_sx + co);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= ((float)mk[e] > 0.f ? 1.f : p.mask_slope);
        }
        if (p.out16 && (!(p.dbg & 1) || v[0] == 12345.678f)) {
          h8 hv;
#pragma unroll
          for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
          *reinterpret_cast<h8*>(p.out16 + n * p.o_sn + oy * p.o_sy + ox * p.o_sx + co) = hv;
        }
      }
    }
    if constexpr (STAT) {  // lanes with equal (hi, pair) hold the same couts for different pixels: fold the 32 pixels, then LDS bins
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float a = gsum[e];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) a += __shfl_xor(a, o, 64);
        gsum[e] = a;
      }
      if (pix == 0) {
#pragma unroll
        for (int pair = 0; pair < 2; ++pair)
#pragma unroll
          for (int e = 0; e < 8; ++e) atomicAdd(&sStat[16 * pair + 8 * hi + e], gsum[8 * pair + e]);
      }
      __syncthreads();
      if (tid < 32 && ct * 32 + tid < p.coutp) atomicAdd(p.stat + (size_t)n * p.coutp + ct * 32 + tid, sStat[tid]);
    }
  }      // tiles
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
  return rows_real >= 1 && ntile_c <= 2;
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
  CSBSR_CHECK(hr_geometry(ksize, c_real, rows_real, p.ch8, p.nks, p.ntile_c), "pack_hr: 1x1 or 3x3, channels must pad to 32 or 56, rows to <= 64");
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
// <= 64 output channels, ReLU / LeakyReLU / no activation, no bias / residual / accumulate / fp32 side output / BatchNorm sums; large
// maps only (the tile grid must fill the chip).
extern "C" int32_t csbsr_conv_hr_eligible(const csbsr_conv_desc_t* d) {
  if (!d || d->transposed || d->KH != d->KW || d->stride != 1 || d->dil != 1) return 0;
  if (!((d->KH == 3 && d->pad == 1) || (d->KH == 1 && d->pad == 0))) return 0;
  if (d->in[1].c != 0 || d->in[0].sx == 0 || (d->in[0].c != 32 && d->in[0].c != 56)) return 0;
  if (d->coutp > 64 || d->OH != d->H || d->OW != d->W) return 0;
  if (d->bias || d->cbias || d->res_mode != CSBSR_RES_NONE || d->accumulate || d->out32 || d->o_lo) return 0;
  if (d->act != CSBSR_ACT_NONE && d->act != CSBSR_ACT_RELU && d->act != CSBSR_ACT_LRELU) return 0;
  if (d->stat_mode == CSBSR_STAT_BN || d->out_scale != 1.0f) return 0;
  if ((long)d->N * d->H * d->W < 256L * 1024) return 0;
  return 1;
}

static half_t* g_hr_zero_page[CSBSR_MAX_DEVICES] = {};

template <int CH8, int TAPS>
static int launch_hr(const ConvHrK& k, hipStream_t st, const half_t* zp) {
  constexpr int SLOTS = (CH8 % 2) ? CH8 : CH8 + 1;
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int NG = (HR_TH + 2 * HALO) * (HR_TW + 2 * HALO) * SLOTS, NINST = (NG + 63) / 64;
  constexpr int SM_BYTES = NINST * 1024 + 16 + 64 * 4;
  static_assert(SM_BYTES <= 80 * 1024, "two workgroups per CU must fit");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_hr_kernel<CH8, true, TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, SM_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_hr_kernel<CH8, false, TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, SM_BYTES);
    attr_set = true;
  }
  // persistent: (workgroups per CU the registers admit) x 256 CUs, a multiple of 8 x ntile_c; never more than there is work
  const unsigned total = k.tiles_x * k.tiles_y * k.N;
  const unsigned unit = 8u * k.ntile_c;
  unsigned g = 256u * (CH8 == 4 ? 3u : 2u);
  if (g > total * k.ntile_c) g = total * k.ntile_c;
  g = (g + unit - 1) / unit * unit;
  dim3 grid(g);
  if (k.stat) hipLaunchKernelGGL((conv_hr_kernel<CH8, true, TAPS>), grid, dim3(256), SM_BYTES, st, k, zp);
  else hipLaunchKernelGGL((conv_hr_kernel<CH8, false, TAPS>), grid, dim3(256), SM_BYTES, st, k, zp);
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
  k.stat = d->stat_mode == CSBSR_STAT_SAMPLE_SUM ? d->stat : nullptr;
  k.tiles_x = (unsigned)((d->W + HR_TW - 1) / HR_TW); k.tiles_y = (unsigned)((d->H + HR_TH - 1) / HR_TH);
  { const char* e = getenv("CSBSR_HR_DBG"); k.dbg = e ? atoi(e) : 0; }
  int dev = 0;
  CSBSR_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < CSBSR_MAX_DEVICES, "conv_hr: no current device");
  if (!g_hr_zero_page[dev]) {
    CSBSR_CHECK(hipMalloc(reinterpret_cast<void**>(&g_hr_zero_page[dev]), 256) == hipSuccess, "conv_hr: zero page alloc failed");
    (void)hipMemset(g_hr_zero_page[dev], 0, 256);
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  g_last_conv_kernel = CONVK_HR;
  if (d->KH == 3) {
    if (d->in[0].c == 32) return launch_hr<4, 9>(k, st, g_hr_zero_page[dev]);
    return launch_hr<7, 9>(k, st, g_hr_zero_page[dev]);
  }
  if (d->in[0].c == 32) return launch_hr<4, 1>(k, st, g_hr_zero_page[dev]);
  return launch_hr<7, 1>(k, st, g_hr_zero_page[dev]);
}
