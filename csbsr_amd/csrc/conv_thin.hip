// Thin-output 3x3 convolution for gfx950: the 3-channel image heads of the path (kb.sr_reconst 128s -> 3, output_conv 512 -> 3 and the
// dgrad side of fe_SR.0, 49 -> 3) at HR resolution.  These layers are pure streaming -- 1 KB of input per pixel for 3 outputs -- but the
// implicit-GEMM kernels gather the input once per tap (9x through L1/L2) and waste 29/32 of every MFMA column tile.
//
// Here the nine taps move into the GEMM's N dimension: one pass over the (halo-extended) input tile computes
//     Y[q][(tap, co)] = sum_ci X[q][ci] * W[co][ci][tap]              (N = 9 * cout <= 32, one MFMA column tile, 27/32 live)
// with the A operand streamed global -> VGPR exactly once, and the convolution output is the shift-and-add
//     out[p][co] = sum_tap Y[p + off(tap)][(tap, co)]
// done from LDS in the epilogue, followed by the common fused epilogue (conv_common.h).  Same packed weights as conv_igemm.hip.
//
// K ordering: the MFMA only needs A and B to agree on which channel sits at which k, so within each group of 32 channels lanes 0-31
// take channels [0,16) and lanes 32-63 take [16,32) -- every lane issues one 32-byte global load per pixel per group (two k-steps),
// and each 64-byte sector is used in full.
#include "conv_common.h"

#define TN_TH 8
#define TN_TW 32
#define TN_HW (TN_TW + 2)
#define TN_HP ((TN_TH + 2) * TN_HW)        // 340 halo pixels
#define TN_MB ((TN_HP + 31) / 32)          // 11 row blocks of 32
#define TN_YLD 33

#define TN_TPW 1                           // tile rows per workgroup: the weight tile is staged once per workgroup

__global__ __launch_bounds__(256) void conv_thin_cout_kernel(const ConvK p, int tiles_x, int tiles_y, int wld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* sW = reinterpret_cast<half_t*>(smem);                                    // [32][wld]
  float* sY = reinterpret_cast<float*>(smem + (size_t)32 * wld * sizeof(half_t));  // [TN_MB*32][TN_YLD]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int tyg = (tiles_y + TN_TPW - 1) / TN_TPW;
  const int ty0 = (b % tyg) * TN_TPW;
  const int n = b / tyg;
  const int CR = p.cout, ntap_rows = 9 * CR;

  // ---- weights -> LDS: sW[(tap, co)][ci] = wt[co][tap * ctot + ci]; rows >= 9*cout and columns >= ctot are zero
  {
    const int chunks_per_row = wld / 8;
    for (int id = tid; id < 32 * chunks_per_row; id += 256) {
      const int row = id / chunks_per_row, ci = (id - row * chunks_per_row) * 8;
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (row < ntap_rows && ci < p.ctot) {
        const int tap = row / CR, co = row - tap * CR;
        v = *reinterpret_cast<const h8*>(p.wt + (size_t)co * p.Kp + (size_t)tap * p.ctot + ci);
      }
      *reinterpret_cast<h8*>(sW + row * wld + ci) = v;
    }
  }

  const int hi = lane >> 5;
  const float slope = p.act == CSBSR_ACT_PRELU ? p.prelu[0] : p.act_slope;
  float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (p.bias)
    for (int co = 0; co < CR; ++co) bias[co] = p.bias[n * p.bias_sn + co];
  bool first = true;
#pragma unroll 1
  for (int ty = ty0; ty < ty0 + TN_TPW && ty < tiles_y; ++ty) {
  // ---- per-lane pixel pointers of this wave's row blocks (wid, wid+4, wid+8)
  const half_t* base0[3];
  const half_t* base1[3];
  bool okp[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int q = (wid + 4 * i) * 32 + (lane & 31);
    const int hy = q / TN_HW, hx = q - hy * TN_HW;
    const int iy = ty * TN_TH + hy - 1, ix = tx * TN_TW + hx - 1;
    okp[i] = (wid + 4 * i) < TN_MB && q < TN_HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const long o0 = okp[i] ? (long)n * p.in[0].sn + (long)iy * p.in[0].sy + (long)ix * p.in[0].sx : 0;
    const long o1 = okp[i] ? (long)n * p.in[1].sn + (long)iy * p.in[1].sy + (long)ix * p.in[1].sx : 0;
    base0[i] = reinterpret_cast<const half_t*>(p.in[0].ptr) + o0;
    base1[i] = reinterpret_cast<const half_t*>(p.in[1].ptr) + o1 - p.c0;
  }
  f16v acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int npair = (p.ctot + 31) / 32;
  h8 a[3][2], an[3][2];
  auto load = [&](int j, h8 (&dst)[3][2]) {
    const int ch = 32 * j + 16 * hi;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const half_t* bp = (ch < p.c0 ? base0[i] : base1[i]) + ch;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (okp[i] && ch + 8 * s + 8 <= p.ctot) v = *reinterpret_cast<const h8*>(bp + 8 * s);
        dst[i][s] = v;
      }
    }
  };
  load(0, a);
  __syncthreads();                      // sW complete (first tile) / previous tile's sY reads done
  first = false;
  for (int j = 0; j < npair; ++j) {
    if (j + 1 < npair) load(j + 1, an);
    const half_t* wrow = sW + (lane & 31) * wld + 32 * j + 16 * hi;
    const h8 b0 = *reinterpret_cast<const h8*>(wrow);
    const h8 b1 = *reinterpret_cast<const h8*>(wrow + 8);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b0, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b1, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { a[i][0] = an[i][0]; a[i][1] = an[i][1]; }
  }

  // ---- Y tile -> LDS (D: col = lane & 31 = (tap, co); rows (r&3) + 8*(r>>2) + 4*(lane>>5))
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (wid + 4 * i >= TN_MB) continue;
    float* yb = sY + (size_t)(wid + 4 * i) * 32 * TN_YLD + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) yb[((r & 3) + 8 * (r >> 2) + 4 * hi) * TN_YLD] = acc[i][r];
  }
  __syncthreads();

  // ---- shift-and-add + fused epilogue: one output pixel per thread
  const int oyl = tid / TN_TW, oxl = tid % TN_TW;
  const int oy = ty * TN_TH + oyl, ox = tx * TN_TW + oxl;
  if (oy >= p.OH || ox >= p.OW) continue;
  float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float* y = sY + (size_t)((oyl + ky) * TN_HW + oxl + kx) * TN_YLD + (ky * 3 + kx) * CR;
      for (int co = 0; co < CR; ++co) v[co] += y[co];
    }
  float ssum[8], ssq[8];
  conv_epilogue_row(p, v, bias, slope, 0, n, oy, ox, ssum, ssq);
  }
}

static int g_conv_thin = 1;
void conv_thin_enable(int on) { g_conv_thin = on; }

bool conv_thin_eligible(const ConvK& k) {
  if (!g_conv_thin || k.transposed) return false;
  if (k.KHt != 3 || k.KWt != 3 || k.stride != 1 || k.dil != 1 || k.pad != 1) return false;
  if (k.cout * 9 > 32 || k.coutp != 8) return false;
  if (k.stat_mode != CSBSR_STAT_NONE) return false;
  if (k.OH != k.H || k.OW != k.W) return false;
  if (k.in[0].sx == 0 || (k.c0 != k.ctot && (k.in[1].sx == 0 || k.c0 % 32 != 0))) return false;
  if (k.ctot < 32 || k.ctot > 1024) return false;
  return true;
}

int conv_thin_launch(const ConvK& k, hipStream_t st) {
  const int tiles_x = (k.OW + TN_TW - 1) / TN_TW, tiles_y = (k.OH + TN_TH - 1) / TN_TH;
  const int wld = round_up(k.ctot, 32) + 8;
  const size_t smem = (size_t)32 * wld * sizeof(half_t) + (size_t)TN_MB * 32 * TN_YLD * sizeof(float);
  static size_t configured = 0;
  if (smem > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_thin_cout_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
        hipSuccess) {
      csbsr_set_error("conv(thin): cannot reserve %zu bytes of LDS", smem);
      return 1;
    }
    configured = smem;
  }
  const int tyg = (tiles_y + TN_TPW - 1) / TN_TPW;
  hipLaunchKernelGGL(conv_thin_cout_kernel, dim3((unsigned)(k.N * tyg * tiles_x)), dim3(256), smem, st, k, tiles_x, tiles_y, wld);
  CSBSR_LAUNCH_CHECK("csbsr_conv_forward(thin)");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Thin-input 3x3 convolution: the mirror case (3-channel image in, 49..512 channels out: fe_SR.0 and the dgrad side of the
// image heads).  The padded implicit GEMM runs K = 9 taps x 8 padded channels = 72 -> 128; here K is the dense (tap, channel)
// index, 27 -> 32 = two MFMA k-steps, the pixel operand is gathered once per tile row from an LDS copy of the halo tile and
// reused for every output-channel tile, and the kernel is bound by writing the output.  D[cout][pixel], register-direct epilogue.
#define TK_WLD 40

__global__ __launch_bounds__(256) void conv_thin_cin_kernel(const ConvK p, int tiles_x, int tiles_y, int CI) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nct = (p.cout + 31) / 32;
  half_t* sWt = reinterpret_cast<half_t*>(smem);                     // [nct*32][TK_WLD], k = tap*CI + c
  half_t* sIn = sWt + (size_t)nct * 32 * TK_WLD;                      // [TN_HP][4]
  float* sBias = reinterpret_cast<float*>(sIn + TN_HP * 4);           // [nct*32]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int tyg = (tiles_y + TN_TPW - 1) / TN_TPW;
  const int ty0 = (b % tyg) * TN_TPW;
  const int n = b / tyg;

  for (int id = tid; id < nct * 32 * (TK_WLD / 8); id += 256) {
    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<h8*>(sWt + id * 8) = z;
  }
  for (int id = tid; id < nct * 32; id += 256) sBias[id] = (p.bias && id < p.cout) ? p.bias[id] : 0.f;
  __syncthreads();
  for (int id = tid; id < nct * 32 * 9; id += 256) {
    const int co = id / 9, tap = id - co * 9;
    const h8 v = *reinterpret_cast<const h8*>(p.wt + (size_t)co * p.Kp + tap * 8);
    for (int c = 0; c < CI; ++c) sWt[co * TK_WLD + tap * CI + c] = v[c];
  }
  const float slope = p.act == CSBSR_ACT_PRELU ? p.prelu[0] : p.act_slope;
  const EpiFast fe = conv_epilogue_fast_setup(p, slope);
#pragma unroll 1
  for (int ty = ty0; ty < ty0 + TN_TPW && ty < tiles_y; ++ty) {
  __syncthreads();                      // previous tile's sIn reads done
  for (int id = tid; id < TN_HP; id += 256) {
    const int hy = id / TN_HW, hx = id - hy * TN_HW;
    const int iy = ty * TN_TH + hy - 1, ix = tx * TN_TW + hx - 1;
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
      v = *reinterpret_cast<const h8*>(reinterpret_cast<const half_t*>(p.in[0].ptr) + (long)n * p.in[0].sn + (long)iy * p.in[0].sy +
                                       (long)ix * p.in[0].sx);
    half_t* d = sIn + id * 4;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = 0;
  }
  __syncthreads();
#pragma unroll 1
  for (int rb = 0; rb < 2; ++rb) {
    const int oyl = wid * 2 + rb, oxl = lane & 31;
    const int oy = ty * TN_TH + oyl, ox = tx * TN_TW + oxl;
    h8 pf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * s + 8 * hi + e;
        const int tap = k / CI, c = k - tap * CI;
        const int ky = tap / 3, kx = tap - ky * 3;
        pf[s][e] = tap < 9 ? sIn[((oyl + ky) * TN_HW + oxl + kx) * 4 + c] : (half_t)0;
      }
    const int nn = (oy < p.OH && ox < p.OW) ? n : -1;
#pragma unroll 4
    for (int ct = 0; ct < nct; ++ct) {
      const half_t* wrow = sWt + (size_t)(ct * 32 + (lane & 31)) * TK_WLD + 8 * hi;
      const h8 w0 = *reinterpret_cast<const h8*>(wrow);
      const h8 w1 = *reinterpret_cast<const h8*>(wrow + 16);
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, pf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, pf[1], acc, 0, 0, 0);
      conv_epilogue_direct_tile(p, acc, ct * 32, slope, nn, oy, ox, sBias, &fe);
    }
  }
  }
}

bool conv_thin_cin_eligible(const ConvK& k, int creal) {
  if (!g_conv_thin || k.transposed) return false;
  if (k.KHt != 3 || k.KWt != 3 || k.stride != 1 || k.dil != 1 || k.pad != 1) return false;
  if (k.ctot != 8 || k.c0 != 8 || creal < 1 || creal > 3 || k.in[0].sx == 0) return false;
  if (k.stat_mode != CSBSR_STAT_NONE || k.cbias) return false;
  if (k.OH != k.H || k.OW != k.W) return false;
  if (k.cout < 32 || k.cout > 1024) return false;
  return true;
}

int conv_thin_cin_launch(const ConvK& k, int creal, hipStream_t st) {
  const int tiles_x = (k.OW + TN_TW - 1) / TN_TW, tiles_y = (k.OH + TN_TH - 1) / TN_TH;
  const int nct = (k.cout + 31) / 32;
  const size_t smem = ((size_t)nct * 32 * TK_WLD + (size_t)TN_HP * 4) * sizeof(half_t) + (size_t)nct * 32 * sizeof(float);
  static size_t configured = 0;
  if (smem > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_thin_cin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
        hipSuccess) {
      csbsr_set_error("conv(thin-in): cannot reserve %zu bytes of LDS", smem);
      return 1;
    }
    configured = smem;
  }
  const int tyg = (tiles_y + TN_TPW - 1) / TN_TPW;
  hipLaunchKernelGGL(conv_thin_cin_kernel, dim3((unsigned)(k.N * tyg * tiles_x)), dim3(256), smem, st, k, tiles_x, tiles_y, creal);
  CSBSR_LAUNCH_CHECK("csbsr_conv_forward(thin-in)");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// The same layers where the epilogue is plain -- no bias, no residual, no mask, no statistics; optionally ACCUMULATING into the output
// (the dgrads of kb.sr_reconst / output_conv into the concatenated feature gradient, fe_SR.0 forward): a lean streaming kernel.  These
// launches are 5 % of the training step and pure HBM streams (read old + write: 6.6 GB at 128 channels, N = 4), which the kernel above
// moves at ~3 TB/s: its general epilogue costs 236 VGPRs (two workgroups per CU) and requests each old-output piece right before its
// use, so ~16 KB per CU are in flight.  Here: one row of 32 pixels per wave, 8 waves, the old-output pieces of the NEXT four 32-cout
// tiles always requested (a rolling register ring refilled right after each piece is consumed), activation as max(t, t * a),
// persistent workgroups with the tiny 3-channel halo tile double-buffered in LDS one tile ahead.
#define TC2_D 4                              // cout tiles of old output in flight per wave
// DACT: this (accumulating) dgrad completes the output gradient of a layer  out = prelu(pre) +- res  (p.mask = its saved output, p.res /
// p.res_mode ITS residual): the unmasked total goes to x.dres as the residual's gradient, what is stored to out16 is dPre = total x
// prelu'(mask -+ res), and the layer's PReLU-slope gradient leaves as one partial sum per workgroup -- that layer's whole
// epilogue-backward pass (csbsr_conv_desc_t::dact_prelu / dres, as csrc/conv_tp.hip does for the 2x2-tap transposed dgrads).
struct Cin2Dact { half_t* dres; long d_sn, d_sy, d_sx; const float* slope; float* part; };
template <bool DACT>
__global__ __launch_bounds__(512, DACT ? 1 : 2) void conv_thin_cin2_kernel(const ConvK p, int tiles_x, int tiles_y, int CI, int has_old, const Cin2Dact x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RD = DACT ? 2 : TC2_D;      // cout tiles of epilogue operands in flight per wave (DACT: three operands per piece)
  const int nct = (p.cout + 31) / 32;
  half_t* sWt = reinterpret_cast<half_t*>(smem);                     // [nct*32][TK_WLD], k = tap*CI + c
  half_t* sIn = sWt + (size_t)nct * 32 * TK_WLD;                      // [2][TN_HP][8]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, pix = lane & 31;
  for (int id = tid; id < nct * 32 * (TK_WLD / 8); id += 512) {
    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<h8*>(sWt + id * 8) = z;
  }
  __syncthreads();
  for (int id = tid; id < nct * 32 * 9; id += 512) {
    const int co = id / 9, tap = id - co * 9;
    const h8 v = *reinterpret_cast<const h8*>(p.wt + (size_t)co * p.Kp + tap * 8);
    for (int c = 0; c < CI; ++c) sWt[co * TK_WLD + tap * CI + c] = v[c];
  }
  const float aslope = p.act == CSBSR_ACT_RELU ? 0.f : (p.act == CSBSR_ACT_LRELU ? p.act_slope : 1.f);
  const float mslope = DACT ? *x.slope : 1.f, rsign = p.res_mode == CSBSR_RES_SUB ? -1.f : 1.f;
  float dpr = 0.f;
  const half_t* in0 = reinterpret_cast<const half_t*>(p.in[0].ptr);
  const unsigned per_img = (unsigned)(tiles_x * tiles_y), total = per_img * (unsigned)p.N;
  // halo pixel of this thread (threads < TN_HP) for a tile: one 16-byte load, out-of-image pixels zero
  auto halo_load = [&](unsigned t) {
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tid < TN_HP && t < total) {
      const int n = t / per_img;
      const unsigned r = t - n * per_img;
      const int hy = tid / TN_HW, hx = tid - hy * TN_HW;
      const int iy = (int)(r / tiles_x) * TN_TH + hy - 1, ix = (int)(r % tiles_x) * TN_TW + hx - 1;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v = *reinterpret_cast<const h8*>(in0 + (long)n * p.in[0].sn + (long)iy * p.in[0].sy + (long)ix * p.in[0].sx);
    }
    return v;
  };
  unsigned t = blockIdx.x;
  if (t >= total) return;
  {
    const h8 v = halo_load(t);
    if (tid < TN_HP) *reinterpret_cast<h8*>(sIn + tid * 8) = v;
  }
  __syncthreads();
  int buf = 0;
  for (; t < total; t += gridDim.x) {
    const h8 hnext = halo_load(t + gridDim.x);                        // the next tile's halo pixel, stored after this tile's MFMAs
    const int n = t / per_img;
    const unsigned r_ = t - n * per_img;
    const int oy = (int)(r_ / tiles_x) * TN_TH + wid, ox = (int)(r_ % tiles_x) * TN_TW + pix;
    const bool live = oy < p.OH && ox < p.OW;
    // the row's pixel operand: K = (tap, channel) dense, 27 -> 32
    const half_t* sI = sIn + buf * (TN_HP * 8);
    h8 pf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * s + 8 * hi + e;
        const int tap = k / CI, c = k - tap * CI;
        const int ky = tap / 3, kx = tap - ky * 3;
        pf[s][e] = tap < 9 ? sI[((wid + ky) * TN_HW + pix + kx) * 8 + c] : (half_t)0;
      }
    // output pointer of this lane: couts 8 hi .. of its pixel; piece (ct, pair) at + 32 ct + 16 pair (dead lanes re-read pixel 0)
    half_t* o = p.out16 + (live ? n * p.o_sn + oy * p.o_sy + ox * p.o_sx : 0) + 8 * hi;
    h8 ring[RD][2];
    h8 ringm[DACT ? RD : 1][2], ringr[DACT ? RD : 1][2];
    const half_t* mo = DACT ? p.mask + (live ? n * p.m_sn + oy * p.m_sy + ox * p.m_sx : 0) + 8 * hi : nullptr;
    const half_t* ro = DACT ? p.res + (live ? n * p.r_sn + oy * p.r_sy + ox * p.r_sx : 0) + 8 * hi : nullptr;
    half_t* dro = DACT ? x.dres + (live ? n * x.d_sn + oy * x.d_sy + ox * x.d_sx : 0) + 8 * hi : nullptr;
    auto piece_ok = [&](int ct, int pair) { return 32 * ct + 16 * pair + 8 * hi < p.coutp; };
    if (has_old) {
#pragma unroll
      for (int j = 0; j < RD; ++j)
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
          const int off = piece_ok(j, pair) ? 32 * j + 16 * pair : 0;
          ring[j][pair] = *reinterpret_cast<const h8*>(o + off);
          if constexpr (DACT) { ringm[j][pair] = *reinterpret_cast<const h8*>(mo + off); ringr[j][pair] = *reinterpret_cast<const h8*>(ro + off); }
        }
    }
    for (int c4 = 0; c4 < nct; c4 += RD) {
#pragma unroll
      for (int j = 0; j < RD; ++j) {
        const int ct = c4 + j;
        const int ctc = ct < nct ? ct : nct - 1;
        const half_t* wrow = sWt + (size_t)(ctc * 32 + pix) * TK_WLD + 8 * hi;
        const h8 w0 = *reinterpret_cast<const h8*>(wrow);
        const h8 w1 = *reinterpret_cast<const h8*>(wrow + 16);
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, pf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, pf[1], acc, 0, 0, 0);
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned a = __float_as_uint(acc[8 * pair + q]), b = __float_as_uint(acc[8 * pair + 4 + q]);
            auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
            v[q] = __uint_as_float(r[0]);
            v[4 + q] = __uint_as_float(r[1]);
          }
          const h8 old = ring[j][pair];
          h8 mk = {0, 0, 0, 0, 0, 0, 0, 0}, rs = {0, 0, 0, 0, 0, 0, 0, 0};
          if constexpr (DACT) { mk = ringm[j][pair]; rs = ringr[j][pair]; }
          // refill this ring slot with the piece RD cout tiles on (clamped: harmless re-read past the last tile)
          if (has_old) {
            const int ctn = ct + RD;
            const int off = (ctn < nct && piece_ok(ctn, pair)) ? 32 * ctn + 16 * pair : 0;
            ring[j][pair] = *reinterpret_cast<const h8*>(o + off);
            if constexpr (DACT) { ringm[j][pair] = *reinterpret_cast<const h8*>(mo + off); ringr[j][pair] = *reinterpret_cast<const h8*>(ro + off); }
          }
          h8 hv, hd;
          const bool st_ok = live && ct < nct && piece_ok(ct, pair);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float tv = v[e] * p.out_scale;
            tv = fmaxf(tv, tv * aslope);
            if (has_old) tv += (float)old[e];
            if constexpr (DACT) {
              const float y = (float)mk[e] - rsign * (float)rs[e];      // the masking layer's activation, rebuilt from its output and residual
              const bool pos = y > 0.f;
              hd[e] = (half_t)(rsign * tv);                               // d(res)
              if (st_ok && !pos) dpr += tv * y;
              tv = pos ? tv : tv * mslope;
            }
            hv[e] = (half_t)tv;
          }
          if (st_ok) {
            *reinterpret_cast<h8*>(o + 32 * ct + 16 * pair) = hv;
            if constexpr (DACT) *reinterpret_cast<h8*>(dro + 32 * ct + 16 * pair) = hd;
          }
        }
      }
    }
    // next tile's halo into the other buffer (nobody reads it during this tile), then one barrier per tile
    if (tid < TN_HP) *reinterpret_cast<h8*>(sIn + (buf ^ 1) * (TN_HP * 8) + tid * 8) = hnext;
    __syncthreads();
    buf ^= 1;
  }
  if constexpr (DACT) {      // the slope gradient: lanes, then the eight waves one after the other (fixed order), one partial per workgroup
    __shared__ float sdp[8];
    for (int o_ = 1; o_ < 64; o_ <<= 1) dpr += __shfl_xor(dpr, o_, 64);
    if (lane == 0) sdp[wid] = dpr;
    __syncthreads();
    if (tid == 0) {
      float a = 0.f;
      for (int w_ = 0; w_ < 8; ++w_) a += sdp[w_];
      x.part[blockIdx.x] = a / mslope;
    }
  }
}

static int g_conv_thin2 = 1;
void conv_thin_cin2_enable(int on) { g_conv_thin2 = on; }

bool conv_thin_cin2_eligible(const ConvK& k, int creal) {
  if (!g_conv_thin2 || !conv_thin_cin_eligible(k, creal)) return false;
  if (k.bias || k.cbias || k.res_mode != CSBSR_RES_NONE || k.mask || k.out32 || !k.out16 || k.o_lo) return false;
  if (k.act != CSBSR_ACT_NONE && k.act != CSBSR_ACT_RELU && k.act != CSBSR_ACT_LRELU) return false;
  if (k.coutp % 8 != 0 || k.o_sx < k.coutp) return false;
  return true;
}

// the accumulating launch that also takes over the epilogue-backward pass of the PReLU + residual layer below (DACT above)
extern "C" int32_t csbsr_conv_thin_dact_eligible(const csbsr_conv_desc_t* d) {
  if (!d || !g_conv_thin2 || d->transposed || !d->mask || !d->mask_prelu || !d->dres || !d->res || d->dact_bias) return 0;
  if (d->res_mode != CSBSR_RES_ADD && d->res_mode != CSBSR_RES_SUB) return 0;
  if (!d->accumulate || d->act != CSBSR_ACT_NONE || d->bias || d->cbias || d->out32 || !d->out16 || d->o_lo || d->r_lo || d->stat_mode != CSBSR_STAT_NONE) return 0;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dil != 1 || d->pad != 1 || d->OH != d->H || d->OW != d->W) return 0;
  if (d->in[1].c != 0 || d->in[0].c != 8 || d->in[0].creal < 1 || d->in[0].creal > 3 || d->in[0].sx == 0) return 0;
  if (d->cout < 32 || d->cout > 1024 || d->coutp % 8 != 0 || d->o_sx < d->coutp || d->dr_sx < d->coutp) return 0;
  return 1;
}

int conv_thin_cin2_launch(const ConvK& k, int creal, hipStream_t st) {
  const int tiles_x = (k.OW + TN_TW - 1) / TN_TW, tiles_y = (k.OH + TN_TH - 1) / TN_TH;
  const int nct = (k.cout + 31) / 32;
  const size_t smem = ((size_t)nct * 32 * TK_WLD + (size_t)2 * TN_HP * 8) * sizeof(half_t);
  static size_t configured = 0;
  if (smem > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_thin_cin2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
      csbsr_set_error("conv(thin-in, streaming): cannot reserve %zu bytes of LDS", smem);
      return 1;
    }
    configured = smem;
  }
  const long total = (long)k.N * tiles_x * tiles_y;
  const unsigned g = (unsigned)(total < 512 ? total : 512);
  hipLaunchKernelGGL(conv_thin_cin2_kernel<false>, dim3(g), dim3(512), smem, st, k, tiles_x, tiles_y, creal, k.accumulate ? 1 : 0, Cin2Dact{});
  CSBSR_LAUNCH_CHECK("csbsr_conv_forward(thin-in, streaming)");
  return 0;
}

int conv_thin_cin2_dact_launch(const ConvK& k, const csbsr_conv_desc_t* d, hipStream_t st) {
  const int tiles_x = (k.OW + TN_TW - 1) / TN_TW, tiles_y = (k.OH + TN_TH - 1) / TN_TH;
  const int nct = (k.cout + 31) / 32;
  const size_t smem = ((size_t)nct * 32 * TK_WLD + (size_t)2 * TN_HP * 8) * sizeof(half_t);
  static size_t configured = 0;
  if (smem > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_thin_cin2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
      csbsr_set_error("conv(thin-in, streaming, dact): cannot reserve %zu bytes of LDS", smem);
      return 1;
    }
    configured = smem;
  }
  const long total = (long)k.N * tiles_x * tiles_y;
  const unsigned g = (unsigned)(total < 256 ? total : 256);      // one 512-thread workgroup per CU (the rings of three operands: 215 registers)
  Cin2Dact x;
  x.dres = reinterpret_cast<half_t*>(d->dres); x.d_sn = d->dr_sn; x.d_sy = d->dr_sy; x.d_sx = d->dr_sx;
  x.slope = d->mask_prelu;
  x.part = csbsr_red_scratch((long)g);
  CSBSR_NEED_SCRATCH(x.part, "conv(thin-in, dact)");
  hipLaunchKernelGGL(conv_thin_cin2_kernel<true>, dim3(g), dim3(512), smem, st, k, tiles_x, tiles_y, d->in[0].creal, 1, x);
  if (d->dact_prelu) { if (int e = csbsr_sum_partials(x.part, (int)g, 1, 1, d->dact_prelu, st)) return e; }
  CSBSR_LAUNCH_CHECK("csbsr_conv_forward(thin-in, streaming, dact)");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Thin-input TRANSPOSED convolution (kernel <= 2 x stride: 2 x 2 taps per output phase) from a 3-channel image: kb.up_conv1 of every
// back-projection stage (DeconvBlock 3 -> 128, 8x8 stride 4, PReLU, + residual; kbpn.py:497-503).  12 MACs per output value: the
// padded implicit GEMM ran it at 22 TFLOP/s (2.8 ms per launch for a 1.3 ms read + write stream).  Streaming VALU kernel: a thread owns
// one output phase column (ox % s) and one channel octet, keeps that phase's 4 taps x 3 channels x 8 couts of weights in registers
// (fp16-rounded, straight from the standard phase-major pack) and walks the pixels of its phase along a group of rows with equal
// oy % s; the 16 octet-threads of a pixel share its four 16-byte input loads and write one contiguous 256-byte pixel.
#define TP_ROW_MAX 1024      // input width the row staging holds (conv_thin_tp_eligible)
template <int CI>
__global__ __launch_bounds__(256) void conv_thin_tp_kernel(const ConvK p, int groups_per_phase) {
  const int tid = threadIdx.x;
  const int c8 = p.coutp >> 3;                          // channel octets (<= 16)
  const int oct = tid % c8, pl = tid / c8;              // pixel lanes: 256 / c8
  const int npl = 256 / c8;
  if (pl >= npl) return;
  const int s = p.stride;
  int b = blockIdx.x;
  const int rg = b % groups_per_phase; b /= groups_per_phase;
  const int py = b % s;
  const int n = b / s;
  const int px = pl % s, sub = pl / s, nsub = npl / s;  // (npl is a multiple of s: launcher)
  const int by = (py + p.pad) / s, bx = (px + p.pad) / s;
  const int ph = py * s + px;
  const int co = oct * 8;
  // weights as fp16 channel pairs for v_dot2_f32_f16 (exact products, fp32 accumulate): (c0, c1) and (c2, 0) per tap -- the input's
  // channel 3 is zero padding
  static_assert(CI == 3, "channel pairs (c0, c1), (c2, pad)");
  h2 w2[4][2][8];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const half_t* wp = p.wt + ((size_t)ph * p.rows_p + co + e) * p.Kp + t * p.ctot;
      w2[t][0][e] = h2{wp[0], wp[1]};
      w2[t][1][e] = h2{wp[2], (half_t)0};
    }
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = (p.bias && co + e < p.cout) ? p.bias[co + e] : 0.f;
  const float slope = p.act == CSBSR_ACT_PRELU ? *p.prelu : (p.act == CSBSR_ACT_RELU ? 0.f : (p.act == CSBSR_ACT_NONE ? 1.f : p.act_slope));
  const float rsign = p.res_mode == CSBSR_RES_ADD ? 1.f : (p.res_mode == CSBSR_RES_SUB ? -1.f : 0.f);
  const half_t* in0 = reinterpret_cast<const half_t*>(p.in[0].ptr) + (long)n * p.in[0].sn;
  const int rows_per_group = (p.OH / s + groups_per_phase - 1) / groups_per_phase;
  // the two input rows an output row reads (3 channels in one 16-byte pixel), staged in LDS once per row with a zero pixel at either end:
  // as per-pixel global loads (L2 hits, but two rounds of L2 latency per pass on the critical path of a kernel with two waves per SIMD)
  // they held the HBM stream of the residual at 2.9 TB/s
  __shared__ h8 srow[2][TP_ROW_MAX + 2];
  for (int r = 0; r < rows_per_group; ++r) {
    const int qy = rg * rows_per_group + r;
    const int oy = qy * s + py;
    if (oy >= p.OH) break;
    __syncthreads();
    for (int i = tid; i < 2 * (p.W + 2); i += 256) {
      const int tr = i / (p.W + 2), ix = i - tr * (p.W + 2) - 1, iy = qy + by - tr;
      h8 v = h8{0, 0, 0, 0, 0, 0, 0, 0};
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = *reinterpret_cast<const h8*>(in0 + (long)iy * p.in[0].sy + (long)ix * p.in[0].sx);
      srow[tr][ix + 1] = v;
    }
    __syncthreads();
    // EIGHT pixels per pass: their residual loads -- the HBM stream of this kernel -- are all requested up front (with four, 2 workgroups
    // x 256 threads x 64 B kept 32 KB per CU in flight: 2.45 TB/s by Little's law, which is what the kernel ran at); the input loads
    // (a 3-channel LR image: L1 / L2 hits) follow four pixels at a time.  Stores to out16 may alias every load as far as the compiler
    // knows, so it would not hoist the next pixels' loads itself.
    for (int qx0 = sub; qx0 * s + px < p.OW; qx0 += 8 * nsub) {
      h8 rr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int ox = (qx0 + u * nsub) * s + px;
        rr[u] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        if (ox < p.OW && rsign != 0.f) rr[u] = *reinterpret_cast<const h8*>(p.res + n * p.r_sn + (long)oy * p.r_sy + (long)ox * p.r_sx + co);
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        h8 xv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int qx = qx0 + (4 * half + u) * nsub, ox = qx * s + px;
          const bool live = ox < p.OW;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int ix = qx + bx - (t & 1);
            xv[u][t] = srow[t >> 1][live ? ix + 1 : 0];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int qx = qx0 + (4 * half + u) * nsub, ox = qx * s + px;
          if (ox >= p.OW) break;
          float acc[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const h2 x01 = h2{xv[u][t][0], xv[u][t][1]}, x23 = h2{xv[u][t][2], (half_t)0};      // (the input's padding lane is never read: 0 * NaN would poison every output)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              acc[e] = __builtin_amdgcn_fdot2(x01, w2[t][0][e], acc[e], false);
              acc[e] = __builtin_amdgcn_fdot2(x23, w2[t][1][e], acc[e], false);
            }
          }
          h8 hv;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t_ = acc[e] * p.out_scale + bias[e];
            t_ = t_ > 0.f ? t_ : t_ * slope;
            if (co + e >= p.cout) t_ = 0.f;
            t_ += rsign * (float)rr[4 * half + u][e];
            hv[e] = (half_t)t_;
          }
          *reinterpret_cast<h8*>(p.out16 + n * p.o_sn + (long)oy * p.o_sy + (long)ox * p.o_sx + co) = hv;
        }
      }
    }
  }
}

bool conv_thin_tp_eligible(const ConvK& k, int creal, bool second_seg) {
  if (!g_conv_thin || !k.transposed || second_seg) return false;
  if (k.KHt != 2 || k.KWt != 2 || k.stride < 2 || k.dil != 1 || k.pad >= k.stride) return false;
  if (k.ctot != 8 || creal != 3 || k.in[0].sx == 0 || k.W > TP_ROW_MAX) return false;
  if (k.OH != k.H * k.stride || k.OW != k.W * k.stride) return false;
  if (k.coutp < 8 || k.coutp > 128 || (256 / (k.coutp / 8)) % k.stride != 0 || 256 % (k.coutp / 8) != 0) return false;
  if (!k.out16 || k.out32 || k.o_lo || k.r_lo || k.cbias || k.mask || k.accumulate || k.stat_mode != CSBSR_STAT_NONE) return false;
  if (k.act == CSBSR_ACT_SIGMOID || (k.res_mode != CSBSR_RES_NONE && k.res_mode != CSBSR_RES_ADD && k.res_mode != CSBSR_RES_SUB)) return false;
  return true;
}

int conv_thin_tp_launch(const ConvK& k, hipStream_t st) {
  // row groups per (sample, row phase): enough workgroups to fill the chip several times over, at least ~4 rows each
  const int rows = k.OH / k.stride;
  int gpp = (8 * 256 + k.N * k.stride - 1) / (k.N * k.stride);
  if (gpp > rows / 4) gpp = rows / 4;
  if (gpp < 1) gpp = 1;
  hipLaunchKernelGGL((conv_thin_tp_kernel<3>), dim3((unsigned)(k.N * k.stride * gpp)), dim3(256), 0, st, k, gpp);
  CSBSR_LAUNCH_CHECK("csbsr_conv_forward(thin transposed)");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Stride-2 transposed convolution from 64 channels INTO <= 3: the dgrad of the detectors' stems (PSPNet: Conv2d 3 -> 64, 7x7 stride 2,
// extractors.py:45; HRNet: 3x3 stride 2, hrnet_backbone.py:303) -- the gradient of the SR image under the segmentation loss.  It ran on the
// 32-cout register-staged implicit GEMM (29 of 32 MFMA columns multiplying zeros, every tap re-gathered): 5.3 ms at B = 8, 30 TF/s, 190 GB/s.
// Here a workgroup owns a 16 x 32 output tile; the (8 + taps) x (16 + taps) input pixels it reads are staged in LDS once (144-byte
// pixel pitch: an odd number of 16-byte slots, conflict-free b128 reads); wave w multiplies the output phase (w / 2, w % 2) -- one tap
// set per wave, so the tap's weights are wave-uniform and come as SCALAR loads (constant address space) straight into the v_dot2
// operands -- two pixels per lane, fp32 accumulate, the common epilogue row.  Phase-packed weights of csbsr_pack_weights (kind 2).
#define TPD_TH 16
#define TPD_TW 32
#define TPD_PITCH 72
#define TPD_IH 12
#define TPD_IW 20
typedef __attribute__((address_space(4))) const unsigned tpd_cu32;

__global__ __launch_bounds__(256) void conv_thin_tpd_kernel(const ConvK p, int tiles_x, int tiles_y, int KH) {
  __shared__ __attribute__((aligned(16))) half_t sIn[TPD_IH * TPD_IW * TPD_PITCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = wid >> 1, px = wid & 1, ph = wid;
  const int by = (py + p.pad) >> 1, bx = (px + p.pad) >> 1, bmin = p.pad >> 1, bmax = (1 + p.pad) >> 1;
  const int IH = TPD_TH / 2 + bmax - bmin + p.KHt - 1, IW = TPD_TW / 2 + bmax - bmin + p.KWt - 1;
  const int qyl = lane >> 4, qxl = lane & 15;
  tpd_cu32* wc = (tpd_cu32*)(unsigned long)(p.wt + (size_t)ph * p.rows_p * p.Kp);
  const int kpd = p.Kp >> 1;                           // dwords per weight row
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = (p.bias && e < p.cout) ? p.bias[e] : 0.f;
  const float slope = p.act == CSBSR_ACT_PRELU ? *p.prelu : p.act_slope;
  const half_t* in0 = reinterpret_cast<const half_t*>(p.in[0].ptr);
  const unsigned per_img = (unsigned)(tiles_x * tiles_y), total = per_img * (unsigned)p.N;
  for (unsigned t = blockIdx.x; t < total; t += gridDim.x) {
    const int n = t / per_img;
    const unsigned r_ = t - n * per_img;
    const int oy0 = (int)(r_ / tiles_x) * TPD_TH, ox0 = (int)(r_ % tiles_x) * TPD_TW;
    const int iyo = oy0 / 2 + bmin - (p.KHt - 1), ixo = ox0 / 2 + bmin - (p.KWt - 1);
    __syncthreads();
    for (int id = tid; id < IH * IW * 8; id += 256) {
      const int pix = id >> 3, oc = id & 7;
      const int ly = pix / IW, lx = pix - ly * IW;
      const int iy = iyo + ly, ix = ixo + lx;
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v = *reinterpret_cast<const h8*>(in0 + (long)n * p.in[0].sn + (long)iy * p.in[0].sy + (long)ix * p.in[0].sx + oc * 8);
      *reinterpret_cast<h8*>(sIn + pix * TPD_PITCH + oc * 8) = v;
    }
    __syncthreads();
    float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll 1
    for (int ty = 0; ty < p.KHt; ++ty) {
      if (((py + p.pad) & 1) + 2 * ty >= KH) break;            // (wave-uniform) the phase has no such kernel row: its packed weights are zeros
      const int ly = qyl + (by - bmin) + (p.KHt - 1) - ty;
#pragma unroll 1
      for (int tx = 0; tx < p.KWt; ++tx) {
        if (((px + p.pad) & 1) + 2 * tx >= KH) break;
        const int lx = qxl + (bx - bmin) + (p.KWt - 1) - tx;
        const half_t* s0 = sIn + (ly * IW + lx) * TPD_PITCH;
        const half_t* s1 = s0 + 4 * IW * TPD_PITCH;
        const int wbase = ((ty * p.KWt + tx) * 64) >> 1;
#pragma unroll 2
        for (int oc = 0; oc < 8; ++oc) {      // (two octets = 24 scalar weight registers at a time: all eight overflowed the SGPR file)
          const h8 x0 = *reinterpret_cast<const h8*>(s0 + oc * 8), x1 = *reinterpret_cast<const h8*>(s1 + oc * 8);
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned wu = wc[r * kpd + wbase + oc * 4 + j];
              const h2 w2 = __builtin_bit_cast(h2, wu);
              acc[0][r] = __builtin_amdgcn_fdot2(h2{x0[2 * j], x0[2 * j + 1]}, w2, acc[0][r], false);
              acc[1][r] = __builtin_amdgcn_fdot2(h2{x1[2 * j], x1[2 * j + 1]}, w2, acc[1][r], false);
            }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oy = oy0 + 2 * (qyl + 4 * j) + py, ox = ox0 + 2 * qxl + px;
      if (oy >= p.OH || ox >= p.OW) continue;
      float v[8] = {acc[j][0], acc[j][1], acc[j][2], 0.f, 0.f, 0.f, 0.f, 0.f}, ssum[8], ssq[8];
      conv_epilogue_row(p, v, bias, slope, 0, n, oy, ox, ssum, ssq);
    }
  }
}

static int g_conv_thin_tpd = 1;
void conv_thin_tpd_enable(int on) { g_conv_thin_tpd = on; }

bool conv_thin_tpd_eligible(const ConvK& k, int KH) {
  if (!g_conv_thin_tpd || !k.transposed || k.stride != 2 || k.dil != 1 || k.KHt != k.KWt || k.KHt > 4 || k.pad < 0) return false;
  if (KH > 2 * k.KHt || TPD_TH / 2 + ((1 + k.pad) >> 1) - (k.pad >> 1) + k.KHt - 1 > TPD_IH) return false;
  if (k.c0 != k.ctot || k.ctot != 64 || k.in[0].sx == 0 || k.cout > 3 || k.coutp != 8 || k.rows_p < 3) return false;
  if (k.cbias || k.mask || k.o_lo || k.r_lo || k.r2_lo || k.stat_mode != CSBSR_STAT_NONE || k.bias_sn || k.fs) return false;
  if (k.OH != 2 * k.H || k.OW != 2 * k.W) return false;
  if ((long)k.N * k.OH * k.OW < 64L * 1024) return false;          // small maps: the general kernel
  return true;
}

int conv_thin_tpd_launch(const ConvK& k, int KH, hipStream_t st) {
  const int tiles_x = (k.OW + TPD_TW - 1) / TPD_TW, tiles_y = (k.OH + TPD_TH - 1) / TPD_TH;
  const long total = (long)k.N * tiles_x * tiles_y;
  const unsigned g = (unsigned)(total < 1024 ? total : 1024);
  hipLaunchKernelGGL(conv_thin_tpd_kernel, dim3(g), dim3(256), 0, st, k, tiles_x, tiles_y, KH);
  CSBSR_LAUNCH_CHECK("csbsr_conv_forward(thin transposed dgrad)");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Strided convolution INTO <= 3 channels from 128 (8x8 stride 4 / 12x12 stride 8): the dgrad of kb.up_conv1 (ConvTranspose2d 3 -> 128, 8x8
// stride 4; kbpn.py:372-374) -- the gradient of the 3-channel LR error image, an 8192-term dot product per output value over a 3.3 GB
// map (N = 4).  On the 32-cout implicit-GEMM tile it ran at 1.8 TB/s with 29 of 32 MFMA columns multiplying zeros.  Here a wave owns four
// consecutive output pixels of a row x all 16 channel octets (lane = 16 * pixel + octet): every tap is one fully coalesced 1 KB load
// (4 x 256 contiguous bytes), multiplied with v_dot2_f32_f16 against the tap's weights for the lane's octet -- the whole packed weight
// matrix sits in LDS ([tap][octet][cout][8], 49 KB for 8x8), read as three 16-byte broadcasts per tap -- and the 16 octet partial sums of
// a pixel meet in a 4-step xor-shuffle.  A workgroup = 4 waves = a 4-row x 16-column tile of outputs: consecutive rows share half of
// their input rows (L1), neighbouring tiles a fifth (L2).  Plain epilogue (out_scale only): fp32 planar and / or fp16 NHWC output.
template <int KS>
__global__ __launch_bounds__(256) void conv_thin_sc_kernel(const ConvK p, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = KS * KS;
  h8* sW = reinterpret_cast<h8*>(smem);                       // [NT][16][3]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int oct = lane & 15, sub = lane >> 4;
  for (int id = tid; id < NT * 16 * 3; id += 256) {
    const int co = id % 3, o = (id / 3) & 15, tap = id / 48;
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (co < p.cout) v = *reinterpret_cast<const h8*>(p.wt + (size_t)co * p.Kp + tap * p.ctot + o * 8);
    sW[id] = v;
  }
  __syncthreads();
  const half_t* in0 = reinterpret_cast<const half_t*>(p.in[0].ptr);
  const int s = p.stride;
  const unsigned per_img = (unsigned)(tiles_x * tiles_y), total = per_img * (unsigned)p.N;
  for (unsigned t = blockIdx.x; t < total; t += gridDim.x) {
    const int n = t / per_img;
    const unsigned r_ = t - n * per_img;
    const int oy = (int)(r_ / tiles_x) * 4 + wid, ox0 = (int)(r_ % tiles_x) * 16;
    if (oy >= p.OH) continue;
    const half_t* inn = in0 + (long)n * p.in[0].sn + oct * 8;
    // two pixel quads per pass: a tap's weights (three 16-byte LDS broadcasts per lane -- the LDS port is this kernel's busiest unit) serve both
#pragma unroll 1
    for (int q = 0; q < 4; q += 2) {
      int oxs[2], ix0s[2];
      bool lives[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { oxs[u] = ox0 + 4 * (q + u) + sub; lives[u] = oxs[u] < p.OW; ix0s[u] = oxs[u] * s - p.pad; }
      const int iy0 = oy * s - p.pad;
      float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll 1
      for (int ky = 0; ky < KS; ++ky) {
        const int iy = iy0 + ky;
        const bool rowin = (unsigned)iy < (unsigned)p.H;
        const half_t* rowp = inn + (long)iy * p.in[0].sy;
        h8 xv[2][KS];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const int ix = ix0s[u] + kx;
            xv[u][kx] = h8{0, 0, 0, 0, 0, 0, 0, 0};
            if (rowin && lives[u] && (unsigned)ix < (unsigned)p.W) xv[u][kx] = *reinterpret_cast<const h8*>(rowp + (long)ix * p.in[0].sx);
          }
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const h8* wp = sW + ((ky * KS + kx) * 16 + oct) * 3;
#pragma unroll
          for (int co = 0; co < 3; ++co) {
            const h8 w = wp[co];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const h2 wj = h2{w[2 * j], w[2 * j + 1]};
              acc[0][co] = __builtin_amdgcn_fdot2(h2{xv[0][kx][2 * j], xv[0][kx][2 * j + 1]}, wj, acc[0][co], false);
              acc[1][co] = __builtin_amdgcn_fdot2(h2{xv[1][kx][2 * j], xv[1][kx][2 * j + 1]}, wj, acc[1][co], false);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int co = 0; co < 3; ++co) {
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) acc[u][co] += __shfl_xor(acc[u][co], o, 64);
          acc[u][co] *= p.out_scale;
        }
        const int ox = oxs[u];
        if (lives[u] && oct == 0) {
          if (p.out32) {
            float* o32 = p.out32 + (long)n * p.o32_sn + (long)oy * p.o32_sy + (long)ox * p.o32_sx;
            for (int co = 0; co < p.cout; ++co) o32[co * p.o32_sc] = acc[u][co];
          }
          if (p.out16) {
            h8 hv = {0, 0, 0, 0, 0, 0, 0, 0};
            hv[0] = (half_t)acc[u][0]; hv[1] = (half_t)acc[u][1]; hv[2] = (half_t)acc[u][2];
            if (p.cout < 3) hv[2] = (half_t)0;
            if (p.cout < 2) hv[1] = (half_t)0;
            *reinterpret_cast<h8*>(p.out16 + n * p.o_sn + (long)oy * p.o_sy + (long)ox * p.o_sx) = hv;
          }
        }
      }
    }
  }
}

static int g_conv_thin_sc = 1;
void conv_thin_sc_enable(int on) { g_conv_thin_sc = on; }

bool conv_thin_sc_eligible(const ConvK& k) {
  if (!g_conv_thin_sc || k.transposed || k.dil != 1) return false;
  if (k.KHt != k.KWt || (k.KHt != 8 && k.KHt != 12) || k.stride < 2 || k.pad < 0 || k.pad >= k.stride) return false;      // 8x8 s4, 12x12 s8 (kbpn.py:22-25)
  if (k.c0 != k.ctot || k.ctot != 128 || k.in[0].sx == 0) return false;
  if (k.cout > 3 || k.coutp != 8) return false;
  if (k.bias || k.cbias || k.act != CSBSR_ACT_NONE || k.res_mode != CSBSR_RES_NONE || k.accumulate || k.mask || k.o_lo) return false;
  if (k.stat_mode != CSBSR_STAT_NONE) return false;
  if (k.OH != (k.H + 2 * k.pad - k.KHt) / k.stride + 1 || k.OW != (k.W + 2 * k.pad - k.KWt) / k.stride + 1) return false;
  if ((long)k.N * k.OH * k.OW < 64L * 1024) return false;          // small maps: the general kernel
  return true;
}

int conv_thin_sc_launch(const ConvK& k, hipStream_t st) {
  const int tiles_x = (k.OW + 15) / 16, tiles_y = (k.OH + 3) / 4;
  const long total = (long)k.N * tiles_x * tiles_y;
  const size_t smem = (size_t)k.KHt * k.KWt * 16 * 3 * 16;
  static LdsAttrOnce a8, a12;
  const unsigned g = (unsigned)(total < 2048 ? total : 2048);
  if (k.KHt == 8) {
    if (int e = csbsr_lds_attr(a8, reinterpret_cast<const void*>(conv_thin_sc_kernel<8>), (int)smem, "conv(thin strided)")) return e;
    hipLaunchKernelGGL((conv_thin_sc_kernel<8>), dim3(g), dim3(256), smem, st, k, tiles_x, tiles_y);
  } else {
    if (int e = csbsr_lds_attr(a12, reinterpret_cast<const void*>(conv_thin_sc_kernel<12>), (int)smem, "conv(thin strided)")) return e;
    hipLaunchKernelGGL((conv_thin_sc_kernel<12>), dim3(g), dim3(256), smem, st, k, tiles_x, tiles_y);
  }
  CSBSR_LAUNCH_CHECK("csbsr_conv_forward(thin strided)");
  return 0;
}
