// fp32 planar (NCHW) image kernels of the path: per-sample blur (pseudo-LR), bicubic resizes, L1 terms,
// exact EDT -> normalised signed distance map, fused boundary-combo segmentation loss.  All HBM-bound
// (3-channel images, 1-channel maps); fp32 throughout because these feed log / division / small differences.
#include "common.h"

static inline int grid_for(long work, int block = 256, int cap = 8192) {
  long b = (work + block - 1) / block;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}
#define ST(s) reinterpret_cast<hipStream_t>(s)

// ------------------------------------------------------------------------------------------- depthwise blur
// y[n,c,oy,ox] = sum_{ky,kx} x[n,c,oy*s-P+ky, ox*s-P+kx] * k[n][ky*K+kx]  (- sub[n,c,oy,ox])
// one workgroup = 16x16 outputs of one (n,c) plane; the input window is staged in LDS.
template <int K>
__global__ __launch_bounds__(256) void blur_fwd_kernel(const float* x, const float* kvec, int C, int H, int W, int OH, int OW,
                                                       int stride, const float* sub, float* y32, half_t* y16, long y16_ld) {
  extern __shared__ float sm[];
  const int P = (K - 1) / 2;
  const int TW = 15 * stride + K;          // staged window side
  float* sK = sm;                          // K*K
  float* sX = sm + K * K;                  // TW*TW
  const int plane = blockIdx.z, n = plane / C, c = plane % C;
  const int oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
  for (int i = threadIdx.x; i < K * K; i += 256) sK[i] = kvec[(long)n * K * K + i];
  const float* xp = x + (long)plane * H * W;
  const int iy0 = oy0 * stride - P, ix0 = ox0 * stride - P;
  for (int i = threadIdx.x; i < TW * TW; i += 256) {
    const int ty = i / TW, tx = i % TW;
    const int iy = iy0 + ty, ix = ix0 + tx;
    sX[i] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? xp[(long)iy * W + ix] : 0.f;
  }
  __syncthreads();
  const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy >= OH || ox >= OW) return;
  float acc = 0.f;
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int kx = 0; kx < K; ++kx) acc += sX[(ty * stride + ky) * TW + tx * stride + kx] * sK[ky * K + kx];
  const long oi = (long)plane * OH * OW + (long)oy * OW + ox;
  if (sub) acc -= sub[oi];
  if (y32) y32[oi] = acc;
  if (y16) y16[(((long)n * OH + oy) * OW + ox) * y16_ld + c] = (half_t)acc;
}

// dx[n,c,iy,ix] (+)= sum_{oy,ox} dy[n,c,oy,ox] * k[n][iy - oy*s + P][ix - ox*s + P]
template <int K>
__global__ void blur_bwd_input_kernel(const float* dy, const float* kvec, float* dx, int accumulate, int N, int C, int H, int W, int OH,
                                      int OW, int stride) {
  const int P = (K - 1) / 2;
  const long total = (long)N * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W); long t = i / W;
    const int iy = (int)(t % H); const long plane = t / H;
    const int n = (int)(plane / C);
    const float* kp = kvec + (long)n * K * K;
    const float* dyp = dy + plane * OH * OW;
    float acc = 0.f;
    // oy*s in [iy+P-K+1, iy+P]
    int oy_lo = (iy + P - K + 1 + stride - 1); oy_lo = oy_lo < 0 ? 0 : oy_lo / stride;
    int oy_hi = (iy + P) / stride; if (oy_hi > OH - 1) oy_hi = OH - 1;
    int ox_lo = (ix + P - K + 1 + stride - 1); ox_lo = ox_lo < 0 ? 0 : ox_lo / stride;
    int ox_hi = (ix + P) / stride; if (ox_hi > OW - 1) ox_hi = OW - 1;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const int ky = iy - oy * stride + P;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const int kx = ix - ox * stride + P;
        acc += dyp[(long)oy * OW + ox] * kp[ky * K + kx];
      }
    }
    dx[i] = accumulate ? dx[i] + acc : acc;
  }
}

// The stride-1 case (the SR loss's pseudo-LR blur of the whole SR batch, sr_loss_functions.py:74-88, once per step at HR): a full 21x21
// correlation with the flipped kernel, 441 MACs per pixel.  The gather form above ran it at 5 TFLOP/s (13 ms per step); here a
// workgroup owns 16 rows x 128 columns, stages the (16+20) x (128+20) window of dy and the 441 taps in LDS, and a thread produces a
// 1 x 8 strip: per kernel row one 28-float slice of the window in registers and 21 x 8 FMAs against broadcast tap reads.
// FWD: the same tiling for the forward blur at stride 1 (unflipped taps; optional subtrahend, fp32 planar and / or fp16 NHWC output)
template <int K, bool FWD>
__global__ __launch_bounds__(256) void blur_bwd_input_s1_kernel(const float* dy, const float* kvec, float* dx, int accumulate, int C, int H, int W,
                                                                int tiles_x, int tiles_y, const float* sub, half_t* y16, long y16_ld) {
  constexpr int P = (K - 1) / 2, TR = 16, TC = 128, WC = TC + K - 1, WR = TR + K - 1, WCP = WC + 4;     // 148 (+4 pad) x 36
  __shared__ __attribute__((aligned(16))) float sK[K * K + 3];
  __shared__ __attribute__((aligned(16))) float sD[WR * WCP];
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const long plane = b / tiles_y;
  const int n = (int)(plane / C);
  // dx[iy][ix] = sum dy[iy + P - ky][ix + P - kx] k[ky][kx] = sum_{a,b} dy[iy - P + a][ix - P + b] kf[a][b], kf = k flipped both ways
  for (int i = tid; i < K * K; i += 256) sK[i] = kvec[(long)n * K * K + (FWD ? i : K * K - 1 - i)];
  const float* dyp = dy + plane * H * W;
  const int y0 = ty * TR, x0 = tx * TC;
  for (int i = tid; i < WR * WC; i += 256) {
    const int r = i / WC, c = i - r * WC;
    const int yy = y0 - P + r, xx = x0 - P + c;
    sD[r * WCP + c] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? dyp[(long)yy * W + xx] : 0.f;
  }
  __syncthreads();
  const int lr = tid >> 4, lc = (tid & 15) * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll 1
  for (int a = 0; a < K; ++a) {
    float w[8 + K - 1];
    const float* row = sD + (lr + a) * WCP + lc;
#pragma unroll
    for (int j = 0; j < (8 + K - 1) / 4; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(row + 4 * j);
      w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int bb = 0; bb < K; ++bb) {
      const float t = sK[a * K + bb];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += w[bb + e] * t;
    }
  }
  const int iy = y0 + lr, ix = x0 + lc;
  if (iy >= H) return;
  const long oi = plane * H * W + (long)iy * W + ix;
  if (FWD) {
    const int c = (int)(plane - (long)n * C);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (ix + e < W) {
        float v = acc[e];
        if (sub) v -= sub[oi + e];
        if (dx) dx[oi + e] = v;
        if (y16) y16[(((long)n * H + iy) * W + ix + e) * y16_ld + c] = (half_t)v;
      }
    return;
  }
  float* o = dx + oi;
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (ix + e < W) o[e] = accumulate ? o[e] + acc[e] : acc[e];
}

// The stride-4 case of the above (the KBlock blur of the x4 model, kbpn.py:497-500, four times per micro-batch at HR), tiled: the
// gather form walked ~30 taps per HR pixel with two global loads and an integer division each (1.95 ms per launch at N = 4, 79 GB/s).
// Per output phase (iy % 4, ix % 4) the backward is a <= 6x6 convolution of the LOW-resolution dy with that phase's taps
// (ky = r + 4 j, r = (py + P) % 4), and all 16 phases of one low-resolution position read the same 7x7 window of dy.  A workgroup
// owns a 16x16 tile of low-resolution positions of one plane: dy tile (22x22, zero outside) and the 441 taps in LDS, a thread keeps
// its 7x7 window in registers and produces its 4x4 block of HR pixels (tap reads are wave-wide broadcasts), stored as four float4 rows.
template <int K>
__global__ __launch_bounds__(256) void blur_bwd_input_s4_kernel(const float* dy, const float* kvec, float* dx, int accumulate, int C, int H, int W,
                                                                int OH, int OW, int tiles_x, int tiles_y) {
  constexpr int P = (K - 1) / 2, S = 4, TL = 16, WIN = 7, TD = TL + WIN - 1;      // 22
  __shared__ float sK[K * K];
  __shared__ float sD[TD * TD];
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const long plane = b / tiles_y;
  const int n = (int)(plane / C);
  for (int i = tid; i < K * K; i += 256) sK[i] = kvec[(long)n * K * K + i];
  const float* dyp = dy + plane * OH * OW;
  const int a0 = ty * TL, b0 = tx * TL;
  for (int i = tid; i < TD * TD; i += 256) {
    const int r = i / TD, c = i - r * TD;
    const int oy = a0 - 3 + r, ox = b0 - 3 + c;
    sD[i] = ((unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW) ? dyp[(long)oy * OW + ox] : 0.f;
  }
  __syncthreads();
  const int la = tid >> 4, lb = tid & 15;
  float win[WIN][WIN];                                  // dy[a - 3 .. a + 3][b - 3 .. b + 3]
#pragma unroll
  for (int r = 0; r < WIN; ++r)
#pragma unroll
    for (int c = 0; c < WIN; ++c) win[r][c] = sD[(la + r) * TD + lb + c];
  const int a = a0 + la, bb = b0 + lb;
  if (a >= OH || bb >= OW) return;
  float* dxp = dx + plane * H * W + (long)(S * a) * W + S * bb;
#pragma unroll
  for (int py = 0; py < S; ++py) {
    constexpr int dummy = 0; (void)dummy;
    const int ry = (py + P) % S, cy = (py + P) / S;     // ky = ry + 4 j, dy row a + cy - j -> window row 3 + cy - j
    float out[S];
#pragma unroll
    for (int px = 0; px < S; ++px) {
      const int rx = (px + P) % S, cx = (px + P) / S;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int ky = ry + S * j;
        if (ky >= K) continue;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int kx = rx + S * i;
          if (kx >= K) continue;
          acc += win[3 + cy - j][3 + cx - i] * sK[ky * K + kx];
        }
      }
      out[px] = acc;
    }
    float4* o = reinterpret_cast<float4*>(dxp + (long)py * W);
    if (accumulate) { const float4 old = *o; out[0] += old.x; out[1] += old.y; out[2] += old.z; out[3] += old.w; }
    *o = make_float4(out[0], out[1], out[2], out[3]);
  }
}

// dk[n][ky*K+kx] += sum_{c,oy,ox} dy[n,c,oy,ox] * x[n,c,oy*s-P+ky, ox*s-P+kx]
// grid: (K*K taps, chunks, N); block reduces over its chunk of (c,oy,ox)
// dk[n][ky*K+kx] = sum_{c,oy,ox} dy[n,c,oy,ox] * x[n,c,oy*s-P+ky, ox*s-P+kx]
// Tiled: a workgroup stages a TOxTO tile of dy and the matching x window in LDS, every thread owns up to two taps and sweeps the
// tile (dy is a wave-wide broadcast read, the x reads of neighbouring taps are neighbouring words), and accumulates over
// ``tiles_per_wg`` tiles before writing one partial row (folded in fixed order afterwards).  (The first version walked the whole plane once per tap with two 64-bit
// divisions per element: 2.1 ms per launch at HR.)
template <int K>
__global__ __launch_bounds__(256) void blur_bwd_kernel_kernel(const float* dy, const float* x, float* dk, int C, int H, int W, int OH,
                                                              int OW, int stride, int TO, int lgTO, int tiles_x, int tiles_y,
                                                              int tiles_per_wg, float* part) {
  extern __shared__ float sm[];
  const int P = (K - 1) / 2;
  const int XW = (TO - 1) * stride + K;
  float* sDy = sm;                 // TO*TO
  float* sX = sm + TO * TO;        // XW*XW
  const int n = blockIdx.y, tid = threadIdx.x;
  const int t0 = tid, t1 = tid + 256;
  const bool v1 = t1 < K * K, v0 = t0 < K * K;
  const int off0 = v0 ? (t0 / K) * XW + (t0 % K) : 0;
  const int off1 = v1 ? (t1 / K) * XW + (t1 % K) : 0;
  float acc0 = 0.f, acc1 = 0.f;
  const int total = C * tiles_y * tiles_x;
  int tile = blockIdx.x * tiles_per_wg;
  const int tend = tile + tiles_per_wg < total ? tile + tiles_per_wg : total;
  for (; tile < tend; ++tile) {
    const int c = tile / (tiles_y * tiles_x);
    const int r = tile - c * tiles_y * tiles_x;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const float* dyp = dy + ((long)n * C + c) * OH * OW;
    const float* xp = x + ((long)n * C + c) * H * W;
    __syncthreads();
    for (int i = tid; i < TO * TO; i += 256) {
      const int oy = ty * TO + (i >> lgTO), ox = tx * TO + (i & (TO - 1));
      sDy[i] = (oy < OH && ox < OW) ? dyp[(long)oy * OW + ox] : 0.f;
    }
    const int iy0 = ty * TO * stride - P, ix0 = tx * TO * stride - P;
    for (int i = tid; i < XW * XW; i += 256) {
      const int ry = i / XW, rx = i - ry * XW;
      const int iy = iy0 + ry, ix = ix0 + rx;
      sX[i] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? xp[(long)iy * W + ix] : 0.f;
    }
    __syncthreads();
    for (int jy = 0; jy < TO; ++jy) {
      const float* xr = sX + jy * stride * XW;
      const float* dr = sDy + jy * TO;
#pragma unroll 8
      for (int jx = 0; jx < TO; ++jx) {
        const float d = dr[jx];
        acc0 += d * xr[jx * stride + off0];
        acc1 += d * xr[jx * stride + off1];
      }
    }
  }
  {                     // row per (sample, workgroup), 512 floats: folded by csbsr_sum_partials
    float* row = part + ((long)n * gridDim.x + blockIdx.x) * 512;
    row[t0] = acc0; row[t1] = acc1;
  }
}

// Stride-1 form (the loss's HR blur, sr_loss_functions.py:93; 80 % of the kernel-gradient time): a thread owns FOUR taps of one kernel row
// (ky, 4 g .. 4 g + 3) and slides a register window along the tile row, so a pixel costs one x word and the broadcast dy word per four
// FMAs instead of three words per two -- the two-taps-per-thread form above is LDS-bound at a sixth of the VALU rate (5.5 ms per config-2
// step at HR, B = 8).  126 of each 128 threads hold taps (21 rows x 6 groups; the sixth group has one tap); the two halves of the workgroup
// take the even / odd rows of the tile and meet in LDS, half 1 into half 0, before the one partial row is written.
template <int K>
__global__ __launch_bounds__(256) void blur_bwd_kernel_s1_kernel(const float* dy, const float* x, int C, int H, int W, int tiles_x, int tiles_y,
                                                                 int tiles_per_wg, float* part) {
  constexpr int TO = 32, P = (K - 1) / 2, XW = TO - 1 + K, NG = (K + 3) / 4;
  static_assert(K * NG <= 128, "one tap group per thread of a half workgroup");
  static_assert(XW % 4 == 0, "16-byte aligned window rows");
  extern __shared__ float sm[];
  float* sDy = sm;                 // TO*TO
  float* sX = sm + TO * TO;        // XW*XW (+ 4 words of slack: the last group's window runs past its row)
  __shared__ float sacc[512];
  const int n = blockIdx.y, tid = threadIdx.x;
  const int half = tid >> 7, u = tid & 127;
  const bool live = u < K * NG;
  const int ky = live ? u / NG : 0, kx0 = live ? 4 * (u % NG) : 0;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int total = C * tiles_y * tiles_x;
  int tile = blockIdx.x * tiles_per_wg;
  const int tend = tile + tiles_per_wg < total ? tile + tiles_per_wg : total;
  for (; tile < tend; ++tile) {
    const int c = tile / (tiles_y * tiles_x);
    const int r = tile - c * tiles_y * tiles_x;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const float* dyp = dy + ((long)n * C + c) * H * W;
    const float* xp = x + ((long)n * C + c) * H * W;
    __syncthreads();
    for (int i = tid; i < TO * TO; i += 256) {
      const int oy = ty * TO + (i >> 5), ox = tx * TO + (i & 31);
      sDy[i] = (oy < H && ox < W) ? dyp[(long)oy * W + ox] : 0.f;
    }
    const int iy0 = ty * TO - P, ix0 = tx * TO - P;
    for (int i = tid; i < XW * XW + 4; i += 256) {
      const int ry = i / XW, rx = i - ry * XW;
      const int iy = iy0 + ry, ix = ix0 + rx;
      sX[i] = (i < XW * XW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? xp[(long)iy * W + ix] : 0.f;
    }
    __syncthreads();
    if (live) {
      for (int jy = half; jy < TO; jy += 2) {
        // four pixels per step from two 16-byte LDS reads (x window words 4 q + 4 .. 4 q + 7, the broadcast dy quad): the window of pixel p is
        // x[p .. p + 3]; kx0 and the row pitch XW = 52 are multiples of 4, so every read is aligned (the one-word-per-pixel form was LDS-issue bound)
        const float4* xr4 = reinterpret_cast<const float4*>(sX + (jy + ky) * XW + kx0);
        const float4* dr4 = reinterpret_cast<const float4*>(sDy + jy * TO);
        float4 cur = xr4[0];
#pragma unroll
        for (int q = 0; q < TO / 4; ++q) {
          const float4 nxt = xr4[q + 1], d = dr4[q];
          a0 += d.x * cur.x; a1 += d.x * cur.y; a2 += d.x * cur.z; a3 += d.x * cur.w;
          a0 += d.y * cur.y; a1 += d.y * cur.z; a2 += d.y * cur.w; a3 += d.y * nxt.x;
          a0 += d.z * cur.z; a1 += d.z * cur.w; a2 += d.z * nxt.x; a3 += d.z * nxt.y;
          a0 += d.w * cur.w; a1 += d.w * nxt.x; a2 += d.w * nxt.y; a3 += d.w * nxt.z;
          cur = nxt;
        }
      }
    }
  }
  const int t = ky * K + kx0;
  if (live && half == 1) {
    sacc[t] = a0;
    if (kx0 + 1 < K) sacc[t + 1] = a1;
    if (kx0 + 2 < K) sacc[t + 2] = a2;
    if (kx0 + 3 < K) sacc[t + 3] = a3;
  }
  __syncthreads();
  if (live && half == 0) {                     // row per (sample, workgroup), 512 floats: folded by csbsr_sum_partials
    float* row = part + ((long)n * gridDim.x + blockIdx.x) * 512;
    row[t] = a0 + sacc[t];
    if (kx0 + 1 < K) row[t + 1] = a1 + sacc[t + 1];
    if (kx0 + 2 < K) row[t + 2] = a2 + sacc[t + 2];
    if (kx0 + 3 < K) row[t + 3] = a3 + sacc[t + 3];
  }
}

#define BLUR_DISPATCH(K, CALL) \
  switch (K) {                 \
    case 21: { constexpr int KK = 21; CALL; break; } \
    case 7: { constexpr int KK = 7; CALL; break; }   \
    case 5: { constexpr int KK = 5; CALL; break; }   \
    default: csbsr_set_error("blur: unsupported kernel size %d", K); return 1; }

extern "C" int csbsr_blur_fwd(const float* x, const float* kvec, int32_t N, int32_t C, int32_t H, int32_t W, int32_t K, int32_t stride,
                              const float* sub, float* y32, void* y16, int64_t y16_ld, csbsr_stream_t s) {
  CSBSR_CHECK(x && kvec && (y32 || y16), "blur_fwd: null");
  const int P = (K - 1) / 2;
  const int OH = (H + 2 * P - K) / stride + 1, OW = (W + 2 * P - K) / stride + 1;
  if (stride == 1 && K == 21) {      // register-blocked 16 x 128 tiles (see blur_bwd_input_s1_kernel): 5.5 -> ~2 ms at HR, B = 8
    const int tiles_x = (W + 127) / 128, tiles_y = (H + 15) / 16;
    hipLaunchKernelGGL((blur_bwd_input_s1_kernel<21, true>), dim3((unsigned)((long)N * C * tiles_y * tiles_x)), dim3(256), 0, ST(s), x, kvec, y32, 0,
                       C, H, W, tiles_x, tiles_y, sub, (half_t*)y16, (long)y16_ld);
    CSBSR_LAUNCH_CHECK("csbsr_blur_fwd");
    return 0;
  }
  const int TW = 15 * stride + K;
  const size_t smem = (size_t)(K * K + TW * TW) * 4;
  dim3 grid((OW + 15) / 16, (OH + 15) / 16, N * C);
  if (smem > 48 * 1024) {      // stride-8 windows (141^2 floats) need the opt-in above the default dynamic LDS limit
    CSBSR_CHECK(smem <= 160 * 1024, "blur_fwd: window does not fit LDS");
    BLUR_DISPATCH(K, CSBSR_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(blur_fwd_kernel<KK>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess,
                                 "blur_fwd: cannot reserve %d bytes of LDS", (int)smem));
  }
  BLUR_DISPATCH(K, hipLaunchKernelGGL((blur_fwd_kernel<KK>), grid, dim3(256), smem, ST(s), x, kvec, C, H, W, OH, OW, stride, sub, y32,
                                      (half_t*)y16, y16_ld));
  CSBSR_LAUNCH_CHECK("csbsr_blur_fwd");
  return 0;
}
extern "C" int csbsr_blur_bwd_input(const float* dy, const float* kvec, float* dx, int32_t accumulate, int32_t N, int32_t C, int32_t H,
                                    int32_t W, int32_t K, int32_t stride, csbsr_stream_t s) {
  CSBSR_CHECK(dy && kvec && dx, "blur_bwd_input: null");
  const int P = (K - 1) / 2;
  const int OH = (H + 2 * P - K) / stride + 1, OW = (W + 2 * P - K) / stride + 1;
  if (stride == 1 && K == 21) {
    const int tiles_x = (W + 127) / 128, tiles_y = (H + 15) / 16;
    hipLaunchKernelGGL((blur_bwd_input_s1_kernel<21, false>), dim3((unsigned)((long)N * C * tiles_y * tiles_x)), dim3(256), 0, ST(s), dy, kvec, dx,
                       accumulate, C, H, W, tiles_x, tiles_y, nullptr, nullptr, 0l);
    CSBSR_LAUNCH_CHECK("csbsr_blur_bwd_input");
    return 0;
  }
  if (stride == 4 && K == 21 && H == 4 * OH && W == 4 * OW && W % 4 == 0) {
    const int tiles_x = (OW + 15) / 16, tiles_y = (OH + 15) / 16;
    hipLaunchKernelGGL((blur_bwd_input_s4_kernel<21>), dim3((unsigned)((long)N * C * tiles_y * tiles_x)), dim3(256), 0, ST(s), dy, kvec, dx,
                       accumulate, C, H, W, OH, OW, tiles_x, tiles_y);
    CSBSR_LAUNCH_CHECK("csbsr_blur_bwd_input");
    return 0;
  }
  BLUR_DISPATCH(K, hipLaunchKernelGGL((blur_bwd_input_kernel<KK>), dim3(grid_for((long)N * C * H * W)), dim3(256), 0, ST(s), dy, kvec, dx,
                                      accumulate, N, C, H, W, OH, OW, stride));
  CSBSR_LAUNCH_CHECK("csbsr_blur_bwd_input");
  return 0;
}
extern "C" int csbsr_blur_bwd_kernel(const float* dy, const float* x, float* dk, int32_t N, int32_t C, int32_t H, int32_t W, int32_t K,
                                     int32_t stride, csbsr_stream_t s) {
  CSBSR_CHECK(dy && x && dk, "blur_bwd_kernel: null");
  const int P = (K - 1) / 2;
  const int OH = (H + 2 * P - K) / stride + 1, OW = (W + 2 * P - K) / stride + 1;
  CSBSR_CHECK(K * K <= 512, "blur_bwd_kernel: at most 512 taps");
  const int TO = stride == 1 ? 32 : (stride <= 4 ? 16 : 8), lgTO = TO == 32 ? 5 : (TO == 16 ? 4 : 3);
  const int XW = (TO - 1) * stride + K;
  const size_t smem = (size_t)(TO * TO + XW * XW) * sizeof(float);
  CSBSR_CHECK(smem <= 64 * 1024, "blur_bwd_kernel: window does not fit LDS");
  const int tiles_x = (OW + TO - 1) / TO, tiles_y = (OH + TO - 1) / TO;
  const int total = C * tiles_x * tiles_y;
  int tpw = (total + 1023) / 1024;          // ~1024 workgroups per sample at most: bounds the partial rows per tap
  if (tpw < 1) tpw = 1;
  dim3 grid((total + tpw - 1) / tpw, N);
  float* part = csbsr_red_scratch((long)N * grid.x * 512);
  CSBSR_NEED_SCRATCH(part, "blur_bwd_kernel");
  if (stride == 1 && K == 21 && OH == H && OW == W) {
    hipLaunchKernelGGL((blur_bwd_kernel_s1_kernel<21>), grid, dim3(256), smem + 16, ST(s), dy, x, C, H, W, tiles_x, tiles_y, tpw, part);
  } else {
    BLUR_DISPATCH(K, hipLaunchKernelGGL((blur_bwd_kernel_kernel<KK>), grid, dim3(256), smem, ST(s), dy, x, dk, C, H, W, OH, OW, stride, TO, lgTO,
                                        tiles_x, tiles_y, tpw, part));
  }
  if (csbsr_sum_partials_batched(part, (int)grid.x, 512, K * K, dk, N, (long)K * K, ST(s))) return 1;
  CSBSR_LAUNCH_CHECK("csbsr_blur_bwd_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------- bicubic
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// out[n,c,oy,ox] += bicubic upsample (align_corners=False, A=-0.75, border-clamped taps) of x
__global__ void bicubic_up_add_kernel(const float* x, float* out, int planes, int H, int W, int scale) {
  const int OH = H * scale, OW = W * scale;
  const long total = (long)planes * OH * OW;
  const float A = -0.75f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW); long t = i / OW;
    const int oy = (int)(t % OH); const long pl = t / OH;
    const float sy = (oy + 0.5f) / scale - 0.5f, sx = (ox + 0.5f) / scale - 0.5f;
    const int iy = (int)floorf(sy), ix = (int)floorf(sx);
    const float ty = sy - iy, tx = sx - ix;
    float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
    float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
    const float* xp = x + pl * H * W;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = iy - 1 + a; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
      float r = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int xx = ix - 1 + b; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
        r += wx[b] * xp[(long)yy * W + xx];
      }
      acc += wy[a] * r;
    }
    out[i] += acc;
  }
}
extern "C" int csbsr_bicubic_up_add(const float* x, float* out, int32_t planes, int32_t H, int32_t W, int32_t scale, csbsr_stream_t s) {
  CSBSR_CHECK(x && out, "bicubic_up_add: null");
  hipLaunchKernelGGL(bicubic_up_add_kernel, dim3(grid_for((long)planes * H * W * scale * scale)), dim3(256), 0, ST(s), x, out, planes, H, W,
                     scale);
  CSBSR_LAUNCH_CHECK("csbsr_bicubic_up_add");
  return 0;
}

// antialiased bicubic down-scale by integer factor f (torch _upsample_bicubic2d_aa): separable, A=-0.5,
// support 2f, weights normalised over the in-image taps.  antialias==0: 4-tap A=-0.75 rule with clamped taps.
__device__ __forceinline__ float aa_filter(float x) {
  const float A = -0.5f;
  x = fabsf(x);
  if (x < 1.f) return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * A;
  return 0.f;
}
// computes tap range [lo,hi) and normalisation for output index o
__device__ __forceinline__ void aa_range(int o, int in, int f, int& lo, int& hi, float& center, float& norm) {
  const float scale = (float)f;
  center = scale * (o + 0.5f);
  const float support = 2.f * scale;
  lo = (int)(center - support + 0.5f); if (lo < 0) lo = 0;
  hi = (int)(center + support + 0.5f); if (hi > in) hi = in;
  float tot = 0.f;
  for (int k = lo; k < hi; ++k) tot += aa_filter((k - center + 0.5f) / scale);
  norm = 1.f / tot;
}
__global__ void aa_down_fwd_kernel(const float* x, float* y, int planes, int H, int W, int f, int antialias) {
  const int OH = H / f, OW = W / f;
  const long total = (long)planes * OH * OW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW); long t = i / OW;
    const int oy = (int)(t % OH); const long pl = t / OH;
    const float* xp = x + pl * H * W;
    float acc = 0.f;
    if (antialias) {
      int ylo, yhi, xlo, xhi; float cy, cx, ny, nx;
      aa_range(oy, H, f, ylo, yhi, cy, ny); aa_range(ox, W, f, xlo, xhi, cx, nx);
      for (int yy = ylo; yy < yhi; ++yy) {
        const float wy = aa_filter((yy - cy + 0.5f) / f) * ny;
        float r = 0.f;
        for (int xx = xlo; xx < xhi; ++xx) r += aa_filter((xx - cx + 0.5f) / f) * nx * xp[(long)yy * W + xx];
        acc += wy * r;
      }
    } else {
      const float A = -0.75f;
      const float sy = (oy + 0.5f) * f - 0.5f, sx = (ox + 0.5f) * f - 0.5f;
      const int iy = (int)floorf(sy), ix = (int)floorf(sx);
      const float ty = sy - iy, tx = sx - ix;
      float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
      float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
      for (int a = 0; a < 4; ++a) {
        int yy = iy - 1 + a; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
        float r = 0.f;
        for (int b = 0; b < 4; ++b) { int xx = ix - 1 + b; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx); r += wx[b] * xp[(long)yy * W + xx]; }
        acc += wy[a] * r;
      }
    }
    y[i] = acc;
  }
}
// adjoint (gather over the outputs whose footprint covers the input pixel)
__global__ void aa_down_bwd_kernel(const float* dy, float* dx, int accumulate, int planes, int H, int W, int f, int antialias) {
  const int OH = H / f, OW = W / f;
  const long total = (long)planes * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W); long t = i / W;
    const int iy = (int)(t % H); const long pl = t / H;
    const float* gp = dy + pl * OH * OW;
    float acc = 0.f;
    int oy_lo = iy / f - 3, oy_hi = iy / f + 3, ox_lo = ix / f - 3, ox_hi = ix / f + 3;
    if (!antialias) {   // clamped taps: border input pixels collect from every output that clamps onto them
      if (iy == 0) oy_lo = 0;
      if (ix == 0) ox_lo = 0;
      if (iy == H - 1) oy_hi = OH - 1;
      if (ix == W - 1) ox_hi = OW - 1;
    }
    oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > OH - 1 ? OH - 1 : oy_hi; ox_hi = ox_hi > OW - 1 ? OW - 1 : ox_hi;
    // separable: the (at most 7) column weights of this input pixel once, then one row weight per candidate output row
    // (the first version evaluated the normalised antialias filter 7 x 8 times per pixel: 2.5 ms per call at HR)
    const bool wide = (ox_hi - ox_lo) > 6;       // clamped-border pixels of the non-antialiased resize at tiny sizes
    float wxv[7];
#pragma unroll
    for (int jx = 0; jx < 7; ++jx) {
      const int ox = ox_lo + jx;
      float wx = 0.f;
      if (ox <= ox_hi && !wide) {
        if (antialias) {
          int lo, hi; float c, nrm; aa_range(ox, W, f, lo, hi, c, nrm);
          if (ix >= lo && ix < hi) wx = aa_filter((ix - c + 0.5f) / f) * nrm;
        } else {
          const float A = -0.75f; const float sx = (ox + 0.5f) * f - 0.5f; const int bx = (int)floorf(sx); const float tx = sx - bx;
          const float w4[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
          for (int b = 0; b < 4; ++b) { int xx = bx - 1 + b; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx); if (xx == ix) wx += w4[b]; }
        }
      }
      wxv[jx] = wx;
    }
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      float wy = 0.f;
      if (antialias) {
        int lo, hi; float c, nrm; aa_range(oy, H, f, lo, hi, c, nrm);
        if (iy >= lo && iy < hi) wy = aa_filter((iy - c + 0.5f) / f) * nrm;
      } else {
        const float A = -0.75f; const float sy = (oy + 0.5f) * f - 0.5f; const int by = (int)floorf(sy); const float ty = sy - by;
        const float w4[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
        for (int a = 0; a < 4; ++a) { int yy = by - 1 + a; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy); if (yy == iy) wy += w4[a]; }
      }
      if (wy == 0.f) continue;
      if (!wide) {
#pragma unroll
        for (int jx = 0; jx < 7; ++jx)
          if (ox_lo + jx <= ox_hi && wxv[jx] != 0.f) acc += wy * wxv[jx] * gp[(long)oy * OW + ox_lo + jx];
      } else {
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {        // non-antialiased only (antialiased windows never exceed 7)
          float wx = 0.f;
          const float A = -0.75f; const float sx = (ox + 0.5f) * f - 0.5f; const int bx = (int)floorf(sx); const float tx = sx - bx;
          const float w4[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
          for (int b = 0; b < 4; ++b) { int xx = bx - 1 + b; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx); if (xx == ix) wx += w4[b]; }
          if (wx != 0.f) acc += wy * wx * gp[(long)oy * OW + ox];
        }
      }
    }
    dx[i] = accumulate ? dx[i] + acc : acc;
  }
}
// Antialiased case, table-driven: the weights are separable and depend only on the input row (column) index, so a first tiny kernel
// writes, for every input row and column, the 7 weights of its candidate outputs (i / f - 3 .. i / f + 3; zero where the output does not
// exist or its footprint misses the pixel) -- the per-pixel kernel above re-derived them with the normalisation loop of aa_range 14
// times per pixel (8.2 ms per step at HR); the main pass is then 49 FMAs per input pixel.
__global__ void aa_weight_tables_kernel(float* tab, int H, int W, int f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H + W) return;
  const bool isrow = i < H;
  const int idx = isrow ? i : i - H, in = isrow ? H : W, on = in / f;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int o = idx / f - 3 + j;
    float w = 0.f;
    if (o >= 0 && o < on) {
      int lo, hi; float c, nrm; aa_range(o, in, f, lo, hi, c, nrm);
      if (idx >= lo && idx < hi) w = aa_filter((idx - c + 0.5f) / f) * nrm;
    }
    tab[(long)i * 8 + j] = w;
  }
  tab[(long)i * 8 + 7] = 0.f;
}
__global__ void aa_down_bwd_tab_kernel(const float* dy, float* dx, const float* tab, int accumulate, int planes, int H, int W, int f) {
  const int OH = H / f, OW = W / f;
  const long total = (long)planes * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W); long t = i / W;
    const int iy = (int)(t % H); const long pl = t / H;
    const float* gp = dy + pl * OH * OW;
    const float4 y0 = *reinterpret_cast<const float4*>(tab + (long)iy * 8), y1 = *reinterpret_cast<const float4*>(tab + (long)iy * 8 + 4);
    const float4 x0 = *reinterpret_cast<const float4*>(tab + (long)(H + ix) * 8), x1 = *reinterpret_cast<const float4*>(tab + (long)(H + ix) * 8 + 4);
    const float wy[7] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z}, wx[7] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z};
    const int oy0 = iy / f - 3, ox0 = ix / f - 3;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      int oy = oy0 + j; oy = oy < 0 ? 0 : (oy > OH - 1 ? OH - 1 : oy);        // (weights of non-existent outputs are zero)
      float r = 0.f;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        int ox = ox0 + k; ox = ox < 0 ? 0 : (ox > OW - 1 ? OW - 1 : ox);
        r += wx[k] * gp[(long)oy * OW + ox];
      }
      acc += wy[j] * r;
    }
    dx[i] = accumulate ? dx[i] + acc : acc;
  }
}
extern "C" int csbsr_aa_bicubic_down_fwd(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t f, int32_t antialias,
                                         csbsr_stream_t s) {
  CSBSR_CHECK(x && y && f >= 1, "aa_down_fwd: bad args");
  hipLaunchKernelGGL(aa_down_fwd_kernel, dim3(grid_for((long)planes * (H / f) * (W / f))), dim3(256), 0, ST(s), x, y, planes, H, W, f, antialias);
  CSBSR_LAUNCH_CHECK("csbsr_aa_bicubic_down_fwd");
  return 0;
}
extern "C" int csbsr_aa_bicubic_down_bwd(const float* dy, float* dx, int32_t accumulate, int32_t planes, int32_t H, int32_t W, int32_t f,
                                         int32_t antialias, csbsr_stream_t s) {
  CSBSR_CHECK(dy && dx && f >= 1, "aa_down_bwd: bad args");
  float* tab = antialias ? csbsr_red_scratch((long)(H + W) * 8) : nullptr;
  if (tab) {
    hipLaunchKernelGGL(aa_weight_tables_kernel, dim3((H + W + 255) / 256), dim3(256), 0, ST(s), tab, H, W, f);
    hipLaunchKernelGGL(aa_down_bwd_tab_kernel, dim3(grid_for((long)planes * H * W)), dim3(256), 0, ST(s), dy, dx, tab, accumulate, planes, H, W, f);
    CSBSR_LAUNCH_CHECK("csbsr_aa_bicubic_down_bwd");
    return 0;
  }
  hipLaunchKernelGGL(aa_down_bwd_kernel, dim3(grid_for((long)planes * H * W)), dim3(256), 0, ST(s), dy, dx, accumulate, planes, H, W, f, antialias);
  CSBSR_LAUNCH_CHECK("csbsr_aa_bicubic_down_bwd");
  return 0;
}

// fp32 single-plane bilinear resize (aux head, align_corners=True: pspnet.py:122) and its adjoint
__device__ __forceinline__ void bil_src32(int o, int in, int out, int align, int& i0, int& i1, float& w1) {
  float src;
  if (align) src = out > 1 ? o * (float)(in - 1) / (float)(out - 1) : 0.f;
  else { src = (o + 0.5f) * ((float)in / (float)out) - 0.5f; if (src < 0.f) src = 0.f; }
  i0 = (int)src; if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  w1 = src - i0;
}
__global__ void bilinear32_fwd_kernel(const float* x, float* y, int planes, int H, int W, int OH, int OW, int align) {
  const long total = (long)planes * OH * OW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW); long t = i / OW;
    const int oy = (int)(t % OH); const long pl = t / OH;
    int y0, y1, x0, x1; float wy, wx;
    bil_src32(oy, H, OH, align, y0, y1, wy); bil_src32(ox, W, OW, align, x0, x1, wx);
    const float* xp = x + pl * H * W;
    y[i] = (1.f - wy) * ((1.f - wx) * xp[(long)y0 * W + x0] + wx * xp[(long)y0 * W + x1]) +
           wy * ((1.f - wx) * xp[(long)y1 * W + x0] + wx * xp[(long)y1 * W + x1]);
  }
}
__global__ void bilinear32_bwd_kernel(const float* dy, float* dx, int planes, int H, int W, int OH, int OW, int align) {
  const long total = (long)planes * H * W;
  // output-per-input ratio of the source mapping (align_corners: (out-1)/(in-1))
  const float ry = (align && H > 1) ? (float)(OH - 1) / (H - 1) : (float)OH / H, rx = (align && W > 1) ? (float)(OW - 1) / (W - 1) : (float)OW / W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W); long t = i / W;
    const int iy = (int)(t % H); const long pl = t / H;
    int oy_lo = (int)floorf((iy - 1.5f) * ry) - 2, oy_hi = (int)ceilf((iy + 1.5f) * ry) + 2;
    int ox_lo = (int)floorf((ix - 1.5f) * rx) - 2, ox_hi = (int)ceilf((ix + 1.5f) * rx) + 2;
    if (iy == 0) oy_lo = 0;
    if (ix == 0) ox_lo = 0;
    if (iy == H - 1) oy_hi = OH - 1;
    if (ix == W - 1) ox_hi = OW - 1;
    oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > OH - 1 ? OH - 1 : oy_hi; ox_hi = ox_hi > OW - 1 ? OW - 1 : ox_hi;
    const float* gp = dy + pl * OH * OW;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float wy; bil_src32(oy, H, OH, align, y0, y1, wy);
      float cy = 0.f;
      if (y0 == iy) cy += 1.f - wy;
      if (y1 == iy) cy += wy;
      if (cy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float wx; bil_src32(ox, W, OW, align, x0, x1, wx);
        float cx = 0.f;
        if (x0 == ix) cx += 1.f - wx;
        if (x1 == ix) cx += wx;
        if (cx != 0.f) acc += cy * cx * gp[(long)oy * OW + ox];
      }
    }
    dx[i] = acc;
  }
}
extern "C" int csbsr_bilinear32_fwd(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t align,
                                    csbsr_stream_t s) {
  CSBSR_CHECK(x && y, "bilinear32_fwd: null");
  hipLaunchKernelGGL(bilinear32_fwd_kernel, dim3(grid_for((long)planes * OH * OW)), dim3(256), 0, ST(s), x, y, planes, H, W, OH, OW, align);
  CSBSR_LAUNCH_CHECK("csbsr_bilinear32_fwd");
  return 0;
}
extern "C" int csbsr_bilinear32_bwd(const float* dy, float* dx, int32_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t align,
                                    csbsr_stream_t s) {
  CSBSR_CHECK(dy && dx, "bilinear32_bwd: null");
  hipLaunchKernelGGL(bilinear32_bwd_kernel, dim3(grid_for((long)planes * H * W)), dim3(256), 0, ST(s), dy, dx, planes, H, W, OH, OW, align);
  CSBSR_LAUNCH_CHECK("csbsr_bilinear32_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------- L1 terms
// sums[n] += sum_{c,h,w} w(n,h,w) |a-b| ;  da (+)= gscale * w * sign(a-b)      (planes of one sample contiguous)
__global__ __launch_bounds__(256) void l1_kernel(const float* a, const float* b, const float* wmap, int C, long hw, float* sums,
                                                 float gscale, const float* gs_n, float* da, int da_acc, int chunks) {
  __shared__ float sred[4];
  const int n = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const long total = (long)C * hw;
  const long per = (total + chunks - 1) / chunks;
  const long beg = chunk * per, end = beg + per < total ? beg + per : total;
  float acc = 0.f;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    const long gi = (long)n * total + i;
    const float d = a[gi] - b[gi];
    const float w = wmap ? wmap[(long)n * hw + i % hw] : 1.f;
    acc += w * fabsf(d);
    if (da) {
      const float g = gscale * (gs_n ? gs_n[n] : 1.f) * w * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
      da[gi] = da_acc ? da[gi] + g : g;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && sums) sums[blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];      // partial per (sample, chunk)
}
extern "C" int csbsr_l1_fwd_bwd(const float* a, const float* b, const float* wmap, int32_t N, int32_t C, int64_t hw, float* sums,
                                float gscale, const float* gs_n, float* da, int32_t da_accumulate, csbsr_stream_t s) {
  CSBSR_CHECK(a && b, "l1: null");
  long total = (long)C * hw;
  int chunks = (int)((total + 65535) / 65536);
  if (chunks < 1) chunks = 1;
  float* part = nullptr;
  if (sums) {
    part = csbsr_red_scratch((long)N * chunks);
    CSBSR_NEED_SCRATCH(part, "l1");
  }
  hipLaunchKernelGGL(l1_kernel, dim3(N * chunks), dim3(256), 0, ST(s), a, b, wmap, C, hw, part, gscale, gs_n, da, da_accumulate, chunks);
  if (sums && csbsr_sum_partials_batched(part, chunks, 1, 1, sums, N, 1, ST(s))) return 1;
  CSBSR_LAUNCH_CHECK("csbsr_l1_fwd_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------- segmentation loss
// sums[n][0..5] = sum bce_elem, sum p*t, sum p*p, sum t*t, sum p*sdf, (unused)
__global__ __launch_bounds__(256) void segloss_reduce_kernel(const float* p, const float* t, const float* sdf, long hw, float* sums,
                                                             float pw0, float pw1, int chunks) {
  __shared__ float sred[4][5];
  const int n = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const long per = (hw + chunks - 1) / chunks;
  const long beg = chunk * per, end = beg + per < hw ? beg + per : hw;
  float a[5] = {0, 0, 0, 0, 0};
  const float sm = 1e-8f;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    const long gi = (long)n * hw + i;
    const float pc = fmaxf(p[gi], sm), tv = t[gi];
    a[0] += -(pw0 * tv * logf(pc + sm) + pw1 * (1.f - tv) * logf(1.f - pc + sm));
    a[1] += pc * tv; a[2] += pc * pc; a[3] += tv * tv; a[4] += pc * sdf[gi];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const float v = wave_sum(a[k]);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5)       // partial row per (sample, chunk)
    sums[(long)blockIdx.x * 8 + threadIdx.x] = sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x];
}
// loss[n] += weight * L ;  dp = gscale[n] * weight * dL/dp   where
// L = alpha * (lw0*bce_mean + lw1*dice)/(lw0+lw1) + (1-alpha) * mean(p*sdf)
__global__ void segloss_finish_kernel(const float* p, const float* t, const float* sdf, int N, long hw, const float* sums, float alpha,
                                      float pw0, float pw1, float lw0, float lw1, float weight, const float* gscale, float* loss,
                                      float* dp, int dp_acc) {
  const float sm = 1e-8f, dsm = 1e-6f;
  const float pws = pw0 + pw1, lws = lw0 + lw1;
  const long total = (long)N * hw;
  for (long gi = (long)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (long)gridDim.x * blockDim.x) {
    const int n = (int)(gi / hw);
    const float* S = sums + n * 8;
    const float num = 2.f * S[1] + dsm, den = S[2] + S[3] + dsm;
    if (gi % hw == 0 && loss) {
      const float bce = S[0] / pws / hw;
      const float dice = 1.f - num / den;
      const float L = alpha * (lw0 * bce + lw1 * dice) / lws + (1.f - alpha) * S[4] / hw;
      loss[n] += weight * L;          // exactly one thread per sample and launch
    }
    if (dp) {
      const float praw = p[gi], tv = t[gi];
      float g = 0.f;
      if (praw >= sm) {   // clamp(min=smooth) passes gradient only where p >= smooth
        const float pc = praw;
        const float dbce = -(pw0 * tv / (pc + sm) - pw1 * (1.f - tv) / (1.f - pc + sm)) / pws / hw;
        const float ddice = -(2.f * tv * den - num * 2.f * pc) / (den * den);
        g = alpha * (lw0 * dbce + lw1 * ddice) / lws + (1.f - alpha) * sdf[gi] / hw;
      }
      g *= weight * (gscale ? gscale[n] : 1.f);
      dp[gi] = dp_acc ? dp[gi] + g : g;
    }
  }
}
extern "C" int csbsr_segloss_reduce(const float* p, const float* t, const float* sdf, int32_t N, int64_t hw, float* sums, float pw0,
                                    float pw1, csbsr_stream_t s) {
  CSBSR_CHECK(p && t && sdf && sums, "segloss_reduce: null");
  int chunks = (int)((hw + 65535) / 65536);
  if (chunks < 1) chunks = 1;
  float* part = csbsr_red_scratch((long)N * chunks * 8);
  CSBSR_NEED_SCRATCH(part, "segloss_reduce");
  hipLaunchKernelGGL(segloss_reduce_kernel, dim3(N * chunks), dim3(256), 0, ST(s), p, t, sdf, (long)hw, part, pw0, pw1, chunks);
  if (csbsr_sum_partials_batched(part, chunks, 8, 5, sums, N, 8, ST(s))) return 1;
  CSBSR_LAUNCH_CHECK("csbsr_segloss_reduce");
  return 0;
}
extern "C" int csbsr_segloss_finish(const float* p, const float* t, const float* sdf, int32_t N, int64_t hw, const float* sums, float alpha,
                                    float pw0, float pw1, float lw0, float lw1, float weight, const float* gscale, float* loss, float* dp,
                                    int32_t dp_accumulate, csbsr_stream_t s) {
  CSBSR_CHECK(p && t && sdf && sums, "segloss_finish: null");
  hipLaunchKernelGGL(segloss_finish_kernel, dim3(grid_for((long)N * hw)), dim3(256), 0, ST(s), p, t, sdf, N, (long)hw, sums, alpha, pw0, pw1,
                     lw0, lw1, weight, gscale, loss, dp, dp_accumulate);
  CSBSR_LAUNCH_CHECK("csbsr_segloss_finish");
  return 0;
}

// d(pre-sigmoid) = dp * p * (1-p) as channel 0 of an fp16 NHWC8 tensor (other channels zero)
__global__ void sigmoid_bwd_to_nhwc8_kernel(const float* dp, const float* p, half_t* out, long npix, float scale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    h8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    const float pv = p[i];
    o[0] = (half_t)(dp[i] * pv * (1.f - pv) * scale);
    *reinterpret_cast<h8*>(out + i * 8) = o;
  }
}
extern "C" int csbsr_sigmoid_bwd_to_nhwc8(const float* dp, const float* p, void* out, int64_t npix, float scale, csbsr_stream_t s) {
  CSBSR_CHECK(dp && p && out, "sigmoid_bwd: null");
  hipLaunchKernelGGL(sigmoid_bwd_to_nhwc8_kernel, dim3(grid_for(npix)), dim3(256), 0, ST(s), dp, p, (half_t*)out, (long)npix, scale);
  CSBSR_LAUNCH_CHECK("csbsr_sigmoid_bwd_to_nhwc8");
  return 0;
}

// ---- 1-channel heads (PSPNet's final 1x1 conv 64 -> 1 + sigmoid at full resolution, pspnet.py:117-121; its aux head 256 -> 1): the
// general kernels pad the single output channel to a 32-row MFMA tile.  Forward: a pixel's C channels (an fp16 hi plane, optionally + the lo
// plane of a split map) against the fp32 weight row in fp32 VALU arithmetic -- LPP = C / 8 lanes per pixel, 16 bytes per lane and plane,
// xor-shuffle fold -- then bias and sigmoid, fp32 planar output: one pass over the input at the streaming rate, and MORE exact than the
// split products it replaces (the weights are not rounded at all).  Input gradient: dX[pixel][c] = dPre[pixel] w[c], one 16-byte store per lane.
template <int LPP>
__global__ __launch_bounds__(256) void head1_fwd_kernel(const half_t* __restrict__ x, long ld, long lo, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int sigmoid, float* __restrict__ out, long npix) {
  const int lane = threadIdx.x & (LPP - 1);
  float wr[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) wr[e] = w[8 * lane + e];
  const float b = bias ? bias[0] : 0.f;
  const long ppb = 256 / LPP;
  for (long px = (long)blockIdx.x * ppb + threadIdx.x / LPP; px < npix; px += (long)gridDim.x * ppb) {
    const half_t* xp = x + px * ld + 8 * lane;
    const h8 hv = *reinterpret_cast<const h8*>(xp);
    float acc = 0.f;
    if (lo) {
      const h8 lv = *reinterpret_cast<const h8*>(xp + lo);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += ((float)hv[e] + (float)lv[e]) * wr[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += (float)hv[e] * wr[e];
    }
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
      const float t = acc + b;
      out[px] = sigmoid ? 1.f / (1.f + __expf(-t)) : t;
    }
  }
}
extern "C" int csbsr_head1_fwd(const void* x, int64_t ld, int64_t lo, int32_t c, const float* w, const float* bias, int32_t sigmoid,
                               float* out, int64_t npix, csbsr_stream_t s) {
  CSBSR_CHECK(x && w && out && (c == 64 || c == 128 || c == 256) && ld >= c, "head1_fwd: bad arguments (64, 128 or 256 channels)");
  const half_t* xp = reinterpret_cast<const half_t*>(x);
  const int lpp = c / 8;
  const unsigned g = (unsigned)grid_for(npix * lpp);
  if (lpp == 8) hipLaunchKernelGGL(head1_fwd_kernel<8>, dim3(g), dim3(256), 0, ST(s), xp, (long)ld, (long)lo, w, bias, sigmoid, out, (long)npix);
  else if (lpp == 16) hipLaunchKernelGGL(head1_fwd_kernel<16>, dim3(g), dim3(256), 0, ST(s), xp, (long)ld, (long)lo, w, bias, sigmoid, out, (long)npix);
  else hipLaunchKernelGGL(head1_fwd_kernel<32>, dim3(g), dim3(256), 0, ST(s), xp, (long)ld, (long)lo, w, bias, sigmoid, out, (long)npix);
  CSBSR_LAUNCH_CHECK("csbsr_head1_fwd");
  return 0;
}
__global__ void head1_bwd_input_kernel(const half_t* __restrict__ dpre, long dpre_ld, const float* __restrict__ w, int c8, half_t* __restrict__ dx,
                                       long ld, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long px = i / c8;
    const int oct = (int)(i - px * c8);
    const float d = (float)dpre[px * dpre_ld];
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)(d * w[8 * oct + e]);
    *reinterpret_cast<h8*>(dx + px * ld + 8 * oct) = o;
  }
}
extern "C" int csbsr_head1_bwd_input(const void* dpre, int64_t dpre_ld, const float* w, int32_t c, void* dx, int64_t ld, int64_t npix,
                                     csbsr_stream_t s) {
  CSBSR_CHECK(dpre && w && dx && c % 8 == 0 && c >= 8 && ld >= c, "head1_bwd_input: bad arguments");
  const long total = (long)npix * (c / 8);
  hipLaunchKernelGGL(head1_bwd_input_kernel, dim3(grid_for(total)), dim3(256), 0, ST(s), reinterpret_cast<const half_t*>(dpre), (long)dpre_ld, w,
                     c / 8, reinterpret_cast<half_t*>(dx), (long)ld, total);
  CSBSR_LAUNCH_CHECK("csbsr_head1_bwd_input");
  return 0;
}

// ------------------------------------------------------------------------------------------- exact EDT / SDF
// Squared Euclidean distance to the nearest zero pixel of `img` (distance_transform_edt semantics) by the exact
// two-pass method: (1) per column, nearest zero above/below (1-D scan); (2) per row, lower envelope of parabolas
// evaluated by brute force over the row held in LDS (W <= 4096).  inv = 1 computes it for the complement.
__global__ void edt_cols_kernel(const float* mask, float* g, int N, int H, int W, int inv) {
  const long total = (long)N * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W); const long n = i / W;
    const float* mp = mask + n * H * W + x;
    float* gp = g + n * H * W + x;
    const float BIG = 1e9f;
    float d = BIG;
    for (int y = 0; y < H; ++y) {
      const bool fg = (mp[(long)y * W] != 0.f) != (inv != 0);
      d = fg ? d + 1.f : 0.f;
      gp[(long)y * W] = d;
    }
    d = BIG;
    for (int y = H - 1; y >= 0; --y) {
      const bool fg = (mp[(long)y * W] != 0.f) != (inv != 0);
      d = fg ? d + 1.f : 0.f;
      const float cur = gp[(long)y * W];
      gp[(long)y * W] = cur < d ? cur : d;
    }
  }
}
// Row pass of the exact EDT: dist(x) = sqrt(min over x' of g(x')^2 + (x - x')^2), g = the column pass's vertical distance.  Every sum is an
// integer below 2^24, so the minimum is exact whatever the search order.  Round 6: the search goes block by block (32 columns) with the
// block minima of g^2 as lower bounds -- a block whose bound  bmin + gap^2  cannot beat the best candidate so far is skipped -- instead of
// expanding column by column until r^2 >= best: with the sparse crack masks of full-size images (0.4 % foreground, nearest crack hundreds of
// pixels away) the column-by-column form cost 5.8 ms per config-2 step (1.5 ms on the periodic masks of the earlier bench data).
__global__ __launch_bounds__(256) void edt_rows_kernel(const float* g, float* dist, int H, int W, float* minmax /*[N][2] as ordered ints*/) {
  extern __shared__ float srow[];      // [W] squares, then [nb] block minima
  __shared__ float smx[4];
  const long row = blockIdx.x;       // n*H + y
  const float* gp = g + row * W;
  const int nb = (W + 31) >> 5;
  float* bmin = srow + W;
  for (int x = threadIdx.x; x < W; x += 256) { const float v = gp[x]; srow[x] = v >= 1e8f ? 1e18f : v * v; }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += 256) {
    float m = 1e18f;
    const int x1 = min(W, 32 * b + 32);
    for (int x = 32 * b; x < x1; ++x) m = fminf(m, srow[x]);
    bmin[b] = m;
  }
  __syncthreads();
  float mx = 0.f;
  for (int x = threadIdx.x; x < W; x += 256) {
    float best = srow[x];
    const int bx = x >> 5;
    auto scan = [&](int b) {
      const int x1 = min(W, 32 * b + 32);
      for (int xx = 32 * b; xx < x1; ++xx) {
        const float d = (float)(x - xx);
        best = fminf(best, srow[xx] + d * d);
      }
    };
    if (bmin[bx] < best) scan(bx);
    for (int d = 1; d < nb; ++d) {
      const int bl = bx - d, br = bx + d;
      const float gl = (float)(x - (32 * bl + 31)), gr = (float)(32 * br - x);      // gap to the nearest column of the block (>= 1)
      const bool lin = bl >= 0 && gl * gl < best, rin = br < nb && gr * gr < best;
      if (!lin && !rin) break;      // both sides out of range or out of reach: the gaps only grow and ``best`` only shrinks
      if (lin && bmin[bl] + gl * gl < best) scan(bl);
      if (rin && bmin[br] + gr * gr < best) scan(br);
    }
    const float dv = sqrtf(best);
    dist[row * W + x] = dv;
    mx = fmaxf(mx, dv);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    atomicMax(reinterpret_cast<int*>(minmax + (row / H)), __float_as_int(mx));   // non-negative floats order as ints
  }
}
// sdf = negdis/max(negdis) - posdis/max(posdis), inner boundary -> 0, all-background sample -> 0
// (min of each distance map is 0 whenever the mask has both classes; compute_sdf1_1 only runs if posmask.any())
__global__ void sdf_combine_kernel(const float* mask, const float* posdis, const float* negdis, const float* pmax, const float* nmax,
                                   float* sdf, int N, int H, int W) {
  const long total = (long)N * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W); long t = i / W;
    const int y = (int)(t % H); const long n = t / H;
    const float pm = pmax[n], nm = nmax[n];
    float v = 0.f;
    if (pm > 0.f) {     // posmask.any()
      const float* mp = mask + n * H * W;
      const bool fg = mp[(long)y * W + x] != 0.f;
      // nm == 0: mask is all foreground -> negdis == 0 everywhere and the reference divides 0/0 (NaN); mirror with 0
      const float nterm = nm > 0.f ? negdis[i] / nm : 0.f;
      v = nterm - posdis[i] / pm;
      if (fg) {
        const bool up = mp[(long)(y > 0 ? y - 1 : y) * W + x] != 0.f, dn = mp[(long)(y < H - 1 ? y + 1 : y) * W + x] != 0.f;
        const bool lf = mp[(long)y * W + (x > 0 ? x - 1 : x)] != 0.f, rt = mp[(long)y * W + (x < W - 1 ? x + 1 : x)] != 0.f;
        if (!(up && dn && lf && rt)) v = 0.f;
      }
    }
    sdf[i] = v;
  }
}
extern "C" int csbsr_sdf(const float* mask, float* sdf, float* scratch /*3*N*H*W + 2*N floats*/, int32_t N, int32_t H, int32_t W,
                         csbsr_stream_t s) {
  CSBSR_CHECK(mask && sdf && scratch, "sdf: null");
  CSBSR_CHECK(W <= 8192, "sdf: row longer than 8192 not supported");
  const long npx = (long)N * H * W;
  float* g = scratch; float* posdis = scratch + npx; float* negdis = scratch + 2 * npx; float* mm = scratch + 3 * npx;
  hipStream_t st = ST(s);
  CSBSR_CHECK(hipMemsetAsync(mm, 0, 2 * N * sizeof(float), st) == hipSuccess, "sdf: memset failed");
  hipLaunchKernelGGL(edt_cols_kernel, dim3(grid_for((long)N * W, 64)), dim3(64), 0, st, mask, g, N, H, W, 0);
  hipLaunchKernelGGL(edt_rows_kernel, dim3(N * H), dim3(256), (W + (W + 31) / 32) * sizeof(float), st, g, posdis, H, W, mm);
  hipLaunchKernelGGL(edt_cols_kernel, dim3(grid_for((long)N * W, 64)), dim3(64), 0, st, mask, g, N, H, W, 1);
  hipLaunchKernelGGL(edt_rows_kernel, dim3(N * H), dim3(256), (W + (W + 31) / 32) * sizeof(float), st, g, negdis, H, W, mm + N);
  hipLaunchKernelGGL(sdf_combine_kernel, dim3(grid_for(npx)), dim3(256), 0, st, mask, posdis, negdis, mm, mm + N, sdf, N, H, W);
  CSBSR_LAUNCH_CHECK("csbsr_sdf");
  return 0;
}
