// Device-side kernels of the data path either side of the hot path (SURVEY.md section 8 "next" rows f1, f2, f4):
//   * anisotropic-Gaussian blur-kernel synthesis for the degradation batch generator  (model/data/blur/blur.py:121-167),
//   * the 99-threshold IoU sweep of the evaluation loop                               (model/engine/inference.py:50-53,111-119),
//   * PSNR / SSIM of a batch of images                                               (model/utils/estimate_metrics.py:89-101,135-191).
// All fp32; HBM-bound single passes.
#include "common.h"

#define ST(s) reinterpret_cast<hipStream_t>(s)

// ------------------------------------------------------------------------------------------- Gaussian blur kernels
// out[n][y][x] = exp(-(a x'^2 + 2 b x' y' + c y'^2)) / sum,  x', y' = linspace(-K/2, K/2, K)  -- GaussianBlur.make().
// params[n] = (sigma_x, sigma_y, theta [rad]).  One workgroup per sample; the exponent is evaluated in fp64 like the reference's numpy.
__global__ __launch_bounds__(256) void gauss_kernels_kernel(const float* params, float* out, int K) {
  __shared__ double sred[256];
  const int n = blockIdx.x, tid = threadIdx.x;
  const double sx = params[n * 3 + 0], sy = params[n * 3 + 1], th = params[n * 3 + 2];
  const double ct = cos(th), st = sin(th);
  const double sx2 = 2.0 * sx * sx, sy2 = 2.0 * sy * sy;
  const double a = ct * ct / sx2 + st * st / sy2, b = st * ct * (1.0 / sy2 - 1.0 / sx2), c = st * st / sx2 + ct * ct / sy2;
  const int rad = K / 2;
  const double step = K > 1 ? 2.0 * rad / (K - 1) : 0.0;
  double part = 0.0;
  for (int i = tid; i < K * K; i += 256) {
    const double x = -rad + step * (i % K), y = -rad + step * (i / K);
    part += exp(-(a * x * x + 2.0 * b * x * y + c * y * y));
  }
  sred[tid] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sred[tid] += sred[tid + o];
    __syncthreads();
  }
  const double inv = 1.0 / sred[0];
  for (int i = tid; i < K * K; i += 256) {
    const double x = -rad + step * (i % K), y = -rad + step * (i / K);
    out[(long)n * K * K + i] = (float)(exp(-(a * x * x + 2.0 * b * x * y + c * y * y)) * inv);
  }
}
extern "C" int csbsr_gaussian_kernels(const float* params, float* out, int32_t N, int32_t K, csbsr_stream_t s) {
  CSBSR_CHECK(params && out && N > 0 && K > 0, "gaussian_kernels: bad args");
  hipLaunchKernelGGL(gauss_kernels_kernel, dim3(N), dim3(256), 0, ST(s), params, out, K);
  CSBSR_LAUNCH_CHECK("csbsr_gaussian_kernels");
  return 0;
}

// ------------------------------------------------------------------------------------------- threshold sweep
// For every sample and every threshold t_i (ascending):  inter[n][i] = #{mask & pred > t_i},  uni[n][i] = #{mask | pred > t_i}.
// One pass: a pixel's pred exceeds exactly the first k thresholds (binary search on the fp32 thresholds the caller built, so the
// comparison is bit-identical to `pred - t > 0`), and lands in bin k of the foreground or the background histogram of its sample;
// suffix sums of the two histograms are the counts.  hist[n][2][T+1] zeroed by the caller; ths in constant-ish global memory.
__global__ __launch_bounds__(256) void iou_hist_kernel(const float* pred, const float* mask, const float* ths, int T, long hw, int chunks,
                                                       unsigned* hist) {
  extern __shared__ unsigned sh[];        // [2][T+1] then the thresholds
  float* sth = reinterpret_cast<float*>(sh + 2 * (T + 1));
  const int n = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  for (int i = threadIdx.x; i < 2 * (T + 1); i += 256) sh[i] = 0;
  for (int i = threadIdx.x; i < T; i += 256) sth[i] = ths[i];
  __syncthreads();
  const long per = (hw + chunks - 1) / chunks;
  const long beg = chunk * per, end = beg + per < hw ? beg + per : hw;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    const float p = pred[(long)n * hw + i];
    int lo = 0, hi = T;                  // k = number of thresholds strictly below p
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (p - sth[mid] > 0.f) lo = mid + 1; else hi = mid; }
    atomicAdd(&sh[(mask[(long)n * hw + i] > 0.5f ? 0 : T + 1) + lo], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * (T + 1); i += 256)
    if (sh[i]) atomicAdd(&hist[(long)n * 2 * (T + 1) + i], sh[i]);
}
__global__ void iou_finish_kernel(const unsigned* hist, int T, float smooth, float* iou, float* inter_out, float* union_out) {
  const int n = blockIdx.x;
  const unsigned* hf = hist + (long)n * 2 * (T + 1);
  const unsigned* hb = hf + (T + 1);
  if (threadIdx.x != 0) return;
  unsigned long pos = 0;
  for (int k = 0; k <= T; ++k) pos += hf[k];
  unsigned long sf = 0, sb = 0;          // suffix sums: pixels exceeding at least i+1 thresholds
  for (int i = T - 1; i >= 0; --i) {
    sf += hf[i + 1]; sb += hb[i + 1];
    const double inter = (double)sf, uni = (double)pos + (double)sb;
    iou[(long)n * T + i] = (float)((inter + smooth) / (uni + smooth));
    if (inter_out) inter_out[(long)n * T + i] = (float)inter;
    if (union_out) union_out[(long)n * T + i] = (float)uni;
  }
}
extern "C" int csbsr_iou_sweep(const float* pred, const float* mask, const float* thresholds, int32_t N, int64_t hw, int32_t T, float smooth,
                               uint32_t* hist /* [N][2][T+1] zeroed */, float* iou /* [N][T] */, float* inter, float* uni, csbsr_stream_t s) {
  CSBSR_CHECK(pred && mask && thresholds && hist && iou && T > 0 && T <= 1024, "iou_sweep: bad args");
  int chunks = (int)((hw + 65535) / 65536);
  if (chunks < 1) chunks = 1;
  const size_t sm = (size_t)(2 * (T + 1)) * 4 + (size_t)T * 4;
  hipLaunchKernelGGL(iou_hist_kernel, dim3(N * chunks), dim3(256), sm, ST(s), pred, mask, thresholds, T, (long)hw, chunks, hist);
  hipLaunchKernelGGL(iou_finish_kernel, dim3(N), dim3(64), 0, ST(s), hist, T, smooth, iou, inter, uni);
  CSBSR_LAUNCH_CHECK("csbsr_iou_sweep");
  return 0;
}

// ------------------------------------------------------------------------------------------- PSNR / SSIM
// sums[n][0] += sum (a-b)^2 ; sums[n][1] += sum ssim_map   over (C, H, W).  SSIM: 11x11 Gaussian window (sigma 1.5), zero padding,
// depthwise, C1 = 0.01^2, C2 = 0.03^2 (estimate_metrics.py:135-162).  Tile 32 x 8 outputs per workgroup, halo tile in LDS, the five
// windowed moments by a separable pass (rows then columns).
#define SS_W 11
#define SS_R 5
#define SS_TX 32
#define SS_TY 8
__global__ __launch_bounds__(256) void psnr_ssim_kernel(const float* a, const float* b, int C, int H, int W, float* sums) {
  __shared__ float sa[SS_TY + 2 * SS_R][SS_TX + 2 * SS_R], sb[SS_TY + 2 * SS_R][SS_TX + 2 * SS_R];
  __shared__ float hm[5][SS_TY + 2 * SS_R][SS_TX];            // row-filtered moments
  __shared__ float g[SS_W];
  __shared__ float sred[4][2];
  const int tid = threadIdx.x;
  const int plane = blockIdx.z;                                // n * C + c
  const int n = plane / C;
  const int x0 = blockIdx.x * SS_TX, y0 = blockIdx.y * SS_TY;
  if (tid < SS_W) {
    float s = 0.f;
    for (int i = 0; i < SS_W; ++i) s += expf(-(float)((i - SS_R) * (i - SS_R)) / (2.f * 1.5f * 1.5f));
    g[tid] = expf(-(float)((tid - SS_R) * (tid - SS_R)) / (2.f * 1.5f * 1.5f)) / s;
  }
  const float* pa = a + (long)plane * H * W;
  const float* pb = b + (long)plane * H * W;
  for (int i = tid; i < (SS_TY + 2 * SS_R) * (SS_TX + 2 * SS_R); i += 256) {
    const int ly = i / (SS_TX + 2 * SS_R), lx = i % (SS_TX + 2 * SS_R);
    const int y = y0 + ly - SS_R, x = x0 + lx - SS_R;
    const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    sa[ly][lx] = in ? pa[(long)y * W + x] : 0.f;
    sb[ly][lx] = in ? pb[(long)y * W + x] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < (SS_TY + 2 * SS_R) * SS_TX; i += 256) {
    const int ly = i / SS_TX, lx = i % SS_TX;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
    for (int k = 0; k < SS_W; ++k) {
      const float va = sa[ly][lx + k], vb = sb[ly][lx + k], w = g[k];
      m0 += w * va; m1 += w * vb; m2 += w * va * va; m3 += w * vb * vb; m4 += w * va * vb;
    }
    hm[0][ly][lx] = m0; hm[1][ly][lx] = m1; hm[2][ly][lx] = m2; hm[3][ly][lx] = m3; hm[4][ly][lx] = m4;
  }
  __syncthreads();
  float acc_se = 0.f, acc_ss = 0.f;
  {
    const int ly = tid / SS_TX, lx = tid % SS_TX;               // 256 threads = 8 x 32 outputs
    const int y = y0 + ly, x = x0 + lx;
    if (y < H && x < W) {
      float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
      for (int k = 0; k < SS_W; ++k) {
        const float w = g[k];
        mu1 += w * hm[0][ly + k][lx]; mu2 += w * hm[1][ly + k][lx];
        e11 += w * hm[2][ly + k][lx]; e22 += w * hm[3][ly + k][lx]; e12 += w * hm[4][ly + k][lx];
      }
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
      const float s1 = e11 - mu1s, s2 = e22 - mu2s, s12 = e12 - mu12;
      acc_ss = ((2.f * mu12 + C1) * (2.f * s12 + C2)) / ((mu1s + mu2s + C1) * (s1 + s2 + C2));
      const float d = sa[ly + SS_R][lx + SS_R] - sb[ly + SS_R][lx + SS_R];
      acc_se = d * d;
    }
  }
  acc_se = wave_sum(acc_se); acc_ss = wave_sum(acc_ss);
  if ((tid & 63) == 0) { sred[tid >> 6][0] = acc_se; sred[tid >> 6][1] = acc_ss; }
  __syncthreads();
  if (tid < 2)       // partial row per workgroup, rows of one sample contiguous: folded in fixed order by csbsr_sum_partials_batched
    sums[(((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2 + tid] = sred[0][tid] + sred[1][tid] + sred[2][tid] + sred[3][tid];
}
__global__ void psnr_ssim_finish_kernel(const float* sums, int N, float count, float* psnr, float* ssim) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float mse = sums[n * 2] / count;
  psnr[n] = 10.f * log10f(1.f / mse);
  ssim[n] = sums[n * 2 + 1] / count;
}
extern "C" int csbsr_psnr_ssim(const float* a, const float* b, int32_t N, int32_t C, int32_t H, int32_t W, float* sums /* [N][2] zeroed */,
                               float* psnr, float* ssim, csbsr_stream_t s) {
  CSBSR_CHECK(a && b && sums && psnr && ssim && N > 0 && C > 0, "psnr_ssim: bad args");
  dim3 grid((W + SS_TX - 1) / SS_TX, (H + SS_TY - 1) / SS_TY, N * C);
  CSBSR_CHECK(grid.y <= 65535 && grid.z <= 65535, "psnr_ssim: image too large for the launch grid");
  const int rows = (int)(C * grid.y * grid.x);
  float* part = csbsr_red_scratch((long)N * rows * 2);
  CSBSR_NEED_SCRATCH(part, "psnr_ssim");
  hipLaunchKernelGGL(psnr_ssim_kernel, grid, dim3(256), 0, ST(s), a, b, C, H, W, part);
  if (csbsr_sum_partials_batched(part, rows, 2, 2, sums, N, 2, ST(s))) return 1;
  hipLaunchKernelGGL(psnr_ssim_finish_kernel, dim3((N + 63) / 64), dim3(64), 0, ST(s), sums, N, (float)((long)C * H * W), psnr, ssim);
  CSBSR_LAUNCH_CHECK("csbsr_psnr_ssim");
  return 0;
}
