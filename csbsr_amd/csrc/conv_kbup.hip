// Backward of kb.up_conv1 -- ConvTranspose2d(3 -> C, 2s x 2s, stride s, pad s/2) + PReLU, whose output is added to the up-block's h
// (/root/reference/model/modeling/kbpn.py:372-374, 405-409) -- in ONE streaming pass over the output gradient.
//
// The layer's pre-activation is a 12-term dot product of a 3-channel LR image (four taps per output phase x three channels), so the
// backward can REBUILD it instead of reading it back: the stand-alone epilogue-backward pass read dOut, the saved output and the residual
// it was added to (to recover sign and value of the pre-activation from out - res) and wrote dPre, 13 GB per launch at N = 4, and the
// weight gradient then read dPre again.  Here a lane owns eight output channels of a pixel, recomputes their pre-activations with the
// forward kernel's own arithmetic (conv_thin_tp_kernel: v_dot2_f32_f16 over the same operands in the same order, so the sign is the
// forward's, not that of a difference of two rounded maps), and from ONE 16-byte load of dOut produces
//     dPre  = dOut * (pre > 0 ? 1 : a)                           (fp16, for the strided dgrad conv_thin_sc that follows)
//     dW   += x[tap, ci] * dPre                                  (fp32 registers: 4 taps x 3 channels x 8 couts per lane)
//     da   += dOut * min(pre, 0)                                 (the PReLU slope's gradient)
// A workgroup = one output row phase py of RB input rows, wave w = column phase px = w: the four waves read whole contiguous output
// rows; the two input rows of an output row are staged in LDS (as in the forward).  The weight-gradient sums leave as fp32 slabs in
// csbsr_conv_wgrad's layout G[split][a = ci (8 rows)][tap][co] -- split = (sample, row block); its 4 x s workgroup-waves own disjoint
// taps -- and csbsr_unpack_wgrad folds them in a fixed order; the slope sums as one partial per workgroup (no atomics).
#include "common.h"
#include "conv_common.h"

#define KBU_RB 4           // input rows per workgroup
#define KBU_U 4            // pixels a lane has in flight
#define KBU_ROW_MAX 1024

struct KbupK {
  const half_t* dout; long d_sn, d_sy, d_sx;
  const half_t* x; long x_sn, x_sy, x_sx;          // [N, h, w, 8] (3 real channels)
  const half_t* wt; int Kp, rows_p;                // the forward's phase-packed weights (csbsr_pack_weights kind 2): [phase][rows_p][Kp], k = tap * 8 + c
  const float* prelu;
  int N, h, w, c8, pad, nrb;
  half_t* dpre; long p_sn, p_sy, p_sx;
  float* slabs; long slab_stride;                  // [N * nrb][8][4 s^2 taps][c8 * 8]; null: no weight / slope gradient (frozen layer)
  float* dprelu_part;                              // [gridDim.x]
};

template <int S>
__global__ __launch_bounds__(64 * S) void thin_tp_bwd_kernel(const KbupK p) {
  __shared__ h8 srow[2][KBU_ROW_MAX + 2];
  __shared__ float sred[S];
  const int tid = threadIdx.x, lane = tid & 63;
  const int px = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c8 = p.c8, npl = 64 / c8;
  const int oct = lane % c8, sub = lane / c8, co = oct * 8;
  int b = blockIdx.x;
  const int py = b % S; b /= S;
  const int rb = b % p.nrb;
  const int n = b / p.nrb;
  const int by = (py + p.pad) / S, bx = (px + p.pad) / S, ph = py * S + px;
  h2 w2[4][2][8];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const half_t* wp = p.wt + ((size_t)ph * p.rows_p + co + e) * p.Kp + t * 8;
      w2[t][0][e] = h2{wp[0], wp[1]};
      w2[t][1][e] = h2{wp[2], (half_t)0};
    }
  const float slope = *p.prelu;
  float wg[4][3][8];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) wg[t][c][e] = 0.f;
  float dpr = 0.f;
  const half_t* xn = p.x + (long)n * p.x_sn;
  const int r_end = (rb + 1) * KBU_RB < p.h ? (rb + 1) * KBU_RB : p.h;
  for (int r = rb * KBU_RB; r < r_end; ++r) {
    const int oy = r * S + py;
    __syncthreads();
    for (int i = tid; i < 2 * (p.w + 2); i += 64 * S) {
      const int tr = i / (p.w + 2), ix = i - tr * (p.w + 2) - 1, iy = r + by - tr;
      h8 v = h8{0, 0, 0, 0, 0, 0, 0, 0};
      if ((unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w) v = *reinterpret_cast<const h8*>(xn + (long)iy * p.x_sy + (long)ix * p.x_sx);
      srow[tr][ix + 1] = v;
    }
    __syncthreads();
    const half_t* drow = p.dout + (long)n * p.d_sn + (long)oy * p.d_sy + co;
    half_t* prow = p.dpre + (long)n * p.p_sn + (long)oy * p.p_sy + co;
    for (int qx0 = sub; qx0 < p.w; qx0 += KBU_U * npl) {
      h8 g[KBU_U];
#pragma unroll
      for (int u = 0; u < KBU_U; ++u) {
        const int qx = qx0 + u * npl;
        g[u] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        if (qx < p.w) g[u] = *reinterpret_cast<const h8*>(drow + (long)(qx * S + px) * p.d_sx);
      }
#pragma unroll
      for (int u = 0; u < KBU_U; ++u) {
        const int qx = qx0 + u * npl;
        if (qx >= p.w) break;
        h8 xv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) xv[t] = srow[t >> 1][qx + bx - (t & 1) + 1];
        float pre[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pre[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {      // (operand order of conv_thin_tp_kernel: the rebuilt value is the forward's, bit for bit)
          const h2 x01 = h2{xv[t][0], xv[t][1]}, x23 = h2{xv[t][2], (half_t)0};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            pre[e] = __builtin_amdgcn_fdot2(x01, w2[t][0][e], pre[e], false);
            pre[e] = __builtin_amdgcn_fdot2(x23, w2[t][1][e], pre[e], false);
          }
        }
        float gd[8];
        h8 hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float go = (float)g[u][e];
          const bool pos = pre[e] > 0.f;
          gd[e] = pos ? go : go * slope;
          dpr += pos ? 0.f : go * pre[e];
          hv[e] = (half_t)gd[e];
        }
        *reinterpret_cast<h8*>(prow + (long)(qx * S + px) * p.p_sx) = hv;
        if (p.slabs) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float xf = (float)xv[t][c];
#pragma unroll
              for (int e = 0; e < 8; ++e) wg[t][c][e] += xf * gd[e];
            }
        }
      }
    }
  }
  if (p.slabs) {
    // fold the pixel lanes of a wave (lanes oct, oct + c8, ..), then lane `oct` writes its 8 couts of the wave's 4 taps x 3 channels
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = wg[t][c][e];
          for (int o = c8; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
          wg[t][c][e] = a;
        }
    if (sub == 0) {
      const int KW = 2 * S, ntap = KW * KW, cp = c8 * 8;
      float* slab = p.slabs + ((size_t)n * p.nrb + rb) * p.slab_stride;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int kh = (py + p.pad) % S + S * (t >> 1), kw = (px + p.pad) % S + S * (t & 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float* o = slab + ((size_t)c * ntap + kh * KW + kw) * cp + co;
          *reinterpret_cast<f4*>(o) = f4{wg[t][c][0], wg[t][c][1], wg[t][c][2], wg[t][c][3]};
          *reinterpret_cast<f4*>(o + 4) = f4{wg[t][c][4], wg[t][c][5], wg[t][c][6], wg[t][c][7]};
        }
      }
    }
    // slope gradient: lanes, then the waves one after the other (fixed order)
    for (int o = 1; o < 64; o <<= 1) dpr += __shfl_xor(dpr, o, 64);
    if (lane == 0) sred[px] = dpr;
    __syncthreads();
    if (tid == 0) {
      float a = 0.f;
      for (int w_ = 0; w_ < S; ++w_) a += sred[w_];
      p.dprelu_part[blockIdx.x] = a;
    }
  }
}

extern "C" int32_t csbsr_thin_tp_backward_slabs(int32_t N, int32_t h) { return N * ((h + KBU_RB - 1) / KBU_RB); }

/* see include/csbsr_hip.h */
extern "C" int csbsr_thin_tp_backward(const void* dout, int64_t d_sn, int64_t d_sy, int64_t d_sx, const void* x, int64_t x_sn, int64_t x_sy,
                                      int64_t x_sx, const void* wt_packed, int32_t cin, int32_t cout, int32_t stride, int32_t pad,
                                      const float* prelu, int32_t N, int32_t h, int32_t w, void* dpre, int64_t p_sn, int64_t p_sy, int64_t p_sx,
                                      float* slabs, float* dprelu_part, csbsr_stream_t s) {
  CSBSR_CHECK(dout && x && wt_packed && prelu && dpre, "thin_tp_backward: null pointer");
  CSBSR_CHECK(cin == 3 && cout % 8 == 0 && cout >= 8 && cout <= 128 && 64 % (cout / 8) == 0, "thin_tp_backward: 3 input channels, 8..128 output channels (a divisor of 512)");
  CSBSR_CHECK(stride == 4 && pad >= 0 && pad < stride && w <= KBU_ROW_MAX, "thin_tp_backward: 8x8 stride-4 layers, input rows of <= %d pixels", KBU_ROW_MAX);
  CSBSR_CHECK((slabs == nullptr) == (dprelu_part == nullptr), "thin_tp_backward: weight-gradient slabs and slope partials go together");
  KbupK k;
  k.dout = reinterpret_cast<const half_t*>(dout); k.d_sn = d_sn; k.d_sy = d_sy; k.d_sx = d_sx;
  k.x = reinterpret_cast<const half_t*>(x); k.x_sn = x_sn; k.x_sy = x_sy; k.x_sx = x_sx;
  k.wt = reinterpret_cast<const half_t*>(wt_packed);
  k.Kp = round_up(4 * 8, 64); k.rows_p = conv_rows_padded(cout);
  k.prelu = prelu; k.N = N; k.h = h; k.w = w; k.c8 = cout / 8; k.pad = pad; k.nrb = (h + KBU_RB - 1) / KBU_RB;
  k.dpre = reinterpret_cast<half_t*>(dpre); k.p_sn = p_sn; k.p_sy = p_sy; k.p_sx = p_sx;
  k.slabs = slabs; k.slab_stride = (long)8 * (4 * stride * stride) * cout; k.dprelu_part = dprelu_part;
  const unsigned grid = (unsigned)(N * k.nrb * stride);
  hipLaunchKernelGGL((thin_tp_bwd_kernel<4>), dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(s), k);
  CSBSR_LAUNCH_CHECK("csbsr_thin_tp_backward");
  return 0;
}
