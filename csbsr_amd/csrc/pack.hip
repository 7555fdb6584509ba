// Weight re-layout kernels: fp32 master weights (the reference's OIHW / IOHW parameter tensors, whose
// state_dict layout is part of the drop-in boundary) -> packed fp16 MFMA operands, and packed fp32 weight
// gradients -> += fp32 OIHW / IOHW .grad.   HBM-bound index shuffles, run once per step per layer.
#include "common.h"

struct PackK {
  const float* w; half_t* dst;
  int kind;            // 0 conv  1 conv flipped+swapped (dgrad of stride-1 conv)  2 transposed (phases)
  int D0, D1, KH, KW, stride, pad;
  int seg0_real, seg0_p, segtot_p, chan_real;
  int row_off, nrows, rows_p, Kp, KHt, KWt, nphase, k_off;
  int split;           // 1: K channels are [hi | hi | lo] blocks of seg0_p each, 2: [hi | lo], 3: [hi | hi], 4: per 32-channel slice [hi 32 | lo 32] (csbsr_pack_weights_split)
  float wscale;        // weights are multiplied by this power of two before the fp16 split (keeps the lo halves out of subnormals)
};

__global__ void pack_weights_kernel(const PackK p) {
  const long total = (long)p.nphase * p.rows_p * p.Kp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % p.Kp);
    const long t = i / p.Kp;
    const int r = (int)(t % p.rows_p);
    const int ph = (int)(t / p.rows_p);
    float v = 0.f;
    int tap = k / p.segtot_p;
    const int cp = k - tap * p.segtot_p;
    // padded channel -> real channel of the K side
    int c = -1, blk = 0;
    if (p.split == 4) {      // K slice j = k / 64 holds units [32 j, 32 j + 32) of the flat (tap, channel) index, hi then lo
      blk = (k >> 5) & 1;
      const int u = (k >> 6) * 32 + (k & 31);
      tap = u / p.seg0_p;
      const int c1 = u - tap * p.seg0_p;
      if (c1 < p.seg0_real) c = c1;
    }
    else if (p.split) { blk = cp / p.seg0_p; const int c1 = cp - blk * p.seg0_p; if (c1 < p.seg0_real) c = c1; }
    else if (cp < p.seg0_p) { if (cp < p.seg0_real) c = cp; }
    else { const int c1 = cp - p.seg0_p; if (p.seg0_real + c1 < p.chan_real) c = p.seg0_real + c1; }
    if (c >= 0) c += p.k_off;
    if (r < p.nrows && c >= 0 && tap < p.KHt * p.KWt) {
      const int jy = tap / p.KWt, jx = tap % p.KWt;
      const int rr = p.row_off + r;
      if (p.kind == 0) {
        v = p.w[(((long)rr * p.D1 + c) * p.KH + jy) * p.KW + jx];
      } else if (p.kind == 1) {
        v = p.w[(((long)c * p.D1 + rr) * p.KH + (p.KH - 1 - jy)) * p.KW + (p.KW - 1 - jx)];
      } else {
        const int py = ph / p.stride, px = ph % p.stride;
        const int kh = (py + p.pad) % p.stride + p.stride * jy;
        const int kw = (px + p.pad) % p.stride + p.stride * jx;
        if (kh < p.KH && kw < p.KW) v = p.w[(((long)c * p.D1 + rr) * p.KH + kh) * p.KW + kw];
      }
    }
    if (p.split) {
      v *= p.wscale;
      const half_t hi = (half_t)v;
      p.dst[i] = blk < ((p.split == 2 || p.split == 4) ? 1 : 2) ? hi : (half_t)(v - (float)hi);
    } else {
      p.dst[i] = (half_t)v;
    }
  }
}

static void pack_geometry(int kind, int D0, int D1, int KH, int KW, int stride, int seg0_real, int seg1_real,
                          int nrows, PackK& p) {
  p.kind = kind; p.D0 = D0; p.D1 = D1; p.KH = KH; p.KW = KW; p.stride = stride;
  p.seg0_real = seg0_real;
  p.seg0_p = round_up(seg0_real, 8);
  p.segtot_p = p.seg0_p + (seg1_real > 0 ? round_up(seg1_real, 8) : 0);
  p.chan_real = seg0_real + seg1_real;
  p.nrows = nrows;
  p.rows_p = round_up(nrows, nrows > 64 ? 128 : 32);   // = conv_rows_padded() of conv_common.h
  p.KHt = kind == 2 ? (KH + stride - 1) / stride : KH;
  p.KWt = kind == 2 ? (KW + stride - 1) / stride : KW;
  p.nphase = kind == 2 ? stride * stride : 1;
  p.Kp = round_up(p.KHt * p.KWt * p.segtot_p, 64);   // = BK of conv_igemm.hip
  p.split = 0; p.wscale = 1.f;
}
// split-fp16 operand: the activation arrives as in[0] = [x_hi | x_lo] (2c channels), in[1] = x_hi (c channels), so per tap the K
// axis is three blocks of c channels holding [w_hi | w_hi | w_lo]:  x_hi w_hi + x_lo w_hi + x_hi w_lo  (x_lo w_lo ~ 2^-22 dropped)
// layout 1 (dgrad of the split-precision detector): the activation gradient is plain fp16, passed as in[0] = in[1] = dY, against
// [w_hi | w_lo]:  dY w_hi + dY w_lo -- the weight's fp16 rounding error is the same for every pixel, so unlike the (incoherent)
// rounding of the gradients it does not average out in the BatchNorm backward sums
// layout 2 (per-layer precision plan): in[0] = [x_hi | x_lo] alone against [w_hi | w_hi]: the activation keeps its ~22 bits, the
// weight its fp16 rounding (two K blocks instead of three)
// layout 3 (fused form of layout 0, csbsr_conv_desc_t::split_fused): per tap and 32-channel slice [w_hi (32) | w_lo (32)] -- one 64-wide
// K slice of the LDS-DMA kernels, staged next to [x_hi (32) | x_lo (32)] and used for all three products
static void pack_geometry_split(int kind, int D0, int D1, int KH, int KW, int stride, int creal, int nrows, int layout, PackK& p) {
  pack_geometry(kind, D0, D1, KH, KW, stride, creal, 0, nrows, p);
  p.segtot_p = (layout == 0 ? 3 : 2) * p.seg0_p;
  p.Kp = round_up(p.KHt * p.KWt * p.segtot_p, 64);
  p.split = layout == 1 ? 2 : (layout == 2 ? 3 : (layout == 3 ? 4 : 1));
}

extern "C" int64_t csbsr_packed_weight_elems(int32_t kind, int32_t D0, int32_t D1, int32_t KH, int32_t KW,
                                             int32_t stride, int32_t seg0_real, int32_t seg1_real, int32_t nrows) {
  PackK p;
  pack_geometry(kind, D0, D1, KH, KW, stride, seg0_real, seg1_real, nrows, p);
  return (int64_t)p.nphase * p.rows_p * p.Kp;
}

extern "C" int csbsr_pack_weights(const float* w, void* dst, int32_t kind, int32_t D0, int32_t D1, int32_t KH,
                                  int32_t KW, int32_t stride, int32_t pad, int32_t seg0_real, int32_t seg1_real,
                                  int32_t row_off, int32_t nrows, int32_t k_off, csbsr_stream_t s) {
  CSBSR_CHECK(w && dst, "pack: null pointer");
  CSBSR_CHECK(kind >= 0 && kind <= 2, "pack: bad kind");
  const int kdim = kind == 0 ? D1 : D0;       // which weight dim the K-side channels index
  const int rdim = kind == 0 ? D0 : D1;
  CSBSR_CHECK(k_off >= 0 && k_off + seg0_real + seg1_real <= kdim, "pack: segments (%d+%d at %d) exceed the contracted dim (%d)", seg0_real, seg1_real, k_off, kdim);
  CSBSR_CHECK(row_off >= 0 && row_off + nrows <= rdim, "pack: row range out of bounds");
  PackK p;
  pack_geometry(kind, D0, D1, KH, KW, stride, seg0_real, seg1_real, nrows, p);
  p.w = w; p.dst = reinterpret_cast<half_t*>(dst); p.pad = pad; p.row_off = row_off; p.k_off = k_off;
  const long total = (long)p.nphase * p.rows_p * p.Kp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(s), p);
  CSBSR_LAUNCH_CHECK("csbsr_pack_weights");
  return 0;
}

extern "C" int64_t csbsr_packed_weight_elems_split(int32_t kind, int32_t D0, int32_t D1, int32_t KH, int32_t KW, int32_t stride,
                                                   int32_t creal, int32_t nrows, int32_t layout) {
  PackK p;
  pack_geometry_split(kind, D0, D1, KH, KW, stride, creal, nrows, layout, p);
  return (int64_t)p.nphase * p.rows_p * p.Kp;
}

extern "C" int csbsr_pack_weights_split(const float* w, void* dst, int32_t kind, int32_t D0, int32_t D1, int32_t KH, int32_t KW,
                                        int32_t stride, int32_t pad, int32_t creal, int32_t row_off, int32_t nrows, int32_t k_off,
                                        float wscale, int32_t layout, csbsr_stream_t s) {
  CSBSR_CHECK(w && dst, "pack_split: null pointer");
  CSBSR_CHECK(kind >= 0 && kind <= 2, "pack_split: bad kind");
  const int kdim = kind == 0 ? D1 : D0, rdim = kind == 0 ? D0 : D1;
  CSBSR_CHECK(k_off >= 0 && k_off + creal <= kdim, "pack_split: channels (%d at %d) exceed the contracted dim (%d)", creal, k_off, kdim);
  CSBSR_CHECK(row_off >= 0 && row_off + nrows <= rdim, "pack_split: row range out of bounds");
  PackK p;
  pack_geometry_split(kind, D0, D1, KH, KW, stride, creal, nrows, layout, p);
  p.w = w; p.dst = reinterpret_cast<half_t*>(dst); p.pad = pad; p.row_off = row_off; p.k_off = k_off; p.wscale = wscale;
  const long total = (long)p.nphase * p.rows_p * p.Kp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(s), p);
  CSBSR_LAUNCH_CHECK("csbsr_pack_weights_split");
  return 0;
}

// G[a][tap][bpad]  ->  grad[a_off + a][b][kh][kw] += ...   (grad is [D0][D1][KH][KW]; a indexes D0 unless
// transpose_ab & 1, in which case a indexes D1 and b indexes D0; transpose_ab & 2: the taps are mirrored, (kh, kw) -> (KH-1-kh, KW-1-kw) --
// together the two bits turn the slab of the MIRRORED problem  sum_pix X[pix][ci] dPre[pix - off(tap)][co]  into an OIHW gradient)
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* g, float* grad, int A, int Breal, int KH, int KW, int seg0_real,
                                                           int seg0_p, int segtot_p, int D0, int D1, int transpose_ab, int b_off, float scale,
                                                           int splits, long slab) {
  // A workgroup = 32 groups of FOUR consecutive elements of the packed layout [a][tap][b_padded] (b_padded is a multiple of 8, so the
  // four share row, tap and segment) x 8 slab lanes: lane s sums slabs s, s + 8, ... with 16-byte loads (32 consecutive groups = 512
  // contiguous bytes per slab), the eight partial sums meet in LDS, lane 0 writes OIHW / IOHW.  (One thread per group walking all
  // 256-1024 slabs two at a time was latency-bound: 65 us per call on 36 KB slabs, 230 calls per step.)
  __shared__ f4 red[256];
  const int ktot = KH * KW * segtot_p;
  const long total4 = (long)A * ktot / 4;
  const f4* g4 = reinterpret_cast<const f4*>(g);
  const long slab4 = slab / 4;
  const int ql = threadIdx.x & 31, sl = threadIdx.x >> 5;
  for (long qb = (long)blockIdx.x * 32; qb < total4; qb += (long)gridDim.x * 32) {
    const long q = qb + ql;
    bool valid = q < total4;
    int a = 0, tap = 0, bp = 0, b = 0;
    if (valid) {
      const long gi = q * 4;
      a = (int)(gi / ktot);
      const int r = (int)(gi - (long)a * ktot);
      tap = r / segtot_p; bp = r - tap * segtot_p;
      if (bp < seg0_p) { b = bp; valid = bp < seg0_real; }
      else { b = seg0_real + (bp - seg0_p); valid = b < Breal; }
    }
    f4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      int sp = sl;
      for (; sp + 8 < splits; sp += 16) { v0 += g4[(long)sp * slab4 + q]; v1 += g4[(long)(sp + 8) * slab4 + q]; }
      if (sp < splits) v0 += g4[(long)sp * slab4 + q];
    }
    red[threadIdx.x] = v0 + v1;
    __syncthreads();
    if (sl == 0 && valid) {
      f4 v = red[ql];
#pragma unroll
      for (int j = 1; j < 8; ++j) v += red[32 * j + ql];
      int kh = tap / KW, kw = tap - kh * KW;
      if (transpose_ab & 2) { kh = KH - 1 - kh; kw = KW - 1 - kw; }
      const int lim = bp < seg0_p ? seg0_real : Breal;      // real channels end inside this group of four?
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (b + e >= lim) break;
        const int bb = b + e + b_off;
        const long di = (transpose_ab & 1) ? ((((long)bb * D1 + a) * KH + kh) * KW + kw) : ((((long)a * D1 + bb) * KH + kh) * KW + kw);
        grad[di] += v[e] * scale;
      }
    }
    __syncthreads();
  }
}

extern "C" int csbsr_unpack_wgrad(const float* g, float* grad, int32_t A, int32_t KH, int32_t KW, int32_t seg0_real,
                                  int32_t seg1_real, int32_t D0, int32_t D1, int32_t transpose_ab, int32_t b_off,
                                  float scale, int32_t splits, int32_t ca_padded, csbsr_stream_t s) {
  CSBSR_CHECK(g && grad, "unpack: null pointer");
  const int seg0_p = round_up(seg0_real, 8);
  const int segtot_p = seg0_p + (seg1_real > 0 ? round_up(seg1_real, 8) : 0);
  const int Breal = seg0_real + seg1_real;
  const long total = (long)A * KH * KW * segtot_p;
  int blocks = (int)((total / 4 + 31) / 32);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(s), g, grad, A,
                     Breal, KH, KW, seg0_real, seg0_p, segtot_p, D0, D1, transpose_ab, b_off, scale, splits,
                     (long)ca_padded * KH * KW * segtot_p);
  CSBSR_LAUNCH_CHECK("csbsr_unpack_wgrad");
  return 0;
}
