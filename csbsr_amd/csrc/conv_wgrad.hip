// Weight-gradient GEMM for gfx950:   G[a][tap][b] = sum over pixels  A[pix][a] * B[pix @ tap][b]
//
// Both operands are channels-last, so the reduction dimension (pixels) is the strided one.  Tiles are staged
// to LDS exactly as loaded ([pixel][channel], 16-byte channel chunks) and the MFMA fragments -- which want 8
// consecutive *pixels* per lane -- are fetched with the gfx950 hardware transpose read ds_read_b64_tr_b16
// (two per fragment), so no shuffle / scalar-LDS transposition is needed.  fp32 accumulate; the pixel range is
// split across grid.z, every split writes its own fp32 slab and csbsr_unpack_wgrad sums the slabs in a fixed order.
//
// Autograd wgrad of F.conv2d / F.conv_transpose2d at the call sites listed in conv_igemm.hip.
#include "common.h"
#include "csbsr_debug.h"

#include "conv_wgrad.h"

template <bool USE_TR>
__device__ __forceinline__ h8 frag_T(const half_t* tile, int ld, int pix0, int ch) {
  // returns {tile[pix0+0..7][ch]}  for the calling lane
  if constexpr (USE_TR) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    // lane supplies the address of 4 contiguous halves: row (i/4), columns 4*(i%4).. of its group's 4x16 block;
    // the group's block starts at channel (ch - i)
    const half_t* p0 = tile + (pix0 + (i >> 2)) * ld + (ch - i) + 4 * (i & 3);
    fp16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p0));
    fp16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p0 + 4 * ld));
    h8 v;
    v[0] = r0[0]; v[1] = r0[1]; v[2] = r0[2]; v[3] = r0[3]; v[4] = r1[0]; v[5] = r1[1]; v[6] = r1[2]; v[7] = r1[3];
    return v;
  } else {
    h8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[(pix0 + e) * ld + ch];
    return v;
  }
}

#ifdef CSBSR_TS
__device__ unsigned long long g_wts[8 * 65536];
extern "C" int csbsr_debug_read_wts(void* dst, long n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wts), n * 8); }
#define WTS_DECL unsigned long long ts_prev = wall_clock64(), ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define WTS(i) do { const unsigned long long t_ = wall_clock64(); ts_acc[i] += t_ - ts_prev; ts_prev = t_; } while (0)
#define WTS_FLUSH do { if (threadIdx.x == 0 && blockIdx.x + gridDim.x * blockIdx.z < 65536) for (int i_ = 0; i_ < 8; ++i_) g_wts[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8 + i_] = ts_acc[i_]; } while (0)
#else
#define WTS_DECL
#define WTS(i)
#define WTS_FLUSH
#endif
template <int BA, int BN, int WA, int WB, bool USE_TR>
__global__ __launch_bounds__(64 * WA * WB) void conv_wgrad_kernel(const WgradK p) {
  WTS_DECL;
  constexpr int NT = 64 * WA * WB;      // threads: every wave owns a 64 x 64 (or smaller) piece of the BA x BN tile
  constexpr int AW_ = BA / WA;          // a-rows per wave
  constexpr int BW_ = BN / WB;          // cols per wave
  constexpr int TA = AW_ / 32, TB = BW_ / 32;
  constexpr int LDA = BA + 32, LDB = BN + 32;      // row stride = 16 dwords mod 64: conflict-free ds_read_b64_tr_b16
  __shared__ __attribute__((aligned(16))) half_t sA[WG_BP * LDA];
  __shared__ __attribute__((aligned(16))) half_t sB[WG_BP * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wa = wid / WB, wb = wid % WB;
  const unsigned ntile = p.tiles_a * p.tiles_b;
  unsigned lt, zsplit;
  if (p.flat) {
    // every tile of a split reads the same pixel range of both operands, so the whole split goes to one XCD (a contiguous chunk of
    // the (split, tile) order): its L2 fetches the range once while the tiles stream through it in step.  With the tiles of a split
    // dealt across the XCDs instead, each XCD pulled in every operand range itself (PMC: 4-15x the operand bytes per launch).
    const unsigned w = xcd_remap(blockIdx.x, ntile * (unsigned)p.splits);
    zsplit = w / ntile; lt = w - zsplit * ntile;
  } else {
    lt = xcd_remap(blockIdx.x, ntile); zsplit = blockIdx.z;
  }
  const int a0 = (lt % p.tiles_a) * BA;
  int col0 = (lt / p.tiles_a) * BN;
  if (p.tap_perm) {
    // 8x8 stride-4 layers with 128 gathered channels: one column tile per tap, 64 tiles = 8 per XCD.  Taps whose kernel offsets agree
    // modulo the stride read the SAME strided pixel set of the gathered side (shifted by whole A-grid pixels), so XCD j takes the
    // taps with ky = j%4 (+4) and kx in {2(j/4), 2(j/4)+1} (+4): each XCD's L2 then holds 1/8 of the gathered tensor, fetched once,
    // instead of every row class being pulled in by two XCDs and every column by all of them (PMC: 9.2 GB per launch at N=4 for
    // 3.5 GB of operands).
    if constexpr (BN == 128) {
      const int j = lt >> 3, r = lt & 7;
      const int ky = (j & 3) + 4 * (r >> 2), kx = 2 * (j >> 2) + (r & 1) + 4 * ((r >> 1) & 1);
      col0 = (ky * 8 + kx) * 128;
    } else {      // 256-column tiles hold the tap pair (ky, kx0), (ky, kx0 + 1): 32 tiles, 4 per XCD
      const int j = lt >> 2, r = lt & 3;
      const int ky = (j & 3) + 4 * (r >> 1), kx0 = 2 * (j >> 2) + 4 * (r & 1);
      col0 = (ky * 8 + kx0) * 128;
    }
  }
  // Strided layers: the tap row ky reads gathered row y*stride + ky*dil - pad, so tiles whose taps differ by a whole stride in ky touch
  // the same (16x larger) gathered rows one A-row apart in time -- 7 steps x every resident workgroup's traffic, far beyond the L2.
  // Each tile's pixel ranges are therefore slid back by floor(ky*dil/stride) A-rows: all taps of a residue class then stream the same
  // gathered rows in the same steps.  (The ranges still partition [0, M): the first split is shorter, the last one longer.)
  long shift = 0;
  if (p.row_shift) shift = (long)((((col0 / p.cbtot) / p.KW) * p.dil) / p.stride) * p.AW;
  long mbeg = (long)zsplit * p.per_split - shift;
  long mend = mbeg + p.per_split;
  if (mbeg < 0) mbeg = 0;
  if (mend > p.M || (int)zsplit == p.splits - 1) mend = p.M;
  if (mbeg >= mend) return;

  // ---- B staging role: chunk ids tid + NT*j : pixel = id/BCH, col chunk = id%BCH
  constexpr int BCH = BN / 8, B_DELTA = NT / BCH;
  constexpr int B_ITERS = WG_BP * BCH / NT;
  static_assert(NT % BCH == 0 && (WG_BP * BCH) % NT == 0, "B staging must tile evenly");
  int b_ky, b_kx;
  const half_t* b_ptr;
  long b_sn, b_sy, b_sx;
  bool b_ok;
  {
    const int col = col0 + (tid % BCH) * 8;     // the column chunk is the same for every j (NT % BCH == 0)
    b_ok = col < p.ktot;
    const int tap = b_ok ? col / p.cbtot : 0;
    const int c = b_ok ? col - tap * p.cbtot : 0;
    b_ky = (tap / p.KW) * p.dil - p.pad;
    b_kx = (tap % p.KW) * p.dil - p.pad;
    const csbsr_seg_t& sg = c < p.cb0 ? p.b[0] : p.b[1];
    b_ptr = reinterpret_cast<const half_t*>(sg.ptr) + (c < p.cb0 ? c : c - p.cb0);
    b_sn = sg.sn; b_sy = sg.sy; b_sx = sg.sx;
  }
  // ---- A staging role: BA/8 chunks per pixel
  constexpr int ACH = BA / 8;
  constexpr int A_ITERS = WG_BP * ACH / NT;
  static_assert(NT % ACH == 0 && (WG_BP * ACH) % NT == 0, "A staging must tile evenly");

  // per-thread running pixel coordinates AND element offsets (no integer division, no 64-bit multiply in the loop: the offsets move by
  // precomputed strides; rebuilding n*sn + y*sy + x*sx per chunk cost ~17 VALU instructions per MFMA in the PMC counts): B chunk j
  // covers pixel tid/BCH + (NT/BCH) j, A chunk i covers pixel tid/ACH + (NT/ACH) i of the step being LOADED
  struct Pix { int n, y, x; long off; };
  auto init_pix = [&](long m, long sn, long sy, long sx) {
    Pix c;
    c.n = (int)(m / ((long)p.AH * p.AW));
    const int rem = (int)(m - (long)c.n * p.AH * p.AW);
    c.y = rem / p.AW; c.x = rem - c.y * p.AW;
    c.off = c.n * sn + c.y * sy + c.x * sx;
    return c;
  };
  constexpr int A_DELTA = NT / ACH;
  // B: element offset of tap (0,0)'s source pixel = n*b_sn + (y*stride)*b_sy + (x*stride)*b_sx ; the tap displacement is a constant
  const long bsx = (long)p.stride * b_sx, bsy = (long)p.stride * b_sy;
  const long b_tap = (long)b_ky * b_sy + (long)b_kx * b_sx;
  const long b_dx = B_DELTA * bsx, b_rowfix = bsy - (long)p.AW * bsx, b_imgfix = b_sn - (long)p.AH * bsy;
  const long a_dx = (long)A_DELTA * p.a_sx, a_rowfix = p.a_sy - (long)p.AW * p.a_sx, a_imgfix = p.a_sn - (long)p.AH * p.a_sy;
  auto advance = [&](Pix& c, int d, long dxs, long rowfix, long imgfix) {
    c.x += d; c.off += dxs;
    while (c.x >= p.AW) { c.x -= p.AW; c.off += rowfix; if (++c.y == p.AH) { c.y = 0; ++c.n; c.off += imgfix; } }
  };
  Pix cb = init_pix(mbeg + tid / BCH, b_sn, bsy, bsx);
  Pix ca_ = init_pix(mbeg + tid / ACH, p.a_sn, p.a_sy, p.a_sx);

  f16v acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  h8 gb[B_ITERS], ga[A_ITERS];
  const int a_ch = (tid % ACH) * 8;
  const half_t* a_base = p.a + a0 + a_ch;
  const half_t* b_base = b_ptr + b_tap;
  const bool a_ok = a0 + a_ch < p.ca;
  auto issue_loads = [&](long m) {     // global -> registers for the step starting at pixel m, then advance the coordinates
#pragma unroll
    for (int j = 0; j < B_ITERS; ++j) {
      const int pix = tid / BCH + B_DELTA * j;
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (b_ok && m + pix < mend) {
        const int by = cb.y * p.stride + b_ky, bx = cb.x * p.stride + b_kx;
        if ((unsigned)by < (unsigned)p.BH && (unsigned)bx < (unsigned)p.BW) v = *reinterpret_cast<const h8*>(b_base + cb.off);
      }
      gb[j] = v;
      advance(cb, B_DELTA, b_dx, b_rowfix, b_imgfix);
    }
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int pix = tid / ACH + A_DELTA * i;
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (a_ok && m + pix < mend) v = *reinterpret_cast<const h8*>(a_base + ca_.off);
      ga[i] = v;
      advance(ca_, A_DELTA, a_dx, a_rowfix, a_imgfix);
    }
  };

  WTS(0);
  issue_loads(mbeg);
  WTS(1);
  for (long m = mbeg; m < mend; m += WG_BP) {
    __syncthreads();   // previous step's fragment reads done
    WTS(2);
#pragma unroll
    for (int j = 0; j < B_ITERS; ++j) {
      const int id = tid + NT * j;
      *reinterpret_cast<h8*>(sB + (id / BCH) * LDB + (id % BCH) * 8) = gb[j];
    }
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) *reinterpret_cast<h8*>(sA + (tid / ACH + A_DELTA * i) * LDA + a_ch) = ga[i];
    WTS(3);
    __syncthreads();
    WTS(4);
    if (m + WG_BP < mend) issue_loads(m + WG_BP);     // next step's HBM loads fly under this step's MFMAs
    WTS(5);
#pragma unroll
    for (int ks = 0; ks < WG_BP / 16; ++ks) {
      const int pix0 = ks * 16 + (lane >> 5) * 8;
      h8 af[TA], bf[TB];
#pragma unroll
      for (int a = 0; a < TA; ++a) af[a] = frag_T<USE_TR>(sA, LDA, pix0, wa * AW_ + a * 32 + (lane & 31));
#pragma unroll
      for (int b = 0; b < TB; ++b) bf[b] = frag_T<USE_TR>(sB, LDB, pix0, wb * BW_ + b * 32 + (lane & 31));
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    WTS(6);
  }

  // ---- epilogue: D[a][col], lane: col = lane%32, rows (r&3)+8*(r>>2)+4*(lane>>5)
  // every (row < ca, col < ktot) element of this split's slab is written exactly once: no atomics, no zero-fill;
  // csbsr_unpack_wgrad sums the slabs
  float* slab = p.g + (size_t)zsplit * p.slab_stride + (size_t)p.row0 * p.ktot;
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int col = col0 + wb * BW_ + b * 32 + (lane & 31);
      if (col >= p.ktot) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = a0 + wa * AW_ + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= p.ca) continue;
        slab[(size_t)row * p.ktot + col] = acc[a][b][r];
      }
    }
  WTS(7);
  WTS_FLUSH;
}

// which kernel the calling thread's last csbsr_conv_wgrad dispatched to (csbsr_debug.h): 0 <128,128>, 1 <128,256>, 2 <64,128>, 3 <32,128>, 4 thin
static thread_local int g_last_wgrad_kernel = -1;
extern "C" int32_t csbsr_debug_last_wgrad_kernel(void) { return g_last_wgrad_kernel; }
static int g_wgrad_use_tr = 1;
static int g_wgrad_thin = 1;
static int g_wgrad_tap_perm = 1;
static int g_wgrad_row_shift = 1;
static int g_wgrad_wide = 1;
static int g_wgrad_extra_lds = 0;   // A/B: dynamic LDS bytes added to the launch to lower the occupancy
static int g_wgrad_flat = 1;     // 0 off, 1 every layer without the tap permutation, 2 every layer
extern "C" void csbsr_debug_set_wgrad_tr(int v) {
  g_wgrad_use_tr = v & 1; g_wgrad_thin = !(v & 2); g_wgrad_tap_perm = !(v & 4); g_wgrad_flat = (v & 8) ? 0 : ((v & 16) ? 2 : 1);
  g_wgrad_row_shift = !(v & 32); g_wgrad_wide = !(v & 64);
  // bit 7: register-staged kernel everywhere; bit 0 clear (scalar LDS transposition) implies it -- the LDS-DMA kernel only has the
  // hardware-transpose read; bits 8..9: LDS-DMA tile menu (256: no 256 x 256 tile, 512: no 128 x 256 tile); bit 10: LDS-DMA kernel only where the
  // 256-row tile applies (register-staged elsewhere: the default until the DMA pieces became inline assembly); bit 21: square tiles
  // for the tap-permuted 8x8 stride-4 layers; bit 22: linear 64-pixel stages (no 2-D stage rectangles)
  g_wgrad_glds = ((v & 128) || !(v & 1)) ? 0 : (1 | ((v & 256) ? 0 : 2) | ((v & 512) ? 4 : 0) | ((v & 1024) ? 0 : 8) | ((v & (1 << 21)) ? 0 : 64) | ((v & (1 << 22)) ? 0 : 128) | ((v & (1 << 23)) ? 0 : 256) | ((v & (1 << 24)) ? 0 : 512));      // bit 24: no 256 x 256 tile for 512 .. 1023 columns (A/B timing)
  g_wgrad_extra_lds = ((v >> 12) & 0xff) * 1024;
}

// ------------------------------------------------------------------------------------------------------------------------
// Thin-A variant: stride-1 "same" conv whose output has so few channels that KH*KW*ca_real <= 32 (the 3-channel image heads:
// kb.sr_reconst, output_conv, the dgrad side of fe_SR.0).  The taps move from the column index into the ROW index,
//     G[(tap, co)][ci] = sum over input pixels q  dPre[q - off(tap)][co] * X[q][ci]
// so one 32-row MFMA tile holds all taps of all output channels (27 of 32 rows live instead of 3), the B operand is the
// unshifted input -- read from HBM exactly once instead of once per tap -- and the A' tile is assembled from the (tiny,
// L2-resident) dPre map.  Same slab layout as the kernel above, so csbsr_unpack_wgrad is unchanged.
template <bool USE_TR>
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(const WgradK p) {
  constexpr int LDA = 32 + 32, LDB = WG_BN + 32;
  __shared__ __attribute__((aligned(16))) half_t sA[WG_BP * LDA];
  __shared__ __attribute__((aligned(16))) half_t sB[WG_BP * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col0 = blockIdx.x * WG_BN;                 // input-channel tile
  const long mbeg = (long)blockIdx.z * p.per_split;
  long mend = mbeg + p.per_split;
  if (mend > p.M) mend = p.M;
  if (mbeg >= mend) return;

  // ---- B staging role (as above, no tap shift): chunk ids tid + 256*j : pixel = id/16, col chunk = id%16
  constexpr int B_ITERS = WG_BP * 16 / 256;
  const half_t* b_ptr;
  long b_sn, b_sy, b_sx;
  bool b_ok;
  {
    const int c = col0 + (tid & 15) * 8;
    b_ok = c < p.cbtot;
    const csbsr_seg_t& sg = (c < p.cb0 || !b_ok) ? p.b[0] : p.b[1];
    b_ptr = reinterpret_cast<const half_t*>(sg.ptr) + (!b_ok ? 0 : (c < p.cb0 ? c : c - p.cb0));
    b_sn = sg.sn; b_sy = sg.sy; b_sx = sg.sx;
  }
  // ---- A' staging role: pixel tid/4, rows 8*(tid%4) .. +7 ; row m = tap * ca_real + co
  int a_dy[8], a_dx[8], a_co[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int m = (tid & 3) * 8 + e;
    const int tap = m / p.ca_real;
    a_co[e] = tap < p.KH * p.KW ? m - tap * p.ca_real : -1;
    a_dy[e] = p.pad - (tap / p.KW) * p.dil;            // dPre pixel = input pixel + (pad - k*dil)
    a_dx[e] = p.pad - (tap % p.KW) * p.dil;
  }
  struct Pix { int n, y, x; };
  auto init_pix = [&](long m) {
    Pix c;
    c.n = (int)(m / ((long)p.BH * p.BW));
    const int rem = (int)(m - (long)c.n * p.BH * p.BW);
    c.y = rem / p.BW; c.x = rem - c.y * p.BW;
    return c;
  };
  auto advance = [&](Pix& c, int d) {
    c.x += d;
    while (c.x >= p.BW) { c.x -= p.BW; if (++c.y == p.BH) { c.y = 0; ++c.n; } }
  };
  Pix cb = init_pix(mbeg + (tid >> 4));
  Pix ca_ = init_pix(mbeg + (tid >> 2));

  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  h8 gb[B_ITERS], ga;
  auto issue_loads = [&](long m) {
#pragma unroll
    for (int j = 0; j < B_ITERS; ++j) {
      const int pix = (tid >> 4) + 16 * j;
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (b_ok && m + pix < mend) v = *reinterpret_cast<const h8*>(b_ptr + cb.n * b_sn + cb.y * b_sy + cb.x * b_sx);
      gb[j] = v;
      advance(cb, 16);
    }
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m + (tid >> 2) < mend) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ay = ca_.y + a_dy[e], ax = ca_.x + a_dx[e];
        if (a_co[e] >= 0 && (unsigned)ay < (unsigned)p.AH && (unsigned)ax < (unsigned)p.AW)
          v[e] = p.a[ca_.n * p.a_sn + ay * p.a_sy + ax * p.a_sx + a_co[e]];
      }
    }
    ga = v;
    advance(ca_, 64);
  };

  issue_loads(mbeg);
  for (long m = mbeg; m < mend; m += WG_BP) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < B_ITERS; ++j) {
      const int id = tid + 256 * j;
      *reinterpret_cast<h8*>(sB + (id >> 4) * LDB + (id & 15) * 8) = gb[j];
    }
    *reinterpret_cast<h8*>(sA + (tid >> 2) * LDA + (tid & 3) * 8) = ga;
    __syncthreads();
    if (m + WG_BP < mend) issue_loads(m + WG_BP);
#pragma unroll
    for (int ks = 0; ks < WG_BP / 16; ++ks) {
      const int pix0 = ks * 16 + (lane >> 5) * 8;
      const h8 af = frag_T<USE_TR>(sA, LDA, pix0, lane & 31);
      const h8 bf = frag_T<USE_TR>(sB, LDB, pix0, wid * 32 + (lane & 31));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
    }
  }
  float* slab = p.g + (size_t)blockIdx.z * p.slab_stride;
  const int ci = col0 + wid * 32 + (lane & 31);
  if (ci >= p.cbtot) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int tap = m / p.ca_real;
    if (tap >= p.KH * p.KW) continue;
    slab[(size_t)(m - tap * p.ca_real) * p.ktot + (size_t)tap * p.cbtot + ci] = acc[r];
  }
}

static bool wgrad_is_thin(const csbsr_wgrad_desc_t* d) {
  const int cr = d->ca_real > 0 ? d->ca_real : d->ca;
  return g_wgrad_thin && d->stride == 1 && d->KH * d->KW * cr <= 32 && d->KH * d->KW > 1 && d->AH == d->BH && d->AW == d->BW &&
         2 * d->pad == d->dil * (d->KH - 1) && d->KH == d->KW && d->b[0].sx != 0 && (d->b[1].c == 0 || d->b[1].sx != 0) &&
         (d->b[0].c + d->b[1].c) >= 64;
}

static int wgrad_tile_a(int ca) { return ca > 64 ? 128 : (ca > 32 ? 64 : 32); }
static int wgrad_tile_n(int ca, int ktot, bool perm8);
// tile shape the dispatch below will use for a problem (the pixel splits, the flat grid and the tap permutation depend on it)
static void wgrad_tiles(int ca, int ktot, bool perm8, int& BA, int& BN) {
  WgradK t{};
  t.ca = ca; t.ktot = ktot; t.tap_perm = perm8 ? 1 : 0;
  if (wgrad_glds_eligible(t) && wgrad_glds_tile_a(t) == 256) { BA = BN = 256; }
  else { BA = wgrad_tile_a(ca); BN = wgrad_tile_n(ca, ktot, perm8); }
}
// 128 x 256 tiles on 8 waves (each wave still owns 64 x 64) for the layers with thousands of columns: the kernel is bound by what a CU
// can load (~16 B/clk, scripts/ts_wgrad.py) and the wide tile moves 48 KB per 64-pixel step for twice the MFMAs of the 32 KB square
// one.  One such workgroup fits per CU (158 VGPRs x 8 waves), so it only pays where the pixel loops are long: +10..16 % on the SFT and
// decoder 3x3 layers, -20 % on a 9-tile 128-channel 3x3, -2 % on the tap-permuted 8x8 stride-4 layers (which keep the square tile).
static int wgrad_tile_n(int ca, int ktot, bool perm8) { return (g_wgrad_wide && ca > 64 && ktot >= 6144 && !perm8) ? 256 : 128; }

// number of pixel-range splits (= fp32 slabs the caller must provide) for a problem
static int32_t wgrad_splits_impl(int32_t ca, int32_t ktot, int64_t M, bool perm8) {
  int BA, BN;
  wgrad_tiles(ca, ktot, perm8, BA, BN);
  if (perm8 && (g_wgrad_glds & 256) && ktot == 8192 && ca == 128) BN = 512;      // (the experimental four-tap tile: 16 tiles)
  const long ntile = (long)((ca + BA - 1) / BA) * ((ktot + BN - 1) / BN);
  long want = (1536 + ntile - 1) / ntile;
  long maxs = (M + WG_BP * 8 - 1) / (WG_BP * 8);
  long splits = want < maxs ? want : maxs;
  const long slab_bytes = (long)ca * ktot * 4;
  const long cap = (192L << 20) / (slab_bytes > 0 ? slab_bytes : 1);      // keep the slab workspace under 192 MiB
  if (splits > cap) splits = cap;
  if (splits > 1024) splits = 1024;
  if (splits < 1) splits = 1;
  const long per_split = ((M + splits - 1) / splits + WG_BP - 1) / WG_BP * WG_BP;
  return (int32_t)((M + per_split - 1) / per_split);
}

extern "C" int32_t csbsr_wgrad_splits(int32_t ca, int32_t ktot, int64_t M) { return wgrad_splits_impl(ca, ktot, M, false); }

static long wgrad_splits_for(long ntile, long slab_elems, long M) {
  long want = (1536 + ntile - 1) / ntile;
  long maxs = (M + WG_BP * 8 - 1) / (WG_BP * 8);
  long splits = want < maxs ? want : maxs;
  const long cap = (192L << 20) / (slab_elems > 0 ? slab_elems * 4 : 1);      // keep the slab workspace under 192 MiB
  if (splits > cap) splits = cap;
  if (splits > 1024) splits = 1024;
  if (splits < 1) splits = 1;
  const long per_split = ((M + splits - 1) / splits + WG_BP - 1) / WG_BP * WG_BP;
  return (M + per_split - 1) / per_split;
}

// splits for a full descriptor (knows about the thin-A kernel, whose grid has KH*KW times fewer column tiles)
extern "C" int32_t csbsr_wgrad_splits_desc(const csbsr_wgrad_desc_t* d) {
  const int cbtot = d->b[0].c + d->b[1].c;
  const int ktot = d->KH * d->KW * cbtot;
  if (wgrad_is_thin(d))
    return (int32_t)wgrad_splits_for((cbtot + WG_BN - 1) / WG_BN, (long)d->ca * ktot, (long)d->N * d->BH * d->BW);
  if (wgrad_hr_eligible(d)) return wgrad_hr_splits(d);
  const bool perm8 = g_wgrad_tap_perm && d->KH == 8 && d->KW == 8 && d->stride == 4 && cbtot == WG_BN && d->ca <= 128;
  return wgrad_splits_impl(d->ca, ktot, (long)d->N * d->AH * d->AW, perm8);
}

// locality switches shared by both kernels: per-tile row shift of the strided layers, flat (split, tile) grid
static void wgrad_locality(WgradK& p, int BA, int BN, int splits) {
  const unsigned ntile = (unsigned)((p.ca + BA - 1) / BA) * (unsigned)((p.ktot + BN - 1) / BN);
  const long per_split = ((p.M + splits - 1) / splits + WG_BP - 1) / WG_BP * WG_BP;
  // the shift must leave the first split non-empty (slabs are written, not accumulated)
  p.row_shift = (g_wgrad_row_shift && p.stride > 1 && (long)((p.KH - 1) * p.dil / p.stride) * p.AW < per_split) ? 1 : 0;
  if (g_wgrad_flat == 2) { p.flat = 1; p.tap_perm = 0; }
  else p.flat = (g_wgrad_flat == 1 && !p.tap_perm && ntile <= 48) ? 1 : 0;   // measured: +5..25 % up to ~40 tiles, -1..2 % for the 100+ tile layers
}

template <int BA, int BN, int WA, int WB>
static int launch_wgrad(const WgradK& k, int splits, hipStream_t st) {
  WgradK p = k;
  p.tiles_a = (unsigned)((k.ca + BA - 1) / BA);
  p.tiles_b = (unsigned)((k.ktot + BN - 1) / BN);
  const unsigned ntile = p.tiles_a * p.tiles_b;
  p.per_split = ((k.M + splits - 1) / splits + WG_BP - 1) / WG_BP * WG_BP;
  if ((int)((k.M + p.per_split - 1) / p.per_split) != splits) {
    csbsr_set_error("wgrad: splits=%d leaves an empty slab; use csbsr_wgrad_splits()", splits);
    return 1;
  }
  p.splits = splits;
  wgrad_locality(p, BA, BN, splits);
  dim3 grid(p.flat ? ntile * splits : ntile, 1, p.flat ? 1 : splits);
  if (g_wgrad_use_tr)
    hipLaunchKernelGGL((conv_wgrad_kernel<BA, BN, WA, WB, true>), grid, dim3(64 * WA * WB), g_wgrad_extra_lds, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<BA, BN, WA, WB, false>), grid, dim3(64 * WA * WB), 0, st, p);
  CSBSR_LAUNCH_CHECK("csbsr_conv_wgrad");
  return 0;
}

extern "C" int csbsr_conv_wgrad(const csbsr_wgrad_desc_t* d, csbsr_stream_t s) {
  CSBSR_CHECK(d && d->a && d->b[0].ptr && d->g, "wgrad: null pointer");
  CSBSR_CHECK(d->ca % 8 == 0 && d->b[0].c % 8 == 0 && d->b[1].c % 8 == 0 && d->b[0].c > 0, "wgrad: channels must be multiples of 8");
  WgradK k;
  k.a = reinterpret_cast<const half_t*>(d->a); k.a_sn = d->a_sn; k.a_sy = d->a_sy; k.a_sx = d->a_sx; k.ca = d->ca;
  k.b[0] = d->b[0]; k.b[1] = d->b[1];
  if (k.b[1].c == 0) k.b[1] = k.b[0];
  k.cb0 = d->b[0].c; k.cbtot = d->b[0].c + d->b[1].c;
  k.N = d->N; k.AH = d->AH; k.AW = d->AW; k.BH = d->BH; k.BW = d->BW;
  k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.dil = d->dil;
  k.g = d->g; k.ktot = d->KH * d->KW * k.cbtot;
  k.tap_perm = (g_wgrad_tap_perm && d->KH == 8 && d->KW == 8 && d->stride == 4 && k.cbtot == WG_BN && d->ca <= 128) ? 1 : 0;
  k.M = (long)d->N * d->AH * d->AW;
  k.slab_stride = (long)d->ca * k.ktot; k.row0 = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  CSBSR_CHECK(d->splits >= 1, "wgrad: splits must come from csbsr_wgrad_splits()");
  if (wgrad_is_thin(d)) {
    k.ca_real = d->ca_real > 0 ? d->ca_real : d->ca;
    k.M = (long)d->N * d->BH * d->BW;
    k.per_split = ((k.M + d->splits - 1) / d->splits + WG_BP - 1) / WG_BP * WG_BP;
    if ((int)((k.M + k.per_split - 1) / k.per_split) != d->splits) {
      csbsr_set_error("wgrad(thin): splits=%d leaves an empty slab; use csbsr_wgrad_splits_desc()", d->splits);
      return 1;
    }
    dim3 grid((k.cbtot + WG_BN - 1) / WG_BN, 1, d->splits);
    g_last_wgrad_kernel = 4;
    if (g_wgrad_use_tr) hipLaunchKernelGGL((conv_wgrad_thin_kernel<true>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((conv_wgrad_thin_kernel<false>), grid, dim3(256), 0, st, k);
    CSBSR_LAUNCH_CHECK("csbsr_conv_wgrad(thin)");
    return 0;
  }
  if (wgrad_hr_eligible(d)) { g_last_wgrad_kernel = 8; return wgrad_hr_launch(d, st); }
  // LDS-DMA kernel for every problem with > 64 A-channels.  (While its DMA pieces were the compiler's global_load_lds builtin, hipcc put
  // an s_waitcnt vmcnt(0) before each stage's first transposing read and the ring never overlapped: the kernel then only matched the
  // register-staged one at equal tile size.  With the pieces as inline assembly, N = 4: SFT 825->384 747 -> 884 TF/s, 384->825 712 -> 808,
  // ResNet 512 728 -> 802, 256 631 -> 701, up_1 1024->256 793 -> 958, 8x8 stride 4 (128 x 256 tap-pair tiles) 650 -> 739.)
  const bool glds_all = wgrad_glds_eligible(k) && (g_wgrad_glds & 8);
  if (wgrad_glds_eligible(k) && (glds_all || wgrad_glds_tile_a(k) == 256)) {
    const int ta = wgrad_glds_tile_a(k), tn = wgrad_glds_tile_n(k);
    g_last_wgrad_kernel = ta == 256 ? 7 : (tn == 512 ? 9 : (tn == 256 ? 6 : 5));
    k.ca_real = 0;
    if (ta != 256) {
      wgrad_locality(k, ta, tn, d->splits);
      return wgrad_glds_launch(k, ta, tn, d->splits, st);
    }
    // rows [0, n256) on 256 x 256 tiles, the remaining (< 256) rows on 128-row tiles: same pixel splits, same slabs
    const int n256 = d->ca / 256 * 256;
    WgradK k1 = k;
    k1.ca = n256;
    wgrad_locality(k1, 256, 256, d->splits);
    int rc = wgrad_glds_launch(k1, 256, 256, d->splits, st);
    if (rc || n256 == d->ca) return rc;
    WgradK k2 = k;
    k2.a = k.a + n256; k2.ca = d->ca - n256; k2.row0 = n256;
    if (glds_all) {
      wgrad_locality(k2, 128, tn, d->splits);
      return wgrad_glds_launch(k2, 128, tn, d->splits, st);
    }
    if (k2.ca > 64) {
      if (wgrad_tile_n(k2.ca, k.ktot, false) == 256) return launch_wgrad<128, 256, 2, 4>(k2, d->splits, st);
      return launch_wgrad<128, 128, 2, 2>(k2, d->splits, st);
    }
    if (k2.ca > 32) return launch_wgrad<64, 128, 2, 2>(k2, d->splits, st);
    return launch_wgrad<32, 128, 1, 4>(k2, d->splits, st);
  }
  if (d->ca > 64) {
    if (wgrad_tile_n(d->ca, k.ktot, k.tap_perm != 0) == 256) { g_last_wgrad_kernel = 1; return launch_wgrad<128, 256, 2, 4>(k, d->splits, st); }
    g_last_wgrad_kernel = 0;
    return launch_wgrad<128, 128, 2, 2>(k, d->splits, st);
  }
  if (d->ca > 32) { g_last_wgrad_kernel = 2; return launch_wgrad<64, 128, 2, 2>(k, d->splits, st); }
  g_last_wgrad_kernel = 3;
  return launch_wgrad<32, 128, 1, 4>(k, d->splits, st);
}
