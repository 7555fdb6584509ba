// Implicit-GEMM convolution / transposed convolution for gfx950.
//
//   D[cout][pixel] = sum_k  Wp[cout][k] * X[pixel @ k]      k = tap * Ctot + channel
//
// MFMA v_mfma_f32_32x32x16_f16: "A" = packed weights (rows = output channels), "B" = gathered input patches
// (columns = output pixels), fp32 accumulators.  One 256-thread workgroup (4 wave64) owns a 128-pixel x BN-channel
// tile; K is walked in 32-wide slices, register-staged global->LDS with double buffering; both LDS tiles are
// [row][32 + 8 pad] halves so the 16-byte fragment reads are bank-conflict free.  The epilogue stages the fp32
// tile through LDS so that every global store / residual load is a full 16-byte, channel-contiguous access.
//
// A transposed convolution (stride s) is run in its gather form: blockIdx.z selects one of the s*s output
// residues ("phases"); within a phase it is an ordinary convolution with ceil(K/s)^2 taps walking the input
// backwards, so conv, deconv and every dgrad share this one kernel (weights re-packed by pack.hip).
//
// Reference call sites this stands in for: F.conv2d / F.conv_transpose2d in
// /root/reference/model/modeling/kbpn.py:241,273-277,513-517 and pspnet_pytorch/{extractors.py:36-38,pspnet.py:30-86}.
#include "common.h"
#include "csbsr_debug.h"
#include "conv_common.h"
#include <type_traits>

#define BM 128
#define BK 64
#define NTAP_MAX 160   // taps per phase the tap tables hold (12 x 12 is the largest kernel on the path)
#define LDS_LD 72  // halves per LDS row: 64 + 8 pad (144 B): conflict-free 16-byte fragment reads

__device__ __attribute__((aligned(16))) const half_t g_zero_line[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // what out-of-image taps read
#ifdef CSBSR_TS
__device__ unsigned long long g_its[8 * 262144];
#define ITS(i) do { if (threadIdx.x == 0 && blockIdx.x < 262144) g_its[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
extern "C" int csbsr_debug_read_its(void* dst, long n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_its), n * 8); }
#else
#define ITS(i)
#endif
template <int BN, int WP, int WC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 ? 2 : (BN == 64 ? 3 : 4)))) void conv_igemm_kernel(const ConvK p) {
  constexpr int PW = BM / WP;      // pixels per wave
  constexpr int CW = BN / WC;      // couts per wave
  constexpr int TP = PW / 32;
  constexpr int TC = CW / 32;
  constexpr int NH = BN == 128 ? 2 : 1;   // epilogue staged in NH halves of HB couts
  constexpr int HB = BN / NH;
  constexpr int OUT_LD = HB + 4;   // fp32 staging row
  constexpr int MAIN_BYTES = (BN + BM) * LDS_LD * 2;
  constexpr int EPI_BYTES = BM * OUT_LD * 4;
  constexpr int SM_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char* smem = smem_dyn;
  half_t* sW = reinterpret_cast<half_t*>(smem);                       // [BN][LDS_LD]
  half_t* sX = sW + BN * LDS_LD;                                      // [BM][LDS_LD]
  int* sRow = reinterpret_cast<int*>(smem + SM_BYTES);                // [BM][3]  n, oy, ox  (n = -1: invalid)
  float* sStat = reinterpret_cast<float*>(smem + SM_BYTES + BM * 3 * 4);  // [2][BN]
  float* sBias = sStat + 2 * BN;                                           // [BN]: fetched before the K loop, not inside the epilogue
  long* sOOff = reinterpret_cast<long*>(sBias + BN);                       // [BM]: element offset of each tile row's output pixel in out16
  long* sTap0 = sOOff + BM;                                                // [NTAP_MAX] element displacement of tap t in input segment 0
  long* sTap1 = sTap0 + NTAP_MAX;                                          // ... in segment 1
  int* sTapYX = reinterpret_cast<int*>(sTap1 + NTAP_MAX);                  // [NTAP_MAX] (ty, tx) packed, -1 past the last tap (K padding)

  ITS(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wp = wid / WC, wc = wid % WC;

  // ---- tile coordinates (XCD-aware: one XCD walks consecutive tiles, cout tiles fastest)
  const unsigned ntile = p.tiles_m * p.tiles_n;
  const unsigned lt = xcd_remap(blockIdx.x, ntile);
  const int tile_n = lt % p.tiles_n, tile_m = lt / p.tiles_n;
  const int cout0 = tile_n * BN;

  // ---- phase geometry
  int py = 0, px = 0, OHp = p.OH, OWp = p.OW, in_step = p.stride, tap_step = p.dil, base_y = -p.pad, base_x = -p.pad;
  int o_step = 1;
  if (p.transposed) {
    py = blockIdx.z / p.stride; px = blockIdx.z % p.stride;
    OHp = (p.OH - py + p.stride - 1) / p.stride;
    OWp = (p.OW - px + p.stride - 1) / p.stride;
    in_step = 1; tap_step = -1; o_step = p.stride;
    base_y = (py + p.pad) / p.stride; base_x = (px + p.pad) / p.stride;
  }
  const long M = p.hw_pad ? (long)p.N * p.hw_pad : (long)p.N * OHp * OWp;      // hw_pad: per-sample padded pixel index (fused per-sample sums)
  const long m0 = (long)tile_m * BM;
  if (m0 >= M) return;
  const half_t* wt = p.wt + (size_t)blockIdx.z * p.rows_p * p.Kp;

  if (tid < BM) {
    long m = m0 + tid;
    int n = -1, oy = 0, ox = 0;
    if (m < M) {
      int rem;
      if (M < (1l << 31)) {      // 32-bit divisions (the 64-bit one is a ~100-instruction routine in front of the prologue barrier)
        const unsigned hw = p.hw_pad ? (unsigned)p.hw_pad : (unsigned)(OHp * OWp);
        n = (int)((unsigned)m / hw); rem = (int)((unsigned)m - (unsigned)n * hw);
      } else {
        const long hw = p.hw_pad ? (long)p.hw_pad : (long)OHp * OWp;
        n = (int)(m / hw); rem = (int)(m - (long)n * hw);
      }
      if (rem >= OHp * OWp) { n = -1; rem = 0; }      // padding position of a per-sample layout
      oy = (int)((unsigned)rem / (unsigned)OWp); ox = rem - oy * OWp;
    }
    sRow[tid * 3 + 0] = n; sRow[tid * 3 + 1] = oy; sRow[tid * 3 + 2] = ox;
    sOOff[tid] = n * p.o_sn + (long)(py + oy * o_step) * p.o_sy + (long)(px + ox * o_step) * p.o_sx;
  }
  if (tid < NTAP_MAX) {        // tap tables: the K loop looks a tap's displacement up instead of carrying 64-bit running sums per thread
    const int ty = tid / p.KWt, tx = tid - ty * p.KWt;
    const bool live = tid < p.KHt * p.KWt;
    sTap0[tid] = live ? (long)tap_step * (ty * p.in[0].sy + tx * p.in[0].sx) : 0;
    sTap1[tid] = live ? (long)tap_step * (ty * p.in[1].sy + tx * p.in[1].sx) : 0;
    sTapYX[tid] = live ? (ty | (tx << 16)) : -1;
  }
  if (tid < 2 * BN) sStat[tid] = 0.f;
  // (a per-sample bias: the tile lies within one sample -- OH * OW is a multiple of the tile, checked by the launcher)
  if (tid < BN) sBias[tid] = (p.bias && cout0 + tid < p.cout) ? p.bias[(p.bias_sn ? (m0 / ((long)OHp * OWp)) * p.bias_sn : 0) + cout0 + tid] : 0.f;
  const float slope = (p.act == CSBSR_ACT_PRELU) ? *p.prelu : p.act_slope;
  __syncthreads();
  ITS(1);

  // ---- per-thread gather state: rows tid/8 + 32 j (j < 4); 16-byte k-segment seg = tid%8 (one K slice = 64 channels of one
  // tap = a full 128-byte line per pixel)
  constexpr int SEGS = BK / 8;                // 8
  constexpr int RPP = 256 / SEGS;             // rows per pass: 32
  constexpr int XCH = BM / RPP;               // 4 patch chunks per thread
  const int seg = tid % SEGS;
  // hoisted out of the K loop: per row a 64-bit element offset for tap (0,0) in each segment and one validity bit per tap;
  // per slice only the thread's (tap, channel) cursor and two running tap displacements change
  long off0[XCH], off1[XCH];
  unsigned long long tapmask[XCH];
  const int ntaps = p.KHt * p.KWt;
  const bool use_mask = ntaps <= 64 && p.Kp >= 12 * BK;     // short-K layers: the precompute costs more than it saves (measured)
  int rn[XCH], riy[XCH], rix[XCH];
#pragma unroll
  for (int j = 0; j < XCH; ++j) {
    const int r = tid / SEGS + RPP * j;
    rn[j] = sRow[r * 3];
    riy[j] = sRow[r * 3 + 1] * in_step + base_y;
    rix[j] = sRow[r * 3 + 2] * in_step + base_x;
    off0[j] = rn[j] * p.in[0].sn + riy[j] * p.in[0].sy + rix[j] * p.in[0].sx;
    off1[j] = rn[j] * p.in[1].sn + riy[j] * p.in[1].sy + rix[j] * p.in[1].sx;
    unsigned long long m = 0;
    if (rn[j] >= 0 && use_mask) {
      int iy = riy[j], ix = rix[j], tkx = 0;          // walk the taps without integer division
      for (int t = 0; t < ntaps; ++t) {
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) m |= 1ull << t;
        ix += tap_step;
        if (++tkx == p.KWt) { tkx = 0; ix = rix[j]; iy += tap_step; }
      }
    }
    tapmask[j] = m;
  }
  // running (tap, channel) cursor of this thread's 16-byte segment.  Per slice the thread does: two table reads for the tap, one
  // pointer select, then per chunk a validity test and a pointer select between the pixel and a zero line -- ~60 VALU instructions
  // where the running-displacement version (branchy next-tap updates, predicated loads with zero-initialised destinations) had
  // ~150, against 4 MFMAs per wave and slice on the 32-cout tile (the K loop of the HR 32-channel layers was VALU-bound).
  int kc = seg * 8, tapi = 0;
  while (kc >= p.ctot) { kc -= p.ctot; ++tapi; }
  const half_t* const zline = g_zero_line;

  constexpr int WCH = (BN * SEGS + 255) / 256;  // weight chunks per thread
  // register-staged slices: two sets in flight for the narrow tiles (a 32-cout slice is 4 MFMAs per wave -- 128 clocks -- against a
  // ~1.3 us load round trip: CSBSR_TS showed the K loop of the HR 32-channel layers waiting one full round trip per slice)
  constexpr bool DEEP = BN <= 32;
  h8 gx0[XCH], gw0[WCH], gx1[DEEP ? XCH : 1], gw1[DEEP ? WCH : 1];

  // weight chunks: a running pointer per chunk (rows past the padded weight rows read the zero line and do not move)
  const half_t* wsrc[WCH];
  int wstep[WCH];
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    const int c = tid + 256 * i;
    const int row = c / SEGS, sg = c % SEGS;
    const bool ok = row < BN && cout0 + row < p.rows_p;
    wsrc[i] = ok ? wt + (size_t)(cout0 + row) * p.Kp + sg * 8 : zline;
    wstep[i] = ok ? BK : 0;
  }
  const half_t* const in0 = reinterpret_cast<const half_t*>(p.in[0].ptr);
  const half_t* const in1 = reinterpret_cast<const half_t*>(p.in[1].ptr) - p.c0;       // indexed by the concatenated channel
  auto load_tile = [&](int kt, auto& gx, auto& gw) {
#pragma unroll
    for (int i = 0; i < WCH; ++i) { gw[i] = *reinterpret_cast<const h8*>(wsrc[i]); wsrc[i] += wstep[i]; }
    // patches
    const bool kvalid = tapi < ntaps;
    const int tq = kvalid ? tapi : 0;
    const int yx = sTapYX[tq];
    const bool s0 = kc < p.c0;
    const half_t* base = (s0 ? in0 + sTap0[tq] : in1 + sTap1[tq]) + kc;
    const int ky = yx & 0xffff, kx = yx >> 16;
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
      bool ok;
      if (use_mask) ok = kvalid && ((tapmask[j] >> (tapi & 63)) & 1ull);
      else {
        const int iy = riy[j] + ky * tap_step, ix = rix[j] + kx * tap_step;
        ok = kvalid && rn[j] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      }
      const half_t* src = ok ? base + (s0 ? off0[j] : off1[j]) : zline;
      gx[j] = *reinterpret_cast<const h8*>(src);
    }
    // advance by one K slice
    kc += BK;
    while (kc >= p.ctot) { kc -= p.ctot; ++tapi; }
  };
  auto store_tile = [&](const auto& gx, const auto& gw) {
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int c = tid + 256 * i;
      const int row = c / SEGS, sg = c % SEGS;
      if (row < BN) *reinterpret_cast<h8*>(sW + row * LDS_LD + sg * 8) = gw[i];
    }
#pragma unroll
    for (int j = 0; j < XCH; ++j) *reinterpret_cast<h8*>(sX + (tid / SEGS + RPP * j) * LDS_LD + seg * 8) = gx[j];
  };

  f16v acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // single LDS buffer + register prefetch: the next slice's HBM loads fly under this slice's 16 MFMAs per wave
  const int nkt = p.Kp / BK;
  const half_t* w_base = sW + (wc * CW + (lane & 31)) * LDS_LD + (lane >> 5) * 8;
  const half_t* x_base = sX + (wp * PW + (lane & 31)) * LDS_LD + (lane >> 5) * 8;
  ITS(2);
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      h8 af[TC], bf[TP];
#pragma unroll
      for (int a = 0; a < TC; ++a) af[a] = *reinterpret_cast<const h8*>(w_base + a * 32 * LDS_LD + ks * 16);
#pragma unroll
      for (int b = 0; b < TP; ++b) bf[b] = *reinterpret_cast<const h8*>(x_base + b * 32 * LDS_LD + ks * 16);
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  };
  load_tile(0, gx0, gw0);
  ITS(3);
  if constexpr (DEEP) {
    if (nkt > 1) load_tile(1, gx1, gw1);
    for (int kt = 0; kt < nkt; kt += 2) {
      if (kt > 0) __syncthreads();              // previous slice's fragment reads done
      store_tile(gx0, gw0);
      __syncthreads();
      if (kt + 2 < nkt) load_tile(kt + 2, gx0, gw0);
      compute();
      if (kt + 1 < nkt) {
        __syncthreads();
        store_tile(gx1, gw1);
        __syncthreads();
        if (kt + 3 < nkt) load_tile(kt + 3, gx1, gw1);
        compute();
      }
    }
  } else {
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt > 0) __syncthreads();              // previous slice's fragment reads done
      store_tile(gx0, gw0);
      __syncthreads();
      if (kt + 1 < nkt) load_tile(kt + 1, gx0, gw0);
      compute();
    }
  }
  ITS(4);
  __syncthreads();

  // ---- epilogue: fp32 tile -> LDS [pixel][cout] in halves of HB couts (keeps LDS <= the main-loop footprint so three
  // workgroups fit per CU), then channel-contiguous 8-wide processing
  float* sO = reinterpret_cast<float*>(smem);
  const EpiFast fe = conv_epilogue_fast_setup(p, slope);
  constexpr int CPR = HB / 8;                 // 8-channel chunks per staged row
  const int cc8 = tid % CPR;                  // fixed per thread (256 % CPR == 0)

#pragma unroll 1
  for (int hh = 0; hh < NH; ++hh) {
    if (hh > 0) lds_barrier();                // previous half fully consumed (LDS only: no wait for its global stores)
#pragma unroll
    for (int a = 0; a < TC; ++a) {
      const int cbase = wc * CW + a * 32;     // wave-uniform
      if (cbase / HB != hh) continue;
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        const int pix = wp * PW + b * 32 + (lane & 31);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cl = cbase - hh * HB + 8 * q + 4 * (lane >> 5);
          f4 v = {acc[a][b][4 * q + 0], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
          *reinterpret_cast<f4*>(sO + pix * OUT_LD + cl) = v;
        }
      }
    }
    lds_barrier();
    if (hh == 0) ITS(5);

    const int co = cout0 + hh * HB + cc8 * 8;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = sBias[hh * HB + cc8 * 8 + e];
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;

    constexpr int RSTEP = 256 / CPR;
    if (fe.ok) {          // straight-line rows (conv_common.h)
      constexpr int RPT = BM / RSTEP, EG = RPT >= 2 ? 2 : RPT;      // (4 rows in flight cost the 64-cout kernel its third wave per SIMD)
      static_assert(RPT % EG == 0, "rows per thread must split into groups");
      auto rows = [&](auto EXTRA, auto BNSTAT) {
#pragma unroll 1
        for (int g = 0; g < RPT / EG; ++g) {
          h8 rr[EG], oo[EG], mm[EG];
          long ooff[EG];
          bool live[EG];
#pragma unroll
          for (int i = 0; i < EG; ++i) {
            const int grow = tid / CPR + (g * EG + i) * RSTEP;
            const int n = sRow[grow * 3];
            live[i] = n >= 0 && co < p.coutp;
            ooff[i] = sOOff[grow] + co;
            rr[i] = h8{0, 0, 0, 0, 0, 0, 0, 0}; oo[i] = h8{0, 0, 0, 0, 0, 0, 0, 0}; mm[i] = h8{1, 1, 1, 1, 1, 1, 1, 1};
            if constexpr (decltype(EXTRA)::value) {
              if (live[i]) {
                if (fe.has_res || fe.has_mask) {
                  const int oyo = py + sRow[grow * 3 + 1] * o_step, oxo = px + sRow[grow * 3 + 2] * o_step;
                  if (fe.has_res) rr[i] = *reinterpret_cast<const h8*>(p.res + n * p.r_sn + oyo * p.r_sy + oxo * p.r_sx + co);
                  if (fe.has_mask) mm[i] = *reinterpret_cast<const h8*>(p.mask + n * p.m_sn + oyo * p.m_sy + oxo * p.m_sx + co);
                }
                if (fe.has_old) oo[i] = *reinterpret_cast<const h8*>(p.out16 + ooff[i]);
              }
            }
          }
#pragma unroll
          for (int i = 0; i < EG; ++i) {
            if (!live[i]) continue;
            const int row = tid / CPR + (g * EG + i) * RSTEP;
            const f4 v0 = *reinterpret_cast<const f4*>(sO + row * OUT_LD + cc8 * 8);
            const f4 v1 = *reinterpret_cast<const f4*>(sO + row * OUT_LD + cc8 * 8 + 4);
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (fe.has_cb) {      // (workgroup-uniform) the pixel's class row joins the bias
              float brow[8];
              conv_class_bias_row(p, bias, co, sRow[row * 3], py + sRow[row * 3 + 1] * o_step, px + sRow[row * 3 + 2] * o_step, brow);
              conv_epilogue_fast_row<decltype(EXTRA)::value, decltype(BNSTAT)::value>(fe, v, brow, co, p.out16 + ooff[i], rr[i], oo[i], ssum, ssq, mm[i]);
              continue;
            }
            conv_epilogue_fast_row<decltype(EXTRA)::value, decltype(BNSTAT)::value>(fe, v, bias, co, p.out16 + ooff[i], rr[i], oo[i], ssum, ssq, mm[i]);
          }
        }
      };
      const bool extra = fe.has_res || fe.has_old || fe.has_mask;
      if (fe.bn) { if (extra) rows(std::true_type{}, std::true_type{}); else rows(std::false_type{}, std::true_type{}); }
      else { if (extra) rows(std::true_type{}, std::false_type{}); else rows(std::false_type{}, std::false_type{}); }
      if (fe.bn) conv_epilogue_flush_stats<CPR, 4>(p, sStat, BN, hh * HB + cc8 * 8, co, ssum, ssq);
      continue;
    }
    // rows in groups of EG: the group's residual / old-output loads are all issued before the first row is combined
    constexpr int RPT = BM / RSTEP, EG = RPT >= 2 ? 2 : 1;
#pragma unroll 1      // one copy of the (large, mode-rich) row code: fully unrolled the kernel was 23 K instructions
    for (int g = 0; g < RPT / EG; ++g) {
      EpiPre pre[EG];
      int rn[EG], roy[EG], rox[EG];
#pragma unroll
      for (int i = 0; i < EG; ++i) {
        const int grow = tid / CPR + (g * EG + i) * RSTEP;
        const int n = sRow[grow * 3];
        rn[i] = (n < 0 || co >= p.coutp) ? -1 : n;
        roy[i] = py + sRow[grow * 3 + 1] * o_step;
        rox[i] = px + sRow[grow * 3 + 2] * o_step;
        if (rn[i] >= 0) conv_epilogue_prefetch(p, co, rn[i], roy[i], rox[i], pre[i]);
      }
#pragma unroll
      for (int i = 0; i < EG; ++i) {
        if (rn[i] < 0) continue;
        const int row = tid / CPR + (g * EG + i) * RSTEP;
        float v[8];
        const f4 v0 = *reinterpret_cast<const f4*>(sO + row * OUT_LD + cc8 * 8);
        const f4 v1 = *reinterpret_cast<const f4*>(sO + row * OUT_LD + cc8 * 8 + 4);
        v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3]; v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
        conv_epilogue_row(p, v, bias, slope, co, rn[i], roy[i], rox[i], ssum, ssq, &pre[i]);
      }
    }

    conv_epilogue_flush_stats<CPR, 4>(p, sStat, BN, hh * HB + cc8 * 8, co, ssum, ssq);
  }

  ITS(6);
  conv_epilogue_store_stats(p, sStat, BN, cout0, (size_t)blockIdx.z * p.tiles_m + tile_m);      // (the last flush ended with a barrier)
}

template <int BN, int WP, int WC>
static int launch_conv(const ConvK& k, int nphase, long maxM, hipStream_t st) {
  ConvK p = k;
  p.tiles_m = (unsigned)((maxM + BM - 1) / BM);
  p.tiles_n = (unsigned)((k.coutp + BN - 1) / BN);
  constexpr int OUT_LD = (BN == 128 ? 64 : BN) + 4;
  constexpr int MAIN_BYTES = (BN + BM) * LDS_LD * 2;
  constexpr int EPI_BYTES = BM * OUT_LD * 4;
  constexpr int SM_BYTES = (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES) + BM * 3 * 4 + 3 * BN * 4 + BM * 8 + NTAP_MAX * 20;
  static LdsAttrOnce attr;
  if (int e = csbsr_lds_attr(attr, reinterpret_cast<const void*>(conv_igemm_kernel<BN, WP, WC>), SM_BYTES, "conv")) return e;
  ConvStatPlan sp;
  if (int e = conv_stat_prepare(p, BM, nphase, sp, st)) return e;
  dim3 grid(p.tiles_m * p.tiles_n, 1, nphase);
  hipLaunchKernelGGL((conv_igemm_kernel<BN, WP, WC>), grid, dim3(256), SM_BYTES, st, p);
  if (int e = conv_stat_finish(p, sp, st)) return e;
  CSBSR_LAUNCH_CHECK("csbsr_conv_forward");
  return 0;
}

// descriptor -> kernel-argument block (validation included); shared with csrc/conv_x3.hip
int conv_desc_to_k(const csbsr_conv_desc_t* d, ConvK& k) {
  CSBSR_CHECK(d && d->in[0].ptr && d->wt, "conv: null pointer");
  CSBSR_CHECK(d->in[0].c > 0 && d->in[0].c % 8 == 0 && d->in[1].c % 8 == 0, "conv: segment channels must be multiples of 8");
  CSBSR_CHECK(d->coutp % 8 == 0 && d->cout <= d->coutp && d->cout > 0, "conv: bad cout/coutp");
  CSBSR_CHECK(d->out16 || d->out32 || d->stat_mode != CSBSR_STAT_NONE, "conv: no output requested");
  CSBSR_CHECK(d->stride >= 1 && d->KH >= 1 && d->KW >= 1, "conv: bad geometry");
  CSBSR_CHECK(!d->transposed || d->dil == 1, "conv: transposed conv supports dilation 1 only");
  CSBSR_CHECK(!d->cbias || (!d->transposed && d->stride == 1), "conv: a position-class bias needs a stride-1 convolution");
  CSBSR_CHECK(!d->cbias || d->cbias_mode == 0 || (d->cbias_mode == 1 && d->OH >= 5 && d->OW >= 5), "conv: cbias_mode 1 needs OH, OW >= 5");
  CSBSR_CHECK(d->act != CSBSR_ACT_PRELU || d->prelu, "conv: PReLU needs a slope pointer");
  CSBSR_CHECK(d->res_mode == CSBSR_RES_NONE || d->res, "conv: res_mode set without res");
  k.in[0] = d->in[0]; k.in[1] = d->in[1];
  if (k.in[1].c == 0) k.in[1] = k.in[0];
  k.N = d->N; k.H = d->H; k.W = d->W; k.OH = d->OH; k.OW = d->OW;
  k.transposed = d->transposed;
  k.stride = d->stride; k.pad = d->pad; k.dil = d->dil;
  k.KHt = d->transposed ? (d->KH + d->stride - 1) / d->stride : d->KH;
  k.KWt = d->transposed ? (d->KW + d->stride - 1) / d->stride : d->KW;
  k.c0 = d->in[0].c; k.ctot = d->in[0].c + d->in[1].c;
  k.Kp = round_up(k.KHt * k.KWt * k.ctot, BK);
  k.rows_p = conv_rows_padded(d->cout);
  k.wt = reinterpret_cast<const half_t*>(d->wt);
  k.cout = d->cout; k.coutp = d->coutp;
  k.out16 = reinterpret_cast<half_t*>(d->out16); k.o_sn = d->o_sn; k.o_sy = d->o_sy; k.o_sx = d->o_sx;
  k.out32 = d->out32; k.o32_sn = d->o32_sn; k.o32_sy = d->o32_sy; k.o32_sx = d->o32_sx; k.o32_sc = d->o32_sc;
  k.bias = d->bias; k.cbias = d->cbias; k.cb_mode = d->cbias_mode; k.act = d->act; k.act_slope = d->act_slope; k.prelu = d->prelu;
  k.res_mode = d->res_mode; k.res = reinterpret_cast<const half_t*>(d->res);
  k.r_sn = d->r_sn; k.r_sy = d->r_sy; k.r_sx = d->r_sx;
  k.res2 = reinterpret_cast<const half_t*>(d->res2); k.r2_sn = d->r2_sn; k.r2_sy = d->r2_sy; k.r2_sx = d->r2_sx;
  CSBSR_CHECK(d->res_mode != CSBSR_RES_FMA || d->res2, "conv: FMA needs res2");
  k.accumulate = d->accumulate; k.stat_mode = d->stat_mode; k.stat = d->stat;
  k.out_scale = d->out_scale;
  k.o_lo = d->o_lo; k.r_lo = d->r_lo; k.r2_lo = d->r2_lo;
  k.mask = reinterpret_cast<const half_t*>(d->mask); k.m_sn = d->m_sn; k.m_sy = d->m_sy; k.m_sx = d->m_sx; k.mask_slope = d->mask_slope;
  k.mask_prelu = d->mask_prelu;
  CSBSR_CHECK(!d->mask_prelu || d->mask, "conv: mask_prelu without a mask");
  CSBSR_CHECK(!d->mask || (d->out16 && !d->o_lo), "conv: the activation mask applies to a plain fp16 output");
  CSBSR_CHECK((!d->dact_bias && !d->dact_prelu && !d->dres) || csbsr_conv_thin_dact_eligible(d),
              "conv: dact_* / dres: csbsr_conv_tp_forward, or the thin-input accumulating dgrad (csbsr_conv_thin_dact_eligible)");
  CSBSR_CHECK(!(d->o_lo && d->accumulate), "conv: a split (hi + lo) output cannot accumulate");
  CSBSR_CHECK(!d->o_lo || d->out16, "conv: o_lo without out16");
  k.tile2d = 0; k.nphase_flat = 0; k.tap_group = 0;
  k.stat_part = nullptr; k.stat_ld = 0; k.hw_pad = 0;
  k.bias_sn = d->bias_sn;
  CSBSR_CHECK(!d->bias_sn || (d->bias && !d->transposed && ((long)d->OH * d->OW) % 256 == 0),
              "conv: a per-sample bias needs a non-transposed layer whose samples are whole pixel tiles (OH * OW a multiple of 256)");
  k.fs = d->split_fused == 2 ? 2 : (d->split_fused ? 1 : 0);      // 2: the fused stage without the x_hi w_lo product (two-product plan)
  CSBSR_CHECK(!k.fs || (d->in[1].c == 0 && d->in[0].c % 16 == 0 && d->in[0].c >= 64 && !d->transposed), "conv: split_fused needs one [hi | lo] segment of 2 x (>= 32, a multiple of 8) channels");
  CSBSR_CHECK(d->stat_mode == CSBSR_STAT_NONE || d->stat, "conv: stat_mode set without stat buffer");
  return 0;
}

// 1 if csbsr_conv_forward would take this descriptor WITH split_fused set (the fused three-product stage exists only in the LDS-DMA
// kernels): the host asks before it packs the [w_hi 32 | w_lo 32] operand and falls back to the three-block form otherwise
extern "C" int32_t csbsr_conv_split_fused_eligible(const csbsr_conv_desc_t* d) {
  if (!d || !d->split_fused) return 0;
  ConvK k;
  if (conv_desc_to_k(d, k)) return 0;
  return conv_glds_eligible(k) ? 1 : 0;
}

extern "C" int csbsr_conv_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s) {
  ConvK k;
  if (int rc = conv_desc_to_k(d, k)) return rc;
  int nphase = 1;
  long maxM;
  if (d->transposed) {
    nphase = d->stride * d->stride;
    maxM = (long)d->N * ((d->OH + d->stride - 1) / d->stride) * ((d->OW + d->stride - 1) / d->stride);
  } else {
    maxM = (long)d->N * d->OH * d->OW;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  const bool split_io = d->o_lo || d->r_lo || d->r2_lo || d->mask;     // the thin kernels have their own epilogues: plain fp16, no mask
  if (d->dres || d->dact_prelu || d->dact_bias) {      // a launch that takes over the layer below's epilogue-backward pass: only the kernels built for it
    CSBSR_CHECK(csbsr_conv_thin_dact_eligible(d), "conv: dact / dres launch not eligible here (csbsr_conv_tp_forward or the thin-input accumulating dgrad)");
    g_last_conv_kernel = CONVK_THIN_CIN2;
    return conv_thin_cin2_dact_launch(k, d, st);
  }
  if (!split_io && conv_thin_eligible(k)) { g_last_conv_kernel = CONVK_THIN_COUT; return conv_thin_launch(k, st); }
  if (!split_io && d->in[1].c == 0 && conv_thin_cin2_eligible(k, d->in[0].creal)) {
    g_last_conv_kernel = CONVK_THIN_CIN2;
    return conv_thin_cin2_launch(k, d->in[0].creal, st);
  }
  if (!split_io && !d->bias_sn && d->in[1].c == 0 && conv_thin_cin_eligible(k, d->in[0].creal)) {
    g_last_conv_kernel = CONVK_THIN_CIN;
    return conv_thin_cin_launch(k, d->in[0].creal, st);
  }
  if (!split_io && !d->bias_sn && conv_thin_tp_eligible(k, d->in[0].creal, d->in[1].c != 0)) { g_last_conv_kernel = CONVK_THIN_TP; return conv_thin_tp_launch(k, st); }
  if (!d->r_lo && !d->r2_lo && conv_thin_sc_eligible(k)) { g_last_conv_kernel = CONVK_THIN_SC; return conv_thin_sc_launch(k, st); }
  if (!split_io && d->in[1].c == 0 && conv_thin_tpd_eligible(k, d->KH)) { g_last_conv_kernel = CONVK_THIN_TPD; return conv_thin_tpd_launch(k, d->KH, st); }
  if (conv_glds_eligible(k)) return conv_glds_launch(k, nphase, maxM, st);
  CSBSR_CHECK(!k.fs, "conv: split_fused launch not eligible for the LDS-DMA kernels (needs > 32 padded output channels)");
  CSBSR_CHECK(k.KHt * k.KWt <= NTAP_MAX, "conv: more taps per phase than the tap tables hold");
  g_last_conv_kernel = k.coutp > 64 ? CONVK_IGEMM128 : (k.coutp > 32 ? CONVK_IGEMM64 : CONVK_IGEMM32);
  if (k.coutp > 64) return launch_conv<128, 2, 2>(k, nphase, maxM, st);
  if (k.coutp > 32) return launch_conv<64, 2, 2>(k, nphase, maxM, st);
  return launch_conv<32, 4, 1>(k, nphase, maxM, st);
}

thread_local int g_last_conv_kernel = -1;
extern "C" int32_t csbsr_debug_last_conv_kernel(void) { return g_last_conv_kernel; }
