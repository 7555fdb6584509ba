// Implicit-GEMM convolution, LDS-DMA variant for the MFMA-bound layers (>= 128 output channels, channel counts that are
// multiples of 64): same math and epilogue as conv_igemm.hip, different data movement.
//
//  * one K slice = 64 channels of ONE tap of ONE input segment, i.e. a full 128-byte line per pixel / per weight row;
//  * both tiles go HBM -> LDS with global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass -- the register-staged
//    kernel is bound by the ~79 B/clk ds_write_b128 path);
//  * the LDS image is lane-linear (1 KiB per wave instruction), so bank conflicts are removed by permuting the SOURCE
//    chunk a lane fetches: position (row, c') holds channel chunk c' ^ ((row >> 1) & 7); fragment reads apply the same XOR
//    (cdna guide rule 21) -> conflict-free ds_read_b128;
//  * NSTAGE-deep ring (3 x 48 KiB for the 256x128 tile, one workgroup of 8 waves per CU = 2 waves per SIMD; 2 x 32 KiB for
//    128x128, two workgroups per CU), counted s_waitcnt vmcnt(N) + raw s_barrier so DMAs stay in flight across barriers;
//  * halo / padding / tile-overhang lanes fetch from a 256-byte zero page instead of branching.
#include "common.h"
#include "csbsr_debug.h"
#include "conv_common.h"
#include <type_traits>

#ifdef CSBSR_TS
__device__ unsigned long long g_ts[8 * 262144];
#define TS(i) do { if (threadIdx.x == 0 && blockIdx.x < 262144) g_ts[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
extern "C" int csbsr_debug_read_ts(void* dst, long n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ts), n * 8); }
#else
#define TS(i)
#endif
// FS (fused split-fp16 input, csbsr_conv_desc_t::split_fused): the input segment is a [hi | lo] plane pair and a K slice holds 32
// channels of x_hi and the same 32 of x_lo (pixel tile) next to 32 of w_hi and of w_lo (weight tile, csbsr_pack_weights_split layout 3).
// The three products of the split arithmetic -- x_hi w_hi + x_lo w_hi + x_hi w_lo -- are all fed from that ONE staged slice: six
// k-slice MFMA groups per stage instead of four, i.e. the 3x MFMA work of the precision mode for 2x the staged bytes (the
// three-block form stages x_hi twice and w_hi twice: 3x).  The kernel is bound by its stage refills, so the
// launch time follows the staged bytes.
// GK (general K walk): channel counts that are not whole 64-channel slices (HRNet-W48's 48 / 96-channel branches and its 720-channel
// concat, /root/reference/model/modeling/hrnet_ocr/backbones/hrnet/hrnet_backbone.py:108-286).  K is the flat (tap, channel) index
// of the packed weights in 8-channel units; a staged slice is 8 consecutive units (FS: 4, for both planes) and may straddle taps, so
// every lane tracks the (tap, channel) of ITS unit, looks the tap's displacement up in a small LDS table and tests that tap's bit
// of the row's validity mask.  One input segment, tap-major order (these layers' tiles re-touch a small window: no L2 issue).
// (Measured and rejected, round 4: the two waves of a SIMD run half a stage apart -- wave groups 0-3 / 4-7 alternating between a LOAD
// section and an MFMA section with two barriers per stage and s_setprio, each group refilling its own half of the pixel tile and half
// of the weight rows where no reader is left, bit-exact and race-free in the kernel tests -- 838 vs 935 TF/s on the ResNet 512 -> 512
// layer, 974 vs 1090 on up_1: the second barrier per stage costs more than the role split buys.  Like round 3's half-stage offset
// of the DMA issue, it says the stage is not waiting for instruction issue of the other wave.)
template <int BM, int NWM, int NSTAGE, int CT = 1, bool FS = false, bool GK = false>     // CT: 128-cout tiles per workgroup (2: a 256-cout tile, every wave
                                                       // 64 px x 128 couts; 0: a 64-cout tile for the 33..64-cout layers, every wave 64 px x 32 couts)
__global__ __launch_bounds__(NWM * 128) void conv_igemm_glds_kernel(const ConvK p, const half_t* __restrict__ zero_page) {
  constexpr int BN = CT ? 128 * CT : 64, BKG = 64;
  constexpr int TA = CT ? 2 * CT : 1;                // 32-cout MFMA tiles per wave
  constexpr int SOW = BN > 128 ? 128 : BN;           // couts of the fp32 tile the epilogue stages at a time
  constexpr int NW = NWM * 2, NT = NW * 64;
  constexpr int XI = BM / 8, WI = BN / 8;            // wave-instructions per stage for the X / W tile
  constexpr int NI = (XI + WI) / NW;                 // per wave
  constexpr int NXI = XI / NW;                       // X instructions per wave (the first NXI of its NI)
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;
  constexpr int OUT_LD = 64 + 4;
  constexpr int EPI_BYTES = BM * SOW * 4;
  constexpr int SM_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
  static_assert(XI % NW == 0 && WI % NW == 0, "tile rows must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // per tile-row tables, filled by ONE thread per row in the prologue (each of the 8 lanes that DMA a row, and each of the 16 threads
  // that later store it, used to redo the 64-bit address products and the tap walk: 26 us of a 200 us workgroup on the 64-tap layers,
  // 2 us of 18 on the transposed ones)
  long* sOff0 = reinterpret_cast<long*>(smem + SM_BYTES);                    // [BM] element offset of tap (0,0) in input segment 0
  long* sOff1 = sOff0 + BM;                                                  // [BM] ... in segment 1
  unsigned long long* sMask = reinterpret_cast<unsigned long long*>(sOff1 + BM);   // [BM] bit t: tap t lands inside the image
  long* sOOff = reinterpret_cast<long*>(sMask + BM);                         // [BM] element offset of the output pixel in out16
  int* sRow = reinterpret_cast<int*>(sOOff + BM);                            // [BM][3] (n, oy, ox) of the phase grid; n = -1: no pixel
  float* sStat = reinterpret_cast<float*>(sRow + BM * 3);                    // [2][BN]
  float* sBias = sStat + 2 * BN;                                             // [BN]: this tile's bias
  long* sDisp = reinterpret_cast<long*>(sBias + BN);                         // GK: [64] element displacement of tap t from tap (0, 0)

  TS(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: LDS destinations / m0 stay on the scalar unit
  const int wm = wid >> 1, wn = wid & 1;
  const unsigned ntile = p.tiles_m * p.tiles_n;
  // Transposed layers: the stride^2 output phases of one pixel tile gather the SAME input pixels (and write interleaved pieces of the
  // same output rows), so the phase is the FASTEST index of a 1-D grid: a tile's phases run side by side on one XCD, the input tile
  // comes from HBM / Infinity Cache once instead of once per phase and the others hit L2.  (grid.z = phase put the phases a whole
  // sweep of the image apart.)
  unsigned lt, zph;
  if (p.nphase_flat > 1) {
    const unsigned w = xcd_remap(blockIdx.x, ntile * (unsigned)p.nphase_flat);
    lt = w / (unsigned)p.nphase_flat; zph = w - lt * (unsigned)p.nphase_flat;
  } else {
    lt = xcd_remap(blockIdx.x, ntile); zph = blockIdx.z;
  }
  const int tile_n = lt % p.tiles_n, tile_m = lt / p.tiles_n;      // cout tile fastest (pixel-fastest: 2.6x less fabric traffic on SFT, 5 % slower)
  const int cout0 = tile_n * BN;

  int py = 0, px = 0, OHp = p.OH, OWp = p.OW, in_step = p.stride, tap_step = p.dil, base_y = -p.pad, base_x = -p.pad, o_step = 1;
  if (p.transposed) {
    py = zph / p.stride; px = zph % p.stride;
    OHp = (p.OH - py + p.stride - 1) / p.stride;
    OWp = (p.OW - px + p.stride - 1) / p.stride;
    in_step = 1; tap_step = -1; o_step = p.stride;
    base_y = (py + p.pad) / p.stride; base_x = (px + p.pad) / p.stride;
  }
  const long M = p.hw_pad ? (long)p.N * p.hw_pad : (long)p.N * OHp * OWp;      // hw_pad: per-sample padded pixel index (fused per-sample sums)
  const long m0 = (long)tile_m * BM;
  // 2-D pixel tiles (16 wide, BM/16 high) for the spatial layers whose output divides evenly: a BM-pixel run of one output row needs
  // KH input rows x (BM*stride + K) columns, the 2-D tile (BM/16*stride + K) x (16*stride + K) -- 2448 instead of 4128 input pixels
  // per 128 outputs for the 8x8 stride-4 layers, 180 instead of 390 for a 3x3
  constexpr int TW2 = 16, TH2 = BM / 16;
  int t2_n = 0, t2_y0 = 0, t2_x0 = 0;
  if (p.tile2d) {
    const int tx_n = p.OW / TW2, ty_n = p.OH / TH2;
    t2_n = tile_m / (tx_n * ty_n);
    const int rem = tile_m - t2_n * (tx_n * ty_n);
    t2_y0 = (rem / tx_n) * TH2; t2_x0 = (rem % tx_n) * TW2;
    if (t2_n >= p.N) return;
  } else if (m0 >= M) return;
  const half_t* wt = p.wt + (size_t)zph * p.rows_p * p.Kp;

  // the bias is fetched now and parked in LDS after the first DMA issue: its HBM round trip used to sit in front of the prologue barrier
  float bias_reg = 0.f;
  if (tid < BN && p.bias && cout0 + tid < p.cout)      // (a per-sample bias: the tile lies within one sample, checked by the launcher)
    bias_reg = p.bias[(p.bias_sn ? (p.tile2d ? (long)t2_n : m0 / ((long)OHp * OWp)) * p.bias_sn : 0) + cout0 + tid];
  TS(7);
  if (tid < BM) {
    long m = m0 + tid;
    int n = -1, oy = 0, ox = 0;
    if (p.tile2d) {
      n = t2_n; oy = t2_y0 + tid / TW2; ox = t2_x0 + tid % TW2;
    } else if (m < M) {
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
    const int iy0 = oy * in_step + base_y, ix0 = ox * in_step + base_x;
    sOff0[tid] = n * p.in[0].sn + iy0 * p.in[0].sy + ix0 * p.in[0].sx;
    sOff1[tid] = n * p.in[1].sn + iy0 * p.in[1].sy + ix0 * p.in[1].sx;
    unsigned long long mk = 0;
    if (n >= 0) {
      // tap (ty, tx) is inside the image iff its row is and its column is: KHt + KWt tests, then one shifted OR per live tap row
      unsigned xm = 0;
      for (int tx = 0, ix = ix0; tx < p.KWt; ++tx, ix += tap_step)
        if ((unsigned)ix < (unsigned)p.W) xm |= 1u << tx;
      for (int ty = 0, iy = iy0, sh = 0; ty < p.KHt; ++ty, iy += tap_step, sh += p.KWt)
        if ((unsigned)iy < (unsigned)p.H) mk |= (unsigned long long)xm << sh;
    }
    sMask[tid] = mk;
    sOOff[tid] = n * p.o_sn + (long)(py + oy * o_step) * p.o_sy + (long)(px + ox * o_step) * p.o_sx;
  }
  if (tid < 2 * BN) sStat[tid] = 0.f;
  if (GK && tid < 64) {
    const int ty = tid / p.KWt, tx = tid - ty * p.KWt;
    sDisp[tid] = tid < p.KHt * p.KWt ? (long)tap_step * (ty * p.in[0].sy + tx * p.in[0].sx) : 0;
  }
  __syncthreads();
  TS(1);

  // ---- per-lane DMA roles.  Instruction j = wid + NW*i of a stage covers tile rows 8j .. 8j+7, lane -> row 8j + lane/8,
  // LDS chunk position c' = lane%8, channel chunk c = c' ^ ((row>>1)&7) = c' ^ ((4*(wid&1) + lane/16) & 7) for every i.
  const int cchunk = (lane & 7) ^ ((4 * (wid & 1) + (lane >> 4)) & 7);
  const int cch = cchunk * 8;          // halves (weight rows; pixel rows of a plain input)
  // FS: chunks 0..3 of a pixel row are 32 channels of the hi plane, chunks 4..7 the same 32 channels of the lo plane (c0 / 2 elements on)
  const int cchx = GK ? (FS ? (cchunk >> 2) * (p.c0 >> 1) : 0) : (FS ? (cchunk & 3) * 8 + (cchunk >> 2) * (p.c0 >> 1) : cch);
  // GK: this lane's 8-channel unit of the slice being issued, as (tap, unit within the tap); U units per tap
  const int gk_U = (FS ? (p.c0 >> 1) : p.c0) >> 3;
  int gk_tap = 0, gk_ch8 = FS ? (cchunk & 3) : cchunk;
  if (GK) { while (gk_ch8 >= gk_U) { gk_ch8 -= gk_U; ++gk_tap; } }
  // Per row: the 64-bit element offset of tap (0,0) in each input segment and a bit per tap saying whether that tap lands inside
  // the image (from the tables above).  Per slice only a wave-uniform base pointer changes.
  long off0[NXI], off1[NXI];
  unsigned long long tapmask[NXI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int r = 8 * (wid + NW * i) + (lane >> 3);
    off0[i] = sOff0[r] + cchx;
    off1[i] = sOff1[r] + cchx;
    tapmask[i] = sMask[r];
  }
  const half_t* wrow[NI - NXI];
#pragma unroll
  for (int i = 0; i < NI - NXI; ++i) {
    const int r = 8 * (wid + NW * (NXI + i) - XI) + (lane >> 3);      // weight-tile row
    const int wr = cout0 + r < p.rows_p ? cout0 + r : p.rows_p - 1;   // rows_p is padded to 128: only the 256-cout tile can reach past it
    wrow[i] = wt + (size_t)wr * p.Kp + cch;
  }
  const half_t* zp = zero_page + (lane & 7) * 8;
  // Scalar running state of the slice being ISSUED.  K is walked CHANNEL-SLICE OUTER, TAP INNER (the packed weights are indexed
  // tap * ctot + channel, so only the order of the 64-wide slices changes): a tile's taps re-touch the same 64-channel planes of the
  // same pixels back to back, so every tap after the first hits L2.  With the taps outer a workgroup streamed its whole pixels x ctot
  // window (426 KB at 832 channels, x 32 workgroups per XCD >> the 4 MB L2) between two touches of a line and every tap went back
  // to the fabric: PMC FETCH_SIZE 39 GB per launch for 1.3 GB of input on the SFT 825 -> 825 conv at N = 4.
  // The wave-uniform source pointer moves by precomputed pixel / row strides between taps (both input segments tracked, their
  // strides may differ); the per-lane row offsets of the current segment sit in offc[].
  const int ntaps = p.KHt * p.KWt;
  int cs = 0, kx = 0, tap = 0, tcount = 0;
  int g_ax = 0, g_ay = 0, g_rx = 0, g_ry = 0;                      // grouped tap order (see issue())
  const int ggx = p.tap_group ? p.KWt / p.stride : 1, ggy = p.tap_group ? p.KHt / p.stride : 1;
  const long rs0 = (long)tap_step * p.in[0].sy, rs1 = (long)tap_step * p.in[1].sy;
  const long tsx0 = (long)tap_step * p.in[0].sx, tsy0 = (long)tap_step * p.in[0].sy - (long)p.KWt * tsx0;
  const long tsx1 = (long)tap_step * p.in[1].sx, tsy1 = (long)tap_step * p.in[1].sy - (long)p.KWt * tsx1;
  const half_t* const xb0_0 = reinterpret_cast<const half_t*>(p.in[0].ptr);          // tap 0, channel 0 of segment 0
  const half_t* const xb1_0 = reinterpret_cast<const half_t*>(p.in[1].ptr) - p.c0;   // tap 0, indexed by the concatenated channel
  const half_t* xb0 = xb0_0;
  const half_t* xb1 = xb1_0;
  bool seg0 = true;
  long offc[NXI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) offc[i] = off0[i];

  // PART / NPARTS: the stage's DMA pieces in NPARTS instalments (GLDS_SPREAD: between the MFMA groups of the stage being multiplied
  // instead of as one burst behind the barrier); the scalar K-walk state advances with the last instalment
  // issue_sel: X pieces [XLO, XHI), W pieces [WLO, WHI) of stage kt; ADV: the scalar K-walk state moves on to the next stage afterwards
  auto issue_sel = [&](int kt, auto XLO, auto XHI, auto WLO, auto WHI, auto ADV) __attribute__((always_inline)) {
    constexpr int xlo = decltype(XLO)::value, xhi = decltype(XHI)::value, wlo = decltype(WLO)::value, whi = decltype(WHI)::value;
    char* sbase = smem + (kt % NSTAGE) * STAGE_BYTES;
    const half_t* xb = (seg0 ? xb0 : xb1) + (FS ? (cs >> 1) : cs);                           // wave-uniform (FS: K slice s = channels 32 s .. of both planes)
    const unsigned long long bit = 1ull << tap;                                             // ntaps <= 64 (eligibility)
    long gk_off = 0;
    bool gk_ok = true;
    if constexpr (GK) {
      gk_ok = gk_tap < ntaps;
      gk_off = sDisp[gk_ok ? gk_tap : 0] + gk_ch8 * 8;
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      if (i < xlo || i >= xhi) continue;
      const half_t* src = GK ? ((gk_ok && ((tapmask[i] >> (gk_tap & 63)) & 1)) ? xb0_0 + offc[i] + gk_off : zp)
                             : ((tapmask[i] & bit) ? xb + offc[i] : zp);
#ifdef CSBSR_GLDS_ABLATE      // timing experiments only (results are garbage): bit 0 = every DMA source folded into one L2-resident 512 KB window
      if ((CSBSR_GLDS_ABLATE & 1) && (tapmask[i] & bit)) src = xb0_0 + ((size_t)(src - xb0_0) & 0x3ffff);
#endif
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sbase + (wid + NW * i) * 1024), 16, 0, 0);
    }
    const int wk = GK ? kt * BKG : tap * p.ctot + cs;                                        // column of this slice in the packed weights
#pragma unroll
    for (int i = 0; i < NI - NXI; ++i) {
      if (i < wlo || i >= whi) continue;
      const half_t* wsrc = wrow[i] + wk;
#ifdef CSBSR_GLDS_ABLATE
      if (CSBSR_GLDS_ABLATE & 1) wsrc = wt + ((size_t)(wsrc - wt) & 0x3ffff);
#endif
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc,
                                       (__attribute__((address_space(3))) void*)(sbase + BM * 128 + (wid + NW * (NXI + i) - XI) * 1024), 16, 0, 0);
    }
    if (!decltype(ADV)::value) return;
    if constexpr (GK) {                       // this lane's unit moves on by one slice (8 units; FS: 4), wrapping into the next tap(s)
      gk_ch8 += FS ? 4 : 8;
      if (gk_ch8 >= gk_U) { gk_ch8 -= gk_U; ++gk_tap; }
      if (gk_ch8 >= gk_U) { gk_ch8 -= gk_U; ++gk_tap; }
      return;
    }
    if (++tcount == ntaps) {                  // next channel slice (wave-uniform branch)
      tcount = 0; tap = 0; kx = 0; xb0 = xb0_0; xb1 = xb1_0;
      g_ax = g_ay = g_rx = g_ry = 0;
      cs += BKG;
      if (seg0 && cs >= p.c0 && cs < p.ctot) {          // crossed into the second input segment (c0 is a multiple of BKG here)
        seg0 = false;
#pragma unroll
        for (int i = 0; i < NXI; ++i) offc[i] = off1[i];
      }
    } else if (!p.tap_group) {
      ++tap;
      xb0 += tsx0; xb1 += tsx1;
      if (++kx == p.KWt) { kx = 0; xb0 += tsy0; xb1 += tsy1; }
    } else {
      // strided layers: taps that agree modulo the stride gather the SAME input pixels (for neighbouring outputs), so they are
      // issued back to back -- (ky, kx) = (ry + s*ay, rx + s*ax), (ax, ay) fastest -- and the class's lines are still in L2 / L1
      // when its next tap wants them.  In raster order the re-touches were 4 and 32 taps apart, a 256-pixel tile's 64-channel
      // window (592 KB) x 32 workgroups per XCD in between: PMC fetch 10 GB per launch for 3.5 GB of input on the 8x8 stride-4
      // layers at N = 4, at which point the layer ran at HBM speed.
      if (++g_ax == ggx) { g_ax = 0; if (++g_ay == ggy) { g_ay = 0; if (++g_rx == p.stride) { g_rx = 0; ++g_ry; } } }
      const int ky = g_ry + p.stride * g_ay, kxx = g_rx + p.stride * g_ax;
      tap = ky * p.KWt + kxx;
      xb0 = xb0_0 + ky * rs0 + kxx * tsx0; xb1 = xb1_0 + ky * rs1 + kxx * tsx1;
    }
  };

  auto issue_part = [&](int kt, auto PART, auto NPARTS) __attribute__((always_inline)) {
    constexpr int part = decltype(PART)::value, nparts = decltype(NPARTS)::value;
    issue_sel(kt, std::integral_constant<int, NXI * part / nparts>{}, std::integral_constant<int, NXI * (part + 1) / nparts>{},
              std::integral_constant<int, (NI - NXI) * part / nparts>{}, std::integral_constant<int, (NI - NXI) * (part + 1) / nparts>{},
              std::integral_constant<bool, part == nparts - 1>{});
  };
  auto issue = [&](int kt) { issue_part(kt, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); };
#ifndef GLDS_SPREAD
#define GLDS_SPREAD 2      // measured (scripts/glds_spread_ab.py, same process): 2 instalments behind the stage's first two MFMA groups +4..7 %, 4 instalments +-0
#endif
  constexpr int SPREAD_ = GLDS_SPREAD > 0 ? GLDS_SPREAD : 1;
  constexpr int SPREAD = (GLDS_SPREAD > 0 && NXI % SPREAD_ == 0 && (NI - NXI) % SPREAD_ == 0) ? GLDS_SPREAD : 0;

  f16v acc[TA][2];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nkt = p.Kp / BKG;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nkt) issue(s);

  if (tid < BN) sBias[tid] = bias_reg;        // read in the epilogue, behind the K loop's barriers
  const float slope = (p.act == CSBSR_ACT_PRELU) ? *p.prelu : p.act_slope;
  // fragment addressing: tile row R, channel chunk c -> byte R*128 + ((c ^ ((R>>1)&7)) << 4)
  const int xr0 = wm * 64 + (lane & 31), wr0 = wn * (32 * TA) + (lane & 31);
  TS(2);
  // ---- the MFMAs of stage kt; hook(PART), PART = 0..3, runs behind the stage's first four MFMA groups (DMA pieces of the refill)
  auto compute = [&](int kt, auto&& hook) __attribute__((always_inline)) {
    const char* xs = smem + (kt % NSTAGE) * STAGE_BYTES;
    const char* ws = xs + BM * 128;
    auto load_w = [&](int ks, h8 (&af)[TA]) __attribute__((always_inline)) {
      const int c = ks * 2 + (lane >> 5);
#pragma unroll
      for (int a = 0; a < TA; ++a) {
        const int R = wr0 + a * 32;
        af[a] = *reinterpret_cast<const h8*>(ws + R * 128 + ((c ^ ((R >> 1) & 7)) << 4));
      }
    };
    auto load_x = [&](int ks, h8 (&bf)[2]) __attribute__((always_inline)) {
      const int c = ks * 2 + (lane >> 5);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int R = xr0 + b * 32;
        bf[b] = *reinterpret_cast<const h8*>(xs + R * 128 + ((c ^ ((R >> 1) & 7)) << 4));
      }
    };
    // fragment double-buffering: the reads of sub-step ks+1 are in flight while the 4 MFMAs of sub-step ks issue
    auto load_frags = [&](int ks, h8 (&af)[TA], h8 (&bf)[2]) __attribute__((always_inline)) {
#ifdef CSBSR_GLDS_ABLATE      // bit 1 = no LDS fragment reads after a tile's first K step (stale fragments)
      if ((CSBSR_GLDS_ABLATE & 2) && (kt | ks)) return;
#endif
      load_w(ks, af); load_x(ks, bf);
    };
    auto mm = [&](const h8 (&af)[TA], const h8 (&bf)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    };
    if constexpr (FS) {
      // chunk pairs ks = 0, 1: channels 0..15 / 16..31 of the hi halves, ks = 2, 3: of the lo halves (pixel AND weight tile)
      // (p.fs == 2, wave-uniform: the x_hi w_lo product is skipped -- the layer's plan keeps the weights' fp16 rounding, round 6 -- : four
      // MFMA groups per stage instead of six from the same staged slice)
      const bool wlo = p.fs != 2;
      h8 af[TA], bh[2], bl[2];
      load_w(0, af); load_x(0, bh);
      mm(af, bh);                      // x_hi w_hi
      hook(std::integral_constant<int, 0>{});
      load_x(2, bl);
      mm(af, bl);                      // x_lo w_hi
      hook(std::integral_constant<int, 1>{});
      if (wlo) {
        load_w(2, af);
        mm(af, bh);                    // x_hi w_lo
      }
      hook(std::integral_constant<int, 2>{});
      load_w(1, af); load_x(1, bh);
      mm(af, bh);
      hook(std::integral_constant<int, 3>{});
      load_x(3, bl);
      mm(af, bl);
      if (wlo) {
        load_w(3, af);
        mm(af, bh);
      }
    } else if constexpr (CT == 2) {      // 8 MFMAs per sub-step cover the next fragment reads; one fragment set keeps the wave inside 256 registers
      auto kslice = [&](auto KS) __attribute__((always_inline)) {
        h8 af[TA], bf[2];
        load_frags(decltype(KS)::value, af, bf);
        mm(af, bf);
        hook(KS);
      };
      kslice(std::integral_constant<int, 0>{}); kslice(std::integral_constant<int, 1>{});
      kslice(std::integral_constant<int, 2>{}); kslice(std::integral_constant<int, 3>{});
    } else {
      h8 af0[TA], bf0[2], af1[TA], bf1[2];
      load_frags(0, af0, bf0);
      load_frags(1, af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      mm(af0, bf0);
      hook(std::integral_constant<int, 0>{});
      load_frags(2, af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      mm(af1, bf1);
      hook(std::integral_constant<int, 1>{});
      load_frags(3, af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      mm(af0, bf0);
      hook(std::integral_constant<int, 2>{});
      __builtin_amdgcn_sched_barrier(0);
      mm(af1, bf1);
      hook(std::integral_constant<int, 3>{});
    }
  };

  for (int kt = 0; kt < nkt; ++kt) {
    // stage kt landed (this wave's DMAs), then everyone's
    const int ahead = (nkt - 1 - kt) < (NSTAGE - 2) ? (nkt - 1 - kt) : (NSTAGE - 2);   // stages still allowed in flight
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt == 0) TS(3);
    // (measured and dropped, r03: the second wave of each SIMD issuing its pieces half a stage later -- so that one wave's ~100 address /
    // issue instructions run under the other's MFMAs -- changed nothing: 920 vs 926 TF/s on the ResNet 512 -> 512 layer)
    const bool refill = kt + NSTAGE - 1 < nkt;
    if (SPREAD == 0 && refill) issue(kt + NSTAGE - 1);      // refills the buffer read in iteration kt-1
    compute(kt, [&](auto PART) __attribute__((always_inline)) {      // instalment PART of SPREAD, after an MFMA group
      if constexpr (SPREAD > 0 && decltype(PART)::value < SPREAD) { if (refill) issue_part(kt + NSTAGE - 1, PART, std::integral_constant<int, SPREAD>{}); }
    });
  }
  TS(4);
  __syncthreads();

  // ---- epilogue staged through LDS in ONE pass: the whole BM x 128 fp32 tile goes into the (now dead) DMA ring -- unpadded rows
  // of 32 16-byte slots with slot' = slot ^ (row & 15) instead of padding, so it fits exactly (64 KB for BM = 128) -- all waves
  // stage at once, one LDS-only barrier, then every thread owns one 8-channel chunk of BM/RSTEP rows.  (Two 64-cout passes with
  // three full barriers, each also draining the previous pass's stores, made the epilogue longer than the 8-slice K loop of the
  // transposed conv.)
  float* sO = reinterpret_cast<float*>(smem);
  constexpr int CPR = SOW / 8;                // 8-channel chunks per staged row (128 couts; 64 for the narrow tile)
  const int cc8 = tid % CPR;
  constexpr int RSTEP = NT / CPR;
  const EpiFast fe = conv_epilogue_fast_setup(p, slope);      // conv_common.h
  // CT == 2: the staged tile is BM x 128 couts at a time.  Pass h stages the two 32-cout MFMA tiles {2h, 2h+1} of EVERY wave (so no
  // wave carries more than half its accumulators across a pass of row code): staged columns 0..63 are couts 64h.. of the waves
  // with wn = 0, columns 64..127 couts 128 + 64h.. of the waves with wn = 1.  Each pass is a compile-time instance (static
  // accumulator indices).
  auto epi_pass = [&](auto HH) {
  constexpr int hh = decltype(HH)::value;
  if (hh > 0) lds_barrier();                 // the previous half's rows are read out
#pragma unroll
  for (int a = 0; a < (TA < 2 ? TA : 2); ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int pix = wm * 64 + b * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int slot = (wn * (SOW / 2) + a * 32 + 8 * q + 4 * (lane >> 5)) >> 2;
        const f16v& t = acc[2 * hh + a][b];
        f4 v = {t[4 * q + 0], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
        *reinterpret_cast<f4*>(sO + pix * SOW + ((slot ^ (pix & 15)) << 2)) = v;
      }
    }
  lds_barrier();
  if (hh == 0) TS(5);
  const int lcol = CT != 2 ? cc8 * 8 : (cc8 >> 3) * 128 + hh * 64 + (cc8 & 7) * 8;      // this thread's 8 couts within the workgroup's BN
  const int co = cout0 + lcol;
  float bias[8], ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bias[e] = sBias[lcol + e]; ssum[e] = ssq[e] = 0.f; }
  if (fe.ok) {
    constexpr int RPT = BM / RSTEP, EG = CT == 2 ? 2 : (RPT < 8 ? RPT : 8);
    static_assert(RPT % EG == 0, "rows per thread must split into groups");
    auto rows = [&](auto EXTRA, auto BNSTAT) {       // EXTRA: a residual and / or the old output is combined in
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
          const f4 v0 = *reinterpret_cast<const f4*>(sO + row * SOW + (((2 * cc8) ^ (row & 15)) << 2));
          const f4 v1 = *reinterpret_cast<const f4*>(sO + row * SOW + (((2 * cc8 + 1) ^ (row & 15)) << 2));
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
    if (fe.bn) conv_epilogue_flush_stats<CPR, NW>(p, sStat, BN, lcol, co, ssum, ssq);
  } else {
    // rows in groups of EG: the group's residual / old-output loads are all issued before the first row is combined
    constexpr int RPT = BM / RSTEP, EG = 2;
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
        const f4 v0 = *reinterpret_cast<const f4*>(sO + row * SOW + (((2 * cc8) ^ (row & 15)) << 2));
        const f4 v1 = *reinterpret_cast<const f4*>(sO + row * SOW + (((2 * cc8 + 1) ^ (row & 15)) << 2));
        v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3]; v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
        conv_epilogue_row(p, v, bias, slope, co, rn[i], roy[i], rox[i], ssum, ssq, &pre[i]);
      }
    }
    conv_epilogue_flush_stats<CPR, NW>(p, sStat, BN, lcol, co, ssum, ssq);
  }
  };      // epi_pass
  epi_pass(std::integral_constant<int, 0>{});
  if constexpr (CT == 2) epi_pass(std::integral_constant<int, 1>{});
  TS(6);
  conv_epilogue_store_stats(p, sStat, BN, cout0, (size_t)zph * p.tiles_m + tile_m);      // (the last flush ended with a barrier)
}

static half_t* g_zero_page = nullptr;
static int g_glds_phase_flat = 1;
static int g_glds_tile2d = 1;
static int g_glds_tap_group = 1;
static int g_glds_wide = 1;
static int g_glds_narrow = 1;
static int g_glds_gk = 1;

template <int BM, int NWM, int NSTAGE, int CT = 1, bool FS = false, bool GK = false>
static int launch_glds(const ConvK& k, int nphase, long maxM, hipStream_t st) {
  ConvK p = k;
  constexpr int BN = CT ? 128 * CT : 64, SOW = BN > 128 ? 128 : BN;
  p.tiles_m = (unsigned)((maxM + BM - 1) / BM);
  p.tile2d = (g_glds_tile2d && !k.transposed && k.KHt * k.KWt > 1 && k.OW % 16 == 0 && k.OH % (BM / 16) == 0) ? 1 : 0;      // same tile count
  p.tap_group = (!GK && g_glds_tap_group && !k.transposed && k.stride > 1 && k.dil == 1 && k.KHt > k.stride && k.KHt % k.stride == 0 &&
                 k.KWt > k.stride && k.KWt % k.stride == 0) ? 1 : 0;
  p.tiles_n = (unsigned)((k.coutp + BN - 1) / BN);
  constexpr int RING = NSTAGE * (BM + BN) * 128;
  constexpr int EPI = BM * SOW * 4;
  constexpr int SM_BYTES = (RING > EPI ? RING : EPI) + BM * (4 * 8 + 3 * 4) + 3 * BN * 4 + (GK ? 64 * 8 : 0);      // ring / staged tile + row tables + statistics + bias (+ GK tap table)
  static_assert(SM_BYTES <= 160 * 1024, "LDS budget");
  static LdsAttrOnce attr;
  if (int e = csbsr_lds_attr(attr, reinterpret_cast<const void*>(conv_igemm_glds_kernel<BM, NWM, NSTAGE, CT, FS, GK>), SM_BYTES, "conv(glds)")) return e;
  if (!g_zero_page) {
    if (hipMalloc(reinterpret_cast<void**>(&g_zero_page), 256) != hipSuccess) { csbsr_set_error("conv(glds): zero page alloc failed"); return 2; }
    (void)hipMemset(g_zero_page, 0, 256);
  }
  p.nphase_flat = (g_glds_phase_flat && nphase > 1) ? nphase : 0;
  ConvStatPlan sp;
  if (int e = conv_stat_prepare(p, BM, nphase, sp, st)) return e;
  dim3 grid(p.tiles_m * p.tiles_n * (p.nphase_flat ? nphase : 1), 1, p.nphase_flat ? 1 : nphase);
  hipLaunchKernelGGL((conv_igemm_glds_kernel<BM, NWM, NSTAGE, CT, FS, GK>), grid, dim3(NWM * 128), SM_BYTES, st, p, g_zero_page);
  if (int e = conv_stat_finish(p, sp, st)) return e;
  CSBSR_LAUNCH_CHECK("csbsr_conv_forward(glds)");
  return 0;
}

static int g_glds_mode = 2;      // 0: off, 1: 128x128 x2 stages only, 2: + 256x128 x3 stages for long-K stride-1 layers
extern "C" void csbsr_debug_set_conv_glds(int mode) {
  g_glds_mode = mode & 7;
  g_glds_phase_flat = (mode & 32) ? 0 : 1;
  g_glds_tap_group = (mode & 128) ? 0 : 1;       // bit 7: raster tap order on the strided layers (A/B timing)
  g_glds_tile2d = (mode & 64) ? 0 : 1;           // bit 6: linear pixel tiles everywhere (A/B timing)     // bit 5: phases back on grid.z (A/B timing)
  g_glds_wide = (mode & 256) ? 0 : 1;            // bit 8: no 256-cout tile (A/B timing)
  g_glds_narrow = (mode & 1024) ? 0 : 1;         // bit 10: no 64-cout tile, 33..64-cout layers back on the register-staged kernel (A/B timing)
  g_glds_gk = (mode & 2048) ? 0 : 1;             // bit 11: no general K walk, channel counts off the 64-channel grid back on the register-staged kernel (A/B timing)
  conv_thin_enable((mode & 16) ? 0 : 1);      // bit 4: route the 3-channel heads through the generic kernel (A/B timing)
  conv_thin_sc_enable((mode & 4096) ? 0 : 1);    // bit 12: the 128 -> 3 strided layers back on the general kernel (A/B timing)
  conv_thin_tpd_enable((mode & 8192) ? 0 : 1);   // bit 13: the stems' 64 -> 3 stride-2 dgrads back on the general kernel (A/B timing)
  conv_thin_cin2_enable((mode & 512) ? 0 : 1);  // bit 9: no streaming variant of the 3-channel-input kernel (A/B timing, tests)
}

static bool conv_glds_general_k(const ConvK& k) { return k.fs ? (k.ctot / 2) % 32 != 0 : k.ctot % 64 != 0; }

// eligibility: MFMA-bound shapes only
bool conv_glds_eligible(const ConvK& k) {
  if (g_glds_mode == 0) return false;
  if (k.coutp <= 32 || (k.coutp <= 64 && !g_glds_narrow)) return false;      // 33..64 couts: the 64-cout tile
  if (conv_glds_general_k(k)) {          // channel counts that are not whole 64-channel slices: the general K walk (one segment, >= 32 channels)
    if (!g_glds_gk || k.c0 != k.ctot || (k.fs ? k.ctot / 2 : k.ctot) < 32) return false;
  } else if (k.c0 != k.ctot && k.c0 % 64 != 0) return false;
  if (k.rows_p % (k.coutp <= 64 ? 64 : 128) != 0) return false;      // packed weights padded to the tile's rows (csbsr_pack_weights does)
  if (k.KHt * k.KWt > 64) return false;       // one validity bit per tap in a 64-bit mask
  return true;
}

int conv_glds_launch(const ConvK& k, int nphase, long maxM, hipStream_t st) {
  // measured (scripts/bench_conv.py): the 8-wave 256x128 tile only pays for long-K layers (SFT 3x3, ResNet 3x3, the 8x8 stride-4
  // gathers); transposed / short-K layers run faster with two 128x128 workgroups per CU
  // (round 5) ... and for short-K layers into >= 384 output channels over >= 4 M pixels (config 5's 64 -> 505 blur_skip layers at HR 1792^2,
  // B = 4: the 256 px x 256 cout tile halves the tile count of an epilogue-bound launch -- dgrad 12.3 -> 10.7 ms, the step 569 -> 551 ms;
  // the same tile on config 2's / config 4's short-K layers, 0.8 M pixels each, measured +-0 / -0.6 %: they keep two 128 x 128 workgroups per CU)
  const bool big = (g_glds_mode == 2 && !k.transposed && k.Kp >= 2304 && maxM >= 256 * 256) ||
                   (g_glds_mode == 2 && !k.transposed && k.Kp >= 512 && k.coutp >= 384 && maxM >= (1l << 22)) ||
                   (g_glds_mode == 3 && maxM >= 256 * 256);     // mode 3 (A/B timing): the 256-row tile wherever it fits
  // 256 px x 256 couts (128 flop per staged byte instead of 85) where the couts fill 256-wide tiles about as well as 128-wide ones
  const int pad128 = (k.coutp + 127) / 128 * 128, pad256 = (k.coutp + 255) / 256 * 256;
  const bool wide = big && g_glds_wide && k.coutp >= 256 && pad256 * 8 <= pad128 * 9 + 64;
  const bool gk = conv_glds_general_k(k);
  const int tile = k.coutp <= 64 ? 0 : (!big ? 1 : (wide ? 3 : 2));
  g_last_conv_kernel = (tile == 0 ? CONVK_GLDS64 : (tile == 1 ? CONVK_GLDS128 : (tile == 3 ? CONVK_GLDS256W : CONVK_GLDS256))) |
                       ((k.fs ? 1 : 0) | (gk ? 2 : 0)) << 8;      // bits 8..: the template instance (FS, GK), one rocprof row each
#define GLDS_DISPATCH(FS_, GK_)                                                                      \
  switch (tile) {                                                                                    \
    case 0: return launch_glds<128, 2, 2, 0, FS_, GK_>(k, nphase, maxM, st);                        \
    case 1: return launch_glds<128, 2, 2, 1, FS_, GK_>(k, nphase, maxM, st);                        \
    case 3: return launch_glds<256, 4, 2, 2, FS_, GK_>(k, nphase, maxM, st);                        \
    default: return launch_glds<256, 4, 3, 1, FS_, GK_>(k, nphase, maxM, st);                       \
  }
  if (k.fs) { if (gk) { GLDS_DISPATCH(true, true) } else { GLDS_DISPATCH(true, false) } }
  if (gk) { GLDS_DISPATCH(false, true) }
  GLDS_DISPATCH(false, false)
#undef GLDS_DISPATCH
}
