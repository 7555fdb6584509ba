// Weight-gradient GEMM, LDS-DMA variant:   G[a][tap][b] = sum over pixels  A[pix][a] * B[pix @ tap][b]
//
// Same math, grid and slab contract as conv_wgrad.hip; different data movement.  The register-staged kernel spends 55 % of a
// workgroup's life issuing the next step's loads (address arithmetic, predicated 16-byte loads into staging VGPRs, a ds_write
// pass): here both operand tiles go HBM -> LDS with global_load_lds_dwordx4 / buffer_load_dwordx4 ... lds into an NSTAGE-deep ring (counted s_waitcnt vmcnt(N) +
// raw s_barrier, DMAs stay in flight across barriers), exactly as conv_igemm_glds.hip stages the forward operands.
//
//  * a stage = 64 reduction pixels of the A tile ([pixel][BA channels]) followed by 64 pixels of the B tile ([pixel][BN columns]);
//    one wave instruction moves 1 KiB = 4 pixel rows of a 128-channel tile / 2 rows of a 256-channel tile;
//  * the MFMA fragments want 8 consecutive PIXELS per lane, so they are read with ds_read_b64_tr_b16 (hardware transpose), whose
//    16-lane groups fetch a 4-pixel x 16-channel block (four 32-byte row pieces).  The LDS image is lane-linear (DMA), rows of
//    256 / 512 bytes = a whole number of bank sweeps, so the four rows of a block would sit on the same banks: the 32-byte unit a
//    lane FETCHES is permuted instead -- position (row, unit') holds channel unit  unit' ^ ((row & 3) << 1)  -- and the fragment
//    reads apply the same XOR (cdna guide rule 21): the 8 row pieces a 32-lane group touches land on 8 distinct 32-byte bank groups;
//  * out-of-image taps, pixels past the split's end and channel / column overhang read as zeros (a 256-byte zero page in the linear-stage
//    form, the buffer descriptor's bounds check in the 2-D-stage form -- see the kernel template);
//  * the DMA pieces are inline assembly: behind its own LDS-DMA builtins hipcc waits vmcnt(0) before the next transposing LDS read, which
//    serialised the ring (see wg_dma16);
//  * tile order, tap permutation, row shift and the flat (split, tile) grid are those of conv_wgrad.hip.
//
// Autograd wgrad of F.conv2d / F.conv_transpose2d at the call sites listed in conv_igemm.hip.
#include <cstdlib>
#include "common.h"
#include "csbsr_debug.h"
#include "conv_wgrad.h"
#include <type_traits>

int g_wgrad_glds = 459 + 512;    // bit 0 kernel enabled, bit 1 256 x 256 tile, bit 2 no 128 x 256 tile, bit 3 every eligible problem, bit 6 128 x 256 tap-pair
                           // tiles for the 8x8 stride-4 layers, bit 7 2-D stage rectangles, bit 8 128 x 512 four-tap tiles for those layers (csbsr_debug_set_wgrad_tr)

// One LDS-DMA piece (64 lanes x 16 bytes -> LDS bytes [lds_addr, lds_addr + 1024)) as inline assembly: behind the compiler's own
// global_load_lds builtin hipcc puts an s_waitcnt vmcnt(0) in front of the next transposing LDS read (it cannot tell the ring stages
// apart), i.e. the stage just requested had to LAND before the current one could be multiplied -- the ring never overlapped within a
// wave.  The pieces' completion is counted by hand below (vmcnt(N) + barrier per stage).
static __device__ __forceinline__ void wg_dma16(const half_t* src, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_addr));
}

// The same piece through a buffer descriptor: per-lane 32-bit byte offset, out-of-range offsets (-1) read as zeros.
typedef int wg_v4i __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void wg_dma16_buf(wg_v4i rs, int voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr));
}
static __device__ __forceinline__ wg_v4i wg_make_rs(const half_t* base) {      // [base, base + 2 GB), from provably uniform halves
  const unsigned long a = reinterpret_cast<unsigned long>(base);
  wg_v4i r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r[2] = 0x7fffffff;
  r[3] = 0x00020000;
  return r;
}

// T2D: the 64 pixels of a stage are a (64 >> tw_log) x (1 << tw_log) rectangle of the A grid instead of 64 consecutive pixels (the host
// picks the widest power-of-two width that divides AW; the rows must divide AH).  A lane's position inside the rectangle -- and with
// it the byte offset of every DMA piece it issues and the tap-shifted coordinates it has to bounds-check -- is then a kernel constant,
// the stage enters through the (uniform, scalar-ALU) base of a buffer descriptor, and a piece costs ~7 vector instructions instead of
// the ~50 of the running (n, y, x) bookkeeping with its wrap-around loops: per stage and wave that code was 330 instructions beside 16
// MFMAs, i.e. the kernel was bound by instruction issue, not by the matrix pipe or the loads.
template <int BA, int BN, int NWA, int NWB, int NSTAGE, bool T2D>
__global__ __launch_bounds__(64 * NWA * NWB) void conv_wgrad_glds_kernel(const WgradK p, const half_t* __restrict__ zero_page) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the inline assembly has no host form: the host pass emits only the launch stub)
  constexpr int NW = NWA * NWB;
  constexpr int AWv = BA / NWA, BWv = BN / NWB;      // rows / columns per wave
  constexpr int TA = AWv / 32, TB = BWv / 32;
  constexpr int BP = WG_BP;                          // 64 pixels per stage
  constexpr int A_BYTES = BP * BA * 2, B_BYTES = BP * BN * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int AI = A_BYTES / 1024, BI = B_BYTES / 1024;      // wave instructions per stage
  constexpr int NIA = AI / NW, NIB = BI / NW, NI = NIA + NIB;  // per wave
  constexpr int CPRA = BA / 8, CPRB = BN / 8;                  // 16-byte chunks per tile row
  constexpr int RPIA = 64 / CPRA, RPIB = 64 / CPRB;            // tile rows per wave instruction
  constexpr int DA = NW * RPIA, DB = NW * RPIB;                // pixel distance between a lane's consecutive instructions
  static_assert(AI % NW == 0 && BI % NW == 0, "stage instructions must split evenly over the waves");
  static_assert(NIA * DA == BP && NIB * DB == BP, "a lane's instructions must cover the stage");
  static_assert(NW % 2 == 0, "swizzle term must be lane-constant");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wid / NWB, wb = wid % NWB;
  const unsigned ntile = p.tiles_a * p.tiles_b;
  unsigned lt, zsplit;
  if (BN == 512 && p.flat == 2) {
    // the four-tap tiles of the 8x8 stride-4 layers, 1-D grid of 16 tiles x (an even number of) splits.  An input pixel is read by the
    // four taps (ky, ky + 4) x (kx, kx + 4) = four DIFFERENT tiles, so an XCD (blockIdx % 8) takes all four tiles of one ky pair, one
    // after the other, for the splits of one parity: every input row then crosses the fabric into ONE L2 (with two tiles per XCD it
    // entered two: 8.2-9.9 GB per launch against 5.2 for the tap-pair tile)
    const unsigned e = blockIdx.x & 7u, i = blockIdx.x >> 3;
    const unsigned t4 = i & 3u;
    zsplit = 2u * (i >> 2) + (e >> 2);
    lt = (((e & 3u) + 4u * (t4 >> 1)) << 1) | (t4 & 1u);      // tile index = 2 ky + column half
  } else if (p.flat) {
    const unsigned w = xcd_remap(blockIdx.x, ntile * (unsigned)p.splits);
    zsplit = w / ntile; lt = w - zsplit * ntile;
  } else {
    lt = xcd_remap(blockIdx.x, ntile); zsplit = blockIdx.z;
  }
  const int a0 = (lt % p.tiles_a) * BA;
  int col0 = (lt / p.tiles_a) * BN;
  if (p.tap_perm) {      // see conv_wgrad.hip: XCD j takes the taps with ky = j%4 (+4), kx in {2(j/4), 2(j/4)+1} (+4)
    if constexpr (BN == 128) {
      const int j = lt >> 3, r = lt & 7;
      const int ky = (j & 3) + 4 * (r >> 2), kx = 2 * (j >> 2) + (r & 1) + 4 * ((r >> 1) & 1);
      col0 = (ky * 8 + kx) * 128;
    } else if constexpr (BN == 512) {      // (natural order: tile = 2 ky + column half; the XCD assignment is the grid mapping above)
    } else {
      const int j = lt >> 2, r = lt & 3;
      const int ky = (j & 3) + 4 * (r >> 1), kx0 = 2 * (j >> 2) + 4 * (r & 1);
      col0 = (ky * 8 + kx0) * 128;
    }
  }
  long shift = 0;
  if (p.row_shift) shift = (long)((((col0 / p.cbtot) / p.KW) * p.dil) / p.stride) * p.AW;
  long mbeg = (long)zsplit * p.per_split - shift;
  long mend = mbeg + p.per_split;
  if (mbeg < 0) mbeg = 0;
  if (mend > p.M || (int)zsplit == p.splits - 1) mend = p.M;
  if (mbeg >= mend) return;

  // ---- per-lane DMA roles.  A: instruction j = wid + NW*i covers tile rows RPIA*j ..; lane -> row RPIA*j + lane/CPRA, LDS chunk
  // position c' = lane % CPRA, fetched channel chunk c = ((c'>>1) ^ ((row & 3) << 1)) << 1 | (c' & 1)  (lane-constant: NW is even)
  const int rowA0 = wid * RPIA + lane / CPRA, rowB0 = wid * RPIB + lane / CPRB;
  const int cpa = lane % CPRA, cpb = lane % CPRB;
  const int chA = ((((cpa >> 1) ^ ((rowA0 & 3) << 1)) << 1) | (cpa & 1)) * 8;
  const int chB = ((((cpb >> 1) ^ ((rowB0 & 3) << 1)) << 1) | (cpb & 1)) * 8;
  const bool a_ok = a0 + chA < p.ca;
  // B column chunk -> (tap, channel, segment): constant per lane
  int b_ky, b_kx;
  const half_t* b_ptr;
  long b_sn, b_sy, b_sx;
  bool b_ok;
  {
    const int col = col0 + chB;
    b_ok = col < p.ktot;
    const int tap = b_ok ? col / p.cbtot : 0;
    const int c = b_ok ? col - tap * p.cbtot : 0;
    b_ky = (tap / p.KW) * p.dil - p.pad;
    b_kx = (tap % p.KW) * p.dil - p.pad;
    const csbsr_seg_t& sg = c < p.cb0 ? p.b[0] : p.b[1];
    b_ptr = reinterpret_cast<const half_t*>(sg.ptr) + (c < p.cb0 ? c : c - p.cb0);
    b_sn = sg.sn; b_sy = sg.sy; b_sx = sg.sx;
  }
  struct Pix { int n, y, x; long off; };
  auto init_pix = [&](long m, long sn, long sy, long sx) {
    Pix c;
    c.n = (int)(m / ((long)p.AH * p.AW));
    const int rem = (int)(m - (long)c.n * p.AH * p.AW);
    c.y = rem / p.AW; c.x = rem - c.y * p.AW;
    c.off = c.n * sn + c.y * sy + c.x * sx;
    return c;
  };
  const long bsx = (long)p.stride * b_sx, bsy = (long)p.stride * b_sy;
  const long b_tap = (long)b_ky * b_sy + (long)b_kx * b_sx;
  const long b_dx = DB * bsx, b_rowfix = bsy - (long)p.AW * bsx, b_imgfix = b_sn - (long)p.AH * bsy;
  const long a_dx = (long)DA * p.a_sx, a_rowfix = p.a_sy - (long)p.AW * p.a_sx, a_imgfix = p.a_sn - (long)p.AH * p.a_sy;
  auto advance = [&](Pix& c, int d, long dxs, long rowfix, long imgfix) {
    c.x += d; c.off += dxs;
    while (c.x >= p.AW) { c.x -= p.AW; c.off += rowfix; if (++c.y == p.AH) { c.y = 0; ++c.n; c.off += imgfix; } }
  };
  Pix cb = init_pix(mbeg + rowB0, b_sn, bsy, bsx);
  Pix ca_ = init_pix(mbeg + rowA0, p.a_sn, p.a_sy, p.a_sx);
  const half_t* a_base = p.a + a0 + chA;
  const half_t* b_base = b_ptr + b_tap;
  const half_t* zp = zero_page + (lane & 7) * 8;
  long m_issue = mbeg;                       // first pixel of the stage being ISSUED

  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem) + (unsigned)wid * 1024u;
  // ---- T2D: kernel-constant piece offsets / tap-shifted coordinates, uniform stage position
  int voffA[NIA], voffB[NIB], cy[NIB], cx[NIB];
  int t_n = 0, t_y = 0, t_x = 0;                       // the stage being ISSUED: image, tile row, tile column
  if (T2D) {
    const int twm = (1 << p.tw_log) - 1;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int r = rowA0 + DA * i, py = r >> p.tw_log, px = r & twm;
      voffA[i] = a_ok ? 2 * (int)(py * p.a_sy + px * p.a_sx + chA) : -1;
    }
    const int c_b = b_ok ? (col0 + chB) % p.cbtot : 0;
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int r = rowB0 + DB * i, py = r >> p.tw_log, px = r & twm;
      cy[i] = b_ok ? py * p.stride + b_ky : 0x40000000;
      cx[i] = px * p.stride + b_kx;
      voffB[i] = 2 * (int)((py * p.stride + b_ky + p.pad) * b_sy + (px * p.stride + b_kx + p.pad) * b_sx + c_b);
    }
    const unsigned s0 = (unsigned)(mbeg / BP), per_img = (unsigned)(p.gx * p.gy);
    t_n = __builtin_amdgcn_readfirstlane((int)(s0 / per_img));
    const unsigned r_ = s0 - (unsigned)t_n * per_img;
    t_y = __builtin_amdgcn_readfirstlane((int)(r_ / (unsigned)p.gx));
    t_x = __builtin_amdgcn_readfirstlane((int)(r_ - (unsigned)t_y * (unsigned)p.gx));
  }
  // (uniform) running position of the stage being issued: descriptor bases and the gathered side's first row / column, stepped by
  // scalar adds -- next rectangle in the row, first one of the next rectangle row, of the next image
  const int tw_ = 1 << p.tw_log, th_ = 64 >> p.tw_log;
  const half_t* base_a = p.a + a0 + t_n * p.a_sn + (long)(t_y * th_) * p.a_sy + (long)(t_x * tw_) * p.a_sx;
  const half_t* base_b = reinterpret_cast<const half_t*>(p.b[0].ptr) + t_n * p.b[0].sn + (long)(t_y * th_ * p.stride - p.pad) * p.b[0].sy +
                         (long)(t_x * tw_ * p.stride - p.pad) * p.b[0].sx;
  int Ys = t_y * th_ * p.stride, Xs = t_x * tw_ * p.stride;
  const long da_x = (long)tw_ * p.a_sx, da_y = (long)th_ * p.a_sy - (long)p.gx * da_x, da_n = p.a_sn - (long)p.gy * th_ * p.a_sy;
  const long db_x = (long)tw_ * p.stride * p.b[0].sx, db_y = (long)th_ * p.stride * p.b[0].sy - (long)p.gx * db_x,
             db_n = p.b[0].sn - (long)p.gy * th_ * p.stride * p.b[0].sy;
  const int dXs = tw_ * p.stride, dYs = th_ * p.stride;
  // PART of NPARTS: the stage's pieces in instalments (behind the first MFMA groups of the stage being multiplied instead of as one
  // burst behind the barrier: WG_SPREAD); the uniform stage position steps on with the last instalment
  auto issue2d_part = [&](int kt, auto PART, auto NPARTS) __attribute__((always_inline)) {
    constexpr int part = decltype(PART)::value, nparts = decltype(NPARTS)::value;
    const unsigned sbase = lds0 + (unsigned)((kt % NSTAGE) * STAGE_BYTES);
    const wg_v4i rsa = wg_make_rs(base_a), rsb = wg_make_rs(base_b);
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      if (i < NIA * part / nparts || i >= NIA * (part + 1) / nparts) continue;
      wg_dma16_buf(rsa, voffA[i], sbase + (unsigned)(NW * i * 1024));
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      if (i < NIB * part / nparts || i >= NIB * (part + 1) / nparts) continue;
      const bool ok = (unsigned)(cy[i] + Ys) < (unsigned)p.BH && (unsigned)(cx[i] + Xs) < (unsigned)p.BW;
      wg_dma16_buf(rsb, ok ? voffB[i] : -1, sbase + (unsigned)(A_BYTES + NW * i * 1024));
    }
    if (part != nparts - 1) return;
    base_a += da_x; base_b += db_x; Xs += dXs;
    if (++t_x == p.gx) {
      t_x = 0; Xs = 0; base_a += da_y; base_b += db_y; Ys += dYs;
      if (++t_y == p.gy) { t_y = 0; Ys = 0; base_a += da_n; base_b += db_n; }
    }
  };
  auto issue2d = [&](int kt) { issue2d_part(kt, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); };
#ifndef WG_SPREAD
#define WG_SPREAD 2
#endif
  // (8-wave tiles only: measured per launch at N = 4, same process, 256 x 256 SFT 958 -> 992 TF/s, ResNet 512 867 -> 945, up_1 1044 -> 1126,
  // 128 x 256 8x8 stride-4 750 -> 768; in the training step 1.85 -> 1.72 ms and 1.93 -> 1.84 ms per launch.  The 128 x 128 tile -- two
  // workgroups per CU, which already overlap each other's bursts -- went 0.60 -> 0.66 ms in the step: burst kept)
  constexpr int SPREAD = (T2D && NW == 8 && WG_SPREAD > 0 && NIA % (WG_SPREAD > 0 ? WG_SPREAD : 1) == 0 && NIB % (WG_SPREAD > 0 ? WG_SPREAD : 1) == 0) ? WG_SPREAD : 0;
  auto issue = [&](int kt) {
    if (T2D) { issue2d(kt); return; }
    const unsigned sbase = lds0 + (unsigned)((kt % NSTAGE) * STAGE_BYTES);
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const bool ok = a_ok && m_issue + rowA0 + DA * i < mend;
      const half_t* src = ok ? a_base + ca_.off : zp;
      wg_dma16(src, sbase + (unsigned)(NW * i * 1024));
      advance(ca_, DA, a_dx, a_rowfix, a_imgfix);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int by = cb.y * p.stride + b_ky, bx = cb.x * p.stride + b_kx;
      const bool ok = b_ok && m_issue + rowB0 + DB * i < mend && (unsigned)by < (unsigned)p.BH && (unsigned)bx < (unsigned)p.BW;
      const half_t* src = ok ? b_base + cb.off : zp;
      wg_dma16(src, sbase + (unsigned)(A_BYTES + NW * i * 1024));
      advance(cb, DB, b_dx, b_rowfix, b_imgfix);
    }
    m_issue += BP;
  };

  f16v acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nkt = (int)((mend - mbeg + BP - 1) / BP);
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nkt) issue(s);

  // ---- fragment addressing.  16-lane group g = lane/16: block of 4 pixel rows x 16 channels; lane i = lane%16 supplies the address
  // of row (i>>2), column quad (i&3) of the block; after the transpose the lane holds channel (block + i) of the 4 rows.
  // byte = row * ROWBYTES + ((unit ^ ((row & 3) << 1)) << 5) + (i & 3) * 8 ; row & 3 == i >> 2 for both reads (pix0 % 8 == 0, +4)
  const int li = lane & 15;
  const int sw = (li >> 2) << 1;
  const int rsub = (lane >> 5) * 8 + (li >> 2);               // row within a 16-pixel sub-step (second read: + 4)
  int offA[TA], offB[TB];
#pragma unroll
  for (int a = 0; a < TA; ++a) {
    const int unit = (wa * AWv + a * 32) / 16 + ((lane >> 4) & 1);
    offA[a] = rsub * (BA * 2) + ((unit ^ sw) << 5) + (li & 3) * 8;
  }
#pragma unroll
  for (int b = 0; b < TB; ++b) {
    const int unit = (wb * BWv + b * 32) / 16 + ((lane >> 4) & 1);
    offB[b] = A_BYTES + rsub * (BN * 2) + ((unit ^ sw) << 5) + (li & 3) * 8;
  }
  auto tr8 = [&](const char* q, int rowbytes) {
    const fp16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(q));
    const fp16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(q + 4 * rowbytes));
    h8 v;
    v[0] = r0[0]; v[1] = r0[1]; v[2] = r0[2]; v[3] = r0[3]; v[4] = r1[0]; v[5] = r1[1]; v[6] = r1[2]; v[7] = r1[3];
    return v;
  };

  for (int kt = 0; kt < nkt; ++kt) {
    const int ahead = (nkt - 1 - kt) < (NSTAGE - 2) ? (nkt - 1 - kt) : (NSTAGE - 2);   // stages still allowed in flight
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool refill = kt + NSTAGE - 1 < nkt;
    if (SPREAD == 0 && refill) issue(kt + NSTAGE - 1);      // refills the buffer read in iteration kt-1
    const char* st = smem + (kt % NSTAGE) * STAGE_BYTES;
    auto sub = [&](auto KS) __attribute__((always_inline)) {
      constexpr int ks = decltype(KS)::value;
      h8 af[TA], bf[TB];
#pragma unroll
      for (int a = 0; a < TA; ++a) af[a] = tr8(st + ks * 16 * (BA * 2) + offA[a], BA * 2);
#pragma unroll
      for (int b = 0; b < TB; ++b) bf[b] = tr8(st + ks * 16 * (BN * 2) + offB[b], BN * 2);
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
      if constexpr (SPREAD > 0 && ks < SPREAD) { if (refill) issue2d_part(kt + NSTAGE - 1, KS, std::integral_constant<int, SPREAD>{}); }
    };
    static_assert(BP / 16 == 4, "four sub-steps per stage");
    sub(std::integral_constant<int, 0>{}); sub(std::integral_constant<int, 1>{});
    sub(std::integral_constant<int, 2>{}); sub(std::integral_constant<int, 3>{});
  }

  // ---- epilogue: D[a][col], lane: col = lane%32, rows (r&3)+8*(r>>2)+4*(lane>>5); every (row < ca, col < ktot) element of this
  // split's slab is written exactly once (csbsr_unpack_wgrad sums the slabs)
  float* slab = p.g + (size_t)zsplit * p.slab_stride + (size_t)p.row0 * p.ktot;
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int col = col0 + wb * BWv + b * 32 + (lane & 31);
      if (col >= p.ktot) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = a0 + wa * AWv + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= p.ca) continue;
        slab[(size_t)row * p.ktot + col] = acc[a][b][r];
      }
    }
#endif
}

static half_t* g_wg_zero_page[CSBSR_MAX_DEVICES] = {};

template <int BA, int BN, int NWA, int NWB, int NSTAGE, bool T2D>
static int launch_wgrad_glds_t(const WgradK& k, int splits, hipStream_t st) {
  WgradK p = k;
  p.tiles_a = (unsigned)((k.ca + BA - 1) / BA);
  p.tiles_b = (unsigned)((k.ktot + BN - 1) / BN);
  const unsigned ntile = p.tiles_a * p.tiles_b;
  p.per_split = ((k.M + splits - 1) / splits + WG_BP - 1) / WG_BP * WG_BP;
  if ((int)((k.M + p.per_split - 1) / p.per_split) != splits) {
    csbsr_set_error("wgrad(glds): splits=%d leaves an empty slab; use csbsr_wgrad_splits_desc()", splits);
    return 1;
  }
  p.splits = splits;
  constexpr int SM_BYTES = NSTAGE * WG_BP * (BA + BN) * 2;
  static_assert(SM_BYTES <= 160 * 1024, "LDS budget");
  static LdsAttrOnce attr;
  if (int e = csbsr_lds_attr(attr, reinterpret_cast<const void*>(conv_wgrad_glds_kernel<BA, BN, NWA, NWB, NSTAGE, T2D>), SM_BYTES, "wgrad(glds)")) return e;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CSBSR_MAX_DEVICES) { csbsr_set_error("wgrad(glds): no current device"); return 2; }
  if (!g_wg_zero_page[dev]) {
    if (hipMalloc(reinterpret_cast<void**>(&g_wg_zero_page[dev]), 256) != hipSuccess) { csbsr_set_error("wgrad(glds): zero page alloc failed"); return 2; }
    (void)hipMemset(g_wg_zero_page[dev], 0, 256);
  }
  if (BN == 512 && ntile == 16 && splits % 2 == 0 && p.tap_perm) p.flat = 2;      // the XCD-by-(ky pair, split parity) grid of the four-tap tiles
  dim3 grid(p.flat ? ntile * splits : ntile, 1, p.flat ? 1 : splits);
  hipLaunchKernelGGL((conv_wgrad_glds_kernel<BA, BN, NWA, NWB, NSTAGE, T2D>), grid, dim3(64 * NWA * NWB), SM_BYTES, st, p, g_wg_zero_page[dev]);
  CSBSR_LAUNCH_CHECK("csbsr_conv_wgrad(glds)");
  return 0;
}

// 2-D stage rectangles (see the kernel): one gathered segment, a power-of-two width <= 64 that divides AW with the matching height
// dividing AH (then every split is a whole number of rectangles), single-row rectangles where the row shift is on, 32-bit piece offsets
static bool wgrad_glds_t2d(WgradK& p) {
  p.tw_log = 0; p.gx = p.gy = 0;
  if (!(g_wgrad_glds & 128) || p.b[1].ptr != p.b[0].ptr || p.cb0 != p.cbtot) return false;
  int tw = 1;
  while (tw < 64 && p.AW % (tw * 2) == 0) tw *= 2;
  const int th = 64 / tw;
  if (tw < 4 || p.AH % th != 0 || (p.row_shift && th != 1)) return false;
  if (p.a_sy * (long)th >= (1l << 29) || p.b[0].sy * (long)(th * p.stride + p.KH * p.dil) >= (1l << 29)) return false;
  int l = 0;
  while ((1 << l) < tw) ++l;
  p.tw_log = l; p.gx = p.AW / tw; p.gy = p.AH / th;
  return true;
}

template <int BA, int BN, int NWA, int NWB, int NSTAGE>
static int launch_wgrad_glds(const WgradK& k, int splits, hipStream_t st) {
  WgradK p = k;
  if (wgrad_glds_t2d(p)) return launch_wgrad_glds_t<BA, BN, NWA, NWB, NSTAGE, true>(p, splits, st);
  return launch_wgrad_glds_t<BA, BN, NWA, NWB, NSTAGE, false>(p, splits, st);
}

bool wgrad_glds_eligible(const WgradK& k) { return g_wgrad_glds != 0 && k.ca > 64; }
// Tile menu (measured at N = 4, scripts/bench_wgrad_ab.sh): 256-row tiles (8 waves, two 64 KB stages) for the multiple-of-256 part of
// the A channels wherever the columns fill 256-wide tiles about as well as 128-wide ones (the caller runs the remaining rows as a
// second, 128-row launch); 128 x 256 for the layers with thousands of columns and for the tap-permuted 8x8 stride-4 layers (one tile =
// the tap pair (ky, kx0), (ky, kx0 + 1): 697 -> 739 TF/s against the square tile); 128 x 128 otherwise (two workgroups per CU: a
// four-stage ring with one workgroup per CU measured 548 instead of 697).
static bool pad_ok(int c, int t) { const int p128 = (c + 127) / 128 * 128, pt = (c + t - 1) / t * t; return pt * 8 <= p128 * 9; }
int wgrad_glds_tile_a(const WgradK& k) {
  // (round 5) 512 .. 1023 columns too, whatever the padding: the mirrored problems of the 64-cout 3x3 layers (9 taps x 64 = 576 columns, rows =
  // 256 / 512 input channels) run three 256-wide column tiles instead of five 128-wide ones -- a quarter of the MFMA work is padding, but every
  // input pixel is staged 3 x instead of 5 x: config 5's 505 -> 64 at HR 11.6 -> 8.6 ms per launch (646 -> 871 TF/s), the decoder's 256 -> 64
  // 2.53 -> 2.25 ms (same-process A/B, scripts/wgrad_mirror_ab.py); in the step config 5 591 -> 570 ms, config 2 +-0.  Whole 256-row tiles only:
  // with a remainder launch (HRNet's 720-channel 1x1 layers at 448^2) config 4 lost 10 ms per step
  if ((g_wgrad_glds & 2) && (g_wgrad_glds & 512) && k.ca >= 256 && k.ca % 256 == 0 && k.ktot >= 512 && k.ktot < 1024 && !k.tap_perm) return 256;
  return (g_wgrad_glds & 2) && k.ca >= 256 && k.ktot >= 1024 && pad_ok(k.ktot, 256) && !k.tap_perm ? 256 : 128;
}
int wgrad_glds_tile_n(const WgradK& k) {
  // bit 8 (off: csbsr_debug_set_wgrad_tr bit 23): a 128 x 512 tile -- four taps of a kernel row of the 8x8 stride-4 layers, every wave 64 x 128:
  // three LDS fragment reads per four MFMAs where the 128 x 256 tap-pair tile's 64 x 64 waves need four -- in two 80 KB stages, the whole
  // LDS of a CU.  N = 4: 2.09 -> 1.92 ms per launch (805 -> 879 TF/s); in the step 1046 -> 1044 ms.
  if ((g_wgrad_glds & 256) && k.tap_perm && k.ktot == 8192 && k.ca == 128) return 512;
  return (k.ktot >= 6144 && (!k.tap_perm || (g_wgrad_glds & 64)) && !(g_wgrad_glds & 4)) ? 256 : 128;
}
int wgrad_glds_launch(const WgradK& k, int ta, int tn, int splits, hipStream_t st) {
  if (tn == 512) {
    return launch_wgrad_glds<128, 512, 2, 4, 2>(k, splits, st);
  }
  if (ta == 256) return launch_wgrad_glds<256, 256, 2, 4, 2>(k, splits, st);
  if (tn == 256) return launch_wgrad_glds<128, 256, 2, 4, 3>(k, splits, st);
  return launch_wgrad_glds<128, 128, 2, 2, 2>(k, splits, st);
}
