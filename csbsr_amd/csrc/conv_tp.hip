// Transposed convolution with a 2 x 2-tap phase decomposition (kernel <= 2 x stride: DBPN's 8x8 stride-4 / 12x12 stride-8 up-projections,
// kbpn.py:230-262 UpBlock / DownBlock deconvs, and -- same geometry -- the dgrads of their strided convolutions), 128 input channels ->
// 128 output channels.  Per output phase (oy % s, ox % s) the layer is a 2x2 convolution of the LOW-resolution input with that phase's
// weights, K = 4 taps x cin = 512: short-K GEMMs that the general LDS-DMA kernel ran at ~520 TF/s in the training step because every
// (tile, phase, tap) re-fetched its pixel operand through L2 (rocprof: 80 % hit rate, the 20 % at fabric latency behind a 2-3 stage ring)
// and paid a 3.3 us prologue + 3.3 us LDS-staged epilogue around 6.7 us of K loop.  Here:
//
//  * one persistent workgroup per CU (4 waves, one per SIMD, ~450 VGPR + AGPR each) owns an 8 x 32 tile of input positions; its
//    (8+2) x (32+2) halo goes HBM -> LDS ONCE (global_load_lds_dwordx4, zero page outside the image) and serves all taps of up to 8
//    phases: the pixel operand crosses the fabric ~1.3 times instead of 64;
//  * a wave owns 32 output channels x all 256 pixels of the tile.  Its weight operand never touches LDS: the weights are packed in
//    MFMA-fragment order (csbsr_pack_weights_tp), a K step's 32 couts x 64 channels are four 16-byte loads per lane from L2 (<= 2 MB
//    per layer, resident), kept three steps ahead in registers.  So the K loop has NO barrier and no hand-counted vmcnt: the only
//    workgroup synchronisation is the halo reload, once per 64 K steps;
//  * pixel fragments are ds_read_b128 at (per-phase lane base + compile-time offset): odd 16-byte pixel pitch, no swizzle, 0 bank
//    conflicts (SQ_LDS_BANK_CONFLICT); one read per MFMA, issued eight MFMAs ahead into the register the MFMA has just consumed;
//  * D = W x X^T orientation with the couts of a 32-row tile permuted in the pack (cout 16 (q/2) + 8 h + 4 (q%2) + j on MFMA row 8 q + 4 h + j): a
//    lane's accumulator registers are two octets of consecutive output channels of one pixel, no cross-lane exchange -- the epilogue (bias, PReLU, residual add / subtract, accumulate, activation-derivative mask) runs in registers with 16-byte operand loads;
//    the finished row is transposed through a wave-private 2.5 KB LDS tile so that its stores are 64 contiguous bytes per pixel (round 6);
//  * the epilogue of phase i is software-pipelined under the K loop of phase i + 1 (second accumulator set): two 8-cout x 32-pixel
//    pieces per K step, their residual / old-output / mask operands loaded one step ahead, their arithmetic cut into chunks that sit
//    in the MFMA shadows (the wave is alone on its SIMD and issues in order; __builtin_amdgcn_sched_barrier pins the order).  Which
//    operands the epilogue reads is a template parameter so that a K step is one basic block; dead lanes of ragged tiles are
//    redirected (loads to offset 0, stores to a sink), never branched around.
//
// Measured (N = 4, 448^2 -> 1792^2, 128 -> 128, 8x8 stride 4): 2.66 ms (633 TF/s) -> 2.00 ms (840 TF/s); in the training step the 21
// launches per micro-batch went 68 -> 45 ms.  rocprof: MFMA pipe 56 % busy in cycles at a ~1.4 GHz effective clock under the counters.
#include "common.h"
#include "conv_common.h"
#include "csbsr_debug.h"
#include <cstdlib>

#define TP_TH 8
#define TP_TW 32
#define TP_HW (TP_TW + 2)
#define TP_HH (TP_TH + 2)
#define TP_NPIX (TP_HH * TP_HW)          // 340 halo pixels
#define TP_STAGE 16384                   // bytes of one K step's weights: 128 couts x 64 channels, [cout tile][k-slice][lane][8]
#define TP_MAXPG 8                       // phases per work item
#define TP_TPITCH 80                     // pitch of a pixel in a wave's output transposition tile: its 32 couts x 2 bytes + 16

struct ConvTpK {
  const half_t* in; long i_sn, i_sy, i_sx;
  int N, H, W;                      // input (low-resolution) size; output = stride x that
  int stride, pad;
  const half_t* wt;                 // [phase][4 taps x NKC][16 KB: cout tile mt, k-slice kk, lane, 8 halves]
  int cout, coutp;
  half_t* out16; long o_sn, o_sy, o_sx;
  const float* bias; int act; float slope; const float* prelu; float out_scale;
  int res_mode; const half_t* res; long r_sn, r_sy, r_sx;
  int accumulate;
  const half_t* mask; long m_sn, m_sy, m_sx; float mask_slope; const float* mask_prelu;
  half_t* dres; long d_sn, d_sy, d_sx;   // HAS_STAT + residual: gradient wrt the masking layer's residual operand (second output)
  float* part; long part_ld;        // HAS_STAT: per-workgroup partial rows [gridDim.x][part_ld]: bias-gradient sums in [0, coutp), PReLU-slope sum at coutp
  unsigned tiles_x, tiles_y, pg, pgroups;     // phases per work item, items per tile
  int dbg;                          // ablation bit (CSBSR_TP_DBG): 1 stores go to the sink
};

__device__ __forceinline__ void tp_wait_barrier_all() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// where one phase of one tile lands: image bases + byte offset of the tile's first pixel of that phase + what part of it is inside
struct TpCtx {
  const char *ob, *rb, *mb, *db;
  unsigned to, tr, tm, td;
  int xlim, ylim;                   // lane live iff pix < xlim; the wave's row nt live iff nt < ylim
};

// HAS_STAT (accumulate / residual + mask launches): also the masking layer's bias / PReLU-slope gradient sums.  The 16 per-lane bias
// sums do not fit next to two accumulator sets in the 256 arch VGPRs (the other 256 registers are AGPRs, which only MFMA results can
// live in), so they live in LDS: the lane read-modify-writes its own slots once per piece.
template <int NKC, bool HAS_RES, bool HAS_ACC, bool HAS_MASK, bool HAS_STAT>
__global__ __launch_bounds__(256) void conv_tp_kernel(const ConvTpK p, const half_t* __restrict__ zero_page, half_t* __restrict__ sink) {
  constexpr int SLOTS = NKC * 8 + 1;                    // 16-byte slots per pixel: odd -> consecutive pixels walk all banks
  constexpr int PITCH = SLOTS * 16;
  constexpr int NG = TP_NPIX * SLOTS;
  constexpr int NINST = (NG + 63) / 64;                 // wave instructions that fill the halo tile
  constexpr int XBYTES = NINST * 1024;
  constexpr int BOFF = XBYTES;                          // 128 biases
  constexpr int DOFF = BOFF + 512;                      // 1 KB landing area for the halo DMA's padding instructions
  constexpr int SOFF = DOFF + 1024;                     // HAS_STAT: per-lane bias-gradient partial sums, [wave][octet half 0..3][lane] float4
  constexpr int TOFF = SOFF + 4 * 256 * 16 + 1024;      // output transposition tiles: [wave][32 pixels][TP_TPITCH]
  constexpr int NKS = 4 * NKC;                          // K steps per phase (64 channels of one tap each)
  constexpr int PPS = 16 / NKS;                         // epilogue pieces (of the previous phase) drained per K step
  constexpr int NFI = (NINST + 3) / 4;
  constexpr int DQ = 256 / SLOTS, DC = 256 % SLOTS;
  constexpr bool PIPE = true;
  constexpr int WD = PIPE ? 3 : 1;                      // weight stages in flight ahead of the MFMAs (WD + 1 register buffers)
  static_assert(NKS % (WD + 1) == 0 && PPS == 2, "buffer indices must be compile-time across phases");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sBias = reinterpret_cast<float*>(smem + BOFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // = the wave's 32-cout tile
  const int pix = lane & 31, hi = lane >> 5;
  const half_t* zp = zero_page + (lane & 7) * 8;
  const int s = p.stride;
  const unsigned per_img = p.tiles_x * p.tiles_y, ntiles = per_img * (unsigned)p.N, items = ntiles * p.pgroups;
  unsigned it = blockIdx.x;
  if (it >= items) return;
  if (tid < 128) sBias[tid] = (p.bias && tid < p.cout) ? p.bias[tid] : 0.f;
  const float slope = p.act == CSBSR_ACT_PRELU ? *p.prelu : (p.act == CSBSR_ACT_RELU ? 0.f : (p.act == CSBSR_ACT_NONE ? 1.f : p.slope));
  const float mslope = (HAS_MASK && p.mask_prelu) ? *p.mask_prelu : p.mask_slope;
  const float rsign = p.res_mode == CSBSR_RES_ADD ? 1.f : (p.res_mode == CSBSR_RES_SUB ? -1.f : 0.f);
  // (which operands the epilogue reads is compile-time: a K step is one basic block)
  constexpr bool has_res = HAS_RES, has_acc = HAS_ACC, has_mask = HAS_MASK;
  constexpr bool has_bias = !(HAS_ACC || HAS_MASK);       // the accumulate / mask variants are dgrad launches: no bias, no activation
  // a lane's share of every output / residual / mask address (bytes, within one image): its pixel column, cout tile, channel octet
  const unsigned lane_o = 2u * (unsigned)((s * pix) * (int)p.o_sx + 32 * wid + 8 * hi);
  const unsigned lane_r = 2u * (unsigned)((s * pix) * (int)p.r_sx + 32 * wid + 8 * hi);
  const unsigned lane_m = 2u * (unsigned)((s * pix) * (int)p.m_sx + 32 * wid + 8 * hi);
  const unsigned row_o = 2u * (unsigned)(s * (int)p.o_sy), row_r = 2u * (unsigned)(s * (int)p.r_sy), row_m = 2u * (unsigned)(s * (int)p.m_sy);
  const unsigned lane_d = 2u * (unsigned)((s * pix) * (int)p.d_sx + 32 * wid + 8 * hi), row_d = 2u * (unsigned)(s * (int)p.d_sy);
  f4* const sSb = reinterpret_cast<f4*>(smem + SOFF) + wid * 256 + lane;      // this lane's four float4 slots: + 64 * (2 * pair + half)
  char* const sink_l = reinterpret_cast<char*>(sink) + lane * 16;

  // halo-tile DMA roles (same for every tile): chunk g = (wid + 4 i) * 64 + lane = (halo pixel q, slot c)
  int f_ty0, f_tx0, f_c0;
  {
    const int g = wid * 64 + lane;
    const int q = g / SLOTS;
    f_c0 = g - q * SLOTS;
    f_ty0 = q / TP_HW;
    f_tx0 = q - f_ty0 * TP_HW;
  }
  const int isy = (int)p.i_sy, isx = (int)p.i_sx;
  const long stage_elems = TP_STAGE / 2;
  // a wave's A operand never touches LDS: its 32 couts x 64 channels of a stage are 4 fragment-ordered KB, one 16-byte load per lane each
  const unsigned wlane = (unsigned)((4 * wid) * 1024 + lane * 16);
  auto load_w = [&](const half_t* src, h8 (&w)[4]) __attribute__((always_inline)) {
    const char* b = reinterpret_cast<const char*>(src);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) w[kk] = *reinterpret_cast<const h8*>(b + (wlane + kk * 1024));
  };
  const char* xl = smem + pix * PITCH + hi * 16;          // B fragments: + per-phase pixel base + compile-time (tap, row, channel) offset

  // ---- epilogue pieces: piece pi = (row nt = pi / 2, octet pair = pi % 2) of a wave's 32 couts x 8 rows x 32 pixels.  The packed
  // weights put cout 16 (q / 2) + 8 h + 4 (q % 2) + j on MFMA row 8 q + 4 h + j, so accumulator registers 8 pair .. 8 pair + 7 of a lane
  // ARE the 8 consecutive couts 32 wid + 16 pair + 8 hi + (0..7) of its pixel (row nt, column pix) -- no cross-lane exchange, and the two
  // half-waves of one store instruction write one contiguous 32-byte segment per pixel; a piece is one such 16-byte octet.  Loads and stores are issued unconditionally (dead lanes read
  // offset 0 of the image / write the sink): no divergent control flow inside a K step.
  auto piece_loads = [&](const TpCtx& c, int pi, h8& r, h8& o, h8& m) __attribute__((always_inline)) {
    const int nt = pi >> 1, pair = pi & 1;
    const bool lv = pix < c.xlim && nt < c.ylim;
    if (has_res) { const unsigned off = lane_r + c.tr + nt * row_r + 32 * pair; r = *reinterpret_cast<const h8*>(c.rb + (lv ? off : 0u)); }
    if (has_acc) { const unsigned off = lane_o + c.to + nt * row_o + 32 * pair; o = *reinterpret_cast<const h8*>(c.ob + (lv ? off : 0u)); }
    if (has_mask) { const unsigned off = lane_m + c.tm + nt * row_m + 32 * pair; m = *reinterpret_cast<const h8*>(c.mb + (lv ? off : 0u)); }
  };
  // The piece arithmetic comes in chunks small enough for one MFMA shadow each (a wave is alone on its SIMD and issues in order, so
  // whatever follows an MFMA in program order runs while the matrix pipe works on it):
  // act chunk c = 0..3: couts c and 4 + c of the piece's octet: scale + bias + activation
  auto act_chunk = [&](const f16v (&pa)[8], int pi, int c, float (&v)[8], f4 (&bq)[2]) __attribute__((always_inline)) {
    const int nt = pi >> 1, pair = pi & 1;
    if (!has_bias) { v[c] = pa[nt][8 * pair + c] * p.out_scale; v[4 + c] = pa[nt][8 * pair + 4 + c] * p.out_scale; return; }
    if (c == 0) {
      const int co = 32 * wid + 16 * pair + 8 * hi;
      bq[0] = *reinterpret_cast<const f4*>(sBias + co); bq[1] = *reinterpret_cast<const f4*>(sBias + co + 4);
    }
    // (output channels >= cout need no masking: their packed weights and LDS biases are zero, and so are the padding channels of
    // the residual / old-output maps)
    float t0 = pa[nt][8 * pair + c] * p.out_scale + bq[0][c], t1 = pa[nt][8 * pair + 4 + c] * p.out_scale + bq[1][c];
    v[c] = t0 > 0.f ? t0 : t0 * slope;
    v[4 + c] = t1 > 0.f ? t1 : t1 * slope;
  };
  // store chunk c = 0..3: residual / old output / activation mask on couts c and 4 + c
  // (HAS_STAT: the masking layer's bias gradient = per-channel sum of the masked result, its PReLU-slope gradient = sum over the
  // pixels with mask <= 0 of (unmasked result) x mask, divided by the slope at the end -- csbsr_epilogue_backward's sums, per lane)
  // The 16 bias-gradient sums of a lane live in LDS (read-modify-write of the lane's own slots once per piece): next to two accumulator
  // sets there is no room for them in the 256 arch VGPRs -- and a loop-carried register for the slope sum alone costs 140 spills, so
  // that one is summed per piece (vd[8] / spl) and added to an LDS slot too.
  float* const sSpL = reinterpret_cast<float*>(smem + SOFF + 4 * 256 * 16) + wid * 64 + lane;
  if (HAS_STAT) {
#pragma unroll
    for (int q = 0; q < 4; ++q) sSb[64 * q] = f4{0.f, 0.f, 0.f, 0.f};
    *sSpL = 0.f;
  }
  // HAS_STAT with a residual operand: the masking layer was out = act(pre) +- res (DownBlock's down_conv2, kbpn.py:254-256), so its
  // activation is (mask -+ res), the residual does NOT enter this launch's result, and d(res) = +- (unmasked result) is a second output
  auto store_chunk = [&](int pair, int c, float (&v)[8], float (&vd)[9], const h8& r, const h8& o, const h8& m) __attribute__((always_inline)) {
    if (HAS_STAT && c == 0) vd[8] = 0.f;
#pragma unroll
    for (int e = c; e < 8; e += 4) {
      if (has_res && !HAS_STAT) v[e] += rsign * (float)r[e];
      if (has_acc) v[e] += (float)o[e];
      if (HAS_STAT && HAS_RES) vd[e] = rsign * v[e];
      if (has_mask) {
        const float mk = (HAS_STAT && HAS_RES) ? (float)m[e] - rsign * (float)r[e] : (float)m[e];
        if (HAS_STAT) vd[8] += mk > 0.f ? 0.f : v[e] * mk;
        v[e] *= (mk > 0.f ? 1.f : mslope);
      }
    }
  };
  // A row's two pieces leave TRANSPOSED: in the accumulator layout a lane owns 16 bytes of its pixel and the neighbour lane the next pixel
  // of the phase (stride x coutp x 2 bytes away), so a store instruction was 64 separate 16-byte write requests.  The two octet pairs of
  // a row go to a wave-private LDS tile [pixel][32 couts] (pitch 80 bytes: conflict-free 16-byte writes), come back as four lanes per
  // pixel, and leave as two stores of 16 pixels x 64 bytes.  Measured in the training step (config 2, same-box ABAB x 4, the 21 launches per
  // micro-batch): 974.1 -> 966.9 ms per step; a plain launch alone (N = 8, nothing but stores in its epilogue) does not move (3.61 -> 3.66
  // ms; 3.14 with the stores sent to the contiguous sink): what the fewer, larger write requests buy is room for the OTHER requests -- the
  // residual / old-output / mask reads of the launches that have them.  (Those reads fetched the same way -- LDS-DMA into a ring, read back
  // in the accumulator layout -- measured worse, 966.9 -> 971.5: the counted wait and the extra LDS reads sit in the K loop; not kept.)
  char* const tt = smem + TOFF + wid * (32 * TP_TPITCH);
  const int rpx = lane >> 2, rch = lane & 3;                     // read-back role: pixel rpx + 16 k, 16-byte chunk rch
  const unsigned lane_o2 = 2u * (unsigned)((s * rpx) * (int)p.o_sx + 32 * wid + 8 * rch), half_o2 = 2u * (unsigned)((s * 16) * (int)p.o_sx);
  auto fin_piece = [&](const TpCtx& c, int pi, const float (&v)[8], const float (&vd)[9]) __attribute__((always_inline)) {
    const int nt = pi >> 1, pair = pi & 1;
    h8 hv;
#pragma unroll
    for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
    *reinterpret_cast<h8*>(tt + pix * TP_TPITCH + pair * 32 + hi * 16) = hv;
    if (HAS_STAT) {
      f4 a = sSb[64 * (2 * pair)], b = sSb[64 * (2 * pair + 1)];
      a += f4{v[0], v[1], v[2], v[3]}; b += f4{v[4], v[5], v[6], v[7]};
      sSb[64 * (2 * pair)] = a; sSb[64 * (2 * pair + 1)] = b;
      *sSpL += vd[8];
    }
    if (HAS_STAT && HAS_RES) {      // (the second output of the three residual + mask launches: as before, 16 bytes per lane)
      h8 hd;
#pragma unroll
      for (int e = 0; e < 8; ++e) hd[e] = (half_t)vd[e];
      const bool lv = pix < c.xlim && nt < c.ylim && 32 * wid + 16 * pair + 8 * hi < p.coutp && !(p.dbg & 1);
      const unsigned offd = lane_d + c.td + nt * row_d + 32 * pair;
      char* dd = lv ? const_cast<char*>(c.db) + offd : sink_l;
      *reinterpret_cast<h8*>(dd) = hd;
    }
  };
  auto fin_read = [&](h8 (&tro)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) tro[k] = *reinterpret_cast<const h8*>(tt + (rpx + 16 * k) * TP_TPITCH + rch * 16);
  };
  auto fin_store = [&](const TpCtx& c, int nt, const h8 (&tro)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool lv = rpx + 16 * k < c.xlim && nt < c.ylim && 32 * wid + 8 * rch < p.coutp && !(p.dbg & 1);
      const unsigned off = lane_o2 + k * half_o2 + c.to + nt * row_o;
      char* dst = lv ? const_cast<char*>(c.ob) + off : sink_l;
      *reinterpret_cast<h8*>(dst) = tro[k];
    }
  };

  TpCtx cur, pend;
  cur.ob = pend.ob = reinterpret_cast<const char*>(p.out16);
  cur.rb = pend.rb = reinterpret_cast<const char*>(has_res ? p.res : p.out16);
  cur.mb = pend.mb = reinterpret_cast<const char*>(has_mask ? p.mask : p.out16);
  cur.db = pend.db = reinterpret_cast<const char*>(p.out16);
  cur.to = cur.tr = cur.tm = cur.td = pend.to = pend.tr = pend.tm = pend.td = 0u;
  if (HAS_STAT) {      // the not-yet-existing pending tile must add exact zeros to the sums: its operand loads hit the zero page (its
    pend.ob = pend.mb = pend.rb = reinterpret_cast<const char*>(zero_page);      // accumulators are zero, its stores go to the sink)
  }
  cur.xlim = cur.ylim = pend.xlim = pend.ylim = 0;        // nothing pending yet: its pieces load offset 0 and store to the sink
  // piece operands are requested LD K steps before the piece retires (LD + 1 register buffers; 8 % (LD + 1) == 0 keeps the indices
  // compile-time across phases): three steps (~1.7 us) ahead when only one operand is read, one step when two are (register budget)
  constexpr int NLOAD = (int)HAS_RES + (int)HAS_ACC + (int)HAS_MASK;
  constexpr int LD = NLOAD <= 1 ? 3 : 1, NB = LD + 1;
  h8 lr[NB][PPS], lo[NB][PPS], lm[NB][PPS];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b < PPS; ++b) lr[a][b] = lo[a][b] = lm[a][b] = h8{0, 0, 0, 0, 0, 0, 0, 0};

  h8 wreg[WD + 1][4];                                     // stage ks of a phase lives in wreg[ks % (WD + 1)]
  {                                                       // the first phase's stages 0 .. WD - 1
    const half_t* w0 = p.wt + (size_t)((it / ntiles) * p.pg) * NKS * stage_elems;
#pragma unroll
    for (int d = 0; d < WD; ++d) load_w(w0 + d * stage_elems, wreg[d]);
  }
  h8 bfr[8];                                              // B fragments of the k-slice in flight: pixel row nt, one read per MFMA

  unsigned pgi = 0, itn = 0;
  int n = 0, Y0 = 0, X0 = 0;
  // ---- one phase: K loop into `acc` while the previous phase's accumulators `pd` drain, two pieces per K step.  No barrier in here:
  // the pixel operand is the (static) LDS tile, the weight operand is private to the wave.
  auto phase = [&](f16v (&acc)[8], const f16v (&pd)[8], unsigned j, bool first_of_item) __attribute__((always_inline)) {
    const unsigned ph = pgi * p.pg + j;
    const int py = ph / s, px = ph - py * s;
    const int by = (py + p.pad) / s, bx = (px + p.pad) / s;
    const bool last_of_item = j + 1 >= p.pg;
    const unsigned phn = !last_of_item ? ph + 1 : (itn < items ? (itn / ntiles) * p.pg : ph);   // (last phase of all: harmless refetch)
    const int pyn = phn / s, pxn = phn - pyn * s;
    const half_t* wcur = p.wt + (size_t)ph * NKS * stage_elems;
    const half_t* wnext = p.wt + (size_t)phn * NKS * stage_elems;
    // rows 0..3 and 4..7 of the tile get their own base register (the 16-bit ds_read offset does not reach over all eight)
    const char* xph = xl + (by * TP_HW + bx) * PITCH;
    const char* xphn = xl + (((pyn + p.pad) / s) * TP_HW + (pxn + p.pad) / s) * PITCH;
    cur.ob = reinterpret_cast<const char*>(p.out16 + n * p.o_sn);
    cur.rb = reinterpret_cast<const char*>(has_res ? p.res + n * p.r_sn : p.out16);
    cur.mb = reinterpret_cast<const char*>(has_mask ? p.mask + n * p.m_sn : p.out16);
    cur.to = 2u * (unsigned)((s * Y0 + py) * (int)p.o_sy + (s * X0 + px) * (int)p.o_sx);
    cur.tr = 2u * (unsigned)((s * Y0 + py) * (int)p.r_sy + (s * X0 + px) * (int)p.r_sx);
    cur.tm = 2u * (unsigned)((s * Y0 + py) * (int)p.m_sy + (s * X0 + px) * (int)p.m_sx);
    if (HAS_STAT && HAS_RES) {
      cur.db = reinterpret_cast<const char*>(p.dres + n * p.d_sn);
      cur.td = 2u * (unsigned)((s * Y0 + py) * (int)p.d_sy + (s * X0 + px) * (int)p.d_sx);
    }
    cur.xlim = p.W - X0; cur.ylim = p.H - Y0;
    // B fragment (row nt) of k-slice (ks, kk) of the phase whose lane base is xb
    auto rdb = [&](const char* xb, int ks, int kk, int nt) __attribute__((always_inline)) {
      const int tap = ks / NKC, kc = ks % NKC;
      const int jy = tap >> 1, jx = tap & 1;
      const int xo = ((1 - jy) * TP_HW + (1 - jx)) * PITCH + kc * 128 + kk * 32;
      return *reinterpret_cast<const h8*>((nt < 4 ? xb : xb + 4 * TP_HW * PITCH) + xo + (nt & 3) * TP_HW * PITCH);
    };
    if (first_of_item) {
      // ---- halo tile -> LDS (every wave has to be past its last read of the old tile)
      __syncthreads();
      const half_t* tbase = p.in + n * p.i_sn + (long)(Y0 - 1) * p.i_sy + (long)(X0 - 1) * p.i_sx;
      int ty = f_ty0, tx = f_tx0, c = f_c0;
#pragma unroll 2
      for (int i = 0; i < NFI; ++i) {
        const int inst = wid + 4 * i;
        const int iy = Y0 - 1 + ty, ix = X0 - 1 + tx;
        const bool ok = ty < TP_HH && c < NKC * 8 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const half_t* src = ok ? tbase + (ty * isy + tx * isx + c * 8) : zp;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + (inst < NINST ? inst * 1024 : DOFF)), 16, 0, 0);
        c += DC; tx += DQ;
        if (c >= SLOTS) { c -= SLOTS; ++tx; }
        if (tx >= TP_HW) { tx -= TP_HW; ++ty; }
        if (tx >= TP_HW) { tx -= TP_HW; ++ty; }
      }
      tp_wait_barrier_all();
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) bfr[nt] = rdb(xph, 0, 0, nt);
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      // the pieces of LD steps ahead: their operands' requests
      if (PIPE)
#pragma unroll
      for (int pp = 0; pp < PPS; ++pp) {
        const int pn_ = (ks + LD) * PPS + pp;
        if (pn_ < 16) piece_loads(pend, pn_, lr[(ks + LD) % NB][pp], lo[(ks + LD) % NB][pp], lm[(ks + LD) % NB][pp]);
        else piece_loads(cur, pn_ - 16, lr[(ks + LD) % NB][pp], lo[(ks + LD) % NB][pp], lm[(ks + LD) % NB][pp]);
      }
      // the weight stage WD steps ahead
      load_w(ks + WD < NKS ? wcur + (ks + WD) * stage_elems : wnext + (ks + WD - NKS) * stage_elems, wreg[(ks + WD) % (WD + 1)]);
      __builtin_amdgcn_sched_barrier(0);
      float v[PPS][8], vdd[HAS_STAT ? PPS : 1][9];     // [8]: the piece's slope-gradient sum
      f4 bq[PPS][2];
      h8 tro[2];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const h8 af = wreg[ks % (WD + 1)][kk];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfr[i], acc[i], 0, 0, 0);
          // the same row's fragment of the next k-slice (of the next step / next phase at the ends; across a tile change it reads
          // the old tile and is replaced after the reload)
          if (kk < 3) bfr[i] = rdb(xph, ks, kk + 1, i);
          else if (ks + 1 < NKS) bfr[i] = rdb(xph, ks + 1, 0, i);
          else bfr[i] = rdb(xphn, 0, 0, i);
          // the pending pieces' arithmetic and stores, one chunk per MFMA shadow, spread over the four k-slices
          // piece 0 (octet pair 0 of row ks): activation in k-slice 0, operands in 1, to the transposition tile at (1, 4); piece 1: (1, 4..7),
          // 2, (2, 4); the row comes back at (2, 6) and leaves at (3, 4) -- five MFMAs between the LDS reads and their use
          if (PIPE) {
            if (kk == 0 && i < 4) act_chunk(pd, ks * PPS, i, v[0], bq[0]);
            if (kk == 1 && i < 4) store_chunk(0, i, v[0], vdd[0], lr[ks % NB][0], lo[ks % NB][0], lm[ks % NB][0]);
            if (kk == 1 && i == 4) fin_piece(pend, ks * PPS, v[0], vdd[0]);
            if (kk == 1 && i >= 4) act_chunk(pd, ks * PPS + 1, i - 4, v[1], bq[1]);
            if (kk == 2 && i < 4) store_chunk(1, i, v[1], vdd[HAS_STAT ? 1 : 0], lr[ks % NB][1], lo[ks % NB][1], lm[ks % NB][1]);
            if (kk == 2 && i == 4) fin_piece(pend, ks * PPS + 1, v[1], vdd[HAS_STAT ? 1 : 0]);
            if (kk == 2 && i == 6) fin_read(tro);
            if (kk == 3 && i == 4) fin_store(pend, ks, tro);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    pend = cur;
  };

  f16v accA[8], accB[8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[a][r] = 0.f;

  for (; it < items; it += gridDim.x) {
    pgi = it / ntiles;
    const unsigned tile = it - pgi * ntiles;
    n = tile / per_img;
    const unsigned r_ = tile - n * per_img;
    Y0 = (r_ / p.tiles_x) * TP_TH; X0 = (r_ % p.tiles_x) * TP_TW;
    itn = it + gridDim.x;
    for (unsigned j = 0; j < p.pg; j += 2) {             // (pg is even: the accumulator sets alternate)
      phase(accA, accB, j, j == 0);
      phase(accB, accA, j + 1, false);
    }
  }
  // ---- the last phase's accumulators (in accB) drain on their own
  if (PIPE)
#pragma unroll
  for (int pi = 0; pi < 16; ++pi) {
    h8 r = h8{0, 0, 0, 0, 0, 0, 0, 0}, o = r, m = r;
    float v[8];
    f4 bq[2];
    piece_loads(pend, pi, r, o, m);
#pragma unroll
    for (int c = 0; c < 4; ++c) act_chunk(accB, pi, c, v, bq);
    float vd[9];
#pragma unroll
    for (int c = 0; c < 4; ++c) store_chunk(pi & 1, c, v, vd, r, o, m);
    fin_piece(pend, pi, v, vd);
    if (pi & 1) {
      h8 tro[2];
      fin_read(tro);
      fin_store(pend, pi >> 1, tro);
    }
  }
  if (HAS_STAT) {
    // lanes of one half-wave hold the same couts (32 wid + 16 pair + 8 hi + e at sb[8 pair + e]) for 32 different pixels
    float sb[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 t = sSb[64 * q];
      sb[4 * q] = t[0]; sb[4 * q + 1] = t[1]; sb[4 * q + 2] = t[2]; sb[4 * q + 3] = t[3];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) sb[e] += __shfl_xor(sb[e], o, 64);
    }
    float sp = wave_sum(*sSpL);
    float* row = p.part + (size_t)blockIdx.x * p.part_ld;
    float* sSp = reinterpret_cast<float*>(smem + DOFF);           // (the halo DMA's padding KB is idle by now)
    __syncthreads();
    if (pix == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = 32 * wid + 16 * (e >> 3) + 8 * hi + (e & 7);
        if (co < p.coutp) row[co] = sb[e];
      }
    }
    if (lane == 0) sSp[wid] = sp;            // the four waves' sums are added in wave order (no LDS atomic: fixed order)
    __syncthreads();
    if (tid == 0) row[p.coutp] = (((sSp[0] + sSp[1]) + sSp[2]) + sSp[3]) / mslope;
  }
}

// ---- weights in K-step order: dst[phase][ks = tap * NKC + kc][mt][kk][lane][e] =
//        W(cout 32 mt + perm(lane % 32), perm(8 q + 4 h + j) = 16 (q / 2) + 8 h + 4 (q % 2) + j, channel 64 kc + 16 kk + 8 (lane / 32) + e, kh = (py + pad) % s + s jy, kw likewise), tap = 2 jy + jx
// W is indexed [contracted channel][row][kh][kw]: ConvTranspose2d's IOHW parameter, or a Conv2d's OIHW parameter seen from its dgrad
// (contracted = the conv's output channels, rows = its input channels) -- csbsr_pack_weights kind 2 in this kernel's order.
struct PackTpK { const float* w; half_t* dst; int D1, KH, KW, stride, pad, nkc, c_real, rows_real, row_off, k_off; };
__global__ void pack_weights_tp_kernel(const PackTpK p, long total) {
  const int nks = 4 * p.nkc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), blk = (int)((i >> 9) & 15);
    const long stage = i >> 13;
    const int ks = (int)(stage % nks), ph = (int)(stage / nks);
    const int mt = blk >> 2, kk = blk & 3;
    const int m = lane & 31;                                 // MFMA row of the A operand -> the cout it carries (see the epilogue)
    const int q_ = m >> 3, h_ = (m >> 2) & 1;
    const int row = 32 * mt + 16 * (q_ >> 1) + 8 * h_ + 4 * (q_ & 1) + (m & 3);
    const int tap = ks / p.nkc, kc = ks % p.nkc;
    const int c = 64 * kc + 16 * kk + 8 * (lane >> 5) + e;
    const int jy = tap >> 1, jx = tap & 1;
    const int py = ph / p.stride, px = ph % p.stride;
    const int kh = (py + p.pad) % p.stride + p.stride * jy, kw = (px + p.pad) % p.stride + p.stride * jx;
    float v = 0.f;
    if (row < p.rows_real && c < p.c_real && kh < p.KH && kw < p.KW)
      v = p.w[(((long)(p.k_off + c) * p.D1 + p.row_off + row) * p.KH + kh) * p.KW + kw];
    p.dst[i] = (half_t)v;
  }
}

static int tp_nkc(int cin_padded) { return 2; }      // 128 padded input channels = two 64-channel K chunks per tap (the only instantiation)

extern "C" int64_t csbsr_packed_weight_elems_tp(int32_t stride, int32_t c_real) {
  const int nkc = tp_nkc(round_up(c_real, 8) <= 64 ? 64 : 128);
  return (int64_t)stride * stride * 4 * nkc * (TP_STAGE / 2);
}

extern "C" int csbsr_pack_weights_tp(const float* w, void* dst, int32_t D0, int32_t D1, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                                     int32_t c_real, int32_t rows_real, int32_t row_off, int32_t k_off, csbsr_stream_t s) {
  CSBSR_CHECK(w && dst, "pack_tp: null pointer");
  CSBSR_CHECK(stride >= 2 && KH == KW && KH > stride && KH <= 2 * stride && pad >= 0 && pad < stride, "pack_tp: needs stride < kernel <= 2 stride");
  CSBSR_CHECK(c_real > 64 && c_real <= 128 && rows_real >= 1 && rows_real <= 128, "pack_tp: at most 128 channels either side");
  CSBSR_CHECK(k_off >= 0 && k_off + c_real <= D0 && row_off >= 0 && row_off + rows_real <= D1, "pack_tp: range out of bounds");
  PackTpK p;
  p.w = w; p.dst = reinterpret_cast<half_t*>(dst); p.D1 = D1; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.nkc = tp_nkc(round_up(c_real, 8) <= 64 ? 64 : 128);
  p.c_real = c_real; p.rows_real = rows_real; p.row_off = row_off; p.k_off = k_off;
  const long total = csbsr_packed_weight_elems_tp(stride, c_real);
  hipLaunchKernelGGL(pack_weights_tp_kernel, dim3((int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(s), p, total);
  CSBSR_LAUNCH_CHECK("csbsr_pack_weights_tp");
  return 0;
}

static int g_conv_tp_mode = 1;      // 0 off, 1 problems that fill the chip, 2 every eligible launch (tests)
extern "C" void csbsr_debug_set_conv_tp(int mode) { g_conv_tp_mode = mode; }

// Which launches take this kernel: transposed, stride < kernel <= 2 stride (2 x 2 taps per phase), output exactly stride x input,
// one plain-fp16 input segment of 128 padded channels (65..128 real), 65..128 (padded: 72..128) output channels, bias / ReLU / leaky / PReLU /
// residual add or subtract / accumulate / activation mask; no fp32 or split outputs, statistics or constant-segment bias.
extern "C" int32_t csbsr_conv_tp_eligible(const csbsr_conv_desc_t* d) {
  if (!d || !g_conv_tp_mode || !d->transposed || d->KH != d->KW || d->stride < 2 || (d->stride & 1) || d->KH <= d->stride || d->KH > 2 * d->stride) return 0;
  if (d->pad < 0 || d->pad >= d->stride || d->dil != 1 || d->OH != d->H * d->stride || d->OW != d->W * d->stride) return 0;
  if (d->in[1].c != 0 || d->in[0].sx == 0 || d->in[0].c != 128) return 0;
  if (d->coutp <= 64 || d->coutp > 128 || !d->out16 || d->out32 || d->cbias || d->o_lo || d->r_lo || d->r2_lo) return 0;
  if (d->stat_mode != CSBSR_STAT_NONE) return 0;
  // residual together with a mask: only as the masking layer's own residual (dres given: its gradient leaves as a second output)
  if (d->res_mode != CSBSR_RES_NONE && (d->accumulate || (d->mask && !d->dres))) return 0;
  if (d->dres && !(d->mask && d->res_mode != CSBSR_RES_NONE && !d->accumulate && d->H % TP_TH == 0 && d->W % TP_TW == 0 &&
                   (long)d->OH * d->dr_sy < (1l << 31))) return 0;
  if ((d->accumulate || d->mask) && (d->bias || d->act != CSBSR_ACT_NONE)) return 0;
  if (d->bias_sn) return 0;                 // (per-sample bias: the general kernels and conv_x3 only)
  // the fused bias / PReLU-slope sums: accumulate + mask launches over whole tiles (a dead lane would add its garbage to the sums)
  if ((d->dact_bias || d->dact_prelu) && !((d->accumulate || d->dres) && d->mask && d->H % TP_TH == 0 && d->W % TP_TW == 0)) return 0;
  if (d->mask_prelu && !d->mask) return 0;
  if (d->res_mode != CSBSR_RES_NONE && d->res_mode != CSBSR_RES_ADD && d->res_mode != CSBSR_RES_SUB) return 0;
  if (d->act == CSBSR_ACT_SIGMOID) return 0;
  // in-image offsets are 32-bit byte offsets
  const long lim = 1l << 31;
  if (d->in[0].sn >= lim || (long)d->OH * d->o_sy >= lim || (d->res_mode != CSBSR_RES_NONE && (long)d->OH * d->r_sy >= lim) ||
      (d->mask && (long)d->OH * d->m_sy >= lim)) return 0;
  if (g_conv_tp_mode == 1 && (long)d->N * d->H * d->W < 128L * TP_TH * TP_TW) return 0;
  return 1;
}

static half_t* g_tp_zero_page[CSBSR_MAX_DEVICES] = {};      // 256 B of zeros + a 1 KB sink for the dead lanes' stores

template <int NKC, bool R, bool A, bool M, bool S>
static int launch_tp(ConvTpK& k, hipStream_t st, half_t* zp, float* dbias, float* dprelu) {
  constexpr int SLOTS = NKC * 8 + 1;
  constexpr int NINST = (TP_NPIX * SLOTS + 63) / 64;
  constexpr int SM_BYTES = NINST * 1024 + 128 * 4 + 1024 + 4 * 256 * 16 + 1024 + 4 * 32 * TP_TPITCH;
  static_assert(SM_BYTES <= 160 * 1024, "LDS budget");
  static LdsAttrOnce attr;
  if (int e = csbsr_lds_attr(attr, reinterpret_cast<const void*>(conv_tp_kernel<NKC, R, A, M, S>), SM_BYTES, "conv_tp")) return e;
  const unsigned items = k.tiles_x * k.tiles_y * k.N * k.pgroups;
  const int ncu = csbsr_cu_budget(st);      // the stream's CU partition (csrc/streams.hip), else the whole device
  const unsigned g = items < (unsigned)ncu ? items : (unsigned)ncu;
  if (S) {
    k.part_ld = k.coutp + 8;
    k.part = csbsr_red_scratch((long)g * k.part_ld);
    CSBSR_CHECK(k.part, "conv_tp: the fused activation-gradient sums need the reduction scratch (csbsr_set_reduction_scratch)");
  }
  hipLaunchKernelGGL((conv_tp_kernel<NKC, R, A, M, S>), dim3(g), dim3(256), SM_BYTES, st, k, zp, zp + 128);
  if (S) {
    if (dbias) csbsr_sum_partials(k.part, (int)g, k.part_ld, k.cout, dbias, st);
    if (dprelu) csbsr_sum_partials(k.part + k.coutp, (int)g, k.part_ld, 1, dprelu, st);
  }
  CSBSR_LAUNCH_CHECK("csbsr_conv_tp_forward");
  return 0;
}

extern "C" int csbsr_conv_tp_forward(const csbsr_conv_desc_t* d, csbsr_stream_t s) {
  CSBSR_CHECK(csbsr_conv_tp_eligible(d), "conv_tp: launch not eligible (see csbsr_conv_tp_eligible)");
  CSBSR_CHECK(d->in[0].ptr && d->wt, "conv_tp: null pointer");
  CSBSR_CHECK(d->act != CSBSR_ACT_PRELU || d->prelu, "conv_tp: PReLU without a slope");
  ConvTpK k;
  k.in = reinterpret_cast<const half_t*>(d->in[0].ptr); k.i_sn = d->in[0].sn; k.i_sy = d->in[0].sy; k.i_sx = d->in[0].sx;
  k.N = d->N; k.H = d->H; k.W = d->W; k.stride = d->stride; k.pad = d->pad;
  k.wt = reinterpret_cast<const half_t*>(d->wt);
  k.cout = d->cout; k.coutp = d->coutp;
  k.out16 = reinterpret_cast<half_t*>(d->out16); k.o_sn = d->o_sn; k.o_sy = d->o_sy; k.o_sx = d->o_sx;
  k.bias = d->bias; k.act = d->act; k.slope = d->act_slope; k.prelu = d->prelu; k.out_scale = d->out_scale;
  k.res_mode = d->res_mode; k.res = reinterpret_cast<const half_t*>(d->res); k.r_sn = d->r_sn; k.r_sy = d->r_sy; k.r_sx = d->r_sx;
  k.accumulate = d->accumulate;
  k.mask = reinterpret_cast<const half_t*>(d->mask); k.m_sn = d->m_sn; k.m_sy = d->m_sy; k.m_sx = d->m_sx; k.mask_slope = d->mask_slope;
  k.mask_prelu = d->mask_prelu; k.part = nullptr; k.part_ld = 0;
  k.dres = reinterpret_cast<half_t*>(d->dres); k.d_sn = d->dr_sn; k.d_sy = d->dr_sy; k.d_sx = d->dr_sx;
  k.tiles_x = (unsigned)((d->W + TP_TW - 1) / TP_TW); k.tiles_y = (unsigned)((d->H + TP_TH - 1) / TP_TH);
  const unsigned nphase = (unsigned)(d->stride * d->stride);
  k.pg = nphase < TP_MAXPG ? nphase : TP_MAXPG;
  while (nphase % k.pg) --k.pg;          // (stride is even: pg is too)
  k.pgroups = nphase / k.pg;
  { const char* e = getenv("CSBSR_TP_DBG"); k.dbg = e ? atoi(e) : 0; }
  int dev = 0;
  CSBSR_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < CSBSR_MAX_DEVICES, "conv_tp: no current device");
  if (!g_tp_zero_page[dev]) {
    CSBSR_CHECK(hipMalloc(reinterpret_cast<void**>(&g_tp_zero_page[dev]), 256 + 1024) == hipSuccess, "conv_tp: zero page alloc failed");
    (void)hipMemset(g_tp_zero_page[dev], 0, 256 + 1024);
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  half_t* zp = g_tp_zero_page[dev];
  const bool r = d->res_mode != CSBSR_RES_NONE, a = d->accumulate != 0, m = d->mask != nullptr;
  // the instantiated epilogues: forward (plain / residual), dgrad (plain / accumulate / mask / accumulate + mask)
  const bool st_ = d->dact_bias || d->dact_prelu;
  g_last_conv_kernel = CONVK_TP | ((r ? 1 : 0) | (a ? 2 : 0) | (m ? 4 : 0) | ((m && (st_ || (r && d->dres))) ? 8 : 0)) << 8;      // bits 8..: <res, acc, mask, sums>
  if (!r && !a && !m) return launch_tp<2, false, false, false, false>(k, st, zp, nullptr, nullptr);
  if (r && !a && !m) return launch_tp<2, true, false, false, false>(k, st, zp, nullptr, nullptr);
  if (!r && a && !m) return launch_tp<2, false, true, false, false>(k, st, zp, nullptr, nullptr);
  if (!r && !a && m) return launch_tp<2, false, false, true, false>(k, st, zp, nullptr, nullptr);
  if (!r && a && m && !st_) return launch_tp<2, false, true, true, false>(k, st, zp, nullptr, nullptr);
  if (!r && a && m && st_) return launch_tp<2, false, true, true, true>(k, st, zp, d->dact_bias, d->dact_prelu);
  if (r && !a && m && d->dres) return launch_tp<2, true, false, true, true>(k, st, zp, d->dact_bias, d->dact_prelu);
  csbsr_set_error("conv_tp: residual together with accumulate / mask is not instantiated");
  return 1;
}
