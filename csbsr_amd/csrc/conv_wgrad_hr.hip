// Weight gradient of the FULL-RESOLUTION 3x3 layers with few channels (the kernel predictor's fe_SR / fe_kernel / fe_cat chains, 32 / 49
// channels at 1792^2, kbpn.py:521-578; the 64-channel decoder head): 24 launches and 4 % of the training step at 240-360 TFLOP/s =
// 1.7 TB/s in the tiled GEMM kernels, whose 32 x 128 tiles stage 26 flop per LDS byte and gather the input once per tap.  These layers
// are HBM-bound by a wide margin (237 GFLOP over 1.6 GB of operands at N = 4), so the kernel is built around reading both operands ONCE:
//
//  * a persistent workgroup walks 8 x 32 pixel tiles; per tile the dPre tile (256 pixels x ca) and the input's (8+2) x (32+2) halo
//    (x cb) go HBM -> LDS with buffer_load ... lds, one piece at a time between the MFMAs of the previous tile (double buffered; the
//    per-lane offsets of every piece are kernel constants, the tile enters through the descriptor base, out-of-image pixels read as
//    zeros through the hardware bounds check);
//  * all nine taps are fed from that one halo tile: D[a][tap][b] += A[pix][a] * B[pix + tap][b], both operands through the
//    transposing LDS read (ds_read_b64_tr_b16), the dPre fragment of 16 pixels shared by the nine taps;
//  * a wave owns one (32 a-channels x 32 b-channels) block pair for all nine taps (144 accumulator registers) and, when there are
//    fewer than four block pairs, a share of the tile's rows; the partial sums of the row shares meet in LDS once, at the end;
//  * every workgroup writes one fp32 slab G[a][tap][b] of the standard layout (csbsr_unpack_wgrad sums them): splits = workgroups.
#include "common.h"
#include "conv_wgrad.h"
#include "csbsr_debug.h"

#define WH_TW 32
#define WH_HW (WH_TW + 2)

struct WgradHrK {
  const half_t* a; long a_sn, a_sy, a_sx;
  const half_t* b; long b_sn, b_sy, b_sx;
  int N, H, W;
  int ca, cb;                  // padded channels the slab rows / columns run over (<= 8 * CA8 / CB8)
  float* g; long slab_stride;
  unsigned tiles_x, tiles_y;
};

// One LDS-DMA piece: 64 lanes x 16 bytes from buffer offset voff (out of range: zeros) to LDS bytes [lds_addr, lds_addr + 1024).
// Written as inline assembly on purpose: after the compiler's own buffer_load ... lds builtin, hipcc puts an s_waitcnt vmcnt(0) in front
// of every transposing LDS read (it cannot tell the ring slots apart), which serialises each piece's HBM round trip with the MFMAs --
// measured 10.7k cycles per tile instead of ~3k.  The completion of the pieces is counted by hand (vmcnt + barrier per tile).
typedef int wh_v4i __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void wh_dma16(wh_v4i rs, int voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr));
}
// buffer descriptor over [base, base + 2 GB), built from provably wave-uniform halves (offsets >= 0x7fffffff read as zeros)
static __device__ __forceinline__ wh_v4i wh_make_rs(const half_t* base) {
  const unsigned long a = reinterpret_cast<unsigned long>(base);
  wh_v4i r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r[2] = 0x7fffffff;
  r[3] = 0x00020000;
  return r;
}

constexpr int wh_pieces(int pix, int c8) { return ((pix * c8 + 63) / 64 + 3) / 4; }      // DMA pieces per wave for a [pix][c8 octets] tile
constexpr int wh_buf_bytes(int CA8, int CB8, int TH) { return (wh_pieces(TH * WH_TW, CA8) + wh_pieces((TH + 2) * WH_HW, CB8)) * 4096; }
constexpr int wh_nparts(int CA8, int CB8) { return 4 / (((CA8 + 3) / 4) * ((CB8 + 3) / 4)); }
constexpr int wh_smem(int CA8, int CB8, int TH, int NBUF) {
  const int ring = NBUF * wh_buf_bytes(CA8, CB8, TH) + 256, red = (wh_nparts(CA8, CB8) - 1) * (4 / wh_nparts(CA8, CB8)) * 9 * 16 * 64 * 4;
  return ring > red ? ring : red;
}

// TH: tile rows (8 or 4); NBUF: ring depth -- the tile NBUF - 1 ahead is in flight while a tile is multiplied: one tile takes ~1.2k MFMA
// cycles, an HBM round trip several thousand, and ~80 KB per CU must be under way to stream at the HBM rate
template <int CA8, int CB8, int WH_TH, int NBUF>
__global__ __launch_bounds__(256) void conv_wgrad_hr_kernel(const WgradHrK p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WH_HH = WH_TH + 2, WH_NPIX = WH_HH * WH_HW;
  constexpr int NA = (CA8 + 3) / 4, NB = (CB8 + 3) / 4, NBP = NA * NB;      // 32-channel blocks, block pairs
  constexpr int NPARTS = 4 / NBP, ROWS = WH_TH / NPARTS;                    // row shares, rows per wave
  constexpr int PA = CA8 * 16, PB = CB8 * 16;                               // LDS bytes per pixel
  constexpr int NPA = wh_pieces(WH_TH * WH_TW, CA8), NPB = wh_pieces(WH_NPIX, CB8), NPW = NPA + NPB;
  constexpr int ABYTES = NPA * 4096, BUF = wh_buf_bytes(CA8, CB8, WH_TH);
  constexpr int G = ROWS * 2, PPG = (NPW + G - 1) / G;                      // (row, k-half) groups per tile, DMA pieces per group
  static_assert(NBP == 1 || NBP == 2 || NBP == 4, "one, two or four block pairs");
  static_assert(PPG <= 9, "at most one piece per tap");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bp = wid % NBP, part = wid / NBP, ab = bp / NB, bb = bp % NB;
  const unsigned per_img = p.tiles_x * p.tiles_y, ntiles = per_img * (unsigned)p.N;

  // ---- DMA roles: piece i of a wave is wave instruction wid + 4 i; its lane fills 16-byte slot g = (wid + 4 i) * 64 + lane
  int voff[NPW], py0[NPW], px0[NPW];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int g = (wid + 4 * i) * 64 + lane, px = g / CA8, c = g - px * CA8;
    const int ty = px / WH_TW, tx = px - ty * WH_TW;
    voff[i] = 2 * (int)(ty * p.a_sy + tx * p.a_sx + c * 8);
    py0[i] = px < WH_TH * WH_TW ? ty : 0x40000000;
    px0[i] = tx;
  }
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int g = (wid + 4 * i) * 64 + lane, hp = g / CB8, c = g - hp * CB8;
    const int ty = hp / WH_HW, tx = hp - ty * WH_HW;
    voff[NPA + i] = 2 * (int)(ty * p.b_sy + tx * p.b_sx + c * 8);
    py0[NPA + i] = hp < WH_NPIX ? ty : 0x40000000;
    px0[NPA + i] = tx;
  }
  auto tile_pos = [&](unsigned t, int& n, int& Y0, int& X0) {
    n = t / per_img;
    const unsigned r = t - n * per_img;
    Y0 = (r / p.tiles_x) * WH_TH; X0 = (r % p.tiles_x) * WH_TW;
  };
  // piece j of tile (n, Y0, X0) into ring buffer `buf`: j < NPA a dPre piece (pixels past the image edge read as zeros), else a halo piece
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem) + (unsigned)wid * 1024u;
  auto issue_piece = [&](wh_v4i rsa, wh_v4i rsb, int Y0, int X0, int j, int buf) __attribute__((always_inline)) {
    if (j < NPA) {
      const bool ok = (unsigned)(py0[j] + Y0) < (unsigned)p.H && (unsigned)(px0[j] + X0) < (unsigned)p.W;
      wh_dma16(rsa, ok ? voff[j] : -1, lds0 + (unsigned)(buf * BUF + 4 * j * 1024));
    } else {
      const bool ok = (unsigned)(py0[j] + Y0 - 1) < (unsigned)p.H && (unsigned)(px0[j] + X0 - 1) < (unsigned)p.W;
      wh_dma16(rsb, ok ? voff[j] : -1, lds0 + (unsigned)(buf * BUF + ABYTES + 4 * (j - NPA) * 1024));
    }
  };
  auto src_a = [&](int n, int Y0, int X0) { return p.a + n * p.a_sn + (long)Y0 * p.a_sy + (long)X0 * p.a_sx; };
  auto src_b = [&](int n, int Y0, int X0) { return p.b + n * p.b_sn + (long)(Y0 - 1) * p.b_sy + (long)(X0 - 1) * p.b_sx; };

  // ---- fragment addressing (transposing reads): 16-lane group = 4 pixel rows x 16 channels, lane i of the group supplies the address
  // of row i / 4, channel quad i % 4; afterwards the lane holds channel (16 (group & 1) + i) of four pixels; lanes 32.. take pixels 8..15
  const int li = lane & 15;
  const int rsub = (lane >> 5) * 8 + (li >> 2);
  const int unit = (lane >> 4) & 1;
  const int aoff = rsub * PA + (ab * 2 + unit) * 32 + (li & 3) * 8 + (part * ROWS) * WH_TW * PA;
  const int boff = ABYTES + rsub * PB + (bb * 2 + unit) * 32 + (li & 3) * 8 + (part * ROWS) * WH_HW * PB;
  auto tr8 = [&](const char* q, int rowbytes) __attribute__((always_inline)) {
    const fp16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(q));
    const fp16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(q + 4 * rowbytes));
    h8 v;
    v[0] = r0[0]; v[1] = r0[1]; v[2] = r0[2]; v[3] = r0[3]; v[4] = r1[0]; v[5] = r1[1]; v[6] = r1[2]; v[7] = r1[3];
    return v;
  };

  f16v acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  unsigned it = blockIdx.x;
  if (it < ntiles) {
    // tile k of this workgroup is tile it + k * gridDim.x; past the last one the pieces refetch the last tile (uniform instruction counts)
    auto tile_at = [&](unsigned t, int& n, int& Y0, int& X0) {
      if (t >= ntiles) t -= ((t - ntiles) / gridDim.x + 1) * gridDim.x;
      tile_pos(t, n, Y0, X0);
    };
#pragma unroll
    for (int k = 0; k < NBUF - 1; ++k) {
      int n, Y0, X0;
      tile_at(it + k * gridDim.x, n, Y0, X0);
      const wh_v4i rsa = wh_make_rs(src_a(n, Y0, X0)), rsb = wh_make_rs(src_b(n, Y0, X0));
#pragma unroll
      for (int j = 0; j < NPW; ++j) issue_piece(rsa, rsb, Y0, X0, j, k);
    }
    int buf = 0;
    for (; it < ntiles; it += gridDim.x) {
      // this tile has landed everywhere (the NBUF - 2 tiles after it may still be in flight) and every wave is done with the buffer
      // of the previous tile: the tile NBUF - 1 ahead goes there, piece by piece, below
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NPW) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int nn, Y0n, X0n;
      tile_at(it + (NBUF - 1) * gridDim.x, nn, Y0n, X0n);
      const wh_v4i rsa = wh_make_rs(src_a(nn, Y0n, X0n)), rsb = wh_make_rs(src_b(nn, Y0n, X0n));
      const int bfill = buf == 0 ? NBUF - 1 : buf - 1;
      const char* sa = smem + buf * BUF + aoff;
      const char* sb = smem + buf * BUF + boff;
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const h8 af = tr8(sa + (r * WH_TW + h * 16) * PA, PA);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const h8 bf = tr8(sb + ((r + ky) * WH_HW + h * 16 + kx) * PB, PB);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[tap], 0, 0, 0);
            const int j = (r * 2 + h) * PPG + tap;
            if (tap < PPG && j < NPW) issue_piece(rsa, rsb, Y0n, X0n, j, bfill);
          }
        }
      buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
  }

  // ---- the row shares of a block pair meet in LDS; share 0 adds them up and writes the slab: every (a < ca, tap, b < cb) exactly once
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last refetch must not land on the sums)
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  if (NPARTS > 1 && part > 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(((part - 1) * NBP + bp) * 144 + t * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (part == 0) {
    float* slab = p.g + (size_t)blockIdx.x * p.slab_stride;
    const int ktot = 9 * p.cb;
    const int b = bb * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[t][r];
#pragma unroll
        for (int q = 1; q < NPARTS; ++q) v += red[(((q - 1) * NBP + bp) * 144 + t * 16 + r) * 64 + lane];
        const int row = ab * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.ca && b < p.cb) slab[(size_t)row * ktot + t * p.cb + b] = v;
      }
  }
#endif
}

static int g_wgrad_hr = 1;      // 0 off, 1 launches of >= 1024 tiles (default), 2 every eligible launch (tests)
extern "C" void csbsr_debug_set_wgrad_hr(int mode) { g_wgrad_hr = mode; }

// tile rows: 8 for 32 + 32 channels, 4 otherwise (the ring of tiles must fit the 160 KB of LDS three or four deep)
static int wh_th(const csbsr_wgrad_desc_t* d) { return (d->ca == 32 && d->b[0].c == 32) ? 8 : 4; }
static long wh_tiles(const csbsr_wgrad_desc_t* d) {
  const int th = wh_th(d);
  return (long)d->N * ((d->AH + th - 1) / th) * ((d->AW + WH_TW - 1) / WH_TW);
}

// 3x3, stride 1, pad 1, ONE gathered segment, both sides 8..64 padded channels in the instantiated octet counts, same spatial size
bool wgrad_hr_eligible(const csbsr_wgrad_desc_t* d) {
  if (!g_wgrad_hr || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1 || d->b[1].c != 0) return false;
  if (d->AH != d->BH || d->AW != d->BW || d->b[0].sx == 0) return false;
  const int ca8 = d->ca / 8, cb8 = d->b[0].c / 8;
  const bool combo = (ca8 == 4 && cb8 == 4) || (ca8 == 4 && cb8 == 7) || (ca8 == 7 && cb8 == 4) || (ca8 == 7 && cb8 == 7) || (ca8 == 8 && cb8 == 8);
  if (!combo) return false;
  if (d->a_sy >= (1l << 31) / 32 || d->b[0].sy >= (1l << 31) / 32) return false;
  if (g_wgrad_hr == 1 && wh_tiles(d) < 2048) return false;
  return true;
}

int32_t wgrad_hr_splits(const csbsr_wgrad_desc_t* d) {
  const long tiles = wh_tiles(d);
  return (int32_t)(tiles < 256 ? tiles : 256);
}

template <int CA8, int CB8, int TH, int NBUF>
static int launch_wgrad_hr(const WgradHrK& k, int splits, hipStream_t st) {
  constexpr int SM_BYTES = wh_smem(CA8, CB8, TH, NBUF);
  static_assert(SM_BYTES <= 160 * 1024, "LDS budget");
  static LdsAttrOnce attr;
  if (int e = csbsr_lds_attr(attr, reinterpret_cast<const void*>(conv_wgrad_hr_kernel<CA8, CB8, TH, NBUF>), SM_BYTES, "wgrad(hr)")) return e;
  hipLaunchKernelGGL((conv_wgrad_hr_kernel<CA8, CB8, TH, NBUF>), dim3(splits), dim3(256), SM_BYTES, st, k);
  CSBSR_LAUNCH_CHECK("csbsr_conv_wgrad(hr)");
  return 0;
}

int wgrad_hr_launch(const csbsr_wgrad_desc_t* d, hipStream_t st) {
  if (d->splits != wgrad_hr_splits(d)) {
    csbsr_set_error("wgrad(hr): splits=%d, expected %d; use csbsr_wgrad_splits_desc()", d->splits, wgrad_hr_splits(d));
    return 1;
  }
  WgradHrK k;
  k.a = reinterpret_cast<const half_t*>(d->a); k.a_sn = d->a_sn; k.a_sy = d->a_sy; k.a_sx = d->a_sx;
  k.b = reinterpret_cast<const half_t*>(d->b[0].ptr); k.b_sn = d->b[0].sn; k.b_sy = d->b[0].sy; k.b_sx = d->b[0].sx;
  k.N = d->N; k.H = d->AH; k.W = d->AW; k.ca = d->ca; k.cb = d->b[0].c;
  k.g = d->g; k.slab_stride = (long)d->ca * 9 * d->b[0].c;
  const int th = wh_th(d);
  k.tiles_x = (unsigned)((d->AW + WH_TW - 1) / WH_TW); k.tiles_y = (unsigned)((d->AH + th - 1) / th);
  const int ca8 = d->ca / 8, cb8 = d->b[0].c / 8;
  if (ca8 == 4 && cb8 == 4) return launch_wgrad_hr<4, 4, 8, 3>(k, d->splits, st);
  if (ca8 == 4 && cb8 == 7) return launch_wgrad_hr<4, 7, 4, 4>(k, d->splits, st);
  if (ca8 == 7 && cb8 == 4) return launch_wgrad_hr<7, 4, 4, 4>(k, d->splits, st);
  if (ca8 == 7 && cb8 == 7) return launch_wgrad_hr<7, 7, 4, 3>(k, d->splits, st);
  return launch_wgrad_hr<8, 8, 4, 3>(k, d->splits, st);
}
