// Kernel-argument block shared by the two weight-gradient kernels (conv_wgrad.hip: register-staged, any channel count;
// conv_wgrad_glds.hip: LDS-DMA, >= 65 A-channels).
#pragma once
#include "common.h"

#define WG_BP 64     // pixels per reduction step
#define WG_BN 128    // columns (tap,channel) per tile
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct WgradK {
  const half_t* a; long a_sn, a_sy, a_sx; int ca;
  csbsr_seg_t b[2]; int cb0, cbtot;
  int N, AH, AW, BH, BW;
  int KH, KW, stride, pad, dil;
  float* g; int ktot;        // row length of G
  long M;                    // N*AH*AW
  long per_split;            // pixels per split (multiple of 32)
  unsigned tiles_a, tiles_b;
  int ca_real;               // real channels of A (thin kernel: rows = (tap, channel))
  int tap_perm;              // 1: XCD-aware tap order of the 8x8 stride-4 layers (see the kernel)
  int flat;                  // 1: 1-D grid over (split, tile): all tiles of one pixel split run on ONE XCD (see the kernel)
  int splits;
  int row_shift;             // 1: per-tile pixel-range shift that aligns the gathered rows of taps a stride apart (see the kernel)
  // a launch may cover only rows [row0, row0 + ca) of the problem's G (the 256-row LDS-DMA tiles take the multiple-of-256 part of
  // the A channels, a second launch the rest): ``a`` already points at channel row0, slabs are slab_stride elements apart
  long slab_stride;          // elements per pixel-split slab = (all A channels) * ktot
  int row0;
  int tw_log, gx, gy;        // LDS-DMA kernel, 2-D stages: log2 of the rectangle width, rectangles per row / per image column
};


// conv_wgrad_glds.hip
bool wgrad_glds_eligible(const WgradK& k);
int wgrad_glds_tile_n(const WgradK& k);                  // 128 or 256 columns per tile (the caller sizes the pixel splits with it)
int wgrad_glds_tile_a(const WgradK& k);                  // 256: rows [0, ca / 256 * 256) run on 256 x 256 tiles, the rest on 128-row tiles
int wgrad_glds_launch(const WgradK& k, int ta, int tn, int splits, hipStream_t st);
extern int g_wgrad_glds;                                 // 0: off (A/B timing, csbsr_debug_set_wgrad_tr bit 7)

// conv_wgrad_hr.hip: full-resolution 3x3 layers with <= 64 channels on both sides (one slab per workgroup)
bool wgrad_hr_eligible(const csbsr_wgrad_desc_t* d);
int32_t wgrad_hr_splits(const csbsr_wgrad_desc_t* d);
int wgrad_hr_launch(const csbsr_wgrad_desc_t* d, hipStream_t st);
