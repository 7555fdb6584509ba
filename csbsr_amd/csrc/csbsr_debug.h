/* Private measurement / A-B hooks of libcsbsr_hip.so -- NOT part of the drop-in C ABI (include/csbsr_hip.h).  They select between
 * kernels that produce identical results, or report which kernel ran; bench.py, scripts/ and the kernel tests bind them through
 * csbsr_amd._lib.DEBUG_SIGNATURES. */
#ifndef CSBSR_DEBUG_H
#define CSBSR_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* kernel the calling thread's last csbsr_conv_forward dispatched to -- 0/1/2 conv_igemm_kernel<32|64|128,..>, 3/4/7
 * conv_igemm_glds_kernel<128,2,2 | 256,4,3 | 256,4,2,2>, 5 conv_thin_cout_kernel, 6 conv_thin_cin_kernel,
 * 8 conv_hr_kernel, 9 conv_tp_kernel, 10 conv_x3_kernel<3>, 11 conv_thin_tp_kernel, 12 conv_x3_kernel<2>, 13 conv_thin_cin2_kernel,
 * 14 conv_igemm_glds_kernel<128,2,2,0>, 15 conv_thin_sc_kernel, 16 conv_thin_tpd_kernel,
 * 17 / 18 conv_x3_kernel<3,1024> / <2,1024> (the instances with the straight-line epilogue rows) (bench.py's roofline block).
 * Bits 8.. = the template instance within the kernel (one rocprofv3 row each): conv_igemm_glds 1 FS (fused split stage) | 2 GK (general K
 * walk); conv_tp 1 res | 2 acc | 4 mask | 8 sums; conv_hr 1 seven-octet input | 2 1x1 | 4 two cout tiles | 8 mask | 16 stat | 32 class bias; conv_x3<3> 1 = the
 * <3,2048> instance */
int32_t csbsr_debug_last_conv_kernel(void);
/* same for csbsr_conv_wgrad: 0 conv_wgrad_kernel<128,128,2,2>, 1 <128,256,2,4>, 2 <64,128,2,2>, 3 <32,128,1,4>, 4 conv_wgrad_thin_kernel,
 * 5 / 6 / 7 / 9 conv_wgrad_glds_kernel<128,128,..> / <128,256,..> / <256,256,..> / <128,512,..>, 8 conv_wgrad_hr_kernel */
int32_t csbsr_debug_last_wgrad_kernel(void);
/* kernel selection only, results are identical:
 *   wgrad: bit0 hardware transpose reads (0 = scalar LDS transposition), 2 no thin kernel, 4 no XCD tap order, 8 no flat grid,
 *          16 flat grid everywhere, 32 no row shift, 64 no 128x256 tile (register-staged kernel), 128 register-staged kernel everywhere
 *          (implied by bit0 = 0), 256 no 256x256 LDS-DMA tile, 512 no 128x256 LDS-DMA tile, 1024 LDS-DMA kernel for every >= 65-row problem (default:
 *          only where it runs the 256x256 tile), bits 12..19 extra dynamic LDS in KiB,
 *          bit 23 no 128 x 512 four-tap tile for the 8x8 stride-4 layers (the 128 x 256 tap-pair tile instead),
 *          bit 24 no 256 x 256 tile for problems of 512 .. 1023 columns (128 x 128 tiles, as through round 4)
 *   conv:  low 3 bits 0 = register-staged kernel only, 1 = 128x128 LDS-DMA tile only, 2 = default, 3 = 256x128 wherever it fits;
 *          16 no thin kernels, 32 phases on grid.z, 64 linear pixel tiles, 128 raster tap order, 256 no 256-cout tile */
void csbsr_debug_set_wgrad_tr(int flags);
void csbsr_debug_set_conv_glds(int mode);
/* phase-decomposed transposed-conv kernel (csrc/conv_tp.hip): 0 never eligible, 1 problems of >= 128 tiles (default), 2 every eligible launch */
void csbsr_debug_set_conv_tp(int mode);
/* wide 3x3 kernel (csrc/conv_x3.hip): 0 never eligible, 1 launches of >= 512 tile x cout-tile items (default), 2 every eligible launch */
void csbsr_debug_set_conv_x3(int mode);
/* Winograd F(2,3)-along-x 3x3 kernel (csrc/conv_x3w.hip): low 3 bits 0 never eligible (default), 1 launches that fill the chip, 2 every eligible
 * launch; bits 3..: smallest padded input-channel count taken in mode 1, in units of 32 (0 = keep the default 128);
 * bits 8..: ablation variant (CSBSR_X3W_ABLATE builds only) */
void csbsr_debug_set_conv_x3w(int mode);
/* narrow-output 3x3 kernel (csrc/conv_x3n.hip): 0 never eligible, 1 launches of >= 512 pixel tiles with >= 256 input channels (default), 2 every eligible launch */
void csbsr_debug_set_conv_x3n(int mode);
/* full-resolution thin 3x3 weight-gradient kernel (csrc/conv_wgrad_hr.hip): 0 never, 1 launches of >= 1024 tiles (default), 2 every eligible launch */
void csbsr_debug_set_wgrad_hr(int mode);
/* CU-partitioned streams (csrc/streams.hip) -- a MEASUREMENT hook, not product: round 5 measured a weight gradient and an HBM-bound link of
 * the dgrad chain on disjoint CU sets against the same launches back to back (scripts/overlap_pair.py, profiles/r05_overlap.json): no
 * split is faster than serial by more than 5 %, most are slower, so the engine keeps ONE stream.  mask: bit i = one CU; bit i lands on
 * XCD i % 8 on gfx950 (a prefix of the bit array spreads over all eight XCDs); nwords 32-bit words.  A stream created here remembers
 * its CU count as its budget; the persistent-grid kernels (conv_x3, conv_tp, conv_hr) size their grids from the budget of the stream
 * they are launched on (none: the device's CU count).  set_cu_budget attaches a budget to any stream, 0 removes it. */
int csbsr_debug_stream_create_cu_mask(void** out, const uint32_t* mask, int32_t nwords);
int csbsr_debug_stream_destroy(void* stream);
int csbsr_debug_stream_set_cu_budget(void* stream, int32_t ncu);
int32_t csbsr_debug_stream_cu_budget(void* stream);
/* where a grid's workgroups ran: out[2 wg] = HW_REG_HW_ID, out[2 wg + 1] = HW_REG_XCC_ID of workgroup wg, each spinning spin_cycles
 * so that the grid spreads over every CU its stream may use (scripts/overlap_pair.py: the CU-mask bit -> XCD map) */
int csbsr_debug_cu_trace(uint32_t* out, int32_t nwg, int32_t spin_cycles, void* stream);
#ifdef __cplusplus
}
#endif
#endif
