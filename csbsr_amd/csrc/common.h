// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16 f16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/csbsr_hip.h"

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

void csbsr_set_error(const char* fmt, ...);
#define CSBSR_CHECK(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      csbsr_set_error(__VA_ARGS__);       \
      return 1;                           \
    }                                     \
  } while (0)
#define CSBSR_LAUNCH_CHECK(name)                                             \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      csbsr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return 2;                                                              \
    }                                                                        \
  } while (0)

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case CSBSR_ACT_RELU: return (v > 0.f || v != v) ? v : 0.f;      // NaN-propagating like the straight-line rows (an overflowed accumulator must reach the optimiser's overflow check)
    case CSBSR_ACT_LRELU:
    case CSBSR_ACT_PRELU: return v > 0.f ? v : v * slope;          // (NaN * slope = NaN)
    case CSBSR_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

// bijective XCD-aware remap of a linear block id: XCD x (= id % 8 as dispatched) owns a contiguous chunk of
// the logical tile order, so neighbouring tiles share one L2 (cdna guide T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, k = bid >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Caller-owned scratch for two-stage reductions (csbsr_set_reduction_scratch).  EVERY floating-point reduction of the library is
// order-fixed: a reducing kernel writes per-workgroup (or per-tile) partial rows here and csbsr_sum_partials* folds them in a fixed
// tree, so two runs on the same input are bit-identical (no fp32 atomics anywhere on the path; besides, device-scope fp32 atomics
// are resolved at the memory side on this part, ~190 ns each and serialised per address: 1500 workgroups x 128 channels of
// bias-gradient atomics cost 290 us on top of a 36 us streaming pass).  No registered scratch = error, never an atomic fallback.
#define CSBSR_MAX_DEVICES 16
#define CSBSR_RED_TAIL (4l << 20)            // floats at the end of the scratch reserved for the second level of csbsr_sum_partials*
float* csbsr_red_scratch(long need_elems);   // the current device's registered scratch if it holds need_elems floats (+ the tail), else nullptr (error set)
// dst[b * dst_bs + j] += sum_{r < rows} part[(b * rows + r) * ld + j],  j < count, b < batch -- fixed summation order
int csbsr_sum_partials_batched(const float* part, int rows, long ld, int count, float* dst, int batch, long dst_bs, hipStream_t st);
static inline int csbsr_sum_partials(const float* part, int nblk, long ld, int count, float* dst, hipStream_t st) {
  return csbsr_sum_partials_batched(part, nblk, ld, count, dst, 1, 0, st);
}
#define CSBSR_NEED_SCRATCH(ptr, what) CSBSR_CHECK((ptr) != nullptr, "%s: reduction scratch missing or too small (csbsr_set_reduction_scratch)", what)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: every launcher keeps one of these per kernel
// instance (a process-wide bool would leave the second GPU of a process with the 64 KB default and a failed launch)
struct LdsAttrOnce { bool done[CSBSR_MAX_DEVICES] = {}; };
static inline int csbsr_lds_attr(LdsAttrOnce& o, const void* fn, int bytes, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CSBSR_MAX_DEVICES) { csbsr_set_error("%s: no current device", what); return 2; }
  if (o.done[dev]) return 0;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    csbsr_set_error("%s: cannot reserve %d bytes of LDS", what, bytes);
    return 2;
  }
  o.done[dev] = true;
  return 0;
}

// CUs a persistent-grid kernel launched on ``st`` may count on (csrc/streams.hip): the stream's CU-mask budget, else the device's CU count
int csbsr_cu_budget(hipStream_t st);

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
