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
    case CSBSR_ACT_RELU: return v > 0.f ? v : 0.f;
    case CSBSR_ACT_LRELU:
    case CSBSR_ACT_PRELU: return v > 0.f ? v : v * slope;
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

// Caller-owned scratch for two-stage per-channel reductions (csbsr_set_reduction_scratch).  Device-scope fp32 atomics are
// resolved at the memory side on this part (~190 ns each, serialised per address: 1500 workgroups x 128 channels of bias-gradient
// atomics cost 290 us on top of a 36 us streaming pass), so the reducing kernels write per-workgroup partial rows here and
// csbsr_sum_partials folds them; without a registered scratch they fall back to atomics.
#define CSBSR_MAX_DEVICES 16
float* csbsr_red_scratch(long need_elems);   // the current device's registered scratch if it holds need_elems floats, else nullptr
int csbsr_sum_partials(const float* part, int nblk, long ld, int count, float* dst, hipStream_t st);   // dst[j] += sum_b part[b*ld+j]

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
