// Microbenchmark: how many bytes per clock one CU can pull into LDS (global_load_lds_dwordx4) or into registers
// (global_load_dwordx4) from an L2- / MALL- / HBM-resident footprint, as a function of the bytes in flight per workgroup and of the
// workgroups per CU.  Answers whether the ~16 B/clk/CU the conv kernels see is a bandwidth limit or in-flight-requests x latency.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_ldsdma.hip -o scripts/ubench_ldsdma && scripts/ubench_ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one "stage" = NI instructions per wave, each 64 lanes x 16 B = 1 KB, lanes contiguous
template <int MODE, int NI, int DEPTH>
__global__ __launch_bounds__(256) void bw_kernel(const char* __restrict__ src, size_t footprint, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const size_t stage_bytes = (size_t)NI * 4 * 1024;                // per workgroup
  const size_t nst = footprint / stage_bytes;
  size_t s = ((size_t)blockIdx.x * 2654435761u) % nst;
  float4 acc = {0, 0, 0, 0};
  auto issue = [&](size_t st, int slot) {
    const char* base = src + st * stage_bytes + (size_t)wid * NI * 1024 + lane * 16;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + j * 1024),
                                         (__attribute__((address_space(3))) void*)(smem + ((slot * 4 + wid) * NI + j) * 1024), 16, 0, 0);
      } else {
        const float4 v = *reinterpret_cast<const float4*>(base + j * 1024);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
  };
  if (MODE == 0) {
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) { issue(s, d); s = (s + 977) % nst; }
    int slot = DEPTH - 1;
    for (int it = 0; it < iters; ++it) {
      issue(s, slot); s = (s + 977) % nst;
      slot = slot + 1 == DEPTH ? 0 : slot + 1;
      // wait for the oldest stage: (DEPTH - 1) * NI instructions may stay in flight
      if constexpr ((DEPTH - 1) * NI == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NI) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && iters < 0) out[0] = smem[0];
  } else {
    for (int it = 0; it < iters + DEPTH - 1; ++it) { issue(s, 0); s = (s + 977) % nst; }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
  }
}

template <int MODE, int NI, int DEPTH>
static void run(const char* name, const char* src, size_t footprint, int wg_per_cu, float* out, int clk_khz) {
  const int iters = 2000;
  const size_t lds = MODE == 0 ? (size_t)DEPTH * NI * 4 * 1024 : 0;
  if (lds > 160 * 1024 / wg_per_cu) return;
  CK(hipFuncSetAttribute((const void*)bw_kernel<MODE, NI, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * wg_per_cu;
  hipLaunchKernelGGL((bw_kernel<MODE, NI, DEPTH>), dim3(grid), dim3(256), lds, 0, src, footprint, 50, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((bw_kernel<MODE, NI, DEPTH>), dim3(grid), dim3(256), lds, 0, src, footprint, iters, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)grid * (iters + DEPTH - 1) * NI * 4 * 1024;
  const double tbs = bytes / (ms * 1e-3) / 1e12;
  printf("%-5s footprint %7.1f MB  stage %3d KB x depth %d  wg/cu %d  in-flight/CU %4d KB : %6.2f TB/s = %5.1f B/clk/CU\n", name,
         footprint / 1048576.0, NI * 4, DEPTH, wg_per_cu, (int)((DEPTH - (MODE == 0 ? 1 : 0)) * NI * 4 * wg_per_cu), tbs,
         bytes / (ms * 1e-3) / 256.0 / (clk_khz * 1e3));
}

int main() {
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  printf("%s: %d CUs, %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
  const size_t maxfp = (size_t)4 << 30;
  char* src; float* out;
  CK(hipMalloc(&src, maxfp)); CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 1, maxfp));
  for (size_t fp : {(size_t)2 << 20, (size_t)24 << 20, (size_t)128 << 20, maxfp}) {
    for (int w : {1, 2, 4}) {
      run<0, 2, 2>("dma", src, fp, w, out, pr.clockRate);
      run<0, 4, 2>("dma", src, fp, w, out, pr.clockRate);
      run<0, 8, 2>("dma", src, fp, w, out, pr.clockRate);
      run<0, 8, 3>("dma", src, fp, w, out, pr.clockRate);
      run<0, 8, 4>("dma", src, fp, w, out, pr.clockRate);
      run<0, 16, 2>("dma", src, fp, w, out, pr.clockRate);
      run<1, 8, 2>("regs", src, fp, w, out, pr.clockRate);
    }
  }
  return 0;
}
