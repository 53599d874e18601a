// Micro-benchmark (tools only): L2 -> CU load throughput per CU for three paths, all CUs busy, L2-resident data.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction)
//   mode 1: global_load_dwordx4 -> VGPR (+ ds_write_b128 when W=1)
//   mode 2: global_load_lds_dword (LDS-DMA, 256 B per wave-instruction)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/ldbench.hip -o tools/ubench/ldbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int ROWB>   // ROWB: contiguous bytes per row touched by one instruction (64 or 128) -- rows are 4608 B apart
__global__ __launch_bounds__(256) void ld_kernel(const char* __restrict__ src, float* __restrict__ sink, int iters, long long span) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = ROWB / 16;                     // lanes per row
  const int row = lane / LPR, ch = lane % LPR;
  // each workgroup walks its own 128-row band; rows 4608 B apart (K = 1152 fp32), advancing ROWB per instruction
  const char* base = src + ((long long)(blockIdx.x % 4) * 128 + wave * 32 + row) * 4608 + ch * 16;
  float4 acc = make_float4(0, 0, 0, 0);
  char* dst = lds + wave * 8192;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const char* g = base + ((it * 8 + u) % (4608 / ROWB)) * ROWB;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, 0, 0);
      } else if (MODE == 2) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(dst + u * 256), 4, 0, 0);
      } else {
        const float4 v = *reinterpret_cast<const float4*>(g);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    if (MODE != 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int MODE, int ROWB>
static void run(const char* name, const char* src, float* sink, int wgs_per_cu) {
  const int iters = 400;
  auto k = ld_kernel<MODE, ROWB>;
  const int lds = 65536 / wgs_per_cu >= 32768 ? 32768 : 32768;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(256 * wgs_per_cu), dim3(256), lds, 0, src, sink, iters, 0LL);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double bytes_per_instr = (MODE == 2) ? 256.0 : 1024.0;
  const double total = 256.0 * wgs_per_cu * 4 * iters * 8 * bytes_per_instr;
  printf("%-44s wgs/cu=%d  %8.1f us  %6.2f TB/s  %5.1f B/clk/CU @2.0GHz  %5.1f clk per wave-instr per CU\n", name, wgs_per_cu, ms * 1e3,
         total / (ms * 1e-3) / 1e12, total / (ms * 1e-3) / 256 / 2.0e9, (ms * 1e-3) * 2.0e9 / (wgs_per_cu * 4.0 * iters * 8));
}

int main() {
  char* src; float* sink;
  const size_t bytes = (size_t)1024 * 128 * 4608 + 65536;
  CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMalloc(&sink, 64));
  for (int w = 1; w <= 2; ++w) {
    run<0, 128>("LDS-DMA dwordx4, 128-B rows", src, sink, w);
    run<0, 64>("LDS-DMA dwordx4, 64-B rows", src, sink, w);
    run<1, 128>("global_load_dwordx4 -> VGPR, 128-B rows", src, sink, w);
    run<1, 64>("global_load_dwordx4 -> VGPR, 64-B rows", src, sink, w);
    run<2, 128>("LDS-DMA dword, 128-B rows", src, sink, w);
  }
  return 0;
}
