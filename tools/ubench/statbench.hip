// Cycles per float4 of the GroupNorm statistics accumulation of the conv epilogue (gemm2.hip), one wave per SIMD:
//   0: fp64 as shipped in round 2 (cvt, add, cvt*cvt+add per value)   1: fp32 plain   2: double-float (TwoSum + exact product) in fp32
//   3: fp64 with the square formed in fp32 pairs?  (no) -> 3: fp64 sum of v and fp64 fma(v, v, acc) with ONE cvt per value
// build: hipcc --offload-arch=gfx950 -O3 -o statbench statbench.hip ; run: ./statbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ in, double* __restrict__ out, long long* __restrict__ cyc, int iters) {
  const int tid = threadIdx.x;
  const float4* src = in + tid;
  double gs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float fs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float hi[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lo[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  __shared__ float4 lds[256 * 8];
  for (int i = 0; i < 8; ++i) lds[i * 256 + tid] = src[i * 256];
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; it += 4) {
    float4 a4s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a4s[u] = lds[((it + u) & 7) * 256 + tid];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float v[4] = {a4s[u].x, a4s[u].y, a4s[u].z, a4s[u].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (MODE == 0) {
          gs[q] += (double)v[q];
          gs[4 + q] += (double)v[q] * (double)v[q];
        } else if (MODE == 1) {
          fs[q] += v[q];
          fs[4 + q] = fmaf(v[q], v[q], fs[4 + q]);
        } else if (MODE == 2) {
          {  // (hi, lo) += v   (TwoSum)
            const float s = hi[q] + v[q];
            const float bb = s - hi[q];
            const float e = (hi[q] - (s - bb)) + (v[q] - bb);
            hi[q] = s;
            lo[q] += e;
          }
          {  // (hi, lo) += v * v   (exact product p + e2, TwoSum of p)
            const float p = v[q] * v[q];
            const float e2 = fmaf(v[q], v[q], -p);
            const float s = hi[4 + q] + p;
            const float bb = s - hi[4 + q];
            const float e = (hi[4 + q] - (s - bb)) + (p - bb);
            hi[4 + q] = s;
            lo[4 + q] += e + e2;
          }
        } else if (MODE == 3) {
          const double d = (double)v[q];
          gs[q] += d;
          gs[4 + q] = fma(d, d, gs[4 + q]);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double r = 0;
  for (int q = 0; q < 8; ++q) r += gs[q] + (double)fs[q] + (double)hi[q] + (double)lo[q];
  out[blockIdx.x * 256 + tid] = r;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = (long long)(t1 - t0);
}

int main() {
  float4* in; double* out; long long* cyc;
  hipMalloc(&in, 256 * 8 * 16); hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 8);
  hipMemset(in, 0x3c, 256 * 8 * 16);
  const int iters = 4096;
  long long h;
#define RUN(M) \
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<M>, dim3(256), dim3(256), 0, 0, in, out, cyc, iters); \
  hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
  printf("mode %d: %.1f s_memtime ticks (100 MHz -> x ~21 shader cycles at 2.1 GHz) per float4 row\n", M, (double)h / iters);
  RUN(0) RUN(1) RUN(2) RUN(3)
  return 0;
}
