// Which bf16 MFMA shape costs the least energy per flop on random operands?  One wave per SIMD, 16 independent accumulator tiles,
// operands held in registers (no memory traffic in the loop): v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip ; run: ./mfma_power <mode 0|1> <seconds> [zeros]
//   (power and clock are sampled from outside: tools/ubench/mfma_power.sh)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    uint4 x = in[(blockIdx.x * 256 + tid) * 8 + i], y = in[(blockIdx.x * 256 + tid) * 8 + 4 + i];
    a[i] = *reinterpret_cast<bf16x8*>(&x);
    b[i] = *reinterpret_cast<bf16x8*>(&y);
  }
  float r = 0.f;
  if (MODE == 0) {
    f32x16 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) r += acc[i][j][e];
  } else {
    f32x4 acc[4][4][4];    // the same 64 accumulator registers x 4 = 256 as mode 0: 64 tiles of 16x16
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) for (int e = 0; e < 4; ++e) acc[i][j][q][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + q) & 3], b[j], acc[i][j][q], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) for (int e = 0; e < 4; ++e) r += acc[i][j][q][e];
  }
  out[blockIdx.x * 256 + tid] = r;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
  const bool zeros = argc > 3;
  const int blocks = 256;
  std::vector<unsigned> h(blocks * 256 * 8 * 4);
  unsigned s = 12345;
  for (auto& v : h) {           // random bf16 pairs with exponents near 1.0 (no inf / nan / denormals)
    s = s * 1664525u + 1013904223u;
    const unsigned lo = 0x3f00u | ((s >> 8) & 0xffu) | ((s >> 1) & 0x8000u), hi = 0x3f00u | ((s >> 16) & 0xffu) | ((s >> 3) & 0x8000u);
    v = zeros ? 0u : (lo | (hi << 16));
  }
  uint4* in; float* out;
  hipMalloc(&in, h.size() * 4); hipMalloc(&out, blocks * 256 * 4);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 20000;
  // flops per launch: mode 0: 16 MFMAs x 32*32*16*2 per iteration per wave; mode 1: 64 x 16*16*32*2 -- the same
  const double flops = (double)blocks * 4 * iters * 16.0 * 32 * 32 * 16 * 2;
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  };
  launch(); hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  int n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) { launch(); hipDeviceSynchronize(); ++n; }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("mode %d (%s)%s: %.0f TFLOP/s bf16 dense over %.1f s\n", mode, mode == 0 ? "32x32x16" : "16x16x32", zeros ? " zeros" : "", flops * n / dt / 1e12, dt);
  return 0;
}
