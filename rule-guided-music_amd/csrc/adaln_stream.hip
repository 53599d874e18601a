// adaln_stream.hip -- the adaLN conditioning of a whole DiT forward as ONE weight-streaming pass.
//
//   mod[N, L] = SiLU(c)[N, D] . W^T[L, D] + b[L]        guided_diffusion/dit.py:333 (adaLN_modulation of every DiTBlockRotary),
//                                                        :374 (final layer); all projections contiguous in the arena (dit.hip)
//
// L = (6 depth + 2) D = 195 840 rows of 1152 floats for DiTRotary_XL_8: 0.9 GB of weights for N <= 32 rows of SiLU(c).  The work is the
// stream, not the arithmetic (7.2 GFLOP at N = 16), so this is an HBM kernel: the bound is 0.9 GB at what one pass over HBM reaches
// (~0.15 ms); the tiled GEMM it replaces (gemm_kernel<32,128,1,4>: weights split hi / lo on the fly, a barrier per K-tile) took 0.25 ms.
//
// Shape of the kernel.  One persistent workgroup per CU, 16 waves.  A wave owns 16 consecutive weight rows at a time (a contiguous 72 KiB
// of HBM) and keeps PF 16-byte loads per lane in flight across the rows' K extent AND across its own task boundary -- the stream never
// drains.  The contraction runs on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate: 37 % matrix-pipe duty at N = 16, no
// VALU beside the loads): A = the weights exactly as they arrive -- lane (i = l % 16, g = l / 16) holds W[row0 + i][16 it + 4 g .. +3],
// one component per MFMA --, B = SiLU(c) from an LDS image built once per workgroup in the lanes' own fragment order (lane l reads
// its 16 bytes at [it][l]: linear, conflict-free), D = 16 x 16 outputs, lane (j = l % 16, g) holds mod[j][row0 + 4 g .. +3]: one 16-byte
// store per task.  N <= 16: one image; 16 < N <= 32: two (NB = 2), the weights still read once.
#include "common.h"

namespace rgm {

namespace {
__device__ __forceinline__ f32x4 ldv(const float* p) {          // 16 bytes from GLOBAL memory as a register vector
  return *(const __attribute__((address_space(1))) f32x4*)p;
}
constexpr int PF = 6;          // 16-byte loads in flight per lane (x 16 waves per CU: 96 KiB per CU)

template <int NIT, int NB>
__global__ __launch_bounds__(1024, 1) void adaln_stream_kernel(const float* __restrict__ cs, const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ out, int N, int L,
                                                               int ldo) {
  constexpr int D = NIT * 16;
  extern __shared__ __attribute__((aligned(16))) char sm_ada[];
  float4* img = reinterpret_cast<float4*>(sm_ada);          // [NB][NIT][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int ntask = L >> 4;
  const int stride = gridDim.x * 16;
  int task = blockIdx.x + gridDim.x * wave;
  // the stream starts before the image is built (the first PF loads of this wave's first task)
  f32x4 w[PF];
  const float* wp = W + ((long long)(task < ntask ? task : 0) * 16 + i16) * D + 4 * g;
#pragma unroll
  for (int u = 0; u < PF; ++u) w[u] = ldv(wp + 16 * u);
  for (int e = tid; e < NB * NIT * 64; e += 1024) {
    const int l = e & 63, it = (e >> 6) % NIT, nb = e / (64 * NIT);
    const int j = nb * 16 + (l & 15), k = it * 16 + 4 * (l >> 4);
    img[e] = j < N ? *reinterpret_cast<const float4*>(cs + (long long)j * D + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  for (; task < ntask; task += stride) {
    const int nxt = task + stride < ntask ? task + stride : task;      // (no next task: re-read this one's first lines, unused)
    const float* np = W + ((long long)nxt * 16 + i16) * D + 4 * g;
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_for<0, NIT>([&](auto itc) {
      constexpr int it = decltype(itc)::value;
      const f32x4 a = w[it % PF];
      // refill the slot: K chunk it + PF of this task, or the next task's first chunks
      if constexpr (it + PF < NIT) w[it % PF] = ldv(wp + 16 * (it + PF));
      else w[it % PF] = ldv(np + 16 * (it + PF - NIT));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float4 b = img[(nb * NIT + it) * 64 + lane];
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b.x, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b.y, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b.z, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b.w, acc[nb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);        // (the scheduler otherwise hoists all NIT image reads in front of the loop and spills them)
    });
    const int col = task * 16 + 4 * g;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) bv = ldv(bias + col);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int j = nb * 16 + i16;
      if (j < N) {
        const f32x4 r = acc[nb] + bv;
        *reinterpret_cast<f32x4*>(out + (long long)j * ldo + col) = r;
      }
    }
    wp = np;
  }
}

template <int NIT, int NB>
int launch_ada(const float* cs, const float* W, const float* bias, float* out, int N, int L, int ldo, hipStream_t s) {
  auto kern = adaln_stream_kernel<NIT, NB>;
  // ONE workgroup per CU, enforced through the LDS request (common.h attn_lds_one_per_cu): the kernel is sized as a persistent workgroup per CU,
  // and it never shares a CU with a twin of itself (tools/isa_lint.py counts it as `single`)
  const size_t lds = attn_lds_one_per_cu((size_t)NB * NIT * 1024);
  // per device of the calling thread (one process per GPU is the rule; a process that drives several devices gets each one's CU count and attribute)
  static int cus_of[16] = {0};
  int dev = 0;
  RGM_CHECK_HIP(hipGetDevice(&dev));
  const int slot = dev & 15;
  if (!cus_of[slot]) {
    int n = 0;
    RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RGM_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus_of[slot] = n > 0 ? n : 256;
  }
  const int cus = cus_of[slot];
  const int ntask = L >> 4;
  const int grid = ntask < cus * 16 ? (ntask + 15) / 16 : cus;      // never more workgroups than tasks / 16
  hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, s, cs, W, bias, out, N, L, ldo);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
}  // namespace

// -> RGM_OK when the streaming kernel took the shape; 1 when it does not cover it (the caller runs its tiled GEMM).
// More than 32 rows: one pass over the weights per 32 rows (C4's 64 candidate rows: two passes, 0.4 ms of a 210 ms step) -- every row is the
// same fixed-order fp32 sum whatever batch it sits in, so a sample's conditioning does not depend on the batch size.
int adaln_stream_launch(const float* cs, const float* W, const float* bias, float* out, int N, int D, int L, int ldo, hipStream_t s) {
  static const int off = getenv("RGM_ADALN_STREAM") ? !atoi(getenv("RGM_ADALN_STREAM")) : 0;     // RGM_ADALN_STREAM=0: the tiled GEMM (A/B runs)
  if (off || N < 1 || N > 256 || (L & 15) || (ldo & 3) || ((uintptr_t)cs & 15) || ((uintptr_t)W & 15) || ((uintptr_t)out & 15) ||
      (bias && ((uintptr_t)bias & 15)))
    return 1;
  if (D != 1152 && D != 384 && D != 768) return 1;
  for (int j0 = 0; j0 < N; j0 += 32) {
    const int n = N - j0 < 32 ? N - j0 : 32;
    const float* c = cs + (long long)j0 * D;
    float* o = out + (long long)j0 * ldo;
    int rc;
    if (D == 1152) rc = n <= 16 ? launch_ada<72, 1>(c, W, bias, o, n, L, ldo, s) : launch_ada<72, 2>(c, W, bias, o, n, L, ldo, s);
    else if (D == 384) rc = n <= 16 ? launch_ada<24, 1>(c, W, bias, o, n, L, ldo, s) : launch_ada<24, 2>(c, W, bias, o, n, L, ldo, s);
    else rc = n <= 16 ? launch_ada<48, 1>(c, W, bias, o, n, L, ldo, s) : launch_ada<48, 2>(c, W, bias, o, n, L, ldo, s);
    if (rc != RGM_OK) return rc;
  }
  return RGM_OK;
}

}  // namespace rgm

extern "C" int rgm_adaln_stream(const float* cs, const float* W, const float* bias, float* mod, int N, int D, int L, void* stream) {
  RGM_REQUIRE(cs && W && mod, "adaln_stream: null tensor");
  const int rc = rgm::adaln_stream_launch(cs, W, bias, mod, N, D, L, L, (hipStream_t)stream);
  if (rc > 0) {
    rgm::set_error("adaln_stream: shape N=%d D=%d L=%d not covered (N <= 256, D in {384, 768, 1152}, L %% 16 == 0, 16-byte aligned pointers)", N, D, L);
    return RGM_ERR_INVALID;
  }
  return rc;
}
