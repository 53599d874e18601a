// gemm2.hip -- bf16x3 GEMM on PRE-SPLIT operands with LDS-DMA staging (the fast path of the bf16x3 arithmetic).
//
// Same contraction and epilogues as gemm.hip (C = epi(alpha * A . B^T), nn.Linear / implicit 3x3 conv of
// guided_diffusion/dit.py and taming/modules/diffusionmodules/model.py), same numerics as its PREC=1 mode
// (a*b ~= ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_bf16, fp32 accumulate) -- but the hi/lo split is done ONCE
// by whoever produces the operand (weights at set_param, activations in the producer's epilogue) instead of by
// every consumer tile.  PMC on the on-the-fly kernel (profiles/r01_gemm_bf16x3_fc1_pmc.txt) showed why: ~9 VALU
// per MFMA (the split) keep the VALU pipe busier (41 %) than the matrix pipe (33 %), and the VGPR round trip
// serialises load-wait / split / ds_write behind the MFMAs of the same wave.
//
// "Split row" format: a logical fp32 row of K elements occupies the same K*4 bytes -- K bf16 `hi` followed by
// K bf16 `lo` (x ~= hi + lo, both round-to-nearest) -- so split tensors drop into fp32-sized buffers and strides.
//
// Structure: global -> LDS by `global_load_lds_dwordx4` (no VGPRs, no VALU): one wave-instruction moves 16 rows x
// 64 B of one plane; the XOR swizzle that keeps the ds_read_b128 fragment reads conflict-free is applied on the
// per-lane SOURCE address (the LDS image of a DMA is lane-linear).  3-stage LDS ring, DMA two K-tiles ahead,
// counted `s_waitcnt vmcnt(N)` + one raw `s_barrier` per K-tile (never __syncthreads: it would drain the DMA queue).
// Out-of-range rows (M / N tails, conv zero padding) read a zero page, so the kernel has no divergent loads.
#include <vector>
#include "common.h"

namespace rgm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  // 16 B per lane, LDS destination = wave-uniform base + lane*16
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN, int ALOAD>
__global__ __launch_bounds__(WM* WN * 64) void gemm2_kernel(GemmParams p, const char* __restrict__ zero_page, int tiles_m,
                                                            int tiles_n) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int STAGE = (BM + BN) * 128;        // bytes per ring stage: [A hi | A lo | B hi | B lo], 64 B per row and plane
  constexpr int SEGS = (BM + BN) / 8;           // 1-KiB DMA segments per stage (16 rows x 64 B of one plane)
  constexpr int SPW = SEGS / NW;                // segments per wave
  static_assert(SEGS % NW == 0, "segments must divide evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char ring[];   // the ONLY shared object (see guide: a 2nd one forces vmcnt(0))

  // ---- blockIdx -> tile (same XCD-contiguous grouped raster as gemm.hip)
  const int nb = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, loc = bid >> 3, q = nb >> 3, r = nb & 7;
  const int sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int grp = sid / per_group;
  const int first_m = grp * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int in_g = sid - grp * per_group;
  const int m0 = (first_m + in_g % gsz) * BM;
  const int n0 = (in_g / gsz) * BN;
  const int z = blockIdx.z;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;

  // ---- per-lane DMA sources: segment s of a stage = plane-major [A hi: BM/16][A lo: BM/16][B hi: BN/16][B lo: BN/16]
  const char* Ab = reinterpret_cast<const char*>(p.A + (long long)z * p.sA);
  const char* Bb = reinterpret_cast<const char*>(p.B + (long long)z * p.sB);
  const int r16 = lane >> 2;                                  // row within the 16-row segment
  const int csrc = ((lane & 3) ^ ((r16 >> 2) & 3)) << 4;      // logical 16-B chunk this lane fetches (swizzle on the source)
  const char* src[SPW];
  int inc[SPW];                                               // bytes to advance per K-tile (0 for zero-page lanes)
  int a_y[SPW], a_x[SPW];
  long long a_img[SPW];
  bool a_row_ok[SPW], is_a[SPW];
  int plane_off[SPW];
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int s = wave + i * NW;
    const bool isA = s < BM / 8;
    const int sp = isA ? s : s - BM / 8;
    const int rows = isA ? BM : BN;
    const int plane = sp >= rows / 16;
    const int row_t = (sp - plane * (rows / 16)) * 16 + r16;  // row within the tile
    is_a[i] = isA;
    if (isA) {
      const int row = m0 + row_t;
      a_row_ok[i] = row < p.M;
      if (ALOAD == 0) {
        plane_off[i] = plane * p.K * 2;
        src[i] = a_row_ok[i] ? Ab + (long long)row * p.lda * 4 + plane_off[i] + csrc : zero_page + csrc;
        inc[i] = a_row_ok[i] ? 64 : 0;
        a_y[i] = a_x[i] = 0;
        a_img[i] = 0;
      } else {  // NHWC split activations: pixel row = Cin bf16 hi | Cin bf16 lo ; source recomputed per tap
        plane_off[i] = plane * p.Cin * 2;
        const int img = row >> (p.logH + p.logW);
        a_y[i] = (row >> p.logW) & (p.H - 1);
        a_x[i] = row & (p.W - 1);
        a_img[i] = (long long)img * (p.H >> p.ups) * (p.W >> p.ups) * p.Cin * 4;
        src[i] = zero_page + csrc;
        inc[i] = 0;
      }
    } else {
      const int row = n0 + row_t;
      const bool ok = row < p.N;
      a_row_ok[i] = ok;
      plane_off[i] = plane * p.K * 2;
      src[i] = ok ? Bb + (long long)row * p.ldb * 4 + plane_off[i] + csrc : zero_page + csrc;
      inc[i] = ok ? 64 : 0;
      a_y[i] = a_x[i] = 0;
      a_img[i] = 0;
    }
  }
  const int cpt = (ALOAD == 1) ? (p.Cin >> 5) : 1;  // K-tiles per 3x3 tap

  auto issue = [&](int kt, int stage) {
    char* dst = ring + stage * STAGE;
    if (ALOAD == 1 && kt % cpt == 0) {               // entering a new tap: re-aim the A segments
      const int tap = kt / cpt;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int Win = p.W >> p.ups;
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
        if (!is_a[i]) continue;
        const int yy = a_y[i] + dy, xx = a_x[i] + dx;
        const bool ok = a_row_ok[i] && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
        src[i] = ok ? Ab + a_img[i] + ((long long)(yy >> p.ups) * Win + (xx >> p.ups)) * p.Cin * 4 + plane_off[i] + csrc
                    : zero_page + csrc;
        inc[i] = ok ? 64 : 0;
      }
    }
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      dma16(src[i], dst + (wave + i * NW) * 1024);
      src[i] += inc[i];
    }
  };

  const int wr = wave / WN, wc = wave - wr * WN;
  const int arow0 = wr * TM * 32, bcol0 = wc * TN * 32;
  const int rq = (l31 >> 2) & 3;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int KT = p.K >> 5;
  issue(0, 0);
  if (KT > 1) issue(1, 1);
  int stage = 0;
  for (int kt = 0; kt < KT; ++kt) {
    // tile kt has landed once at most the NEXT tile's SPW segments of this wave are still in flight
    if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's part of tile kt is in LDS; everyone is done reading stage (kt-1)%3
    if (kt + 2 < KT) issue(kt + 2, stage == 0 ? 2 : stage - 1);   // (kt+2)%3 == (stage+2)%3
    const char* Ah = ring + stage * STAGE;
    const char* Bh = Ah + BM * 128;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int co = ((2 * st + hh) ^ rq) << 4;
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ro = (arow0 + i * 32 + l31) * 64 + co;
        ah[i] = *reinterpret_cast<const bf16x8*>(Ah + ro);
        al[i] = *reinterpret_cast<const bf16x8*>(Ah + BM * 64 + ro);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int ro = (bcol0 + i * 32 + l31) * 64 + co;
        bh[i] = *reinterpret_cast<const bf16x8*>(Bh + ro);
        bl[i] = *reinterpret_cast<const bf16x8*>(Bh + BN * 64 + ro);
      }
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in) {
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[im], bh[in], acc[im][in], 0, 0, 0);
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[im], bl[in], acc[im][in], 0, 0, 0);
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[im], bh[in], acc[im][in], 0, 0, 0);
        }
    }
    stage = stage == 2 ? 0 : stage + 1;
  }

  // ---- epilogue (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)); optional split output
  float* __restrict__ Cb = p.C + (long long)z * p.sC;
  const float* resb = p.res ? p.res + (long long)z * p.sRes : nullptr;
  const float* biasb = p.bias ? p.bias + (long long)z * p.sBias : nullptr;
#pragma unroll
  for (int im = 0; im < TM; ++im) {
#pragma unroll
    for (int in = 0; in < TN; ++in) {
      const int col = n0 + bcol0 + in * 32 + l31;
      if (col >= p.N) continue;
      const float bv = biasb ? biasb[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + arow0 + im * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (row >= p.M) continue;
        float v = acc[im][in][e] * p.alpha + bv;
        if (p.act == 1) v = silu_f(v);
        else if (p.act == 2) v = gelu_tanh_f(v);
        if (p.gate) v *= p.gate[(long long)(row / p.rows_per_gate) * p.gate_ld + col];
        if (resb) v += resb[(long long)row * p.ldres + col];
        if (p.out_split) {   // row = N bf16 hi | N bf16 lo in the same ldc*4 bytes
          __bf16* rowp = reinterpret_cast<__bf16*>(Cb + (long long)row * p.ldc);
          const __bf16 hi = (__bf16)v;
          rowp[col] = hi;
          rowp[p.N + col] = (__bf16)(v - (float)hi);
        } else {
          Cb[(long long)row * p.ldc + col] = v;
        }
      }
    }
  }
}

static char* g_zero_page = nullptr;

struct Prof2 {
  hipEvent_t a, b;
  int tile;
  double flops;
};
static bool g2_prof_on = false;
static std::vector<Prof2> g2_prof;

template <int BM, int BN, int WM, int WN>
static int launch2(const GemmParams& p, hipStream_t s, int tile_id) {
  if (!g_zero_page) {
    RGM_CHECK_HIP(hipMalloc(&g_zero_page, 4096));
    RGM_CHECK_HIP(hipMemset(g_zero_page, 0, 4096));
  }
  const int tm = cdiv(p.M, BM), tn = cdiv(p.N, BN);
  const size_t lds = (size_t)3 * (BM + BN) * 128;
  static bool attr0 = false, attr1 = false;
  auto k0 = gemm2_kernel<BM, BN, WM, WN, 0>;
  auto k1 = gemm2_kernel<BM, BN, WM, WN, 1>;
  if (lds > 65536) {
    if (p.aload == 0 && !attr0) {
      RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr0 = true;
    }
    if (p.aload == 1 && !attr1) {
      RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr1 = true;
    }
  }
  dim3 grid(tm * tn, 1, p.batch), block(WM * WN * 64);
  Prof2 rec{};
  if (g2_prof_on) {
    RGM_CHECK_HIP(hipEventCreate(&rec.a));
    RGM_CHECK_HIP(hipEventCreate(&rec.b));
    rec.tile = 40 + tile_id + (p.aload ? 10 : 0);
    rec.flops = 2.0 * p.M * (double)p.N * p.K * p.batch;
    RGM_CHECK_HIP(hipEventRecord(rec.a, s));
  }
  if (p.aload == 0)
    hipLaunchKernelGGL(k0, grid, block, lds, s, p, (const char*)g_zero_page, tm, tn);
  else
    hipLaunchKernelGGL(k1, grid, block, lds, s, p, (const char*)g_zero_page, tm, tn);
  RGM_LAUNCH_CHECK();
  if (g2_prof_on) {
    RGM_CHECK_HIP(hipEventRecord(rec.b, s));
    g2_prof.push_back(rec);
  }
  return RGM_OK;
}

// A and B in split-row format (see top).  tile: 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 5 = 256x128 (8 waves)
int gemm2_launch(const GemmParams& p, hipStream_t s) {
  RGM_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && (p.K & 31) == 0, "gemm2: bad shape M=%d N=%d K=%d (K%%32)", p.M, p.N, p.K);
  RGM_REQUIRE(p.aload == 0 || (p.Cin % 32 == 0 && p.K == 9 * p.Cin), "gemm2: implicit conv needs Cin%%32==0, K=9*Cin");
  RGM_REQUIRE(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 && (p.lda & 3) == 0 && (p.ldb & 3) == 0,
              "gemm2: operands must be 16-byte aligned with ld%%4==0");
  int tile = p.tile;
  if (tile == 0) {
    const long long work = (long long)p.M * p.N * p.batch;
    tile = work >= (long long)8192 * 2048 ? 1 : (work >= (long long)1024 * 1152 ? 2 : 3);
  }
  switch (tile) {
    case 1: return launch2<128, 128, 2, 2>(p, s, 1);
    case 2: return launch2<128, 64, 2, 2>(p, s, 2);
    case 3: return launch2<64, 64, 2, 2>(p, s, 3);
    case 5: return launch2<256, 128, 4, 2>(p, s, 5);
    default: break;
  }
  set_error("gemm2: unknown tile %d", tile);
  return RGM_ERR_INVALID;
}

// x (rows, K) fp32 [row stride ld] -> split rows in place-compatible layout (out may alias x only if they are equal)
__global__ void split_rows_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows, int K, int ld_in, int ld_out) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;   // over rows * K/4
  const int kq = K >> 2;
  if (i >= rows * kq) return;
  const long long row = i / kq;
  const int c = (int)(i - row * kq) * 4;
  const float4 v = *reinterpret_cast<const float4*>(x + row * ld_in + c);
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 hi, lo;
  hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
  lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
  lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
  __bf16* rowp = reinterpret_cast<__bf16*>(out + row * ld_out);
  *reinterpret_cast<bf16x4*>(rowp + c) = hi;
  *reinterpret_cast<bf16x4*>(rowp + K + c) = lo;
}

int split_rows_launch(const float* x, float* out, long long rows, int K, int ld_in, int ld_out, hipStream_t s) {
  RGM_REQUIRE(x && out && rows > 0 && K > 0 && (K & 3) == 0 && x != out, "split_rows: bad arguments (out-of-place, K%%4==0)");
  const long long total = rows * (K >> 2);
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, rows, K, ld_in, ld_out);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

void gemm2_prof(bool on) { g2_prof_on = on; }
void gemm2_prof_reset() {
  for (auto& r : g2_prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g2_prof.clear();
}
int gemm2_prof_report(int kernel, int* launches, double* total_ms, double* total_flops) {
  int n = 0;
  double ms = 0.0, fl = 0.0;
  for (auto& r : g2_prof) {
    if (r.tile != kernel) continue;
    RGM_CHECK_HIP(hipEventSynchronize(r.b));
    float e = 0.f;
    RGM_CHECK_HIP(hipEventElapsedTime(&e, r.a, r.b));
    ms += e;
    fl += r.flops;
    ++n;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  return RGM_OK;
}

}  // namespace rgm

// Split a (rows, K) fp32 matrix into the split-row format consumed by rgm_gemm_split (out-of-place).
extern "C" int rgm_split_rows(const float* x, float* out, int64_t rows, int K, void* stream) {
  return rgm::split_rows_launch(x, out, rows, K, K, K, (hipStream_t)stream);
}

// C[M,N] = act(A . B^T + bias) with A (M,K) and B (N,K) in split-row format; tile as in gemm2_launch; out_split -> C split too.
extern "C" int rgm_gemm_split(const float* A_split, const float* B_split, float* C, int M, int N, int K, const float* bias, int act,
                              int tile, int out_split, void* stream) {
  RGM_REQUIRE(A_split && B_split && C, "gemm_split: null operand");
  rgm::GemmParams g;
  g.A = A_split; g.lda = K; g.B = B_split; g.ldb = K; g.C = C; g.ldc = N;
  g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act; g.tile = tile; g.out_split = out_split;
  return rgm::gemm2_launch(g, (hipStream_t)stream);
}
