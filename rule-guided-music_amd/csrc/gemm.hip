// gemm.hip -- fp32 GEMM on the CDNA4 matrix cores: C = epi(alpha * A[M,K] . B[N,K]^T).
//
// This one kernel family carries every dense contraction of the hot path:
//   nn.Linear of the DiT blocks (qkv / proj / fc1 / fc2 / adaLN / embedders / final)
//       guided_diffusion/dit.py:219-227, :263-288, :326, :333, :372-376
//   1x1 convs and the 3x3 convs of the taming decoder as an implicit GEMM over NHWC activations
//       taming/modules/diffusionmodules/model.py:38-53, :78-137, :140-192
//
// Why fp32 MFMA (v_mfma_f32_32x32x2_f32): the reference computes in fp32 and the contract is 1e-3
// relative on latents after 28 blocks x 50..1000 steps; the f32-input MFMA is bitwise an fmaf chain
// (MI355X_MICROARCH: 157 TF peak = 1/16 of bf16) so parity is a re-association question only.
//
// Structure (wave64, 4 waves / workgroup):
//   * block tile BM x BN x 32, each wave owns TM x TN accumulators of 32x32 (16 VGPR each);
//   * A and B are both K-contiguous ("B^T form" -- nn.Linear weights are [out,in]), staged
//     global -> VGPR (16 B / lane, one full 128-B line per 8 lanes) -> LDS, double buffered, one
//     barrier per K-tile, the next tile's global loads issued before the current tile's MFMAs;
//   * LDS rows are 32 floats = 8 slots of 16 B; slot' = slot ^ ((row >> 1) & 7) makes both the
//     8-lane ds_write_b128 groups and the non-contiguous 16-lane ds_read_b128 groups conflict-free;
//   * fragment trick: the MFMA sums over its two k-slots (lane halves); which k lands in which slot
//     is free as long as A and B agree, so lane-half h reads k = 8j+4h .. 8j+4h+3 with ONE
//     ds_read_b128 and feeds element s to step s (steps cover {s, 4+s}) -- no b32 reads, no shuffles;
//   * blockIdx -> tile map: bijective XCD remap (block b runs on XCD b%8) so each XCD's private L2
//     serves a contiguous band of tiles, rastered in groups of 8 M-tiles so co-resident blocks share
//     A / B panels;
//   * fused epilogue: bias, SiLU / GELU(tanh), adaLN gate, residual add (in place allowed).
#include <vector>
#include "common.h"

namespace rgm {

typedef split_t bf16x8 __attribute__((ext_vector_type(8)));
typedef split_t bf16x4 __attribute__((ext_vector_type(4)));

// fp32 -> (hi, lo) bf16 pair with x ~= hi + lo to ~2^-17 relative (both round-to-nearest-even, v_cvt_pk_bf16_f32)
__device__ __forceinline__ void split_bf16(const float4& v, bf16x4& hi, bf16x4& lo) {
#ifdef RGM_EXPERIMENT_NOSPLIT   // timing experiment only (wrong numerics): what if the operands arrived pre-split?
  const uint2 t = make_uint2(__builtin_amdgcn_perm(__float_as_uint(v.y), __float_as_uint(v.x), 0x07060302u),
                             __builtin_amdgcn_perm(__float_as_uint(v.w), __float_as_uint(v.z), 0x07060302u));
  hi = *reinterpret_cast<const bf16x4*>(&t);
  lo = hi;
  return;
#endif
  hi[0] = (split_t)v.x; hi[1] = (split_t)v.y; hi[2] = (split_t)v.z; hi[3] = (split_t)v.w;
  lo[0] = (split_t)(v.x - (float)hi[0]);
  lo[1] = (split_t)(v.y - (float)hi[1]);
  lo[2] = (split_t)(v.z - (float)hi[2]);
  lo[3] = (split_t)(v.w - (float)hi[3]);
}

// PREC 0: exact fp32 products on v_mfma_f32_32x32x2_f32 (157 TF peak).
// PREC 1: "bf16x3" -- every operand is split into hi+lo bf16 while it is staged to LDS and each product is
//         a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: relative error
//         ~2e-5 per product instead of 6e-8, at 3/16 of the fp32-MFMA issue time per FLOP (833 TF equivalent peak).
template <int BM, int BN, int WM, int WN, int ALOAD, int PREC>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmParams p, int tiles_m, int tiles_n) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int RPP = NT / 8;  // rows staged per pass
  constexpr int PA = BM / RPP, PB = BN / RPP;
  static_assert(PA >= 1 && PB >= 1 && TM >= 1 && TN >= 1, "tile/wave shape");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BM][32]
  float* Bs = smem + 2 * BM * 32;   // [2][BN][32]

  // ---- blockIdx -> (tile_m, tile_n): XCD-contiguous bands, grouped raster
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
  const float* __restrict__ Ab = p.A + (long long)z * p.sA;
  const float* __restrict__ Bb = p.B + (long long)z * p.sB;

  const int tid = threadIdx.x;
  const int srow = tid >> 3, slot = tid & 7;
  const int ssw = (srow >> 1) & 7;

  // ---- per-thread staging coordinates
  long long a_off[PA];
  bool a_ok[PA];
  int a_y[PA], a_x[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = m0 + srow + i * RPP;
    a_ok[i] = row < p.M;
    if (ALOAD == 0) {
      a_off[i] = (long long)row * p.lda + slot * 4;
      a_y[i] = a_x[i] = 0;
    } else {
      const int img = row >> (p.logH + p.logW);
      a_y[i] = (row >> p.logW) & (p.H - 1);
      a_x[i] = row & (p.W - 1);
      // input plane: (H >> ups) x (W >> ups) for the (up)convolutions, 2H x 2W for the stride-2 downsample (ups == -1)
      a_off[i] = (p.ups < 0 ? (long long)img * (2 * p.H) * (2 * p.W) : (long long)img * (p.H >> p.ups) * (p.W >> p.ups)) * p.Cin + slot * 4;
    }
  }
  long long b_off[PB];
  bool b_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = n0 + srow + i * RPP;
    b_ok[i] = row < p.N;
    b_off[i] = (long long)row * p.ldb + slot * 4;
  }
  const int cpt = (ALOAD == 1) ? (p.Cin >> 5) : 1;  // k-tiles per 3x3 tap

  float4 ra0[PA], rb0[PB];
  auto gload = [&](int kt, float4 (&ra)[PA], float4 (&rb)[PB]) {
    if (ALOAD == 0) {
#pragma unroll
      for (int i = 0; i < PA; ++i)
        ra[i] = a_ok[i] ? *reinterpret_cast<const float4*>(Ab + a_off[i] + kt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const int tap = kt / cpt;
      const int c0 = (kt - tap * cpt) << 5;
      if (p.ups >= 0) {   // 3x3, pad 1, optionally on the nearest-x2 upsampled input (taming Upsample, model.py:38-53)
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int Win = p.W >> p.ups;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const int yy = a_y[i] + dy, xx = a_x[i] + dx;
          const bool ok = a_ok[i] && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
          const long long off = a_off[i] + ((long long)(yy >> p.ups) * Win + (xx >> p.ups)) * p.Cin + c0;
          ra[i] = ok ? *reinterpret_cast<const float4*>(Ab + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {            // 3x3, stride 2, zero pad (0,1,0,1) on a 2H x 2W input (taming Downsample, model.py:56-75)
        const int dy = tap / 3, dx = tap - (tap / 3) * 3;
        const int Hin = 2 * p.H, Win = 2 * p.W;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const int yy = 2 * a_y[i] + dy, xx = 2 * a_x[i] + dx;
          const bool ok = a_ok[i] && yy < Hin && xx < Win;
          const long long off = a_off[i] + ((long long)yy * Win + xx) * p.Cin + c0;
          ra[i] = ok ? *reinterpret_cast<const float4*>(Ab + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
      rb[i] = b_ok[i] ? *reinterpret_cast<const float4*>(Bb + b_off[i] + kt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  // PREC 1 LDS image (same bytes as fp32): per operand tile [hi rows | lo rows], a row = 32 bf16 = 64 B = 4 chunks of
  // 16 B (8 k each); chunk' = chunk ^ ((row >> 2) & 3) keeps the 16-lane ds_read_b128 groups conflict-free.
  const int bsw = (srow >> 2) & 3;
  auto lstore = [&](int buf, const float4 (&ra)[PA], const float4 (&rb)[PB]) {
    float* Ad = As + buf * BM * 32;
    float* Bd = Bs + buf * BN * 32;
    if (PREC == 0) {
#pragma unroll
      for (int i = 0; i < PA; ++i)
        *reinterpret_cast<float4*>(Ad + (srow + i * RPP) * 32 + ((slot ^ ssw) << 2)) = ra[i];
#pragma unroll
      for (int i = 0; i < PB; ++i)
        *reinterpret_cast<float4*>(Bd + (srow + i * RPP) * 32 + ((slot ^ ssw) << 2)) = rb[i];
    } else {
      char* Ah = reinterpret_cast<char*>(Ad);
      char* Bh = reinterpret_cast<char*>(Bd);
      const int coff = ((((slot >> 1) ^ bsw) << 4) + ((slot & 1) << 3));
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        bf16x4 hi, lo;
        split_bf16(ra[i], hi, lo);
        const int ro = (srow + i * RPP) * 64 + coff;
        *reinterpret_cast<bf16x4*>(Ah + ro) = hi;
        *reinterpret_cast<bf16x4*>(Ah + BM * 64 + ro) = lo;
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        bf16x4 hi, lo;
        split_bf16(rb[i], hi, lo);
        const int ro = (srow + i * RPP) * 64 + coff;
        *reinterpret_cast<bf16x4*>(Bh + ro) = hi;
        *reinterpret_cast<bf16x4*>(Bh + BN * 64 + ro) = lo;
      }
    }
  };

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wr = wave / WN, wc = wave - wr * WN;
  const int arow0 = wr * TM * 32, bcol0 = wc * TN * 32;
  const int rsw = (l31 >> 1) & 7;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int KT = p.K >> 5;
  auto compute = [&](int buf) {
    const float* Asb = As + buf * BM * 32;
    const float* Bsb = Bs + buf * BN * 32;
    if (PREC == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 a[TM], b[TN];
        const int so = ((2 * j + hh) ^ rsw) << 2;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(Asb + (arow0 + i * 32 + l31) * 32 + so);
#pragma unroll
        for (int i = 0; i < TN; ++i) b[i] = *reinterpret_cast<const f32x4*>(Bsb + (bcol0 + i * 32 + l31) * 32 + so);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int im = 0; im < TM; ++im)
#pragma unroll
            for (int in = 0; in < TN; ++in)
              acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[im][s], b[in][s], acc[im][in], 0, 0, 0);
      }
    } else {
      const char* Ah = reinterpret_cast<const char*>(Asb);
      const char* Bh = reinterpret_cast<const char*>(Bsb);
      const int rq = (l31 >> 2) & 3;
#pragma unroll
      for (int st = 0; st < 2; ++st) {          // two k-steps of 16 per 32-wide K tile; lane half hh holds 8 of the 16
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
            acc[im][in] = RGM_MFMA_SPLIT_32x32x16(al[im], bh[in], acc[im][in], 0, 0, 0);
            acc[im][in] = RGM_MFMA_SPLIT_32x32x16(ah[im], bl[in], acc[im][in], 0, 0, 0);
            acc[im][in] = RGM_MFMA_SPLIT_32x32x16(ah[im], bh[in], acc[im][in], 0, 0, 0);
          }
      }
    }
  };
  // one K-tile of prefetch: the next tile's global loads are issued before this tile's MFMAs and written to the other
  // LDS buffer after them.  (A two-tile-deep register prefetch was measured for bf16x3: -17 % -- the extra 24-32
  // VGPRs cost a wave of occupancy and the kernel is not latency-bound.)
  gload(0, ra0, rb0);
  lstore(0, ra0, rb0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload(kt + 1, ra0, rb0);
    compute(buf);
    if (kt + 1 < KT) lstore(buf ^ 1, ra0, rb0);
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float* __restrict__ Cb = p.C + (long long)z * p.sC;
  const float* resb = p.res ? p.res + (long long)z * p.sRes : nullptr;
  const float* biasb = p.bias ? p.bias + (long long)z * p.sBias : nullptr;
  const float* auxb = p.aux ? p.aux + (long long)z * p.sAux : nullptr;
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
        else if (p.act == 3) v *= gelu_tanh_grad_f(auxb[(long long)row * p.ldaux + col]);
        else if (p.act == 4) v *= silu_grad_f(auxb[(long long)row * p.ldaux + col]);
        if (p.gate) v *= p.gate[(long long)(row / p.rows_per_gate) * p.gate_ld + col];
        if (resb) v += resb[(long long)row * p.ldres + col];
        if (p.out_split) {   // split-row output (common.h split_idx) feeding a pre-split consumer (gemm2.hip)
          split_t* rowp = reinterpret_cast<split_t*>(Cb + (long long)row * p.ldc);
          const split_t hi = (split_t)v;
          rowp[split_idx(col)] = hi;
          rowp[split_idx(col) + 32] = (split_t)(v - (float)hi);
        } else {
          Cb[(long long)row * p.ldc + col] = v;
        }
      }
    }
  }
}

// ---- optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg)
struct ProfRec {
  hipEvent_t a, b;
  int tile;
  double flops;
};
static bool g_prof_on = false;
static int g_default_prec = 0;   // 0 fp32 MFMA (parity default), 1 bf16x3
static std::vector<ProfRec> g_prof;

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmParams& p, hipStream_t s, int tile_id) {
  const int tm = cdiv(p.M, BM), tn = cdiv(p.N, BN);
  const size_t lds = (size_t)2 * (BM + BN) * 32 * sizeof(float);
  dim3 grid(tm * tn, 1, p.batch), block(WM * WN * 64);
  if (lds > 65536) {   // 256x128 tile: 96 KiB of dynamic LDS needs the opt-in (once per instantiation)
    static bool done[2][2] = {{false, false}, {false, false}};
    const int pr = (p.prec < 0 ? g_default_prec : p.prec) ? 1 : 0, al = p.aload ? 1 : 0;
    if (!done[pr][al]) {
      const void* fn = pr ? (al ? (const void*)gemm_kernel<BM, BN, WM, WN, 1, 1> : (const void*)gemm_kernel<BM, BN, WM, WN, 0, 1>)
                          : (al ? (const void*)gemm_kernel<BM, BN, WM, WN, 1, 0> : (const void*)gemm_kernel<BM, BN, WM, WN, 0, 0>);
      RGM_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      done[pr][al] = true;
    }
  }
  ProfRec rec{};
  if (g_prof_on) {
    RGM_CHECK_HIP(hipEventCreate(&rec.a));
    RGM_CHECK_HIP(hipEventCreate(&rec.b));
    rec.tile = tile_id + (p.aload ? 10 : 0) + ((p.prec < 0 ? g_default_prec : p.prec) ? 20 : 0);
    rec.flops = 2.0 * p.M * (double)p.N * p.K * p.batch;
    RGM_CHECK_HIP(hipEventRecord(rec.a, s));
  }
  const int prec = (p.prec < 0 ? g_default_prec : p.prec) ? 1 : 0;
  if (prec == 0) {
    if (p.aload == 0)
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 0, 0>), grid, block, lds, s, p, tm, tn);
    else
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1, 0>), grid, block, lds, s, p, tm, tn);
  } else {
    if (p.aload == 0)
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 0, 1>), grid, block, lds, s, p, tm, tn);
    else
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1, 1>), grid, block, lds, s, p, tm, tn);
  }
  RGM_LAUNCH_CHECK();
  if (g_prof_on) {
    RGM_CHECK_HIP(hipEventRecord(rec.b, s));
    g_prof.push_back(rec);
  }
  return RGM_OK;
}

static double wave_eff(int M, int N, int bm, int bn, int batch) {
  const double tiles = (double)cdiv(M, bm) * cdiv(N, bn) * batch;
  const double rounds = tiles / 256.0;
  const double useful = ((double)M * N * batch) / (tiles * bm * bn);  // padding waste
  return useful * rounds / (double)((long long)(rounds + 0.999999));
}

int gemm_launch(const GemmParams& p, hipStream_t s) {
  RGM_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && (p.K & 31) == 0, "gemm: bad shape M=%d N=%d K=%d (K%%32)", p.M, p.N, p.K);
  RGM_REQUIRE(p.aload == 0 || (p.Cin % 32 == 0 && p.K == 9 * p.Cin), "gemm: implicit conv needs Cin%%32==0, K=9*Cin");
  RGM_REQUIRE(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 && (p.lda & 3) == 0 && (p.ldb & 3) == 0,
              "gemm: operands must be 16-byte aligned with ld%%4==0");
  RGM_REQUIRE(!p.out_split || ((p.N & 31) == 0 && (p.ldc & 31) == 0), "gemm: split-row output needs N%%32==0 (N=%d)", p.N);
  int tile = p.tile;
  const int prec = (p.prec < 0 ? g_default_prec : p.prec) ? 1 : 0;
  if (tile == 0) {
    if (p.M <= 64) tile = 4;
    else if (prec == 1) {
      // bf16x3: the MFMA work per byte staged is 5x shorter than fp32, so operand reuse (tile area) matters more
      // than the last CU-round.  Measured on MI355X (tools/gemm_sweep.py): 128x64 wins from M~2k up (187-245 TF),
      // 64x64 below (small grids), 128x128 only pays at M >= 16k.
      const long long work = (long long)p.M * p.N * p.batch;
      tile = work >= (long long)16384 * 4096 ? 1 : (work >= (long long)2048 * 1152 ? 2 : 3);
      // short-K dense GEMMs over millions of rows (the VAE's 1x1 nin_shortcut convs and their input gradients) are HBM-bound:
      // the smaller tile keeps more loads in flight per CU (tools/nin_shapes.py: 2.9 vs 2.3 TB/s at K = 256)
      if (tile == 1 && !p.aload && p.K <= 512) tile = 2;
    } else {
      // fp32: pick the tile that wastes the fewest CU-rounds; smaller tiles pay more L2->LDS traffic
      const double e1 = wave_eff(p.M, p.N, 128, 128, p.batch) * 0.90;
      const double e2 = wave_eff(p.M, p.N, 128, 64, p.batch) * 0.98;
      const double e3 = wave_eff(p.M, p.N, 64, 64, p.batch) * 0.96;
      tile = 1;
      double best = e1;
      if (e2 > best) { best = e2; tile = 2; }
      if (e3 > best) { best = e3; tile = 3; }
    }
  }
  switch (tile) {
    case 1: return launch_cfg<128, 128, 2, 2>(p, s, 1);
    case 2: return launch_cfg<128, 64, 2, 2>(p, s, 2);
    case 3: return launch_cfg<64, 64, 2, 2>(p, s, 3);
    case 4: return launch_cfg<32, 128, 1, 4>(p, s, 4);
    case 5: return launch_cfg<256, 128, 4, 2>(p, s, 5);
    default: break;
  }
  set_error("gemm: unknown tile %d", tile);
  return RGM_ERR_INVALID;
}

}  // namespace rgm

// Profiling hooks: with profiling on, every GEMM launch is bracketed by two hipEvents on ITS stream.
// kernel ids: 1..4 = dense tiles (128x128, 128x64, 64x64, 32x128), 11..14 = the same tiles with the implicit-conv loader.
// Default arithmetic of every GEMM that does not ask for one explicitly: 0 = exact fp32 MFMA, 1 = bf16x3 split.
extern "C" int rgm_set_gemm_precision(int prec) {
  RGM_REQUIRE(prec >= 0 && prec <= 2, "set_gemm_precision: %d (0 = fp32, 1 = bf16x3 split on the fly, 2 = bf16x3 with pre-split operands where available)", prec);
  rgm::g_default_prec = prec;
  return RGM_OK;
}
extern "C" int rgm_get_gemm_precision(void) { return rgm::g_default_prec; }

extern "C" int rgm_prof_enable(int on) {
  rgm::g_prof_on = on != 0;
  rgm::gemm2_prof(on != 0);
  return RGM_OK;
}
extern "C" int rgm_prof_reset(void) {
  for (auto& r : rgm::g_prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  rgm::g_prof.clear();
  rgm::gemm2_prof_reset();
  return RGM_OK;
}
// sums over the recorded launches of kernel id `kernel`: launches, total milliseconds, total algorithmic FLOPs (2MNK)
extern "C" int rgm_prof_report(int kernel, int* launches, double* total_ms, double* total_flops) {
  if (kernel >= 40) return rgm::gemm2_prof_report(kernel, launches, total_ms, total_flops);   // gemm2.hip kernels
  int n = 0;
  double ms = 0.0, fl = 0.0;
  for (auto& r : rgm::g_prof) {
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
