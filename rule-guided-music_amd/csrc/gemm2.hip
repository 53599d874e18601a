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
// "Split row" format (common.h split_idx): a logical fp32 row of K elements keeps its K*4 bytes; every block of 32
// elements is one 128-byte line [32 bf16 hi | 32 bf16 lo] (x ~= hi + lo, both round-to-nearest) -- so split tensors
// drop into fp32-sized buffers and strides, and one row of one 32-wide K tile is exactly one cache line (a layout
// with separate hi / lo planes made every DMA piece touch 16 half-used lines: measured ~31 cycles per piece per CU).
//
// Structure: global -> LDS by `global_load_lds_dwordx4` (no VGPRs, no VALU): one wave-instruction moves 8 rows x
// 128 B; the XOR swizzle that keeps the ds_read_b128 fragment reads conflict-free is applied on the per-lane
// SOURCE address (the LDS image of a DMA is lane-linear).  3-stage LDS ring, DMA two K-tiles ahead,
// counted `s_waitcnt vmcnt(N)` + one raw `s_barrier` per K-tile (never __syncthreads: it would drain the DMA queue).
// Out-of-range rows (M / N tails, conv zero padding) read a zero page, so the kernel has no divergent loads.
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include "common.h"
#include "gemm2_body.h"
#include "ln_body.h"

namespace rgm {

template <int BM, int BN, int WM, int WN, int ALOAD, int NSTAGE, int DBG = 0, int PIPE = 0>
__global__ __launch_bounds__(WM* WN * 64 * (PIPE == 4 ? 2 : 1)) void gemm2_kernel(GemmParams p, const char* __restrict__ zero_page, int tiles_m,
                                                            int tiles_n, int exp, long long* __restrict__ dbg = nullptr) {
  gemm2_body<BM, BN, WM, WN, ALOAD, NSTAGE, DBG, PIPE>(p, zero_page, tiles_m, tiles_n, exp, dbg, (int)blockIdx.x, (int)blockIdx.z,
                                                       (int)(gridDim.x >> 1));
}

// Two tile shapes in ONE launch: workgroups [0, nbig) compute 128x128 tiles of `pb` (whole CU-rounds of the big tile, the
// efficient shape), workgroups [nbig, gridDim.x) compute SBM x SBN tiles of `ps` (the leftover columns).  Workgroups are
// dispatched in index order, so the launch ENDS on small tiles: the last, partly filled round costs a fraction of a big
// tile's time instead of a whole one (fc1 at B = 16: 1024 + 128 tiles of 128x128 = 2.25 rounds that cost ~3; here 2 rounds
// + 256 tiles of 128x64).  No K split, no partial sums: every output element is still one workgroup's fixed-order sum.
template <int SBM, int SBN>
__global__ __launch_bounds__(256) void gemm2_dual_kernel(GemmParams pb, GemmParams ps, const char* __restrict__ zero_page, int tmb, int tnb,
                                                         int tms, int tns, int nbig, int exp) {
  if ((int)blockIdx.x < nbig)
    gemm2_body<128, 128, 2, 2, 0, 2, 0, 3>(pb, zero_page, tmb, tnb, exp, nullptr, (int)blockIdx.x, 0, -1);
  else
    gemm2_body<SBM, SBN, 2, 2, 0, (SBM == 64 ? 3 : 2), 0, 3>(ps, zero_page, tms, tns, exp, nullptr, (int)blockIdx.x - nbig, 0, -1);
}

static char* g_zero_page = nullptr;
// rgm_set_big_tiles: mode 0 = the heuristics never pick the one-wave-per-SIMD kernels (tiles 71 / 72), 1 = they do (default); min_tiles =
// tiles a launch must have before the VAE convs take them (default 256 = one round of the chip; parity tests set 1 so that the small
// golden inputs run through the big-tile kernels too).  RGM_BIG_TILES=0 in the environment: same as mode 0 (A/B runs).
static int g_big_tiles = getenv("RGM_BIG_TILES") ? atoi(getenv("RGM_BIG_TILES")) : 1;
static int g_big_min_tiles = 256;
int big_tiles_mode() { return g_big_tiles; }
int big_tiles_min() { return g_big_min_tiles; }
static long long* g_dbg = nullptr;   // set by rgm_gemm2_dbg: stamped kernel variant (tools/gemm_stamp.py)
// timing experiments only (wrong results): RGM_GEMM2_EXP=1 no DMA after the prologue, =2 DMA + barriers only
static int g_exp = RGM_EXP_ENV("RGM_GEMM2_EXP");

struct Prof2 {
  hipEvent_t a, b;
  int tile;
  double flops;
  double bytes;     // algorithmic HBM bytes of the launch: every operand read once, the output written once
};
static bool g2_prof_on = false;
static std::vector<Prof2> g2_prof;

template <int BM, int BN, int WM, int WN, int NSTAGE = 3, int PIPE = 0>
static int launch2(const GemmParams& p, hipStream_t s, int tile_id) {
  if (!g_zero_page) {
    RGM_CHECK_HIP(hipMalloc(&g_zero_page, 4096));
    RGM_CHECK_HIP(hipMemset(g_zero_page, 0, 4096));
  }
  const int tm = cdiv(p.M, BM), tn = cdiv(p.N, BN);
  size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
  // wide wave tiles (256x288 = 4 waves of 64 x 288; gemm2_body.h SBLO): dense operands, the plain epilogue (bias, SiLU / GELU, fp32 or split rows),
  // four 32 x 288 slabs + the bias tile in LDS behind the K loop
  constexpr bool WIDE = PIPE == 5 && ((BM / (WM * 32)) * (BN / (WN * 32)) > 16 || 64 % ((BN / (WN * 32)) * 8) != 0);   // (gemm2_body.h: the linear slab epilogue)
  if (WIDE) {
    RGM_REQUIRE(!p.aload && !p.gate && !p.res && !p.stats && !p.aux && !p.C2 && p.act < 3 && !p.ln_out && ((p.N | p.ldc) & 7) == 0 &&
                (((uintptr_t)p.C | (uintptr_t)p.bias) & 15) == 0,
                "gemm2: the %dx%d tile takes dense operands and the plain epilogue (bias, SiLU / GELU, fp32 or split rows; N %% 8 == 0)", BM, BN);
    const size_t epi = (size_t)WM * WN * 32 * (BN / WN) * 4 + (size_t)BN * 4;
    if (epi > lds) lds = epi;
  }
  // (measured, round 5: XL-28 forward at B = 2 / 3 / 4 / 6 / 8 4.10 / 4.73 / 5.30 / 6.34 / 7.86 ms without, 4.22 / 4.69 / 5.29 / 6.60 / 7.82 with a
  // distance of 8 -- unlike the 144-column kernel, whose loaders were the in-situ bottleneck, these tiles do not gain: off)
  static const int p4_pf = getenv("RGM_P4_PF") ? atoi(getenv("RGM_P4_PF")) : 0;
  const bool pf4 = PIPE == 4 && p4_pf > 0 && lds + 1024 <= 160 * 1024 && p.batch == 1;
  if (pf4) lds += 1024;                        // the consumers' prefetch scratch (gemm2_body.h)
  const size_t lds_attr = (PIPE == 4 && (size_t)NSTAGE * (BM + BN) * 128 + 1024 <= 160 * 1024) ? (size_t)NSTAGE * (BM + BN) * 128 + 1024 : lds;   // the attribute is set once
  static bool attr0 = false, attr1 = false;
  auto k0 = gemm2_kernel<BM, BN, WM, WN, 0, NSTAGE, 0, PIPE>;
  auto k1 = gemm2_kernel<BM, BN, WM, WN, (PIPE == 4 || WIDE) ? 0 : 1, NSTAGE, 0, PIPE>;
  auto k2 = gemm2_kernel<BM, BN, WM, WN, (PIPE == 5 && !WIDE) ? 2 : 0, NSTAGE, 0, PIPE>;   // implicit conv with channel-block-major K (PIPE 5 only)
  RGM_REQUIRE(!p.conv_kmajor || (p.aload == 1 && p.Cin % 32 == 0), "gemm2: conv_kmajor is a property of the implicit 3x3 conv (aload == 1)");
  RGM_REQUIRE(PIPE != 4 || p.aload == 0, "gemm2: the loader/consumer kernels take dense operands only");
  if (lds_attr > 65536) {
    if (p.aload == 0 && !attr0) {
      RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_attr));
      attr0 = true;
    }
    if (p.aload == 1 && !attr1) {
      RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_attr));
      if (PIPE == 5) RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_attr));
      attr1 = true;
    }
  }
  dim3 grid(tm * tn, 1, p.batch), block(WM * WN * 64 * (PIPE == 4 ? 2 : 1));
  GemmParams pr = p;
  pr.pf_kt = pf4 ? p4_pf : 0;
  {
    // -1 (default): lane pairs everywhere but on the one-wave-per-SIMD tiles, which give a lane EIGHT columns in the plain split-row epilogue
    // (2; gemm2_body.h plain_rows8: C2 12.17 -> 12.11 ms same box) and 2 x 8-byte stores elsewhere; 0 / 1 / 2: forced (A/B runs)
    static const int sp = getenv("RGM_SPLIT_PAIR") ? atoi(getenv("RGM_SPLIT_PAIR")) : -1;
    pr.split_pair = sp < 0 ? (PIPE == 5 ? 2 : 1) : sp;
  }
  if (PIPE == 5) {
    static const int st_plain = getenv("RGM_ST_PLAIN") ? atoi(getenv("RGM_ST_PLAIN")) : 0;
    pr.st_plain = p.out_split ? (st_plain >> 1) & 1 : st_plain & 1;
  }
  if (PIPE == 5 && p.raster_group == 0) {
    // every XCD computes tm * tn / 8 contiguous tiles of the raster: a gm x gn block of them reads gm A panels and gn B panels
    // through that XCD's L2 -- fewest for gm ~ gn (weighted by the tile sides).  In-situ PMC, C2 step: 2.9x the algorithmic bytes
    // on the K slices of fc2 with the fixed 8-row sweep (an XCD's 10 tiles = 8 rows x 1.25 columns)
    const double per_xcd = (double)tm * tn / 8.0;
    int g = (int)(sqrt(per_xcd * (double)BN / (double)BM) + 0.5);
    static const int fixed = RGM_EXP_ENV("RGM_RASTER_GROUP");       // experiments (common.h): a fixed sweep height for A/B runs
    if (fixed > 0) g = fixed;
    pr.raster_group = g < 1 ? 1 : (g > tm ? tm : g);
  }
  RGM_REQUIRE(!p.gate || p.rows_per_gate >= 32 || BM * BN <= 128 * 128 || PIPE != 5,   // (the 256x128 tiles of PIPE 0 / 3 run linear_rows: any gate period)
              "gemm2: the one-wave-per-SIMD tiles take at most two gate rows per 32-row slab (rows_per_gate %d < 32)", p.rows_per_gate);
  Prof2 rec{};
  if (g2_prof_on) {
    RGM_CHECK_HIP(hipEventCreate(&rec.a));
    RGM_CHECK_HIP(hipEventCreate(&rec.b));
    rec.tile = 40 + tile_id + (p.aload ? 10 : 0);
    rec.flops = 2.0 * p.M * (double)p.N * p.K * p.batch;
    // A once (the implicit conv reads its NHWC input, not the 9-tap im2col matrix), B once, C once; K slices (batch) write partials
    const double a_elems = p.aload ? (double)p.M * p.Cin / (double)(1 << (2 * p.ups)) : (double)p.M * p.K * p.batch;
    rec.bytes = 4.0 * (a_elems + (double)p.N * p.K * p.batch + (double)p.M * p.N * p.batch);
    RGM_CHECK_HIP(hipEventRecord(rec.a, s));
  }
#ifdef RGM_GEMM2_STAMPS   // make CXXFLAGS+=-DRGM_GEMM2_STAMPS: also build the s_memtime-stamped kernels (tools/gemm_stamp.py)
  static const int dbg_tile = getenv("RGM_GEMM2_DBG_TILE") ? atoi(getenv("RGM_GEMM2_DBG_TILE")) : 0;   // stamp launches of this tile id only (in-situ stamps)
  if (p.aload == 0 && g_dbg && (dbg_tile == 0 || dbg_tile == tile_id)) {
    auto kd = gemm2_kernel<BM, BN, WM, WN, 0, NSTAGE, 1, PIPE>;
    static bool attrd = false;
    if (lds > 65536 && !attrd) {
      RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attrd = true;
    }
    hipLaunchKernelGGL(kd, grid, block, lds, s, pr, (const char*)g_zero_page, tm, tn, g_exp, g_dbg);
  } else if (PIPE == 5 && p.conv_kmajor && g_dbg) {          // the channel-block-major conv of the one-wave-per-SIMD kernels (tools/conv_stamp.py)
    auto kd = gemm2_kernel<BM, BN, WM, WN, (PIPE == 5 && !WIDE) ? 2 : 0, NSTAGE, 1, PIPE>;
    static bool attrd2 = false;
    if (lds > 65536 && !attrd2) {
      RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attrd2 = true;
    }
    hipLaunchKernelGGL(kd, grid, block, lds, s, pr, (const char*)g_zero_page, tm, tn, g_exp, g_dbg);
  } else
#endif
  if (p.aload == 0)
    hipLaunchKernelGGL(k0, grid, block, lds, s, pr, (const char*)g_zero_page, tm, tn, g_exp, (long long*)nullptr);
  else if (PIPE == 5 && p.conv_kmajor)
    hipLaunchKernelGGL(k2, grid, block, lds, s, pr, (const char*)g_zero_page, tm, tn, g_exp, (long long*)nullptr);
  else
    hipLaunchKernelGGL(k1, grid, block, lds, s, pr, (const char*)g_zero_page, tm, tn, g_exp, (long long*)nullptr);
  RGM_LAUNCH_CHECK();
  if (g2_prof_on) {
    RGM_CHECK_HIP(hipEventRecord(rec.b, s));
    g2_prof.push_back(rec);
  }
  return RGM_OK;
}

// big + small tiles in one launch (gemm2_dual_kernel): returns the number of leading columns that fill whole rounds of 512
// 128x128 workgroups when the columns left over are at most 0.3 round -- otherwise 0 (tools/gemm_sweep.py, tile code 48)
static int dual_big_columns(const GemmParams& p) {
  if (p.aload || p.batch != 1 || p.stats || p.act >= 3 || (p.M & 127) || (p.N & 127)) return 0;
  const int tm = p.M >> 7, tn = p.N >> 7;
  int best = 0;
  for (int c = tn - 1; c >= 1; --c) {              // c big column-tiles: tm * c must be whole rounds
    if (((long long)tm * c) % 512) continue;
    const long long left = (long long)tm * (tn - c);
    if (left <= 154) best = c * 128;               // <= 0.3 round of big tiles left
    break;
  }
  return best;
}

template <int SBM, int SBN>
static int launch_dual(const GemmParams& p, int nb_cols, hipStream_t s) {
  if (!g_zero_page) {
    RGM_CHECK_HIP(hipMalloc(&g_zero_page, 4096));
    RGM_CHECK_HIP(hipMemset(g_zero_page, 0, 4096));
  }
  GemmParams pb = p, ps = p;
  pb.N = nb_cols;
  ps.N = p.N - nb_cols;
  ps.B = p.B + (long long)nb_cols * p.ldb;
  ps.C = p.C + nb_cols;                            // split-row output: a 128-column block is 128 floats wide as well
  if (p.bias) ps.bias = p.bias + nb_cols;
  if (p.res) ps.res = p.res + nb_cols;
  if (p.gate) ps.gate = p.gate + nb_cols;
  const int tmb = cdiv(pb.M, 128), tnb = cdiv(pb.N, 128), tms = cdiv(ps.M, SBM), tns = cdiv(ps.N, SBN);
  auto k = gemm2_dual_kernel<SBM, SBN>;
  static bool attr = false;
  if (!attr) {
    RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    attr = true;
  }
  const int rec = gemm2_prof_begin(88, 2.0 * p.M * (double)p.N * p.K, s);
  hipLaunchKernelGGL(k, dim3(tmb * tnb + tms * tns), dim3(256), 65536, s, pb, ps, (const char*)g_zero_page, tmb, tnb, tms, tns, tmb * tnb, g_exp);
  RGM_LAUNCH_CHECK();
  gemm2_prof_end(rec, s);
  return RGM_OK;
}

// ---- deterministic split-K for small grids (M <= ~1k rows: B = 2..8 latents, the per-rank SCG batches).  A K-split IS a
// batched GEMM: slice s reads A / B at column offset s*K/S (a stride of K/S floats, rows keep their ld) and writes its raw
// partial to P[s][M][N]; this kernel then sums the S partials in a fixed order and applies the epilogue (bias, act, gate,
// residual, split-row output) -- no atomics, bit-reproducible.  One float4 per thread.
__global__ void splitk_reduce_kernel(const float* __restrict__ P, GemmParams p, int S) {
  const long long i4 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int nq = p.N >> 2;
  if (i4 >= (long long)p.M * nq) return;
  const int row = (int)(i4 / nq), col = (int)(i4 - (long long)row * nq) * 4;
  const long long MN = (long long)p.M * p.N;
  float4 a = *reinterpret_cast<const float4*>(P + (long long)row * p.N + col);
  for (int sidx = 1; sidx < S; ++sidx) {
    const float4 b = *reinterpret_cast<const float4*>(P + sidx * MN + (long long)row * p.N + col);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  float v[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
  if (p.bias) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (p.act == 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = silu_f(v[q]);
  } else if (p.act == 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = gelu_tanh_fast_f(v[q]);
  }
  if (p.gate) {
    const float4 g = *reinterpret_cast<const float4*>(p.gate + (long long)(row / p.rows_per_gate) * p.gate_ld + col);
    v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w;
  }
  if (p.res) {
    const float4 r = *reinterpret_cast<const float4*>(p.res + (long long)row * p.ldres + col);
    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
  }
  if (p.out_split) {
    typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      hi[q] = (split_t)v[q];
      lo[q] = (split_t)(v[q] - (float)hi[q]);
    }
    split_t* rowp = reinterpret_cast<split_t*>(p.C + (long long)row * p.ldc);
    *reinterpret_cast<bf16x4*>(rowp + split_idx(col)) = hi;
    *reinterpret_cast<bf16x4*>(rowp + split_idx(col) + 32) = lo;
  } else {
    *reinterpret_cast<float4*>(p.C + (long long)row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// number of K slices for a dense, unbatched GEMM whose 128x64 grid leaves most of the chip idle; 1 = do not split
static int splitk_factor(const GemmParams& p) {
  if (!p.sk_ws) return 1;                        // the partial sums live in caller-provided scratch (include/rgm.h conventions)
  if (p.C2) return 1;                            // two outputs: the plain epilogue only
  if (p.aload || p.batch != 1 || p.tile != 0 || p.act >= 3 || (p.N & 3) || (p.ldc & 3) || (p.ldres & 3) || (p.gate_ld & 3)) return 1;
  if ((((uintptr_t)p.C | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.gate) & 15) != 0) return 1;
  const long long t64 = (long long)cdiv(p.M, 128) * cdiv(p.N, 64);
  const int KT = p.K >> 5;
  if (t64 >= 384 || KT < 72) return 1;          // measured (tools/gemm_sweep.py, SWEEP_SHAPES=small): pays for K = 4608 only
  int best = 1;
  for (int S = 2; S <= 8; ++S) {
    if (KT % S != 0 || KT / S < 9) continue;
    if ((size_t)S * p.M * p.N * sizeof(float) + GEMM_SK_FLAG_BYTES > p.sk_ws_bytes) break;   // the partial sums must fit the caller's scratch: fall back to fewer slices
    best = S;
    if (t64 * S >= 512) break;
  }
  return best;
}

static int g_t144 = getenv("RGM_T144") ? atoi(getenv("RGM_T144")) : 31;   // which grids take the 128x144 tiles (bit mask, gemm2_launch)
static int g_co_min = getenv("RGM_CO_MIN_TILES") ? atoi(getenv("RGM_CO_MIN_TILES")) : 100;   // co-scheduled launches (GemmParams::co_sched)
static int g_co_kt = getenv("RGM_CO_KT") ? atoi(getenv("RGM_CO_KT")) : 36;
static int g_fuse_reduce_ln = getenv("RGM_FUSE_REDUCE_LN") ? atoi(getenv("RGM_FUSE_REDUCE_LN")) : 1;
static long long g_fused_reduce_ln_launches = 0;

// splitk_reduce_kernel with one wave per output row, followed by the next adaLN-LayerNorm of that row (GemmParams::ln_out).  The row
// arithmetic is splitk_reduce_kernel's (act 0, fp32 rows), the LayerNorm is ln_mod_kernel's (dit_kernels.hip) with the same lane <->
// column mapping and the same reduction order: what lands in ln_out is what the separate kernel writes.
template <int MAXV, int S>
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(const float* __restrict__ P, GemmParams p) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  splitk_reduce_ln_row<MAXV, S>(P, p, row);
}

// A and B in split-row format (see top).  tile: 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 5 = 256x128 (8 waves)
int gemm2_launch(const GemmParams& p, hipStream_t s) {
  RGM_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && (p.K & 31) == 0, "gemm2: bad shape M=%d N=%d K=%d (K%%32)", p.M, p.N, p.K);
  RGM_REQUIRE(p.aload == 0 || (p.Cin % 32 == 0 && p.K == 9 * p.Cin), "gemm2: implicit conv needs Cin%%32==0, K=9*Cin");
  RGM_REQUIRE(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 && (p.lda & 3) == 0 && (p.ldb & 3) == 0,
              "gemm2: operands must be 16-byte aligned with ld%%4==0");
  RGM_REQUIRE(!p.out_split || ((p.N & 31) == 0 && (p.ldc & 31) == 0), "gemm2: split-row output needs N%%32==0 (N=%d)", p.N);
  RGM_REQUIRE(!p.C2 || (p.out_split && !p.gate && !p.res && !p.stats && !p.aload && p.act < 3 && !p.ln_out && p.batch == 1 && (p.ldc2 & 31) == 0 &&
                        ((p.N | p.ldc) & 3) == 0 && (((uintptr_t)p.C | (uintptr_t)p.C2 | (uintptr_t)p.bias) & 15) == 0),
              "gemm2: the second output (C2) belongs to the plain split-row epilogue: bias + activation, 16-byte aligned rows, no gate / residual / statistics");
  RGM_REQUIRE(!p.stats || ((p.stats_gw == 4 || p.stats_gw == 8 || p.stats_gw == 16) && p.N % 64 == 0 && p.batch == 1 &&
                           ((p.tile == 0 && p.aload) || p.tile == 21 || p.tile == 22 || p.tile == 43 || p.tile == 44 || p.tile == 71 || p.tile == 72) &&
                           ((p.N | p.ldc | p.ldres | p.gate_ld | p.ldaux) & 3) == 0 &&
                           (((uintptr_t)p.C | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.gate | (uintptr_t)p.aux) & 15) == 0),
              "gemm2: GroupNorm partial sums need a fixed tile height (heuristic conv tiles: 128 rows; 71: 256; 72: 512), N%%64==0, group width 4/8/16 and 16-byte aligned rows");
  // ---- 256x256 tiles, one wave per SIMD (tile 71, PIPE 5): 380-420 TFLOP/s per full round of 256 workgroups against 300-360 for
  // the 128x128 kernels (tools/gemm_sweep.py: qkv at B = 16 86 us against 101, fc1 at n.B = 64 417 against 464), but ONE workgroup
  // per CU: a launch costs ceil(tiles / 256) rounds of ~(33 + 1.44 KT) us whatever the last round's fill.  Three ways to keep the
  // rounds full: the whole GEMM when its last round is at least 84 % full; K slices (as a batch, reduced by splitk_reduce_kernel)
  // when the tiles fill less than one round and K is long (fc2 at B = 16: 80 tiles x 3 slices); whole rounds of column tiles on
  // this kernel and the leftover columns through the heuristic again (fc1 at B = 16: 16 x 16 tiles + 512 columns on 128x64 tiles).
  int S = 1;
  int sk_tile = 44;
  // ---- 128x144 tiles (gemm144.hip, tile 81) at the samplers' small batches.  g_t144 bit 1: a grid that is ONE full round of them (224-256
  // tiles) where 128x128 tiles give more than a round -- fc1 at B = 4: 8 x 32 = 256 against 288, 43.7 -> 33.8 us in isolation (cold weights,
  // tools/gemm_sweep.py 100 181), forward 5.73 -> 5.47 ms; bit 8: K slices ON these tiles where even they leave most CUs idle -- fc2 at
  // B = 2 .. 4: 48-64 tiles x 4 slices, forward 4.39 -> 4.30 ms at B = 2, 5.47 -> 5.32 at B = 4.  Measured and NOT taken (same sweeps,
  // in the forward): fc2 at B = 16 unsliced (one round of 256 tiles, 108 against 128 us in isolation -- but the K-slice path's reduce also
  // writes the next LayerNorm: forward 12.17 -> 12.74 ms), proj at B = 16 (12.17 -> 12.08 and 13.11 -> 13.19 on two boxes: noise),
  // fc1 at B = 8 as two rounds (no change), fc2 at B = 8 as 128 tiles x 2 slices (8.18 -> 8.43 ms).
  if (p.tile == 0 && g_t144 && p.batch == 1 && p.M < 2048 && gemm144_supports(p)) {
    const long long t144 = (long long)cdiv(p.M, 128) * (p.N / 144);
    const long long t128 = (long long)cdiv(p.M, 128) * cdiv(p.N, 128);
    if ((g_t144 & 1) && t144 >= 224 && t144 <= 256 && t128 > 256) {
      GemmParams q = p;
      q.tile = 81;
      q.ln_out = nullptr;
      return gemm2_launch(q, s);
    }
  }
  // bits 2 / 4 (round 5): ONE round of 144-column tiles at M >= 2048 -- proj at B = 16 (bit 2: K-tiles < 72) and fc2 at B = 16 unsliced (bit 4:
  // its LayerNorm then runs as its own kernel).  With 8-row raster sweeps and the L2 prefetch (gemm144.hip) they are 43 / 115 us in the forward
  // against 49 (128x64 tiles) / 109 + 28 (256x256 K slices + reduce-LayerNorm): C2 12.83 -> 11.97 ms per step on one box.  Also for each
  // of two half batches in flight (co_sched; RGM_T144_CO=0 for A/B runs): C3 28.86 -> 28.27 ms, B = 32 forward 22.55 -> 22.25.
  static const int t144_co = getenv("RGM_T144_CO") ? atoi(getenv("RGM_T144_CO")) : 1;
  if (p.tile == 0 && (g_t144 & 6) && p.batch == 1 && p.M >= 2048 && (!p.co_sched || t144_co) && gemm144_supports(p)) {
    const long long t144 = (long long)cdiv(p.M, 128) * (p.N / 144);
    const int KT = p.K >> 5;
    if (t144 >= 224 && t144 <= 256 && (long long)cdiv(p.M, 256) * cdiv(p.N, 256) < 140 && ((KT < 72 && (g_t144 & 2)) || (KT >= 72 && (g_t144 & 4)))) {
      GemmParams q = p;
      q.tile = 81;
      q.ln_out = nullptr;
      return gemm2_launch(q, s);
    }
  }
  if (p.tile == 0 && g_t144 && p.batch == 1 && p.M < 2048 && gemm144_supports(p)) {
    const long long t144 = (long long)cdiv(p.M, 128) * (p.N / 144);
    const int KT = p.K >> 5;
    // bit 16 (round 6): the same for proj (36 K-tiles) when its reduce can write the second adaLN-LayerNorm of the block (ln_out) -- 64 tiles x 4
    // slices of 9 K-tiles at B = 4 instead of 144 tiles of 128 x 64 on 256 CUs: forward-step 5.03 -> 4.96 ms same box.  Full rounds only: 192 tiles
    // (B = 3: 48 x 4, B = 2: 32 x 6) measured level to behind (4.55 -> 4.66 ms at B = 3), profiles/r06_b4_proj_slices_ab.txt
    const bool long_k = (g_t144 & 8) && KT >= 72, proj_k = (g_t144 & 16) && KT >= 36 && KT < 72 && p.ln_out && p.gate && p.res && !p.co_sched;
    if ((long_k || proj_k) && p.sk_ws && t144 <= 64 && p.act == 0 && !p.out_split) {
      int best = 1;
      const int min_kt = long_k ? 18 : 9;
      for (int c = 2; c <= 8; ++c) {
        if (KT % c || KT / c < min_kt || t144 * c > 256) continue;
        if ((size_t)c * p.M * p.N * sizeof(float) + GEMM_SK_FLAG_BYTES > p.sk_ws_bytes) continue;
        best = c;
      }
      if (best > 1 && t144 * best >= (long_k ? 192 : 224)) {
        S = best;
        sk_tile = 81;
      }
    }
  }
  const bool big_ok = S == 1 && p.tile == 0 && g_big_tiles && !p.aload && p.batch == 1 && !p.stats && p.act < 3 && !p.aux && p.M >= 2048 &&
                      (!p.gate || p.rows_per_gate >= 32) &&     // the big tiles' gate / residual epilogue: at most two gate rows per 32-row slab

                      ((p.N | p.ldc | p.ldres | p.gate_ld) & 3) == 0 &&
                      (((uintptr_t)p.C | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.gate) & 15) == 0;
  // ---- 256x288 tiles (tile 74, round 6): N = 4608 = 16 x 288 -- fc1 of DiT-XL -- at M = 4096 is ONE round of 256 of them where the 256x256
  // tiling needs 288 (a 16 x 16 main launch + 512 columns on 128x64 tiles: 93 + 23 us in the C2 forward).  Taken when the grid is a FULL round
  // (250 .. 256 tiles: at B = 15, 240 tiles, it measured behind -- 11.54 against 11.48 ms), every tile is whole and the epilogue is the plain one;
  // also for each of two half batches in flight (B = 32: C3 26.95 -> 26.80 ms same box).  Same-box C2: 11.59 -> 11.34 ms
  // (profiles/r06_c2_tile288_ab.txt).  RGM_T288: bit 0 single stream, bit 1 beside a second stream's launches; 0 = off (A/B runs).
  static const int g_t288 = getenv("RGM_T288") ? atoi(getenv("RGM_T288")) : 3;
  if (big_ok && (p.co_sched ? (g_t288 & 2) : (g_t288 & 1)) && !p.gate && !p.res && !p.C2 && !p.ln_out && p.M % 256 == 0 && p.N % 288 == 0 &&
      (p.ldc & 7) == 0) {
    const long long t288 = (long long)(p.M / 256) * (p.N / 288), t256 = (long long)cdiv(p.M, 256) * cdiv(p.N, 256);
    if (t288 >= 250 && t288 <= 256 && t256 > 256) {
      GemmParams q = p;
      q.tile = 74;
      return gemm2_launch(q, s);
    }
  }
  // (256x224 tiles for qkv -- N = 3456 at M = 4096 as 16 x 16 = 256 tiles instead of 224 of 256x256 with 32 CUs idle -- were built, bit-identical, and
  // measured BEHIND: C2 11.83 -> 11.90 ms, C3 27.14 -> 27.36 same box, profiles/r06_c2_tile224_ab.txt; removed)
  if (big_ok && p.co_sched) {
    // Two half batches in flight (dit.hip): the other stream's kernels fill the CUs a partial round leaves, so what counts is the work per
    // tile, not the fill of the launch's last round.  Whole GEMM on 256x256 tiles from g_co_min tiles up (no column split: fc1's 288 tiles
    // at M = 4096 go out as ONE launch); long K as slices of about g_co_kt K-tiles.
    const int tm = cdiv(p.M, 256), tn = cdiv(p.N, 256), KT = p.K >> 5;
    const long long total = (long long)tm * tn;
    const double waste = (double)((long long)tn * 256 - p.N) / ((double)tn * 256);
    if (waste <= 0.12) {
      if (KT >= 72 && p.sk_ws && !p.C2 && total < 200) {
        int best = 1;
        for (int c = 2; c <= 8; ++c) {
          if (KT % c || KT / c < g_co_kt || total * c > 256) continue;
          if ((size_t)c * p.M * p.N * sizeof(float) + GEMM_SK_FLAG_BYTES > p.sk_ws_bytes) continue;
          best = c;
        }
        if (best > 1) { S = best; sk_tile = 71; }
      }
      if (S == 1 && total >= g_co_min) {
        GemmParams q = p;
        q.tile = 71;
        return gemm2_launch(q, s);
      }
    }
  }
  if (big_ok && !(p.co_sched && S > 1)) {
    const int tm = cdiv(p.M, 256), tn = cdiv(p.N, 256), KT = p.K >> 5;
    const long long total = (long long)tm * tn;
    const long long rounds = (total + 255) / 256;
    const double waste = (double)((long long)tn * 256 - p.N) / ((double)tn * 256);      // columns of the edge tiles beyond N
    // one (partial) round: from 140 tiles up the 256x256 kernel beats every 128-row kernel (B = 32 sweep, profiles/r03_tile71_sweep_b32.txt:
    // proj 160 tiles 74 us against 89 / 103, fc2 240 against 307 / 264); several rounds: the last one at least 84 % full
    // (fc1 at B = 8, 144 tiles, GELU + split output, cold weights: 74.4 us against 82.8-92.4 on the 128-row kernels, tools/fc1_mid_tiles.py;
    // qkv at B = 8, 112 tiles, stays on one round of 128x128)
    const bool fills = rounds == 1 ? total >= 140 : (double)total / (double)(rounds * 256) >= 0.84;
    if (fills && waste <= 0.11) {
      GemmParams q = p;
      q.tile = 71;
      return gemm2_launch(q, s);
    }
    if (total < 200 && KT >= 72 && p.sk_ws && !p.C2 && waste <= 0.12) {
      int best = 1;
      for (int c = 2; c <= 8; ++c) {
        if (KT % c || KT / c < 24 || total * c > 256) continue;
        if ((size_t)c * p.M * p.N * sizeof(float) + GEMM_SK_FLAG_BYTES > p.sk_ws_bytes) continue;
        best = c;
      }
      if (best > 1 && total * best >= 200) {
        S = best;
        sk_tile = 71;
      }
    }
    if (S == 1 && total > 256 && tm <= 256) {
      const int tn_main = (int)((total / 256) * 256 / tm);      // column tiles that fill whole rounds
      const long long main_tiles = (long long)tm * tn_main;
      const int n_main = tn_main * 256;
      if (tn_main >= 1 && n_main < p.N && (double)main_tiles / (double)(((main_tiles + 255) / 256) * 256) >= 0.9) {
        GemmParams pm = p, pr = p;
        pm.ln_out = pr.ln_out = nullptr;            // column blocks: no whole rows in either launch
        pm.N = n_main;
        pm.tile = 71;
        pr.N = p.N - n_main;
        pr.B = p.B + (long long)n_main * p.ldb;
        pr.C = p.C + n_main;                       // split-row output: a 256-column block is 256 floats wide as well
        if (p.C2) pr.C2 = p.C2 + n_main;
        if (p.bias) pr.bias = p.bias + n_main;
        if (p.res) pr.res = p.res + n_main;
        if (p.gate) pr.gate = p.gate + n_main;
        RGM_TRY(gemm2_launch(pm, s));
        return gemm2_launch(pr, s);
      }
    }
    // the same along M when N is narrow (proj / fc2, 5 column tiles, at C5's 112-window batches: 560 tiles = 2.19 rounds): whole
    // rounds of row tiles on this kernel, the leftover rows through the heuristic again.  A gate row never straddles the cut.
    if (S == 1 && total > 256 && tn <= 128) {
      const int tm_main = (int)((total / 256) * 256 / tn);
      const long long main_tiles = (long long)tm_main * tn;
      const long long rows_main = (long long)tm_main * 256;
      if (tm_main >= 1 && rows_main < p.M && (double)main_tiles / (double)(((main_tiles + 255) / 256) * 256) >= 0.9 &&
          (!p.gate || (p.rows_per_gate > 0 && rows_main % p.rows_per_gate == 0))) {
        GemmParams pm = p, pr = p;
        pm.ln_out = pr.ln_out = nullptr;            // the caller's fused LayerNorm needs ONE launch to own every row
        pm.M = (int)rows_main;
        pm.tile = 71;
        pr.M = p.M - (int)rows_main;
        pr.A = p.A + rows_main * p.lda;
        pr.C = p.C + rows_main * p.ldc;
        if (p.C2) pr.C2 = p.C2 + rows_main * p.ldc2;
        if (p.res) pr.res = p.res + rows_main * p.ldres;
        if (p.gate) pr.gate = p.gate + (rows_main / p.rows_per_gate) * p.gate_ld;
        RGM_TRY(gemm2_launch(pm, s));
        return gemm2_launch(pr, s);
      }
    }
  }
  if (S == 1 && !p.stats) S = splitk_factor(p);
  if (S == 1 && p.tile == 0 && p.sk_ws && !p.C2 && !p.stats && !p.aload && p.batch == 1 && p.act < 3 && (p.K >> 5) >= 96 &&
      ((p.N | p.ldc | p.ldres | p.gate_ld) & 3) == 0 && (((uintptr_t)p.C | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.gate) & 15) == 0) {
    // long-K GEMM on a grid that leaves the second 128x128 workgroup slot of most CUs empty (fc2 at B = 16: 288 tiles on 512
    // slots, 144 K-tiles each): K slices as a batch fill whole rounds.  Cost model in K-tile times, from tools/gemm_sweep.py
    // (fc2: 157 / 162 / 138 / 156 / 158 us at S = 1 / 2 / 3 / 4 / 6): rounds(S) * KT / S plus ~10 per slice for the partial
    // sums' round trip through the reduce kernel.
    const long long t128 = (long long)cdiv(p.M, 128) * cdiv(p.N, 128);
    const int KT = p.K >> 5;
    if (t128 >= 192 && t128 < 512) {
      double best = (double)KT;                    // unsplit: one (partial) round of KT K-tiles
      for (int c = 2; c <= 6; ++c) {
        if (KT % c || KT / c < 24) continue;
        if ((size_t)c * p.M * p.N * sizeof(float) + GEMM_SK_FLAG_BYTES > p.sk_ws_bytes) continue;
        const double cost = (double)((t128 * c + 511) / 512) * (KT / c) + 10.0 * c;
        if (cost < best * 0.95) { best = cost; S = c; sk_tile = 43; }
      }
    }
  }
  if (p.tile >= 200 && p.tile < 217 && p.sk_ws && !p.stats) {   // experiments: 200 + S slices on 128x128 tiles (tools/gemm_sweep.py)
    S = p.tile - 200;
    sk_tile = 43;
    RGM_REQUIRE(S >= 2 && (p.K >> 5) % S == 0, "gemm2: K-tiles %d do not divide into %d slices", p.K >> 5, S);
  }
  if (S > 1) {
    const size_t need = (size_t)S * p.M * p.N * sizeof(float);
    RGM_REQUIRE(p.sk_ws_bytes >= GEMM_SK_FLAG_BYTES + need, "gemm2: split-K scratch %zu bytes < %zu", p.sk_ws_bytes, GEMM_SK_FLAG_BYTES + need);
    float* partial = reinterpret_cast<float*>(static_cast<char*>(p.sk_ws) + GEMM_SK_FLAG_BYTES);
    GemmParams q = p;                     // the K slices as a batch: raw partial sums, no epilogue
    q.K = p.K / S;
    q.batch = S;
    q.sA = q.K; q.sB = q.K;
    q.C = partial; q.ldc = p.N; q.sC = (long long)p.M * p.N;
    q.sk_ws = nullptr; q.sk_ws_bytes = 0;
    q.bias = nullptr; q.act = 0; q.alpha = 1.0f; q.gate = nullptr; q.res = nullptr; q.out_split = 0;
    q.tile = sk_tile;
    RGM_TRY(gemm2_launch(q, s));
    // the reduce holds whole rows: with GemmParams::ln_out it also writes the next adaLN-LayerNorm of each row (same values as
    // ln_mod_kernel on the reduced rows; rgm_set_fuse_reduce_ln(0) / RGM_FUSE_REDUCE_LN=0 keep the two kernels apart: A/B runs, the parity test)
    if (g_fuse_reduce_ln && p.ln_out && p.ln_shift && p.ln_scale && p.act == 0 && !p.out_split && (p.N & 3) == 0 && p.N <= 1280 && p.ldc == p.N &&
        (p.ln_mod_ld & 3) == 0 && (((uintptr_t)p.ln_shift | (uintptr_t)p.ln_scale | (uintptr_t)p.ln_out) & 15) == 0 &&
        ((p.ldres | p.gate_ld) & 3) == 0 && (((uintptr_t)p.C | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.gate) & 15) == 0) {
      const dim3 grid((unsigned)cdiv(p.M, 4)), block(256);
      bool launched = true;
      switch (S) {           // the slice counts the heuristics produce for K = 4608 (144 K-tiles); anything else: two kernels
      case 2: hipLaunchKernelGGL((splitk_reduce_ln_kernel<5, 2>), grid, block, 0, s, (const float*)partial, p); break;
      case 3: hipLaunchKernelGGL((splitk_reduce_ln_kernel<5, 3>), grid, block, 0, s, (const float*)partial, p); break;
      case 4: hipLaunchKernelGGL((splitk_reduce_ln_kernel<5, 4>), grid, block, 0, s, (const float*)partial, p); break;
      case 6: hipLaunchKernelGGL((splitk_reduce_ln_kernel<5, 6>), grid, block, 0, s, (const float*)partial, p); break;
      case 8: hipLaunchKernelGGL((splitk_reduce_ln_kernel<5, 8>), grid, block, 0, s, (const float*)partial, p); break;
      default: launched = false;
      }
      if (launched) {
        RGM_LAUNCH_CHECK();
        if (p.ln_done) *p.ln_done = 1;
        ++g_fused_reduce_ln_launches;
        return RGM_OK;
      }
    }
    const long long total4 = (long long)p.M * (p.N >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, (const float*)partial, p, S);
    RGM_LAUNCH_CHECK();
    return RGM_OK;
  }
  int tile = p.tile;
  if (tile == 48 || tile == 49) {   // whole rounds of 128x128 tiles + the leftover columns on 128x64 (48) / 64x64 (49) in one launch.
    // Not the heuristic's choice: fc1 at B = 16 runs 129-135 us this way in isolation (tools/gemm_sweep.py; 145 / 152 us for the
    // one-shape 128x128 / 128x64 launches) but 162-166 us inside the forward, where the 128x64 launch holds 149 us
    // (tools/insitu_probe.py: operands from the Infinity Cache / HBM favour 3 workgroups per CU).
    const int nb_cols = dual_big_columns(p);
    RGM_REQUIRE(nb_cols > 0, "gemm2: tile %d (big + small tiles) does not apply to M=%d N=%d", tile, p.M, p.N);
    return tile == 49 ? launch_dual<64, 64>(p, nb_cols, s) : launch_dual<128, 64>(p, nb_cols, s);
  }
  if (tile == 0) {
    // tools/gemm_sweep.py on MI355X: cross-iteration pipeline (PIPE 3) at 128x128 (2 workgroups per CU) once the grid
    // fills at least one round of the chip, at 128x64 (3 per CU) below that; grids that do not even fill the CUs
    // once (B = 2: M = 512 rows) are latency-bound and take the loader/consumer kernel with its 3-stage ring
    const long long t128 = (long long)cdiv(p.M, 128) * cdiv(p.N, 128) * p.batch;
    const long long t64 = (long long)cdiv(p.M, 128) * cdiv(p.N, 64) * p.batch;
    // fraction of the CU-rounds a grid fills (512 resident 128x128 workgroups, 768 of 128x64): at B = 16 fc1 has
    // 1152 / 2304 tiles = 2.25 (75 %) / 3.0 (100 %) rounds, qkv 864 / 1728 = 1.69 (84 %) / 2.25 (75 %)
    auto fill = [](long long tiles, long long slots) { return (double)tiles / (double)(((tiles + slots - 1) / slots) * slots); };
    if (t128 >= 512 && fill(t128, 512) * 1.15 >= fill(t64, 768)) tile = 43;   // 128x128 is ~15 % ahead per tile (B = 32 sweep)
    else if (p.aload || t128 > 512) tile = 44;
    // one round of 128x128 at two workgroups per CU where 128x64 tiles would need two (qkv at B = 8: 432 tiles, 51 us against 77 us)
    // (in situ, 288 / 576 tiles both ways: fc1 at B = 4, wide and short, 51.6 us on 128x128 against 58.8; proj at B = 16, tall, 54.1 against 52.2)
    else if (t128 > 256) tile = (t64 > 768 || p.N >= 4 * p.M) ? 43 : 44;
    // at most one 128x128 workgroup per CU (B <= 4): nothing else on the CU hides the HBM latency of the weights -> loader/consumer
    // kernels with 4-6 stage rings, the largest tile that still gives every CU one (SWEEP_SHAPES=small SWEEP_COLD=1 sweep)
    else if (t64 > 256) tile = 54;
    else if (t64 > 128) tile = 56;
    else tile = 57;
    // implicit-conv loader: the per-piece pixel bookkeeping pushes the cross-iteration pipeline at 128x128 over 256
    // registers (one wave per SIMD) -> the single-set pipeline (PIPE 1) there
    if (p.aload && tile == 43) tile = 21;
  }
  switch (tile) {
    case 1: return launch2<128, 128, 2, 2>(p, s, 1);
    case 2: return launch2<128, 64, 2, 2>(p, s, 2);
    case 3: return launch2<64, 64, 2, 2>(p, s, 3);
    case 5: return launch2<256, 128, 4, 2>(p, s, 5);
    // software-pipelined bodies (PIPE): fragments double-buffered in registers, DMA pieces spread between the MFMAs
    case 21: return launch2<128, 128, 2, 2, 2, 1>(p, s, 21);
    case 22: return launch2<128, 64, 2, 2, 2, 1>(p, s, 22);
    // cross-iteration register pipeline (PIPE == 3), 3- and 2-stage rings
    case 43: return launch2<128, 128, 2, 2, 2, 3>(p, s, 43);   // 64 KB: 2 per CU
    case 44: return launch2<128, 64, 2, 2, 2, 3>(p, s, 44);    // 48 KB: 3 per CU
    case 45: return launch2<256, 128, 4, 2, 2, 3>(p, s, 45);   // 96 KB, 8 waves
    case 46: return launch2<64, 64, 2, 2, 3, 3>(p, s, 46);     // 48 KB: 3 per CU
    // loader/consumer split (PIPE == 4): NW MFMA waves + NW DMA waves, 3-stage ring
    case 51: return launch2<128, 128, 2, 2, 3, 4>(p, s, 51);   // 96 KB: 1 per CU
    case 52: return launch2<128, 64, 2, 2, 3, 4>(p, s, 52);    // 72 KB: 2 per CU
    // deep rings for grids of at most one workgroup per CU
    case 53: return launch2<128, 64, 2, 2, 6, 4>(p, s, 53);    // 144 KB
    case 54: return launch2<128, 128, 2, 2, 4, 4>(p, s, 54);   // 128 KB
    case 55: return launch2<128, 128, 2, 2, 5, 4>(p, s, 55);   // 160 KB
    case 56: return launch2<128, 64, 2, 2, 4, 4>(p, s, 56);    // 96 KB
    case 57: return launch2<64, 64, 2, 2, 6, 4>(p, s, 57);     // 96 KB
    case 58: return launch2<64, 64, 2, 2, 3, 4>(p, s, 58);     // 48 KB: 3 per CU
    // one wave per SIMD, 128x128 per wave (PIPE == 5)
    case 71: return launch2<256, 256, 2, 2, 2, 5>(p, s, 71);   // 128 KB: 1 per CU, 512 registers
    case 73: return launch2<128, 256, 1, 4, 2, 5>(p, s, 73);   // 96 KB: 128x64 wave tiles, for M of a few thousand rows (B = 8: the shapes B = 16 has at 256 rows)
    case 81: return gemm144_launch(p, s);                     // gemm144.hip: 128x144 tiles on 16x16x32 MFMAs (N % 144 == 0)
    case 74: return launch2<256, 288, 4, 1, 2, 5>(p, s, 74);   // 145 KB: 64x288 wave tiles (288 accumulator registers) -- fc1 of DiT-XL at M = 4096 as ONE round of 256 tiles
    case 72: return launch2<512, 128, 4, 1, 2, 5>(p, s, 72);   // 160 KB (all of the LDS): the same 128x128 wave tiles for N = 128 (VAE convs at 128 channels)
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
  typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 hi, lo;
  hi[0] = (split_t)v.x; hi[1] = (split_t)v.y; hi[2] = (split_t)v.z; hi[3] = (split_t)v.w;
  lo[0] = (split_t)(v.x - (float)hi[0]); lo[1] = (split_t)(v.y - (float)hi[1]);
  lo[2] = (split_t)(v.z - (float)hi[2]); lo[3] = (split_t)(v.w - (float)hi[3]);
  split_t* rowp = reinterpret_cast<split_t*>(out + row * ld_out);
  *reinterpret_cast<bf16x4*>(rowp + split_idx(c)) = hi;
  *reinterpret_cast<bf16x4*>(rowp + split_idx(c) + 32) = lo;
}

int split_rows_launch(const float* x, float* out, long long rows, int K, int ld_in, int ld_out, hipStream_t s) {
  RGM_REQUIRE(x && out && rows > 0 && K > 0 && (K & 3) == 0 && x != out, "split_rows: bad arguments (out-of-place, K%%4==0)");
  const long long total = rows * (K >> 2);
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, rows, K, ld_in, ld_out);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

size_t gemm2_scratch_bytes(int M, int N) {
  // deterministic split-K: up to 8 slices of a grid below 384 128x64 tiles (splitk_factor), never less than the 32 MiB the K-sliced
  // fc2 of a batch of 16 takes with room to spare.  The first GEMM_SK_FLAG_BYTES are reserved (the partial sums start behind them).
  const long long rows = (long long)cdiv(384, cdiv(N, 64)) * 128;
  const size_t splitk = (size_t)8 * (size_t)(M < rows ? M : rows) * N * sizeof(float);
  const size_t floor_ = (size_t)32 << 20;
  return GEMM_SK_FLAG_BYTES + (splitk > floor_ ? splitk : floor_);
}

void gemm2_prof(bool on) { g2_prof_on = on; }
// bracket one launch with two events on its stream (no-ops while profiling is off)
int gemm2_prof_begin(int id, double flops, hipStream_t s) {
  if (!g2_prof_on) return -1;
  Prof2 rec{};
  if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) return -1;
  rec.tile = id;
  rec.flops = flops;
  rec.bytes = 0.0;
  (void)hipEventRecord(rec.a, s);
  g2_prof.push_back(rec);
  return (int)g2_prof.size() - 1;
}
void gemm2_prof_set_bytes(int idx, double bytes) {      // algorithmic bytes of the launch (rgm_prof_bytes)
  if (idx >= 0 && idx < (int)g2_prof.size()) g2_prof[idx].bytes = bytes;
}
void gemm2_prof_end(int idx, hipStream_t s) {
  if (idx >= 0 && idx < (int)g2_prof.size()) (void)hipEventRecord(g2_prof[idx].b, s);
}
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

// algorithmic bytes summed over the recorded launches of `kernel` (0 for kernels that do not record them)
double gemm2_prof_bytes(int kernel) {
  double b = 0.0;
  for (auto& r : g2_prof)
    if (r.tile == kernel) b += r.bytes;
  return b;
}

// raw per-launch records of the pre-split kernels, in launch order (tools/insitu_probe.py): kernel id, milliseconds, FLOPs
int gemm2_prof_dump(int cap, int* ids, double* ms, double* flops) {
  int n = 0;
  for (auto& r : g2_prof) {
    if (n >= cap) break;
    if (hipEventSynchronize(r.b) != hipSuccess) break;
    float e = 0.f;
    if (hipEventElapsedTime(&e, r.a, r.b) != hipSuccess) break;
    ids[n] = r.tile;
    ms[n] = e;
    flops[n] = r.flops;
    ++n;
  }
  return n;
}

}  // namespace rgm

extern "C" int rgm_set_big_tiles(int mode, int min_tiles) {
  RGM_REQUIRE((mode == 0 || mode == 1) && min_tiles >= 1, "set_big_tiles: mode %d (0 / 1), min_tiles %d (>= 1)", mode, min_tiles);
  rgm::g_big_tiles = mode;
  rgm::g_big_min_tiles = min_tiles;
  return RGM_OK;
}

// K-sliced GEMMs with GemmParams::ln_out (fc2 of a DiT block): 1 = the reduce kernel also writes the next LayerNorm (default), 0 = two kernels
extern "C" int rgm_set_fuse_reduce_ln(int on) {
  RGM_REQUIRE(on == 0 || on == 1, "set_fuse_reduce_ln: %d (0 / 1)", on);
  rgm::g_fuse_reduce_ln = on;
  return RGM_OK;
}
extern "C" long long rgm_fused_reduce_ln_launches(void) { return rgm::g_fused_reduce_ln_launches; }

// Sum of the algorithmic HBM bytes (operands read once, output written once) of the recorded launches of a pre-split kernel id.
extern "C" double rgm_prof_bytes(int kernel) { return rgm::gemm2_prof_bytes(kernel); }

// Profiling aid: the per-launch records behind rgm_prof_report for the pre-split GEMM kernels, in launch order; returns the count.
extern "C" int rgm_prof_dump(int cap, int* ids, double* ms, double* flops) {
  if (!ids || !ms || !flops || cap <= 0) return 0;
  return rgm::gemm2_prof_dump(cap, ids, ms, flops);
}

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

// Timing instrumentation (tools only): mode 1 = use the s_memtime-stamped kernels from now on, mode 2 = copy the
// 8 waves x 8 counters (cycles summed over the K loop of the middle workgroup) to out64 (64 entries), mode 0 = off.
extern "C" int rgm_gemm2_dbg(int mode, long long* out64) {
  using namespace rgm;
  if (mode == 1) {
#ifndef RGM_GEMM2_STAMPS
    RGM_REQUIRE(false, "gemm2_dbg: library built without -DRGM_GEMM2_STAMPS (make EXTRA=-DRGM_GEMM2_STAMPS)");
#endif
    if (!g_dbg) RGM_CHECK_HIP(hipMalloc(&g_dbg, 64 * sizeof(long long)));
    RGM_CHECK_HIP(hipMemset(g_dbg, 0, 64 * sizeof(long long)));
  } else if (mode == 2) {
    RGM_REQUIRE(g_dbg && out64, "gemm2_dbg: not enabled");
    RGM_CHECK_HIP(hipDeviceSynchronize());
    RGM_CHECK_HIP(hipMemcpy(out64, g_dbg, 64 * sizeof(long long), hipMemcpyDeviceToHost));
  } else {
    if (g_dbg) (void)hipFree(g_dbg);
    g_dbg = nullptr;
  }
  return RGM_OK;
}

// Strided variants: rows of A / B / C / the split image may be padded (ld in elements, ld % 32 == 0 for split rows).
// A row stride that is a multiple of 2 KiB puts one K-slice of many rows on few L2 channels; +32 elements avoids it.
extern "C" int rgm_split_rows_ld(const float* x, int ld_in, float* out, int ld_out, int64_t rows, int K, void* stream) {
  return rgm::split_rows_launch(x, out, rows, K, ld_in, ld_out, (hipStream_t)stream);
}
extern "C" int rgm_gemm_split_ld(const float* A_split, int lda, const float* B_split, int ldb, float* C, int ldc, int M, int N, int K,
                                 const float* bias, int act, int tile, int out_split, void* stream) {
  RGM_REQUIRE(A_split && B_split && C, "gemm_split_ld: null operand");
  rgm::GemmParams g;
  g.A = A_split; g.lda = lda; g.B = B_split; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act; g.tile = tile; g.out_split = out_split;
  return rgm::gemm2_launch(g, (hipStream_t)stream);
}

// Bytes of scratch a caller should provide for pre-split GEMMs of up to M rows and N columns so that the heuristic may K-slice them
// (deterministic split-K: S partial results + one fixed-order reduce kernel).  16-byte aligned caller memory, no initialisation needed.
extern "C" size_t rgm_gemm_scratch_bytes(int M, int N) { return rgm::gemm2_scratch_bytes(M, N); }

// rgm_gemm_split with caller-provided split-K scratch: tile 0 lets the heuristic pick (K slices when they pay).
extern "C" int rgm_gemm_split_ws(const float* A_split, const float* B_split, float* C, int M, int N, int K, const float* bias, int act,
                                 int tile, int out_split, void* ws, size_t ws_bytes, void* stream) {
  RGM_REQUIRE(A_split && B_split && C && ws, "gemm_split_ws: null operand");
  rgm::GemmParams g;
  g.A = A_split; g.lda = K; g.B = B_split; g.ldb = K; g.C = C; g.ldc = N;
  g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act; g.tile = tile; g.out_split = out_split;
  g.sk_ws = ws; g.sk_ws_bytes = ws_bytes;
  return rgm::gemm2_launch(g, (hipStream_t)stream);
}

// The general entry of the pre-split GEMM family: every fused epilogue the DiT block uses (alpha, bias, activation, per-sample
// adaLN gate, residual that may alias C, split-row output), explicit row strides, explicit tile (0 = heuristic, 7x = the
// one-wave-per-SIMD kernels) and the caller's split-K scratch (may be NULL: decompositions that need it are then not chosen; forced
// ones fail).  What guided_diffusion/dit.py:332-336 computes per block, in one launch.
extern "C" int rgm_gemm_split_epi(const float* A_split, int lda, const float* B_split, int ldb, float* C, int ldc, int M, int N, int K,
                                  const float* bias, int act, float alpha, const float* gate, int gate_ld, int rows_per_gate,
                                  const float* res, int ldres, int tile, int out_split, void* ws, size_t ws_bytes, void* stream) {
  RGM_REQUIRE(A_split && B_split && C, "gemm_split_epi: null operand");
  rgm::GemmParams g;
  g.A = A_split; g.lda = lda; g.B = B_split; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act; g.alpha = alpha; g.tile = tile; g.out_split = out_split;
  g.gate = gate; g.gate_ld = gate_ld; g.rows_per_gate = rows_per_gate > 0 ? rows_per_gate : 1;
  g.res = res; g.ldres = ldres;
  g.sk_ws = ws; g.sk_ws_bytes = ws ? ws_bytes : 0;
  return rgm::gemm2_launch(g, (hipStream_t)stream);
}
