// gemm3.hip -- persistent loader/consumer bf16x3 GEMM on PRE-SPLIT operands (dense A (M,K), B (N,K), split rows).
//
// Same contraction, numerics (a*b ~= al*bh + ah*bl + ah*bh on v_mfma_f32_32x32x16_bf16, fp32 accumulate, that term
// order per accumulator) and epilogues as gemm2.hip; what changes is who does what and for how long:
//
//   * ONE workgroup per CU, alive for the whole launch: NW "consumer" waves (MFMA + ds_read only) and NW "loader"
//     waves (global_load_lds only), one of each per SIMD.  A wave can issue an LDS-DMA piece only every ~70-80
//     cycles (tools/gemm_stamp.py; 25 GB/s per wave), which in the symmetric kernels sat between the MFMAs of the
//     same wave: 8 x 70 = 560 cycles beside 768 MFMA cycles per K-tile at 128x128.
//   * the K-tiles of ALL the workgroup's output tiles form one stream through an S-stage LDS ring (S = 4 at 128x128: all 160 KB): while the consumers
//     multiply K-tile g (from registers) and run a tile's epilogue, the loaders already have g+1 landed and g+2 ..
//     g+S-1 in flight -- of the NEXT output tile when g is a tile's last.  No per-tile prologue (a first DMA round trip, ~4 k
//     cycles), no workgroup launch per tile, and the epilogue's global stores overlap the next tile's loads.
//   * one s_barrier per K-tile is the only synchronisation: barrier g says "K-tile g+1 is in LDS and every consumer has
//     finished reading K-tile g", which is exactly what lets the loaders overwrite g's stage with g+3... g+2's data.
//   * the epilogue stages through its own LDS slab (the ring holds live data), then bias / act / gate / residual and
//     16-byte stores per lane as in gemm2.hip.
#include <stdlib.h>
#include <vector>
#include "common.h"

namespace rgm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16_g3(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16,
                                   0, 0);
}

template <int BM, int BN, int WM, int WN, int S>
__global__ __launch_bounds__(WM* WN * 128) void gemm3_kernel(GemmParams p, const char* __restrict__ zero_page, int tiles_m,
                                                             int tiles_n, int exp) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int STAGE = (BM + BN) * 128;        // bytes per ring stage: BM A rows then BN B rows, one 128-B line each
  constexpr int SEGS = (BM + BN) / 8;           // 1-KiB DMA pieces per stage (8 rows x 128 B)
  constexpr int LSEG = SEGS / NW;               // pieces per loader wave and K-tile
  constexpr int WCOLS = TN * 32;                // columns of a consumer wave's sub-tile
  static_assert(SEGS % NW == 0 && (S - 2) * LSEG <= 63 && S >= 3 && S <= 6, "pieces must divide over the loaders and fit vmcnt");
  extern __shared__ __attribute__((aligned(16))) char ring[];   // S stages, then NW epilogue slabs of 32 x WCOLS floats

  const int ntiles = tiles_m * tiles_n;
  const int z = blockIdx.z;
  const int KT = p.K >> 5;
  int my_tiles = 0;
  if ((int)blockIdx.x < ntiles) my_tiles = (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
  const int G = my_tiles * KT;                  // K-tiles this workgroup streams
  if (G == 0) return;

  // linear tile id -> (m0, n0): the XCD-contiguous grouped raster of gemm.hip / gemm2.hip (workgroup b lives on XCD b % 8
  // and, with gridDim.x a multiple of 8, so do all its tiles)
  auto coords = [&](int t, int& m0, int& n0) {
    const int xcd = t & 7, loc = t >> 3, q = ntiles >> 3, r = ntiles & 7;
    const int sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int grp = sid / per_group;
    const int first_m = grp * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int in_g = sid - grp * per_group;
    m0 = (first_m + in_g % gsz) * BM;
    n0 = (in_g / gsz) * BN;
  };

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const char* Ab = reinterpret_cast<const char*>(p.A + (long long)z * p.sA);
  const char* Bb = reinterpret_cast<const char*>(p.B + (long long)z * p.sB);

  if (wave >= NW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wave - NW;
    const int r8 = lane >> 3;
    const char* src[LSEG];
    int inc[LSEG];
    int lt = blockIdx.x, lkt = 0;
    auto aim = [&](int t) {                     // per-lane source of every piece at k = 0 of output tile t
      int m0, n0;
      coords(t, m0, n0);
#pragma unroll
      for (int i = 0; i < LSEG; ++i) {
        const int sgm = lw + i * NW;
        const bool isA = sgm < BM / 8;
        const int row_l = sgm * 8 + r8;                          // LDS row within the stage
        const int cs = ((lane & 7) ^ ((row_l >> 1) & 7)) << 4;   // swizzle on the source side (see gemm2.hip)
        const int row = isA ? m0 + row_l : n0 + row_l - BM;
        const bool ok = isA ? row < p.M : row < p.N;
        src[i] = ok ? (isA ? Ab + (long long)row * p.lda * 4 : Bb + (long long)row * p.ldb * 4) + cs : zero_page + cs;
        inc[i] = ok ? 128 : 0;
      }
    };
    auto issue = [&](char* dst) {               // one K-tile of this wave's pieces, then advance the stream
      static_for<0, LSEG>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        dma16_g3(src[i], dst + (lw + i * NW) * 1024);
        src[i] += inc[i];
      });
      if (++lkt == KT) {
        lkt = 0;
        lt += gridDim.x;
        if (lt < ntiles) aim(lt);
      }
    };
    // wait until at most `r` K-tiles' worth of this wave's pieces are still in flight (r is small and uniform)
    auto wait_tiles = [&](int r) {
      switch (r) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LSEG) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LSEG <= 63 ? 2 * LSEG : 63) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LSEG <= 63 ? 3 * LSEG : 63) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LSEG <= 63 ? 4 * LSEG : 63) : "memory"); break;
      }
    };
    aim(lt);
    // stage of K-tile g is g % S.  Prologue: K-tiles 0 .. S-2 in flight, wait for K-tile 0.
    int issued = 0;                              // K-tiles issued so far
    for (; issued < S - 1 && issued < G; ++issued) issue(ring + issued * STAGE);
    wait_tiles(issued - 1);
    __builtin_amdgcn_s_barrier();               // barrier P: K-tile 0 is in LDS
    int sn = S - 1;                              // stage of the next K-tile to issue
    for (int g = 0; g + 1 < G; ++g) {
      if (issued < G) {                          // K-tile g+S-1 into the stage K-tile g-1 left: free since barrier g-1
        issue(ring + sn * STAGE);
        ++issued;
        sn = sn + 1 == S ? 0 : sn + 1;
      }
      wait_tiles(issued - (g + 2));              // own share of K-tile g+1 has landed; younger tiles may still fly
      __builtin_amdgcn_s_barrier();             // barrier g
    }
    return;
  }

  // -------------------------------------------------------------------- consumer waves
  const int wr = wave / WN, wc = wave - wr * WN;
  const int arow0 = wr * TM * 32, bcol0 = wc * TN * 32;
  const int rq = (l31 >> 1) & 7;                 // read-side swizzle (tile row offsets are multiples of 16)
  constexpr int NM = TM * TN * 3;                // MFMAs per k16 step
  constexpr int NRD = 2 * (TM + TN) * 2;         // ds_read_b128 per K-tile
  constexpr int RPM = (NRD + NM - 1) / NM;       // reads dropped into one MFMA gap
  struct Frags {
    bf16x8 a[2][TM][2], b[2][TN][2];             // [k16 step][fragment][hi, lo]
  };
  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  auto read_one = [&](Frags& f, const char* As, auto jc) {
    constexpr int j = decltype(jc)::value;       // st-major; A fragments (hi, lo) then B fragments (hi, lo)
    constexpr int st = j / (2 * (TM + TN)), r = j % (2 * (TM + TN));
    constexpr int fi = r / 2, lo = r % 2;
    const int chunk = ((4 * lo + 2 * st + hh) ^ rq) << 4;
    if constexpr (fi < TM) {
      f.a[st][fi][lo] = *reinterpret_cast<const bf16x8*>(As + (arow0 + fi * 32 + l31) * 128 + chunk);
    } else {
      f.b[st][fi - TM][lo] = *reinterpret_cast<const bf16x8*>(As + BM * 128 + (bcol0 + (fi - TM) * 32 + l31) * 128 + chunk);
    }
  };
  auto mfma_step = [&](const Frags& f, auto stc, bool prefetch, Frags& nxt, const char* As_next) {
    constexpr int st = decltype(stc)::value;
    static_for<0, NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int t = m / (TM * TN), im = (m % (TM * TN)) / TN, in = m % TN;
      acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[st][im][t == 0 ? 1 : 0], f.b[st][in][t == 1 ? 1 : 0], acc[im][in], 0, 0, 0);
      if constexpr (st == 1) {
        if (prefetch) {
          static_for<0, RPM>([&](auto rc) {
            constexpr int j = m * RPM + decltype(rc)::value;
            if constexpr (j < NRD) read_one(nxt, As_next, std::integral_constant<int, j>{});
          });
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- epilogue of one output tile (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
  float* __restrict__ Cb = p.C + (long long)z * p.sC;
  const float* resb = p.res ? p.res + (long long)z * p.sRes : nullptr;
  const float* biasb = p.bias ? p.bias + (long long)z * p.sBias : nullptr;
  const bool vec = ((p.N | p.ldc | p.ldres | p.gate_ld) & 3) == 0 &&
                   (((uintptr_t)Cb | (uintptr_t)resb | (uintptr_t)biasb | (uintptr_t)p.gate) & 15) == 0;
  float* stg = reinterpret_cast<float*>(ring + S * STAGE) + wave * (32 * WCOLS);
  auto finish = [&](float v, int row, int col) -> float {   // scalar tail of the epilogue (fallback path)
    if (p.act == 1) v = silu_f(v);
    else if (p.act == 2) v = gelu_tanh_fast_f(v);
    if (p.gate) v *= p.gate[(long long)(row / p.rows_per_gate) * p.gate_ld + col];
    if (resb) v += resb[(long long)row * p.ldres + col];
    return v;
  };
  auto epilogue = [&](int m0, int n0) {
    if (vec) {
      constexpr int LPR = WCOLS / 4, RPI = 64 / LPR;   // lanes per row, rows per wave-instruction
      const int lr = lane / LPR, lc = (lane % LPR) * 4;
      const int col = n0 + bcol0 + lc;
      const bool col_ok = col < p.N;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (biasb && col_ok) bv = *reinterpret_cast<const float4*>(biasb + col);
      static_for<0, TM>([&](auto im_c) {
        constexpr int im = decltype(im_c)::value;
        static_for<0, TN>([&](auto in_c) {
          constexpr int in = decltype(in_c)::value;
#pragma unroll
          for (int e = 0; e < 16; ++e) stg[((e & 3) + 8 * (e >> 2) + 4 * hh) * WCOLS + in * 32 + l31] = acc[im][in][e];
        });
#pragma unroll
        for (int j = 0; j < 32 / RPI; ++j) {
          const int r = j * RPI + lr;
          const int row = m0 + arow0 + im * 32 + r;
          const float4 a4 = *reinterpret_cast<const float4*>(stg + r * WCOLS + lc);   // same wave wrote it: LDS ops are in order
          if (row < p.M && col_ok) {
            float v[4] = {a4.x * p.alpha + bv.x, a4.y * p.alpha + bv.y, a4.z * p.alpha + bv.z, a4.w * p.alpha + bv.w};
            if (p.act == 1) {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) v[q4] = silu_f(v[q4]);
            } else if (p.act == 2) {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) v[q4] = gelu_tanh_fast_f(v[q4]);
            }
            if (p.gate) {
              const float4 g4 = *reinterpret_cast<const float4*>(p.gate + (long long)(row / p.rows_per_gate) * p.gate_ld + col);
              v[0] *= g4.x; v[1] *= g4.y; v[2] *= g4.z; v[3] *= g4.w;
            }
            if (resb) {
              const float4 r4 = *reinterpret_cast<const float4*>(resb + (long long)row * p.ldres + col);
              v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if (p.out_split) {   // split-row output (common.h split_idx): 4 hi then, 32 further, 4 lo
              typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
              bf16x4 hi, lo;
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                hi[q4] = (__bf16)v[q4];
                lo[q4] = (__bf16)(v[q4] - (float)hi[q4]);
              }
              __bf16* rowp = reinterpret_cast<__bf16*>(Cb + (long long)row * p.ldc);
              *reinterpret_cast<bf16x4*>(rowp + split_idx(col)) = hi;
              *reinterpret_cast<bf16x4*>(rowp + split_idx(col) + 32) = lo;
            } else {
              *reinterpret_cast<float4*>(Cb + (long long)row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        }
      });
    } else {
      static_for<0, TM>([&](auto im_c) {
        static_for<0, TN>([&](auto in_c) {
          constexpr int im = decltype(im_c)::value, in = decltype(in_c)::value;
          const int col = n0 + bcol0 + in * 32 + l31;
          if (col < p.N) {
            const float bv = biasb ? biasb[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = m0 + arow0 + im * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
              if (row < p.M) {
                const float v = finish(acc[im][in][e] * p.alpha + bv, row, col);
                if (p.out_split) {
                  __bf16* rowp = reinterpret_cast<__bf16*>(Cb + (long long)row * p.ldc);
                  const __bf16 hi = (__bf16)v;
                  rowp[split_idx(col)] = hi;
                  rowp[split_idx(col) + 32] = (__bf16)(v - (float)hi);
                } else {
                  Cb[(long long)row * p.ldc + col] = v;
                }
              }
            }
          }
        });
      });
    }
  };

  Frags f0, f1;
  int t = blockIdx.x, m0, n0, kt = 0, s1 = 1;
  coords(t, m0, n0);
  zero_acc();
  __builtin_amdgcn_s_barrier();                  // barrier P (loaders: K-tile 0 landed)
  static_for<0, NRD>([&](auto jc) { read_one(f0, ring, jc); });
  auto iter = [&](Frags& cur, Frags& nxt, int g) {
    const bool more = g + 1 < G;
    if (exp != 1) mfma_step(cur, std::integral_constant<int, 0>{}, false, nxt, nullptr);
    if (more) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();              // barrier g: K-tile g+1 in LDS, everyone is done reading K-tile g
    }
    __builtin_amdgcn_sched_barrier(0);
    if (exp != 1) mfma_step(cur, std::integral_constant<int, 1>{}, more, nxt, ring + s1 * STAGE);
    s1 = s1 + 1 == S ? 0 : s1 + 1;
    if (++kt == KT) {                            // the tile is complete; the loaders are already two K-tiles into the next
      epilogue(m0, n0);
      kt = 0;
      t += gridDim.x;
      if (t < ntiles) {
        coords(t, m0, n0);
        zero_acc();
      }
    }
  };
  for (int g = 0; g < G; g += 2) {
    iter(f0, f1, g);
    if (g + 1 < G) iter(f1, f0, g + 1);
  }
}

static char* g3_zero_page = nullptr;
static int g3_num_cu = 0;

template <int BM, int BN, int WM, int WN, int S>
static int launch3(const GemmParams& p, hipStream_t s, int tile_id) {
  if (!g3_zero_page) {
    RGM_CHECK_HIP(hipMalloc(&g3_zero_page, 4096));
    RGM_CHECK_HIP(hipMemset(g3_zero_page, 0, 4096));
    int dev = 0;
    hipDeviceProp_t prop;
    RGM_CHECK_HIP(hipGetDevice(&dev));
    RGM_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    g3_num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  constexpr int NW = WM * WN, TN = BN / (WN * 32);
  const int tm = cdiv(p.M, BM), tn = cdiv(p.N, BN);
  const size_t lds = (size_t)S * (BM + BN) * 128 + (size_t)NW * 32 * TN * 32 * 4;
  auto k = gemm3_kernel<BM, BN, WM, WN, S>;
  static bool attr = false;
  if (!attr) {
    RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  int gx = tm * tn < g3_num_cu ? tm * tn : g3_num_cu;      // one persistent workgroup per CU
  if (gx >= 8) gx &= ~7;                                     // keep a workgroup's tiles on its XCD (coords)
  dim3 grid(gx, 1, p.batch), block(NW * 128);
  const int rec = gemm2_prof_begin(40 + tile_id, 2.0 * p.M * (double)p.N * p.K * p.batch, s);
  static const int g3_exp = RGM_EXP_ENV("RGM_GEMM3_EXP");   // timing experiment (common.h): 1 = DMA + barriers only
  hipLaunchKernelGGL(k, grid, block, lds, s, p, (const char*)g3_zero_page, tm, tn, g3_exp);
  RGM_LAUNCH_CHECK();
  gemm2_prof_end(rec, s);
  return RGM_OK;
}

// tile: 61 = 128x128, 62 = 128x64 (both 4 MFMA + 4 DMA waves)
int gemm3_launch(const GemmParams& p, hipStream_t s, int tile) {
  RGM_REQUIRE(p.aload == 0, "gemm3: dense operands only");
  switch (tile) {
    case 61: return launch3<128, 128, 2, 2, 4>(p, s, 61);   // 4 x 32 KB ring + 32 KB epilogue slabs = all 160 KB
    case 62: return launch3<128, 64, 2, 2, 5>(p, s, 62);    // 5 x 24 KB + 16 KB
    case 63: return launch3<128, 128, 2, 2, 3>(p, s, 63);
    default: break;
  }
  set_error("gemm3: unknown tile %d", tile);
  return RGM_ERR_INVALID;
}

}  // namespace rgm
