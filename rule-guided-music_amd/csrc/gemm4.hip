// gemm4.hip -- persistent STREAM-K bf16x3 GEMM on PRE-SPLIT operands (dense A (M,K), B (N,K), split rows; gemm2.hip's format).
//
// Same contraction, numerics (a*b ~= al*bh + ah*bl + ah*bh on v_mfma_f32_32x32x16_bf16, fp32 accumulate, that term order
// per accumulator) and epilogues as gemm2.hip's 128x128 cross-iteration pipeline (tile 43), which it reuses as its K loop.
// What changes is the decomposition.  At the sampler's batch (B = 16 -> M = 4096 rows) the nn.Linear GEMMs of a DiT block
// (guided_diffusion/dit.py:263-288 qkv / proj, timm Mlp fc1 / fc2) are 864 .. 1152 tiles of 128x128 on 512 resident
// workgroups: 1.7 .. 2.25 CU-rounds that cost 2 .. 2.8 round-times (profiles/r01: 0.81x), and every tile pays its own
// prologue (a first DMA round trip) and launch slot.  Here:
//
//   * the grid is exactly the resident set (2 workgroups per CU), alive for the whole launch;
//   * hybrid "data-parallel + stream-K" decomposition: with G = gridDim.x workgroups and tiles = F*G + R, every workgroup
//     first computes F WHOLE tiles (round r: XCD-contiguous raster tile chunk + r*G/8 + its slot, so that the workgroups
//     that run together work on neighbouring tiles at the same K position and share the A / B panels in L2 exactly like
//     the tiled launch), then the (tile, K-tile) iteration space of the R leftover tiles is cut into G EQUAL contiguous
//     ranges, one per workgroup ("stream-K"): every CU does the same number of MFMAs whatever M, N, K are;
//   * a workgroup's K-tiles form ONE stream through its 2-stage LDS ring: the DMA of the next tile's first K-tiles is
//     issued during the last MFMAs of the current tile and lands during its epilogue (own 8 KB staging slab);
//   * a tile cut between workgroups is finished by the workgroup that holds its FIRST K-tile (it reaches the tile last):
//     the others store their raw fp32 accumulators write-through (sc1) to a workspace slot and raise a flag; the
//     finisher polls the flag (relaxed), takes ONE agent-scope acquire, adds the slots in workgroup order (fixed
//     summation order: bit-reproducible for a given grid) and runs the epilogue.  Guide recipe G16/R1: placement-
//     independent, every spin bounded (a timeout raises the error word and the launch completes with wrong data, never hangs).
//     Consumed flags are reset by the finisher, so a workspace zeroed once (rgm_dit_forward does, per forward) stays valid
//     across the launches that follow on the same stream.
#include <stdlib.h>
#include "common.h"

namespace rgm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16_g4(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16,
                                   0, 0);
}

constexpr int G4_SLOT_BYTES = 128 * 128 * 4;     // one raw accumulator tile
constexpr int G4_FLAG_BYTES = 4096;              // flags[gridDim.x] + error word, ahead of the slots
constexpr unsigned G4_SPIN_LIMIT = 1u << 22;
constexpr int G4_ERR_WORD = G4_FLAG_BYTES / 4 - 1;   // last word of the flag block: set when a finisher's bounded spin ran out (rgm_gemm_streamk_status)

// EDGE: M or N is not a multiple of 128 (rows beyond the edge read a zero page; per-piece validity is carried per lane)
template <bool EDGE>
__global__ __launch_bounds__(256, 2) void gemm4_kernel(GemmParams p, const char* __restrict__ zero_page, int tiles_m, int tiles_n,
                                                      char* __restrict__ ws) {
  constexpr int BM = 128, BN = 128, NW = 4, WN = 2, TM = 2, TN = 2;
  constexpr int STAGE = (BM + BN) * 128;        // bytes per ring stage: BM A rows then BN B rows, one 128-B line each
  constexpr int SPW = (BM + BN) / 8 / NW;       // 1-KiB DMA pieces per wave and K-tile
  constexpr int NM = TM * TN * 3;               // MFMAs per k16 step
  constexpr int WCOLS = TN * 32;
  static_assert(SPW <= NM, "one DMA piece per MFMA slot of the second k16 step");
  extern __shared__ __attribute__((aligned(16))) char ring[];   // 2 stages, then 4 epilogue slabs of 8 x WCOLS floats

  const int KT = p.K >> 5;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x, G8 = G >> 3;
  const int F = ntiles / G, R = ntiles - F * G;  // whole-tile rounds, leftover tiles
  const int total = R * KT;                      // stream-K space: the leftover tiles' K-tiles (< 2^31, checked by the launcher)
  const int rq_ = total / G, rr_ = total - rq_ * G;   // the first rr_ workgroups take rq_ + 1 K-tiles, the others rq_
  auto range_begin = [&](int wi) __attribute__((always_inline)) { return wi * rq_ + min(wi, rr_); };
  // workgroup b lives on XCD b % 8 (observed; used for speed only): every XCD owns one contiguous chunk of the tiles / stream
  const int xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
  const int w = xcd * G8 + loc;
  const int it0 = range_begin(w), it1 = range_begin(w + 1);
  const int sk_t0 = total ? it0 / KT : 0, sk_k0 = total ? it0 - sk_t0 * KT : 0;
  const int Gw = F * KT + (it1 - it0);           // K-tiles this workgroup streams
  if (Gw <= 0) return;
  // segment i of this workgroup's stream: i < F a whole tile of round i, then the tiles its stream-K range touches
  auto tile_of = [&](int i) __attribute__((always_inline)) { return i < F ? xcd * F * G8 + i * G8 + loc : F * G + sk_t0 + (i - F); };

  auto coords = [&](int t, int& m0, int& n0) __attribute__((always_inline)) {   // grouped raster: 8 row-tiles per column sweep
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int in_g = t - grp * per_group;
    m0 = (first_m + in_g % gsz) * BM;
    n0 = (in_g / gsz) * BN;
  };

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Bb = reinterpret_cast<const char*>(p.B);
  unsigned* flags = reinterpret_cast<unsigned*>(ws);
  char* slots = ws + G4_FLAG_BYTES;

  // ---------------------------------------------------------------- DMA side: this wave's pieces of the K-tile stream
  const char* src[SPW];
  int inc[EDGE ? SPW : 1];
  auto aim = [&](int t, int kt) __attribute__((always_inline)) {                 // per-lane source of every piece at K-tile kt of output tile t
    int m0, n0;
    coords(t, m0, n0);
    int ln = lane;
    asm volatile("" : "+v"(ln));                                 // keep the per-piece address arithmetic out of the K loop's live set
    const int r8 = ln >> 3;
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      const int s = wave + i * NW;
      const bool isA = s < BM / 8;
      const int row_l = s * 8 + r8;                              // LDS row within the stage
      const int cs = ((ln & 7) ^ ((row_l >> 1) & 7)) << 4;       // XOR swizzle applied on the source side (gemm2.hip)
      const int row = isA ? m0 + row_l : n0 + row_l - BM;
      const bool ok = !EDGE || (isA ? row < p.M : row < p.N);
      src[i] = ok ? (isA ? Ab + (long long)row * p.lda * 4 : Bb + (long long)row * p.ldb * 4) + (long long)kt * 128 + cs : zero_page + cs;
      if (EDGE) inc[i] = ok ? 128 : 0;
    }
  };
  int l_seg = 0, l_kt = F == 0 ? sk_k0 : 0;      // the DMA side's position in the stream: (segment, K-tile) of the next issue
  bool l_aim = false;
  auto advance = [&]() __attribute__((always_inline)) {                           // the K-tile just issued is behind us
    if (++l_kt == KT) {
      ++l_seg;
      l_kt = l_seg == F ? sk_k0 : 0;
      l_aim = true;
    }
  };

  // ---------------------------------------------------------------- MFMA side
  const int wr = wave / WN, wc = wave - wr * WN;
  const int arow0 = wr * TM * 32, bcol0 = wc * TN * 32;
  const int rq = (l31 >> 1) & 7;                 // read-side swizzle (tile row offsets are multiples of 16)
  struct Frags {
    bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
  };
  f32x16 acc[TM][TN];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  auto load_frags = [&](Frags& f, const char* As) __attribute__((always_inline)) {
    const char* Bs = As + BM * 128;
    static_for<0, 2>([&](auto sc) {
      constexpr int st = decltype(sc)::value;
      const int ch = ((2 * st + hh) ^ rq) << 4, cl = ((4 + 2 * st + hh) ^ rq) << 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ro = (arow0 + i * 32 + l31) * 128;
        f.ah[st][i] = *reinterpret_cast<const bf16x8*>(As + ro + ch);
        f.al[st][i] = *reinterpret_cast<const bf16x8*>(As + ro + cl);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int ro = (bcol0 + i * 32 + l31) * 128;
        f.bh[st][i] = *reinterpret_cast<const bf16x8*>(Bs + ro + ch);
        f.bl[st][i] = *reinterpret_cast<const bf16x8*>(Bs + ro + cl);
      }
    });
  };
  // one k16 step of MFMAs from registers; in the second step the SPW pieces of the next-but-one K-tile go out one per MFMA
  auto mfma_step = [&](const Frags& f, auto stc, bool issue, char* dst) __attribute__((always_inline)) {
    constexpr int st = decltype(stc)::value;
    static_for<0, NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int t = m / (TM * TN), im = (m % (TM * TN)) / TN, in = m % TN;
      // per accumulator the term order stays al*bh, ah*bl, ah*bh (same rounding sequence as the other kernels)
      acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 0 ? f.al[st][im] : f.ah[st][im], t == 1 ? f.bl[st][in] : f.bh[st][in],
                                                            acc[im][in], 0, 0, 0);
      if constexpr (st == 1 && m < SPW) {
        if (issue) {
          dma16_g4(src[m], dst + (wave + m * NW) * 1024);
          if constexpr (EDGE) src[m] += inc[m]; else src[m] += 128;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---------------------------------------------------------------- epilogue of one output tile / hand-off of a cut tile
  // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  16 rows of the wave's 64 columns at a time go
  // through the wave's own LDS slab so that every lane owns 4 consecutive columns: bias / gate / residual reads and the C
  // stores are 16 B per lane, a full 128-B line per 8 lanes.
  // Stream-K: a tile cut between workgroups is emitted in the same row-major order -- PUBLISH stores the raw accumulators
  // write-through (sc1) to this workgroup's slot (a plain 128 x 128 fp32 tile); the finishing workgroup adds the `ncontrib`
  // slots of the workgroups after it, in workgroup order, where it reads its own staged values (fixed summation order).
  // (the launcher admits only 16-byte aligned rows: N, ldc, ldres, gate_ld multiples of 4, pointers 16-byte aligned)
  float* __restrict__ Cb = p.C;
  auto emit = [&](int m0, int n0, bool publish, int ncontrib) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));                        // epilogue addressing is recomputed here, not carried through the K loop
    const int l31 = ln & 31, hh = ln >> 5;
    float* stg = reinterpret_cast<float*>(ring + 2 * STAGE) + wave * (8 * WCOLS);
    const float* first_slot = reinterpret_cast<const float*>(slots + (long long)(w + 1) * G4_SLOT_BYTES);
    {
      constexpr int LPR = WCOLS / 4, RPI = 64 / LPR;   // 16 lanes per row, 4 rows per wave-instruction
      const int lr = ln / LPR, lc = (ln % LPR) * 4;
      const int col = n0 + bcol0 + lc;
      const bool col_ok = col < p.N;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!publish && p.bias && col_ok) bv = *reinterpret_cast<const float4*>(p.bias + col);
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slots + (long long)w * G4_SLOT_BYTES, 0, G4_SLOT_BYTES, 0x00020000);
      static_for<0, TM * 4>([&](auto hc) {
        constexpr int im = decltype(hc)::value >> 2, oct = decltype(hc)::value & 3;    // rows 8*oct .. 8*oct+7 of 32-row block im
        static_for<0, TN>([&](auto in_c) {
          constexpr int in = decltype(in_c)::value;
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4)               // register e = 4*oct + e4 holds row 8*oct + e4 + 4*hh
            stg[(e4 + 4 * hh) * WCOLS + in * 32 + l31] = acc[im][in][oct * 4 + e4];
        });
#pragma unroll
        for (int j = 0; j < 8 / RPI; ++j) {
          const int r = j * RPI + lr;
          const int trow = arow0 + im * 32 + oct * 8 + r;                              // row / first column inside the tile
          const int tcol = bcol0 + lc;
          const int row = m0 + trow;
          float4 a4 = *reinterpret_cast<const float4*>(stg + r * WCOLS + lc);   // same wave wrote it: LDS ops are in order
          if (publish) {
            const f32x4 v4 = {a4.x, a4.y, a4.z, a4.w};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), rs, (trow * BN + tcol) * 4, 0, 16 /* sc1: write-through */);
            continue;
          }
          for (int c = 0; c < ncontrib; ++c) {
            const float4 s4 = *reinterpret_cast<const float4*>(first_slot + (long long)c * (G4_SLOT_BYTES / 4) + trow * BN + tcol);
            a4.x += s4.x; a4.y += s4.y; a4.z += s4.z; a4.w += s4.w;
          }
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
            if (p.res) {
              const float4 r4 = *reinterpret_cast<const float4*>(p.res + (long long)row * p.ldres + col);
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
    }
  };

  // the finisher of a cut tile: wait for the slots of the workgroups after this one (G16 / R1: relaxed poll of ONE word by
  // one wave, ONE agent-scope acquire, barrier), returns how many there are; release_slots() hands their flags back
  auto wait_partials = [&](int sk_tile) __attribute__((always_inline)) {   // sk_tile: index among the leftover tiles
    const int tile_end = (sk_tile + 1) * KT;
    int n = 0;
    for (int w2 = w + 1; w2 < G && range_begin(w2) < tile_end; ++w2) ++n;
    if (wave == 0) {
      for (int c = 0; c < n; ++c) {
        unsigned spins = 0;
        while (__hip_atomic_load(flags + w + 1 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > G4_SPIN_LIMIT) {                          // never hang: flag the launch instead
            if (lane == 0) __hip_atomic_store(flags + G4_ERR_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // ONE acquire after the matches drops this CU's stale lines
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    return n;
  };
  auto release_slots = [&](int n) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every wave has read the slots
    __builtin_amdgcn_s_barrier();
    if (tid < n) __hip_atomic_store(flags + w + 1 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto publish_done = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains (also the ring DMA in flight: harmless)
    __builtin_amdgcn_s_barrier();
    if (tid == 0) __hip_atomic_store(flags + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // ---------------------------------------------------------------- the stream
  Frags f0, f1;
  int c_seg = 0, seg_k0 = l_kt, m0, n0;
  coords(tile_of(0), m0, n0);
  zero_acc();
  aim(tile_of(0), l_kt);
  static_for<0, SPW>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    dma16_g4(src[i], ring + (wave + i * NW) * 1024);
    if constexpr (EDGE) src[i] += inc[i]; else src[i] += 128;
  });
  advance();
  if (Gw > 1) {
    if (l_aim) {
      aim(tile_of(l_seg), l_kt);
      l_aim = false;
    }
    static_for<0, SPW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dma16_g4(src[i], ring + STAGE + (wave + i * NW) * 1024);
      if constexpr (EDGE) src[i] += inc[i]; else src[i] += 128;
    });
    advance();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPW) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  load_frags(f0, ring);

  // one K-tile: 12 MFMAs of k16 step 0, the barrier that hands K-tile g+1 over, 12 MFMAs of step 1 with the DMA of K-tile g+2
  auto body = [&](Frags& cur, Frags& nxt, int g, bool last) __attribute__((always_inline)) {
    const bool more1 = g + 1 < Gw, more2 = g + 2 < Gw;
    if (more2 && l_aim) {                          // K-tile g+2 opens a new output tile: re-aim the pieces
      aim(tile_of(l_seg), l_kt);
      l_aim = false;
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(cur, std::integral_constant<int, 0>{}, false, nullptr);
    if (more1) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own pieces of K-tile g+1 have landed
      __builtin_amdgcn_s_barrier();               // K-tile g+1 complete in LDS; nobody still reads K-tile g's stage
      // the next K-tile's fragments are requested here, a k16 step ahead of their use -- except across the end of an
      // output tile's segment, where they would sit in registers through the epilogue (one 500-cycle LDS round trip per tile)
      if (!last) load_frags(nxt, ring + ((g + 1) & 1) * STAGE);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(cur, std::integral_constant<int, 1>{}, more2, ring + (g & 1) * STAGE);
    if (more2) advance();
  };
  int g = 0;
  while (true) {
    // the segment of output tile tile_of(c_seg) this workgroup owns: K-tiles seg_k0 .. seg_k0 + n_seg - 1; f0 holds the first one's fragments
    const int n_seg = min(KT - seg_k0, Gw - g);
    for (int j = 0; j < n_seg; j += 2) {
      body(f0, f1, g + j, j + 1 == n_seg);
      if (j + 1 < n_seg) body(f1, f0, g + j + 1, j + 2 == n_seg);
    }
    g += n_seg;
    const bool publish = seg_k0 != 0;             // not the tile's first K-tile: another workgroup finishes it
    int nc = 0;
    if (!publish && n_seg != KT) nc = wait_partials(sk_t0 + (c_seg - F));   // cut tile: the later workgroups' parts, in workgroup order
    emit(m0, n0, publish, nc);
    if (publish) publish_done();
    else if (nc) release_slots(nc);
    if (g >= Gw) break;
    ++c_seg;
    seg_k0 = c_seg == F ? sk_k0 : 0;
    coords(tile_of(c_seg), m0, n0);
    zero_acc();
    load_frags(f0, ring + (g & 1) * STAGE);       // K-tile g landed before the barrier of the segment's last K-tile
  }
}

static char* g4_zero_page = nullptr;
static int g4_grid = 0;

size_t gemm4_workspace_bytes() { return (size_t)G4_FLAG_BYTES + (size_t)512 * G4_SLOT_BYTES; }

static int g4_mode = 1;   // rgm_set_streamk: 0 never, 1 heuristic, 2 whenever the operands allow it
void gemm4_set_mode(int mode) { g4_mode = mode; }
int gemm4_get_mode() { return g4_mode; }

// When it pays (tools/gemm_sweep.py + bench.py A/B, MI355X): the persistent whole-tile rounds run ~6 % above the tiled launch
// (fc1 / fc2 at M = 16384: 369 / 360 vs 347 / 339 TFLOP/s); the stream-K remainder costs every workgroup a slot round trip
// (~25 us of fabric traffic when all 512 publish at once), which eats the balance gain at M = 4096 (B = 16: C2 62.0 vs 64.7
// steps/s with fc1 on this kernel) and leaves +1.8 % on the C3 step (B = 32).  Hence: M >= 8192 and a remainder of at most
// 15 % of a workgroup's K-tiles (or none).
bool gemm4_eligible(const GemmParams& p) {
  if (g4_mode == 0) return false;
  if (!p.sk_ws || p.sk_ws_bytes < gemm4_workspace_bytes()) return false;
  if (p.aload || p.batch != 1 || p.stats || p.act > 2 || p.aux) return false;
  if (((p.N | p.ldc | p.ldres | p.gate_ld) & 3) != 0 || (((uintptr_t)p.C | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.gate) & 15) != 0)
    return false;                                  // the epilogue moves 16 bytes per lane
  const long long tiles = (long long)cdiv(p.M, 128) * cdiv(p.N, 128);
  const int KT = p.K >> 5;
  if (tiles * KT >= (1LL << 31)) return false;
  if (g4_mode == 2) return tiles * KT >= 512;
  const long long F = tiles / 512, R = tiles - F * 512;
  if (F == 0 || p.M < 8192) return false;
  const double sk = (double)R * KT / 512.0, all = (double)F * KT + sk;
  return R == 0 || (sk >= 8.0 && sk <= 0.15 * all);
}

int gemm4_launch(const GemmParams& p, hipStream_t s) {
  RGM_REQUIRE(p.aload == 0 && p.batch == 1 && !p.stats && p.act <= 2, "gemm4: dense unbatched operands, act 0..2 only");
  RGM_REQUIRE(p.sk_ws && p.sk_ws_bytes >= gemm4_workspace_bytes(), "gemm4: stream-K workspace missing (%zu bytes needed)",
              gemm4_workspace_bytes());
  RGM_REQUIRE(((uintptr_t)p.sk_ws & 15) == 0, "gemm4: workspace must be 16-byte aligned");
  RGM_REQUIRE(((p.N | p.ldc | p.ldres | p.gate_ld) & 3) == 0 &&
                  (((uintptr_t)p.C | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.gate) & 15) == 0,
              "gemm4: rows of C / residual / gate / bias must be 16-byte aligned (N, ld %% 4 == 0)");
  if (!g4_zero_page) {
    RGM_CHECK_HIP(hipMalloc(&g4_zero_page, 4096));
    RGM_CHECK_HIP(hipMemset(g4_zero_page, 0, 4096));
    int dev = 0;
    hipDeviceProp_t prop;
    RGM_CHECK_HIP(hipGetDevice(&dev));
    RGM_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g4_grid = cus * 2 > 512 ? 512 : (cus * 2) & ~7;          // the resident set: 2 workgroups per CU, a multiple of the 8 XCDs
    for (const void* k : {reinterpret_cast<const void*>(gemm4_kernel<false>), reinterpret_cast<const void*>(gemm4_kernel<true>)}) {
      RGM_CHECK_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768 + 8192));
      int per_cu = 0;
      RGM_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, 2 * 32768 + 8192));
      RGM_REQUIRE(per_cu >= 2, "gemm4: only %d workgroup(s) per CU are resident; the stream-K grid needs 2", per_cu);
    }
  }
  const int tm = cdiv(p.M, 128), tn = cdiv(p.N, 128);
  const long long total = (long long)tm * tn * (p.K >> 5);
  RGM_REQUIRE(total < (1LL << 31), "gemm4: %lld K-tiles exceed the 31-bit stream index", total);
  int grid = g4_grid;
  if (total < grid) grid = (int)(total < 8 ? 8 : total & ~7LL);   // tiny problems: no more workgroups than K-tiles
  const int rec = gemm2_prof_begin(47, 2.0 * p.M * (double)p.N * p.K, s);
  if ((p.M & 127) || (p.N & 127))
    hipLaunchKernelGGL(gemm4_kernel<true>, dim3(grid), dim3(256), 2 * 32768 + 8192, s, p, (const char*)g4_zero_page, tm, tn, (char*)p.sk_ws);
  else
    hipLaunchKernelGGL(gemm4_kernel<false>, dim3(grid), dim3(256), 2 * 32768 + 8192, s, p, (const char*)g4_zero_page, tm, tn, (char*)p.sk_ws);
  RGM_LAUNCH_CHECK();
  gemm2_prof_end(rec, s);
  return RGM_OK;
}

}  // namespace rgm

// Bytes of the stream-K workspace (flags + one raw accumulator slot per resident workgroup); zero its first 4096 bytes once
// before the first launch that uses it (the kernels hand every flag back).
extern "C" size_t rgm_gemm_streamk_workspace_bytes(void) { return rgm::gemm4_workspace_bytes(); }

// The stream-K finisher spins a bounded number of times for its contributors' slots; if a spin ever runs out (co-residency broken by
// a co-running kernel) it raises the workspace's error word and the launch completes with wrong data.  This reads that word back
// (synchronises `stream`): RGM_OK, or RGM_ERR_STATE after re-zeroing the whole flag block so that the workspace is usable again.
// rgm_dit_forward checks the same word asynchronously (dit.hip); call this at a sync point when driving the GEMM entries directly.
extern "C" int rgm_gemm_streamk_status(void* ws, void* stream) {
  RGM_REQUIRE(ws, "gemm_streamk_status: null workspace");
  unsigned err = 0;
  RGM_CHECK_HIP(hipMemcpyAsync(&err, static_cast<char*>(ws) + rgm::G4_ERR_WORD * 4, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  RGM_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (err == 0) return RGM_OK;
  RGM_CHECK_HIP(hipMemsetAsync(ws, 0, rgm::G4_FLAG_BYTES, (hipStream_t)stream));
  rgm::set_error("stream-K GEMM: a finisher timed out waiting for a partial tile (results of that launch are invalid); flags re-zeroed");
  return RGM_ERR_STATE;
}

// 0: never use the persistent stream-K kernel, 1: heuristic (default), 2: whenever the operands allow it (experiments)
extern "C" int rgm_set_streamk(int mode) {
  RGM_REQUIRE(mode >= 0 && mode <= 2, "set_streamk: mode %d", mode);
  rgm::gemm4_set_mode(mode);
  return RGM_OK;
}

// rgm_gemm_split with a caller-provided stream-K workspace: tile 0 lets the heuristic pick (stream-K when it pays), 47 forces it.
// The flag words are zeroed here on the stream (a standalone call may be the workspace's first user).
extern "C" int rgm_gemm_split_ws(const float* A_split, const float* B_split, float* C, int M, int N, int K, const float* bias, int act,
                                 int tile, int out_split, void* ws, size_t ws_bytes, void* stream) {
  RGM_REQUIRE(A_split && B_split && C && ws, "gemm_split_ws: null operand");
  rgm::GemmParams g;
  g.A = A_split; g.lda = K; g.B = B_split; g.ldb = K; g.C = C; g.ldc = N;
  g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act; g.tile = tile; g.out_split = out_split;
  g.sk_ws = ws; g.sk_ws_bytes = ws_bytes;
  RGM_CHECK_HIP(hipMemsetAsync(ws, 0, rgm::G4_FLAG_BYTES, (hipStream_t)stream));
  return rgm::gemm2_launch(g, (hipStream_t)stream);
}

// The general entry of the pre-split GEMM family: every fused epilogue the DiT block uses (alpha, bias, activation, per-sample
// adaLN gate, residual that may alias C, split-row output), explicit row strides, explicit tile (0 = heuristic, 47 = stream-K,
// 7x = the 256x256 kernels of gemm5.hip) and the caller's stream-K / split-K scratch (may be NULL: decompositions that need it
// are then not chosen; forced ones fail).  What guided_diffusion/dit.py:332-336 computes per block, in one launch.
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
  if (ws) RGM_CHECK_HIP(hipMemsetAsync(ws, 0, rgm::G4_FLAG_BYTES, (hipStream_t)stream));
  return rgm::gemm2_launch(g, (hipStream_t)stream);
}
