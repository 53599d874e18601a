// gemm144.hip -- pre-split GEMM on 128 x 144 output tiles (16x16x32 MFMAs): the tile for grids the 32-column-granular kernels of
// gemm2.hip cannot balance.
//
// Every N of DiTRotary_XL_8 (1152, 3456, 4608) is a multiple of 144 = 1152 / 8, and M = 256 B rows: fc1 at B = 4 is 8 x 32 = 256 tiles
// -- one per CU -- where 128x128 gives 288 (32 CUs carry two, the launch takes two tile-times) and 128x64 576 (0.75 of the 768 slots);
// proj at B = 16 is 32 x 8 = 256 where 128x64 gives 576 latency-bound tiles at 0.24 of the MFMA peak.  144 columns are nine 16-wide
// MFMA tiles, so the kernel multiplies with v_mfma_f32_16x16x32_bf16 (same rate as 32x32x16, K = 32 = one 128-B split-row line per
// instruction): a wave owns 32 rows x 144 columns = 2 x 9 accumulator tiles (72 registers).
//
// Operands: A (M, K) and B (N, K) in split-row format ([32 hi | 32 lo] per 128-B line, common.h), C = A . B^T; the same GemmParams
// epilogue subset the DiT forward uses (bias, SiLU / GELU, split output, gate, residual, K slices as a batch).  Structure = gemm2.hip's
// loader/consumer split (PIPE 4): waves 4-7 issue all LDS-DMA (36 KiB stages: 128 A rows, 144 B rows, 16 rows of padding so that
// every loader wave owns 9 pieces), waves 0-3 multiply from registers and fetch the next tile's 22 fragments between the MFMAs of the
// last two thirds of a tile; one barrier per K-tile.  The product is computed transposed (the B fragment is the MFMA's A operand): a lane
// then holds FOUR CONSECUTIVE COLUMNS of one output row and the epilogue reads bias / gate / residual and writes C 16 B per lane
// straight from the accumulators -- no pass through LDS.  Term order per accumulator: al*bh, ah*bl, ah*bh (as everywhere).
#include "common.h"

namespace rgm {
void gemm2_prof_set_bytes(int idx, double bytes);      // gemm2.hip
namespace {

typedef split_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifdef RGM_SPLIT_F16
#define RGM_MFMA_SPLIT_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
#define RGM_MFMA_SPLIT_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {   // 16 B per lane, LDS destination = wave-uniform base + lane*16
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

constexpr int BM = 128, BN = 144, PAD = 16;
constexpr int ROWS = BM + BN + PAD;           // LDS rows per stage
constexpr int STAGE = ROWS * 128;             // 36 KiB
constexpr int LSEG = ROWS / 8 / 4;            // 1-KiB pieces per loader wave and K-tile: 9
constexpr int TM = 2, TN = 9;                 // 16x16 accumulator tiles per consumer wave (32 rows x 144 columns)

// DBG: s_memtime stamps of the middle workgroup (tools/g144_stamp.py; rgm_gemm144_dbg), 8 slots per wave:
//   consumers: 0 barrier wait (arrival -> release), 1 whole K loop, 2 prologue (entry -> K loop), 3 epilogue, 7 K-tiles
//   loaders:   0 barrier wait, 1 whole K loop, 2 prologue, 4 piece issue, 5 landing wait, 7 K-tiles
template <int NSTAGE, int DBG = 0, int EXP = 0, int PF = 0>
__global__ __launch_bounds__(512) void gemm144_kernel(GemmParams p, const char* __restrict__ zero_page, int tiles_m, int tiles_n,
                                                      long long* __restrict__ dbg = nullptr, int pf_a = 1) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_entry = 0;
  auto now = [&]() {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
  };
  unsigned long long rt_entry = 0;
  if (DBG) {
    t_entry = now();
    rt_entry = __builtin_amdgcn_s_memrealtime();
  }
  const int z = blockIdx.z;
  // XCD-contiguous raster, row tiles fastest: an XCD's tiles are a few column panels of B (weights: read once, by one L2) x all row tiles
  const int bid = blockIdx.x, nb = tiles_m * tiles_n;
  const int xcd = bid & 7, loc = bid >> 3, q = nb >> 3, r = nb & 7;
  const int sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  // sweeps of G row tiles (p.raster_group, host: min(tiles_m, 8)): an XCD's 32 consecutive tiles are G row tiles x 32 / G column panels -- at
  // M = 4096 (32 x 8 tiles) 8 x 4, i.e. a quarter of A and half of B through each L2 where whole columns (G = tiles_m) pull ALL of A through
  // every one of the eight (in situ, proj at B = 16: 151 MB of fills for 24 MB of operands)
  const int G = p.raster_group > 0 ? p.raster_group : tiles_m;
  const int per_group = G * tiles_n, grp = sid / per_group, first_m = grp * G;
  const int gsz = min(tiles_m - first_m, G), in_g = sid - grp * per_group;
  const int m0 = (first_m + in_g % gsz) * BM, n0 = (in_g / gsz) * BN;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int KT = p.K >> 5;
  const char* Ab = reinterpret_cast<const char*>(p.A + (long long)z * p.sA);
  const char* Bb = reinterpret_cast<const char*>(p.B + (long long)z * p.sB);

  if (wave >= 4) {
    // ---- loader waves.  Piece s of a stage = LDS rows 8s..8s+7; lane = (row r8, physical 16-B chunk pc) fetches logical chunk
    // pc ^ ((row >> 1) & 7) of its row's line (the read-side swizzle of the fragment reads, applied on the source side)
    const int lw = wave - 4, r8 = lane >> 3;
    const char* src[LSEG];
    int inc[LSEG];
#pragma unroll
    for (int i = 0; i < LSEG; ++i) {
      const int row_l = (lw + i * 4) * 8 + r8;
      const int cs = ((lane & 7) ^ ((row_l >> 1) & 7)) << 4;
      const bool isA = row_l < BM;
      const int row = isA ? m0 + row_l : n0 + row_l - BM;
      const bool ok = isA ? row < p.M : (row_l < BM + BN && row < p.N);
      src[i] = ok ? (isA ? Ab + (long long)row * p.lda * 4 : Bb + (long long)row * p.ldb * 4) + cs : zero_page + cs;
      inc[i] = ok ? 128 : 0;
    }
    auto issue_tile = [&](char* dst) {
      static_for<0, LSEG>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        dma16(src[i], dst + (lw + i * 4) * 1024);
        src[i] += inc[i];
      });
    };
    auto wait_flying = [&](int tiles) {                 // wave-uniform: at most `tiles` of the newest tiles may still fly
      static_for<0, NSTAGE - 1>([&](auto c) {
        if (tiles == decltype(c)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(c)::value * LSEG) : "memory");
      });
    };
    // (Prologue, stamped in the C2 forward: ~4 k cycles until the addresses are set up -- kernel arguments and code arrive cold --, ~2.5 k for
    // the first tile's nine pieces to be issued and land, ~1 k more for the next two.  Releasing the consumers as soon as tile 0 is in, the other
    // two tiles issued behind barrier P, moved the barrier from ~8.9 k to ~7.9 k cycles and the C2 step by nothing: 11.50 / 11.48 ms over three
    // alternating pairs, B = 4 4.84 -> 4.86; removed.)
    if (DBG) tacc[3] = now() - t_entry;                 // prologue split: addresses set up ...
    static_for<0, NSTAGE - 1>([&](auto c) {
      if (decltype(c)::value < KT) issue_tile(ring + decltype(c)::value * STAGE);
    });
    wait_flying(min(NSTAGE - 2, KT - 1));
    if (DBG) tacc[6] = now() - t_entry;                 // ... the first tiles issued, tile 0 landed
    __builtin_amdgcn_s_barrier();                       // barrier P: tile 0 is in LDS
    int s2 = NSTAGE - 1;
    unsigned long long t0 = 0;
    if (DBG) {
      t0 = now();
      tacc[2] = t0 - t_entry;
    }
    for (int kt = 0; kt + 1 < KT; ++kt) {
      unsigned long long ta = 0, tb = 0, tc = 0;
      if (DBG) ta = now();
      if (kt + NSTAGE - 1 < KT && !(EXP & 4)) issue_tile(ring + s2 * STAGE);   // into tile kt-1's stage: every consumer read it before barrier kt-1
      if (DBG) tb = now();
      wait_flying(min(NSTAGE - 2, KT - 2 - kt));
      if (DBG) tc = now();
      if constexpr (!(EXP & 1)) __builtin_amdgcn_s_barrier();   // barrier kt: tile kt+1 is in LDS
      if (DBG) {
        tacc[4] += tb - ta;
        tacc[5] += tc - tb;
        tacc[0] += now() - tc;
      }
      s2 = s2 == NSTAGE - 1 ? 0 : s2 + 1;
    }
    if (DBG) {
      tacc[1] = now() - t0;
      tacc[7] = KT;
      if (blockIdx.x == gridDim.x / 2 && blockIdx.z == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) dbg[wave * 8 + i] = (long long)tacc[i];
    }
    return;
  }

  // ---- consumer waves
  const int l15 = lane & 15, kb = lane >> 4;
  const int rq = (l15 >> 1) & 7;                        // tile row offsets are multiples of 16
  const int arow0 = wave * 32;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- PF: operand lines of K-tile kt + PF pulled into this XCD's L2 ahead of the loaders (round 5).  In the forward the loaders' LDS-DMA
  // requests miss L2 (weights from HBM, A from the Infinity Cache) and their ISSUE backs up behind the outstanding misses -- 950-1300 cycles
  // per K-tile against 420 on warm operands (tools/g144_insitu_stamp.py), the consumers wait 370-710 at every barrier.  A request that
  // costs nothing to wait for hides it: one 4-byte LDS-DMA per line into a scratch KiB behind the ring, issued by the consumer waves (their
  // vmcnt is otherwise unused) PF K-tiles ahead.  The tiles of an XCD share panels (raster above): the gsz row tiles of a sweep share a B
  // panel, the XCD's column panels share an A panel; each takes its share of the rows, so a line is requested once per XCD.
  const char* pf_src = zero_page;
  int pf_inc = 0, pf_left = 0;
  char* pf_dst = ring + NSTAGE * STAGE + wave * 256;
  if constexpr (PF > 0) {
    const int q8 = nb >> 3;                                  // tiles per XCD (r == 0)
    const bool aligned = r == 0 && q8 % gsz == 0 && (per_group % q8 == 0 || q8 % per_group == 0);
    int nA = aligned ? min(q8 / gsz, tiles_n) : 1;
    nA = nA >= 8 ? 8 : nA >= 4 ? 4 : nA >= 2 ? 2 : 1;
    const int nB = aligned ? gsz : 1;
    const int iA = (in_g / gsz) % nA, iB = in_g % gsz % nB;
    // A only where the XCD's tiles share it (a K slice's row panel is read once per XCD: nothing to share, and 128 more requests per K-tile
    // overload the L1 fill path: 1793 -> 2243 cycles per K-tile on fc2's slices at B = 4) and when asked for (p.tile_flags bit 0: A/B runs)
    const int a_cnt = (nA > 1 && pf_a) ? BM / nA : 0, b_cnt = (BN + nB - 1) / nB;
    const int slot = wave * 64 + lane;
    if (slot < a_cnt) {
      const int row = m0 + iA * a_cnt + slot;
      if (row < p.M) { pf_src = Ab + (long long)row * p.lda * 4; pf_inc = 128; }
    } else if (slot < a_cnt + b_cnt) {
      const int rl = iB * b_cnt + slot - a_cnt, row = n0 + rl;
      if (rl < BN && row < p.N) { pf_src = Bb + (long long)row * p.ldb * 4; pf_inc = 128; }
    }
    // K-tiles NSTAGE - 1 .. PF - 1 while the loaders fetch the first ones; from then on one per K-tile
    pf_src += (long long)pf_inc * (NSTAGE - 1);
    pf_left = KT - (NSTAGE - 1);
    for (int i = NSTAGE - 1; i < PF; ++i) {
      if (pf_left > 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pf_src, (__attribute__((address_space(3))) void*)pf_dst, 4, 0, 0);
        pf_src += pf_inc;
      }
      --pf_left;
    }
  }
  // (Requesting the tile's RESIDUAL lines in the last K-tiles as well was measured -- 12.10 vs 12.10 ms per C2 step, the epilogue is bound by its
  // stores -- and removed.)  Two halves for two MFMA gaps: the request, then the pointer.
  auto prefetch_issue = [&]() {
    if constexpr (PF > 0) {
      if (pf_left > 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pf_src, (__attribute__((address_space(3))) void*)pf_dst, 4, 0, 0);
    }
  };
  auto prefetch_advance = [&]() {
    if constexpr (PF > 0) {
      pf_src += pf_inc;
      --pf_left;
    }
  };
  // Fragments of one K-tile: row l15 of the 16-row tile, k = 8 kb .. 8 kb + 7.  Term t of an accumulator (t-major MFMA order) multiplies
  //   t = 0: a.lo x b.hi     t = 1: a.hi x b.lo     t = 2: a.hi x b.hi
  // so the lo halves are dead after their term and are overwritten with the NEXT tile's while the tile is still being multiplied; only
  // the hi halves (live to the end) are double-buffered: 72 + 2 x 44 + 44 = 204 registers (a second full set does not fit the 256 of
  // two waves per SIMD).
  struct Half {
    bf16x8 a[TM], b[TN];
  };
  constexpr int NT = TM * TN;                           // 18 MFMAs per term
  auto rd = [&](bf16x8& dst, const char* st, int row, int lo) {
    dst = *reinterpret_cast<const bf16x8*>(st + row * 128 + (((4 * lo + kb) ^ rq) << 4));
  };
  const int ar = arow0 + l15, br = BM + l15;
  // term T of the tile whose halves are (hi, lo); the reads riding on its MFMAs (from stage `nst`, the next tile):
  //   T = 1: next a.lo (2), then the next tile's hi halves (2 + 9) into `nhi`       T = 2: next b.lo (9)
  auto term = [&](auto tc, const Half& hi, Half& lo, Half& nhi, const char* nst, auto prec) {
    constexpr int T = decltype(tc)::value;
    constexpr bool PRE = decltype(prec)::value != 0;
    static_for<0, NT>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int im = m / TN, in = m % TN;
      // transposed product: D[n][m] -> the lane holds columns 4 kb .. 4 kb + 3 of row l15
      if constexpr (T == 0) acc[im][in] = RGM_MFMA_SPLIT_16x16x32(hi.b[in], lo.a[im], acc[im][in], 0, 0, 0);
      if constexpr (T == 1) acc[im][in] = RGM_MFMA_SPLIT_16x16x32(lo.b[in], hi.a[im], acc[im][in], 0, 0, 0);
      if constexpr (T == 2) acc[im][in] = RGM_MFMA_SPLIT_16x16x32(hi.b[in], hi.a[im], acc[im][in], 0, 0, 0);
      if constexpr (PRE && T == 1 && !(EXP & 2)) {
        if constexpr (m < TM) rd(lo.a[m], nst, ar + m * 16, 1);
        else if constexpr (m < 2 * TM) rd(nhi.a[m - TM], nst, ar + (m - TM) * 16, 0);
        else if constexpr (m < 2 * TM + TN) rd(nhi.b[m - 2 * TM], nst, br + (m - 2 * TM) * 16, 0);
      }
      if constexpr (PRE && T == 2 && !(EXP & 2)) {
        // b.lo[in] is free once the LAST row tile's term-1 MFMA has used it: all of term 1 is behind us
        if constexpr (m < TN) rd(lo.b[m], nst, br + m * 16, 1);
      }
      // the L2 prefetch of this K-tile rides in a gap that carries nothing else (behind the barrier it was 10 more instructions in the one
      // gap where the MFMA pipe drains anyway: waitcnt + barrier)
      if constexpr (PRE && T == 2 && m == TN + 2) prefetch_issue();
      if constexpr (PRE && T == 2 && m == TN + 5) prefetch_advance();
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  static_assert(2 * TM + TN <= NT && TN <= NT, "the next tile's reads fit between the MFMAs of terms 1 and 2");
  Half h0, h1, lo;
  if (DBG) tacc[4] = now() - t_entry;                   // prologue split: set-up and first prefetches done ...
  __builtin_amdgcn_s_barrier();                         // barrier P (loaders: tile 0 landed)
  if (DBG) tacc[5] = now() - t_entry;                   // ... barrier P passed (then: tile 0's 22 fragments)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    rd(h0.a[i], ring, ar + i * 16, 0);
    rd(lo.a[i], ring, ar + i * 16, 1);
  }
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    rd(h0.b[i], ring, br + i * 16, 0);
    rd(lo.b[i], ring, br + i * 16, 1);
  }
  int s1 = 1;
  // a tile with a successor (its fragments are fetched on the way) / the last tile.  Kept apart -- and the loop below peeled -- so that no
  // path carries a stale fragment set to a merge point: the register allocator would keep it alive (346 registers instead of ~210)
  auto iter_more = [&](Half& cur, Half& nxt) {
    term(I0{}, cur, lo, nxt, nullptr, I0{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long ta = 0;
    if (DBG) ta = now();
    if constexpr (!(EXP & 1)) __builtin_amdgcn_s_barrier();   // tile kt+1 in LDS; every consumer read tile kt during tile kt-1
    if (DBG) tacc[0] += now() - ta;
    __builtin_amdgcn_sched_barrier(0);
    const char* nst = ring + s1 * STAGE;
    term(I1{}, cur, lo, nxt, nst, I1{});
    term(I2{}, cur, lo, nxt, nst, I1{});
    s1 = s1 == NSTAGE - 1 ? 0 : s1 + 1;
  };
  auto iter_last = [&](Half& cur) {
    term(I0{}, cur, lo, cur, nullptr, I0{});
    term(I1{}, cur, lo, cur, nullptr, I0{});
    term(I2{}, cur, lo, cur, nullptr, I0{});
  };
  unsigned long long t_loop = 0;
  if (DBG) {
    t_loop = now();
    tacc[2] = t_loop - t_entry;
  }
  int kt = 0;
  for (; kt + 2 < KT; kt += 2) {
    iter_more(h0, h1);
    iter_more(h1, h0);
  }
  if (KT - kt == 2) {
    iter_more(h0, h1);
    iter_last(h1);
  } else {
    iter_last(h0);
  }
  if (DBG) {
    const unsigned long long t = now();
    tacc[1] = t - t_loop;
    t_loop = t;
  }

  // ---- epilogue, straight from the accumulators: acc[im][in][e] = C[m0 + arow0 + 16 im + l15][n0 + 16 in + 4 kb + e]
  float* __restrict__ Cb = p.C + (long long)z * p.sC;
  const float* resb = p.res ? p.res + (long long)z * p.sRes : nullptr;
  const float* biasb = p.bias ? p.bias + (long long)z * p.sBias : nullptr;
  const int colb = n0 + 4 * kb;
  float4 bv[TN];
#pragma unroll
  for (int in = 0; in < TN; ++in) bv[in] = biasb ? *reinterpret_cast<const float4*>(biasb + colb + in * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
  typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int im = 0; im < TM; ++im) {
    const int row = m0 + arow0 + im * 16 + l15;
    const bool ok = row < p.M;
    float4 g4[TN], r4[TN];
    const bool reads = (p.gate || resb) && ok;
    if (reads) {
      const float* gp = p.gate ? p.gate + (long long)(row / p.rows_per_gate) * p.gate_ld + colb : nullptr;
      const float* rp = resb ? resb + (long long)row * p.ldres + colb : nullptr;
#pragma unroll
      for (int in = 0; in < TN; ++in) {
        g4[in] = gp ? *reinterpret_cast<const float4*>(gp + in * 16) : make_float4(1.f, 1.f, 1.f, 1.f);
        r4[in] = rp ? *reinterpret_cast<const float4*>(rp + in * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int in = 0; in < TN; ++in) {
      const int col = colb + in * 16;
      float v[4] = {acc[im][in][0] * p.alpha + bv[in].x, acc[im][in][1] * p.alpha + bv[in].y, acc[im][in][2] * p.alpha + bv[in].z,
                    acc[im][in][3] * p.alpha + bv[in].w};
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_fast_f(v[e]);
      }
      if (reads) {
        v[0] = v[0] * g4[in].x + r4[in].x; v[1] = v[1] * g4[in].y + r4[in].y;
        v[2] = v[2] * g4[in].z + r4[in].z; v[3] = v[3] * g4[in].w + r4[in].w;
      }
      if (p.out_split) {     // (before the row guard: both lanes of a pair -- same row, kb ^ 1 -- take part in the exchange)
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hi[e] = (split_t)v[e];
          lo[e] = (split_t)(v[e] - (float)hi[e]);
        }
        split_t* rowp = reinterpret_cast<split_t*>(Cb + (long long)row * p.ldc);
        if (ok) store_split4_maybe_pair<16>(rowp, col, hi, lo);     // 16 bytes per lane: the hi halves of 8 columns (kb even) / their lo halves
        continue;
      }
      if (!ok) continue;
      {
        *reinterpret_cast<float4*>(Cb + (long long)row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  if (DBG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tacc[3] = now() - t_loop;
    tacc[7] = KT;
    if (blockIdx.x == gridDim.x / 2 && blockIdx.z == 0 && lane == 0)
      for (int i = 0; i < 8; ++i) dbg[wave * 8 + i] = (long long)tacc[i];
    // every workgroup (first 4096 of z = 0): entry and exit in shader cycles and on the 100 MHz wall clock, K loop, epilogue
    if (wave == 0 && lane == 0 && blockIdx.z == 0 && blockIdx.x < 4096) {
      long long* w = dbg + 64 + 8 * blockIdx.x;
      const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      w[0] = (long long)t_entry; w[1] = (long long)now(); w[2] = (long long)rt_entry; w[3] = (long long)rt;
      w[4] = (long long)tacc[1]; w[5] = (long long)tacc[3]; w[6] = (long long)tacc[2]; w[7] = bid;
    }
  }
}

constexpr size_t DBG_SLOTS = 64 + 8 * 4096;
char* g_zero144 = nullptr;
long long* g_dbg144 = nullptr;
}  // namespace

// shapes and epilogues this kernel carries (gemm2_launch asks before it picks tile 81)
bool gemm144_supports(const GemmParams& p) {
  if (p.aload || p.stats || p.act >= 3 || p.aux || p.C2 || p.N % BN != 0 || p.K % 32 != 0 || p.K < 64) return false;
  if (((p.N | p.ldc | p.ldres | p.gate_ld) & 3) != 0) return false;
  const uintptr_t al = (uintptr_t)p.C | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.gate;
  if ((al & 15) != 0 || ((p.sC | p.sRes | p.sBias) & 3) != 0) return false;
  if (p.gate && p.rows_per_gate <= 0) return false;
  return true;
}

int gemm144_launch(const GemmParams& p, hipStream_t s) {
  RGM_REQUIRE(gemm144_supports(p), "gemm144: unsupported call (N=%d must be a multiple of 144; dense operands; act 0-2)", p.N);
  if (!g_zero144) {
    RGM_CHECK_HIP(hipMalloc(&g_zero144, 4096));
    RGM_CHECK_HIP(hipMemset(g_zero144, 0, 4096));
  }
  constexpr int NSTAGE = 4;      // 144 KiB of the 160: one workgroup per CU (profiler id 135)
  constexpr int PFD = 8;         // L2 prefetch distance in K-tiles (kernel: PF).  Same box, C2 step / B = 4 forward: 4 -> 12.58 / 5.50 ms, 8 -> 12.12 / 5.30, 16 -> 12.32 / 5.40
  static const int pf_on = getenv("RGM_G144_PF") ? atoi(getenv("RGM_G144_PF")) : 3;   // bit 0: prefetch, bit 1: A panels too (where shared)
  const int tm = cdiv(p.M, BM), tn = p.N / BN;
  const size_t lds = (size_t)NSTAGE * STAGE + 1024;      // + the prefetch's scratch KiB
  GemmParams pr = p;
  if (pr.raster_group <= 0) {
    static const int fixed = getenv("RGM_G144_RASTER") ? atoi(getenv("RGM_G144_RASTER")) : 0;   // A/B runs
    pr.raster_group = fixed > 0 ? (fixed < tm ? fixed : tm) : (tm < 8 ? tm : 8);
  }
  auto k = pf_on ? gemm144_kernel<NSTAGE, 0, 0, PFD> : gemm144_kernel<NSTAGE>;
  static bool attr = false;
  if (!attr) {
    RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm144_kernel<NSTAGE, 0, 0, PFD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm144_kernel<NSTAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  if (g_dbg144) {                // stamped variant (tools/g144_stamp.py)
    static const int exp = getenv("RGM_G144_EXP") ? atoi(getenv("RGM_G144_EXP")) : 0;   // timing only (wrong results): 1 no barriers, 2 no fragment reads, 4 no DMA
    auto launch_dbg = [&](auto kd) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kd, dim3((unsigned)(tm * tn), 1, (unsigned)p.batch), dim3(512), lds, s, pr, (const char*)g_zero144, tm, tn, g_dbg144, pf_on >> 1);
    };
    switch (exp) {            // (with the L2 prefetch, as the product kernel runs)
      case 1: launch_dbg(gemm144_kernel<NSTAGE, 1, 1, PFD>); break;
      case 2: launch_dbg(gemm144_kernel<NSTAGE, 1, 2, PFD>); break;
      case 4: launch_dbg(gemm144_kernel<NSTAGE, 1, 4, PFD>); break;
      case 7: launch_dbg(gemm144_kernel<NSTAGE, 1, 7, PFD>); break;
      default:
        if (pf_on) launch_dbg(gemm144_kernel<NSTAGE, 1, 0, PFD>);
        else launch_dbg(gemm144_kernel<NSTAGE, 1, 0>);
    }
    RGM_LAUNCH_CHECK();
    return RGM_OK;
  }
  const int pi = gemm2_prof_begin(135, 2.0 * p.M * (double)p.N * p.K * p.batch, s);
  // A once, B once, C once (+ the residual it is added to): what bench.py prices the in-situ HBM traffic against
  gemm2_prof_set_bytes(pi, 4.0 * ((double)p.M * p.K * p.batch + (double)p.N * p.K * p.batch + (double)p.M * p.N * p.batch * (p.res ? 2.0 : 1.0)));
  hipLaunchKernelGGL(k, dim3((unsigned)(tm * tn), 1, (unsigned)p.batch), dim3(512), lds, s, pr, (const char*)g_zero144, tm, tn, (long long*)nullptr, pf_on >> 1);
  gemm2_prof_end(pi, s);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
}  // namespace rgm

// tools/g144_stamp.py: mode 1 arms the stamped kernel, 2 copies the 64 slots out, 0 disarms
extern "C" int rgm_gemm144_dbg(int mode, long long* out64) {
  using namespace rgm;
  if (mode == 1) {
    if (!g_dbg144) RGM_CHECK_HIP(hipMalloc(&g_dbg144, DBG_SLOTS * sizeof(long long)));
    RGM_CHECK_HIP(hipMemset(g_dbg144, 0, DBG_SLOTS * sizeof(long long)));
  } else if (mode == 2 || mode == 3) {           // 3: + 8 slots per workgroup (the first 4096): 64 + 8 x 4096 values
    RGM_REQUIRE(g_dbg144 && out64, "gemm144_dbg: not armed");
    RGM_CHECK_HIP(hipDeviceSynchronize());
    RGM_CHECK_HIP(hipMemcpy(out64, g_dbg144, (mode == 3 ? DBG_SLOTS : 64) * sizeof(long long), hipMemcpyDeviceToHost));
  } else {
    if (g_dbg144) (void)hipFree(g_dbg144);
    g_dbg144 = nullptr;
  }
  return RGM_OK;
}
