// gemm2_body.h -- the tile body of the pre-split bf16x3 GEMM (gemm2.hip: design notes there), as a header: gemm2.hip instantiates it in
// one-shape launches, chain.hip inside the persistent launch of a whole DiT forward (COH = 1: tile addressed directly, every output
// stored device-coherent so that another workgroup of the SAME launch may consume it).
#pragma once
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include "common.h"

namespace rgm {

typedef split_t bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  // 16 B per lane, LDS destination = wave-uniform base + lane*16
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// the kernel body: output tile `bid` (raster order) of batch element `z`.  A function so that ONE launch can mix tile shapes
// (gemm2_dual_kernel below); gemm2_kernel is the plain one-shape wrapper.
template <int BM, int BN, int WM, int WN, int ALOAD, int NSTAGE, int DBG = 0, int PIPE = 0, int COH = 0>
__device__ __forceinline__ void gemm2_body(const GemmParams& p, const char* __restrict__ zero_page, int tiles_m, int tiles_n, int exp,
                                           long long* __restrict__ dbg, const int bid, const int z, const int rec_bid, const int tid_in = -1) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int STAGE = (BM + BN) * 128;        // bytes per ring stage: BM A rows then BN B rows, one 128-B line each
  constexpr int SEGS = (BM + BN) / 8;           // 1-KiB DMA pieces per stage (8 rows x 128 B)
  constexpr int SPW = SEGS / NW;                // pieces per wave
  static_assert(SEGS % NW == 0, "pieces must divide evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char ring[];   // the ONLY shared object (see guide: a 2nd one forces vmcnt(0))
  unsigned long long t_entry = 0;
  if (DBG) {
    t_entry = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- blockIdx -> tile (same XCD-contiguous grouped raster as gemm.hip)
  const int nb = tiles_m * tiles_n;
  const int xcd = bid & 7, loc = bid >> 3, q = nb >> 3, r = nb & 7;
  const int sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int GROUP = (PIPE == 5 && p.raster_group > 0) ? p.raster_group : 8;
  const int per_group = GROUP * tiles_n;
  const int grp = sid / per_group;
  const int first_m = grp * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int in_g = sid - grp * per_group;
  // COH (chain.hip): the caller names the tile -- bid = row tile * tiles_n + column tile -- and consumes the output inside the same launch
  const int m0 = COH ? (bid / tiles_n) * BM : (first_m + in_g % gsz) * BM;
  const int n0 = COH ? (bid % tiles_n) * BN : (in_g / gsz) * BN;

  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;      // (chain.hip hands in an opaque copy: see there)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;

  // ---- per-lane DMA sources.  Piece s of a stage = LDS rows 8s..8s+7; lane = (row r8, physical 16-B chunk pc).
  // The XOR swizzle that keeps the ds_read_b128 fragment reads conflict-free (chunk ^= (row >> 1) & 7, rows are
  // 128 B = 32 banks apart) is applied on the SOURCE side: the lane fetches logical chunk pc ^ swz of its line, so
  // the 8 lanes of a row still cover one whole 128-B cache line.
  const char* Ab = reinterpret_cast<const char*>(p.A + (long long)z * p.sA);
  const char* Bb = reinterpret_cast<const char*>(p.B + (long long)z * p.sB);
  // The instruction cache is cold at every launch and the code up to the first DMA is fetched line by line while the whole
  // chip waits (tools/gemm_stamp.py: 4-6k cycles of prologue): everything only the epilogue needs is computed after the K loop.
  // vector epilogue (below) needs 16-B aligned rows; uniform over the workgroup
  auto vector_epilogue = [&]() {
    const uintptr_t zb = (uintptr_t)(p.C + (long long)z * p.sC) | (p.res ? (uintptr_t)(p.res + (long long)z * p.sRes) : 0) |
                         (p.bias ? (uintptr_t)(p.bias + (long long)z * p.sBias) : 0) | (uintptr_t)p.gate |
                         (p.aux ? (uintptr_t)(p.aux + (long long)z * p.sAux) : 0);
    return ((p.N | p.ldc | p.ldres | p.gate_ld | p.ldaux) & 3) == 0 && (zb & 15) == 0 && exp != 5;
  };
  const int KT = p.K >> 5;

  if constexpr (PIPE == 4) {
    // Loader waves (waves NW..2NW-1): all the DMA of the workgroup and nothing else.  One wave issues an LDS-DMA piece
    // only every ~70-80 cycles whatever the TA load (tools/gemm_stamp.py; the guide's "ldsdma-fill": 25 GB/s per
    // loader wave), which is what sat between the MFMAs of the other kernels; NW loaders beside NW MFMA waves give the
    // 40 B/clk/CU a 128x128 tile needs without touching the MFMA waves' streams.  NSTAGE-deep ring: after barrier kt-1
    // (consumers are done with tile kt-1) a loader issues its share of tile kt+NSTAGE-1 into that stage, waits until its
    // share of tile kt+1 has landed (vmcnt((NSTAGE-2)*LSEG): only the tiles behind it may still fly) and joins barrier
    // kt.  3 stages cover a K-tile's worth of MFMA time; grids of at most one workgroup per CU (B <= 4) have nothing
    // else to hide the HBM latency of their weights behind and take 4-6 stages (tiles 53-56).
    constexpr int LSEG = SEGS / NW;                         // pieces per loader wave and tile
    static_assert(NSTAGE >= 3 && ALOAD == 0 && SEGS % NW == 0 && (NSTAGE - 2) * LSEG < 64, "PIPE 4: dense operands, ring of 3+ stages");
    if (wave >= NW) {
      const int lw = wave - NW;
      const int r8l = lane >> 3;
      const char* lsrc[LSEG];
      int linc[LSEG];
      static_assert((BM / 8) % NW == 0, "piece i of every wave is an A piece or a B piece");
#pragma unroll
      for (int i = 0; i < LSEG; ++i) {
        const int sgm = lw + i * NW;
        const bool isA = i < BM / 8 / NW;                     // == sgm < BM / 8, known per piece
        const int row_l = sgm * 8 + r8l;
        const int row_t = isA ? row_l : row_l - BM;
        const int cs = ((lane & 7) ^ ((row_l >> 1) & 7)) << 4;
        const int row = (isA ? m0 : n0) + row_t;
        const bool ok = isA ? row < p.M : row < p.N;
        lsrc[i] = ok ? (isA ? Ab + (long long)row * p.lda * 4 : Bb + (long long)row * p.ldb * 4) + cs : zero_page + cs;
        linc[i] = ok ? 128 : 0;
      }
      auto issue_tile = [&](char* dst) {
        static_for<0, LSEG>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          dma16(lsrc[i], dst + (lw + i * NW) * 1024);
          lsrc[i] += linc[i];
        });
      };
      auto wait_flying = [&](int tiles) {                // wave-uniform: at most `tiles` of the newest tiles may still fly
        static_for<0, NSTAGE - 1>([&](auto c) {
          if (tiles == decltype(c)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(c)::value * LSEG) : "memory");
        });
      };
      static_for<0, NSTAGE - 1>([&](auto c) {
        if (decltype(c)::value < KT) issue_tile(ring + decltype(c)::value * STAGE);
      });
      wait_flying(min(NSTAGE - 2, KT - 1));
      __builtin_amdgcn_s_barrier();                     // barrier P: tile 0 is in LDS
      int s2 = NSTAGE - 1;
      for (int kt = 0; kt + 1 < KT; ++kt) {
        if (kt + NSTAGE - 1 < KT && exp != 1) issue_tile(ring + s2 * STAGE);
        wait_flying(min(NSTAGE - 2, KT - 2 - kt));
        __builtin_amdgcn_s_barrier();                   // barrier kt: tile kt+1 is in LDS
        s2 = s2 == NSTAGE - 1 ? 0 : s2 + 1;
      }
      if (vector_epilogue()) __syncthreads();           // the consumers' epilogue barrier
      return;
    }
  }

  const int r8 = lane >> 3;
  const char* src[SPW];
  int inc[SPW];                                               // bytes to advance per K-tile (0 for zero-page lanes)
  int csrc[SPW];
  int a_y[SPW], a_x[SPW];
  long long a_img[SPW];
  bool a_row_ok[SPW], is_a[SPW];
  static_assert((BM / 8) % NW == 0, "piece i of every wave is an A piece or a B piece");
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int s = wave + i * NW;
    const bool isA = i < BM / 8 / NW;                         // == s < BM / 8, known per piece (no branch on the wave id)
    const int row_l = s * 8 + r8;                             // LDS row within the stage
    const int row_t = isA ? row_l : row_l - BM;               // row within the A / B tile
    csrc[i] = ((lane & 7) ^ ((row_l >> 1) & 7)) << 4;
    is_a[i] = isA;
    if (isA) {
      const int row = m0 + row_t;
      a_row_ok[i] = row < p.M;
      if (ALOAD == 0) {
        src[i] = a_row_ok[i] ? Ab + (long long)row * p.lda * 4 + csrc[i] : zero_page + csrc[i];
        inc[i] = a_row_ok[i] ? 128 : 0;
        a_y[i] = a_x[i] = 0;
        a_img[i] = 0;
      } else if (ALOAD == 2) {
        // channel-block-major K (GemmParams::conv_kmajor) on the one-wave-per-SIMD kernels: K-tile kt = (kc, tap) = (kt / 9, kt % 9) reads
        // the 128-byte line kc of the neighbour pixel `tap`.  Per piece: the byte offset of the CENTRE pixel's source line 0 (a_img), a 9-bit
        // mask of the taps whose neighbour exists (a_y) and, for the nearest-x2 upsampling convs (the source of (y, x) is (y >> 1, x >> 1)),
        // the parities of y and x (a_x): the tap's source offset is then one of two wave-uniform values per axis, picked by parity
        const int y = (row >> p.logW) & (p.H - 1), x = row & (p.W - 1);
        const int img = row >> (p.logH + p.logW);
        const int Hin = p.H >> p.ups, Win = p.W >> p.ups;
        a_img[i] = ((((long long)img * Hin + (y >> p.ups)) * Win + (x >> p.ups)) * p.Cin) * 4 + csrc[i];
        int m9 = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
          if (a_row_ok[i] && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) m9 |= 1 << tap;
        }
        a_y[i] = m9;
        a_x[i] = (y & 1) | ((x & 1) << 1);
        src[i] = zero_page + csrc[i];
        inc[i] = 0;
      } else {  // NHWC split activations: a pixel = Cin/32 lines of [32 hi | 32 lo]; source recomputed per tap
        const int img = row >> (p.logH + p.logW);
        a_y[i] = (row >> p.logW) & (p.H - 1);
        a_x[i] = row & (p.W - 1);
        a_img[i] = (long long)img * (p.H >> p.ups) * (p.W >> p.ups) * p.Cin * 4;
        src[i] = zero_page + csrc[i];
        inc[i] = 0;
      }
    } else {
      const int row = n0 + row_t;
      const bool ok = row < p.N;
      a_row_ok[i] = ok;
      src[i] = ok ? Bb + (long long)row * p.ldb * 4 + csrc[i] : zero_page + csrc[i];
      inc[i] = ok ? 128 : 0;
      a_y[i] = a_x[i] = 0;
      a_img[i] = 0;
    }
  }
  const int cpt = (ALOAD == 1) ? (p.Cin >> 5) : 1;  // K-tiles per 3x3 tap
  // Implicit conv (ALOAD 1): re-aim the A pieces.  Tap-major K (the weights as repacked [cout][9][cin]): once per tap, the pieces then walk
  // the pixel's Cin/32 lines.  Channel-block-major K (GemmParams::conv_kmajor, weights [cout][cin/32][9][32]): K-tile kt = (kc, tap) =
  // (kt / 9, kt % 9) reads line kc of the tap's neighbour, so every K-tile is re-aimed -- the nine taps of a channel block re-read the
  // same input lines within nine K-tiles, i.e. from the XCD's L2 (the one-wave-per-SIMD kernels have their own fast path: ALOAD 2).
  // Both orders sum the same products; every conv of the pre-split mode uses ONE order so that results do not depend on the tile shape.
  auto conv_retarget = [&](int kt) {
    if (ALOAD != 1) return;
    const bool km = p.conv_kmajor != 0;
    if (!km && kt % cpt != 0) return;
    const int tap = km ? kt % 9 : kt / cpt;
    const long long line = km ? (long long)(kt / 9) * 128 : 0;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const int Win = p.W >> p.ups;
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      if (!is_a[i]) continue;
      const int yy = a_y[i] + dy, xx = a_x[i] + dx;
      const bool ok = a_row_ok[i] && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      src[i] = ok ? Ab + a_img[i] + ((long long)(yy >> p.ups) * Win + (xx >> p.ups)) * p.Cin * 4 + line + csrc[i]
                  : zero_page + csrc[i];
      inc[i] = (ok && !km) ? 128 : 0;
    }
  };

  auto issue = [&](int kt, int stage) {
    char* dst = ring + stage * STAGE;
    conv_retarget(kt);               // entering a new tap (or, channel-block-major, every K-tile): re-aim the A segments
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      dma16(src[i], dst + (wave + i * NW) * 1024);
      src[i] += inc[i];
    }
  };

  const int wr = wave / WN, wc = wave - wr * WN;
  const int arow0 = wr * TM * 32, bcol0 = wc * TN * 32;
  const int rq = (l31 >> 1) & 7;                 // read-side swizzle (tile row offsets are multiples of 16)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // DBG: per-wave s_memtime deltas summed over the K loop (segments: DMA wait, barrier, DMA issue, read0, mfma0, read1, mfma1)
  unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tp = 0;
  const bool rec = DBG && bid == rec_bid;
#ifndef RGM_G2_DMA_EVERY2
#define RGM_G2_DMA_EVERY2 0      // A/B builds: 1 = the wide tile's DMA pieces two MFMAs apart (as before round 6's last change)
#endif
#define RGM_STAMP(i)                                               \
  if (DBG) {                                                       \
    __builtin_amdgcn_sched_barrier(0);                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
    tacc[i] += now_ - tp;                                          \
    tp = now_;                                                     \
    __builtin_amdgcn_sched_barrier(0);                             \
  }
  if constexpr (PIPE == 4) {
    // Consumer waves of the loader/consumer split: MFMAs from registers, the next tile's 2*(TM+TN)*2 fragment reads
    // dropped between the MFMAs of the second k16 step, one barrier per K-tile; no VMEM in the loop.
    constexpr int NM = TM * TN * 3;
    constexpr int NRD = 2 * (TM + TN) * 2;                 // ds_read_b128 per tile
    constexpr int RPM = (NRD + NM - 1) / NM;               // reads per MFMA gap
    struct Frags {
      bf16x8 a[2][TM][2], b[2][TN][2];                     // [k16 step][frag][hi, lo]
    };
    auto read_one = [&](Frags& f, const char* As, auto jc) {
      constexpr int j = decltype(jc)::value;               // read index: st-major, A frags (hi, lo) then B frags (hi, lo)
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
        acc[im][in] = RGM_MFMA_SPLIT_32x32x16(f.a[st][im][t == 0 ? 1 : 0], f.b[st][in][t == 1 ? 1 : 0], acc[im][in], 0, 0, 0);
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
    Frags f0, f1;
    // B-panel lines of K-tile kt + p.pf_kt requested into this XCD's L2 ahead of the loaders (as gemm144.hip's PF: in the forward the weights
    // come from HBM and the loaders' issue backs up behind their own outstanding misses).  The row tiles of a raster sweep share a B panel and
    // sit on one XCD: each requests its share of the BN rows -- one 4-byte LDS-DMA per line into a scratch KiB behind the ring (launch2 adds
    // it where it fits), by the consumer waves, whose vmcnt is otherwise unused.
    const char* pf_src = zero_page;
    int pf_inc = 0, pf_left = 0;
    char* pf_dst = ring + NSTAGE * STAGE + wave * 256;
    bool pf_wave = false;
    if (p.pf_kt > 0 && !COH) {
      const int b_cnt = (BN + gsz - 1) / gsz;
      const int slot = wave * 64 + lane, rl = (in_g % gsz) * b_cnt + slot, row = n0 + rl;
      pf_wave = wave * 64 < b_cnt;
      if (slot < b_cnt && rl < BN && row < p.N) {
        pf_src = Bb + (long long)row * p.ldb * 4 + (long long)(NSTAGE - 1) * 128;
        pf_inc = 128;
      }
      pf_left = pf_wave ? KT - (NSTAGE - 1) : 0;
      for (int i = NSTAGE - 1; i < p.pf_kt; ++i) {
        if (pf_left > 0) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pf_src, (__attribute__((address_space(3))) void*)pf_dst, 4, 0, 0);
          pf_src += pf_inc;
        }
        --pf_left;
      }
    }
    __builtin_amdgcn_s_barrier();                         // barrier P (loader: tile 0 landed)
    static_for<0, NRD>([&](auto jc) { read_one(f0, ring, jc); });
    int s1 = 1;
    if (DBG) {
      tp = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tacc[2] = tp - t_entry;
    }
    auto iter = [&](Frags& cur, Frags& nxt, int kt) {
      const bool more1 = kt + 1 < KT;
      mfma_step(cur, std::integral_constant<int, 0>{}, false, nxt, nullptr);
      RGM_STAMP(4)
      if (more1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RGM_STAMP(0)
        __builtin_amdgcn_s_barrier();                     // tile kt+1 in LDS; every consumer is done reading tile kt
        RGM_STAMP(1)
        if (pf_left > 0) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pf_src, (__attribute__((address_space(3))) void*)pf_dst, 4, 0, 0);
          pf_src += pf_inc;
        }
        --pf_left;
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(cur, std::integral_constant<int, 1>{}, more1, nxt, ring + s1 * STAGE);
      RGM_STAMP(6)
      s1 = s1 == NSTAGE - 1 ? 0 : s1 + 1;
    };
    for (int kt = 0; kt < KT; kt += 2) {
      iter(f0, f1, kt);
      if (kt + 1 < KT) iter(f1, f0, kt + 1);
    }
  } else
  if constexpr (PIPE == 5) {
    // ONE wave per SIMD, 128x128 accumulators per wave (256x256 per workgroup, 256 AGPRs), everything else hidden behind the wave's
    // own MFMA stream.  Why: at 128x128 per workgroup (64x64 per wave) a K-tile is 24 MFMAs against 16 fragment reads + 8 DMA
    // pieces per wave and 42 B/clk/CU of LDS-DMA (70 % of what the texture addresser delivers: DESIGN 4/4b, the operand stream
    // alone takes longer than the MFMAs); a 128x128 wave tile makes it 96 MFMAs against 32 reads + 16 pieces and 21 B/clk/CU.
    // There is no second wave on the SIMD to fill a stall, so a K-tile is two phases of 48 MFMAs, each carrying the LDS reads
    // of the NEXT k16 step (register double buffer, 2 x 64 VGPRs) and, in the second phase, the DMA of the next-but-one K-tile:
    //   phase A(kt): MFMAs of (kt, step 0) | reads of (kt, step 1)
    //   wait own DMA of tile kt+1, own reads of tile kt; s_barrier   (tile kt+1 complete in LDS, tile kt's stage is free)
    //   phase B(kt): MFMAs of (kt, step 1) | reads of (kt+1, step 0) | DMA of tile kt+2 into tile kt's stage
    // so every DMA has 1.5-3 k cycles to land and every fragment 1.5 k.  The last two K-tiles run peeled bodies without the
    // reads / DMA they do not need: no branch sits between the MFMAs of the steady state.
    static_assert(NSTAGE == 2, "PIPE 5: 2-stage ring");
    constexpr int NM = TM * TN * 3;                       // MFMAs per k16 step
    constexpr int NRD = 2 * (TM + TN);                    // ds_read_b128 per k16 step
    auto retarget = [&](int kt) { conv_retarget(kt); };
    // MFMA slots per read / DMA piece: 3 for the 128x128 wave tile (48 MFMAs a phase, 16 reads, 16-20 pieces), 2 for 128x64 (24 / 12 / 12)
    constexpr int EV = (NRD * 3 <= NM && SPW * 3 <= 2 * NM) ? 3 : 2;
    static_assert(NRD * EV <= NM && (EV == 3 ? SPW * 3 <= 2 * NM : SPW * 2 <= NM), "reads and DMA pieces must fit the MFMA slots of a phase");
    // SBLO (wave tiles of more than 16 accumulator tiles: the 64 x 288 wave tile of the 256x288 workgroup tile, 288 accumulator registers):
    // two full fragment sets (2 x 88 registers) do not fit beside them.  Term t of a k16 step multiplies  t = 0: a.lo x b.hi,  t = 1: a.hi x b.lo,
    // t = 2: a.hi x b.hi  (t-major), so a.lo is dead after the first third of a phase and b.lo after the second: only the hi halves are
    // double-buffered, the next step's lo halves are read into the SAME registers once their term is behind (as gemm144.hip does).
    constexpr bool SBLO = TM * TN > 16;
    static_assert(!SBLO || ALOAD == 0, "the wide wave tiles take dense operands only");
    // 18 accumulator tiles are 288 registers: 256 of them ARE the AGPRs, the last two tiles live in VGPRs.  Left to the allocator the loop
    // shuffled accumulators between the two files (1072 v_accvgpr moves + 205 scratch accesses per K-tile): the MFMAs of this path are
    // asm statements whose constraint names the file of each accumulator.
    auto mfma_pin = [&](auto ag_c, f32x16& c, const bf16x8& a, const bf16x8& b) __attribute__((always_inline)) {
#ifdef RGM_SPLIT_F16
      if constexpr (decltype(ag_c)::value) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#else
      if constexpr (decltype(ag_c)::value) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#endif
    };
    struct FragsK {
      bf16x8 a[TM][SBLO ? 1 : 2], b[TN][SBLO ? 1 : 2];    // [frag][hi, lo] of one k16 step (SBLO: hi only)
    };
    bf16x8 alo[SBLO ? TM : 1], blo[SBLO ? TN : 1];        // SBLO: the single set of lo halves
    auto read_one = [&](FragsK& f, const char* As, auto stc, auto jc) {
      constexpr int st = decltype(stc)::value, j = decltype(jc)::value;
      if constexpr (SBLO) {
        // read index j: 0 .. TM + TN - 1 the hi halves (A then B), then the TM a.lo, then the TN b.lo
        constexpr int lo = j >= TM + TN ? 1 : 0;
        constexpr int fi = lo ? j - (TM + TN) : j;
        const int chunk = ((4 * lo + 2 * st + hh) ^ rq) << 4;
        if constexpr (fi < TM) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(As + (arow0 + fi * 32 + l31) * 128 + chunk);
          if constexpr (lo) alo[fi] = v;
          else f.a[fi][0] = v;
        } else {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(As + BM * 128 + (bcol0 + (fi - TM) * 32 + l31) * 128 + chunk);
          if constexpr (lo) blo[fi - TM] = v;
          else f.b[fi - TM][0] = v;
        }
      } else {
      constexpr int fi = j / 2, lo = j % 2;
      const int chunk = ((4 * lo + 2 * st + hh) ^ rq) << 4;
      if constexpr (fi < TM) {
        f.a[fi][lo] = *reinterpret_cast<const bf16x8*>(As + (arow0 + fi * 32 + l31) * 128 + chunk);
      } else {
        f.b[fi - TM][lo] = *reinterpret_cast<const bf16x8*>(As + BM * 128 + (bcol0 + (fi - TM) * 32 + l31) * 128 + chunk);
      }
      }
    };
    // ALOAD == 2: the A pieces of a K-tile (tap, kc) point at line kc of the tap's neighbour pixel, or at the zero page
    constexpr int APIECES = BM / 8 / NW;                  // the first APIECES pieces of a wave are A pieces
    // wave-uniform description of a K-tile's tap: source-pixel steps for even / odd y and x (without upsampling both are dy / dx;
    // with it a step only crosses into the next source pixel from the matching parity), in bytes, plus the channel block's line
    struct TapStep {
      int tap;
      int row_e, row_o, col_e, col_o;      // bytes
      long long line;
    };
    auto tap_step = [&](int tap, int kc) {
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int Win = p.W >> p.ups;
      const int pix = p.Cin * 4, rowb = Win * pix;
      TapStep t;
      t.tap = tap;
      t.row_e = (p.ups ? (dy < 0 ? -1 : 0) : dy) * rowb;
      t.row_o = (p.ups ? (dy > 0 ? 1 : 0) : dy) * rowb;
      t.col_e = (p.ups ? (dx < 0 ? -1 : 0) : dx) * pix;
      t.col_o = (p.ups ? (dx > 0 ? 1 : 0) : dx) * pix;
      t.line = (long long)kc * 128;
      return t;
    };
    auto aim_piece = [&](auto ic, const TapStep& t) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < APIECES) {
        const int d = ((a_x[i] & 1) ? t.row_o : t.row_e) + ((a_x[i] & 2) ? t.col_o : t.col_e);
        src[i] = ((a_y[i] >> t.tap) & 1) ? Ab + a_img[i] + t.line + d : zero_page + csrc[i];
      }
    };
    // the same in two halves for the gaps behind two consecutive MFMAs: one piece's 12 VALU instructions in ONE gap take longer than the MFMA
    // they hide behind (stamps in the decode, 512x128 tile: phase A 2170 cycles against 1618 on dense operands; the ISA shows 16 gaps of 12
    // fillers each beside 32 gaps of 0-2).  aim_a: the tap's source-pixel offset and whether the neighbour exists; aim_b: the address.
    int aim_d[APIECES > 0 ? APIECES : 1], aim_ok[APIECES > 0 ? APIECES : 1];
    const char* aim_p[APIECES > 0 ? APIECES : 1];
    auto aim_a = [&](auto ic, const TapStep& t) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < APIECES) {
        aim_d[i] = ((a_x[i] & 1) ? t.row_o : t.row_e) + ((a_x[i] & 2) ? t.col_o : t.col_e);
        aim_ok[i] = (a_y[i] >> t.tap) & 1;
        aim_p[i] = Ab + a_img[i] + t.line;
        asm volatile("" : "+v"(aim_d[i]), "+v"(aim_ok[i]), "+v"(aim_p[i]));      // computed HERE: the compiler otherwise sinks all of it into aim_b's gap
      }
    };
    auto aim_b = [&](auto ic, const TapStep& t) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < APIECES) src[i] = aim_ok[i] ? aim_p[i] + aim_d[i] : zero_page + csrc[i];
    };
    auto aim_all = [&](int tap, int kc) {
      const TapStep t = tap_step(tap, kc);
      static_for<0, APIECES>([&](auto ic) { aim_piece(ic, t); });
    };
    // one k16 step: NM MFMAs from `cur`; READ: the NRD reads of step `stn` of the tile at `rd` into `nxt`; DMA: this wave's SPW
    // pieces of the next-but-one tile into `dst`; AIM (ALOAD 2, phases without DMA): the A pieces' sources of the next-but-one tile
    auto phase_aim = [&](const FragsK& cur, FragsK& nxt, auto readc, auto dmac, auto stnc, const char* rd, char* dst, auto aimc,
                         const TapStep& aim) {
      constexpr bool READ = decltype(readc)::value != 0, DMA = decltype(dmac)::value != 0;
      constexpr bool AIM = decltype(aimc)::value != 0;
      static_assert(!(AIM && DMA), "the sources are re-aimed in the phase that does not issue them");
      static_for<0, NM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int t = m / (TM * TN), im = (m % (TM * TN)) / TN, in = m % TN;
        // per accumulator the term order stays al*bh, ah*bl, ah*bh (same rounding sequence as the other kernels)
        if constexpr (SBLO) {
          using AG = std::integral_constant<bool, (im * TN + in) < 16>;      // the first 16 accumulator tiles: AGPRs
          if constexpr (t == 0) mfma_pin(AG{}, acc[im][in], alo[im], cur.b[in][0]);
          if constexpr (t == 1) mfma_pin(AG{}, acc[im][in], cur.a[im][0], blo[in]);
          if constexpr (t == 2) mfma_pin(AG{}, acc[im][in], cur.a[im][0], cur.b[in][0]);
          // reads of the next step at the even MFMA slots: the hi halves from slot 0 on, a.lo behind term 0 (slot TM * TN / 2 + ...), b.lo
          // behind term 1 -- each into registers whose last reader has issued
          if constexpr (READ && m % 2 == 0) {
            constexpr int sl = m / 2;
            constexpr int S_ALO = (TM * TN + 1) / 2 > TM + TN ? (TM * TN + 1) / 2 : TM + TN;      // first slot at or behind the end of term 0
            constexpr int S_BLO = TM * TN;                                                       // slot 2 TM TN / 2: the end of term 1
            static_assert(S_ALO + TM <= S_BLO && S_BLO + TN <= (NM + 1) / 2, "the lo reads must fit behind their terms");
            if constexpr (sl < TM + TN) read_one(nxt, rd, stnc, std::integral_constant<int, sl>{});
            else if constexpr (sl >= S_ALO && sl < S_ALO + TM) read_one(nxt, rd, stnc, std::integral_constant<int, TM + TN + (sl - S_ALO)>{});
            else if constexpr (sl >= S_BLO && sl < S_BLO + TN) read_one(nxt, rd, stnc, std::integral_constant<int, 2 * TM + TN + (sl - S_BLO)>{});
          }
        } else {
        acc[im][in] = RGM_MFMA_SPLIT_32x32x16(cur.a[im][t == 0 ? 1 : 0], cur.b[in][t == 1 ? 1 : 0], acc[im][in], 0, 0, 0);
        if constexpr (READ && m % EV == 0 && m / EV < NRD) read_one(nxt, rd, stnc, std::integral_constant<int, m / EV>{});
        }
        if constexpr (AIM && EV == 3) {
          if constexpr (m % 3 == 1 && m / 3 < APIECES) aim_a(std::integral_constant<int, m / 3>{}, aim);
          if constexpr (m % 3 == 2 && m / 3 < APIECES) aim_b(std::integral_constant<int, m / 3>{}, aim);
        } else if constexpr (AIM && m % EV == 1 && m / EV < APIECES) {
          aim_piece(std::integral_constant<int, m / EV>{}, aim);
        }
        // DMA pieces: one per DEV MFMA slots.  The wide wave tile (SBLO: 54 MFMAs a phase, 17 pieces, reads at the even slots) had them behind every
        // second MFMA -- in situ a piece then cost the phase 34 cycles against 18 in the 256x256 tile, whose pieces sit three MFMAs apart
        // (stamps: phase B 2305 cycles against a floor of 1728); three apart here too.
        constexpr int DEV = (SBLO && SPW * 3 <= NM && !RGM_G2_DMA_EVERY2) ? 3 : EV;
        if constexpr (DMA && m % DEV == 1 && m / DEV < SPW) {
          constexpr int i = m / DEV;
          dma16(src[i], dst + (wave + i * NW) * 1024);
          src[i] += inc[i];
        }
        if constexpr (DMA && EV == 3 && m % 3 == 2 && NM / 3 + m / 3 < SPW) {      // tiles with more than NM / 3 pieces per wave (512x128: 20)
          constexpr int i = NM / 3 + m / 3;
          dma16(src[i], dst + (wave + i * NW) * 1024);
          src[i] += inc[i];
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    using Y = std::integral_constant<int, 1>;
    using Nn = std::integral_constant<int, 0>;
    auto phase = [&](const FragsK& cur, FragsK& nxt, auto readc, auto dmac, auto stnc, const char* rd, char* dst) {
      phase_aim(cur, nxt, readc, dmac, stnc, rd, dst, Nn{}, TapStep{});
    };
    FragsK f0, f1;
    // (tap, kc) of the K-tile whose A sources are aimed next (ALOAD 2): K-tile kt = (kt % 9, kt / 9)
    int tap_n = 0, kc_n = 0;
    auto next_ktile = [&]() {
      if (++tap_n == 9) {
        tap_n = 0;
        ++kc_n;
      }
    };
    if (ALOAD == 2) {
      aim_all(0, 0);
      next_ktile();
    }
    retarget(0);
    static_for<0, SPW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dma16(src[i], ring + (wave + i * NW) * 1024);
      src[i] += inc[i];
    });
    if (KT > 1) {
      if (ALOAD == 2) {
        aim_all(tap_n, kc_n);
        next_ktile();
      }
      retarget(1);
      static_for<0, SPW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        dma16(src[i], ring + STAGE + (wave + i * NW) * 1024);
        src[i] += inc[i];
      });
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    static_for<0, NRD>([&](auto jc) { read_one(f0, ring, Nn{}, jc); });
    auto handover = [&]() {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own pieces of the next tile landed, own reads of this tile done
      __builtin_amdgcn_s_barrier();
    };
    if (DBG) {
      tp = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tacc[2] = tp - t_entry;                             // prologue: entry -> K loop
    }
    int kt = 0;
    for (; kt + 2 < KT; ++kt) {                           // steady state: tiles kt+1 and kt+2 exist
      char* cs = ring + (kt & 1) * STAGE;
      char* ns = ring + ((kt + 1) & 1) * STAGE;
      if constexpr (ALOAD == 2) {                            // K-tile kt + 2's A sources, computed in the gaps of this phase's MFMAs
        phase_aim(f0, f1, Y{}, Nn{}, Y{}, cs, nullptr, Y{}, tap_step(tap_n, kc_n));
        next_ktile();
      } else {
        phase(f0, f1, Y{}, Nn{}, Y{}, cs, nullptr);
      }
      RGM_STAMP(4)
      retarget(kt + 2);
      if (DBG) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        RGM_STAMP(0)
        __builtin_amdgcn_s_barrier();
        RGM_STAMP(1)
      } else {
        handover();
      }
      phase(f1, f0, Y{}, Y{}, Nn{}, ns, cs);
      RGM_STAMP(6)
    }
    if (kt + 1 < KT) {                                    // last but one: nothing left to fetch
      char* cs = ring + (kt & 1) * STAGE;
      char* ns = ring + ((kt + 1) & 1) * STAGE;
      phase(f0, f1, Y{}, Nn{}, Y{}, cs, nullptr);
      handover();
      phase(f1, f0, Y{}, Nn{}, Nn{}, ns, nullptr);
      ++kt;
    }
    {                                                     // last tile
      char* cs = ring + (kt & 1) * STAGE;
      phase(f0, f1, Y{}, Nn{}, Y{}, cs, nullptr);
      phase(f1, f0, Nn{}, Nn{}, Nn{}, nullptr, nullptr);
    }
    RGM_STAMP(6)
  } else
  if constexpr (PIPE == 3) {
    // Cross-iteration register pipeline: the fragments of K-tile kt+1 are requested (behind the barrier that says the
    // tile has landed) BEFORE the second k16 step of tile kt is multiplied, into a second register set, so neither the
    // LDS round trip nor the barrier skew is exposed -- a wave's stream is MFMA, MFMA, ... with the DMA pieces of tile
    // kt+2 dropped in between.  3-stage ring: kt+2 is issued at the top of iteration kt (1.5-2 K-tiles of latency
    // cover); 2-stage ring: kt+2 reuses tile kt's stage, so it is issued after the mid-iteration barrier.
    constexpr int NM = TM * TN * 3;                       // MFMAs per k16 step
    constexpr int P0 = (NSTAGE == 3) ? (SPW < NM / 2 ? SPW : NM / 2) : 0;   // pieces issued during step 0
    constexpr int P1 = SPW - P0;                                             // ... during step 1
    static_assert(P1 <= NM, "not enough MFMA slots to spread the DMA pieces");
    struct Frags {
      bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    };
    auto retarget = [&](int kt) { conv_retarget(kt); };
    auto load_frags = [&](Frags& f, const char* As) {
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
    // one k16 step of MFMAs from registers; DMA pieces [PB, PB + PN) of the next-but-one tile dropped in between
    auto mfma_step = [&](const Frags& f, auto stc, auto pbc, auto pnc, bool issue, char* dst) {
      constexpr int st = decltype(stc)::value, PB = decltype(pbc)::value, PN = decltype(pnc)::value;
      constexpr int EVERY = (PN <= NM / 2) ? 2 : 1;
      static_for<0, NM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int t = m / (TM * TN), im = (m % (TM * TN)) / TN, in = m % TN;
        // per accumulator the term order stays al*bh, ah*bl, ah*bh (same rounding sequence as the other kernels)
        acc[im][in] = RGM_MFMA_SPLIT_32x32x16(t == 0 ? f.al[st][im] : f.ah[st][im], t == 1 ? f.bl[st][in] : f.bh[st][in],
                                                              acc[im][in], 0, 0, 0);
        if constexpr ((m % EVERY) == EVERY - 1 && (m / EVERY) < PN) {
          constexpr int i = PB + m / EVERY;
          if (issue) {
            dma16(src[i], dst + (wave + i * NW) * 1024);
            src[i] += inc[i];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    Frags f0, f1;
    retarget(0);
    static_for<0, SPW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dma16(src[i], ring + (wave + i * NW) * 1024);
      src[i] += inc[i];
    });
    if (KT > 1) {
      retarget(1);
      static_for<0, SPW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        dma16(src[i], ring + STAGE + (wave + i * NW) * 1024);
        src[i] += inc[i];
      });
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    load_frags(f0, ring);
    int s1 = 1 % NSTAGE, s2 = 2 % NSTAGE;                  // ring stages of tiles kt+1, kt+2
    if (DBG) {
      tp = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tacc[2] = tp - t_entry;                       // prologue: entry -> K loop
    }
    auto iter = [&](Frags& cur, Frags& nxt, int kt) {
      const bool more1 = kt + 1 < KT, more2 = kt + 2 < KT && exp != 1;
      char* dst2 = ring + s2 * STAGE;
      if (more2) retarget(kt + 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(cur, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, P0>{}, more2, dst2);
      RGM_STAMP(4)
      if (more1) {
        // own pieces of tile kt+1 have landed once only the P0 just-issued pieces of tile kt+2 are outstanding
        if (more2 && P0 > 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P0) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        RGM_STAMP(0)
        __builtin_amdgcn_s_barrier();   // tile kt+1 complete in LDS; nobody still reads tile kt's stage
        RGM_STAMP(1)
        load_frags(nxt, ring + s1 * STAGE);
        if (DBG) { RGM_STAMP(3) }      // (stamping waits for the fragments: the un-stamped kernel does not)
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(cur, std::integral_constant<int, 1>{}, std::integral_constant<int, P0>{}, std::integral_constant<int, P1>{}, more2, dst2);
      RGM_STAMP(6)
      s1 = s1 + 1 == NSTAGE ? 0 : s1 + 1;
      s2 = s2 + 1 == NSTAGE ? 0 : s2 + 1;
    };
    for (int kt = 0; kt < KT; kt += 2) {
      iter(f0, f1, kt);
      if (kt + 1 < KT) iter(f1, f0, kt + 1);
    }
  } else
  if constexpr (PIPE) {
    // Software-pipelined body (2-stage ring): per K-tile ONE exposed LDS round trip -- both k16 steps' fragments are
    // requested up front into two register sets -- and the next tile's DMA pieces are issued one per two MFMAs
    // (a burst of 8 pieces right after the barrier cost each wave ~800 cycles in the TA queue: tools/gemm_stamp.py).
    static_assert(NSTAGE == 2, "PIPE 1: 2-stage ring only");
    constexpr int NM = TM * TN * 3;                       // MFMAs per k16 step
    auto retarget = [&](int kt) { conv_retarget(kt); };
    retarget(0);
    static_for<0, SPW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dma16(src[i], ring + (wave + i * NW) * 1024);
      src[i] += inc[i];
    });
    int stage = 0;
    if (DBG) {
      tp = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tacc[2] = tp - t_entry;                       // prologue: entry -> K loop
    }
    for (int kt = 0; kt < KT; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RGM_STAMP(0)
      __builtin_amdgcn_s_barrier();   // tile kt is in LDS; everyone is done reading the other stage
      RGM_STAMP(1)
      const char* As = ring + stage * STAGE;
      const char* Bs = As + BM * 128;
      char* nxt = ring + (stage ^ 1) * STAGE;
      const bool more = (kt + 1 < KT) && exp != 1;
      bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
      static_for<0, 2>([&](auto sc) {
        constexpr int st = decltype(sc)::value;
        const int ch = ((2 * st + hh) ^ rq) << 4, cl = ((4 + 2 * st + hh) ^ rq) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int ro = (arow0 + i * 32 + l31) * 128;
          ah[st][i] = *reinterpret_cast<const bf16x8*>(As + ro + ch);
          al[st][i] = *reinterpret_cast<const bf16x8*>(As + ro + cl);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int ro = (bcol0 + i * 32 + l31) * 128;
          bh[st][i] = *reinterpret_cast<const bf16x8*>(Bs + ro + ch);
          bl[st][i] = *reinterpret_cast<const bf16x8*>(Bs + ro + cl);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if (more) retarget(kt + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (DBG) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RGM_STAMP(3)
      }
      static_for<0, 2 * NM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int st = m / NM, t = (m % NM) / (TM * TN), im = ((m % NM) % (TM * TN)) / TN, in = (m % NM) % TN;
        // per accumulator the term order stays al*bh, ah*bl, ah*bh (same rounding sequence as the other kernels)
        acc[im][in] = RGM_MFMA_SPLIT_32x32x16(t == 0 ? al[st][im] : ah[st][im], t == 1 ? bl[st][in] : bh[st][in],
                                                              acc[im][in], 0, 0, 0);
        if constexpr ((m & 1) == 1 && (m >> 1) < SPW) {
          constexpr int i = m >> 1;
          if (more) {
            dma16(src[i], nxt + (wave + i * NW) * 1024);
            src[i] += inc[i];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG && m == NM - 1) { RGM_STAMP(4) }
        if constexpr (DBG && m == 2 * NM - 1) { RGM_STAMP(6) }
      });
      stage ^= 1;
    }
  } else {
  issue(0, 0);
  if (NSTAGE == 3 && KT > 1) issue(1, 1);
  int stage = 0;
  if (DBG) {
    tp = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  for (int kt = 0; kt < KT; ++kt) {
    if (DBG) {   // same schedule as below, stamped
      if (NSTAGE == 3) {
        if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      RGM_STAMP(0)
      __builtin_amdgcn_s_barrier();
      RGM_STAMP(1)
      if (NSTAGE == 3) {
        if (kt + 2 < KT && exp != 1) issue(kt + 2, stage == 0 ? 2 : stage - 1);
      } else {
        if (kt + 1 < KT && exp != 1) issue(kt + 1, stage ^ 1);
      }
      RGM_STAMP(2)
    } else if (NSTAGE == 3) {
      // tile kt has landed once at most the NEXT tile's SPW segments of this wave are still in flight
      if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // every wave's part of tile kt is in LDS; everyone is done reading stage (kt-1)%3
      if (kt + 2 < KT && exp != 1) issue(kt + 2, stage == 0 ? 2 : stage - 1);   // (kt+2)%3 == (stage+2)%3
    } else {
      // 2-stage ring (half the LDS -> twice the co-resident workgroups): tile kt is the only DMA in flight here
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // tile kt is in LDS; everyone is done reading stage (kt-1)%2 = (kt+1)%2
      if (kt + 1 < KT && exp != 1) issue(kt + 1, stage ^ 1);
    }
    const char* As = ring + stage * STAGE;
    const char* Bs = As + BM * 128;
    if (DBG || exp != 2)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int ch = ((2 * st + hh) ^ rq) << 4, cl = ((4 + 2 * st + hh) ^ rq) << 4;   // hi / lo chunk of this k16 step
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ro = (arow0 + i * 32 + l31) * 128;
        ah[i] = *reinterpret_cast<const bf16x8*>(As + ro + ch);
        al[i] = *reinterpret_cast<const bf16x8*>(As + ro + cl);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int ro = (bcol0 + i * 32 + l31) * 128;
        bh[i] = *reinterpret_cast<const bf16x8*>(Bs + ro + ch);
        bl[i] = *reinterpret_cast<const bf16x8*>(Bs + ro + cl);
      }
      if (DBG) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (st == 0) { RGM_STAMP(3) } else { RGM_STAMP(5) }
      }
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in) {
          acc[im][in] = RGM_MFMA_SPLIT_32x32x16(al[im], bh[in], acc[im][in], 0, 0, 0);
          acc[im][in] = RGM_MFMA_SPLIT_32x32x16(ah[im], bl[in], acc[im][in], 0, 0, 0);
          acc[im][in] = RGM_MFMA_SPLIT_32x32x16(ah[im], bh[in], acc[im][in], 0, 0, 0);
        }
      if (DBG) {
        if (st == 0) { RGM_STAMP(4) } else { RGM_STAMP(6) }
      }
    }
    stage = (NSTAGE == 3) ? (stage == 2 ? 0 : stage + 1) : (stage ^ 1);
  }

  }   // !PIPE
  const unsigned long long t_loop_end = tp;
  float* __restrict__ Cb = p.C + (long long)z * p.sC;
  const float* resb = p.res ? p.res + (long long)z * p.sRes : nullptr;
  const float* biasb = p.bias ? p.bias + (long long)z * p.sBias : nullptr;
  const float* auxb = p.aux ? p.aux + (long long)z * p.sAux : nullptr;   // act 3 / 4: pre-activation whose derivative multiplies the result
  if constexpr (PIPE == 5 && (TN * 32 > 256 || 64 % (TN * 8) != 0)) {
    // ---- wave tiles whose rows are not a power-of-two number of lanes (64 x 288: the 256x288 workgroup tile, fc1 of DiT-XL at M = 4096 as ONE round
    // of 256 tiles; 64 x 224: the 256x224 tile, qkv likewise).  The plain epilogue only (bias, SiLU / GELU, fp32 or split rows -- launch2 checks):
    // a wave's 32-row slab is WCOLS / 8 = 36 (28) units of 8 columns per row, so the slab is walked linearly -- unit u = lane + 64 k is row
    // u / 36, columns 8 (u % 36) .. + 7 -- with the bias tile in LDS (a per-iteration global load would queue behind the previous iteration's
    // stores: one vmcnt).
    static_assert(NW * 32 * TN * 32 * 4 + BN * 4 <= 160 * 1024, "wide epilogue: four slabs + the bias tile must fit the LDS");
    constexpr int WCOLS = TN * 32, UPR = WCOLS / 8, UNITS = 32 * UPR, KIT = UNITS / 64;
    static_assert(UNITS % 64 == 0, "a slab is a whole number of wave-instructions");
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));                              // (as below: lane coordinates re-derived behind an opaque copy)
    const int lane = tid_e & 63, l31 = tid_e & 31, hh = (tid_e >> 5) & 1;
    float* stg = reinterpret_cast<float*>(ring) + wave * (32 * WCOLS);
    float* bias_l = reinterpret_cast<float*>(ring) + NW * 32 * WCOLS;
    __syncthreads();                                             // every wave is done reading the last stage
    for (int c = tid_e; c < BN / 4; c += NW * 64) {
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (biasb && n0 + 4 * c < p.N) b4 = ldg16(biasb + n0 + 4 * c);
      *reinterpret_cast<float4*>(bias_l + 4 * c) = b4;
    }
    __syncthreads();
    typedef split_t bf16x8_t __attribute__((ext_vector_type(8)));
    auto wide_rows = [&](auto act_c, auto split_c, const float* slab, int row_base) {
      constexpr int ACT = decltype(act_c)::value;
      constexpr bool SPLIT = decltype(split_c)::value != 0;
      constexpr int U8 = 2;
      static_assert(KIT % U8 == 0, "unroll");
#pragma unroll 1
      for (int k0 = 0; k0 < KIT; k0 += U8) {
        float4 a8[U8][2], b8[U8][2];
        int rows[U8], cols[U8];
#pragma unroll
        for (int u = 0; u < U8; ++u) {
          const int unit = lane + 64 * (k0 + u);
          const int r = unit / UPR, c8 = (unit - r * UPR) * 8;
          rows[u] = row_base + r;
          cols[u] = c8;
          const float* sp = slab + r * WCOLS + c8;
          a8[u][0] = *reinterpret_cast<const float4*>(sp);
          a8[u][1] = *reinterpret_cast<const float4*>(sp + 4);
          b8[u][0] = *reinterpret_cast<const float4*>(bias_l + bcol0 + c8);
          b8[u][1] = *reinterpret_cast<const float4*>(bias_l + bcol0 + c8 + 4);
        }
#pragma unroll
        for (int u = 0; u < U8; ++u) {
          const int row = rows[u], col8 = n0 + bcol0 + cols[u];
          if (row < p.M && col8 < p.N) {
            float v[8] = {a8[u][0].x * p.alpha + b8[u][0].x, a8[u][0].y * p.alpha + b8[u][0].y, a8[u][0].z * p.alpha + b8[u][0].z,
                          a8[u][0].w * p.alpha + b8[u][0].w, a8[u][1].x * p.alpha + b8[u][1].x, a8[u][1].y * p.alpha + b8[u][1].y,
                          a8[u][1].z * p.alpha + b8[u][1].z, a8[u][1].w * p.alpha + b8[u][1].w};
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) v[q8] = ACT == 1 ? silu_f(v[q8]) : (ACT == 2 ? gelu_tanh_fast_f(v[q8]) : v[q8]);
            if constexpr (SPLIT) {
              bf16x8_t hi, lo;
#pragma unroll
              for (int q8 = 0; q8 < 8; ++q8) {
                hi[q8] = (split_t)v[q8];
                lo[q8] = (split_t)(v[q8] - (float)hi[q8]);
              }
              split_t* rowp = reinterpret_cast<split_t*>(Cb + (long long)row * p.ldc);
              bf16x8_t* dh = reinterpret_cast<bf16x8_t*>(rowp + split_idx(col8));
              bf16x8_t* dl = reinterpret_cast<bf16x8_t*>(rowp + split_idx(col8) + 32);
              if (p.st_plain) {                                   // (A/B runs: RGM_ST_PLAIN, launch2)
                *dh = hi;
                *dl = lo;
              } else {
                __builtin_nontemporal_store(hi, dh);
                __builtin_nontemporal_store(lo, dl);
              }
            } else {
              const f32x4 v0 = {v[0], v[1], v[2], v[3]}, v1 = {v[4], v[5], v[6], v[7]};
              f32x4* d0 = reinterpret_cast<f32x4*>(Cb + (long long)row * p.ldc + col8);
              if (p.st_plain) {
                d0[0] = v0;
                d0[1] = v1;
              } else {
                __builtin_nontemporal_store(v0, d0);
                __builtin_nontemporal_store(v1, d0 + 1);
              }
            }
          }
        }
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    static_for<0, TM>([&](auto im_c) {
      constexpr int im = decltype(im_c)::value;
      static_for<0, TN>([&](auto in_c) {
        constexpr int in = decltype(in_c)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) stg[((e & 3) + 8 * (e >> 2) + 4 * hh) * WCOLS + in * 32 + l31] = acc[im][in][e];
      });
      const int row_base = m0 + arow0 + im * 32;
      if (p.out_split) {
        if (p.act == 0) wide_rows(I0{}, I1{}, stg, row_base);
        else if (p.act == 1) wide_rows(I1{}, I1{}, stg, row_base);
        else wide_rows(I2{}, I1{}, stg, row_base);
      } else {
        if (p.act == 0) wide_rows(I0{}, I0{}, stg, row_base);
        else if (p.act == 1) wide_rows(I1{}, I0{}, stg, row_base);
        else wide_rows(I2{}, I0{}, stg, row_base);
      }
    });
  } else {
  const bool vec = vector_epilogue();
  // ---- epilogue (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)); optional split output
  // Vector path: the accumulators of one 32-row slab go through the (now idle) LDS ring so that every lane owns 4
  // consecutive columns of a row -> bias / gate / residual are read and C is written 16 B per lane, a full 128-B line
  // per 8 lanes, instead of 64 dword stores of two half-lines each (tools/gemm_stamp.py: the scalar epilogue cost
  // 18-34k cycles per tile, a third of the tile's lifetime).
  if (vec) {
    // lane coordinates re-derived from the thread index behind an opaque copy: the one-wave-per-SIMD kernels have no register to carry
    // them through the K loop, and the compiler parked `hh` in scratch -- one scratch_load + s_waitcnt vmcnt(0) in front of every row
    // loop, i.e. every slab waited for the previous slab's stores to retire
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane = tid_e & 63, l31 = tid_e & 31, hh = (tid_e >> 5) & 1;
    constexpr int WCOLS = TN * 32, LPR = WCOLS / 4, RPI = 64 / LPR;   // lanes per row, rows per wave-instruction
    static_assert(NW * 32 * WCOLS * 4 <= NSTAGE * STAGE, "epilogue slab must fit in the ring");
    __syncthreads();                                                  // every wave is done reading the last stage
    // all TM slabs of a wave staged at once when the ring has the room (every tile but 256x128): ONE row loop per tile
    constexpr bool ALL_IM = (size_t)TM * NW * 32 * WCOLS * 4 <= (size_t)NSTAGE * STAGE;
    float* stg = reinterpret_cast<float*>(ring) + wave * ((ALL_IM ? TM : 1) * 32 * WCOLS);
    const int lr = lane / LPR, lc = (lane % LPR) * 4;
    const int col = n0 + bcol0 + lc;
    const bool col_ok = col < p.N;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (biasb && col_ok) bv = ldg16(biasb + col);
    double gs[8] = {0., 0., 0., 0., 0., 0., 0., 0.};   // p.stats: this lane's column sums / sums of squares (fp64: see common.h)
    constexpr int NJ = 32 / RPI;
    auto write_slab = [&](auto im_c, float* slab) {
      constexpr int im = decltype(im_c)::value;
      static_for<0, TN>([&](auto in_c) {
        constexpr int in = decltype(in_c)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * hh) * WCOLS + in * 32 + l31] = acc[im][in][e];
      });
    };
    // Output stores.  The one-wave-per-SIMD kernels (PIPE 5) finish a whole round of 256 KB tiles at the same moment and their
    // epilogue runs at the chip's write rate: non-temporal stores (the 57-76 MB of a qkv / fc1 output pass through the 32 MB of L2
    // anyway) take 1.7-4.4 % off those launches (tools/which_kernel.py, same box: 86.9 -> 85.4 us at 224 tiles, 98.3 -> 94.0 at 256).
    auto out16 = [&](float* dst, const float (&v)[4]) {
      if constexpr (COH) {
        const f32x4 nv = {v[0], v[1], v[2], v[3]};
        store16_sc1(dst, nv);
      } else if constexpr (PIPE == 5) {
        const f32x4 nv = {v[0], v[1], v[2], v[3]};
        if (p.st_plain) *reinterpret_cast<f32x4*>(dst) = nv;          // (A/B runs: RGM_ST_PLAIN, launch2)
        else __builtin_nontemporal_store(nv, reinterpret_cast<f32x4*>(dst));
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      }
    };
    // split rows: 2 x 8 bytes per lane (hi halves, lo halves of its 4 columns), or lanes (2k, 2k+1) paired to one 16-byte store each
    // (p.split_pair, launch2).  Pairing halves the store instructions but pays two DPP moves and four selects per row in VALU: it wins where
    // other waves hide that (the LayerNorm rows, the 144-column kernel's two waves per SIMD) and LOSES on the one-wave-per-SIMD tiles --
    // 256x256 epilogue, isolated: 33.8 -> 39.4 k cycles with split rows, 39.0 -> 48.5 k with GELU (tools/gemm_stamp.py SPLIT=1) -- so those keep 2 x 8.
    typedef split_t bf16x4_t __attribute__((ext_vector_type(4)));
    auto split_store = [&](split_t* rowp, int col, const bf16x4_t& hi, const bf16x4_t& lo) {
      if (p.split_pair == 1) {
        if (PIPE == 5 && !p.st_plain) store_split4_pair<true>(rowp, col, hi, lo);
        else store_split4_pair<false>(rowp, col, hi, lo);
      } else if (PIPE == 5 && !p.st_plain) {
        __builtin_nontemporal_store(hi, reinterpret_cast<bf16x4_t*>(rowp + split_idx(col)));
        __builtin_nontemporal_store(lo, reinterpret_cast<bf16x4_t*>(rowp + split_idx(col) + 32));
      } else {
        *reinterpret_cast<bf16x4_t*>(rowp + split_idx(col)) = hi;
        *reinterpret_cast<bf16x4_t*>(rowp + split_idx(col) + 32) = lo;
      }
    };
    auto store_row = [&](int row, const float (&v)[4]) {
      if (p.out_split) {   // split-row output (common.h split_idx): 4 hi then, 32 further, 4 lo
        typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 hi, lo;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          hi[q4] = (split_t)v[q4];
          lo[q4] = (split_t)(v[q4] - (float)hi[q4]);
        }
        split_t* rowp = reinterpret_cast<split_t*>(Cb + (long long)row * p.ldc);
        if constexpr (COH) {
          store_split4_pair_sc1<1>(rowp, col, hi, lo);      // lanes (2k, 2k+1) own columns 8k' .. 8k'+7 of the same row
        } else {
          split_store(rowp, col, hi, lo);
        }
      } else if (exp != 4) {
        out16(Cb + (long long)row * p.ldc + col, v);
      }
    };
    // Three row bodies, chosen by uniform branches.  The general one carries every epilogue variant (activations and their
    // derivatives, gate, residual, statistics, split output): ~2.5 KB of code, so its row loop is ROLLED -- unrolled 4-16
    // times it was 20-40 KB of straight-line code, and the instruction cache is cold at every launch: a workgroup of the
    // first round (all of them on small grids) spent 11-25k cycles fetching it (tools/gemm_stamp.py; a second pass over the
    // same code in the same kernel runs in a third of the time).  A rolled loop with global loads in it waits vmcnt(0) every
    // iteration = for its own previous store, so the two cases the DiT forward runs get their own bodies:
    //   plain   (bias / SiLU / GELU / split output, nothing read per row): rolled and branch-free, the bias load is retired
    //           before the loop -> stores are fire-and-forget;
    //   linear  (act 0 + gate and/or residual: attention proj, fc2, the residual convs of the VAE with their statistics): a slab's
    //           gate / residual reads are all issued before its first row is finished, small unrolled body.
    const bool reads_rows = p.gate || resb || p.act >= 3;
    const bool plain = !reads_rows && !p.stats;
    const bool linear = !plain && p.act == 0 && reads_rows;
    // rows [0, nj * RPI) of `slab`; slab row 0 is global row row0 - lr
    auto rolled_rows = [&](const float* slab, int row0, int nj) {
      if (plain) {
        asm volatile("" : "+v"(bv.x), "+v"(bv.y), "+v"(bv.z), "+v"(bv.w));   // the bias has landed: no VMEM wait inside the loop
        auto plain_rows = [&](auto act_c, auto split_c) {      // branch-free body per (activation, output format)
          constexpr int ACT = decltype(act_c)::value;
          constexpr bool SPLIT = decltype(split_c)::value != 0;
          // one wave per SIMD (PIPE 5) has nobody to hide an iteration's LDS round trip + activation chain behind: U rows per
          // iteration, all slab reads first (the other kernels keep the rolled body: their cost is the cold instruction cache)
          constexpr int U = PIPE == 5 ? 4 : 1;
#pragma unroll 1
          for (int j0 = 0; j0 < nj; j0 += U) {
            float4 a4s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) a4s[u] = *reinterpret_cast<const float4*>(slab + ((j0 + u) * RPI + lr) * WCOLS + lc);   // same wave wrote it: LDS ops are in order
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int row = row0 + (j0 + u) * RPI;
              const float4 a4 = a4s[u];
              if (row < p.M && col_ok) {
                float v[4] = {a4.x * p.alpha + bv.x, a4.y * p.alpha + bv.y, a4.z * p.alpha + bv.z, a4.w * p.alpha + bv.w};
                if constexpr (SPLIT) {
                  if (p.C2) out16(Cb + (long long)row * p.ldc + col, v);       // second output: the pre-activation as fp32 rows in C
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) v[q4] = ACT == 1 ? silu_f(v[q4]) : (ACT == 2 ? gelu_tanh_fast_f(v[q4]) : v[q4]);
                if constexpr (SPLIT) {
                  typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
                  bf16x4 hi, lo;
#pragma unroll
                  for (int q4 = 0; q4 < 4; ++q4) {
                    hi[q4] = (split_t)v[q4];
                    lo[q4] = (split_t)(v[q4] - (float)hi[q4]);
                  }
                  split_t* rowp = p.C2 ? reinterpret_cast<split_t*>(p.C2 + (long long)z * p.sC + (long long)row * p.ldc2)
                                       : reinterpret_cast<split_t*>(Cb + (long long)row * p.ldc);
                  if constexpr (COH) {
                    store_split4_pair_sc1<1>(rowp, col, hi, lo);
                  } else {
                    split_store(rowp, col, hi, lo);
                  }
                } else {
                  if (exp != 4) out16(Cb + (long long)row * p.ldc + col, v);
                }
              }
            }
          }
        };
        // split rows with EIGHT columns per lane (p.split_pair == 2; the one-wave-per-SIMD tiles): 16 lanes per row, 4 rows per instruction --
        // a lane stores the 8 hi halves (16 bytes) and the 8 lo halves (16 bytes) of its columns itself: half the store instructions of the
        // 4-column mapping without the lane exchange of the paired variant (its DPP moves + selects are VALU time nobody hides here)
        auto plain_rows8 = [&](auto act_c) {
          constexpr int ACT = decltype(act_c)::value;
          constexpr int LPR8 = WCOLS / 8, RPI8 = 64 / LPR8;
          const int lr8 = lane / LPR8, lc8 = (lane % LPR8) * 8;
          const int col8 = n0 + bcol0 + lc8;
          const bool ok8 = col8 < p.N;
          float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
          if (biasb && ok8) {
            b0 = ldg16(biasb + col8);
            b1 = ldg16(biasb + col8 + 4);
          }
          asm volatile("" : "+v"(b0.x), "+v"(b0.y), "+v"(b0.z), "+v"(b0.w), "+v"(b1.x), "+v"(b1.y), "+v"(b1.z), "+v"(b1.w));
          const int rbase = row0 - lr + lr8;                      // row0 carries the 4-column mapping's row offset
          const int nj8 = nj * RPI / RPI8;
          typedef split_t bf16x8_t __attribute__((ext_vector_type(8)));
          constexpr int U8 = 2;
#pragma unroll 1
          for (int j0 = 0; j0 < nj8; j0 += U8) {
            float4 a8[U8][2];
#pragma unroll
            for (int u = 0; u < U8; ++u) {
              const float* sp = slab + ((j0 + u) * RPI8 + lr8) * WCOLS + lc8;
              a8[u][0] = *reinterpret_cast<const float4*>(sp);
              a8[u][1] = *reinterpret_cast<const float4*>(sp + 4);
            }
#pragma unroll
            for (int u = 0; u < U8; ++u) {
              const int row = rbase + (j0 + u) * RPI8;
              if (row < p.M && ok8) {
                float v[8] = {a8[u][0].x * p.alpha + b0.x, a8[u][0].y * p.alpha + b0.y, a8[u][0].z * p.alpha + b0.z, a8[u][0].w * p.alpha + b0.w,
                              a8[u][1].x * p.alpha + b1.x, a8[u][1].y * p.alpha + b1.y, a8[u][1].z * p.alpha + b1.z, a8[u][1].w * p.alpha + b1.w};
                if (p.C2) {
                  const float v0[4] = {v[0], v[1], v[2], v[3]}, v1[4] = {v[4], v[5], v[6], v[7]};
                  out16(Cb + (long long)row * p.ldc + col8, v0);
                  out16(Cb + (long long)row * p.ldc + col8 + 4, v1);
                }
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) v[q8] = ACT == 1 ? silu_f(v[q8]) : (ACT == 2 ? gelu_tanh_fast_f(v[q8]) : v[q8]);
                bf16x8_t hi, lo;
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) {
                  hi[q8] = (split_t)v[q8];
                  lo[q8] = (split_t)(v[q8] - (float)hi[q8]);
                }
                split_t* rowp = p.C2 ? reinterpret_cast<split_t*>(p.C2 + (long long)z * p.sC + (long long)row * p.ldc2)
                                     : reinterpret_cast<split_t*>(Cb + (long long)row * p.ldc);
                bf16x8_t* dh = reinterpret_cast<bf16x8_t*>(rowp + split_idx(col8));
                bf16x8_t* dl = reinterpret_cast<bf16x8_t*>(rowp + split_idx(col8) + 32);
                if (p.st_plain) {
                  *dh = hi;
                  *dl = lo;
                } else {
                  __builtin_nontemporal_store(hi, dh);
                  __builtin_nontemporal_store(lo, dl);
                }
              }
            }
          }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (PIPE == 5 && !COH && p.out_split && p.split_pair == 2) {
          if (p.act == 0) plain_rows8(I0{});
          else if (p.act == 1) plain_rows8(I1{});
          else plain_rows8(I2{});
        } else
        if (p.out_split) {
          if (p.act == 0) plain_rows(I0{}, I1{});
          else if (p.act == 1) plain_rows(I1{}, I1{});
          else plain_rows(I2{}, I1{});
        } else {
          if (p.act == 0) plain_rows(I0{}, I0{});
          else if (p.act == 1) plain_rows(I1{}, I0{});
          else plain_rows(I2{}, I0{});
        }
      } else {
        asm volatile("" : "+v"(bv.x), "+v"(bv.y), "+v"(bv.z), "+v"(bv.w));   // as above: without per-row reads (conv + statistics) no VMEM wait is left in the loop
#pragma unroll 1
        for (int j = 0; j < nj; ++j) {
          const int r = j * RPI + lr, row = row0 + j * RPI;
          const float4 a4 = *reinterpret_cast<const float4*>(slab + r * WCOLS + lc);
          if (row < p.M && col_ok) {
            float v[4] = {a4.x * p.alpha + bv.x, a4.y * p.alpha + bv.y, a4.z * p.alpha + bv.z, a4.w * p.alpha + bv.w};
            if (p.act == 1) {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) v[q4] = silu_f(v[q4]);
            } else if (p.act == 2) {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) v[q4] = gelu_tanh_fast_f(v[q4]);
            } else if (p.act == 3 || p.act == 4) {   // backward through an activation: times gelu'(aux) / silu'(aux)
              const float4 x4 = *reinterpret_cast<const float4*>(auxb + (long long)row * p.ldaux + col);
              const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) v[q4] *= (p.act == 3 ? gelu_tanh_grad_f(xs[q4]) : silu_grad_f(xs[q4]));
            }
            if (p.gate) {
              const float4 g = ldg16(p.gate + (long long)(row / p.rows_per_gate) * p.gate_ld + col);
              v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w;
            }
            if (resb) {
              const float4 rr = ldg16(resb + (long long)row * p.ldres + col);
              v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
            }
            if (p.stats) {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                gs[q4] += (double)v[q4];
                gs[4 + q4] += (double)v[q4] * (double)v[q4];
              }
            }
            store_row(row, v);
          }
        }
      }
    };
    auto linear_rows = [&](const float* slab, int row0) {      // one 32-row slab
      float4 g4[NJ], r4[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = row0 + j * RPI;
        g4[j] = make_float4(1.f, 1.f, 1.f, 1.f);
        r4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < p.M && col_ok) {
          if (p.gate) g4[j] = ldg16(p.gate + (long long)(row / p.rows_per_gate) * p.gate_ld + col);
          if (resb && exp != 8) r4[j] = ldg16(resb + (long long)row * p.ldres + col);
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = j * RPI + lr, row = row0 + j * RPI;
        const float4 a4 = *reinterpret_cast<const float4*>(slab + r * WCOLS + lc);
        if (row < p.M && col_ok) {
          const float v[4] = {(a4.x * p.alpha + bv.x) * g4[j].x + r4[j].x, (a4.y * p.alpha + bv.y) * g4[j].y + r4[j].y,
                              (a4.z * p.alpha + bv.z) * g4[j].z + r4[j].z, (a4.w * p.alpha + bv.w) * g4[j].w + r4[j].w};
          if (p.stats && exp != 7) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              gs[q4] += (double)v[q4];
              gs[4 + q4] += (double)v[q4] * (double)v[q4];
            }
          }
          store_row(row, v);
        }
      }
    };
    // The big tiles (one slab at a time, 4 slabs per wave): linear_rows unrolled 16 x 4 times is ~100 KB of straight-line code that one
    // wave per SIMD executes at the speed its instructions arrive -- 19 k cycles per slab on the residual convs of the VAE whatever was
    // removed from it (tools/conv_stamp.py with RGM_GEMM2_EXP 4 / 7 / 8 / 9: no stores / sums / residual loads / slab reads, 77-85 k cycles
    // of epilogue after 150 k of K loop every time).  Rolled instead: the slab's residual rows are requested up front as before, parked in
    // a second LDS slab beside the accumulators', and a 4-row loop reads both.  A 32-row slab meets at most two gate rows
    // (rows_per_gate >= 32): both are loaded before the loop.
    constexpr bool STAGED_LINEAR = !ALL_IM && (size_t)2 * NW * 32 * WCOLS * 4 <= (size_t)NSTAGE * STAGE;
    // The order of the memory operations is the point (vmcnt retires in order, loads and stores alike).  Per slab:
    //   accumulators -> LDS slab | wait for this slab's residual rows (LDS-DMA into a second slab: no registers, no compiler-placed
    //   wait) | pass 1, LDS -> LDS: (acc * alpha + bias) * gate + residual, GroupNorm sums | LDS-DMA of the NEXT slab's residual rows |
    //   pass 2: LDS -> global stores.
    // The next slab's rows are requested before this slab's stores are issued, so the wait for them is s_waitcnt vmcnt(<stores of
    // one slab>) and never waits for a store.  With the residual loads behind the previous slab's stores (linear_rows) every slab
    // sat out its own round trip AND the previous slab's last store: 15-19 k cycles per slab (tools/conv_stamp.py).
    auto staged_dma_res = [&](int row0, float* rslab) {
      if constexpr (COH) asm volatile("" ::: "memory");   // (COH stores are asm statements: keep the DMA where the counted waits expect it)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = row0 + j * RPI;
        const float* src = (resb && row < p.M && col_ok && exp != 8) ? resb + (long long)row * p.ldres + col
                                                                      : reinterpret_cast<const float*>(zero_page) + lc;
        dma16(src, reinterpret_cast<char*>(rslab) + j * 1024);
      }
      if constexpr (COH) asm volatile("" ::: "memory");
    };
    auto staged_pass1 = [&](float* slab, const float* rslab, int row0, float4 g_lo, float4 g_hi, int bnd) {
      constexpr int U = 4;
      static_assert(NJ % U == 0, "row loop unroll must divide the rows of a slab");
#pragma unroll 1
      for (int j0 = 0; j0 < NJ; j0 += U) {
        float4 a4s[U], r4s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          a4s[u] = *reinterpret_cast<const float4*>(slab + ((j0 + u) * RPI + lr) * WCOLS + lc);
          r4s[u] = *reinterpret_cast<const float4*>(rslab + ((j0 + u) * RPI + lr) * WCOLS + lc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int row = row0 + (j0 + u) * RPI;
          const float4 a4 = a4s[u], rr = r4s[u];
          const float4 g = row < bnd ? g_lo : g_hi;
          const float v[4] = {(a4.x * p.alpha + bv.x) * g.x + rr.x, (a4.y * p.alpha + bv.y) * g.y + rr.y,
                              (a4.z * p.alpha + bv.z) * g.z + rr.z, (a4.w * p.alpha + bv.w) * g.w + rr.w};
          if (p.stats && exp != 7 && row < p.M && col_ok) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              gs[q4] += (double)v[q4];
              gs[4 + q4] += (double)v[q4] * (double)v[q4];
            }
          }
          *reinterpret_cast<float4*>(slab + ((j0 + u) * RPI + lr) * WCOLS + lc) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    };
    auto staged_pass2 = [&](auto full_c, const float* slab, int row0) {
      constexpr bool FULL = decltype(full_c)::value != 0;
      constexpr int U = 4;
#pragma unroll 1
      for (int j0 = 0; j0 < NJ; j0 += U) {
        float4 a4s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a4s[u] = *reinterpret_cast<const float4*>(slab + ((j0 + u) * RPI + lr) * WCOLS + lc);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int row = row0 + (j0 + u) * RPI;
          if (FULL || (row < p.M && col_ok)) {
            const float v[4] = {a4s[u].x, a4s[u].y, a4s[u].z, a4s[u].w};
            store_row(row, v);
          }
        }
      }
    };
    auto reduce_stats = [&](bool coherent) {   // GroupNorm partial sums of this tile (uniform branch): lanes -> waves -> groups, all in a fixed order
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) gs[k] += __shfl_xor(gs[k], o, 64);
      }
      __syncthreads();                                                // every wave is done with its staging slab
      double* sred = reinterpret_cast<double*>(ring);                 // [wave][quad of its WCOLS columns][8]
      if (lr == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) sred[(wave * LPR + lane) * 8 + k] = gs[k];
      }
      __syncthreads();
      const int qpg = p.stats_gw >> 2;                                // column quads per group
      const int ngrp = BN / p.stats_gw;
      if (tid < ngrp && n0 + tid * p.stats_gw < p.N) {
        double sum = 0., sq = 0.;
        for (int qq = 0; qq < qpg; ++qq) {
          const int quad = tid * qpg + qq;                            // quad index inside the BN columns of the tile
          const int wc_ = quad / LPR, l = quad - wc_ * LPR;
          for (int wr_ = 0; wr_ < WM; ++wr_) {
            const double* r = sred + ((wr_ * WN + wc_) * LPR + l) * 8;
            sum += (r[0] + r[1]) + (r[2] + r[3]);
            sq += (r[4] + r[5]) + (r[6] + r[7]);
          }
        }
        double* o2 = p.stats + ((long long)(m0 / BM) * (p.N / p.stats_gw) + n0 / p.stats_gw + tid) * 2;
        if (coherent) {   // read by other workgroups of this launch: device-coherent stores (written through; no cache flush needed)
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(o2), (unsigned long long)__double_as_longlong(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(o2) + 1, (unsigned long long)__double_as_longlong(sq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          o2[0] = sum;
          o2[1] = sq;
        }
      }
    };
    bool gn_done = false;
    if constexpr (PIPE == 5 && ALOAD == 2 && !ALL_IM) {
      if (p.gn_count) {
        // ---- GroupNorm + swish of this conv's output inside the launch (GemmParams::gn_count).  Pass A: the tile's sums from the
        // accumulators (no store) -> partials to p.stats -> arrive at the image's counter and wait for the image's other tiles
        // (bounded) -> mean / rstd from all partials in tile order -> pass B: normalise, swish, split rows.
        const int row_w2 = m0 + arow0 + lr;
        static_for<0, TM>([&](auto im_c) {
          constexpr int im = decltype(im_c)::value;
          write_slab(im_c, stg);
          const int row0 = row_w2 + im * 32;
#pragma unroll 1
          for (int j0 = 0; j0 < NJ; j0 += 4) {
            float4 a4s[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a4s[u] = *reinterpret_cast<const float4*>(stg + ((j0 + u) * RPI + lr) * WCOLS + lc);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int row = row0 + (j0 + u) * RPI;
              if (row < p.M && col_ok) {
                const float v[4] = {a4s[u].x * p.alpha + bv.x, a4s[u].y * p.alpha + bv.y, a4s[u].z * p.alpha + bv.z, a4s[u].w * p.alpha + bv.w};
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                  gs[q4] += (double)v[q4];
                  gs[4 + q4] += (double)v[q4] * (double)v[q4];
                }
              }
            }
          }
        });
        // No fence: an agent-scope fence writes the L2 back (dirty with every tile's output) -- 176 against 168 ms per 64-latent decode
        // when each tile fenced twice.  The partials go out as device-coherent stores, are complete when vmcnt retires them, and only
        // then does the tile arrive; the readers use device-coherent loads for the counter and the partials and read nothing else.
        reduce_stats(true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* okf = reinterpret_cast<int*>(ring + 32768);  // behind reduce_stats' scratch
        const int tile_m = m0 / BM, tile_n = n0 / BN;
        if (tid == 0) {
          unsigned* cnt = p.gn_count + (long long)(tile_m / p.gn_tiles) * tiles_n + tile_n;
          __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          int ok = 0;
          if (!p.gn_force_fail) {
            while (true) {
              if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)p.gn_tiles) {
                ok = 1;
                break;
              }
              // 1 ms of the 100 MHz clock (a handful of tile times: 36-K-tile conv tiles run 100-200 us): a sibling tile is not resident --
              // another stream holds its CU.  Every such tile is counted by gn_fixup_kernel (rgm_gn_fallback_tiles); round 4 waited 4 ms.
              if (__builtin_amdgcn_s_memrealtime() - t0 > 100000ull) break;
              __builtin_amdgcn_s_sleep(16);
            }
          }
          *okf = ok;
          if (!ok) p.gn_fail[(long long)tile_m * tiles_n + tile_n] = 1;
        }
        __syncthreads();
        const int ok = *okf;
        // (mean, rstd) of the tile's groups: one thread per group sums the image's partials in tile order (the sums of
        // gn_finalize_tiles_kernel) and leaves the pair in LDS
        float2* mr = reinterpret_cast<float2*>(ring + 32768 + 64);
        {
          const int ngr = p.N / p.stats_gw, ngrp_t = BN / p.stats_gw;
          if (ok && tid < ngrp_t && n0 + tid * p.stats_gw < p.N) {
            const int g = n0 / p.stats_gw + tid;
            const unsigned long long* part = reinterpret_cast<const unsigned long long*>(p.stats) +
                                             ((long long)(tile_m / p.gn_tiles) * p.gn_tiles * ngr + g) * 2;
            double s_ = 0., ss_ = 0.;
            for (int t = 0; t < p.gn_tiles; ++t) {
              s_ += __longlong_as_double((long long)__hip_atomic_load(part + (long long)t * ngr * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
              ss_ += __longlong_as_double((long long)__hip_atomic_load(part + (long long)t * ngr * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            const double mean_d = s_ / p.gn_n;
            double var = ss_ / p.gn_n - mean_d * mean_d;
            if (var < 0.0) var = 0.0;
            mr[tid] = make_float2((float)mean_d, (float)(1.0 / sqrt(var + (double)p.gn_eps)));
          }
        }
        __syncthreads();                                 // the pairs are in LDS; every wave has read the flag
        float mean = 0.f, rstd = 1.f;
        float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && col_ok) {
          const float2 m2 = mr[(col - n0) / p.stats_gw];
          mean = m2.x;
          rstd = m2.y;
          ga = *reinterpret_cast<const float4*>(p.gn_gamma + col);
          be = *reinterpret_cast<const float4*>(p.gn_beta + col);
        }
        __syncthreads();                                 // ... and the pairs: pass B may overwrite the slabs
        static_for<0, TM>([&](auto im_c) {
          constexpr int im = decltype(im_c)::value;
          write_slab(im_c, stg);
          const int row0 = row_w2 + im * 32;
#pragma unroll 1
          for (int j0 = 0; j0 < NJ; j0 += 4) {
            float4 a4s[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a4s[u] = *reinterpret_cast<const float4*>(stg + ((j0 + u) * RPI + lr) * WCOLS + lc);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int row = row0 + (j0 + u) * RPI;
              if (row < p.M && col_ok) {
                const float v[4] = {a4s[u].x * p.alpha + bv.x, a4s[u].y * p.alpha + bv.y, a4s[u].z * p.alpha + bv.z, a4s[u].w * p.alpha + bv.w};
                if (ok) {
                  float o[4] = {(v[0] - mean) * rstd * ga.x + be.x, (v[1] - mean) * rstd * ga.y + be.y, (v[2] - mean) * rstd * ga.z + be.z,
                                (v[3] - mean) * rstd * ga.w + be.w};
                  if (p.gn_swish) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) o[q4] = silu_fast_f(o[q4]);
                  }
                  store_row(row, o);
                } else {
                  out16(Cb + (long long)row * p.ldc + col, v);     // raw fp32 rows: gn_fixup converts the tile in place
                }
              }
            }
          }
        });
        gn_done = true;
      }
    }
    const int row_w = m0 + arow0 + lr;                          // this lane's row in slab row lr of the wave's first slab
    if (gn_done) {
    } else if constexpr (ALL_IM) {
      static_for<0, TM>([&](auto im_c) { write_slab(im_c, stg + decltype(im_c)::value * 32 * WCOLS); });
      if (linear) {
        static_for<0, TM>([&](auto im_c) { linear_rows(stg + decltype(im_c)::value * 32 * WCOLS, row_w + decltype(im_c)::value * 32); });
      } else {
        rolled_rows(stg, row_w, TM * NJ);
      }
    } else {
      auto estamp = [&](int i) {                                 // DBG: cycles since the end of the K loop (tools/gemm_stamp.py)
        if (DBG && rec && lane == 0) {
          __builtin_amdgcn_sched_barrier(0);
          const unsigned long long now_ = __builtin_amdgcn_s_memtime();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          dbg[32 + wave * 8 + i] = (long long)(now_ - t_loop_end);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      estamp(0);
      if (STAGED_LINEAR && linear) {       // (launch2 requires rows_per_gate >= 32 of these tiles)
        if constexpr (STAGED_LINEAR) {
          float* rslab = reinterpret_cast<float*>(ring) + (NW + wave) * 32 * WCOLS;
          const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
          // gate rows of all TM slabs up front (a 32-row slab meets at most two): nothing is outstanding yet, so the wait for them is free
          float4 g_lo[TM], g_hi[TM];
          int bnd[TM];
          static_for<0, TM>([&](auto im_c) {
            constexpr int im = decltype(im_c)::value;
            g_lo[im] = g_hi[im] = make_float4(1.f, 1.f, 1.f, 1.f);
            bnd[im] = 0x7fffffff;
            if (p.gate) {
              const int g0 = (m0 + arow0 + im * 32) / p.rows_per_gate, glast = (p.M - 1) / p.rows_per_gate;
              bnd[im] = (g0 + 1) * p.rows_per_gate;
              if (col_ok) {
                g_lo[im] = ldg16(p.gate + (long long)min(g0, glast) * p.gate_ld + col);
                g_hi[im] = ldg16(p.gate + (long long)min(g0 + 1, glast) * p.gate_ld + col);
              }
            }
          });
          staged_dma_res(row_w, rslab);
          static_for<0, TM>([&](auto im_c) {
            constexpr int im = decltype(im_c)::value;
            write_slab(im_c, stg);
            if (im == 0) {
              if (DBG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              estamp(1);
            }
            // this slab's residual rows have landed: only the stores of the previous slab's pass 2 were issued after their DMA
            // (the counted waits below assume exactly ONE store per row and lane -- store_row's out16, or its 16-byte half of a lane pair's
            // split row -- and TWO for unpaired split rows, behind the DMA; the experiment variants that drop stores wait for everything)
            if (im == 0 || !full || exp != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (p.out_split && p.split_pair != 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ) : "memory");
            staged_pass1(stg, rslab, row_w + im * 32, g_lo[im], g_hi[im], bnd[im]);
            if constexpr (im + 1 < TM) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // pass 1 has read the residual slab: the DMA may overwrite it
              staged_dma_res(row_w + (im + 1) * 32, rslab);
            }
            if (full) staged_pass2(std::integral_constant<int, 1>{}, stg, row_w + im * 32);
            else staged_pass2(std::integral_constant<int, 0>{}, stg, row_w + im * 32);
            if (im == 0) estamp(2);
            if (im == 1) estamp(3);
          });
        }
      } else
      static_for<0, TM>([&](auto im_c) {
        write_slab(im_c, stg);
        if (decltype(im_c)::value == 0) {
          if (DBG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          estamp(1);
        }
        if constexpr (STAGED_LINEAR) {
          rolled_rows(stg, row_w + decltype(im_c)::value * 32, NJ);
        } else {
          if (linear) linear_rows(stg, row_w + decltype(im_c)::value * 32);
          else rolled_rows(stg, row_w + decltype(im_c)::value * 32, NJ);
        }
        if (decltype(im_c)::value == 0) estamp(2);
        if (decltype(im_c)::value == 1) estamp(3);
      });
      estamp(4);
      if (DBG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        estamp(5);
      }
    }
    if (p.stats && !gn_done) reduce_stats(false);
  } else
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
            float v = acc[im][in][e] * p.alpha + bv;
            if (p.act == 1) v = silu_f(v);
            else if (p.act == 2) v = gelu_tanh_f(v);
            else if (p.act == 3) v *= gelu_tanh_grad_f(auxb[(long long)row * p.ldaux + col]);
            else if (p.act == 4) v *= silu_grad_f(auxb[(long long)row * p.ldaux + col]);
            if (p.gate) v *= p.gate[(long long)(row / p.rows_per_gate) * p.gate_ld + col];
            if (resb) v += resb[(long long)row * p.ldres + col];
            if (p.out_split) {   // split-row output (common.h split_idx)
              split_t* rowp = reinterpret_cast<split_t*>(Cb + (long long)row * p.ldc);
              const split_t hi = (split_t)v;
              rowp[split_idx(col)] = hi;
              rowp[split_idx(col) + 32] = (split_t)(v - (float)hi);
            } else {
              if (exp != 4) Cb[(long long)row * p.ldc + col] = v;
              else if (v == 123.456f) Cb[0] = v;   // timing experiment: keep the math, drop the stores
            }
          }
        }
      }
    });
  });
  }   // narrow wave tiles
  if (DBG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (PIPE) tacc[5] = t_end - t_loop_end;       // epilogue: K loop end -> C stores retired
    if (rec && lane == 0) {
#pragma unroll
      for (int i = 0; i < 7; ++i) dbg[wave * 8 + i] = (long long)tacc[i];
      dbg[wave * 8 + 7] = KT;
    }
  }
#undef RGM_STAMP
}

}  // namespace rgm
