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

namespace rgm {

typedef split_t bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  // 16 B per lane, LDS destination = wave-uniform base + lane*16
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// the kernel body: output tile `bid` (raster order) of batch element `z`.  A function so that ONE launch can mix tile shapes
// (gemm2_dual_kernel below); gemm2_kernel is the plain one-shape wrapper.
template <int BM, int BN, int WM, int WN, int ALOAD, int NSTAGE, int DBG = 0, int PIPE = 0>
__device__ __forceinline__ void gemm2_body(const GemmParams& p, const char* __restrict__ zero_page, int tiles_m, int tiles_n, int exp,
                                           long long* __restrict__ dbg, const int bid, const int z, const int rec_bid) {
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
  const int m0 = (first_m + in_g % gsz) * BM;
  const int n0 = (in_g / gsz) * BN;

  const int tid = threadIdx.x;
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
    struct FragsK {
      bf16x8 a[TM][2], b[TN][2];                          // [frag][hi, lo] of one k16 step
    };
    auto read_one = [&](FragsK& f, const char* As, auto stc, auto jc) {
      constexpr int st = decltype(stc)::value, j = decltype(jc)::value;
      constexpr int fi = j / 2, lo = j % 2;
      const int chunk = ((4 * lo + 2 * st + hh) ^ rq) << 4;
      if constexpr (fi < TM) {
        f.a[fi][lo] = *reinterpret_cast<const bf16x8*>(As + (arow0 + fi * 32 + l31) * 128 + chunk);
      } else {
        f.b[fi - TM][lo] = *reinterpret_cast<const bf16x8*>(As + BM * 128 + (bcol0 + (fi - TM) * 32 + l31) * 128 + chunk);
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
        acc[im][in] = RGM_MFMA_SPLIT_32x32x16(cur.a[im][t == 0 ? 1 : 0], cur.b[in][t == 1 ? 1 : 0], acc[im][in], 0, 0, 0);
        if constexpr (READ && m % EV == 0 && m / EV < NRD) read_one(nxt, rd, stnc, std::integral_constant<int, m / EV>{});
        if constexpr (AIM && m % EV == 1 && m / EV < APIECES) aim_piece(std::integral_constant<int, m / EV>{}, aim);
        if constexpr (DMA && m % EV == 1 && m / EV < SPW) {
          constexpr int i = m / EV;
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
    if (biasb && col_ok) bv = *reinterpret_cast<const float4*>(biasb + col);
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
    typedef split_t bf16x4_t __attribute__((ext_vector_type(4)));
    auto out16 = [&](float* dst, const float (&v)[4]) {
      if constexpr (PIPE == 5) {
        const f32x4 nv = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(nv, reinterpret_cast<f32x4*>(dst));
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      }
    };
    auto out8 = [&](split_t* dst, const bf16x4_t& v) {
      if constexpr (PIPE == 5) __builtin_nontemporal_store(v, reinterpret_cast<bf16x4_t*>(dst));
      else *reinterpret_cast<bf16x4_t*>(dst) = v;
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
        out8(rowp + split_idx(col), hi);
        out8(rowp + split_idx(col) + 32, lo);
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
                  out8(rowp + split_idx(col), hi);
                  out8(rowp + split_idx(col) + 32, lo);
                } else {
                  if (exp != 4) out16(Cb + (long long)row * p.ldc + col, v);
                }
              }
            }
          }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
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
              const float4 g = *reinterpret_cast<const float4*>(p.gate + (long long)(row / p.rows_per_gate) * p.gate_ld + col);
              v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w;
            }
            if (resb) {
              const float4 rr = *reinterpret_cast<const float4*>(resb + (long long)row * p.ldres + col);
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
          if (p.gate) g4[j] = *reinterpret_cast<const float4*>(p.gate + (long long)(row / p.rows_per_gate) * p.gate_ld + col);
          if (resb && exp != 8) r4[j] = *reinterpret_cast<const float4*>(resb + (long long)row * p.ldres + col);
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
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = row0 + j * RPI;
        const float* src = (resb && row < p.M && col_ok && exp != 8) ? resb + (long long)row * p.ldres + col
                                                                      : reinterpret_cast<const float*>(zero_page) + lc;
        dma16(src, reinterpret_cast<char*>(rslab) + j * 1024);
      }
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
              if (__builtin_amdgcn_s_memrealtime() - t0 > 400000ull) break;     // 4 ms of the 100 MHz clock: a sibling tile is not resident
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
                g_lo[im] = *reinterpret_cast<const float4*>(p.gate + (long long)min(g0, glast) * p.gate_ld + col);
                g_hi[im] = *reinterpret_cast<const float4*>(p.gate + (long long)min(g0 + 1, glast) * p.gate_ld + col);
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
            // (the counted waits below assume exactly ONE store per row and lane for fp32 rows and TWO for split rows -- store_row's out16 /
            // out8 pair -- behind the DMA; the experiment variants that drop stores wait for everything)
            if (im == 0 || !full || exp != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (p.out_split) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
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
  const size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
  static bool attr0 = false, attr1 = false;
  auto k0 = gemm2_kernel<BM, BN, WM, WN, 0, NSTAGE, 0, PIPE>;
  auto k1 = gemm2_kernel<BM, BN, WM, WN, PIPE == 4 ? 0 : 1, NSTAGE, 0, PIPE>;
  auto k2 = gemm2_kernel<BM, BN, WM, WN, PIPE == 5 ? 2 : 0, NSTAGE, 0, PIPE>;   // implicit conv with channel-block-major K (PIPE 5 only)
  RGM_REQUIRE(!p.conv_kmajor || (p.aload == 1 && p.Cin % 32 == 0), "gemm2: conv_kmajor is a property of the implicit 3x3 conv (aload == 1)");
  RGM_REQUIRE(PIPE != 4 || p.aload == 0, "gemm2: the loader/consumer kernels take dense operands only");
  if (lds > 65536) {
    if (p.aload == 0 && !attr0) {
      RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr0 = true;
    }
    if (p.aload == 1 && !attr1) {
      RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (PIPE == 5) RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr1 = true;
    }
  }
  dim3 grid(tm * tn, 1, p.batch), block(WM * WN * 64 * (PIPE == 4 ? 2 : 1));
  GemmParams pr = p;
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
  if (p.aload == 0 && g_dbg) {
    auto kd = gemm2_kernel<BM, BN, WM, WN, 0, NSTAGE, 1, PIPE>;
    static bool attrd = false;
    if (lds > 65536 && !attrd) {
      RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attrd = true;
    }
    hipLaunchKernelGGL(kd, grid, block, lds, s, pr, (const char*)g_zero_page, tm, tn, g_exp, g_dbg);
  } else if (PIPE == 5 && p.conv_kmajor && g_dbg) {          // the channel-block-major conv of the one-wave-per-SIMD kernels (tools/conv_stamp.py)
    auto kd = gemm2_kernel<BM, BN, WM, WN, PIPE == 5 ? 2 : 0, NSTAGE, 1, PIPE>;
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

static int g_t144 = getenv("RGM_T144") ? atoi(getenv("RGM_T144")) : 9;   // which grids take the 128x144 tiles (bit mask, gemm2_launch)
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
  const int lane = threadIdx.x & 63;
  const int nv = p.N >> 2;
  const long long MN = (long long)p.M * p.N;
  // one wave holds the row and there are only M waves (1024 at B = 4): every load of the row -- S partial sums, bias, gate, residual
  // per chunk -- is issued before the first sum (compile-time S and MAXV), or the wave walks through 5 S dependent round trips
  float4 part[MAXV][S], bq[MAXV], gq[MAXV], rq[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    const int col = c * 4;
    const bool ok = c < nv;
#pragma unroll
    for (int sidx = 0; sidx < S; ++sidx)
      part[i][sidx] = ok ? *reinterpret_cast<const float4*>(P + sidx * MN + (long long)row * p.N + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    bq[i] = (ok && p.bias) ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    gq[i] = (ok && p.gate) ? *reinterpret_cast<const float4*>(p.gate + (long long)(row / p.rows_per_gate) * p.gate_ld + col) : make_float4(1.f, 1.f, 1.f, 1.f);
    rq[i] = (ok && p.res) ? *reinterpret_cast<const float4*>(p.res + (long long)row * p.ldres + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nv) {
      const int col = c * 4;
      float4 a = part[i][0];
#pragma unroll
      for (int sidx = 1; sidx < S; ++sidx) {
        const float4 b = part[i][sidx];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      float w[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
      if (p.bias) { w[0] += bq[i].x; w[1] += bq[i].y; w[2] += bq[i].z; w[3] += bq[i].w; }
      if (p.gate) { w[0] *= gq[i].x; w[1] *= gq[i].y; w[2] *= gq[i].z; w[3] *= gq[i].w; }
      if (p.res) { w[0] += rq[i].x; w[1] += rq[i].y; w[2] += rq[i].z; w[3] += rq[i].w; }
      v[i] = make_float4(w[0], w[1], w[2], w[3]);
      *reinterpret_cast<float4*>(p.C + (long long)row * p.ldc + col) = v[i];
    }
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float D = (float)p.N;
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / D + p.ln_eps);
  const long long mo = (long long)(row / p.ln_rows_per_batch) * p.ln_mod_ld;
  float4* orow = reinterpret_cast<float4*>(p.ln_out + (long long)row * p.N);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c >= nv) continue;
    float4 y = make_float4((v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd);
    const float4 sc = reinterpret_cast<const float4*>(p.ln_scale + mo)[c], sh = reinterpret_cast<const float4*>(p.ln_shift + mo)[c];
    y = make_float4(y.x * (1.f + sc.x) + sh.x, y.y * (1.f + sc.y) + sh.y, y.z * (1.f + sc.z) + sh.z, y.w * (1.f + sc.w) + sh.w);
    if (p.ln_out_split) {
      typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 hi, lo;
      hi[0] = (split_t)y.x; hi[1] = (split_t)y.y; hi[2] = (split_t)y.z; hi[3] = (split_t)y.w;
      lo[0] = (split_t)(y.x - (float)hi[0]); lo[1] = (split_t)(y.y - (float)hi[1]);
      lo[2] = (split_t)(y.z - (float)hi[2]); lo[3] = (split_t)(y.w - (float)hi[3]);
      split_t* rp = reinterpret_cast<split_t*>(p.ln_out + (long long)row * p.N);
      const int si = split_idx(c * 4);
      *reinterpret_cast<bf16x4*>(rp + si) = hi;
      *reinterpret_cast<bf16x4*>(rp + si + 32) = lo;
    } else {
      orow[c] = y;
    }
  }
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
    const int KT = p.K >> 5;
    if ((g_t144 & 1) && t144 >= 224 && t144 <= 256 && t128 > 256) {
      GemmParams q = p;
      q.tile = 81;
      q.ln_out = nullptr;
      return gemm2_launch(q, s);
    }
    if ((g_t144 & 8) && p.sk_ws && t144 <= 64 && KT >= 72 && p.act == 0 && !p.out_split) {
      int best = 1;
      for (int c = 2; c <= 8; ++c) {
        if (KT % c || KT / c < 18 || t144 * c > 256) continue;
        if ((size_t)c * p.M * p.N * sizeof(float) + GEMM_SK_FLAG_BYTES > p.sk_ws_bytes) continue;
        best = c;
      }
      if (best > 1 && t144 * best >= 192) {
        S = best;
        sk_tile = 81;
      }
    }
  }
  const bool big_ok = S == 1 && p.tile == 0 && g_big_tiles && !p.aload && p.batch == 1 && !p.stats && p.act < 3 && !p.aux && p.M >= 2048 &&
                      (!p.gate || p.rows_per_gate >= 32) &&     // the big tiles' gate / residual epilogue: at most two gate rows per 32-row slab

                      ((p.N | p.ldc | p.ldres | p.gate_ld) & 3) == 0 &&
                      (((uintptr_t)p.C | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.gate) & 15) == 0;
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
