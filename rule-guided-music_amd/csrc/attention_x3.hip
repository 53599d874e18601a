// attention_x3.hip -- the RotaryAttention core of attention.hip in bf16x3 arithmetic (the bf16x3 / bf16x3_presplit modes).
//
// Same algorithm, reference (guided_diffusion/dit.py:263-277, rotary-embedding-torch 0.3.2) and work decomposition
// as attention.hip -- one workgroup per (sample, head), K and V of the head resident in LDS, single pass, scores
// computed transposed (S^T = K . Q^T) so that the probabilities stay in registers as the B operand of
// O^T = V^T . P^T -- but both products run on v_mfma_f32_32x32x16_bf16 with every operand split x ~= hi + lo
// (a*b ~= al*bh + ah*bl + ah*bh, fp32 accumulate), like the GEMMs of these modes.  The fp32 kernel spends 43 of its
// 64 us (T = 256, hd = 72) in v_mfma_f32_32x32x2_f32; the same contraction is 5.3x fewer matrix-pipe cycles here.
//
// LDS images (both split once while staging, 16-byte slots, odd slot strides -> conflict-free ds_read_b128 groups):
//   K   [key][ hi: KP bf16 | lo: KP bf16 | pad ]      KP = hd rounded up to 16 (zeros beyond hd), A operand of S^T:
//       lane (key, half) reads d = 16j + 8*half .. +7 of both planes;
//   V^T [d][ hi: TP bf16 | lo: TP bf16 | pad ]        keys PERMUTED inside every 32-group so that the 8 keys a lane's
//       probabilities cover in one k16 step (C/D layout of S^T: register r <-> key (r&3) + 8*(r>>2) + 4*half) are 16
//       contiguous bytes: position 16*h2 + 8*half + j  <->  key (j&3) + 8*(2*h2 + (j>>2)) + 4*half.
#include <stdlib.h>
#include "common.h"
#include "attention_x3_body.h"

namespace rgm {

template <int HD, int NKT>
__global__ __launch_bounds__(512) void rotary_attention_x3_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                                  const float* __restrict__ cos_tab,
                                                                  const float* __restrict__ sin_tab, int T, int heads, int rot_half,
                                                                  float* __restrict__ lse, int out_split, int qgroups) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  attn_x3_body<HD, NKT>(smem3, qkv, o, cos_tab, sin_tab, T, heads, rot_half, lse, out_split, qgroups, (int)blockIdx.x);
}

static int g_attn_pairs = getenv("RGM_ATTN_PAIRS") ? atoi(getenv("RGM_ATTN_PAIRS")) : 0;   // rgm_set_attn_pairs

template <int HD, int NKT>
static int launch_attn_x3(const float* qkv, float* o, const float* ct, const float* st, int N, int T, int heads, int rot_half,
                          float* lse, int out_split, hipStream_t s) {
  constexpr int KP = (HD + 15) / 16 * 16, TP = NKT * 32;
  size_t lds = (size_t)TP * (KP * 4 + 16) + (size_t)HD * (TP * 4 + 16);
  // ONE workgroup per CU, enforced (common.h attn_prepare_kernel).  At T <= 128 the images take <= 80 KiB, waves 4-7 have no
  // query tile and retire right after the staging barrier, and a second workgroup would move in beside waves 0-3: the co-residency under
  // which this kernel returned wrong rows in round 3 (1 launch in ~10 at N = 48, T = 128).
  static const int allow_two = RGM_EXP_ENV("RGM_ATTN_TWO_PER_CU");      // experiments only (common.h): reproduce the hazard
  auto kern = rotary_attention_x3_kernel<HD, NKT>;
  // Round 5: with the two-phase Q prologue (attention_x3_body.h) the probe of DESIGN 4h stays clean at TWO workgroups per CU -- 0 wrong
  // workgroups of 4.6 million (3 x 1000 launches x 1536, profiles/r05_attn_hazard_two_phase_n96.txt; the interleaved prologue on the same
  // box: 22 of 40 launches wrong).  rgm_set_attn_pairs(1) lifts the guard for THIS instantiation only (hd = 72, T <= 128: C5's half
  // windows): four waves (one per query tile) and the exact 79 KiB, so that two workgroups share a CU.
  if (HD == 72 && NKT == 4 && g_attn_pairs && !allow_two) {
    static bool attr_p = false;
    if (!attr_p) RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_p = true;
    const int nqt_p = (T + 31) / 32;
    hipLaunchKernelGGL(kern, dim3(N * heads), dim3(64 * nqt_p), lds, s, qkv, o, ct, st, T, heads, rot_half, lse, out_split, 1);
    RGM_LAUNCH_CHECK();
    return RGM_OK;
  }
  if (allow_two) {
    static bool attr2 = false;
    if (!attr2) RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr2 = true;
  } else {
    lds = attn_lds_one_per_cu(lds);
    static bool prepared = false;
    if (!prepared) RGM_TRY(attn_prepare_kernel(kern, 512, lds, "rotary_attention_x3_kernel"));
    prepared = true;
  }
  // nine query tiles on eight waves (T = 257) in a grid that does not fill the chip: two workgroups per (sample, head), one tile per wave
  const int nqt = (T + 31) / 32;
  const int split = attn_split_mode();
  const int qgroups = (nqt > 8 && (split < 0 ? (long long)N * heads <= ATTN_SPLIT_MAX_PAIRS : split == 1)) ? 2 : 1;
  hipLaunchKernelGGL(kern, dim3(N * heads * qgroups), dim3(512), lds, s, qkv, o, ct, st, T, heads, rot_half, lse, out_split, qgroups);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Key-blocked variant for 128 < T <= 256 (the DiT's own shape).  The single-pass kernel above runs its phases in series and every
// workgroup of the launch (one per CU: K and V of a head fill the LDS) is in the same phase at the same time: 14 us of global
// reads at the chip's bandwidth, then 10 us of MFMA with the memory idle (tools/attn_stamps.py).  Here the keys go through two
// LDS buffers in blocks of 64: block b+2's rows are requested before block b+1's products start and written to LDS (rotated,
// split, V transposed) after them, so reads and MFMAs overlap inside every workgroup; the softmax is the running-maximum form
// (m, l per query; the output accumulators are rescaled by 2^(m_old - m_new) once per block, exact for the first block).
// Same operand layouts, same MFMA order inside a block, same outputs (o, lse) as the kernel above.
template <int HD>
__global__ __launch_bounds__(512) void rotary_attention_x3_blocked_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                                          const float* __restrict__ cos_tab,
                                                                          const float* __restrict__ sin_tab, int T, int heads, int rot_half,
                                                                          float* __restrict__ lse, int out_split, int stagger, int qsplit) {
  constexpr int KP = (HD + 15) / 16 * 16;
  constexpr int KS = KP / 16;
  constexpr int DT = (HD + 31) / 32;
  constexpr int KB = 64;                    // keys per block = 2 key tiles
  constexpr int KROW = KP * 4 + 16;
  constexpr int VROW = KB * 4 + 16;         // V^T row of a block: 64 hi | 64 lo | pad (17 slots: odd)
  constexpr int KBYTES = KB * KROW, BUF = KBYTES + HD * VROW;
  constexpr int CPR = KP / 4;               // float4 chunks per padded K row
  constexpr int CHUNKS = KB * CPR;          // per block
  constexpr int SLOTS = (CHUNKS + 511) / 512;
  static_assert((KROW / 16) % 2 == 1 && (VROW / 16) % 2 == 1, "slot strides must be odd");
  extern __shared__ __attribute__((aligned(16))) char smem3[];

  // qsplit (1, 2, 4; round 5): the 256 queries of a (sample, head) over qsplit workgroups -- every one stages all of K / V (its eight waves
  // share that work as before), its first 8 / qsplit waves own 32 queries each and the others skip the MFMA sections.  Small batches leave
  // most CUs without a (sample, head) otherwise (B = 4: 64 workgroups).  Per query the arithmetic and its order are unchanged: identical rows.
  const int pair = blockIdx.x / qsplit, qs = blockIdx.x - pair * qsplit;
  const int n = pair / heads, head = pair - n * heads;
  const int D = heads * HD, D3 = 3 * D;
  const float* base = qkv + (long long)n * T * D3 + head * HD;
  const int tid = threadIdx.x;
  const int R = 2 * rot_half;
  typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int nb = (T + KB - 1) / KB;
#ifdef RGM_EXPERIMENTS   // staggered starts: workgroup classes (blockIdx % groups) begin `delay` x 1024 cycles (~0.5 us) apart (RGM_ATTN_STAGGER=groups*100+delay)
  if (stagger) {
    const int groups = stagger / 100, delay = stagger % 100;
    for (int i = 0; i < (int)(blockIdx.x % groups) * delay; ++i) __builtin_amdgcn_s_sleep(16);
  }
#endif

  // ---- staging registers of one block: SLOTS chunks of K (+ their rotary factors) and VSLOTS chunks of V per thread.
  // K chunks are indexed key-major (consecutive lanes = consecutive 16-byte chunks of a row: coalesced reads, row-major LDS writes).
  // V chunks are indexed so that ONE ds_write_b16 of a wave covers 32 keys of a 32-group x 2 channel chunks of opposite parity:
  // the V^T row stride is 4 (mod 8) dwords (16-byte slots, odd slot stride for the ds_read_b128 side), so the banks of a write are
  // 16 * (chunk & 1) + 4 i + pos / 2 -- 32 different ones for that set of lanes, and only 2 for the 18 chunks of one key, which is what
  // key-major lanes gave it (9-way conflicts on every one of the 8 transposing writes per chunk: most of the staging time).
  constexpr int VCH = (HD + 3) / 4;          // 16-byte chunks of a V row
  constexpr int VPAIRS = (VCH + 1) / 2;
  constexpr int VITEMS = (KB / 32) * VPAIRS * 64;
  constexpr int VSLOTS = (VITEMS + 511) / 512;
  float4 kq[SLOTS], vq[VSLOTS];
  // Rotary factors: ONE LDS table (c0, s0, c1, s1) per (position, 4-channel rotary chunk), built once per workgroup.  Read straight
  // from the global cos / sin tables they were 4 dword loads per K chunk and per Q chunk -- 60 % of the load instructions of the
  // kernel, and the texture addresser takes a wave's 64 dwords no faster than its 64 float4s (tools/attn_stamps.py: the "memory"
  // phases were bound by the number of load instructions, not by bytes or latency).
  const int NRC = R >> 2;
  float4* cs_lds = reinterpret_cast<float4*>(smem3 + 2 * BUF);
  auto v_item = [&](int sl, int& key, int& d0) {     // false: no chunk in this slot
    const int w = tid + sl * 512;
    const int wv = w >> 6, ln = w & 63;
    const int g32 = wv % (KB / 32), it = wv / (KB / 32);
    key = g32 * 32 + (ln & 31);
    d0 = (2 * it + (ln >> 5)) * 4;
    return w < VITEMS && d0 < HD;
  };
  auto request = [&](int b) {
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      const int c = tid + sl * 512;
      const int key = c / CPR, d0 = (c - key * CPR) * 4, kg = b * KB + key;
      kq[sl] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < CHUNKS && kg < T && d0 < HD) kq[sl] = *reinterpret_cast<const float4*>(base + (long long)kg * D3 + D + d0);
    }
#pragma unroll
    for (int sl = 0; sl < VSLOTS; ++sl) {
      int key, d0;
      vq[sl] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v_item(sl, key, d0) && b * KB + key < T) vq[sl] = *reinterpret_cast<const float4*>(base + (long long)(b * KB + key) * D3 + 2 * D + d0);
    }
  };
  auto deposit = [&](char* buf, int b) {     // rotate, split, write K rows and the permuted V^T rows of block b
    char* Ks = buf;
    char* Vt = buf + KBYTES;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      const int c = tid + sl * 512;
      if (c >= CHUNKS) continue;
      const int key = c / CPR, d0 = (c - key * CPR) * 4;
      const float4 x = kq[sl];
      float4 f = make_float4(1.f, 0.f, 1.f, 0.f);                                              // (1, 0) outside the rotary channels
      if (d0 < R) f = cs_lds[min(b * KB + key, T - 1) * NRC + (d0 >> 2)];
      const float kr[4] = {x.x * f.x - x.y * f.y, x.y * f.x + x.x * f.y, x.z * f.z - x.w * f.w, x.w * f.z + x.z * f.w};
      bf16x4 hi, lo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hi[i] = (split_t)kr[i];
        lo[i] = (split_t)(kr[i] - (float)hi[i]);
      }
      char* krp = Ks + key * KROW + d0 * 2;
      *reinterpret_cast<bf16x4*>(krp) = hi;
      *reinterpret_cast<bf16x4*>(krp + KP * 2) = lo;
    }
#pragma unroll
    for (int sl = 0; sl < VSLOTS; ++sl) {
      int key, d0;
      if (!v_item(sl, key, d0)) continue;
      const int k32 = key & 31;
      const int half = (k32 >> 2) & 1, blk = k32 >> 3;
      const int pos = (key & ~31) + 16 * (blk >> 1) + 8 * half + 4 * (blk & 1) + (k32 & 3);
      const float vs[4] = {vq[sl].x, vq[sl].y, vq[sl].z, vq[sl].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const split_t vh = (split_t)vs[i];
        char* vr = Vt + (d0 + i) * VROW + pos * 2;
        *reinterpret_cast<split_t*>(vr) = vh;
        *reinterpret_cast<split_t*>(vr + KB * 2) = (split_t)(vs[i] - (float)vh);
      }
    }
  };

#ifdef RGM_ATTN_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_attn_real[2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  ATTN_STAMP(0)
  request(0);
  // ---- Q fragments of this wave's 32 queries (8 waves x 32 = 256 >= T): lane (query l31, half hh) holds Q[q][16j + 8hh .. +7].
  // Requested HERE, before the table loop: that loop's LDS writes wait for its own loads and, loads returning in order, for block 0's
  // as well -- Q requested after it was a second full round trip of every workgroup of the launch at the same moment.
  const int wq = 8 / qsplit;
  const bool active = wave < wq;                               // wave-uniform
  const int q = active ? (qs * wq + wave) * 32 + l31 : T;
  const int qc = min(q, T - 1);
  float4 qraw[KS][2];
  {
    const float* qp = base + (long long)qc * D3;
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int d0 = 16 * j + 8 * hh + 4 * u;
        qraw[j][u] = d0 < HD ? *reinterpret_cast<const float4*>(qp + d0) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  {
    // entry e = (position, chunk) holds the factors of channels 4 chunk .. +3 = table elements 2e, 2e+1: both tables are read as
    // contiguous float4 (two entries each) when they are 16-byte aligned
    const int ne = T * NRC;
    const bool al = (((uintptr_t)cos_tab | (uintptr_t)sin_tab) & 15) == 0;
    for (int e2 = tid; 2 * e2 < ne; e2 += 512) {
      const int e = 2 * e2;
      if (al && e + 1 < ne) {
        const float4 c = reinterpret_cast<const float4*>(cos_tab)[e2], sn = reinterpret_cast<const float4*>(sin_tab)[e2];
        cs_lds[e] = make_float4(c.x, sn.x, c.y, sn.y);
        cs_lds[e + 1] = make_float4(c.z, sn.z, c.w, sn.w);
      } else {
        cs_lds[e] = make_float4(cos_tab[2 * e], sin_tab[2 * e], cos_tab[2 * e + 1], sin_tab[2 * e + 1]);
        if (e + 1 < ne) cs_lds[e + 1] = make_float4(cos_tab[2 * e + 2], sin_tab[2 * e + 2], cos_tab[2 * e + 3], sin_tab[2 * e + 3]);
      }
    }
  }
  const float scale = rsqrtf((float)HD) * 1.44269504088896340736f;   // log2 domain (see above)
  bf16x8 qh[KS], ql[KS];
  {
    __syncthreads();                         // the rotary table is complete
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      float v8[8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int d0 = 16 * j + 8 * hh + 4 * u;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d0 < HD) {
          v = qraw[j][u];
          if (d0 < R) {
            const float4 f = cs_lds[qc * NRC + (d0 >> 2)];
            const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            v.x = x0 * f.x - x1 * f.y;
            v.y = x1 * f.x + x0 * f.y;
            v.z = x2 * f.z - x3 * f.w;
            v.w = x3 * f.z + x2 * f.w;
          }
        }
        v8[4 * u] = v.x * scale; v8[4 * u + 1] = v.y * scale; v8[4 * u + 2] = v.z * scale; v8[4 * u + 3] = v.w * scale;
      }
      split8(v8, qh[j], ql[j]);
    }
  }
  ATTN_STAMP(1)
  deposit(smem3, 0);
  ATTN_STAMP(2)
  if (nb > 1) request(1);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  ATTN_STAMP(3)
  f32x16 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
  int vrow[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) vrow[dt] = min(dt * 32 + l31, HD - 1) * VROW + 16 * hh;
  float m_run = -INFINITY, l_run = 0.f;

  for (int b = 0; b < nb; ++b) {
    const char* Ks = smem3 + (b & 1) * BUF;
    const char* Vt = Ks + KBYTES;
    if (active) {
    // ---- S^T of the block's two key tiles
    f32x16 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.f;
      const char* kp = Ks + (kt * 32 + l31) * KROW + 16 * hh;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(kp + 32 * j);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(kp + 32 * j + KP * 2);
        sacc[kt] = RGM_MFMA_SPLIT_32x32x16(kl, qh[j], sacc[kt], 0, 0, 0);
        sacc[kt] = RGM_MFMA_SPLIT_32x32x16(kh, ql[j], sacc[kt], 0, 0, 0);
        sacc[kt] = RGM_MFMA_SPLIT_32x32x16(kh, qh[j], sacc[kt], 0, 0, 0);
      }
    }
    if (b < 4) { ATTN_STAMP(4 + 3 * b) }
    // ---- running softmax: register e of tile kt is key b*64 + kt*32 + (e&3) + 8*(e>>2) + 4*hh
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int k0 = b * KB + kt * 32;
      if (k0 + 32 > T) {                   // ragged or empty tile (wave-uniform)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (k0 + (e & 3) + 8 * (e >> 2) + 4 * hh >= T) sacc[kt][e] = -INFINITY;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[kt][e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);                     // finite from block 0 on (key 0 is never masked)
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // 0 for the first block (m_run = -inf), 1 when nothing grew
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(sacc[kt][e] - m_new);
        sacc[kt][e] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    if (b > 0) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
    }
    // ---- O^T += V^T . P^T over the block's 64 keys
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        float p8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p8[j] = sacc[kt][8 * h2 + j];
        bf16x8 ph, pl;
        split8(p8, ph, pl);
        const int koff = (kt * 32 + 16 * h2) * 2;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 vh = *reinterpret_cast<const bf16x8*>(Vt + vrow[dt] + koff);
          const bf16x8 vl = *reinterpret_cast<const bf16x8*>(Vt + vrow[dt] + koff + KB * 2);
          oacc[dt] = RGM_MFMA_SPLIT_32x32x16(vl, ph, oacc[dt], 0, 0, 0);
          oacc[dt] = RGM_MFMA_SPLIT_32x32x16(vh, pl, oacc[dt], 0, 0, 0);
          oacc[dt] = RGM_MFMA_SPLIT_32x32x16(vh, ph, oacc[dt], 0, 0, 0);
        }
      }
    }
    }   // active
    __builtin_amdgcn_sched_barrier(0);
    if (b < 4) { ATTN_STAMP(5 + 3 * b) }
    // ---- the next block (requested one iteration ago) goes to the other buffer -- last read in iteration b-1, behind a barrier;
    // the one after it is requested now and flies through the next iteration's MFMAs
    if (b + 1 < nb) {
      deposit(smem3 + ((b + 1) & 1) * BUF, b + 1);
      if (b + 2 < nb) request(b + 2);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
    if (b < 3) { ATTN_STAMP(6 + 3 * b) }
  }

  const float inv = 1.0f / l_run;
  if (lse && hh == 0 && q < T) lse[((long long)n * heads + head) * T + q] = (m_run + log2f(l_run)) * 0.693147180559945309417f;
  if (q < T) {
    float* op = o + ((long long)n * T + q) * D + head * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * hh;
        if (d < HD) {
          const float4 ov = make_float4(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
          if (out_split) {
            bf16x4 hi, lo;
            hi[0] = (split_t)ov.x; hi[1] = (split_t)ov.y; hi[2] = (split_t)ov.z; hi[3] = (split_t)ov.w;
            lo[0] = (split_t)(ov.x - (float)hi[0]); lo[1] = (split_t)(ov.y - (float)hi[1]);
            lo[2] = (split_t)(ov.z - (float)hi[2]); lo[3] = (split_t)(ov.w - (float)hi[3]);
            split_t* rp = reinterpret_cast<split_t*>(o + ((long long)n * T + q) * D);
            store_split4_maybe_pair<32>(rp, head * HD + d, hi, lo);   // lanes l / l + 32 (hh = 0 / 1) hold one 8-aligned group of the same row: 16 bytes each
          } else {
            *reinterpret_cast<float4*>(op + d) = ov;
          }
        }
      }
  }
  ATTN_STAMP(15)
#ifdef RGM_ATTN_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_attn_real[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace rgm
thread_local int g_attn_co_sched = 0;   // dit.hip (the issuing host thread): the launches of a forward run as two half batches are being issued
namespace rgm {
template <int HD>
static int launch_attn_x3_blocked(const float* qkv, float* o, const float* ct, const float* st, int N, int T, int heads, int rot_half,
                                  float* lse, int out_split, hipStream_t s) {
  constexpr int KP = (HD + 15) / 16 * 16;
  // two block buffers + the rotary table; never less than the one-workgroup-per-CU request (hd = 64 with a short rotary table would fit twice)
  const size_t need = 2 * ((size_t)64 * (KP * 4 + 16) + (size_t)HD * (64 * 4 + 16)) + (size_t)T * (rot_half / 2) * 16;
  RGM_REQUIRE(need <= 160 * 1024, "attention: %zu bytes of LDS", need);
  const size_t lds = attn_lds_one_per_cu(need);
  auto kern = rotary_attention_x3_blocked_kernel<HD>;
  static size_t prepared_lds = 0;      // the occupancy check is per LDS size class: the request grows with T
  if (prepared_lds != lds) RGM_TRY(attn_prepare_kernel(kern, 512, lds, "rotary_attention_x3_blocked_kernel"));
  prepared_lds = lds;
  static const int stagger = RGM_EXP_ENV("RGM_ATTN_STAGGER");
  // query split: as many workgroups per (sample, head) -- 4, 2 or 1 -- as keep the launch within one round of the 256 CUs (RGM_ATTN_QSPLIT: A/B runs)
  const char* qs_str = getenv("RGM_ATTN_QSPLIT");            // read per launch: the parity test switches it inside one process
  const int qs_env = qs_str ? atoi(qs_str) : 0;
  // Measured (tools/attn_time.py, identical outputs): B = 1 / 2 / 4 / 8 launches 21.6 / 21.9 / 22.9 / 25.2 us unsplit, 15.6 / 16.8 / 18.6 / 23.7
  // at the best split -- up to 128 workgroups; 256 lose again (B = 4 x 4: 19.4, B = 8 x 4: 36.9).  Beside a second stream's launches (the
  // forward as two half batches: g_attn_co_sched, set by dit.hip) the CUs a small attention launch leaves are what the other half's GEMMs
  // run on: splitting to 128 workgroups there made the B = 4 / 8 forwards 7-8 % slower, so at most 64.
  const int pairs = N * heads, cap = g_attn_co_sched ? 64 : 128;
  int qsplit = qs_env == 1 || qs_env == 2 || qs_env == 4 ? qs_env : (pairs * 4 <= cap ? 4 : pairs * 2 <= cap ? 2 : 1);
  hipLaunchKernelGGL(kern, dim3(pairs * qsplit), dim3(512), lds, s, qkv, o, ct, st, T, heads, rot_half, lse, out_split, stagger, qsplit);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// same contract as rotary_attention_launch (attention.hip)
int rotary_attention_x3_launch(const float* qkv, float* o, const float* cos_tab, const float* sin_tab, int N, int T, int heads,
                               int hd, int rot_half, hipStream_t s, float* lse, int out_split) {
  RGM_REQUIRE(N > 0 && T > 0 && T <= 288, "attention: T=%d out of range (1..288)", T);
  RGM_REQUIRE((2 * rot_half) % 4 == 0 && 2 * rot_half <= hd, "attention: rotary dim %d", 2 * rot_half);
  const int nkt = (T + 31) / 32;
  static const int blocked = getenv("RGM_ATTN_BLOCKED") ? atoi(getenv("RGM_ATTN_BLOCKED")) : 1;
  if (blocked && T > 128 && T <= 256) {     // the key-blocked kernel overlaps its reads with its MFMAs
    if (hd == 72) return launch_attn_x3_blocked<72>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    if (hd == 64) return launch_attn_x3_blocked<64>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
  }
  if (hd == 72) {
    if (nkt <= 4) return launch_attn_x3<72, 4>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    if (nkt <= 8) return launch_attn_x3<72, 8>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    set_error("attention: head_dim 72 supports T <= 256 (K+V of one head must fit the 160 KiB LDS), got %d", T);
    return RGM_ERR_INVALID;
  }
  if (hd == 64) {
    if (nkt <= 4) return launch_attn_x3<64, 4>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    if (nkt <= 5) return launch_attn_x3<64, 5>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    if (nkt <= 8) return launch_attn_x3<64, 8>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    return launch_attn_x3<64, 9>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
  }
  set_error("attention: head_dim %d not supported (64, 72)", hd);
  return RGM_ERR_INVALID;
}

int rotary_attention_fwd(const float* qkv, float* o, const float* cos_tab, const float* sin_tab, int N, int T, int heads, int hd,
                         int rot_half, hipStream_t s, int out_split, float* lse) {
  if (rgm_get_gemm_precision() != 0) return rotary_attention_x3_launch(qkv, o, cos_tab, sin_tab, N, T, heads, hd, rot_half, s, lse, out_split);
  return rotary_attention_launch(qkv, o, cos_tab, sin_tab, N, T, heads, hd, rot_half, s, lse, out_split);
}

int attn_pairs_set(int on) {
  const int prev = g_attn_pairs;
  g_attn_pairs = on;
  return prev;
}

#ifdef RGM_ATTN_STAMPS
int attn_stamps_copy(long long* out) {
  RGM_CHECK_HIP(hipDeviceSynchronize());
  RGM_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_stamps), sizeof(long long) * 128));
  RGM_CHECK_HIP(hipMemcpyFromSymbol(out + 128, HIP_SYMBOL(g_attn_real), sizeof(long long) * 2048));
  return RGM_OK;
}
#endif

}  // namespace rgm
#ifdef RGM_ATTN_STAMPS
extern "C" int rgm_attn_stamps(long long* out128)  /* 128 phase stamps + 2048 entry/exit real-time stamps */ { return rgm::attn_stamps_copy(out128); }
#endif
// Two workgroups per CU for the short-sequence attention at head_dim 72 (T <= 128): 1 = on, 0 = one per CU (the round-3 guard; default).
// Returns the previous setting.
extern "C" int rgm_set_attn_pairs(int on) { return rgm::attn_pairs_set(on != 0); }
