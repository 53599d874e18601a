// attention_x3_body.h -- the single-pass bf16x3 attention of attention_x3.hip (design notes there) as a device function: the one-launch
// kernel wraps it, chain.hip runs it as a work item of the persistent DiT forward.
#pragma once
#include <stdlib.h>
#include "common.h"

namespace rgm {

typedef split_t bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hi[i] = (split_t)v[i];
    lo[i] = (split_t)(v[i] - (float)hi[i]);
  }
}

#ifdef RGM_ATTN_STAMPS   // phase timing experiment (tools/attn_stamps.py; make EXTRA="-DRGM_EXPERIMENTS -DRGM_ATTN_STAMPS")
__device__ long long g_attn_stamps[16 * 8];
__device__ long long g_attn_real[2 * 1024];   // s_memrealtime (100 MHz) at entry / exit of wave 0 of every workgroup
#define ATTN_STAMP(i)                                                                       \
  if (blockIdx.x == 7) {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime();                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    if ((threadIdx.x & 63) == 0) g_attn_stamps[(threadIdx.x >> 6) * 16 + (i)] = now_;       \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  }
#else
#define ATTN_STAMP(i)
#endif

#ifdef RGM_ATTN_HAZARD_DBG   // tools/ubench/attn_hazard.hip: where a workgroup ran (CU, LDS base), when, and what its barrier saw
struct AttnDbg {
  unsigned hw_id, lds_alloc, xcc_id, arrivals;
  unsigned long long t0, t1;
};
__device__ AttnDbg* g_attn_dbg = nullptr;   // [grid]
__device__ int* g_attn_cnt = nullptr;       // [grid], zeroed by the host: staging-complete arrivals counted through global memory
__device__ float* g_attn_dump = nullptr;    // [grid][waves][ATTN_DUMP_ITEMS][64]: Q fragments, max, sum, scores, probabilities of every lane
#define ATTN_DUMP_ITEMS 42
#define ATTN_DUMP(item, val)                                                                                             \
  if (g_attn_dump) g_attn_dump[(((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * ATTN_DUMP_ITEMS + (item)) * 64 + lane] = (val);
#endif

// The single-pass kernel's body for workgroup `bidx` of the (sample, head, query group) grid; smem3 = the workgroup's dynamic LDS.
// COH = 1 (chain.hip, out_split only): the output rows are stored device-coherent, 16 bytes per lane (lanes l / l + 32 pair their halves).
template <int HD, int NKT, int COH = 0>
__device__ __forceinline__ void attn_x3_body(char* smem3, const float* __restrict__ qkv, float* __restrict__ o,
                                             const float* __restrict__ cos_tab, const float* __restrict__ sin_tab, int T, int heads,
                                             int rot_half, float* __restrict__ lse, int out_split, int qgroups, const int bidx, const int tid_in = -1) {
  constexpr int KP = (HD + 15) / 16 * 16;   // padded contraction length of QK^T
  constexpr int KS = KP / 16;               // k16 steps of QK^T
  constexpr int DT = (HD + 31) / 32;        // 32-wide output-channel tiles
  constexpr int TP = NKT * 32;              // padded key count
  constexpr int KROW = KP * 4 + 16;         // bytes per K row  (KP*4/16 is even -> +1 slot makes the stride odd)
  constexpr int VROW = TP * 4 + 16;         // bytes per V^T row (TP*4/16 = 8*NKT is even)
  static_assert((KROW / 16) % 2 == 1 && (VROW / 16) % 2 == 1, "slot strides must be odd");
  char* Ks = smem3;                         // [TP][KROW]
  char* Vt = smem3 + TP * KROW;             // [HD][VROW]

  // qgroups > 1: the query tiles of a (sample, head) are shared out over that many workgroups (each stages K and V itself) -- the
  // classifiers' 257 tokens are 9 tiles on 8 waves, two tile-times in one workgroup, and their (sample, head) grids leave CUs idle
  const int bh = bidx / qgroups, qgrp = bidx - bh * qgroups;
  const int n = bh / heads, head = bh - n * heads;
  const int D = heads * HD, D3 = 3 * D;
  const float* base = qkv + (long long)n * T * D3 + head * HD;
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;
  const int R = 2 * rot_half;
  typedef split_t bf16x4 __attribute__((ext_vector_type(4)));

#ifdef RGM_ATTN_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_attn_real[2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  ATTN_STAMP(0)
#ifdef RGM_ATTN_HAZARD_DBG
  if (g_attn_dbg && tid == 0) {
    AttnDbg& d = g_attn_dbg[blockIdx.x];
    d.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    d.lds_alloc = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    d.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    d.t0 = __builtin_amdgcn_s_memrealtime();
  }
#endif
  // ---- stage K (rotated, split) and V (split, transposed, keys permuted); padded keys / channels are zeros
  // The workgroup is 64 * (query tiles, at most 8) threads: every wave owns a query tile (launch_attn_x3).
  const int nthr = blockDim.x;
  constexpr int CPR = KP / 4;               // float4 chunks per padded K row
  // U chunks per thread and iteration, every load of the U (K, V and the rotary factors) requested before the first is used: with one
  // chunk per iteration a 256-thread workgroup (chain.hip: nobody else on the CU) walked 20 dependent round trips -- 35 of the 52 us of a
  // (sample, head) item (tools/chain_spans.py)
  constexpr int U = 5;
  for (int c0 = tid; c0 < TP * CPR; c0 += U * nthr) {
    float4 kvs[U], vvs[U];
    float2 cfs[U][2];     // (c0, c1), (s0, s1)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * nthr;
      const int key = c / CPR, ch = c - key * CPR, d0 = ch * 4;   // consecutive lanes -> one row's chunks (coalesced global reads)
      kvs[u] = vvs[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      cfs[u][0] = make_float2(1.f, 1.f);
      cfs[u][1] = make_float2(0.f, 0.f);
      if (c < TP * CPR && key < T && d0 < HD) {
        const float* rowp = base + (long long)key * D3;
        kvs[u] = ldg16(rowp + D + d0);
        vvs[u] = ldg16(rowp + 2 * D + d0);
        if (d0 < R) {
          const int pi = key * rot_half + (d0 >> 1);
          cfs[u][0] = make_float2(ldg4(cos_tab + pi), ldg4(cos_tab + pi + 1));
          cfs[u][1] = make_float2(ldg4(sin_tab + pi), ldg4(sin_tab + pi + 1));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * nthr;
      if (c >= TP * CPR) continue;
      const int key = c / CPR, ch = c - key * CPR, d0 = ch * 4;
      float4 kv = kvs[u];
      const float4 vv = vvs[u];
      if (key < T && d0 < HD && d0 < R) {
        const float c0f = cfs[u][0].x, c1 = cfs[u][0].y, s0 = cfs[u][1].x, s1 = cfs[u][1].y;
        const float x0 = kv.x, x1 = kv.y, x2 = kv.z, x3 = kv.w;
        kv.x = x0 * c0f - x1 * s0;
        kv.y = x1 * c0f + x0 * s0;
        kv.z = x2 * c1 - x3 * s1;
        kv.w = x3 * c1 + x2 * s1;
      }
      {
        bf16x4 hi, lo;
        hi[0] = (split_t)kv.x; hi[1] = (split_t)kv.y; hi[2] = (split_t)kv.z; hi[3] = (split_t)kv.w;
        lo[0] = (split_t)(kv.x - (float)hi[0]); lo[1] = (split_t)(kv.y - (float)hi[1]);
        lo[2] = (split_t)(kv.z - (float)hi[2]); lo[3] = (split_t)(kv.w - (float)hi[3]);
        char* kr = Ks + key * KROW + d0 * 2;
        *reinterpret_cast<bf16x4*>(kr) = hi;
        *reinterpret_cast<bf16x4*>(kr + KP * 2) = lo;
      }
      if (d0 < HD) {
        // key -> position inside its 32-group: key = (j&3) + 8*(2*h2 + (j>>2)) + 4*half  <->  pos = 16*h2 + 8*half + j
        const int k32 = key & 31;
        const int half = (k32 >> 2) & 1, blk = k32 >> 3;                 // blk = 2*h2 + (j>>2)
        const int pos = (key & ~31) + 16 * (blk >> 1) + 8 * half + 4 * (blk & 1) + (k32 & 3);
        const float vs[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const split_t hi = (split_t)vs[i];
          char* vr = Vt + (d0 + i) * VROW + pos * 2;
          *reinterpret_cast<split_t*>(vr) = hi;
          *reinterpret_cast<split_t*>(vr + TP * 2) = (split_t)(vs[i] - (float)hi);
        }
      }
    }
  }
  ATTN_STAMP(1)
#ifdef RGM_ATTN_HAZARD_DBG
  if (g_attn_cnt) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((tid & 63) == 0) __hip_atomic_fetch_add(&g_attn_cnt[blockIdx.x], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
  __syncthreads();
#ifdef RGM_ATTN_HAZARD_DBG
  if (g_attn_cnt && g_attn_dbg && tid == 0)
    g_attn_dbg[blockIdx.x].arrivals = __hip_atomic_load(&g_attn_cnt[blockIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
#endif
  ATTN_STAMP(2)

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int nwaves = nthr >> 6;
  // scores are kept in the log2 domain (log2(e) folded into the query scale): p = 2^(s - max) is ONE v_exp_f32 per element
  // instead of the 8-op double-float exp of the fp32 kernel (its ~2e-7 argument error is 100x below the bf16x3 product error)
  const float scale = rsqrtf((float)HD) * 1.44269504088896340736f;
  const int nqt = (T + 31) >> 5;

  const int per_grp = (nqt + qgroups - 1) / qgroups, qt_end = min(nqt, (qgrp + 1) * per_grp);
  for (int qt = qgrp * per_grp + wave; qt < qt_end; qt += nwaves) {
    const int q = qt * 32 + l31;
    const int qc = min(q, T - 1);
    // ---- Q fragments: lane (query l31, half hh) holds Q[q][16j + 8hh .. +7], rotated, pre-scaled, split
    bf16x8 qh[KS], ql[KS];
    // Two-phase Q prologue (round 5; the round-4 probe flavour, now the product): every load of the prologue is issued and RETIRED before
    // the first value is used.  The short-sequence hazard of DESIGN 4h was a read of a global_load_dwordx2's second destination register
    // two instructions behind its s_waitcnt; with one wait in front of ALL uses that window does not exist (0 wrong rows in 100 launches at
    // two workgroups per CU, profiles/r04_attn_hazard_*).  The one-workgroup-per-CU guard stays.  Same arithmetic, same values.
    // -DRGM_ATTN_ONE_PHASE_Q restores the interleaved prologue (tools/ubench/attn_hazard.hip reproduces the hazard with it).
#ifndef RGM_ATTN_ONE_PHASE_Q
    if (true) {
      const float* qp = base + (long long)qc * D3;
      float4 qraw[KS][2];
      float2 cf[KS][2][2];
#pragma unroll
      for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int d0 = 16 * j + 8 * hh + 4 * u;
          qraw[j][u] = make_float4(0.f, 0.f, 0.f, 0.f);
          cf[j][u][0] = cf[j][u][1] = make_float2(1.f, 1.f);
          if (d0 < HD) {
            qraw[j][u] = ldg16(qp + d0);
            if (d0 < R) {
              const int pi = qc * rot_half + (d0 >> 1);
              cf[j][u][0] = ldg8(cos_tab + pi);
              cf[j][u][1] = ldg8(sin_tab + pi);
            }
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int d0 = 16 * j + 8 * hh + 4 * u;
          float4 v = qraw[j][u];
          if (d0 < R) {
            const float c0 = cf[j][u][0].x, c1 = cf[j][u][0].y, s0 = cf[j][u][1].x, s1 = cf[j][u][1].y;
            const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            v.x = x0 * c0 - x1 * s0;
            v.y = x1 * c0 + x0 * s0;
            v.z = x2 * c1 - x3 * s1;
            v.w = x3 * c1 + x2 * s1;
          }
          v8[4 * u] = v.x * scale; v8[4 * u + 1] = v.y * scale; v8[4 * u + 2] = v.z * scale; v8[4 * u + 3] = v.w * scale;
        }
        split8(v8, qh[j], ql[j]);
      }
    } else
#endif
    {
      const float* qp = base + (long long)qc * D3;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int d0 = 16 * j + 8 * hh + 4 * u;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (d0 < HD) {
            v = *reinterpret_cast<const float4*>(qp + d0);
            if (d0 < R) {
              const int pi = qc * rot_half + (d0 >> 1);
              const float c0 = cos_tab[pi], s0 = sin_tab[pi], c1 = cos_tab[pi + 1], s1 = sin_tab[pi + 1];
              const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
              v.x = x0 * c0 - x1 * s0;
              v.y = x1 * c0 + x0 * s0;
              v.z = x2 * c1 - x3 * s1;
              v.w = x3 * c1 + x2 * s1;
            }
          }
          v8[4 * u] = v.x * scale; v8[4 * u + 1] = v.y * scale; v8[4 * u + 2] = v.z * scale; v8[4 * u + 3] = v.w * scale;
        }
        split8(v8, qh[j], ql[j]);
      }
    }
    ATTN_STAMP(3)
    // ---- S^T[key][query] = K . Q^T
    f32x16 sacc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
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
    ATTN_STAMP(4)
    // ---- softmax over keys: register e of tile kt is key kt*32 + (e&3) + 8*(e>>2) + 4*hh
    float mx = -INFINITY;
    const int ktr = T >> 5, tr = T & 31;   // ragged tile index / valid keys in it (wave-uniform)
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt * 32 >= T) {                  // tile entirely past the sequence
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[kt][e] = -INFINITY;
      } else if (kt == ktr) {              // the one ragged tile: 16 lane masks shared by all kt
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if ((e & 3) + 8 * (e >> 2) + 4 * hh >= tr) sacc[kt][e] = -INFINITY;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[kt][e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(sacc[kt][e] - mx);   // masked scores: 2^(-inf) = 0
        sacc[kt][e] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    // natural-log sum-exp of the scaled scores, saved for the backward (scores here are in the log2 domain)
    if (lse && hh == 0 && q < T) lse[((long long)n * heads + head) * T + q] = (mx + log2f(sum)) * 0.693147180559945309417f;
    ATTN_STAMP(5)
#ifdef RGM_ATTN_HAZARD_DUMP
    const float dbg_mx = mx, dbg_sum = sum;
#endif
    // ---- O^T[d][query] = V^T . P^T ; A operand = V^T rows (d = lane&31), B operand = the probability registers, split
    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
    int vrow[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vrow[dt] = min(dt * 32 + l31, HD - 1) * VROW + 16 * hh;   // rows >= hd: discarded outputs
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
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
          const bf16x8 vl = *reinterpret_cast<const bf16x8*>(Vt + vrow[dt] + koff + TP * 2);
          oacc[dt] = RGM_MFMA_SPLIT_32x32x16(vl, ph, oacc[dt], 0, 0, 0);
          oacc[dt] = RGM_MFMA_SPLIT_32x32x16(vh, pl, oacc[dt], 0, 0, 0);
          oacc[dt] = RGM_MFMA_SPLIT_32x32x16(vh, ph, oacc[dt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the V^T reads of later key tiles from being hoisted (spills)
    }
    ATTN_STAMP(6)
    // ---- store: lane = query (row), registers 4g..4g+3 = 4 consecutive channels
    if (q < T) {
      float* op = o + ((long long)n * T + q) * D + head * HD;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * hh;
          if (d < HD) {
            const float4 ov = make_float4(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
            if (out_split) {   // split-row format (common.h split_idx): A operand of the pre-split proj GEMM
              bf16x4 hi, lo;
              hi[0] = (split_t)ov.x; hi[1] = (split_t)ov.y; hi[2] = (split_t)ov.z; hi[3] = (split_t)ov.w;
              lo[0] = (split_t)(ov.x - (float)hi[0]); lo[1] = (split_t)(ov.y - (float)hi[1]);
              lo[2] = (split_t)(ov.z - (float)hi[2]); lo[3] = (split_t)(ov.w - (float)hi[3]);
              split_t* rp = reinterpret_cast<split_t*>(o + ((long long)n * T + q) * D);
              if constexpr (COH) {   // lanes l (hh = 0: channels 8g .. +3) and l + 32 (hh = 1: +4 .. +7) hold one 8-aligned group (HD % 8 == 0)
                store_split4_pair_sc1<32>(rp, head * HD + d, hi, lo);
              } else {
                store_split4_maybe_pair<32>(rp, head * HD + d, hi, lo);   // the same pairing, ordinary stores
              }
            } else {
              *reinterpret_cast<float4*>(op + d) = ov;
            }
          }
        }
    }
#ifdef RGM_ATTN_HAZARD_DUMP   // the Q fragments as the S^T MFMAs saw them (stored here, behind the output: stores in the prologue hide the failure;
                              // the 40 extra live registers shift the schedule too -- the run that named the element was made at commit 7be1b0c)
    if constexpr (KS <= 5) {
#pragma unroll
      for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          ATTN_DUMP(j * 8 + w, __uint_as_float(reinterpret_cast<const unsigned*>(&qh[j])[w]))
          ATTN_DUMP(j * 8 + 4 + w, __uint_as_float(reinterpret_cast<const unsigned*>(&ql[j])[w]))
        }
    }
    ATTN_DUMP(40, dbg_mx)
    ATTN_DUMP(41, dbg_sum)
#endif
    ATTN_STAMP(7)
  }
#ifdef RGM_ATTN_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_attn_real[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#ifdef RGM_ATTN_HAZARD_DBG
  if (g_attn_dbg && tid == 0) g_attn_dbg[blockIdx.x].t1 = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace rgm
