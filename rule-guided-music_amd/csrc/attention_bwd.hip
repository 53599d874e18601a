// attention_bwd.hip -- input-gradient of the RotaryAttention core (classifier guidance backward, a7).
//
// Reference: the autograd of guided_diffusion/dit.py:263-277 as used by
// guided_diffusion/condition_functions.py:58-85 (th.autograd.grad of the classifier log-prob w.r.t. x_t).
// Weights are frozen at sampling time, so only d(qkv) is needed.  Given dO and the forward's saved qkv, O and
// per-query log-sum-exp:
//     P = exp(S - lse),  D = rowsum(dO * O),  dS = P * (dO V^T - D),
//     dQ_rot = scale * dS K_rot,  dK_rot = dS^T (scale * Q_rot),  dV = P^T dO,  then un-rotate dQ, dK.
//
// Two kernels per (sample, head), both single-pass over 32-wide tiles with fp32 MFMA and the same
// "transposed scores" register trick as the forward (attention.hip):
//   * dq kernel : K, V resident in LDS; a wave owns 32 queries (columns of S^T), streams key tiles;
//                 dP^T = V dO^T lands in the same C layout as P^T, dS^T registers feed dQ^T = K^T dS^T directly;
//   * dkv kernel: Q (pre-scaled), dO, lse, D resident in LDS; a wave owns 32 keys (columns of S), streams query
//                 tiles; P and dS registers feed dV^T = dO^T P and dK^T = Q^T dS directly.
// No atomics: every output element is owned by exactly one wave, so the result is deterministic.
#include <stdlib.h>
#include "common.h"

namespace rgm {

__device__ __forceinline__ float exp_le0(float x) {
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f;
  x = fmaxf(x, -104.0f);
  const float t = x * L2E_HI;
  float r = fmaf(x, L2E_HI, -t);
  r = fmaf(x, L2E_LO, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.693147182464599609375f, e);
}

// ---- bf16x3 arithmetic of the backward (round 6; the bf16x3 / bf16x3_presplit modes): the same five contractions on v_mfma_f32_32x32x16_bf16,
// every operand split hi + lo when it is fetched (a*b ~= al*bh + ah*bl + ah*bh, fp32 accumulate -- the forward's and the GEMMs' arithmetic),
// with the register layouts of the fp32 kernels: 16 channels (or 16 keys / queries) per MFMA instead of 2 -- 36 (dq) / 48 (dkv) MFMAs of 32
// cycles per tile pair where the fp32 path issues 96 / 128 of 64.  The LDS images are pre-split rows (PsImg below); only the registers that
// become B operands (Q / dO resp. K / V fragments once per tile, P and dS per tile pair) are split by the wave that holds them.
typedef split_t bsplit8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void bwd_split8(const float* v, bsplit8& hi, bsplit8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hi[i] = (split_t)v[i];
    lo[i] = (split_t)(v[i] - (float)hi[i]);
  }
}
#ifdef RGM_SPLIT_F16
#define RGM_BWD_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
#define RGM_BWD_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
// acc += A . B with both operands split: term order al*bh, ah*bl, ah*bh (as everywhere)
__device__ __forceinline__ void mfma_x3(f32x16& acc, const bsplit8& ah, const bsplit8& al, const bsplit8& bh, const bsplit8& bl) {
  acc = RGM_BWD_MFMA(al, bh, acc, 0, 0, 0);
  acc = RGM_BWD_MFMA(ah, bl, acc, 0, 0, 0);
  acc = RGM_BWD_MFMA(ah, bh, acc, 0, 0, 0);
}
// ---- LDS images of the x3 kernels: a row keeps its (HD + 4) * 4 bytes but holds [HD hi halves | HD lo halves | 16 bytes of pad] -- split ONCE
// by the thread that stages the element (the on-the-fly split of round 6's first version was repeated by each of the eight waves for the tile
// it read: the kernels were VALU-bound).  Row stride in 16-byte slots: 17 (hd 64) / 19 (hd 72), odd -> conflict-free ds_read_b128 groups.
template <int HD>
struct PsImg {
  static constexpr int RS = (HD + 4) * 4;                    // bytes per row (the fp32 image's HDP floats)
  static __device__ __forceinline__ void st4(float* img, int row, int d0, const float4& v) {
    typedef split_t h4 __attribute__((ext_vector_type(4)));
    h4 hi, lo;
    hi[0] = (split_t)v.x; hi[1] = (split_t)v.y; hi[2] = (split_t)v.z; hi[3] = (split_t)v.w;
    lo[0] = (split_t)(v.x - (float)hi[0]); lo[1] = (split_t)(v.y - (float)hi[1]);
    lo[2] = (split_t)(v.z - (float)hi[2]); lo[3] = (split_t)(v.w - (float)hi[3]);
    char* rp = reinterpret_cast<char*>(img) + row * RS + 2 * d0;
    *reinterpret_cast<h4*>(rp) = hi;
    *reinterpret_cast<h4*>(rp + 2 * HD) = lo;
  }
  // channels d0 .. d0 + 7 of `row` (d0 a multiple of 8) as the split fragment; zero = the chunk lies beyond the row's channels
  static __device__ __forceinline__ void row8(const float* img, int row, int d0, bool zero, bsplit8& hi, bsplit8& lo) {
    const char* rp = reinterpret_cast<const char*>(img) + row * RS + 2 * d0;
    hi = *reinterpret_cast<const bsplit8*>(rp);
    lo = *reinterpret_cast<const bsplit8*>(rp + 2 * HD);
    if (zero) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { hi[i] = (split_t)0.f; lo[i] = (split_t)0.f; }
    }
  }
  // the transposed fragment: channel `col` of the 8 rows a lane's C/D registers 8 h2 .. 8 h2 + 7 stand for -- row (j & 3) + 8 (2 h2 + (j >> 2))
  // + 4 hh of the 32-row tile starting at row `r0`: slot j of the B operand built from those registers meets slot j here
  static __device__ __forceinline__ void col8(const float* img, int r0, int col, int h2, int hh, bsplit8& hi, bsplit8& lo) {
    const char* cp = reinterpret_cast<const char*>(img) + (long long)r0 * RS + 2 * col;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const char* ep = cp + ((j & 3) + 8 * (2 * h2 + (j >> 2)) + 4 * hh) * RS;
      hi[j] = *reinterpret_cast<const split_t*>(ep);
      lo[j] = *reinterpret_cast<const split_t*>(ep + 2 * HD);
    }
  }
  // the element / four elements as floats again (hi + lo: 2^-17 of the fp32 value; the lone-token paths' plain sums)
  static __device__ __forceinline__ float ld1(const float* img, int row, int d) {
    const char* ep = reinterpret_cast<const char*>(img) + row * RS + 2 * d;
    return (float)*reinterpret_cast<const split_t*>(ep) + (float)*reinterpret_cast<const split_t*>(ep + 2 * HD);
  }
  static __device__ __forceinline__ float4 ld4(const float* img, int row, int d0) {
    typedef split_t h4 __attribute__((ext_vector_type(4)));
    const char* rp = reinterpret_cast<const char*>(img) + row * RS + 2 * d0;
    const h4 hi = *reinterpret_cast<const h4*>(rp), lo = *reinterpret_cast<const h4*>(rp + 2 * HD);
    return make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]);
  }
};
// one accessor for both image formats (X3: pre-split rows; else fp32 rows of HDP floats)
template <int HD, bool X3>
__device__ __forceinline__ float4 img_ld4(const float* img, int row, int d0) {
  if constexpr (X3) return PsImg<HD>::ld4(img, row, d0);
  else return *reinterpret_cast<const float4*>(img + row * (HD + 4) + d0);
}
template <int HD, bool X3>
__device__ __forceinline__ float img_ld1(const float* img, int row, int d) {
  if constexpr (X3) return PsImg<HD>::ld1(img, row, d);
  else return img[row * (HD + 4) + d];
}

__device__ __forceinline__ float4 rotate4(float4 v, const float* __restrict__ ct, const float* __restrict__ st, int pi, bool inverse) {
  const float c0 = ct[pi], c1 = ct[pi + 1];
  float s0 = st[pi], s1 = st[pi + 1];
  if (inverse) { s0 = -s0; s1 = -s1; }
  return make_float4(v.x * c0 - v.y * s0, v.y * c0 + v.x * s0, v.z * c1 - v.w * s1, v.w * c1 + v.z * s1);
}

// ------------------------------------------------------------------------------------------- dQ
// SPLIT: one workgroup per (sample, head, QUERY TILE) instead of per (sample, head): its eight waves share the tile's key loop (wave w takes
// key tiles w, w + 8, ...), the partial dQ tiles meet in LDS (over the K / V images, behind a barrier) and are summed in wave order.  For
// the grids that leave the chip empty -- the classifiers at the samplers' batches: 6 heads x B samples of 257 tokens = 9 query tiles on 8
// waves, i.e. two tile-times on 24 .. 192 of 256 CUs -- this turns 2 x 9 serial key tiles into 2 and fills the CUs (launch_bwd).
// d(qkv) rows go out as fp32 or -- for a pre-split dgrad GEMM right behind (dit.hip grad chain) -- as split rows (common.h split_idx);
// col = column of the value's first channel inside the 3 D wide row, a multiple of 4 (the four channels share a 32-block)
__device__ __forceinline__ void dqkv_store4(float* __restrict__ dqkv, long long row, int D3, int col, const float4& v, int osplit) {
  if (osplit) {
    typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, lo;
    hi[0] = (split_t)v.x; hi[1] = (split_t)v.y; hi[2] = (split_t)v.z; hi[3] = (split_t)v.w;
    lo[0] = (split_t)(v.x - (float)hi[0]); lo[1] = (split_t)(v.y - (float)hi[1]);
    lo[2] = (split_t)(v.z - (float)hi[2]); lo[3] = (split_t)(v.w - (float)hi[3]);
    split_t* rp = reinterpret_cast<split_t*>(dqkv + row * D3);
    *reinterpret_cast<bf16x4*>(rp + split_idx(col)) = hi;
    *reinterpret_cast<bf16x4*>(rp + split_idx(col) + 32) = lo;
  } else {
    *reinterpret_cast<float4*>(dqkv + row * D3 + col) = v;
  }
}
__device__ __forceinline__ void dqkv_store1(float* __restrict__ dqkv, long long row, int D3, int col, float v, int osplit) {
  if (osplit) {
    split_t* rp = reinterpret_cast<split_t*>(dqkv + row * D3);
    const split_t hi = (split_t)v;
    rp[split_idx(col)] = hi;
    rp[split_idx(col) + 32] = (split_t)(v - (float)hi);
  } else {
    dqkv[row * D3 + col] = v;
  }
}

template <int HD, int NKT, bool SPLIT, bool X3 = false>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                          const float* __restrict__ d_o, const float* __restrict__ lse,
                                                          float* __restrict__ dqkv, const float* __restrict__ cos_tab,
                                                          const float* __restrict__ sin_tab, int T, int heads, int rot_half, int lone, int osplit) {
  constexpr int HDP = HD + 4, KB = HD / 8, DT = (HD + 31) / 32, TP = NKT * 32;   // hd = 72: the third channel tile is partial
  // (its operand reads run past a row into the next row / the following array: finite data feeding accumulator rows that are never stored)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;              // [TP][HDP] rotated keys
  float* Vs = smem + TP * HDP;   // [TP][HDP]
  const int nqt_all = (T + 31) >> 5;
  const int bh = SPLIT ? blockIdx.x / nqt_all : blockIdx.x;
  const int n = bh / heads, head = bh - n * heads;
  const int D = heads * HD, D3 = 3 * D, R = 2 * rot_half;
  const float* base = qkv + (long long)n * T * D3 + head * HD;
  const int tid = threadIdx.x;
  constexpr int CPR = HD / 4;
  // every load of the staging -- the rows' chunks and their rotary factors -- is requested before the first is used: chunk by chunk the
  // loop walked TP * CPR / 512 = 9 dependent round trips (rows, then the cos / sin entries) before the first MFMA of the launch
  constexpr int NIT = (TP * CPR + 511) / 512;
  {
    float4 kvs[NIT], vvs[NIT];
    float cf[NIT][4];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c = tid + i * 512;
      const int key = c / CPR, d0 = (c - key * CPR) * 4;
      kvs[i] = vvs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      cf[i][0] = cf[i][1] = 1.f;
      cf[i][2] = cf[i][3] = 0.f;
      if (c < TP * CPR && key < T) {
        const float* rowp = base + (long long)key * D3;
        kvs[i] = *reinterpret_cast<const float4*>(rowp + D + d0);
        vvs[i] = *reinterpret_cast<const float4*>(rowp + 2 * D + d0);
        if (d0 < R) {
          const int pi = key * rot_half + (d0 >> 1);
          cf[i][0] = cos_tab[pi]; cf[i][1] = cos_tab[pi + 1];
          cf[i][2] = sin_tab[pi]; cf[i][3] = sin_tab[pi + 1];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c = tid + i * 512;
      if (c >= TP * CPR) continue;
      const int key = c / CPR, d0 = (c - key * CPR) * 4;
      const float4 x = kvs[i];
      const float4 kv = make_float4(x.x * cf[i][0] - x.y * cf[i][2], x.y * cf[i][0] + x.x * cf[i][2],
                                    x.z * cf[i][1] - x.w * cf[i][3], x.w * cf[i][1] + x.z * cf[i][3]);   // (1, 0) outside the rotary channels: x itself
      if constexpr (X3) {
        PsImg<HD>::st4(Ks, key, d0, kv);
        PsImg<HD>::st4(Vs, key, d0, vvs[i]);
      } else {
        *reinterpret_cast<float4*>(Ks + key * HDP + d0) = kv;
        *reinterpret_cast<float4*>(Vs + key * HDP + d0) = vvs[i];
      }
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const float scale = rsqrtf((float)HD);
  const int ktr = T >> 5, tr = T & 31;
  // lone (T % 32 == 1, the classifiers' 257 tokens): the last query tile holds ONE query -- as a ninth MFMA tile it is a second round for
  // wave 0, i.e. the whole kernel's second tile-time.  One wave computes that query's row with plain FMAs instead (below).
  const int nqt = ((T + 31) >> 5) - (lone ? 1 : 0);
  const int qt0 = SPLIT ? blockIdx.x - bh * nqt_all : wave;
  for (int qt = qt0; qt < nqt; qt += SPLIT ? nqt_all : 8) {
    const int q = qt * 32 + l31, qc = min(q, T - 1);
    const long long orow = ((long long)n * T + qc) * D + head * HD;
    f32x4 qf[X3 ? 1 : KB], dof[X3 ? 1 : KB];
    constexpr int KS = (HD + 15) / 16;                       // k16 steps of the contractions over the channels (hd = 72: the fifth is half empty)
    bsplit8 qh[X3 ? KS : 1], ql[X3 ? KS : 1], gh[X3 ? KS : 1], gl[X3 ? KS : 1];
    float dsum = 0.f;
    if constexpr (X3) {   // lane (query l31, half hh) holds channels 16 j + 8 hh .. + 7 of Q (rotated, scaled) and dO: B operands of S^T and dP^T
      const float* qp = base + (long long)qc * D3;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        float q8[8], g8[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int d0 = 16 * j + 8 * hh + 4 * u;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f), g = v;
          if (d0 < HD) {
            v = *reinterpret_cast<const float4*>(qp + d0);
            if (d0 < R) v = rotate4(v, cos_tab, sin_tab, qc * rot_half + (d0 >> 1), false);
            g = *reinterpret_cast<const float4*>(d_o + orow + d0);
            const float4 ov = *reinterpret_cast<const float4*>(o + orow + d0);
            dsum += (g.x * ov.x + g.y * ov.y) + (g.z * ov.z + g.w * ov.w);
          }
          q8[4 * u] = v.x * scale; q8[4 * u + 1] = v.y * scale; q8[4 * u + 2] = v.z * scale; q8[4 * u + 3] = v.w * scale;
          g8[4 * u] = g.x; g8[4 * u + 1] = g.y; g8[4 * u + 2] = g.z; g8[4 * u + 3] = g.w;
        }
        bwd_split8(q8, qh[j], ql[j]);
        bwd_split8(g8, gh[j], gl[j]);
      }
    } else {
      const float* qp = base + (long long)qc * D3;
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int d0 = 8 * j + 4 * hh;
        float4 v = *reinterpret_cast<const float4*>(qp + d0);
        if (d0 < R) v = rotate4(v, cos_tab, sin_tab, qc * rot_half + (d0 >> 1), false);
        qf[j][0] = v.x * scale; qf[j][1] = v.y * scale; qf[j][2] = v.z * scale; qf[j][3] = v.w * scale;
        const float4 g = *reinterpret_cast<const float4*>(d_o + orow + d0);
        const float4 ov = *reinterpret_cast<const float4*>(o + orow + d0);
        dof[j][0] = g.x; dof[j][1] = g.y; dof[j][2] = g.z; dof[j][3] = g.w;
        dsum += (g.x * ov.x + g.y * ov.y) + (g.z * ov.z + g.w * ov.w);
      }
    }
    dsum += __shfl_xor(dsum, 32, 64);                     // D[q] = sum_d dO[q][d] O[q][d]
    const float lq = lse[((long long)n * heads + head) * T + qc];
    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
#pragma unroll 1
    for (int kt = SPLIT ? wave : 0; kt < NKT; kt += SPLIT ? 8 : 1) {
      if (kt * 32 >= T) break;
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
      if constexpr (X3) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          const bool past = 16 * j + 8 * hh >= HD;            // (hd = 72: the read runs into the lo halves / the pad)
          bsplit8 kh, kl, vh, vl;
          PsImg<HD>::row8(Ks, kt * 32 + l31, 16 * j + 8 * hh, past, kh, kl);
          PsImg<HD>::row8(Vs, kt * 32 + l31, 16 * j + 8 * hh, past, vh, vl);
          mfma_x3(s, kh, kl, qh[j], ql[j]);                   // S^T[key][query]
          mfma_x3(dp, vh, vl, gh[j], gl[j]);                  // dP^T[key][query]
        }
      } else {
      const float* kp = Ks + (kt * 32 + l31) * HDP + 4 * hh;
      const float* vp = Vs + (kt * 32 + l31) * HDP + 4 * hh;
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 8 * j);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(vp + 8 * j);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[u], qf[j][u], s, 0, 0, 0);      // S^T[key][query]
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[u], dof[j][u], dp, 0, 0, 0);   // dP^T[key][query]
        }
      }
      }
      f32x16 ds;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float p = exp_le0(s[e] - lq);
        if (kt == ktr && (e & 3) + 8 * (e >> 2) + 4 * hh >= tr) p = 0.f;               // ragged last key tile
        ds[e] = p * (dp[e] - dsum);
      }
      if constexpr (X3) {   // dQ^T[d][query] += K^T[d][key] dS^T[key][query]: the registers of dS^T are the B operand, K^T comes column-wise from the image
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          float d8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) d8[j] = ds[8 * h2 + j];
          bsplit8 dsh, dsl;
          bwd_split8(d8, dsh, dsl);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            bsplit8 ah, al;
            PsImg<HD>::col8(Ks, kt * 32, min(dt * 32 + l31, HD - 1), h2, hh, ah, al);   // (channels past hd: rows of dQ^T nobody stores)
            mfma_x3(dq[dt], ah, al, dsh, dsl);
          }
        }
      } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float* kr = Ks + (kt * 32 + (u & 3) + 8 * (u >> 2) + 4 * hh) * HDP + l31;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[dt * 32], ds[u], dq[dt], 0, 0, 0);
        if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      }
    }
    if constexpr (SPLIT) {   // partial dQ of this wave's key tiles -> LDS [wave][query][HDP] over the K image; fixed-order sum; scale, un-rotate, store
      __syncthreads();       // every wave is done with K / V
      float* part = smem + (size_t)wave * 32 * HDP;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * hh;
          if (d < HD) *reinterpret_cast<float4*>(part + l31 * HDP + d) = make_float4(dq[dt][4 * g], dq[dt][4 * g + 1], dq[dt][4 * g + 2], dq[dt][4 * g + 3]);
        }
      __syncthreads();
      for (int e = tid; e < 32 * (HD / 4); e += 512) {
        const int qi = e / (HD / 4), d = (e - qi * (HD / 4)) * 4, qq = qt * 32 + qi;
        if (qq >= T) continue;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const float4 b = *reinterpret_cast<const float4*>(smem + (size_t)w * 32 * HDP + qi * HDP + d);
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float4 v = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
        if (d < R) v = rotate4(v, cos_tab, sin_tab, qq * rot_half + (d >> 1), true);
        dqkv_store4(dqkv, (long long)n * T + qq, D3, head * HD + d, v, osplit);
      }
    } else
    if (q < T) {   // dQ^T[d][query]: lane = query row, registers 4g..4g+3 = channels dt*32 + 8g + 4hh ..+3
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * hh;
          if (d >= HD) continue;
          float4 v = make_float4(dq[dt][4 * g] * scale, dq[dt][4 * g + 1] * scale, dq[dt][4 * g + 2] * scale, dq[dt][4 * g + 3] * scale);
          if (d < R) v = rotate4(v, cos_tab, sin_tab, q * rot_half + (d >> 1), true);
          dqkv_store4(dqkv, (long long)n * T + q, D3, head * HD + d, v, osplit);
        }
    }
  }
  if constexpr (!SPLIT) {
    if (lone && wave == 7) {
      // dQ of the lone last query q* = T - 1 without MFMA: phase A, lane = key: s = q* . K[key], dp = dO* . V[key], ds = p (dp - D*);
      // phase B, lane = channel: dq*[d] = sum_key ds[key] K[key][d].  Scratch rows sit behind the K / V images (launch_bwd asks for them).
      float* scr = smem + 2 * TP * HDP;          // qrow[HDP] | grow[HDP] | ds[TP] | out[HDP]
      float* qrow = scr;
      float* grow = scr + HDP;
      float* dsr = scr + 2 * HDP;
      float* outr = dsr + TP;
      const int qs_ = T - 1;
      const long long orow = ((long long)n * T + qs_) * D + head * HD;
      float dpart = 0.f;
      for (int d0 = lane * 4; d0 < HD; d0 += 256) {
        float4 v = *reinterpret_cast<const float4*>(base + (long long)qs_ * D3 + d0);
        if (d0 < R) v = rotate4(v, cos_tab, sin_tab, qs_ * rot_half + (d0 >> 1), false);
        *reinterpret_cast<float4*>(qrow + d0) = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
        const float4 g = *reinterpret_cast<const float4*>(d_o + orow + d0);
        const float4 ov = *reinterpret_cast<const float4*>(o + orow + d0);
        *reinterpret_cast<float4*>(grow + d0) = g;
        dpart += (g.x * ov.x + g.y * ov.y) + (g.z * ov.z + g.w * ov.w);
      }
      const float dstar = wave_sum(dpart);
      const float lq = lse[((long long)n * heads + head) * T + qs_];
      for (int key = lane; key < TP; key += 64) {
        float sacc = 0.f, dpa = 0.f;
        if (key < T) {
#pragma unroll 4
          for (int d0 = 0; d0 < HD; d0 += 4) {
            const float4 kq = img_ld4<HD, X3>(Ks, key, d0), vq = img_ld4<HD, X3>(Vs, key, d0);
            const float4 qq = *reinterpret_cast<const float4*>(qrow + d0), gq = *reinterpret_cast<const float4*>(grow + d0);
            sacc += (kq.x * qq.x + kq.y * qq.y) + (kq.z * qq.z + kq.w * qq.w);
            dpa += (vq.x * gq.x + vq.y * gq.y) + (vq.z * gq.z + vq.w * gq.w);
          }
        }
        dsr[key] = key < T ? exp_le0(sacc - lq) * (dpa - dstar) : 0.f;
      }
      for (int d = lane; d < HD; d += 64) {        // (same wave wrote dsr: LDS operations of a wave are in order)
        float a = 0.f;
        for (int key = 0; key < T; ++key) a = fmaf(dsr[key], img_ld1<HD, X3>(Ks, key, d), a);
        outr[d] = a * scale;
      }
      for (int d0 = lane * 4; d0 < HD; d0 += 256) {
        float4 v = *reinterpret_cast<const float4*>(outr + d0);
        if (d0 < R) v = rotate4(v, cos_tab, sin_tab, qs_ * rot_half + (d0 >> 1), true);
        dqkv_store4(dqkv, (long long)n * T + qs_, D3, head * HD + d0, v, osplit);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- dK, dV
// SPLIT: one workgroup per (sample, head, KEY TILE); wave w takes query tiles w, w + 8, ...; partial dK / dV tiles summed through LDS
template <int HD, int NKT, bool SPLIT, bool X3 = false>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                           const float* __restrict__ d_o, const float* __restrict__ lse,
                                                           float* __restrict__ dqkv, const float* __restrict__ cos_tab,
                                                           const float* __restrict__ sin_tab, int T, int heads, int rot_half, int lone, int osplit) {
  constexpr int HDP = HD + 4, KB = HD / 8, DT = (HD + 31) / 32, TP = NKT * 32;   // hd = 72: the third channel tile is partial
  // (its operand reads run past a row into the next row / the following array: finite data feeding accumulator rows that are never stored)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                   // [TP][HDP] rotated, pre-scaled queries
  float* Gs = smem + TP * HDP;        // [TP][HDP] dO
  float* Ls = smem + 2 * TP * HDP;    // [TP] lse
  float* Ds = Ls + TP;                // [TP] D = rowsum(dO * O)
  const int nkt_all = (T + 31) >> 5;
  const int bh = SPLIT ? blockIdx.x / nkt_all : blockIdx.x;
  const int n = bh / heads, head = bh - n * heads;
  const int D = heads * HD, D3 = 3 * D, R = 2 * rot_half;
  const float* base = qkv + (long long)n * T * D3 + head * HD;
  const int tid = threadIdx.x;
  const float scale = rsqrtf((float)HD);
  constexpr int CPR = HD / 4;
  constexpr int NIT = (TP * CPR + 511) / 512;      // as in the dq kernel: every load of the staging requested before the first is used
  {
    float4 qvs[NIT], gvs[NIT];
    float cf[NIT][4];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c = tid + i * 512;
      const int qi = c / CPR, d0 = (c - qi * CPR) * 4;
      qvs[i] = gvs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      cf[i][0] = cf[i][1] = 1.f;
      cf[i][2] = cf[i][3] = 0.f;
      if (c < TP * CPR && qi < T) {
        qvs[i] = *reinterpret_cast<const float4*>(base + (long long)qi * D3 + d0);
        gvs[i] = *reinterpret_cast<const float4*>(d_o + ((long long)n * T + qi) * D + head * HD + d0);
        if (d0 < R) {
          const int pi = qi * rot_half + (d0 >> 1);
          cf[i][0] = cos_tab[pi]; cf[i][1] = cos_tab[pi + 1];
          cf[i][2] = sin_tab[pi]; cf[i][3] = sin_tab[pi + 1];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c = tid + i * 512;
      if (c >= TP * CPR) continue;
      const int qi = c / CPR, d0 = (c - qi * CPR) * 4;
      const float4 x = qvs[i];
      float4 qv = make_float4(x.x * cf[i][0] - x.y * cf[i][2], x.y * cf[i][0] + x.x * cf[i][2],
                              x.z * cf[i][1] - x.w * cf[i][3], x.w * cf[i][1] + x.z * cf[i][3]);
      qv = make_float4(qv.x * scale, qv.y * scale, qv.z * scale, qv.w * scale);
      if constexpr (X3) {
        PsImg<HD>::st4(Qs, qi, d0, qv);
        PsImg<HD>::st4(Gs, qi, d0, gvs[i]);
      } else {
        *reinterpret_cast<float4*>(Qs + qi * HDP + d0) = qv;
        *reinterpret_cast<float4*>(Gs + qi * HDP + d0) = gvs[i];
      }
    }
  }
  for (int qi = tid; qi < TP; qi += 512) {
    float l = 0.f, dd = 0.f;
    if (qi < T) {
      l = lse[((long long)n * heads + head) * T + qi];
      const float* gp = d_o + ((long long)n * T + qi) * D + head * HD;
      const float* op = o + ((long long)n * T + qi) * D + head * HD;
      for (int d = 0; d < HD; d += 4) {
        const float4 g = *reinterpret_cast<const float4*>(gp + d), ov = *reinterpret_cast<const float4*>(op + d);
        dd += (g.x * ov.x + g.y * ov.y) + (g.z * ov.z + g.w * ov.w);
      }
    }
    Ls[qi] = l;
    Ds[qi] = dd;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int qtr = T >> 5, tr = T & 31;
  // lone (T % 32 == 1): the last token is a tile of its own on both axes.  As a KEY it would be a second round for wave 0 -> one wave
  // computes its dK / dV rows with plain FMAs (below); as a QUERY it would be a ninth MFMA iteration of every wave's loop -> a rank-1
  // update of the accumulators instead.
  const int nkt = ((T + 31) >> 5) - (lone ? 1 : 0);
  const int kt0 = SPLIT ? blockIdx.x - bh * nkt_all : wave;
  for (int kt = kt0; kt < nkt; kt += SPLIT ? nkt_all : 8) {
    const int key = kt * 32 + l31, kc = min(key, T - 1);
    f32x4 kf[KB], vf[KB];
    constexpr int KS = (HD + 15) / 16;
    bsplit8 kh[X3 ? KS : 1], kl[X3 ? KS : 1], vh[X3 ? KS : 1], vl[X3 ? KS : 1];   // lane (key l31, half hh): channels 16 j + 8 hh .. + 7 (B operands of S and dP)
    if constexpr (X3) {
      const float* rowp = base + (long long)kc * D3;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        float k8[8], v8[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int d0 = 16 * j + 8 * hh + 4 * u;
          float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
          if (d0 < HD) {
            kv = *reinterpret_cast<const float4*>(rowp + D + d0);
            if (d0 < R) kv = rotate4(kv, cos_tab, sin_tab, kc * rot_half + (d0 >> 1), false);
            vv = *reinterpret_cast<const float4*>(rowp + 2 * D + d0);
          }
          k8[4 * u] = kv.x; k8[4 * u + 1] = kv.y; k8[4 * u + 2] = kv.z; k8[4 * u + 3] = kv.w;
          v8[4 * u] = vv.x; v8[4 * u + 1] = vv.y; v8[4 * u + 2] = vv.z; v8[4 * u + 3] = vv.w;
        }
        bwd_split8(k8, kh[j], kl[j]);
        bwd_split8(v8, vh[j], vl[j]);
      }
    }
    auto load_kf = [&]() __attribute__((always_inline)) {          // the fp32 fragments: the fp32 MFMAs' B operands; the lone-query update's rows
      const float* rowp = base + (long long)kc * D3;
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int d0 = 8 * j + 4 * hh;
        float4 kv = *reinterpret_cast<const float4*>(rowp + D + d0);
        if (d0 < R) kv = rotate4(kv, cos_tab, sin_tab, kc * rot_half + (d0 >> 1), false);
        const float4 vv = *reinterpret_cast<const float4*>(rowp + 2 * D + d0);
        kf[j][0] = kv.x; kf[j][1] = kv.y; kf[j][2] = kv.z; kf[j][3] = kv.w;
        vf[j][0] = vv.x; vf[j][1] = vv.y; vf[j][2] = vv.z; vf[j][3] = vv.w;
      }
    };
    if constexpr (!X3) load_kf();
    const bool key_ok = key < T;
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }
#pragma unroll 1
    for (int qt = SPLIT ? wave : 0; qt < NKT; qt += SPLIT ? 8 : 1) {
      if (qt * 32 >= T - (lone ? 1 : 0)) break;
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
      if constexpr (X3) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          const bool past = 16 * j + 8 * hh >= HD;
          bsplit8 qfh, qfl, gfh, gfl;
          PsImg<HD>::row8(Qs, qt * 32 + l31, 16 * j + 8 * hh, past, qfh, qfl);
          PsImg<HD>::row8(Gs, qt * 32 + l31, 16 * j + 8 * hh, past, gfh, gfl);
          mfma_x3(s, qfh, qfl, kh[j], kl[j]);                  // S[query][key]
          mfma_x3(dp, gfh, gfl, vh[j], vl[j]);                 // dP[query][key]
        }
      } else {
      const float* qp = Qs + (qt * 32 + l31) * HDP + 4 * hh;
      const float* gp = Gs + (qt * 32 + l31) * HDP + 4 * hh;
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const f32x4 qf = *reinterpret_cast<const f32x4*>(qp + 8 * j);
        const f32x4 gf = *reinterpret_cast<const f32x4*>(gp + 8 * j);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[u], kf[j][u], s, 0, 0, 0);     // S[query][key]
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[u], vf[j][u], dp, 0, 0, 0);   // dP[query][key]
        }
      }
      }
      f32x16 p, ds;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qrow = (e & 3) + 8 * (e >> 2) + 4 * hh;
        float pv = exp_le0(s[e] - Ls[qt * 32 + qrow]);
        if (!key_ok || (qt == qtr && qrow >= tr)) pv = 0.f;                          // padded key column / ragged query tile
        p[e] = pv;
        ds[e] = pv * (dp[e] - Ds[qt * 32 + qrow]);
      }
      if constexpr (X3) {   // dV^T[d][key] += dO^T[d][query] P[query][key], dK^T[d][key] += Q^T[d][query] dS[query][key]
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          float p8[8], d8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            p8[j] = p[8 * h2 + j];
            d8[j] = ds[8 * h2 + j];
          }
          bsplit8 ph, pl, dsh, dsl;
          bwd_split8(p8, ph, pl);
          bwd_split8(d8, dsh, dsl);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            bsplit8 ah, al;
            const int dcol = min(dt * 32 + l31, HD - 1);          // (channels past hd: accumulator rows nobody stores)
            PsImg<HD>::col8(Gs, qt * 32, dcol, h2, hh, ah, al);
            mfma_x3(dv[dt], ah, al, ph, pl);
            PsImg<HD>::col8(Qs, qt * 32, dcol, h2, hh, ah, al);
            mfma_x3(dk[dt], ah, al, dsh, dsl);
          }
        }
      } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int qrow = qt * 32 + (u & 3) + 8 * (u >> 2) + 4 * hh;
        const float* gr = Gs + qrow * HDP + l31;
        const float* qr = Qs + qrow * HDP + l31;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[dt * 32], p[u], dv[dt], 0, 0, 0);    // dV^T[d][key]
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[dt * 32], ds[u], dk[dt], 0, 0, 0);   // dK_rot^T[d][key]
        }
        if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      }
    }
    if constexpr (!SPLIT) {
      if (lone) {   // the lone last query q* against this wave's 32 keys: s, dp from the K / V fragments the lanes hold, then
        if constexpr (X3) load_kf();              // (not kept through the query loop in the x3 kernel: 64 registers)
        const int qs_ = T - 1;                    // dV^T[d][key] += dO[q*][d] p[key], dK^T[d][key] += Q[q*][d] ds[key]
        float sp = 0.f, dpp = 0.f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          const float4 qq = img_ld4<HD, X3>(Qs, qs_, 8 * j + 4 * hh), gq = img_ld4<HD, X3>(Gs, qs_, 8 * j + 4 * hh);
          sp += (kf[j][0] * qq.x + kf[j][1] * qq.y) + (kf[j][2] * qq.z + kf[j][3] * qq.w);
          dpp += (vf[j][0] * gq.x + vf[j][1] * gq.y) + (vf[j][2] * gq.z + vf[j][3] * gq.w);
        }
        sp += __shfl_xor(sp, 32, 64);
        dpp += __shfl_xor(dpp, 32, 64);
        const float pv = key_ok ? exp_le0(sp - Ls[qs_]) : 0.f;
        const float dsv = pv * (dpp - Ds[qs_]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d = dt * 32 + 8 * g + 4 * hh;          // registers 4g..4g+3 of tile dt = channels d..d+3 (reads past hd: never stored)
            const float4 gq = img_ld4<HD, X3>(Gs, qs_, d), qq = img_ld4<HD, X3>(Qs, qs_, d);
            dv[dt][4 * g] = fmaf(gq.x, pv, dv[dt][4 * g]); dv[dt][4 * g + 1] = fmaf(gq.y, pv, dv[dt][4 * g + 1]);
            dv[dt][4 * g + 2] = fmaf(gq.z, pv, dv[dt][4 * g + 2]); dv[dt][4 * g + 3] = fmaf(gq.w, pv, dv[dt][4 * g + 3]);
            dk[dt][4 * g] = fmaf(qq.x, dsv, dk[dt][4 * g]); dk[dt][4 * g + 1] = fmaf(qq.y, dsv, dk[dt][4 * g + 1]);
            dk[dt][4 * g + 2] = fmaf(qq.z, dsv, dk[dt][4 * g + 2]); dk[dt][4 * g + 3] = fmaf(qq.w, dsv, dk[dt][4 * g + 3]);
          }
      }
    }
    if constexpr (SPLIT) {   // partial dK / dV -> LDS [wave][dK | dV][key][HDP] over the Q / dO images (launch_bwd sizes the request for it)
      __syncthreads();
      float* part = smem + (size_t)wave * 2 * 32 * HDP;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * hh;
          if (d < HD) {
            *reinterpret_cast<float4*>(part + l31 * HDP + d) = make_float4(dk[dt][4 * g], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]);
            *reinterpret_cast<float4*>(part + 32 * HDP + l31 * HDP + d) = make_float4(dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
          }
        }
      __syncthreads();
      for (int e = tid; e < 2 * 32 * (HD / 4); e += 512) {
        const int which = e / (32 * (HD / 4)), r = e - which * 32 * (HD / 4);
        const int ki = r / (HD / 4), d = (r - ki * (HD / 4)) * 4, kk = kt * 32 + ki;
        if (kk >= T) continue;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const float4 b = *reinterpret_cast<const float4*>(smem + (size_t)w * 2 * 32 * HDP + which * 32 * HDP + ki * HDP + d);
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (which == 0 && d < R) a = rotate4(a, cos_tab, sin_tab, kk * rot_half + (d >> 1), true);
        dqkv_store4(dqkv, (long long)n * T + kk, D3, head * HD + (which == 0 ? D : 2 * D) + d, a, osplit);
      }
    } else
    if (key_ok) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * hh;
          if (d >= HD) continue;
          float4 kv = make_float4(dk[dt][4 * g], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]);
          if (d < R) kv = rotate4(kv, cos_tab, sin_tab, key * rot_half + (d >> 1), true);
          dqkv_store4(dqkv, (long long)n * T + key, D3, head * HD + D + d, kv, osplit);
          dqkv_store4(dqkv, (long long)n * T + key, D3, head * HD + 2 * D + d, make_float4(dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]), osplit);
        }
    }
  }
  if constexpr (!SPLIT) {
    if (lone && wave == 7) {
      // dK, dV of the lone last key k* = T - 1 without MFMA: phase A, lane = query: p = exp(Q[q] . K* - lse[q]), ds = p (dO[q] . V* - D[q]);
      // phase B, lane = channel: dV*[d] = sum_q p[q] dO[q][d], dK*[d] = sum_q ds[q] Q[q][d].  Scratch rows behind lse / D.
      float* scr = Ds + TP;                      // krow[HDP] | vrow[HDP] | p[TP] | ds[TP] | out[HDP]
      float* krow = scr;
      float* vrow = scr + HDP;
      float* pr = scr + 2 * HDP;
      float* dsr = pr + TP;
      float* outr = dsr + TP;
      const int ks_ = T - 1;
      const float* rowp = base + (long long)ks_ * D3;
      for (int d0 = lane * 4; d0 < HD; d0 += 256) {
        float4 kv = *reinterpret_cast<const float4*>(rowp + D + d0);
        if (d0 < R) kv = rotate4(kv, cos_tab, sin_tab, ks_ * rot_half + (d0 >> 1), false);
        *reinterpret_cast<float4*>(krow + d0) = kv;
        *reinterpret_cast<float4*>(vrow + d0) = *reinterpret_cast<const float4*>(rowp + 2 * D + d0);
      }
      for (int qi = lane; qi < TP; qi += 64) {
        float sacc = 0.f, dpa = 0.f;
        if (qi < T) {
#pragma unroll 4
          for (int d0 = 0; d0 < HD; d0 += 4) {
            const float4 qq = img_ld4<HD, X3>(Qs, qi, d0), gq = img_ld4<HD, X3>(Gs, qi, d0);
            const float4 kq = *reinterpret_cast<const float4*>(krow + d0), vq = *reinterpret_cast<const float4*>(vrow + d0);
            sacc += (qq.x * kq.x + qq.y * kq.y) + (qq.z * kq.z + qq.w * kq.w);
            dpa += (gq.x * vq.x + gq.y * vq.y) + (gq.z * vq.z + gq.w * vq.w);
          }
        }
        const float pv = qi < T ? exp_le0(sacc - Ls[qi]) : 0.f;
        pr[qi] = pv;
        dsr[qi] = pv * (dpa - Ds[qi < T ? qi : 0]);
      }
      for (int d = lane; d < HD; d += 64) {
        float av = 0.f, ak = 0.f;
        for (int qi = 0; qi < T; ++qi) {
          av = fmaf(pr[qi], img_ld1<HD, X3>(Gs, qi, d), av);
          ak = fmaf(dsr[qi], img_ld1<HD, X3>(Qs, qi, d), ak);
        }
        dqkv_store1(dqkv, (long long)n * T + ks_, D3, head * HD + 2 * D + d, av, osplit);
        outr[d] = ak;
      }
      for (int d0 = lane * 4; d0 < HD; d0 += 256) {
        float4 v = *reinterpret_cast<const float4*>(outr + d0);
        if (d0 < R) v = rotate4(v, cos_tab, sin_tab, ks_ * rot_half + (d0 >> 1), true);
        dqkv_store4(dqkv, (long long)n * T + ks_, D3, head * HD + D + d0, v, osplit);
      }
    }
  }
}

static int g_attn_split = -1;   // rgm_set_attn_split: -1 auto, 0 never, 1 always

template <int HD, int NKT, bool X3>
static int launch_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, const float* ct,
                      const float* st, int N, int T, int heads, int rot_half, hipStream_t s, int osplit) {
  constexpr int TP = NKT * 32, HDP = HD + 4;
  // one workgroup per CU (common.h attn_prepare_kernel): the same single-pass structure as the forward kernels
  const size_t images = (size_t)2 * TP * HDP * sizeof(float);
  const int nt = (T + 31) / 32;
  // per-tile workgroups when (sample, head) workgroups leave most CUs idle: the classifiers at C4's batch (24 pairs) or on one row of a
  // sharded step (6).  Every per-tile workgroup stages the head's K / V (Q / dO) again, which costs more than it buys from ~120 pairs on
  // (tools/cls_time.py, value-and-gradient of DiTRotary-S/8-cls: B = 1 4.58 -> 2.34 ms, B = 4 4.89 -> 2.61, B = 16 6.09 -> 5.69,
  // B = 24 7.17 -> 8.09, B = 32 7.99 -> 9.65)
  const bool split = g_attn_split < 0 ? (long long)N * heads <= ATTN_SPLIT_MAX_PAIRS && nt > 1 : g_attn_split == 1;
  if (split) {
    const size_t lds_q = attn_lds_one_per_cu(images > (size_t)8 * 32 * HDP * 4 ? images : (size_t)8 * 32 * HDP * 4);
    const size_t lds_kv = attn_lds_one_per_cu((images > (size_t)16 * 32 * HDP * 4 ? images : (size_t)16 * 32 * HDP * 4) + (size_t)2 * TP * sizeof(float));
    auto kq = attn_bwd_dq_kernel<HD, NKT, true, X3>;
    auto kkv = attn_bwd_dkv_kernel<HD, NKT, true, X3>;
    static bool prepared = false;
    if (!prepared) {
      RGM_TRY(attn_prepare_kernel(kq, 512, lds_q, "attn_bwd_dq_kernel (per query tile)"));
      RGM_TRY(attn_prepare_kernel(kkv, 512, lds_kv, "attn_bwd_dkv_kernel (per key tile)"));
      prepared = true;
    }
    hipLaunchKernelGGL(kq, dim3(N * heads * nt), dim3(512), lds_q, s, qkv, o, d_o, lse, dqkv, ct, st, T, heads, rot_half, 0, osplit);
    RGM_LAUNCH_CHECK();
    hipLaunchKernelGGL(kkv, dim3(N * heads * nt), dim3(512), lds_kv, s, qkv, o, d_o, lse, dqkv, ct, st, T, heads, rot_half, 0, osplit);
    RGM_LAUNCH_CHECK();
    return RGM_OK;
  }
  // the lone-token paths (T % 32 == 1 with more than one tile: the classifiers' 257 tokens) keep a few scratch rows behind the images
  const size_t scr_q = (size_t)(3 * HDP + TP) * sizeof(float), scr_kv = (size_t)(3 * HDP + 2 * TP) * sizeof(float);
  const size_t lds_q = attn_lds_one_per_cu(images + scr_q);
  const size_t lds_kv = attn_lds_one_per_cu(images + (size_t)2 * TP * sizeof(float) + scr_kv);
  static const int lone_off = getenv("RGM_ATTN_LONE") ? !atoi(getenv("RGM_ATTN_LONE")) : 0;      // RGM_ATTN_LONE=0: every tile through the MFMAs (A/B)
  const int lone = (!lone_off && (T & 31) == 1 && nt > 1 && lds_kv <= 160 * 1024) ? 1 : 0;
  auto kq = attn_bwd_dq_kernel<HD, NKT, false, X3>;
  auto kkv = attn_bwd_dkv_kernel<HD, NKT, false, X3>;
  static bool prepared = false;
  if (!prepared) {
    RGM_TRY(attn_prepare_kernel(kq, 512, lds_q <= 160 * 1024 ? lds_q : attn_lds_one_per_cu(images), "attn_bwd_dq_kernel"));
    RGM_TRY(attn_prepare_kernel(kkv, 512, lds_kv <= 160 * 1024 ? lds_kv : attn_lds_one_per_cu(images + (size_t)2 * TP * sizeof(float)), "attn_bwd_dkv_kernel"));
    prepared = true;
  }
  const size_t use_q = lds_q <= 160 * 1024 ? lds_q : attn_lds_one_per_cu(images);
  const size_t use_kv = lds_kv <= 160 * 1024 ? lds_kv : attn_lds_one_per_cu(images + (size_t)2 * TP * sizeof(float));
  hipLaunchKernelGGL(kq, dim3(N * heads), dim3(512), use_q, s, qkv, o, d_o, lse, dqkv, ct, st, T, heads, rot_half, lone, osplit);
  RGM_LAUNCH_CHECK();
  hipLaunchKernelGGL(kkv, dim3(N * heads), dim3(512), use_kv, s, qkv, o, d_o, lse, dqkv, ct, st, T, heads, rot_half, lone, osplit);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

int rotary_attention_bwd_launch(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                                const float* cos_tab, const float* sin_tab, int N, int T, int heads, int hd, int rot_half,
                                hipStream_t s, int osplit) {
  RGM_REQUIRE(!osplit || (heads * hd) % 32 == 0, "attention backward: split rows need 3 x heads x head_dim in 32-blocks");
  RGM_REQUIRE(hd == 64 || hd == 72, "attention backward: head_dim %d (64 = the S/B family, 72 = XL)", hd);
  RGM_REQUIRE(T > 0 && T <= 288, "attention backward: T=%d", T);
  const int nkt = (T + 31) / 32;
  // the bf16x3 modes run the x3 kernels (the forward attention and every GEMM of the chain already compute that way); fp32 mode and
  // RGM_ATTN_BWD_X3=0 (A/B runs) the fp32 MFMAs
  static const int x3_off = getenv("RGM_ATTN_BWD_X3") ? !atoi(getenv("RGM_ATTN_BWD_X3")) : 0;
  const bool x3 = rgm_get_gemm_precision() != 0 && !x3_off;
#define RGM_BWD_GO(HDv, NKTv)                                                                                                       \
  return x3 ? launch_bwd<HDv, NKTv, true>(qkv, o, d_o, lse, dqkv, cos_tab, sin_tab, N, T, heads, rot_half, s, osplit)               \
            : launch_bwd<HDv, NKTv, false>(qkv, o, d_o, lse, dqkv, cos_tab, sin_tab, N, T, heads, rot_half, s, osplit)
  if (hd == 72) {   // XL eps-network (DPS guidance): Q/dO resp. K/V of one head + lse/D = 157.7 KB of LDS at T = 256
    RGM_REQUIRE(nkt <= 8, "attention backward: head_dim 72 supports T <= 256, got %d", T);
    if (nkt <= 4) { RGM_BWD_GO(72, 4); }
    RGM_BWD_GO(72, 8);
  }
  if (nkt <= 4) { RGM_BWD_GO(64, 4); }
  if (nkt <= 5) { RGM_BWD_GO(64, 5); }
  if (nkt <= 8) { RGM_BWD_GO(64, 8); }
  RGM_BWD_GO(64, 9);
#undef RGM_BWD_GO
}

int attn_split_mode() { return g_attn_split; }
void attn_set_split(int mode) { g_attn_split = mode; }

}  // namespace rgm

// Attention kernels of the classifier path (T = 257: 9 tiles on 8 waves; a few dozen (sample, head) pairs on 256 CUs): -1 (default) = one
// workgroup per (sample, head, tile) whenever (sample, head) workgroups would not fill the chip, 0 = never, 1 = always (A/B runs, tests)
extern "C" int rgm_set_attn_split(int mode) {
  RGM_REQUIRE(mode >= -1 && mode <= 1, "set_attn_split: %d (-1 auto, 0 never, 1 always)", mode);
  rgm::attn_set_split(mode);
  return RGM_OK;
}

extern "C" int rgm_rotary_attention_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                                        const float* cos_tab, const float* sin_tab, int N, int T, int heads, int hd,
                                        int rot_half, void* stream) {
  RGM_REQUIRE(qkv && o && d_o && lse && dqkv && cos_tab && sin_tab, "attention backward: null tensor");
  return rgm::rotary_attention_bwd_launch(qkv, o, d_o, lse, dqkv, cos_tab, sin_tab, N, T, heads, hd, rot_half, (hipStream_t)stream, 0);
}
