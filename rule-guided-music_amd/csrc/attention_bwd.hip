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
template <int HD, int NKT, bool SPLIT>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                          const float* __restrict__ d_o, const float* __restrict__ lse,
                                                          float* __restrict__ dqkv, const float* __restrict__ cos_tab,
                                                          const float* __restrict__ sin_tab, int T, int heads, int rot_half) {
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
  for (int c = tid; c < TP * CPR; c += 512) {
    const int key = c / CPR, d0 = (c - key * CPR) * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (key < T) {
      const float* rowp = base + (long long)key * D3;
      kv = *reinterpret_cast<const float4*>(rowp + D + d0);
      vv = *reinterpret_cast<const float4*>(rowp + 2 * D + d0);
      if (d0 < R) kv = rotate4(kv, cos_tab, sin_tab, key * rot_half + (d0 >> 1), false);
    }
    *reinterpret_cast<float4*>(Ks + key * HDP + d0) = kv;
    *reinterpret_cast<float4*>(Vs + key * HDP + d0) = vv;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const float scale = rsqrtf((float)HD);
  const int nqt = (T + 31) >> 5, ktr = T >> 5, tr = T & 31;
  const int qt0 = SPLIT ? blockIdx.x - bh * nqt_all : wave;
  for (int qt = qt0; qt < nqt; qt += SPLIT ? nqt : 8) {
    const int q = qt * 32 + l31, qc = min(q, T - 1);
    const long long orow = ((long long)n * T + qc) * D + head * HD;
    f32x4 qf[KB], dof[KB];
    float dsum = 0.f;
    {
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
      f32x16 ds;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float p = exp_le0(s[e] - lq);
        if (kt == ktr && (e & 3) + 8 * (e >> 2) + 4 * hh >= tr) p = 0.f;               // ragged last key tile
        ds[e] = p * (dp[e] - dsum);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float* kr = Ks + (kt * 32 + (u & 3) + 8 * (u >> 2) + 4 * hh) * HDP + l31;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[dt * 32], ds[u], dq[dt], 0, 0, 0);
        if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
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
        *reinterpret_cast<float4*>(dqkv + ((long long)n * T + qq) * D3 + head * HD + d) = v;
      }
    } else
    if (q < T) {   // dQ^T[d][query]: lane = query row, registers 4g..4g+3 = channels dt*32 + 8g + 4hh ..+3
      float* op = dqkv + ((long long)n * T + q) * D3 + head * HD;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * hh;
          if (d >= HD) continue;
          float4 v = make_float4(dq[dt][4 * g] * scale, dq[dt][4 * g + 1] * scale, dq[dt][4 * g + 2] * scale, dq[dt][4 * g + 3] * scale);
          if (d < R) v = rotate4(v, cos_tab, sin_tab, q * rot_half + (d >> 1), true);
          *reinterpret_cast<float4*>(op + d) = v;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------- dK, dV
// SPLIT: one workgroup per (sample, head, KEY TILE); wave w takes query tiles w, w + 8, ...; partial dK / dV tiles summed through LDS
template <int HD, int NKT, bool SPLIT>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                           const float* __restrict__ d_o, const float* __restrict__ lse,
                                                           float* __restrict__ dqkv, const float* __restrict__ cos_tab,
                                                           const float* __restrict__ sin_tab, int T, int heads, int rot_half) {
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
  for (int c = tid; c < TP * CPR; c += 512) {
    const int qi = c / CPR, d0 = (c - qi * CPR) * 4;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
    if (qi < T) {
      qv = *reinterpret_cast<const float4*>(base + (long long)qi * D3 + d0);
      if (d0 < R) qv = rotate4(qv, cos_tab, sin_tab, qi * rot_half + (d0 >> 1), false);
      qv = make_float4(qv.x * scale, qv.y * scale, qv.z * scale, qv.w * scale);
      gv = *reinterpret_cast<const float4*>(d_o + ((long long)n * T + qi) * D + head * HD + d0);
    }
    *reinterpret_cast<float4*>(Qs + qi * HDP + d0) = qv;
    *reinterpret_cast<float4*>(Gs + qi * HDP + d0) = gv;
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
  const int nkt = (T + 31) >> 5, qtr = T >> 5, tr = T & 31;
  const int kt0 = SPLIT ? blockIdx.x - bh * nkt_all : wave;
  for (int kt = kt0; kt < nkt; kt += SPLIT ? nkt : 8) {
    const int key = kt * 32 + l31, kc = min(key, T - 1);
    f32x4 kf[KB], vf[KB];
    {
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
    }
    const bool key_ok = key < T;
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }
#pragma unroll 1
    for (int qt = SPLIT ? wave : 0; qt < NKT; qt += SPLIT ? 8 : 1) {
      if (qt * 32 >= T) break;
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
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
      f32x16 p, ds;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qrow = (e & 3) + 8 * (e >> 2) + 4 * hh;
        float pv = exp_le0(s[e] - Ls[qt * 32 + qrow]);
        if (!key_ok || (qt == qtr && qrow >= tr)) pv = 0.f;                          // padded key column / ragged query tile
        p[e] = pv;
        ds[e] = pv * (dp[e] - Ds[qt * 32 + qrow]);
      }
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
        *reinterpret_cast<float4*>(dqkv + ((long long)n * T + kk) * D3 + head * HD + (which == 0 ? D : 2 * D) + d) = a;
      }
    } else
    if (key_ok) {
      float* op = dqkv + ((long long)n * T + key) * D3 + head * HD;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * hh;
          if (d >= HD) continue;
          float4 kv = make_float4(dk[dt][4 * g], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]);
          if (d < R) kv = rotate4(kv, cos_tab, sin_tab, key * rot_half + (d >> 1), true);
          *reinterpret_cast<float4*>(op + D + d) = kv;
          *reinterpret_cast<float4*>(op + 2 * D + d) = make_float4(dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
        }
    }
  }
}

static int g_attn_split = -1;   // rgm_set_attn_split: -1 auto, 0 never, 1 always

template <int HD, int NKT>
static int launch_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, const float* ct,
                      const float* st, int N, int T, int heads, int rot_half, hipStream_t s) {
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
    auto kq = attn_bwd_dq_kernel<HD, NKT, true>;
    auto kkv = attn_bwd_dkv_kernel<HD, NKT, true>;
    static bool prepared = false;
    if (!prepared) {
      RGM_TRY(attn_prepare_kernel(kq, 512, lds_q, "attn_bwd_dq_kernel (per query tile)"));
      RGM_TRY(attn_prepare_kernel(kkv, 512, lds_kv, "attn_bwd_dkv_kernel (per key tile)"));
      prepared = true;
    }
    hipLaunchKernelGGL(kq, dim3(N * heads * nt), dim3(512), lds_q, s, qkv, o, d_o, lse, dqkv, ct, st, T, heads, rot_half);
    RGM_LAUNCH_CHECK();
    hipLaunchKernelGGL(kkv, dim3(N * heads * nt), dim3(512), lds_kv, s, qkv, o, d_o, lse, dqkv, ct, st, T, heads, rot_half);
    RGM_LAUNCH_CHECK();
    return RGM_OK;
  }
  const size_t lds_q = attn_lds_one_per_cu(images);
  const size_t lds_kv = attn_lds_one_per_cu(images + (size_t)2 * TP * sizeof(float));
  auto kq = attn_bwd_dq_kernel<HD, NKT, false>;
  auto kkv = attn_bwd_dkv_kernel<HD, NKT, false>;
  static bool prepared = false;
  if (!prepared) {
    RGM_TRY(attn_prepare_kernel(kq, 512, lds_q, "attn_bwd_dq_kernel"));
    RGM_TRY(attn_prepare_kernel(kkv, 512, lds_kv, "attn_bwd_dkv_kernel"));
    prepared = true;
  }
  hipLaunchKernelGGL(kq, dim3(N * heads), dim3(512), lds_q, s, qkv, o, d_o, lse, dqkv, ct, st, T, heads, rot_half);
  RGM_LAUNCH_CHECK();
  hipLaunchKernelGGL(kkv, dim3(N * heads), dim3(512), lds_kv, s, qkv, o, d_o, lse, dqkv, ct, st, T, heads, rot_half);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

int rotary_attention_bwd_launch(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                                const float* cos_tab, const float* sin_tab, int N, int T, int heads, int hd, int rot_half,
                                hipStream_t s) {
  RGM_REQUIRE(hd == 64 || hd == 72, "attention backward: head_dim %d (64 = the S/B family, 72 = XL)", hd);
  RGM_REQUIRE(T > 0 && T <= 288, "attention backward: T=%d", T);
  const int nkt = (T + 31) / 32;
  if (hd == 72) {   // XL eps-network (DPS guidance): Q/dO resp. K/V of one head + lse/D = 157.7 KB of LDS at T = 256
    RGM_REQUIRE(nkt <= 8, "attention backward: head_dim 72 supports T <= 256, got %d", T);
    if (nkt <= 4) return launch_bwd<72, 4>(qkv, o, d_o, lse, dqkv, cos_tab, sin_tab, N, T, heads, rot_half, s);
    return launch_bwd<72, 8>(qkv, o, d_o, lse, dqkv, cos_tab, sin_tab, N, T, heads, rot_half, s);
  }
  if (nkt <= 4) return launch_bwd<64, 4>(qkv, o, d_o, lse, dqkv, cos_tab, sin_tab, N, T, heads, rot_half, s);
  if (nkt <= 5) return launch_bwd<64, 5>(qkv, o, d_o, lse, dqkv, cos_tab, sin_tab, N, T, heads, rot_half, s);
  if (nkt <= 8) return launch_bwd<64, 8>(qkv, o, d_o, lse, dqkv, cos_tab, sin_tab, N, T, heads, rot_half, s);
  return launch_bwd<64, 9>(qkv, o, d_o, lse, dqkv, cos_tab, sin_tab, N, T, heads, rot_half, s);
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
  return rgm::rotary_attention_bwd_launch(qkv, o, d_o, lse, dqkv, cos_tab, sin_tab, N, T, heads, hd, rot_half, (hipStream_t)stream);
}
