// ln_body.h -- the row bodies of the DiT forward's two LayerNorm kernels as device functions: ln_mod_kernel (dit_kernels.hip) and
// splitk_reduce_ln_kernel (gemm2.hip) wrap them one row per wave; chain.hip runs them as work items of the persistent forward.
#pragma once
#include "common.h"

namespace rgm {

// one wave = one row of ln_mod_kernel (dit_kernels.hip), in two steps so that a wave with several rows (chain.hip: four waves per CU and
// nobody else to hide a row's round trip behind) can request all of them before it finishes the first: ln_mod_load, then ln_mod_finish.
// COH = 1 (chain.hip; split output, D % 8 == 0): device-coherent 16-byte stores
template <int MAXV>
struct LnRow {
  float4 v[MAXV];
};
template <int MAXV>
__device__ __forceinline__ void ln_mod_load(LnRow<MAXV>& st, const float* __restrict__ x, const int row, int D, const int lane_in = -1) {
  const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63);
  const int nv = D >> 2;
  const float* xr = x + (long long)row * D;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    const float4 t = ldg16(xr + 4 * (c < nv ? c : nv - 1));            // unconditional load of a valid address: no branch per chunk
    st.v[i] = c < nv ? t : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
// the modulation row of a sample, loaded once for all the rows a wave holds (chain.hip): MOD = nullptr -> read per row as ln_mod_kernel does
template <int MAXV>
struct LnMod {
  float4 sc[MAXV], sh[MAXV];
};
template <int MAXV>
__device__ __forceinline__ void ln_mod_load_mod(LnMod<MAXV>& md, const float* __restrict__ shift, const float* __restrict__ scale, int D,
                                                long long mo, const int lane_in = -1) {
  const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63);
  const int nv = D >> 2;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    const int cc = c < nv ? c : nv - 1;
    md.sc[i] = ldg16(scale + mo + 4 * cc);
    md.sh[i] = ldg16(shift + mo + 4 * cc);
  }
}
template <int MAXV, int COH = 0, int HAS_MOD = 0>
__device__ __forceinline__ void ln_mod_finish(const LnRow<MAXV>& st, float* __restrict__ out, const int row, int D,
                                              float eps, const float* __restrict__ weight,
                                              const float* __restrict__ bias, const float* __restrict__ shift,
                                              const float* __restrict__ scale, int mod_ld, int rows_per_batch,
                                              int out_split, const LnMod<MAXV>& MOD = LnMod<MAXV>{}, const int lane_in = -1) {
  const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63);
  const int nv = D >> 2;
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    v[i] = st.v[i];
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  const long long mo = (long long)(row / rows_per_batch) * mod_ld;
  float4* orow = reinterpret_cast<float4*>(out + (long long)row * D);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c >= nv) continue;
    float4 y = make_float4((v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd);
    if (weight) {
      const float4 w = reinterpret_cast<const float4*>(weight)[c], b = reinterpret_cast<const float4*>(bias)[c];
      y = make_float4(y.x * w.x + b.x, y.y * w.y + b.y, y.z * w.z + b.z, y.w * w.w + b.w);
    }
    if (scale) {
      float4 sc, sh;
      if constexpr (HAS_MOD) {
        sc = MOD.sc[i];
        sh = MOD.sh[i];
      } else {
        sc = reinterpret_cast<const float4*>(scale + mo)[c];
        sh = reinterpret_cast<const float4*>(shift + mo)[c];
      }
      y = make_float4(y.x * (1.f + sc.x) + sh.x, y.y * (1.f + sc.y) + sh.y, y.z * (1.f + sc.z) + sh.z, y.w * (1.f + sc.w) + sh.w);
    }
    if (out_split) {   // split-row format (common.h split_idx) for the pre-split GEMM path
      typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 hi, lo;
      hi[0] = (split_t)y.x; hi[1] = (split_t)y.y; hi[2] = (split_t)y.z; hi[3] = (split_t)y.w;
      lo[0] = (split_t)(y.x - (float)hi[0]); lo[1] = (split_t)(y.y - (float)hi[1]);
      lo[2] = (split_t)(y.z - (float)hi[2]); lo[3] = (split_t)(y.w - (float)hi[3]);
      split_t* rp = reinterpret_cast<split_t*>(out + (long long)row * D);
      if constexpr (COH) {
        store_split4_pair_sc1<1>(rp, c * 4, hi, lo);       // lanes (2k, 2k+1) hold chunks (2j, 2j+1): one 8-aligned group
      } else {
        store_split4_maybe_pair<1>(rp, c * 4, hi, lo);     // the same pairing, ordinary stores: 16 bytes per lane instead of 2 x 8
      }
    } else {
      orow[c] = y;
    }
  }
}
template <int MAXV, int COH = 0>
__device__ __forceinline__ void ln_mod_row(const float* __restrict__ x, float* __restrict__ out, const int row, int D,
                                           float eps, const float* __restrict__ weight,
                                           const float* __restrict__ bias, const float* __restrict__ shift,
                                           const float* __restrict__ scale, int mod_ld, int rows_per_batch,
                                           int out_split) {
  LnRow<MAXV> st;
  ln_mod_load<MAXV>(st, x, row, D);
  ln_mod_finish<MAXV, COH>(st, out, row, D, eps, weight, bias, shift, scale, mod_ld, rows_per_batch, out_split);
}


// one wave = one row of splitk_reduce_ln_kernel (gemm2.hip).  LN = 0: the reduce alone (the last block of a forward has no next LayerNorm).
// COH = 1 (chain.hip): the reduced row and the LayerNorm row are stored device-coherent, 16 bytes per lane.
// (two steps like ln_mod_row: splitk_reduce_load -- the S partial sums and the residual of a row; the bias and the gate row, shared by the
// rows of a sample, come with splitk_reduce_shared -- then splitk_reduce_ln_finish)
template <int MAXV, int S>
struct RedRow {
  float4 part[MAXV][S], rq[MAXV];
};
template <int MAXV>
struct RedShared {
  float4 bq[MAXV], gq[MAXV];
};
template <int MAXV>
__device__ __forceinline__ void splitk_reduce_shared(RedShared<MAXV>& sh, const GemmParams& p, const int row, const int lane_in = -1) {
  const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63);
  const int nv = p.N >> 2;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    const int col = c * 4;
    const bool ok = c < nv;
    sh.bq[i] = (ok && p.bias) ? ldg16(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    sh.gq[i] = (ok && p.gate) ? ldg16(p.gate + (long long)(row / p.rows_per_gate) * p.gate_ld + col) : make_float4(1.f, 1.f, 1.f, 1.f);
    (void)ok;
  }
}
template <int MAXV, int S>
__device__ __forceinline__ void splitk_reduce_load(RedRow<MAXV, S>& st, const float* __restrict__ P, const GemmParams& p, const int row, const int lane_in = -1) {
  const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63);
  const int nv = p.N >> 2;
  const long long MN = (long long)p.M * p.N;
  // one wave holds the row and there are only M waves (1024 at B = 4): every load of the row -- S partial sums, bias, gate, residual
  // per chunk -- is issued before the first sum (compile-time S and MAXV), or the wave walks through 5 S dependent round trips
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    const bool ok = c < nv;
    const int col = (ok ? c : nv - 1) * 4;                  // unconditional loads of a valid address (no branch per chunk); unused when !ok
#pragma unroll
    for (int sidx = 0; sidx < S; ++sidx) {
      const float4 t = ldg16(P + sidx * MN + (long long)row * p.N + col);
      st.part[i][sidx] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    st.rq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.res) {
      const float4 t = ldg16(p.res + (long long)row * p.ldres + col);
      st.rq[i] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
template <int MAXV, int S, int COH = 0, int LN = 1, int HAS_MOD = 0>
__device__ __forceinline__ void splitk_reduce_ln_finish(const RedRow<MAXV, S>& st, const RedShared<MAXV>& sh, const GemmParams& p, const int row,
                                                        const LnMod<MAXV>& MOD = LnMod<MAXV>{}, const int lane_in = -1) {
  const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63);
  const int nv = p.N >> 2;
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nv) {
      const int col = c * 4;
      float4 a = st.part[i][0];
#pragma unroll
      for (int sidx = 1; sidx < S; ++sidx) {
        const float4 b = st.part[i][sidx];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      float w[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
      if (p.bias) { w[0] += sh.bq[i].x; w[1] += sh.bq[i].y; w[2] += sh.bq[i].z; w[3] += sh.bq[i].w; }
      if (p.gate) { w[0] *= sh.gq[i].x; w[1] *= sh.gq[i].y; w[2] *= sh.gq[i].z; w[3] *= sh.gq[i].w; }
      if (p.res) { w[0] += st.rq[i].x; w[1] += st.rq[i].y; w[2] += st.rq[i].z; w[3] += st.rq[i].w; }
      v[i] = make_float4(w[0], w[1], w[2], w[3]);
      if constexpr (COH) {
        const f32x4 wv = {w[0], w[1], w[2], w[3]};
        store16_sc1(p.C + (long long)row * p.ldc + col, wv);
      } else {
        *reinterpret_cast<float4*>(p.C + (long long)row * p.ldc + col) = v[i];
      }
    }
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  if constexpr (!LN) return;
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
    float4 sc, shf;
    if constexpr (HAS_MOD) {
      sc = MOD.sc[i];
      shf = MOD.sh[i];
    } else {
      sc = reinterpret_cast<const float4*>(p.ln_scale + mo)[c];
      shf = reinterpret_cast<const float4*>(p.ln_shift + mo)[c];
    }
    y = make_float4(y.x * (1.f + sc.x) + shf.x, y.y * (1.f + sc.y) + shf.y, y.z * (1.f + sc.z) + shf.z, y.w * (1.f + sc.w) + shf.w);
    if (p.ln_out_split) {
      typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 hi, lo;
      hi[0] = (split_t)y.x; hi[1] = (split_t)y.y; hi[2] = (split_t)y.z; hi[3] = (split_t)y.w;
      lo[0] = (split_t)(y.x - (float)hi[0]); lo[1] = (split_t)(y.y - (float)hi[1]);
      lo[2] = (split_t)(y.z - (float)hi[2]); lo[3] = (split_t)(y.w - (float)hi[3]);
      split_t* rp = reinterpret_cast<split_t*>(p.ln_out + (long long)row * p.N);
      if constexpr (COH) {
        store_split4_pair_sc1<1>(rp, c * 4, hi, lo);
      } else {
        store_split4_maybe_pair<1>(rp, c * 4, hi, lo);
      }
    } else {
      orow[c] = y;
    }
  }
}
template <int MAXV, int S, int COH = 0, int LN = 1>
__device__ __forceinline__ void splitk_reduce_ln_row(const float* __restrict__ P, const GemmParams& p, const int row) {
  RedRow<MAXV, S> st;
  RedShared<MAXV> sh;
  splitk_reduce_shared<MAXV>(sh, p, row);
  splitk_reduce_load<MAXV, S>(st, P, p, row);
  splitk_reduce_ln_finish<MAXV, S, COH, LN>(st, sh, p, row);
}


}  // namespace rgm
