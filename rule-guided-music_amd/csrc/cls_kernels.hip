// cls_kernels.hip -- bandwidth-bound pieces of the classifier-guidance backward (input gradient only).
//
// Reference: autograd of guided_diffusion/dit.py:803-831 under guided_diffusion/condition_functions.py:58-85.
//   ln_mod_bwd     d/dx of  LN(x) * (1 + scale) + shift   (or affine LN): one wave per row, row in registers
//   gate_rows      dx * gate[b]            (adaLN gates are constants w.r.t. x)
//   act_rows       SiLU / GELU(tanh) forward on a saved pre-activation
//   mse_grad / ce_grad   d(log p)/d(logits) * classifier_scale, written into a 32-column zero-padded buffer
//   scatter_rows   head gradient -> token rows of the residual-stream gradient (cls row, or mean-pool broadcast)
#include "common.h"

namespace rgm {

__device__ __forceinline__ void store4(float* __restrict__ out, long long row, int D, int c, float4 v, int out_split) {
  if (out_split) {
    typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, lo;
    hi[0] = (split_t)v.x; hi[1] = (split_t)v.y; hi[2] = (split_t)v.z; hi[3] = (split_t)v.w;
    lo[0] = (split_t)(v.x - (float)hi[0]); lo[1] = (split_t)(v.y - (float)hi[1]);
    lo[2] = (split_t)(v.z - (float)hi[2]); lo[3] = (split_t)(v.w - (float)hi[3]);
    split_t* rp = reinterpret_cast<split_t*>(out + row * D);
    *reinterpret_cast<bf16x4*>(rp + split_idx(c)) = hi;
    *reinterpret_cast<bf16x4*>(rp + split_idx(c) + 32) = lo;
  } else {
    *reinterpret_cast<float4*>(out + row * D + c) = v;
  }
}

template <int MAXV>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ res, float* __restrict__ out, int M, int D,
                                                         float eps, const float* __restrict__ weight,
                                                         const float* __restrict__ scale, int mod_ld, int rows_per_batch,
                                                         const float* __restrict__ gate2, float* __restrict__ gated) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const int nv = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (long long)row * D);
  const float4* gr = reinterpret_cast<const float4*>(dy + (long long)row * D);
  const long long mo = (long long)(row / rows_per_batch) * mod_ld;
  float4 v[MAXV], g[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    v[i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    g[i] = c < nv ? gr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nv) {
      if (scale) {
        const float4 sc = reinterpret_cast<const float4*>(scale + mo)[c];
        g[i] = make_float4(g[i].x * (1.f + sc.x), g[i].y * (1.f + sc.y), g[i].z * (1.f + sc.z), g[i].w * (1.f + sc.w));
      } else if (weight) {
        const float4 w = reinterpret_cast<const float4*>(weight)[c];
        g[i] = make_float4(g[i].x * w.x, g[i].y * w.y, g[i].z * w.z, g[i].w * w.w);
      }
    }
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      v[i] = make_float4(v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean);
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      v[i] = make_float4(v[i].x * rstd, v[i].y * rstd, v[i].z * rstd, v[i].w * rstd);   // xhat
      sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      sgx += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
    }
  }
  const float mg = wave_sum(sg) / (float)D, mgx = wave_sum(sgx) / (float)D;
  float4* orow = reinterpret_cast<float4*>(out + (long long)row * D);
  const float4* rrow = res ? reinterpret_cast<const float4*>(res + (long long)row * D) : nullptr;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c >= nv) continue;
    float4 o = make_float4((g[i].x - mg - v[i].x * mgx) * rstd, (g[i].y - mg - v[i].y * mgx) * rstd,
                           (g[i].z - mg - v[i].z * mgx) * rstd, (g[i].w - mg - v[i].w * mgx) * rstd);
    if (rrow) {
      const float4 r = rrow[c];
      o = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w);
    }
    orow[c] = o;
    if (gated) {   // the next GEMM's operand in the same pass: out * gate2 (gate_rows_kernel's product) as split rows
      const float4 g2 = reinterpret_cast<const float4*>(gate2 + mo)[c];
      store4(gated, row, D, c * 4, make_float4(o.x * g2.x, o.y * g2.y, o.z * g2.z, o.w * g2.w), 1);
    }
  }
}

int ln_mod_bwd_launch(const float* dy, const float* x, const float* res, float* out, int M, int D, float eps,
                      const float* weight, const float* scale, int mod_ld, int rows_per_batch, hipStream_t s, const float* gate2,
                      float* gated) {
  RGM_REQUIRE(M > 0 && (D & 3) == 0 && D <= 2048, "ln_mod_bwd: D=%d", D);
  RGM_REQUIRE(!gated || (gate2 && (D & 31) == 0), "ln_mod_bwd: the gated split output needs a gate and D%%32==0");
  if (rows_per_batch <= 0) rows_per_batch = 1;
  dim3 grid(cdiv(M, 4)), block(256);
  const int nv = D / 4;
  if (nv <= 128)
    hipLaunchKernelGGL(ln_mod_bwd_kernel<2>, grid, block, 0, s, dy, x, res, out, M, D, eps, weight, scale, mod_ld, rows_per_batch, gate2, gated);
  else if (nv <= 320)
    hipLaunchKernelGGL(ln_mod_bwd_kernel<5>, grid, block, 0, s, dy, x, res, out, M, D, eps, weight, scale, mod_ld, rows_per_batch, gate2, gated);
  else
    hipLaunchKernelGGL(ln_mod_bwd_kernel<8>, grid, block, 0, s, dy, x, res, out, M, D, eps, weight, scale, mod_ld, rows_per_batch, gate2, gated);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// write 4 consecutive values of a row either as fp32 or in the split-row format of the pre-split GEMMs (common.h split_idx)

__global__ void gate_rows_kernel(const float* __restrict__ dx, const float* __restrict__ gate, float* __restrict__ out,
                                 long long total4, int D, int gate_ld, int rows_per_batch, int out_split) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int q = D >> 2;
  const long long row = i / q;
  const int c = (int)(i - row * q) * 4;
  const float4 a = *reinterpret_cast<const float4*>(dx + row * D + c);
  const float4 g = *reinterpret_cast<const float4*>(gate + (row / rows_per_batch) * gate_ld + c);
  store4(out, row, D, c, make_float4(a.x * g.x, a.y * g.y, a.z * g.z, a.w * g.w), out_split);
}
int gate_rows_launch(const float* dx, const float* gate, float* out, int M, int D, int gate_ld, int rows_per_batch, hipStream_t s,
                     int out_split) {
  RGM_REQUIRE((D & 3) == 0 && (gate_ld & 3) == 0 && (!out_split || (D & 31) == 0), "gate_rows: D=%d gate_ld=%d", D, gate_ld);
  const long long total4 = (long long)M * (D >> 2);
  hipLaunchKernelGGL(gate_rows_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, dx, gate, out, total4, D, gate_ld,
                     rows_per_batch, out_split);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// out = act(in) over rows of D values (act 1 silu, 2 gelu-tanh); optionally written as split rows
__global__ void act_rows_kernel(const float* __restrict__ in, float* __restrict__ out, long long total4, int D, int act, int out_split) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int q = D >> 2;
  const long long row = i / q;
  const int c = (int)(i - row * q) * 4;
  const float4 v = *reinterpret_cast<const float4*>(in + row * D + c);
  const float4 r = act == 1 ? make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w))
                            : make_float4(gelu_tanh_f(v.x), gelu_tanh_f(v.y), gelu_tanh_f(v.z), gelu_tanh_f(v.w));
  store4(out, row, D, c, r, out_split);
}
int act_rows_launch(const float* in, float* out, long long rows, int D, int act, hipStream_t s, int out_split) {
  RGM_REQUIRE((D & 3) == 0 && (!out_split || (D & 31) == 0), "act_rows: D=%d", D);
  const long long total4 = rows * (D >> 2);
  hipLaunchKernelGGL(act_rows_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, in, out, total4, D, act, out_split);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// dl[r][0:32) : -2 (logits - target) * scale for k < K, 0 padding         (grad of -sum (logits-target)^2)
__global__ void mse_grad_kernel(const float* __restrict__ logits, const float* __restrict__ target, float* __restrict__ dl,
                                int rows, int K, int Kp, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * Kp) return;
  const int r = i / Kp, k = i - r * Kp;
  dl[i] = k < K ? -2.0f * (logits[r * K + k] - target[r * K + k]) * scale : 0.f;
}
// dl[r][k] = (onehot(target[r]) - softmax(logits[r]))[k] * scale            (grad of -CE)
__global__ void ce_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, float* __restrict__ dl,
                               int rows, int K, int Kp, float scale) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float mx = -INFINITY;
  for (int k = 0; k < K; ++k) mx = fmaxf(mx, logits[r * K + k]);
  float sum = 0.f;
  for (int k = 0; k < K; ++k) sum += expf(logits[r * K + k] - mx);
  const int tg = (int)target[r];
  for (int k = 0; k < Kp; ++k)
    dl[r * Kp + k] = k < K ? ((k == tg ? 1.f : 0.f) - expf(logits[r * K + k] - mx) / sum) * scale : 0.f;
}
int loss_grad_launch(const float* logits, const void* target, float* dl, int rows, int K, int Kp, float scale, int kind, hipStream_t s) {
  if (kind == 0)
    hipLaunchKernelGGL(mse_grad_kernel, dim3(cdiv(rows * Kp, 256)), dim3(256), 0, s, logits, (const float*)target, dl, rows, K, Kp, scale);
  else
    hipLaunchKernelGGL(ce_grad_kernel, dim3(cdiv(rows, 64)), dim3(64), 0, s, logits, (const int64_t*)target, dl, rows, K, Kp, scale);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// dx[n*T + first + g*per + j][:] = src[n*groups + g][:] / per  for j < per ; every other row of dx = 0
__global__ void scatter_rows_kernel(const float* __restrict__ src, float* __restrict__ dx, int N, int T, int D, int first, int groups,
                                    int per) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)N * T * D) return;
  const int d = (int)(i % D);
  const long long row = i / D;
  const int tok = (int)(row % T), n = (int)(row / T);
  const int rel = tok - first;
  float v = 0.f;
  if (rel >= 0 && rel < groups * per) v = src[((long long)n * groups + rel / per) * D + d] / (float)per;
  dx[i] = v;
}
int scatter_rows_launch(const float* src, float* dx, int N, int T, int D, int first, int groups, int per, hipStream_t s) {
  const long long total = (long long)N * T * D;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dx, N, T, D, first, groups, per);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

}  // namespace rgm
