// dit_kernels.hip -- the bandwidth-bound glue of the DiT forward (HBM-bound; no MFMA here).
//
//   layernorm_modulate   LN(eps, no affine) * (1 + scale) + shift      guided_diffusion/dit.py:25-26, :334-335, :374
//                        (optional affine LN for the classifier head, dit.py:770, :828)
//   patchify / unpatchify FlattenPatchify1D.forward dit.py:219-224 ; DiTRotary.unpatchify dit.py:608-616
//   timestep_sincos      TimestepEmbedder.timestep_embedding dit.py:47-65
//   cond_finish          c = t_emb (+ y_embedder table row) ; SiLU(c)  dit.py:627-629, :333
//   fill_cls / gather / mean-pool rows for the classifier heads        dit.py:813, :817-828
//
// All are one pass over their tensor with 16-byte lanes; LN keeps the row in registers (one wave
// per row, shuffle reductions), so x is read once and the modulated row written once.
#include <stdlib.h>
#include "common.h"
#include "ln_body.h"

namespace rgm {

template <int MAXV>
__global__ __launch_bounds__(256) void ln_mod_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int D,
                                                     float eps, const float* __restrict__ weight,
                                                     const float* __restrict__ bias, const float* __restrict__ shift,
                                                     const float* __restrict__ scale, int mod_ld, int rows_per_batch,
                                                     int out_split, int preload) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  if (preload && scale && !weight) {
    // adaLN rows (every LayerNorm of a DiT block): the sample's shift / scale row is requested WITH the row itself -- read behind the two
    // wave reductions it was a second dependent round trip of every wave of the launch (one wave per row: nothing else hides it)
    LnRow<MAXV> st;
    LnMod<MAXV> md;
    ln_mod_load<MAXV>(st, x, row, D);
    ln_mod_load_mod<MAXV>(md, shift, scale, D, (long long)(row / rows_per_batch) * mod_ld);
    ln_mod_finish<MAXV, 0, 1>(st, out, row, D, eps, nullptr, nullptr, shift, scale, mod_ld, rows_per_batch, out_split, md);
    return;
  }
  ln_mod_row<MAXV>(x, out, row, D, eps, weight, bias, shift, scale, mod_ld, rows_per_batch, out_split);
}

int layernorm_modulate_launch(const float* x, float* out, int M, int D, float eps, const float* weight, const float* bias,
                              const float* shift, const float* scale, int mod_ld, int rows_per_batch, hipStream_t s,
                              int out_split) {
  RGM_REQUIRE(M > 0 && D > 0 && (D & 3) == 0 && D <= 2048, "layernorm: D=%d must be a multiple of 4, <= 2048", D);
  RGM_REQUIRE((scale == nullptr) == (shift == nullptr) && (weight == nullptr) == (bias == nullptr), "layernorm: shift/scale and weight/bias come in pairs");
  RGM_REQUIRE(scale == nullptr || ((mod_ld & 3) == 0 && ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0), "layernorm: modulation rows must be 16-byte aligned");
  if (rows_per_batch <= 0) rows_per_batch = 1;
  dim3 grid(cdiv(M, 4)), block(256);
  const int nv = D / 4;
  static const int preload = getenv("RGM_LN_PRELOAD") ? atoi(getenv("RGM_LN_PRELOAD")) : 1;     // 0: shift / scale read behind the reductions (A/B runs)
  if (nv <= 128)
    hipLaunchKernelGGL(ln_mod_kernel<2>, grid, block, 0, s, x, out, M, D, eps, weight, bias, shift, scale, mod_ld, rows_per_batch, out_split, preload);
  else if (nv <= 320)
    hipLaunchKernelGGL(ln_mod_kernel<5>, grid, block, 0, s, x, out, M, D, eps, weight, bias, shift, scale, mod_ld, rows_per_batch, out_split, preload);
  else
    hipLaunchKernelGGL(ln_mod_kernel<8>, grid, block, 0, s, x, out, M, D, eps, weight, bias, shift, scale, mod_ld, rows_per_batch, out_split, preload);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// x (N,C,H,W) -> tok (N*H*W/P, P*C): token = h*(W/P) + w/P, feature = (w%P)*C + c
__global__ void patchify_kernel(const float* __restrict__ x, float* __restrict__ tok, int N, int C, int H, int W, int P) {
  const long long total = (long long)N * C * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    // i indexes the OUTPUT (contiguous writes): [n][h][w][c]
    const int c = i % C;
    long long r = i / C;
    const int w = r % W;
    r /= W;
    const int h = r % H;
    const int n = r / H;
    tok[i] = x[(((long long)n * C + c) * H + h) * W + w];
  }
}

// tok (N*T, P*OC) -> out (N,OC,H,W), inverse of the map above (out indexed contiguously)
__global__ void unpatchify_kernel(const float* __restrict__ tok, float* __restrict__ out, int N, int OC, int H, int W) {
  const long long total = (long long)N * OC * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int w = i % W;
    long long r = i / W;
    const int h = r % H;
    r /= H;
    const int c = r % OC;
    const int n = r / OC;
    out[i] = tok[(((long long)n * H + h) * W + w) * OC + c];
  }
}

// emb[n][0:half] = cos(t*freq), emb[n][half:] = sin(t*freq)
__global__ void timestep_sincos_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ emb, int N, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * half) return;
  const int n = i / half, k = i - n * half;
  const float a = (float)t[n] * freqs[k];
  emb[(long long)n * 2 * half + k] = cosf(a);
  emb[(long long)n * 2 * half + half + k] = sinf(a);
}

// cs = SiLU(c + table[y])   (c updated in place too when c_out != nullptr)
__global__ void cond_finish_kernel(const float* __restrict__ c, const float* __restrict__ table, const int32_t* __restrict__ y,
                                   float* __restrict__ cs, int N, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D) return;
  const int n = i / D, d = i - n * D;
  float v = c[i];
  if (table && y) v += table[(long long)y[n] * D + d];
  cs[i] = silu_f(v);
}

// x[n*T + 0][:] = cls[:]
__global__ void fill_cls_kernel(const float* __restrict__ cls, float* __restrict__ x, int N, int T, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D) return;
  const int n = i / D, d = i - n * D;
  x[(long long)n * T * D + d] = cls[d];
}

// out[n][g][:] = mean over `per` consecutive token rows starting at x[n*T + first + g*per]; per==1 is a row gather
__global__ void pool_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int T, int D, int first, int groups, int per) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * groups * D) return;
  const int d = i % D;
  const int g = (i / D) % groups;
  const int n = i / (D * groups);
  const float* p = x + ((long long)n * T + first + (long long)g * per) * D + d;
  float s = 0.f;
  for (int k = 0; k < per; ++k) s += p[(long long)k * D];
  out[i] = per == 1 ? s : s / (float)per;
}

// out[b][c*out_ld + r] = in[b][r*Cc + c]  (batched transpose through a padded LDS tile)
__global__ void transpose_batched_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc, int out_ld) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const float* ib = in + (long long)b * R * Cc;
  float* ob = out + (long long)b * Cc * out_ld;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    if (r < R && c < Cc) tile[i][threadIdx.x] = ib[(long long)r * Cc + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < Cc) ob[(long long)c * out_ld + r] = tile[threadIdx.x][i];
  }
}

int transpose_launch(const float* in, float* out, int R, int Cc, int out_ld, int batch, hipStream_t s) {
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(cdiv(Cc, 32), cdiv(R, 32), batch), dim3(32, 8), 0, s, in, out, R, Cc, out_ld);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

int patchify_launch(const float* x, float* tok, int N, int C, int H, int W, int P, hipStream_t s) {
  const long long total = (long long)N * C * H * W;
  hipLaunchKernelGGL(patchify_kernel, dim3((int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, s, x, tok, N, C, H, W, P);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
int unpatchify_launch(const float* tok, float* out, int N, int OC, int H, int W, hipStream_t s) {
  const long long total = (long long)N * OC * H * W;
  hipLaunchKernelGGL(unpatchify_kernel, dim3((int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, s, tok, out, N, OC, H, W);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
int timestep_sincos_launch(const int64_t* t, const float* freqs, float* emb, int N, int half, hipStream_t s) {
  hipLaunchKernelGGL(timestep_sincos_kernel, dim3(cdiv(N * half, 256)), dim3(256), 0, s, t, freqs, emb, N, half);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
int cond_finish_launch(const float* c, const float* table, const int32_t* y, float* cs, int N, int D, hipStream_t s) {
  hipLaunchKernelGGL(cond_finish_kernel, dim3(cdiv(N * D, 256)), dim3(256), 0, s, c, table, y, cs, N, D);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
int fill_cls_launch(const float* cls, float* x, int N, int T, int D, hipStream_t s) {
  hipLaunchKernelGGL(fill_cls_kernel, dim3(cdiv(N * D, 256)), dim3(256), 0, s, cls, x, N, T, D);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
int pool_rows_launch(const float* x, float* out, int N, int T, int D, int first, int groups, int per, hipStream_t s) {
  hipLaunchKernelGGL(pool_rows_kernel, dim3(cdiv(N * groups * D, 256)), dim3(256), 0, s, x, out, N, T, D, first, groups, per);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

}  // namespace rgm
