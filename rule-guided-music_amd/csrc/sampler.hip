// sampler.hip -- the per-step elementwise work of the DDPM / DDIM / SCG sampler, fused.
//
// Reference: guided_diffusion/gaussian_diffusion.py
//   :359-364 _predict_xstart_from_eps, :380-385 _predict_eps_from_xstart, :228-250 q_posterior_mean_variance
//   :252-357 p_mean_variance (EPSILON mean, FIXED_LARGE variance), :387-407 condition_mean,
//   :467-489 condition_score, :635-735 p_sample, :881-976 ddim_sample, :491-554 scg_sample
//   (candidate expansion :509-514, argmax-select :539-554), :1331-1344 _extract_into_tensor.
//
// The reference issues ~12 tiny ATen launches and ~10 host->device table uploads per step; here a
// step is ONE launch over the latent (32 KiB per sample -- pure launch-latency territory), with the
// float32-cast schedule tables resident on the device and indexed by the per-sample timestep.
// Noise comes from a counter-based Philox4x32-10 stream (seed, offset): every rank of a multi-GPU
// SCG run can regenerate any candidate's noise without communication.
#include "common.h"

namespace rgm {

// ------------------------------------------------------------------------------------- philox
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1;
  c[3] = (uint32_t)p0;
  c[0] = n0;
  c[2] = n2;
}

__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t seed, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}

// the 4 normals of Philox block `ctr`: Box-Muller on (0,1] x [0,1)
__device__ __forceinline__ void philox_normals(uint64_t ctr, uint64_t seed, float (&z)[4]) {
  uint32_t r[4];
  philox4x32_10(ctr, seed, r);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)(r[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(r[2 * h + 1] >> 8) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    z[2 * h] = rad * cs;
    z[2 * h + 1] = rad * sn;
  }
}

// out[i] ~ N(0,1); element i is a pure function of (seed, offset + i): counter = (offset+i)/4, lane (offset+i)%4
__global__ void randn_kernel(float* __restrict__ out, long long n, uint64_t seed, uint64_t offset) {
  const long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x;  // one Philox block = 4 normals
  const uint64_t first = offset >> 2;
  const long long nblk = (long long)(((offset + (uint64_t)n + 3) >> 2) - first);
  if (q >= nblk) return;
  float z[4];
  philox_normals(first + (uint64_t)q, seed, z);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long g = (long long)((first + (uint64_t)q) * 4 + j) - (long long)offset;
    if (g >= 0 && g < n) out[g] = z[j];
  }
}

// ------------------------------------------------------------------------------------- steps
struct StepTables {          // device float32 copies of the float64 schedule tables (len T')
  const float* sqrt_recip_ac;
  const float* sqrt_recipm1_ac;
  const float* post_c1;
  const float* post_c2;
  const float* var;           // FIXED_LARGE variance
  const float* logvar;
  const float* ac;
  const float* ac_prev;
};

// DDPM ancestral step.  One thread per element; E = elements per sample.
//   x0 = c1*x - c2*eps (clip) ; mean = pc1*x0 + pc2*x (+ var*grad) ; sample = mean + [t>t_end]*exp(.5 logvar)*noise
// noise == nullptr -> sample = mean (SCG: candidates are drawn by scg_candidates instead).
// Learned variances (ModelVarType.LEARNED_RANGE / LEARNED, :299-313): `vv` holds the network's second output half per
// element; with min_log / max_log tables log var = frac * max_log + (1 - frac) * min_log, frac = (vv + 1) / 2 (LEARNED_RANGE);
// with the tables NULL vv IS the log-variance (LEARNED).  vv == NULL: the fixed tables.
__global__ void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ grad,
                                 const float* __restrict__ noise, const int64_t* __restrict__ t, StepTables tb, int clip,
                                 int t_end, float* __restrict__ sample, float* __restrict__ pred_xstart,
                                 float* __restrict__ g_out, long long total, int E, const float* __restrict__ vv = nullptr,
                                 const float* __restrict__ min_log = nullptr, const float* __restrict__ max_log = nullptr,
                                 float* __restrict__ g_elem = nullptr) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / E);
  const int ti = (int)t[b];
  const float xv = x[i];
  float x0 = tb.sqrt_recip_ac[ti] * xv - tb.sqrt_recipm1_ac[ti] * eps[i];
  if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
  float mean = tb.post_c1[ti] * x0 + tb.post_c2[ti] * xv;
  float logvar = tb.logvar[ti], var = tb.var[ti];
  if (vv) {
    if (min_log) {
      const float frac = (vv[i] + 1.f) / 2.f;
      logvar = frac * max_log[ti] + (1.f - frac) * min_log[ti];
    } else {
      logvar = vv[i];
    }
    var = expf(logvar);
  }
  if (grad) mean = mean + var * grad[i];
  const float g = expf(0.5f * logvar);
  float s = mean;
  if (noise) s = mean + (ti > t_end ? 1.f : 0.f) * g * noise[i];
  sample[i] = s;
  pred_xstart[i] = x0;
  if (g_out && (i % E) == 0) g_out[b] = g;
  if (g_elem) g_elem[i] = g;          // learned variances: the noise scale is a tensor (SCG candidates: exp(0.5 log_variance), :708)
}

// DDIM step (eta general; the CLI always runs eta = 1).  grad != nullptr applies condition_score first.
__global__ void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ eps_in, const float* __restrict__ grad,
                                 const float* __restrict__ noise, const int64_t* __restrict__ t, StepTables tb, int clip,
                                 int t_end, float eta, float* __restrict__ sample, float* __restrict__ pred_xstart,
                                 float* __restrict__ g_out, long long total, int E) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / E);
  const int ti = (int)t[b];
  const float xv = x[i];
  const float c1 = tb.sqrt_recip_ac[ti], c2 = tb.sqrt_recipm1_ac[ti];
  const float ab = tb.ac[ti], abp = tb.ac_prev[ti];
  float x0 = c1 * xv - c2 * eps_in[i];
  if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
  if (grad) {
    float e = (c1 * xv - x0) / c2;
    e = e - sqrtf(1.f - ab) * grad[i];
    x0 = c1 * xv - c2 * e;
  }
  const float e2 = (c1 * xv - x0) / c2;
  const float sigma = eta * sqrtf((1.f - abp) / (1.f - ab)) * sqrtf(1.f - ab / abp);
  const float mean = x0 * sqrtf(abp) + sqrtf(1.f - abp - sigma * sigma) * e2;
  float s = mean;
  if (noise) s = mean + (ti != t_end ? 1.f : 0.f) * sigma * noise[i];
  sample[i] = s;
  pred_xstart[i] = x0;
  if (g_out && (i % E) == 0) g_out[b] = sigma;
}

// cand[k][b][:] = mean[b][:] + g[b] * noise[k][b][:]   (gaussian_diffusion.py:509-514); k in [0, n_local)
__global__ void scg_candidates_kernel(const float* __restrict__ mean, const float* __restrict__ g,
                                      const float* __restrict__ noise, float* __restrict__ cand, int n, int B, int E) {
  const long long total = (long long)n * B * E;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long be = i % ((long long)B * E);
  cand[i] = mean[be] + g[be / E] * noise[i];
}

// x0 = c1[t]*x - c2[t]*eps  (optionally * inv_scale: the 1/scale_factor of _decode :1350)
__global__ void xstart_from_eps_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                       const int64_t* __restrict__ t, StepTables tb, float out_scale,
                                       float* __restrict__ out, long long total, int E) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ti = (int)t[i / E];
  out[i] = (tb.sqrt_recip_ac[ti] * x[i] - tb.sqrt_recipm1_ac[ti] * eps[i]) * out_scale;
}

// Replacement-based conditioning of scripts/edit.py (p_mean_variance :293-298): the x0 estimate is overwritten by the
// ground-truth latent where mask == 1 and the eps estimate recomputed from it:
//   x0 = clip(c1*x - c2*eps);  x0r = mask*gt + (1-mask)*x0;  eps' = (c1*x - x0r) / c2
__global__ void edit_replace_eps_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ gt,
                                        const float* __restrict__ mask, const int64_t* __restrict__ t, StepTables tb, int clip,
                                        float* __restrict__ out, long long total, int E) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ti = (int)t[i / E];
  const float c1 = tb.sqrt_recip_ac[ti], c2 = tb.sqrt_recipm1_ac[ti];
  float x0 = c1 * x[i] - c2 * eps[i];
  if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
  const float m = mask[i];
  const float x0r = m * gt[i] + (1.f - m) * x0;
  out[i] = (c1 * x[i] - x0r) / c2;
}

// max_ind[b] = first argmax_k total[k][b] ; out[b][:] = cand[max_ind[b]][b][:]   (:539-554)
__global__ void scg_select_kernel(const float* __restrict__ cand, const float* __restrict__ total, float* __restrict__ out,
                                  int64_t* __restrict__ max_ind, int n, int B, int E) {
  const int b = blockIdx.y;
  int best = 0;
  float bv = total[b];
  for (int k = 1; k < n; ++k) {
    const float v = total[(long long)k * B + b];
    if (v > bv) { bv = v; best = k; }   // strict '>' keeps the FIRST maximum (torch argmax tie-break)
  }
  // NaN handling: torch.argmax treats NaN as maximal; reproduce (first NaN wins)
  for (int k = 0; k < n; ++k)
    if (total[(long long)k * B + b] != total[(long long)k * B + b]) { best = k; break; }
  if (cand && out) {
    const float* src = cand + ((long long)best * B + b) * E;
    float* dst = out + (long long)b * E;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x) dst[i] = src[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && max_ind) max_ind[b] = best;
}

// Winner regeneration of the sharded SCG step: the (n, B) score table is identical on every rank after the all-gather, the
// winning candidate may have been scored on another rank, and candidate k of sample b is mean[b] + g[b] * z with z the
// Philox normals at stream positions base + (k*B + b)*E + e -- so every rank rebuilds the winners itself, on the device,
// from the argmax indices (no latent traffic, no host read-back of max_ind).  Segment-wise selection (dc.base > 0,
// gaussian_diffusion.py:562-592): row h of the (C, H, W) latent belongs to segment h / seg_rows and takes that segment's
// winner, max_ind is (S, B); seg_rows >= H is the plain case (S = 1).
__global__ void scg_rebuild_kernel(const float* __restrict__ mean, const float* __restrict__ g, const int64_t* __restrict__ max_ind,
                                   uint64_t seed, uint64_t base, float* __restrict__ out, int B, int E, int H, int W, int seg_rows,
                                   int g_per_elem) {
  const long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x;   // 4 consecutive elements (E % 4 == 0, W % 4 == 0)
  const long long total4 = (long long)B * E / 4;
  if (q >= total4) return;
  const long long i = q * 4;
  const int b = (int)(i / E), e = (int)(i - (long long)b * E);
  const int seg = ((e / W) % H) / seg_rows;
  const int64_t k = max_ind[(long long)seg * B + b];
  const uint64_t pos = base + ((uint64_t)k * B + b) * (uint64_t)E + (uint64_t)e;
  float z0[4], z1[4];
  philox_normals(pos >> 2, seed, z0);
  const int lane = (int)(pos & 3);
  if (lane) philox_normals((pos >> 2) + 1, seed, z1);
  const float gb = g_per_elem ? 0.f : g[b];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int l = lane + j;
    const float z = l < 4 ? z0[l & 3] : z1[l & 3];
    out[i + j] = mean[i + j] + (g_per_elem ? g[i + j] : gb) * z;
  }
}

}  // namespace rgm

using namespace rgm;

static StepTables make_tables(const float* const* tabs) {
  StepTables t;
  t.sqrt_recip_ac = tabs[0];
  t.sqrt_recipm1_ac = tabs[1];
  t.post_c1 = tabs[2];
  t.post_c2 = tabs[3];
  t.var = tabs[4];
  t.logvar = tabs[5];
  t.ac = tabs[6];
  t.ac_prev = tabs[7];
  return t;
}

extern "C" int rgm_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  RGM_REQUIRE(out && n >= 0, "randn: bad arguments");
  if (n == 0) return RGM_OK;
  const long long nblk = (long long)(((offset + (uint64_t)n + 3) >> 2) - (offset >> 2));
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, (long long)n, seed, offset);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_ddpm_step(const float* x, const float* eps, const float* grad, const float* noise, const int64_t* t,
                             const float* const* tables, int clip_denoised, int t_end, float* sample, float* pred_xstart,
                             float* g_out, int N, int E, void* stream) {
  RGM_REQUIRE(x && eps && t && tables && sample && pred_xstart && N > 0 && E > 0, "ddpm_step: bad arguments");
  const long long total = (long long)N * E;
  hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, eps, grad,
                     noise, t, make_tables(tables), clip_denoised, t_end, sample, pred_xstart, g_out, total, E, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_ddpm_step_learned(const float* x, const float* eps, const float* var_values, const float* min_log_tab,
                                     const float* max_log_tab, const float* grad, const float* noise, const int64_t* t,
                                     const float* const* tables, int clip_denoised, int t_end, float* sample, float* pred_xstart, int N,
                                     int E, void* stream) {
  RGM_REQUIRE(x && eps && var_values && t && tables && sample && pred_xstart && N > 0 && E > 0, "ddpm_step_learned: bad arguments");
  RGM_REQUIRE((min_log_tab == nullptr) == (max_log_tab == nullptr), "ddpm_step_learned: min / max log-variance tables come together");
  const long long total = (long long)N * E;
  hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, eps, grad, noise, t,
                     make_tables(tables), clip_denoised, t_end, sample, pred_xstart, (float*)nullptr, total, E, var_values, min_log_tab,
                     max_log_tab, (float*)nullptr);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// The same step, also writing the per-element noise scale g = exp(0.5 log_variance) (N, E): what p_sample hands scg_sample as
// g_coeff at a learned-variance step (gaussian_diffusion.py:706-711).
extern "C" int rgm_ddpm_step_learned_g(const float* x, const float* eps, const float* var_values, const float* min_log_tab,
                                       const float* max_log_tab, const float* grad, const float* noise, const int64_t* t,
                                       const float* const* tables, int clip_denoised, int t_end, float* sample, float* pred_xstart,
                                       float* g_elem, int N, int E, void* stream) {
  RGM_REQUIRE(x && eps && var_values && t && tables && sample && pred_xstart && g_elem && N > 0 && E > 0, "ddpm_step_learned_g: bad arguments");
  RGM_REQUIRE((min_log_tab == nullptr) == (max_log_tab == nullptr), "ddpm_step_learned_g: min / max log-variance tables come together");
  const long long total = (long long)N * E;
  hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, eps, grad, noise, t,
                     make_tables(tables), clip_denoised, t_end, sample, pred_xstart, (float*)nullptr, total, E, var_values, min_log_tab,
                     max_log_tab, g_elem);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_ddim_step(const float* x, const float* eps, const float* grad, const float* noise, const int64_t* t,
                             const float* const* tables, int clip_denoised, int t_end, float eta, float* sample,
                             float* pred_xstart, float* g_out, int N, int E, void* stream) {
  RGM_REQUIRE(x && eps && t && tables && sample && pred_xstart && N > 0 && E > 0, "ddim_step: bad arguments");
  const long long total = (long long)N * E;
  hipLaunchKernelGGL(ddim_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, eps, grad,
                     noise, t, make_tables(tables), clip_denoised, t_end, eta, sample, pred_xstart, g_out, total, E);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_scg_candidates(const float* mean, const float* g, const float* noise, float* cand, int n, int B, int E,
                                  void* stream) {
  RGM_REQUIRE(mean && g && noise && cand && n > 0 && B > 0 && E > 0, "scg_candidates: bad arguments");
  const long long total = (long long)n * B * E;
  hipLaunchKernelGGL(scg_candidates_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean, g, noise, cand, n, B, E);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_xstart_from_eps(const float* x, const float* eps, const int64_t* t, const float* const* tables,
                                   float out_scale, float* out, int N, int E, void* stream) {
  RGM_REQUIRE(x && eps && t && tables && out && N > 0 && E > 0, "xstart_from_eps: bad arguments");
  const long long total = (long long)N * E;
  hipLaunchKernelGGL(xstart_from_eps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, eps, t,
                     make_tables(tables), out_scale, out, total, E);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_edit_replace_eps(const float* x, const float* eps, const float* gt, const float* mask, const int64_t* t,
                                    const float* const* tables, int clip_denoised, float* eps_out, int N, int E, void* stream) {
  RGM_REQUIRE(x && eps && gt && mask && t && tables && eps_out && N > 0 && E > 0, "edit_replace_eps: bad arguments");
  const long long total = (long long)N * E;
  hipLaunchKernelGGL(edit_replace_eps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, eps, gt, mask,
                     t, make_tables(tables), clip_denoised, eps_out, total, E);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_scg_select(const float* cand, const float* total_logp, float* out, int64_t* max_ind, int n, int B, int E,
                              void* stream) {
  RGM_REQUIRE(total_logp && n > 0 && B > 0 && E > 0 && ((cand == nullptr) == (out == nullptr)), "scg_select: bad arguments");
  hipLaunchKernelGGL(scg_select_kernel, dim3(cdiv(E, 1024) > 64 ? 64 : cdiv(E, 1024), B), dim3(256), 0, (hipStream_t)stream, cand,
                     total_logp, out, max_ind, n, B, E);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_scg_rebuild(const float* mean, const float* g, const int64_t* max_ind, uint64_t seed, uint64_t base, float* out,
                               int B, int E, int H, int W, int seg_rows, void* stream) {
  RGM_REQUIRE(mean && g && max_ind && out && B > 0 && E > 0 && H > 0 && W > 0 && seg_rows > 0, "scg_rebuild: bad arguments");
  RGM_REQUIRE(E % (H * W) == 0 && W % 4 == 0, "scg_rebuild: E = %d is not C x H x W with H = %d, W = %d (W %% 4 == 0)", E, H, W);
  const long long total4 = (long long)B * E / 4;
  hipLaunchKernelGGL(scg_rebuild_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean, g, max_ind,
                     seed, base, out, B, E, H, W, seg_rows, 0);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// rgm_scg_rebuild with a per-element noise scale g (B, E) -- learned variances.
extern "C" int rgm_scg_rebuild_g(const float* mean, const float* g_elem, const int64_t* max_ind, uint64_t seed, uint64_t base, float* out,
                                 int B, int E, int H, int W, int seg_rows, void* stream) {
  RGM_REQUIRE(mean && g_elem && max_ind && out && B > 0 && E > 0 && H > 0 && W > 0 && seg_rows > 0, "scg_rebuild_g: bad arguments");
  RGM_REQUIRE(E % (H * W) == 0 && W % 4 == 0, "scg_rebuild_g: E = %d is not C x H x W with H = %d, W = %d (W %% 4 == 0)", E, H, W);
  const long long total4 = (long long)B * E / 4;
  hipLaunchKernelGGL(scg_rebuild_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean, g_elem, max_ind,
                     seed, base, out, B, E, H, W, seg_rows, 1);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
