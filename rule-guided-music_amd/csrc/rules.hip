// rules.hip -- built-in rule programs on the decoded piano roll and their losses (HBM-bound byte/count work).
//
// Reference: music_rule_guidance/music_rules.py:23-26 (piano_like), :29-43 (total_pitch_class_histogram),
// :46-83 (note_density), :86-94 (note_density_class); music_rule_guidance/rule_maps.py:17-22 (losses).
// Rolls are (N, C, 128, T) float32 in [-1, 1]; only channel 0 (notes) is read.
//
// Like the reference, the kernels WRITE into the caller's roll: rows outside the piano range [21,108]
// become -1 (piano_like works on a view) and note_density additionally snaps values < -0.95 to -1 before
// binarising.  Later rules of the same SCG step observe those writes, so they are reproduced, not skipped.
// note_density is integer counting behind hard thresholds -> bit-exact; pitch_hist is a float sum.
#include "common.h"

namespace rgm {
constexpr int MIN_PIANO = 21, MAX_PIANO = 108;

// grid (128 pitches, N): sum over time of (x+1)/2 for one pitch row; out-of-range rows are overwritten with -1
__global__ __launch_bounds__(256) void pitch_rowsum_kernel(float* __restrict__ roll, float* __restrict__ rowsum, int C, int T) {
  const int p = blockIdx.x, n = blockIdx.y;
  float* row = roll + ((long long)n * C * 128 + p) * T;
  const bool valid = p >= MIN_PIANO && p <= MAX_PIANO;
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    if (valid) s += (row[t] + 1.0f) / 2.0f;
    else row[t] = -1.0f;
  }
  __shared__ float sh[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) rowsum[n * 128 + p] = valid ? (sh[0] + sh[1]) + (sh[2] + sh[3]) : 0.f;
}

// one thread per sample: fold 128 pitches (+4 zero pad) into 12 classes, normalise by (sum + 1e-12)
__global__ void pitch_fold_kernel(const float* __restrict__ rowsum, float* __restrict__ out, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float h[12];
  float tot = 0.f;
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    float s = 0.f;
    for (int o = 0; o < 11; ++o) {
      const int p = o * 12 + c;
      if (p < 128) s += rowsum[n * 128 + p];
    }
    h[c] = s;
    tot += s;
  }
  const float d = tot + 1e-12f;
#pragma unroll
  for (int c = 0; c < 12; ++c) out[n * 12 + c] = h[c] / d;
}

// grid (ceil(T/256), N), block 256 = 256 consecutive time columns of one sample.
// Per column: #active pitches and whether any pitch has an onset (active now, inactive in the previous column).
// out[n][w] = mean count over window w ; out[n][nwin + w] = (#onset columns in window w) / hscale
__global__ __launch_bounds__(256) void note_density_kernel(float* __restrict__ roll, float* __restrict__ out, int C, int T,
                                                           int interval, float hscale) {
  __shared__ int vcnt[256], hcnt[256];
  const int n = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  float* base = roll + (long long)n * C * 128 * T;
  int v = 0, on = 0;
  if (t < T) {
    for (int p = 0; p < 128; ++p) {
      float* row = base + (long long)p * T;
      if (p < MIN_PIANO || p > MAX_PIANO) {
        row[t] = -1.0f;
        continue;
      }
      const float x = row[t];
      const bool act = !(x < -0.95f);                    // x<-0.95 -> -1 -> 0 ; everything else >= 0.025 -> 1
      const bool prev = t > 0 ? !(row[t - 1] < -0.95f) : false;   // zero pad on the left; value is threshold-stable
      if (!act) row[t] = -1.0f;
      v += act ? 1 : 0;
      on |= (act && !prev) ? 1 : 0;
    }
  }
  vcnt[threadIdx.x] = v;
  hcnt[threadIdx.x] = on;
  __syncthreads();
  const int wpb = 256 / interval;                        // windows per block (interval divides 256)
  if (threadIdx.x < wpb) {
    const int w = blockIdx.x * wpb + threadIdx.x;
    const int nwin = T / interval;
    if (w < nwin) {
      int sv = 0, shh = 0;
      for (int k = 0; k < interval; ++k) {
        sv += vcnt[threadIdx.x * interval + k];
        shh += hcnt[threadIdx.x * interval + k];
      }
      out[(long long)n * 2 * nwin + w] = (float)sv / (float)interval;
      out[(long long)n * 2 * nwin + nwin + w] = (float)shh / hscale;
    }
  }
}

// torch.bucketize(v, bounds) (right=False): first index i with bounds[i] >= v ; out int64
__global__ void bucketize_kernel(const float* __restrict__ v, const float* __restrict__ bounds, int nb, int64_t* __restrict__ out,
                                 int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = v[i];
  int k = 0;
  while (k < nb && bounds[k] < x) ++k;
  out[i] = k;
}

// out[r] = mean_k (a[r][k] - b[r][k])^2    |    out[r] = mean_k [a != b]
__global__ void row_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int rows, int K,
                                int zero_one) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    const float x = a[(long long)r * K + k], y = b[(long long)r * K + k];
    if (zero_one) s += (x != y) ? 1.f : 0.f;
    else { const float d = x - y; s += d * d; }
  }
  out[r] = s / (float)K;
}
// DPS through the differentiable pitch histogram (condition_functions.py:122-126 rule_x0_mse_dummy on music_rules.py:29-43):
// one thread per sample: hist as pitch_fold_kernel, logp = -scale * sum (hist - target)^2 and
// dh[c] = d logp / d (unnormalised class sum c) = (u_c - sum_k u_k hist_k) / (tot + 1e-12), u = -2 scale (hist - target)
__global__ void pitch_fold_vag_kernel(const float* __restrict__ rowsum, const float* __restrict__ target, float scale,
                                      float* __restrict__ hist_out, float* __restrict__ logp, float* __restrict__ dh, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float h[12];
  float tot = 0.f;
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    float s = 0.f;
    for (int o = 0; o < 11; ++o) {
      const int p = o * 12 + c;
      if (p < 128) s += rowsum[n * 128 + p];
    }
    h[c] = s;
    tot += s;
  }
  const float d = tot + 1e-12f;
  float lp = 0.f, dot = 0.f, u[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const float hc = h[c] / d;
    const float e = hc - target[n * 12 + c];
    lp += e * e;
    u[c] = -2.0f * scale * e;
    dot += u[c] * hc;
    h[c] = hc;
  }
  logp[n] = -scale * lp;
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    if (hist_out) hist_out[n * 12 + c] = h[c];
    dh[n * 12 + c] = (u[c] - dot) / d;
  }
}

// d_roll (N,C,128,T): channel 0, piano rows get 0.5 * dh[pitch % 12] (the (x+1)/2 rescale), everything else 0
__global__ void pitch_grad_fill_kernel(const float* __restrict__ dh, float* __restrict__ droll, long long total, int C, int T) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long r = i / T;
  const int p = (int)(r % 128);
  const long long nc = r / 128;
  const int ch = (int)(nc % C);
  const long long n = nc / C;
  droll[i] = (ch == 0 && p >= MIN_PIANO && p <= MAX_PIANO) ? 0.5f * dh[n * 12 + p % 12] : 0.0f;
}

// get_chords' device-side preamble (music_rules.py:97-110): piano_like mask and the < -0.95 background snap WRITTEN into channel 0
// of the roll, then (x + 1) / 2 * 127, clamp to [0, 127], truncate -> the integer piano roll the host chord analyser reads
__global__ void chord_quantise_kernel(float* __restrict__ roll, uint8_t* __restrict__ out, long long total, int C, int T) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;   // over N*128*T
  if (i >= total) return;
  const int t = (int)(i % T);
  const int p = (int)((i / T) % 128);
  const long long n = i / ((long long)T * 128);
  float* px = roll + ((n * C) * 128 + p) * T + t;
  float x = *px;
  if (p < MIN_PIANO || p > MAX_PIANO || x < -0.95f) {
    x = -1.0f;
    *px = x;
  }
  float v = (x + 1.0f) / 2.0f * 127.0f;
  v = fminf(fmaxf(v, 0.0f), 127.0f);
  out[i] = (uint8_t)(int)v;
}

}  // namespace rgm

using namespace rgm;

extern "C" int rgm_rule_pitch_hist(float* roll, float* out, float* scratch, int N, int C, int T, void* stream) {
  RGM_REQUIRE(roll && out && scratch && N > 0 && C > 0 && T > 0, "pitch_hist: bad arguments (scratch: N*128 floats)");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(pitch_rowsum_kernel, dim3(128, N), dim3(256), 0, s, roll, scratch, C, T);
  RGM_LAUNCH_CHECK();
  hipLaunchKernelGGL(pitch_fold_kernel, dim3(cdiv(N, 64)), dim3(64), 0, s, scratch, out, N);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// log p = -scale * ||pitch_hist(roll) - target||^2 per sample and its gradient w.r.t. the roll (N,C,128,T).  Like
// rgm_rule_pitch_hist the non-piano rows of channel 0 of `roll` are overwritten with -1.  scratch: N*(128+12) floats.
extern "C" int rgm_rule_pitch_hist_vag(float* roll, const float* target, float scale, float* hist, float* logp, float* d_roll,
                                       float* scratch, int N, int C, int T, void* stream) {
  RGM_REQUIRE(roll && target && logp && d_roll && scratch && N > 0 && C > 0 && T > 0, "pitch_hist_vag: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  float* dh = scratch + (size_t)N * 128;
  hipLaunchKernelGGL(pitch_rowsum_kernel, dim3(128, N), dim3(256), 0, s, roll, scratch, C, T);
  RGM_LAUNCH_CHECK();
  hipLaunchKernelGGL(pitch_fold_vag_kernel, dim3(cdiv(N, 64)), dim3(64), 0, s, scratch, target, scale, hist, logp, dh, N);
  RGM_LAUNCH_CHECK();
  const long long total = (long long)N * C * 128 * T;
  hipLaunchKernelGGL(pitch_grad_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dh, d_roll, total, C, T);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_rule_note_density(float* roll, float* out, int N, int C, int T, int interval, float hscale, void* stream) {
  RGM_REQUIRE(roll && out && N > 0 && C > 0 && T > 0, "note_density: bad arguments");
  RGM_REQUIRE(interval > 0 && 256 % interval == 0 && T % interval == 0, "note_density: interval %d must divide 256 and T=%d", interval, T);
  hipLaunchKernelGGL(note_density_kernel, dim3(cdiv(T, 256), N), dim3(256), 0, (hipStream_t)stream, roll, out, C, T, interval, hscale);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_rule_chord_quantise(float* roll, uint8_t* out, int N, int C, int T, void* stream) {
  RGM_REQUIRE(roll && out && N > 0 && C > 0 && T > 0, "chord_quantise: bad arguments");
  const long long total = (long long)N * 128 * T;
  hipLaunchKernelGGL(chord_quantise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, roll, out, total, C, T);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_bucketize(const float* v, const float* bounds, int nb, int64_t* out, int n, void* stream) {
  RGM_REQUIRE(v && bounds && out && n > 0 && nb > 0, "bucketize: bad arguments");
  hipLaunchKernelGGL(bucketize_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, v, bounds, nb, out, n);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_row_loss(const float* a, const float* b, float* out, int rows, int K, int zero_one, void* stream) {
  RGM_REQUIRE(a && b && out && rows > 0 && K > 0, "row_loss: bad arguments");
  hipLaunchKernelGGL(row_loss_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, rows, K, zero_one);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
