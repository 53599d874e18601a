// vae.hip -- taming KL-VAE (f8, z=4ch): weight arena, glue kernels, the decode schedule, the encoder (editing) and the
// decoder's input-gradient pass (DPS through a rule on the decoded roll).
//
// Reference: taming/models/klvae_pedal.py:80-85 (AutoencoderKL.decode = post_quant_conv -> Decoder),
// taming/modules/diffusionmodules/model.py:436-537 (Decoder), :78-137 (ResnetBlock), :140-192 (AttnBlock),
// :38-53 (Upsample), :29-35 (swish, GroupNorm(32, eps 1e-6)); tile gather / re-assembly of
// guided_diffusion/gaussian_diffusion.py:1347-1358 (_decode); uint8 quantisation of
// guided_diffusion/midi_util.py:59-63.  Config: ch 128, ch_mult (1,2,2,4), 2 res blocks, no up-attention.
//
// Layout: activations are NHWC ([tile][y][x][c], c contiguous) so that a 3x3 conv is an implicit GEMM
// whose K axis (tap, cin) is contiguous for both operands (gemm.hip, aload=1); conv weights are repacked
// once at set_param to [cout][tap][cin].  ~99.9 % of the 114.5 GFLOP per tile runs in that GEMM kernel;
// nearest-x2 upsampling is folded into the consumer conv's gather (the 4x larger tensor is never
// materialised), residual adds ride the GEMM epilogue.  GroupNorm statistics are accumulated in fp64
// (deterministic two-level reduction), then normalise+affine+swish is one float4 pass.
// Latent squares are gathered straight from the (N,4,H,16) latent (incl. 1/scale_factor) and the last conv
// scatters straight into the (N,3,128,8H) roll, so _decode's permute/chunk/concat never touch HBM.
#include <map>
#include <string>
#include <vector>
#include "common.h"

namespace rgm {

// ---------------------------------------------------------------------------------- glue kernels
// pq[m][i][j][co] = b[co] + sum_ci w[co][ci] * in[off(m) + ci*sc + i*si + j*sj] * in_scale   (post_quant_conv 1x1)
__global__ void vae_gather_pq_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b,
                                     float* __restrict__ pq, int M, int Nb, long long n_stride, long long s_stride,
                                     long long sc, long long si, long long sj, float in_scale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over M*256 pixels
  if (idx >= M * 256) return;
  const int m = idx >> 8, i = (idx >> 4) & 15, j = idx & 15;
  const int s = m / Nb, n = m - s * Nb;
  const float* p = in + n * n_stride + s * s_stride + i * si + j * sj;
  float v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = p[c * sc] * in_scale;
  float4 o;
  float* op = reinterpret_cast<float*>(&o);
#pragma unroll
  for (int co = 0; co < 4; ++co) op[co] = b[co] + w[co * 4 + 0] * v[0] + w[co * 4 + 1] * v[1] + w[co * 4 + 2] * v[2] + w[co * 4 + 3] * v[3];
  reinterpret_cast<float4*>(pq)[idx] = o;
}

// conv_in 3x3 (4 -> Cout) on the 16x16 squares; weights repacked [Cout][9][4]; out NHWC
__global__ void vae_conv_in_kernel(const float* __restrict__ pq, const float* __restrict__ w, const float* __restrict__ b,
                                   float* __restrict__ out, int M, int Cout) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;  // over M*256*Cout, co fastest
  if (idx >= (long long)M * 256 * Cout) return;
  const int co = (int)(idx % Cout);
  const int pix = (int)(idx / Cout);
  const int m = pix >> 8, y = (pix >> 4) & 15, x = pix & 15;
  float acc = b[co];
  const float4* wq = reinterpret_cast<const float4*>(w + (long long)co * 36);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    if ((unsigned)yy < 16u && (unsigned)xx < 16u) {
      const float4 a = reinterpret_cast<const float4*>(pq)[(m << 8) + (yy << 4) + xx];
      const float4 ww = wq[tap];
      acc += (a.x * ww.x + a.y * ww.y) + (a.z * ww.z + a.w * ww.w);
    }
  }
  out[idx] = acc;
}

// Encoder conv_in 3x3 (3 -> Cout, pad 1) on 128x128 piano-roll tiles (taming Encoder.conv_in, model.py:357-361):
// x NCHW (M,3,128,128), weights in their checkpoint layout [Cout][3][3][3], out NHWC [M][128][128][Cout]
__global__ void vae_enc_conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                       float* __restrict__ out, int M, int Cout) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;  // over M*16384*Cout, co fastest
  if (idx >= (long long)M * 16384 * Cout) return;
  const int co = (int)(idx % Cout);
  const long long pix = idx / Cout;
  const int m = (int)(pix >> 14), y = (int)((pix >> 7) & 127), xx0 = (int)(pix & 127);
  float acc = b[co];
  const float* wc = w + (long long)co * 27;
#pragma unroll
  for (int ci = 0; ci < 3; ++ci) {
    const float* plane = x + ((long long)m * 3 + ci) * 16384;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int yy = y + tap / 3 - 1, xx = xx0 + tap % 3 - 1;
      if ((unsigned)yy < 128u && (unsigned)xx < 128u) acc = fmaf(plane[(yy << 7) + xx], wc[ci * 9 + tap], acc);
    }
  }
  out[idx] = acc;
}

// quant_conv 1x1 (8 -> 8) on the encoder output (klvae_pedal.py:61-63): h NHWC rows [M*256][8] -> moments NCHW (M,8,16,16)
__global__ void vae_enc_quant_kernel(const float* __restrict__ h8, const float* __restrict__ w, const float* __restrict__ b,
                                     float* __restrict__ out, int M) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over M*8*256, pixel fastest
  if (idx >= M * 8 * 256) return;
  const int pix = idx & 255, co = (idx >> 8) & 7, m = idx >> 11;
  const float* r = h8 + ((long long)m * 256 + pix) * 8;
  float acc = b[co];
#pragma unroll
  for (int ci = 0; ci < 8; ++ci) acc = fmaf(r[ci], w[co * 8 + ci], acc);
  out[idx] = acc;
}

// GroupNorm partial sums in fp64: grid (chunks, M); x NHWC [M][P][C]; part[m][chunk][32][2]
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part, int P, int C,
                                                         int chunks) {
  __shared__ double sh[32][2];
  const int m = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) sh[tid >> 1][tid & 1] = 0.0;
  __syncthreads();
  const int q = C >> 2;              // float4 per pixel
  const int ppi = 256 / q;           // pixels per block-iteration (q in {32, 64, 128})
  const int cq = tid % q, p_off = tid / q;
  const int p0 = (int)((long long)P * ch / chunks), p1 = (int)((long long)P * (ch + 1) / chunks);
  const float4* xb = reinterpret_cast<const float4*>(x + (long long)m * P * C);
  double s = 0.0, ss = 0.0;
  for (int p = p0 + p_off; p < p1; p += ppi) {
    const float4 v = xb[(long long)p * q + cq];
    s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
  }
  const int g = (cq * 4) / (C >> 5);
  atomicAdd(&sh[g][0], s);           // LDS fp64 atomics: order varies, but fp64 sums of <=2^18 fp32 squares
  atomicAdd(&sh[g][1], ss);          // differ only beyond float precision of the final mean / rstd
  __syncthreads();
  if (tid < 64) part[(((long long)m * chunks + ch) * 32 + (tid >> 1)) * 2 + (tid & 1)] = sh[tid >> 1][tid & 1];
}

__global__ void gn_finalize_kernel(const double* __restrict__ part, float* __restrict__ stats, int M, int chunks, double count,
                                   float eps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over M*32
  if (idx >= M * 32) return;
  const int m = idx >> 5, g = idx & 31;
  double s = 0.0, ss = 0.0;
  for (int c = 0; c < chunks; ++c) {
    s += part[(((long long)m * chunks + c) * 32 + g) * 2];
    ss += part[(((long long)m * chunks + c) * 32 + g) * 2 + 1];
  }
  const double mean = s / count;
  double var = ss / count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[idx * 2] = (float)mean;
  stats[idx * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// The same statistics from the per-tile partial sums the producing conv GEMM's epilogue left (gemm2.hip, GemmParams::stats):
// tp[(m * tiles + t) * 32 + g][2] fp64 sums over 128 pixels x (C/32) channels; summed over the image's tiles in a fixed order
__global__ void gn_finalize_tiles_kernel(const double* __restrict__ tp, float* __restrict__ stats, int M, int tiles, double count,
                                         float eps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over M*32
  if (idx >= M * 32) return;
  const int m = idx >> 5, g = idx & 31;
  double s = 0.0, ss = 0.0;
  for (int t = 0; t < tiles; ++t) {
    const double* r = tp + (((long long)m * tiles + t) * 32 + g) * 2;
    s += r[0];
    ss += r[1];
  }
  const double mean = s / count;
  double var = ss / count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[idx * 2] = (float)mean;
  stats[idx * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = act((x - mean) * rstd * gamma + beta), act = swish (1) or identity (0); NHWC float4 pass
__global__ void gn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, long long total4, int P, int C,
                                int swish, int out_split, float* __restrict__ raw_split) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int q = C >> 2;
  const int c4 = (int)(i % q);
  const int m = (int)(i / ((long long)q * P));
  const int g = (c4 * 4) / (C >> 5);
  const float mean = stats[(m * 32 + g) * 2], rstd = stats[(m * 32 + g) * 2 + 1];
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[c4], be = reinterpret_cast<const float4*>(beta)[c4];
  if (raw_split) {   // the input itself as split rows, in the same pass (the 1x1 shortcut of a channel-changing ResnetBlock reads it)
    typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, lo;
    hi[0] = (split_t)v.x; hi[1] = (split_t)v.y; hi[2] = (split_t)v.z; hi[3] = (split_t)v.w;
    lo[0] = (split_t)(v.x - (float)hi[0]); lo[1] = (split_t)(v.y - (float)hi[1]);
    lo[2] = (split_t)(v.z - (float)hi[2]); lo[3] = (split_t)(v.w - (float)hi[3]);
    split_t* px = reinterpret_cast<split_t*>(raw_split + (i / q) * C);
    *reinterpret_cast<bf16x4*>(px + split_idx(c4 * 4)) = hi;
    *reinterpret_cast<bf16x4*>(px + split_idx(c4 * 4) + 32) = lo;
  }
  float4 o;
  o.x = (v.x - mean) * rstd * ga.x + be.x;
  o.y = (v.y - mean) * rstd * ga.y + be.y;
  o.z = (v.z - mean) * rstd * ga.z + be.z;
  o.w = (v.w - mean) * rstd * ga.w + be.w;
  if (swish) {
    o.x = silu_fast_f(o.x); o.y = silu_fast_f(o.y); o.z = silu_fast_f(o.z); o.w = silu_fast_f(o.w);
  }
  if (out_split) {   // split-row pixels (common.h split_idx) for the pre-split conv GEMM
    typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, lo;
    hi[0] = (split_t)o.x; hi[1] = (split_t)o.y; hi[2] = (split_t)o.z; hi[3] = (split_t)o.w;
    lo[0] = (split_t)(o.x - (float)hi[0]); lo[1] = (split_t)(o.y - (float)hi[1]);
    lo[2] = (split_t)(o.z - (float)hi[2]); lo[3] = (split_t)(o.w - (float)hi[3]);
    split_t* px = reinterpret_cast<split_t*>(y + (i / q) * C);
    *reinterpret_cast<bf16x4*>(px + split_idx(c4 * 4)) = hi;
    *reinterpret_cast<bf16x4*>(px + split_idx(c4 * 4) + 32) = lo;
  } else {
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

// The same pass with EIGHT channels of a pixel per lane and two such units per lane, all four 16-byte loads issued before the first
// result: a split-row store is then 16 B of hi halves + 16 B of lo halves per lane (whole 32-B sectors instead of 8-B pieces) and twice
// the bytes are in flight per wave.  Same arithmetic per element as gn_apply_kernel (identical values); C % 8 == 0.
__global__ __launch_bounds__(256) void gn_apply8_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, long long total8,
                                                        long long half, int P, int C, int swish, int out_split, float* __restrict__ raw_split) {
  typedef split_t bf16x8 __attribute__((ext_vector_type(8)));
  const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i0 >= half) return;
  const int q = C >> 3;
  long long idx[2] = {i0, i0 + half};
  float4 v[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (idx[u] < total8) {
      v[u][0] = reinterpret_cast<const float4*>(x)[idx[u] * 2];
      v[u][1] = reinterpret_cast<const float4*>(x)[idx[u] * 2 + 1];
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long long i = idx[u];
    if (i >= total8) continue;
    const int c8 = (int)(i % q);
    const int m = (int)(i / ((long long)q * P));
    const int gw = C >> 5;                       // channels per group: 4, 8 or 16 -> the 8 channels of a lane meet one or two groups
    const int g0 = (c8 * 8) / gw, g1 = (c8 * 8 + 4) / gw;
    const float mean0 = stats[(m * 32 + g0) * 2], rstd0 = stats[(m * 32 + g0) * 2 + 1];
    const float mean1 = stats[(m * 32 + g1) * 2], rstd1 = stats[(m * 32 + g1) * 2 + 1];
    const float4 ga0 = reinterpret_cast<const float4*>(gamma)[c8 * 2], ga1 = reinterpret_cast<const float4*>(gamma)[c8 * 2 + 1];
    const float4 be0 = reinterpret_cast<const float4*>(beta)[c8 * 2], be1 = reinterpret_cast<const float4*>(beta)[c8 * 2 + 1];
    const float in[8] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w, v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
    const long long pix = i / q;
    const int si = split_idx(c8 * 8);
    if (raw_split) {
      bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        hi[e] = (split_t)in[e];
        lo[e] = (split_t)(in[e] - (float)hi[e]);
      }
      split_t* px = reinterpret_cast<split_t*>(raw_split + pix * C);
      *reinterpret_cast<bf16x8*>(px + si) = hi;
      *reinterpret_cast<bf16x8*>(px + si + 32) = lo;
    }
    const float gam[8] = {ga0.x, ga0.y, ga0.z, ga0.w, ga1.x, ga1.y, ga1.z, ga1.w};
    const float bet[8] = {be0.x, be0.y, be0.z, be0.w, be1.x, be1.y, be1.z, be1.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float mean = e < 4 ? mean0 : mean1, rstd = e < 4 ? rstd0 : rstd1;
      o[e] = (in[e] - mean) * rstd * gam[e] + bet[e];
      if (swish) o[e] = silu_fast_f(o[e]);
    }
    if (out_split) {
      bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        hi[e] = (split_t)o[e];
        lo[e] = (split_t)(o[e] - (float)hi[e]);
      }
      split_t* px = reinterpret_cast<split_t*>(y + pix * C);
      *reinterpret_cast<bf16x8*>(px + si) = hi;
      *reinterpret_cast<bf16x8*>(px + si + 32) = lo;
    } else {
      reinterpret_cast<float4*>(y)[i * 2] = make_float4(o[0], o[1], o[2], o[3]);
      reinterpret_cast<float4*>(y)[i * 2 + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
}

// Tiles of a conv launch with GemmParams::gn_count whose wait for the image's other tiles ran out wrote their RAW fp32 rows (and raised
// fail[tile]): this pass finalises the image's statistics from the per-tile partial sums (the arithmetic of gn_finalize_tiles_kernel)
// and converts such a tile to normalised split rows in place -- a 128-byte line holds the same 32 elements in both formats and its
// eight lanes read it before any of them writes.  One workgroup per tile; it leaves at once when the flag is down (the normal case).
__global__ __launch_bounds__(256) void gn_fixup_kernel(float* __restrict__ y, const int* __restrict__ fail, const double* __restrict__ tp,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int tiles_n,
                                                       int rows, int bn, int C, int tiles_per_img, double count, float eps, int swish,
                                                       unsigned long long* __restrict__ fallbacks) {
  const int tile = blockIdx.x;
  if (!fail[tile]) return;
  if (threadIdx.x == 0) atomicAdd(fallbacks, 1ull);      // rgm_gn_fallback_tiles: a tile whose wait ran out is otherwise only visible as lost time
  typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const int qpr = bn >> 2;                               // column quads per tile row: 32 or 64
  const int quad = threadIdx.x % qpr, r0 = threadIdx.x / qpr, rstep = 256 / qpr;
  const int col = tile_n * bn + quad * 4;
  const int gw = C >> 5, g = col / gw;
  const long long img_tile0 = (long long)(tile_m / tiles_per_img) * tiles_per_img;
  double s = 0., ss = 0.;
  for (int t = 0; t < tiles_per_img; ++t) {
    const double* r = tp + ((img_tile0 + t) * 32 + g) * 2;
    s += r[0];
    ss += r[1];
  }
  const double mean_d = s / count;
  double var = ss / count - mean_d * mean_d;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float4 ga = *reinterpret_cast<const float4*>(gamma + col), be = *reinterpret_cast<const float4*>(beta + col);
  for (int r = r0; r < rows; r += rstep) {
    float* rowp = y + ((long long)tile_m * rows + r) * C;
    const float4 v = *reinterpret_cast<const float4*>(rowp + col);
    float o[4] = {(v.x - mean) * rstd * ga.x + be.x, (v.y - mean) * rstd * ga.y + be.y, (v.z - mean) * rstd * ga.z + be.z,
                  (v.w - mean) * rstd * ga.w + be.w};
    if (swish) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = silu_fast_f(o[e]);
    }
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (split_t)o[e];
      lo[e] = (split_t)(o[e] - (float)hi[e]);
    }
    __builtin_amdgcn_s_waitcnt(0);                       // the line's eight lanes have their values before the first store
    __builtin_amdgcn_wave_barrier();
    split_t* px = reinterpret_cast<split_t*>(rowp);
    *reinterpret_cast<bf16x4*>(px + split_idx(col)) = hi;
    *reinterpret_cast<bf16x4*>(px + split_idx(col) + 32) = lo;
  }
}

// rows softmax, one wave per row, cols <= 1024 and % 4 == 0
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int rows, int cols) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* r = s + (long long)row * cols;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, r[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float e = expf(r[c] - mx);
    r[c] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int c = lane; c < cols; c += 64) r[c] *= inv;
}

// conv_out 3x3 (C=128 -> 3) on 128x128; x NHWC (post GN+swish); weights [3][9][128].
// A block is one image row; a half-wave (32 lanes = the 128 channels as float4) walks 16 consecutive pixels with a sliding
// 3x3 window in registers (3 coalesced 512-B loads per pixel instead of 27 strided ones), its lane's 108 weights in registers,
// and reduces the three outputs over the 32 lanes; lane j keeps pixel j, so the row is written as 64-B runs.
// Writes roll[n][co][y][s*128 + x] (tile m = s*Nb + n) and optionally the uint8 roll.
__global__ __launch_bounds__(256) void vae_conv_out_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ roll,
                                                           uint8_t* __restrict__ u8, int M, int Nb, int Tt, float thr) {
  constexpr int C = 128, Q = C / 4;
  const int lane = threadIdx.x & 63, c4 = lane & 31;
  const int seg = (threadIdx.x >> 6) * 2 + (lane >> 5);      // 8 segments of 16 pixels
  const int m = blockIdx.x >> 7, y = blockIdx.x & 127, x0 = seg * 16;
  float4 wr[3][9];
#pragma unroll
  for (int co = 0; co < 3; ++co)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wr[co][tap] = *reinterpret_cast<const float4*>(w + (co * 9 + tap) * C + c4 * 4);
  const float4* xb = reinterpret_cast<const float4*>(x + (long long)m * 16384 * C) + c4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  auto ld = [&](int yy, int xx) { return ((unsigned)yy < 128u && (unsigned)xx < 128u) ? xb[((yy << 7) + xx) * Q] : zero; };
  float4 win[3][3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    win[dy][0] = ld(y + dy - 1, x0 - 1);
    win[dy][1] = ld(y + dy - 1, x0);
  }
  float keep[3] = {0.f, 0.f, 0.f};
#pragma unroll 2
  for (int j = 0; j < 16; ++j) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) win[dy][2] = ld(y + dy - 1, x0 + j + 1);
    float acc[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      float a = 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float4 v = win[dy][dx], ww = wr[co][dy * 3 + dx];
          a += (v.x * ww.x + v.y * ww.y) + (v.z * ww.z + v.w * ww.w);
        }
      acc[co] = a;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
#pragma unroll
      for (int co = 0; co < 3; ++co) acc[co] += __shfl_xor(acc[co], o, 64);
    }
    if (c4 == j) {
      keep[0] = acc[0]; keep[1] = acc[1]; keep[2] = acc[2];
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      win[dy][0] = win[dy][1];
      win[dy][1] = win[dy][2];
    }
  }
  if (c4 >= 16) return;
  const int xx0 = x0 + c4;
  const int s = m / Nb, n = m - s * Nb;
  const long long plane = (long long)128 * Tt;
  const long long o = (long long)n * 3 * plane + (long long)y * Tt + s * 128 + xx0;
#pragma unroll
  for (int co = 0; co < 3; ++co) {
    const float r = keep[co] + b[co];
    if (roll) roll[o + co * plane] = r;
    if (u8) {  // midi_util.py:59-63, output layout (B,128,T,3)
      float v = r <= thr ? -1.0f : r;
      v = fminf(fmaxf((v + 1.0f) * 63.5f, 0.0f), 127.0f);
      u8[((long long)n * 128 + y) * Tt * 3 + (long long)(s * 128 + xx0) * 3 + co] = (uint8_t)v;
    }
  }
}


// conv_out, LDS-tiled (round 4): a workgroup computes TWO image rows (256 lanes = 256 output pixels).  The four input rows it needs go
// through LDS 32 channels at a time as [8 channel quads][4 rows][130 pixels] float4 (one quad plane is padded by a float4: the eight
// lanes of a 128-B global line then write eight different bank groups) -- every input value is fetched once per workgroup (2x over the
// launch instead of the 3x of one-row workgroups plus nothing but coalesced 128-B lines), a lane owns a PIXEL and keeps its three
// outputs in registers (no cross-lane reduction: the old kernel spent 15 shuffles per pixel), the 3 x 9 x 128 weights are wave-uniform
// and come through the scalar cache.  Same sums in another order than vae_conv_out_kernel: 3.67 -> ~1.3 ms per 512-square decode.
// NORM: x is the RAW residual stream and norm_out + swish (ref taming model.py:535-537: h = norm_out(h); h = nonlinearity(h); conv_out(h))
// are applied as the values are staged -- gn_apply's expression, so the conv sees the values the separate pass would have written; every
// value is staged by two workgroups (four rows for two output rows), i.e. normalised twice instead of being written and read once.
template <bool NORM>
__global__ __launch_bounds__(256) void vae_conv_out_tiled_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ b, float* __restrict__ roll,
                                                                 uint8_t* __restrict__ u8, int M, int Nb, int Tt, float thr,
                                                                 const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta) {
  constexpr int C = 128, CQ = 8, ROWS = 4, PX = 130, PLANE = ROWS * PX + 1;   // float4 units
  __shared__ float4 tile[CQ * PLANE];
  const int tid = threadIdx.x;
  const int m = blockIdx.x >> 6, y0 = (blockIdx.x & 63) * 2;
  const int r = tid >> 7, xx = tid & 127;                 // this lane's output pixel: row y0 + r, column xx
  const float4* xb = reinterpret_cast<const float4*>(x + (long long)m * 16384 * C);
  float acc[3] = {0.f, 0.f, 0.f};
  for (int cb = 0; cb < C / 32; ++cb) {
    if (cb) __syncthreads();
    // stage rows y0-1 .. y0+2, pixels -1 .. 128, channels 32 cb .. +31: consecutive lanes = the 8 quads of one pixel (one 128-B line)
    // (all 17 loads of a lane in flight before the first LDS write: a rolled load -> wait -> write loop was 64 serial round trips per workgroup)
    constexpr int NLD = (ROWS * PX * CQ + 255) / 256;
    float4 stage[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 256;
      const int pp = i >> 3, row = pp / PX, px = pp - row * PX;
      const int yy = y0 - 1 + row, xs = px - 1;
      stage[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < ROWS * PX * CQ && (unsigned)yy < 128u && (unsigned)xs < 128u) stage[k] = xb[((yy << 7) + xs) * (C / 4) + cb * 8 + (i & 7)];
    }
    if constexpr (NORM) {   // lane = channel quad cb * 8 + (tid & 7) for every k: one group (4 channels per group at C = 128), one gamma / beta quad
      const int cq = cb * 8 + (tid & 7);
      const float mean = stats[(m * 32 + cq) * 2], rstd = stats[(m * 32 + cq) * 2 + 1];
      const float4 ga = reinterpret_cast<const float4*>(gamma)[cq], be = reinterpret_cast<const float4*>(beta)[cq];
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * 256;
        const int pp = i >> 3, row = pp / PX, px = pp - row * PX;
        const int yy = y0 - 1 + row, xs = px - 1;
        if (i < ROWS * PX * CQ && (unsigned)yy < 128u && (unsigned)xs < 128u) {      // the conv's zero padding stays zero
          float4 o;
          o.x = silu_fast_f((stage[k].x - mean) * rstd * ga.x + be.x);
          o.y = silu_fast_f((stage[k].y - mean) * rstd * ga.y + be.y);
          o.z = silu_fast_f((stage[k].z - mean) * rstd * ga.z + be.z);
          o.w = silu_fast_f((stage[k].w - mean) * rstd * ga.w + be.w);
          stage[k] = o;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 256;
      const int pp = i >> 3, row = pp / PX, px = pp - row * PX;
      if (i < ROWS * PX * CQ) tile[(i & 7) * PLANE + row * PX + px] = stage[k];
    }
    __syncthreads();
    const float* wc = w + cb * 32;
#pragma unroll 1
    for (int q = 0; q < CQ; ++q) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float4 v = tile[q * PLANE + (r + dy) * PX + xx + dx];
#pragma unroll
          for (int co = 0; co < 3; ++co) {
            const float* ww = wc + (co * 9 + dy * 3 + dx) * C + q * 4;      // wave-uniform: scalar loads
            acc[co] = fmaf(v.x, ww[0], acc[co]);
            acc[co] = fmaf(v.y, ww[1], acc[co]);
            acc[co] = fmaf(v.z, ww[2], acc[co]);
            acc[co] = fmaf(v.w, ww[3], acc[co]);
          }
        }
    }
  }
  const int y = y0 + r;
  const int s = m / Nb, n = m - s * Nb;
  const long long plane = (long long)128 * Tt;
  const long long o = (long long)n * 3 * plane + (long long)y * Tt + s * 128 + xx;
#pragma unroll
  for (int co = 0; co < 3; ++co) {
    const float rr = acc[co] + b[co];
    if (roll) roll[o + co * plane] = rr;
    if (u8) {  // midi_util.py:59-63, output layout (B,128,T,3)
      float v = rr <= thr ? -1.0f : rr;
      v = fminf(fmaxf((v + 1.0f) * 63.5f, 0.0f), 127.0f);
      u8[((long long)n * 128 + y) * Tt * 3 + (long long)(s * 128 + xx) * 3 + co] = (uint8_t)v;
    }
  }
}

// [Cout][Cin][3][3] -> [Cout][9][Cin]
__global__ void repack_conv3_kernel(const float* __restrict__ in, float* __restrict__ out, int Cout, int Cin) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)Cout * Cin * 9) return;
  const int ci = (int)(i % Cin);
  const int tap = (int)((i / Cin) % 9);
  const int co = (int)(i / ((long long)Cin * 9));
  out[i] = in[((long long)co * Cin + ci) * 9 + tap];
}

// [Cout][9][Cin] -> [Cout][Cin/32][9][32]: the K order of the one-wave-per-SIMD conv kernels (gemm2.hip PIPE 5, GemmParams::conv_kmajor):
// the nine taps of one 32-channel block are consecutive K-tiles, so the same input lines are re-read within 9 K-tiles instead of once
// per Cin/32 K-tiles -- they then come from the XCD's L2 instead of the fabric (in-situ PMC, 128-channel convs of a 64-candidate decode:
// 47.7 GB fetched per launch = the whole im2col matrix, L2 hit 36 %, the launch at 5 TB/s)
__global__ void repack_conv3_kmajor_kernel(const float* __restrict__ in, float* __restrict__ out, int Cout, int Cin) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)Cout * Cin * 9) return;
  const int ci = (int)(i % Cin);
  const int tap = (int)((i / Cin) % 9);
  const long long co = i / ((long long)Cin * 9);
  out[(co * (Cin / 32) + ci / 32) * 288 + tap * 32 + (ci & 31)] = in[i];
}

// x<=thr -> -1 ; clamp((x+1)*63.5, 0, 127) -> uint8 ; (B,3,128,T) -> (B,128,T,3)
__global__ void quantise_roll_kernel(const float* __restrict__ roll, uint8_t* __restrict__ u8, int B, int T, float thr) {
  const long long total = (long long)B * 128 * T * 3;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % 3);
  long long r = i / 3;
  const int t = (int)(r % T);
  r /= T;
  const int p = (int)(r % 128);
  const int b = (int)(r / 128);
  float v = roll[(((long long)b * 3 + c) * 128 + p) * T + t];
  v = v <= thr ? -1.0f : v;
  v = fminf(fmaxf((v + 1.0f) * 63.5f, 0.0f), 127.0f);
  u8[i] = (uint8_t)v;
}


// ------------------------------------------------------------------- decoder backward (input gradients only)
// GroupNorm(+swish) backward, pass 1: per (tile, group) sums of dxhat and dxhat*xhat in fp64, where
// xhat = (x-mean)*rstd, y = xhat*gamma+beta, dxhat = dz * swish'(y) * gamma.  Same grid / chunking as gn_partial_kernel.
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                             const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, double* __restrict__ part, int P,
                                                             int C, int chunks, int swish) {
  __shared__ double sh[32][2];
  const int m = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) sh[tid >> 1][tid & 1] = 0.0;
  __syncthreads();
  const int q = C >> 2;
  const int ppi = 256 / q;
  const int cq = tid % q, p_off = tid / q;
  const int p0 = (int)((long long)P * ch / chunks), p1 = (int)((long long)P * (ch + 1) / chunks);
  const float4* xb = reinterpret_cast<const float4*>(x + (long long)m * P * C);
  const float4* db = reinterpret_cast<const float4*>(dz + (long long)m * P * C);
  const int g = (cq * 4) / (C >> 5);
  const float mean = stats[(m * 32 + g) * 2], rstd = stats[(m * 32 + g) * 2 + 1];
  const float4 ga4 = reinterpret_cast<const float4*>(gamma)[cq], be4 = reinterpret_cast<const float4*>(beta)[cq];
  const float ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, be[4] = {be4.x, be4.y, be4.z, be4.w};
  double s = 0.0, ss = 0.0;
  for (int p = p0 + p_off; p < p1; p += ppi) {
    const float4 v4 = xb[(long long)p * q + cq], d4 = db[(long long)p * q + cq];
    const float v[4] = {v4.x, v4.y, v4.z, v4.w}, d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (v[k] - mean) * rstd;
      const float dy = swish ? d[k] * silu_grad_f(xh * ga[k] + be[k]) : d[k];
      const float dxh = dy * ga[k];
      s += (double)dxh;
      ss += (double)dxh * (double)xh;
    }
  }
  atomicAdd(&sh[g][0], s);
  atomicAdd(&sh[g][1], ss);
  __syncthreads();
  if (tid < 64) part[(((long long)m * chunks + ch) * 32 + (tid >> 1)) * 2 + (tid & 1)] = sh[tid >> 1][tid & 1];
}

__global__ void gn_bwd_finalize_kernel(const double* __restrict__ part, float* __restrict__ sums, int M, int chunks, double count) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over M*32
  if (idx >= M * 32) return;
  const int m = idx >> 5, g = idx & 31;
  double s = 0.0, ss = 0.0;
  for (int c = 0; c < chunks; ++c) {
    s += part[(((long long)m * chunks + c) * 32 + g) * 2];
    ss += part[(((long long)m * chunks + c) * 32 + g) * 2 + 1];
  }
  sums[idx * 2] = (float)(s / count);
  sums[idx * 2 + 1] = (float)(ss / count);
}

// pass 2: dx = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat*xhat)) (+ add); may run in place on dz
__global__ void gn_bwd_apply_kernel(const float* __restrict__ x, const float* dz, const float* __restrict__ stats,
                                    const float* __restrict__ sums, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* add, float* out, long long total4, int P, int C,
                                    int swish, int out_split) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int q = C >> 2;
  const int c4 = (int)(i % q);
  const int m = (int)(i / ((long long)q * P));
  const int g = (c4 * 4) / (C >> 5);
  const float mean = stats[(m * 32 + g) * 2], rstd = stats[(m * 32 + g) * 2 + 1];
  const float m1 = sums[(m * 32 + g) * 2], m2 = sums[(m * 32 + g) * 2 + 1];
  const float4 v4 = reinterpret_cast<const float4*>(x)[i], d4 = reinterpret_cast<const float4*>(dz)[i];
  const float4 ga4 = reinterpret_cast<const float4*>(gamma)[c4], be4 = reinterpret_cast<const float4*>(beta)[c4];
  const float v[4] = {v4.x, v4.y, v4.z, v4.w}, d[4] = {d4.x, d4.y, d4.z, d4.w};
  const float ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, be[4] = {be4.x, be4.y, be4.z, be4.w};
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float xh = (v[k] - mean) * rstd;
    const float dy = swish ? d[k] * silu_grad_f(xh * ga[k] + be[k]) : d[k];
    o[k] = rstd * (dy * ga[k] - m1 - xh * m2);
  }
  if (add) {
    const float4 a = reinterpret_cast<const float4*>(add)[i];
    o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
  }
  if (out_split) {   // split-row pixels for the pre-split input-gradient conv (out must not alias dz then)
    typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (split_t)o[k];
      lo[k] = (split_t)(o[k] - (float)hi[k]);
    }
    split_t* px = reinterpret_cast<split_t*>(out + (i / q) * C);
    *reinterpret_cast<bf16x4*>(px + split_idx(c4 * 4)) = hi;
    *reinterpret_cast<bf16x4*>(px + split_idx(c4 * 4) + 32) = lo;
  } else {
    reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// conv_out backward: d_in[m][y][x][c] = sum_{co,ky,kx} d_roll[n][co][y-ky+1][s*128 + x-kx+1] * w[co][ky*3+kx][c]
// (the roll gradient is read in the (N,3,128,T) layout the forward scatters to); one thread per (pixel, 4 channels)
__global__ __launch_bounds__(256) void vae_conv_out_bwd_kernel(const float* __restrict__ droll, const float* __restrict__ w,
                                                               float* __restrict__ dx, int M, int Nb, int Tt) {
  constexpr int C = 128;
  __shared__ __attribute__((aligned(16))) float ws[3 * 9 * C];
  for (int i = threadIdx.x; i < 3 * 9 * C; i += 256) ws[i] = w[i];
  __syncthreads();
  const long long idx = blockIdx.x * 256LL + threadIdx.x;     // over M*16384*32 (grid exact)
  const int c4 = (int)(idx & 31);
  const int pix = (int)(idx >> 5);
  const int m = pix >> 14, y = (pix >> 7) & 127, x0 = pix & 127;
  const int s = m / Nb, n = m - s * Nb;
  const long long plane = 128LL * Tt;
  const float* base = droll + (long long)n * 3 * plane + s * 128;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int yo = y - tap / 3 + 1, xo = x0 - tap % 3 + 1;
    if ((unsigned)yo >= 128u || (unsigned)xo >= 128u) continue;
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      const float d = base[co * plane + (long long)yo * Tt + xo];
      const float4 ww = *reinterpret_cast<const float4*>(ws + (co * 9 + tap) * C + c4 * 4);
      acc.x = fmaf(d, ww.x, acc.x); acc.y = fmaf(d, ww.y, acc.y); acc.z = fmaf(d, ww.z, acc.z); acc.w = fmaf(d, ww.w, acc.w);
    }
  }
  reinterpret_cast<float4*>(dx)[idx] = acc;
}

// backward of the nearest x2 upsampling folded into the upsample conv: out[m][y][x][c] = sum of the 2x2 block of in
__global__ void sumpool2_kernel(const float* __restrict__ in, float* __restrict__ out, long long total4, int H, int q) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;   // over M*H*H*q (H = output size)
  if (i >= total4) return;
  const int c4 = (int)(i % q);
  const long long pix = i / q;
  const int x = (int)(pix % H), y = (int)((pix / H) % H);
  const long long m = pix / ((long long)H * H);
  const float4* b = reinterpret_cast<const float4*>(in) + ((m * 2 * H + 2 * y) * 2 * H + 2 * x) * q + c4;
  const float4 a0 = b[0], a1 = b[q], a2 = b[2LL * H * q], a3 = b[2LL * H * q + q];
  reinterpret_cast<float4*>(out)[i] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                                                  (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
}

// conv_in (4 -> Cout, 3x3) and post_quant_conv (1x1, 4 -> 4) backward plus the scatter back into the latent layout the
// forward gathered from (incl. in_scale): one wave per 16x16-square pixel, lanes stride the Cout channels of dy
__global__ __launch_bounds__(256) void vae_conv_in_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              const float* __restrict__ wpq, float* __restrict__ dlat, int M,
                                                              int Cout, int Nb, long long n_stride, long long s_stride,
                                                              long long sc, long long si, long long sj, float in_scale) {
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= M * 256) return;
  const int lane = threadIdx.x & 63;
  const int m = pix >> 8, y = (pix >> 4) & 15, x = pix & 15;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int tap = 0; tap < 9; ++tap) {
    const int yo = y - tap / 3 + 1, xo = x - tap % 3 + 1;
    if ((unsigned)yo >= 16u || (unsigned)xo >= 16u) continue;
    const float* d = dy + ((long long)(m << 8) + (yo << 4) + xo) * Cout;
    for (int co = lane; co < Cout; co += 64) {
      const float g = d[co];
      const float4 ww = *reinterpret_cast<const float4*>(w + ((long long)co * 9 + tap) * 4);
      a0 = fmaf(g, ww.x, a0); a1 = fmaf(g, ww.y, a1); a2 = fmaf(g, ww.z, a2); a3 = fmaf(g, ww.w, a3);
    }
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
  if (lane == 0) {
    const int s = m / Nb, n = m - s * Nb;
    float* p = dlat + n * n_stride + s * s_stride + y * si + x * sj;
    const float dpq[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      p[c * sc] = in_scale * (wpq[0 * 4 + c] * dpq[0] + wpq[1 * 4 + c] * dpq[1] + wpq[2 * 4 + c] * dpq[2] + wpq[3 * 4 + c] * dpq[3]);
  }
}

// softmax backward on rows, in place on dP: dS = alpha * P * (dP - sum_j dP_j P_j); one wave per row
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP, int rows,
                                                               int cols, float alpha) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = P + (long long)row * cols;
  float* d = dP + (long long)row * cols;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot = fmaf(d[c], p[c], dot);
  dot = wave_sum(dot);
  for (int c = lane; c < cols; c += 64) d[c] = alpha * p[c] * (d[c] - dot);
}

// weights of the input-gradient conv: rep [Cout][9][Cin] (arena layout) -> out [Cin][9][Cout] with the taps mirrored
__global__ void repack_conv3_dgrad_kernel(const float* __restrict__ rep, float* __restrict__ out, int Cout, int Cin) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)Cout * Cin * 9) return;
  const int co = (int)(i % Cout);
  const int tap = (int)((i / Cout) % 9);
  const int ci = (int)(i / ((long long)Cout * 9));
  out[i] = rep[((long long)co * 9 + (8 - tap)) * Cin + ci];
}

}  // namespace rgm

using namespace rgm;

struct VSlot {
  size_t off = 0, numel = 0;
  bool set = false;
  int conv3 = 0;  // 1: repack [co][ci][3][3] -> [co][9][ci]
  int lin = 0;    // 1: 1x1 conv weight [cout][cin] of the decoder that also keeps a split-row copy "<key>.S" (pre-split GEMM, gemm2.hip)
  int cout = 0, cin = 0;
  bool derived = false;  // "<key>.S": split-row copy of a 3x3 weight for the pre-split conv GEMM (gemm2.hip), not a parameter
  int group = 0;         // 0: decoder + post_quant_conv (needed by decode), 1: encoder + quant_conv (needed by encode)
};

struct rgm_vae {
  std::map<std::string, VSlot> slots;
  float* arena = nullptr;
  size_t arena_floats = 0;
  float* stage = nullptr;  // staging for repack
  size_t stage_floats = 0;
  int ch = 128;
  int cur_group = 0;       // group given to the slots being registered
  const float* p(const std::string& k) const { return arena + slots.at(k).off; }
  // input-gradient copies of the decoder weights (rgm_vae_enable_grad): 3x3 convs as [Cin][9 mirrored][Cout], 1x1 as [Cin][Cout]
  float* garena = nullptr;
  std::map<std::string, size_t> goff;
  bool grad_ready = false;
  const float* gp(const std::string& k) const { return garena + goff.at(k); }
};

static void vslot(rgm_vae* h, const std::string& key, size_t numel, int conv3 = 0, int cout = 0, int cin = 0, int lin = 0) {
  VSlot s;
  s.off = h->arena_floats;
  s.numel = numel;
  s.conv3 = conv3;
  s.lin = (lin && cin % 32 == 0 && cout % 32 == 0 && h->cur_group == 0) ? 1 : 0;
  s.cout = cout;
  s.cin = cin;
  s.group = h->cur_group;
  h->slots[key] = s;
  h->arena_floats += (numel + 3) / 4 * 4;
  if (numel > h->stage_floats) h->stage_floats = numel;
  if (s.lin) {
    VSlot d;
    d.off = h->arena_floats;
    d.numel = numel;
    d.derived = true;
    d.set = true;
    h->slots[key + ".S"] = d;
    h->arena_floats += (numel + 3) / 4 * 4;
  }
  if (conv3 && cin % 32 == 0 && h->cur_group == 0) {   // the encoder convs stay on the on-the-fly kernel
    VSlot d;
    d.off = h->arena_floats;
    d.numel = numel;
    d.derived = true;
    d.set = true;
    h->slots[key + ".S"] = d;
    h->arena_floats += (numel + 3) / 4 * 4;
    VSlot k = d;                                        // the same weight, channel-block-major K, split rows (repack_conv3_kmajor_kernel)
    k.off = h->arena_floats;
    h->slots[key + ".K"] = k;
    h->arena_floats += (numel + 3) / 4 * 4;
  }
}

static void res_slots(rgm_vae* h, const std::string& p, int cin, int cout) {
  vslot(h, p + "norm1.weight", cin);
  vslot(h, p + "norm1.bias", cin);
  vslot(h, p + "conv1.weight", (size_t)cout * cin * 9, 1, cout, cin);
  vslot(h, p + "conv1.bias", cout);
  vslot(h, p + "norm2.weight", cout);
  vslot(h, p + "norm2.bias", cout);
  vslot(h, p + "conv2.weight", (size_t)cout * cout * 9, 1, cout, cout);
  vslot(h, p + "conv2.bias", cout);
  if (cin != cout) {
    vslot(h, p + "nin_shortcut.weight", (size_t)cout * cin, 0, cout, cin, 1);
    vslot(h, p + "nin_shortcut.bias", cout);
  }
}

static const int CH_MULT[4] = {1, 2, 2, 4};

extern "C" int rgm_vae_create(rgm_vae** out) {
  RGM_REQUIRE(out, "vae_create: null argument");
  rgm_vae* h = new rgm_vae();
  const int ch = h->ch;
  vslot(h, "post_quant_conv.weight", 16);
  vslot(h, "post_quant_conv.bias", 4);
  int bi = ch * CH_MULT[3];
  const std::string d = "decoder.";
  vslot(h, d + "conv_in.weight", (size_t)bi * 4 * 9, 1, bi, 4);
  vslot(h, d + "conv_in.bias", bi);
  res_slots(h, d + "mid.block_1.", bi, bi);
  vslot(h, d + "mid.attn_1.norm.weight", bi);
  vslot(h, d + "mid.attn_1.norm.bias", bi);
  for (const char* nm : {"q", "k", "v", "proj_out"}) {
    vslot(h, d + "mid.attn_1." + nm + ".weight", (size_t)bi * bi, 0, bi, bi, 1);
    vslot(h, d + "mid.attn_1." + nm + ".bias", bi);
  }
  res_slots(h, d + "mid.block_2.", bi, bi);
  for (int lvl = 3; lvl >= 0; --lvl) {
    const int bo = ch * CH_MULT[lvl];
    for (int ib = 0; ib < 3; ++ib) {
      res_slots(h, d + "up." + std::to_string(lvl) + ".block." + std::to_string(ib) + ".", bi, bo);
      bi = bo;
    }
    if (lvl != 0) {
      vslot(h, d + "up." + std::to_string(lvl) + ".upsample.conv.weight", (size_t)bi * bi * 9, 1, bi, bi);
      vslot(h, d + "up." + std::to_string(lvl) + ".upsample.conv.bias", bi);
    }
  }
  vslot(h, d + "norm_out.weight", bi);
  vslot(h, d + "norm_out.bias", bi);
  vslot(h, d + "conv_out.weight", (size_t)3 * bi * 9, 1, 3, bi);
  vslot(h, d + "conv_out.bias", 3);
  {  // Encoder + quant_conv (taming Encoder, model.py:342-433; klvae_pedal.py:30-31): optional, used by rgm_vae_encode
    h->cur_group = 1;
    const std::string e = "encoder.";
    vslot(h, e + "conv_in.weight", (size_t)ch * 3 * 9);
    vslot(h, e + "conv_in.bias", ch);
    int ei = ch;
    for (int lvl = 0; lvl < 4; ++lvl) {
      const int eo = ch * CH_MULT[lvl];
      for (int ib = 0; ib < 2; ++ib) {
        res_slots(h, e + "down." + std::to_string(lvl) + ".block." + std::to_string(ib) + ".", ei, eo);
        ei = eo;
      }
      if (lvl != 3) {
        vslot(h, e + "down." + std::to_string(lvl) + ".downsample.conv.weight", (size_t)ei * ei * 9, 1, ei, ei);
        vslot(h, e + "down." + std::to_string(lvl) + ".downsample.conv.bias", ei);
      }
    }
    res_slots(h, e + "mid.block_1.", ei, ei);
    vslot(h, e + "mid.attn_1.norm.weight", ei);
    vslot(h, e + "mid.attn_1.norm.bias", ei);
    for (const char* nm : {"q", "k", "v", "proj_out"}) {
      vslot(h, e + "mid.attn_1." + nm + ".weight", (size_t)ei * ei);
      vslot(h, e + "mid.attn_1." + nm + ".bias", ei);
    }
    res_slots(h, e + "mid.block_2.", ei, ei);
    vslot(h, e + "norm_out.weight", ei);
    vslot(h, e + "norm_out.bias", ei);
    vslot(h, e + "conv_out.weight", (size_t)8 * ei * 9, 1, 8, ei);
    vslot(h, e + "conv_out.bias", 8);
    vslot(h, "quant_conv.weight", 64);
    vslot(h, "quant_conv.bias", 8);
    h->cur_group = 0;
  }
  RGM_CHECK_HIP(hipMalloc(&h->arena, h->arena_floats * sizeof(float)));
  RGM_CHECK_HIP(hipMalloc(&h->stage, h->stage_floats * sizeof(float)));
  *out = h;
  return RGM_OK;
}

extern "C" void rgm_vae_destroy(rgm_vae* h) {
  if (!h) return;
  if (h->arena) (void)hipFree(h->arena);
  if (h->stage) (void)hipFree(h->stage);
  if (h->garena) (void)hipFree(h->garena);
  delete h;
}

extern "C" int rgm_vae_has_param(rgm_vae* h, const char* key) {
  if (!h || !key) return 0;
  auto it = h->slots.find(key);
  return it != h->slots.end() && !it->second.derived ? 1 : 0;
}

extern "C" int rgm_vae_set_param(rgm_vae* h, const char* key, const void* dptr, const int64_t* shape, int ndim) {
  RGM_REQUIRE(h && key && dptr, "vae_set_param: null argument");
  auto it = h->slots.find(key);
  RGM_REQUIRE(it != h->slots.end() && !it->second.derived, "vae_set_param: unknown key '%s'", key);
  size_t numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
  VSlot& s = it->second;
  RGM_REQUIRE(numel == s.numel, "vae_set_param: '%s' has %zu elements, expected %zu", key, numel, s.numel);
  if (s.conv3) {
    RGM_CHECK_HIP(hipMemcpy(h->stage, dptr, numel * sizeof(float), hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL(repack_conv3_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, 0, h->stage, h->arena + s.off, s.cout, s.cin);
    RGM_LAUNCH_CHECK();
    auto sp = h->slots.find(std::string(key) + ".S");
    if (sp != h->slots.end())   // rows = cout, K = 9*cin: the 32-blocks of a split row never straddle a tap (cin % 32 == 0)
      RGM_TRY(split_rows_launch(h->arena + s.off, h->arena + sp->second.off, s.cout, 9 * s.cin, 9 * s.cin, 9 * s.cin, 0));
    auto kp = h->slots.find(std::string(key) + ".K");
    if (kp != h->slots.end()) {                          // stage <- kc-major fp32, then split rows into the slot
      hipLaunchKernelGGL(repack_conv3_kmajor_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, 0, h->arena + s.off, h->stage, s.cout, s.cin);
      RGM_LAUNCH_CHECK();
      RGM_TRY(split_rows_launch(h->stage, h->arena + kp->second.off, s.cout, 9 * s.cin, 9 * s.cin, 9 * s.cin, 0));
    }
    RGM_CHECK_HIP(hipStreamSynchronize(0));
  } else {
    RGM_CHECK_HIP(hipMemcpy(h->arena + s.off, dptr, numel * sizeof(float), hipMemcpyDeviceToDevice));
    if (s.lin) {
      RGM_TRY(split_rows_launch(h->arena + s.off, h->arena + h->slots.at(std::string(key) + ".S").off, s.cout, s.cin, s.cin, s.cin, 0));
      RGM_CHECK_HIP(hipStreamSynchronize(0));
    }
  }
  s.set = true;
  h->grad_ready = false;   // the input-gradient copies are stale: rgm_vae_enable_grad again
  return RGM_OK;
}

extern "C" int rgm_vae_missing_params(rgm_vae* h) {
  if (!h) return -1;
  int n = 0;
  for (auto& kv : h->slots) n += (kv.second.set || kv.second.group != 0) ? 0 : 1;   // what decode needs
  return n;
}

extern "C" int rgm_vae_encoder_missing_params(rgm_vae* h) {
  if (!h) return -1;
  int n = 0;
  for (auto& kv : h->slots) n += (kv.second.set || kv.second.group != 1) ? 0 : 1;   // what encode needs
  return n;
}

// Builds the weight copies the decoder's input-gradient pass multiplies with (decoder 3x3 convs with >= 32 channels on
// both sides, nin_shortcut and the attention 1x1 convs).  Call after the decoder parameters are set; set_param invalidates.
extern "C" int rgm_vae_enable_grad(rgm_vae* h) {
  RGM_REQUIRE(h, "vae_enable_grad: null handle");
  RGM_REQUIRE(rgm_vae_missing_params(h) == 0, "vae_enable_grad: %d decoder parameters not set", rgm_vae_missing_params(h));
  if (h->grad_ready) return RGM_OK;
  if (!h->garena) {
    size_t off = 0;
    for (auto& kv : h->slots) {
      const VSlot& s = kv.second;
      const std::string& k = kv.first;
      if (s.derived || s.group != 0 || k.compare(0, 8, "decoder.") != 0) continue;
      const bool c3 = s.conv3 && s.cin % 32 == 0 && s.cout % 32 == 0;
      const bool c1 = !s.conv3 && k.size() > 7 && k.compare(k.size() - 7, 7, ".weight") == 0 &&
                      (k.find("nin_shortcut") != std::string::npos || k.find("attn_1.q.") != std::string::npos ||
                       k.find("attn_1.k.") != std::string::npos || k.find("attn_1.v.") != std::string::npos ||
                       k.find("attn_1.proj_out.") != std::string::npos);
      if (!c3 && !c1) continue;
      h->goff[k + ".T"] = off;
      off += (s.numel + 3) / 4 * 4;
      if (c3) {   // split-row copy for the pre-split kernel (rows = Cin, K = 9*Cout: a 32-block never straddles a tap)
        h->goff[k + ".TS"] = off;
        off += (s.numel + 3) / 4 * 4;
      }
    }
    RGM_CHECK_HIP(hipMalloc(&h->garena, off * sizeof(float)));
  }
  for (auto& kv : h->goff) {
    if (kv.first.compare(kv.first.size() - 3, 3, ".TS") == 0) continue;   // written with its ".T" twin
    const std::string key = kv.first.substr(0, kv.first.size() - 2);
    const VSlot& s = h->slots.at(key);
    float* dst = h->garena + kv.second;
    if (s.conv3) {
      hipLaunchKernelGGL(repack_conv3_dgrad_kernel, dim3((unsigned)((s.numel + 255) / 256)), dim3(256), 0, 0, h->arena + s.off, dst,
                         s.cout, s.cin);
      RGM_LAUNCH_CHECK();
      RGM_TRY(split_rows_launch(dst, h->garena + h->goff.at(key + ".TS"), s.cin, 9 * s.cout, 9 * s.cout, 9 * s.cout, 0));
    } else {   // [Cout][Cin] -> [Cin][Cout]; 1x1 slots do not record their shape: bias length = Cout
      const int cout = (int)h->slots.at(key.substr(0, key.size() - 6) + "bias").numel;
      const int cin = (int)(s.numel / cout);
      RGM_TRY(transpose_launch(h->arena + s.off, dst, cout, cin, cout, 1, 0));
    }
  }
  RGM_CHECK_HIP(hipStreamSynchronize(0));
  h->grad_ready = true;
  return RGM_OK;
}

namespace {
constexpr int GN_CHUNKS = 16;
constexpr int GN_SLOTS = 16;   // conv launches per decode that may normalise their own output (14 conv1's in the decoder)
struct VPlan {
  float *pq, *b0, *b1, *b2, *q, *k, *v, *vt, *sc, *stats;
  double *part, *tpart;
  unsigned* gn_count;     // arrival counters of a conv launch that normalises its own output: one per (image, column tile)
  int* gn_fail;           // per-tile flags of that launch (a tile whose wait ran out wrote raw rows)
  size_t bytes;
};
VPlan vplan(int M, void* ws) {
  VPlan p{};
  char* base = (char*)ws;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = base + off;
    off += align_up(bytes, 256);
    return r;
  };
  const size_t big = (size_t)M * 128 * 128 * 256 * sizeof(float);
  p.pq = (float*)take((size_t)M * 256 * 4 * sizeof(float));
  p.b0 = (float*)take(big);
  p.b1 = (float*)take(big);
  p.b2 = (float*)take(big);
  const size_t tok = (size_t)M * 256 * 512 * sizeof(float);
  p.q = (float*)take(tok);
  p.k = (float*)take(tok);
  p.v = (float*)take(tok);
  p.vt = (float*)take(tok);
  p.sc = (float*)take((size_t)M * 256 * 256 * sizeof(float));
  p.stats = (float*)take((size_t)M * 32 * 2 * sizeof(float));
  p.part = (double*)take((size_t)M * GN_CHUNKS * 32 * 2 * sizeof(double));
  p.tpart = (double*)take((size_t)M * 128 * 32 * 2 * sizeof(double));   // per-tile GroupNorm sums from a conv epilogue (<= 128 row tiles)
  // GN_SLOTS launches of a decode each own a slice (M * 2 counters + M * 64 flags, contiguous): ONE memset when the first of them is met
  p.gn_count = (unsigned*)take((size_t)GN_SLOTS * M * 66 * sizeof(unsigned));
  p.gn_fail = nullptr;
  p.bytes = off;
  return p;
}

// GroupNorm + swish inside the producing conv (GemmParams::gn_count): 0 = never, 1 = where the launch qualifies (conv3), 2 = the same with
// every tile forced onto the fallback (tests of gn_fixup_kernel)
static int g_gn_fuse = getenv("RGM_GN_FUSE") ? atoi(getenv("RGM_GN_FUSE")) : 1;
static long long g_gn_fused_launches = 0;
static unsigned long long* g_gn_fallbacks = nullptr;      // device counter: tiles that took the raw-row fallback (gn_fixup_kernel)

struct Ctx {
  rgm_vae* h;
  VPlan p;
  int M;
  hipStream_t s;
  int split = 0;   // bf16x3_presplit: GroupNorm writes split rows, the 3x3 convs run on gemm2.hip
  // GroupNorm statistics fused into the producing conv: the pre-split conv GEMM leaves per-tile sums of its output in p.tpart;
  // a group_norm that is the very next op on that tensor finalises them instead of re-reading the tensor from HBM
  int seq = 0, tp_seq = -1;
  const float* tp_for = nullptr;
  int final_split = 0;           // the next same-channel ResnetBlock writes its OUTPUT as split rows only (its one consumer is the upsampling conv)
  int gn_slot = 0;               // next free slice of the plan's counters / flags (GemmParams::gn_count); zeroed when slot 0 is taken
  int tp_rows = 128;             // tile height of the conv that left those sums (128: heuristic tiles; 256 / 512: the 256x256 / 512x128 kernels)
};

// stats: where (mean, rstd) of the M x 32 groups go (kept for the backward when given; default the shared scratch)
// y == nullptr: statistics only (the consumer normalises as it loads: conv_out_launch)
int group_norm(Ctx& c, const float* x, float* y, int P, int C, const std::string& key, int swish, int out_split = 0,
               float* stats = nullptr, float* raw_split = nullptr) {
  if (!stats) stats = c.p.stats;
  const int my = ++c.seq;
  if (c.tp_for == x && c.tp_seq == my - 1) {
    hipLaunchKernelGGL(gn_finalize_tiles_kernel, dim3(cdiv(c.M * 32, 256)), dim3(256), 0, c.s, c.p.tpart, stats, c.M, P / c.tp_rows,
                       (double)P * (C / 32), 1e-6f);
    RGM_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(gn_partial_kernel, dim3(GN_CHUNKS, c.M), dim3(256), 0, c.s, x, c.p.part, P, C, GN_CHUNKS);
    RGM_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(c.M * 32, 256)), dim3(256), 0, c.s, c.p.part, stats, c.M, GN_CHUNKS,
                       (double)P * (C / 32), 1e-6f);
    RGM_LAUNCH_CHECK();
  }
  if (!y) return RGM_OK;
  static const int apply8 = getenv("RGM_GN_APPLY8") ? atoi(getenv("RGM_GN_APPLY8")) : 1;   // 0: the four-channels-per-lane kernel (A/B runs)
  if (apply8 && C % 128 == 0) {
    const long long total8 = (long long)c.M * P * C / 8, half = (total8 + 1) / 2;
    hipLaunchKernelGGL(gn_apply8_kernel, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, c.s, x, y, stats,
                       c.h->p(key + ".weight"), c.h->p(key + ".bias"), total8, half, P, C, swish, out_split, raw_split);
    RGM_LAUNCH_CHECK();
    return RGM_OK;
  }
  const long long total4 = (long long)c.M * P * C / 4;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, c.s, x, y, stats,
                     c.h->p(key + ".weight"), c.h->p(key + ".bias"), total4, P, C, swish, out_split, raw_split);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// out[M*H*W, Cout] = conv3x3(in NHWC [M, H>>ups, W>>ups, Cin]) + bias (+ res)
// in_split: `in` holds split-row pixels (group_norm(..., out_split=1) or split_rows) -> pre-split LDS-DMA kernel
// fuse_norm (optional): key of the GroupNorm that consumes this conv's output and nothing else does -- applied (with swish) inside the
// launch when it qualifies; *fused then tells the caller that `out` already holds the normalised split rows.
int conv3(Ctx& c, const float* in, float* out, int H, int Cin, int Cout, const std::string& key, int ups, const float* res,
          int in_split = 0, const std::string* fuse_norm = nullptr, int* fused = nullptr, int out_split = 0) {
  GemmParams g;
  g.A = in; g.B = c.h->p(key + (in_split ? ".weight.S" : ".weight")); g.ldb = 9 * Cin; g.C = out; g.ldc = Cout;
  g.M = c.M * H * H; g.N = Cout; g.K = 9 * Cin; g.lda = Cin;
  g.bias = c.h->p(key + ".bias");
  g.res = res; g.ldres = Cout;
  g.aload = 1; g.H = H; g.W = H; g.Cin = Cin; g.logH = ilog2(H); g.logW = ilog2(H); g.ups = ups;
  const int my = ++c.seq;
  if (in_split) {
    g.out_split = out_split;                 // split rows instead of fp32 rows (same bytes): a consumer that is a pre-split GEMM reads them as they are
    g.tile = RGM_EXP_ENV("RGM_CONV_TILE");   // 0 = gemm2's heuristic (timing experiments: common.h)
    int rows = 128;
    if (g.tile == 0 && big_tiles_mode()) {
      // One wave per SIMD, 128x128 accumulators per wave (gemm2.hip PIPE 5): 256x256 tiles for the 256- / 512-channel convs, 512x128
      // for the 128-channel ones, once the grid fills at least one round of the chip (a 64-candidate decode: 32 ... 512 rounds);
      // tiles must not straddle an image (per-image GroupNorm sums), so 16x16 maps (256 pixels) only take the 256-row tile
      const long long P = (long long)H * H;
      const long long min_tiles = big_tiles_min();
      if ((Cout == 256 || Cout == 512) && P % 256 == 0 && ((long long)g.M / 256) * (Cout / 256) >= min_tiles) {
        g.tile = 71;
        rows = 256;
      } else if (Cout == 128 && P % 512 == 0 && (long long)g.M / 512 >= min_tiles) {
        g.tile = 72;
        rows = 512;
      }
    }
    // channel-block-major K for EVERY pre-split conv (weights: the .K copy), whatever tile the launch takes: results then do not depend
    // on the batch size through the tile choice, and the nine taps of a channel block re-read their input lines from L2
    static const int kmajor = getenv("RGM_CONV_KMAJOR") ? atoi(getenv("RGM_CONV_KMAJOR")) : 1;   // 0: tap-major (A/B runs)
    static const int kmajor_ups = getenv("RGM_CONV_KMAJOR_UPS") ? atoi(getenv("RGM_CONV_KMAJOR_UPS")) : 0;
    if (kmajor && (ups == 0 || kmajor_ups)) {
      g.B = c.h->p(key + ".weight.K");
      g.conv_kmajor = 1;
    }
    if ((g.tile == 0 || g.tile == 71 || g.tile == 72) && Cout % 128 == 0 && Cout <= 512 && (H * H) % rows == 0) {   // tiles never straddle an image
      g.stats = c.p.tpart;
      g.stats_gw = Cout / 32;
      c.tp_for = out;
      c.tp_seq = my;
      c.tp_rows = rows;
    }
    // GroupNorm + swish in this launch's epilogue: one-wave-per-SIMD tiles with channel-block-major K, no residual, and an image's row
    // tiles must be dispatched together -- one column tile, whole images per XCD chunk of the raster (tiles / 8 a multiple of the tiles of
    // an image), at most 32 tiles per image (the CUs of an XCD) on the 256-CU part -- or be a whole image each.
    if (fuse_norm && fused && g_gn_fuse && g.stats && g.conv_kmajor && !res && (g.tile == 71 || g.tile == 72)) {
      const int bn = g.tile == 71 ? 256 : 128;
      const long long tm = g.M / rows, tn = (Cout + bn - 1) / bn, tpi = (long long)H * H / rows;
      static int cus = -1;
      if (cus < 0) {
        int dev = 0;
        RGM_CHECK_HIP(hipGetDevice(&dev));
        RGM_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      }
      const bool together = tpi == 1 || (tn == 1 && tpi <= 32 && cus == 256 && tm % 8 == 0 && (tm / 8) % tpi == 0);
      if (together && c.gn_slot < GN_SLOTS && Cout % bn == 0 && tm * tn <= (long long)c.M * 64 && (tm / tpi) * tn <= (long long)c.M * 2) {
        if (c.gn_slot == 0) RGM_CHECK_HIP(hipMemsetAsync(c.p.gn_count, 0, (size_t)GN_SLOTS * c.M * 66 * sizeof(unsigned), c.s));
        g.gn_count = c.p.gn_count + (size_t)c.gn_slot * c.M * 66;
        g.gn_fail = reinterpret_cast<int*>(g.gn_count + (size_t)c.M * 2);
        ++c.gn_slot;
        g.gn_tiles = (int)tpi;
        g.gn_gamma = c.h->p(*fuse_norm + ".weight");
        g.gn_beta = c.h->p(*fuse_norm + ".bias");
        g.gn_n = (double)H * H * (Cout / 32);
        g.gn_eps = 1e-6f;
        g.gn_swish = 1;
        g.gn_force_fail = g_gn_fuse == 2 ? 1 : 0;
        g.out_split = 1;
        if (!g_gn_fallbacks) {
          RGM_CHECK_HIP(hipMalloc(&g_gn_fallbacks, sizeof(unsigned long long)));
          RGM_CHECK_HIP(hipMemset(g_gn_fallbacks, 0, sizeof(unsigned long long)));
        }
        RGM_TRY(gemm2_launch(g, c.s));
        hipLaunchKernelGGL(gn_fixup_kernel, dim3((unsigned)(tm * tn)), dim3(256), 0, c.s, out, (const int*)g.gn_fail, (const double*)c.p.tpart,
                           g.gn_gamma, g.gn_beta, (int)tn, rows, bn, Cout, (int)tpi, g.gn_n, g.gn_eps, 1, g_gn_fallbacks);
        RGM_LAUNCH_CHECK();
        *fused = 1;
        ++g_gn_fused_launches;
        c.tp_for = nullptr;                 // the sums belong to a tensor that no longer exists in raw form
        return RGM_OK;
      }
    }
    return gemm2_launch(g, c.s);
  }
  return gemm_launch(g, c.s);
}

// in_split: `in` holds split rows -> the pre-split LDS-DMA GEMM on the weight's ".S" copy (decoder 1x1 convs with Cin, Cout % 32 == 0)
int conv1(Ctx& c, const float* in, float* out, int rows, int Cin, int Cout, const std::string& key, const float* res, int in_split = 0) {
  ++c.seq;
  GemmParams g;
  g.A = in; g.lda = Cin; g.B = c.h->p(key + ".weight"); g.ldb = Cin; g.C = out; g.ldc = Cout;
  g.M = rows; g.N = Cout; g.K = Cin; g.bias = c.h->p(key + ".bias");
  g.res = res; g.ldres = Cout;
  if (in_split) {
    g.B = c.h->p(key + ".weight.S");
    return gemm2_launch(g, c.s);
  }
  return gemm_launch(g, c.s);
}
static bool has_split_1x1(const Ctx& c, const std::string& key) {
  static const int on = getenv("RGM_VAE_SPLIT_1X1") ? atoi(getenv("RGM_VAE_SPLIT_1X1")) : 1;   // 0: the fp32-operand kernel (A/B runs)
  return on && c.split && c.h->slots.count(key + ".weight.S") != 0;
}

// ResnetBlock (taming model.py:78-137): out = shortcut(x) + conv2(swish(norm2(b))), b = conv1(swish(norm1(x))).
// t1 is scratch for the normalised maps; hbuf (Cin != Cout only) holds conv2's output and may alias b; out may alias x when
// Cin == Cout.  st1/st2 keep the GroupNorm statistics (the backward pass reads them together with x and b).
int resnet(Ctx& c, const float* x, float* b, float* hbuf, float* out, float* t1, int H, int Cin, int Cout, const std::string& key,
           float* st1 = nullptr, float* st2 = nullptr) {
  const int P = H * H;
  RGM_TRY(group_norm(c, x, t1, P, Cin, key + "norm1", 1, c.split, st1));
  if (c.split && !st2 && Cin == Cout) {   // plain decode: norm2 + swish inside conv1's launch where it qualifies (the saving decode keeps raw b)
    const std::string n2 = key + "norm2";
    int fused = 0;
    RGM_TRY(conv3(c, t1, b, H, Cin, Cout, key + "conv1", 0, nullptr, 1, &n2, &fused));
    if (fused) return conv3(c, b, out, H, Cout, Cout, key + "conv2", 0, x, 1, nullptr, nullptr, c.final_split);   // b holds swish(norm2(conv1(.))) as split rows
  } else {
    RGM_TRY(conv3(c, t1, b, H, Cin, Cout, key + "conv1", 0, nullptr, c.split));
  }
  RGM_TRY(group_norm(c, b, t1, P, Cout, key + "norm2", 1, c.split, st2));
  if (Cin == Cout) return conv3(c, t1, out, H, Cout, Cout, key + "conv2", 0, x, c.split, nullptr, nullptr, c.split ? c.final_split : 0);  // out = conv2(.) + x
  RGM_TRY(conv3(c, t1, hbuf, H, Cout, Cout, key + "conv2", 0, nullptr, c.split));
  return conv1(c, x, out, c.M * P, Cin, Cout, key + "nin_shortcut", hbuf);                  // out = nin(x) + h
}
// the three-buffer in-place form the plain decode / encode schedules use: cur <- block(cur)
int resnet(Ctx& c, float*& cur, float*& t1, float*& t2, int H, int Cin, int Cout, const std::string& key) {
  if (Cin == Cout) return resnet(c, cur, t2, nullptr, cur, t1, H, Cin, Cout, key);
  if (has_split_1x1(c, key + "nin_shortcut")) {
    // pre-split arithmetic: the shortcut runs on the LDS-DMA GEMM too.  norm1's pass over x also leaves x as split rows (t2), the
    // shortcut turns them into nin(x) in x's own buffer, and conv2 adds that in place: three buffers, no extra pass over x.
    const int P = H * H;
    RGM_TRY(group_norm(c, cur, t1, P, Cin, key + "norm1", 1, 1, nullptr, t2));
    RGM_TRY(conv1(c, t2, cur, c.M * P, Cin, Cout, key + "nin_shortcut", nullptr, 1));            // cur <- nin(x)
    const std::string n2 = key + "norm2";
    int fused = 0;
    RGM_TRY(conv3(c, t1, t2, H, Cin, Cout, key + "conv1", 0, nullptr, 1, &n2, &fused));            // t2 <- conv1(swish(norm1(x)))
    if (fused) return conv3(c, t2, cur, H, Cout, Cout, key + "conv2", 0, cur, 1);                   // (t2 already swish(norm2(.)) as split rows)
    RGM_TRY(group_norm(c, t2, t1, P, Cout, key + "norm2", 1, 1));
    return conv3(c, t1, cur, H, Cout, Cout, key + "conv2", 0, cur, 1);                              // cur <- conv2(.) + nin(x)
  }
  RGM_TRY(resnet(c, cur, t2, t2, t1, t1, H, Cin, Cout, key));   // conv2 has consumed t1 before nin writes it
  std::swap(cur, t1);
  return RGM_OK;
}

// AttnBlock (taming model.py:140-192) on the 256 tokens of a 16x16 map, single head of width C: cur <- cur + proj(attn(norm(cur)))
// (out may alias x; q, k, v and the softmax stay in the plan's buffers, where the backward pass finds them)
int attn_block(Ctx& c, const float* cur, float* out, float* t1, const std::string& a, int C, float* st = nullptr) {
  const int M = c.M, rows = M * 256;
  hipStream_t s = c.s;
  const int sp = has_split_1x1(c, a + "q") ? 1 : 0;   // pre-split arithmetic: the four 1x1 convs on the LDS-DMA GEMM
  RGM_TRY(group_norm(c, cur, t1, 256, C, a + "norm", 0, sp, st));
  RGM_TRY(conv1(c, t1, c.p.q, rows, C, C, a + "q", nullptr, sp));
  RGM_TRY(conv1(c, t1, c.p.k, rows, C, C, a + "k", nullptr, sp));
  RGM_TRY(conv1(c, t1, c.p.v, rows, C, C, a + "v", nullptr, sp));
  GemmParams g;  // scores[m] = q[m] . k[m]^T * C^-0.5
  g.A = c.p.q; g.lda = C; g.sA = 256LL * C; g.B = c.p.k; g.ldb = C; g.sB = 256LL * C;
  g.C = c.p.sc; g.ldc = 256; g.sC = 256LL * 256; g.M = 256; g.N = 256; g.K = C; g.batch = M;
  g.alpha = 1.0f / sqrtf((float)C);
  RGM_TRY(gemm_launch(g, s));
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, c.p.sc, rows, 256);
  RGM_LAUNCH_CHECK();
  RGM_TRY(transpose_launch(c.p.v, c.p.vt, 256, C, 256, M, s));
  GemmParams o;  // o[m] = p[m] . v[m]  (B^T form: vt[m] is [C][256])
  o.A = c.p.sc; o.lda = 256; o.sA = 256LL * 256; o.B = c.p.vt; o.ldb = 256; o.sB = 256LL * C;
  o.C = t1; o.ldc = C; o.sC = 256LL * C; o.M = 256; o.N = C; o.K = 256; o.batch = M;
  RGM_TRY(gemm_launch(o, s));
  if (sp) {   // vt is free once p.v has been formed: the attention output as split rows
    RGM_TRY(split_rows_launch(t1, c.p.vt, rows, C, C, C, s));
    return conv1(c, c.p.vt, out, rows, C, C, a + "proj_out", cur, 1);
  }
  return conv1(c, t1, out, rows, C, C, a + "proj_out", cur);  // x + proj_out(o)
}
}  // namespace

// GroupNorm + swish of a ResnetBlock's conv1 output inside the conv launch (0 = never; 1 = where the launch qualifies, the default; 2 = the
// same with every tile on the fallback path -- raw rows + gn_fixup_kernel -- for its test).  *prev (optional): the previous mode.
extern "C" int rgm_set_gn_fuse(int mode, int* prev) {
  RGM_REQUIRE(mode >= 0 && mode <= 2, "set_gn_fuse: %d (0 / 1 / 2)", mode);
  if (prev) *prev = g_gn_fuse;
  g_gn_fuse = mode;
  return RGM_OK;
}
extern "C" long long rgm_gn_fused_launches(void) { return g_gn_fused_launches; }
// tiles of fused GroupNorm launches that gave up waiting for their image's other tiles and took the raw-row fallback since the last reset
// (synchronises the device; reset != 0 zeroes the counter afterwards).  0 on an idle device; forced-fallback launches (rgm_set_gn_fuse(2))
// count every tile.  -1 on a HIP error.
extern "C" long long rgm_gn_fallback_tiles(int reset) {
  if (!g_gn_fallbacks) return 0;
  unsigned long long v = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(&v, g_gn_fallbacks, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (reset && hipMemset(g_gn_fallbacks, 0, sizeof(v)) != hipSuccess) return -1;
  return (long long)v;
}

extern "C" size_t rgm_vae_workspace_bytes(const rgm_vae* h, int M) {
  if (!h || M <= 0) return 0;
  return vplan(M, nullptr).bytes;
}

// in: element (tile m = s*Nb + n, channel c, pitch i, time j) at in[n*n_stride + s*s_stride + c*sc + i*si + j*sj] * in_scale
// conv_out (+ bias, scatter into the roll, optional uint8 quantise): one launcher for the plain and the saving decode
// stats (optional): t1 is the RAW input of norm_out and (mean, rstd) of its M x 32 groups are in `stats`: the tiled kernel normalises as it stages
static int conv_out_launch(const rgm_vae* h, const float* t1, float* roll, uint8_t* u8, int M, int Nb, int Tt, float thr, hipStream_t s,
                           const float* stats = nullptr) {
  static const int tiled = getenv("RGM_CONV_OUT_TILED") ? atoi(getenv("RGM_CONV_OUT_TILED")) : 1;   // 0: the one-row kernel (A/B)
  if (stats)
    hipLaunchKernelGGL(vae_conv_out_tiled_kernel<true>, dim3(M * 64), dim3(256), 0, s, t1, h->p("decoder.conv_out.weight"),
                       h->p("decoder.conv_out.bias"), roll, u8, M, Nb, Tt, thr, stats, h->p("decoder.norm_out.weight"), h->p("decoder.norm_out.bias"));
  else if (tiled)
    hipLaunchKernelGGL(vae_conv_out_tiled_kernel<false>, dim3(M * 64), dim3(256), 0, s, t1, h->p("decoder.conv_out.weight"),
                       h->p("decoder.conv_out.bias"), roll, u8, M, Nb, Tt, thr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
  else
    hipLaunchKernelGGL(vae_conv_out_kernel, dim3(M * 128), dim3(256), 0, s, t1, h->p("decoder.conv_out.weight"),
                       h->p("decoder.conv_out.bias"), roll, u8, M, Nb, Tt, thr);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

static int decode_impl(rgm_vae* h, const float* in, int Nb, int S, long long n_stride, long long s_stride, long long sc,
                       long long si, long long sj, float in_scale, float* roll, uint8_t* u8, float thr, void* ws,
                       size_t ws_bytes, hipStream_t s) {
  RGM_REQUIRE(h && in && (roll || u8) && Nb > 0 && S > 0, "vae_decode: bad arguments");
  if (rgm_vae_missing_params(h) != 0) {
    std::string miss;
    for (auto& kv : h->slots)
      if (!kv.second.set && kv.second.group == 0 && miss.size() < 200) miss += kv.first + " ";
    set_error("vae_decode: %d parameters not set: %s", rgm_vae_missing_params(h), miss.c_str());
    return RGM_ERR_STATE;
  }
  const int M = Nb * S;
  Ctx c{h, vplan(M, ws), M, s, rgm_get_gemm_precision() == 2 ? 1 : 0};
  if (!ws || c.p.bytes > ws_bytes) {
    set_error("vae_decode: workspace %zu bytes < required %zu", ws_bytes, c.p.bytes);
    return RGM_ERR_WORKSPACE;
  }
  const std::string d = "decoder.";
  hipLaunchKernelGGL(vae_gather_pq_kernel, dim3(cdiv(M * 256, 256)), dim3(256), 0, s, in, h->p("post_quant_conv.weight"),
                     h->p("post_quant_conv.bias"), c.p.pq, M, Nb, n_stride, s_stride, sc, si, sj, in_scale);
  RGM_LAUNCH_CHECK();
  float *cur = c.p.b0, *t1 = c.p.b1, *t2 = c.p.b2;
  int C = 512;
  {
    const long long tot = (long long)M * 256 * C;
    hipLaunchKernelGGL(vae_conv_in_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, c.p.pq, h->p(d + "conv_in.weight"),
                       h->p(d + "conv_in.bias"), cur, M, C);
    RGM_LAUNCH_CHECK();
  }
  RGM_TRY(resnet(c, cur, t1, t2, 16, C, C, d + "mid.block_1."));
  RGM_TRY(attn_block(c, cur, cur, t1, d + "mid.attn_1.", C));
  RGM_TRY(resnet(c, cur, t1, t2, 16, C, C, d + "mid.block_2."));
  int H = 16;
  for (int lvl = 3; lvl >= 0; --lvl) {
    const int bo = h->ch * CH_MULT[lvl];
    static const int presplit_ups = getenv("RGM_UPS_PRESPLIT") ? atoi(getenv("RGM_UPS_PRESPLIT")) : 1;   // 0: the split_rows pass (A/B runs)
    bool ups_in_split = false;
    for (int ib = 0; ib < 3; ++ib) {
      // the last block of a level that is followed by an upsampling conv has that conv as its ONLY consumer: it writes split rows straight away
      c.final_split = (ib == 2 && lvl != 0 && c.split && presplit_ups && C == bo) ? 1 : 0;
      ups_in_split = c.final_split != 0;
      RGM_TRY(resnet(c, cur, t1, t2, H, C, bo, d + "up." + std::to_string(lvl) + ".block." + std::to_string(ib) + "."));
      c.final_split = 0;
      C = bo;
    }
    if (lvl != 0) {
      H *= 2;
      if (c.split && ups_in_split) {
        RGM_TRY(conv3(c, cur, t1, H, C, C, d + "up." + std::to_string(lvl) + ".upsample.conv", 1, nullptr, 1));
      } else if (c.split) {   // the upsample conv reads the raw residual stream: one HBM pass turns it into split rows
        RGM_TRY(split_rows_launch(cur, t2, (long long)c.M * (H / 2) * (H / 2), C, C, C, s));
        RGM_TRY(conv3(c, t2, t1, H, C, C, d + "up." + std::to_string(lvl) + ".upsample.conv", 1, nullptr, 1));
      } else {
        RGM_TRY(conv3(c, cur, t1, H, C, C, d + "up." + std::to_string(lvl) + ".upsample.conv", 1, nullptr));
      }
      std::swap(cur, t1);
    }
  }
  // norm_out + swish inside conv_out's staging (g_gn_fuse, C = 128, the tiled kernel): no pass over the largest tensor of the decode
  static const int tiled = getenv("RGM_CONV_OUT_TILED") ? atoi(getenv("RGM_CONV_OUT_TILED")) : 1;
  static const int out_fuse = getenv("RGM_NORM_OUT_FUSE") ? atoi(getenv("RGM_NORM_OUT_FUSE")) : 1;   // 0: the separate pass (A/B runs)
  if (g_gn_fuse && out_fuse && tiled && C == 128) {
    RGM_TRY(group_norm(c, cur, nullptr, 128 * 128, C, d + "norm_out", 1));
    ++g_gn_fused_launches;
    return conv_out_launch(h, cur, roll, u8, M, Nb, S * 128, thr, s, c.p.stats);
  }
  RGM_TRY(group_norm(c, cur, t1, 128 * 128, C, d + "norm_out", 1));
  return conv_out_launch(h, t1, roll, u8, M, Nb, S * 128, thr, s);
}

extern "C" int rgm_vae_decode(rgm_vae* h, const float* z, float* out, int M, void* ws, size_t ws_bytes, void* stream) {
  // z (M,4,16,16) [c][pitch i][time j]; out (M,3,128,128): every tile is its own "sample" with one segment
  return decode_impl(h, z, M, 1, 4 * 256, 0, 256, 16, 1, 1.0f, out, nullptr, -0.95f, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int rgm_vae_decode_latent(rgm_vae* h, const float* latent, float inv_scale, float* roll, uint8_t* roll_u8,
                                     float threshold, int N, int H, void* ws, size_t ws_bytes, void* stream) {
  // latent (N,4,H,16) [c][time][pitch]; square s covers time rows 16s..16s+15 (gaussian_diffusion.py:1351-1355)
  RGM_REQUIRE(H > 0 && H % 16 == 0, "vae_decode_latent: H=%d must be a multiple of 16", H);
  return decode_impl(h, latent, N, H / 16, 4LL * H * 16, 256, (long long)H * 16, 1, 16, inv_scale, roll, roll_u8, threshold, ws,
                     ws_bytes, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------- decode with saves + input-gradient pass
// The workspace of the grad entry points = the plain plan (scratch, attention tensors) followed by a tape: the input x and
// the conv1 output b of every ResnetBlock, the attention / upsample inputs and every GroupNorm's (mean, rstd).  Its
// layout is a pure function of M, so the backward call re-derives every pointer from the workspace it is handed.
namespace {
struct GNode {
  int kind = 0;            // 0 ResnetBlock, 1 AttnBlock, 2 Upsample conv
  const float* x = nullptr;
  float *b = nullptr, *out = nullptr, *st1 = nullptr, *st2 = nullptr;
  int H = 0, Cin = 0, Cout = 0;   // H: spatial size of the node's output
  std::string key;
};
struct GradPlan {
  VPlan p;
  std::vector<GNode> nodes;
  float *x0 = nullptr, *st_out = nullptr, *sums = nullptr;
  const float* x_last = nullptr;
  float *at[7] = {}, *as[3] = {};   // attention-backward scratch: token-sized / score-sized
  size_t bytes = 0;
};
GradPlan gradplan(const rgm_vae* h, int M, void* ws) {
  GradPlan g;
  g.p = vplan(M, ws);
  char* base = (char*)ws;
  size_t off = g.p.bytes;
  auto take = [&](size_t floats) {
    char* r = base + off;
    off += align_up(floats * sizeof(float), 256);
    return (float*)r;
  };
  const std::string d = "decoder.";
  int C = 512, H = 16;
  g.x0 = take((size_t)M * 256 * C);
  const float* cur = g.x0;
  auto res = [&](const std::string& key, int Cin, int Cout) {
    GNode n;
    n.kind = 0; n.x = cur; n.H = H; n.Cin = Cin; n.Cout = Cout; n.key = key;
    n.b = take((size_t)M * H * H * Cout);
    n.out = take((size_t)M * H * H * Cout);
    n.st1 = take((size_t)M * 64);
    n.st2 = take((size_t)M * 64);
    cur = n.out;
    g.nodes.push_back(n);
  };
  res(d + "mid.block_1.", C, C);
  {
    GNode n;
    n.kind = 1; n.x = cur; n.H = 16; n.Cin = n.Cout = C; n.key = d + "mid.attn_1.";
    n.out = take((size_t)M * 256 * C);
    n.st1 = take((size_t)M * 64);
    cur = n.out;
    g.nodes.push_back(n);
  }
  res(d + "mid.block_2.", C, C);
  for (int lvl = 3; lvl >= 0; --lvl) {
    const int bo = h->ch * CH_MULT[lvl];
    for (int ib = 0; ib < 3; ++ib) {
      res(d + "up." + std::to_string(lvl) + ".block." + std::to_string(ib) + ".", C, bo);
      C = bo;
    }
    if (lvl != 0) {
      H *= 2;
      GNode n;
      n.kind = 2; n.x = cur; n.H = H; n.Cin = n.Cout = C; n.key = d + "up." + std::to_string(lvl) + ".upsample.conv";
      n.out = take((size_t)M * H * H * C);
      cur = n.out;
      g.nodes.push_back(n);
    }
  }
  g.x_last = cur;
  g.st_out = take((size_t)M * 64);
  g.sums = take((size_t)M * 64);
  for (auto& a : g.at) a = take((size_t)M * 256 * 512);
  for (auto& a : g.as) a = take((size_t)M * 256 * 256);
  g.bytes = off;
  return g;
}

int check_grad_call(rgm_vae* h, const GradPlan& g, void* ws, size_t ws_bytes, const char* who) {
  if (!h->grad_ready) {
    set_error("%s: call rgm_vae_enable_grad after loading the decoder parameters", who);
    return RGM_ERR_STATE;
  }
  if (!ws || g.bytes > ws_bytes) {
    set_error("%s: workspace %zu bytes < required %zu (rgm_vae_grad_workspace_bytes)", who, ws_bytes, g.bytes);
    return RGM_ERR_WORKSPACE;
  }
  return RGM_OK;
}

// dx = GroupNorm(+swish) backward of dz at input x (+ add); out may alias dz
int group_norm_bwd(Ctx& c, const GradPlan& g, const float* x, const float* dz, const float* stats, float* out, int P, int C,
                   const std::string& key, int swish, const float* add, int out_split = 0) {
  const float *ga = c.h->p(key + ".weight"), *be = c.h->p(key + ".bias");
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(GN_CHUNKS, c.M), dim3(256), 0, c.s, x, dz, stats, ga, be, c.p.part, P, C, GN_CHUNKS, swish);
  RGM_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(cdiv(c.M * 32, 256)), dim3(256), 0, c.s, c.p.part, g.sums, c.M, GN_CHUNKS,
                     (double)P * (C / 32));
  RGM_LAUNCH_CHECK();
  const long long total4 = (long long)c.M * P * C / 4;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, c.s, x, dz, stats, g.sums, ga, be, add,
                     out, total4, P, C, swish, out_split);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// input gradient of a 3x3 conv (forward Cin -> Cout at H x H): dx[M*H*H, Cin] = conv3x3(dy, mirrored W^T)
// (in_split: dy holds split-row pixels -> pre-split LDS-DMA kernel on the ".TS" weights)
int conv3_dgrad(Ctx& c, const float* dy, float* dx, int H, int Cin, int Cout, const std::string& key, int in_split = 0) {
  GemmParams g;
  g.A = dy; g.B = c.h->gp(key + (in_split ? ".weight.TS" : ".weight.T")); g.ldb = 9 * Cout; g.C = dx; g.ldc = Cin;
  g.M = c.M * H * H; g.N = Cin; g.K = 9 * Cout; g.lda = Cout;
  g.aload = 1; g.H = H; g.W = H; g.Cin = Cout; g.logH = ilog2(H); g.logW = ilog2(H); g.ups = 0;
  return in_split ? gemm2_launch(g, c.s) : gemm_launch(g, c.s);
}
// input gradient of a 1x1 conv: dx[rows, Cin] = dy[rows, Cout] . W (+ res)
int conv1_dgrad(Ctx& c, const float* dy, float* dx, int rows, int Cin, int Cout, const std::string& key, const float* res) {
  GemmParams g;
  g.A = dy; g.lda = Cout; g.B = c.h->gp(key + ".weight.T"); g.ldb = Cout; g.C = dx; g.ldc = Cin;
  g.M = rows; g.N = Cin; g.K = Cout; g.res = res; g.ldres = Cin;
  return gemm_launch(g, c.s);
}
int bgemm(Ctx& c, const float* A, int lda, long long sA, const float* B, int ldb, long long sB, float* C, int ldc, long long sC,
          int M, int N, int K) {
  GemmParams g;
  g.A = A; g.lda = lda; g.sA = sA; g.B = B; g.ldb = ldb; g.sB = sB; g.C = C; g.ldc = ldc; g.sC = sC;
  g.M = M; g.N = N; g.K = K; g.batch = c.M;
  return gemm_launch(g, c.s);
}
}  // namespace

extern "C" size_t rgm_vae_grad_workspace_bytes(const rgm_vae* h, int M) {
  if (!h || M <= 0) return 0;
  return gradplan(h, M, nullptr).bytes;
}

// Same decode as decode_impl (same kernels in the same order, hence the same roll) with every tensor the backward needs kept.
static int decode_save_impl(rgm_vae* h, const float* in, int Nb, int S, long long n_stride, long long s_stride, long long sc,
                            long long si, long long sj, float in_scale, float* roll, void* ws, size_t ws_bytes, hipStream_t s) {
  RGM_REQUIRE(h && in && roll && Nb > 0 && S > 0, "vae_decode_save: bad arguments");
  const int M = Nb * S;
  GradPlan g = gradplan(h, M, ws);
  RGM_TRY(check_grad_call(h, g, ws, ws_bytes, "vae_decode_save"));
  Ctx c{h, g.p, M, s, rgm_get_gemm_precision() == 2 ? 1 : 0};
  hipLaunchKernelGGL(vae_gather_pq_kernel, dim3(cdiv(M * 256, 256)), dim3(256), 0, s, in, h->p("post_quant_conv.weight"),
                     h->p("post_quant_conv.bias"), c.p.pq, M, Nb, n_stride, s_stride, sc, si, sj, in_scale);
  RGM_LAUNCH_CHECK();
  {
    const long long tot = (long long)M * 256 * 512;
    hipLaunchKernelGGL(vae_conv_in_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, c.p.pq, h->p("decoder.conv_in.weight"),
                       h->p("decoder.conv_in.bias"), g.x0, M, 512);
    RGM_LAUNCH_CHECK();
  }
  float *t1 = c.p.b1, *t2 = c.p.b2;
  for (const GNode& n : g.nodes) {
    if (n.kind == 0) {
      RGM_TRY(resnet(c, n.x, n.b, t2, n.out, t1, n.H, n.Cin, n.Cout, n.key, n.st1, n.st2));
    } else if (n.kind == 1) {
      RGM_TRY(attn_block(c, n.x, n.out, t1, n.key, n.Cin, n.st1));
    } else if (c.split) {
      RGM_TRY(split_rows_launch(n.x, t2, (long long)M * (n.H / 2) * (n.H / 2), n.Cin, n.Cin, n.Cin, s));
      RGM_TRY(conv3(c, t2, n.out, n.H, n.Cin, n.Cin, n.key, 1, nullptr, 1));
    } else {
      RGM_TRY(conv3(c, n.x, n.out, n.H, n.Cin, n.Cin, n.key, 1, nullptr));
    }
  }
  static const int tiled = getenv("RGM_CONV_OUT_TILED") ? atoi(getenv("RGM_CONV_OUT_TILED")) : 1;
  static const int out_fuse = getenv("RGM_NORM_OUT_FUSE") ? atoi(getenv("RGM_NORM_OUT_FUSE")) : 1;
  if (g_gn_fuse && out_fuse && tiled) {   // as in the plain decode: statistics only (kept for the backward), norm_out inside conv_out's staging
    RGM_TRY(group_norm(c, g.x_last, nullptr, 128 * 128, 128, "decoder.norm_out", 1, 0, g.st_out));
    return conv_out_launch(h, g.x_last, roll, nullptr, M, Nb, S * 128, -0.95f, s, g.st_out);
  }
  RGM_TRY(group_norm(c, g.x_last, t1, 128 * 128, 128, "decoder.norm_out", 1, 0, g.st_out));
  return conv_out_launch(h, t1, roll, nullptr, M, Nb, S * 128, -0.95f, s);   // the same kernel as the plain decode: identical rolls
}

// d_in = (d roll / d in)^T d_roll for the decode decode_save_impl ran on this workspace
static int decode_vjp_impl(rgm_vae* h, const float* d_roll, int Nb, int S, long long n_stride, long long s_stride, long long sc,
                           long long si, long long sj, float in_scale, float* d_in, void* ws, size_t ws_bytes, hipStream_t s) {
  RGM_REQUIRE(h && d_roll && d_in && Nb > 0 && S > 0, "vae_decode_vjp: bad arguments");
  const int M = Nb * S;
  GradPlan g = gradplan(h, M, ws);
  RGM_TRY(check_grad_call(h, g, ws, ws_bytes, "vae_decode_vjp"));
  Ctx c{h, g.p, M, s, rgm_get_gemm_precision() == 2 ? 1 : 0};
  float *gr = c.p.b0, *s1 = c.p.b1, *s2 = c.p.b2;
  const int sp = c.split;   // pre-split mode: the conv input gradients read split rows (one extra HBM pass where no producer writes them)
  hipLaunchKernelGGL(vae_conv_out_bwd_kernel, dim3((unsigned)(M * 16384LL * 32 / 256)), dim3(256), 0, s, d_roll,
                     h->p("decoder.conv_out.weight"), gr, M, Nb, S * 128);
  RGM_LAUNCH_CHECK();
  RGM_TRY(group_norm_bwd(c, g, g.x_last, gr, g.st_out, gr, 128 * 128, 128, "decoder.norm_out", 1, nullptr));
  for (auto it = g.nodes.rbegin(); it != g.nodes.rend(); ++it) {
    const GNode& n = *it;
    const int P = n.H * n.H;
    if (n.kind == 2) {            // conv on the nearest-upsampled map: full-resolution input gradient, then fold the 2x2 blocks
      if (sp) RGM_TRY(split_rows_launch(gr, s2, (long long)M * P, n.Cin, n.Cin, n.Cin, s));
      RGM_TRY(conv3_dgrad(c, sp ? s2 : gr, s1, n.H, n.Cin, n.Cin, n.key, sp));
      const int Hh = n.H / 2, q = n.Cin / 4;
      const long long total4 = (long long)M * Hh * Hh * q;
      hipLaunchKernelGGL(sumpool2_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, s1, s2, total4, Hh, q);
      RGM_LAUNCH_CHECK();
      std::swap(gr, s2);
    } else if (n.kind == 0) {
      if (sp) RGM_TRY(split_rows_launch(gr, s2, (long long)M * P, n.Cout, n.Cout, n.Cout, s));
      RGM_TRY(conv3_dgrad(c, sp ? s2 : gr, s1, n.H, n.Cout, n.Cout, n.key + "conv2", sp));                    // d swish(norm2(b))
      RGM_TRY(group_norm_bwd(c, g, n.b, s1, n.st2, sp ? s2 : s1, P, n.Cout, n.key + "norm2", 1, nullptr, sp)); // d b
      RGM_TRY(conv3_dgrad(c, sp ? s2 : s1, sp ? s1 : s2, n.H, n.Cin, n.Cout, n.key + "conv1", sp));          // d swish(norm1(x))
      float* da = sp ? s1 : s2;
      if (n.Cin == n.Cout) {
        RGM_TRY(group_norm_bwd(c, g, n.x, da, n.st1, da, P, n.Cin, n.key + "norm1", 1, gr));                   // + identity shortcut
      } else {
        RGM_TRY(group_norm_bwd(c, g, n.x, da, n.st1, da, P, n.Cin, n.key + "norm1", 1, nullptr));
        RGM_TRY(conv1_dgrad(c, gr, da, M * P, n.Cin, n.Cout, n.key + "nin_shortcut", da));                     // + nin_shortcut^T
      }
      if (sp) std::swap(gr, s1); else std::swap(gr, s2);
    } else {                      // AttnBlock: out = x + proj(softmax(alpha q k^T) v), q/k/v = 1x1 convs of norm(x)
      const int C = n.Cin, rows = M * 256;
      const long long tk = 256LL * C, sq = 256LL * 256;
      const float *q = c.p.q, *k = c.p.k, *v = c.p.v, *Pm = c.p.sc;
      float *d_o = g.at[0], *d_oT = g.at[1], *dv = g.at[2], *kT = g.at[3], *dq = g.at[4], *qT = g.at[5], *dk = g.at[6];
      float *dS = g.as[0], *Pt = g.as[1], *dSt = g.as[2];
      RGM_TRY(conv1_dgrad(c, gr, d_o, rows, C, C, n.key + "proj_out", nullptr));
      RGM_TRY(bgemm(c, d_o, C, tk, v, C, tk, dS, 256, sq, 256, 256, C));                  // dP[i][j] = d_o[i] . v[j]
      hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, Pm, dS, rows, 256, 1.0f / sqrtf((float)C));
      RGM_LAUNCH_CHECK();
      RGM_TRY(transpose_launch(Pm, Pt, 256, 256, 256, M, s));
      RGM_TRY(transpose_launch(d_o, d_oT, 256, C, 256, M, s));
      RGM_TRY(bgemm(c, Pt, 256, sq, d_oT, 256, tk, dv, C, tk, 256, C, 256));              // dv[j] = sum_i P[i][j] d_o[i]
      RGM_TRY(transpose_launch(k, kT, 256, C, 256, M, s));
      RGM_TRY(bgemm(c, dS, 256, sq, kT, 256, tk, dq, C, tk, 256, C, 256));                // dq[i] = sum_j dS[i][j] k[j]
      RGM_TRY(transpose_launch(dS, dSt, 256, 256, 256, M, s));
      RGM_TRY(transpose_launch(q, qT, 256, C, 256, M, s));
      RGM_TRY(bgemm(c, dSt, 256, sq, qT, 256, tk, dk, C, tk, 256, C, 256));               // dk[j] = sum_i dS[i][j] q[i]
      RGM_TRY(conv1_dgrad(c, dq, s1, rows, C, C, n.key + "q", nullptr));
      RGM_TRY(conv1_dgrad(c, dk, s1, rows, C, C, n.key + "k", s1));
      RGM_TRY(conv1_dgrad(c, dv, s1, rows, C, C, n.key + "v", s1));
      RGM_TRY(group_norm_bwd(c, g, n.x, s1, n.st1, s1, 256, C, n.key + "norm", 0, gr));
      std::swap(gr, s1);
    }
  }
  hipLaunchKernelGGL(vae_conv_in_bwd_kernel, dim3(cdiv(M * 256, 4)), dim3(256), 0, s, gr, h->p("decoder.conv_in.weight"),
                     h->p("post_quant_conv.weight"), d_in, M, 512, Nb, n_stride, s_stride, sc, si, sj, in_scale);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// decode_latent keeping what rgm_vae_decode_latent_vjp needs in `ws` (rgm_vae_grad_workspace_bytes(h, N*H/16) bytes, left
// untouched by the caller between the two calls).  roll (N,3,128,8H) is the same as rgm_vae_decode_latent's.
extern "C" int rgm_vae_decode_latent_save(rgm_vae* h, const float* latent, float inv_scale, float* roll, int N, int H, void* ws,
                                          size_t ws_bytes, void* stream) {
  RGM_REQUIRE(H > 0 && H % 16 == 0, "vae_decode_latent_save: H=%d must be a multiple of 16", H);
  return decode_save_impl(h, latent, N, H / 16, 4LL * H * 16, 256, (long long)H * 16, 1, 16, inv_scale, roll, ws, ws_bytes,
                          (hipStream_t)stream);
}
// d_latent (N,4,H,16) = (d roll / d latent)^T d_roll, d_roll (N,3,128,8H): reference autograd through _decode
// (gaussian_diffusion.py:1347-1358) -> AutoencoderKL.decode, as the dps_rule branch of condition_mean differentiates it (:425-433)
extern "C" int rgm_vae_decode_latent_vjp(rgm_vae* h, const float* d_roll, float inv_scale, float* d_latent, int N, int H, void* ws,
                                         size_t ws_bytes, void* stream) {
  RGM_REQUIRE(H > 0 && H % 16 == 0, "vae_decode_latent_vjp: H=%d must be a multiple of 16", H);
  return decode_vjp_impl(h, d_roll, N, H / 16, 4LL * H * 16, 256, (long long)H * 16, 1, 16, inv_scale, d_latent, ws, ws_bytes,
                         (hipStream_t)stream);
}

extern "C" int rgm_quantise_roll(const float* roll, uint8_t* out_u8, int B, int T, float threshold, void* stream) {
  RGM_REQUIRE(roll && out_u8 && B > 0 && T > 0, "quantise_roll: bad arguments");
  const long long total = (long long)B * 128 * T * 3;
  hipLaunchKernelGGL(quantise_roll_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, roll, out_u8, B, T, threshold);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

// Encoder forward (taming Encoder.forward, model.py:404-433, + quant_conv): x (M,3,128,128) -> moments (M,8,16,16)
// (mean = channels 0..3, logvar = 4..7; AutoencoderKL.encode_save, klvae_pedal.py:61-68 with range_fix=False).
extern "C" int rgm_vae_encode(rgm_vae* h, const float* x, float* moments, int M, void* ws, size_t ws_bytes, void* stream) {
  RGM_REQUIRE(h && x && moments && M > 0, "vae_encode: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (rgm_vae_encoder_missing_params(h) != 0) {
    std::string miss;
    for (auto& kv : h->slots)
      if (!kv.second.set && kv.second.group == 1 && miss.size() < 200) miss += kv.first + " ";
    set_error("vae_encode: %d encoder parameters not set: %s", rgm_vae_encoder_missing_params(h), miss.c_str());
    return RGM_ERR_STATE;
  }
  Ctx c{h, vplan(M, ws), M, s, 0};
  if (!ws || c.p.bytes > ws_bytes) {
    set_error("vae_encode: workspace %zu bytes < required %zu", ws_bytes, c.p.bytes);
    return RGM_ERR_WORKSPACE;
  }
  const std::string e = "encoder.";
  float *cur = c.p.b0, *t1 = c.p.b1, *t2 = c.p.b2;
  int C = h->ch, H = 128;
  {
    const long long tot = (long long)M * 16384 * C;
    hipLaunchKernelGGL(vae_enc_conv_in_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, x, h->p(e + "conv_in.weight"),
                       h->p(e + "conv_in.bias"), cur, M, C);
    RGM_LAUNCH_CHECK();
  }
  for (int lvl = 0; lvl < 4; ++lvl) {
    const int eo = h->ch * CH_MULT[lvl];
    for (int ib = 0; ib < 2; ++ib) {
      RGM_TRY(resnet(c, cur, t1, t2, H, C, eo, e + "down." + std::to_string(lvl) + ".block." + std::to_string(ib) + "."));
      C = eo;
    }
    if (lvl != 3) {   // Downsample: zero pad (0,1,0,1) + 3x3 stride 2 (model.py:56-75) = the ups == -1 loader of gemm.hip
      H /= 2;
      RGM_TRY(conv3(c, cur, t1, H, C, C, e + "down." + std::to_string(lvl) + ".downsample.conv", -1, nullptr));
      std::swap(cur, t1);
    }
  }
  RGM_TRY(resnet(c, cur, t1, t2, 16, C, C, e + "mid.block_1."));
  RGM_TRY(attn_block(c, cur, cur, t1, e + "mid.attn_1.", C));
  RGM_TRY(resnet(c, cur, t1, t2, 16, C, C, e + "mid.block_2."));
  RGM_TRY(group_norm(c, cur, t1, 256, C, e + "norm_out", 1));
  RGM_TRY(conv3(c, t1, t2, 16, C, 8, e + "conv_out", 0, nullptr));          // [M*256][8]
  hipLaunchKernelGGL(vae_enc_quant_kernel, dim3(cdiv(M * 8 * 256, 256)), dim3(256), 0, s, t2, h->p("quant_conv.weight"),
                     h->p("quant_conv.bias"), moments, M);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
