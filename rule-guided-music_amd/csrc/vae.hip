// vae.hip -- taming KL-VAE decoder (f8, z=4ch): weight arena, glue kernels and the decode schedule.
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

// y = act((x - mean) * rstd * gamma + beta), act = swish (1) or identity (0); NHWC float4 pass
__global__ void gn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, long long total4, int P, int C,
                                int swish, int out_split) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int q = C >> 2;
  const int c4 = (int)(i % q);
  const int m = (int)(i / ((long long)q * P));
  const int g = (c4 * 4) / (C >> 5);
  const float mean = stats[(m * 32 + g) * 2], rstd = stats[(m * 32 + g) * 2 + 1];
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[c4], be = reinterpret_cast<const float4*>(beta)[c4];
  float4 o;
  o.x = (v.x - mean) * rstd * ga.x + be.x;
  o.y = (v.y - mean) * rstd * ga.y + be.y;
  o.z = (v.z - mean) * rstd * ga.z + be.z;
  o.w = (v.w - mean) * rstd * ga.w + be.w;
  if (swish) {
    o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w);
  }
  if (out_split) {   // split-row pixels (common.h split_idx) for the pre-split conv GEMM
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, lo;
    hi[0] = (__bf16)o.x; hi[1] = (__bf16)o.y; hi[2] = (__bf16)o.z; hi[3] = (__bf16)o.w;
    lo[0] = (__bf16)(o.x - (float)hi[0]); lo[1] = (__bf16)(o.y - (float)hi[1]);
    lo[2] = (__bf16)(o.z - (float)hi[2]); lo[3] = (__bf16)(o.w - (float)hi[3]);
    __bf16* px = reinterpret_cast<__bf16*>(y + (i / q) * C);
    *reinterpret_cast<bf16x4*>(px + split_idx(c4 * 4)) = hi;
    *reinterpret_cast<bf16x4*>(px + split_idx(c4 * 4) + 32) = lo;
  } else {
    reinterpret_cast<float4*>(y)[i] = o;
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

// conv_out 3x3 (C=128 -> 3) on 128x128; x NHWC (post GN+swish); weights [3][9][128] staged in LDS.
// One thread per pixel; writes roll[n][co][y][s*128 + x] (tile m = s*Nb + n) and optionally the uint8 roll.
__global__ __launch_bounds__(256) void vae_conv_out_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ roll,
                                                           uint8_t* __restrict__ u8, int M, int Nb, int Tt, float thr) {
  constexpr int C = 128;
  __shared__ __attribute__((aligned(16))) float ws[3 * 9 * C];
  for (int i = threadIdx.x; i < 3 * 9 * C; i += 256) ws[i] = w[i];
  __syncthreads();
  const int pix = blockIdx.x * 256 + threadIdx.x;  // over M*128*128 (grid exact)
  const int m = pix >> 14, y = (pix >> 7) & 127, xx0 = pix & 127;
  float a0 = b[0], a1 = b[1], a2 = b[2];
  const float4* xb = reinterpret_cast<const float4*>(x + (long long)m * 16384 * C);
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = y + tap / 3 - 1, xx = xx0 + tap % 3 - 1;
    if ((unsigned)yy >= 128u || (unsigned)xx >= 128u) continue;
    const float4* px = xb + ((yy << 7) + xx) * (C / 4);
    const float4* w0 = reinterpret_cast<const float4*>(ws + (0 * 9 + tap) * C);
    const float4* w1 = reinterpret_cast<const float4*>(ws + (1 * 9 + tap) * C);
    const float4* w2 = reinterpret_cast<const float4*>(ws + (2 * 9 + tap) * C);
#pragma unroll 8
    for (int c = 0; c < C / 4; ++c) {
      const float4 v = px[c];
      const float4 p0 = w0[c], p1 = w1[c], p2 = w2[c];
      a0 += (v.x * p0.x + v.y * p0.y) + (v.z * p0.z + v.w * p0.w);
      a1 += (v.x * p1.x + v.y * p1.y) + (v.z * p1.z + v.w * p1.w);
      a2 += (v.x * p2.x + v.y * p2.y) + (v.z * p2.z + v.w * p2.w);
    }
  }
  const int s = m / Nb, n = m - s * Nb;
  const long long plane = (long long)128 * Tt;
  const long long o = (long long)n * 3 * plane + (long long)y * Tt + s * 128 + xx0;
  const float r[3] = {a0, a1, a2};
#pragma unroll
  for (int co = 0; co < 3; ++co) {
    if (roll) roll[o + co * plane] = r[co];
    if (u8) {  // midi_util.py:59-63, output layout (B,128,T,3)
      float v = r[co] <= thr ? -1.0f : r[co];
      v = fminf(fmaxf((v + 1.0f) * 63.5f, 0.0f), 127.0f);
      u8[((long long)n * 128 + y) * Tt * 3 + (long long)(s * 128 + xx0) * 3 + co] = (uint8_t)v;
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

}  // namespace rgm

using namespace rgm;

struct VSlot {
  size_t off = 0, numel = 0;
  bool set = false;
  int conv3 = 0;  // 1: repack [co][ci][3][3] -> [co][9][ci]
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
};

static void vslot(rgm_vae* h, const std::string& key, size_t numel, int conv3 = 0, int cout = 0, int cin = 0) {
  VSlot s;
  s.off = h->arena_floats;
  s.numel = numel;
  s.conv3 = conv3;
  s.cout = cout;
  s.cin = cin;
  s.group = h->cur_group;
  h->slots[key] = s;
  h->arena_floats += (numel + 3) / 4 * 4;
  if (numel > h->stage_floats) h->stage_floats = numel;
  if (conv3 && cin % 32 == 0 && h->cur_group == 0) {   // the encoder convs stay on the on-the-fly kernel
    VSlot d;
    d.off = h->arena_floats;
    d.numel = numel;
    d.derived = true;
    d.set = true;
    h->slots[key + ".S"] = d;
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
    vslot(h, p + "nin_shortcut.weight", (size_t)cout * cin);
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
    vslot(h, d + "mid.attn_1." + nm + ".weight", (size_t)bi * bi);
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
    RGM_CHECK_HIP(hipStreamSynchronize(0));
  } else {
    RGM_CHECK_HIP(hipMemcpy(h->arena + s.off, dptr, numel * sizeof(float), hipMemcpyDeviceToDevice));
  }
  s.set = true;
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

namespace {
constexpr int GN_CHUNKS = 16;
struct VPlan {
  float *pq, *b0, *b1, *b2, *q, *k, *v, *vt, *sc, *stats;
  double* part;
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
  p.bytes = off;
  return p;
}

struct Ctx {
  rgm_vae* h;
  VPlan p;
  int M;
  hipStream_t s;
  int split = 0;   // bf16x3_presplit: GroupNorm writes split rows, the 3x3 convs run on gemm2.hip
};

int group_norm(Ctx& c, const float* x, float* y, int P, int C, const std::string& key, int swish, int out_split = 0) {
  hipLaunchKernelGGL(gn_partial_kernel, dim3(GN_CHUNKS, c.M), dim3(256), 0, c.s, x, c.p.part, P, C, GN_CHUNKS);
  RGM_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(c.M * 32, 256)), dim3(256), 0, c.s, c.p.part, c.p.stats, c.M, GN_CHUNKS,
                     (double)P * (C / 32), 1e-6f);
  RGM_LAUNCH_CHECK();
  const long long total4 = (long long)c.M * P * C / 4;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, c.s, x, y, c.p.stats,
                     c.h->p(key + ".weight"), c.h->p(key + ".bias"), total4, P, C, swish, out_split);
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
int conv3(Ctx& c, const float* in, float* out, int H, int Cin, int Cout, const std::string& key, int ups, const float* res,
          int in_split = 0) {
  GemmParams g;
  g.A = in; g.B = c.h->p(key + (in_split ? ".weight.S" : ".weight")); g.ldb = 9 * Cin; g.C = out; g.ldc = Cout;
  g.M = c.M * H * H; g.N = Cout; g.K = 9 * Cin; g.lda = Cin;
  g.bias = c.h->p(key + ".bias");
  g.res = res; g.ldres = Cout;
  g.aload = 1; g.H = H; g.W = H; g.Cin = Cin; g.logH = ilog2(H); g.logW = ilog2(H); g.ups = ups;
  if (in_split) g.tile = RGM_EXP_ENV("RGM_CONV_TILE");   // 0 = gemm2's heuristic (timing experiments: common.h)
  return in_split ? gemm2_launch(g, c.s) : gemm_launch(g, c.s);
}

int conv1(Ctx& c, const float* in, float* out, int rows, int Cin, int Cout, const std::string& key, const float* res) {
  GemmParams g;
  g.A = in; g.lda = Cin; g.B = c.h->p(key + ".weight"); g.ldb = Cin; g.C = out; g.ldc = Cout;
  g.M = rows; g.N = Cout; g.K = Cin; g.bias = c.h->p(key + ".bias");
  g.res = res; g.ldres = Cout;
  return gemm_launch(g, c.s);
}

// x (cur) -> result buffer; uses the two other rotating buffers as scratch. Returns which buffer holds the result.
int resnet(Ctx& c, float*& cur, float*& t1, float*& t2, int H, int Cin, int Cout, const std::string& key) {
  const int P = H * H;
  RGM_TRY(group_norm(c, cur, t1, P, Cin, key + "norm1", 1, c.split));
  RGM_TRY(conv3(c, t1, t2, H, Cin, Cout, key + "conv1", 0, nullptr, c.split));
  RGM_TRY(group_norm(c, t2, t1, P, Cout, key + "norm2", 1, c.split));
  if (Cin == Cout) {
    RGM_TRY(conv3(c, t1, cur, H, Cout, Cout, key + "conv2", 0, cur, c.split));  // in place: out = conv2(.) + x
  } else {
    RGM_TRY(conv3(c, t1, t2, H, Cout, Cout, key + "conv2", 0, nullptr, c.split));
    RGM_TRY(conv1(c, cur, t1, c.M * P, Cin, Cout, key + "nin_shortcut", t2));  // t1 = nin(x) + h
    std::swap(cur, t1);
  }
  return RGM_OK;
}

// AttnBlock (taming model.py:140-192) on the 256 tokens of a 16x16 map, single head of width C: cur <- cur + proj(attn(norm(cur)))
int attn_block(Ctx& c, float* cur, float* t1, const std::string& a, int C) {
  const int M = c.M, rows = M * 256;
  hipStream_t s = c.s;
  RGM_TRY(group_norm(c, cur, t1, 256, C, a + "norm", 0));
  RGM_TRY(conv1(c, t1, c.p.q, rows, C, C, a + "q", nullptr));
  RGM_TRY(conv1(c, t1, c.p.k, rows, C, C, a + "k", nullptr));
  RGM_TRY(conv1(c, t1, c.p.v, rows, C, C, a + "v", nullptr));
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
  return conv1(c, t1, cur, rows, C, C, a + "proj_out", cur);  // x + proj_out(o), in place
}
}  // namespace

extern "C" size_t rgm_vae_workspace_bytes(const rgm_vae* h, int M) {
  if (!h || M <= 0) return 0;
  return vplan(M, nullptr).bytes;
}

// in: element (tile m = s*Nb + n, channel c, pitch i, time j) at in[n*n_stride + s*s_stride + c*sc + i*si + j*sj] * in_scale
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
  RGM_TRY(attn_block(c, cur, t1, d + "mid.attn_1.", C));
  RGM_TRY(resnet(c, cur, t1, t2, 16, C, C, d + "mid.block_2."));
  int H = 16;
  for (int lvl = 3; lvl >= 0; --lvl) {
    const int bo = h->ch * CH_MULT[lvl];
    for (int ib = 0; ib < 3; ++ib) {
      RGM_TRY(resnet(c, cur, t1, t2, H, C, bo, d + "up." + std::to_string(lvl) + ".block." + std::to_string(ib) + "."));
      C = bo;
    }
    if (lvl != 0) {
      H *= 2;
      if (c.split) {   // the upsample conv reads the raw residual stream: one HBM pass turns it into split rows
        RGM_TRY(split_rows_launch(cur, t2, (long long)c.M * (H / 2) * (H / 2), C, C, C, s));
        RGM_TRY(conv3(c, t2, t1, H, C, C, d + "up." + std::to_string(lvl) + ".upsample.conv", 1, nullptr, 1));
      } else {
        RGM_TRY(conv3(c, cur, t1, H, C, C, d + "up." + std::to_string(lvl) + ".upsample.conv", 1, nullptr));
      }
      std::swap(cur, t1);
    }
  }
  RGM_TRY(group_norm(c, cur, t1, 128 * 128, C, d + "norm_out", 1));
  hipLaunchKernelGGL(vae_conv_out_kernel, dim3(M * 64), dim3(256), 0, s, t1, h->p(d + "conv_out.weight"), h->p(d + "conv_out.bias"),
                     roll, u8, M, Nb, S * 128, thr);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
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
  RGM_TRY(attn_block(c, cur, t1, e + "mid.attn_1.", C));
  RGM_TRY(resnet(c, cur, t1, t2, 16, C, C, e + "mid.block_2."));
  RGM_TRY(group_norm(c, cur, t1, 256, C, e + "norm_out", 1));
  RGM_TRY(conv3(c, t1, t2, 16, C, 8, e + "conv_out", 0, nullptr));          // [M*256][8]
  hipLaunchKernelGGL(vae_enc_quant_kernel, dim3(cdiv(M * 8 * 256, 256)), dim3(256), 0, s, t2, h->p("quant_conv.weight"),
                     h->p("quant_conv.bias"), moments, M);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
