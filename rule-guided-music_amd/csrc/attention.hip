// attention.hip -- RotaryAttention core of DiTBlockRotary, fp32, one workgroup per (sample, head).
//
// Reference: guided_diffusion/dit.py:263-277 (qkv split, rotary on q and k, F.scaled_dot_product_attention
// with scale head_dim**-0.5, no mask, dropout 0) and rotary-embedding-torch==0.3.2
// rotate_queries_or_keys (interleaved pairs on the first `2*rot_half` channels, position = token index).
//
// Shape regime: T <= 288 tokens (256 patches, +1 cls token for classifiers, 128 for DiffCollage half
// windows), head_dim 72 (XL) or 64 (S/B).  The whole K and V of one head fit the 160 KiB LDS of a
// CDNA4 CU (256 x 76 x 4 B + 256 x 72 x 4 B = 148 KiB), so the kernel is single-pass: no online
// softmax, no rescaling -- each wave keeps the full score strip of its 32 queries in registers.
//
//   * 8 waves per workgroup; wave w owns query tiles w, w+8, ... (32 queries each);
//   * scores are computed TRANSPOSED, S^T = K . Q^T with v_mfma_f32_32x32x2_f32: the C layout then
//     has query = lane&31 (a column) and keys spread over the 16 accumulator registers, so the
//     softmax max / sum over keys is in-lane plus ONE cross-half shuffle, and the probabilities are
//     already in the B-operand layout of O^T = V^T . P^T (register s of a key tile is exactly the
//     k-slot pair the MFMA wants) -- P never leaves the register file;
//   * K rows are padded to hd+4 floats: 16-byte slot stride 19 (or 17) is odd, so the 16-lane
//     ds_read_b128 groups of the A-operand reads are conflict-free; V^T operands are ds_read_b32 with
//     consecutive lanes on consecutive channels;
//   * rotary (cos/sin table lookup) and the softmax scale are applied while K goes to LDS / Q to VGPRs,
//     so q, k are read from HBM exactly once and no rotated copy is ever written back.
#include "common.h"

namespace rgm {

// exp(x) for x <= 0 without ocml's range-check compares (each costs an SGPR-pair mask; 128 of them
// per strip spill the scalar file): exp2 of the product x*log2(e) carried in two floats, ~1-2 ulp.
__device__ __forceinline__ float exp_neg(float x) {
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f;
  x = fmaxf(x, -104.0f);   // masked scores are -inf: (-inf)*c + inf would be NaN below; 2^-150 flushes to exactly 0
  const float t = x * L2E_HI;
  float r = fmaf(x, L2E_HI, -t);
  r = fmaf(x, L2E_LO, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.693147182464599609375f, e);
}

template <int HD, int NKT>
__global__ __launch_bounds__(512) void rotary_attention_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                               const float* __restrict__ cos_tab,
                                                               const float* __restrict__ sin_tab, int T, int heads,
                                                               int rot_half, float* __restrict__ lse, int out_split) {
  constexpr int HDP = HD + 4;          // padded K row (floats)
  constexpr int KB = HD / 8;           // k-blocks of 8 in QK^T
  constexpr int DT = (HD + 31) / 32;   // 32-wide output-channel tiles
  constexpr int TP = NKT * 32;         // padded key count
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;                    // [TP][HD]  (+ slack: reads of channels >= HD run into Ks)
  float* Ks = smem + TP * HD;          // [TP][HDP]

  const int n = blockIdx.x / heads, head = blockIdx.x - n * heads;
  const int D = heads * HD, D3 = 3 * D;
  const float* base = qkv + (long long)n * T * D3 + head * HD;
  const int tid = threadIdx.x;
  const int R = 2 * rot_half;

  // ---- stage K (rotated) and V into LDS; zero the padded key rows
  constexpr int CPR = HD / 4;  // float4 chunks per row
  for (int c = tid; c < TP * CPR; c += 512) {
    const int key = c / CPR, ch = c - key * CPR, d0 = ch * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (key < T) {
      const float* rowp = base + (long long)key * D3;
      kv = *reinterpret_cast<const float4*>(rowp + D + d0);
      vv = *reinterpret_cast<const float4*>(rowp + 2 * D + d0);
      if (d0 < R) {
        const int pi = key * rot_half + (d0 >> 1);
        const float c0 = cos_tab[pi], s0 = sin_tab[pi], c1 = cos_tab[pi + 1], s1 = sin_tab[pi + 1];
        const float x0 = kv.x, x1 = kv.y, x2 = kv.z, x3 = kv.w;
        kv.x = x0 * c0 - x1 * s0;
        kv.y = x1 * c0 + x0 * s0;
        kv.z = x2 * c1 - x3 * s1;
        kv.w = x3 * c1 + x2 * s1;
      }
    }
    *reinterpret_cast<float4*>(Ks + key * HDP + d0) = kv;
    *reinterpret_cast<float4*>(Vs + key * HD + d0) = vv;
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const float scale = rsqrtf((float)HD);
  const int nqt = (T + 31) >> 5;

  for (int qt = wave; qt < nqt; qt += 8) {
    const int q = qt * 32 + l31;
    const int qc = min(q, T - 1);
    // ---- Q fragment: lane (query l31, half hh) holds Q[q][8j+4hh .. +3], rotated and pre-scaled
    f32x4 qf[KB];
    {
      const float* qp = base + (long long)qc * D3;
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int d0 = 8 * j + 4 * hh;
        float4 v = *reinterpret_cast<const float4*>(qp + d0);
        if (d0 < R) {
          const int pi = qc * rot_half + (d0 >> 1);
          const float c0 = cos_tab[pi], s0 = sin_tab[pi], c1 = cos_tab[pi + 1], s1 = sin_tab[pi + 1];
          const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
          v.x = x0 * c0 - x1 * s0;
          v.y = x1 * c0 + x0 * s0;
          v.z = x2 * c1 - x3 * s1;
          v.w = x3 * c1 + x2 * s1;
        }
        qf[j][0] = v.x * scale;
        qf[j][1] = v.y * scale;
        qf[j][2] = v.z * scale;
        qf[j][3] = v.w * scale;
      }
    }
    // ---- S^T[key][query] = K . Q^T
    f32x16 sacc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.f;
      const float* kp = Ks + (kt * 32 + l31) * HDP + 4 * hh;
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 8 * j);
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[j][s], sacc[kt], 0, 0, 0);
      }
    }
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
        const float pv = exp_neg(sacc[kt][e] - mx);
        sacc[kt][e] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (lse && hh == 0 && q < T) lse[((long long)n * heads + head) * T + q] = mx + logf(sum);   // saved for the backward
    // ---- O^T[d][query] = V^T . P^T ; A operand = V[key][d] with d = lane&31, B operand = P registers
    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float* vp = Vs + (kt * 32 + (s & 3) + 8 * (s >> 2) + 4 * hh) * HD + l31;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[dt * 32], sacc[kt][s], oacc[dt], 0, 0, 0);
        if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the 384 V reads from being hoisted (spills)
      }
    }
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
              typedef split_t bf16x4 __attribute__((ext_vector_type(4)));
              bf16x4 hi, lo;
              hi[0] = (split_t)ov.x; hi[1] = (split_t)ov.y; hi[2] = (split_t)ov.z; hi[3] = (split_t)ov.w;
              lo[0] = (split_t)(ov.x - (float)hi[0]); lo[1] = (split_t)(ov.y - (float)hi[1]);
              lo[2] = (split_t)(ov.z - (float)hi[2]); lo[3] = (split_t)(ov.w - (float)hi[3]);
              split_t* rp = reinterpret_cast<split_t*>(o + ((long long)n * T + q) * D);
              const int si = split_idx(head * HD + d);      // d % 4 == 0: the 4 elements share a 32-block
              *reinterpret_cast<bf16x4*>(rp + si) = hi;
              *reinterpret_cast<bf16x4*>(rp + si + 32) = lo;
            } else {
              *reinterpret_cast<float4*>(op + d) = ov;
            }
          }
        }
    }
  }
}

template <int HD, int NKT>
static int launch_attn(const float* qkv, float* o, const float* ct, const float* st, int N, int T, int heads,
                       int rot_half, float* lse, int out_split, hipStream_t s) {
  constexpr int TP = NKT * 32;
  // V strip + K strip; channel reads of the last (partial) 32-wide tile run past a V row into the
  // next row / the K strip, which is finite data feeding discarded accumulator rows only.
  const size_t lds = attn_lds_one_per_cu((size_t)(TP * HD + TP * (HD + 4)) * sizeof(float));   // one workgroup per CU (common.h)
  auto kern = rotary_attention_kernel<HD, NKT>;
  static bool prepared = false;
  if (!prepared) RGM_TRY(attn_prepare_kernel(kern, 512, lds, "rotary_attention_kernel"));
  prepared = true;
  hipLaunchKernelGGL(kern, dim3(N * heads), dim3(512), lds, s, qkv, o, ct, st, T, heads, rot_half, lse, out_split);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

int rotary_attention_launch(const float* qkv, float* o, const float* cos_tab, const float* sin_tab, int N, int T,
                            int heads, int hd, int rot_half, hipStream_t s, float* lse, int out_split) {
  RGM_REQUIRE(N > 0 && T > 0 && T <= 288, "attention: T=%d out of range (1..288)", T);
  RGM_REQUIRE((2 * rot_half) % 4 == 0 && 2 * rot_half <= hd, "attention: rotary dim %d", 2 * rot_half);
  const int nkt = (T + 31) / 32;
  if (hd == 72) {
    if (nkt <= 4) return launch_attn<72, 4>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    if (nkt <= 8) return launch_attn<72, 8>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    set_error("attention: head_dim 72 supports T <= 256 (K+V of one head must fit the 160 KiB LDS), got %d", T);
    return RGM_ERR_INVALID;
  }
  if (hd == 64) {
    if (nkt <= 4) return launch_attn<64, 4>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    if (nkt <= 5) return launch_attn<64, 5>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    if (nkt <= 8) return launch_attn<64, 8>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
    return launch_attn<64, 9>(qkv, o, cos_tab, sin_tab, N, T, heads, rot_half, lse, out_split, s);
  }
  set_error("attention: head_dim %d not supported (64, 72)", hd);
  return RGM_ERR_INVALID;
}

}  // namespace rgm
