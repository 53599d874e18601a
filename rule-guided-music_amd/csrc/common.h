// common.h -- shared helpers for librgm_hip (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/rgm.h"

namespace rgm {

void set_error(const char* fmt, ...);

#define RGM_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      rgm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return RGM_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

#define RGM_REQUIRE(cond, ...)                   \
  do {                                           \
    if (!(cond)) {                               \
      rgm::set_error(__VA_ARGS__);               \
      return RGM_ERR_INVALID;                    \
    }                                            \
  } while (0)

#define RGM_LAUNCH_CHECK()                                                     \
  do {                                                                         \
    hipError_t _e = hipGetLastError();                                         \
    if (_e != hipSuccess) {                                                    \
      rgm::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return RGM_ERR_HIP;                                                      \
    }                                                                          \
  } while (0)

#define RGM_TRY(expr)          \
  do {                         \
    int _s = (expr);           \
    if (_s != RGM_OK) return _s; \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Element type of the hi / lo halves of the "x3" arithmetic (x ~= hi + lo, three MFMAs per product, fp32 accumulate) -- a BUILD constant:
//   default        __bf16   : 8 significand bits each, 2^-16 per product, fp32's exponent range (rounds 1-3; modes "bf16x3", "bf16x3_presplit")
//   -DRGM_SPLIT_F16 _Float16 : 11 significand bits each, 2^-22 per product at the same MFMA rate (v_mfma_f32_32x32x16_f16), but fp16's
//                              range: |x| > 65504 overflows, lo halves of |x| < 0.125 are subnormal (still <= 3e-8 absolute).  Evaluated in
//                              round 4 as librgm_hip_f16.so (make F16=1); rgm_split_dtype() tells which build is loaded.
#ifdef RGM_SPLIT_F16
typedef _Float16 split_t;
#define RGM_MFMA_SPLIT_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 split_t;
#define RGM_MFMA_SPLIT_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif

// "Split row" format (pre-split bf16x3 operands, gemm2.hip): a logical fp32 row of K elements keeps its K*4 bytes;
// every block of 32 elements becomes one 128-byte line [32 bf16 hi | 32 bf16 lo] (x ~= hi + lo, round-to-nearest each).
// Returns the bf16 index of element `col`'s hi part within the row; its lo part is 32 further.
__host__ __device__ __forceinline__ int split_idx(int col) { return ((col >> 5) << 6) + (col & 31); }

// Timing experiments (elimination runs that produce WRONG results: no DMA, no MFMA, no stores ...) are compiled in only
// with -DRGM_EXPERIMENTS (make EXTRA=-DRGM_EXPERIMENTS) and then selected by environment variables; a normal build
// ignores those variables.
#ifdef RGM_EXPERIMENTS
#define RGM_EXP_ENV(name) (getenv(name) ? atoi(getenv(name)) : 0)
#else
#define RGM_EXP_ENV(name) 0
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) -- guarantees static register indexing where
// `#pragma unroll` on a large nest is only a request
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Device-coherent ("write-through", sc1) stores: what a workgroup of a persistent launch publishes for ANOTHER workgroup of the same
// launch (chain.hip).  The bytes leave the XCD's L2 when vmcnt retires the store -- no cache write-back (an agent-scope release fence
// writes the whole dirty L2 back: MI355X_MICROARCH.md "publish-large", 8.2 against 3.0 us per 64 KB), so the hand-off is
//   sc1 stores -> s_waitcnt vmcnt(0) -> barrier -> relaxed agent-scope counter add      (producer)
//   relaxed agent-scope poll -> agent-scope acquire fence (L1 invalidate) -> barrier -> plain loads / LDS-DMA   (consumer).
// 16 bytes per lane: narrower sc1 stores are one fabric write each (a dwordx2 costs 2.7x per byte) -- 8-byte pieces of split rows are
// paired over two lanes first (store_split4_pair_sc1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void store16_sc1(void* p, const f32x4& v) {
  // s_nop: a store of more than 8 bytes still reads its data registers for a wait state or two after issue, and the compiler's hazard
  // recogniser does not see a store inside an asm statement -- without it the next VALU write of those registers corrupted the last
  // four lanes of every 16 (found on the attention output: rows 12-15 / 28-31 of a query tile, run-to-run different).
  // No "memory" clobber: a clobber pins every load behind the previous store (the row items of chain.hip walked one round trip per
  // 16-byte chunk that way); volatile asm statements keep their order among themselves, and where an epilogue counts on the order of its
  // LDS-DMA loads against these stores (gemm2_body.h staged_dma_res) it fences the loads with compiler barriers of its own.
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v));
}
typedef split_t split_x4 __attribute__((ext_vector_type(4)));
// A lane holds 4 consecutive elements (columns col .. col + 3, col % 4 == 0) of a split row as hi / lo halves; its partner lane
// (lane ^ XOR) holds the other 4 of the same 8-aligned group.  The lane with (col & 4) == 0 stores the 8 hi halves (16 bytes), the other
// the 8 lo halves: two 16-byte coherent stores instead of four 8-byte ones.  Both lanes of a pair must call (shuffle inside).
template <int XOR>
__device__ __forceinline__ void store_split4_pair_sc1(split_t* rowp, int col, const split_x4& hi, const split_x4& lo) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const bool even = (col & 4) == 0;
  const u32x2 mine_hi = __builtin_bit_cast(u32x2, hi), mine_lo = __builtin_bit_cast(u32x2, lo);
  const u32x2 send = even ? mine_lo : mine_hi;           // the even lane needs the partner's hi, the odd lane the partner's lo
  u32x2 recv;
  recv[0] = (unsigned)__shfl_xor((int)send[0], XOR, 64);
  recv[1] = (unsigned)__shfl_xor((int)send[1], XOR, 64);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 out = even ? u32x4{mine_hi[0], mine_hi[1], recv[0], recv[1]} : u32x4{recv[0], recv[1], mine_lo[0], mine_lo[1]};
  split_t* dst = rowp + split_idx(col & ~7) + (even ? 0 : 32);
  store16_sc1(dst, __builtin_bit_cast(f32x4, out));
}

// The same pairing for ordinary (non-coherent) stores, neighbouring lanes (lane ^ 1, exchanged with a DPP quad permute: no LDS traffic).
// An 8-byte store costs an issue slot like a 16-byte one, and a write-bound epilogue is store-ISSUE-bound: the 256x256 tile's epilogue takes
// 33.8 k cycles with split-row output as 2 x 8 bytes per lane against 25.5 k for the same bytes as fp32 rows (tools/gemm_stamp.py, SPLIT=1).
// NT: non-temporal (the one-wave-per-SIMD kernels' 57-76 MB outputs).  Both lanes of a pair must call.
template <bool NT, int XOR = 1>
__device__ __forceinline__ void store_split4_pair(split_t* rowp, int col, const split_x4& hi, const split_x4& lo) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const bool even = (col & 4) == 0;
  const u32x2 mine_hi = __builtin_bit_cast(u32x2, hi), mine_lo = __builtin_bit_cast(u32x2, lo);
  const u32x2 send = even ? mine_lo : mine_hi;
  u32x2 recv;
  if constexpr (XOR == 1) {
    recv[0] = (unsigned)__builtin_amdgcn_mov_dpp((int)send[0], 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]: lane ^ 1
    recv[1] = (unsigned)__builtin_amdgcn_mov_dpp((int)send[1], 0xB1, 0xF, 0xF, true);
  } else {                                                                              // partner in another DPP row (gemm144: lane ^ 16)
    recv[0] = (unsigned)__shfl_xor((int)send[0], XOR, 64);
    recv[1] = (unsigned)__shfl_xor((int)send[1], XOR, 64);
  }
  const u32x4 out = even ? u32x4{mine_hi[0], mine_hi[1], recv[0], recv[1]} : u32x4{recv[0], recv[1], mine_lo[0], mine_lo[1]};
  f32x4* dst = reinterpret_cast<f32x4*>(rowp + split_idx(col & ~7) + (even ? 0 : 32));
  if constexpr (NT) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, out), dst);
  else *dst = __builtin_bit_cast(f32x4, out);
}

// The row kernels (LayerNorm bodies, attention output, the 144-column GEMM): paired unless the build says otherwise (-DRGM_ROW_SPLIT_PAIR=0: the
// A/B build of tools/ab_lib.sh)
#ifndef RGM_ROW_SPLIT_PAIR
#define RGM_ROW_SPLIT_PAIR 1
#endif
template <int XOR>
__device__ __forceinline__ void store_split4_maybe_pair(split_t* rowp, int col, const split_x4& hi, const split_x4& lo) {
  if constexpr (RGM_ROW_SPLIT_PAIR) {
    store_split4_pair<false, XOR>(rowp, col, hi, lo);
  } else {
    *reinterpret_cast<split_x4*>(rowp + split_idx(col)) = hi;
    *reinterpret_cast<split_x4*>(rowp + split_idx(col) + 32) = lo;
  }
}

// 16-byte load from GLOBAL memory, said so: a pointer that reaches a device function through a descriptor in memory (chain.hip) is a
// generic pointer to the compiler -- flat loads, which count on lgkmcnt as well and serialise against the LDS traffic around them
__device__ __forceinline__ float4 ldg16(const float* q) {
  const f32x4 v = *(const __attribute__((address_space(1))) f32x4*)q;
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float2 ldg8(const float* q) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = *(const __attribute__((address_space(1))) f32x2_t*)q;
  return make_float2(v[0], v[1]);
}
__device__ __forceinline__ float ldg4(const float* q) {
  return *(const __attribute__((address_space(1))) float*)q;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// the same function through v_exp_f32 + v_rcp_f32 (~2e-7 relative): the decoder's GroupNorm-swish, which since round 4 also runs inside conv
// epilogues (64-256 values per lane on the tile's critical path).  Every swish of the VAE uses THIS form, fused or not, so that a decode does
// not depend on which launches qualified for the fusion.  x -> -inf: x * rcp(inf) = -0; x -> +inf: x * rcp(1) = x.
__device__ __forceinline__ float silu_fast_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950409f));
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float c = 0.7978845608028654f;  // sqrt(2/pi)
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}
// same function as gelu_tanh_f written as x * sigmoid(2u): one v_exp_f32 + one v_rcp_f32 instead of libm tanhf
// (~3e-7 relative; used by the pre-split GEMM epilogue where 64 of them per lane sit on the tile's critical path)
__device__ __forceinline__ float gelu_tanh_fast_f(float x) {
  const float w = x * __builtin_fmaf(x * x, -0.0713548163f * 1.4426950409f, -1.5957691216f * 1.4426950409f);   // -2u * log2(e)
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float c = 0.7978845608028654f;
  const float u = c * (x + 0.044715f * x * x * x);
  const float th = tanhf(u);
  return 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * c * (1.0f + 3.0f * 0.044715f * x * x);
}
__device__ __forceinline__ float silu_grad_f(float x) {
  const float sg = 1.0f / (1.0f + expf(-x));
  return sg * (1.0f + x * (1.0f - sg));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------
// GEMM  C = epi(alpha * A . B^T): see gemm.hip
// ---------------------------------------------------------------------------------------------
struct GemmParams {
  const float* A = nullptr;
  const float* B = nullptr;
  float* C = nullptr;
  int M = 0, N = 0, K = 0;
  int lda = 0, ldb = 0, ldc = 0;
  long long sA = 0, sB = 0, sC = 0;  // batch strides in elements (blockIdx.z)
  int batch = 1;
  // epilogue
  const float* bias = nullptr;   // [N]
  long long sBias = 0;
  int act = 0;                   // 0 none, 1 silu, 2 gelu(tanh)
  float alpha = 1.0f;
  const float* gate = nullptr;   // gate[(row / rows_per_gate) * gate_ld + col]
  int gate_ld = 0, rows_per_gate = 1;
  const float* res = nullptr;    // res[row * ldres + col] (+ batch * sRes); may alias C
  int ldres = 0;
  long long sRes = 0;
  // act 3 / 4: multiply by gelu_tanh'(aux) / silu'(aux) (backward through an activation); aux[row*ldaux + col]
  const float* aux = nullptr;
  int ldaux = 0;
  long long sAux = 0;
  // implicit-GEMM 3x3 conv A operand (aload == 1): A is NHWC [img][Hin][Win][Cin], M = imgs*H*W
  int aload = 0;
  int H = 0, W = 0, Cin = 0, logH = 0, logW = 0, ups = 0;
  // K order of the implicit conv: 0 = (tap, cin) as the weights are repacked [cout][9][cin]; 1 = (cin / 32, tap, cin % 32) -- the nine
  // taps of a 32-channel block are consecutive K-tiles (gemm2.hip PIPE 5 only; weights [cout][cin/32][9][32]).  The three nearest-x2 upsampling
  // convs keep order 0 (measured better); the ALOAD 2 loader also implements order 1 for them, behind RGM_CONV_KMAJOR_UPS (experiments)
  int conv_kmajor = 0;
  // tile override for experiments: 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 32x128
  int tile = 0;
  // gemm2 heuristic: a launch of comparable size runs beside this one on a second stream (dit.hip: the blocks as two half batches), so a
  // partial round of one-workgroup-per-CU tiles is not idle time -- the 256x256 kernel is taken from far fewer tiles (gemm2_launch)
  int co_sched = 0;
  // gemm2 raster: row-tiles per column sweep of the XCD-contiguous grouped raster (0 = 8).  The one-wave-per-SIMD kernels set it to
  // ~sqrt(tiles per XCD): an XCD's tiles then form a near-square block and its L2 fetches the fewest operand panels
  int raster_group = 0;
  // one-wave-per-SIMD tiles: ordinary instead of non-temporal output stores (set by launch2 from RGM_ST_PLAIN: bit 0 fp32 rows, bit 1 split rows)
  int st_plain = 0;
  // gemm2 epilogues: split rows as lane pairs, 16 bytes per lane (set by launch2: RGM_SPLIT_PAIR, default by kernel family)
  int split_pair = 0;
  // loader/consumer tiles (PIPE 4): L2 prefetch distance of the B panel in K-tiles, 0 = off (set by launch2: RGM_P4_PF, and only where the scratch KiB fits)
  int pf_kt = 0;
  // K-slice launches of gemm2_launch (fc2 of a DiT block): the reduce kernel holds whole output rows, so it can also write the NEXT
  // adaLN-LayerNorm of that row -- LN(row, ln_eps) * (1 + ln_scale) + ln_shift, the arithmetic of ln_mod_kernel, to ln_out -- and
  // save that kernel's launch and its read of the row.  Optional: *ln_done (host) is set to 1 only when the launch took this route.
  float* ln_out = nullptr;
  const float* ln_shift = nullptr;
  const float* ln_scale = nullptr;
  int ln_mod_ld = 0, ln_rows_per_batch = 1, ln_out_split = 0;
  float ln_eps = 0.f;
  int* ln_done = nullptr;
  // arithmetic: -1 library default (rgm_set_gemm_precision), 0 fp32 MFMA, 1 bf16x3 split
  int prec = -1;
  // gemm2 (pre-split operands): write C in split-row format (N bf16 hi | N bf16 lo per row) for the next GEMM
  int out_split = 0;
  // gemm2, plain epilogue (bias + activation, nothing read per row) with out_split: a SECOND output -- C2 (row stride ldc2) takes the split rows
  // of act(v) while C takes the fp32 rows of the pre-activation v = alpha * acc + bias (the value-and-gradient chain of a classifier keeps
  // fc1's pre-activation for the backward and feeds fc2 its GELU: one launch instead of GEMM + an elementwise pass)
  float* C2 = nullptr;
  int ldc2 = 0;
  // gemm2, 128-row tiles: also write per-tile GroupNorm partial sums of the OUTPUT (after bias / residual):
  // stats[(tile_m * (N / stats_gw) + col / stats_gw) * 2 + {0, 1}] = sum, sum of squares over the tile's rows of the stats_gw
  // (4, 8 or 16) channels of group col / stats_gw -- fp64, fixed-order tree, no atomics (deterministic, and the same to 1e-16
  // whatever tile shape the heuristic picked: results stay independent of the batch size); vae.hip group_norm consumes them
  double* stats = nullptr;
  int stats_gw = 0;
  // gemm2, one-wave-per-SIMD conv tiles (PIPE 5, ALOAD 2) with `stats`: GroupNorm + swish of the OUTPUT applied in this launch's epilogue
  // (the conv1 -> norm2 pair of a ResnetBlock: taming model.py:117-126).  A tile first leaves its partial sums in `stats`, arrives at the
  // counter of its (image, column tile) -- gn_count[(row tile / gn_tiles) * column tiles + column tile], zeroed by the caller -- and waits
  // (bounded) until all gn_tiles row tiles of the image have arrived; it then sums the image's partials in tile order (the arithmetic of
  // gn_finalize_tiles_kernel), normalises its accumulators, applies swish and writes split rows.  A tile whose wait runs out writes its raw
  // fp32 rows instead and raises gn_fail[row tile * column tiles + column tile]; vae.hip's gn_fixup pass converts such tiles in place.
  unsigned* gn_count = nullptr;
  int* gn_fail = nullptr;
  int gn_tiles = 0;              // row tiles per image (image pixels / tile rows)
  const float* gn_gamma = nullptr;
  const float* gn_beta = nullptr;
  double gn_n = 0.0;             // elements per group: pixels per image x channels per group
  float gn_eps = 0.f;
  int gn_swish = 0;
  int gn_force_fail = 0;         // tests: every tile takes the fallback
  // caller-provided scratch of the deterministic split-K (gemm2_scratch_bytes): the first 4096 bytes are reserved, the rest holds the
  // K slices' partial sums.  nullptr: no K slicing.
  void* sk_ws = nullptr;
  size_t sk_ws_bytes = 0;
};
int gemm_launch(const GemmParams& p, hipStream_t stream);
// gemm2.hip: the same contraction on operands already in split-row format (K bf16 hi | K bf16 lo per row)
int gemm2_launch(const GemmParams& p, hipStream_t stream);
// gemm144.hip: 128x144 tiles (16x16x32 MFMAs) for N % 144 == 0 -- tile 81 of gemm2_launch
bool gemm144_supports(const GemmParams& p);
int gemm144_launch(const GemmParams& p, hipStream_t stream);
constexpr size_t GEMM_SK_FLAG_BYTES = 4096;   // reserved head of the scratch (the split-K partial sums start behind it)
// scratch a caller must provide for GEMMs of up to M rows and N columns to use the deterministic split-K
size_t gemm2_scratch_bytes(int M, int N);
int gemm2_prof_begin(int id, double flops, hipStream_t s);
void gemm2_prof_end(int idx, hipStream_t s);
int split_rows_launch(const float* x, float* out, long long rows, int K, int ld_in, int ld_out, hipStream_t s);
void gemm2_prof(bool on);
void gemm2_prof_reset();
int gemm2_prof_report(int kernel, int* launches, double* total_ms, double* total_flops);
double gemm2_prof_bytes(int kernel);
int big_tiles_mode();   // rgm_set_big_tiles (gemm2.hip)
int big_tiles_min();

// elementwise / reductions (elementwise.hip)
int layernorm_modulate_launch(const float* x, float* out, int M, int D, float eps, const float* weight,
                              const float* bias, const float* shift, const float* scale, int mod_ld,
                              int rows_per_batch, hipStream_t s, int out_split = 0);
int rotary_attention_launch(const float* qkv, float* o, const float* cos_tab, const float* sin_tab, int N,
                            int T, int heads, int hd, int rot_half, hipStream_t s, float* lse = nullptr, int out_split = 0);
// the same forward in bf16x3 arithmetic (attention_x3.hip)
int rotary_attention_x3_launch(const float* qkv, float* o, const float* cos_tab, const float* sin_tab, int N, int T, int heads,
                               int hd, int rot_half, hipStream_t s, float* lse = nullptr, int out_split = 0);
// forward attention by mode: fp32 MFMA in fp32 mode, bf16x3 otherwise (rgm_set_gemm_precision); lse (optional) for the backward
int rotary_attention_fwd(const float* qkv, float* o, const float* cos_tab, const float* sin_tab, int N, int T, int heads, int hd,
                         int rot_half, hipStream_t s, int out_split = 0, float* lse = nullptr);
// attention backward (attention_bwd.hip): dqkv (N*T, 3*heads*hd) from dO, the saved qkv / O / lse
int rotary_attention_bwd_launch(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                                const float* cos_tab, const float* sin_tab, int N, int T, int heads, int hd,
                                int rot_half, hipStream_t s, int osplit = 0);   // osplit: d(qkv) as split rows (a pre-split dgrad GEMM reads them next)
int transpose_launch(const float* in, float* out, int R, int Cc, int out_ld, int batch, hipStream_t s);
int attn_split_mode();   // rgm_set_attn_split (attention_bwd.hip): -1 auto, 0 never, 1 always
constexpr int ATTN_SPLIT_MAX_PAIRS = 96;   // auto: per-tile attention workgroups up to this many (sample, head) pairs (B <= 16 for the 6-head classifiers)

// ---------------------------------------------------------------------------------------------
// ONE workgroup of an attention kernel per CU -- a correctness requirement, not a tuning choice.
// Two workgroups of rotary_attention_x3_kernel<72,4> on one CU (T <= 128: their K / V images take 80 KiB each), started together and
// running the same instruction stream on the same SIMDs, make the one dispatched second (LDS base != 0) read, in lanes 48-63 of one
// wave, the OLD contents of the second destination register of a global_load_dwordx2 two instructions behind its s_waitcnt vmcnt(0)
// (the rotary factors of Q chunk (j = 1, u = 0): v.z = x2*c1 - x3*s1 comes out as -x3*s1, 16 queries of the head wrong by ~0.1;
// tools/ubench/attn_hazard.hip reproduces it standalone, ~1 workgroup in 4000; profiles/r04_attn_hazard_*.txt).  Alone on its CU -- or
// beside a foreign workgroup -- the kernel has never produced a wrong value.  Every attention launcher therefore asks for more than half
// of the 160 KiB LDS and checks with the occupancy API, once per instantiation, that the runtime agrees.
// ---------------------------------------------------------------------------------------------
constexpr size_t ATTN_ONE_PER_CU_LDS = 80 * 1024 + 512;
inline size_t attn_lds_one_per_cu(size_t lds) { return lds < ATTN_ONE_PER_CU_LDS ? ATTN_ONE_PER_CU_LDS : lds; }
// sets the dynamic-LDS limit of `kern` and requires that exactly one workgroup of (threads, lds) fits a CU
template <class Kern>
int attn_prepare_kernel(Kern kern, int threads, size_t lds, const char* what) {
  RGM_REQUIRE(lds >= ATTN_ONE_PER_CU_LDS && lds <= 160 * 1024, "%s: %zu bytes of LDS (one workgroup per CU needs %zu .. %d)", what, lds,
              ATTN_ONE_PER_CU_LDS, 160 * 1024);
  RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  RGM_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds));
  RGM_REQUIRE(per_cu == 1, "%s: %d workgroups of %d threads / %zu bytes of LDS fit a CU; the attention kernels must run one per CU", what,
              per_cu, threads, lds);
  return RGM_OK;
}

}  // namespace rgm
