// capi.hip -- error plumbing and the thin extern "C" wrappers around the building-block kernels.
#include <string.h>
#include "common.h"

namespace rgm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace rgm

using namespace rgm;

extern "C" int rgm_version(void) { return 100; }
extern "C" const char* rgm_last_error(void) { return g_err; }

extern "C" int rgm_gemm(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                        const float* bias, int act, float alpha, const float* gate, int gate_ld, int rows_per_gate,
                        const float* res, int ldres, void* stream) {
  RGM_REQUIRE(A && B && C, "gemm: null operand");
  GemmParams g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.bias = bias; g.act = act; g.alpha = alpha;
  g.gate = gate; g.gate_ld = gate_ld; g.rows_per_gate = rows_per_gate > 0 ? rows_per_gate : 1;
  g.res = res; g.ldres = ldres;
  return gemm_launch(g, (hipStream_t)stream);
}

extern "C" int rgm_gemm_tile(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                             const float* bias, int act, int tile, void* stream) {
  RGM_REQUIRE(A && B && C, "gemm: null operand");
  GemmParams g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act; g.tile = tile & 15; g.prec = (tile >> 4) ? (tile >> 4) - 1 : -1;
  return gemm_launch(g, (hipStream_t)stream);
}

extern "C" int rgm_layernorm_modulate(const float* x, float* out, int M, int D, float eps, const float* weight,
                                      const float* bias, const float* shift, const float* scale, int mod_ld,
                                      int rows_per_batch, void* stream) {
  RGM_REQUIRE(x && out, "layernorm: null tensor");
  return layernorm_modulate_launch(x, out, M, D, eps, weight, bias, shift, scale, mod_ld, rows_per_batch, (hipStream_t)stream);
}

extern "C" int rgm_rotary_attention(const float* qkv, float* o, const float* cos_tab, const float* sin_tab, int N, int T,
                                    int heads, int hd, int rot_half, void* stream) {
  RGM_REQUIRE(qkv && o && cos_tab && sin_tab, "attention: null tensor");
  return rotary_attention_fwd(qkv, o, cos_tab, sin_tab, N, T, heads, hd, rot_half, (hipStream_t)stream);
}

// the same forward, also writing the per-query log-sum-exp (N * heads * T) the backward needs (what the classifiers' saved-activation pass calls)
extern "C" int rgm_rotary_attention_lse(const float* qkv, float* o, float* lse, const float* cos_tab, const float* sin_tab, int N, int T,
                                        int heads, int hd, int rot_half, void* stream) {
  RGM_REQUIRE(qkv && o && lse && cos_tab && sin_tab, "attention: null tensor");
  return rotary_attention_fwd(qkv, o, cos_tab, sin_tab, N, T, heads, hd, rot_half, (hipStream_t)stream, 0, lse);
}

// 0: the x3 modes split into bf16 halves (default build), 1: into fp16 halves (-DRGM_SPLIT_F16 build: librgm_hip_f16.so)
extern "C" int rgm_split_dtype(void) {
#ifdef RGM_SPLIT_F16
  return 1;
#else
  return 0;
#endif
}
