/* rgm.h -- C ABI of librgm_hip.so: the MI355X (gfx950) native layer under the reference's Python API.
 *
 * The reference (yjhuangcd/rule-guided-music) has no FFI: its hot path sits behind Python call
 * signatures.  The drop-in keeps those signatures in rule-guided-music_amd/ and
 * binds THIS header with ctypes (rule-guided-music_amd/rgm/native.py); INTEGRATION.md shows the stub.
 * Each entry point cites the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; tensors are dense, row-major,
 *     float32 unless stated; all work is enqueued on the hipStream_t passed in (as void*), nothing
 *     synchronises the device, nothing allocates except *_create / *_set_param (weight arena);
 *   - return 0 (RGM_OK) or a negative rgm_status; never throws; rgm_last_error() gives the text
 *     (thread-local, valid until the next call on that thread);
 *   - handles are bound to the device current at *_create, not thread-safe (one per rank);
 *   - workspace is caller-provided memory of at least *_workspace_bytes(), 256-byte aligned.
 */
#ifndef RGM_H
#define RGM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  RGM_OK = 0,
  RGM_ERR_INVALID = -1,   /* bad argument / shape / unknown key */
  RGM_ERR_HIP = -2,       /* a HIP runtime call failed */
  RGM_ERR_WORKSPACE = -3, /* workspace too small */
  RGM_ERR_STATE = -4      /* handle not ready (missing parameters) */
} rgm_status;

int rgm_version(void);
const char* rgm_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * DiTRotary eps-network / DiTRotaryClassifier            guided_diffusion/dit.py:538-634, :735-831
 * ---------------------------------------------------------------------------------------------- */
typedef struct rgm_dit rgm_dit;

typedef struct {
  int32_t depth;        /* 28 for DiTRotary_XL_8 (dit.py:902) */
  int32_t hidden;       /* 1152 */
  int32_t heads;        /* 16 */
  int32_t patch;        /* 8  (FlattenPatchify1D, dit.py:200-227) */
  int32_t in_ch;        /* 4  */
  int32_t out_ch;       /* 4  (learn_sigma False) ; ignored for classifiers */
  int32_t width;        /* input_size[1] = 16 */
  int32_t n_embed;      /* rows of y_embedder.embedding_table (num_classes + 1), 0 = unconditional */
  int32_t kind;         /* 0 eps-net, 1 classifier (cls token + head), 2 chord classifier (key + chord heads) */
  int32_t n_out;        /* classifier: num_classes of classifier_head */
  int32_t max_tokens;   /* largest sequence length that will be used (2*H, +1 for classifiers) */
} rgm_dit_cfg;

int rgm_dit_create(const rgm_dit_cfg* cfg, rgm_dit** out);
void rgm_dit_destroy(rgm_dit* h);
/* Copy one state_dict tensor (key exactly as in the reference module's state_dict()) into
 * the library's weight arena.  dptr: device float32; the caller keeps ownership of its tensor.
 * Replaces nn.Module.load_state_dict on scripts/sample_rule.py:71-73, :100-102. */
int rgm_dit_set_param(rgm_dit* h, const char* key, const void* dptr, const int64_t* shape, int ndim);
/* number of parameters still unset (0 == ready) */
int rgm_dit_missing_params(rgm_dit* h);
size_t rgm_dit_workspace_bytes(const rgm_dit* h, int N, int H);
/* DiTRotary.forward(x, t, y) dit.py:618-634.  x (N,in_ch,H,width); t (N) int64 (already re-spaced);
 * y (N) int32 row of the label table or NULL; eps (N,out_ch,H,width). */
int rgm_dit_forward(rgm_dit* h, const float* x, const int64_t* t, const int32_t* y, float* eps,
                    int N, int H, void* ws, size_t ws_bytes, void* stream);
/* DiTRotaryClassifier.forward dit.py:803-831.  logits (N,n_out) [kind 1]  or  key (N,25) + chord
 * (N,H/width,n_out) [kind 2; key_out may be NULL]. */
int rgm_dit_classify(rgm_dit* h, const float* x, const int64_t* t, float* logits, float* key_out,
                     int N, int H, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Building blocks (exported for parity tests and for composing other callers)
 * ---------------------------------------------------------------------------------------------- */
/* C[M,N] = act(alpha * A[M,K] . B[N,K]^T + bias[N]) (* gate) (+ res); fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * act: 0 none, 1 SiLU, 2 GELU(tanh).  gate (or NULL): gate[(row / rows_per_gate) * gate_ld + col].
 * res (or NULL, may alias C): res[row * ldres + col].  K % 32 == 0.  nn.Linear / 1x1 conv. */
int rgm_gemm(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
             const float* bias, int act, float alpha, const float* gate, int gate_ld, int rows_per_gate,
             const float* res, int ldres, void* stream);
/* out = LN(x; eps, no affine) * (1 + scale[b]) + shift[b]   (dit.py:25-26, :334-335); b = row / rows_per_batch;
 * shift/scale rows have stride mod_ld.  weight/bias (affine LN, dit.py:770) optional; shift/scale optional. */
int rgm_layernorm_modulate(const float* x, float* out, int M, int D, float eps, const float* weight,
                           const float* bias, const float* shift, const float* scale, int mod_ld,
                           int rows_per_batch, void* stream);
/* RotaryAttention core (dit.py:263-277): qkv (N*T, 3*heads*hd) -> o (N*T, heads*hd); rotary on the first
 * 2*rot_half channels of q,k with cos/sin tables (T, rot_half); softmax scale hd^-0.5. hd in {64,72}. */
int rgm_rotary_attention(const float* qkv, float* o, const float* cos_tab, const float* sin_tab,
                         int N, int T, int heads, int hd, int rot_half, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RGM_H */
