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
/* The adaLN modulation of U (t, y) pairs: rows[U][(6 depth + 2) hidden] = adaLN_modulation(SiLU(t_embedder(t) + y_embedder(y))) of every
 * block and the final layer (dit.py:621-628, :333, :374) -- exactly what rgm_dit_forward computes for a sample with that timestep and label,
 * row by row (a row's arithmetic does not depend on the batch it sits in).  One pass over the adaLN weights (0.9 GB for XL) per 32 pairs.
 * A sampler that knows its schedule (gaussian_diffusion.py:833-880: the loop visits fixed timesteps) asks for the next steps' rows at once
 * and passes them to rgm_dit_forward_cond: sample i takes rows[idx[i]] (idx: device int32, 0 <= idx[i] < U).  ws as for N = U rows. */
int rgm_dit_cond_rows(rgm_dit* h, const int64_t* t, const int32_t* y, int U, int H, float* rows, void* ws, size_t ws_bytes, void* stream);
int rgm_dit_forward_cond(rgm_dit* h, const float* x, const float* rows, const int32_t* idx, int U, float* eps,
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
/* adaLN conditioning of a whole forward in one weight-streaming pass (csrc/adaln_stream.hip): mod[N, L] = cs[N, D] . W^T[L, D] + bias[L],
 * cs = SiLU(c) -- the Linear of every block's adaLN_modulation (ref guided_diffusion/dit.py:318-322, 333) and of the final layer (:366-369,
 * 374), whose weights the DiT handles keep contiguous.  Exact fp32 (v_mfma_f32_16x16x4_f32); one pass over the weights per 32 rows.  N <= 256, D in {384, 768, 1152}, L % 16 == 0,
 * 16-byte aligned pointers; RGM_ERR_INVALID for any other shape (rgm_dit_forward then runs its tiled GEMM instead). */
int rgm_adaln_stream(const float* cs, const float* W, const float* bias, float* mod, int N, int D, int L, void* stream);
/* RotaryAttention core (dit.py:263-277): qkv (N*T, 3*heads*hd) -> o (N*T, heads*hd); rotary on the first
 * 2*rot_half channels of q,k with cos/sin tables (T, rot_half); softmax scale hd^-0.5. hd in {64,72}. */
int rgm_rotary_attention(const float* qkv, float* o, const float* cos_tab, const float* sin_tab,
                         int N, int T, int heads, int hd, int rot_half, void* stream);
/* the same forward, also writing lse (N*heads*T): log-sum-exp of the scaled scores of every query, the saved quantity of the backward */
int rgm_rotary_attention_lse(const float* qkv, float* o, float* lse, const float* cos_tab, const float* sin_tab,
                             int N, int T, int heads, int hd, int rot_half, void* stream);

/* Arithmetic of the GEMM family (every nn.Linear / conv of the path):
 *   0  exact fp32 products on v_mfma_f32_32x32x2_f32 (default; 157 TFLOP/s peak);
 *   1  "bf16x3": operands split hi+lo bf16 while staged to LDS, a*b ~= ah*bh + ah*bl + al*bh on
 *      v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~2e-5 relative per product; 833 TFLOP/s equivalent peak).
 *   2  "bf16x3_presplit": same products, hi/lo split done once by the producer of each operand; the DiT backbone
 *      GEMMs and the VAE 3x3 convs run on the LDS-DMA kernel (rgm_gemm_split), everything else as mode 1.
 * Process-wide default used by all handles; results stay within the 1e-3 latent tolerance in every mode (tests). */
int rgm_set_gemm_precision(int prec);
int rgm_get_gemm_precision(void);
/* Element type of the hi / lo halves the bf16x3 modes split fp32 operands into -- a build constant of the library: 0 = bf16 (default
 * build), 1 = fp16 (make F16=1 -> librgm_hip_f16.so: 2^-22 instead of 2^-16 per product at the same MFMA rate, fp16's exponent range). */
int rgm_split_dtype(void);
/* Split-row format of the pre-split bf16x3 path (precision 2, "bf16x3_presplit"): a logical fp32 row of K values
 * keeps its K*4 bytes; every block of 32 values is one 128-byte line [32 bf16 hi | 32 bf16 lo] (x ~= hi + lo, both
 * round-to-nearest; K % 32 == 0).  rgm_split_rows converts (rows,K) fp32 -> split (out-of-place); rgm_gemm_split is
 * the LDS-DMA kernel (csrc/gemm2.hip) on operands already in that format (A (M,K), B (N,K) both split):
 * C = act(A . B^T + bias), optionally written split as well (N % 32 == 0).  tile: 0 auto; 1 128x128, 2 128x64,
 * 3 64x64, 5 256x128 (3-stage ring); 21 / 22 = 128x128 / 128x64 with the single-set pipeline; 43 / 44 / 45 / 46 =
 * 128x128 / 128x64 / 256x128 / 64x64 with the cross-iteration register pipeline; 51 / 52 = 128x128 / 128x64 with
 * the loader/consumer split (4 MFMA waves + 4 DMA waves, 3-stage ring). */
int rgm_split_rows(const float* x, float* out, int64_t rows, int K, void* stream);
int rgm_gemm_split(const float* A_split, const float* B_split, float* C, int M, int N, int K, const float* bias,
                   int act, int tile, int out_split, void* stream);
/* the same with explicit row strides (elements; multiples of 32 for split rows): padded rows keep one K-slice of
 * many rows from landing on a few L2 channels when K*4 is a multiple of 2 KiB */
int rgm_split_rows_ld(const float* x, int ld_in, float* out, int ld_out, int64_t rows, int K, void* stream);
int rgm_gemm_split_ld(const float* A_split, int lda, const float* B_split, int ldb, float* C, int ldc, int M, int N, int K,
                      const float* bias, int act, int tile, int out_split, void* stream);
/* The one-wave-per-SIMD kernels (tiles 71 = 256x256, 72 = 512x128; csrc/gemm2.hip PIPE 5): mode 0 keeps the heuristics off them, 1 (default)
 * lets them choose; min_tiles = tiles a VAE conv launch must have before it takes them (default 256: one round of the chip). */
int rgm_set_big_tiles(int mode, int min_tiles);
/* K-sliced fc2 of a DiT block (rgm_dit_forward, pre-split arithmetic): 1 (default) = the kernel that reduces the K slices also writes the
 * next block's adaLN-LayerNorm of each row (ref guided_diffusion/dit.py:334-336 -- same values as the separate LayerNorm launch,
 * tests/test_gpu_fullsize.py), 0 = separate launches.  rgm_fused_reduce_ln_launches: how many launches took the fused route so far. */
int rgm_set_fuse_reduce_ln(int on);
long long rgm_fused_reduce_ln_launches(void);
/* adaLN conditioning of a forward (ref guided_diffusion/dit.py:332, 372: every block's adaLN_modulation(c), 0.9 GB of weights for N rows):
 * 1 = block 0's slice in front of block 0, the rest on a side stream owned by the handle, forked from and joined to the caller's
 * stream by events (still stream-ordered for the caller; capturable) while block 0 computes; 0 (default; measured equal or better) = one
 * GEMM in front of block 0. */
int rgm_set_adaln_overlap(int on);
/* GroupNorm + swish of a ResnetBlock's conv1 output (ref taming/modules/diffusionmodules/model.py:117-126: h = conv1(.); h = norm2(h);
 * h = nonlinearity(h)) inside the conv launch of the decoder: the tiles of an image leave their partial sums, meet at an L2-resident
 * counter (bounded wait) and write normalised split rows -- the separate HBM pass over the tensor disappears.  mode 0 = never, 1 = where
 * the launch qualifies (default), 2 = every tile on the fallback path (raw rows + in-place fix-up kernel; tests).  *prev (optional)
 * receives the previous mode; rgm_gn_fused_launches() counts the launches that took the route. */
int rgm_set_gn_fuse(int mode, int* prev);
long long rgm_gn_fused_launches(void);
/* Tiles of those fused launches whose bounded wait (1 ms) for the image's other tiles ran out -- they write raw rows and a fix-up pass
 * converts them, same values -- since the last reset; synchronises the device; reset != 0 zeroes the counter.  0 on an idle device:
 * a non-zero count says another stream held CUs while a decode ran (each such tile costs up to 1 ms). */
long long rgm_gn_fallback_tiles(int reset);
/* Blocks of an eps-network forward (ref guided_diffusion/dit.py:618-634: samples are independent inside a block) as TWO half batches, the
 * second on a side stream owned by the handle, forked from and joined to the caller's stream by events (stream-ordered for the caller):
 * one half's kernels fill the CUs the other half's last round of one-workgroup-per-CU tiles leaves idle.  Batches of at least
 * min_batch samples take it; 0 = never; -1 (default) = the batch sizes where a same-box sweep found it ahead (2, 5..9, 17..39, 57..).
 * *prev (optional) receives the previous setting. */
int rgm_set_dit_halves(int min_batch, int* prev);
/* Short-sequence attention of the bf16x3 modes at head_dim 72 (T <= 128: C5's half windows; ref guided_diffusion/dit.py:263-288) with TWO
 * workgroups per CU (four waves and 79 KiB of LDS each) instead of the one-per-CU guard of round 3: the guard answered a hazard of the
 * interleaved Q prologue, which the two-phase prologue removed (profiles/r05_attn_hazard_two_phase_n96.txt).  1 = on,
 * 0 = guard (default).  Returns the previous setting. */
int rgm_set_attn_pairs(int on);
/* Deterministic split-K of the pre-split GEMM (csrc/gemm2.hip: K slices as a batch + one fixed-order reduce kernel, which for the fc2
 * of a DiT block also writes the next adaLN-LayerNorm): the scratch is caller memory like every other workspace,
 * rgm_gemm_scratch_bytes(M, N) bytes for GEMMs of up to M rows and N columns, 16-byte aligned, no initialisation.  tile 0 lets the
 * heuristic choose. */
size_t rgm_gemm_scratch_bytes(int M, int N);
int rgm_gemm_split_ws(const float* A_split, const float* B_split, float* C, int M, int N, int K, const float* bias, int act,
                      int tile, int out_split, void* ws, size_t ws_bytes, void* stream);
/* The general pre-split entry: C = (act(alpha * A . B^T + bias)) * gate + res with explicit row strides (elements), the per-sample
 * adaLN gate gate[(row / rows_per_gate) * gate_ld + col] and a residual that may alias C (the proj / fc2 epilogue of a DiT block,
 * guided_diffusion/dit.py:332-336), explicit tile as in rgm_gemm_split (71.. = the one-wave-per-SIMD tiles) and the
 * caller's scratch (NULL: no workspace-backed decomposition).  The tiles above 128x128 (5, 45, 71, 72, 73) take a gate only with
 * rows_per_gate >= 32 (RGM_ERR_INVALID otherwise); tile 0 keeps finer gates on the 128-row kernels. */
int rgm_gemm_split_epi(const float* A_split, int lda, const float* B_split, int ldb, float* C, int ldc, int M, int N, int K,
                       const float* bias, int act, float alpha, const float* gate, int gate_ld, int rows_per_gate,
                       const float* res, int ldres, int tile, int out_split, void* ws, size_t ws_bytes, void* stream);
/* Same as rgm_gemm without gate/residual but with an explicit tile shape in the low 4 bits (1: 128x128,
 * 2: 128x64, 3: 64x64, 4: 32x128, 0: auto) and an explicit precision in bits 4.. (0: library default,
 * 1: fp32, 2: bf16x3) -- used by the parity tests and tile-selection experiments. */
int rgm_gemm_tile(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                  const float* bias, int act, int tile, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sampler step kernels                                   guided_diffusion/gaussian_diffusion.py
 * `tables_host`: HOST array of 8 DEVICE pointers to the float32-cast schedule tables (length = number of
 * re-spaced timesteps), in this order: sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod,
 * posterior_mean_coef1, posterior_mean_coef2, model_variance (FIXED_LARGE, :316-329), model_log_variance,
 * alphas_cumprod, alphas_cumprod_prev  -- i.e. what _extract_into_tensor (:1331-1344) would upload per call.
 * t: (N) int64 indices into those tables.  E = elements per sample.
 * ---------------------------------------------------------------------------------------------- */
/* Counter-based N(0,1): element i depends only on (seed, offset+i) (Philox4x32-10 + Box-Muller).
 * Replaces th.randn / th.randn_like at :846, :699, :715, :944, :512. */
int rgm_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* p_mean_variance + condition_mean + p_sample update (:252-357, :402-407, :698-703):
 *   x0 = c1 x - c2 eps (clip) ; mean = pc1 x0 + pc2 x (+ var * grad) ; sample = mean + [t > t_end] exp(.5 logvar) noise.
 * grad / noise may be NULL (noise NULL -> sample = mean).  g_out (N) or NULL receives exp(.5 logvar). */
int rgm_ddpm_step(const float* x, const float* eps, const float* grad, const float* noise, const int64_t* t,
                  const float* const* tables_host, int clip_denoised, int t_end, float* sample, float* pred_xstart,
                  float* g_out, int N, int E, void* stream);
/* The same step with LEARNED variances (learn_sigma=True checkpoints; p_mean_variance :299-313): var_values (N,E) is the second
 * half of the network's output; min_log_tab / max_log_tab (device float32 tables: posterior_log_variance_clipped, log(betas))
 * give log var = frac max + (1 - frac) min, frac = (v + 1) / 2 (LEARNED_RANGE); both NULL: var_values is the log-variance (LEARNED). */
int rgm_ddpm_step_learned(const float* x, const float* eps, const float* var_values, const float* min_log_tab,
                          const float* max_log_tab, const float* grad, const float* noise, const int64_t* t,
                          const float* const* tables_host, int clip_denoised, int t_end, float* sample, float* pred_xstart, int N,
                          int E, void* stream);
/* ... also writing the per-element noise scale g_elem (N,E) = exp(0.5 log_variance): the tensor g_coeff p_sample hands scg_sample at
 * a learned-variance step (:706-711).  Per-element candidates: rgm_scg_candidates(mean, g_elem, noise, cand, n, N*E, 1). */
int rgm_ddpm_step_learned_g(const float* x, const float* eps, const float* var_values, const float* min_log_tab,
                            const float* max_log_tab, const float* grad, const float* noise, const int64_t* t,
                            const float* const* tables_host, int clip_denoised, int t_end, float* sample, float* pred_xstart,
                            float* g_elem, int N, int E, void* stream);
/* ddim_sample (:881-952) incl. condition_score (:467-489) when grad != NULL; g_out receives sigma. */
int rgm_ddim_step(const float* x, const float* eps, const float* grad, const float* noise, const int64_t* t,
                  const float* const* tables_host, int clip_denoised, int t_end, float eta, float* sample,
                  float* pred_xstart, float* g_out, int N, int E, void* stream);
/* scg_sample candidate expansion (:509-514): cand[k][b] = mean[b] + g[b] * noise[k][b], k < n. */
int rgm_scg_candidates(const float* mean, const float* g, const float* noise, float* cand, int n, int B, int E,
                       void* stream);
/* _predict_xstart_from_eps (:359-364) times out_scale (the 1/scale_factor of _decode :1350). */
int rgm_xstart_from_eps(const float* x, const float* eps, const int64_t* t, const float* const* tables_host,
                        float out_scale, float* out, int N, int E, void* stream);
/* Replacement-based conditioning of the editing path (p_mean_variance :293-298): x0 = clip(predict_xstart(x, eps)),
 * x0r = mask*gt + (1-mask)*x0, eps_out = predict_eps_from_xstart(x, x0r).  gt, mask (N,E) like x; eps_out may alias eps. */
int rgm_edit_replace_eps(const float* x, const float* eps, const float* gt, const float* mask, const int64_t* t,
                         const float* const* tables_host, int clip_denoised, float* eps_out, int N, int E, void* stream);
/* scg_sample selection (:539-554): max_ind[b] = first argmax_k total[k][b]; out[b] = cand[max_ind[b]][b].
 * cand and out may both be NULL (index only); max_ind may be NULL. */
int rgm_scg_select(const float* cand, const float* total_logp, float* out, int64_t* max_ind, int n, int B, int E,
                   void* stream);
/* Sharded SCG (SURVEY 8e option i; replaces the gather `sample[max_ind, arange(B)]` of :553-554 / :587-589 when the winner
 * was scored on another rank): out[b] = mean[b] + g[b] * z, z = the rgm_randn stream (seed) at positions
 * base + (k*B + b)*E + e with k = max_ind[seg][b]; the latent row h of (C,H,W) belongs to segment h / seg_rows
 * (max_ind is (S,B), S = ceil(H / seg_rows); seg_rows >= H: one winner per sample).  No host read of max_ind. */
int rgm_scg_rebuild(const float* mean, const float* g, const int64_t* max_ind, uint64_t seed, uint64_t base, float* out,
                    int B, int E, int H, int W, int seg_rows, void* stream);
/* the same with a per-element noise scale g_elem (B,E) (learned variances) */
int rgm_scg_rebuild_g(const float* mean, const float* g_elem, const int64_t* max_ind, uint64_t seed, uint64_t base, float* out,
                      int B, int E, int H, int W, int seg_rows, void* stream);

/* ------------------------------------------------------------------------------------------------
 * taming KL-VAE decoder (f8-all-onset config)            taming/models/klvae_pedal.py:80-85,
 *                                                        taming/modules/diffusionmodules/model.py:436-537
 * ---------------------------------------------------------------------------------------------- */
typedef struct rgm_vae rgm_vae;
int rgm_vae_create(rgm_vae** out);
void rgm_vae_destroy(rgm_vae* h);
/* key as in the Lightning checkpoint's ["state_dict"] ("decoder.*", "post_quant_conv.*"; klvae_pedal.py:50-59).
 * 3x3 conv weights are repacked to [cout][tap][cin] on the device.  "encoder.*" / "quant_conv.*" are optional (only
 * rgm_vae_encode needs them); other prefixes (loss.) are not ours: query with rgm_vae_has_param and skip them
 * (strict=False semantics). */
int rgm_vae_set_param(rgm_vae* h, const char* key, const void* dptr, const int64_t* shape, int ndim);
int rgm_vae_has_param(rgm_vae* h, const char* key);
int rgm_vae_missing_params(rgm_vae* h);            /* decoder + post_quant_conv parameters still unset (what decode needs) */
int rgm_vae_encoder_missing_params(rgm_vae* h);    /* encoder + quant_conv parameters still unset (what encode needs) */
size_t rgm_vae_workspace_bytes(const rgm_vae* h, int M /* number of 16x16 latent squares */);
/* AutoencoderKL.decode(z): z (M,4,16,16) -> out (M,3,128,128). */
int rgm_vae_decode(rgm_vae* h, const float* z, float* out, int M, void* ws, size_t ws_bytes, void* stream);
/* _decode (gaussian_diffusion.py:1347-1358) fused: latent (N,4,H,16) * inv_scale -> H/16 squares per sample ->
 * roll (N,3,128,8H) float32 (may be NULL) and/or roll_u8 (N,128,8H,3) uint8 with the background threshold and
 * truncating cast of decode_sample_for_midi (midi_util.py:59-63) (may be NULL). */
int rgm_vae_decode_latent(rgm_vae* h, const float* latent, float inv_scale, float* roll, uint8_t* roll_u8,
                          float threshold, int N, int H, void* ws, size_t ws_bytes, void* stream);
/* AutoencoderKL.encode_save(x, range_fix=False) (klvae_pedal.py:61-68; Encoder.forward model.py:404-433 incl. the
 * stride-2 Downsample convs :56-75, then quant_conv): x (M,3,128,128) piano-roll tiles in [-1,1] -> moments
 * (M,8,16,16) = mean (channels 0..3) | logvar (4..7).  Workspace as for decode with the same M. */
int rgm_vae_encode(rgm_vae* h, const float* x, float* moments, int M, void* ws, size_t ws_bytes, void* stream);
/* Decoder input gradient -- what torch.autograd computes when the reference's dps_rule branch differentiates
 * rule(_decode(x0_hat)) w.r.t. the latent (gaussian_diffusion.py:415-433 with guidance.nn False; _decode :1347-1358;
 * Decoder.forward model.py:506-537).  Two stream-ordered calls sharing one workspace of
 * rgm_vae_grad_workspace_bytes(h, N*H/16) bytes that the caller leaves untouched in between:
 *   rgm_vae_decode_latent_save: the same roll as rgm_vae_decode_latent, keeping every ResnetBlock's input and conv1
 *     output, the attention tensors and all GroupNorm statistics (~105 MB per 16x16 square) in `ws`;
 *   rgm_vae_decode_latent_vjp:  d_latent (N,4,H,16) = (d roll / d latent)^T d_roll, d_roll (N,3,128,8H).
 * rgm_vae_enable_grad builds the mirrored/transposed weight copies the backward GEMMs read (once, after set_param;
 * ~0.2 GB); the grad calls return RGM_ERR_STATE without it.  Input gradients only -- no parameter gradients. */
int rgm_vae_enable_grad(rgm_vae* h);
size_t rgm_vae_grad_workspace_bytes(const rgm_vae* h, int M /* number of 16x16 latent squares */);
int rgm_vae_decode_latent_save(rgm_vae* h, const float* latent, float inv_scale, float* roll, int N, int H, void* ws,
                               size_t ws_bytes, void* stream);
int rgm_vae_decode_latent_vjp(rgm_vae* h, const float* d_roll, float inv_scale, float* d_latent, int N, int H, void* ws,
                              size_t ws_bytes, void* stream);
/* midi_util.py:59-63 on an existing roll: (B,3,128,T) float32 -> (B,128,T,3) uint8. */
int rgm_quantise_roll(const float* roll, uint8_t* out_u8, int B, int T, float threshold, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rule programs (FUNC_DICT / LOSS_DICT built-ins)        music_rule_guidance/music_rules.py, rule_maps.py
 * roll (N,C,128,T) float32, channel 0 is read AND WRITTEN (piano_like mask / background threshold, like the
 * reference's in-place view writes).
 * ---------------------------------------------------------------------------------------------- */
/* total_pitch_class_histogram (:29-43): out (N,12); scratch N*128 floats. */
int rgm_rule_pitch_hist(float* roll, float* out, float* scratch, int N, int C, int T, void* stream);
/* rule_x0_mse_dummy for pitch_hist (condition_functions.py:122-126): logp (N) = -scale * ||pitch_hist(roll) - target||^2
 * and d_roll (N,C,128,T) = d logp / d roll (0.5 * d/dh on channel 0's piano rows, 0 elsewhere); hist (N,12) optional;
 * scratch N*140 floats.  Writes the piano_like mask into roll like rgm_rule_pitch_hist. */
int rgm_rule_pitch_hist_vag(float* roll, const float* target, float scale, float* hist, float* logp, float* d_roll,
                            float* scratch, int N, int C, int T, void* stream);
/* note_density (:46-83): out (N, 2*T/interval) = [vertical..., horizontal...]; interval divides 256 and T. */
int rgm_rule_note_density(float* roll, float* out, int N, int C, int T, int interval, float hscale, void* stream);
/* get_chords' preamble (music_rules.py:97-110): piano_like mask and < -0.95 -> -1 written into channel 0 of roll, then
 * clamp((x+1)/2*127, 0, 127) truncated -> out (N,128,T) uint8: the integer roll the host chord analyser (music21) reads. */
int rgm_rule_chord_quantise(float* roll, uint8_t* out, int N, int C, int T, void* stream);
/* torch.bucketize(v, bounds) as used by note_density_class (:86-94): out int64. */
int rgm_bucketize(const float* v, const float* bounds, int nb, int64_t* out, int n, void* stream);
/* mse_loss_mean / zero_one_loss_mean (rule_maps.py:17-22) over the last dim: a,b (rows,K) -> out (rows). */
int rgm_row_loss(const float* a, const float* b, float* out, int rows, int K, int zero_one, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Classifier guidance backward                            guided_diffusion/condition_functions.py:58-85
 * ---------------------------------------------------------------------------------------------- */
/* Value and input-gradient of a DiTRotaryClassifier log-probability in one call (replaces
 * th.autograd.grad(log_probs.sum(), x_in) * classifier_scale; weights are frozen, so only dgrad runs):
 *   loss_kind 0 (kind-1 handle): log p = -sum_k (logits - target)^2, target float32 (N, n_out)    [grad_nn_zt_mse]
 *   loss_kind 1 (kind-2 handle): log p = -sum_w CE(chord_logits[w], target[w]), target int64 (N, H/width)
 *                                                                                 [grad_nn_zt_chord, both=False]
 *   loss_kind 1 (kind-1 handle): log p = log softmax(logits)[target], target int64 (N,)             [grad_nn_zt_xentropy :46-56]
 * grad_x (N,in_ch,H,width) = d(sum log p)/dx * scale;  logits_out (N[,H/width], n_out) or NULL. */
size_t rgm_dit_grad_workspace_bytes(const rgm_dit* h, int N, int H);
int rgm_dit_cls_value_and_grad(rgm_dit* h, const float* x, const int64_t* t, const void* target, int loss_kind,
                               float scale, float* logits_out, float* grad_x, int N, int H, void* ws,
                               size_t ws_bytes, void* stream);
/* Input gradient of the eps-network -- the autograd step of DPS guidance (condition_mean, gaussian_diffusion.py:415-465:
 * th.autograd.grad(log_probs.sum(), xt) through pred_xstart(xt, model(xt))): eps_out = model(x, t, y) (optional) and
 * grad_x = (d eps / d x)^T g_eps, via the saved-activation forward and the same dgrad chain as the classifiers.
 * rgm_dit_enable_grad(h) first: it keeps W^T copies of the eps-network's Linear weights (1.8 GB at XL); workspace =
 * rgm_dit_grad_workspace_bytes(h, N, H).  Two phases that may be separate calls on the same workspace: forward
 * (x, t [, y] given; g_eps/grad_x NULL) saves the activations and writes eps_out; backward (x NULL; g_eps, grad_x given)
 * consumes them -- DPS forms g_eps from the classifier's gradient at x0(eps) in between. */
int rgm_dit_enable_grad(rgm_dit* h);
int rgm_dit_vjp(rgm_dit* h, const float* x, const int64_t* t, const int32_t* y, const float* g_eps, float* eps_out,
                float* grad_x, int N, int H, void* ws, size_t ws_bytes, void* stream);
/* d(qkv) of the RotaryAttention core from dO, the saved qkv, O and per-query log-sum-exp (N, heads, T); hd = 64. */
/* Attention launches of the classifier path (DiTRotary-S/8-cls: 257 tokens = 9 tiles of 32 on 8 waves, 6 heads x the sampler's batch of
 * (sample, head) pairs on 256 CUs; ref guided_diffusion/dit.py:803-831 + condition_functions.py:58-64): -1 (default) = per-tile workgroups
 * (backward: one per query / key tile, partial sums through LDS in a fixed order; forward: two per (sample, head)) whenever the
 * (sample, head) grid would leave CUs idle, 0 = never, 1 = always.  Same values to the last bit of the summation order. */
int rgm_set_attn_split(int mode);
int rgm_rotary_attention_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                             const float* cos_tab, const float* sin_tab, int N, int T, int heads, int hd,
                             int rot_half, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py roofline leg): with profiling on, every GEMM launch is bracketed by two
 * hipEvents recorded on the launch stream.  kernel ids: gemm.hip = tile (1..5: 128x128 / 128x64 / 64x64 /
 * 32x128 / 256x128) + 10 with the implicit 3x3-conv loader + 20 in bf16x3; gemm2.hip = 40 + tile (as in
 * rgm_gemm_split) + 10 with the conv loader.
 * rgm_gemm2_dbg: s_memtime stamps of the gemm2 K loop (tools/gemm_stamp.py; library built with
 * -DRGM_GEMM2_STAMPS): mode 1 arm, 2 copy 64 counters to out64, 0 off.
 * ---------------------------------------------------------------------------------------------- */
int rgm_prof_enable(int on);
int rgm_prof_reset(void);
int rgm_prof_report(int kernel, int* launches, double* total_ms, double* total_flops);
/* algorithmic HBM bytes (every operand read once, the output written once) summed over the recorded launches of a pre-split kernel id */
double rgm_prof_bytes(int kernel);
/* per-launch records of the pre-split GEMM kernels in launch order (kernel id, milliseconds, algorithmic FLOPs); returns the count */
int rgm_prof_dump(int cap, int* ids, double* ms, double* flops);
int rgm_gemm2_dbg(int mode, long long* out64);
/* the same for the 128x144 kernel (gemm144.hip, tools/g144_stamp.py): always built; 8 waves x 8 slots (cycles of the middle workgroup).
 * mode 1 arms, 2 copies the 64 slots out, 3 copies 64 + 8 x 4096 values (per workgroup of the stamped launch: entry / exit in shader cycles,
 * entry / exit on the 100 MHz wall clock, K loop, epilogue, prologue cycles, raster id -- tools/g144_insitu_stamp.py ALLWG=1), 0 disarms */
int rgm_gemm144_dbg(int mode, long long* out64);

/* ------------------------------------------------------------------------------------------------
 * DiffCollage long-sequence composition                   diff_collage/w_img.py, condind_long.py, condind_circle.py
 * ---------------------------------------------------------------------------------------------- */
/* split_wimg (:8-24): img (B,C,h,W) -> wins (B*n,C,h,128), window i at columns [i*(128-overlap), +128), read
 * circularly (column >= W wraps) so the circular variant needs no concatenated copy.  halves (B*n,C,h,overlap)
 * or NULL receives the right `overlap` columns of every window (input of the half-window eps call). */
int rgm_collage_split(const float* img, float* wins, float* halves, int B, int C, int h, int W, int n, int overlap,
                      void* stream);
/* get_eps_t_fn tail (condind_long.py:36-50 / condind_circle.py:57-82): out = fold-sum of full_i minus half_i
 * (i < n-1) on the right overlaps; circle != 0 averages the wrapped seam; is_avg divides by the coverage
 * count (avg_merge_wimg is_avg=True).  half may be NULL (plain merge). */
int rgm_collage_merge(const float* full, const float* half, float* out, int B, int C, int h, int n, int overlap,
                      int circle, int is_avg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RGM_H */
