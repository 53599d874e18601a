// dit.hip -- DiTRotary eps-network and DiTRotaryClassifier: weight arena + forward schedule.
//
// Reference: guided_diffusion/dit.py:538-634 (DiTRotary), :735-831 (DiTRotaryClassifier),
// :315-336 (DiTBlockRotary), :359-376 (FinalLayerPatch1D), :33-70 (TimestepEmbedder), :73-100 (LabelEmbedder).
// The host side here only sequences kernels on the caller's stream; all arithmetic lives in
// gemm.hip / attention.hip / dit_kernels.hip.
//
// HBM layout
//   * weights: ONE arena, fp32, nn.Linear layout [out][in] (K-contiguous = the GEMM's B^T operand
//     as-is).  The adaLN projections of all blocks (+ the final layer's) are placed back to back, so
//     the conditioning of the whole network is ONE skinny GEMM  mod[N, (6*depth+2)*D] = SiLU(c) . W^T
//     per forward instead of depth+1 launches that each re-stream their weights for 16 rows;
//   * activations: caller workspace; residual stream x[N*T, D] is updated in place by the gated
//     epilogues of proj / fc2, xm / qkv / attn / hidden are scratch strips reused by every block.
#include <map>
#include <stdlib.h>
#include <string>
#include <vector>
#include <math.h>
#include "common.h"

extern thread_local int g_attn_co_sched;   // attention_x3.hip
namespace rgm {
int patchify_launch(const float* x, float* tok, int N, int C, int H, int W, int P, hipStream_t s);
int unpatchify_launch(const float* tok, float* out, int N, int OC, int H, int W, hipStream_t s);
int timestep_sincos_launch(const int64_t* t, const float* freqs, float* emb, int N, int half, hipStream_t s);
int cond_finish_launch(const float* c, const float* table, const int32_t* y, float* cs, int N, int D, hipStream_t s);
int fill_cls_launch(const float* cls, float* x, int N, int T, int D, hipStream_t s);
int pool_rows_launch(const float* x, float* out, int N, int T, int D, int first, int groups, int per, hipStream_t s);
int ln_mod_bwd_launch(const float* dy, const float* x, const float* res, float* out, int M, int D, float eps,
                      const float* weight, const float* scale, int mod_ld, int rows_per_batch, hipStream_t s,
                      const float* gate2 = nullptr, float* gated = nullptr);   // gated: also out * gate2[row / rows_per_batch] as split rows
int gate_rows_launch(const float* dx, const float* gate, float* out, int M, int D, int gate_ld, int rows_per_batch, hipStream_t s,
                     int out_split = 0);
int act_rows_launch(const float* in, float* out, long long rows, int D, int act, hipStream_t s, int out_split = 0);
int loss_grad_launch(const float* logits, const void* target, float* dl, int rows, int K, int Kp, float scale, int kind, hipStream_t s);
int scatter_rows_launch(const float* src, float* dx, int N, int T, int D, int first, int groups, int per, hipStream_t s);
int adaln_stream_launch(const float* cs, const float* W, const float* bias, float* out, int N, int D, int L, int ldo, hipStream_t s);   // adaln_stream.hip
}  // namespace rgm

using namespace rgm;

struct Slot {
  size_t off = 0;     // floats from arena base
  size_t numel = 0;
  bool set = false;
  int t_rows = 0, t_cols = 0, t_ld = 0;   // ".T" slots: transposed copy of a [t_cols(out)][t_rows(in)] weight, row stride t_ld
  bool in_t = false;                      // lives in arena_t (the W^T copies an eps-network gets from rgm_dit_enable_grad)
};

static int g_adaln_overlap = getenv("RGM_ADALN_OVERLAP") ? atoi(getenv("RGM_ADALN_OVERLAP")) : 0;   // measured: no gain

// The blocks of an eps-network forward as TWO half batches on two streams (rows of different samples never meet inside a block): one
// half's kernels fill the CUs the other half's last tile round leaves idle, and their write-bound epilogues fall under the other half's
// K loops.  g_dit_halves (rgm_set_dit_halves / RGM_DIT_HALVES): -1 = where the same-box sweep of tools/halves_exp.py found it ahead
// (profiles/r04_halves_sweep.txt, r05_halves_sweep.txt: B = 2, 5..9 and from 17 up -- 2..14 % -- but behind at 10..14, and even at 16, where
// ONE round of 256x256 tiles per GEMM leaves nothing to overlap); 0 = never; n > 0 = every batch of at least n samples.
static int g_dit_halves = getenv("RGM_DIT_HALVES") ? atoi(getenv("RGM_DIT_HALVES")) : -1;
static bool dit_halves_for(int N) {
  if (g_dit_halves == 0 || N < 2) return false;
  if (g_dit_halves > 0) return N >= g_dit_halves;
  // (round 5, after the 144-column tiles' raster / prefetch: profiles/r05_halves_sweep.txt -- B = 40 / 48 / 56 4.4 / 2.7 / 3.3 % ahead as halves
  // now, 10 .. 16 still 5-7 % behind; with a (sample, head)'s queries split over workgroups (attention_x3.hip) B = 4 is 5.12 ms as ONE batch
  // against 5.30 as halves, B = 2 / 3 are level: profiles/r05_halves_sweep_small.txt)
  return N == 2 || (N >= 5 && N <= 9) || N >= 17;
}
// parts of a split forward (2 .. 4; read once: the workspace plan depends on it).  Three and four parts measured behind two at every batch
// size but B = 112 (profiles/r04_halves_parts.txt): the parts of a forward re-read the weights and shrink the tile grids
static const int g_dit_parts = getenv("RGM_DIT_PARTS") ? atoi(getenv("RGM_DIT_PARTS")) : 2;
// where in block 0 of the first part the other parts are released: 0 = with it, 1 .. 5 = behind its qkv / attention / proj / second LayerNorm /
// fc1 (the parts then run out of phase: one's short-K GEMMs beside the other's long-K ones).  Measured: no gain (profiles/r04_halves_stagger.txt)
static int dit_stagger() {
  static const int v = getenv("RGM_DIT_STAGGER") ? atoi(getenv("RGM_DIT_STAGGER")) : 0;
  return v >= 0 && v <= 5 ? v : 0;       // the release must happen exactly once per forward
}

struct rgm_dit {
  rgm_dit_cfg cfg{};
  int device = 0;
  std::map<std::string, Slot> slots;
  float* arena = nullptr;
  size_t arena_floats = 0;
  float* tfreqs = nullptr;   // [128] timestep frequencies
  float* cos_tab = nullptr;  // [max_tokens][rot_half]
  float* sin_tab = nullptr;
  bool rotary_ready = false;
  int rot_half = 0, hd = 0;
  size_t ada_rows = 0;       // (6*depth + 2 or 0) * D
  float* arena_t = nullptr;  // eps-network only, allocated by rgm_dit_enable_grad: W^T copies for the input-gradient (DPS)
  size_t arena_t_floats = 0;
  void* sk_ws = nullptr;     // scratch of the call in progress (points into the caller's workspace; see GemmParams::sk_ws)
  size_t sk_ws_bytes = 0;
  // adaLN conditioning of blocks 1.. (0.9 GB of weights for N rows: HBM-bound, 0.27 ms of a C2 step) on a side stream, forked from and
  // joined to the caller's stream by events, while block 0 runs (rgm_set_adaln_overlap)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t side_x[2] = {nullptr, nullptr};       // third / fourth part of a forward split into more than two parts (g_dit_parts)
  hipEvent_t ev_join_x[2] = {nullptr, nullptr};

  const float* p(const std::string& k) const { return arena + slots.at(k).off; }
  float* sp(const Slot& sl) const { return (sl.in_t ? arena_t : arena) + sl.off; }
};

static void add_slot(rgm_dit* h, const std::string& key, size_t numel) {
  Slot s;
  s.off = h->arena_floats;
  s.numel = numel;
  h->slots[key] = s;
  h->arena_floats += (numel + 3) / 4 * 4;  // keep every tensor 16-byte aligned
}

// transposed copy W^T [in][out_padded] of a Linear weight [out][in]: the B operand of the dgrad GEMM dX = dY . W
static void add_t_slot(rgm_dit* h, const std::string& key, int out_f, int in_f) {
  const int ld = (out_f + 31) / 32 * 32;
  Slot s;
  s.off = h->arena_floats;
  s.numel = (size_t)in_f * ld;
  s.set = true;                    // filled as a side effect of setting `key`
  s.t_rows = in_f;
  s.t_cols = out_f;
  s.t_ld = ld;
  h->slots[key + ".T"] = s;
  h->arena_floats += s.numel;
}

// split-row copy of a ".T" slot (W^T [in][ld]): B operand of the pre-split dgrad GEMM dX = dY . W
static void add_ts_slot(rgm_dit* h, const std::string& key, bool in_t = false, size_t* cursor = nullptr) {
  const Slot& t = h->slots.at(key + ".T");
  Slot s;
  s.off = cursor ? *cursor : h->arena_floats;
  s.numel = t.numel;
  s.set = true;
  s.t_rows = t.t_rows; s.t_cols = t.t_cols; s.t_ld = t.t_ld;
  s.in_t = in_t;
  h->slots[key + ".TS"] = s;
  if (cursor) *cursor += s.numel;
  else h->arena_floats += s.numel;
}

// split-row copy (gemm2.hip format) of a Linear weight [out][in]: B operand of the pre-split bf16x3 GEMM path
static void add_s_slot(rgm_dit* h, const std::string& key, int out_f, int in_f) {
  Slot s;
  s.off = h->arena_floats;
  s.numel = (size_t)out_f * in_f;
  s.set = true;
  s.t_rows = out_f;
  s.t_cols = in_f;
  h->slots[key + ".S"] = s;
  h->arena_floats += s.numel;
}

extern "C" int rgm_dit_create(const rgm_dit_cfg* c, rgm_dit** out) {
  RGM_REQUIRE(c && out, "dit_create: null argument");
  RGM_REQUIRE(c->hidden % c->heads == 0 && c->hidden % 32 == 0, "dit_create: hidden=%d heads=%d", c->hidden, c->heads);
  const int hd = c->hidden / c->heads;
  RGM_REQUIRE(hd == 64 || hd == 72, "dit_create: head_dim %d (64 or 72 supported)", hd);
  RGM_REQUIRE((c->in_ch * c->patch) % 32 == 0, "dit_create: in_ch*patch=%d must be a multiple of 32", c->in_ch * c->patch);
  RGM_REQUIRE(c->kind >= 0 && c->kind <= 2, "dit_create: kind %d", c->kind);
  RGM_REQUIRE(c->max_tokens > 0 && c->max_tokens <= 288, "dit_create: max_tokens %d", c->max_tokens);
  rgm_dit* h = new rgm_dit();
  h->cfg = *c;
  h->hd = hd;
  h->rot_half = (int)(hd * 0.5) / 2;  // rotary_dim = int(head_dim * 0.5), dit.py:571
  RGM_CHECK_HIP(hipGetDevice(&h->device));
  const size_t D = c->hidden;
  const int pc = c->in_ch * c->patch;
  if (c->kind != 0) add_slot(h, "cls_token", D);
  add_slot(h, "x_embedder.MLP.0.weight", 256 * (size_t)pc);
  add_slot(h, "x_embedder.MLP.0.bias", 256);
  add_slot(h, "x_embedder.MLP.2.weight", D * 256);
  add_slot(h, "x_embedder.MLP.2.bias", D);
  add_slot(h, "t_embedder.mlp.0.weight", D * 256);
  add_slot(h, "t_embedder.mlp.0.bias", D);
  add_slot(h, "t_embedder.mlp.2.weight", D * D);
  add_slot(h, "t_embedder.mlp.2.bias", D);
  if (c->kind == 0 && c->n_embed > 0) add_slot(h, "y_embedder.embedding_table.weight", (size_t)c->n_embed * D);
  add_slot(h, "rotary_emb.freqs", h->rot_half);
  // adaLN projections, contiguous across blocks (+ final layer) -> one GEMM
  for (int i = 0; i < c->depth; ++i) add_slot(h, "blocks." + std::to_string(i) + ".adaLN_modulation.1.weight", 6 * D * D);
  if (c->kind == 0) add_slot(h, "final_layer.adaLN_modulation.1.weight", 2 * D * D);
  for (int i = 0; i < c->depth; ++i) add_slot(h, "blocks." + std::to_string(i) + ".adaLN_modulation.1.bias", 6 * D);
  if (c->kind == 0) add_slot(h, "final_layer.adaLN_modulation.1.bias", 2 * D);
  h->ada_rows = (size_t)(6 * c->depth + (c->kind == 0 ? 2 : 0)) * D;
  for (int i = 0; i < c->depth; ++i) {
    const std::string b = "blocks." + std::to_string(i) + ".";
    add_slot(h, b + "attn.qkv.weight", 3 * D * D);
    add_slot(h, b + "attn.qkv.bias", 3 * D);
    add_slot(h, b + "attn.proj.weight", D * D);
    add_slot(h, b + "attn.proj.bias", D);
    add_slot(h, b + "mlp.fc1.weight", 4 * D * D);
    add_slot(h, b + "mlp.fc1.bias", 4 * D);
    add_slot(h, b + "mlp.fc2.weight", 4 * D * D);
    add_slot(h, b + "mlp.fc2.bias", D);
  }
  if (c->kind == 0) {
    add_slot(h, "final_layer.linear.weight", (size_t)c->patch * c->out_ch * D);
    add_slot(h, "final_layer.linear.bias", (size_t)c->patch * c->out_ch);
  } else {
    add_slot(h, "norm.weight", D);
    add_slot(h, "norm.bias", D);
    add_slot(h, "classifier_head.0.weight", D / 4 * D);
    add_slot(h, "classifier_head.0.bias", D / 4);
    add_slot(h, "classifier_head.2.weight", (size_t)c->n_out * (D / 4));
    add_slot(h, "classifier_head.2.bias", c->n_out);
    if (c->kind == 2) {
      add_slot(h, "norm_key.weight", D);
      add_slot(h, "norm_key.bias", D);
      add_slot(h, "classifier_head_key.0.weight", D / 4 * D);
      add_slot(h, "classifier_head_key.0.bias", D / 4);
      add_slot(h, "classifier_head_key.2.weight", 25 * (D / 4));
      add_slot(h, "classifier_head_key.2.bias", 25);
    }
  }
  {
    const int Di = c->hidden;
    for (int i = 0; i < c->depth; ++i) {
      const std::string b = "blocks." + std::to_string(i) + ".";
      add_s_slot(h, b + "attn.qkv.weight", 3 * Di, Di);
      add_s_slot(h, b + "attn.proj.weight", Di, Di);
      add_s_slot(h, b + "mlp.fc1.weight", 4 * Di, Di);
      add_s_slot(h, b + "mlp.fc2.weight", Di, 4 * Di);
    }
  }
  if (c->kind != 0) {              // classifiers are differentiated w.r.t. their input (guidance): keep W^T too
    const int Di = c->hidden;
    add_t_slot(h, "x_embedder.MLP.0.weight", 256, pc);
    add_t_slot(h, "x_embedder.MLP.2.weight", Di, 256);
    for (int i = 0; i < c->depth; ++i) {
      const std::string b = "blocks." + std::to_string(i) + ".";
      add_t_slot(h, b + "attn.qkv.weight", 3 * Di, Di);
      add_t_slot(h, b + "attn.proj.weight", Di, Di);
      add_t_slot(h, b + "mlp.fc1.weight", 4 * Di, Di);
      add_t_slot(h, b + "mlp.fc2.weight", Di, 4 * Di);
      for (const char* w : {"attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"}) add_ts_slot(h, b + w);
    }
    add_t_slot(h, "classifier_head.0.weight", Di / 4, Di);
    add_t_slot(h, "classifier_head.2.weight", c->n_out, Di / 4);
  }
  RGM_CHECK_HIP(hipMalloc(&h->arena, h->arena_floats * sizeof(float)));
  RGM_CHECK_HIP(hipMemset(h->arena, 0, h->arena_floats * sizeof(float)));
  // timestep frequencies, float32 op order of dit.py:59-61
  float tf[128];
  const float a = (float)(-log(10000.0));
  for (int k = 0; k < 128; ++k) tf[k] = expf((a * (float)k) / 128.0f);
  RGM_CHECK_HIP(hipMalloc(&h->tfreqs, sizeof(tf)));
  RGM_CHECK_HIP(hipMemcpy(h->tfreqs, tf, sizeof(tf), hipMemcpyHostToDevice));
  const size_t tab = (size_t)c->max_tokens * h->rot_half * sizeof(float);
  RGM_CHECK_HIP(hipMalloc(&h->cos_tab, tab));
  RGM_CHECK_HIP(hipMalloc(&h->sin_tab, tab));
  *out = h;
  return RGM_OK;
}

extern "C" void rgm_dit_destroy(rgm_dit* h) {
  if (!h) return;
  if (h->arena) (void)hipFree(h->arena);
  if (h->arena_t) (void)hipFree(h->arena_t);
  if (h->tfreqs) (void)hipFree(h->tfreqs);
  if (h->cos_tab) (void)hipFree(h->cos_tab);
  if (h->sin_tab) (void)hipFree(h->sin_tab);
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  for (int i = 0; i < 2; ++i) {
    if (h->side_x[i]) (void)hipStreamDestroy(h->side_x[i]);
    if (h->ev_join_x[i]) (void)hipEventDestroy(h->ev_join_x[i]);
  }
  delete h;
}

extern "C" int rgm_dit_set_param(rgm_dit* h, const char* key, const void* dptr, const int64_t* shape, int ndim) {
  RGM_REQUIRE(h && key && dptr, "dit_set_param: null argument");
  std::string k(key);
  // blocks.N.attn.rotary_emb.freqs alias the shared rotary_emb.freqs Parameter (dit.py:571-575)
  const std::string suffix = "attn.rotary_emb.freqs";
  if (k.size() > suffix.size() && k.compare(k.size() - suffix.size(), suffix.size(), suffix) == 0) k = "rotary_emb.freqs";
  auto it = h->slots.find(k);
  RGM_REQUIRE(it != h->slots.end(), "dit_set_param: unknown key '%s'", key);
  size_t numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
  RGM_REQUIRE(numel == it->second.numel, "dit_set_param: '%s' has %zu elements, expected %zu", key, numel, it->second.numel);
  RGM_CHECK_HIP(hipMemcpy(h->arena + it->second.off, dptr, numel * sizeof(float), hipMemcpyDeviceToDevice));
  it->second.set = true;
  auto tt = h->slots.find(k + ".T");
  if (tt != h->slots.end()) {
    const Slot& ts = tt->second;
    RGM_TRY(transpose_launch(h->arena + it->second.off, h->sp(ts), ts.t_cols, ts.t_rows, ts.t_ld, 1, 0));
    auto t2 = h->slots.find(k + ".TS");
    if (t2 != h->slots.end()) RGM_TRY(split_rows_launch(h->sp(ts), h->sp(t2->second), ts.t_rows, ts.t_ld, ts.t_ld, ts.t_ld, 0));
    RGM_CHECK_HIP(hipStreamSynchronize(0));
  }
  auto ss = h->slots.find(k + ".S");
  if (ss != h->slots.end()) {
    const Slot& sl = ss->second;
    RGM_TRY(split_rows_launch(h->arena + it->second.off, h->arena + sl.off, sl.t_rows, sl.t_cols, sl.t_cols, sl.t_cols, 0));
    RGM_CHECK_HIP(hipStreamSynchronize(0));
  }
  if (k == "rotary_emb.freqs") {
    // rotary-embedding-torch 0.3.2: angle = pos * freq in float32, then cos / sin
    std::vector<float> fr(h->rot_half), ct((size_t)h->cfg.max_tokens * h->rot_half), st(ct.size());
    RGM_CHECK_HIP(hipMemcpy(fr.data(), dptr, fr.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int t = 0; t < h->cfg.max_tokens; ++t)
      for (int j = 0; j < h->rot_half; ++j) {
        const float ang = (float)t * fr[j];
        ct[(size_t)t * h->rot_half + j] = cosf(ang);
        st[(size_t)t * h->rot_half + j] = sinf(ang);
      }
    RGM_CHECK_HIP(hipMemcpy(h->cos_tab, ct.data(), ct.size() * sizeof(float), hipMemcpyHostToDevice));
    RGM_CHECK_HIP(hipMemcpy(h->sin_tab, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
    h->rotary_ready = true;
  }
  return RGM_OK;
}

extern "C" int rgm_dit_missing_params(rgm_dit* h) {
  if (!h) return -1;
  int n = 0;
  for (auto& kv : h->slots) n += kv.second.set ? 0 : 1;
  return n;
}

// W^T copies of an eps-network's Linear weights, in their own allocation, so that the input-gradient entry point
// (rgm_dit_vjp: DPS guidance) can run its dgrad GEMMs.  Classifier handles have them from the start.  May be called
// before or after the parameters are set; costs (4 + 4 + 1 + 3) D^2 floats per block (1.8 GB at XL depth 28).
extern "C" int rgm_dit_enable_grad(rgm_dit* h) {
  RGM_REQUIRE(h, "dit_enable_grad: null handle");
  if (h->cfg.kind != 0 || h->arena_t) return RGM_OK;
  const rgm_dit_cfg& c = h->cfg;
  const int D = c.hidden, pc = c.in_ch * c.patch;
  std::vector<std::string> keys;
  auto add = [&](const std::string& key, int out_f, int in_f) {
    const int ld = (out_f + 31) / 32 * 32;
    Slot s;
    s.off = h->arena_t_floats;
    s.numel = (size_t)in_f * ld;
    s.set = true;
    s.t_rows = in_f; s.t_cols = out_f; s.t_ld = ld;
    s.in_t = true;
    h->slots[key + ".T"] = s;
    h->arena_t_floats += s.numel;
    keys.push_back(key);
  };
  add("x_embedder.MLP.0.weight", 256, pc);
  add("x_embedder.MLP.2.weight", D, 256);
  for (int i = 0; i < c.depth; ++i) {
    const std::string b = "blocks." + std::to_string(i) + ".";
    add(b + "attn.qkv.weight", 3 * D, D);
    add(b + "attn.proj.weight", D, D);
    add(b + "mlp.fc1.weight", 4 * D, D);
    add(b + "mlp.fc2.weight", D, 4 * D);
  }
  add("final_layer.linear.weight", c.patch * c.out_ch, D);
  for (int i = 0; i < c.depth; ++i) {
    const std::string b = "blocks." + std::to_string(i) + ".";
    for (const char* w : {"attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"}) add_ts_slot(h, b + w, true, &h->arena_t_floats);
  }
  RGM_CHECK_HIP(hipMalloc(&h->arena_t, h->arena_t_floats * sizeof(float)));
  RGM_CHECK_HIP(hipMemset(h->arena_t, 0, h->arena_t_floats * sizeof(float)));
  for (auto& k : keys) {
    const Slot& base = h->slots.at(k);
    if (!base.set) continue;                       // set_param fills the copy when the weight arrives
    const Slot& ts = h->slots.at(k + ".T");
    RGM_TRY(transpose_launch(h->arena + base.off, h->sp(ts), ts.t_cols, ts.t_rows, ts.t_ld, 1, 0));
    auto t2 = h->slots.find(k + ".TS");
    if (t2 != h->slots.end()) RGM_TRY(split_rows_launch(h->sp(ts), h->sp(t2->second), ts.t_rows, ts.t_ld, ts.t_ld, ts.t_ld, 0));
  }
  RGM_CHECK_HIP(hipStreamSynchronize(0));
  return RGM_OK;
}

namespace {
struct Ws {
  char* base;
  size_t off = 0, cap;
  Ws(void* b, size_t c) : base((char*)b), cap(c) {}
  float* take(size_t floats) {
    float* p = (float*)(base + off);
    off += align_up(floats * sizeof(float), 256);
    return p;
  }
};

struct Plan {
  int N, H, T0, T, M0, M;
  size_t L;
  float *tok_in, *h1, *x, *xm, *qkv, *ao, *hid, *temb, *c1, *c, *cs, *mod, *tok_out, *pool, *pooln, *z1;
  char* sk;          // split-K scratch of the pre-split GEMMs (gemm2_scratch_bytes)
  char* sk2;         // the same for the second half batch when the blocks run as two half batches on two streams
  char* sk_x[2];     // ... and for a third / fourth part
  size_t sk_bytes;
  size_t bytes;
};

Plan make_plan(const rgm_dit* h, int N, int H, void* ws, size_t cap) {
  const rgm_dit_cfg& c = h->cfg;
  Plan p{};
  p.N = N;
  p.H = H;
  p.T0 = H * c.width / c.patch;
  p.T = p.T0 + (c.kind != 0 ? 1 : 0);
  p.M0 = N * p.T0;
  p.M = N * p.T;
  p.L = h->ada_rows;
  const size_t D = c.hidden;
  Ws w(ws, cap);
  p.tok_in = w.take((size_t)p.M0 * c.in_ch * c.patch);
  p.h1 = w.take((size_t)p.M0 * 256);
  p.x = w.take((size_t)p.M * D);
  p.xm = w.take((size_t)p.M * D);
  p.qkv = w.take((size_t)p.M * 3 * D);
  p.ao = w.take((size_t)p.M * D);
  p.hid = w.take((size_t)p.M * 4 * D);
  p.temb = w.take((size_t)N * 256);
  p.c1 = w.take((size_t)N * D);
  p.c = w.take((size_t)N * D);
  p.cs = w.take((size_t)N * D);
  p.mod = w.take((size_t)N * p.L);
  p.tok_out = w.take((size_t)p.M0 * c.patch * (c.kind == 0 ? c.out_ch : 1));
  const size_t groups = 1 + (size_t)(H / c.width);
  p.pool = w.take((size_t)N * groups * D);
  p.pooln = w.take((size_t)N * groups * D);
  p.z1 = w.take((size_t)N * groups * (D / 4));
  p.sk_bytes = gemm2_scratch_bytes(p.M, 4 * (int)D);
  p.sk = reinterpret_cast<char*>(w.take(p.sk_bytes / sizeof(float)));
  p.sk2 = reinterpret_cast<char*>(w.take(p.sk_bytes / sizeof(float)));
  for (int i = 0; i < 2; ++i) p.sk_x[i] = g_dit_parts > 2 + i ? reinterpret_cast<char*>(w.take(p.sk_bytes / sizeof(float))) : nullptr;
  p.bytes = w.off;
  return p;
}

int lin(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K, int act,
        hipStream_t s) {
  GemmParams g;
  g.A = A; g.lda = lda; g.B = W; g.ldb = K; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act;
  return gemm_launch(g, s);
}

int lin_gated(const float* A, int lda, const float* W, const float* bias, float* X, int M, int N, int K,
              const float* gate, int gate_ld, int rows_per_gate, hipStream_t s) {
  GemmParams g;
  g.A = A; g.lda = lda; g.B = W; g.ldb = K; g.C = X; g.ldc = N;
  g.M = M; g.N = N; g.K = K; g.bias = bias;
  g.gate = gate; g.gate_ld = gate_ld; g.rows_per_gate = rows_per_gate;
  g.res = X; g.ldres = N;
  return gemm_launch(g, s);
}

// adaLN conditioning mod[N, L] = SiLU(c) . W_ada^T + b of ALL blocks (+ final layer): the weight-streaming kernel (adaln_stream.hip: exact fp32,
// one pass over the 0.9 GB at HBM rate) for up to 32 rows, the tiled GEMM otherwise
int cond_mod(const float* cs, const float* W, const float* bias, float* mod, int N, int D, int L, hipStream_t s) {
  const int rc = adaln_stream_launch(cs, W, bias, mod, N, D, L, L, s);
  if (rc <= 0) return rc;
  return lin(cs, D, W, bias, mod, L, N, L, D, 0, s);
}

// embedders + blocks; leaves the residual stream in plan.x and SiLU(c) modulation in plan.mod
// mod[i][:] = rows[idx[i]][:]  (conditioning rows computed ahead, rgm_dit_cond_rows; idx clamped to the table)
__global__ void gather_cond_rows_kernel(const float* __restrict__ rows, const int32_t* __restrict__ idx, float* __restrict__ mod, int L4, int U) {
  const int i = blockIdx.y;
  int u = idx[i];
  u = u < 0 ? 0 : (u >= U ? U - 1 : u);
  const float4* src = reinterpret_cast<const float4*>(rows) + (size_t)u * L4;
  float4* dst = reinterpret_cast<float4*>(mod) + (size_t)i * L4;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < L4; j += gridDim.x * blockDim.x) dst[j] = src[j];
}

// SiLU(t_embedder(t) + y_embedder(y)) for p.N rows -> p.cs  (dit.py:621-628)
static int cond_vector(rgm_dit* h, const Plan& p, const int64_t* t, const int32_t* y, hipStream_t s) {
  const rgm_dit_cfg& c = h->cfg;
  const int D = c.hidden;
  RGM_TRY(timestep_sincos_launch(t, h->tfreqs, p.temb, p.N, 128, s));
  RGM_TRY(lin(p.temb, 256, h->p("t_embedder.mlp.0.weight"), h->p("t_embedder.mlp.0.bias"), p.c1, D, p.N, D, 256, 1, s));
  RGM_TRY(lin(p.c1, D, h->p("t_embedder.mlp.2.weight"), h->p("t_embedder.mlp.2.bias"), p.c, D, p.N, D, D, 0, s));
  const float* ytab = (c.kind == 0 && c.n_embed > 0 && y) ? h->p("y_embedder.embedding_table.weight") : nullptr;
  return cond_finish_launch(p.c, ytab, y, p.cs, p.N, D, s);
}

// cond_rows / cond_idx (optional, eps-network): the adaLN modulation of every sample is row cond_idx[i] of a table computed ahead
// (rgm_dit_cond_rows) instead of a pass over the 0.9 GB of adaLN weights in this forward
int run_backbone(rgm_dit* h, const Plan& p, const float* x, const int64_t* t, const int32_t* y, hipStream_t s,
                 const float* cond_rows = nullptr, const int32_t* cond_idx = nullptr, int cond_U = 0) {
  const rgm_dit_cfg& c = h->cfg;
  const int D = c.hidden, pc = c.in_ch * c.patch, T = p.T, L = (int)p.L;
  RGM_TRY(patchify_launch(x, p.tok_in, p.N, c.in_ch, p.H, c.width, c.patch, s));
  RGM_TRY(lin(p.tok_in, pc, h->p("x_embedder.MLP.0.weight"), h->p("x_embedder.MLP.0.bias"), p.h1, 256, p.M0, 256, pc, 1, s));
  if (c.kind == 0) {
    RGM_TRY(lin(p.h1, 256, h->p("x_embedder.MLP.2.weight"), h->p("x_embedder.MLP.2.bias"), p.x, D, p.M0, D, 256, 0, s));
  } else {  // rows 1..T0 of every sample; row 0 is the cls token (dit.py:813)
    GemmParams g;
    g.A = p.h1; g.lda = 256; g.sA = (long long)p.T0 * 256;
    g.B = h->p("x_embedder.MLP.2.weight"); g.ldb = 256;
    g.C = p.x + D; g.ldc = D; g.sC = (long long)T * D;
    g.M = p.T0; g.N = D; g.K = 256; g.batch = p.N;
    g.bias = h->p("x_embedder.MLP.2.bias");
    RGM_TRY(gemm_launch(g, s));
    RGM_TRY(fill_cls_launch(h->p("cls_token"), p.x, p.N, T, D, s));
  }
  if (!cond_rows) RGM_TRY(cond_vector(h, p, t, y, s));
  // adaLN conditioning of the whole forward: mod[N, L] = cs . W_ada^T + b (all blocks' projections are contiguous in the arena).  Block 0
  // needs only its own 6 D columns up front; the other 0.9 GB of weights stream on the handle's side stream while block 0 computes
  // (the matrix pipes are the limit there and HBM idles) and are joined back before block 0's fc2, whose reduce writes block 1's LayerNorm.
  bool joined = true;
  if (cond_rows) {
    hipLaunchKernelGGL(gather_cond_rows_kernel, dim3(48, (unsigned)p.N), dim3(256), 0, s, cond_rows, cond_idx, p.mod, L / 4, cond_U);   // (L / 4 = 48 960 float4 for XL: 4 per thread)
    RGM_LAUNCH_CHECK();
  } else if (g_adaln_overlap && c.depth > 1 && L > 6 * D) {
    if (!h->side) {
      RGM_CHECK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
      RGM_CHECK_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
      RGM_CHECK_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    const float* W = h->p("blocks.0.adaLN_modulation.1.weight");
    const float* bv = h->p("blocks.0.adaLN_modulation.1.bias");
    RGM_TRY(lin(p.cs, D, W, bv, p.mod, L, p.N, 6 * D, D, 0, s));
    RGM_CHECK_HIP(hipEventRecord(h->ev_fork, s));
    RGM_CHECK_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    RGM_TRY(lin(p.cs, D, W + (size_t)6 * D * D, bv + 6 * D, p.mod + 6 * D, L, p.N, L - 6 * D, D, 0, h->side));
    RGM_CHECK_HIP(hipEventRecord(h->ev_join, h->side));
    joined = false;
  } else {
    RGM_TRY(cond_mod(p.cs, h->p("blocks.0.adaLN_modulation.1.weight"), h->p("blocks.0.adaLN_modulation.1.bias"), p.mod, p.N, D, L, s));
  }
  auto join = [&]() -> int {
    if (!joined) RGM_CHECK_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
    joined = true;
    return RGM_OK;
  };
  const bool v2 = rgm_get_gemm_precision() == 2;   // bf16x3 with pre-split operands: producers emit split rows, gemm2 consumes
  static const int dit_exp = RGM_EXP_ENV("RGM_DIT_EXP");   // timing experiments (common.h): 1 = fc1 without GELU/split, 2 = block-0 weights everywhere
  if (v2) {
    // every producer (adaLN-LayerNorm, attention, fc1's GELU epilogue) writes split rows; all four GEMMs run on the LDS-DMA kernel
    // (gemm2.hip picks the tile).  A "part" is a contiguous range of samples: the whole batch on the caller's stream, or -- from
    // g_dit_halves samples up -- two half batches, the second on the handle's side stream (every buffer is row-major over samples, so a
    // part is a pointer offset; each part has its own split-K scratch).  Block i of both parts is enqueued before block i + 1 of either.
    struct Part { int n0, n; hipStream_t st; char* sk; int xm_ready; };
    Part parts[4];
    int nparts = 1;
    parts[0] = Part{0, p.N, s, p.sk, 0};
    const bool halves = c.kind == 0 && dit_halves_for(p.N) && joined;
    if (halves) {
      if (!h->side) {
        RGM_CHECK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        RGM_CHECK_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        RGM_CHECK_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
      }
      nparts = g_dit_parts < 2 ? 2 : (g_dit_parts > 4 ? 4 : g_dit_parts);
      if (nparts > p.N) nparts = p.N;
      for (int k = 2; k < nparts; ++k) {
        if (!h->side_x[k - 2]) {
          RGM_CHECK_HIP(hipStreamCreateWithFlags(&h->side_x[k - 2], hipStreamNonBlocking));
          RGM_CHECK_HIP(hipEventCreateWithFlags(&h->ev_join_x[k - 2], hipEventDisableTiming));
        }
      }
      int n0 = 0;
      for (int k = 0; k < nparts; ++k) {                 // contiguous, near-equal: the first N % nparts parts take one sample more
        int n = p.N / nparts + (k < p.N % nparts ? 1 : 0);
        static const int first_n = (getenv("RGM_DIT_FIRST_PART") ? atoi(getenv("RGM_DIT_FIRST_PART")) : 0);      // experiments: an unequal pair (first part's samples)
        if (first_n > 0 && first_n < p.N && nparts == 2) n = k == 0 ? first_n : p.N - first_n;
        parts[k] = Part{n0, n, k == 0 ? s : (k == 1 ? h->side : h->side_x[k - 2]), k == 0 ? p.sk : (k == 1 ? p.sk2 : p.sk_x[k - 2]), 0};
        n0 += n;
      }
    }
    auto release = [&](int i, int k, int point) -> int {   // the second half may start: everything in front of it on s is done
      if (halves && i == 0 && k == 0 && point == dit_stagger()) {
        RGM_CHECK_HIP(hipEventRecord(h->ev_fork, s));
        for (int j = 1; j < nparts; ++j) RGM_CHECK_HIP(hipStreamWaitEvent(parts[j].st, h->ev_fork, 0));
      }
      return RGM_OK;
    };
    // (a lambda: an error return between the fork and the join below must not leave the side streams running into the shared workspace --
    // the caller's next launch on `s` would race with them)
    auto run_blocks = [&]() -> int {
    for (int i = 0; i < c.depth; ++i) {
      const std::string b = "blocks." + std::to_string((dit_exp & 2) ? 0 : i) + ".";
      for (int k = 0; k < nparts; ++k) {
        Part& q = parts[k];
        const size_t r0 = (size_t)q.n0 * T;
        const int Mq = q.n * T;
        const float* m = p.mod + (size_t)q.n0 * L + (size_t)i * 6 * D;   // this part's first sample, block i
        float* x = p.x + r0 * D;
        float* xm = p.xm + r0 * D;
        float* qkv = p.qkv + r0 * 3 * D;
        float* ao = p.ao + r0 * D;
        float* hid = p.hid + r0 * 4 * D;
        // next_mod (fc2 only): shift of the NEXT block's first adaLN-LayerNorm (its scale is D further).  A K-sliced fc2 then writes that
        // LayerNorm from its reduce kernel (GemmParams::ln_out) and *ln_done tells the loop to skip the separate launch.
        auto lin2 = [&](const float* A, const std::string& wkey, const float* bias, float* C, int N, int K, int act, int out_split,
                        const float* gate, const float* res, int tile, const float* next_mod = nullptr, int* ln_done = nullptr) {
          GemmParams g;
          g.tile = tile;
          g.co_sched = halves ? 1 : 0;
          g.A = A; g.lda = K; g.B = h->p(wkey + ".S"); g.ldb = K; g.C = C; g.ldc = N;
          g.M = Mq; g.N = N; g.K = K; g.bias = bias; g.act = act; g.out_split = out_split;
          g.sk_ws = q.sk; g.sk_ws_bytes = p.sk_bytes;
          if (gate) { g.gate = gate; g.gate_ld = L; g.rows_per_gate = T; g.res = res; g.ldres = N; }
          if (next_mod) {
            g.ln_out = xm; g.ln_shift = next_mod; g.ln_scale = next_mod + D; g.ln_mod_ld = L; g.ln_rows_per_batch = T;
            g.ln_out_split = 1; g.ln_eps = 1e-6f; g.ln_done = ln_done;
          }
          return gemm2_launch(g, q.st);
        };
        RGM_TRY(release(i, k, 0));
        if (!q.xm_ready) RGM_TRY(layernorm_modulate_launch(x, xm, Mq, D, 1e-6f, nullptr, nullptr, m, m + D, L, T, q.st, 1));
        q.xm_ready = 0;
        RGM_TRY(lin2(xm, b + "attn.qkv.weight", h->p(b + "attn.qkv.bias"), qkv, 3 * D, D, 0, 0, nullptr, nullptr, RGM_EXP_ENV("RGM_QKV_TILE")));
        RGM_TRY(release(i, k, 1));
        g_attn_co_sched = halves ? 1 : 0;          // attention_x3.hip: how far to split a (sample, head) over workgroups
        const int attn_rc = rotary_attention_fwd(qkv, ao, h->cos_tab, h->sin_tab, q.n, T, c.heads, h->hd, h->rot_half, q.st, 1);
        g_attn_co_sched = 0;
        RGM_TRY(attn_rc);
        RGM_TRY(release(i, k, 2));
        // (a K-sliced proj -- small batches, gemm2.hip g_t144 bit 16 -- writes the block's second adaLN-LayerNorm from its reduce kernel)
        int ln2_done = 0;
        RGM_TRY(lin2(ao, b + "attn.proj.weight", h->p(b + "attn.proj.bias"), x, D, D, 0, 0, m + 2 * D, x, 0, m + 3 * D, &ln2_done));
        RGM_TRY(release(i, k, 3));
        if (!ln2_done) RGM_TRY(layernorm_modulate_launch(x, xm, Mq, D, 1e-6f, nullptr, nullptr, m + 3 * D, m + 4 * D, L, T, q.st, 1));
        RGM_TRY(release(i, k, 4));
        RGM_TRY(lin2(xm, b + "mlp.fc1.weight", h->p(b + "mlp.fc1.bias"), hid, 4 * D, D, (dit_exp & 1) ? 0 : 2, (dit_exp & 1) ? 0 : 1, nullptr, nullptr,
                     RGM_EXP_ENV("RGM_FC1_TILE")));
        RGM_TRY(release(i, k, 5));
        if (!halves) RGM_TRY(join());          // block i + 1's shift / scale feed the fused reduce + LayerNorm of this fc2
        RGM_TRY(lin2(hid, b + "mlp.fc2.weight", h->p(b + "mlp.fc2.bias"), x, D, 4 * D, 0, 0, m + 5 * D, x, RGM_EXP_ENV("RGM_FC2_TILE"),
                     i + 1 < c.depth ? m + 6 * D : nullptr, &q.xm_ready));
      }
    }
    return RGM_OK;
    };
    const int blocks_rc = run_blocks();
    if (blocks_rc != RGM_OK) {
      g_attn_co_sched = 0;
      for (int k = 1; k < nparts; ++k) (void)hipStreamSynchronize(parts[k].st);
      if (!joined && h->side) (void)hipStreamSynchronize(h->side);
      return blocks_rc;
    }
    for (int k = 1; k < nparts; ++k) {
      hipEvent_t ev = k == 1 ? h->ev_join : h->ev_join_x[k - 2];
      RGM_CHECK_HIP(hipEventRecord(ev, parts[k].st));
      RGM_CHECK_HIP(hipStreamWaitEvent(s, ev, 0));
    }
    RGM_TRY(join());
    return RGM_OK;
  }
  for (int i = 0; i < c.depth; ++i) {
    const std::string b = "blocks." + std::to_string((dit_exp & 2) ? 0 : i) + ".";
    const float* m = p.mod + (size_t)i * 6 * D;
    if (i > 0) RGM_TRY(join());
    RGM_TRY(layernorm_modulate_launch(p.x, p.xm, p.M, D, 1e-6f, nullptr, nullptr, m, m + D, L, T, s));
    RGM_TRY(lin(p.xm, D, h->p(b + "attn.qkv.weight"), h->p(b + "attn.qkv.bias"), p.qkv, 3 * D, p.M, 3 * D, D, 0, s));
    RGM_TRY(rotary_attention_fwd(p.qkv, p.ao, h->cos_tab, h->sin_tab, p.N, T, c.heads, h->hd, h->rot_half, s));
    RGM_TRY(lin_gated(p.ao, D, h->p(b + "attn.proj.weight"), h->p(b + "attn.proj.bias"), p.x, p.M, D, D, m + 2 * D, L, T, s));
    RGM_TRY(layernorm_modulate_launch(p.x, p.xm, p.M, D, 1e-6f, nullptr, nullptr, m + 3 * D, m + 4 * D, L, T, s));
    RGM_TRY(lin(p.xm, D, h->p(b + "mlp.fc1.weight"), h->p(b + "mlp.fc1.bias"), p.hid, 4 * D, p.M, 4 * D, D, 2, s));
    RGM_TRY(lin_gated(p.hid, 4 * D, h->p(b + "mlp.fc2.weight"), h->p(b + "mlp.fc2.bias"), p.x, p.M, D, 4 * D, m + 5 * D, L, T, s));
  }
  RGM_TRY(join());              // depth 1: nothing above waited
  return RGM_OK;
}

// shapes and parameters of a call (everything but the workspace)
int check_call(rgm_dit* h, int N, int H) {
  RGM_REQUIRE(h && N > 0 && H > 0, "dit: bad arguments");
  const rgm_dit_cfg& c = h->cfg;
  RGM_REQUIRE((H * c.width) % c.patch == 0, "dit: H*W=%d not divisible by patch", H * c.width);
  const int T = H * c.width / c.patch + (c.kind != 0 ? 1 : 0);
  RGM_REQUIRE(T <= c.max_tokens, "dit: %d tokens exceed max_tokens=%d given at create", T, c.max_tokens);
  if (rgm_dit_missing_params(h) != 0 || !h->rotary_ready) {
    std::string miss;
    for (auto& kv : h->slots)
      if (!kv.second.set && miss.size() < 200) miss += kv.first + " ";
    set_error("dit: %d parameters not set: %s", rgm_dit_missing_params(h), miss.c_str());
    return RGM_ERR_STATE;
  }
  return RGM_OK;
}

int check_ready(rgm_dit* h, int N, int H, size_t ws_bytes, const void* ws, Plan* plan) {
  RGM_TRY(check_call(h, N, H));
  *plan = make_plan(h, N, H, const_cast<void*>(ws), ws_bytes);
  if (plan->bytes > ws_bytes || ws == nullptr) {
    set_error("dit: workspace %zu bytes < required %zu", ws_bytes, plan->bytes);
    return RGM_ERR_WORKSPACE;
  }
  RGM_REQUIRE(((uintptr_t)ws & 255) == 0, "dit: workspace must be 256-byte aligned");
  return RGM_OK;
}
}  // namespace

extern "C" size_t rgm_dit_workspace_bytes(const rgm_dit* h, int N, int H) {
  if (!h || N <= 0 || H <= 0) return 0;
  return make_plan(h, N, H, nullptr, 0).bytes;
}

static int forward_tail(rgm_dit* h, const Plan& p, float* eps, int N, int H, hipStream_t s);

extern "C" int rgm_dit_forward(rgm_dit* h, const float* x, const int64_t* t, const int32_t* y, float* eps, int N, int H,
                               void* ws, size_t ws_bytes, void* stream) {
  RGM_REQUIRE(h && h->cfg.kind == 0, "dit_forward: handle is not an eps-network");
  RGM_REQUIRE(x && t && eps, "dit_forward: null tensor");
  Plan p;
  RGM_TRY(check_ready(h, N, H, ws_bytes, ws, &p));
  hipStream_t s = (hipStream_t)stream;
  RGM_TRY(run_backbone(h, p, x, t, y, s));
  return forward_tail(h, p, eps, N, H, s);
}

// The adaLN modulation of U (t, y) pairs -- rows[U][(6 depth + 2) hidden], what rgm_dit_forward computes for a sample with that timestep and
// label -- in one pass over the adaLN weights per 32 pairs (adaln_stream.hip).  A sampler that knows its schedule asks for the next steps'
// rows at once and hands them to rgm_dit_forward_cond: the 0.9 GB of weights are then read once per group of steps, not once per step.
extern "C" int rgm_dit_cond_rows(rgm_dit* h, const int64_t* t, const int32_t* y, int U, int H, float* rows, void* ws, size_t ws_bytes,
                                 void* stream) {
  RGM_REQUIRE(h && h->cfg.kind == 0, "dit_cond_rows: handle is not an eps-network");
  RGM_REQUIRE(t && rows && ((uintptr_t)rows & 15) == 0, "dit_cond_rows: null or unaligned tensor");
  Plan p;
  RGM_TRY(check_ready(h, U, H, ws_bytes, ws, &p));
  hipStream_t s = (hipStream_t)stream;
  RGM_TRY(cond_vector(h, p, t, y, s));
  return cond_mod(p.cs, h->p("blocks.0.adaLN_modulation.1.weight"), h->p("blocks.0.adaLN_modulation.1.bias"), rows, U, h->cfg.hidden, (int)p.L, s);
}

// rgm_dit_forward with the conditioning of sample i taken from rows[idx[i]] (rgm_dit_cond_rows; idx on the device, 0 <= idx[i] < U)
extern "C" int rgm_dit_forward_cond(rgm_dit* h, const float* x, const float* rows, const int32_t* idx, int U, float* eps, int N, int H,
                                    void* ws, size_t ws_bytes, void* stream) {
  RGM_REQUIRE(h && h->cfg.kind == 0, "dit_forward_cond: handle is not an eps-network");
  RGM_REQUIRE(x && rows && idx && eps && U > 0 && ((uintptr_t)rows & 15) == 0, "dit_forward_cond: null or unaligned tensor");
  Plan p;
  RGM_TRY(check_ready(h, N, H, ws_bytes, ws, &p));
  hipStream_t s = (hipStream_t)stream;
  RGM_TRY(run_backbone(h, p, x, nullptr, nullptr, s, rows, idx, U));
  return forward_tail(h, p, eps, N, H, s);
}

static int forward_tail(rgm_dit* h, const Plan& p, float* eps, int N, int H, hipStream_t s) {
  const rgm_dit_cfg& c = h->cfg;
  const int D = c.hidden, L = (int)p.L;
  const float* m = p.mod + (size_t)c.depth * 6 * D;
  RGM_TRY(layernorm_modulate_launch(p.x, p.xm, p.M, D, 1e-6f, nullptr, nullptr, m, m + D, L, p.T, s));
  const int po = c.patch * c.out_ch;
  RGM_TRY(lin(p.xm, D, h->p("final_layer.linear.weight"), h->p("final_layer.linear.bias"), p.tok_out, po, p.M, po, D, 0, s));
  RGM_TRY(unpatchify_launch(p.tok_out, eps, N, c.out_ch, H, c.width, s));
  return RGM_OK;
}

static int run_head(rgm_dit* h, const Plan& p, const char* norm, const char* head, int rows, int n_out, float* out,
                    hipStream_t s) {
  const int D = h->cfg.hidden;
  const std::string nk(norm), hk(head);
  RGM_TRY(layernorm_modulate_launch(p.pool, p.pooln, rows, D, 1e-5f, h->p(nk + ".weight"), h->p(nk + ".bias"), nullptr, nullptr, 0, 1, s));
  RGM_TRY(lin(p.pooln, D, h->p(hk + ".0.weight"), h->p(hk + ".0.bias"), p.z1, D / 4, rows, D / 4, D, 1, s));
  RGM_TRY(lin(p.z1, D / 4, h->p(hk + ".2.weight"), h->p(hk + ".2.bias"), out, n_out, rows, n_out, D / 4, 0, s));
  return RGM_OK;
}

extern "C" int rgm_dit_classify(rgm_dit* h, const float* x, const int64_t* t, float* logits, float* key_out, int N, int H,
                                void* ws, size_t ws_bytes, void* stream) {
  RGM_REQUIRE(h && h->cfg.kind != 0, "dit_classify: handle is not a classifier");
  RGM_REQUIRE(x && t && logits, "dit_classify: null tensor");
  Plan p;
  RGM_TRY(check_ready(h, N, H, ws_bytes, ws, &p));
  hipStream_t s = (hipStream_t)stream;
  const rgm_dit_cfg& c = h->cfg;
  const int D = c.hidden;
  RGM_TRY(run_backbone(h, p, x, t, nullptr, s));
  if (c.kind == 1) {
    RGM_TRY(pool_rows_launch(p.x, p.pool, N, p.T, D, 0, 1, 1, s));
    return run_head(h, p, "norm", "classifier_head", N, c.n_out, logits, s);
  }
  if (key_out) {
    RGM_TRY(pool_rows_launch(p.x, p.pool, N, p.T, D, 0, 1, 1, s));
    RGM_TRY(run_head(h, p, "norm_key", "classifier_head_key", N, 25, key_out, s));
  }
  const int n_token = H / c.width;
  RGM_REQUIRE(n_token > 0 && p.T0 % n_token == 0, "dit_classify: H=%d gives %d chord windows", H, n_token);
  RGM_TRY(pool_rows_launch(p.x, p.pool, N, p.T, D, 1, n_token, p.T0 / n_token, s));
  return run_head(h, p, "norm", "classifier_head", N * n_token, c.n_out, logits, s);
}

// =====================================================================================================
// Classifier guidance: value and input gradient in one call (forward with saved activations + dgrad chain).
// Replaces th.autograd.grad(log_probs.sum(), x_in) of condition_functions.py:58-85.
// =====================================================================================================
namespace {
struct GPlan {
  int N, H, T0, T, M0, M, L, Kp, groups, cls;   // cls = 1 when row 0 of every sample is the class token (classifiers)
  float *tok_in, *zpre, *h1, *temb, *c1, *c, *cs, *mod;
  float *xs, *x1s, *qkvs, *aos, *pres, *lses;   // per-block saves (xs has depth+1 entries)
  float *xm, *hid, *dx, *dx1, *t1, *dbig, *dqkv, *dsmall;
  float *pool, *pooln, *z1pre, *z1, *logits, *dl, *dz1, *dpooln, *dpool, *dz, *dtin;
  char* sk;          // split-K scratch of the pre-split GEMMs (gemm2_scratch_bytes)
  size_t sk_bytes;
  size_t bytes;
};

GPlan gplan(const rgm_dit* h, int N, int H, void* ws) {
  const rgm_dit_cfg& c = h->cfg;
  GPlan p{};
  p.N = N; p.H = H;
  p.T0 = H * c.width / c.patch;
  p.cls = c.kind != 0 ? 1 : 0;
  p.T = p.T0 + p.cls;
  p.M0 = N * p.T0;
  p.M = N * p.T;
  p.L = (int)h->ada_rows;
  p.Kp = (c.n_out + 31) / 32 * 32;
  p.groups = c.kind == 2 ? H / c.width : 1;
  const size_t D = c.hidden, M = p.M, dep = c.depth;
  Ws w(ws, 0);
  p.tok_in = w.take((size_t)p.M0 * c.in_ch * c.patch);
  p.zpre = w.take((size_t)p.M0 * 256);
  p.h1 = w.take((size_t)p.M0 * 256);
  p.temb = w.take((size_t)N * 256);
  p.c1 = w.take(N * D); p.c = w.take(N * D); p.cs = w.take(N * D);
  p.mod = w.take((size_t)N * p.L);
  p.xs = w.take((dep + 1) * M * D);
  p.x1s = w.take(dep * M * D);
  p.qkvs = w.take(dep * M * 3 * D);
  p.aos = w.take(dep * M * D);
  p.pres = w.take(dep * M * 4 * D);
  p.lses = w.take(dep * (size_t)N * c.heads * p.T);
  p.xm = w.take(M * D);
  p.hid = w.take(M * 4 * D);
  p.dx = w.take(M * D); p.dx1 = w.take(M * D); p.t1 = w.take(M * D);
  p.dbig = w.take(M * 4 * D);
  p.dqkv = w.take(M * 3 * D);
  p.dsmall = w.take(M * D);
  const size_t R = (size_t)N * p.groups + (c.kind == 0 ? (size_t)p.M0 : 0);   // eps-network: pool doubles as the token-space output
  p.pool = w.take(R * D); p.pooln = w.take(R * D);
  p.z1pre = w.take(R * (D / 4)); p.z1 = w.take(R * (D / 4));
  p.logits = w.take(R * c.n_out);
  p.dl = w.take(R * p.Kp);
  p.dz1 = w.take(R * (D / 4)); p.dpooln = w.take(R * D); p.dpool = w.take(R * D);
  p.dz = w.take((size_t)p.M0 * 256);
  p.dtin = w.take((size_t)p.M0 * c.in_ch * c.patch);
  p.sk_bytes = gemm2_scratch_bytes(p.M, 4 * (int)D);
  p.sk = reinterpret_cast<char*>(w.take(p.sk_bytes / sizeof(float)));
  p.bytes = w.off;
  return p;
}

// dX[M, in] = dY[M, out(ld lda)] . W  using the stored W^T ([in][out_padded]); optional act-grad epilogue
int dgrad(rgm_dit* h, const std::string& wkey, const float* dY, int lda, float* dX, int ldc, int M, const float* aux, int ldaux,
          int act, hipStream_t s) {
  const Slot& ts = h->slots.at(wkey + ".T");
  GemmParams g;
  g.A = dY; g.lda = lda; g.B = h->sp(ts); g.ldb = ts.t_ld; g.C = dX; g.ldc = ldc;
  g.M = M; g.N = ts.t_rows; g.K = ts.t_ld;
  g.aux = aux; g.ldaux = ldaux; g.act = act;
  return gemm_launch(g, s);
}

// the same on the pre-split LDS-DMA kernel: dY is in split-row format, W^T comes from the ".TS" copy; dX fp32 or split
int dgrad2(rgm_dit* h, const std::string& wkey, const float* dY_split, int lda, float* dX, int ldc, int M, const float* aux, int ldaux,
           int act, int out_split, hipStream_t s) {
  const Slot& ts = h->slots.at(wkey + ".TS");
  GemmParams g;
  g.A = dY_split; g.lda = lda; g.B = h->sp(ts); g.ldb = ts.t_ld; g.C = dX; g.ldc = ldc;
  g.M = M; g.N = ts.t_rows; g.K = ts.t_ld;
  g.aux = aux; g.ldaux = ldaux; g.act = act; g.out_split = out_split;
  return gemm2_launch(g, s);
}

// Linear on the pre-split kernel: A split rows, weight from the ".S" copy; optional adaLN gate + residual
int lin_split(rgm_dit* h, const float* A_split, const std::string& wkey, const float* bias, float* C, int M, int N, int K, int act,
              int out_split, const float* gate, int gate_ld, int rows_per_gate, const float* res, hipStream_t s) {
  GemmParams g;
  g.A = A_split; g.lda = K; g.B = h->p(wkey + ".S"); g.ldb = K; g.C = C; g.ldc = N;
  g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act; g.out_split = out_split;
  g.sk_ws = h->sk_ws; g.sk_ws_bytes = h->sk_ws_bytes;
  if (gate) { g.gate = gate; g.gate_ld = gate_ld; g.rows_per_gate = rows_per_gate; g.res = res; g.ldres = N; }
  return gemm2_launch(g, s);
}
}  // namespace

extern "C" size_t rgm_dit_grad_workspace_bytes(const rgm_dit* h, int N, int H) {
  if (!h || N <= 0 || H <= 0) return 0;
  return gplan(h, N, H, nullptr).bytes;
}

// ---- the three shared pieces of the input-gradient paths (classifier guidance, eps-network VJP)
static int grad_forward(rgm_dit* h, const GPlan& p, const float* x, const int64_t* t, const int32_t* y, hipStream_t s) {
  const rgm_dit_cfg& c = h->cfg;
  const int N = p.N, H = p.H, D = c.hidden, pc = c.in_ch * c.patch, T = p.T, L = p.L, M = p.M;
  const size_t MD = (size_t)M * D;
  // ---------------- forward, keeping what the backward needs
  RGM_TRY(patchify_launch(x, p.tok_in, N, c.in_ch, H, c.width, c.patch, s));
  RGM_TRY(lin(p.tok_in, pc, h->p("x_embedder.MLP.0.weight"), h->p("x_embedder.MLP.0.bias"), p.zpre, 256, p.M0, 256, pc, 0, s));
  RGM_TRY(act_rows_launch(p.zpre, p.h1, p.M0, 256, 1, s));
  {
    GemmParams g;
    g.A = p.h1; g.lda = 256; g.sA = (long long)p.T0 * 256;
    g.B = h->p("x_embedder.MLP.2.weight"); g.ldb = 256;
    g.C = p.xs + (size_t)p.cls * D; g.ldc = D; g.sC = (long long)T * D;
    g.M = p.T0; g.N = D; g.K = 256; g.batch = N;
    g.bias = h->p("x_embedder.MLP.2.bias");
    RGM_TRY(gemm_launch(g, s));
    if (p.cls) RGM_TRY(fill_cls_launch(h->p("cls_token"), p.xs, N, T, D, s));
  }
  RGM_TRY(timestep_sincos_launch(t, h->tfreqs, p.temb, N, 128, s));
  RGM_TRY(lin(p.temb, 256, h->p("t_embedder.mlp.0.weight"), h->p("t_embedder.mlp.0.bias"), p.c1, D, N, D, 256, 1, s));
  RGM_TRY(lin(p.c1, D, h->p("t_embedder.mlp.2.weight"), h->p("t_embedder.mlp.2.bias"), p.c, D, N, D, D, 0, s));
  {
    const float* ytab = (c.kind == 0 && c.n_embed > 0 && y) ? h->p("y_embedder.embedding_table.weight") : nullptr;
    RGM_TRY(cond_finish_launch(p.c, ytab, y, p.cs, N, D, s));
  }
  RGM_TRY(cond_mod(p.cs, h->p("blocks.0.adaLN_modulation.1.weight"), h->p("blocks.0.adaLN_modulation.1.bias"), p.mod, N, D, L, s));
  const size_t lse_sz = (size_t)N * c.heads * T;
  const bool v2 = rgm_get_gemm_precision() == 2;
  h->sk_ws = p.sk;
  h->sk_ws_bytes = p.sk_bytes;
  for (int i = 0; i < c.depth; ++i) {
    const std::string b = "blocks." + std::to_string(i) + ".";
    const float* m = p.mod + (size_t)i * 6 * D;
    float* xi = p.xs + i * MD;
    float* x1 = p.x1s + i * MD;
    float* xn = p.xs + (i + 1) * MD;
    float* qkv = p.qkvs + i * MD * 3;
    float* ao = p.aos + i * MD;
    float* pre = p.pres + i * MD * 4;
    if (v2) {   // bf16x3_presplit: producers write split rows, every GEMM of the block runs on the LDS-DMA kernel
      RGM_TRY(layernorm_modulate_launch(xi, p.xm, M, D, 1e-6f, nullptr, nullptr, m, m + D, L, T, s, 1));
      RGM_TRY(lin_split(h, p.xm, b + "attn.qkv.weight", h->p(b + "attn.qkv.bias"), qkv, M, 3 * D, D, 0, 0, nullptr, 0, 1, nullptr, s));
      RGM_TRY(rotary_attention_fwd(qkv, ao, h->cos_tab, h->sin_tab, N, T, c.heads, h->hd, h->rot_half, s, 0, p.lses + i * lse_sz));
      RGM_TRY(split_rows_launch(ao, p.t1, M, D, D, D, s));          // the backward needs O in fp32, proj its split image
      RGM_TRY(lin_split(h, p.t1, b + "attn.proj.weight", h->p(b + "attn.proj.bias"), x1, M, D, D, 0, 0, m + 2 * D, L, T, xi, s));
      RGM_TRY(layernorm_modulate_launch(x1, p.xm, M, D, 1e-6f, nullptr, nullptr, m + 3 * D, m + 4 * D, L, T, s, 1));
      static const int fc1_dual = getenv("RGM_FC1_DUAL") ? atoi(getenv("RGM_FC1_DUAL")) : 1;   // 0: GEMM + elementwise pass (A/B runs)
      if (fc1_dual) {   // one launch: the pre-activation (fp32, kept for the backward) and its GELU as split rows (fc2's operand)
        GemmParams g;
        g.A = p.xm; g.lda = D; g.B = h->p(b + "mlp.fc1.weight.S"); g.ldb = D; g.C = pre; g.ldc = 4 * D;
        g.M = M; g.N = 4 * D; g.K = D; g.bias = h->p(b + "mlp.fc1.bias"); g.act = 2; g.out_split = 1;
        g.C2 = p.hid; g.ldc2 = 4 * D;
        RGM_TRY(gemm2_launch(g, s));
      } else {
        RGM_TRY(lin_split(h, p.xm, b + "mlp.fc1.weight", h->p(b + "mlp.fc1.bias"), pre, M, 4 * D, D, 0, 0, nullptr, 0, 1, nullptr, s));
        RGM_TRY(act_rows_launch(pre, p.hid, M, 4 * D, 2, s, 1));
      }
      RGM_TRY(lin_split(h, p.hid, b + "mlp.fc2.weight", h->p(b + "mlp.fc2.bias"), xn, M, D, 4 * D, 0, 0, m + 5 * D, L, T, x1, s));
      continue;
    }
    RGM_TRY(layernorm_modulate_launch(xi, p.xm, M, D, 1e-6f, nullptr, nullptr, m, m + D, L, T, s));
    RGM_TRY(lin(p.xm, D, h->p(b + "attn.qkv.weight"), h->p(b + "attn.qkv.bias"), qkv, 3 * D, M, 3 * D, D, 0, s));
    RGM_TRY(rotary_attention_fwd(qkv, ao, h->cos_tab, h->sin_tab, N, T, c.heads, h->hd, h->rot_half, s, 0, p.lses + i * lse_sz));
    {
      GemmParams g;
      g.A = ao; g.lda = D; g.B = h->p(b + "attn.proj.weight"); g.ldb = D; g.C = x1; g.ldc = D;
      g.M = M; g.N = D; g.K = D; g.bias = h->p(b + "attn.proj.bias");
      g.gate = m + 2 * D; g.gate_ld = L; g.rows_per_gate = T; g.res = xi; g.ldres = D;
      RGM_TRY(gemm_launch(g, s));
    }
    RGM_TRY(layernorm_modulate_launch(x1, p.xm, M, D, 1e-6f, nullptr, nullptr, m + 3 * D, m + 4 * D, L, T, s));
    RGM_TRY(lin(p.xm, D, h->p(b + "mlp.fc1.weight"), h->p(b + "mlp.fc1.bias"), pre, 4 * D, M, 4 * D, D, 0, s));
    RGM_TRY(act_rows_launch(pre, p.hid, M, 4 * D, 2, s));
    {
      GemmParams g;
      g.A = p.hid; g.lda = 4 * D; g.B = h->p(b + "mlp.fc2.weight"); g.ldb = 4 * D; g.C = xn; g.ldc = D;
      g.M = M; g.N = D; g.K = 4 * D; g.bias = h->p(b + "mlp.fc2.bias");
      g.gate = m + 5 * D; g.gate_ld = L; g.rows_per_gate = T; g.res = x1; g.ldres = D;
      RGM_TRY(gemm_launch(g, s));
    }
  }
  return RGM_OK;
}

// in: p.dx = gradient of the residual stream after the last block; out: p.dx = gradient of the embedded tokens
static int grad_blocks_backward(rgm_dit* h, const GPlan& p, hipStream_t s) {
  const rgm_dit_cfg& c = h->cfg;
  const int N = p.N, D = c.hidden, T = p.T, L = p.L, M = p.M;
  const size_t MD = (size_t)M * D;
  const size_t lse_sz = (size_t)N * c.heads * T;
  // ---------------- blocks, last to first
  for (int i = c.depth - 1; i >= 0; --i) {
    const std::string b = "blocks." + std::to_string(i) + ".";
    const float* m = p.mod + (size_t)i * 6 * D;
    const float* xi = p.xs + i * MD;
    const float* x1 = p.x1s + i * MD;
    if (rgm_get_gemm_precision() == 2) {   // same chain on the pre-split kernel: every GEMM operand is produced as split rows
      // the gated split operand of a dgrad GEMM comes out of the LayerNorm backward in front of it (one pass less each); only the very first
      // one, behind the head's backward, is a pass of its own (RGM_BWD_GATE_FUSE=0: all of them, for A/B runs)
      static const int gate_fuse = getenv("RGM_BWD_GATE_FUSE") ? atoi(getenv("RGM_BWD_GATE_FUSE")) : 1;
      if (!gate_fuse || i == c.depth - 1) RGM_TRY(gate_rows_launch(p.dx, m + 5 * D, p.t1, M, D, L, T, s, 1));
      RGM_TRY(dgrad2(h, b + "mlp.fc2.weight", p.t1, D, p.dbig, 4 * D, M, p.pres + i * MD * 4, 4 * D, 3, 1, s));
      RGM_TRY(dgrad2(h, b + "mlp.fc1.weight", p.dbig, 4 * D, p.dsmall, D, M, nullptr, 0, 0, 0, s));
      if (gate_fuse) {
        RGM_TRY(ln_mod_bwd_launch(p.dsmall, x1, p.dx, p.dx1, M, D, 1e-6f, nullptr, m + 4 * D, L, T, s, m + 2 * D, p.t1));
      } else {
        RGM_TRY(ln_mod_bwd_launch(p.dsmall, x1, p.dx, p.dx1, M, D, 1e-6f, nullptr, m + 4 * D, L, T, s));
        RGM_TRY(gate_rows_launch(p.dx1, m + 2 * D, p.t1, M, D, L, T, s, 1));
      }
      RGM_TRY(dgrad2(h, b + "attn.proj.weight", p.t1, D, p.dsmall, D, M, nullptr, 0, 0, 0, s));
      static const int bwd_split = getenv("RGM_ATTN_BWD_SPLIT") ? atoi(getenv("RGM_ATTN_BWD_SPLIT")) : 1;   // 0: fp32 rows + a split pass (A/B runs)
      if (bwd_split && D % 32 == 0) {   // d(qkv) leaves the attention backward as split rows: the operand of the dgrad GEMM below
        RGM_TRY(rotary_attention_bwd_launch(p.qkvs + i * MD * 3, p.aos + i * MD, p.dsmall, p.lses + i * lse_sz, p.dbig, h->cos_tab,
                                            h->sin_tab, N, T, c.heads, h->hd, h->rot_half, s, 1));
      } else {
        RGM_TRY(rotary_attention_bwd_launch(p.qkvs + i * MD * 3, p.aos + i * MD, p.dsmall, p.lses + i * lse_sz, p.dqkv, h->cos_tab,
                                            h->sin_tab, N, T, c.heads, h->hd, h->rot_half, s));
        RGM_TRY(split_rows_launch(p.dqkv, p.dbig, M, 3 * D, 3 * D, 3 * D, s));
      }
      RGM_TRY(dgrad2(h, b + "attn.qkv.weight", p.dbig, 3 * D, p.dsmall, D, M, nullptr, 0, 0, 0, s));
      if (gate_fuse && i > 0)     // ... with block i - 1's fc2 gate: the operand of the next iteration's first dgrad
        RGM_TRY(ln_mod_bwd_launch(p.dsmall, xi, p.dx1, p.dx, M, D, 1e-6f, nullptr, m + D, L, T, s, m - 6 * D + 5 * D, p.t1));
      else
        RGM_TRY(ln_mod_bwd_launch(p.dsmall, xi, p.dx1, p.dx, M, D, 1e-6f, nullptr, m + D, L, T, s));
      continue;
    }
    RGM_TRY(gate_rows_launch(p.dx, m + 5 * D, p.t1, M, D, L, T, s));                                   // d f2 = g2 * dx
    RGM_TRY(dgrad(h, b + "mlp.fc2.weight", p.t1, D, p.dbig, 4 * D, M, p.pres + i * MD * 4, 4 * D, 3, s)); // d pre = (. W2) * gelu'
    RGM_TRY(dgrad(h, b + "mlp.fc1.weight", p.dbig, 4 * D, p.dsmall, D, M, nullptr, 0, 0, s));            // d m2
    RGM_TRY(ln_mod_bwd_launch(p.dsmall, x1, p.dx, p.dx1, M, D, 1e-6f, nullptr, m + 4 * D, L, T, s));      // dx1 = dx + LN'
    RGM_TRY(gate_rows_launch(p.dx1, m + 2 * D, p.t1, M, D, L, T, s));                                   // d a = g1 * dx1
    RGM_TRY(dgrad(h, b + "attn.proj.weight", p.t1, D, p.dsmall, D, M, nullptr, 0, 0, s));               // d o
    RGM_TRY(rotary_attention_bwd_launch(p.qkvs + i * MD * 3, p.aos + i * MD, p.dsmall, p.lses + i * lse_sz, p.dqkv, h->cos_tab,
                                        h->sin_tab, N, T, c.heads, h->hd, h->rot_half, s));
    RGM_TRY(dgrad(h, b + "attn.qkv.weight", p.dqkv, 3 * D, p.dsmall, D, M, nullptr, 0, 0, s));          // d m1
    RGM_TRY(ln_mod_bwd_launch(p.dsmall, xi, p.dx1, p.dx, M, D, 1e-6f, nullptr, m + D, L, T, s));         // dx = dx1 + LN'
  }
  return RGM_OK;
}

// in: p.dx (token gradients); out: grad_x (N,in_ch,H,width)
static int grad_embed_backward(rgm_dit* h, const GPlan& p, float* grad_x, hipStream_t s) {
  const rgm_dit_cfg& c = h->cfg;
  const int N = p.N, H = p.H, D = c.hidden, pc = c.in_ch * c.patch, T = p.T;
  // ---------------- patch embedder backward (token rows 1..T0 of every sample), then un-patchify
  {
    const Slot& ts = h->slots.at("x_embedder.MLP.2.weight.T");
    GemmParams g;
    g.A = p.dx + (size_t)p.cls * D; g.lda = D; g.sA = (long long)T * D;
    g.B = h->sp(ts); g.ldb = ts.t_ld;
    g.C = p.dz; g.ldc = 256; g.sC = (long long)p.T0 * 256;
    g.M = p.T0; g.N = 256; g.K = D; g.batch = N;
    g.aux = p.zpre; g.ldaux = 256; g.sAux = (long long)p.T0 * 256; g.act = 4;
    RGM_TRY(gemm_launch(g, s));
  }
  RGM_TRY(dgrad(h, "x_embedder.MLP.0.weight", p.dz, 256, p.dtin, pc, p.M0, nullptr, 0, 0, s));
  RGM_TRY(unpatchify_launch(p.dtin, grad_x, N, c.in_ch, H, c.width, s));
  return RGM_OK;
}

// loss_kind 0: log p = -sum_k (logits - target)^2, target float (N, n_out)                  [grad_nn_zt_mse]
// loss_kind 1: log p = -sum_windows CE(chord_logits, target), target int64 (N, H/width)     [grad_nn_zt_chord, both=False]
//              on a plain classifier (kind 1): log p = log softmax(logits)[target], target int64 (N,)  [grad_nn_zt_xentropy]
// grad_x (N,in_ch,H,width) = d(sum log p)/dx * scale ; logits_out (N[,H/width], n_out) optional.
extern "C" int rgm_dit_cls_value_and_grad(rgm_dit* h, const float* x, const int64_t* t, const void* target, int loss_kind,
                                          float scale, float* logits_out, float* grad_x, int N, int H, void* ws,
                                          size_t ws_bytes, void* stream) {
  RGM_REQUIRE(h && h->cfg.kind != 0, "cls_value_and_grad: handle is not a classifier");
  RGM_REQUIRE(x && t && target && grad_x, "cls_value_and_grad: null tensor");
  RGM_REQUIRE((loss_kind == 0 && h->cfg.kind == 1) || (loss_kind == 1 && (h->cfg.kind == 2 || h->cfg.kind == 1)),
              "cls_value_and_grad: loss_kind %d does not match classifier kind %d", loss_kind, h->cfg.kind);
  RGM_TRY(check_call(h, N, H));
  const rgm_dit_cfg& c = h->cfg;
  GPlan p = gplan(h, N, H, ws);
  if (!ws || p.bytes > ws_bytes) {
    set_error("cls_value_and_grad: workspace %zu bytes < required %zu", ws_bytes, p.bytes);
    return RGM_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int D = c.hidden, T = p.T, M = p.M;
  const size_t MD = (size_t)M * D;
  RGM_TRY(grad_forward(h, p, x, t, nullptr, s));
  const float* xf = p.xs + (size_t)c.depth * MD;
  // ---------------- head forward + loss gradient
  const int rows = N * p.groups, per = c.kind == 2 ? p.T0 / p.groups : 1, first = c.kind == 2 ? 1 : 0;
  RGM_TRY(pool_rows_launch(xf, p.pool, N, T, D, first, p.groups, per, s));
  RGM_TRY(layernorm_modulate_launch(p.pool, p.pooln, rows, D, 1e-5f, h->p("norm.weight"), h->p("norm.bias"), nullptr, nullptr, 0, 1, s));
  RGM_TRY(lin(p.pooln, D, h->p("classifier_head.0.weight"), h->p("classifier_head.0.bias"), p.z1pre, D / 4, rows, D / 4, D, 0, s));
  RGM_TRY(act_rows_launch(p.z1pre, p.z1, rows, D / 4, 1, s));
  float* logits = logits_out ? logits_out : p.logits;
  RGM_TRY(lin(p.z1, D / 4, h->p("classifier_head.2.weight"), h->p("classifier_head.2.bias"), logits, c.n_out, rows, c.n_out, D / 4, 0, s));
  RGM_TRY(loss_grad_launch(logits, target, p.dl, rows, c.n_out, p.Kp, scale, loss_kind, s));
  // ---------------- head backward -> gradient of the residual stream
  RGM_TRY(dgrad(h, "classifier_head.2.weight", p.dl, p.Kp, p.dz1, D / 4, rows, p.z1pre, D / 4, 4, s));
  RGM_TRY(dgrad(h, "classifier_head.0.weight", p.dz1, D / 4, p.dpooln, D, rows, nullptr, 0, 0, s));
  RGM_TRY(ln_mod_bwd_launch(p.dpooln, p.pool, nullptr, p.dpool, rows, D, 1e-5f, h->p("norm.weight"), nullptr, 0, 1, s));
  RGM_TRY(scatter_rows_launch(p.dpool, p.dx, N, T, D, first, p.groups, per, s));
  RGM_TRY(grad_blocks_backward(h, p, s));
  return grad_embed_backward(h, p, grad_x, s);
}

// Input gradient of the eps-network (DPS guidance, gaussian_diffusion.py:415-465): eps = model(x, t, y) with saved
// activations, then grad_x = (d eps / d x)^T g_eps through final layer, blocks and embedder.  eps_out may be NULL.
extern "C" int rgm_dit_vjp(rgm_dit* h, const float* x, const int64_t* t, const int32_t* y, const float* g_eps, float* eps_out,
                           float* grad_x, int N, int H, void* ws, size_t ws_bytes, void* stream) {
  RGM_REQUIRE(h && h->cfg.kind == 0, "dit_vjp: handle is not an eps-network");
  // two phases sharing the workspace: forward (x, t given: activations are saved, eps_out written) and backward (g_eps,
  // grad_x given); a caller that needs eps before it can form g_eps (DPS: g_eps depends on the classifier's gradient at
  // x0(eps)) makes two calls -- (x, t, y, NULL, eps, NULL) then (NULL, NULL, NULL, g_eps, NULL, grad_x) -- with the same N, H, ws.
  RGM_REQUIRE((x && t) || (g_eps && grad_x), "dit_vjp: nothing to do");
  RGM_REQUIRE(!g_eps == !grad_x, "dit_vjp: g_eps and grad_x come together");
  RGM_REQUIRE(h->arena_t, "dit_vjp: call rgm_dit_enable_grad first (W^T copies are not kept by default)");
  RGM_TRY(check_call(h, N, H));
  const rgm_dit_cfg& c = h->cfg;
  GPlan p = gplan(h, N, H, ws);
  if (!ws || p.bytes > ws_bytes) {
    set_error("dit_vjp: workspace %zu bytes < required %zu", ws_bytes, p.bytes);
    return RGM_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int D = c.hidden, L = p.L, M = p.M, po = c.patch * c.out_ch;
  const float* xf = p.xs + (size_t)c.depth * M * D;
  const float* mf = p.mod + (size_t)c.depth * 6 * D;          // final-layer shift | scale
  if (x) {
    RGM_TRY(grad_forward(h, p, x, t, y, s));
    if (eps_out) {
      float* tok = p.pool;                                     // (M, patch*out_ch)
      RGM_TRY(layernorm_modulate_launch(xf, p.xm, M, D, 1e-6f, nullptr, nullptr, mf, mf + D, L, p.T, s));
      RGM_TRY(lin(p.xm, D, h->p("final_layer.linear.weight"), h->p("final_layer.linear.bias"), tok, po, M, po, D, 0, s));
      RGM_TRY(unpatchify_launch(tok, eps_out, N, c.out_ch, H, c.width, s));
    }
  }
  if (!g_eps) return RGM_OK;
  // final layer backward: d tok = patchify(g_eps) (unpatchify is a permutation), d xm = d tok . W, dx = LN'(d xm)
  RGM_TRY(patchify_launch(g_eps, p.pooln, N, c.out_ch, H, c.width, c.patch, s));
  RGM_TRY(dgrad(h, "final_layer.linear.weight", p.pooln, po, p.dsmall, D, M, nullptr, 0, 0, s));
  RGM_TRY(ln_mod_bwd_launch(p.dsmall, xf, nullptr, p.dx, M, D, 1e-6f, nullptr, mf + D, L, p.T, s));
  RGM_TRY(grad_blocks_backward(h, p, s));
  return grad_embed_backward(h, p, grad_x, s);
}



// 1: the adaLN conditioning of blocks 1.. runs on the handle's side stream under block 0 (fork / join by events on the caller's stream:
// still stream-ordered for the caller, capturable); 0 (default): one GEMM in front of block 0.  Same-box A/B at C2: 12.67 ms (0) vs
// 12.72 ms (1) -- the one-wave-per-SIMD GEMMs own every register of their CUs, the side launch only runs in their gaps
// Eps-network forwards of at least `min_batch` samples run their blocks as two half batches on two streams (0: never; -1, the default:
// the batch sizes where it measured ahead).  Returns the previous value through *prev when given.
extern "C" int rgm_set_dit_halves(int min_batch, int* prev) {
  RGM_REQUIRE(min_batch >= -1, "set_dit_halves: %d", min_batch);
  if (prev) *prev = g_dit_halves;
  g_dit_halves = min_batch;
  return RGM_OK;
}

extern "C" int rgm_set_adaln_overlap(int on) {
  RGM_REQUIRE(on == 0 || on == 1, "set_adaln_overlap: %d (0 / 1)", on);
  g_adaln_overlap = on;
  return RGM_OK;
}
