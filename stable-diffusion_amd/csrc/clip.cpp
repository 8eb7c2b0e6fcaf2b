// CLIP text transformer executor (the arithmetic behind FrozenCLIPEmbedder.forward,
// stable_diffusion.py:2553-2579: token ids -> transformers' CLIPTextModel -> last_hidden_state).
//
// Token + position embeddings, then `layers` pre-LayerNorm blocks
//   x += out_proj(causal_attention(q_proj(n), k_proj(n), v_proj(n))),  n = LayerNorm1(x)
//   x += fc2(quick_gelu(fc1(LayerNorm2(x))))
// and a final LayerNorm.  q/k/v projections run as one GEMM over the row-stacked weights; the
// residual adds ride in the GEMM epilogues; the residual stream stays in the compute dtype.
// Rows are (sample, token) pairs, so every GEMM has M = batch * tokens.
#include <math.h>

#include <memory>

#include "../../include/sdmi.h"
#include "executor.h"

namespace {
struct ClipLayerW {
  std::string prefix;
  const float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
  void* qkv_w = nullptr; float* qkv_b = nullptr;     // [3 hidden][hidden], bias [3 hidden]
  PackedConv out, fc1, fc2;
};
}  // namespace

struct sdmi_clip_text {
  sdmi_clip_cfg cfg;
  int dtype = SDMI_F16;
  std::vector<WeightSpec> specs;
  WeightStore store;
  DeviceMem mem;
  Workspace ws;
  bool finalized = false;
  std::vector<ClipLayerW> layers;
  const float *tok = nullptr, *pos = nullptr, *fin_g = nullptr, *fin_b = nullptr;

  ~sdmi_clip_text() { mem.release_all(); }
  void add_spec(const std::string& key, std::vector<int64_t> shape) { specs.push_back({key, std::move(shape)}); }

  void build_structure() {
    const int d = cfg.hidden, f = cfg.intermediate;
    add_spec("text_model.embeddings.token_embedding.weight", {cfg.vocab_size, d});
    add_spec("text_model.embeddings.position_embedding.weight", {cfg.max_positions, d});
    for (int i = 0; i < cfg.layers; ++i) {
      ClipLayerW l;
      l.prefix = "text_model.encoder.layers." + std::to_string(i);
      for (const char* n : {"q_proj", "k_proj", "v_proj", "out_proj"}) {
        add_spec(l.prefix + ".self_attn." + n + ".weight", {d, d});
        add_spec(l.prefix + ".self_attn." + n + ".bias", {d});
      }
      add_spec(l.prefix + ".layer_norm1.weight", {d}); add_spec(l.prefix + ".layer_norm1.bias", {d});
      add_spec(l.prefix + ".mlp.fc1.weight", {f, d}); add_spec(l.prefix + ".mlp.fc1.bias", {f});
      add_spec(l.prefix + ".mlp.fc2.weight", {d, f}); add_spec(l.prefix + ".mlp.fc2.bias", {d});
      add_spec(l.prefix + ".layer_norm2.weight", {d}); add_spec(l.prefix + ".layer_norm2.bias", {d});
      layers.push_back(l);
    }
    add_spec("text_model.final_layer_norm.weight", {d});
    add_spec("text_model.final_layer_norm.bias", {d});
  }

  int finalize(hipStream_t s) {
    if (finalized) { sdmi_set_error("finalize: already finalized"); return SDMI_ESTATE; }
    SDMI_TRY(store.check_complete(specs));
    SDMI_TRY(conv_gemm_init());
    SDMI_TRY(attention_init());
    SDMI_TRY(groupnorm_static_init());
    const int d = cfg.hidden, f = cfg.intermediate;
    const size_t esz = sdmi_dtype_size(dtype);
    std::vector<std::string> drop;
    auto P = [&](PackedConv* pc, const std::string& prefix, int cout, int cin) {
      pc->cin = cin; pc->cout = cout; pc->taps = 1;
      SDMI_TRY(mem.alloc(&pc->w, (size_t)cout * cin * esz));
      SDMI_TRY(pack_conv_weight(store.get(prefix + ".weight"), pc->w, cout, cin, 1, dtype, s));
      pc->bias = store.get(prefix + ".bias");
      drop.push_back(prefix + ".weight");
      return (int)SDMI_OK;
    };
    tok = store.get("text_model.embeddings.token_embedding.weight");
    pos = store.get("text_model.embeddings.position_embedding.weight");
    fin_g = store.get("text_model.final_layer_norm.weight");
    fin_b = store.get("text_model.final_layer_norm.bias");
    for (auto& l : layers) {
      const std::string a = l.prefix + ".self_attn.";
      l.ln1_g = store.get(l.prefix + ".layer_norm1.weight"); l.ln1_b = store.get(l.prefix + ".layer_norm1.bias");
      l.ln2_g = store.get(l.prefix + ".layer_norm2.weight"); l.ln2_b = store.get(l.prefix + ".layer_norm2.bias");
      SDMI_TRY(mem.alloc(&l.qkv_w, (size_t)3 * d * d * esz));
      void* qb;
      SDMI_TRY(mem.alloc(&qb, (size_t)3 * d * 4));
      l.qkv_b = (float*)qb;
      int part = 0;
      for (const char* n : {"q_proj", "k_proj", "v_proj"}) {
        SDMI_TRY(pack_conv_weight(store.get(a + n + ".weight"), (char*)l.qkv_w + (size_t)part * d * d * esz, d, d, 1,
                                  dtype, s));
        SDMI_CHECK_HIP(hipMemcpyAsync(l.qkv_b + (size_t)part * d, store.get(a + n + ".bias"), (size_t)d * 4,
                                      hipMemcpyDeviceToDevice, s));
        drop.push_back(a + n + ".weight"); drop.push_back(a + n + ".bias");
        ++part;
      }
      SDMI_TRY(P(&l.out, a + "out_proj", d, d));
      SDMI_TRY(P(&l.fc1, l.prefix + ".mlp.fc1", f, d));
      SDMI_TRY(P(&l.fc2, l.prefix + ".mlp.fc2", d, f));
    }
    SDMI_CHECK_HIP(hipStreamSynchronize(s));
    for (const auto& k : drop) store.drop(k);
    finalized = true;
    return SDMI_OK;
  }

  int linear(const void* a, int K, const void* w, const float* bias, int N, void* out, int M, const void* resid,
             hipStream_t s) {
    ConvGemmArgs g;
    g.a0 = a; g.c0 = K; g.w = w; g.out = out; g.ldo = N; g.bias = bias; g.resid = resid; g.ldr = N;
    g.B = 1; g.Hin = 1; g.Win = M; g.Ho = 1; g.Wo = M; g.taps = 1; g.N = N; g.dtype = dtype;
    g.splits = conv_gemm_pick_splits(M, N, K, dtype);
    if (g.splits > 1) g.partial = (float*)ws.get("splitk", (size_t)g.splits * M * N * 4);
    if (ws.dry) return SDMI_OK;
    return conv_gemm(g, s);
  }

  int run(const int64_t* ids, float* out, int B, int T, hipStream_t s) {
    const int d = cfg.hidden, f = cfg.intermediate, M = B * T;
    const size_t esz = sdmi_dtype_size(dtype);
    void* x = ws.get("x", (size_t)M * d * esz);
    void* n = ws.get("n", (size_t)M * d * esz);
    void* qkv = ws.get("qkv", (size_t)M * 3 * d * esz);
    void* att = ws.get("att", (size_t)M * d * esz);
    void* hid = ws.get("hid", (size_t)M * f * esz);
    if (!ws.dry) SDMI_TRY(embed_tokens(ids, tok, pos, x, M, T, d, cfg.vocab_size, dtype, s));
    const int hd = d / cfg.heads;
    for (const ClipLayerW& l : layers) {
      if (!ws.dry) SDMI_TRY(layernorm(x, n, l.ln1_g, l.ln1_b, M, d, cfg.ln_eps, dtype, s));
      SDMI_TRY(linear(n, d, l.qkv_w, l.qkv_b, 3 * d, qkv, M, nullptr, s));
      if (!ws.dry) {
        AttentionArgs a;
        a.q = qkv; a.k = (const char*)qkv + (size_t)d * esz; a.v = (const char*)qkv + (size_t)2 * d * esz;
        a.ldq = a.ldk = a.ldv = 3 * d; a.out = att; a.ldo = d;
        a.B = B; a.heads = cfg.heads; a.Nq = T; a.Nk = T; a.d = hd;
        a.scale = (float)pow((double)hd, -0.5); a.dtype = dtype; a.causal = 1;
        SDMI_TRY(attention(a, s));
      }
      SDMI_TRY(linear(att, d, l.out.w, l.out.bias, d, x, M, x, s));
      if (!ws.dry) SDMI_TRY(layernorm(x, n, l.ln2_g, l.ln2_b, M, d, cfg.ln_eps, dtype, s));
      SDMI_TRY(linear(n, d, l.fc1.w, l.fc1.bias, f, hid, M, nullptr, s));
      if (!ws.dry) SDMI_TRY(quick_gelu(hid, (int64_t)M * f, dtype, s));
      SDMI_TRY(linear(hid, f, l.fc2.w, l.fc2.bias, d, x, M, x, s));
    }
    if (ws.dry) return SDMI_OK;
    if (dtype == SDMI_F32) return layernorm(x, out, fin_g, fin_b, M, d, cfg.ln_eps, dtype, s);
    SDMI_TRY(layernorm(x, n, fin_g, fin_b, M, d, cfg.ln_eps, dtype, s));
    return convert_to_f32(n, out, (int64_t)M * d, dtype, s);
  }

  int forward(const int64_t* ids, float* out, int B, int T, hipStream_t s) {
    if (!finalized) { sdmi_set_error("forward before finalize"); return SDMI_ESTATE; }
    SDMI_REQUIRE(ids && out && B > 0 && T > 0, "clip forward: bad arguments");
    SDMI_REQUIRE(T <= cfg.max_positions, "clip forward: %d tokens exceed the %d positions", T, cfg.max_positions);
    ws.dry = true; ws.need.clear();
    SDMI_TRY(run(ids, out, B, T, s));
    ws.dry = false;
    if (!ws.fits()) {
      SDMI_CHECK_HIP(hipStreamSynchronize(s));
      SDMI_TRY(ws.commit());
    }
    return run(ids, out, B, T, s);
  }
};

extern "C" {

int sdmi_clip_text_create(const sdmi_clip_cfg* cfg, sdmi_clip_text** out) {
  SDMI_REQUIRE(cfg && out, "clip create: null argument");
  SDMI_REQUIRE(cfg->compute_dtype >= 0 && cfg->compute_dtype <= 2, "clip create: bad compute_dtype");
  SDMI_REQUIRE(cfg->vocab_size > 0 && cfg->layers > 0 && cfg->heads > 0 && cfg->max_positions > 0,
               "clip create: empty configuration");
  SDMI_REQUIRE(cfg->hidden % 64 == 0 && cfg->intermediate % 64 == 0 && cfg->hidden % cfg->heads == 0,
               "clip create: hidden and intermediate sizes must be multiples of 64");
  const int hd = cfg->hidden / cfg->heads;
  SDMI_REQUIRE(hd % 32 == 0 && (hd / 32 & (hd / 32 - 1)) == 0 && hd <= 2048, "clip create: head size %d unsupported", hd);
  std::unique_ptr<sdmi_clip_text> c(new sdmi_clip_text());
  c->cfg = *cfg;
  c->dtype = cfg->compute_dtype;
  c->store.mem = &c->mem;
  c->ws.mem = &c->mem;
  c->build_structure();
  *out = c.release();
  return SDMI_OK;
}
int sdmi_clip_text_num_weights(const sdmi_clip_text* c) { return c ? (int)c->specs.size() : 0; }
int sdmi_clip_text_weight_info(const sdmi_clip_text* c, int index, const char** key, int64_t* shape4, int* ndim) {
  SDMI_REQUIRE(c && index >= 0 && index < (int)c->specs.size(), "weight_info: bad index");
  const WeightSpec& s = c->specs[index];
  if (key) *key = s.key.c_str();
  if (ndim) *ndim = (int)s.shape.size();
  if (shape4)
    for (size_t i = 0; i < 4; ++i) shape4[i] = i < s.shape.size() ? s.shape[i] : 1;
  return SDMI_OK;
}
int sdmi_clip_text_set_weight(sdmi_clip_text* c, const char* key, const void* dev_ptr, const int64_t* shape, int ndim,
                              int dtype, void* stream) {
  SDMI_REQUIRE(c && key && dev_ptr && shape, "set_weight: null argument");
  if (c->finalized) { sdmi_set_error("set_weight after finalize"); return SDMI_ESTATE; }
  return c->store.set(c->specs, key, dev_ptr, shape, ndim, dtype, (hipStream_t)stream);
}
int sdmi_clip_text_finalize(sdmi_clip_text* c, void* stream) {
  SDMI_REQUIRE(c, "finalize: null handle");
  return c->finalize((hipStream_t)stream);
}
int sdmi_clip_text_forward(sdmi_clip_text* c, const int64_t* token_ids, void* out, int batch, int tokens, void* stream) {
  SDMI_REQUIRE(c, "forward: null handle");
  return c->forward(token_ids, (float*)out, batch, tokens, (hipStream_t)stream);
}
void sdmi_clip_text_destroy(sdmi_clip_text* c) { delete c; }

}  // extern "C"
