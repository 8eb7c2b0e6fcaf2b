// UNet denoiser executor: builds the block list from the configuration (same construction rules as
// the reference's UNetModel.__init__, stable_diffusion.py:3334-3496), owns packed weights, the
// per-prompt context K/V cache and the activation workspace, and enqueues one forward as a sequence
// of gfx950 kernels on the caller's stream (optionally replayed as a HIP graph).
//
// Layout decisions
//   * activations are NHWC in the compute dtype, so a feature map *is* its token matrix
//     [B*H*W][C]: the reference's `b c h w <-> b (h w) c` rearranges (:2891, :2894) vanish and 1x1
//     convolutions / Linear layers are the same GEMM;
//   * the skip-connection concat (:3545) is never materialised: GroupNorm and the implicit-GEMM
//     loader read the two source tensors directly;
//   * nearest-x2 upsampling (:3062) is folded into the following conv's loader;
//   * q/k/v of self-attention come from one fused [3d][d] GEMM; GEGLU is an epilogue of the d->8d GEMM;
//   * all 22 per-ResBlock time-embedding projections (:3217) are one batched tiny-M launch;
//   * context K/V (:2781-2782) are projected once per prompt (set_context) for all 16 blocks.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <memory>

#include "../../include/sdmi.h"
#include "executor.h"

namespace {

struct ResW {
  std::string prefix;
  int cin = 0, cout = 0;
  const float *gn1_g = nullptr, *gn1_b = nullptr, *gn2_g = nullptr, *gn2_b = nullptr;
  PackedConv conv1, conv2, skip;
  // 16-bit modes: the 1x1 skip_connection as extra K columns of the second 3x3 convolution ([cout][9 cout + cin],
  // bias = the sum of both), so that skip(x) + conv(h) (stable_diffusion.py:3234) is one launch
  PackedConv conv2f;
  bool has_skip = false;
  int emb_off = 0;          // column offset into the batched embedding projection
};

struct TBlockW {
  std::string prefix;
  const float *ln_g[3] = {nullptr, nullptr, nullptr}, *ln_b[3] = {nullptr, nullptr, nullptr};
  PackedConv qkv, out1, q2, kv2, out2, ff1, ff2;
  float* ff1_bias = nullptr;       // permuted to the GEGLU row order
  // LayerNorm folded into qkv / q2 / ff1 (16-bit modes): per output column sum_k of the packed weights (which then
  // carry gamma); the PackedConv's bias is b' = b + W beta
  float *qkv_c = nullptr, *q2_c = nullptr, *ff1_c = nullptr;
  void* kv_cache = nullptr;        // [B*ntok][2*inner]
  // cross-attention folded per prompt (16-bit modes): per sample G [heads*80][d] (scores = LayerNorm(x) G^T, keys
  // padded to 80 per head) with its bias g0 and column sums Gc ([heads*80] each), and U [d][heads*80] (the head
  // outputs already through to_out); see xattn_fold.hip
  void *xG = nullptr, *xU = nullptr;
  float *xg0 = nullptr, *xGc = nullptr;
};

struct STW {
  std::string prefix;
  int ch = 0, heads = 0, dh = 0, inner = 0;
  const float *gn_g = nullptr, *gn_b = nullptr;
  PackedConv proj_in, proj_out;
  std::vector<TBlockW> blocks;
};

enum LayerKind { L_CONV_IN, L_RES, L_ST, L_DOWN, L_UP };
struct Layer {
  LayerKind kind;
  int idx;               // index into the kind's vector
};

struct SimpleConv { std::string prefix; int ch = 0; PackedConv conv; };

constexpr int XSEG = 80;  // keys per head in the folded cross-attention (xattn_fold.hip)

struct Tensor {           // NHWC activation
  void* p = nullptr;
  int H = 0, W = 0, C = 0;
};

}  // namespace

struct sdmi_unet {
  sdmi_unet_cfg cfg;
  int dtype = SDMI_F16;
  int temb = 0;
  std::vector<WeightSpec> specs;
  WeightStore store;
  DeviceMem mem;
  Workspace ws;
  bool finalized = false;

  std::vector<ResW> res;
  std::vector<STW> sts;
  std::vector<SimpleConv> downs, ups;
  std::vector<std::vector<Layer>> input_blocks, output_blocks;
  std::vector<Layer> middle;
  std::vector<int> skip_channels;
  int final_ch = 0;

  // packed small weights
  float* inconv_w = nullptr; const float* inconv_b = nullptr;
  float* outconv_w = nullptr; const float* outconv_b = nullptr;
  const float *out_gn_g = nullptr, *out_gn_b = nullptr;
  const float *te0_w = nullptr, *te0_b = nullptr, *te2_w = nullptr, *te2_b = nullptr;
  float* emb_w_all = nullptr; float* emb_b_all = nullptr; int emb_total = 0;

  // context cache
  int ctx_batch = 0, ctx_ntok = 0;
  void* ctx_conv = nullptr; size_t ctx_conv_cap = 0;

  // every sample of a forward carries the same timestep (promise of the caller): embedding MLP on one row
  int uniform_t = 0;
  int emb_rows = 0;
  // samples b and b + B/2 of a forward have identical x, timestep and concat inputs and differ only in
  // their context (classifier-free guidance pairs; promise of the caller): everything up to the first
  // cross-attention is computed for one half and stored twice
  int cfg_dup = 0;
  int next_dup_rows = 0;        // consumed by the next gemm()
  // LayerNorm folded into the Linear behind it (decided at finalize: 16-bit compute types; SDMI_LN_FOLD=0 keeps
  // the separate kernel); next_ln_* are consumed by the next gemm()
  bool fold_ln = false;
  const float* next_ln_stats = nullptr;
  const float* next_ln_colsum = nullptr;
  int next_ln_nparts = 0;
  // LayerNorm row sums left by the GEMM that produces the LayerNorm's input (16-bit modes with the fold):
  // the next gemm() writes them here, last_rs_parts pairs per row
  bool producer_stats = false;
  float* next_row_stats_out = nullptr;
  int last_rs_parts = 0;
  // cross-attention as two GEMMs with per-prompt folded weights (decided at finalize: 16-bit modes with the
  // LayerNorm fold; SDMI_XATTN_FOLD=0 keeps to_q -> attention -> to_out).
  // next_w_rows / next_vec_per_sample / next_softmax are consumed by the next gemm()
  bool xfold = false;
  int xfold_min_d = 640;
  int next_w_rows = 0, next_vec_per_sample = 0, next_softmax = 0;
  // extra pointwise sources of the next gemm() (a skip_connection folded into a 3x3 convolution)
  const void *next_e0 = nullptr, *next_e1 = nullptr;
  int next_ec0 = 0, next_ec1 = 0;

  // graph replay
  int graph_mode = 1;
  struct GraphKey {
    const void* x; const void* c0; const void* c1; const void* t; void* out;
    int n_concat, cc0, cc1, B, H, W, uniform_t, cfg_dup; hipStream_t stream;
    bool operator==(const GraphKey& o) const { return memcmp(this, &o, sizeof(GraphKey)) == 0; }
  };
  GraphKey gkey; bool ghave = false;
  GraphKey gseen; bool gseen_have = false;   // key of the last eager forward: a graph is recorded on its second sighting
  hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
  hipStream_t cap_stream = nullptr;      // capture happens here; replay goes to the caller's stream
  int graph_nodes = 0;                   // kernel launches of the recorded forward

  ~sdmi_unet() {
    drop_graph();
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
    mem.release_all();
  }
  void drop_graph() {
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (graph) (void)hipGraphDestroy(graph);
    gexec = nullptr; graph = nullptr; ghave = false; graph_nodes = 0;
  }

  // ---- structure -----------------------------------------------------------------------------
  void add_spec(const std::string& key, std::vector<int64_t> shape) { specs.push_back({key, std::move(shape)}); }

  void heads_for(int ch, int* nh, int* dh) const {
    int h = cfg.num_heads;
    int d;
    if (cfg.num_head_channels == -1) d = ch / h;
    else { h = ch / cfg.num_head_channels; d = cfg.num_head_channels; }
    if (cfg.legacy) d = ch / h;
    *nh = h; *dh = d;
  }

  Layer add_res(const std::string& prefix, int cin, int cout) {
    ResW r; r.prefix = prefix; r.cin = cin; r.cout = cout; r.has_skip = cin != cout;
    r.emb_off = emb_total; emb_total += cout;
    add_spec(prefix + ".in_layers.0.weight", {cin}); add_spec(prefix + ".in_layers.0.bias", {cin});
    add_spec(prefix + ".in_layers.2.weight", {cout, cin, 3, 3}); add_spec(prefix + ".in_layers.2.bias", {cout});
    add_spec(prefix + ".emb_layers.1.weight", {cout, temb}); add_spec(prefix + ".emb_layers.1.bias", {cout});
    add_spec(prefix + ".out_layers.0.weight", {cout}); add_spec(prefix + ".out_layers.0.bias", {cout});
    add_spec(prefix + ".out_layers.3.weight", {cout, cout, 3, 3}); add_spec(prefix + ".out_layers.3.bias", {cout});
    if (r.has_skip) {
      add_spec(prefix + ".skip_connection.weight", {cout, cin, 1, 1});
      add_spec(prefix + ".skip_connection.bias", {cout});
    }
    res.push_back(r);
    return {L_RES, (int)res.size() - 1};
  }

  Layer add_st(const std::string& prefix, int ch) {
    STW s; s.prefix = prefix; s.ch = ch;
    heads_for(ch, &s.heads, &s.dh);
    s.inner = s.heads * s.dh;
    const int d = s.inner, cd = cfg.context_dim;
    add_spec(prefix + ".norm.weight", {ch}); add_spec(prefix + ".norm.bias", {ch});
    add_spec(prefix + ".proj_in.weight", {d, ch, 1, 1}); add_spec(prefix + ".proj_in.bias", {d});
    for (int b = 0; b < cfg.transformer_depth; ++b) {
      TBlockW tb; tb.prefix = prefix + ".transformer_blocks." + std::to_string(b);
      const std::string& q = tb.prefix;
      add_spec(q + ".attn1.to_q.weight", {d, d}); add_spec(q + ".attn1.to_k.weight", {d, d});
      add_spec(q + ".attn1.to_v.weight", {d, d});
      add_spec(q + ".attn1.to_out.0.weight", {d, d}); add_spec(q + ".attn1.to_out.0.bias", {d});
      add_spec(q + ".ff.net.0.proj.weight", {8 * d, d}); add_spec(q + ".ff.net.0.proj.bias", {8 * d});
      add_spec(q + ".ff.net.2.weight", {d, 4 * d}); add_spec(q + ".ff.net.2.bias", {d});
      add_spec(q + ".attn2.to_q.weight", {d, d}); add_spec(q + ".attn2.to_k.weight", {d, cd});
      add_spec(q + ".attn2.to_v.weight", {d, cd});
      add_spec(q + ".attn2.to_out.0.weight", {d, d}); add_spec(q + ".attn2.to_out.0.bias", {d});
      for (int n = 1; n <= 3; ++n) {
        add_spec(q + ".norm" + std::to_string(n) + ".weight", {d});
        add_spec(q + ".norm" + std::to_string(n) + ".bias", {d});
      }
      s.blocks.push_back(tb);
    }
    add_spec(prefix + ".proj_out.weight", {ch, d, 1, 1}); add_spec(prefix + ".proj_out.bias", {ch});
    sts.push_back(s);
    return {L_ST, (int)sts.size() - 1};
  }

  Layer add_simple(std::vector<SimpleConv>& v, LayerKind kind, const std::string& prefix, int ch) {
    SimpleConv c; c.prefix = prefix; c.ch = ch;
    add_spec(prefix + ".weight", {ch, ch, 3, 3}); add_spec(prefix + ".bias", {ch});
    v.push_back(c);
    return {kind, (int)v.size() - 1};
  }

  int build_structure() {
    const int mc = cfg.model_channels;
    temb = 4 * mc;
    add_spec("time_embed.0.weight", {temb, mc}); add_spec("time_embed.0.bias", {temb});
    add_spec("time_embed.2.weight", {temb, temb}); add_spec("time_embed.2.bias", {temb});
    add_spec("input_blocks.0.0.weight", {mc, cfg.in_channels, 3, 3}); add_spec("input_blocks.0.0.bias", {mc});
    input_blocks.push_back({{L_CONV_IN, 0}});
    std::vector<int> chans = {mc};
    int ch = mc, ds = 1, idx = 1;
    auto attn_here = [&](int d) {
      for (int i = 0; i < cfg.n_attention_resolutions; ++i)
        if (cfg.attention_resolutions[i] == d) return true;
      return false;
    };
    for (int level = 0; level < cfg.n_channel_mult; ++level) {
      const int mult = cfg.channel_mult[level];
      for (int r = 0; r < cfg.num_res_blocks; ++r) {
        std::vector<Layer> layers;
        const std::string bp = "input_blocks." + std::to_string(idx);
        layers.push_back(add_res(bp + ".0", ch, mult * mc));
        ch = mult * mc;
        if (attn_here(ds)) layers.push_back(add_st(bp + ".1", ch));
        input_blocks.push_back(layers);
        chans.push_back(ch);
        ++idx;
      }
      if (level != cfg.n_channel_mult - 1) {
        input_blocks.push_back({add_simple(downs, L_DOWN, "input_blocks." + std::to_string(idx) + ".0.op", ch)});
        chans.push_back(ch);
        ++idx;
        ds *= 2;
      }
    }
    skip_channels = chans;
    middle.push_back(add_res("middle_block.0", ch, ch));
    middle.push_back(add_st("middle_block.1", ch));
    middle.push_back(add_res("middle_block.2", ch, ch));
    idx = 0;
    for (int level = cfg.n_channel_mult - 1; level >= 0; --level) {
      const int mult = cfg.channel_mult[level];
      for (int i = 0; i <= cfg.num_res_blocks; ++i) {
        const int ich = chans.back(); chans.pop_back();
        std::vector<Layer> layers;
        const std::string bp = "output_blocks." + std::to_string(idx);
        layers.push_back(add_res(bp + ".0", ch + ich, mult * mc));
        ch = mult * mc;
        if (attn_here(ds)) layers.push_back(add_st(bp + "." + std::to_string(layers.size()), ch));
        if (level && i == cfg.num_res_blocks) {
          layers.push_back(add_simple(ups, L_UP, bp + "." + std::to_string(layers.size()) + ".conv", ch));
          ds /= 2;
        }
        output_blocks.push_back(layers);
        ++idx;
      }
    }
    final_ch = ch;
    add_spec("out.0.weight", {final_ch}); add_spec("out.0.bias", {final_ch});
    add_spec("out.2.weight", {cfg.out_channels, mc, 3, 3}); add_spec("out.2.bias", {cfg.out_channels});
    return SDMI_OK;
  }

  // ---- weight packing --------------------------------------------------------------------------
  int pack(PackedConv* pc, const std::string& wkey, const std::string& bkey, int cout, int cin, int taps,
           hipStream_t s) {
    pc->cin = cin; pc->cout = cout; pc->taps = taps;
    SDMI_TRY(mem.alloc(&pc->w, (size_t)cout * cin * taps * sdmi_dtype_size(dtype)));
    SDMI_TRY(pack_conv_weight(store.get(wkey), pc->w, cout, cin, taps, dtype, s));
    pc->bias = bkey.empty() ? nullptr : store.get(bkey);
    return SDMI_OK;
  }
  // several (rows_i, cin) matrices stacked along the output dimension into one GEMM weight
  int pack_stacked(PackedConv* pc, const std::vector<std::string>& wkeys, int rows_each, int cin, hipStream_t s) {
    pc->cin = cin; pc->cout = rows_each * (int)wkeys.size(); pc->taps = 1; pc->bias = nullptr;
    const size_t esz = sdmi_dtype_size(dtype);
    SDMI_TRY(mem.alloc(&pc->w, (size_t)pc->cout * cin * esz));
    for (size_t i = 0; i < wkeys.size(); ++i)
      SDMI_TRY(pack_conv_weight(store.get(wkeys[i]), (char*)pc->w + i * (size_t)rows_each * cin * esz, rows_each,
                                cin, 1, dtype, s));
    return SDMI_OK;
  }

  int finalize(hipStream_t s) {
    if (finalized) { sdmi_set_error("finalize: already finalized"); return SDMI_ESTATE; }
    SDMI_TRY(store.check_complete(specs));
    SDMI_TRY(conv_gemm_init());
    SDMI_TRY(elementwise_init());
    SDMI_TRY(attention_init());
    SDMI_TRY(groupnorm_static_init());
    std::vector<std::string> packed_keys;
    {
      const char* e = getenv("SDMI_LN_FOLD");
      fold_ln = dtype != SDMI_F32 && !(e && atoi(e) == 0);
      const char* e2 = getenv("SDMI_LN_PRODUCER_STATS");
      producer_stats = fold_ln && !(e2 && atoi(e2) == 0);
      const char* e3 = getenv("SDMI_XATTN_FOLD");
      xfold = fold_ln && !(e3 && atoi(e3) == 0);
      if (const char* e4 = getenv("SDMI_XATTN_FOLD_MIN_D")) xfold_min_d = atoi(e4);
    }
    // scratch for one folded fp32 matrix (the widest is the GEGLU projection, 8d x d) and its bias
    float *fold_w = nullptr, *fold_b = nullptr;
    if (fold_ln) {
      size_t widest = 0;
      for (auto& st : sts) widest = std::max(widest, (size_t)8 * st.inner * st.inner);
      void* q;
      SDMI_TRY(mem.alloc(&q, widest * 4)); fold_w = (float*)q;
      SDMI_TRY(mem.alloc(&q, (size_t)8 * 4096 * 4)); fold_b = (float*)q;
    }
    // W diag(gamma) packed into `dst` rows [row0, row0 + rows), b' = W beta (+ bias) into bprime[row0 ...]
    auto fold_into = [&](void* dst, float* bprime, int row0, const std::string& wkey, const std::string& bkey,
                         const float* g, const float* b, int rows, int cin) -> int {
      SDMI_TRY(fold_ln_weight(store.get(wkey), g, b, bkey.empty() ? nullptr : store.get(bkey), fold_w, bprime + row0,
                              rows, cin, s));
      return pack_conv_weight(fold_w, (char*)dst + (size_t)row0 * cin * sdmi_dtype_size(dtype), rows, cin, 1, dtype, s);
    };
    auto alloc_f = [&](float** out, size_t n) -> int {
      void* q;
      SDMI_TRY(mem.alloc(&q, n * 4));
      *out = (float*)q;
      return SDMI_OK;
    };
    auto P = [&](PackedConv* pc, const std::string& prefix, int cout, int cin, int taps, bool bias = true) {
      packed_keys.push_back(prefix + ".weight");
      return pack(pc, prefix + ".weight", bias ? prefix + ".bias" : std::string(), cout, cin, taps, s);
    };
    te0_w = store.get("time_embed.0.weight"); te0_b = store.get("time_embed.0.bias");
    te2_w = store.get("time_embed.2.weight"); te2_b = store.get("time_embed.2.bias");
    {
      void* p;
      SDMI_TRY(mem.alloc(&p, (size_t)9 * cfg.in_channels * cfg.model_channels * 4));
      inconv_w = (float*)p;
      SDMI_TRY(pack_inconv_weight(store.get("input_blocks.0.0.weight"), inconv_w, cfg.model_channels,
                                  cfg.in_channels, s));
      inconv_b = store.get("input_blocks.0.0.bias");
      packed_keys.push_back("input_blocks.0.0.weight");
      SDMI_TRY(mem.alloc(&p, (size_t)9 * cfg.model_channels * cfg.out_channels * 4));
      outconv_w = (float*)p;
      SDMI_TRY(pack_outconv_weight(store.get("out.2.weight"), outconv_w, cfg.out_channels, cfg.model_channels, s));
      outconv_b = store.get("out.2.bias");
      packed_keys.push_back("out.2.weight");
      out_gn_g = store.get("out.0.weight"); out_gn_b = store.get("out.0.bias");
      SDMI_TRY(mem.alloc(&p, (size_t)emb_total * temb * 4)); emb_w_all = (float*)p;
      SDMI_TRY(mem.alloc(&p, (size_t)emb_total * 4)); emb_b_all = (float*)p;
    }
    for (auto& r : res) {
      const std::string& p = r.prefix;
      r.gn1_g = store.get(p + ".in_layers.0.weight"); r.gn1_b = store.get(p + ".in_layers.0.bias");
      r.gn2_g = store.get(p + ".out_layers.0.weight"); r.gn2_b = store.get(p + ".out_layers.0.bias");
      SDMI_TRY(P(&r.conv1, p + ".in_layers.2", r.cout, r.cin, 9));
      SDMI_TRY(P(&r.conv2, p + ".out_layers.3", r.cout, r.cout, 9));
      if (r.has_skip) SDMI_TRY(P(&r.skip, p + ".skip_connection", r.cout, r.cin, 1));
      if (r.has_skip && dtype != SDMI_F32 && !(getenv("SDMI_FUSE_SKIP") && atoi(getenv("SDMI_FUSE_SKIP")) == 0)) {
        const size_t esz = sdmi_dtype_size(dtype);
        const size_t k2 = (size_t)9 * r.cout, kf = k2 + r.cin;
        r.conv2f.cin = r.cout; r.conv2f.cout = r.cout; r.conv2f.taps = 9;
        SDMI_TRY(mem.alloc(&r.conv2f.w, (size_t)r.cout * kf * esz));
        SDMI_CHECK_HIP(hipMemcpy2DAsync(r.conv2f.w, kf * esz, r.conv2.w, k2 * esz, k2 * esz, r.cout,
                                        hipMemcpyDeviceToDevice, s));
        SDMI_CHECK_HIP(hipMemcpy2DAsync((char*)r.conv2f.w + k2 * esz, kf * esz, r.skip.w, (size_t)r.cin * esz,
                                        (size_t)r.cin * esz, r.cout, hipMemcpyDeviceToDevice, s));
        void* fb;
        SDMI_TRY(mem.alloc(&fb, (size_t)r.cout * 4));
        SDMI_TRY(axpby(r.conv2.bias, r.skip.bias, 1.0f, 1.0f, (float*)fb, r.cout, s));
        r.conv2f.bias = (float*)fb;
      }
      SDMI_CHECK_HIP(hipMemcpyAsync(emb_w_all + (size_t)r.emb_off * temb, store.get(p + ".emb_layers.1.weight"),
                                    (size_t)r.cout * temb * 4, hipMemcpyDeviceToDevice, s));
      SDMI_CHECK_HIP(hipMemcpyAsync(emb_b_all + r.emb_off, store.get(p + ".emb_layers.1.bias"), (size_t)r.cout * 4,
                                    hipMemcpyDeviceToDevice, s));
      packed_keys.push_back(p + ".emb_layers.1.weight");
      packed_keys.push_back(p + ".emb_layers.1.bias");
    }
    for (auto& st : sts) {
      const std::string& p = st.prefix;
      const int d = st.inner;
      st.gn_g = store.get(p + ".norm.weight"); st.gn_b = store.get(p + ".norm.bias");
      SDMI_TRY(P(&st.proj_in, p + ".proj_in", d, st.ch, 1));
      SDMI_TRY(P(&st.proj_out, p + ".proj_out", st.ch, d, 1));
      for (auto& tb : st.blocks) {
        const std::string& q = tb.prefix;
        for (int n = 0; n < 3; ++n) {
          tb.ln_g[n] = store.get(q + ".norm" + std::to_string(n + 1) + ".weight");
          tb.ln_b[n] = store.get(q + ".norm" + std::to_string(n + 1) + ".bias");
        }
        std::vector<std::string> qkv = {q + ".attn1.to_q.weight", q + ".attn1.to_k.weight", q + ".attn1.to_v.weight"};
        std::vector<std::string> kv = {q + ".attn2.to_k.weight", q + ".attn2.to_v.weight"};
        SDMI_TRY(pack_stacked(&tb.kv2, kv, d, cfg.context_dim, s));
        SDMI_TRY(P(&tb.out1, q + ".attn1.to_out.0", d, d, 1));
        SDMI_TRY(P(&tb.out2, q + ".attn2.to_out.0", d, d, 1));
        SDMI_TRY(P(&tb.ff2, q + ".ff.net.2", d, 4 * d, 1));
        tb.ff1.cin = d; tb.ff1.cout = 8 * d; tb.ff1.taps = 1;
        SDMI_TRY(mem.alloc(&tb.ff1.w, (size_t)8 * d * d * sdmi_dtype_size(dtype)));
        SDMI_TRY(alloc_f(&tb.ff1_bias, (size_t)8 * d));
        const size_t esz_w = sdmi_dtype_size(dtype);
        if (fold_ln) {
          SDMI_REQUIRE(8 * d <= 8 * 4096, "%s: inner dimension %d too wide for the LayerNorm fold", q.c_str(), d);
          // norm1 -> fused q/k/v projection
          tb.qkv.cin = d; tb.qkv.cout = 3 * d; tb.qkv.taps = 1;
          SDMI_TRY(mem.alloc(&tb.qkv.w, (size_t)3 * d * d * esz_w));
          float* b;
          SDMI_TRY(alloc_f(&b, (size_t)3 * d));
          for (int i = 0; i < 3; ++i)
            SDMI_TRY(fold_into(tb.qkv.w, b, i * d, qkv[i], "", tb.ln_g[0], tb.ln_b[0], d, d));
          tb.qkv.bias = b;
          SDMI_TRY(alloc_f(&tb.qkv_c, (size_t)3 * d));
          SDMI_TRY(packed_rowsum(tb.qkv.w, tb.qkv_c, 3 * d, d, dtype, s));
          // norm2 -> cross-attention query projection
          tb.q2.cin = d; tb.q2.cout = d; tb.q2.taps = 1;
          SDMI_TRY(mem.alloc(&tb.q2.w, (size_t)d * d * esz_w));
          SDMI_TRY(alloc_f(&b, (size_t)d));
          SDMI_TRY(fold_into(tb.q2.w, b, 0, q + ".attn2.to_q.weight", "", tb.ln_g[1], tb.ln_b[1], d, d));
          tb.q2.bias = b;
          SDMI_TRY(alloc_f(&tb.q2_c, (size_t)d));
          SDMI_TRY(packed_rowsum(tb.q2.w, tb.q2_c, d, d, dtype, s));
          // norm3 -> GEGLU projection: fold in checkpoint row order, then the (value, gate) interleave
          SDMI_TRY(fold_ln_weight(store.get(q + ".ff.net.0.proj.weight"), tb.ln_g[2], tb.ln_b[2],
                                  store.get(q + ".ff.net.0.proj.bias"), fold_w, fold_b, 8 * d, d, s));
          SDMI_TRY(pack_geglu_weight(fold_w, fold_b, tb.ff1.w, tb.ff1_bias, 4 * d, d, dtype, s));
          SDMI_TRY(alloc_f(&tb.ff1_c, (size_t)8 * d));
          SDMI_TRY(packed_rowsum(tb.ff1.w, tb.ff1_c, 8 * d, d, dtype, s));
          packed_keys.push_back(q + ".attn2.to_q.weight");
        } else {
          SDMI_TRY(pack_stacked(&tb.qkv, qkv, d, d, s));
          SDMI_TRY(P(&tb.q2, q + ".attn2.to_q", d, d, 1, false));
          SDMI_TRY(pack_geglu_weight(store.get(q + ".ff.net.0.proj.weight"), store.get(q + ".ff.net.0.proj.bias"),
                                     tb.ff1.w, tb.ff1_bias, 4 * d, d, dtype, s));
        }
        packed_keys.insert(packed_keys.end(), qkv.begin(), qkv.end());
        packed_keys.insert(packed_keys.end(), kv.begin(), kv.end());
        tb.ff1.bias = tb.ff1_bias;
        packed_keys.push_back(q + ".ff.net.0.proj.weight");
        packed_keys.push_back(q + ".ff.net.0.proj.bias");
      }
    }
    for (auto& c : downs) SDMI_TRY(P(&c.conv, c.prefix, c.ch, c.ch, 9));
    for (auto& c : ups) SDMI_TRY(P(&c.conv, c.prefix, c.ch, c.ch, 9));
    SDMI_CHECK_HIP(hipStreamSynchronize(s));
    for (const auto& k : packed_keys) store.drop(k);
    if (fold_w) mem.release(fold_w);
    if (fold_b) mem.release(fold_b);
    finalized = true;
    return SDMI_OK;
  }

  // ---- per-prompt context cache ------------------------------------------------------------------
  int set_context(const float* ctx, int batch, int ntok, hipStream_t s) {
    if (!finalized) { sdmi_set_error("set_context before finalize"); return SDMI_ESTATE; }
    SDMI_REQUIRE(ctx && batch > 0 && ntok > 0, "set_context: bad arguments");
    const size_t esz = sdmi_dtype_size(dtype);
    const size_t rows = (size_t)batch * ntok;
    if (batch != ctx_batch || ntok != ctx_ntok) {
      drop_graph();
      if (ctx_conv) mem.release(ctx_conv);
      SDMI_TRY(mem.alloc(&ctx_conv, rows * cfg.context_dim * esz));
      for (auto& st : sts)
        for (auto& tb : st.blocks) {
          if (tb.kv_cache) mem.release(tb.kv_cache);
          SDMI_TRY(mem.alloc(&tb.kv_cache, rows * 2 * st.inner * esz));
          for (void** q : {&tb.xG, &tb.xU, (void**)&tb.xg0, (void**)&tb.xGc})
            if (*q) { mem.release(*q); *q = nullptr; }
          if (xfold && st.inner >= xfold_min_d && ntok <= XSEG && (st.heads * XSEG) % 160 == 0) {
            const size_t seg_rows = (size_t)batch * st.heads * XSEG;
            void* q;
            SDMI_TRY(mem.alloc(&tb.xG, seg_rows * st.inner * esz));
            SDMI_TRY(mem.alloc(&tb.xU, seg_rows * st.inner * esz));
            SDMI_TRY(mem.alloc(&q, seg_rows * 4)); tb.xg0 = (float*)q;
            SDMI_TRY(mem.alloc(&q, seg_rows * 4)); tb.xGc = (float*)q;
          }
        }
      ctx_batch = batch; ctx_ntok = ntok;
    }
    const void* src = ctx;
    if (dtype != SDMI_F32) {
      SDMI_TRY(convert_f32(ctx, ctx_conv, (int64_t)(rows * cfg.context_dim), dtype, s));
      src = ctx_conv;
    }
    for (auto& st : sts)
      for (auto& tb : st.blocks) {
        ConvGemmArgs g;
        g.a0 = src; g.c0 = cfg.context_dim; g.w = tb.kv2.w; g.out = tb.kv_cache; g.ldo = 2 * st.inner;
        g.B = 1; g.Hin = 1; g.Win = (int)rows; g.Ho = 1; g.Wo = (int)rows; g.N = 2 * st.inner; g.dtype = dtype;
        SDMI_TRY(conv_gemm(g, s));
        if (tb.xG) {
          // scale = dim_head ** -0.5 evaluated in double and rounded to fp32 (as attn() passes it), times log2(e)
          const float scale = (float)pow((double)st.dh, -0.5);
          const float c0 = (float)((double)scale * 1.4426950408889634);
          SDMI_TRY(xattn_fold(tb.kv_cache, tb.q2.w, tb.q2.bias, tb.out2.w, tb.xG, tb.xg0, tb.xU, batch, ntok, st.heads,
                              st.dh, st.inner, c0, dtype, s));
          SDMI_TRY(packed_rowsum(tb.xG, tb.xGc, batch * st.heads * XSEG, st.inner, dtype, s));
        }
      }
    return SDMI_OK;
  }

  // ---- forward ---------------------------------------------------------------------------------
  struct Fwd {
    const float* x; const float* concat[2]; int cc[2]; int n_concat;
    const int64_t* t; float* out; int B, H, W; hipStream_t s;
  };

  size_t act_bytes(int B, int H, int W, int C) const { return (size_t)B * H * W * C * sdmi_dtype_size(dtype); }

  int gemm(const void* a0, int c0, const void* a1, int c1, const PackedConv& pc, void* out, int ldo, int B, int Hin,
           int Win, int Ho, int Wo, int stride, int up, const float* rowadd, int ldra, const void* resid, int ldr,
           int geglu, const float* bias_override, hipStream_t s) {
    ConvGemmArgs g;
    g.a0 = a0; g.c0 = c0; g.a1 = a1; g.c1 = c1; g.w = pc.w; g.out = out; g.ldo = ldo;
    g.bias = bias_override ? bias_override : pc.bias; g.rowadd = rowadd; g.ldra = ldra; g.resid = resid; g.ldr = ldr;
    g.B = B; g.Hin = Hin; g.Win = Win; g.Ho = Ho; g.Wo = Wo; g.taps = pc.taps; g.stride = stride; g.up = up;
    g.N = pc.cout; g.geglu = geglu; g.dtype = dtype;
    g.dup_rows = next_dup_rows; next_dup_rows = 0;
    g.ln_stats = next_ln_stats; g.ln_colsum = next_ln_colsum; g.ln_nparts = next_ln_nparts;
    next_ln_stats = next_ln_colsum = nullptr; next_ln_nparts = 0;
    g.row_stats_out = next_row_stats_out; next_row_stats_out = nullptr;
    g.w_rows = next_w_rows; g.vec_per_sample = next_vec_per_sample; g.softmax_seg = next_softmax;
    next_w_rows = next_vec_per_sample = next_softmax = 0;
    g.e0 = next_e0; g.ec0 = next_e0 ? next_ec0 : 0; g.e1 = next_e1; g.ec1 = next_e1 ? next_ec1 : 0;
    next_e0 = next_e1 = nullptr; next_ec0 = next_ec1 = 0;
    const int M = B * Ho * Wo, K = pc.taps * (c0 + (a1 ? c1 : 0)) + g.ec0 + g.ec1;
    g.splits = g.w_rows > 0 ? 1 : conv_gemm_pick_splits(M, pc.cout, K, dtype, geglu != 0);
    if (g.splits > 1) g.partial = (float*)ws.get("splitk", (size_t)g.splits * M * pc.cout * 4);
    last_rs_parts = g.row_stats_out ? conv_gemm_row_stat_parts(M, pc.cout, K, dtype, geglu != 0, g.splits, g.w_rows) : 0;
    if (ws.dry) return SDMI_OK;
    return conv_gemm(g, s);
  }
  // token GEMM (taps = 1): rows x K -> rows x N
  int linear(const void* a, int K, const PackedConv& pc, void* out, int ldo, int rows, const void* resid, int ldr,
             int geglu, hipStream_t s) {
    return gemm(a, K, nullptr, 0, pc, out, ldo, 1, 1, rows, 1, rows, 1, 0, nullptr, 0, resid, ldr, geglu, nullptr, s);
  }

  int gn(const Tensor& x0, const Tensor* x1, int B, const float* g, const float* b, float eps, int silu, void* out,
         hipStream_t s) {
    GroupNormArgs a;
    a.x0 = x0.p; a.c0 = x0.C; a.x1 = x1 ? x1->p : nullptr; a.c1 = x1 ? x1->C : 0;
    a.B = B; a.HW = x0.H * x0.W; a.groups = 32; a.eps = eps; a.gamma = g; a.beta = b; a.silu = silu; a.out = out;
    a.scratch = (float*)ws.get("gn_scratch", groupnorm_scratch_floats(B, a.HW, 32) * 4);
    a.dtype = dtype;
    if (ws.dry) return SDMI_OK;
    return groupnorm(a, s);
  }

  // dup: compute samples 0..B/2-1 only and store the block's output for both halves
  int run_res(const ResW& r, const Tensor& x0, const Tensor* x1, Tensor* y, const std::string& out_role,
              const float* emb_all, const Fwd& f, bool dup = false) {
    const int H = x0.H, W = x0.W;
    const int B = dup ? f.B / 2 : f.B;
    const int dup_rows = dup ? B * H * W : 0;
    const int cin = x0.C + (x1 ? x1->C : 0);
    SDMI_REQUIRE(cin == r.cin, "%s: got %d input channels, expected %d", r.prefix.c_str(), cin, r.cin);
    SDMI_REQUIRE(r.has_skip || !x1, "%s: identity skip over a concatenated input is not supported", r.prefix.c_str());
    void* nrm = ws.get("nrm", act_bytes(B, H, W, cin > r.cout ? cin : r.cout));
    void* t1 = ws.get("t1", act_bytes(B, H, W, r.cout));
    y->p = ws.get(out_role, act_bytes(f.B, H, W, r.cout)); y->H = H; y->W = W; y->C = r.cout;
    SDMI_TRY(gn(x0, x1, B, r.gn1_g, r.gn1_b, 1e-5f, 1, nrm, f.s));
    SDMI_TRY(gemm(nrm, cin, nullptr, 0, r.conv1, t1, r.cout, B, H, W, H, W, 1, 0, emb_all + r.emb_off,
                  emb_rows == 1 && B > 1 ? -1 : emb_total,
                  nullptr, 0, 0, nullptr, f.s));
    Tensor t1t{t1, H, W, r.cout};
    SDMI_TRY(gn(t1t, nullptr, B, r.gn2_g, r.gn2_b, 1e-5f, 1, nrm, f.s));
    if (r.has_skip && r.conv2f.w && conv_gemm_extra_ok(B * H * W, r.cout, 9 * r.cout + cin, dtype)) {
      // skip_connection(x) + conv(h) as one GEMM: x (both halves of a concatenated input) behind the nine taps
      next_e0 = x0.p; next_ec0 = x0.C;
      if (x1) { next_e1 = x1->p; next_ec1 = x1->C; }
      next_dup_rows = dup_rows;
      SDMI_TRY(gemm(nrm, r.cout, nullptr, 0, r.conv2f, y->p, r.cout, B, H, W, H, W, 1, 0, nullptr, 0, nullptr, 0, 0,
                    nullptr, f.s));
    } else if (r.has_skip) {
      SDMI_TRY(gemm(x0.p, x0.C, x1 ? x1->p : nullptr, x1 ? x1->C : 0, r.skip, y->p, r.cout, B, H, W, H, W, 1, 0,
                    nullptr, 0, nullptr, 0, 0, nullptr, f.s));
      next_dup_rows = dup_rows;
      SDMI_TRY(gemm(nrm, r.cout, nullptr, 0, r.conv2, y->p, r.cout, B, H, W, H, W, 1, 0, nullptr, 0, y->p, r.cout, 0,
                    nullptr, f.s));
    } else {
      next_dup_rows = dup_rows;
      SDMI_TRY(gemm(nrm, r.cout, nullptr, 0, r.conv2, y->p, r.cout, B, H, W, H, W, 1, 0, nullptr, 0, x0.p, x0.C, 0,
                    nullptr, f.s));
    }
    return SDMI_OK;
  }

  int attn(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int B,
           int heads, int nq, int nk, int d, hipStream_t s) {
    if (ws.dry) return SDMI_OK;
    AttentionArgs a;
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out = out; a.ldo = ldo;
    a.B = B; a.heads = heads; a.Nq = nq; a.Nk = nk; a.d = d; a.scale = 1.0f / sqrtf((float)d); a.dtype = dtype;
    // the reference multiplies by dim_head ** -0.5 evaluated in double and rounded to fp32
    a.scale = (float)pow((double)d, -0.5);
    return attention(a, s);
  }

  int ln(const void* x, void* out, const float* g, const float* b, int rows, int d, hipStream_t s) {
    if (ws.dry) return SDMI_OK;
    return layernorm(x, out, g, b, rows, d, 1e-5f, dtype, s);
  }

  // dup: the two halves of the batch enter identical; the first block's self-attention, its output
  // projection and the cross-attention query projection run on one half and are stored twice
  int run_st(const STW& st, const Tensor& x, Tensor* y, const std::string& out_role, const Fwd& f, bool dup = false) {
    const int B = f.B, H = x.H, W = x.W, N = H * W, rows = B * N, d = st.inner;
    const int Bh = dup ? B / 2 : B, rows_h = Bh * N;
    const size_t esz = sdmi_dtype_size(dtype);
    SDMI_REQUIRE(x.C == st.ch, "%s: got %d channels, expected %d", st.prefix.c_str(), x.C, st.ch);
    void* nrm = ws.get("nrm", act_bytes(B, H, W, st.ch));
    void* xt = ws.get("xt", (size_t)rows * d * esz);
    void* lnb = fold_ln ? nullptr : ws.get("ln", (size_t)rows * d * esz);
    // row statistics of xt: (rstd, rstd * mean) from the statistics kernel, or up to 16 (sum, sum of squares) pairs
    // per row left by the GEMM that wrote xt
    float* lnst = fold_ln ? (float*)ws.get("lnstat", (size_t)rows * (producer_stats ? 16 : 1) * 2 * 4) : nullptr;
    int xt_parts = 0, xt_stat_rows = 0;       // what lnst holds for the current xt (0 parts: nothing)
    // a linear() that (re)writes xt, leaving the row sums behind when the next reader is a folded LayerNorm
    auto to_xt = [&](const void* a, int K, const PackedConv& pc, int nrows, const void* resid, int dup_rows,
                     bool read_by_ln = true) -> int {
      if (producer_stats && read_by_ln) next_row_stats_out = lnst;
      next_dup_rows = dup_rows;
      SDMI_TRY(linear(a, K, pc, xt, d, nrows, resid, d, 0, f.s));
      xt_parts = producer_stats && read_by_ln ? last_rs_parts : 0;
      xt_stat_rows = nrows + dup_rows;
      SDMI_REQUIRE(xt_parts <= 16, "row statistics: %d parts per row", xt_parts);
      return SDMI_OK;
    };
    const int xseg_n = st.heads * XSEG;                   // width of the folded cross-attention's probability rows
    // the folded cross-attention needs the folded context and tile shapes that divide a sample's token count
    // and pays from d = 640 up: the two GEMMs cost 2 (80 + 80) heads d flops per token against 4 d d + 4 ntok d, and
    // at d = 320 the three-launch path is as fast (measured; SDMI_XATTN_FOLD_MIN_D overrides)
    const bool xf = xfold && d >= xfold_min_d && !st.blocks.empty() && st.blocks[0].xG != nullptr &&
                    conv_gemm_per_sample_ok(rows, xseg_n, d, dtype, N, true) &&
                    conv_gemm_per_sample_ok(rows, d, xseg_n, dtype, N, false);
    void* qkv = ws.get("qkv", (size_t)rows * std::max(3 * d, xseg_n) * esz);
    void* att = ws.get("att", (size_t)rows * d * esz);
    void* ffm = ws.get("ffm", (size_t)rows * 4 * d * esz);
    y->p = ws.get(out_role, act_bytes(B, H, W, st.ch)); y->H = H; y->W = W; y->C = st.ch;
    SDMI_TRY(gn(x, nullptr, Bh, st.gn_g, st.gn_b, 1e-6f, 0, nrm, f.s));
    SDMI_TRY(to_xt(nrm, st.ch, st.proj_in, rows_h, nullptr, 0));
    // Linear(LayerNorm(xt)): either the normalised copy goes through HBM (fp32 mode), or only the row statistics
    // do and the GEMM reads xt itself (weights carry gamma, the epilogue applies mean, rstd and W beta)
    auto ln_linear = [&](const TBlockW& tb, int which, const float* colsum, const PackedConv& pc, void* out, int ldo,
                         int nrows, int geglu, int dup_rows) -> int {
      const void* a = xt;
      if (fold_ln) {
        if (xt_parts > 0 && xt_stat_rows >= nrows) {
          next_ln_nparts = xt_parts;
        } else {
          if (!ws.dry) SDMI_TRY(layernorm_stats(xt, lnst, nrows, d, 1e-5f, dtype, f.s));
          xt_parts = 0;
        }
        next_ln_stats = lnst; next_ln_colsum = colsum;
      } else {
        SDMI_TRY(ln(xt, lnb, tb.ln_g[which], tb.ln_b[which], nrows, d, f.s));
        a = lnb;
      }
      next_dup_rows = dup_rows;
      return linear(a, d, pc, out, ldo, nrows, nullptr, 0, geglu, f.s);
    };
    bool first = true;
    for (const auto& tb : st.blocks) {
      const bool half = dup && first;             // later blocks see diverged halves
      const int bq = half ? Bh : B, rq = half ? rows_h : rows;
      first = false;
      // self-attention
      SDMI_TRY(ln_linear(tb, 0, tb.qkv_c, tb.qkv, qkv, 3 * d, rq, 0, 0));
      SDMI_TRY(attn(qkv, 3 * d, (char*)qkv + (size_t)d * esz, 3 * d, (char*)qkv + (size_t)2 * d * esz, 3 * d, att, d,
                    bq, st.heads, N, N, st.dh, f.s));
      SDMI_TRY(to_xt(att, d, tb.out1, rq, xt, half ? rows_h : 0));
      if (xf) {
        // cross-attention as two GEMMs with this prompt's folded weights: probabilities = softmax per head of
        // LayerNorm(xt) G^T (+ g0) in the first one's epilogue, xt += P U^T + bias in the second
        PackedConv pg; pg.w = tb.xG; pg.cin = d; pg.cout = xseg_n; pg.taps = 1; pg.bias = tb.xg0;
        PackedConv pu; pu.w = tb.xU; pu.cin = xseg_n; pu.cout = d; pu.taps = 1; pu.bias = tb.out2.bias;
        next_w_rows = N; next_vec_per_sample = 1; next_softmax = 1;
        SDMI_TRY(ln_linear(tb, 1, tb.xGc, pg, qkv, xseg_n, rows, 0, 0));
        next_w_rows = N;
        SDMI_TRY(to_xt(qkv, xseg_n, pu, rows, xt, 0));
      } else {
      // cross-attention against the cached context projections
      SDMI_TRY(ln_linear(tb, 1, tb.q2_c, tb.q2, qkv, d, rq, 0, half ? rows_h : 0));
      SDMI_TRY(attn(qkv, d, tb.kv_cache, 2 * d, (char*)tb.kv_cache + (size_t)d * esz, 2 * d, att, d, B, st.heads, N,
                    ctx_ntok, st.dh, f.s));
      SDMI_TRY(to_xt(att, d, tb.out2, rows, xt, 0));
      }
      // GEGLU feed-forward
      SDMI_TRY(ln_linear(tb, 2, tb.ff1_c, tb.ff1, ffm, 4 * d, rows, 1, 0));
      SDMI_TRY(to_xt(ffm, 4 * d, tb.ff2, rows, xt, 0, &tb != &st.blocks.back()));
    }
    SDMI_TRY(linear(xt, d, st.proj_out, y->p, st.ch, rows, x.p, st.ch, 0, f.s));
    return SDMI_OK;
  }

  // dup_block: this block is [ResBlock, SpatialTransformer] directly behind the first convolution and
  // the batch halves are identical on entry
  int run_layers(const std::vector<Layer>& layers, Tensor h, const Tensor* skip, Tensor* out,
                 const std::string& block_role, const float* emb_all, const Fwd& f, bool dup_block = false) {
    for (size_t li = 0; li < layers.size(); ++li) {
      const bool last = li + 1 == layers.size();
      const std::string role = last ? block_role : (li % 2 == 0 ? std::string("tmp_a") : std::string("tmp_b"));
      const Layer& l = layers[li];
      Tensor y;
      switch (l.kind) {
        case L_CONV_IN: {
          y.p = ws.get(role, act_bytes(f.B, f.H, f.W, cfg.model_channels));
          y.H = f.H; y.W = f.W; y.C = cfg.model_channels;
          if (!ws.dry) {
            InConvArgs a;
            a.src[0] = f.x; a.c[0] = cfg.in_channels;
            for (int i = 0; i < f.n_concat; ++i) { a.src[i + 1] = f.concat[i]; a.c[i + 1] = f.cc[i]; a.c[0] -= f.cc[i]; }
            a.w = inconv_w; a.bias = inconv_b; a.out = y.p; a.B = f.B; a.H = f.H; a.W = f.W;
            a.Cout = cfg.model_channels; a.dtype = dtype;
            SDMI_TRY(in_conv(a, f.s));
          }
          break;
        }
        case L_RES:
          SDMI_TRY(run_res(res[l.idx], h, li == 0 ? skip : nullptr, &y, role, emb_all, f, dup_block && li == 0));
          break;
        case L_ST:
          SDMI_TRY(run_st(sts[l.idx], h, &y, role, f, dup_block && li == 1));
          break;
        case L_DOWN: {
          const SimpleConv& c = downs[l.idx];
          y.H = (h.H + 2 - 3) / 2 + 1; y.W = (h.W + 2 - 3) / 2 + 1; y.C = c.ch;
          y.p = ws.get(role, act_bytes(f.B, y.H, y.W, c.ch));
          SDMI_TRY(gemm(h.p, h.C, nullptr, 0, c.conv, y.p, c.ch, f.B, h.H, h.W, y.H, y.W, 2, 0, nullptr, 0, nullptr, 0,
                        0, nullptr, f.s));
          break;
        }
        case L_UP: {
          const SimpleConv& c = ups[l.idx];
          y.H = 2 * h.H; y.W = 2 * h.W; y.C = c.ch;
          y.p = ws.get(role, act_bytes(f.B, y.H, y.W, c.ch));
          SDMI_TRY(gemm(h.p, h.C, nullptr, 0, c.conv, y.p, c.ch, f.B, h.H, h.W, y.H, y.W, 1, 1, nullptr, 0, nullptr, 0,
                        0, nullptr, f.s));
          break;
        }
      }
      h = y;
    }
    *out = h;
    return SDMI_OK;
  }

  int run(const Fwd& f) {
    const int B = f.B;
    float* temb0 = (float*)ws.get("temb0", (size_t)B * cfg.model_channels * 4);
    float* temb1 = (float*)ws.get("temb1", (size_t)B * temb * 4);
    float* emb = (float*)ws.get("emb", (size_t)B * temb * 4);
    float* emb_all = (float*)ws.get("emb_all", (size_t)B * emb_total * 4);
    emb_rows = uniform_t ? 1 : B;          // rows of the embedding tables actually computed
    if (!ws.dry) {
      SDMI_TRY(timestep_embedding(f.t, temb0, emb_rows, cfg.model_channels, 10000.f, f.s));
      SDMI_TRY(small_linear(temb0, te0_w, te0_b, temb1, emb_rows, cfg.model_channels, temb, 0, 1, f.s));
      SDMI_TRY(small_linear(temb1, te2_w, te2_b, emb, emb_rows, temb, temb, 0, 0, f.s));
      SDMI_TRY(small_linear(emb, emb_w_all, emb_b_all, emb_all, emb_rows, temb, emb_total, 1, 0, f.s));
    }
    std::vector<Tensor> hs;
    Tensor h;
    for (size_t i = 0; i < input_blocks.size(); ++i) {
      Tensor y;
      const std::vector<Layer>& L = input_blocks[i];
      const bool dup_block = cfg_dup && i == 1 && f.B % 2 == 0 && f.n_concat == 0 && L.size() == 2 &&
                             L[0].kind == L_RES && L[1].kind == L_ST && !res[L[0].idx].has_skip;
      SDMI_TRY(run_layers(L, h, nullptr, &y, "in" + std::to_string(i), emb_all, f, dup_block));
      hs.push_back(y);
      h = y;
    }
    {
      Tensor y;
      SDMI_TRY(run_layers(middle, h, nullptr, &y, "mid", emb_all, f));
      h = y;
    }
    for (size_t i = 0; i < output_blocks.size(); ++i) {
      Tensor skip = hs.back(); hs.pop_back();
      SDMI_REQUIRE(skip.H == h.H && skip.W == h.W, "output block %zu: skip is %dx%d but h is %dx%d (latent size "
                   "must be divisible by %d)", i, skip.H, skip.W, h.H, h.W, 1 << (cfg.n_channel_mult - 1));
      Tensor y;
      SDMI_TRY(run_layers(output_blocks[i], h, &skip, &y, "out" + std::to_string(i), emb_all, f));
      h = y;
    }
    void* nrm = ws.get("nrm", act_bytes(B, h.H, h.W, h.C));
    SDMI_TRY(gn(h, nullptr, B, out_gn_g, out_gn_b, 1e-5f, 1, nrm, f.s));
    if (!ws.dry)
      SDMI_TRY(out_conv(nrm, outconv_w, outconv_b, f.out, B, h.H, h.W, h.C, cfg.out_channels, dtype, f.s));
    return SDMI_OK;
  }

  // One ResBlock or SpatialTransformer of the layer list on its own, through the same code as forward()
  int run_block(int kind, int index, const void* x, const void* x_skip, int c_skip, const float* emb, void* out,
                int B, int H, int W, hipStream_t s) {
    if (!finalized) { sdmi_set_error("run_block before finalize"); return SDMI_ESTATE; }
    SDMI_REQUIRE(x && out && B > 0 && H > 0 && W > 0, "run_block: bad arguments");
    SDMI_REQUIRE(kind == 0 || kind == 1, "run_block: kind must be 0 (ResBlock) or 1 (SpatialTransformer)");
    SDMI_REQUIRE(index >= 0 && index < (int)(kind == 0 ? res.size() : sts.size()), "run_block: no block %d", index);
    SDMI_REQUIRE(kind == 0 ? emb != nullptr : ctx_batch == B,
                 kind == 0 ? "run_block: a ResBlock needs emb" : "run_block: context was set for another batch");
    SDMI_REQUIRE(kind == 0 || !x_skip, "run_block: only a ResBlock takes a second input");
    Fwd f;
    memset(&f, 0, sizeof(f));
    f.B = B; f.H = H; f.W = W; f.s = s;
    auto body = [&]() -> int {
      Tensor y;
      if (kind == 0) {
        const ResW& r = res[index];
        float* emb_all = (float*)ws.get("emb_all", (size_t)B * emb_total * 4);
        emb_rows = B;
        if (!ws.dry) SDMI_TRY(small_linear(emb, emb_w_all, emb_b_all, emb_all, B, temb, emb_total, 1, 0, s));
        Tensor x0{const_cast<void*>(x), H, W, r.cin - (x_skip ? c_skip : 0)};
        Tensor x1{const_cast<void*>(x_skip), H, W, c_skip};
        SDMI_TRY(run_res(r, x0, x_skip ? &x1 : nullptr, &y, "blk_out", emb_all, f));
      } else {
        Tensor x0{const_cast<void*>(x), H, W, sts[index].ch};
        SDMI_TRY(run_st(sts[index], x0, &y, "blk_out", f));
      }
      if (!ws.dry)
        SDMI_CHECK_HIP(hipMemcpyAsync(out, y.p, act_bytes(B, H, W, y.C), hipMemcpyDeviceToDevice, s));
      return SDMI_OK;
    };
    ws.dry = true; ws.need.clear();
    SDMI_TRY(body());
    ws.dry = false;
    if (!ws.fits()) {
      drop_graph();
      SDMI_CHECK_HIP(hipStreamSynchronize(s));
      SDMI_TRY(ws.commit());
    }
    return body();
  }

  int forward(const Fwd& f) {
    if (!finalized) { sdmi_set_error("forward before finalize"); return SDMI_ESTATE; }
    SDMI_REQUIRE(f.x && f.t && f.out && f.B > 0 && f.H > 0 && f.W > 0, "forward: bad arguments");
    SDMI_REQUIRE(ctx_batch == f.B, "forward: context was set for batch %d but forward got batch %d", ctx_batch, f.B);
    int csum = 0;
    for (int i = 0; i < f.n_concat; ++i) csum += f.cc[i];
    SDMI_REQUIRE(f.n_concat >= 0 && f.n_concat <= 2 && csum < cfg.in_channels, "forward: bad concat channels");
    // size the workspace with a dry run; (re)allocate only when something grew
    ws.dry = true; ws.need.clear();
    SDMI_TRY(run(f));
    ws.dry = false;
    if (!ws.fits()) {
      drop_graph();
      SDMI_CHECK_HIP(hipStreamSynchronize(f.s));
      SDMI_TRY(ws.commit());
    }
    if (!graph_mode) return run(f);
    GraphKey k;
    memset(&k, 0, sizeof(k));
    k.x = f.x; k.c0 = f.n_concat > 0 ? f.concat[0] : nullptr; k.c1 = f.n_concat > 1 ? f.concat[1] : nullptr;
    k.t = f.t; k.out = f.out; k.n_concat = f.n_concat; k.cc0 = f.cc[0]; k.cc1 = f.cc[1];
    k.B = f.B; k.H = f.H; k.W = f.W; k.uniform_t = uniform_t; k.cfg_dup = cfg_dup; k.stream = nullptr;
    if (!(ghave && k == gkey)) {
      // Recording and instantiating a graph costs more than an eager forward, so a key earns its graph by
      // coming back: a caller that hands over fresh buffers on every call (new x / t / out addresses) runs
      // eagerly throughout, a sampling loop with persistent buffers replays from its third step on.
      if (!(gseen_have && k == gseen)) {
        gseen = k; gseen_have = true;
        return run(f);
      }
      drop_graph();
      // the caller's stream may be the legacy default stream, which cannot be captured: record the
      // launch sequence on a private stream (nothing executes there) and replay it on f.s
      if (!cap_stream) SDMI_CHECK_HIP(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
      Fwd fc = f;
      fc.s = cap_stream;
      SDMI_CHECK_HIP(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
      const int rc = run(fc);
      hipGraph_t g = nullptr;
      const hipError_t e = hipStreamEndCapture(cap_stream, &g);
      if (rc != SDMI_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
      if (e != hipSuccess) { sdmi_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return SDMI_EHIP; }
      graph = g;
      SDMI_CHECK_HIP(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
      {
        size_t n = 0;
        SDMI_CHECK_HIP(hipGraphGetNodes(graph, nullptr, &n));
        graph_nodes = (int)n;
      }
      gkey = k; ghave = true;
    }
    SDMI_CHECK_HIP(hipGraphLaunch(gexec, f.s));
    return SDMI_OK;
  }
};

extern "C" {

int sdmi_unet_create(const sdmi_unet_cfg* cfg, sdmi_unet** out) {
  SDMI_REQUIRE(cfg && out, "unet_create: null argument");
  SDMI_REQUIRE(cfg->n_channel_mult > 0 && cfg->n_channel_mult <= 8 && cfg->n_attention_resolutions >= 0 &&
               cfg->n_attention_resolutions <= 8, "unet_create: bad list lengths");
  SDMI_REQUIRE(cfg->compute_dtype >= 0 && cfg->compute_dtype <= 2, "unet_create: bad compute_dtype");
  SDMI_REQUIRE(cfg->context_dim > 0 && cfg->transformer_depth > 0, "unet_create: spatial transformer needs context_dim");
  SDMI_REQUIRE(cfg->num_heads != -1 || cfg->num_head_channels != -1,
               "Either num_heads or num_head_channels has to be set.");
  SDMI_REQUIRE(cfg->model_channels % 64 == 0, "unet_create: model_channels must be a multiple of 64");
  SDMI_REQUIRE(cfg->in_channels > 0 && cfg->in_channels <= 64 && (cfg->out_channels == 3 || cfg->out_channels == 4),
               "unet_create: in_channels <= 64 and out_channels in {3,4}");
  std::unique_ptr<sdmi_unet> u(new sdmi_unet());
  u->cfg = *cfg;
  u->dtype = cfg->compute_dtype;
  u->store.mem = &u->mem;
  u->ws.mem = &u->mem;
  SDMI_TRY(u->build_structure());
  *out = u.release();
  return SDMI_OK;
}

int sdmi_unet_num_weights(const sdmi_unet* u) { return u ? (int)u->specs.size() : 0; }

int sdmi_unet_weight_info(const sdmi_unet* u, int index, const char** key, int64_t* shape4, int* ndim) {
  SDMI_REQUIRE(u && index >= 0 && index < (int)u->specs.size(), "weight_info: bad index");
  const WeightSpec& s = u->specs[index];
  if (key) *key = s.key.c_str();
  if (ndim) *ndim = (int)s.shape.size();
  if (shape4)
    for (size_t i = 0; i < 4; ++i) shape4[i] = i < s.shape.size() ? s.shape[i] : 1;
  return SDMI_OK;
}

int sdmi_unet_set_weight(sdmi_unet* u, const char* key, const void* dev_ptr, const int64_t* shape, int ndim,
                         int dtype, void* stream) {
  SDMI_REQUIRE(u && key && dev_ptr && shape, "set_weight: null argument");
  if (u->finalized) { sdmi_set_error("set_weight after finalize"); return SDMI_ESTATE; }
  return u->store.set(u->specs, key, dev_ptr, shape, ndim, dtype, (hipStream_t)stream);
}

int sdmi_unet_finalize(sdmi_unet* u, void* stream) {
  SDMI_REQUIRE(u, "finalize: null handle");
  return u->finalize((hipStream_t)stream);
}

int sdmi_unet_set_context(sdmi_unet* u, const void* ctx, int batch, int ntok, void* stream) {
  SDMI_REQUIRE(u, "set_context: null handle");
  return u->set_context((const float*)ctx, batch, ntok, (hipStream_t)stream);
}

int sdmi_unet_forward(sdmi_unet* u, const void* x, const void* const* concat, const int* concat_channels,
                      int n_concat, const int64_t* t, void* out, int batch, int h, int w, void* stream) {
  SDMI_REQUIRE(u, "forward: null handle");
  SDMI_REQUIRE(n_concat >= 0 && n_concat <= 2 && (n_concat == 0 || (concat && concat_channels)),
               "forward: bad concat arguments");
  sdmi_unet::Fwd f;
  f.x = (const float*)x; f.concat[0] = f.concat[1] = nullptr; f.cc[0] = f.cc[1] = 0; f.n_concat = n_concat;
  for (int i = 0; i < n_concat; ++i) { f.concat[i] = (const float*)concat[i]; f.cc[i] = concat_channels[i]; }
  f.t = t; f.out = (float*)out; f.B = batch; f.H = h; f.W = w; f.s = (hipStream_t)stream;
  return u->forward(f);
}

int sdmi_unet_run_block(sdmi_unet* u, int kind, int index, const void* x, const void* x_skip, int c_skip,
                        const float* emb, void* out, int batch, int h, int w, void* stream) {
  SDMI_REQUIRE(u, "run_block: null handle");
  return u->run_block(kind, index, x, x_skip, c_skip, emb, out, batch, h, w, (hipStream_t)stream);
}

int64_t sdmi_unet_device_bytes(const sdmi_unet* u) { return u ? (int64_t)u->mem.bytes : 0; }

int sdmi_unet_set_uniform_timesteps(sdmi_unet* u, int enable) {
  SDMI_REQUIRE(u, "set_uniform_timesteps: null handle");
  u->uniform_t = enable ? 1 : 0;
  return SDMI_OK;
}

int sdmi_unet_set_cfg_duplicate(sdmi_unet* u, int enable) {
  SDMI_REQUIRE(u, "set_cfg_duplicate: null handle");
  u->cfg_dup = enable ? 1 : 0;
  return SDMI_OK;
}

int sdmi_unet_graph_launches(const sdmi_unet* u) { return u ? u->graph_nodes : 0; }

int sdmi_unet_set_graph_mode(sdmi_unet* u, int enable) {
  SDMI_REQUIRE(u, "set_graph_mode: null handle");
  u->graph_mode = enable ? 1 : 0;
  if (!enable) u->drop_graph();
  return SDMI_OK;
}

void sdmi_unet_destroy(sdmi_unet* u) { delete u; }

}  // extern "C"
