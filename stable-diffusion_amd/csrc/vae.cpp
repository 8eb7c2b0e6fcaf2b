// VAE executors.
// Decoder (AutoencoderKL.decode -> Decoder.forward, stable_diffusion.py:1416-1419, :496-527):
// post_quant_conv -> conv_in -> mid (ResnetBlock, single-head attention, ResnetBlock) -> up levels
// (ResnetBlocks + nearest-x2 + conv) -> GroupNorm + swish + conv_out.
// Encoder (AutoencoderKL.encode -> Encoder.forward, :1410-1414, :393-420): conv_in -> down levels
// (ResnetBlocks + DownsampleEncoder :88-99 = zero pad right/bottom, conv 3x3 stride 2) -> mid ->
// GroupNorm + swish + conv_out -> quant_conv; the result is the (mean | logvar) moments tensor.
//
// Same kernels and NHWC layout as the UNet.  The mid-block attention (one head of width C over all
// H*W tokens, :161-183) does not fit a register-resident flash tile, so it runs as two GEMMs
// around a row softmax: S = q k^T (fp32 out), P = softmax(S * C^-0.5), O = P v^T.  v^T comes out of
// a GEMM with swapped operands (weights as the row operand), and v's bias is added after the P
// product, which is exact because every row of P sums to one.
#include <math.h>

#include <memory>

#include "../../include/sdmi.h"
#include "executor.h"

namespace {

struct VResW {
  std::string prefix;
  int cin = 0, cout = 0;
  const float *gn1_g = nullptr, *gn1_b = nullptr, *gn2_g = nullptr, *gn2_b = nullptr;
  PackedConv conv1, conv2, skip;
  bool has_skip = false;
};
struct VAttnW {
  std::string prefix;
  int ch = 0;
  const float *gn_g = nullptr, *gn_b = nullptr;
  PackedConv q, k, v, proj;
};
struct VUpW { std::string prefix; int ch = 0; PackedConv conv; };
enum VKind { V_RES, V_ATTN, V_UP, V_DOWN };
struct VLayer { VKind kind; int idx; };
struct VT { void* p = nullptr; int H = 0, W = 0, C = 0; };

}  // namespace

struct VaeNet {
  bool is_encoder = false;
  sdmi_vae_cfg cfg;
  int dtype = SDMI_F16;
  std::vector<WeightSpec> specs;
  WeightStore store;
  DeviceMem mem;
  Workspace ws;
  bool finalized = false;
  std::vector<VResW> res;
  std::vector<VAttnW> attns;
  std::vector<VUpW> ups;
  std::vector<VLayer> layers;
  int block_in = 0, final_ch = 0, final_out = 0;
  float* inconv_w = nullptr; const float* inconv_b = nullptr;
  float* outconv_w = nullptr; const float* outconv_b = nullptr;
  const float *pq_w = nullptr, *pq_b = nullptr, *no_g = nullptr, *no_b = nullptr;

  ~VaeNet() { mem.release_all(); }
  void add_spec(const std::string& key, std::vector<int64_t> shape) { specs.push_back({key, std::move(shape)}); }

  void add_res(const std::string& p, int cin, int cout) {
    VResW r; r.prefix = p; r.cin = cin; r.cout = cout; r.has_skip = cin != cout;
    add_spec(p + ".norm1.weight", {cin}); add_spec(p + ".norm1.bias", {cin});
    add_spec(p + ".conv1.weight", {cout, cin, 3, 3}); add_spec(p + ".conv1.bias", {cout});
    add_spec(p + ".norm2.weight", {cout}); add_spec(p + ".norm2.bias", {cout});
    add_spec(p + ".conv2.weight", {cout, cout, 3, 3}); add_spec(p + ".conv2.bias", {cout});
    if (r.has_skip) {
      add_spec(p + ".nin_shortcut.weight", {cout, cin, 1, 1}); add_spec(p + ".nin_shortcut.bias", {cout});
    }
    res.push_back(r);
    layers.push_back({V_RES, (int)res.size() - 1});
  }
  void add_attn(const std::string& p, int ch) {
    VAttnW a; a.prefix = p; a.ch = ch;
    add_spec(p + ".norm.weight", {ch}); add_spec(p + ".norm.bias", {ch});
    for (const char* n : {"q", "k", "v", "proj_out"}) {
      add_spec(p + "." + n + ".weight", {ch, ch, 1, 1}); add_spec(p + "." + n + ".bias", {ch});
    }
    attns.push_back(a);
    layers.push_back({V_ATTN, (int)attns.size() - 1});
  }

  int build_encoder_structure() {
    const int nres = cfg.n_ch_mult;
    add_spec("encoder.conv_in.weight", {cfg.ch, cfg.out_ch, 3, 3});
    add_spec("encoder.conv_in.bias", {cfg.ch});
    int bi = cfg.ch;
    for (int lvl = 0; lvl < nres; ++lvl) {
      const int bo = cfg.ch * cfg.ch_mult[lvl];
      for (int i = 0; i < cfg.num_res_blocks; ++i) {
        add_res("encoder.down." + std::to_string(lvl) + ".block." + std::to_string(i), bi, bo);
        bi = bo;
      }
      if (lvl != nres - 1) {
        VUpW u; u.prefix = "encoder.down." + std::to_string(lvl) + ".downsample.conv"; u.ch = bi;
        add_spec(u.prefix + ".weight", {bi, bi, 3, 3}); add_spec(u.prefix + ".bias", {bi});
        ups.push_back(u);
        layers.push_back({V_DOWN, (int)ups.size() - 1});
      }
    }
    add_res("encoder.mid.block_1", bi, bi);
    add_attn("encoder.mid.attn_1", bi);
    add_res("encoder.mid.block_2", bi, bi);
    final_ch = bi;
    block_in = cfg.ch;
    add_spec("encoder.norm_out.weight", {bi}); add_spec("encoder.norm_out.bias", {bi});
    add_spec("encoder.conv_out.weight", {2 * cfg.z_channels, bi, 3, 3});
    add_spec("encoder.conv_out.bias", {2 * cfg.z_channels});
    add_spec("quant_conv.weight", {2 * cfg.embed_dim, 2 * cfg.z_channels, 1, 1});
    add_spec("quant_conv.bias", {2 * cfg.embed_dim});
    return SDMI_OK;
  }

  int build_structure() {
    if (is_encoder) return build_encoder_structure();
    const int nres = cfg.n_ch_mult;
    block_in = cfg.ch * cfg.ch_mult[nres - 1];
    add_spec("post_quant_conv.weight", {cfg.z_channels, cfg.embed_dim, 1, 1});
    add_spec("post_quant_conv.bias", {cfg.z_channels});
    add_spec("decoder.conv_in.weight", {block_in, cfg.z_channels, 3, 3});
    add_spec("decoder.conv_in.bias", {block_in});
    int bi = block_in;
    add_res("decoder.mid.block_1", bi, bi);
    add_attn("decoder.mid.attn_1", bi);
    add_res("decoder.mid.block_2", bi, bi);
    for (int lvl = nres - 1; lvl >= 0; --lvl) {
      const int bo = cfg.ch * cfg.ch_mult[lvl];
      for (int i = 0; i <= cfg.num_res_blocks; ++i) {
        add_res("decoder.up." + std::to_string(lvl) + ".block." + std::to_string(i), bi, bo);
        bi = bo;
      }
      if (lvl != 0) {
        VUpW u; u.prefix = "decoder.up." + std::to_string(lvl) + ".upsample.conv"; u.ch = bi;
        add_spec(u.prefix + ".weight", {bi, bi, 3, 3}); add_spec(u.prefix + ".bias", {bi});
        ups.push_back(u);
        layers.push_back({V_UP, (int)ups.size() - 1});
      }
    }
    final_ch = bi;
    add_spec("decoder.norm_out.weight", {bi}); add_spec("decoder.norm_out.bias", {bi});
    add_spec("decoder.conv_out.weight", {cfg.out_ch, bi, 3, 3}); add_spec("decoder.conv_out.bias", {cfg.out_ch});
    return SDMI_OK;
  }

  int finalize(hipStream_t s) {
    if (finalized) { sdmi_set_error("finalize: already finalized"); return SDMI_ESTATE; }
    SDMI_TRY(store.check_complete(specs));
    SDMI_TRY(conv_gemm_init());
    SDMI_TRY(groupnorm_static_init());
    SDMI_TRY(elementwise_init());
    std::vector<std::string> drop;
    auto P = [&](PackedConv* pc, const std::string& prefix, int cout, int cin, int taps) {
      pc->cin = cin; pc->cout = cout; pc->taps = taps;
      SDMI_TRY(mem.alloc(&pc->w, (size_t)cout * cin * taps * sdmi_dtype_size(dtype)));
      SDMI_TRY(pack_conv_weight(store.get(prefix + ".weight"), pc->w, cout, cin, taps, dtype, s));
      pc->bias = store.get(prefix + ".bias");
      drop.push_back(prefix + ".weight");
      return (int)SDMI_OK;
    };
    const std::string half = is_encoder ? "encoder" : "decoder";
    const std::string pq = is_encoder ? "quant_conv" : "post_quant_conv";
    const int in_c = is_encoder ? cfg.out_ch : cfg.z_channels;           // channels of the NCHW fp32 input
    final_out = is_encoder ? 2 * cfg.z_channels : cfg.out_ch;            // channels the last 3x3 conv writes
    pq_w = store.get(pq + ".weight"); pq_b = store.get(pq + ".bias");
    void* p;
    SDMI_TRY(mem.alloc(&p, (size_t)9 * in_c * block_in * 4)); inconv_w = (float*)p;
    SDMI_TRY(pack_inconv_weight(store.get(half + ".conv_in.weight"), inconv_w, block_in, in_c, s));
    inconv_b = store.get(half + ".conv_in.bias");
    SDMI_TRY(mem.alloc(&p, (size_t)9 * final_ch * final_out * 4)); outconv_w = (float*)p;
    SDMI_TRY(pack_outconv_weight(store.get(half + ".conv_out.weight"), outconv_w, final_out, final_ch, s));
    outconv_b = store.get(half + ".conv_out.bias");
    no_g = store.get(half + ".norm_out.weight"); no_b = store.get(half + ".norm_out.bias");
    drop.push_back(half + ".conv_in.weight"); drop.push_back(half + ".conv_out.weight");
    for (auto& r : res) {
      const std::string& q = r.prefix;
      r.gn1_g = store.get(q + ".norm1.weight"); r.gn1_b = store.get(q + ".norm1.bias");
      r.gn2_g = store.get(q + ".norm2.weight"); r.gn2_b = store.get(q + ".norm2.bias");
      SDMI_TRY(P(&r.conv1, q + ".conv1", r.cout, r.cin, 9));
      SDMI_TRY(P(&r.conv2, q + ".conv2", r.cout, r.cout, 9));
      if (r.has_skip) SDMI_TRY(P(&r.skip, q + ".nin_shortcut", r.cout, r.cin, 1));
    }
    for (auto& a : attns) {
      a.gn_g = store.get(a.prefix + ".norm.weight"); a.gn_b = store.get(a.prefix + ".norm.bias");
      SDMI_TRY(P(&a.q, a.prefix + ".q", a.ch, a.ch, 1));
      SDMI_TRY(P(&a.k, a.prefix + ".k", a.ch, a.ch, 1));
      SDMI_TRY(P(&a.v, a.prefix + ".v", a.ch, a.ch, 1));
      SDMI_TRY(P(&a.proj, a.prefix + ".proj_out", a.ch, a.ch, 1));
    }
    for (auto& u : ups) SDMI_TRY(P(&u.conv, u.prefix, u.ch, u.ch, 9));
    SDMI_CHECK_HIP(hipStreamSynchronize(s));
    for (const auto& k : drop) store.drop(k);
    finalized = true;
    return SDMI_OK;
  }

  size_t act_bytes(int B, int H, int W, int C) const { return (size_t)B * H * W * C * sdmi_dtype_size(dtype); }

  // up: 0 plain, 1 nearest x2 ahead of the conv, -1 the encoder's downsample (pad right/bottom, stride 2)
  int gemm(const void* a0, int c0, const void* w, const float* bias, int N, int taps, void* out, int ldo, int B,
           int Hin, int Win, int Ho, int Wo, int up, const void* resid, int ldr, int out_f32, hipStream_t s) {
    ConvGemmArgs g;
    g.a0 = a0; g.c0 = c0; g.w = w; g.out = out; g.ldo = ldo; g.bias = bias; g.resid = resid; g.ldr = ldr;
    g.B = B; g.Hin = Hin; g.Win = Win; g.Ho = Ho; g.Wo = Wo; g.taps = taps; g.N = N;
    g.stride = up < 0 ? 2 : 1; g.pad_lo = up < 0 ? 0 : 1; g.up = up > 0 ? 1 : 0;
    g.dtype = dtype; g.out_f32 = out_f32;
    const int M = B * Ho * Wo;
    g.splits = conv_gemm_pick_splits(M, N, taps * c0, dtype);
    if (g.splits > 1) g.partial = (float*)ws.get("splitk", (size_t)g.splits * M * N * 4);
    if (ws.dry) return SDMI_OK;
    return conv_gemm(g, s);
  }
  int gn(const VT& x, int B, const float* g, const float* b, int silu, void* out, hipStream_t s) {
    GroupNormArgs a;
    a.x0 = x.p; a.c0 = x.C; a.B = B; a.HW = x.H * x.W; a.groups = 32; a.eps = 1e-6f; a.gamma = g; a.beta = b;
    a.silu = silu; a.out = out; a.dtype = dtype;
    a.scratch = (float*)ws.get("gn_scratch", groupnorm_scratch_floats(B, a.HW, 32) * 4);
    if (ws.dry) return SDMI_OK;
    return groupnorm(a, s);
  }

  int run_res(const VResW& r, const VT& x, VT* y, const std::string& role, int B, hipStream_t s) {
    const int H = x.H, W = x.W;
    SDMI_REQUIRE(x.C == r.cin, "%s: got %d channels, expected %d", r.prefix.c_str(), x.C, r.cin);
    void* nrm = ws.get("nrm", act_bytes(B, H, W, r.cin > r.cout ? r.cin : r.cout));
    void* t1 = ws.get("t1", act_bytes(B, H, W, r.cout));
    y->p = ws.get(role, act_bytes(B, H, W, r.cout)); y->H = H; y->W = W; y->C = r.cout;
    SDMI_TRY(gn(x, B, r.gn1_g, r.gn1_b, 1, nrm, s));
    SDMI_TRY(gemm(nrm, r.cin, r.conv1.w, r.conv1.bias, r.cout, 9, t1, r.cout, B, H, W, H, W, 0, nullptr, 0, 0, s));
    VT t1t{t1, H, W, r.cout};
    SDMI_TRY(gn(t1t, B, r.gn2_g, r.gn2_b, 1, nrm, s));
    if (r.has_skip) {
      SDMI_TRY(gemm(x.p, r.cin, r.skip.w, r.skip.bias, r.cout, 1, y->p, r.cout, B, H, W, H, W, 0, nullptr, 0, 0, s));
      SDMI_TRY(gemm(nrm, r.cout, r.conv2.w, r.conv2.bias, r.cout, 9, y->p, r.cout, B, H, W, H, W, 0, y->p, r.cout, 0, s));
    } else {
      SDMI_TRY(gemm(nrm, r.cout, r.conv2.w, r.conv2.bias, r.cout, 9, y->p, r.cout, B, H, W, H, W, 0, x.p, r.cin, 0, s));
    }
    return SDMI_OK;
  }

  int run_attn(const VAttnW& a, const VT& x, VT* y, const std::string& role, int B, hipStream_t s) {
    const int H = x.H, W = x.W, N = H * W, C = a.ch;
    const size_t esz = sdmi_dtype_size(dtype);
    void* nrm = ws.get("nrm", act_bytes(B, H, W, C));
    void* q = ws.get("vq", act_bytes(B, H, W, C));
    void* k = ws.get("vk", act_bytes(B, H, W, C));
    void* vt = ws.get("vvt", (size_t)C * N * esz);             // v^T of one sample: [C][N]
    float* sc = (float*)ws.get("vscores", (size_t)N * N * 4);   // one sample at a time
    void* pr = ws.get("vprobs", (size_t)N * N * esz);
    void* o = ws.get("vo", act_bytes(B, H, W, C));
    y->p = ws.get(role, act_bytes(B, H, W, C)); y->H = H; y->W = W; y->C = C;
    SDMI_TRY(gn(x, B, a.gn_g, a.gn_b, 0, nrm, s));
    SDMI_TRY(gemm(nrm, C, a.q.w, a.q.bias, C, 1, q, C, 1, 1, B * N, 1, B * N, 0, nullptr, 0, 0, s));
    SDMI_TRY(gemm(nrm, C, a.k.w, a.k.bias, C, 1, k, C, 1, 1, B * N, 1, B * N, 0, nullptr, 0, 0, s));
    const float scale = (float)pow((double)C, -0.5);
    // The P v product below contracts over the tokens in whole K-tiles.  Token counts that are not (a latent whose
    // H * W is no multiple of 64; never the case behind the UNet, whose latents are multiples of 8 per side, but the
    // reference's autoencoder takes any size): v as a plain projection and the vector-ALU attention kernel with one
    // head of width C, which has no such granularity.
    const int bk = dtype == SDMI_F32 ? 32 : 64;
    if (N % bk != 0) {
      void* v = ws.get("vv", act_bytes(B, H, W, C));
      SDMI_TRY(gemm(nrm, C, a.v.w, a.v.bias, C, 1, v, C, 1, 1, B * N, 1, B * N, 0, nullptr, 0, 0, s));
      if (!ws.dry) {
        AttentionArgs at;
        at.q = q; at.ldq = C; at.k = k; at.ldk = C; at.v = v; at.ldv = C; at.out = o; at.ldo = C;
        at.B = B; at.heads = 1; at.Nq = N; at.Nk = N; at.d = C; at.scale = scale; at.dtype = dtype; at.impl = 1;
        SDMI_TRY(attention(at, s));
      }
      SDMI_TRY(gemm(o, C, a.proj.w, a.proj.bias, C, 1, y->p, C, 1, 1, B * N, 1, B * N, 0, x.p, C, 0, s));
      return SDMI_OK;
    }
    for (int b = 0; b < B; ++b) {
      const char* hb = (const char*)nrm + (size_t)b * N * C * esz;
      const char* qb = (const char*)q + (size_t)b * N * C * esz;
      const char* kb = (const char*)k + (size_t)b * N * C * esz;
      char* ob = (char*)o + (size_t)b * N * C * esz;
      // v^T[c][tok] = sum_k Wv[c][k] h[tok][k]   (bias deferred): rows = Wv, "weights" = activations
      SDMI_TRY(gemm(a.v.w, C, hb, nullptr, N, 1, vt, N, 1, 1, C, 1, C, 0, nullptr, 0, 0, s));
      // S = q k^T (fp32)
      SDMI_TRY(gemm(qb, C, kb, nullptr, N, 1, sc, N, 1, 1, N, 1, N, 0, nullptr, 0, 1, s));
      if (!ws.dry) SDMI_TRY(softmax_rows(sc, pr, N, N, scale, dtype, s));
      // O = P v + b_v
      SDMI_TRY(gemm(pr, N, vt, a.v.bias, C, 1, ob, C, 1, 1, N, 1, N, 0, nullptr, 0, 0, s));
    }
    SDMI_TRY(gemm(o, C, a.proj.w, a.proj.bias, C, 1, y->p, C, 1, 1, B * N, 1, B * N, 0, x.p, C, 0, s));
    return SDMI_OK;
  }

  // decoder: z [B][z_channels][H][W] -> image [B][out_ch][8H][8W]; encoder: image [B][out_ch][H][W] ->
  // moments [B][2 embed_dim][H/8][W/8]
  int run(const float* z, float z_scale, float* out, int B, int H, int W, hipStream_t s) {
    const int in_c = is_encoder ? cfg.out_ch : cfg.z_channels;
    float* zq = is_encoder ? nullptr : (float*)ws.get("zq", (size_t)B * cfg.z_channels * H * W * 4);
    VT h;
    h.p = ws.get("h_a", act_bytes(B, H, W, block_in)); h.H = H; h.W = W; h.C = block_in;
    if (!ws.dry) {
      if (!is_encoder) SDMI_TRY(pointwise_nchw(z, pq_w, pq_b, zq, B, cfg.embed_dim, cfg.z_channels, H * W, z_scale, s));
      InConvArgs a;
      a.src[0] = is_encoder ? z : zq; a.c[0] = in_c; a.w = inconv_w; a.bias = inconv_b; a.out = h.p;
      a.B = B; a.H = H; a.W = W; a.Cout = block_in; a.dtype = dtype;
      SDMI_TRY(in_conv(a, s));
    }
    int flip = 1;
    for (const VLayer& l : layers) {
      const std::string role = flip ? "h_b" : "h_a";
      VT y;
      if (l.kind == V_RES) {
        SDMI_TRY(run_res(res[l.idx], h, &y, role, B, s));
      } else if (l.kind == V_ATTN) {
        SDMI_TRY(run_attn(attns[l.idx], h, &y, role, B, s));
      } else if (l.kind == V_UP) {
        const VUpW& u = ups[l.idx];
        y.H = 2 * h.H; y.W = 2 * h.W; y.C = u.ch;
        y.p = ws.get(role, act_bytes(B, y.H, y.W, u.ch));
        SDMI_TRY(gemm(h.p, h.C, u.conv.w, u.conv.bias, u.ch, 9, y.p, u.ch, B, h.H, h.W, y.H, y.W, 1, nullptr, 0, 0, s));
      } else {
        const VUpW& u = ups[l.idx];
        y.H = (h.H + 1 - 3) / 2 + 1; y.W = (h.W + 1 - 3) / 2 + 1; y.C = u.ch;
        y.p = ws.get(role, act_bytes(B, y.H, y.W, u.ch));
        SDMI_TRY(gemm(h.p, h.C, u.conv.w, u.conv.bias, u.ch, 9, y.p, u.ch, B, h.H, h.W, y.H, y.W, -1, nullptr, 0, 0, s));
      }
      h = y;
      flip ^= 1;
    }
    void* nrm = ws.get("nrm", act_bytes(B, h.H, h.W, h.C));
    SDMI_TRY(gn(h, B, no_g, no_b, 1, nrm, s));
    if (is_encoder) {
      float* mom = (float*)ws.get("moments", (size_t)B * final_out * h.H * h.W * 4);
      if (!ws.dry) {
        SDMI_TRY(out_conv(nrm, outconv_w, outconv_b, mom, B, h.H, h.W, h.C, final_out, dtype, s));
        SDMI_TRY(pointwise_nchw(mom, pq_w, pq_b, out, B, final_out, 2 * cfg.embed_dim, h.H * h.W, 1.0f, s));
      }
    } else if (!ws.dry) {
      SDMI_TRY(out_conv(nrm, outconv_w, outconv_b, out, B, h.H, h.W, h.C, final_out, dtype, s));
    }
    return SDMI_OK;
  }

  int forward(const float* z, float z_scale, float* out, int B, int H, int W, hipStream_t s) {
    if (!finalized) { sdmi_set_error("forward before finalize"); return SDMI_ESTATE; }
    SDMI_REQUIRE(z && out && B > 0 && H > 0 && W > 0, "vae forward: bad arguments");
    if (is_encoder)
      SDMI_REQUIRE(H % (1 << (cfg.n_ch_mult - 1)) == 0 && W % (1 << (cfg.n_ch_mult - 1)) == 0,
                   "vae encoder: image size %dx%d must be divisible by %d", H, W, 1 << (cfg.n_ch_mult - 1));
    ws.dry = true; ws.need.clear();
    SDMI_TRY(run(z, z_scale, out, B, H, W, s));
    ws.dry = false;
    if (!ws.fits()) {
      SDMI_CHECK_HIP(hipStreamSynchronize(s));
      SDMI_TRY(ws.commit());
    }
    return run(z, z_scale, out, B, H, W, s);
  }
};

struct sdmi_vae_decoder : VaeNet {};
struct sdmi_vae_encoder : VaeNet {};

namespace {
template <typename H>
int create_vae(const sdmi_vae_cfg* cfg, H** out, bool encoder) {
  SDMI_REQUIRE(cfg && out, "vae create: null argument");
  SDMI_REQUIRE(cfg->n_ch_mult > 0 && cfg->n_ch_mult <= 8, "vae create: bad ch_mult length");
  SDMI_REQUIRE(cfg->compute_dtype >= 0 && cfg->compute_dtype <= 2, "vae create: bad compute_dtype");
  SDMI_REQUIRE(cfg->ch % 64 == 0 && (cfg->out_ch == 3 || cfg->out_ch == 4) && cfg->z_channels <= 64,
               "vae create: ch must be a multiple of 64, image channels 3 or 4");
  if (encoder)
    SDMI_REQUIRE(2 * cfg->z_channels == 8, "vae encoder: z_channels must be 4 (the last conv writes 8 channels)");
  std::unique_ptr<H> v(new H());
  v->is_encoder = encoder;
  v->cfg = *cfg;
  v->dtype = cfg->compute_dtype;
  v->store.mem = &v->mem;
  v->ws.mem = &v->mem;
  SDMI_TRY(v->build_structure());
  *out = v.release();
  return SDMI_OK;
}
int vae_weight_info(const VaeNet* v, int index, const char** key, int64_t* shape4, int* ndim) {
  SDMI_REQUIRE(v && index >= 0 && index < (int)v->specs.size(), "weight_info: bad index");
  const WeightSpec& s = v->specs[index];
  if (key) *key = s.key.c_str();
  if (ndim) *ndim = (int)s.shape.size();
  if (shape4)
    for (size_t i = 0; i < 4; ++i) shape4[i] = i < s.shape.size() ? s.shape[i] : 1;
  return SDMI_OK;
}
int vae_set_weight(VaeNet* v, const char* key, const void* dev_ptr, const int64_t* shape, int ndim, int dtype,
                   void* stream) {
  SDMI_REQUIRE(v && key && dev_ptr && shape, "set_weight: null argument");
  if (v->finalized) { sdmi_set_error("set_weight after finalize"); return SDMI_ESTATE; }
  return v->store.set(v->specs, key, dev_ptr, shape, ndim, dtype, (hipStream_t)stream);
}
}  // namespace

extern "C" {

int sdmi_vae_decoder_create(const sdmi_vae_cfg* cfg, sdmi_vae_decoder** out) { return create_vae(cfg, out, false); }
int sdmi_vae_encoder_create(const sdmi_vae_cfg* cfg, sdmi_vae_encoder** out) { return create_vae(cfg, out, true); }
int sdmi_vae_encoder_num_weights(const sdmi_vae_encoder* v) { return v ? (int)v->specs.size() : 0; }
int sdmi_vae_encoder_weight_info(const sdmi_vae_encoder* v, int index, const char** key, int64_t* shape4, int* ndim) {
  return vae_weight_info(v, index, key, shape4, ndim);
}
int sdmi_vae_encoder_set_weight(sdmi_vae_encoder* v, const char* key, const void* dev_ptr, const int64_t* shape,
                                int ndim, int dtype, void* stream) {
  return vae_set_weight(v, key, dev_ptr, shape, ndim, dtype, stream);
}
int sdmi_vae_encoder_finalize(sdmi_vae_encoder* v, void* stream) {
  SDMI_REQUIRE(v, "finalize: null handle");
  return v->finalize((hipStream_t)stream);
}
int sdmi_vae_encoder_forward(sdmi_vae_encoder* v, const void* x, void* moments, int batch, int h, int w,
                             void* stream) {
  SDMI_REQUIRE(v, "forward: null handle");
  return v->forward((const float*)x, 1.0f, (float*)moments, batch, h, w, (hipStream_t)stream);
}
void sdmi_vae_encoder_destroy(sdmi_vae_encoder* v) { delete v; }
int sdmi_gaussian_sample(const void* moments, const void* noise, float scale, void* z, int batch, int channels,
                         int hw, void* stream) {
  return gaussian_sample((const float*)moments, (const float*)noise, scale, (float*)z, batch, channels, hw,
                         (hipStream_t)stream);
}

int sdmi_vae_decoder_num_weights(const sdmi_vae_decoder* v) { return v ? (int)v->specs.size() : 0; }

int sdmi_vae_decoder_weight_info(const sdmi_vae_decoder* v, int index, const char** key, int64_t* shape4, int* ndim) {
  return vae_weight_info(v, index, key, shape4, ndim);
}

int sdmi_vae_decoder_set_weight(sdmi_vae_decoder* v, const char* key, const void* dev_ptr, const int64_t* shape,
                                int ndim, int dtype, void* stream) {
  return vae_set_weight(v, key, dev_ptr, shape, ndim, dtype, stream);
}

int sdmi_vae_decoder_finalize(sdmi_vae_decoder* v, void* stream) {
  SDMI_REQUIRE(v, "finalize: null handle");
  return v->finalize((hipStream_t)stream);
}

int sdmi_vae_decoder_forward(sdmi_vae_decoder* v, const void* z, float z_scale, void* out, int batch, int h, int w,
                             void* stream) {
  SDMI_REQUIRE(v, "forward: null handle");
  return v->forward((const float*)z, z_scale, (float*)out, batch, h, w, (hipStream_t)stream);
}

void sdmi_vae_decoder_destroy(sdmi_vae_decoder* v) { delete v; }

}  // extern "C"
