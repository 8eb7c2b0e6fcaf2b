// Host-side launchers of the gfx950 kernels.  All of them enqueue on `stream` and return at once;
// none allocates, frees or synchronises.  Activations are NHWC ("token-major": [rows][channels]).
#pragma once
#include "common.h"

// ---- optional per-kernel-class timing with HIP events (eager launches only; off by default) -------
enum { PROF_CONV_GEMM = 0, PROF_ATTENTION = 1, PROF_GROUPNORM = 2, PROF_LAYERNORM = 3, PROF_OTHER = 4, PROF_NCLS = 5 };
void prof_begin(int cls, hipStream_t stream, double flops, double bytes, const char* label = nullptr);
void prof_end(int cls, hipStream_t stream);
struct ProfScope {
  int cls; hipStream_t s;
  ProfScope(int c, hipStream_t st, double flops, double bytes, const char* label = nullptr) : cls(c), s(st) {
    prof_begin(c, st, flops, bytes, label);
  }
  ~ProfScope() { prof_end(cls, s); }
};

// ---- implicit-GEMM convolution / linear ---------------------------------------------------------
// out[m][n] = epilogue( sum_k A(m,k) * Wp[n][k] ),  m = output pixel (b, oy, ox), n = out channel,
// k = (tap, channel of the (possibly two-source) input).  Wp is the packed weight [N][K].
struct ConvGemmArgs {
  const void* a0 = nullptr;     // first input, NHWC [B][Hin][Win][c0]
  const void* a1 = nullptr;     // optional second input (channel concat), NHWC [B][Hin][Win][c1]
  int c0 = 0, c1 = 0;
  const void* w = nullptr;      // packed weights [N][taps*(c0+c1)], element type = dtype
  void* out = nullptr;          // [M][ldo] (dtype)
  int ldo = 0;
  const float* bias = nullptr;  // [N] fp32 or null
  const float* rowadd = nullptr;// [B][ldra] fp32 or null: added per sample (time-embedding projection)
  int ldra = 0;                 // row stride of rowadd (0 = N, < 0 = every sample reads row 0)
  int out_f32 = 0;              // 1: store fp32 regardless of dtype (attention scores)
  const void* resid = nullptr;  // [M][ldr] (dtype) or null; may alias out
  int ldr = 0;
  int B = 1, Hin = 1, Win = 1;  // input spatial size (before the optional x2 nearest upsample)
  int Ho = 1, Wo = 1;           // output spatial size
  int taps = 1;                 // 1 (pointwise / linear) or 9 (3x3, pad 1)
  int stride = 1;               // 1 or 2
  int pad_lo = 1;               // 3x3: zero rows/columns ahead of the image (the trailing side always has 1);
                                // 0 = F.pad(x, (0,1,0,1)) + conv(padding=0), the VAE encoder's downsample
  int up = 0;                   // 1: input is nearest-upsampled x2 on the fly
  int N = 0;                    // output channels (rows of Wp)
  int geglu = 0;                // 1: Wp rows interleave (value,gate) 16-row groups; out has N/2 columns
  int dtype = SDMI_F16;
  // split-K: when splits > 1 the kernel writes fp32 partial sums to `partial` ([splits][M][N]) and
  // conv_gemm() launches the reduction+epilogue kernel itself.
  int splits = 1;
  float* partial = nullptr;
  // LayerNorm folded into this GEMM (token GEMMs only): the operand rows are the raw x, `w` carries gamma, `bias`
  // is b' = b + W beta, ln_colsum[n] = sum_k w[n][k] and ln_stats[m] = (rstd_m, rstd_m * mean_m) from
  // layernorm_stats(); the sums become rstd_m * (S_mn - mean_m * colsum_n) + b'_n ahead of the rest of the epilogue
  const float* ln_stats = nullptr;
  const float* ln_colsum = nullptr;
  // ln_stats in the form a producing GEMM leaves it (row_stats_out below): ln_nparts pairs (sum, sum of squares)
  // per row, folded and turned into (rstd, rstd * mean) with ln_eps where they are used; 0 = layernorm_stats() form
  int ln_nparts = 0;
  float ln_eps = 1e-5f;
  // not null: the launch also leaves, per output row, conv_gemm_row_stat_parts(...) pairs (sum, sum of squares of
  // the stored values over the row's N columns) at row_stats_out[row * parts ...] (rows m + dup_rows as well), for a
  // LayerNorm folded into the GEMM that reads this output.  Not with GEGLU or fp32 outputs.
  float* row_stats_out = nullptr;
  // > 0: every output row m is also stored at row m + dup_rows of `out` (two samples that are known to
  // be identical so far are computed once); not with GEGLU or fp32 outputs
  int dup_rows = 0;
  // extra pointwise sources behind the taps (a 1x1 convolution over other tensors summed into this launch, e.g. a
  // ResBlock's skip_connection into its second 3x3 convolution): NHWC [B][Ho][Wo][ec0] (and [ec1]), read at the
  // output pixel; the packed weight rows continue with their columns: K = taps * (c0 + c1) + ec0 + ec1.
  // 16-bit types on the large-tile kernels only (conv_gemm_extra_ok).
  const void* e0 = nullptr; const void* e1 = nullptr;
  int ec0 = 0, ec1 = 0;
  // > 0: `w` holds one [N][K] matrix per sample, back to back, and output row m uses matrix m / w_rows (token GEMMs
  // in 16-bit types, no split-K; conv_gemm_per_sample_ok() tells whether a tile shape divides w_rows).  With
  // vec_per_sample, bias and ln_colsum are per sample as well ([samples][N]).
  int w_rows = 0;
  int vec_per_sample = 0;
  // 1 (per-sample launches only): every 80-column segment of an output row is replaced by its base-2 softmax
  int softmax_seg = 0;
};
int conv_gemm(const ConvGemmArgs& a, hipStream_t stream);
bool conv_gemm_extra_ok(int M, int N, int K, int dtype);     // K = all columns, the extra sources' included
bool conv_gemm_per_sample_ok(int M, int N, int K, int dtype, int rows_per_sample, bool seg);
int conv_gemm_init();   // one-time kernel attribute setup; call outside stream capture
// smallest split count worth using for this shape on a 256-CU part (1 = no split)
int conv_gemm_pick_splits(int M, int N, int K, int dtype, bool geglu = false);
// pairs per row that a launch of this shape with `splits` writes to row_stats_out
int conv_gemm_row_stat_parts(int M, int N, int K, int dtype, bool geglu, int splits, int rows_per_sample = 0);

// ---- normalisation ---------------------------------------------------------------------------
struct GroupNormArgs {
  const void* x0 = nullptr;     // NHWC [B][HW][c0]
  const void* x1 = nullptr;     // optional second source (channel concat) [B][HW][c1]
  int c0 = 0, c1 = 0;
  int B = 1, HW = 1;
  int groups = 32;
  float eps = 1e-5f;
  const float* gamma = nullptr; // [C] fp32
  const float* beta = nullptr;  // [C] fp32
  int silu = 0;
  void* out = nullptr;          // NHWC [B][HW][C]
  float* scratch = nullptr;     // >= groupnorm_scratch_floats(B, HW) floats
  int dtype = SDMI_F16;
};
size_t groupnorm_scratch_floats(int B, int HW, int groups);
int groupnorm(const GroupNormArgs& a, hipStream_t stream);
// same, with the library's built-in per-GPU scratch (a.scratch is ignored); allocates nothing
int groupnorm_static(const GroupNormArgs& a, hipStream_t stream);
int groupnorm_static_init(float** out = nullptr);      // resolves the built-in scratch of the current device (not under capture)

int layernorm(const void* x, void* out, const float* gamma, const float* beta, int rows, int d,
              float eps, int dtype, hipStream_t stream);

// per-row (rstd, rstd * mean) of nn.LayerNorm, for GEMMs that apply the normalisation in their epilogue
int layernorm_stats(const void* x, float* stats, int rows, int d, float eps, int dtype, hipStream_t stream);

// ---- attention -------------------------------------------------------------------------------
struct AttentionArgs {
  const void* q = nullptr; int ldq = 0;   // [B][Nq][ldq], head h at columns [h*d, (h+1)*d)
  const void* k = nullptr; int ldk = 0;   // [B][Nk][ldk]
  const void* v = nullptr; int ldv = 0;   // [B][Nk][ldv]
  void* out = nullptr;     int ldo = 0;   // [B][Nq][ldo]
  int B = 1, heads = 1, Nq = 0, Nk = 0, d = 0;
  float scale = 1.f;
  int dtype = SDMI_F16;
  int impl = 0;                           // 0 = auto, 1 = force the scalar (fp32-math) kernel
  int causal = 0;                         // 1: query i attends keys 0..i only (text transformer)
};
int attention(const AttentionArgs& a, hipStream_t stream);
int attention_init();   // one-time kernel attribute setup; call outside stream capture
int elementwise_init(); // same, for the first-convolution kernel

// ---- small / elementwise ---------------------------------------------------------------------
// emb[b][:] = [cos(t*f) | sin(t*f)], f_k = exp(-ln(max_period) k / half)            (fp32 out)
int timestep_embedding(const int64_t* t, float* out, int B, int dim, float max_period, hipStream_t s);
// y[m][n] = act_out( sum_k act_in(x[m][k]) * W[n][k] + b[n] ), fp32 everywhere, tiny M
// act: 0 none, 1 SiLU
int small_linear(const float* x, const float* W, const float* bias, float* y, int M, int K, int N,
                 int act_in, int act_out, hipStream_t s);

// first convolution: NCHW fp32 sources (channel-concatenated) -> NHWC dtype, 3x3 pad 1 stride 1
struct InConvArgs {
  const float* src[3] = {nullptr, nullptr, nullptr};   // each [B][c_i][H][W]
  int c[3] = {0, 0, 0};
  const float* w = nullptr;      // packed [9*Cin][Cout] fp32 (tap-major, then input channel)
  const float* bias = nullptr;   // [Cout]
  void* out = nullptr;           // NHWC [B][H][W][Cout]
  int B = 1, H = 1, W = 1, Cout = 0;
  int dtype = SDMI_F16;
};
int in_conv(const InConvArgs& a, hipStream_t s);
// last convolution: NHWC dtype -> NCHW fp32, 3x3 pad 1, Cout in {3, 4, 8}
int out_conv(const void* x, const float* w /*[Cout][9][Cin] fp32*/, const float* bias, float* out,
             int B, int H, int W, int Cin, int Cout, int dtype, hipStream_t s);
// pointwise NCHW fp32 conv with tiny channel counts: out = W*(scale*x) + b  (post_quant_conv)
int pointwise_nchw(const float* x, const float* w, const float* bias, float* out, int B, int Cin,
                   int Cout, int HW, float scale, hipStream_t s);

// posterior sample of the VAE encoder: z = scale * (mean + exp(0.5 clamp(logvar)) * noise); noise null = mode
int gaussian_sample(const float* moments, const float* noise, float scale, float* z, int B, int C, int HW,
                    hipStream_t s);

// fused classifier-free guidance + DDIM update (all fp32, NCHW or any layout: purely elementwise)
//   e = e_u + scale (e_c - e_u);  pred_x0 = (x - sqrt_1m_at e)/sqrt(a_t);
//   x_prev = sqrt(a_prev) pred_x0 + sqrt(1 - a_prev - sigma^2) e + sigma * noise * temperature
// eps points at [e_u | e_c] (each n elements) when cfg != 0, else at e (n elements).
int ddim_step(const float* x, const float* eps, int cfg, float scale, float a_t, float a_prev,
              float sigma, float sqrt_1m_at, const float* noise, float temperature, float* x_prev,
              float* pred_x0, int64_t n, hipStream_t s);
// DDPM ancestral update with the schedule tables' entries at the step's timestep as scalars (fp32):
//   x0 = c_recip x - c_recipm1 eps (clamped to [-1,1] if clip); x_prev = coef1 x0 + coef2 x + noise_coef (noise temperature)
int ddpm_step(const float* x, const float* eps, float c_recip, float c_recipm1, int clip, float coef1, float coef2,
              float noise_coef, const float* noise, float temperature, float* x_prev, float* pred_x0, int64_t n,
              hipStream_t s);
// out = ca * a + cb * b   (stochastic_encode / q_sample with per-call scalars; fp32)
int cfg_combine(const float* eps, float scale, float* out, int64_t n, hipStream_t s);
int axpby(const float* a, const float* b, float ca, float cb, float* out, int64_t n, hipStream_t s);
// known-region blend of the DDIM loop (fp32): out = (ca x0 + cb noise) * mask + (1 - mask) * img; mask has 1 or C channels
int q_sample_blend(const float* x0, const float* noise, const float* mask, const float* img, float ca, float cb,
                   float* out, int B, int C, int HW, int mask_channels, hipStream_t s);
// row softmax over fp32 scores -> dtype probabilities (VAE mid-block attention, two-pass path)
int softmax_rows(const float* s, void* p, int rows, int cols, float scale, int dtype, hipStream_t st);
// out[i] = clamp((x[i]+1)/2, 0, 1)
int to_image(const float* x, float* out, int64_t n, hipStream_t s);
int to_image_u8(const float* x, unsigned char* out, int B, int C, int HW, hipStream_t s);
int inpaint_composite(const float* image, const float* mask, const float* pred, float* out, int B, int C, int HW,
                      hipStream_t s);

// ---- weight packing (device side, run once at load) ---------------------------------------------
// OIHW fp32 -> [O][tap][I] dtype (tap = ky*3+kx); also serves (O,I) linear weights with taps = 1
int pack_conv_weight(const float* w, void* out, int O, int I, int taps, int dtype, hipStream_t s);
// rows of a (2F, I) fp32 matrix -> (value,gate)-interleaved 16-row groups; same permutation on bias
int pack_geglu_weight(const float* w, const float* b, void* wout, float* bout, int F, int I, int dtype,
                      hipStream_t s);
// LayerNorm -> Linear folding, run once at load: w_out = w diag(gamma) (fp32 [N][K]), b_out = bias + w beta;
// packed_rowsum: c[n] = sum_k wp[n][k] of a packed matrix in its stored precision
int fold_ln_weight(const float* w, const float* gamma, const float* beta, const float* bias, float* w_out,
                   float* b_out, int N, int K, hipStream_t s);
int packed_rowsum(const void* wp, float* c, int N, int K, int dtype, hipStream_t s);
// OIHW fp32 -> [tap][I][O] fp32 for in_conv
int pack_inconv_weight(const float* w, float* out, int O, int I, hipStream_t s);
// OIHW fp32 -> [O][tap][I] fp32 for out_conv
int pack_outconv_weight(const float* w, float* out, int O, int I, hipStream_t s);
// cross-attention folded per prompt (xattn_fold.hip): from the cached K | V projections kv [B*ntok][2*d], the
// gamma-carrying query weights wq [d][d] (rows (head, e)) with qb = Wq beta, and the output weights wo [d][d]:
// G [B][heads*80][d], g0 [B][heads*80] (pad columns -30000), U [B][d][heads*80]; c0 = scale * log2(e)
int xattn_fold(const void* kv, const void* wq, const float* qb, const void* wo, void* G, float* g0, void* U, int B,
               int ntok, int heads, int dh, int d, float c0, int dtype, hipStream_t s);
// text transformer: x[row][:] = tok[ids[row]][:] + pos[row % T_len][:] (fp32 tables, dtype out); x * sigmoid(1.702 x)
int embed_tokens(const int64_t* ids, const float* tok, const float* pos, void* out, int rows, int T_len, int d,
                 int vocab, int dtype, hipStream_t s);
int quick_gelu(void* x, int64_t n, int dtype, hipStream_t s);
int convert_f32(const float* x, void* out, int64_t n, int dtype, hipStream_t s);
int convert_to_f32(const void* x, float* out, int64_t n, int dtype, hipStream_t s);
