// C-ABI glue for the per-op entry points of libsdmi (see include/sdmi.h) and the error channel.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/sdmi.h"
#include "ops.h"

static thread_local char g_err[2048] = "";

void sdmi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- profiler -----------------------------------------------------------------------------------
#include <string.h>
#include <vector>
namespace {
struct ProfEntry { hipEvent_t a, b; int cls; double flops, bytes; char label[64]; };
bool g_prof_on = false;
std::vector<ProfEntry> g_prof_entries;
std::vector<hipEvent_t> g_prof_pool;
double g_prof_flops[PROF_NCLS], g_prof_bytes[PROF_NCLS];
int64_t g_prof_launches[PROF_NCLS];
const char* const kProfNames[PROF_NCLS] = {"conv_gemm", "attention", "groupnorm", "layernorm", "other"};
hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void prof_begin(int cls, hipStream_t stream, double flops, double bytes, const char* label) {
  if (!g_prof_on) return;
  ProfEntry e{prof_event(), prof_event(), cls, flops, bytes, {0}};
  if (label) strncpy(e.label, label, sizeof(e.label) - 1);
  (void)hipEventRecord(e.a, stream);
  g_prof_entries.push_back(e);
  g_prof_flops[cls] += flops; g_prof_bytes[cls] += bytes; g_prof_launches[cls] += 1;
}
void prof_end(int cls, hipStream_t stream) {
  if (!g_prof_on) return;
  for (size_t i = g_prof_entries.size(); i-- > 0;)
    if (g_prof_entries[i].cls == cls) { (void)hipEventRecord(g_prof_entries[i].b, stream); return; }
}

extern "C" {

int sdmi_prof_enable(int on) {
  g_prof_on = on != 0;
  return SDMI_OK;
}
int sdmi_prof_reset(void) {
  for (auto& e : g_prof_entries) { g_prof_pool.push_back(e.a); g_prof_pool.push_back(e.b); }
  g_prof_entries.clear();
  memset(g_prof_flops, 0, sizeof(g_prof_flops));
  memset(g_prof_bytes, 0, sizeof(g_prof_bytes));
  memset(g_prof_launches, 0, sizeof(g_prof_launches));
  return SDMI_OK;
}
int sdmi_prof_get(const char* kernel_class, double* ms_total, int64_t* launches, double* flops, double* bytes) {
  SDMI_REQUIRE(kernel_class, "prof_get: null class");
  int cls = -1;
  for (int i = 0; i < PROF_NCLS; ++i)
    if (strcmp(kernel_class, kProfNames[i]) == 0) cls = i;
  SDMI_REQUIRE(cls >= 0, "prof_get: unknown class '%s'", kernel_class);
  SDMI_CHECK_HIP(hipDeviceSynchronize());
  double ms = 0.0;
  for (const auto& e : g_prof_entries)
    if (e.cls == cls) {
      float t = 0.f;
      SDMI_CHECK_HIP(hipEventElapsedTime(&t, e.a, e.b));
      ms += t;
    }
  if (ms_total) *ms_total = ms;
  if (launches) *launches = g_prof_launches[cls];
  if (flops) *flops = g_prof_flops[cls];
  if (bytes) *bytes = g_prof_bytes[cls];
  return SDMI_OK;
}

int sdmi_prof_calibrate(int n, void* stream, double* us_per_bracket) {
  SDMI_REQUIRE(n > 0 && us_per_bracket, "prof_calibrate: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  std::vector<hipEvent_t> ev(2 * (size_t)n);
  for (auto& e : ev) e = prof_event();
  for (int i = 0; i < n; ++i) {
    SDMI_CHECK_HIP(hipEventRecord(ev[2 * i], s));
    SDMI_CHECK_HIP(hipEventRecord(ev[2 * i + 1], s));
  }
  SDMI_CHECK_HIP(hipStreamSynchronize(s));
  double tot = 0.0;
  for (int i = 0; i < n; ++i) {
    float t = 0.f;
    SDMI_CHECK_HIP(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
    tot += t;
  }
  for (auto& e : ev) g_prof_pool.push_back(e);
  *us_per_bracket = tot * 1e3 / n;
  return SDMI_OK;
}

int sdmi_prof_dump(const char* path) {
  SDMI_REQUIRE(path, "prof_dump: null path");
  SDMI_CHECK_HIP(hipDeviceSynchronize());
  FILE* f = fopen(path, "w");
  SDMI_REQUIRE(f, "prof_dump: cannot open '%s'", path);
  fprintf(f, "class,label,us,flops,bytes\n");
  for (const auto& e : g_prof_entries) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) t = -1.f;
    fprintf(f, "%s,%s,%.3f,%.0f,%.0f\n", kProfNames[e.cls], e.label, t * 1e3, e.flops, e.bytes);
  }
  fclose(f);
  return SDMI_OK;
}

int sdmi_version(void) { return SDMI_VERSION; }
const char* sdmi_last_error(void) { return g_err; }

int sdmi_ddim_step(const void* x, const void* eps, int cfg, float scale, float a_t, float a_prev, float sigma,
                   float sqrt_one_minus_at, const void* noise, float temperature, void* x_prev, void* pred_x0,
                   int64_t n, void* stream) {
  return ddim_step((const float*)x, (const float*)eps, cfg, scale, a_t, a_prev, sigma, sqrt_one_minus_at,
                   (const float*)noise, temperature, (float*)x_prev, (float*)pred_x0, n, (hipStream_t)stream);
}

int sdmi_ddpm_step(const void* x, const void* eps, float sqrt_recip_acp, float sqrt_recipm1_acp, int clip_denoised,
                   float mean_coef1, float mean_coef2, float noise_coef, const void* noise, float temperature,
                   void* x_prev, void* pred_x0, int64_t n, void* stream) {
  return ddpm_step((const float*)x, (const float*)eps, sqrt_recip_acp, sqrt_recipm1_acp, clip_denoised, mean_coef1,
                   mean_coef2, noise_coef, (const float*)noise, temperature, (float*)x_prev, (float*)pred_x0, n,
                   (hipStream_t)stream);
}

int sdmi_cfg_combine(const void* eps, float scale, void* out, int64_t n, void* stream) {
  return cfg_combine((const float*)eps, scale, (float*)out, n, (hipStream_t)stream);
}

int sdmi_axpby(const void* a, const void* b, float ca, float cb, void* out, int64_t n, void* stream) {
  return axpby((const float*)a, (const float*)b, ca, cb, (float*)out, n, (hipStream_t)stream);
}

int sdmi_q_sample_blend(const void* x0, const void* noise, const void* mask, const void* img, float sqrt_acp,
                        float sqrt_one_minus_acp, void* out, int batch, int channels, int hw, int mask_channels,
                        void* stream) {
  return q_sample_blend((const float*)x0, (const float*)noise, (const float*)mask, (const float*)img, sqrt_acp,
                        sqrt_one_minus_acp, (float*)out, batch, channels, hw, mask_channels, (hipStream_t)stream);
}

int sdmi_to_image_u8(const void* x, void* out, int batch, int channels, int hw, void* stream) {
  return to_image_u8((const float*)x, (unsigned char*)out, batch, channels, hw, (hipStream_t)stream);
}
int sdmi_inpaint_composite(const void* image, const void* mask, const void* predicted, void* out, int batch,
                           int channels, int hw, void* stream) {
  return inpaint_composite((const float*)image, (const float*)mask, (const float*)predicted, (float*)out, batch,
                           channels, hw, (hipStream_t)stream);
}
int sdmi_to_image(const void* x, void* out, int64_t n, void* stream) {
  return to_image((const float*)x, (float*)out, n, (hipStream_t)stream);
}

int64_t sdmi_groupnorm_scratch_bytes(int batch, int groups) {
  return (int64_t)groupnorm_scratch_floats(batch, 0, groups) * 4;
}

int sdmi_groupnorm(const void* x0, int c0, const void* x1, int c1, int batch, int hw, int groups, float eps,
                   const float* gamma, const float* beta, int silu, void* out, void* scratch, int dtype,
                   void* stream) {
  GroupNormArgs a;
  a.x0 = x0; a.c0 = c0; a.x1 = x1; a.c1 = c1; a.B = batch; a.HW = hw; a.groups = groups; a.eps = eps;
  a.gamma = gamma; a.beta = beta; a.silu = silu; a.out = out; a.scratch = (float*)scratch; a.dtype = dtype;
  return groupnorm(a, (hipStream_t)stream);
}

int sdmi_groupnorm_static(const void* x0, int c0, const void* x1, int c1, int batch, int hw, int groups, float eps,
                          const float* gamma, const float* beta, int silu, void* out, int dtype, void* stream) {
  GroupNormArgs a;
  a.x0 = x0; a.c0 = c0; a.x1 = x1; a.c1 = c1; a.B = batch; a.HW = hw; a.groups = groups; a.eps = eps;
  a.gamma = gamma; a.beta = beta; a.silu = silu; a.out = out; a.dtype = dtype;
  return groupnorm_static(a, (hipStream_t)stream);
}

int sdmi_layernorm(const void* x, void* out, const float* gamma, const float* beta, int rows, int d, float eps,
                   int dtype, void* stream) {
  return layernorm(x, out, gamma, beta, rows, d, eps, dtype, (hipStream_t)stream);
}

// workspace layout of sdmi_layernorm_linear: [folded fp32 W][b' (source order)][b' (packed order)][colsum][row stats]
// [packed W][split-K slabs], every part rounded up to 256 bytes
namespace {
struct LnLinearWs { size_t fw, fb, pb, cs, st, pw, sk, total; };
LnLinearWs ln_linear_ws(int rows, int k, int n, int splits, int dtype) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  LnLinearWs w;
  size_t o = 0;
  w.fw = o; o += up((size_t)n * k * 4);
  w.fb = o; o += up((size_t)n * 4);
  w.pb = o; o += up((size_t)n * 4);
  w.cs = o; o += up((size_t)n * 4);
  w.st = o; o += up((size_t)rows * 8);
  w.pw = o; o += up((size_t)n * k * sdmi_dtype_size(dtype));
  w.sk = o; o += up(splits > 1 ? (size_t)splits * rows * n * 4 : 0);
  w.total = o;
  return w;
}
}  // namespace

int64_t sdmi_layernorm_linear_workspace_bytes(int rows, int k, int n, int splits, int dtype) {
  return (int64_t)ln_linear_ws(rows, k, n, splits, dtype).total;
}

int sdmi_layernorm_linear(const void* x, const float* w, const float* bias, const float* gamma, const float* beta,
                          float eps, void* out, int rows, int k, int n, int geglu, int splits, void* workspace,
                          int dtype, void* stream) {
  SDMI_REQUIRE(x && w && gamma && beta && out && workspace, "layernorm_linear: null pointer");
  SDMI_REQUIRE(!geglu || (bias && n % 32 == 0), "layernorm_linear: GEGLU needs a bias and n a multiple of 32");
  hipStream_t s = (hipStream_t)stream;
  const LnLinearWs ws = ln_linear_ws(rows, k, n, splits, dtype);
  char* base = (char*)workspace;
  float* fw = (float*)(base + ws.fw); float* fb = (float*)(base + ws.fb); float* pb = (float*)(base + ws.pb);
  float* cs = (float*)(base + ws.cs); float* st = (float*)(base + ws.st);
  void* pw = base + ws.pw;
  SDMI_TRY(fold_ln_weight(w, gamma, beta, bias, fw, fb, n, k, s));
  if (geglu) SDMI_TRY(pack_geglu_weight(fw, fb, pw, pb, n / 2, k, dtype, s));
  else SDMI_TRY(pack_conv_weight(fw, pw, n, k, 1, dtype, s));
  SDMI_TRY(packed_rowsum(pw, cs, n, k, dtype, s));
  SDMI_TRY(layernorm_stats(x, st, rows, k, eps, dtype, s));
  ConvGemmArgs a;
  a.a0 = x; a.c0 = k; a.w = pw; a.bias = geglu ? pb : fb; a.out = out; a.ldo = geglu ? n / 2 : n;
  a.B = 1; a.Hin = 1; a.Win = rows; a.Ho = 1; a.Wo = rows; a.taps = 1; a.N = n; a.geglu = geglu; a.dtype = dtype;
  a.splits = splits; a.partial = splits > 1 ? (float*)(base + ws.sk) : nullptr;
  a.ln_stats = st; a.ln_colsum = cs;
  return conv_gemm(a, s);
}

int sdmi_pack_weight(const float* w, void* out, int out_ch, int in_ch, int taps, int dtype, void* stream) {
  return pack_conv_weight(w, out, out_ch, in_ch, taps, dtype, (hipStream_t)stream);
}

int sdmi_pack_geglu(const float* w, const float* b, void* wout, float* bout, int inner, int in_ch, int dtype,
                    void* stream) {
  return pack_geglu_weight(w, b, wout, bout, inner, in_ch, dtype, (hipStream_t)stream);
}

int sdmi_conv2d(const void* a0, int c0, const void* a1, int c1, const void* w, const float* bias,
                const float* rowadd, const void* resid, int ldr, void* out, int ldo, int batch, int hin, int win,
                int taps, int stride, int pad_lo, int up, int n_out, int geglu, int splits, void* partial,
                int dtype, void* stream) {
  ConvGemmArgs a;
  a.a0 = a0; a.c0 = c0; a.a1 = a1; a.c1 = c1; a.w = w; a.bias = bias; a.rowadd = rowadd;
  a.resid = resid; a.ldr = ldr; a.out = out; a.ldo = ldo; a.B = batch; a.Hin = hin; a.Win = win;
  a.taps = taps; a.stride = stride; a.up = up; a.N = n_out; a.geglu = geglu; a.splits = splits;
  a.partial = (float*)partial; a.dtype = dtype;
  const int hup = up ? 2 * hin : hin, wup = up ? 2 * win : win;
  a.pad_lo = pad_lo;
  if (taps == 9) { a.Ho = (hup + pad_lo + 1 - 3) / stride + 1; a.Wo = (wup + pad_lo + 1 - 3) / stride + 1; }
  else { a.Ho = (hup - 1) / stride + 1; a.Wo = (wup - 1) / stride + 1; }
  return conv_gemm(a, (hipStream_t)stream);
}

int sdmi_conv2d_splits(int m, int n, int k, int dtype) { return conv_gemm_pick_splits(m, n, k, dtype); }

int sdmi_conv_in(const void* const* src, const int* channels, int n_src, const float* w, const float* bias,
                 float* wpack, void* out, int batch, int h, int width, int cout, int dtype, void* stream) {
  SDMI_REQUIRE(src && channels && n_src >= 1 && n_src <= 3 && w && bias && wpack && out, "conv_in: bad arguments");
  InConvArgs a;
  int cin = 0;
  for (int i = 0; i < n_src; ++i) { a.src[i] = (const float*)src[i]; a.c[i] = channels[i]; cin += channels[i]; }
  SDMI_TRY(pack_inconv_weight(w, wpack, cout, cin, (hipStream_t)stream));
  a.w = wpack; a.bias = bias; a.out = out; a.B = batch; a.H = h; a.W = width; a.Cout = cout; a.dtype = dtype;
  return in_conv(a, (hipStream_t)stream);
}

int sdmi_conv_out(const void* x, const float* w, const float* bias, float* wpack, float* out, int batch, int h,
                  int width, int cin, int cout, int dtype, void* stream) {
  SDMI_REQUIRE(x && w && bias && wpack && out, "conv_out: null pointer");
  SDMI_TRY(pack_outconv_weight(w, wpack, cout, cin, (hipStream_t)stream));
  return out_conv(x, wpack, bias, out, batch, h, width, cin, cout, dtype, (hipStream_t)stream);
}

int sdmi_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo,
                   int batch, int heads, int nq, int nk, int d, float scale, int dtype, int impl, void* stream) {
  AttentionArgs a;
  a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out = out; a.ldo = ldo;
  a.B = batch; a.heads = heads; a.Nq = nq; a.Nk = nk; a.d = d; a.scale = scale; a.dtype = dtype; a.impl = impl & 1; a.causal = (impl >> 1) & 1;
  return attention(a, (hipStream_t)stream);
}

int sdmi_timestep_embedding(const int64_t* t, float* out, int batch, int dim, float max_period, void* stream) {
  return timestep_embedding(t, out, batch, dim, max_period, (hipStream_t)stream);
}

int sdmi_small_linear(const float* x, const float* w, const float* bias, float* y, int m, int k, int n, int act_in,
                      int act_out, void* stream) {
  return small_linear(x, w, bias, y, m, k, n, act_in, act_out, (hipStream_t)stream);
}

int sdmi_convert_from_f32(const float* x, void* out, int64_t n, int dtype, void* stream) {
  return convert_f32(x, out, n, dtype, (hipStream_t)stream);
}

int sdmi_convert_to_f32(const void* x, float* out, int64_t n, int dtype, void* stream) {
  return convert_to_f32(x, out, n, dtype, (hipStream_t)stream);
}

}  // extern "C"
