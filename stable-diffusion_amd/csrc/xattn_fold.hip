// Cross-attention against a fixed text context, folded once per prompt (CrossAttention.forward,
// stable_diffusion.py:2776-2797, with context = the prompt's hidden states).
//
// For one head h of one sample b, with K = ctx Wk^T and V = ctx Wv^T already projected (the per-prompt cache):
//
//   scores[m][k] = scale * sum_e q[m][(h,e)] K[k][(h,e)],  q = LayerNorm(x) Wq^T
//               = rstd_m (x_m . G[k] - mean_m c[k]) + g0[k]
//       G[k][j]  = c0 sum_e K[k][(h,e)] (Wq diag gamma)[(h,e)][j]        c0 = scale log2(e)
//       c[k]     = sum_j G[k][j]
//       g0[k]    = c0 sum_e K[k][(h,e)] (Wq beta)[(h,e)]
//   out[m][n]   = sum_h sum_k P_h[m][k] U[n][(h,k)] + bias[n] + x[m][n]
//       U[n][(h,k)] = sum_e V[k][(h,e)] Wo[n][(h,e)]
//
// so the query projection, the attention and the output projection of the block become two GEMMs with
// per-sample weights G ([heads*80][d], keys padded to 80 per head) and U ([d][heads*80]) and a softmax over
// 80-column segments between them.  The kernels below build G, g0 and U; c comes from packed_rowsum on the
// rounded G.  Plain fp32 vector arithmetic: this runs once per prompt, not per step.
#include "ops.h"

namespace {

constexpr int XSEG = 80;            // keys per head segment (text context of at most 80 tokens)

// G and g0.  grid (B * heads, ceil(d / 256)), 256 threads: thread j owns output column j of the 80 rows of (b, h).
// kv: [B * ntok][2 * inner] (K in columns 0 .. inner-1); wq: packed [inner][d] (rows (h, e)), already carrying gamma;
// qb: [inner] = Wq beta.
template <typename T>
__global__ __launch_bounds__(256) void xattn_fold_g_kernel(const T* kv, const T* wq, const float* qb, T* G, float* g0,
                                                           int ntok, int heads, int dh, int d, float c0) {
  extern __shared__ __attribute__((aligned(16))) float ks[];      // [80][dh], scaled by c0, rows beyond ntok zero
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int inner = heads * dh;
  for (int i = threadIdx.x; i < XSEG * dh; i += blockDim.x) {
    const int k = i / dh, e = i - k * dh;
    ks[i] = k < ntok ? c0 * to_f32(kv[((size_t)b * ntok + k) * 2 * inner + h * dh + e]) : 0.f;
  }
  __syncthreads();
  if (blockIdx.y == 0 && threadIdx.x < XSEG) {
    const int k = threadIdx.x;
    float s = 0.f;
    for (int e = 0; e < dh; ++e) s = fmaf(ks[k * dh + e], qb[h * dh + e], s);
    // columns beyond the context: a bias that no real score comes near, so their probability is exactly zero
    g0[((size_t)b * heads + h) * XSEG + k] = k < ntok ? s : -30000.f;
  }
  const int j = blockIdx.y * 256 + threadIdx.x;
  if (j >= d) return;
  float acc[XSEG];
#pragma unroll
  for (int k = 0; k < XSEG; ++k) acc[k] = 0.f;
  for (int e = 0; e < dh; e += 4) {
    const T* wrow = wq + (size_t)(h * dh + e) * d + j;
    const float w0 = to_f32(wrow[0]), w1 = to_f32(wrow[d]), w2 = to_f32(wrow[2 * (size_t)d]), w3 = to_f32(wrow[3 * (size_t)d]);
#pragma unroll
    for (int k = 0; k < XSEG; ++k) {
      const float4 kk = *reinterpret_cast<const float4*>(ks + k * dh + e);
      acc[k] = fmaf(kk.x, w0, fmaf(kk.y, w1, fmaf(kk.z, w2, fmaf(kk.w, w3, acc[k]))));
    }
  }
  T* out = G + ((size_t)(b * heads + h) * XSEG) * d + j;
#pragma unroll
  for (int k = 0; k < XSEG; ++k) out[(size_t)k * d] = from_f32<T>(acc[k]);
}

// U.  grid (B * heads, ceil(d / 256)), 256 threads: thread n owns the 80 entries (h, 0..79) of row n of U[b].
// kv: V in columns inner .. 2 inner - 1; wo: packed [d][inner] (columns (h, e)).
template <typename T>
__global__ __launch_bounds__(256) void xattn_fold_u_kernel(const T* kv, const T* wo, T* U, int ntok, int heads, int dh,
                                                           int d) {
  extern __shared__ __attribute__((aligned(16))) float vs[];      // [80][dh], rows beyond ntok zero
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int inner = heads * dh;
  for (int i = threadIdx.x; i < XSEG * dh; i += blockDim.x) {
    const int k = i / dh, e = i - k * dh;
    vs[i] = k < ntok ? to_f32(kv[((size_t)b * ntok + k) * 2 * inner + inner + h * dh + e]) : 0.f;
  }
  __syncthreads();
  const int n = blockIdx.y * 256 + threadIdx.x;
  if (n >= d) return;
  float acc[XSEG];
#pragma unroll
  for (int k = 0; k < XSEG; ++k) acc[k] = 0.f;
  typedef typename VecOf<T>::type vec_t;
  for (int e = 0; e < dh; e += 8) {
    const vec_t wv = *reinterpret_cast<const vec_t*>(wo + (size_t)n * inner + h * dh + e);
    float w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = to_f32(wv[i]);
#pragma unroll
    for (int k = 0; k < XSEG; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(vs + k * dh + e);
      const float4 c = *reinterpret_cast<const float4*>(vs + k * dh + e + 4);
      float s = acc[k];
      s = fmaf(a.x, w[0], s); s = fmaf(a.y, w[1], s); s = fmaf(a.z, w[2], s); s = fmaf(a.w, w[3], s);
      s = fmaf(c.x, w[4], s); s = fmaf(c.y, w[5], s); s = fmaf(c.z, w[6], s); s = fmaf(c.w, w[7], s);
      acc[k] = s;
    }
  }
  T* out = U + ((size_t)b * d + n) * (heads * XSEG) + h * XSEG;
#pragma unroll
  for (int k0 = 0; k0 < XSEG; k0 += 8) {
    vec_t o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = from_f32<T>(acc[k0 + i]);
    *reinterpret_cast<vec_t*>(out + k0) = o;
  }
}

template <typename T>
int launch_fold(const void* kv, const void* wq, const float* qb, const void* wo, void* G, float* g0, void* U, int B,
                int ntok, int heads, int dh, int d, float c0, hipStream_t s) {
  const size_t lds = (size_t)XSEG * dh * sizeof(float);
  SDMI_REQUIRE(lds <= 64 * 1024, "xattn_fold: head size %d too large", dh);
  const dim3 grid(B * heads, (d + 255) / 256);
  hipLaunchKernelGGL(xattn_fold_g_kernel<T>, grid, dim3(256), lds, s, (const T*)kv, (const T*)wq, qb, (T*)G, g0, ntok,
                     heads, dh, d, c0);
  hipLaunchKernelGGL(xattn_fold_u_kernel<T>, grid, dim3(256), lds, s, (const T*)kv, (const T*)wo, (T*)U, ntok, heads, dh,
                     d);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

}  // namespace

int xattn_fold(const void* kv, const void* wq, const float* qb, const void* wo, void* G, float* g0, void* U, int B,
               int ntok, int heads, int dh, int d, float c0, int dtype, hipStream_t s) {
  SDMI_REQUIRE(kv && wq && qb && wo && G && g0 && U, "xattn_fold: null pointer");
  SDMI_REQUIRE(ntok > 0 && ntok <= XSEG && dh % 8 == 0 && d == heads * dh && d % 8 == 0,
               "xattn_fold: needs at most %d context tokens and d = heads * dh (got ntok %d, heads %d, dh %d, d %d)",
               XSEG, ntok, heads, dh, d);
  if (dtype == SDMI_F16) return launch_fold<half_t>(kv, wq, qb, wo, G, g0, U, B, ntok, heads, dh, d, c0, s);
  if (dtype == SDMI_BF16) return launch_fold<bf16_t>(kv, wq, qb, wo, G, g0, U, B, ntok, heads, dh, d, c0, s);
  sdmi_set_error("xattn_fold: 16-bit compute types only");
  return SDMI_EINVAL;
}
