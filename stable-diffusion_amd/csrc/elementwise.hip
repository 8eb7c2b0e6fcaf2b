// Small and elementwise gfx950 kernels of the denoising path: timestep embedding, tiny-M linears, the fused
// classifier-free-guidance + DDIM update, row softmax, image post-scale and weight packing.
#include "ops.h"
#include <math.h>

namespace {

inline unsigned grid_for(size_t n, int block = 256, unsigned cap = 16384) {
  size_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  return (unsigned)(g > cap ? cap : g);
}

// ---- timestep embedding ------------------------------------------------------------------------
__global__ void timestep_embedding_kernel(const int64_t* t, float* out, int B, int dim, float neg_log_period) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * half) return;
  const int b = idx / half, k = idx - b * half;
  // The exponent is formed in fp32 with the reference's operation order (scalar * k, then / half);
  // exp / cos / sin are then evaluated in double and rounded once, i.e. correctly rounded fp32.
  const float a = (neg_log_period * (float)k) / (float)half;
  const float freq = (float)exp((double)a);
  const float arg = (float)t[b] * freq;
  out[(size_t)b * dim + k] = (float)cos((double)arg);
  out[(size_t)b * dim + half + k] = (float)sin((double)arg);
  if ((dim & 1) && k == 0) out[(size_t)b * dim + dim - 1] = 0.f;
}

// ---- tiny-M linear -------------------------------------------------------------------------------
// y[m][n] = act_out(sum_k act_in(x[m][k]) W[n][k] + bias[n]) for a handful of rows (time embedding MLP,
// the per-block embedding projections).  The weight matrix is the only large operand and is read
// exactly once: a workgroup parks its MB rows of x (activation applied) in LDS, every wave streams
// NPW weight rows at a time with 16-byte loads (lane <-> k), multiplies them against all MB rows and
// folds the lanes at the end.
constexpr int SL_NPW = 4;          // weight rows per wave and pass
constexpr int SL_PASSES = 2;       // passes per wave
constexpr int SL_THREADS = 1024;   // 16 waves share one LDS copy of x: enough loads in flight per CU
constexpr int SL_WAVES = SL_THREADS / 64;

template <int MB, int NPW = SL_NPW, int PASSES = SL_PASSES>
__global__ __launch_bounds__(SL_THREADS) void small_linear_kernel(const float* x, const float* W, const float* bias,
                                                                  float* y, int M, int K, int N, int act_in,
                                                                  int act_out) {
  extern __shared__ __attribute__((aligned(16))) float sl_x[];          // [MB][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mb = blockIdx.y * MB;
  for (int m = 0; m < MB; ++m)
    for (int k = tid; k < K; k += SL_THREADS) {
      float v = mb + m < M ? x[(size_t)(mb + m) * K + k] : 0.f;
      if (act_in == 1) v = silu_f(v);
      sl_x[m * K + k] = v;
    }
  __syncthreads();
  for (int pass = 0; pass < PASSES; ++pass) {
    const int n0 = ((blockIdx.x * PASSES + pass) * SL_WAVES + wave) * NPW;
    if (n0 >= N) return;
    float acc[NPW][MB];
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[j][m] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      float4 w[NPW];
#pragma unroll
      for (int j = 0; j < NPW; ++j)
        w[j] = *reinterpret_cast<const float4*>(W + (size_t)min(n0 + j, N - 1) * K + k);
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const float4 xv = *reinterpret_cast<const float4*>(sl_x + m * K + k);
#pragma unroll
        for (int j = 0; j < NPW; ++j)
          acc[j][m] += xv.x * w[j].x + xv.y * w[j].y + xv.z * w[j].z + xv.w * w[j].w;
      }
    }
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const float sum = wave_sum(acc[j][m]);
        if (lane == 0 && n0 + j < N && mb + m < M) {
          float v = sum + (bias ? bias[n0 + j] : 0.f);
          if (act_out == 1) v = silu_f(v);
          y[(size_t)(mb + m) * N + n0 + j] = v;
        }
      }
  }
}

__global__ void pointwise_nchw_kernel(const float* x, const float* w, const float* bias, float* out, int B,
                                      int Cin, int Cout, int HW, float scale) {
  const size_t total = (size_t)B * Cout * HW;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int px = (int)(idx % HW), co = (int)((idx / HW) % Cout), b = (int)(idx / ((size_t)HW * Cout));
    float acc = bias[co];
    for (int ci = 0; ci < Cin; ++ci) acc += w[co * Cin + ci] * (scale * x[((size_t)b * Cin + ci) * HW + px]);
    out[idx] = acc;
  }
}

// ---- posterior sample (DiagonalGaussianDistribution :624-638, get_first_stage_encoding :4264-4271) ----
// moments: NCHW fp32 [B][2C][HW] (mean | logvar); z = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise);
// noise == nullptr gives the mode (mean).  Every product and sum is rounded separately, as the
// reference's elementwise torch ops are.
__global__ void gaussian_sample_kernel(const float* moments, const float* noise, float scale, float* z, int B, int C,
                                       int HW) {
#pragma clang fp contract(off)
  const size_t total = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / ((size_t)C * HW), r = i - b * (size_t)C * HW;
    const float mean = moments[b * 2 * C * HW + r];
    float v = mean;
    if (noise) {
      float lv = moments[b * 2 * C * HW + (size_t)C * HW + r];
      lv = fminf(fmaxf(lv, -30.0f), 20.0f);
      const float sd = expf(0.5f * lv);
      v = mean + sd * noise[i];
    }
    z[i] = scale * v;
  }
}

// ---- sampler -----------------------------------------------------------------------------------
__global__ void ddim_step_kernel(const float* x, const float* eps, int cfg, float scale, float sqrt_at,
                                 float sqrt_aprev, float dir_coef, float sigma, float sqrt_1m_at,
                                 const float* noise, float temperature, float* x_prev, float* pred_x0, int64_t n) {
  // every product and sum is rounded separately, as the reference's elementwise torch ops are
#pragma clang fp contract(off)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float e;
    if (cfg) {
      const float eu = eps[i], ec = eps[n + i];
      e = eu + scale * (ec - eu);
    } else {
      e = eps[i];
    }
    const float xv = x[i];
    // same operation order as the reference: (x - s*e) / sqrt(a_t); sqrt(a_prev)*p0 + dir*e + noise
    const float p0 = (xv - sqrt_1m_at * e) / sqrt_at;
    const float nz = noise ? sigma * noise[i] * temperature : 0.f;
    const float xp = sqrt_aprev * p0 + dir_coef * e + nz;
    if (pred_x0) pred_x0[i] = p0;
    x_prev[i] = xp;
  }
}

// classifier-free-guidance combine on its own (e_uncond + scale * (e_cond - e_uncond), :5347), for callers that
// put a score corrector between the combine and the update
__global__ void cfg_combine_kernel(const float* eps, float scale, float* out, int64_t n) {
#pragma clang fp contract(off)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float eu = eps[i], ec = eps[n + i];
    out[i] = eu + scale * (ec - eu);
  }
}

// DDPM ancestral update: x0 from eps, the posterior mean of q(x_{t-1} | x_t, x0) and the noise term.
// The per-timestep table entries arrive as scalars (the sampling loops use one t for the whole batch).
__global__ void ddpm_step_kernel(const float* x, const float* eps, float c_recip, float c_recipm1, int clip,
                                 float coef1, float coef2, float noise_coef, const float* noise,
                                 float temperature, float* x_prev, float* pred_x0, int64_t n) {
#pragma clang fp contract(off)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float xv = x[i];
    float p0 = c_recip * xv - c_recipm1 * eps[i];
    if (clip) p0 = fminf(fmaxf(p0, -1.0f), 1.0f);
    const float mean = coef1 * p0 + coef2 * xv;
    const float nz = noise ? noise_coef * (noise[i] * temperature) : 0.f;
    if (pred_x0) pred_x0[i] = p0;
    x_prev[i] = mean + nz;
  }
}

__global__ void axpby_kernel(const float* a, const float* b, float ca, float cb, float* out, int64_t n) {
#pragma clang fp contract(off)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = ca * a[i] + cb * b[i];
}

// known-region blend of the DDIM loop: the known latent x0 is noised to the step's timestep (q_sample) and
// written over the region mask marks; mask has 1 or `C` channels (broadcast over channels like the torch op)
__global__ void q_sample_blend_kernel(const float* x0, const float* noise, const float* mask, const float* img,
                                      float ca, float cb, float* out, int B, int C, int HW, int MC) {
  // every product and sum is rounded separately, as the reference's elementwise torch ops are
#pragma clang fp contract(off)
  const size_t total = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / ((size_t)C * HW), r = i - b * (size_t)C * HW;
    const size_t c = r / HW, px = r - c * HW;
    const float m = mask[(b * MC + (MC == 1 ? 0 : c)) * HW + px];
    const float q = ca * x0[i] + cb * noise[i];
    out[i] = q * m + (1.0f - m) * img[i];
  }
}

__global__ void to_image_kernel(const float* x, float* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = fminf(fmaxf((x[i] + 1.0f) / 2.0f, 0.f), 1.f);
}

// ---- row softmax (one block per row) ------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, T* p, int cols, float scale) {
  __shared__ float red[4];
  const float* row = s + (size_t)blockIdx.x * cols;
  T* orow = p + (size_t)blockIdx.x * cols;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, row[c] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wid] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float l = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) l += __expf(row[c] * scale - m);
  l = wave_sum(l);
  if (lane == 0) red[wid] = l;
  __syncthreads();
  l = red[0] + red[1] + red[2] + red[3];
  const float inv = 1.0f / l;
  for (int c = threadIdx.x; c < cols; c += 256) orow[c] = from_f32<T>(__expf(row[c] * scale - m) * inv);
}

// ---- packing --------------------------------------------------------------------------------
template <typename T>
__global__ void pack_conv_weight_kernel(const float* w, T* out, int O, int I, int taps) {
  const size_t total = (size_t)O * I * taps;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % I), tap = (int)((idx / I) % taps), o = (int)(idx / ((size_t)I * taps));
    out[idx] = from_f32<T>(w[((size_t)o * I + i) * taps + tap]);
  }
}

// packed row p: group = p/32, within = p%32; within<16 -> value row group*16+within, else gate row
template <typename T>
__global__ void pack_geglu_weight_kernel(const float* w, const float* b, T* wout, float* bout, int F, int I) {
  const size_t total = (size_t)2 * F * I;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % I), pr = (int)(idx / I);
    const int grp = pr >> 5, within = pr & 31;
    const int src = within < 16 ? grp * 16 + within : F + grp * 16 + (within - 16);
    wout[idx] = from_f32<T>(w[(size_t)src * I + i]);
    if (i == 0) bout[pr] = b[src];
  }
}

// LayerNorm folded into the Linear behind it: y = LN(x) W^T + b = rstd (x W'^T - mean c) + b' with
// W'[n][k] = gamma[k] W[n][k], c[n] = sum_k W'[n][k] (over the values as the GEMM will see them, see
// packed_rowsum_kernel) and b'[n] = b[n] + sum_k beta[k] W[n][k].  One wave per output row, lanes fold in a
// fixed order.
__global__ __launch_bounds__(256) void fold_ln_weight_kernel(const float* w, const float* gamma, const float* beta,
                                                             const float* bias, float* w_out, float* b_out, int N,
                                                             int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float wv = w[(size_t)n * K + k];
    w_out[(size_t)n * K + k] = wv * gamma[k];
    acc += beta[k] * wv;
  }
  acc = wave_sum(acc);
  if (lane == 0) b_out[n] = acc + (bias ? bias[n] : 0.f);
}

// c[n] = sum_k Wp[n][k] over a packed [N][K] matrix in its stored precision
template <typename T>
__global__ __launch_bounds__(256) void packed_rowsum_kernel(const T* wp, float* c, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += to_f32<T>(wp[(size_t)n * K + k]);
  acc = wave_sum(acc);
  if (lane == 0) c[n] = acc;
}

__global__ void pack_inconv_weight_kernel(const float* w, float* out, int O, int I) {
  const size_t total = (size_t)O * I * 9;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int o = (int)(idx % O), i = (int)((idx / O) % I), tap = (int)(idx / ((size_t)O * I));
    out[idx] = w[((size_t)o * I + i) * 9 + tap];
  }
}

// token + position embedding lookup: x[b][t][:] = tok[ids[b][t]][:] + pos[t][:]   (fp32 tables -> T)
template <typename T>
__global__ void embed_tokens_kernel(const int64_t* ids, const float* tok, const float* pos, T* out, int rows,
                                    int T_len, int d, int vocab) {
  const int dv = d / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * dv;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / dv), c = (int)(i - (int64_t)row * dv) * 4;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);       // the host wrapper rejects out-of-range ids
    const float4 a = *reinterpret_cast<const float4*>(tok + (size_t)id * d + c);
    const float4 b = *reinterpret_cast<const float4*>(pos + (size_t)(row % T_len) * d + c);
    T* o = out + (size_t)row * d + c;
    o[0] = from_f32<T>(a.x + b.x); o[1] = from_f32<T>(a.y + b.y);
    o[2] = from_f32<T>(a.z + b.z); o[3] = from_f32<T>(a.w + b.w);
  }
}

// x * sigmoid(1.702 x), in place
template <typename T>
__global__ void quick_gelu_kernel(T* x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = to_f32<T>(x[i]);
    x[i] = from_f32<T>(v / (1.0f + __expf(-1.702f * v)));
  }
}

template <typename T>
__global__ void convert_f32_kernel(const float* x, T* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = from_f32<T>(x[i]);
}
template <typename T>
__global__ void convert_to_f32_kernel(const T* x, float* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = to_f32<T>(x[i]);
}

}  // namespace

int timestep_embedding(const int64_t* t, float* out, int B, int dim, float max_period, hipStream_t s) {
  SDMI_REQUIRE(t && out && B > 0 && dim >= 2, "timestep_embedding: bad arguments");
  const int n = B * (dim / 2);
  const float neg_log_period = (float)(-log((double)max_period));
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, out, B, dim,
                     neg_log_period);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

constexpr size_t SMALL_LINEAR_MAX_LDS = 144 * 1024;
int elementwise_init();

int small_linear(const float* x, const float* W, const float* bias, float* y, int M, int K, int N, int act_in,
                 int act_out, hipStream_t s) {
  SDMI_REQUIRE(x && W && y && M > 0 && K > 0 && N > 0, "small_linear: bad arguments");
  SDMI_REQUIRE(K % 4 == 0, "small_linear: K=%d must be a multiple of 4", K);
  SDMI_TRY(elementwise_init());
  const int per_block = SL_WAVES * SL_NPW * SL_PASSES;
  const dim3 block(SL_THREADS);
  // few output columns (the time-embedding MLP: 1280): one weight row per wave and one pass, or 10 workgroups would
  // stream the whole matrix
  if (M <= 2 && N <= 4096) {
    SDMI_REQUIRE((size_t)2 * K * 4 <= SMALL_LINEAR_MAX_LDS, "small_linear: K=%d too large", K);
    if (M == 1)
      hipLaunchKernelGGL((small_linear_kernel<1, 1, 1>), dim3((N + SL_WAVES - 1) / SL_WAVES, 1), block, (size_t)K * 4, s,
                         x, W, bias, y, M, K, N, act_in, act_out);
    else
      hipLaunchKernelGGL((small_linear_kernel<2, 1, 1>), dim3((N + SL_WAVES - 1) / SL_WAVES, 1), block, (size_t)2 * K * 4,
                         s, x, W, bias, y, M, K, N, act_in, act_out);
    SDMI_CHECK_HIP(hipGetLastError());
    return SDMI_OK;
  }
#define SL_LAUNCH(MB_)                                                                                    \
  {                                                                                                       \
    SDMI_REQUIRE((size_t)(MB_) * K * 4 <= SMALL_LINEAR_MAX_LDS, "small_linear: K=%d too large", K);     \
    hipLaunchKernelGGL(small_linear_kernel<MB_>, dim3((N + per_block - 1) / per_block, (M + (MB_) - 1) / (MB_)), \
                       block, (size_t)(MB_) * K * 4, s, x, W, bias, y, M, K, N, act_in, act_out);       \
  }
  if (M == 1) SL_LAUNCH(1)
  else if (M <= 2) SL_LAUNCH(2)
  else if (M <= 8) SL_LAUNCH(8)
  else SL_LAUNCH(16)
#undef SL_LAUNCH
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int boundary_conv_init();

// Raise the dynamic-LDS limits of the kernels in this file and of the boundary convolutions once, outside of
// stream capture.
int elementwise_init() {
  static bool done = false;
  if (done) return SDMI_OK;
  SDMI_TRY(boundary_conv_init());
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)small_linear_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)SMALL_LINEAR_MAX_LDS));
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)small_linear_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)SMALL_LINEAR_MAX_LDS));
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)small_linear_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)SMALL_LINEAR_MAX_LDS));
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)small_linear_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)SMALL_LINEAR_MAX_LDS));
  // the narrow-output instantiations park up to two rows of x (K beyond 8192 exceeds the default 64 KiB)
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)small_linear_kernel<1, 1, 1>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LINEAR_MAX_LDS));
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)small_linear_kernel<2, 1, 1>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LINEAR_MAX_LDS));
  done = true;
  return SDMI_OK;
}

int gaussian_sample(const float* moments, const float* noise, float scale, float* z, int B, int C, int HW,
                    hipStream_t s) {
  SDMI_REQUIRE(moments && z && B > 0 && C > 0 && HW > 0, "gaussian_sample: bad arguments");
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3(grid_for((size_t)B * C * HW)), dim3(256), 0, s, moments, noise,
                     scale, z, B, C, HW);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int pointwise_nchw(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout,
                   int HW, float scale, hipStream_t s) {
  SDMI_REQUIRE(x && w && bias && out, "pointwise_nchw: null pointer");
  const size_t total = (size_t)B * Cout * HW;
  hipLaunchKernelGGL(pointwise_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, w, bias, out, B, Cin, Cout,
                     HW, scale);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int ddim_step(const float* x, const float* eps, int cfg, float scale, float a_t, float a_prev, float sigma,
              float sqrt_1m_at, const float* noise, float temperature, float* x_prev, float* pred_x0, int64_t n,
              hipStream_t s) {
  SDMI_REQUIRE(x && eps && x_prev && n > 0, "ddim_step: bad arguments");
  // scalar prefactors in fp32, exactly as the reference's fp32 tensors: sqrt(a_t), sqrt(a_prev),
  // sqrt(1 - a_prev - sigma^2)
  const float sqrt_at = sqrtf(a_t);
  const float sqrt_aprev = sqrtf(a_prev);
  const float dir_coef = sqrtf(1.0f - a_prev - sigma * sigma);
  hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for((size_t)n, 256, 2048)), dim3(256), 0, s, x, eps, cfg, scale,
                     sqrt_at, sqrt_aprev, dir_coef, sigma, sqrt_1m_at, noise, temperature, x_prev, pred_x0, n);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int ddpm_step(const float* x, const float* eps, float c_recip, float c_recipm1, int clip, float coef1, float coef2,
              float noise_coef, const float* noise, float temperature, float* x_prev, float* pred_x0, int64_t n,
              hipStream_t s) {
  SDMI_REQUIRE(x && eps && x_prev && n > 0, "ddpm_step: bad arguments");
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for((size_t)n, 256, 2048)), dim3(256), 0, s, x, eps, c_recip,
                     c_recipm1, clip, coef1, coef2, noise_coef, noise, temperature, x_prev, pred_x0, n);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int q_sample_blend(const float* x0, const float* noise, const float* mask, const float* img, float ca, float cb,
                   float* out, int B, int C, int HW, int mask_channels, hipStream_t s) {
  SDMI_REQUIRE(x0 && noise && mask && img && out && B > 0 && C > 0 && HW > 0, "q_sample_blend: bad arguments");
  SDMI_REQUIRE(mask_channels == 1 || mask_channels == C, "q_sample_blend: mask has %d channels, expected 1 or %d",
               mask_channels, C);
  hipLaunchKernelGGL(q_sample_blend_kernel, dim3(grid_for((size_t)B * C * HW, 256, 2048)), dim3(256), 0, s, x0, noise,
                     mask, img, ca, cb, out, B, C, HW, mask_channels);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int cfg_combine(const float* eps, float scale, float* out, int64_t n, hipStream_t s) {
  SDMI_REQUIRE(eps && out && n > 0, "cfg_combine: bad arguments");
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for((size_t)n, 256, 2048)), dim3(256), 0, s, eps, scale, out, n);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int axpby(const float* a, const float* b, float ca, float cb, float* out, int64_t n, hipStream_t s) {
  SDMI_REQUIRE(a && b && out && n > 0, "axpby: bad arguments");
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for((size_t)n, 256, 2048)), dim3(256), 0, s, a, b, ca, cb, out, n);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

// uint8 HWC image as the scripts save it (txt2img.py:307, :316-317): trunc(255 * clamp((x + 1) / 2, 0, 1)),
// NCHW fp32 in, NHWC uint8 out; each product rounded to fp32 as numpy does.
__global__ void to_image_u8_kernel(const float* x, unsigned char* out, int B, int C, int HW) {
#pragma clang fp contract(off)
  const size_t total = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t px = (i / C) % HW, b = i / ((size_t)C * HW);
    float v = (x[(b * C + c) * HW + px] + 1.0f) / 2.0f;
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    out[i] = (unsigned char)(255.0f * v);
  }
}

// inpainted = (1 - mask) * clamp((image + 1) / 2, 0, 1) + mask * predicted  (inpaint_with_text.py:286-291);
// image, predicted: NCHW fp32 [B][C][HW] (predicted already in [0,1]); mask: [B][1][HW]
__global__ void inpaint_composite_kernel(const float* image, const float* mask, const float* pred, float* out, int B,
                                         int C, int HW) {
#pragma clang fp contract(off)
  const size_t total = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i % HW, b = i / ((size_t)C * HW);
    const float m = mask[b * HW + px];
    float im = (image[i] + 1.0f) / 2.0f;
    im = fminf(fmaxf(im, 0.0f), 1.0f);
    out[i] = (1.0f - m) * im + m * pred[i];
  }
}

int to_image_u8(const float* x, unsigned char* out, int B, int C, int HW, hipStream_t s) {
  SDMI_REQUIRE(x && out && B > 0 && C > 0 && HW > 0, "to_image_u8: bad arguments");
  hipLaunchKernelGGL(to_image_u8_kernel, dim3(grid_for((size_t)B * C * HW, 256, 4096)), dim3(256), 0, s, x, out, B, C,
                     HW);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int inpaint_composite(const float* image, const float* mask, const float* pred, float* out, int B, int C, int HW,
                      hipStream_t s) {
  SDMI_REQUIRE(image && mask && pred && out && B > 0 && C > 0 && HW > 0, "inpaint_composite: bad arguments");
  hipLaunchKernelGGL(inpaint_composite_kernel, dim3(grid_for((size_t)B * C * HW, 256, 4096)), dim3(256), 0, s, image,
                     mask, pred, out, B, C, HW);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int to_image(const float* x, float* out, int64_t n, hipStream_t s) {
  SDMI_REQUIRE(x && out && n > 0, "to_image: bad arguments");
  hipLaunchKernelGGL(to_image_kernel, dim3(grid_for((size_t)n, 256, 4096)), dim3(256), 0, s, x, out, n);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int softmax_rows(const float* sc, void* p, int rows, int cols, float scale, int dtype, hipStream_t st) {
  SDMI_REQUIRE(sc && p && rows > 0 && cols > 0, "softmax_rows: bad arguments");
  SDMI_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL(softmax_rows_kernel<T>, dim3(rows), dim3(256), 0, st, sc, (T*)p, cols, scale));
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int pack_conv_weight(const float* w, void* out, int O, int I, int taps, int dtype, hipStream_t s) {
  SDMI_REQUIRE(w && out, "pack_conv_weight: null pointer");
  const size_t total = (size_t)O * I * taps;
  SDMI_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL(pack_conv_weight_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, w,
                                         (T*)out, O, I, taps));
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int fold_ln_weight(const float* w, const float* gamma, const float* beta, const float* bias, float* w_out,
                   float* b_out, int N, int K, hipStream_t s) {
  SDMI_REQUIRE(w && gamma && beta && w_out && b_out && N > 0 && K > 0, "fold_ln_weight: bad arguments");
  hipLaunchKernelGGL(fold_ln_weight_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w, gamma, beta, bias, w_out, b_out, N, K);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
int packed_rowsum(const void* wp, float* c, int N, int K, int dtype, hipStream_t s) {
  SDMI_REQUIRE(wp && c && N > 0 && K > 0, "packed_rowsum: bad arguments");
  SDMI_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(packed_rowsum_kernel<T>, dim3((N + 3) / 4), dim3(256), 0, s,
                                                   (const T*)wp, c, N, K));
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
int pack_geglu_weight(const float* w, const float* b, void* wout, float* bout, int F, int I, int dtype,
                      hipStream_t s) {
  SDMI_REQUIRE(w && b && wout && bout && F % 16 == 0, "pack_geglu_weight: bad arguments");
  const size_t total = (size_t)2 * F * I;
  SDMI_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL(pack_geglu_weight_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, w, b,
                                         (T*)wout, bout, F, I));
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int pack_inconv_weight(const float* w, float* out, int O, int I, hipStream_t s) {
  SDMI_REQUIRE(w && out, "pack_inconv_weight: null pointer");
  hipLaunchKernelGGL(pack_inconv_weight_kernel, dim3(grid_for((size_t)O * I * 9)), dim3(256), 0, s, w, out, O, I);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int pack_outconv_weight(const float* w, float* out, int O, int I, hipStream_t s) {
  SDMI_REQUIRE(w && out, "pack_outconv_weight: null pointer");
  hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(grid_for((size_t)O * I * 9)), dim3(256), 0, s, w, out, O,
                     I, 9);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int embed_tokens(const int64_t* ids, const float* tok, const float* pos, void* out, int rows, int T_len, int d,
                 int vocab, int dtype, hipStream_t s) {
  SDMI_REQUIRE(ids && tok && pos && out && rows > 0 && T_len > 0 && d % 4 == 0 && vocab > 0, "embed_tokens: bad arguments");
  SDMI_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL(embed_tokens_kernel<T>, dim3(grid_for((size_t)rows * (d / 4))), dim3(256), 0,
                                         s, ids, tok, pos, (T*)out, rows, T_len, d, vocab));
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int quick_gelu(void* x, int64_t n, int dtype, hipStream_t s) {
  SDMI_REQUIRE(x && n > 0, "quick_gelu: bad arguments");
  SDMI_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL(quick_gelu_kernel<T>, dim3(grid_for((size_t)n)), dim3(256), 0, s, (T*)x, n));
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int convert_f32(const float* x, void* out, int64_t n, int dtype, hipStream_t s) {
  SDMI_REQUIRE(x && out && n > 0, "convert_f32: bad arguments");
  SDMI_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL(convert_f32_kernel<T>, dim3(grid_for((size_t)n)), dim3(256), 0, s, x,
                                         (T*)out, n));
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int convert_to_f32(const void* x, float* out, int64_t n, int dtype, hipStream_t s) {
  SDMI_REQUIRE(x && out && n > 0, "convert_to_f32: bad arguments");
  SDMI_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL(convert_to_f32_kernel<T>, dim3(grid_for((size_t)n)), dim3(256), 0, s,
                                         (const T*)x, out, n));
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
