// GroupNorm(+SiLU) and LayerNorm for NHWC / token-major activations on gfx950.  HBM-bound kernels:
// 16-byte vector loads/stores, fp32 statistics, double-precision combination of partial sums.
//
// GroupNorm runs as three launches:
//   1. gn_partial : grid (chunks, B); every thread owns one 16-byte channel vector and walks a strip
//                   of pixels, keeping per-channel sum / sum-of-squares in registers; the block folds
//                   them into per-group partial sums in LDS and writes [B][chunk][G][2] floats.
//   2. gn_finalize: one thread per (b, group): combines the chunk partials in double -> mean, rstd.
//   3. gn_apply   : elementwise normalise * gamma + beta (+ SiLU), 16 bytes per thread.
// The input may be the channel-concatenation of two tensors (UNet skip connections); groups may
// straddle the seam, which is why statistics are accumulated per channel first.
#include "ops.h"

namespace {

constexpr int GN_MAX_CHUNKS = 256;

template <typename T>
__device__ __forceinline__ void load_vec(const T* p, float (&f)[VecOf<T>::N]) {
  typedef typename VecOf<T>::type V;
  const V v = *reinterpret_cast<const V*>(p);
#pragma unroll
  for (int e = 0; e < VecOf<T>::N; ++e) f[e] = to_f32<T>(v[e]);
}
template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float (&f)[VecOf<T>::N]) {
  typedef typename VecOf<T>::type V;
  V v;
#pragma unroll
  for (int e = 0; e < VecOf<T>::N; ++e) v[e] = from_f32<T>(f[e]);
  *reinterpret_cast<V*>(p) = v;
}

struct GnParams {
  const void* x0; const void* x1; int c0, c1, C;
  int B, HW, groups, cpg;
  int chunks, pix_per_chunk;
  float eps;
  const float* gamma; const float* beta;
  int silu;
  void* out;
  float* partial;   // [B][chunks][G][2]
  float* stats;     // [B][G][2]  (mean, rstd)
};

constexpr int GN_LDS_SLOTS = 1024;   // >= max(blockDim, C / elements-per-vector)

template <typename T>
__global__ void gn_partial_kernel(GnParams p) {
  constexpr int EPV = VecOf<T>::N;
  // per-(pixel lane, vector) partial sums: [0]=sum,[1]=sumsq of the vector's first group, [2],[3] of
  // its second group (a 16-byte vector spans at most two groups)
  __shared__ float ls[4][GN_LDS_SLOTS];
  const int nvec = p.C / EPV;
  const int chunk = blockIdx.x, b = blockIdx.y;
  // blockDim is nvec * lanes_pix when the vectors of one pixel fit a block (every thread owns one
  // vector and every lanes_pix-th pixel); otherwise a thread strides over vectors and takes all pixels.
  const bool fits = (int)blockDim.x >= nvec;
  const int lanes_pix = fits ? (int)blockDim.x / nvec : 1;
  const int pix_lane = fits ? (int)threadIdx.x / nvec : 0;
  const int vstride = fits ? nvec : (int)blockDim.x;
  const int p_begin = chunk * p.pix_per_chunk;
  const int p_end = min(p.HW, p_begin + p.pix_per_chunk);
  for (int v = fits ? (int)threadIdx.x % nvec : (int)threadIdx.x; v < nvec; v += vstride) {
    const int c = v * EPV;
    const T* src; int cs, cc;
    if (c < p.c0) { src = (const T*)p.x0; cs = p.c0; cc = c; }
    else { src = (const T*)p.x1; cs = p.c1; cc = c - p.c0; }
    float s[EPV], q[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) { s[e] = 0.f; q[e] = 0.f; }
    for (int px = p_begin + pix_lane; px < p_end; px += lanes_pix) {
      float f[EPV];
      load_vec<T>(src + ((size_t)b * p.HW + px) * cs + cc, f);
#pragma unroll
      for (int e = 0; e < EPV; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
    }
    const int boundary = (c / p.cpg + 1) * p.cpg;
    float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      if (c + e < boundary) { sa += s[e]; qa += q[e]; } else { sb += s[e]; qb += q[e]; }
    }
    const int slot = pix_lane * nvec + v;
    ls[0][slot] = sa; ls[1][slot] = qa; ls[2][slot] = sb; ls[3][slot] = qb;
  }
  __syncthreads();
  // fixed-order fold per group: results do not depend on thread timing
  if ((int)threadIdx.x < p.groups) {
    const int g = threadIdx.x;
    const int v_lo = (g * p.cpg) / EPV, v_hi = ((g + 1) * p.cpg - 1) / EPV;
    float s = 0.f, q = 0.f;
    for (int pl = 0; pl < lanes_pix; ++pl)
      for (int v = v_lo; v <= v_hi; ++v) {
        const int slot = pl * nvec + v;
        const int g0 = (v * EPV) / p.cpg;
        if (g0 == g) { s += ls[0][slot]; q += ls[1][slot]; }
        else { s += ls[2][slot]; q += ls[3][slot]; }
      }
    float* dst = p.partial + (((size_t)b * p.chunks + chunk) * p.groups + g) * 2;
    dst[0] = s;
    dst[1] = q;
  }
}

__global__ void gn_finalize_kernel(GnParams p) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.B * p.groups) return;
  const int b = idx / p.groups, g = idx - b * p.groups;
  double s = 0.0, q = 0.0;
  for (int c = 0; c < p.chunks; ++c) {
    const float* src = p.partial + (((size_t)b * p.chunks + c) * p.groups + g) * 2;
    s += (double)src[0]; q += (double)src[1];
  }
  const double n = (double)p.cpg * (double)p.HW;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0.0) var = 0.0;
  p.stats[idx * 2 + 0] = (float)mean;
  p.stats[idx * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnParams p) {
  constexpr int EPV = VecOf<T>::N;
  const int nvec = p.C / EPV;
  const size_t total = (size_t)p.B * p.HW * nvec;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = idx / nvec;                 // b*HW + px
    const int v = (int)(idx - pix * nvec);
    const int b = (int)(pix / p.HW);
    const int c = v * EPV;
    const T* src; int cs, cc;
    if (c < p.c0) { src = (const T*)p.x0; cs = p.c0; cc = c; }
    else { src = (const T*)p.x1; cs = p.c1; cc = c - p.c0; }
    float f[EPV];
    load_vec<T>(src + pix * cs + cc, f);
    const int g0 = c / p.cpg;
    const int boundary = (g0 + 1) * p.cpg;
    const float* st = p.stats + ((size_t)b * p.groups + g0) * 2;
    const float m0 = st[0], r0 = st[1];
    float m1 = m0, r1 = r0;
    if (boundary < c + EPV) { m1 = st[2]; r1 = st[3]; }
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const bool first = c + e < boundary;
      float y = (f[e] - (first ? m0 : m1)) * (first ? r0 : r1);
      y = y * p.gamma[c + e] + p.beta[c + e];
      if (p.silu) y = silu_f(y);
      f[e] = y;
    }
    store_vec<T>((T*)p.out + pix * p.C + c, f);
  }
}

template <typename T>
int launch_gn(const GroupNormArgs& a, hipStream_t stream) {
  constexpr int EPV = VecOf<T>::N;
  GnParams p;
  p.x0 = a.x0; p.x1 = a.x1; p.c0 = a.c0; p.c1 = a.x1 ? a.c1 : 0; p.C = p.c0 + p.c1;
  p.B = a.B; p.HW = a.HW; p.groups = a.groups;
  SDMI_REQUIRE(a.x0 && a.out && a.gamma && a.beta && a.scratch, "groupnorm: null pointer");
  SDMI_REQUIRE(p.groups > 0 && p.groups <= 64 && p.C % p.groups == 0, "groupnorm: C=%d groups=%d", p.C, p.groups);
  p.cpg = p.C / p.groups;
  SDMI_REQUIRE(p.c0 % EPV == 0 && p.c1 % EPV == 0, "groupnorm: channels must be multiples of %d", EPV);
  SDMI_REQUIRE(p.cpg >= EPV || 2 * p.cpg == EPV || p.cpg == EPV,
               "groupnorm: a %d-element vector may span at most two groups (cpg=%d)", EPV, p.cpg);
  SDMI_REQUIRE(p.B > 0 && p.HW > 0, "groupnorm: empty input");
  p.eps = a.eps; p.gamma = a.gamma; p.beta = a.beta; p.silu = a.silu; p.out = a.out;
  int chunks = (p.HW + 63) / 64;
  if (chunks > GN_MAX_CHUNKS) chunks = GN_MAX_CHUNKS;
  p.pix_per_chunk = (p.HW + chunks - 1) / chunks;
  p.chunks = (p.HW + p.pix_per_chunk - 1) / p.pix_per_chunk;
  p.partial = a.scratch;
  p.stats = a.scratch + (size_t)p.B * GN_MAX_CHUNKS * p.groups * 2;
  const int nvec = p.C / EPV;
  SDMI_REQUIRE(nvec <= GN_LDS_SLOTS, "groupnorm: C=%d too wide", p.C);
  int threads = nvec <= 256 ? nvec * (256 / nvec) : 256;
  ProfScope prof(PROF_GROUPNORM, stream, 0.0, 2.0 * p.B * (double)p.HW * p.C * sizeof(T));
  hipLaunchKernelGGL(gn_partial_kernel<T>, dim3(p.chunks, p.B), dim3(threads), 0, stream, p);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((p.B * p.groups + 63) / 64), dim3(64), 0, stream, p);
  const size_t total = (size_t)p.B * p.HW * nvec;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gn_apply_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

// ---- LayerNorm: one wave per row, row held in registers, two-pass (mean, then centred variance) --
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, T* out, const float* gamma,
                                                        const float* beta, int rows, int d, float eps) {
  constexpr int EPV = VecOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = d / EPV;
  float f[MAXV][EPV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nvec) {
      load_vec<T>(x + (size_t)row * d + v * EPV, f[i]);
#pragma unroll
      for (int e = 0; e < EPV; ++e) s += f[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nvec) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) { const float t = f[i][e] - mean; q += t * t; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nvec) {
      float y[EPV];
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        const int c = v * EPV + e;
        y[e] = (f[i][e] - mean) * rstd * gamma[c] + beta[c];
      }
      store_vec<T>(out + (size_t)row * d + v * EPV, y);
    }
  }
}

template <typename T>
int launch_ln(const void* x, void* out, const float* gamma, const float* beta, int rows, int d,
              float eps, hipStream_t stream) {
  constexpr int EPV = VecOf<T>::N;
  SDMI_REQUIRE(x && out && gamma && beta, "layernorm: null pointer");
  SDMI_REQUIRE(d % EPV == 0 && d > 0 && rows > 0, "layernorm: d=%d must be a positive multiple of %d", d, EPV);
  const int nvec = d / EPV;
  const dim3 grid((rows + 3) / 4), block(256);
  ProfScope prof(PROF_LAYERNORM, stream, 0.0, 2.0 * rows * (double)d * sizeof(T));
  if (nvec <= 64 * 3)
    hipLaunchKernelGGL((layernorm_kernel<T, 3>), grid, block, 0, stream, (const T*)x, (T*)out, gamma, beta, rows, d, eps);
  else if (nvec <= 64 * 8)
    hipLaunchKernelGGL((layernorm_kernel<T, 8>), grid, block, 0, stream, (const T*)x, (T*)out, gamma, beta, rows, d, eps);
  else {
    sdmi_set_error("layernorm: d=%d too wide", d);
    return SDMI_EINVAL;
  }
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

}  // namespace

size_t groupnorm_scratch_floats(int B, int HW, int groups) {
  (void)HW;
  return (size_t)B * GN_MAX_CHUNKS * groups * 2 + (size_t)B * groups * 2;
}

int groupnorm(const GroupNormArgs& a, hipStream_t stream) {
  SDMI_DISPATCH_DTYPE(a.dtype, T, return launch_gn<T>(a, stream));
  return SDMI_OK;
}

int layernorm(const void* x, void* out, const float* gamma, const float* beta, int rows, int d,
              float eps, int dtype, hipStream_t stream) {
  SDMI_DISPATCH_DTYPE(dtype, T, return launch_ln<T>(x, out, gamma, beta, rows, d, eps, stream));
  return SDMI_OK;
}
