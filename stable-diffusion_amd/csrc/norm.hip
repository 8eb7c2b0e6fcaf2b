// GroupNorm(+SiLU) and LayerNorm for NHWC / token-major activations on gfx950.  HBM-bound kernels:
// 16-byte vector loads/stores, fp32 statistics, double-precision combination of partial sums.
//
// GroupNorm of a small tensor is ONE launch (gn_group_kernel below: a workgroup per (sample, group), the slice
// held in registers).  Larger tensors take two launches:
//   1. gn_partial : grid (strips, B); every thread owns one 16-byte channel vector and walks a strip
//                   of pixels, keeping per-channel sum / sum-of-squares in registers; the block folds
//                   them into per-group partial sums in LDS (fixed order) and writes [B][strip][G][2].
//   2. gn_apply   : same grid; each block combines the strip partials of its sample in double ->
//                   mean, rstd, then normalises its strip with per-thread precomputed scale/shift
//                   (+ SiLU), 16-byte loads and stores.
// The input may be the channel-concatenation of two tensors (UNet skip connections); groups may
// straddle the seam, which is why statistics are accumulated per channel first.
#include "ops.h"
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

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

// Thread layout shared by both kernels: blockDim = nvec * lanes_pix when one pixel's vectors fit a
// block (a thread owns ONE channel vector and walks every lanes_pix-th pixel of the block's strip);
// wider tensors make a thread stride over vectors and take every pixel.
struct GnLayout {
  int nvec, lanes_pix, pix_lane, vstride, v_first, p_begin, p_end;
};
template <int EPV>
__device__ __forceinline__ GnLayout gn_layout(const GnParams& p, int chunk) {
  GnLayout L;
  L.nvec = p.C / EPV;
  const bool fits = (int)blockDim.x >= L.nvec;
  L.lanes_pix = fits ? (int)blockDim.x / L.nvec : 1;
  L.pix_lane = fits ? (int)threadIdx.x / L.nvec : 0;
  L.vstride = fits ? L.nvec : (int)blockDim.x;
  L.v_first = fits ? (int)threadIdx.x - L.pix_lane * L.nvec : (int)threadIdx.x;
  L.p_begin = chunk * p.pix_per_chunk;
  L.p_end = min(p.HW, L.p_begin + p.pix_per_chunk);
  return L;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(GnParams p) {
  constexpr int EPV = VecOf<T>::N;
  // per-(pixel lane, vector) partial sums: [0]=sum,[1]=sumsq of the vector's first group, [2],[3] of
  // its second group (a 16-byte vector spans at most two groups)
  __shared__ float ls[4][GN_LDS_SLOTS];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const GnLayout L = gn_layout<EPV>(p, chunk);
  for (int v = L.v_first; v < L.nvec; v += L.vstride) {
    const int c = v * EPV;
    const T* src; int cs, cc;
    if (c < p.c0) { src = (const T*)p.x0; cs = p.c0; cc = c; }
    else { src = (const T*)p.x1; cs = p.c1; cc = c - p.c0; }
    src += (size_t)b * p.HW * cs + cc;
    float s[EPV], q[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) { s[e] = 0.f; q[e] = 0.f; }
    int px = L.p_begin + L.pix_lane;
    // four independent 16-byte loads in flight per thread
    for (; px + 3 * L.lanes_pix < L.p_end; px += 4 * L.lanes_pix) {
      float f0[EPV], f1[EPV], f2[EPV], f3[EPV];
      load_vec<T>(src + (size_t)px * cs, f0);
      load_vec<T>(src + (size_t)(px + L.lanes_pix) * cs, f1);
      load_vec<T>(src + (size_t)(px + 2 * L.lanes_pix) * cs, f2);
      load_vec<T>(src + (size_t)(px + 3 * L.lanes_pix) * cs, f3);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        s[e] += (f0[e] + f1[e]) + (f2[e] + f3[e]);
        q[e] += (f0[e] * f0[e] + f1[e] * f1[e]) + (f2[e] * f2[e] + f3[e] * f3[e]);
      }
    }
    for (; px < L.p_end; px += L.lanes_pix) {
      float f[EPV];
      load_vec<T>(src + (size_t)px * cs, f);
#pragma unroll
      for (int e = 0; e < EPV; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
    }
    const int boundary = (c / p.cpg + 1) * p.cpg;
    float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      if (c + e < boundary) { sa += s[e]; qa += q[e]; } else { sb += s[e]; qb += q[e]; }
    }
    const int slot = L.pix_lane * L.nvec + v;
    ls[0][slot] = sa; ls[1][slot] = qa; ls[2][slot] = sb; ls[3][slot] = qb;
  }
  __syncthreads();
  // fixed-order fold per group: results do not depend on thread timing
  if ((int)threadIdx.x < p.groups) {
    const int g = threadIdx.x;
    const int v_lo = (g * p.cpg) / EPV, v_hi = ((g + 1) * p.cpg - 1) / EPV;
    float s = 0.f, q = 0.f;
    for (int pl = 0; pl < L.lanes_pix; ++pl)
      for (int v = v_lo; v <= v_hi; ++v) {
        const int slot = pl * L.nvec + v;
        const int g0 = (v * EPV) / p.cpg;
        if (g0 == g) { s += ls[0][slot]; q += ls[1][slot]; }
        else { s += ls[2][slot]; q += ls[3][slot]; }
      }
    float* dst = p.partial + (((size_t)b * p.chunks + chunk) * p.groups + g) * 2;
    dst[0] = s;
    dst[1] = q;
  }
}

// Second pass: every block first folds the chunk partials of its sample into (mean, rstd) per group
// (8 slices per group in parallel, combined in double in a fixed order), then normalises its strip
// with per-thread precomputed scale/shift: y = x * (rstd*gamma) + (beta - mean*rstd*gamma) [+ SiLU].
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnParams p) {
  constexpr int EPV = VecOf<T>::N;
  __shared__ double red[2][8][64];
  __shared__ float s_mean[64], s_rstd[64];
  const int chunk = blockIdx.x, b = blockIdx.y;
  {
    const int t = threadIdx.x;
    for (int u = t; u < 8 * p.groups; u += (int)blockDim.x) {
      const int g = u % p.groups, slice = u / p.groups;
      double s = 0.0, q = 0.0;
      const float2* src = reinterpret_cast<const float2*>(p.partial) + ((size_t)b * p.chunks * p.groups + g);
      const size_t cstep = (size_t)p.groups;
      int c = slice;
      // four loads in flight; the sum keeps its fixed order
      for (; c + 24 < p.chunks; c += 32) {
        const float2 v0 = src[c * cstep], v1 = src[(c + 8) * cstep], v2 = src[(c + 16) * cstep], v3 = src[(c + 24) * cstep];
        s += (double)v0.x; q += (double)v0.y;
        s += (double)v1.x; q += (double)v1.y;
        s += (double)v2.x; q += (double)v2.y;
        s += (double)v3.x; q += (double)v3.y;
      }
      for (; c < p.chunks; c += 8) {
        const float2 v = src[c * cstep];
        s += (double)v.x; q += (double)v.y;
      }
      red[0][slice][g] = s; red[1][slice][g] = q;
    }
    __syncthreads();
    if (t < p.groups) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { s += red[0][k][t]; q += red[1][k][t]; }
      const double n = (double)p.cpg * (double)p.HW;
      const double mean = s / n;
      double var = q / n - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[t] = (float)mean;
      s_rstd[t] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
  }
  const GnLayout L = gn_layout<EPV>(p, chunk);
  for (int v = L.v_first; v < L.nvec; v += L.vstride) {
    const int c = v * EPV;
    const T* src; int cs, cc;
    if (c < p.c0) { src = (const T*)p.x0; cs = p.c0; cc = c; }
    else { src = (const T*)p.x1; cs = p.c1; cc = c - p.c0; }
    src += (size_t)b * p.HW * cs + cc;
    T* dst = (T*)p.out + (size_t)b * p.HW * p.C + c;
    const int g0 = c / p.cpg;
    const int boundary = (g0 + 1) * p.cpg;
    float sc[EPV], sh[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const int g = (c + e < boundary) ? g0 : g0 + 1;
      const float rg = s_rstd[g] * p.gamma[c + e];
      sc[e] = rg;
      sh[e] = p.beta[c + e] - s_mean[g] * rg;
    }
    int px = L.p_begin + L.pix_lane;
    // four pixels per trip: four independent 16-byte loads in flight per thread
    for (; px + 3 * L.lanes_pix < L.p_end; px += 4 * L.lanes_pix) {
      float f[4][EPV];
#pragma unroll
      for (int u = 0; u < 4; ++u) load_vec<T>(src + (size_t)(px + u * L.lanes_pix) * cs, f[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          f[u][e] = fmaf(f[u][e], sc[e], sh[e]);
          if (p.silu) f[u][e] = silu_f(f[u][e]);
        }
        store_vec<T>(dst + (size_t)(px + u * L.lanes_pix) * p.C, f[u]);
      }
    }
    for (; px < L.p_end; px += L.lanes_pix) {
      float f[EPV];
      load_vec<T>(src + (size_t)px * cs, f);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        f[e] = fmaf(f[e], sc[e], sh[e]);
        if (p.silu) f[e] = silu_f(f[e]);
      }
      store_vec<T>(dst + (size_t)px * p.C, f);
    }
  }
}

// One-launch GroupNorm for tensors whose (sample, group) slice fits the registers of one workgroup (every
// GroupNorm of the UNet): a workgroup owns one (sample, group), reads its HW x cpg elements ONCE, 4 bytes per
// lane (dpp = 4-byte words per pixel; the block size is a multiple of dpp, so a thread keeps one word column j
// and walks pixels p0, p0 + pstep, ...; consecutive threads cover consecutive words of the slice), takes the
// mean and then the centred variance from the registers (two fixed-order folds: lanes, then waves through LDS)
// and writes the normalised values.  No partial sums leave the workgroup, the tensor is read once instead of
// twice, and results do not depend on timing.
struct GnGroupGeo {
  int dpp, pstep, threads, iters, map;     // map: 0 natural, 1 = a sample's groups on one XCD, 2 = G/8 adjacent groups per XCD
};

template <typename T, int MAXI>
__global__ __launch_bounds__(1024) void gn_group_kernel(GnParams p, GnGroupGeo geo) {
  constexpr int EPD = 4 / (int)sizeof(T);
  __shared__ float red[2][16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = (int)blockDim.x >> 6;
  // workgroups go to the XCDs round-robin by id: keep the workgroups that share 128-byte lines (neighbouring
  // groups of one sample) on one XCD, so that its L2 fetches each line once
  const int id = blockIdx.x;
  int b, g;
  if (geo.map == 1) {
    const int k = id >> 3;
    b = (id & 7) + 8 * (k / p.groups); g = k % p.groups;
  } else if (geo.map == 2) {
    const int k = id >> 3, gpx = p.groups >> 3;
    g = (id & 7) * gpx + k % gpx; b = k / gpx;
  } else {
    b = id / p.groups; g = id % p.groups;
  }
  const int j = t % geo.dpp, p0 = t / geo.dpp;
  const int c = g * p.cpg + j * EPD;
  const uint32_t* src; uint32_t rs;       // row stride in 4-byte words
  if (c < p.c0) { src = reinterpret_cast<const uint32_t*>((const T*)p.x0 + (size_t)b * p.HW * p.c0 + c); rs = p.c0 / EPD; }
  else { src = reinterpret_cast<const uint32_t*>((const T*)p.x1 + (size_t)b * p.HW * p.c1 + (c - p.c0)); rs = p.c1 / EPD; }
  uint32_t d[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int px = p0 + i * geo.pstep;
    d[i] = px < p.HW ? src[(uint32_t)px * rs] : 0u;
  }
  auto elem = [](uint32_t w, int e) -> float {
    if constexpr (EPD == 1) return __builtin_bit_cast(float, w);
    else return to_f32<T>(__builtin_bit_cast(T, (unsigned short)(e ? (w >> 16) : (w & 0xffffu))));
  };
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
#pragma unroll
    for (int e = 0; e < EPD; ++e) s += elem(d[i], e);       // words past the tensor are zero
  // the registers keep the packed words only: each pass unpacks them again
#pragma unroll
  for (int i = 0; i < MAXI; ++i) asm volatile("" : "+v"(d[i]));
  s = wave_sum(s);
  if (lane == 0) red[0][wave] = s;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < nw; ++w) tot += red[0][w];
  const float inv_n = 1.0f / ((float)p.cpg * (float)p.HW);
  const float mean = tot * inv_n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
    if (p0 + i * geo.pstep < p.HW) {
#pragma unroll
      for (int e = 0; e < EPD; ++e) { const float u = elem(d[i], e) - mean; q = fmaf(u, u, q); }
    }
#pragma unroll
  for (int i = 0; i < MAXI; ++i) asm volatile("" : "+v"(d[i]));
  q = wave_sum(q);
  if (lane == 0) red[1][wave] = q;
  __syncthreads();
  float qt = 0.f;
  for (int w = 0; w < nw; ++w) qt += red[1][w];
  const float rstd = 1.0f / sqrtf(qt * inv_n + p.eps);
  float sc[EPD], sh[EPD];
#pragma unroll
  for (int e = 0; e < EPD; ++e) {
    sc[e] = rstd * p.gamma[c + e];
    sh[e] = fmaf(-mean, sc[e], p.beta[c + e]);
  }
  uint32_t* dst = reinterpret_cast<uint32_t*>((T*)p.out + (size_t)b * p.HW * p.C + c);
  const uint32_t os = p.C / EPD;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int px = p0 + i * geo.pstep;
    if (px < p.HW) {
      float y[EPD];
#pragma unroll
      for (int e = 0; e < EPD; ++e) {
        y[e] = fmaf(elem(d[i], e), sc[e], sh[e]);
        if (p.silu) y[e] = silu_f(y[e]);
      }
      uint32_t w;
      if constexpr (EPD == 1) w = __builtin_bit_cast(uint32_t, y[0]);
      else w = (uint32_t)__builtin_bit_cast(unsigned short, from_f32<T>(y[0])) |
               ((uint32_t)__builtin_bit_cast(unsigned short, from_f32<T>(y[EPD - 1])) << 16);
      dst[(uint32_t)px * os] = w;
    }
  }
}

// Geometry of the one-launch kernel, or threads == 0 when the tensor does not fit it.
template <typename T>
GnGroupGeo gn_group_geo(const GnParams& p) {
  constexpr int EPD = 4 / (int)sizeof(T);
  GnGroupGeo geo = {0, 0, 0, 0, 0};
  // A workgroup touches every 128-byte line of its pixels for cpg * sizeof(T) bytes of it; the lines it shares
  // with the neighbouring groups come through L2 once per group.  Beyond this much line traffic per workgroup
  // (KiB; measured: slices of the 64x64 level lose, everything smaller gains) the two-launch path is faster.
  static const long max_line_kib = [] { const char* e = getenv("SDMI_GN_ONE_LAUNCH_KB"); return e ? atol(e) : 256L; }();
  if (p.cpg % EPD != 0 || p.c0 % EPD != 0 || p.c1 % EPD != 0) return geo;
  const size_t chunk_bytes = (size_t)p.cpg * sizeof(T);
  const size_t line_bytes = (size_t)p.HW * (chunk_bytes >= 128 ? chunk_bytes : 128);
  if (line_bytes > (size_t)max_line_kib * 1024) return geo;
  if ((size_t)p.HW * (size_t)p.C * sizeof(T) >= ((size_t)1 << 31)) return geo;      // 32-bit word offsets inside a sample
  const int dpp = p.cpg / EPD;
  // whole waves and a multiple of dpp: a thread's word column never changes
  int unit = dpp;
  while (unit % 64 != 0) unit += dpp;
  if (unit > 1024) return geo;
  const int threads = (1024 / unit) * unit, pstep = threads / dpp;
  const int iters = (p.HW + pstep - 1) / pstep;
  if (iters > (sizeof(T) == 4 ? 32 : 48)) return geo;      // more words per thread than the registers hold without spilling
  geo.dpp = dpp; geo.pstep = pstep; geo.threads = threads; geo.iters = iters;
  geo.map = (p.B % 8 == 0) ? 1 : ((p.groups % 8 == 0) ? 2 : 0);
  return geo;
}

template <typename T>
void launch_gn_group(const GnParams& p, const GnGroupGeo& geo, hipStream_t stream) {
  const dim3 grid(p.B * p.groups), block(geo.threads);
#define SDMI_GN_GROUP(MAXI) hipLaunchKernelGGL((gn_group_kernel<T, MAXI>), grid, block, 0, stream, p, geo)
  if (geo.iters <= 4) SDMI_GN_GROUP(4);
  else if (geo.iters <= 8) SDMI_GN_GROUP(8);
  else if (geo.iters <= 12) SDMI_GN_GROUP(12);
  else if (geo.iters <= 16) SDMI_GN_GROUP(16);
  else if (geo.iters <= 24) SDMI_GN_GROUP(24);
  else if (geo.iters <= 32) SDMI_GN_GROUP(32);
  else SDMI_GN_GROUP(48);
#undef SDMI_GN_GROUP
}

template <typename T>
int launch_gn(const GroupNormArgs& a, hipStream_t stream) {
  constexpr int EPV = VecOf<T>::N;
  GnParams p;
  p.x0 = a.x0; p.x1 = a.x1; p.c0 = a.c0; p.c1 = a.x1 ? a.c1 : 0; p.C = p.c0 + p.c1;
  p.B = a.B; p.HW = a.HW; p.groups = a.groups;
  SDMI_REQUIRE(a.x0 && a.out && a.gamma && a.beta && a.scratch, "groupnorm: null pointer");
  SDMI_REQUIRE(p.groups > 0 && p.groups <= 32 && p.C % p.groups == 0, "groupnorm: C=%d groups=%d", p.C, p.groups);
  p.cpg = p.C / p.groups;
  SDMI_REQUIRE(p.c0 % EPV == 0 && p.c1 % EPV == 0, "groupnorm: channels must be multiples of %d", EPV);
  SDMI_REQUIRE(p.cpg >= EPV || 2 * p.cpg == EPV || p.cpg == EPV,
               "groupnorm: a %d-element vector may span at most two groups (cpg=%d)", EPV, p.cpg);
  SDMI_REQUIRE(p.B > 0 && p.HW > 0, "groupnorm: empty input");
  p.eps = a.eps; p.gamma = a.gamma; p.beta = a.beta; p.silu = a.silu; p.out = a.out;
  char label[64];
  snprintf(label, sizeof(label), "B%d HW%d C%d", p.B, p.HW, p.C);
  const GnGroupGeo geo = gn_group_geo<T>(p);
  if (geo.threads) {
    p.chunks = 0; p.pix_per_chunk = 0; p.partial = nullptr; p.stats = nullptr;
    ProfScope prof(PROF_GROUPNORM, stream, 0.0, 2.0 * p.B * (double)p.HW * p.C * sizeof(T), label);
    launch_gn_group<T>(p, geo, stream);
    SDMI_CHECK_HIP(hipGetLastError());
    return SDMI_OK;
  }
  const int nvec = p.C / EPV;
  SDMI_REQUIRE(nvec <= GN_LDS_SLOTS, "groupnorm: C=%d too wide", p.C);
  const int threads = nvec <= 256 ? nvec * (256 / nvec) : 256;
  const int lanes_pix = nvec <= 256 ? 256 / nvec : 1;
  // about two workgroups per CU in all (more, smaller strips pay the per-block statistics prologue
  // too often; measured best at 512), each strip at least 4 pixels per pixel lane
  static const int target_blocks = [] { const char* e = getenv("SDMI_GN_BLOCKS"); return e ? atoi(e) : 512; }();
  int chunks = (target_blocks + p.B - 1) / p.B;
  const int max_by_work = (p.HW + 4 * lanes_pix - 1) / (4 * lanes_pix);
  if (chunks > max_by_work) chunks = max_by_work;
  if (chunks > GN_MAX_CHUNKS) chunks = GN_MAX_CHUNKS;
  // every second-pass block folds its sample's chunk partials before it starts: with few samples a long
  // list costs more (serial round trips) than the extra blocks gain
  static const int chunk_cap = [] { const char* e = getenv("SDMI_GN_CHUNK_CAP"); return e ? atoi(e) : 128; }();
  if (chunks > chunk_cap) chunks = chunk_cap;
  if (chunks < 1) chunks = 1;
  p.pix_per_chunk = (p.HW + chunks - 1) / chunks;
  p.chunks = (p.HW + p.pix_per_chunk - 1) / p.pix_per_chunk;
  p.partial = a.scratch;
  p.stats = nullptr;
  ProfScope prof(PROF_GROUPNORM, stream, 0.0, 2.0 * p.B * (double)p.HW * p.C * sizeof(T), label);
  hipLaunchKernelGGL(gn_partial_kernel<T>, dim3(p.chunks, p.B), dim3(threads), 0, stream, p);
  hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(p.chunks, p.B), dim3(threads), 0, stream, p);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

// ---- LayerNorm: one wave per R rows, rows held in registers, two-pass (mean, then centred variance).
// R rows are loaded before the first reduction so that several 16-byte loads per lane are in flight and
// the shuffle chains of the rows interleave.
template <typename T, int MAXV, int R>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, T* out, const float* gamma,
                                                        const float* beta, int rows, int d, float eps) {
  constexpr int EPV = VecOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  const int nvec = d / EPV;
  float f[R][MAXV][EPV];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = 0.f;
    const int row = min(row0 + r, rows - 1);        // tail rows recompute the last row and do not store
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + i * 64;
      if (v < nvec) load_vec<T>(x + (size_t)row * d + v * EPV, f[r][i]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (lane + i * 64 < nvec) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) s[r] += f[r][i][e];
      }
  float mean[R], q[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) / (float)d;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    q[r] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (lane + i * 64 < nvec) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) { const float t = f[r][i][e] - mean[r]; q[r] += t * t; }
      }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(wave_sum(q[r]) / (float)d + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nvec) {
      float g[EPV], b[EPV];
#pragma unroll
      for (int e = 0; e < EPV; e += 4) {
        *reinterpret_cast<float4*>(g + e) = *reinterpret_cast<const float4*>(gamma + v * EPV + e);
        *reinterpret_cast<float4*>(b + e) = *reinterpret_cast<const float4*>(beta + v * EPV + e);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (row0 + r < rows) {
          float y[EPV];
#pragma unroll
          for (int e = 0; e < EPV; ++e) y[e] = (f[r][i][e] - mean[r]) * rstd[r] * g[e] + b[e];
          store_vec<T>(out + (size_t)(row0 + r) * d + v * EPV, y);
        }
      }
    }
  }
}

// Row statistics only: (rstd, rstd * mean) per row.  LPR lanes share a row (lane q takes the 16-byte vectors
// q, q + LPR, ... of it, so a load instruction covers LPR * 16 contiguous bytes of each of its rows) and a wave
// works on 64 / LPR rows at once: with 8 lanes per row the cross-lane fold is three steps for eight rows
// instead of six per row.  Sums are taken about the row's first element (shifted data), which keeps
// E[t^2] - E[t]^2 free of cancellation for rows with a large common offset; nothing but the two running sums
// stays in registers.
template <typename T, int LPR>
__global__ __launch_bounds__(256) void layernorm_stats_kernel(const T* x, float2* stats, int rows, int d, float eps) {
  constexpr int EPV = VecOf<T>::N, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, r = lane / LPR, q = lane % LPR;
  const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW + r;
  const T* xr = x + (size_t)min(row, rows - 1) * d;
  const int nvec = d / EPV;
  const float shift = to_f32<T>(xr[0]);
  float s = 0.f, ss = 0.f;
  int v = q;
  for (; v + 3 * LPR < nvec; v += 4 * LPR) {          // four independent 16-byte loads in flight per lane
    float f[4][EPV];
#pragma unroll
    for (int u = 0; u < 4; ++u) load_vec<T>(xr + (v + LPR * u) * EPV, f[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < EPV; ++e) { const float t = f[u][e] - shift; s += t; ss = fmaf(t, t, ss); }
  }
  for (; v < nvec; v += LPR) {
    float f[EPV];
    load_vec<T>(xr + v * EPV, f);
#pragma unroll
    for (int e = 0; e < EPV; ++e) { const float t = f[e] - shift; s += t; ss = fmaf(t, t, ss); }
  }
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
  const float m1 = s / (float)d;
  const float var = fmaxf(ss / (float)d - m1 * m1, 0.f);
  const float rstd = rsqrtf(var + eps);
  if (q == 0 && row < rows) stats[row] = make_float2(rstd, rstd * (shift + m1));
}

template <typename T>
int launch_ln_stats(const void* x, float* stats, int rows, int d, float eps, hipStream_t stream) {
  constexpr int EPV = VecOf<T>::N;
  SDMI_REQUIRE(x && stats, "layernorm_stats: null pointer");
  SDMI_REQUIRE(d % EPV == 0 && d > 0 && rows > 0, "layernorm_stats: d=%d must be a positive multiple of %d", d, EPV);
  char label[32];
  snprintf(label, sizeof(label), "stats rows%d d%d", rows, d);
  ProfScope prof(PROF_LAYERNORM, stream, 0.0, (double)rows * d * sizeof(T), label);
  float2* st = reinterpret_cast<float2*>(stats);
  // lanes per row: as few as still give the chip about 4096 waves (16 per CU), and no more than the row has vectors
  int lpr = 8;
  while (lpr < 64 && (long)rows * lpr / 64 < 4096 && 2 * lpr <= d / EPV) lpr *= 2;
  const int rpb = 4 * (64 / lpr);           // rows per 256-thread workgroup
  const dim3 grid((rows + rpb - 1) / rpb), block(256);
  if (lpr == 8) hipLaunchKernelGGL((layernorm_stats_kernel<T, 8>), grid, block, 0, stream, (const T*)x, st, rows, d, eps);
  else if (lpr == 16) hipLaunchKernelGGL((layernorm_stats_kernel<T, 16>), grid, block, 0, stream, (const T*)x, st, rows, d, eps);
  else if (lpr == 32) hipLaunchKernelGGL((layernorm_stats_kernel<T, 32>), grid, block, 0, stream, (const T*)x, st, rows, d, eps);
  else hipLaunchKernelGGL((layernorm_stats_kernel<T, 64>), grid, block, 0, stream, (const T*)x, st, rows, d, eps);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

template <typename T>
int launch_ln(const void* x, void* out, const float* gamma, const float* beta, int rows, int d,
              float eps, hipStream_t stream) {
  constexpr int EPV = VecOf<T>::N;
  SDMI_REQUIRE(x && out && gamma && beta, "layernorm: null pointer");
  SDMI_REQUIRE(d % EPV == 0 && d > 0 && rows > 0, "layernorm: d=%d must be a positive multiple of %d", d, EPV);
  const int nvec = d / EPV;
  char label[32];
  snprintf(label, sizeof(label), "rows%d d%d", rows, d);
  ProfScope prof(PROF_LAYERNORM, stream, 0.0, 2.0 * rows * (double)d * sizeof(T), label);
  const dim3 block(256);
  auto grid = [&](int r) { return dim3((rows + 4 * r - 1) / (4 * r)); };
  // rows per wave: enough loads in flight for narrow rows, fewer when there are too few rows to fill the chip
  if (nvec <= 64 && rows >= 16384)
    hipLaunchKernelGGL((layernorm_kernel<T, 1, 4>), grid(4), block, 0, stream, (const T*)x, (T*)out, gamma, beta, rows, d, eps);
  else if (nvec <= 64)
    hipLaunchKernelGGL((layernorm_kernel<T, 1, 1>), grid(1), block, 0, stream, (const T*)x, (T*)out, gamma, beta, rows, d, eps);
  else if (nvec <= 64 * 2 && rows >= 8192)
    hipLaunchKernelGGL((layernorm_kernel<T, 2, 2>), grid(2), block, 0, stream, (const T*)x, (T*)out, gamma, beta, rows, d, eps);
  else if (nvec <= 64 * 3)
    hipLaunchKernelGGL((layernorm_kernel<T, 3, 1>), grid(1), block, 0, stream, (const T*)x, (T*)out, gamma, beta, rows, d, eps);
  else if (nvec <= 64 * 8)
    hipLaunchKernelGGL((layernorm_kernel<T, 8, 1>), grid(1), block, 0, stream, (const T*)x, (T*)out, gamma, beta, rows, d, eps);
  else {
    sdmi_set_error("layernorm: d=%d too wide", d);
    return SDMI_EINVAL;
  }
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

}  // namespace

// ---- GroupNorm for callers that cannot allocate -----------------------------------------------------
// The two GroupNorm launches hand the per-strip partial sums to each other through a scratch buffer that
// sdmi_groupnorm takes from its caller.  The variant below owns that buffer instead: a file-scope device
// array, one copy per GPU at that GPU's own address, held until the process ends.  In host code the
// array's name is only a handle for hipGetSymbolAddress; everything that touches the memory (the kernel
// argument) uses the address that call returned for the current device.
constexpr int GN_STATIC_MAX_BATCH = 32;
constexpr int GN_STATIC_MAX_GROUPS = 32;
constexpr size_t GN_STATIC_FLOATS =
    (size_t)GN_STATIC_MAX_BATCH * GN_MAX_CHUNKS * GN_STATIC_MAX_GROUPS * 2 + (size_t)GN_STATIC_MAX_BATCH * GN_STATIC_MAX_GROUPS * 2;
static __device__ __align__(256) float scratch[GN_STATIC_FLOATS];      // 2 MiB + 8 KiB

namespace {
constexpr int GN_MAX_DEVICES = 64;
float* g_gn_static_ptr[GN_MAX_DEVICES];          // device address of `scratch` on each GPU, resolved on first use
std::mutex g_gn_static_mu;
}  // namespace

// Resolve the current device's copy of `scratch` (once per device).  hipGetSymbolAddress sets the device variable
// up lazily, which must not first happen inside a stream capture: the model handles call this from finalize, and a
// caller that uses the entry point on its own makes its first call per GPU outside capture.
int groupnorm_static_init(float** out) {
  int dev = -1;
  SDMI_CHECK_HIP(hipGetDevice(&dev));
  SDMI_REQUIRE(dev >= 0 && dev < GN_MAX_DEVICES, "groupnorm_static: device ordinal %d not supported", dev);
  std::lock_guard<std::mutex> lock(g_gn_static_mu);
  if (!g_gn_static_ptr[dev]) {
    void* sym = nullptr;
    SDMI_CHECK_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)));     // this device's copy of the array
    SDMI_REQUIRE(sym != nullptr, "groupnorm_static: the scratch symbol has no address on device %d", dev);
    g_gn_static_ptr[dev] = (float*)sym;
  }
  if (out) *out = g_gn_static_ptr[dev];
  return SDMI_OK;
}

int groupnorm_static(const GroupNormArgs& a_in, hipStream_t stream) {
  SDMI_REQUIRE(a_in.B > 0 && a_in.B <= GN_STATIC_MAX_BATCH,
               "groupnorm_static: batch %d is beyond the built-in scratch (at most %d samples per call; use "
               "sdmi_groupnorm with a caller-provided scratch buffer for more)", a_in.B, GN_STATIC_MAX_BATCH);
  SDMI_REQUIRE(a_in.groups > 0 && a_in.groups <= GN_STATIC_MAX_GROUPS,
               "groupnorm_static: %d groups are beyond the built-in scratch (at most %d)", a_in.groups,
               GN_STATIC_MAX_GROUPS);
  SDMI_REQUIRE(groupnorm_scratch_floats(a_in.B, a_in.HW, a_in.groups) <= GN_STATIC_FLOATS,
               "groupnorm_static: the call needs more scratch than is built in");
  float* ptr = nullptr;
  SDMI_TRY(groupnorm_static_init(&ptr));
  GroupNormArgs a = a_in;
  a.scratch = ptr;      // every partial sum the second launch reads is written by the first: nothing to clear
  return groupnorm(a, stream);
}

size_t groupnorm_scratch_floats(int B, int HW, int groups) {
  (void)HW;
  return (size_t)B * GN_MAX_CHUNKS * groups * 2 + (size_t)B * groups * 2;
}

int groupnorm(const GroupNormArgs& a, hipStream_t stream) {
  SDMI_DISPATCH_DTYPE(a.dtype, T, return launch_gn<T>(a, stream));
  return SDMI_OK;
}

int layernorm_stats(const void* x, float* stats, int rows, int d, float eps, int dtype, hipStream_t stream) {
  SDMI_DISPATCH_DTYPE(dtype, T, return launch_ln_stats<T>(x, stats, rows, d, eps, stream));
  return SDMI_OK;
}

int layernorm(const void* x, void* out, const float* gamma, const float* beta, int rows, int d,
              float eps, int dtype, hipStream_t stream) {
  SDMI_DISPATCH_DTYPE(dtype, T, return launch_ln<T>(x, out, gamma, beta, rows, d, eps, stream));
  return SDMI_OK;
}
