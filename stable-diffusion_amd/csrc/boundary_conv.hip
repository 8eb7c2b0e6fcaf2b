// The two convolutions on the NCHW fp32 boundary of the UNet and the VAE, on the gfx950 matrix cores.
//
//   in_conv   first convolution: NCHW fp32 sources (channel-concatenated, Cin <= 9) -> NHWC activations.
//             K = 9 Cin is tiny and the inputs are fp32, so the product runs on v_mfma_f32_16x16x4_f32
//             (exact fp32 products, a k-ordered fmaf chain): rows = output channels, columns = 16 pixels.
//             A wave keeps the weights of its 64 output channels in registers for the whole kernel and walks
//             groups of 16 pixels; output rows are paired so that a lane ends up with 8 consecutive channels of
//             one pixel (one 16-byte store, 64 contiguous bytes per pixel and wave).
//   out_conv  last convolution: NHWC activations (Cin = 128 / 320) -> NCHW fp32 with Cout <= 16.  One wave per
//             filter tap (9 waves per workgroup); a wave holds its tap's weights as MFMA A operands (rows =
//             output channels, padded to 16), streams the 16 pixels' channel vectors straight from global
//             memory as B operands (16 bytes per lane and MFMA) and leaves its partial sums in LDS; 16 x Cout
//             threads add bias and the nine partials in tap order and write 64-byte runs along W.
//             16-bit activations: the fp32 weights are split into (hi, lo) halves, two MFMAs per step, so
//             the weights keep ~22 bits although the operands are 16 bit.  fp32 activations: 16x16x4 fp32 MFMA.
//
// Shapes outside these (Cout not a multiple of 64, other Cin) run on the scalar kernels below.
#include "ops.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct InConvParams {
  const float* src[3]; int c[3]; int Cin;
  const float* w; const float* bias; void* out;
  int B, H, W, Cout;
};

// ---- first convolution, MFMA ------------------------------------------------------------------------
template <typename T, int CIN>
__global__ __launch_bounds__(512) void in_conv_mfma_kernel(InConvParams p) {
  constexpr int K = 9 * CIN, NS = (K + 3) / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int cbase = wave * 64;
  // tile q = 2u + hf; MFMA row i of tile q is output channel cbase + 32u + 8(i >> 2) + 4hf + (i & 3), so that the
  // accumulator rows 4g .. 4g+3 of tiles (u,0), (u,1) are the 8 consecutive channels cbase + 32u + 8g .. +7
  float wreg[NS][4];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int k = 4 * s + g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = cbase + 32 * (q >> 1) + 8 * (j >> 2) + 4 * (q & 1) + (j & 3);
      wreg[s][q] = k < K ? p.w[(size_t)k * p.Cout + ch] : 0.f;
    }
  }
  f32x4 bias[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    bias[q] = *reinterpret_cast<const f32x4*>(p.bias + cbase + 32 * (q >> 1) + 8 * g + 4 * (q & 1));
  const int HW = p.H * p.W;
  const long npix = (long)p.B * HW;
  const long ngroups = (npix + 15) / 16;
  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long pix = grp * 16 + j;
    const bool ok = pix < npix;
    const long pp = ok ? pix : 0;
    const int b = (int)(pp / HW), rem = (int)(pp - (long)b * HW);
    const int y = rem / p.W, x = rem - y * p.W;
    float xv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      int tap, ci;
      if constexpr (CIN == 4) { tap = s; ci = g; }
      else { const int k = 4 * s + g; tap = k / CIN; ci = k - tap * CIN; }
      const int ty = tap / 3;
      const int iy = y + ty - 1, ix = x + (tap - 3 * ty) - 1;
      const bool valid = ok && tap < 9 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const float* sp; int cs, cl;
      if (ci < p.c[0]) { sp = p.src[0]; cs = p.c[0]; cl = ci; }
      else if (ci < p.c[0] + p.c[1]) { sp = p.src[1]; cs = p.c[1]; cl = ci - p.c[0]; }
      else { sp = p.src[2]; cs = p.c[2]; cl = ci - p.c[0] - p.c[1]; }
      const float* addr = valid ? sp + (((size_t)b * cs + cl) * p.H + iy) * p.W + ix : p.src[0];
      const float v = *addr;
      xv[s] = valid ? v : 0.f;
    }
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = bias[q];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s][q], xv[s], acc[q], 0, 0, 0);
    if (ok) {
      T* o = (T*)p.out + (size_t)pix * p.Cout + cbase + 8 * g;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if constexpr (sizeof(T) == 2) {
          typedef T Tx8 __attribute__((ext_vector_type(8)));
          Tx8 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = from_f32<T>(acc[2 * u][e]); v[4 + e] = from_f32<T>(acc[2 * u + 1][e]); }
          *reinterpret_cast<Tx8*>(o + 32 * u) = v;
        } else {
          *reinterpret_cast<f32x4*>((float*)o + 32 * u) = acc[2 * u];
          *reinterpret_cast<f32x4*>((float*)o + 32 * u + 4) = acc[2 * u + 1];
        }
      }
    }
  }
}

// ---- first convolution, scalar (any Cin <= 64, Cout % 8 == 0) ----------------------------------------
// A thread owns 8 consecutive output channels of one pixel and a workgroup walks pixels with the
// whole [9 Cin][Cout] fp32 weight block parked in LDS.  Summation order per output: bias, then taps in
// order, sources in order, channels in order (the same order as the MFMA kernel's k index).
template <typename T>
__global__ __launch_bounds__(256) void in_conv_kernel(InConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float in_conv_w[];
  const int tid = threadIdx.x;
  const int nw = 9 * p.Cin * p.Cout;
  for (int i = tid * 4; i < nw; i += 256 * 4)
    *reinterpret_cast<float4*>(in_conv_w + i) = *reinterpret_cast<const float4*>(p.w + i);
  __syncthreads();
  const int tpp = p.Cout / 8;                 // threads per pixel
  const int ppb = 256 / tpp;                  // pixels per workgroup pass
  const int tp = tid % tpp, pl = tid / tpp;
  if (pl >= ppb) return;
  const int co = tp * 8;
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = p.bias[co + e];
  const size_t npix = (size_t)p.B * p.H * p.W;
  const size_t hw = (size_t)p.H * p.W;
  for (size_t pix = (size_t)blockIdx.x * ppb + pl; pix < npix; pix += (size_t)gridDim.x * ppb) {
    const int x = (int)(pix % p.W), y = (int)((pix / p.W) % p.H), b = (int)(pix / hw);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias[e];
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
      if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) continue;
      int cbase = 0;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (p.c[s] == 0) continue;
        const float* sp = p.src[s] + ((size_t)b * p.c[s] * p.H + iy) * p.W + ix;
        for (int ci = 0; ci < p.c[s]; ++ci) {
          const float xv = sp[(size_t)ci * hw];
          const float* wr = in_conv_w + (size_t)(tap * p.Cin + cbase + ci) * p.Cout + co;
          const float4 w0 = *reinterpret_cast<const float4*>(wr), w1 = *reinterpret_cast<const float4*>(wr + 4);
          acc[0] += xv * w0.x; acc[1] += xv * w0.y; acc[2] += xv * w0.z; acc[3] += xv * w0.w;
          acc[4] += xv * w1.x; acc[5] += xv * w1.y; acc[6] += xv * w1.z; acc[7] += xv * w1.w;
        }
        cbase += p.c[s];
      }
    }
    typedef T Tx8 __attribute__((ext_vector_type(8)));
    if constexpr (sizeof(T) == 2) {
      Tx8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(acc[e]);
      *reinterpret_cast<Tx8*>((T*)p.out + pix * p.Cout + co) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) ((T*)p.out)[pix * p.Cout + co + e] = from_f32<T>(acc[e]);
    }
  }
}

// ---- last convolution, MFMA ---------------------------------------------------------------------------
template <typename T> struct Mma16;
template <> struct Mma16<half_t> {
  typedef half_t v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma16<bf16_t> {
  typedef bf16_t v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

template <> struct Mma16<float> {
  typedef float v8 __attribute__((ext_vector_type(4)));       // placeholder: the fp32 path uses the 16x16x4 MFMA
};

constexpr int OUT_CONV_THREADS = 9 * 64;

// Pixel groups are dealt to the workgroups so that the workgroups of one XCD (blockIdx % 8; the grid is a
// multiple of 8) walk one contiguous band of the image together: the nine taps' overlapping reads of a row
// then meet in that XCD's L2 instead of being fetched by all eight.  A workgroup loads the next group's
// activations before it waits at the barrier of the current one.
template <typename T, int CIN>
__global__ __launch_bounds__(OUT_CONV_THREADS) void out_conv_mfma_kernel(const T* x, const float* w, const float* bias,
                                                                         float* out, int B, int H, int W, int Cout) {
  __shared__ __attribute__((aligned(16))) float part[2][9][16][16];     // [buffer][tap][pixel][output channel]
  const int lane = threadIdx.x & 63;
  const int tap = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
  const int HW = H * W;
  const long npix = (long)B * HW;
  const long ngroups = (npix + 15) / 16;
  const long per_xcd = (ngroups + 7) / 8;
  const long first = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const long last = min(ngroups, (long)((blockIdx.x & 7) + 1) * per_xcd);
  const long stride = gridDim.x >> 3;
  const float* wrow = w + ((size_t)min(j, Cout - 1) * 9 + tap) * CIN;     // A row j = output channel j (rows >= Cout: zero)
  const bool row_ok = j < Cout;

  constexpr bool F32 = sizeof(T) == 4;
  // 16-bit: step s covers channels 32s .. 32s+31, lane group g holds channels 32s + 8g .. +7 of both operands.
  // fp32: block u, element e is channel 16u + 4g + e on both operands (the lane's 16 contiguous bytes of x).
  constexpr int NV = F32 ? CIN / 16 : CIN / 32;                           // 16-byte vectors of x per lane and pixel
  typedef typename VecOf<T>::type xvec_t;
  typedef typename Mma16<T>::v8 v8;
  xvec_t whi[NV], wlo[NV];
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    constexpr int E = VecOf<T>::N;
    float wv[E];
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(wrow + E * (4 * s + g) + e);
      wv[e] = q.x; wv[e + 1] = q.y; wv[e + 2] = q.z; wv[e + 3] = q.w;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float v = row_ok ? wv[e] : 0.f;
      const T h = from_f32<T>(v);
      whi[s][e] = h;
      wlo[s][e] = from_f32<T>(v - to_f32<T>(h));        // fp32: zero, unused
    }
  }

  xvec_t xv[NV];
  bool valid = false;
  auto issue = [&](long grp) {
    const long pix = grp * 16 + j;
    const long pp = pix < npix ? pix : 0;
    const int b = (int)(pp / HW), rem = (int)(pp - (long)b * HW);
    const int y = rem / W, xx = rem - y * W;
    const int iy = y + dy, ix = xx + dx;
    valid = pix < npix && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const T* xp = valid ? x + (((size_t)b * H + iy) * W + ix) * CIN + VecOf<T>::N * g : x;
#pragma unroll
    for (int s = 0; s < NV; ++s) xv[s] = *reinterpret_cast<const xvec_t*>(xp + 4 * VecOf<T>::N * s);
  };
  if (first < last) issue(first);
  int buf = 0;
  for (long grp = first; grp < last; grp += stride) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!valid) {
#pragma unroll
      for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int e = 0; e < VecOf<T>::N; ++e) xv[s][e] = (T)0.f;
    }
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      if constexpr (F32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(whi[s][e], xv[s][e], acc, 0, 0, 0);
      } else {
        acc = Mma16<T>::mma(__builtin_bit_cast(v8, whi[s]), __builtin_bit_cast(v8, xv[s]), acc);
        acc = Mma16<T>::mma(__builtin_bit_cast(v8, wlo[s]), __builtin_bit_cast(v8, xv[s]), acc);
      }
    }
    if (grp + stride < last) issue(grp + stride);       // in flight across the barrier and the reduction below
    *reinterpret_cast<f32x4*>(&part[buf][tap][j][4 * g]) = acc;
    __syncthreads();
    if ((int)threadIdx.x < 16 * Cout) {
      const int o = threadIdx.x >> 4, jj = threadIdx.x & 15;
      const long px = grp * 16 + jj;
      if (px < npix) {
        float sum = bias[o];
#pragma unroll
        for (int t = 0; t < 9; ++t) sum += part[buf][t][jj][o];
        const int bb = (int)(px / HW), r2 = (int)(px - (long)bb * HW);
        out[((size_t)bb * Cout + o) * HW + r2] = sum;
      }
    }
    buf ^= 1;       // the next group's partials go to the other buffer: one barrier per group is enough
  }
}

// ---- last convolution, scalar: one wave per output pixel (any Cin % 8 == 0, Cout in {3, 4, 8}) ------------
template <typename T, int COUT>
__global__ __launch_bounds__(256) void out_conv_kernel(const T* x, const float* w, const float* bias, float* out,
                                                       int B, int H, int W, int Cin) {
  constexpr int EPV = VecOf<T>::N;
  typedef typename VecOf<T>::type V;
  const int lane = threadIdx.x & 63;
  const size_t pix = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (size_t)B * H * W) return;
  const int xx = (int)(pix % W), yy = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
  const int nvec = Cin / EPV;
  for (int tap = 0; tap < 9; ++tap) {
    const int iy = yy + tap / 3 - 1, ix = xx + tap % 3 - 1;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    const T* xp = x + (((size_t)b * H + iy) * W + ix) * Cin;
    for (int v = lane; v < nvec; v += 64) {
      const V xv = *reinterpret_cast<const V*>(xp + v * EPV);
#pragma unroll
      for (int o = 0; o < COUT; ++o) {
        const float* wp = w + ((size_t)o * 9 + tap) * Cin + v * EPV;
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[o] += to_f32<T>(xv[e]) * wp[e];
      }
    }
  }
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    const float s = wave_sum(acc[o]);
    if (lane == 0) out[(((size_t)b * COUT + o) * H + yy) * W + xx] = s + bias[o];
  }
}

constexpr size_t IN_CONV_MAX_LDS = 144 * 1024;

template <typename T>
int launch_in_conv(const InConvParams& p, hipStream_t s) {
  const size_t npix = (size_t)p.B * p.H * p.W;
  // SDMI_BOUNDARY_SCALAR=1 (tests; read on every call): always take the scalar kernels
  const char* e = getenv("SDMI_BOUNDARY_SCALAR");
  const bool scalar = e && atoi(e) != 0;
  if (!scalar && p.Cout % 64 == 0 && p.Cout <= 512 && (p.Cin == 3 || p.Cin == 4 || p.Cin == 9)) {
    size_t groups = (npix + 15) / 16;
    const unsigned grid = (unsigned)(groups < 1024 ? groups : 1024);
    if (p.Cin == 3) hipLaunchKernelGGL((in_conv_mfma_kernel<T, 3>), dim3(grid), dim3(p.Cout), 0, s, p);
    else if (p.Cin == 4) hipLaunchKernelGGL((in_conv_mfma_kernel<T, 4>), dim3(grid), dim3(p.Cout), 0, s, p);
    else hipLaunchKernelGGL((in_conv_mfma_kernel<T, 9>), dim3(grid), dim3(p.Cout), 0, s, p);
    return SDMI_OK;
  }
  const size_t lds = (size_t)9 * p.Cin * p.Cout * sizeof(float);
  SDMI_REQUIRE(lds <= IN_CONV_MAX_LDS, "in_conv: %d x %d weights do not fit in LDS", p.Cin, p.Cout);
  const int ppb = 256 / (p.Cout / 8);
  SDMI_REQUIRE(ppb >= 1, "in_conv: Cout=%d too wide for the scalar kernel", p.Cout);
  size_t blocks = (npix + ppb - 1) / ppb;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(in_conv_kernel<T>, dim3((unsigned)blocks), dim3(256), lds, s, p);
  return SDMI_OK;
}

template <typename T>
int launch_out_conv(const void* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cin,
                    int Cout, hipStream_t s) {
  const size_t pix = (size_t)B * H * W;
  const char* e = getenv("SDMI_BOUNDARY_SCALAR");
  const bool scalar = e && atoi(e) != 0;
  if (!scalar && Cout >= 1 && Cout <= 16 && (Cin == 128 || Cin == 320)) {
    // as many workgroups as stay resident (9 waves each; registers allow one per CU at Cin = 320, two at 128),
    // a multiple of 8 for the XCD bands
    const size_t groups = (pix + 15) / 16;
    size_t want = Cin == 128 ? 512 : 256;
    if (want > groups) want = groups;
    const unsigned grid = (unsigned)((want + 7) / 8 * 8);
    if (Cin == 128)
      hipLaunchKernelGGL((out_conv_mfma_kernel<T, 128>), dim3(grid), dim3(OUT_CONV_THREADS), 0, s, (const T*)x, w, bias,
                         out, B, H, W, Cout);
    else
      hipLaunchKernelGGL((out_conv_mfma_kernel<T, 320>), dim3(grid), dim3(OUT_CONV_THREADS), 0, s, (const T*)x, w, bias,
                         out, B, H, W, Cout);
    return SDMI_OK;
  }
  dim3 grid((unsigned)((pix + 3) / 4));
  if (Cout == 4)
    hipLaunchKernelGGL((out_conv_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)x, w, bias, out, B, H, W, Cin);
  else if (Cout == 3)
    hipLaunchKernelGGL((out_conv_kernel<T, 3>), grid, dim3(256), 0, s, (const T*)x, w, bias, out, B, H, W, Cin);
  else if (Cout == 8)
    hipLaunchKernelGGL((out_conv_kernel<T, 8>), grid, dim3(256), 0, s, (const T*)x, w, bias, out, B, H, W, Cin);
  else {
    sdmi_set_error("out_conv: Cout=%d unsupported for Cin=%d (3, 4 or 8)", Cout, Cin);
    return SDMI_EINVAL;
  }
  return SDMI_OK;
}

}  // namespace

// Raise the dynamic-LDS limit of the scalar first-convolution kernels once, outside of stream capture.
int boundary_conv_init() {
  static bool done = false;
  if (done) return SDMI_OK;
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)in_conv_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)IN_CONV_MAX_LDS));
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)in_conv_kernel<half_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)IN_CONV_MAX_LDS));
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)in_conv_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)IN_CONV_MAX_LDS));
  done = true;
  return SDMI_OK;
}

int in_conv(const InConvArgs& a, hipStream_t s) {
  InConvParams p;
  p.Cin = 0;
  for (int i = 0; i < 3; ++i) {
    p.src[i] = a.src[i]; p.c[i] = a.src[i] ? a.c[i] : 0; p.Cin += p.c[i];
  }
  SDMI_REQUIRE(a.src[0] && a.w && a.bias && a.out && p.Cin > 0 && p.Cin <= 64, "in_conv: bad arguments");
  SDMI_REQUIRE(p.c[0] > 0 && (p.c[1] > 0 || p.c[2] == 0), "in_conv: sources must be given in order");
  p.w = a.w; p.bias = a.bias; p.out = a.out; p.B = a.B; p.H = a.H; p.W = a.W; p.Cout = a.Cout;
  SDMI_REQUIRE(a.Cout % 8 == 0 && a.Cout <= 2048, "in_conv: Cout=%d must be a multiple of 8, at most 2048", a.Cout);
  SDMI_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0, "in_conv: empty input");
  SDMI_TRY(boundary_conv_init());
  char label[64];
  snprintf(label, sizeof(label), "in_conv B%d %dx%d C%d->%d", a.B, a.H, a.W, p.Cin, a.Cout);
  ProfScope prof(PROF_OTHER, s, 2.0 * 9 * p.Cin * (double)a.Cout * a.B * a.H * a.W,
                 (double)a.B * a.H * a.W * (p.Cin * 4.0 + a.Cout * (double)sdmi_dtype_size(a.dtype)), label);
  SDMI_DISPATCH_DTYPE(a.dtype, T, SDMI_TRY(launch_in_conv<T>(p, s)));
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int out_conv(const void* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cin,
             int Cout, int dtype, hipStream_t s) {
  SDMI_REQUIRE(x && w && bias && out, "out_conv: null pointer");
  SDMI_REQUIRE(Cin % 8 == 0 && B > 0 && H > 0 && W > 0, "out_conv: Cin must be a multiple of 8, sizes positive");
  char label[64];
  snprintf(label, sizeof(label), "out_conv B%d %dx%d C%d->%d", B, H, W, Cin, Cout);
  ProfScope prof(PROF_OTHER, s, 2.0 * 9 * Cin * (double)Cout * B * H * W,
                 (double)B * H * W * (Cin * (double)sdmi_dtype_size(dtype) + Cout * 4.0), label);
  SDMI_DISPATCH_DTYPE(dtype, T, SDMI_TRY(launch_out_conv<T>(x, w, bias, out, B, H, W, Cin, Cout, s)));
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
