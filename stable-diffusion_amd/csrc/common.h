// Shared declarations for the gfx950 kernels and the host-side executors of libsdmi.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <string>

// ---- element types ---------------------------------------------------------------------------
enum { SDMI_F32 = 0, SDMI_F16 = 1, SDMI_BF16 = 2 };

typedef _Float16 half_t;
typedef __bf16 bf16_t;

static inline size_t sdmi_dtype_size(int dt) { return dt == SDMI_F32 ? 4 : 2; }

// ---- error handling: no exception crosses the C boundary ---------------------------------------
// error codes (SDMI_OK, SDMI_EINVAL, ...) come from the public header
#include "../../include/sdmi.h"

void sdmi_set_error(const char* fmt, ...);

#define SDMI_CHECK_HIP(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      sdmi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return SDMI_EHIP;                                                                   \
    }                                                                                     \
  } while (0)

#define SDMI_REQUIRE(cond, ...)                                                           \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      sdmi_set_error(__VA_ARGS__);                                                        \
      return SDMI_EINVAL;                                                                 \
    }                                                                                     \
  } while (0)

#define SDMI_TRY(expr)                                                                    \
  do {                                                                                    \
    int _r = (expr);                                                                      \
    if (_r != SDMI_OK) return _r;                                                         \
  } while (0)

// ---- device helpers ----------------------------------------------------------------------------
#ifdef __HIPCC__
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

template <typename T> struct VecOf;   // 16-byte vector of T
template <> struct VecOf<float>  { typedef float  type __attribute__((ext_vector_type(4))); enum { N = 4 }; };
template <> struct VecOf<half_t> { typedef half_t type __attribute__((ext_vector_type(8))); enum { N = 8 }; };
template <> struct VecOf<bf16_t> { typedef bf16_t type __attribute__((ext_vector_type(8))); enum { N = 8 }; };

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// val * GELU(gate) for the GEGLU epilogues.  16-bit outputs: two values per instruction on the packed fp32 pipe,
// erf(g / sqrt 2) = g P(g^2) / Q(g^2) with cubic P and Q (fitted on |g| <= 3.3 sqrt 2, where erf has reached
// 1 - 3e-6; |error| <= 2.7e-6, the result stays inside (-1, 1)): one reciprocal and no exponential per value, about
// two thirds of the vector instructions of the series form with an exponential (Abramowitz & Stegun 7.1.26) that it
// replaced (the GELU of a GEGLU projection at M 65536, N 2560 runs with the matrix cores idle: 45 of its 178 us).
typedef float sdmi_f32x2 __attribute__((ext_vector_type(2)));
typedef float sdmi_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sdmi_f32x2 geglu2_fast(sdmi_f32x2 val, sdmi_f32x2 g) {
  sdmi_f32x2 gc;
  gc.x = __builtin_amdgcn_fmed3f(g.x, -4.666904755831213f, 4.666904755831213f);
  gc.y = __builtin_amdgcn_fmed3f(g.y, -4.666904755831213f, 4.666904755831213f);
  const sdmi_f32x2 t = gc * gc;
  sdmi_f32x2 p = t * (sdmi_f32x2)(6.622132928e-05f) + (sdmi_f32x2)(7.633776298e-03f);
  p = p * t + (sdmi_f32x2)(5.445629777e-02f);
  p = p * t + (sdmi_f32x2)(7.978868118e-01f);
  sdmi_f32x2 q = t * (sdmi_f32x2)(1.161323832e-03f) + (sdmi_f32x2)(2.368698296e-02f);
  q = q * t + (sdmi_f32x2)(2.349381700e-01f);
  q = q * t + (sdmi_f32x2)(1.0f);
  sdmi_f32x2 r;
  r.x = __builtin_amdgcn_rcpf(q.x);
  r.y = __builtin_amdgcn_rcpf(q.y);
  const sdmi_f32x2 e = (p * gc) * r;                       // erf(g / sqrt 2)
  const sdmi_f32x2 hv = (val * g) * (sdmi_f32x2)(0.5f);
  return hv * e + hv;                                      // val * 0.5 g (1 + erf)
}
template <typename T> __device__ __forceinline__ sdmi_f32x4 geglu4_for(sdmi_f32x4 val, sdmi_f32x4 gate) {
  if constexpr (sizeof(T) == 2) {
    const sdmi_f32x2 lo = geglu2_fast(val.xy, gate.xy), hi = geglu2_fast(val.zw, gate.zw);
    return (sdmi_f32x4){lo.x, lo.y, hi.x, hi.y};
  } else {
    return (sdmi_f32x4){val.x * gelu_erf_f(gate.x), val.y * gelu_erf_f(gate.y), val.z * gelu_erf_f(gate.z),
                        val.w * gelu_erf_f(gate.w)};
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#endif

// Dispatch a templated launcher on the runtime element type.
#define SDMI_DISPATCH_DTYPE(dt, T, ...)                                  \
  switch (dt) {                                                          \
    case SDMI_F32:  { typedef float  T; __VA_ARGS__; break; }            \
    case SDMI_F16:  { typedef half_t T; __VA_ARGS__; break; }            \
    case SDMI_BF16: { typedef bf16_t T; __VA_ARGS__; break; }            \
    default: sdmi_set_error("bad dtype %d", (int)(dt)); return SDMI_EINVAL; \
  }
