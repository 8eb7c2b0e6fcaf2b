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
// GELU for 16-bit outputs: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// output rounding of fp16 / bf16) -- one rcp, one exp2 and a degree-5 polynomial, no branches.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float erf_abs = fmaf(-poly * t, e, 1.0f);          // erf(|x| / sqrt 2)
  const float half_x = 0.5f * x;
  return fmaf(fabsf(half_x), erf_abs, half_x);               // 0.5 x (1 + sign(x) erf|.|)
}
template <typename T> __device__ __forceinline__ float gelu_for(float x) {
  if constexpr (sizeof(T) == 2) return gelu_erf_fast(x);
  else return gelu_erf_f(x);
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
