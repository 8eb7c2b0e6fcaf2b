// Does a SIMD overlap MFMA work of one wave with VALU work of another?  (gfx950 microbenchmark)
//   mode 0: waves 0-3 issue MFMAs, waves 4-7 idle      mode 1: waves 0-3 idle, waves 4-7 issue VALU
//   mode 2: both at once (unsynchronised)               mode 3: both, every wave alternates MFMA and VALU blocks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma_block(f32x16 (&acc)[4], h8 a, h8 b) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
}
__device__ __forceinline__ void valu_block(float (&v)[8], float c) {
#pragma unroll
  for (int i = 0; i < 64; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], c, 0.5f);
}
__device__ __forceinline__ void exp_block(float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]);
}

__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, float c, long long* cyc) {
  const int wid = threadIdx.x >> 6;
  const long long t0 = __builtin_readcyclecounter();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f); b[i] = (_Float16)(i * 0.1f); }
  const bool mf = (mode == 0 || mode == 2) ? wid < 4 : false;
  const bool va = (mode == 1 || mode == 2) ? wid >= 4 : false;
  const bool ex = (mode == 4) ? wid >= 4 : (mode == 5 ? wid >= 4 : false);
  const bool mf5 = mode == 5 && wid < 4;
  if (mode == 3) {
    for (int it = 0; it < iters; ++it) { mfma_block(acc, a, b); valu_block(v, c); }
  } else if (mf || mf5) {
    for (int it = 0; it < iters; ++it) mfma_block(acc, a, b);
  } else if (va) {
    for (int it = 0; it < iters; ++it) valu_block(v, c);
  } else if (ex) {
    for (int it = 0; it < iters; ++it) exp_block(v);
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  const long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wid] = t1 - t0;     // shader-clock cycles of this wave
}

int main() {
  float* out;
  const int blocks = 256;
  if (hipMalloc(&out, blocks * 512 * 4) != hipSuccess) return 1;
  long long* cyc; long long hc[8];
  if (hipMalloc(&cyc, 64) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char* names[] = {"mfma only (4 waves)", "valu only (4 waves)", "mfma waves + valu waves", "each wave alternates (8 waves)",
                         "exp only (4 waves)", "mfma waves + exp waves"};
  for (int mode = 0; mode < 6; ++mode) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, mode, 100, out, 0.999f, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, mode, iters, out, 0.999f, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
    printf("mode %d %-34s %8.3f ms  -> %6.1f ns per iteration; cycles/iter wave0 %.1f wave4 %.1f  (%.2f GHz)\n", mode,
           names[mode], ms, ms * 1e6 / iters, (double)hc[0] / iters, (double)hc[4] / iters,
           (double)(hc[0] > hc[4] ? hc[0] : hc[4]) / (ms * 1e6));
  }
  return 0;
}
