// Per-kernel floor of a chain of dependent launches replayed from a HIP graph on one stream:
//   empty kernels, kernels that rewrite a small buffer (dirty lines to write back at every kernel boundary),
//   and the same chain launched eagerly.  Build: hipcc --offload-arch=gfx950 -O3 graph_floor.hip -o graph_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel() {}
__global__ void touch_kernel(float4* buf, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = buf[i];
    v.x += 1.f;
    buf[i] = v;
  }
}

template <typename F>
static int run(const char* name, int n, hipStream_t s, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) launch();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  const int reps = 20;
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  for (int r = 0; r < 3; ++r) { for (int i = 0; i < n; ++i) launch(); }
  CK(hipStreamSynchronize(s));
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { for (int i = 0; i < n; ++i) launch(); }
  CK(hipStreamSynchronize(s));
  double us_e = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("%-34s graph %7.2f us per kernel | eager %7.2f us per kernel\n", name, us / n, us_e / n);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int n = 400;
  float4* buf; const size_t bytes = 64u << 20; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  if (run("empty, 1 block", n, s, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); })) return 1;
  if (run("empty, 1024 blocks x 256", n, s, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, s); })) return 1;
  for (size_t mb : {1, 4, 16, 64}) {
    char name[64]; snprintf(name, sizeof(name), "read+write %zu MiB, 1024 x 256", mb);
    const size_t nvec = (mb << 20) / 16;
    if (run(name, n, s, [&] { hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, s, buf, nvec); })) return 1;
  }
  return 0;
}
