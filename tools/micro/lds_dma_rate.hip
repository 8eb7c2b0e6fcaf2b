// LDS-DMA issue rate on one CU: global_load_lds_dwordx4 (64-bit address per lane) against
// buffer_load_dwordx4 ... lds (descriptor + 32-bit offset per lane), same access pattern as the GEMM loader
// (a wave instruction moves 8 rows x 128 bytes), data resident in L2.  One 512-thread workgroup per CU.
// Build: hipcc --offload-arch=gfx950 -O3 lds_dma_rate.hip -o lds_dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ROW_BYTES = 640;          // row pitch of the source (a 320-channel fp16 tensor)
constexpr int ROWS = 4096;              // rows a workgroup cycles through (2.6 MB per workgroup's view, shared: L2-resident)

template <int MODE>
__global__ __launch_bounds__(512) void dma_kernel(const char* src, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lrow = lane >> 3, chunk = lane & 7;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, ROWS * ROW_BYTES, 0x00020000);
  char* dst = smem + wid * 1024;
  unsigned off = (unsigned)((wid * 8 + lrow) * ROW_BYTES + chunk * 16);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned o = (off + u * 64 * ROW_BYTES) % (unsigned)(ROWS * ROW_BYTES - 1024);
      if (MODE == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o),
                                         (__attribute__((address_space(3))) void*)(dst + u * 8192), 16, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + u * 8192), 16, o, 0, 0, 0);
    }
    off += 128;
    if (off >= (unsigned)(ROWS * ROW_BYTES - 64 * 8 * ROW_BYTES)) off -= (unsigned)(ROWS * ROW_BYTES - 64 * 8 * ROW_BYTES);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (reinterpret_cast<unsigned*>(smem)[tid] == 0x12345678u) *sink = 1;
}

template <int MODE>
static int run(const char* name, const char* src, unsigned* sink) {
  const int iters = 2000, blocks = 256;
  CK(hipFuncSetAttribute((const void*)dma_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(dma_kernel<MODE>, dim3(blocks), dim3(512), 65536, 0, src, 50, sink);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(dma_kernel<MODE>, dim3(blocks), dim3(512), 65536, 0, src, iters, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes_per_cu = (double)iters * 8 * 8 * 1024;
  printf("%-28s %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip  (%.1f ns per wave instruction per CU)\n", name, ms,
         bytes_per_cu / ms / 1e6, bytes_per_cu * blocks / ms / 1e9, ms * 1e6 / (iters * 64.0));
  return 0;
}

int main() {
  char* src; unsigned* sink;
  CK(hipMalloc(&src, (size_t)ROWS * ROW_BYTES)); CK(hipMemset(src, 1, (size_t)ROWS * ROW_BYTES));
  CK(hipMalloc(&sink, 4));
  for (int r = 0; r < 2; ++r) {
    if (run<0>("global_load_lds_dwordx4", src, sink)) return 1;
    if (run<1>("buffer_load_dwordx4 lds", src, sink)) return 1;
  }
  return 0;
}
