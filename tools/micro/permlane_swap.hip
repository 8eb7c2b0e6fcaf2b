#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + threadIdx.x] = s[0]; out[192 + threadIdx.x] = s[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int j = 0; j < 4; ++j) { for (int i = 0; i < 64; i += 4) printf("%u ", h[j * 64 + i]); printf("\n"); }
  return 0;
}
