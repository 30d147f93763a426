// prints what v_permlane16_swap / v_permlane32_swap do to (a, b) = (lane, 100 + lane)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  const auto p = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  const auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = p[0]; o[64 + threadIdx.x] = p[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h)); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"p16[0]", "p16[1]", "p32[0]", "p32[1]"};
  for (int v = 0; v < 4; ++v) { printf("%s:", nm[v]); for (int i = 0; i < 64; i += 8) printf(" %u", h[64 * v + i]); printf("\n"); }
  return 0;
}
