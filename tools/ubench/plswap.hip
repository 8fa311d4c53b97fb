// semantics of v_permlane32_swap_b32 (gfx950) through __builtin_amdgcn_permlane32_swap(a, b, false, false)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 512); unsigned h[128];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("a = lane, b = lane + 100;  r = permlane32_swap(a, b)\n");
  for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r[0] = %3u  r[1] = %3u\n", l, h[l], h[64 + l]);
  return 0;
}
