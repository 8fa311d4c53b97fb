// ds_read_b64_tr_b16 rate for lane addresses  (i>>2)*S + (i&3)*8 + (g&1)*C + (g>>1)*H   (g = lane>>4, i = lane&15)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDSP(T) __attribute__((address_space(3))) T
__global__ __launch_bounds__(512, 1) void k(int iters, int S, int C, int H, int* sink) {
  extern __shared__ char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  for (int t = threadIdx.x; t < 65536 / 4; t += 512) reinterpret_cast<int*>(sm)[t] = t;
  __syncthreads();
  const char* p = sm + (i >> 2) * S + (i & 3) * 8 + (g & 1) * C + (g >> 1) * H + 64 * (wave & 3);
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const unsigned a0 = (unsigned)(size_t)(LDSP(char)*)p;
  u2 v[8];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(v[r]) : "v"(a0 + (r & 3) * 8192));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (v[0][0] == 12345u && v[7][1] == 999u) sink[0] = v[3][0];
}
int main() {
  int* s; (void)hipMalloc(&s, 4);
  const int iters = 20000;
  const int cases[][3] = {{32, 128, 256}, {2176, 32, 1088}, {1056, 32, 4224}, {272, 32, 1088}, {544, 32, 2176}, {1088, 32, 544}, {264 * 8, 32, 264 * 4}, {1088, 64, 544}, {1088, 128, 544},
                          {272, 32, 1088}, {272, 128, 1088}, {272, 32, 128}, {128, 32, 64}, {256, 32, 64}, {256, 32, 128}, {64, 32, 256},
                          {64, 32, 1024}, {272, 64, 1088}, {528, 32, 2112}, {1056, 32, 528}, {288, 32, 1152}, {288, 32, 144}, {320, 32, 160}, {160, 32, 640}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, iters, c[0], c[1], c[2], s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, iters, c[0], c[1], c[2], s);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("S %5d C %4d H %5d : %.2f ns per wave read\n", c[0], c[1], c[2], ms * 1e6 / (iters * 8.0));
  }
  return 0;
}
