// Bank conflicts of ds_read_b64_tr_b16 for the dW-GEMM operand reads: a 16-lane group (g = lane>>4, i = lane&15) reads
// rows base + RS*(i>>2) (+ RG*(g>>1)), 8-B chunk (i&3) of the 32-B column block (g&1) of an image with row stride ROWB.
// RS = rows between the four edges of a quad, RG = rows between the two k-halves of a wave read.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDSP(T) __attribute__((address_space(3))) T
template <int ROWB, int RS, int RG>
__global__ __launch_bounds__(512, 1) void k(int iters, long long* out, int* sink) {
  extern __shared__ char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  for (int t = threadIdx.x; t < 64 * ROWB / 4; t += 512) reinterpret_cast<int*>(sm)[t] = t;
  __syncthreads();
  const char* p = sm + (RS * (i >> 2) + RG * (g >> 1)) * ROWB + (16 * (g & 1) + 4 * (i & 3)) * 2 + 64 * (wave & 3);
  s16x4 acc = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP(s16x4)*)(p + (r & 1) * ROWB + (r >> 1) * 16 * ROWB));
      acc += v;
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (acc[0] == 12345 && acc[1] == 999) sink[0] = acc[2];
}
// linear addresses: MODE 0 ds_read_b64_tr_b16 at lane*8, 1 plain ds_read_b64 at lane*8, 2 ds_read_b128 at lane*16
template <int MODE>
__global__ __launch_bounds__(512, 1) void klin(int iters, long long* out, int* sink) {
  extern __shared__ char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = threadIdx.x; t < 16384 / 4; t += 512) reinterpret_cast<int*>(sm)[t] = t;
  __syncthreads();
  const char* p = sm + wave * 1024 + lane * (MODE == 2 ? 16 : 8);
  int acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE == 0) { const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP(s16x4)*)(p + (r & 1) * 512)); acc += v[0] + v[3]; }
      else if (MODE == 1) { typedef int i2 __attribute__((ext_vector_type(2))); const i2 v = *reinterpret_cast<const volatile i2*>(p + (r & 1) * 512); acc += v[0] + v[1]; }
      else { typedef int i4 __attribute__((ext_vector_type(4))); const i4 v = *reinterpret_cast<const volatile i4*>(p); acc += v[0] + v[3]; }
    }
  }
  if (acc == 12345) sink[0] = acc;
}
template <int MODE>
void runlin(long long* d, int* s, const char* name) {
  const int iters = 20000;
  hipLaunchKernelGGL((klin<MODE>), dim3(256), dim3(512), 16384, 0, iters, d, s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((klin<MODE>), dim3(256), dim3(512), 16384, 0, iters, d, s);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %.3f ms  = %.2f ns per wave read (8 waves per CU reading)\n", name, ms, ms * 1e6 / (iters * 8.0));
}
template <int ROWB, int RS, int RG>
void run(long long* d, int* s, const char* name) {
  const int iters = 20000;
  hipLaunchKernelGGL((k<ROWB, RS, RG>), dim3(256), dim3(512), 64 * ROWB, 0, iters, d, s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ROWB, RS, RG>), dim3(256), dim3(512), 64 * ROWB, 0, iters, d, s);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %.3f ms  = %.2f ns per wave read (8 waves per CU reading)\n", name, ms, ms * 1e6 / (iters * 8.0));
}
int main() {
  long long* d; int* s;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 4);
  run<272, 2, 8>(d, s, "G image now: stride 272, quad rows 2 apart, halves 8");
  run<272, 4, 2>(d, s, "G image new: stride 272, quad rows 4 apart, halves 2");
  run<264, 8, 4>(d, s, "Z image: stride 264, quad rows 8 apart, halves 4");
  run<528, 2, 8>(d, s, "dw8 images now: stride 528, quad rows 2 apart");
  run<528, 4, 2>(d, s, "dw8 images new: stride 528, quad rows 4 apart");
  run<272, 1, 4>(d, s, "stride 272, plain rows (no permutation)");
  runlin<0>(d, s, "linear: ds_read_b64_tr_b16 at lane*8");
  runlin<1>(d, s, "linear: ds_read_b64 at lane*8");
  runlin<2>(d, s, "linear: ds_read_b128 at lane*16");
  return 0;
}
