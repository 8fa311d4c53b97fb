// VALU issue rates on gfx950: cycles per wave-instruction for v_fma_f32 / v_pk_fma_f32 / ds_read_b128,
// one or two waves per SIMD, many independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND, int NCH>
__global__ __launch_bounds__(512, 1) void k(int iters, long long* out, float* sink) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float r = 0.f;
  long long t0 = __builtin_amdgcn_s_memtime();
  if (KIND == 0) {
    float a[NCH];
    for (int j = 0; j < NCH; ++j) a[j] = lane + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) a[j] = __builtin_fmaf(a[j], 1.0001f, 0.5f);
    }
    for (int j = 0; j < NCH; ++j) r += a[j];
  } else if (KIND == 1) {
    f32x2 a[NCH];
    for (int j = 0; j < NCH; ++j) a[j] = f32x2{(float)lane, (float)j};
    const f32x2 m = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) a[j] = __builtin_elementwise_fma(a[j], m, c);
    }
    for (int j = 0; j < NCH; ++j) r += a[j][0] + a[j][1];
  } else {
    float4 acc[NCH];
    for (int j = 0; j < NCH; ++j) acc[j] = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(&lds[((j * 64 + lane) * 4 + (i & 1) * 4096) & 8191]);
        acc[j].x += v.x;
      }
    }
    for (int j = 0; j < NCH; ++j) r += acc[j].x;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (r == 12345.678f) sink[0] = r;
}
template <int KIND, int NCH>
void run(long long* d, float* s, int threads) {
  const int iters = 1000;
  hipLaunchKernelGGL((k<KIND, NCH>), dim3(256), dim3(threads), 0, 0, iters, d, s);
  hipLaunchKernelGGL((k<KIND, NCH>), dim3(256), dim3(threads), 0, 0, iters, d, s);
  long long h;
  (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const char* nm[] = {"v_fma_f32", "v_pk_fma_f32", "ds_read_b128+add"};
  printf("%-18s chains %2d waves/SIMD %d: %.2f cycles per instruction per wave\n", nm[KIND], NCH, threads / 256,
         (double)h / (iters * NCH));
}
int main() {
  long long* d; float* s;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 4);
  run<0, 16>(d, s, 256); run<0, 32>(d, s, 256); run<0, 16>(d, s, 512);
  run<1, 16>(d, s, 256); run<1, 32>(d, s, 256); run<1, 16>(d, s, 512);
  run<2, 8>(d, s, 256); run<2, 16>(d, s, 256); run<2, 8>(d, s, 512);
  return 0;
}
