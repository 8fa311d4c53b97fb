// Does a partner wave's VALU stream progress beside a bf16 MFMA stream when the MFMA wave leaves issue gaps?
// 512 threads = 2 waves per SIMD: waves 4-7 run v_mfma_f32_32x32x16_bf16 (two independent accumulators) with
// GAP x `s_nop 7` behind every MFMA, waves 0-3 run 8 independent v_fma_f32 chains.  Also: same-wave interleave of NV
// v_fma_f32 behind every MFMA.  Times from s_memtime (constant 100 MHz clock) scaled by the measured wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int GAP, int NV, int VALU_ON, int MFMA_ON>
__global__ __launch_bounds__(512, 1) void k(int iters, long long* out, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float r = 0.f;
  long long t0 = clock64();
  if (wave < 4) {
    if (VALU_ON) {
      float a[8];
      for (int j = 0; j < 8; ++j) a[j] = lane + j;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = __builtin_fmaf(a[j], 1.0001f, 0.5f);
      }
      for (int j = 0; j < 8; ++j) r += a[j];
    }
  } else if (MFMA_ON) {
    f32x16 c[2] = {};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)(float)(lane * 0.5f); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = lane + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j & 1], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < GAP; ++g) { asm volatile("s_nop 7"); }
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    r = c[0][0] + c[1][0];
    for (int j = 0; j < 8; ++j) r += v[j];
  }
  long long t1 = clock64();
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
  if (r == 12345.678f) sink[0] = r;
}
template <int GAP, int NV, int VALU_ON, int MFMA_ON>
void run(long long* d, float* s) {
  const int iters = 4000;
  hipLaunchKernelGGL((k<GAP, NV, VALU_ON, MFMA_ON>), dim3(256), dim3(512), 0, 0, iters, d, s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<GAP, NV, VALU_ON, MFMA_ON>), dim3(256), dim3(512), 0, 0, iters, d, s);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[8];
  (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("gap %d nv %2d valu %d mfma %d: kernel %.3f ms | VALU wave %.2f ns/instr  MFMA wave %.2f ns/MFMA (clock64 ticks: valu %lld mfma %lld)\n",
         GAP, NV, VALU_ON, MFMA_ON, ms, VALU_ON ? ms * 1e6 * ((double)h[0] / (double)(h[0] > h[4] ? h[0] : h[4])) / (iters * 32.0) : 0.0,
         MFMA_ON ? ms * 1e6 * ((double)h[4] / (double)(h[0] > h[4] ? h[0] : h[4])) / (iters * 4.0) : 0.0, h[0], h[4]);
}
int main() {
  long long* d; float* s;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 4);
  run<0, 0, 1, 0>(d, s);     // VALU alone
  run<0, 0, 0, 1>(d, s);     // MFMA alone, back to back
  run<0, 0, 1, 1>(d, s);     // both, MFMA back to back
  run<1, 0, 1, 1>(d, s); run<2, 0, 1, 1>(d, s); run<3, 0, 1, 1>(d, s); run<4, 0, 1, 1>(d, s);
  run<3, 0, 0, 1>(d, s); run<4, 0, 0, 1>(d, s);
  run<0, 2, 0, 1>(d, s); run<0, 4, 0, 1>(d, s); run<0, 6, 0, 1>(d, s); run<0, 8, 0, 1>(d, s);   // same-wave interleave
  return 0;
}
