// Does an fp32 MFMA stream in one wave overlap with a VALU stream in the partner wave of the same SIMD?
// 512-thread workgroup = 2 waves per SIMD; waves 0-3 run VALU (v_fma_f32 / v_pk_fma_f32 / LDS reads),
// waves 4-7 run MFMAs.  One workgroup per CU, timed with s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512, 1) void k(int mode_valu, int mode_mfma, int iters, long long* out, float* sink, int prio) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  if (prio == 1 && wave < 4) __builtin_amdgcn_s_setprio(3);
  if (prio == 2 && wave >= 4) __builtin_amdgcn_s_setprio(3);
  long long t0 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  if (wave < 4) {
    if (mode_valu == 1) {           // 8 independent v_fma_f32 chains
      float a[8];
      for (int j = 0; j < 8; ++j) a[j] = lane + j;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __builtin_fmaf(a[j], 1.0001f, 0.5f);
      }
      for (int j = 0; j < 8; ++j) r += a[j];
    } else if (mode_valu == 2) {    // 8 independent v_pk_fma_f32 chains
      f32x2 a[8];
      for (int j = 0; j < 8; ++j) a[j] = f32x2{(float)lane, (float)j};
      const f32x2 m = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __builtin_elementwise_fma(a[j], m, c);
      }
      for (int j = 0; j < 8; ++j) r += a[j][0] + a[j][1];
    } else if (mode_valu == 3) {    // ds_read_b128 stream
      float4 acc = {0, 0, 0, 0};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float4 v = *reinterpret_cast<float4*>(&lds[((i * 8 + j) * 64 + lane * 4) & 4095]);
          acc.x += v.x;
        }
      }
      r = acc.x;
    }
  } else {
    if (mode_mfma == 1) {           // 16x16x4 f32, 4 independent accumulators
      f32x4 c[4] = {};
      float a = lane, b = lane * 0.5f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j & 3], 0, 0, 0);
      }
      r = c[0][0] + c[1][0] + c[2][0] + c[3][0];
    } else if (mode_mfma == 2) {    // 32x32x2 f32, 2 independent accumulators
      f32x16 c[2] = {};
      float a = lane, b = lane * 0.5f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j & 1], 0, 0, 0);
      }
      r = c[0][0] + c[1][0];
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
  if (r == 12345.678f) sink[0] = r;
}

int main() {
  long long* d; float* s;
  hipMalloc(&d, 64); hipMalloc(&s, 4);
  const int iters = 2000;
  for (int prio = 0; prio <= 2; ++prio)
  for (int mv = 1; mv <= 3; ++mv)
    for (int mm = 1; mm <= 1; ++mm) {
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mv, mm, iters, d, s, prio);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mv, mm, iters, d, s, prio);
      long long h[8];
      hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      printf("prio %d valu_mode %d mfma_mode %d: VALU wave %.1f cyc/instr   MFMA wave %.1f cyc/instr\n", prio, mv, mm,
             mv ? (double)h[0] / (iters * 8) : 0.0, mm ? (double)h[4] / (iters * 8) : 0.0);
    }
  return 0;
}
