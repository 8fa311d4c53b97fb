// What v_mfma_f32_32x32x16_f16 does with the inputs a two-piece fp16 split would feed it:
//  (1) subnormal fp16 inputs: honoured or flushed?   A = 1.0, B = 2^-20 (subnormal) -> C = 2^-20 or 0
//  (2) a product of two 11-bit significands is exact in the fp32 accumulator: (1+2^-10)^2 = 1 + 2^-9 + 2^-20
//  (3) rate against the bf16 instruction of the same shape (all CUs busy, 2 waves per SIMD, back to back)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float* o, float av, float bv) {
  f16x8 a, b;
  const int l = threadIdx.x;
  for (int i = 0; i < 8; ++i) {
    const int kslot = 8 * (l >> 5) + i;
    a[i] = kslot == 0 ? (_Float16)av : (_Float16)0.f;
    b[i] = kslot == 0 ? (_Float16)bv : (_Float16)0.f;
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  o[l] = c[0];
}

template <int F16>
__global__ __launch_bounds__(512) void rate(float* o, int steps) {
  f16x8 a, b; bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * threadIdx.x); b[i] = (_Float16)1.f; ab[i] = (__bf16)(0.001f * threadIdx.x); bb[i] = (__bf16)1.f; }
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
  for (int s = 0; s < steps; ++s) {
    if (F16) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
    }
  }
  o[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  struct { float a, b; const char* name; } cases[] = {
      {1.0f, ldexpf(1.f, -20), "1.0 x 2^-20 (subnormal B)"},
      {ldexpf(1.f, -20), 1.0f, "2^-20 x 1.0 (subnormal A)"},
      {ldexpf(1.f, -24), 1.0f, "2^-24 x 1.0 (smallest subnormal A)"},
      {ldexpf(1.f, -20), ldexpf(1.f, -20), "2^-20 x 2^-20 (both subnormal)"},
      {ldexpf(3.f, -16), 1024.0f, "3*2^-16 x 1024 (subnormal with 2 bits)"},
      {1.0f + ldexpf(1.f, -10), 1.0f + ldexpf(1.f, -10), "(1+2^-10)^2"}};
  for (auto& cs : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, cs.a, cs.b);
    float h[64]; (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%-40s C = %.9g   (exact %.9g)\n", cs.name, h[0], (double)cs.a * (double)cs.b);
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int steps = 20000;
  for (int f16 = 0; f16 < 2; ++f16)
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      if (f16) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(512), 0, 0, d, steps);
      else hipLaunchKernelGGL(rate<0>, dim3(256), dim3(512), 0, 0, d, steps);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      // per SIMD: 2 waves x steps x 4 MFMAs
      printf("%s 32x32x16: %.2f ns per MFMA per SIMD, %.0f TFLOP/s chip\n", f16 ? "f16 " : "bf16", ms * 1e6 / (2.0 * steps * 4),
             256.0 * 8 * steps * 4 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) * 1e-12);
    }
  return 0;
}
