// How does v_mfma_f32_32x32x16_bf16 round its accumulate?  C0 = 1.0, every MFMA adds ONE product x = 0.75 ulp(1.0)
// (A = 1.0 in k-slot 0 of row 0.., B = x in k-slot 0): round-to-nearest climbs by 1 ulp per step, truncation stays at 1.0.
// Second experiment: x = 0.25 ulp: nearest stays, "round up" would climb.  Third: 16 products of 0.75/16 ulp each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* o, float x, int nk, int steps, float c0) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.f; b[i] = (__bf16)0.f; }
  const int l = threadIdx.x;
  for (int i = 0; i < 8; ++i) {
    const int kslot = 8 * (l >> 5) + i;
    if (kslot < nk) { a[i] = (__bf16)1.0f; b[i] = (__bf16)x; }
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = c0;
  for (int s = 0; s < steps; ++s) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  o[l] = c[0];
}
int main() {
  float* d; (void)hipMalloc(&d, 256);
  const float ulp = ldexpf(1.f, -23);
  struct { float x; int nk; const char* name; } cases[] = {
      {0.75f * ulp, 1, "1 product of 0.75 ulp"}, {0.25f * ulp, 1, "1 product of 0.25 ulp"},
      {0.75f * ulp / 16, 16, "16 products of 0.75/16 ulp"}, {0.5f * ulp, 1, "1 product of 0.5 ulp (tie)"},
      {-0.75f * ulp, 1, "1 product of -0.75 ulp"}, {-0.25f * ulp, 1, "1 product of -0.25 ulp"}};
  for (auto& cs : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, cs.x, cs.nk, 100, 1.0f);
    float h[64]; (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%-32s after 100 steps: C = 1 + %.3f ulp   (exact %.3f ulp)\n", cs.name, (h[0] - 1.0f) / ulp, 100.0 * cs.x * (cs.nk > 1 ? cs.nk : 1) / ulp);
  }
  return 0;
}
