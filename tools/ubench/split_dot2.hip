// Exact 3-way bf16 split of fp32 pairs: the shift/and/subtract form used so far against a form whose residuals come
// from v_dot2c_f32_bf16 (r = x - piece in ONE instruction: dot2((piece_lo, piece_hi), (-1, 0), x)).  Checks that both
// give bit-identical pieces on random and special inputs, and times them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ void split_ref(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk(s0, s1);
}
__device__ __forceinline__ float sub_lo(float x, unsigned p) {   // x - bf16(low half of p)
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, 0x0000BF80u), x, false);
}
__device__ __forceinline__ float sub_hi(float x, unsigned p) {   // x - bf16(high half of p)
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, 0xBF800000u), x, false);
}
__device__ __forceinline__ void split_dot(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk(x0, x1);
  const float r0 = sub_lo(x0, h), r1 = sub_hi(x1, h);
  m = cvt_pk(r0, r1);
  const float s0 = sub_lo(r0, m), s1 = sub_hi(r1, m);
  l = cvt_pk(s0, s1);
}
__global__ void check(const float* x, int n, unsigned* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  unsigned h, m, l, h2, m2, l2;
  split_ref(x[2 * i], x[2 * i + 1], h, m, l);
  split_dot(x[2 * i], x[2 * i + 1], h2, m2, l2);
  if (h != h2 || m != m2 || l != l2) atomicAdd(bad, 1u);
}
template <int KIND>
__global__ __launch_bounds__(256) void rate(int iters, float* sink, long long* out) {
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.37f + j;
  unsigned acc = 0;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      unsigned h, m, l;
      if (KIND == 0) split_ref(v[j], v[j + 1], h, m, l); else split_dot(v[j], v[j + 1], h, m, l);
      acc ^= h ^ m ^ l;
      v[j] += 1.0f; v[j + 1] += 0.5f;
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (acc == 0x12345u) sink[0] = acc;
}
int main() {
  const int n = 1 << 22;
  float* hx = (float*)malloc(n * 4);
  srand(7);
  for (int i = 0; i < n; ++i) {
    unsigned b = ((unsigned)rand() << 16) ^ (unsigned)rand();
    float f; memcpy(&f, &b, 4);
    if (!std::isfinite(f)) f = 1.0f / (1 + i % 97);
    hx[i] = f;
  }
  hx[0] = 0.f; hx[1] = -0.f; hx[2] = 1e-38f; hx[3] = 3e38f; hx[4] = 1.0f; hx[5] = 1.00390625f; hx[6] = -255.99998f; hx[7] = 1e-45f;
  float* dx; unsigned* bad; float* sink; long long* out;
  hipMalloc(&dx, n * 4); hipMalloc(&bad, 4); hipMalloc(&sink, 4); hipMalloc(&out, 8);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(check, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, bad);
  unsigned hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("pairs whose pieces differ between the two forms: %u of %d (random bit patterns incl. denormals)\n", hb, n / 2);
  for (int kind = 0; kind < 2; ++kind) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, iters, sink, out); else hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, iters, sink, out);
    hipEventRecord(e0);
    if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, iters, sink, out); else hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, iters, sink, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms for %d pair-splits per lane\n", kind ? "dot2 form" : "shift/and form", ms, iters * 4);
  }
  return 0;
}
