#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, float* o, int nbytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  for (int i = 0; i < 8; ++i) {
    i4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, i * 1024, 0);
    f4 f = __builtin_bit_cast(f4, v);
    float* d = o + (i * 64 + threadIdx.x) * 4;
    d[0] = f.x; d[1] = f.y; d[2] = f.z; d[3] = f.w;
  }
}
int main() {
  const int n = 8 * 64 * 4;
  std::vector<float> h(n), out(n);
  for (int i = 0; i < n; ++i) h[i] = i;
  float *d, *o;
  (void)hipMalloc(&d, n * 4); (void)hipMalloc(&o, n * 4);
  (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n * 4);
  (void)hipMemcpy(out.data(), o, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) if (out[i] != h[i]) { if (bad < 5) printf("mismatch %d: %f vs %f\n", i, out[i], h[i]); ++bad; }
  printf("bad = %d of %d\n", bad, n);
  return 0;
}
