// What does ds_read_b64_tr_b16 deliver?  LDS holds sm[i] = i (16-bit); every lane passes its own address.
// case 0: lane l -> element 4*l (lane-linear 8-B chunks);  case 1: 16-lane group g, lane i in group ->
// row 4*g + (i>>2) of a [16][STRIDE] image, columns 4*(i&3)..+3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(s16x4* o, int mode, int stride) {
  __shared__ short sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  const int off = mode == 0 ? 4 * l : (4 * g + (i >> 2)) * stride + 4 * (i & 3);
  o[l] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + off));
}
int main() {
  s16x4* d; hipMalloc(&d, 64 * 8);
  short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode, 40);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
