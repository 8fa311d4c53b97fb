// LDS store THROUGHPUT by workgroup size and stores in flight per wait (round 5): how fast can the edge backward's piece
// images be written?  One workgroup per CU; every wave writes `nw` stores, then s_waitcnt lgkmcnt(0); reports bytes per ns per CU.
//   pat 0  ds_write_b64, G-image map (hx_prow_g, stride 272)      pat 1  ds_write_b64 contiguous
//   pat 2  ds_write_b128 contiguous                                pat 3  ds_write_b32 contiguous
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ int prow_g(int e) { const int hi = e >> 4, a = (e >> 2) & 3, b = e & 3; return 16 * hi + 4 * b + a; }
template <int NW, int PAT>
__global__ __launch_bounds__(1024) void k(int iters, int* sink) {
  extern __shared__ char sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 7, zk = wave & 3, zrt = wave >> 2, half = lane >> 5;
  const int row = 32 * zrt + (lane & 31);
  unsigned acc = tid;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      if (PAT == 0) *reinterpret_cast<u2*>(sm + prow_g(row) * 272 + (32 * zk + 4 * half + 8 * (q & 3)) * 2 + (q >> 2) * 17408) = u2{(unsigned)it, acc};
      if (PAT == 1) *reinterpret_cast<u2*>(sm + (tid & 511) * 8 + (q & 15) * 4096) = u2{(unsigned)it, acc};
      if (PAT == 2) *reinterpret_cast<u4*>(sm + (tid & 511) * 16 + (q & 7) * 8192) = u4{(unsigned)it, acc, 1u, 2u};
      if (PAT == 3) *reinterpret_cast<unsigned*>(sm + (tid & 511) * 4 + (q & 31) * 2048) = acc + it;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 0x12345u) sink[0] = acc;
}
template <int NW, int PAT>
void run(int threads, int* s) {
  const int iters = 4000;
  hipLaunchKernelGGL((k<NW, PAT>), dim3(256), dim3(threads), 131072, 0, 10, s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NW, PAT>), dim3(256), dim3(threads), 131072, 0, iters, s);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * NW * threads * (PAT == 2 ? 16 : PAT == 3 ? 4 : 8);
  printf("pat %d threads %4d stores/wait %2d : %6.1f B/ns per CU  (%.1f ns per wave store)\n", PAT, threads, NW, bytes / (ms * 1e6),
         ms * 1e6 / (iters * NW));
}
int main() {
  int* s; (void)hipMalloc(&s, 4);
  for (int th : {512, 1024}) {
    run<4, 0>(th, s); run<8, 0>(th, s); run<16, 0>(th, s); run<32, 0>(th, s);
    run<8, 1>(th, s); run<32, 1>(th, s);
    run<8, 2>(th, s); run<32, 2>(th, s);
    run<32, 3>(th, s);
  }
  return 0;
}
