// LDS bank-conflict cycles of the NON-transposing accesses of edge_bwd_h2_kernel (the transposing reads measure 0 conflict
// cycles in their kernel layouts: tools/ubench/trbank2.hip + tools/pmc_trbank.sh).  Each launch repeats one access pattern
// with the kernel's own lane -> address map (8 waves, wave w: zk = w & 3, zrt = w >> 2, lane: row = 32 zrt + (l & 31),
// half = l >> 5); run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT (tools/pmc_ldswr.sh).
//   0  G-image piece write   ds_write_b64   prow_g(row) * 272 + (32 zk + 4 half + 8 q) * 2          (hx_img_write<272>)
//   1  Z-image piece write   ds_write_b64   prow_z(row) * 264 + (32 zk + 4 half + 8 q) * 2          (hx_img_write<264>)
//   2  G rows, dZ B operand  ds_read_b128   prow_g(row) * 272 + 16 half + 32 ks                      (hx_dz_gemm)
//   3  fp32 staging write    ds_write_b128  row * 528 + (32 zk + 4 half + 8 q) * 4                   (head1)
//   4  staging column read   ds_read_b32    r * 528 + 4 (tid & 127), r = 16 (tid >> 7) + k           (head2: dWo)
//   6  G-image pieces as 16-byte writes: lane (row, half) holds columns 8 (2p + half) .. + 7 after a v_permlane32_swap  ds_write_b128
//   7  control: 8 contiguous bytes per lane                                                           ds_write_b64
//   5  G-image write, rows r and r + 16 a further 16 B apart (stride 272, + 16 B for rows >= 16 of a 32-row block): a variant
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ int prow_z(int e) { const int hi = e >> 4, a = (e >> 2) & 3, b = e & 3; return 2 * (a + 4 * b) + (hi & 1) + 32 * (hi >> 1); }
__device__ int prow_g(int e) { const int hi = e >> 4, a = (e >> 2) & 3, b = e & 3; return 16 * hi + 4 * b + a; }
__global__ __launch_bounds__(512, 1) void k(int iters, int pat, int* sink) {
  extern __shared__ char sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, zk = wave & 3, zrt = wave >> 2, half = lane >> 5;
  const int row = 32 * zrt + (lane & 31);
  for (int t = tid; t < 65536 / 4; t += 512) reinterpret_cast<int*>(sm)[t] = t;
  __syncthreads();
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (pat == 0) { *reinterpret_cast<u2*>(sm + prow_g(row) * 272 + (32 * zk + 4 * half + 8 * q) * 2) = u2{(unsigned)it, acc}; }
      else if (pat == 1) { *reinterpret_cast<u2*>(sm + prow_z(row) * 264 + (32 * zk + 4 * half + 8 * q) * 2) = u2{(unsigned)it, acc}; }
      else if (pat == 2) { const u4 v = *reinterpret_cast<const u4*>(sm + prow_g(row) * 272 + 16 * half + 32 * (q + 4 * (it & 1))); acc += v[0] ^ v[3]; }
      else if (pat == 3) { *reinterpret_cast<u4*>(sm + row * 528 + (32 * zk + 4 * half + 8 * q) * 4) = u4{(unsigned)it, acc, 1u, 2u}; }
      else if (pat == 4) { acc += *reinterpret_cast<const unsigned*>(sm + (16 * (tid >> 7) + q + 4 * (it & 3)) * 528 + 4 * (tid & 127)); }
      else if (pat == 6) { *reinterpret_cast<u4*>(sm + prow_g(row) * 272 + (32 * zk + 8 * (2 * (q & 1) + half)) * 2 + (q >> 1) * 17408) = u4{(unsigned)it, acc, 1u, 2u}; }
      else if (pat == 7) { *reinterpret_cast<u2*>(sm + (tid & 511) * 8 + q * 4096) = u2{(unsigned)it, acc}; }
      else { const int pr = prow_g(row); *reinterpret_cast<u2*>(sm + pr * 272 + ((pr >> 4) & 1) * 16 + (32 * zk + 4 * half + 8 * q) * 2) = u2{(unsigned)it, acc}; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 0x12345u) sink[0] = acc;
}
int main() {
  int* s; (void)hipMalloc(&s, 4);
  const int iters = 20000;
  for (int pat = 0; pat < 8; ++pat) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, iters, pat, s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, iters, pat, s);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("pattern %d : %.2f ns per wave access\n", pat, ms * 1e6 / (iters * 4.0));
  }
  return 0;
}
