// Second-stage reductions and the small transposed product used by the embedding / head weight
// gradients.  Deterministic (fixed summation order), no atomics.
#pragma once
#include <hip/hip_runtime.h>

#include "ng_common.h"

namespace ng {

// blockmax[blockIdx.x] = max over the 256-thread block of m (m >= 0).  max is exact and order-free.
__device__ __forceinline__ void block_max_store(float m, float* __restrict__ blockmax) {
  __shared__ float bm_red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) bm_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) blockmax[blockIdx.x] = fmaxf(fmaxf(bm_red[0], bm_red[1]), fmaxf(bm_red[2], bm_red[3]));
}

// out[map(idx)] = sum_z partial[z][idx].   One 1024-thread block per 64 consecutive elements; the
// 16 waves split z, so every lane has nz/16 independent, fully coalesced loads in flight (the
// one-thread-per-element form was latency-bound: 60-120 us for a 12K-element gradient).
//   w_map = 0: identity;  w_map = 1: MPLayer weight, idx = k*Nout + m with k = ne*F + l -> (l*F+m)*E+ne
// Same reduction with segments: the summed vector is cut into pieces that land in different tensors (weight and
// bias gradients of several layers): one launch instead of a reduction plus one device copy per tensor.
constexpr int REDUCE_MAX_SEG = 16;
struct ReduceSegs {
  int n;
  int begin[REDUCE_MAX_SEG];     // first index of the segment in the summed vector
  int len[REDUCE_MAX_SEG];
  float* dst[REDUCE_MAX_SEG];
};

// One reduction: what reduce_z_kernel / reduce_z_seg_kernel / reduce_batch_kernel all execute for block `blk` of the
// job (the summation order — and with it every bit of the result — is the same whichever kernel runs it).
struct ReduceJob {
  const float* partial;
  float* out;                    // plain / mapped form (sg == nullptr)
  int64_t n_elem, z_stride;
  int nz, w_map, F, E, Nout;
  int narrow;                    // NG_REDUCE=narrow: 64 elements per block whatever the size (tests compare the two forms' bits)
  // w_map == 3, "outer" job (round 6; queued by defer_outer_job): out[c*F + f] = sum_z aux[z*E + c] * partial[z*F + f] — the embedding
  // weight gradient of a molecule-sized call straight from atoms [nz][E = C] and dh0 [nz][F], no first-stage launch
  const float* aux = nullptr;
};

// Jobs of 32 K elements and more (the weight gradients of the default width: 196,608 elements x ~40 partials) run 256
// elements per block, four consecutive ones per lane: a wave's load is 1 KB contiguous instead of 256 B, a quarter of the
// requests for the same bytes (reduce_batch_kernel 150 -> ~80 us per F = 256 step).  Per element the order of the sum — wave w
// adds z = w, w + 16, ..., then the sixteen waves' sums in wave order — is that of the narrow form: the same bits.
__host__ __device__ inline bool reduce_job_wide(const ReduceJob& j) {
  return !j.narrow && j.n_elem >= 32768 && (j.n_elem & 3) == 0 && (j.z_stride & 3) == 0 && ((uintptr_t)j.partial & 15) == 0;
}
__host__ __device__ inline unsigned reduce_job_blocks(const ReduceJob& j) {
  const int64_t per = reduce_job_wide(j) ? 256 : 64;
  return (unsigned)((j.n_elem + per - 1) / per);
}
__device__ __forceinline__ int64_t reduce_out_index(const ReduceJob& j, int64_t idx) {
  if (j.w_map == 1) {          // MPLayer weight, idx = k*Nout + m with k = ne*F + l -> (l*F+m)*E+ne
    const int k = (int)(idx / j.Nout), m = (int)(idx % j.Nout);
    const int ne = k / j.F, l = k % j.F;
    return ((int64_t)l * j.F + m) * j.E + ne;
  }
  if (j.w_map == 2) {          // MPLayer weight from h^T B: idx = k*Nout + l with k = ne*F + m
    const int k = (int)(idx / j.Nout), l = (int)(idx % j.Nout);
    const int ne = k / j.F, m = k % j.F;
    return ((int64_t)l * j.F + m) * j.E + ne;
  }
  return idx;
}

__device__ __forceinline__ void reduce_job_block_wide(const ReduceJob& j, unsigned blk, float4 (*red)[64]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t idx = ((int64_t)blk * 64 + lane) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < j.n_elem) {
    const float* p = j.partial + idx;
    int z = w;
    for (; z + 112 < j.nz; z += 128) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (int64_t)(z + 16 * u) * j.z_stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; z < j.nz; z += 16) {
      const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)z * j.z_stride);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && idx < j.n_elem) {
    float4 t = red[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 v = red[k][lane]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    if (j.w_map == 0 && ((uintptr_t)j.out & 15) == 0) {
      *reinterpret_cast<float4*>(j.out + idx) = t;
    } else {
      j.out[reduce_out_index(j, idx)] = t.x; j.out[reduce_out_index(j, idx + 1)] = t.y;
      j.out[reduce_out_index(j, idx + 2)] = t.z; j.out[reduce_out_index(j, idx + 3)] = t.w;
    }
  }
}

__device__ __forceinline__ void reduce_job_block(const ReduceJob& j, const ReduceSegs* sg, unsigned blk, float (*red)[64]) {
  if (!sg && reduce_job_wide(j)) { reduce_job_block_wide(j, blk, reinterpret_cast<float4(*)[64]>(red)); return; }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blk * 64 + lane;
  float s = 0.f;
  if (j.w_map == 3) {
    if (idx < j.n_elem) {
      const int c = (int)(idx / j.F), f = (int)(idx % j.F);
      const float* x = j.aux + c;
      const float* y = j.partial + f;
      int z = w;
      for (; z + 112 < j.nz; z += 128) {
        float xv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { xv[u] = x[(int64_t)(z + 16 * u) * j.E]; yv[u] = y[(int64_t)(z + 16 * u) * j.F]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += xv[u] * yv[u];
      }
      for (; z < j.nz; z += 16) s += x[(int64_t)z * j.E] * y[(int64_t)z * j.F];
    }
  } else if (idx < j.n_elem) {
    // eight loads in flight per lane, added in z order (the sum is the one of the plain loop, bit for bit): with one
    // load per trip the kernel ran at the memory LATENCY — 40 us for the 68 MB of a training step's partials
    const float* p = j.partial + idx;
    int z = w;
    for (; z + 112 < j.nz; z += 128) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(z + 16 * u) * j.z_stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < j.nz; z += 16) s += p[(int64_t)z * j.z_stride];
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && idx < j.n_elem) {
    float t = red[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][lane];
    if (sg) {
#pragma unroll
      for (int k = 0; k < REDUCE_MAX_SEG; ++k)
        if (k < sg->n && idx >= sg->begin[k] && idx < sg->begin[k] + sg->len[k]) sg->dst[k][idx - sg->begin[k]] = t;
      return;
    }
    j.out[j.w_map == 3 ? idx : reduce_out_index(j, idx)] = t;
  }
}

static __global__ __launch_bounds__(1024) void reduce_z_kernel(const float* __restrict__ partial, int nz,
                                                        int64_t n_elem, float* __restrict__ out,
                                                        int w_map, int F, int E, int Nout,
                                                        int64_t z_stride, int narrow) {
  __shared__ __attribute__((aligned(16))) float red[16 * 4][64];      // [16][64] floats, or float4 (wide jobs)
  const ReduceJob j{partial, out, n_elem, z_stride, nz, w_map, F, E, Nout, narrow};
  reduce_job_block(j, nullptr, blockIdx.x, red);
}

static inline void launch_reduce_z(hipStream_t st, const float* partial, int nz, int64_t n_elem, float* out,
                            int w_map = 0, int F = 0, int E = 0, int Nout = 1, int64_t z_stride = 0) {
  const ReduceJob j{partial, out, n_elem, z_stride ? z_stride : n_elem, nz, w_map, F, E, Nout, sw().reduce_narrow ? 1 : 0};
  hipLaunchKernelGGL(reduce_z_kernel, dim3(reduce_job_blocks(j)), dim3(1024), 0, st, partial,
                     nz, n_elem, out, w_map, F, E, Nout, j.z_stride, j.narrow);
}

static __global__ __launch_bounds__(1024) void reduce_z_seg_kernel(const float* __restrict__ partial, int nz,
                                                                   int64_t n_elem, int64_t z_stride, ReduceSegs sg) {
  __shared__ float red[16][64];
  const ReduceJob j{partial, nullptr, n_elem, z_stride, nz, 0, 0, 0, 1, 1};
  reduce_job_block(j, &sg, blockIdx.x, red);
}

static inline void launch_reduce_z_seg(hipStream_t st, const float* partial, int nz, int64_t n_elem,
                                       int64_t z_stride, const ReduceSegs& sg) {
  hipLaunchKernelGGL(reduce_z_seg_kernel, dim3((unsigned)cdiv(n_elem, 64)), dim3(1024), 0, st, partial, nz,
                     n_elem, z_stride, sg);
}

// Deferred form (ng_defer_reductions, capi.hip): while deferral is on, a producer takes its partial buffer from the
// context's reduction arena (deferred_partials: stays valid until the flush, nullptr = deferral off) and hands the
// reduction to reduce_or_defer / reduce_seg_or_defer, which queue it; ng_flush_reductions runs every queued job in ONE
// launch (reduce_batch_kernel).  With deferral off both run the reduction at once, as before.  The backward of a
// training step has seven such reductions of 5-12 us each behind kernels that leave the GPU idle meanwhile.
float* deferred_partials(ng_ctx* ctx, size_t floats);
int reduce_or_defer(ng_ctx* ctx, hipStream_t st, const float* partial, int nz, int64_t n_elem, float* out, int w_map = 0,
                    int F = 0, int E = 0, int Nout = 1, int64_t z_stride = 0);
// caller_owned: `partial` is not from the arena but the caller keeps it alive until the flush (head_loss_reduce)
int reduce_seg_or_defer(ng_ctx* ctx, hipStream_t st, const float* partial, int nz, int64_t n_elem, int64_t z_stride,
                        const ReduceSegs& sg, bool caller_owned = false);
int flush_reductions(ng_ctx* ctx, hipStream_t st);
// out[c*F + f] = sum_{i < n} X[i*C + c] * Y[i*F + f] as a job of the deferred batch (no launch of its own); false: deferral is
// off or n is too long for a single-stage sum — the caller runs its two-stage form.  X, Y must stay valid until the flush.
constexpr int64_t OUTER_JOB_MAX_ROWS = 2048;
bool defer_outer_job(ng_ctx* ctx, hipStream_t st, const float* X, const float* Y, int64_t n, int C, int F, float* out);

// partial[blk][a*B + b] = sum_{rows of blk} X(row, a) * Y(row, b)      (A <= 32, any B)
// rows are staged 64 at a time in LDS; thread t owns items t, t+256, ... (<= SMALL_TN_ITEMS each);
// blockIdx.y selects a batch of 256*SMALL_TN_ITEMS items.
constexpr int SMALL_TN_ITEMS = 12;

template <class FX, class FY>
__global__ __launch_bounds__(256) void small_tn_kernel(int64_t N, int A, int B,
                                                       int64_t rows_per_block, FX fx, FY fy,
                                                       float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Xs = sm;           // [64][A]
  float* Ys = sm + 64 * A;  // [64][B]
  const int items = A * B;
  const int item0 = blockIdx.y * 256 * SMALL_TN_ITEMS;
  float acc[SMALL_TN_ITEMS];
#pragma unroll
  for (int j = 0; j < SMALL_TN_ITEMS; ++j) acc[j] = 0.f;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
  for (int64_t rb = r0; rb < r1; rb += 64) {
    const int nr = (int)(r1 - rb < 64 ? r1 - rb : 64);
    __syncthreads();
    for (int t = threadIdx.x; t < nr * A; t += 256) Xs[t] = fx(rb + t / A, t % A);
    for (int t = threadIdx.x; t < nr * B; t += 256) Ys[t] = fy(rb + t / B, t % B);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SMALL_TN_ITEMS; ++j) {
      const int it = item0 + threadIdx.x + 256 * j;
      if (it < items) {
        const int a = it / B, b = it % B;
        float s = acc[j];
        for (int r = 0; r < nr; ++r) s += Xs[r * A + a] * Ys[r * B + b];
        acc[j] = s;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < SMALL_TN_ITEMS; ++j) {
    const int it = item0 + threadIdx.x + 256 * j;
    if (it < items) partial[(int64_t)blockIdx.x * items + it] = acc[j];
  }
}

}  // namespace ng
