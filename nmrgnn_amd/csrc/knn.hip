// Neighbour-list builder on the GPU: the step in front of the hot path
// (nmrgnn/library.py:106-117 -> nmrdata.parse_universe; eval-struct's "MDAnalysis" timing bucket,
// nmrgnn/main.py:236-243).  Frames of a trajectory are independent graphs of the same n atoms.
//
// Brute force per frame, tiled through LDS: thread = one query atom, candidates stream through a
// [TILE] position tile shared by the workgroup, the K best (distance, index) pairs live in registers
// as a sorted list updated by a fully unrolled, branch-free insertion.  Ties keep the lower index
// first (candidates are visited in ascending index and comparisons are strict).  At protein sizes
// (n ~ 3-5k, 8-24 M pairs per frame) this is far below a millisecond per frame; frames of 16384 atoms
// and more take the cell grid of knn_cells.hip (same lists).
//
// Conventions (ours; nmrdata is not part of the reference tree): self excluded, ascending distance,
// distances * scale (0.1: Angstrom -> nm), unused slots (n-1 < K) are (0, 0.0); nlist holds
// frame-offset (batch-global) indices, inv_degree = 1/#(local index > 0) as library.py:115-116.
#include <string>

#include "ng_common.h"
#include "ng_internal.h"

namespace ng {

constexpr int KNN_TILE = 1024;

// the distance expression of every kNN kernel (knn_cells.hip has the same one): identical lists need identical rounding
__device__ __forceinline__ float knn_dist2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

template <int KMAX>
__global__ __launch_bounds__(256) void knn_kernel(int n, int K, float scale,
                                                  const float* __restrict__ pos,      // [G][n][3]
                                                  int32_t* __restrict__ nlist,        // [G*n][K]
                                                  float* __restrict__ edges,          // [G*n][K]
                                                  float* __restrict__ inv_degree) {   // [G*n]
  __shared__ float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
  const int frame = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float* fp = pos + (int64_t)frame * n * 3;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (i < n) { qx = fp[3 * i]; qy = fp[3 * i + 1]; qz = fp[3 * i + 2]; }
  float bd[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { bd[k] = INFINITY; bi[k] = 0; }

  for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
    const int cnt = min(KNN_TILE, n - t0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      sx[t] = fp[3 * (t0 + t)]; sy[t] = fp[3 * (t0 + t) + 1]; sz[t] = fp[3 * (t0 + t) + 2];
    }
    __syncthreads();
    if (i < n) {
      for (int t = 0; t < cnt; ++t) {
        const float dx = sx[t] - qx, dy = sy[t] - qy, dz = sz[t] - qz;
        const float d2 = knn_dist2(dx, dy, dz);
        const int j = t0 + t;
        if (d2 < bd[KMAX - 1] && j != i) {
#pragma unroll
          for (int k = KMAX - 1; k >= 1; --k) {
            const bool shift = bd[k - 1] > d2;          // old element k-1 moves up
            const bool here = !shift && bd[k] > d2;     // candidate lands in slot k
            bi[k] = shift ? bi[k - 1] : (here ? j : bi[k]);
            bd[k] = shift ? bd[k - 1] : (here ? d2 : bd[k]);
          }
          if (bd[0] > d2) { bd[0] = d2; bi[0] = j; }
        }
      }
    }
  }
  if (i >= n) return;
  const int64_t row = (int64_t)frame * n + i;
  int deg = 0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K) {
      const bool ok = bd[k] < INFINITY;
      nlist[row * K + k] = ok ? frame * n + bi[k] : 0;
      edges[row * K + k] = ok ? sqrtf(bd[k]) * scale : 0.f;
      deg += (ok && bi[k] > 0) ? 1 : 0;
    }
  }
  inv_degree[row] = deg > 0 ? 1.0f / (float)deg : 0.f;
}

// The same search with S = 8 lanes per query atom: lane s of a group scans the candidates t = s (mod 8) of every tile
// into its own sorted list, then the eight lists are merged by K rounds of "smallest head wins" over DPP.  Work per
// query is unchanged; the serial chain per thread is 8x shorter and a frame spreads over 8x as many workgroups —
// what a single molecule-sized frame needs (2770 atoms: 0.47 ms -> see tools/knn_time.py).  Order and ties are
// those of the one-lane kernel: (distance, index) ascending.
// S = 8 or 16 lanes per query (one DPP row at most): a single molecule-sized frame (2770 queries) fills 87 / 173 workgroups
template <int KMAX, int S>
__global__ __launch_bounds__(256) void knn_kernel_s8(int n, int K, float scale, const float* __restrict__ pos,
                                                     int32_t* __restrict__ nlist, float* __restrict__ edges,
                                                     float* __restrict__ inv_degree) {
  __shared__ float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
  const int frame = blockIdx.y;
  const int sl = threadIdx.x & (S - 1);
  const int i = blockIdx.x * (256 / S) + threadIdx.x / S;
  const float* fp = pos + (int64_t)frame * n * 3;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (i < n) { qx = fp[3 * i]; qy = fp[3 * i + 1]; qz = fp[3 * i + 2]; }
  float bd[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { bd[k] = INFINITY; bi[k] = 0x7fffffff; }

  for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
    const int cnt = min(KNN_TILE, n - t0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      sx[t] = fp[3 * (t0 + t)]; sy[t] = fp[3 * (t0 + t) + 1]; sz[t] = fp[3 * (t0 + t) + 2];
    }
    __syncthreads();
    if (i < n) {
      for (int t = sl; t < cnt; t += S) {
        const float dx = sx[t] - qx, dy = sy[t] - qy, dz = sz[t] - qz;
        const float d2 = knn_dist2(dx, dy, dz);
        const int j = t0 + t;
        if (d2 < bd[KMAX - 1] && j != i) {
#pragma unroll
          for (int k = KMAX - 1; k >= 1; --k) {
            const bool shift = bd[k - 1] > d2;
            const bool here = !shift && bd[k] > d2;
            bi[k] = shift ? bi[k - 1] : (here ? j : bi[k]);
            bd[k] = shift ? bd[k - 1] : (here ? d2 : bd[k]);
          }
          if (bd[0] > d2) { bd[0] = d2; bi[0] = j; }
        }
      }
    }
  }
  // merge: round k takes the lexicographically smallest (distance, index) head of the eight lanes
  float rd[KMAX];
  int ri[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    float md = bd[0];
    int mi = bi[0];
#define NG_KNN_STEP(CTRL)                                                                                        \
    {                                                                                                            \
      const float od = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, md), CTRL, 0xf, 0xf, false)); \
      const int oi = __builtin_amdgcn_update_dpp(0, mi, CTRL, 0xf, 0xf, false);                                 \
      const bool take = od < md || (od == md && oi < mi);                                                        \
      md = take ? od : md;                                                                                       \
      mi = take ? oi : mi;                                                                                       \
    }
    NG_KNN_STEP(0xB1)    // quad_perm 1,0,3,2
    NG_KNN_STEP(0x4E)    // quad_perm 2,3,0,1
    NG_KNN_STEP(0x141)   // row_half_mirror: the other quad of the group of eight
    if (S >= 16) { NG_KNN_STEP(0x128) }   // row_ror:8: the other eight lanes of the DPP row
#undef NG_KNN_STEP
    rd[k] = md; ri[k] = mi;
    const bool mine = bi[0] == mi && bd[0] == md;
#pragma unroll
    for (int q = 0; q < KMAX - 1; ++q) {
      bd[q] = mine ? bd[q + 1] : bd[q];
      bi[q] = mine ? bi[q + 1] : bi[q];
    }
    bd[KMAX - 1] = mine ? INFINITY : bd[KMAX - 1];
    bi[KMAX - 1] = mine ? 0x7fffffff : bi[KMAX - 1];
  }
  if (i >= n || sl != 0) return;
  const int64_t row = (int64_t)frame * n + i;
  int deg = 0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K) {
      const bool ok = rd[k] < INFINITY;
      nlist[row * K + k] = ok ? frame * n + ri[k] : 0;
      edges[row * K + k] = ok ? sqrtf(rd[k]) * scale : 0.f;
      deg += (ok && ri[k] > 0) ? 1 : 0;
    }
  }
  inv_degree[row] = deg > 0 ? 1.0f / (float)deg : 0.f;
}

}  // namespace ng

extern "C" int ng_knn_graph(ng_ctx* ctx, void* stream, int G, int n, int K, float scale,
                            const float* pos, int32_t* nlist, float* edges, float* inv_degree) {
  using namespace ng;
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  NG_REQUIRE(ctx, K >= 1 && K <= 64, "knn: neighbour count must be in [1,64]");
  NG_REQUIRE(ctx, G >= 0 && n >= 0, "knn: negative size");
  NG_REQUIRE(ctx, (int64_t)G * n < (int64_t)1 << 31, "knn: batch exceeds int32 indices");
  NG_REQUIRE(ctx, G <= 65535, "knn: at most 65535 frames per call");
  if (G == 0 || n == 0) return NG_OK;
  // large frames: cell grid (knn_cells.hip), the same lists in O(n) instead of O(n^2); NG_KNN=cells / brute force the choice
  if (!sw().knn_brute && !sw().knn_serial && knn_cells_supported(G, n, K) && (n >= 16384 || sw().knn_cells))
    return knn_cells(ctx, st, G, n, K, scale, pos, nlist, edges, inv_degree);
  ProfScope ps(ctx, st, "knn_graph");
  const dim3 grid((unsigned)cdiv(n, 256), (unsigned)G), block(256);
  if (K <= 16 && !sw().knn_serial)      // NG_KNN=serial: one lane per query (the first kernel)
  {
    const int64_t nq = (int64_t)G * n;          // few queries: more lanes per query, so that the launch still fills the chip
    if (nq <= 16384)
      hipLaunchKernelGGL((knn_kernel_s8<16, 16>), dim3((unsigned)cdiv(n, 16), (unsigned)G), block, 0, st, n, K, scale, pos,
                         nlist, edges, inv_degree);
    else
      hipLaunchKernelGGL((knn_kernel_s8<16, 8>), dim3((unsigned)cdiv(n, 32), (unsigned)G), block, 0, st, n, K, scale, pos,
                         nlist, edges, inv_degree);
  }
  else if (K <= 16)
    hipLaunchKernelGGL(knn_kernel<16>, grid, block, 0, st, n, K, scale, pos, nlist, edges, inv_degree);
  else if (K <= 32)
    hipLaunchKernelGGL(knn_kernel<32>, grid, block, 0, st, n, K, scale, pos, nlist, edges, inv_degree);
  else
    hipLaunchKernelGGL(knn_kernel<64>, grid, block, 0, st, n, K, scale, pos, nlist, edges, inv_degree);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}
