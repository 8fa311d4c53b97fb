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

// ---- one WAVE per query atom, for molecule-sized calls (round 4).  A single 2770-atom frame gives the lane kernels above 693
// waves that each walk 173 candidates with a 64-instruction branch-free insertion paid at almost every step (some lane of the
// wave always qualifies): 48 us, a quarter of a one-frame call, all of it the serial chain of one wave.  Here the 64 lanes of a
// wave take 64 candidates per step for ONE query and nothing is inserted before the bar is low:
//   A  every lane computes the keys of its candidates (key = bits(d^2) << 32 | index: the (distance, index) order of the other
//      kernels, so the lists come out the same bit for bit) into registers and keeps its smallest;
//   B  the K-th smallest of the 64 lane minima is an upper bound of the K-th smallest key overall (K distinct candidates lie at or
//      below it) — found by ranking the 64 minima against each other;
//   C  only keys at or below that bound are inserted (a few more than K) into a sorted list held ACROSS the lanes (lane k = k-th
//      smallest): one wave-wide shift and two compares per insertion, whatever K is.
// The frame's positions are staged in LDS once per workgroup (n <= 4096); the four waves of a workgroup take four queries.
typedef unsigned long long knn_u64;
__device__ __forceinline__ knn_u64 knn_readlane64(knn_u64 v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((knn_u64)hi << 32) | lo;
}
// lane l gets lane l - 1's value, lane 0 gets 0 (DPP wave_shr:1)
__device__ __forceinline__ knn_u64 knn_shr1(knn_u64 v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x138, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, false);
  return ((knn_u64)hi << 32) | lo;
}
constexpr int KNN_WAVE_MAXN = 4096;

template <int STEPS>
__global__ __launch_bounds__(256) void knn_wave_kernel(int n, int K, float scale, const float* __restrict__ pos,
                                                       int32_t* __restrict__ nlist, float* __restrict__ edges,
                                                       float* __restrict__ inv_degree) {
  extern __shared__ float spos[];                 // [3][64 * STEPS]
  constexpr int NP = 64 * STEPS;
  float* sx = spos; float* sy = spos + NP; float* sz = spos + 2 * NP;
  const int frame = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* fp = pos + (int64_t)frame * n * 3;
  for (int t = threadIdx.x; t < n; t += 256) { sx[t] = fp[3 * t]; sy[t] = fp[3 * t + 1]; sz[t] = fp[3 * t + 2]; }
  __syncthreads();
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;                             // uniform over the wave
  const float qx = sx[i], qy = sy[i], qz = sz[i];
  // A: keys of this lane's candidates t = 64 s + lane, and their minimum
  knn_u64 key[STEPS];
  knn_u64 mn = ~0ull;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int t = 64 * s + lane;
    const int tc = min(t, n - 1);
    const float d2 = knn_dist2(sx[tc] - qx, sy[tc] - qy, sz[tc] - qz);
    knn_u64 k = ((knn_u64)__builtin_bit_cast(unsigned, d2) << 32) | (unsigned)t;
    if (t >= n || t == i) k = ~0ull;
    key[s] = k;
    mn = k < mn ? k : mn;
  }
  // B: the K-th smallest lane minimum (keys of real candidates are distinct; absent ones are ~0 and rank last)
  knn_u64 tau = ~0ull;
  {
    int rank = 0;
    for (int b = 0; b < 64; ++b) rank += knn_readlane64(mn, b) < mn ? 1 : 0;
    const unsigned long long hit = __ballot(rank == K - 1 && mn != ~0ull);
    if (hit) tau = knn_readlane64(mn, __builtin_ctzll(hit));
  }
  // C: insert what lies at or below the bound
  knn_u64 list = ~0ull, kth = ~0ull;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    unsigned long long m = __ballot(key[s] <= tau && key[s] != ~0ull);
    while (m) {
      const int b = __builtin_ctzll(m);
      m &= m - 1;
      const knn_u64 c = knn_readlane64(key[s], b);
      if (c < kth) {
        const knn_u64 prev = knn_shr1(list);
        list = c < prev ? prev : (c < list ? c : list);
        kth = knn_readlane64(list, K - 1);
      }
    }
  }
  const int64_t row = (int64_t)frame * n + i;
  const bool ok = lane < K && list != ~0ull;
  const int idx = (int)(unsigned)list;
  const float d2 = __builtin_bit_cast(float, (unsigned)(list >> 32));
  if (lane < K) {
    nlist[row * K + lane] = ok ? frame * n + idx : 0;
    edges[row * K + lane] = ok ? sqrtf(d2) * scale : 0.f;
  }
  const int deg = __popcll(__ballot(ok && idx > 0));
  if (lane == 0) inv_degree[row] = deg > 0 ? 1.0f / (float)deg : 0.f;
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
  // molecule-sized calls: one wave per query (NG_KNN=serial / lanes: the earlier kernels for every size)
  if (!sw().knn_serial && !sw().knn_lanes && n <= KNN_WAVE_MAXN && (int64_t)G * n <= 16384) {
    const dim3 gw((unsigned)cdiv(n, 4), (unsigned)G);
    if (n <= 1024) hipLaunchKernelGGL((knn_wave_kernel<16>), gw, block, 3 * 64 * 16 * 4, st, n, K, scale, pos, nlist, edges, inv_degree);
    else if (n <= 2048) hipLaunchKernelGGL((knn_wave_kernel<32>), gw, block, 3 * 64 * 32 * 4, st, n, K, scale, pos, nlist, edges, inv_degree);
    else if (n <= 3072) hipLaunchKernelGGL((knn_wave_kernel<48>), gw, block, 3 * 64 * 48 * 4, st, n, K, scale, pos, nlist, edges, inv_degree);
    else hipLaunchKernelGGL((knn_wave_kernel<64>), gw, block, 3 * 64 * 64 * 4, st, n, K, scale, pos, nlist, edges, inv_degree);
  } else if (K <= 16 && !sw().knn_serial)      // 8 / 16 lanes per query
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
