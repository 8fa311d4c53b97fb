// K nearest neighbours through a uniform cell grid: the neighbour search of ng_knn_graph for LARGE frames
// (nmrgnn/library.py:106-117 -> nmrdata.parse_universe; SURVEY §8(f1) "a cell-list kNN HIP kernel is the natural
// follow-up").  The brute-force kernels of knn.hip visit n candidates per query: fine at protein size (2770 atoms:
// 35 us), 1e10 pair evaluations at 100 k atoms.  Here a query visits the atoms of the cells around it:
//   1. bounding box of the frame; cell edge L = 0.5 (K / density)^(1/3) (two shells of cells then usually settle the K
//      nearest); the grid is clipped to `cap` cells per frame
//   2. counting sort of the atoms by cell (integer atomics, ng_exclusive_scan_i32): sorted (x, y, z, index) records
//   3. one thread per SORTED atom (neighbouring threads sit in the same cell and walk the same ranges): shells of cells
//      r = 0, 1, 2, ... around its cell — a row of cells along x is one contiguous range of records — until the K-th
//      distance found is within r L of the query (everything unvisited is at least that far away) or the grid is exhausted
// Result: EXACTLY the lists of the brute-force kernels — (distance, index) ascending, ties to the lower index, the same
// fp32 distance expression (knn_dist2) — whatever the grid; tests compare the two bit for bit.
#include <algorithm>

#include "ng_common.h"
#include "ng_internal.h"

extern "C" int ng_exclusive_scan_i32(ng_ctx*, void*, int64_t, const int32_t*, int32_t*);

namespace ng {

struct KcGrid {          // per frame
  float ox, oy, oz;      // lower corner
  float inv_l, l;        // 1 / cell edge, cell edge
  int nx, ny, nz;
};

// the distance expression of every kNN kernel (knn.hip uses the same one): bit-identical orders need identical rounding
__device__ __forceinline__ float knn_dist2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

__device__ __forceinline__ int kc_coord(float x, float o, float inv_l, int n) {
  return min(max((int)floorf((x - o) * inv_l), 0), n - 1);
}

// ---- 1. bounding box (KC_BOX_BLOCKS partial boxes per frame), then the grid of every frame -----------------------------
constexpr int KC_BOX_BLOCKS = 64;
#ifndef KC_EDGE
#define KC_EDGE 0.50f
#endif

__device__ __forceinline__ void kc_box_reduce(float (&lo)[3], float (&hi)[3], float (*red)[16]) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[a][w] = lo[a]; red[3 + a][w] = hi[a]; }
  __syncthreads();
}

__global__ __launch_bounds__(256) void kc_box_kernel(int n, const float* __restrict__ pos, float* __restrict__ boxes) {
  __shared__ float red[6][16];
  const float* fp = pos + (int64_t)blockIdx.y * n * 3;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += KC_BOX_BLOCKS * 256)
#pragma unroll
    for (int a = 0; a < 3; ++a) { const float v = fp[3 * i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
  kc_box_reduce(lo, hi, red);
  if (threadIdx.x < 6) {
    float v = red[threadIdx.x][0];
    for (int q = 1; q < 4; ++q) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][q]) : fmaxf(v, red[threadIdx.x][q]);
    boxes[((int64_t)blockIdx.y * KC_BOX_BLOCKS + blockIdx.x) * 6 + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(64) void kc_grid_kernel(int n, int K, int cap, const float* __restrict__ boxes, KcGrid* __restrict__ grids) {
  __shared__ float red[6][16];
  const float* b = boxes + ((int64_t)blockIdx.x * KC_BOX_BLOCKS + threadIdx.x) * 6;
  float lo[3] = {b[0], b[1], b[2]}, hi[3] = {b[3], b[4], b[5]};
  kc_box_reduce(lo, hi, red);
  if (threadIdx.x == 0) {
    const float ex = fmaxf(red[3][0] - red[0][0], 1e-3f), ey = fmaxf(red[4][0] - red[1][0], 1e-3f), ez = fmaxf(red[5][0] - red[2][0], 1e-3f);
    // mean density of the box (a protein does not fill its box: the true local density is higher, the cells then hold more
    // atoms than aimed for — slower, never wrong)
    const float rho = (float)n / (ex * ey * ez);
    // cell edge = KC_EDGE (K / rho)^(1/3) = 0.8 x the radius of the sphere that holds K atoms at that density: most queries
    // settle after the second shell (125 cells, ~8 K candidates).  Measured per 110,800-atom frame: 0.40 -> 235 us, 0.50 ->
    // 202, 0.60 -> 232, 0.72 (one shell for an interior atom, but a wave of 64 queries always has one that needs two) -> 320
    float l = KC_EDGE * cbrtf((float)K / rho);
    int nx, ny, nz;
    for (;;) {
      nx = (int)fminf(ex / l, 4.0e6f) + 1; ny = (int)fminf(ey / l, 4.0e6f) + 1; nz = (int)fminf(ez / l, 4.0e6f) + 1;
      if ((int64_t)nx * ny * nz <= cap) break;
      l *= 1.26f;                                  // a factor of two in cell volume
    }
    KcGrid g;
    g.ox = red[0][0]; g.oy = red[1][0]; g.oz = red[2][0];
    g.l = l; g.inv_l = 1.0f / l;
    g.nx = nx; g.ny = ny; g.nz = nz;
    grids[blockIdx.x] = g;
  }
}

// ---- 2. counting sort by cell ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kc_count_kernel(int n, int cap, const float* __restrict__ pos, const KcGrid* __restrict__ grids,
                                                       int32_t* __restrict__ cell_of, int32_t* __restrict__ count) {
  const int frame = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const KcGrid g = grids[frame];
  const float* p = pos + ((int64_t)frame * n + i) * 3;
  const int cx = kc_coord(p[0], g.ox, g.inv_l, g.nx), cy = kc_coord(p[1], g.oy, g.inv_l, g.ny), cz = kc_coord(p[2], g.oz, g.inv_l, g.nz);
  const int c = (cz * g.ny + cy) * g.nx + cx;
  cell_of[(int64_t)frame * n + i] = c;
  atomicAdd(count + (int64_t)frame * cap + c, 1);
}

// records land in their cell's range in arrival order: the search result does not depend on it (explicit (distance,
// index) order), so the unordered fill is deterministic where it matters
__global__ __launch_bounds__(256) void kc_fill_kernel(int n, int cap, const float* __restrict__ pos, const int32_t* __restrict__ cell_of,
                                                      const int32_t* __restrict__ start, int32_t* __restrict__ cursor,
                                                      float4* __restrict__ rec) {
  const int frame = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t row = (int64_t)frame * n + i;
  const int64_t cell = (int64_t)frame * cap + cell_of[row];
  const int slot = start[cell] + atomicAdd(cursor + cell, 1);
  const float* p = pos + row * 3;
  rec[slot] = make_float4(p[0], p[1], p[2], __builtin_bit_cast(float, i));
}

// ---- 3. the search ------------------------------------------------------------------------------------------------------
// A list entry is ONE 64-bit key, (bits of the squared distance) << 32 | index: squared distances are non-negative floats,
// whose bit patterns order like the values, so key order IS (distance, index) order — one compare per slot instead of
// three.  The insertion appears once in the kernel (one loop over the one or two record ranges of a row of cells): with
// a copy per call site, and four more for a four-deep prefetch, the kernel was instruction-fetch bound (6 us per ROW).
template <int KMAX>
__global__ __launch_bounds__(256) void kc_query_kernel(int n, int K, int cap, float scale, const KcGrid* __restrict__ grids,
                                                       const int32_t* __restrict__ start, const float4* __restrict__ rec,
                                                       int32_t* __restrict__ nlist, float* __restrict__ edges,
                                                       float* __restrict__ inv_degree) {
  const int frame = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;      // sorted slot inside the frame
  if (p >= n) return;
  const KcGrid g = grids[frame];
  const float4 q = rec[(int64_t)frame * n + p];
  const int i = __builtin_bit_cast(int, q.w);
  const int cx = kc_coord(q.x, g.ox, g.inv_l, g.nx), cy = kc_coord(q.y, g.oy, g.inv_l, g.ny), cz = kc_coord(q.z, g.oz, g.inv_l, g.nz);
  const int32_t* st = start + (int64_t)frame * cap;
  constexpr uint64_t EMPTY = ((uint64_t)0x7f800000u << 32) | 0x7fffffffu;      // (inf, no index)
  uint64_t key[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) key[k] = EMPTY;

  const int rmax = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));
  for (int r = 0; r <= rmax; ++r) {
    for (int dz = -r; dz <= r; ++dz) {
      const int z = cz + dz;
      if (z < 0 || z >= g.nz) continue;
      for (int dy = -r; dy <= r; ++dy) {
        const int y = cy + dy;
        if (y < 0 || y >= g.ny) continue;
        const int rowc = (z * g.ny + y) * g.nx;
        // a whole row of the shell (cells cx-r .. cx+r are one range of records), or — an interior row — only its two end cells
        const bool whole = abs(dz) == r || abs(dy) == r;
        const int xl = cx - r, xr = cx + r;
        int a0, b0, a1 = 0, b1 = 0;
        if (whole) {
          a0 = st[rowc + max(xl, 0)]; b0 = st[rowc + min(xr, g.nx - 1) + 1];
        } else {
          a0 = xl >= 0 ? st[rowc + xl] : 0; b0 = xl >= 0 ? st[rowc + xl + 1] : 0;
          a1 = xr < g.nx ? st[rowc + xr] : 0; b1 = xr < g.nx ? st[rowc + xr + 1] : 0;
        }
        const int len0 = b0 - a0, total = len0 + (b1 - a1);
        for (int u = 0; u < total; ++u) {
          const float4 c = rec[u < len0 ? a0 + u : a1 + (u - len0)];
          const float d2 = knn_dist2(c.x - q.x, c.y - q.y, c.z - q.z);
          const unsigned j = __builtin_bit_cast(unsigned, c.w);
          const uint64_t kk = ((uint64_t)__builtin_bit_cast(unsigned, d2) << 32) | j;
          if (kk < key[KMAX - 1] && (int)j != i) {
#pragma unroll
            for (int k = KMAX - 1; k >= 1; --k) {
              const bool shift = kk < key[k - 1];                 // old element k-1 moves up
              key[k] = shift ? key[k - 1] : (kk < key[k] ? kk : key[k]);
            }
            key[0] = kk < key[0] ? kk : key[0];
          }
        }
      }
    }
    // every atom outside the cube of shells 0..r is at least r L from the query (it is at least r whole cells away along
    // some axis; 0.1 % off for the rounding of the cell assignment).  Enough once the K-th distance is inside that.
    const float reach = (float)r * g.l * 0.999f;
    uint64_t kth = EMPTY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) kth = k == K - 1 ? key[k] : kth;      // (a run-time index would put the list on the stack)
    if (__builtin_bit_cast(float, (unsigned)(kth >> 32)) <= reach * reach) break;
  }

  const int64_t row = (int64_t)frame * n + i;
  int deg = 0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K) {
      const float d2 = __builtin_bit_cast(float, (unsigned)(key[k] >> 32));
      const int j = (int)(unsigned)key[k];
      const bool ok = d2 < INFINITY;
      nlist[row * K + k] = ok ? frame * n + j : 0;
      edges[row * K + k] = ok ? sqrtf(d2) * scale : 0.f;
      deg += (ok && j > 0) ? 1 : 0;
    }
  }
  inv_degree[row] = deg > 0 ? 1.0f / (float)deg : 0.f;
}

// cells per frame: about n / 4 (the grid kernel shrinks its grid to fit), a power of two in [64, 2^21]
static int kc_cell_cap(int n) {
  int cap = 64;
  while (cap < 4 * (int64_t)n && cap < (1 << 22)) cap <<= 1;
  return cap;
}

bool knn_cells_supported(int G, int n, int K) {
  return K >= 1 && K <= 64 && n >= 64 && (int64_t)G * kc_cell_cap(n) <= ((int64_t)1 << 26);
}

int knn_cells(ng_ctx* ctx, hipStream_t st, int G, int n, int K, float scale, const float* pos, int32_t* nlist, float* edges,
              float* inv_degree) {
  const int cap = kc_cell_cap(n);
  const int64_t rows = (int64_t)G * n, cells = (int64_t)G * cap;
  // scratch: records [rows] float4 | cell_of [rows] | count [cells] | start [cells + 1] | grids [G] | partial boxes [G][64][6]
  const size_t bytes = (size_t)rows * 16 + (size_t)rows * 4 + (size_t)cells * 4 + (size_t)(cells + 1) * 4 + (size_t)G * (sizeof(KcGrid) + KC_BOX_BLOCKS * 6 * 4) + 256;
  char* ws = (char*)workspace(ctx, bytes);
  if (!ws) return NG_ERR_NOMEM;
  float4* rec = reinterpret_cast<float4*>(ws);
  int32_t* cell_of = reinterpret_cast<int32_t*>(ws + (size_t)rows * 16);
  int32_t* count = cell_of + rows;
  int32_t* start = count + cells;
  KcGrid* grids = reinterpret_cast<KcGrid*>(start + cells + 1 + ((cells + 1) & 1));
  float* boxes = reinterpret_cast<float*>(grids + G);
  const dim3 grid((unsigned)cdiv(n, 256), (unsigned)G), block(256);
  {
    ProfScope ps(ctx, st, "knn_cells_sort");
    NG_HIP(ctx, hipMemsetAsync(count, 0, (size_t)cells * 4, st));
    hipLaunchKernelGGL(kc_box_kernel, dim3(KC_BOX_BLOCKS, (unsigned)G), dim3(256), 0, st, n, pos, boxes);
    hipLaunchKernelGGL(kc_grid_kernel, dim3((unsigned)G), dim3(64), 0, st, n, K, cap, boxes, grids);
    hipLaunchKernelGGL(kc_count_kernel, grid, block, 0, st, n, cap, pos, grids, cell_of, count);
    NG_HIP(ctx, hipGetLastError());
    const int rc = ng_exclusive_scan_i32(ctx, st, cells, count, start);
    if (rc) return rc;
    NG_HIP(ctx, hipMemsetAsync(count, 0, (size_t)cells * 4, st));      // now the fill cursors
    hipLaunchKernelGGL(kc_fill_kernel, grid, block, 0, st, n, cap, pos, cell_of, start, count, rec);
    NG_HIP(ctx, hipGetLastError());
  }
  ProfScope ps(ctx, st, "knn_cells_query");
  if (K <= 16)
    hipLaunchKernelGGL(kc_query_kernel<16>, grid, block, 0, st, n, K, cap, scale, grids, start, rec, nlist, edges, inv_degree);
  else if (K <= 32)
    hipLaunchKernelGGL(kc_query_kernel<32>, grid, block, 0, st, n, K, cap, scale, grids, start, rec, nlist, edges, inv_degree);
  else
    hipLaunchKernelGGL(kc_query_kernel<64>, grid, block, 0, st, n, K, cap, scale, grids, start, rec, nlist, edges, inv_degree);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

}  // namespace ng
