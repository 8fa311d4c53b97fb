// Per-batch graph preprocessing on the engine side (round 3): the incoming-edge (CSC) lists of the deterministic
// backward scatter and the compute-side copy of the padded neighbour lists, built by the library instead of a chain of
// torch sort / bincount / cumsum launches.  The reference feeds a NEW graph every step (nmrgnn/library.py:88-89), so
// this work is per step, not per data set.
//
//   entry id  eid = i*K + j (padded lists, slots with edges == 0 dropped: they carry e == 0 exactly, model.py:261)
//             or the CSR entry index (row_ptr form: every entry is live)
//   csc_ptr[t] .. csc_ptr[t+1]: the entries whose neighbour is atom t, in ASCENDING entry id — the order a stable sort by
//   target gives, which every bit-for-bit test of the backward relies on.
//
// A stable counting sort in five small launches: histogram by target (integer atomics: order-free), exclusive scan
// (block-local + block sums + offsets), unordered fill through per-target cursors, and a per-target rank sort of each
// short segment (in-degree ~ K) that restores ascending entry order.  Nothing floating-point is reduced here, so the
// atomics cannot make the result depend on the schedule.
#include <string>

#include "ng_common.h"

namespace ng {

constexpr int GL_BLOCK = 256;
constexpr int GL_SCAN_ITEMS = 4;                       // per thread
constexpr int GL_SCAN_TILE = GL_BLOCK * GL_SCAN_ITEMS;  // 1024 counts per scan block

// one thread per entry: histogram of live targets; the compute-side list (padded slots -> the atom itself)
// An entry whose target lies outside [0, N) is treated like a padded slot (dropped from the incoming lists, compute-side
// index = the atom itself): with validate=False nothing else stands between a bad index and an out-of-bounds atomic.
__global__ __launch_bounds__(GL_BLOCK) void gl_count_kernel(int64_t n_entries, int64_t N, int K, const int32_t* __restrict__ nlist,
                                                            const float* __restrict__ edges, int32_t* __restrict__ nlist_c,
                                                            int32_t* __restrict__ count) {
  const int64_t eid = (int64_t)blockIdx.x * GL_BLOCK + threadIdx.x;
  if (eid >= n_entries) return;
  const int32_t t = nlist[eid];
  const bool live = (edges == nullptr || edges[eid] > 0.f) && t >= 0 && t < N;
  if (nlist_c) nlist_c[eid] = live ? t : (int32_t)(eid / K);
  if (live) atomicAdd(&count[t], 1);
}

// exclusive scan, step 1: per block of GL_SCAN_TILE counts, local exclusive scan in place + the block total
__global__ __launch_bounds__(GL_BLOCK) void gl_scan_local_kernel(int64_t n, int32_t* __restrict__ data, int32_t* __restrict__ block_sum) {
  __shared__ int32_t s[GL_BLOCK];
  const int64_t base = (int64_t)blockIdx.x * GL_SCAN_TILE + (int64_t)threadIdx.x * GL_SCAN_ITEMS;
  int32_t v[GL_SCAN_ITEMS], sum = 0;
#pragma unroll
  for (int k = 0; k < GL_SCAN_ITEMS; ++k) { v[k] = base + k < n ? data[base + k] : 0; sum += v[k]; }
  s[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < GL_BLOCK; off <<= 1) {
    const int32_t add = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += add;
    __syncthreads();
  }
  int32_t run = s[threadIdx.x] - sum;       // exclusive prefix of this thread inside the block
#pragma unroll
  for (int k = 0; k < GL_SCAN_ITEMS; ++k) {
    if (base + k < n) data[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == GL_BLOCK - 1) block_sum[blockIdx.x] = s[GL_BLOCK - 1];
}

// step 2 (one block): exclusive scan of the block totals, any count of them
__global__ __launch_bounds__(GL_BLOCK) void gl_scan_sums_kernel(int nblocks, int32_t* __restrict__ block_sum) {
  __shared__ int32_t s[GL_BLOCK];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblocks; b0 += GL_BLOCK) {
    const int i = b0 + threadIdx.x;
    const int32_t v = i < nblocks ? block_sum[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < GL_BLOCK; off <<= 1) {
      const int32_t add = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nblocks) block_sum[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == GL_BLOCK - 1) carry += s[GL_BLOCK - 1];
    __syncthreads();
  }
}

// step 3: csc_ptr[i] = local prefix + block offset = cursor[i]   (csc_ptr[n], the total, is written by gl_sort_kernel:
// after the fill the cursor of the last target stands at it)
__global__ __launch_bounds__(GL_BLOCK) void gl_scan_apply_kernel(int64_t n, const int32_t* __restrict__ local, const int32_t* __restrict__ block_sum,
                                                                 int32_t* __restrict__ csc_ptr, int32_t* __restrict__ cursor) {
  const int64_t i = (int64_t)blockIdx.x * GL_BLOCK + threadIdx.x;
  if (i < n) {
    const int32_t p = local[i] + block_sum[i / GL_SCAN_TILE];
    csc_ptr[i] = p;
    cursor[i] = p;
  }
}

// unordered fill: entry eid goes to the next free slot of its target
__global__ __launch_bounds__(GL_BLOCK) void gl_fill_kernel(int64_t n_entries, int64_t N, const int32_t* __restrict__ nlist,
                                                           const float* __restrict__ edges, int32_t* __restrict__ cursor,
                                                           int32_t* __restrict__ tmp) {
  const int64_t eid = (int64_t)blockIdx.x * GL_BLOCK + threadIdx.x;
  if (eid >= n_entries) return;
  if (edges != nullptr && !(edges[eid] > 0.f)) return;
  const int32_t t = nlist[eid];
  if (t < 0 || t >= N) return;
  const int pos = atomicAdd(&cursor[t], 1);
  tmp[pos] = (int32_t)eid;
}

// Per target: its segment of tmp in ascending entry id -> csc_edge.  Round 5: a 16-LANE GROUP per target instead of one thread
// (round 4: registers for segments up to 32, an O(d^2) loop over memory in one thread beyond — 188 us average, 827 us at worst per
// call in the bench trace, where a handful of the 131,072 random targets have more than 32 incoming edges, and milliseconds for a
// solvated ion or a coarse cutoff).  A group ranks its segment sixteen entries at a time against sixteen entries passed round the
// group (ds_bpermute): (len / 16)^2 x 16 exchanges.  Segments longer than GL_SORT_GROUP_MAX are left to the whole workgroup after a
// barrier: bitonic sort in LDS up to GL_SORT_LDS entries, rank sort from LDS chunks beyond.  Same output order as before (entry
// ids are distinct, so the order is total): lists bit-identical.
constexpr int GL_SORT_GROUP_MAX = 64, GL_SORT_LDS = 4096, GL_GROUPS = GL_BLOCK / 16;
__global__ __launch_bounds__(GL_BLOCK) void gl_sort_kernel(int64_t n, const int32_t* __restrict__ cursor, int32_t* __restrict__ csc_ptr,
                                                           const int32_t* __restrict__ tmp, int32_t* __restrict__ csc_edge) {
  __shared__ int32_t s_key[GL_SORT_LDS];
  __shared__ int s_long[GL_GROUPS];
  __shared__ int s_nlong;
  const int g = threadIdx.x >> 4, gl = threadIdx.x & 15;
  const int64_t t = (int64_t)blockIdx.x * GL_GROUPS + g;
  if (threadIdx.x == 0) s_nlong = 0;
  __syncthreads();
  if (t < n) {
    const int p0 = csc_ptr[t], p1 = cursor[t];      // after the fill a cursor stands at the end of its segment
    const int len = p1 - p0;
    if (len <= GL_SORT_GROUP_MAX) {
      for (int a = 0; a < len; a += 16) {
        const int32_t v = a + gl < len ? tmp[p0 + a + gl] : 0x7fffffff;
        int rank = 0;
        for (int b = 0; b < len; b += 16) {
          const int32_t u = b + gl < len ? tmp[p0 + b + gl] : 0x7fffffff;
#pragma unroll
          for (int r = 0; r < 16; ++r) rank += __shfl(u, r, 16) < v ? 1 : 0;     // entry ids are distinct; the padding ranks last
        }
        if (a + gl < len) csc_edge[p0 + rank] = v;
      }
    } else if (gl == 0) {
      s_long[atomicAdd(&s_nlong, 1)] = g;
    }
  }
  __syncthreads();
  // (csc_ptr[n] is written at the very end: the last block's groups read csc_ptr[t] for t < n only)
  const int nlong = s_nlong;
  for (int i = 0; i < nlong; ++i) {
    const int64_t tl = (int64_t)blockIdx.x * GL_GROUPS + s_long[i];
    const int p0 = csc_ptr[tl], len = cursor[tl] - p0;
    if (len <= GL_SORT_LDS) {
      int m = 64;
      while (m < len) m <<= 1;
      for (int j = threadIdx.x; j < m; j += GL_BLOCK) s_key[j] = j < len ? tmp[p0 + j] : 0x7fffffff;
      __syncthreads();
      for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          // one compare-exchange per PAIR (pair q: lower element x = 2 j (q / j) + q % j, partner x + j), four pairs in flight
          // per thread: as a loop over elements with a dependent read -> write per iteration a step took 1.7 us
#pragma unroll 4
          for (int q = threadIdx.x; q < (m >> 1); q += GL_BLOCK) {
            const int x = ((q & ~(j - 1)) << 1) | (q & (j - 1)), y = x + j;
            const int32_t kx = s_key[x], ky = s_key[y];
            const bool up = (x & k) == 0;
            if ((kx > ky) == up) { s_key[x] = ky; s_key[y] = kx; }
          }
          __syncthreads();
        }
      for (int j = threadIdx.x; j < len; j += GL_BLOCK) csc_edge[p0 + j] = s_key[j];
      __syncthreads();
    } else {
      // beyond the LDS: every thread ranks its entries against the segment, staged through LDS in chunks (O(len^2 / 256) per thread)
      for (int a0 = 0; a0 < len; a0 += GL_BLOCK) {
        const int a = a0 + threadIdx.x;
        const int32_t v = a < len ? tmp[p0 + a] : 0x7fffffff;
        int rank = 0;
        for (int b0 = 0; b0 < len; b0 += GL_SORT_LDS) {
          const int cl = min(GL_SORT_LDS, len - b0);
          __syncthreads();
          for (int j = threadIdx.x; j < cl; j += GL_BLOCK) s_key[j] = tmp[p0 + b0 + j];
          __syncthreads();
          for (int j = 0; j < cl; ++j) rank += s_key[j] < v ? 1 : 0;
        }
        if (a < len) csc_edge[p0 + rank] = v;
      }
      __syncthreads();
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) csc_ptr[n] = cursor[n - 1];
}

// offsets of a scan over n + 1 elements applied in place; element n (a zero before the scan) ends up as the total
__global__ __launch_bounds__(GL_BLOCK) void gl_scan_offsets_kernel(int64_t n1, int32_t* __restrict__ data, const int32_t* __restrict__ block_sum) {
  const int64_t i = (int64_t)blockIdx.x * GL_BLOCK + threadIdx.x;
  if (i < n1) data[i] += block_sum[i / GL_SCAN_TILE];
}

// short vectors: one workgroup, one launch (a molecule-sized frame: the multi-launch scan above costs 18 us of
// boundaries for 3k integers)
__global__ __launch_bounds__(1024) void gl_scan_small_kernel(int n, const int32_t* __restrict__ in, int32_t* __restrict__ out) {
  __shared__ int32_t s[1024];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < n; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int32_t v = i < n ? in[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int32_t add = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < n) out[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += s[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = carry;
}

// ---- live-edge partition (round 4): a STABLE partition of the slots of a padded list into live (edges > 0) and dead,
// the row order of the compacted edge kernels.  Three launches: live count per block of 1024 slots, exclusive scan of
// the block counts (+ the total = n_live), fill.  Integer work only; the order is the slot order by construction.
__global__ __launch_bounds__(GL_BLOCK) void gl_live_count_kernel(int64_t n, const float* __restrict__ edges, int32_t* __restrict__ bcnt) {
  __shared__ int32_t s[GL_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * GL_SCAN_TILE + (int64_t)threadIdx.x * GL_SCAN_ITEMS;
  int32_t c = 0;
#pragma unroll
  for (int k = 0; k < GL_SCAN_ITEMS; ++k) c += (base + k < n && edges[base + k] > 0.f) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t t = 0;
    for (int w = 0; w < GL_BLOCK / 64; ++w) t += s[w];
    bcnt[blockIdx.x] = t;
  }
}

// one block: exclusive scan of the block counts in place, total -> *total
__global__ __launch_bounds__(GL_BLOCK) void gl_live_scan_kernel(int nblocks, int32_t* __restrict__ bcnt, int32_t* __restrict__ total) {
  __shared__ int32_t s[GL_BLOCK];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblocks; b0 += GL_BLOCK) {
    const int i = b0 + threadIdx.x;
    const int32_t v = i < nblocks ? bcnt[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < GL_BLOCK; off <<= 1) {
      const int32_t add = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nblocks) bcnt[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == GL_BLOCK - 1) carry += s[GL_BLOCK - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// slot g: live -> perm[rank] = g, pos[g] = rank, d_c[rank] = edges[g];  dead -> perm[n_live + (g - live slots before g)] = g,
// pos[g] = -1
__global__ __launch_bounds__(GL_BLOCK) void gl_live_fill_kernel(int64_t n, const float* __restrict__ edges, const int32_t* __restrict__ boff,
                                                                const int32_t* __restrict__ total, int32_t* __restrict__ perm,
                                                                int32_t* __restrict__ pos, float* __restrict__ d_c) {
  __shared__ int32_t s[GL_BLOCK];
  const int64_t base = (int64_t)blockIdx.x * GL_SCAN_TILE + (int64_t)threadIdx.x * GL_SCAN_ITEMS;
  float d[GL_SCAN_ITEMS];
  int32_t c = 0;
#pragma unroll
  for (int k = 0; k < GL_SCAN_ITEMS; ++k) { d[k] = base + k < n ? edges[base + k] : 0.f; c += d[k] > 0.f ? 1 : 0; }
  s[threadIdx.x] = c;
  __syncthreads();
  for (int off = 1; off < GL_BLOCK; off <<= 1) {
    const int32_t add = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += add;
    __syncthreads();
  }
  int32_t rank = boff[blockIdx.x] + s[threadIdx.x] - c;      // live slots before this thread's first slot
  const int32_t n_live = *total;
#pragma unroll
  for (int k = 0; k < GL_SCAN_ITEMS; ++k) {
    const int64_t g = base + k;
    if (g >= n) break;
    if (d[k] > 0.f) {
      perm[rank] = (int32_t)g; pos[g] = rank; d_c[rank] = d[k];
      ++rank;
    } else {
      perm[n_live + (g - rank)] = (int32_t)g; pos[g] = -1;
    }
  }
}

// ---- one graph per call (the reference's own granularity, nmrgnn/library.py:88-89): both builders in ONE launch each.  A 256-atom
// graph has 4096 slots; the multi-launch forms above spend their time on launch boundaries (nine launches, ~40 us of a 0.37-ms
// training step: profiles/r04n_one_graph_trace.txt).
constexpr int GL_SMALL_THREADS = 1024, GL_SMALL_SLOTS = 16384, GL_SMALL_TARGETS = 2048;

// live partition of <= GL_SMALL_SLOTS slots: thread t owns slots [t * per, t * per + per)
__device__ __forceinline__ void gl_live_small_body(int n, const float* __restrict__ edges, int32_t* __restrict__ perm,
                                                   int32_t* __restrict__ pos, float* __restrict__ d_c,
                                                   int32_t* __restrict__ total, int32_t* s /* [GL_SMALL_THREADS / 64] */) {
  const int per = (n + GL_SMALL_THREADS - 1) / GL_SMALL_THREADS;      // <= 16
  const int b0 = threadIdx.x * per, b1 = min(n, b0 + per);
  int32_t c = 0;
  for (int g = b0; g < b1; ++g) c += edges[g] > 0.f ? 1 : 0;
  // inclusive scan of the per-thread counts: shuffles inside a wave, one barrier for the wave totals (round 5; the
  // Hillis-Steele form over LDS took twenty barriers)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int32_t incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s[wv] = incl;
  __syncthreads();
  int32_t wbase = 0, n_live = 0;
  for (int w = 0; w < GL_SMALL_THREADS / 64; ++w) {
    const int32_t v = s[w];
    if (w < wv) wbase += v;
    n_live += v;
  }
  int32_t rank = wbase + incl - c;
  for (int g = b0; g < b1; ++g) {
    const float d = edges[g];
    if (d > 0.f) { perm[rank] = g; pos[g] = rank; d_c[rank] = d; ++rank; }
    else { perm[n_live + (g - rank)] = g; pos[g] = -1; }
  }
  if (threadIdx.x == 0) *total = n_live;
}

__global__ __launch_bounds__(GL_SMALL_THREADS) void gl_live_small_kernel(int n, const float* __restrict__ edges, int32_t* __restrict__ perm,
                                                                         int32_t* __restrict__ pos, float* __restrict__ d_c,
                                                                         int32_t* __restrict__ total) {
  __shared__ int32_t s[GL_SMALL_THREADS / 64];
  gl_live_small_body(n, edges, perm, pos, d_c, total, s);
}

// incoming lists of <= GL_SMALL_TARGETS targets and <= GL_SMALL_SLOTS entries: histogram (LDS integer atomics: order-free), scan,
// unordered fill through LDS cursors, per-target rank sort back into ascending entry id — the stable order of the big form
__device__ __forceinline__ void gl_lists_small_body(int n_entries, int N, int K, const int32_t* __restrict__ nlist,
                                                    const float* __restrict__ edges, int32_t* __restrict__ nlist_c,
                                                    int32_t* __restrict__ csc_ptr, int32_t* __restrict__ csc_edge,
                                                    int32_t* s_cnt, int32_t* s_cur, int32_t* s_scan, int32_t* s_tmp) {
  for (int t = threadIdx.x; t < N; t += GL_SMALL_THREADS) s_cnt[t] = 0;
  __syncthreads();
  for (int eid = threadIdx.x; eid < n_entries; eid += GL_SMALL_THREADS) {
    const int32_t t = nlist[eid];
    const bool live = (edges == nullptr || edges[eid] > 0.f) && t >= 0 && t < N;
    if (nlist_c) nlist_c[eid] = live ? t : eid / K;
    if (live) atomicAdd(&s_cnt[t], 1);
  }
  __syncthreads();
  // exclusive scan over the N counts: thread t owns targets [2t, 2t + 2); wave scans by DPP-free shuffles, one barrier for the
  // wave totals (round 5: the Hillis-Steele form took 2 log2(threads) barriers — a third of the kernel's 30 us)
  const int t0 = 2 * threadIdx.x;
  const int32_t c0 = t0 < N ? s_cnt[t0] : 0, c1 = t0 + 1 < N ? s_cnt[t0 + 1] : 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int32_t incl = c0 + c1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_scan[wv] = incl;
  __syncthreads();
  int32_t wbase = 0, total = 0;
  for (int w = 0; w < GL_SMALL_THREADS / 64; ++w) {
    const int32_t v = s_scan[w];
    if (w < wv) wbase += v;
    total += v;
  }
  const int32_t p0 = wbase + incl - c0 - c1;
  if (t0 < N) { csc_ptr[t0] = p0; s_cur[t0] = p0; }
  if (t0 + 1 < N) { csc_ptr[t0 + 1] = p0 + c0; s_cur[t0 + 1] = p0 + c0; }
  if (threadIdx.x == 0) csc_ptr[N] = total;
  __syncthreads();
  for (int eid = threadIdx.x; eid < n_entries; eid += GL_SMALL_THREADS) {
    const int32_t t = nlist[eid];
    if ((edges != nullptr && !(edges[eid] > 0.f)) || t < 0 || t >= N) continue;
    s_tmp[atomicAdd(&s_cur[t], 1)] = eid;
  }
  __syncthreads();
  // per-target rank sort back into ascending entry id, a 16-lane group per target (as gl_sort_kernel): one thread per target
  // walked its d^2 comparisons alone
  const int g = threadIdx.x >> 4, gl = threadIdx.x & 15;
  for (int t = g; t < N; t += GL_SMALL_THREADS / 16) {
    const int q1 = s_cur[t], len = s_cnt[t], q0 = q1 - len;
    for (int a = 0; a < len; a += 16) {
      const int32_t v = a + gl < len ? s_tmp[q0 + a + gl] : 0x7fffffff;
      int rank = 0;
      for (int b = 0; b < len; b += 16) {
        const int32_t u = b + gl < len ? s_tmp[q0 + b + gl] : 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 16; ++r) rank += __shfl(u, r, 16) < v ? 1 : 0;
      }
      if (a + gl < len) csc_edge[q0 + rank] = v;
    }
  }
}

__global__ __launch_bounds__(GL_SMALL_THREADS) void gl_lists_small_kernel(int n_entries, int N, int K, const int32_t* __restrict__ nlist,
                                                                          const float* __restrict__ edges, int32_t* __restrict__ nlist_c,
                                                                          int32_t* __restrict__ csc_ptr, int32_t* __restrict__ csc_edge) {
  __shared__ int32_t s_cnt[GL_SMALL_TARGETS], s_cur[GL_SMALL_TARGETS], s_scan[GL_SMALL_THREADS / 64];
  extern __shared__ int32_t s_tmp[];          // [n_entries]
  gl_lists_small_body(n_entries, N, K, nlist, edges, nlist_c, csc_ptr, csc_edge, s_cnt, s_cur, s_scan, s_tmp);
}

// both builders of a one-graph call in ONE launch (round 6: ng_build_graph_lists): the live partition reads only `edges`, the
// incoming lists only `nlist` / `edges` — neither reads what the other writes
__global__ __launch_bounds__(GL_SMALL_THREADS) void gl_both_small_kernel(int n_entries, int N, int K, const int32_t* __restrict__ nlist,
                                                                         const float* __restrict__ edges, int32_t* __restrict__ nlist_c,
                                                                         int32_t* __restrict__ csc_ptr, int32_t* __restrict__ csc_edge,
                                                                         int32_t* __restrict__ perm, int32_t* __restrict__ pos,
                                                                         float* __restrict__ d_c, int32_t* __restrict__ n_live) {
  __shared__ int32_t s_cnt[GL_SMALL_TARGETS], s_cur[GL_SMALL_TARGETS], s_scan[GL_SMALL_THREADS / 64], s_live[GL_SMALL_THREADS / 64];
  extern __shared__ int32_t s_tmp[];          // [n_entries]
  gl_live_small_body(n_entries, edges, perm, pos, d_c, n_live, s_live);
  gl_lists_small_body(n_entries, N, K, nlist, edges, nlist_c, csc_ptr, csc_edge, s_cnt, s_cur, s_scan, s_tmp);
}

}  // namespace ng

using namespace ng;

extern "C" int ng_build_live_edges(ng_ctx* ctx, void* stream, int64_t n_slots, const float* edges, int32_t* perm, int32_t* pos,
                                   float* d_c, int32_t* n_live) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, n_slots >= 0 && n_slots < ((int64_t)1 << 31) - GL_SCAN_TILE, "live edges: slot count out of range");
  NG_REQUIRE(ctx, n_live && (n_slots == 0 || (edges && perm && pos && d_c)), "live edges: arguments");
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard dg(ctx->device);
  if (n_slots == 0) { NG_HIP(ctx, hipMemsetAsync(n_live, 0, sizeof(int32_t), st)); return NG_OK; }
  if (n_slots <= GL_SMALL_SLOTS) {      // one graph: one launch
    ProfScope ps(ctx, st, "live_edges");
    hipLaunchKernelGGL(gl_live_small_kernel, dim3(1), dim3(GL_SMALL_THREADS), 0, st, (int)n_slots, edges, perm, pos, d_c, n_live);
    NG_HIP(ctx, hipGetLastError());
    return NG_OK;
  }
  const int nb = (int)cdiv(n_slots, GL_SCAN_TILE);
  int32_t* bcnt = (int32_t*)aux_workspace(ctx, (size_t)(nb + 16) * sizeof(int32_t));
  if (!bcnt) return NG_ERR_HIP;
  ProfScope ps(ctx, st, "live_edges");
  hipLaunchKernelGGL(gl_live_count_kernel, dim3(nb), dim3(GL_BLOCK), 0, st, n_slots, edges, bcnt);
  hipLaunchKernelGGL(gl_live_scan_kernel, dim3(1), dim3(GL_BLOCK), 0, st, nb, bcnt, n_live);
  hipLaunchKernelGGL(gl_live_fill_kernel, dim3(nb), dim3(GL_BLOCK), 0, st, n_slots, edges, bcnt, n_live, perm, pos, d_c);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

static bool gl_small_shape(int64_t N, int K, int64_t n_entries) {
  return N >= 1 && N <= GL_SMALL_TARGETS && n_entries <= GL_SMALL_SLOTS && n_entries > 0 && K > 0;
}

// ng_build_incoming_lists + ng_build_live_edges of the padded form as one call; one launch at molecule size
extern "C" int ng_build_graph_lists(ng_ctx* ctx, void* stream, int64_t N, int K, const int32_t* nlist, const float* edges,
                                    int32_t* nlist_c, int32_t* csc_ptr, int32_t* csc_edge, int32_t* perm, int32_t* pos, float* d_c,
                                    int32_t* n_live) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, N >= 0 && K > 0 && N * (int64_t)K < ((int64_t)1 << 31), "graph lists: padded form, sizes in range");
  NG_REQUIRE(ctx, csc_ptr && csc_edge && perm && pos && d_c && n_live && (N == 0 || (nlist && edges)), "graph lists: arguments");
  const int64_t n_entries = N * K;
  if (gl_small_shape(N, K, n_entries)) {
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard dg(ctx->device);
    ProfScope ps(ctx, st, "graph_lists");
    hipLaunchKernelGGL(gl_both_small_kernel, dim3(1), dim3(GL_SMALL_THREADS), (size_t)n_entries * sizeof(int32_t), st, (int)n_entries,
                       (int)N, K, nlist, edges, nlist_c, csc_ptr, csc_edge, perm, pos, d_c, n_live);
    NG_HIP(ctx, hipGetLastError());
    return NG_OK;
  }
  const int rc = ng_build_incoming_lists(ctx, stream, N, K, n_entries, nlist, edges, nlist_c, csc_ptr, csc_edge);
  if (rc) return rc;
  return ng_build_live_edges(ctx, stream, n_entries, edges, perm, pos, d_c, n_live);
}

extern "C" int ng_graph_lists_one_launch(int64_t N, int K) { return gl_small_shape(N, K, N * (int64_t)K) ? 1 : 0; }

// out[0..n] = exclusive prefix sums of in[0..n-1] (out[n] = total): row_ptr from a degree vector, on the device
extern "C" int ng_exclusive_scan_i32(ng_ctx* ctx, void* stream, int64_t n, const int32_t* in, int32_t* out) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, n >= 0 && n < ((int64_t)1 << 31) - 1 && out && (in || n == 0), "exclusive scan: arguments");
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard dg(ctx->device);
  const int64_t n1 = n + 1;
  const int nb = (int)cdiv(n1, GL_SCAN_TILE);
  int32_t* bsum = (int32_t*)aux_workspace(ctx, (size_t)(nb + 16) * sizeof(int32_t));
  if (!bsum) return NG_ERR_HIP;
  ProfScope ps(ctx, st, "exclusive_scan");
  if (n <= 32768 && in != out) {
    hipLaunchKernelGGL(gl_scan_small_kernel, dim3(1), dim3(1024), 0, st, (int)n, in, out);
    NG_HIP(ctx, hipGetLastError());
    return NG_OK;
  }
  if (n > 0) NG_HIP(ctx, hipMemcpyAsync(out, in, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  NG_HIP(ctx, hipMemsetAsync(out + n, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(gl_scan_local_kernel, dim3(nb), dim3(GL_BLOCK), 0, st, n1, out, bsum);
  hipLaunchKernelGGL(gl_scan_sums_kernel, dim3(1), dim3(GL_BLOCK), 0, st, nb, bsum);
  hipLaunchKernelGGL(gl_scan_offsets_kernel, dim3((unsigned)cdiv(n1, GL_BLOCK)), dim3(GL_BLOCK), 0, st, n1, out, bsum);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" size_t ng_incoming_lists_scratch_bytes(int64_t N, int64_t n_entries) {
  const int64_t nb = cdiv(std::max<int64_t>(N, 1), GL_SCAN_TILE);
  return (size_t)(2 * N + nb + n_entries + 16) * sizeof(int32_t);
}

extern "C" int ng_build_incoming_lists(ng_ctx* ctx, void* stream, int64_t N, int K, int64_t n_entries, const int32_t* nlist,
                                       const float* edges, int32_t* nlist_c, int32_t* csc_ptr, int32_t* csc_edge) {
  NG_REQUIRE(ctx, N >= 0 && n_entries >= 0 && n_entries < ((int64_t)1 << 31), "incoming lists: sizes out of range");
  NG_REQUIRE(ctx, csc_ptr && csc_edge, "incoming lists: csc_ptr / csc_edge required");
  NG_REQUIRE(ctx, nlist_c == nullptr || K > 0, "incoming lists: nlist_c needs the padded form (K > 0)");
  NG_REQUIRE(ctx, K == 0 || n_entries == N * K, "incoming lists: padded form has N*K entries");
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard dg(ctx->device);
  if (N == 0) { NG_HIP(ctx, hipMemsetAsync(csc_ptr, 0, sizeof(int32_t), st)); return NG_OK; }
  if (N <= GL_SMALL_TARGETS && n_entries <= GL_SMALL_SLOTS && n_entries > 0 && K > 0) {      // one graph: one launch, no scratch
    ProfScope ps(ctx, st, "incoming_lists");
    hipLaunchKernelGGL(gl_lists_small_kernel, dim3(1), dim3(GL_SMALL_THREADS), (size_t)n_entries * sizeof(int32_t), st, (int)n_entries,
                       (int)N, K, nlist, edges, nlist_c, csc_ptr, csc_edge);
    NG_HIP(ctx, hipGetLastError());
    return NG_OK;
  }
  const int nb = (int)cdiv(N, GL_SCAN_TILE);
  int32_t* ws = (int32_t*)aux_workspace(ctx, ng_incoming_lists_scratch_bytes(N, n_entries));
  if (!ws) return NG_ERR_HIP;
  int32_t* count = ws;                 // [N]  counts -> local prefixes
  int32_t* cursor = count + N;         // [N]
  int32_t* bsum = cursor + N;          // [nb]
  int32_t* tmp = bsum + nb;            // [n_entries]
  ProfScope ps(ctx, st, "incoming_lists");
  NG_HIP(ctx, hipMemsetAsync(count, 0, (size_t)N * sizeof(int32_t), st));
  if (n_entries > 0)
    hipLaunchKernelGGL(gl_count_kernel, dim3((unsigned)cdiv(n_entries, GL_BLOCK)), dim3(GL_BLOCK), 0, st, n_entries, N, K, nlist, edges,
                       nlist_c, count);
  hipLaunchKernelGGL(gl_scan_local_kernel, dim3(nb), dim3(GL_BLOCK), 0, st, N, count, bsum);
  hipLaunchKernelGGL(gl_scan_sums_kernel, dim3(1), dim3(GL_BLOCK), 0, st, nb, bsum);
  hipLaunchKernelGGL(gl_scan_apply_kernel, dim3((unsigned)cdiv(N, GL_BLOCK)), dim3(GL_BLOCK), 0, st, N, count, bsum, csc_ptr,
                     cursor);
  if (n_entries > 0)
    hipLaunchKernelGGL(gl_fill_kernel, dim3((unsigned)cdiv(n_entries, GL_BLOCK)), dim3(GL_BLOCK), 0, st, n_entries, N, nlist, edges,
                       cursor, tmp);
  hipLaunchKernelGGL(gl_sort_kernel, dim3((unsigned)cdiv(N, GL_GROUPS)), dim3(GL_BLOCK), 0, st, N, cursor, csc_ptr, tmp, csc_edge);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}
