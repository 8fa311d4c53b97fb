// LDS-window neighbour aggregation: the gather + edge-weighted segment sum of MPLayer
//     A[i][n][l] = sum_j e[i][j][n] * src[nlist[i][j]][l]          (nmrgnn/layers.py:33 + ij-part of 39-40)
// and its transposed form over incoming-edge lists (backward to the nodes)
//     B[t][n][m] = sum_{p in csc[t]} e[p][n] * src[p / K][m].
//
// Molecule batches have index-local neighbour lists (every neighbour of an atom lies in the atom's
// own graph), so a workgroup that owns a run of atoms first finds the [lo, hi] range of rows its lists
// reference, stages that window of `src` in LDS with coalesced, deep-in-flight loads (each row is
// read from HBM/L2 ONCE per workgroup instead of once per referencing edge), and then gathers from
// LDS.  A run whose window does not fit (whole proteins with spatial kNN lists) falls back to
// gathering from global memory — same kernel, workgroup-uniform branch.
//
// Layout: F/4 lanes per atom (float4 of features each), 256/(F/4) atoms per round; a 16-lane group
// reads one 256-B row = all 64 LDS banks once, and ds_read_b128's lane groups take disjoint 16-B slots
// of different rows, so the LDS gathers are conflict-free without padding.
#include <algorithm>

#include "mfma_gemm.cuh"
#include "ng_internal.h"

namespace ng {

constexpr int AW_MAX_E = 8;
constexpr int AW_LDS_BYTES = 64 * 1024;   // window budget per workgroup -> two workgroups per CU

struct AggWinArgs {
  int64_t N;
  int K, F;
  int atoms_per_tile;      // run of atoms owned by one workgroup
  int win_rows;            // AW_LDS_BYTES / (F*4)
  const float* src;        // [N][F]
  const int32_t* nlist;    // fixed-K lists  (RAGGED = false)
  const int32_t* ptr;      // csc_ptr[N+1]   (RAGGED = true)
  const int32_t* eids;     // csc_edge[nnz]
  const float* e;          // [N*K][E]
  float* A;                // [N][E][F]
};

template <int E, bool RAGGED>
__global__ __launch_bounds__(256) void aggregate_window_kernel(AggWinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float win[];
  __shared__ int s_red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c4n = a.F / 4;
  const int apr = 256 / c4n;   // atoms per round
  const int64_t i0 = (int64_t)blockIdx.x * a.atoms_per_tile;
  const int64_t i1 = std::min<int64_t>(i0 + a.atoms_per_tile, a.N);

  // ---- 1. range of source rows referenced by this run
  int lo = 0x7fffffff, hi = -1;
  if (!RAGGED) {
    const int64_t b = i0 * a.K, en = i1 * a.K;
    for (int64_t t = b + tid; t < en; t += 256) {
      const int v = a.nlist[t];
      lo = min(lo, v); hi = max(hi, v);
    }
  } else {
    const int b = a.ptr[i0], en = a.ptr[i1];
    for (int t = b + tid; t < en; t += 256) {
      const int v = a.eids[t] / a.K;
      lo = min(lo, v); hi = max(hi, v);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
  }
  if (lane == 0) { s_red[0][wave] = lo; s_red[1][wave] = hi; }
  __syncthreads();
  lo = min(min(s_red[0][0], s_red[0][1]), min(s_red[0][2], s_red[0][3]));
  hi = max(max(s_red[1][0], s_red[1][1]), max(s_red[1][2], s_red[1][3]));
  const bool use_win = hi >= lo && (hi - lo + 1) <= a.win_rows;

  // ---- 2. stage the window (whole rows, coalesced float4, all loads of a thread in flight at once)
  const float4* src4 = reinterpret_cast<const float4*>(a.src);
  float4* win4 = reinterpret_cast<float4*>(win);
  if (use_win) {
    const int n4 = (hi - lo + 1) * c4n;
    const float4* g = src4 + (int64_t)lo * c4n;
    for (int t = tid; t < n4; t += 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (t + u * 256 < n4) ? g[t + u * 256] : f4zero();
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (t + u * 256 < n4) win4[t + u * 256] = v[u];
    }
  }
  __syncthreads();

  // ---- 3. gather + weighted sum
  const int al = tid / c4n, c = tid % c4n;
  float4* A4 = reinterpret_cast<float4*>(a.A);
  for (int64_t base = i0; base < i1; base += apr) {
    const int64_t i = base + al;
    if (i >= i1) continue;
    float4 acc[E];
#pragma unroll
    for (int n = 0; n < E; ++n) acc[n] = f4zero();
    int p0, p1;
    if (!RAGGED) { p0 = 0; p1 = a.K; } else { p0 = a.ptr[i]; p1 = a.ptr[i + 1]; }
    for (int q0 = p0; q0 < p1; q0 += 8) {
      int row[8];
      int64_t eoff[8];
      float4 hv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + u < p1 ? q0 + u : p1 - 1;
        if (!RAGGED) {
          row[u] = a.nlist[i * a.K + q];
          eoff[u] = (i * a.K + q) * E;
        } else {
          const int eid = a.eids[q];
          row[u] = eid / a.K;
          eoff[u] = (int64_t)eid * E;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        hv[u] = use_win ? win4[(row[u] - lo) * c4n + c] : src4[(int64_t)row[u] * c4n + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (q0 + u < p1) {
#pragma unroll
          for (int n = 0; n < E; ++n) {
            const float ev = a.e[eoff[u] + n];
            acc[n].x += ev * hv[u].x; acc[n].y += ev * hv[u].y;
            acc[n].z += ev * hv[u].z; acc[n].w += ev * hv[u].w;
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < E; ++n) A4[(i * E + n) * c4n + c] = acc[n];
  }
}

template <bool RAGGED>
static int launch_agg_window(ng_ctx* ctx, hipStream_t st, int E, const AggWinArgs& a, const char* tag) {
  const dim3 grid((unsigned)cdiv(a.N, a.atoms_per_tile));
  const size_t lds = (size_t)a.win_rows * a.F * 4;
  ProfScope ps(ctx, st, tag);
#define NG_AW(EE)                                                                                    \
  case EE:                                                                                           \
    hipLaunchKernelGGL((aggregate_window_kernel<EE, RAGGED>), grid, dim3(256), lds, st, a);          \
    break;
  switch (E) { NG_AW(1) NG_AW(2) NG_AW(3) NG_AW(4) NG_AW(5) NG_AW(6) NG_AW(7) NG_AW(8) }
#undef NG_AW
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

bool aggregate_window_supported(int F, int E) {
  return F % 4 == 0 && F >= 16 && F <= 1024 && (256 % (F / 4)) == 0 && E >= 1 && E <= AW_MAX_E;
}

static AggWinArgs make_args(int64_t N, int K, int F) {
  AggWinArgs a{};
  a.N = N; a.K = K; a.F = F;
  a.win_rows = AW_LDS_BYTES / (F * 4);
  // a run as long as the window: for 256-atom molecules at F = 64 one workgroup owns one molecule
  a.atoms_per_tile = std::max(a.win_rows, 256 / (F / 4));
  return a;
}

int aggregate_window(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* src,
                     const int32_t* nlist, const float* e, float* A) {
  if (N == 0) return NG_OK;
  AggWinArgs a = make_args(N, K, F);
  a.src = src; a.nlist = nlist; a.e = e; a.A = A;
  return launch_agg_window<false>(ctx, st, E, a, "mp_aggregate");
}

int aggregate_window_csc(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* src,
                         const int32_t* csc_ptr, const int32_t* csc_edge, const float* e, float* B) {
  if (N == 0) return NG_OK;
  AggWinArgs a = make_args(N, K, F);
  a.src = src; a.ptr = csc_ptr; a.eids = csc_edge; a.e = e; a.A = B;
  return launch_agg_window<true>(ctx, st, E, a, "mp_aggregate_csc");
}

}  // namespace ng
