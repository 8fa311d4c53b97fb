// MPLayer over CSR (variable-degree) neighbour lists, and the distance-cutoff graph builder that produces them.
//
// The reference feeds padded [N,K] lists (nmrgnn/library.py:106-117) whose padded slots carry edges == 0 and, through
// the edge mask (nmrgnn/model.py:251,261), contribute exactly 0 to the contraction of nmrgnn/layers.py:39-40.  Dropping
// them gives the CSR form of SURVEY §8(b):  row i owns the entries p in [row_ptr[i], row_ptr[i+1]),  col[p] = neighbour
// atom,  dist[p] = distance;  edge features e[nnz][E] come from ng_edge_mlp_fwd on the flat dist array.
//
//   forward    A[i][n][:]  = sum_{p in row i} e[p][n] * h[col[p]][:]           (gather, this file)
//              h'          = act(v * A Wp) (+ h)                               (GEMM, gemm_ops / gemm_h2)
//   backward   dP = dH*act'(S)*v ; dw = A^T dP ; dA = dP Wp^T                  (GEMMs)
//              de[p][n]   (+)= <dA[row(p)][n][:], h[col[p]][:]>                (gather-dot, this file)
//              dh[t][:]    = dH[t][:] + sum_{q in csc[t]} sum_n e[p_q][n] dA[row(p_q)][n][:]   (pull scatter, this file)
//
// The kernels take the row extent either from row_ptr or, when row_ptr == nullptr, as the fixed stride K (row i =
// [i*K, (i+1)*K)): the padded layout is the special case, which is also how edge_feature_size > 8 is served for
// padded lists.  Edge features are processed in chunks of at most 8 (EC) so that edge_feature_size = 64
// (nmrgnn/model.py:23) runs through the same code.  Summation order inside a row is the entry order, so on lists that
// differ only by dropped zero-weight slots the CSR and the padded generic paths agree bit for bit.
#include <algorithm>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "reduce.cuh"

namespace ng {

namespace {

struct RowRange {
  const int32_t* row_ptr;   // [N+1] or nullptr
  int K;                    // fixed stride when row_ptr == nullptr
  __device__ __forceinline__ void get(int64_t i, int64_t& p0, int64_t& p1) const {
    if (row_ptr) { p0 = row_ptr[i]; p1 = row_ptr[i + 1]; }
    else { p0 = i * K; p1 = p0 + K; }
  }
};

// A[i][n0+n][:] = sum_p e[p][n0+n] * h[col[p]][:]     F/4 lanes per atom, 8 row gathers in flight per lane
template <int EC>
__global__ __launch_bounds__(256) void csr_aggregate_kernel(int64_t N, int F, int E, int n0, RowRange rr,
                                                            const float* __restrict__ h,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ e, float* __restrict__ A) {
  const int c4n = F / 4;
  const int apb = 256 / c4n;
  const int a = threadIdx.x / c4n, c4 = threadIdx.x % c4n;
  const int64_t i = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * apb + a;   // contiguous atom ranges per XCD (one L2)
  if (i >= N) return;
  int64_t p0, p1;
  rr.get(i, p0, p1);
  float4 acc[EC];
#pragma unroll
  for (int n = 0; n < EC; ++n) acc[n] = f4zero();
  const float4* h4 = reinterpret_cast<const float4*>(h);
  for (int64_t q0 = p0; q0 < p1; q0 += 8) {
    float4 hv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t p = q0 + u < p1 ? q0 + u : p1 - 1;
      hv[u] = h4[(int64_t)col[p] * c4n + c4];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (q0 + u < p1) {
        const float* ep = e + (q0 + u) * E + n0;
#pragma unroll
        for (int n = 0; n < EC; ++n) {
          const float ev = ep[n];
          acc[n].x += ev * hv[u].x; acc[n].y += ev * hv[u].y;
          acc[n].z += ev * hv[u].z; acc[n].w += ev * hv[u].w;
        }
      }
    }
  }
  typedef float nt4 __attribute__((ext_vector_type(4)));     // written once, read by a later kernel
  nt4* A4 = reinterpret_cast<nt4*>(A);
#pragma unroll
  for (int n = 0; n < EC; ++n)
    __builtin_nontemporal_store(nt4{acc[n].x, acc[n].y, acc[n].z, acc[n].w}, A4 + (i * E + n0 + n) * c4n + c4);
}

// de[p][n0+n] (+)= sum_l dA[i][n0+n][l] * h[col[p]][l]   for the entries p of row i
template <int EC>
__global__ __launch_bounds__(256) void csr_edge_grad_kernel(int64_t N, int F, int E, int n0, RowRange rr,
                                                            const float* __restrict__ h,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ dA, float* __restrict__ de,
                                                            int accumulate) {
  const int c4n = F / 4;   // power of two <= 64: the lanes of an atom sit inside one wave
  const int apb = 256 / c4n;
  const int a = threadIdx.x / c4n, c4 = threadIdx.x % c4n;
  const int64_t i = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * apb + a;
  const bool live = i < N;
  const int64_t ii = live ? i : 0;
  int64_t p0, p1;
  rr.get(ii, p0, p1);
  if (!live) p1 = p0;
  // every lane of a wave must take part in the shuffles: iterate to the longest row of the wave's atoms
  int len = (int)(p1 - p0), maxlen = len;
  for (int off = 32; off >= c4n; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
  const float4* dA4 = reinterpret_cast<const float4*>(dA);
  const float4* h4 = reinterpret_cast<const float4*>(h);
  float4 g[EC];
#pragma unroll
  for (int n = 0; n < EC; ++n) g[n] = dA4[(ii * E + n0 + n) * c4n + c4];
  for (int j = 0; j < maxlen; ++j) {
    const bool has = j < len;
    const int64_t p = has ? p0 + j : (len > 0 ? p0 : 0);
    const float4 hv = (has || len > 0) ? h4[(int64_t)col[p] * c4n + c4] : f4zero();
    float part[EC];
#pragma unroll
    for (int n = 0; n < EC; ++n) part[n] = g[n].x * hv.x + g[n].y * hv.y + g[n].z * hv.z + g[n].w * hv.w;
    for (int off = c4n >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int n = 0; n < EC; ++n) part[n] += __shfl_xor(part[n], off, 64);
    }
    if (has && c4 == 0) {
#pragma unroll
      for (int n = 0; n < EC; ++n) {
        const int64_t o = p * E + n0 + n;
        de[o] = accumulate ? de[o] + part[n] : part[n];
      }
    }
  }
}

// dh_in[t][:] = base + sum_{q in csc[t]} sum_n e[eid][n0+n] * dA[row(eid)][n0+n][:]
//   base = dh_out[t] for the first feature chunk, dh_in[t] (running sum) for the following ones
template <int EC>
__global__ __launch_bounds__(256) void csr_scatter_pull_kernel(int64_t N, int F, int E, int n0, int K,
                                                               const int32_t* __restrict__ row_of,
                                                               const int32_t* __restrict__ csc_ptr,
                                                               const int32_t* __restrict__ csc_edge,
                                                               const float* __restrict__ e,
                                                               const float* __restrict__ dA,
                                                               const float* __restrict__ base,
                                                               float* __restrict__ dh_in) {
  const int c4n = F / 4;
  const int apb = 256 / c4n;
  const int a = threadIdx.x / c4n, c4 = threadIdx.x % c4n;
  const int64_t t = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * apb + a;
  if (t >= N) return;
  const float4* dA4 = reinterpret_cast<const float4*>(dA);
  float4 acc = reinterpret_cast<const float4*>(base)[t * c4n + c4];
  const int q0 = csc_ptr[t], q1 = csc_ptr[t + 1];
  for (int q = q0; q < q1; ++q) {
    const int eid = csc_edge[q];
    const int64_t src = row_of ? row_of[eid] : eid / K;
    const float* ep = e + (int64_t)eid * E + n0;
#pragma unroll
    for (int n = 0; n < EC; ++n) {
      const float ev = ep[n];
      const float4 v = dA4[(src * E + n0 + n) * c4n + c4];
      acc.x += ev * v.x; acc.y += ev * v.y; acc.z += ev * v.z; acc.w += ev * v.w;
    }
  }
  reinterpret_cast<float4*>(dh_in)[t * c4n + c4] = acc;
}

// ---- wide-lane forms for F >= 128 (the reference's default width): LPA = F/16 lanes per atom, each lane owns four
// float4 of the row (float4 index c + LPA t, t = 0..3, so a load instruction still covers LPA x 16 contiguous bytes).
// The per-edge dot products are reduced over 8 or 16 lanes on the DPP network (3-4 VALU adds) instead of over 64 lanes
// with six ds_bpermute shuffles each (37.7 M LDS operations per launch, 67 % of the wave cycles parked at F = 256),
// and four rows are gathered before the first is consumed.
template <int CTRL>
__device__ __forceinline__ float csr_dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int LPA>
__device__ __forceinline__ float csr_group_sum(float v) {     // sum over aligned groups of LPA lanes, valid in every lane
  if (LPA == 16) {
    v = csr_dpp_add<0x128>(v);     // row_ror:8
    v = csr_dpp_add<0x124>(v);     // row_ror:4
    v = csr_dpp_add<0x122>(v);     // row_ror:2
    v = csr_dpp_add<0x121>(v);     // row_ror:1
  } else {
    v = csr_dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = csr_dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = csr_dpp_add<0x141>(v);     // row_half_mirror
  }
  return v;
}

template <int EC, int LPA>
__global__ __launch_bounds__(256) void csr_edge_grad_wide_kernel(int64_t N, int E, RowRange rr,
                                                                 const float* __restrict__ h,
                                                                 const int32_t* __restrict__ col,
                                                                 const float* __restrict__ dA, float* __restrict__ de,
                                                                 int accumulate) {
  constexpr int C4N = 4 * LPA, APB = 256 / LPA;
  const int a = threadIdx.x / LPA, c = threadIdx.x % LPA;
  const int64_t i = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * APB + a;
  const bool live = i < N;
  const int64_t ii = live ? i : 0;
  int64_t p0, p1;
  rr.get(ii, p0, p1);
  if (!live) p1 = p0;
  const int len = (int)(p1 - p0);
  int maxlen = len;                               // DPP rows are shared: every lane of the wave walks the longest row
  for (int off = 32; off >= LPA; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
  const float4* dA4 = reinterpret_cast<const float4*>(dA);
  const float4* h4 = reinterpret_cast<const float4*>(h);
  float4 g[EC][4];
#pragma unroll
  for (int n = 0; n < EC; ++n)
#pragma unroll
    for (int t = 0; t < 4; ++t) g[n][t] = dA4[(ii * E + n) * C4N + c + LPA * t];
  for (int j0 = 0; j0 < maxlen; j0 += 4) {
    float4 hv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = len > 0 ? p0 + min(j0 + u, len - 1) : 0;
      const int64_t r = len > 0 ? (int64_t)col[p] : 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) hv[u][t] = h4[r * C4N + c + LPA * t];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float mine = 0.f;
#pragma unroll
      for (int n = 0; n < EC; ++n) {
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          part += g[n][t].x * hv[u][t].x + g[n][t].y * hv[u][t].y + g[n][t].z * hv[u][t].z + g[n][t].w * hv[u][t].w;
        part = csr_group_sum<LPA>(part);
        if (c == n) mine = part;                  // lane n of the atom's group stores feature n: EC adjacent floats
      }
      if (j0 + u < len && c < EC) {
        const int64_t o = (p0 + j0 + u) * E + c;
        de[o] = accumulate ? de[o] + mine : mine;
      }
    }
  }
}

// rec (optional, E <= 3): per incoming-edge entry {source row (int bits), e0, e1, e2} — one 16-byte load instead of
// the index chain csc_edge -> row_of -> e
template <int EC, int LPA>
__global__ __launch_bounds__(256) void csr_scatter_pull_wide_kernel(int64_t N, int E, int K,
                                                                    const int32_t* __restrict__ row_of,
                                                                    const int32_t* __restrict__ csc_ptr,
                                                                    const int32_t* __restrict__ csc_edge,
                                                                    const float* __restrict__ e,
                                                                    const float4* __restrict__ rec,
                                                                    const float* __restrict__ dA,
                                                                    const float* __restrict__ base,
                                                                    float* __restrict__ dh_in) {
  constexpr int C4N = 4 * LPA, APB = 256 / LPA;
  const int a = threadIdx.x / LPA, c = threadIdx.x % LPA;
  const int64_t t = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * APB + a;
  if (t >= N) return;
  const float4* dA4 = reinterpret_cast<const float4*>(dA);
  float4 acc[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) acc[tt] = reinterpret_cast<const float4*>(base)[t * C4N + c + LPA * tt];
  const int q0 = csc_ptr[t], q1 = csc_ptr[t + 1];
  for (int q = q0; q < q1; q += 2) {
    float ev[2][EC];
    float4 v[2][EC][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int qq = min(q + u, q1 - 1);
      int64_t src;
      if (rec) {
        const float4 r = rec[qq];
        src = __float_as_int(r.x);
        const float rv[3] = {r.y, r.z, r.w};
#pragma unroll
        for (int n = 0; n < EC; ++n) ev[u][n] = n < 3 ? rv[n < 3 ? n : 0] : 0.f;
      } else {
        const int eid = csc_edge[qq];
        src = row_of ? row_of[eid] : eid / K;
#pragma unroll
        for (int n = 0; n < EC; ++n) ev[u][n] = e[(int64_t)eid * E + n];
      }
      if (q + u >= q1) {
#pragma unroll
        for (int n = 0; n < EC; ++n) ev[u][n] = 0.f;
      }
#pragma unroll
      for (int n = 0; n < EC; ++n)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) v[u][n][tt] = dA4[(src * E + n) * C4N + c + LPA * tt];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int n = 0; n < EC; ++n)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          acc[tt].x += ev[u][n] * v[u][n][tt].x; acc[tt].y += ev[u][n] * v[u][n][tt].y;
          acc[tt].z += ev[u][n] * v[u][n][tt].z; acc[tt].w += ev[u][n] * v[u][n][tt].w;
        }
  }
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) reinterpret_cast<float4*>(dh_in)[t * C4N + c + LPA * tt] = acc[tt];
}

#define NG_EC_SWITCH(EC, CALL)   \
  switch (EC) {                  \
    case 1: { CALL(1) } break;   \
    case 2: { CALL(2) } break;   \
    case 3: { CALL(3) } break;   \
    case 4: { CALL(4) } break;   \
    case 5: { CALL(5) } break;   \
    case 6: { CALL(6) } break;   \
    case 7: { CALL(7) } break;   \
    default: { CALL(8) } break;  \
  }

bool csr_shape_ok(int F) { return F % 4 == 0 && F >= 16 && F <= 256 && (256 % (F / 4)) == 0; }

}  // namespace

int csr_aggregate(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* h, const int32_t* row_ptr,
                  const int32_t* col, const float* e, float* A) {
  NG_REQUIRE(ctx, csr_shape_ok(F), "mp (csr): F in {16,32,64,128,256}");
  NG_REQUIRE(ctx, E >= 1 && E <= 64, "mp (csr): edge_feature_size in [1,64]");
  if (N == 0) return NG_OK;
  ProfScope ps(ctx, st, "mp_aggregate_csr");
  const int apb = 256 / (F / 4);
  const dim3 grid((unsigned)cdiv(N, apb));
  const RowRange rr{row_ptr, K};
  for (int n0 = 0; n0 < E; n0 += 8) {
    const int ec = std::min(8, E - n0);
#define CALL(EE) hipLaunchKernelGGL((csr_aggregate_kernel<EE>), grid, dim3(256), 0, st, N, F, E, n0, rr, h, col, e, A);
    NG_EC_SWITCH(ec, CALL)
#undef CALL
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

int csr_edge_grad(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* h, const int32_t* row_ptr,
                  const int32_t* col, const float* dA, float* de, int accumulate) {
  if (N == 0) return NG_OK;
  ProfScope ps(ctx, st, "mp_edge_grad");
  // padded lists of a batch of small graphs at the default width: slab windows of h in LDS (mp_win.hip)
  if (!row_ptr && agg_win_supported(F, E, K) && N >= 4096 && ctx->graph_span > 0 && ctx->graph_span <= agg_win_rows())
    return egrad_win(ctx, st, N, K, F, E, h, col, dA, de, accumulate);
  const RowRange rr{row_ptr, K};
  if ((F == 128 || F == 256) && E <= 4) {          // wide-lane form: DPP reductions over F/16 lanes
    const int lpa = F / 16;
    const dim3 gridw((unsigned)cdiv(N, 256 / lpa));
#define CALLW(EE)                                                                                                  \
  if (lpa == 16) hipLaunchKernelGGL((csr_edge_grad_wide_kernel<EE, 16>), gridw, dim3(256), 0, st, N, E, rr, h, col, dA, de, accumulate); \
  else hipLaunchKernelGGL((csr_edge_grad_wide_kernel<EE, 8>), gridw, dim3(256), 0, st, N, E, rr, h, col, dA, de, accumulate);
    switch (E) { case 1: { CALLW(1) } break; case 2: { CALLW(2) } break; case 3: { CALLW(3) } break; default: { CALLW(4) } break; }
#undef CALLW
    NG_HIP(ctx, hipGetLastError());
    return NG_OK;
  }
  const int apb = 256 / (F / 4);
  const dim3 grid((unsigned)cdiv(N, apb));
  for (int n0 = 0; n0 < E; n0 += 8) {
    const int ec = std::min(8, E - n0);
#define CALL(EE) \
  hipLaunchKernelGGL((csr_edge_grad_kernel<EE>), grid, dim3(256), 0, st, N, F, E, n0, rr, h, col, dA, de, accumulate);
    NG_EC_SWITCH(ec, CALL)
#undef CALL
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

// ---- window pull (round 5): the backward scatter-sum of the default width with the dA rows in LDS ------------------------------
//   out[t][l] = base[t][l] + sum over the incoming edges (i -> t) of sum_n e_n * dA[i][n][l]        (layers.py:33-40)
// The kernel above pulls 3 KB of dA per incoming edge through L2 — 6.3 GB per layer of the bench batch, 0.37-0.38 ms, "0.23 of
// the HBM roofline" for 0.7 GB of algorithmic traffic: the bound is the L2, every dA row is fetched ~16 times.  Edges never
// leave a graph, so the sources of a run of consecutive targets lie in a short row range (the graph: 256 rows in the bench
// batch; ng_ctx_set_graph_span tells the library), and here each of those rows crosses HBM -> LDS exactly once.
// (First form, measured and replaced: column slabs of 32 with a 288-row window per slab — the targets' accumulators and the
// records were re-read for each of the eight slabs: 0.32-0.35 ms per launch, its skeleton alone 0.16.)
constexpr int PW_THREADS = 1024, PW_T = 256, PW_BR = 16, PW_REC = 4096, PW_SPAN = 288;
struct PullWinArgs {
  int64_t N;
  int F;                      // 256
  int64_t ntiles;
  int tiles_per_wg;
  const int32_t* csc_ptr;     // [N + 1]
  const float4* rec;          // [entries] {source row (int bits), e0, e1, e2} in CSC order = ascending source per target
  const float* dA;            // [N][E][F]
  const float* base;          // [N][F]
  float* out;                 // [N][F]
};

// A workgroup (one per CU, sixteen waves) takes tiles of 256 consecutive targets.  Per tile: every thread owns 4 float4 of
// FOUR target rows (16 lanes per target, 64 targets per pass, 4 passes: 64 accumulator registers — the tile's whole
// [256][256] gradient block lives in registers), the tile's records are staged in LDS once (64 KB), and the source rows
// [lo, hi] of the tile are walked in blocks of 24 full rows x E (72 KB, the next block travelling in registers meanwhile).
// The records of a target are in ascending source order (CSC order = ascending entry id = source-major), so each (thread,
// pass) just advances a pointer through its target's list as the blocks go by.  dA, base and out cross HBM once, whole rows.
// Measured (same box, ms per launch at 512 x 256 atoms, profiles/r05c_pull_win.txt): through L2 0.384 | this kernel 0.335 | its
// parts: dA loads 0.02 (hidden), sums out of LDS 0.16 (29 % of the LDS read rate: chains of dependent LDS round trips, one
// edge per pass and block), everything else 0.11.  Not kept: the four passes advancing together (348 B/lane of scratch); one
// wave per target row with wave-uniform lists (0.63: sixteen waves, one edge each in flight).
template <int EC>
__global__ __launch_bounds__(PW_THREADS, 1) void pull_win_kernel(PullWinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pw_smem[];
  constexpr int F4 = 64;                                                   // float4 per row (F = 256)
  constexpr int WIN4 = PW_BR * EC * F4;                                     // float4 per block
  constexpr int PF = (WIN4 + PW_THREADS - 1) / PW_THREADS;
  float4* win4 = reinterpret_cast<float4*>(pw_smem);                      // [PW_BR][EC][64]
  float4* srec = win4 + WIN4;                                             // [PW_REC]
  int* red = reinterpret_cast<int*>(srec + PW_REC);                       // [2][16] wave partials, [32..33] lo / hi
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g16 = tid >> 4, l16 = tid & 15;
  const float4* dA4 = reinterpret_cast<const float4*>(a.dA);
#pragma unroll 1
  for (int64_t tile = (int64_t)blockIdx.x * a.tiles_per_wg; tile < min(((int64_t)blockIdx.x + 1) * a.tiles_per_wg, a.ntiles); ++tile) {
    const int64_t r0 = tile * PW_T, r1 = min(r0 + PW_T, a.N);
    const int Q0 = a.csc_ptr[r0], Q1 = a.csc_ptr[r1];
    const int nrec = Q1 - Q0;
    __syncthreads();      // the tile before is done with srec / win4 / red
    // records of the tile -> LDS (the first PW_REC; a longer list is finished from memory), their source range on the way
    int lo = 0x7fffffff, hi = -1;
    for (int q = tid; q < nrec; q += PW_THREADS) {
      const float4 r = a.rec[Q0 + q];
      if (q < PW_REC) srec[q] = r;
      const int sidx = __float_as_int(r.x);
      lo = min(lo, sidx); hi = max(hi, sidx);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    if (lane == 0) { red[wave] = lo; red[16 + wave] = hi; }
    __syncthreads();
    if (tid == 0) {
      int l = red[0], h = red[16];
      for (int w = 1; w < PW_THREADS / 64; ++w) { l = min(l, red[w]); h = max(h, red[16 + w]); }
      red[32] = l; red[33] = h;
    }
    __syncthreads();
    lo = red[32]; hi = red[33];
    // accumulators = base rows; list pointers
    float4 acc[4][4];
    int qp[4], qe[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t tg = r0 + 64 * p + g16;
      const bool live = tg < r1;
      const int64_t tc = live ? tg : r0;
      qp[p] = a.csc_ptr[tc] - Q0;
      qe[p] = live ? a.csc_ptr[tc + 1] - Q0 : qp[p];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = reinterpret_cast<const float4*>(a.base)[tc * F4 + 16 * j + l16];
    }
    auto blk_load = [&](float4 (&pf)[PF], int b0) {
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        const int x = tid + k * PW_THREADS;
        const int r = x / (EC * F4), rem = x % (EC * F4);
        const int64_t g = min((int64_t)b0 + r, a.N - 1);      // rows past the end are never referenced
        pf[k] = x < WIN4 ? dA4[g * (EC * F4) + rem] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    float4 pf[PF];
    if (hi >= lo) blk_load(pf, lo);
#pragma unroll 1
    for (int b0 = lo; b0 <= hi; b0 += PW_BR) {
      __syncthreads();      // readers of the block before are done
#pragma unroll
      for (int k = 0; k < PF; ++k)
        if (tid + k * PW_THREADS < WIN4) win4[tid + k * PW_THREADS] = pf[k];
      __syncthreads();
      if (b0 + PW_BR <= hi) blk_load(pf, b0 + PW_BR);
      const int b1 = b0 + PW_BR;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        while (qp[p] < qe[p]) {
          const float4 r = qp[p] < PW_REC ? srec[qp[p]] : a.rec[Q0 + qp[p]];
          const int sidx = __float_as_int(r.x);
          if (sidx >= b1) break;
          const float ev[3] = {r.y, r.z, r.w};
          if (sidx < b0) {
            // a record behind the block: the target's records do not ascend by source (lists of ng_build_incoming_lists
            // do; ng_mp_layer_bwd_rec also takes caller-built ones).  Its row comes from memory — same products, and no
            // read in front of the staged block (round-5 advisor finding).
            const float4* gr = dA4 + (int64_t)sidx * (EC * F4) + l16;
#pragma unroll
            for (int n = 0; n < EC; ++n)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float4 v = gr[n * F4 + 16 * j];
                acc[p][j].x = fmaf(ev[n], v.x, acc[p][j].x); acc[p][j].y = fmaf(ev[n], v.y, acc[p][j].y);
                acc[p][j].z = fmaf(ev[n], v.z, acc[p][j].z); acc[p][j].w = fmaf(ev[n], v.w, acc[p][j].w);
              }
            ++qp[p];
            continue;
          }
          const float4* wr = win4 + (sidx - b0) * (EC * F4) + l16;
#pragma unroll
          for (int n = 0; n < EC; ++n) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 v = wr[n * F4 + 16 * j];
              acc[p][j].x = fmaf(ev[n], v.x, acc[p][j].x); acc[p][j].y = fmaf(ev[n], v.y, acc[p][j].y);
              acc[p][j].z = fmaf(ev[n], v.z, acc[p][j].z); acc[p][j].w = fmaf(ev[n], v.w, acc[p][j].w);
            }
            __builtin_amdgcn_sched_barrier(0);      // (four row quarters in flight, not twelve)
          }
          ++qp[p];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t tg = r0 + 64 * p + g16;
      if (tg < r1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) reinterpret_cast<float4*>(a.out)[tg * F4 + 16 * j + l16] = acc[p][j];
      }
    }
  }
}

bool pull_win_ok(ng_ctx* ctx, int64_t N, int F, int E, const float* rec) {
  return rec != nullptr && E >= 1 && E <= 3 && F == 256 && N >= 4096 && ctx->graph_span > 0 && ctx->graph_span <= PW_SPAN;
}

int pull_win(ng_ctx* ctx, hipStream_t st, int64_t N, int F, int E, const int32_t* csc_ptr, const float* rec, const float* dA,
             const float* base, float* out) {
  PullWinArgs a;
  a.N = N; a.F = F; a.ntiles = cdiv(N, PW_T);
  a.tiles_per_wg = (int)std::max<int64_t>(1, cdiv(a.ntiles, (int64_t)ctx->num_cu));
  a.csc_ptr = csc_ptr; a.rec = reinterpret_cast<const float4*>(rec); a.dA = dA; a.base = base; a.out = out;
  const int grid = (int)cdiv(a.ntiles, a.tiles_per_wg);
  const size_t lds = (size_t)PW_BR * E * F * 4 + (size_t)PW_REC * 16 + 64 * 4;
  switch (E) {
    case 1: hipLaunchKernelGGL((pull_win_kernel<1>), dim3(grid), dim3(PW_THREADS), lds, st, a); break;
    case 2: hipLaunchKernelGGL((pull_win_kernel<2>), dim3(grid), dim3(PW_THREADS), lds, st, a); break;
    default: hipLaunchKernelGGL((pull_win_kernel<3>), dim3(grid), dim3(PW_THREADS), lds, st, a); break;
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

int csr_scatter_pull(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const int32_t* row_of,
                     const int32_t* csc_ptr, const int32_t* csc_edge, const float* e, const float* rec, const float* dA,
                     const float* dh_out, float* dh_in) {
  if (N == 0) return NG_OK;
  ProfScope ps(ctx, st, "mp_scatter_pull");
  if (pull_win_ok(ctx, N, F, E, rec)) return pull_win(ctx, st, N, F, E, csc_ptr, rec, dA, dh_out, dh_in);
  if ((F == 128 || F == 256) && E <= 4) {
    const int lpa = F / 16;
    const dim3 gridw((unsigned)cdiv(N, 256 / lpa));
    const float4* rec4 = E <= 3 ? reinterpret_cast<const float4*>(rec) : nullptr;
#define CALLW(EE)                                                                                                   \
  if (lpa == 16) hipLaunchKernelGGL((csr_scatter_pull_wide_kernel<EE, 16>), gridw, dim3(256), 0, st, N, E, K, row_of, csc_ptr, csc_edge, e, rec4, dA, dh_out, dh_in); \
  else hipLaunchKernelGGL((csr_scatter_pull_wide_kernel<EE, 8>), gridw, dim3(256), 0, st, N, E, K, row_of, csc_ptr, csc_edge, e, rec4, dA, dh_out, dh_in);
    switch (E) { case 1: { CALLW(1) } break; case 2: { CALLW(2) } break; case 3: { CALLW(3) } break; default: { CALLW(4) } break; }
#undef CALLW
    NG_HIP(ctx, hipGetLastError());
    return NG_OK;
  }
  const int apb = 256 / (F / 4);
  const dim3 grid((unsigned)cdiv(N, apb));
  for (int n0 = 0; n0 < E; n0 += 8) {
    const int ec = std::min(8, E - n0);
    const float* base = n0 == 0 ? dh_out : dh_in;
#define CALL(EE)                                                                                              \
  hipLaunchKernelGGL((csr_scatter_pull_kernel<EE>), grid, dim3(256), 0, st, N, F, E, n0, K, row_of, csc_ptr, \
                     csc_edge, e, dA, base, dh_in);
    NG_EC_SWITCH(ec, CALL)
#undef CALL
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

// dP[i][m] = dH[i][m] * act'(S[i][m]) * v[i]   (SURVEY App. B), formed ONCE per layer: the dA GEMM, the three k-tiles of
// the dw GEMM and (before) their loaders each recomputed it from dH and S — a third of the dw GEMM's 1.2 GB of reads
// blockmax (optional): max |dP| per block, the input of the fp16 GEMMs' power-of-two gradient scale (gemm_h2.hip)
__global__ __launch_bounds__(256) void mp_dp_kernel(int64_t N, int F, int act, const float* __restrict__ dH,
                                                    const float* __restrict__ S, const float* __restrict__ v,
                                                    float* __restrict__ dP, float* __restrict__ blockmax) {
  const int c4n = F / 4;
  const int64_t total = N * c4n;
  float amax = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / c4n;
    float4 d = reinterpret_cast<const float4*>(dH)[t];
    if (S) {
      const float4 s = reinterpret_cast<const float4*>(S)[t];
      d.x *= act_grad_from_out(act, s.x); d.y *= act_grad_from_out(act, s.y);
      d.z *= act_grad_from_out(act, s.z); d.w *= act_grad_from_out(act, s.w);
    }
    const float r = v[i];
    d.x *= r; d.y *= r; d.z *= r; d.w *= r;
    reinterpret_cast<float4*>(dP)[t] = d;
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(d.x), fabsf(d.y))), fmaxf(fabsf(d.z), fabsf(d.w)));
  }
  if (blockmax) block_max_store(amax, blockmax);
}

// generic MPLayer forward over (row_ptr | K) lists: repack w, aggregate, GEMM with the epilogue of layers.py:42 + model.py:167
int mp_generic_fwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, int act, int residual, const float* h,
                   const int32_t* row_ptr, const int32_t* col, const float* e, const float* inv_degree, const float* w,
                   float* h_out, float* A_save, float* s_save) {
  const int64_t KF = (int64_t)E * F;
  if (E <= 3 && (mp_gg_supported(N, F, E, row_ptr ? 0 : K) || mp_gw_infer_ok(ctx, N, K, F, E, row_ptr != nullptr, A_save != nullptr)))     // gather-GEMM (gemm_h2.hip): the aggregate only as a by-product when asked for
    return mp_gg_fwd(ctx, st, N, K, F, E, act, residual, h, row_ptr, col, e, inv_degree, w, h_out, s_save, A_save);
  float* ws = (float*)workspace(ctx, (size_t)(KF * F + (A_save ? 0 : N * KF)) * 4);
  if (!ws) return NG_ERR_NOMEM;
  const float* Wp = nullptr;
  float* A = A_save ? A_save : ws + KF * F;
  int rc = mp_plain_weights(ctx, st, N, F, E, w, ws, 0, &Wp);
  if (rc) return rc;
  rc = csr_aggregate(ctx, st, N, K, F, E, h, row_ptr, col, e, A);
  if (rc) return rc;
  return dense_fwd(ctx, st, N, (int)KF, F, act, A, Wp, nullptr, inv_degree, residual ? h : nullptr, h_out, s_save,
                   "mp_update_fwd");
}

int mp_generic_bwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, int act, const float* h,
                   const int32_t* row_ptr, const int32_t* col, const int32_t* row_of, const float* e,
                   const float* inv_degree, const float* w, const float* A_save, const float* s_save,
                   const int32_t* csc_ptr, const int32_t* csc_edge, const float* dh_out, float* dh_in, float* de,
                   int de_accum, float* dw, const float* csc_rec, int64_t nnz) {
  const int64_t KF = (int64_t)E * F;
  const size_t dw_scr = dense_dw_scratch_floats(ctx, N, (int)KF, F, false);
  // node side as a gather-GEMM over dP rows (gemm_h2.hip) needs the incoming-edge records; built here when the caller has none
  const bool gg = E <= 3 && N > 0 && mp_gg_supported(N, F, E, 0);
  const int64_t n_ent = row_ptr ? nnz : N * K;
  const size_t rec_floats = gg && !csc_rec ? (size_t)std::max<int64_t>(n_ent, 1) * 4 : 0;
  float* ws = (float*)workspace(ctx, (size_t)(KF * F + N * KF + N * F + dw_scr + (A_save ? 0 : N * KF) + rec_floats) * 4);
  if (!ws) return NG_ERR_NOMEM;
  const float* Wp = nullptr;
  float* dA = ws + KF * F;
  float* dP = dA + N * KF;
  float* scr = dP + N * F;
  int rc = mp_plain_weights(ctx, st, N, F, E, w, ws, 1, &Wp);
  if (rc) return rc;
  const bool A_save_given = A_save != nullptr;
  if (!A_save) {   // the caller did not keep the forward aggregate: rebuild it
    float* Ar = scr + dw_scr;
    rc = row_ptr ? csr_aggregate(ctx, st, N, K, F, E, h, row_ptr, col, e, Ar) : mp_aggregate_padded(ctx, st, N, K, F, E, h, col, e, Ar);
    if (rc) return rc;
    A_save = Ar;
  }
  // the same dP feeds both products; its power-of-two scale (gemm_h2.hip) comes from block maxima the dP kernel
  // writes on the side — no extra pass over the tensor
  const float* gsc = nullptr;
  if (N > 0) {
    ProfScope ps(ctx, st, "mp_dP");
    const int64_t work = N * (F / 4);
    int cap = 0;
    float* bmax = dense_grad_uses_h2(N, (int)KF, F) ? gemm_grad_blockmax(ctx, &cap) : nullptr;
    const int nblk = (int)std::min<int64_t>(cdiv(work, 256), 256 * 16);
    hipLaunchKernelGGL(mp_dp_kernel, dim3((unsigned)nblk), dim3(256), 0, st, N, F, act, dh_out,
                       act == NG_ACT_NONE ? nullptr : s_save, inv_degree, dP, bmax);
    NG_HIP(ctx, hipGetLastError());
    if (bmax) {
      rc = gemm_grad_scale_from_blocks(ctx, st, nblk, &gsc);
      if (rc) return rc;
    }
  }
  rc = dense_dw(ctx, st, N, (int)KF, F, NG_ACT_NONE, A_save, dP, nullptr, nullptr, dw, nullptr, 1, F, E, scr, "mp_dw", gsc);
  if (rc) return rc;
  rc = dense_dx(ctx, st, N, (int)KF, F, NG_ACT_NONE, dP, nullptr, nullptr, Wp, nullptr, dA, "mp_dA", gsc);
  if (rc) return rc;
  rc = csr_edge_grad(ctx, st, N, K, F, E, h, row_ptr, col, dA, de, de_accum);
  if (rc) return rc;
  // node side: dh_in = dh_out + pull of dA over the incoming edges — at the default width as a gather-GEMM over dP rows
  // (1 KB per incoming edge instead of 3 KB of dA; gemm_h2.hip)
  if (gg) {
    if (!csc_rec) {
      float* rec = scr + dw_scr + (A_save_given ? 0 : N * KF);
      rc = mp_win_records(ctx, st, N, K, E, csc_ptr, csc_edge, e, rec, row_of, n_ent);
      if (rc) return rc;
      csc_rec = rec;
    }
    return mp_gg_pull(ctx, st, N, F, E, dP, csc_ptr, csc_rec, w, dh_out, dh_in, gsc);
  }
  return csr_scatter_pull(ctx, st, N, K, F, E, row_of, csc_ptr, csc_edge, e, csc_rec, dA, dh_out, dh_in);
}

// ------------------------------------------------------------------------------------------------------------------
// Distance-cutoff graphs (BASELINE configs[4] "variable degree"): every OTHER atom of the same frame closer than
// `cutoff` (Angstrom) is a neighbour; rows are written in ascending neighbour index ("CSR-sorted": consecutive
// entries gather consecutive rows).  Two passes over LDS-tiled positions, one thread per query atom:
//   count:  deg[row]                               -> the caller's exclusive scan gives row_ptr
//   fill :  col (batch-global), dist*scale, inv_degree = 1/#(local neighbour index > 0)  (library.py:115-116)
constexpr int CUT_TILE = 1024;
// ONE explicit fused-multiply-add chain for the squared distance (the expression of knn.hip's knn_dist2): the count and
// the fill pass — and the one-thread and 16-lane forms — must agree bit for bit on `d2 < cutoff2`, because rows are sized
// by the count and then filled; left to -ffp-contract two instantiations may round differently by an ulp.
__device__ __forceinline__ float cut_dist2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

template <bool FILL>
__global__ __launch_bounds__(256) void cutoff_kernel(int n, float cutoff2, float scale, const float* __restrict__ pos,
                                                     int32_t* __restrict__ deg, const int32_t* __restrict__ row_ptr,
                                                     int32_t* __restrict__ col, float* __restrict__ dist,
                                                     float* __restrict__ inv_degree, int32_t* __restrict__ row_of) {
  __shared__ float sx[CUT_TILE], sy[CUT_TILE], sz[CUT_TILE];
  const int frame = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float* fp = pos + (int64_t)frame * n * 3;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (i < n) { qx = fp[3 * i]; qy = fp[3 * i + 1]; qz = fp[3 * i + 2]; }
  const int64_t row = (int64_t)frame * n + i;
  int cnt = 0, cnt_pos = 0;
  int64_t out = 0, lim = 0;          // a row never writes past its own extent, whatever the count pass saw
  if (FILL && i < n) { out = row_ptr[row]; lim = row_ptr[row + 1]; }
  for (int t0 = 0; t0 < n; t0 += CUT_TILE) {
    const int m = min(CUT_TILE, n - t0);
    __syncthreads();
    for (int t = threadIdx.x; t < m; t += 256) {
      sx[t] = fp[3 * (t0 + t)]; sy[t] = fp[3 * (t0 + t) + 1]; sz[t] = fp[3 * (t0 + t) + 2];
    }
    __syncthreads();
    if (i < n) {
      for (int t = 0; t < m; ++t) {
        const float dx = sx[t] - qx, dy = sy[t] - qy, dz = sz[t] - qz;
        const float d2 = cut_dist2(dx, dy, dz);
        const int j = t0 + t;
        if (d2 < cutoff2 && j != i) {
          if (FILL && out + cnt < lim) {
            col[out + cnt] = frame * n + j;
            dist[out + cnt] = sqrtf(d2) * scale;
            if (row_of) row_of[out + cnt] = (int32_t)row;
          }
          ++cnt;
          cnt_pos += j > 0 ? 1 : 0;
        }
      }
    }
  }
  if (i >= n) return;
  if (FILL) inv_degree[row] = cnt_pos > 0 ? 1.0f / (float)cnt_pos : 0.f;
  else deg[row] = cnt;
}

// The same with 16 lanes per query atom (round 3): lane s of an atom's group tests candidates s, s + 16, ..; a wave
// ballot per 16-candidate chunk gives the hits in ascending candidate order, so rows come out exactly as the
// one-thread-per-atom kernel writes them.  One thread per atom left a 2770-atom frame with 11 workgroups walking 2770
// candidates each, its hits stored one by one: 80 us (count) + 295 us (fill) per frame; this form: 256 threads = 16 atoms.
template <bool FILL>
__global__ __launch_bounds__(256) void cutoff_s16_kernel(int n, float cutoff2, float scale, const float* __restrict__ pos,
                                                         int32_t* __restrict__ deg, const int32_t* __restrict__ row_ptr,
                                                         int32_t* __restrict__ col, float* __restrict__ dist,
                                                         float* __restrict__ inv_degree, int32_t* __restrict__ row_of) {
  __shared__ float sx[CUT_TILE], sy[CUT_TILE], sz[CUT_TILE];
  const int frame = blockIdx.y;
  const int s = threadIdx.x & 15, grp = (threadIdx.x & 63) >> 4;
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
  const float* fp = pos + (int64_t)frame * n * 3;
  const int ic = i < n ? i : n - 1;
  const float qx = fp[3 * ic], qy = fp[3 * ic + 1], qz = fp[3 * ic + 2];
  const int64_t row = (int64_t)frame * n + i;
  int cnt = 0, cnt_pos = 0;
  int64_t out = 0, lim = 0;          // a row never writes past its own extent, whatever the count pass saw
  if (FILL && i < n) { out = row_ptr[row]; lim = row_ptr[row + 1]; }
  for (int t0 = 0; t0 < n; t0 += CUT_TILE) {
    const int m = min(CUT_TILE, n - t0);
    __syncthreads();
    for (int t = threadIdx.x; t < m; t += 256) {
      sx[t] = fp[3 * (t0 + t)]; sy[t] = fp[3 * (t0 + t) + 1]; sz[t] = fp[3 * (t0 + t) + 2];
    }
    __syncthreads();
    for (int c0 = 0; c0 < m; c0 += 64) {          // wave-uniform trip count: every lane takes part in the ballots
      // four 16-candidate chunks per trip: their LDS reads are in flight together (a ballot per chunk in a plain loop
      // serialises on the LDS latency of each)
      bool hit[4];
      float d2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = c0 + 16 * u + s;
        hit[u] = false;
        d2[u] = 0.f;
        if (t < m && i < n) {
          const float dx = sx[t] - qx, dy = sy[t] - qy, dz = sz[t] - qz;
          d2[u] = cut_dist2(dx, dy, dz);
          hit[u] = d2[u] < cutoff2 && t0 + t != i;
        }
      }
      if (!FILL) {        // counting needs no order: per-lane tallies, summed over the 16 lanes at the end
#pragma unroll
        for (int u = 0; u < 4; ++u) cnt += hit[u] ? 1 : 0;
        continue;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = t0 + c0 + 16 * u + s;
        const unsigned long long b = __ballot(hit[u]);
        const unsigned mine = (unsigned)(b >> (16 * grp)) & 0xFFFFu;
        const int p = cnt + __popc(mine & ((1u << s) - 1u));
        if (FILL && hit[u] && out + p < lim) {
          col[out + p] = frame * n + j;
          dist[out + p] = sqrtf(d2[u]) * scale;
          if (row_of) row_of[out + p] = (int32_t)row;
        }
        cnt += __popc(mine);
        // local neighbour index > 0 (library.py:115-116): candidate 0 of the frame does not count
        cnt_pos += __popc(t0 + c0 + 16 * u == 0 ? (mine & ~1u) : mine);
      }
    }
  }
  if (!FILL) {
    cnt += __shfl_xor(cnt, 1, 64); cnt += __shfl_xor(cnt, 2, 64); cnt += __shfl_xor(cnt, 4, 64); cnt += __shfl_xor(cnt, 8, 64);
  }
  if (i >= n || s != 0) return;
  if (FILL) inv_degree[row] = cnt_pos > 0 ? 1.0f / (float)cnt_pos : 0.f;
  else deg[row] = cnt;
}

// One WAVE per query atom, for molecule-sized calls (round 4; the counterpart of knn.hip: knn_wave_kernel): the 64 lanes test 64
// candidates per step, a ballot gives the hits in ascending candidate order — the order of the other cutoff kernels, same
// distance expression, so the rows are the same bit for bit — and the frame's positions are staged in LDS once per workgroup
// (n <= 4096).  A 2770-atom frame: count 27-38 us + fill 36 us (16 lanes per atom) -> a few us each.
constexpr int CUT_WAVE_MAXN = 4096;
template <bool FILL>
__global__ __launch_bounds__(256) void cutoff_wave_kernel(int n, float cutoff2, float scale, const float* __restrict__ pos,
                                                          int32_t* __restrict__ deg, const int32_t* __restrict__ row_ptr,
                                                          int32_t* __restrict__ col, float* __restrict__ dist,
                                                          float* __restrict__ inv_degree, int32_t* __restrict__ row_of) {
  extern __shared__ float cw_pos[];               // [3][n]
  float* sx = cw_pos; float* sy = cw_pos + n; float* sz = cw_pos + 2 * n;
  const int frame = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* fp = pos + (int64_t)frame * n * 3;
  for (int t = threadIdx.x; t < n; t += 256) { sx[t] = fp[3 * t]; sy[t] = fp[3 * t + 1]; sz[t] = fp[3 * t + 2]; }
  __syncthreads();
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;                             // uniform over the wave
  const float qx = sx[i], qy = sy[i], qz = sz[i];
  const int64_t row = (int64_t)frame * n + i;
  int64_t out = 0, lim = 0;
  if (FILL) { out = row_ptr[row]; lim = row_ptr[row + 1]; }
  int cnt = 0, cnt_pos = 0;
  for (int tb = 0; tb < n; tb += 64) {
    const int t = tb + lane;
    const int tc = min(t, n - 1);
    const float d2 = cut_dist2(sx[tc] - qx, sy[tc] - qy, sz[tc] - qz);
    const bool hit = t < n && t != i && d2 < cutoff2;
    const unsigned long long m = __ballot(hit);
    if (FILL && hit) {
      const int64_t p = out + cnt + __popcll(m & ((1ull << lane) - 1ull));
      if (p < lim) {                              // a row never writes past its own extent
        col[p] = frame * n + t;
        dist[p] = sqrtf(d2) * scale;
        if (row_of) row_of[p] = (int32_t)row;
      }
    }
    cnt += __popcll(m);
    cnt_pos += __popcll(tb == 0 ? (m & ~1ull) : m);      // local neighbour index > 0 (library.py:115-116)
  }
  if (lane != 0) return;
  if (FILL) inv_degree[row] = cnt_pos > 0 ? 1.0f / (float)cnt_pos : 0.f;
  else deg[row] = cnt;
}

}  // namespace ng

using namespace ng;

// ===================================================================================== C ABI
extern "C" int ng_mp_aggregate_csr(ng_ctx* ctx, void* stream, int64_t N, int F, int E, const float* h,
                                   const int32_t* row_ptr, const int32_t* col, const float* e, float* A) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, row_ptr && col, "ng_mp_aggregate_csr: row_ptr and col are required");
  return csr_aggregate(ctx, (hipStream_t)stream, N, 0, F, E, h, row_ptr, col, e, A);
}

extern "C" int ng_mp_layer_fwd_csr(ng_ctx* ctx, void* stream, int64_t N, int64_t nnz, int F, int E, int act,
                                   int residual, const float* h, const int32_t* row_ptr, const int32_t* col,
                                   const float* e, const float* inv_degree, const float* w, float* h_out,
                                   float* A_save, float* s_save) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, row_ptr && (col || nnz == 0), "ng_mp_layer_fwd_csr: row_ptr and col are required");
  NG_REQUIRE(ctx, (E * F) % 8 == 0, "mp_layer: (E*F) % 8");
  NG_REQUIRE(ctx, nnz >= 0 && nnz < ((int64_t)1 << 31), "mp_layer (csr): nnz must fit int32");
  return mp_generic_fwd(ctx, (hipStream_t)stream, N, 0, F, E, act, residual, h, row_ptr, col, e, inv_degree, w, h_out,
                        A_save, s_save);
}

extern "C" int ng_mp_layer_bwd_csr(ng_ctx* ctx, void* stream, int64_t N, int64_t nnz, int F, int E, int act,
                                   const float* h, const int32_t* row_ptr, const int32_t* col, const int32_t* row_of,
                                   const float* e, const float* inv_degree, const float* w, const float* A_save,
                                   const float* s_save, const int32_t* csc_ptr, const int32_t* csc_edge,
                                   const float* dh_out, float* dh_in, float* de, int de_accum, float* dw) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, row_ptr && (nnz == 0 || (col && row_of)), "ng_mp_layer_bwd_csr: row_ptr, col and row_of are required");
  NG_REQUIRE(ctx, act == NG_ACT_NONE || s_save, "mp_layer_bwd: s_save required for an activation");
  NG_REQUIRE(ctx, nnz >= 0 && nnz < ((int64_t)1 << 31), "mp_layer (csr): nnz must fit int32");
  return mp_generic_bwd(ctx, (hipStream_t)stream, N, 0, F, E, act, h, row_ptr, col, row_of, e, inv_degree, w, A_save,
                        s_save, csc_ptr, csc_edge, dh_out, dh_in, de, de_accum, dw, nullptr, nnz);
}

extern "C" int ng_cutoff_count(ng_ctx* ctx, void* stream, int G, int n, float cutoff, const float* pos,
                               int32_t* deg) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, G >= 0 && n >= 0 && cutoff > 0.f, "cutoff graph: sizes >= 0, cutoff > 0");
  NG_REQUIRE(ctx, (int64_t)G * n < (int64_t)1 << 31 && G <= 65535, "cutoff graph: batch too large");
  if (G == 0 || n == 0) return NG_OK;
  ProfScope ps(ctx, (hipStream_t)stream, "cutoff_count");
  if (!sw().knn_serial && !sw().knn_lanes && n <= CUT_WAVE_MAXN && (int64_t)G * n <= 16384)      // molecule-sized: one wave per atom
    hipLaunchKernelGGL((cutoff_wave_kernel<false>), dim3((unsigned)cdiv(n, 4), (unsigned)G), dim3(256), (size_t)3 * n * 4,
                       (hipStream_t)stream, n, cutoff * cutoff, 1.0f, pos, deg, nullptr, nullptr, nullptr, nullptr, nullptr);
  else if (sw().knn_serial)
    hipLaunchKernelGGL((cutoff_kernel<false>), dim3((unsigned)cdiv(n, 256), (unsigned)G), dim3(256), 0,
                       (hipStream_t)stream, n, cutoff * cutoff, 1.0f, pos, deg, nullptr, nullptr, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((cutoff_s16_kernel<false>), dim3((unsigned)cdiv(n, 16), (unsigned)G), dim3(256), 0,
                       (hipStream_t)stream, n, cutoff * cutoff, 1.0f, pos, deg, nullptr, nullptr, nullptr, nullptr, nullptr);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_cutoff_fill_rows(ng_ctx* ctx, void* stream, int G, int n, float cutoff, float scale, const float* pos,
                                   const int32_t* row_ptr, int32_t* col, float* dist, float* inv_degree, int32_t* row_of);
extern "C" int ng_cutoff_fill(ng_ctx* ctx, void* stream, int G, int n, float cutoff, float scale, const float* pos,
                              const int32_t* row_ptr, int32_t* col, float* dist, float* inv_degree) {
  return ng_cutoff_fill_rows(ctx, stream, G, n, cutoff, scale, pos, row_ptr, col, dist, inv_degree, nullptr);
}

// the same, also writing row_of[nnz] (the row of every entry: what the CSR backward walks) when it is not NULL
extern "C" int ng_cutoff_fill_rows(ng_ctx* ctx, void* stream, int G, int n, float cutoff, float scale, const float* pos,
                                   const int32_t* row_ptr, int32_t* col, float* dist, float* inv_degree, int32_t* row_of) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, G >= 0 && n >= 0 && cutoff > 0.f, "cutoff graph: sizes >= 0, cutoff > 0");
  NG_REQUIRE(ctx, (int64_t)G * n < (int64_t)1 << 31 && G <= 65535, "cutoff graph: batch too large");
  if (G == 0 || n == 0) return NG_OK;
  ProfScope ps(ctx, (hipStream_t)stream, "cutoff_fill");
  if (!sw().knn_serial && !sw().knn_lanes && n <= CUT_WAVE_MAXN && (int64_t)G * n <= 16384)
    hipLaunchKernelGGL((cutoff_wave_kernel<true>), dim3((unsigned)cdiv(n, 4), (unsigned)G), dim3(256), (size_t)3 * n * 4,
                       (hipStream_t)stream, n, cutoff * cutoff, scale, pos, nullptr, row_ptr, col, dist, inv_degree, row_of);
  else if (sw().knn_serial)
    hipLaunchKernelGGL((cutoff_kernel<true>), dim3((unsigned)cdiv(n, 256), (unsigned)G), dim3(256), 0,
                       (hipStream_t)stream, n, cutoff * cutoff, scale, pos, nullptr, row_ptr, col, dist, inv_degree, row_of);
  else
    hipLaunchKernelGGL((cutoff_s16_kernel<true>), dim3((unsigned)cdiv(n, 16), (unsigned)G), dim3(256), 0,
                       (hipStream_t)stream, n, cutoff * cutoff, scale, pos, nullptr, row_ptr, col, dist, inv_degree, row_of);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}
