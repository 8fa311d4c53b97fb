// MPLayer forward / backward for atom_feature_size == 64 as a short chain of simple kernels:
//   high-occupancy, XCD-aware gather kernels (HBM/L2-bound)  +  tall-skinny MFMA products with the
//   weights resident in registers (tall_gemm.hip).
// Reference: nmrgnn/layers.py:26-46 (MPLayer.call), nmrgnn/model.py:165-167 (residual); backward math
// in SURVEY App. B, with the scatter  dh[nlist[i,j]] += sum_n e_ijn dA_iln  rewritten as an aggregation
// of dP rows over the INCOMING edges followed by a dense product (see mp_fused.hip header).
//
// Why not the fully fused kernels of mp_fused.hip: measured at the bench shape they are parked on
// memory 50-70 % of the time (two 4-wave workgroups per CU cannot hide the dependent gather latency),
// while these 32-waves-per-CU kernels keep an L2 hit rate > 90 % (tiles of one molecule share an XCD)
// and run the aggregation at 57 % of the 8 TB/s HBM peak.
//
//   forward    A  = agg(h; nlist, e)                      [gather]     A_save
//              h' = act(v * A Wp) + h                     [tall GEMM]  S_save
//   backward   dP = dH * act'(S) * v ; dA = dP Wp^T       [tall GEMM, prologue]
//              de[i,j,n] (+)= <dA[i,n,:], h[nlist[i,j],:]> [gather]
//              dw = A^T dP                                [TN GEMM, split over rows]
//              B  = agg_csc(dP; csc, e)                   [gather over incoming edges]
//              dh = dH + B Wq                             [tall GEMM]
#include <algorithm>

#include "mfma_gemm.cuh"
#include "ng_internal.h"

namespace ng {

constexpr int SF = 64;          // feature width
constexpr int SC4 = SF / 4;     // lanes per atom
constexpr int SAPB = 256 / SC4; // atoms per 256-thread block

bool mp_split_enabled(int F, int E) {
  const char* v = getenv("NG_MP_PATH");
  if (v && std::string(v) != "split" && std::string(v) != "win") return false;   // win: forward only so far
  return F == SF && E >= 1 && E <= 3;
}

// A[i][n][:] = sum_j e[i][j][n] * src[nlist[i][j]][:]
template <int E>
__global__ __launch_bounds__(256) void split_agg_kernel(int64_t N, int K, const float* __restrict__ src,
                                                        const int32_t* __restrict__ nlist,
                                                        const float* __restrict__ e,
                                                        float* __restrict__ A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int32_t* s_nl = reinterpret_cast<int32_t*>(smem_raw);          // [16*K]
  float* s_e = reinterpret_cast<float*>(smem_raw) + SAPB * K;    // [16*K*E]
  // XCD-aware: workgroup b runs on XCD b % 8 -> each XCD owns a contiguous eighth of the atoms, so
  // all tiles of a molecule share one L2 (gathered rows miss once, then hit)
  const int64_t i0 = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * SAPB;
  const int n_at = (int)std::min<int64_t>(SAPB, N - i0);
  for (int t = threadIdx.x; t < n_at * K; t += 256) s_nl[t] = nlist[i0 * K + t];
  for (int t = threadIdx.x; t < n_at * K * E; t += 256) s_e[t] = e[i0 * K * E + t];
  __syncthreads();
  const int a = threadIdx.x >> 4, c = threadIdx.x & 15;
  if (a >= n_at) return;
  float4 acc[E];
#pragma unroll
  for (int n = 0; n < E; ++n) acc[n] = f4zero();
  const float4* s4 = reinterpret_cast<const float4*>(src);
  for (int j0 = 0; j0 < K; j0 += 8) {
    float4 hv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u < K ? j0 + u : K - 1;
      hv[u] = s4[(int64_t)s_nl[a * K + j] * SC4 + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (j0 + u < K) {
#pragma unroll
        for (int n = 0; n < E; ++n) {
          const float ev = s_e[(a * K + j0 + u) * E + n];
          acc[n].x += ev * hv[u].x; acc[n].y += ev * hv[u].y;
          acc[n].z += ev * hv[u].z; acc[n].w += ev * hv[u].w;
        }
      }
    }
  }
  float4* A4 = reinterpret_cast<float4*>(A);
#pragma unroll
  for (int n = 0; n < E; ++n) A4[((i0 + a) * E + n) * SC4 + c] = acc[n];
}

// B[t][n][:] = sum_{p in csc[t]} e[p][n] * src[p / K][:]      (p = edge id i*K + j with nlist[i][j] == t)
template <int E>
__global__ __launch_bounds__(256) void split_agg_csc_kernel(int64_t N, int K,
                                                            const float* __restrict__ src,
                                                            const int32_t* __restrict__ ptr,
                                                            const int32_t* __restrict__ eids,
                                                            const float* __restrict__ e,
                                                            float* __restrict__ B) {
  const int a = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int64_t t = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * SAPB + a;
  if (t >= N) return;
  float4 acc[E];
#pragma unroll
  for (int n = 0; n < E; ++n) acc[n] = f4zero();
  const float4* s4 = reinterpret_cast<const float4*>(src);
  const int p0 = ptr[t], p1 = ptr[t + 1];
  for (int q0 = p0; q0 < p1; q0 += 8) {
    int eid[8];
    float4 hv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) eid[u] = eids[q0 + u < p1 ? q0 + u : p1 - 1];
#pragma unroll
    for (int u = 0; u < 8; ++u) hv[u] = s4[(int64_t)(eid[u] / K) * SC4 + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (q0 + u < p1) {
#pragma unroll
        for (int n = 0; n < E; ++n) {
          const float ev = e[(int64_t)eid[u] * E + n];
          acc[n].x += ev * hv[u].x; acc[n].y += ev * hv[u].y;
          acc[n].z += ev * hv[u].z; acc[n].w += ev * hv[u].w;
        }
      }
    }
  }
  float4* B4 = reinterpret_cast<float4*>(B);
#pragma unroll
  for (int n = 0; n < E; ++n) B4[(t * E + n) * SC4 + c] = acc[n];
}

// de[i][j][n] (+)= <dA[i][n][:], h[nlist[i][j]][:]>
template <int E>
__global__ __launch_bounds__(256) void split_edge_grad_kernel(int64_t N, int K,
                                                              const float* __restrict__ h,
                                                              const int32_t* __restrict__ nlist,
                                                              const float* __restrict__ dA,
                                                              float* __restrict__ de, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float s_de[];   // [16][K*E]
  const int a = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int64_t i0 = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * SAPB;
  const int64_t i = i0 + a;
  const bool live = i < N;
  const int64_t ii = live ? i : 0;
  const int KE = K * E;
  const float4* dA4 = reinterpret_cast<const float4*>(dA);
  const float4* h4 = reinterpret_cast<const float4*>(h);
  float4 g[E];
#pragma unroll
  for (int n = 0; n < E; ++n) g[n] = dA4[(ii * E + n) * SC4 + c];
  for (int j0 = 0; j0 < K; j0 += 8) {
    float4 hv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u < K ? j0 + u : K - 1;
      hv[u] = h4[(int64_t)nlist[ii * K + j] * SC4 + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float part[E];
#pragma unroll
      for (int n = 0; n < E; ++n)
        part[n] = g[n].x * hv[u].x + g[n].y * hv[u].y + g[n].z * hv[u].z + g[n].w * hv[u].w;
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
        for (int n = 0; n < E; ++n) part[n] += __shfl_xor(part[n], off, 64);
      }
      if (c == 0 && j0 + u < K) {
#pragma unroll
        for (int n = 0; n < E; ++n) s_de[a * KE + (j0 + u) * E + n] = part[n];
      }
    }
  }
  __syncthreads();
  // coalesced copy-out of the block's 16 x K*E gradient block
  const int n_at = (int)std::min<int64_t>(SAPB, N - i0);
  for (int t = threadIdx.x; t < n_at * KE; t += 256) {
    const int64_t o = i0 * KE + t;
    de[o] = accumulate ? de[o] + s_de[t] : s_de[t];
  }
}

// ---- second-generation gather kernels ----------------------------------------------------------------
// B[t][n][:] over incoming edges, with the block's edge ids and edge features staged in LDS first:
// the ids are read coalesced, the dependent e[eid] gathers happen once up front (all in flight
// together) instead of sitting in the inner loop behind every row gather.
constexpr int CSC_CAP = 1024;   // staged incoming edges per 16-atom block (falls back to global beyond)

template <int E>
__global__ __launch_bounds__(256) void split_agg_csc2_kernel(int64_t N, int K,
                                                             const float* __restrict__ src,
                                                             const int32_t* __restrict__ ptr,
                                                             const int32_t* __restrict__ eids,
                                                             const float* __restrict__ e,
                                                             float* __restrict__ B) {
  __shared__ int s_src[CSC_CAP];
  __shared__ float s_e[CSC_CAP * E];
  const int a = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int64_t t0 = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * SAPB;
  const int64_t t1 = std::min<int64_t>(t0 + SAPB, N);
  const int p_lo = ptr[t0], p_hi = ptr[t1];
  const int cnt = p_hi - p_lo;
  const bool staged = cnt <= CSC_CAP;
  if (staged) {
    for (int q = threadIdx.x; q < cnt; q += 256) {
      const int eid = eids[p_lo + q];
      s_src[q] = eid / K;
#pragma unroll
      for (int n = 0; n < E; ++n) s_e[q * E + n] = e[(int64_t)eid * E + n];
    }
  }
  __syncthreads();
  const int64_t t = t0 + a;
  if (t >= N) return;
  float4 acc[E];
#pragma unroll
  for (int n = 0; n < E; ++n) acc[n] = f4zero();
  const float4* s4 = reinterpret_cast<const float4*>(src);
  const int p0 = ptr[t], p1 = ptr[t + 1];
  for (int q0 = p0; q0 < p1; q0 += 8) {
    int row[8];
    float ev[8][E];
    float4 hv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + u < p1 ? q0 + u : p1 - 1;
      if (staged) {
        row[u] = s_src[q - p_lo];
#pragma unroll
        for (int n = 0; n < E; ++n) ev[u][n] = s_e[(q - p_lo) * E + n];
      } else {
        const int eid = eids[q];
        row[u] = eid / K;
#pragma unroll
        for (int n = 0; n < E; ++n) ev[u][n] = e[(int64_t)eid * E + n];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) hv[u] = s4[(int64_t)row[u] * SC4 + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (q0 + u < p1) {
#pragma unroll
        for (int n = 0; n < E; ++n) {
          acc[n].x += ev[u][n] * hv[u].x; acc[n].y += ev[u][n] * hv[u].y;
          acc[n].z += ev[u][n] * hv[u].z; acc[n].w += ev[u][n] * hv[u].w;
        }
      }
    }
  }
  float4* B4 = reinterpret_cast<float4*>(B);
#pragma unroll
  for (int n = 0; n < E; ++n) B4[(t * E + n) * SC4 + c] = acc[n];
}

// de[i][j][n] (+)= <dA[i][n][:], h[nlist[i][j]][:]> with FOUR lanes per atom (16 features each,
// interleaved in 16-B units so that the 4 lanes of an atom read 64 contiguous bytes per load): the
// cross-lane reduction shrinks from 4 ds_bpermute stages to 2 quad-permute DPP adds per value.
__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  return v;
}

template <int E>
__global__ __launch_bounds__(256) void split_edge_grad2_kernel(int64_t N, int K,
                                                               const float* __restrict__ h,
                                                               const int32_t* __restrict__ nlist,
                                                               const float* __restrict__ dA,
                                                               float* __restrict__ de, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float s_de[];   // [64][K*E]
  constexpr int APB = 64;
  const int a = threadIdx.x >> 2, c = threadIdx.x & 3;
  const int64_t i0 = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * APB;
  const int64_t i = i0 + a;
  const bool live = i < N;
  const int64_t ii = live ? i : 0;
  const int KE = K * E;
  const float4* dA4 = reinterpret_cast<const float4*>(dA);
  const float4* h4 = reinterpret_cast<const float4*>(h);
  // lane c owns float4 columns c, c+4, c+8, c+12 of the 16 in a row
  float4 g[E][4];
#pragma unroll
  for (int n = 0; n < E; ++n)
#pragma unroll
    for (int u = 0; u < 4; ++u) g[n][u] = dA4[(ii * E + n) * SC4 + c + 4 * u];
  for (int j0 = 0; j0 < K; j0 += 4) {
    float4 hv[4][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = j0 + jj < K ? j0 + jj : K - 1;
      const int64_t row = nlist[ii * K + j];
#pragma unroll
      for (int u = 0; u < 4; ++u) hv[jj][u] = h4[row * SC4 + c + 4 * u];
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
      for (int n = 0; n < E; ++n) {
        float p = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          p += g[n][u].x * hv[jj][u].x + g[n][u].y * hv[jj][u].y + g[n][u].z * hv[jj][u].z +
               g[n][u].w * hv[jj][u].w;
        p = quad_sum(p);
        if (c == 0 && j0 + jj < K) s_de[a * KE + (j0 + jj) * E + n] = p;
      }
    }
  }
  __syncthreads();
  const int n_at = (int)std::min<int64_t>(APB, N - i0);
  for (int t = threadIdx.x; t < n_at * KE; t += 256) {
    const int64_t o = i0 * KE + t;
    de[o] = accumulate ? de[o] + s_de[t] : s_de[t];
  }
}

#define NG_E_SWITCH(E, CALL)  \
  switch (E) {                \
    case 1: { CALL(1) } break; \
    case 2: { CALL(2) } break; \
    case 3: { CALL(3) } break; \
  }

int mp_split_fwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, int residual,
                 const float* h, const int32_t* nlist, const float* e, const float* inv_degree,
                 const float* w, float* h_out, float* A_save, float* s_save) {
  const int KF = E * SF;
  float* ws = (float*)workspace(ctx, (size_t)(KF * SF + (A_save ? 0 : N * KF)) * 4);
  if (!ws) return NG_ERR_NOMEM;
  float* Wfrag = ws;
  float* A = A_save ? A_save : ws + KF * SF;
  int rc = mp_pack(ctx, st, E, 0, w, Wfrag);
  if (rc) return rc;
  {
    ProfScope ps(ctx, st, "mp_aggregate");
    const dim3 grid((unsigned)cdiv(N, SAPB));
    const size_t lds = (size_t)SAPB * K * (1 + E) * 4;
#define CALL(EE) hipLaunchKernelGGL((split_agg_kernel<EE>), grid, dim3(256), lds, st, N, K, h, nlist, e, A);
    NG_E_SWITCH(E, CALL)
#undef CALL
    NG_HIP(ctx, hipGetLastError());
  }
  TallArgs a{};
  a.N = N; a.X = A; a.ldx = KF; a.k_valid = KF; a.Wfrag = Wfrag; a.rowscale = inv_degree; a.act = act;
  a.S_save = s_save; a.resid = residual ? h : nullptr; a.out = h_out; a.ldo = SF; a.n_valid = SF;
  return tall_gemm(ctx, st, KF, SF, a, false, "mp_update_fwd");
}

int mp_split_bwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, const float* h,
                 const int32_t* nlist, const float* e, const float* inv_degree, const float* w,
                 const float* A_save, const float* s_save, const int32_t* csc_ptr,
                 const int32_t* csc_edge, const float* dh_out, float* dh_in, float* de, int de_accum,
                 float* dw, const float* csc_rec) {
  const int KF = E * SF;
  const size_t dw_scr = tall_tn_scratch_floats(ctx, KF);
  // scratch: two packed weight copies | dP [N,64] | dA / B [N,KF] (shared) | dw partials
  float* ws = (float*)workspace(ctx, (size_t)(2 * KF * SF + N * SF + N * KF + dw_scr + 64) * 4);
  if (!ws) return NG_ERR_NOMEM;
  float* WfragT = ws;
  float* WfragN = WfragT + KF * SF;
  float* dP = WfragN + KF * SF;
  float* dAB = dP + N * SF;
  float* scr = dAB + N * KF;
  float* dummy = scr + dw_scr;
  // NG_MP_BWD=split keeps the two-kernel edge gradient (dA to HBM, then the gather-dot); default is the
  // window-resident fused kernel of mp_win_bwd.hip when the neighbour count allows it
  const char* bsel = getenv("NG_MP_BWD");
  const bool win_edge = mp_win_bwd_supported(SF, E, K) && !(bsel && std::string(bsel) == "split");
  const bool win_node = win_edge && !(bsel && std::string(bsel) == "edge") && N * K * 4 <= N * KF &&
                        mp_win_node_scratch_floats(ctx, E) <= dw_scr;
  int rc;
  if (win_edge && win_node) {
    rc = mpw_pack2(ctx, st, E, w, 2, WfragT, 1, WfragN);     // both weight images in one launch
    if (rc) return rc;
  } else {
    rc = win_edge ? mpw_pack(ctx, st, E, 2, w, WfragT) : mp_pack(ctx, st, E, 2, w, WfragT);
    if (rc) return rc;
    rc = win_node ? mpw_pack(ctx, st, E, 1, w, WfragN) : mp_pack(ctx, st, E, 1, w, WfragN);
    if (rc) return rc;
  }
  if (win_edge) {
    rc = mp_win_bwd_edge(ctx, st, N, K, E, act, h, nlist, inv_degree, WfragT, s_save, dh_out, dP, de, de_accum,
                         dummy);
    if (rc) return rc;
    if (win_node) {   // incoming-edge aggregate, dh GEMM and dw in one kernel; dAB's space holds the records
      if (!csc_rec) {
        rc = mp_win_records(ctx, st, N, K, E, csc_ptr, csc_edge, e, dAB);
        if (rc) return rc;
        csc_rec = dAB;
      }
      return mp_win_bwd_node(ctx, st, N, E, h, dP, csc_ptr, csc_rec, WfragN, dh_out, dh_in, dw, scr, dummy);
    }
  } else {   // dP = dH * act'(S) * v (kept) ;  dA = dP Wp^T
    TallArgs a{};
    a.N = N; a.X = dh_out; a.ldx = SF; a.k_valid = SF;
    a.S_in = act == NG_ACT_NONE ? nullptr : s_save; a.rs_in = inv_degree; a.act_in = act; a.dP_out = dP;
    a.Wfrag = WfragT; a.act = NG_ACT_NONE; a.out = dAB; a.ldo = KF; a.n_valid = KF;
    rc = tall_gemm(ctx, st, SF, KF, a, true, "mp_dA");
    if (rc) return rc;
  }
  if (N > 0 && !win_edge) {
    ProfScope ps(ctx, st, "mp_edge_grad");
    const dim3 grid((unsigned)cdiv(N, 64));
    const size_t lds = (size_t)64 * K * E * 4;
#define CALL(EE)                                                                                      \
  hipLaunchKernelGGL((split_edge_grad2_kernel<EE>), grid, dim3(256), lds, st, N, K, h, nlist, dAB, de, \
                     de_accum);
    NG_E_SWITCH(E, CALL)
#undef CALL
    NG_HIP(ctx, hipGetLastError());
  }
  if (N > 0) {   // B = aggregation of dP over the incoming edges
    ProfScope ps(ctx, st, "mp_aggregate_csc");
    const dim3 grid((unsigned)cdiv(N, SAPB));
#define CALL(EE)                                                                                       \
  hipLaunchKernelGGL((split_agg_csc2_kernel<EE>), grid, dim3(256), 0, st, N, K, dP, csc_ptr, csc_edge, e, \
                     dAB);
    NG_E_SWITCH(E, CALL)
#undef CALL
    NG_HIP(ctx, hipGetLastError());
  }
  // dw[l][m][n] = sum_i v_i A[i][(n,l)] dP'[i][m] = sum_t h[t][l] B[t][(n,m)]  — the aggregate A of the
  // forward pass is not needed: the incoming-edge aggregate B of dP carries the same sum
  rc = tall_tn(ctx, st, N, dAB, KF, KF, h, SF, SF, nullptr, NG_ACT_NONE, dw, nullptr, 2, SF, E, scr, "mp_dw");
  if (rc) return rc;
  TallArgs b{};
  b.N = N; b.X = dAB; b.ldx = KF; b.k_valid = KF; b.Wfrag = WfragN; b.act = NG_ACT_NONE;
  b.resid = dh_out; b.out = dh_in; b.ldo = SF; b.n_valid = SF;
  return tall_gemm(ctx, st, KF, SF, b, false, "mp_dh");
}

}  // namespace ng
