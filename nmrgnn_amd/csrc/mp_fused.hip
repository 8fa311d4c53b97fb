// Fused MPLayer kernels for atom_feature_size == 64 (the bench architecture), edge_feature_size <= 3.
// Reference: nmrgnn/layers.py:26-46 (MPLayer.call) + residual of nmrgnn/model.py:165-167.
//
//   forward   h'[i,:] = act( v_i * sum_{n,l} A[i,n,l] W[l,:,n] ) + h[i,:],
//             A[i,n,l] = sum_j e[i,j,n] h[nlist[i,j], l]
//   One persistent kernel per layer: gather + edge-weighted segment sum straight into an LDS tile
//   [64 atoms][E*64] (never to HBM unless training keeps it for dw), fp32 MFMA against the weight slab
//   held in registers for the whole launch, epilogue (inv_degree, activation, residual) from the
//   accumulators.
//
//   backward to nodes uses THE SAME kernel on the transposed graph:
//     dh[t,l] = dH[t,l] + sum_{n,m} W[l,m,n] * B[t,n,m],   B[t,n,m] = sum_{(i,j)->t} e[i,j,n] dP[i,m]
//   i.e. an aggregation of dP rows over the INCOMING edges of t (CSC lists, ragged) followed by the
//   same kind of GEMM — 256-B row gathers instead of the 768-B dA rows a literal scatter would pull.
//
//   backward to edges: dA tile = dP W^T by MFMA into LDS, then de[i,j,n] = <dA[i,n,:], h[nlist[i,j],:]>.
//   dA never reaches HBM.  dP = dH * act'(S) * v is formed once here and written for the other two.
#include <algorithm>

#include "mfma_gemm.cuh"
#include "ng_internal.h"

namespace ng {

constexpr int MF = 64;          // feature width handled here
constexpr int MTM = 64;         // atoms per tile
constexpr int MLPA = MF / 4;    // lanes per atom in the gather phases (float4 each)

bool mp_fused_supported(int F, int E) { return F == MF && E >= 1 && E <= 3; }

// ---- weight fragment packing --------------------------------------------------------------------
// mode 0 (forward):       Wsrc(k = n*F + l, o = m) = w[l][m][n]      out[ms][t][lane][s], o = 32ms+(lane&31)
// mode 1 (back to nodes): Wsrc(k = n*F + m, o = l) = w[l][m][n]
// mode 2 (dA = dP Wp^T):  Wsrc(k = m, o = n*F + l) = w[l][m][n]      (contraction over m, F/8 t-steps)
__global__ void mp_pack_kernel(int E, int mode, const float* __restrict__ w, float* __restrict__ out) {
  const int KF = E * MF;
  const int kdim = mode == 2 ? MF : KF;     // contraction length
  const int odim = mode == 2 ? KF : MF;     // output width
  const int total = kdim * odim;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int r = idx;
    const int s = r & 3; r >>= 2;
    const int lane = r & 63; r >>= 6;
    const int nt = kdim / 8;
    const int t = r % nt;
    const int ms = r / nt;
    const int k = 8 * t + 4 * (lane >> 5) + s;
    const int o = 32 * ms + (lane & 31);
    float v;
    if (mode == 0) {
      const int n = k / MF, l = k % MF;
      v = w[(l * MF + o) * E + n];
    } else if (mode == 1) {
      const int n = k / MF, m = k % MF;
      v = w[(o * MF + m) * E + n];
    } else {
      const int n = o / MF, l = o % MF;
      v = w[(l * MF + k) * E + n];
    }
    out[idx] = v;
  }
}

struct MpFusedArgs {
  int64_t N;
  int K;
  const float* src;        // [N][64] rows to gather (h, or dP for the transposed pass)
  const int32_t* nlist;    // fixed-K lists            (RAGGED = false)
  const int32_t* ptr;      // incoming-edge lists      (RAGGED = true): ptr[N+1], eids[nnz]
  const int32_t* eids;
  const float* e;          // [N*K][E]
  const float* Wfrag;      // packed, mode 0 or 1
  const float* rowscale;   // [N] or nullptr
  const float* resid;      // [N][64] or nullptr
  float* out;              // [N][64]
  float* A_save;           // [N][E*64] or nullptr
  float* S_save;           // [N][64] or nullptr
  int act;
};

template <int E, bool RAGGED>
__global__ __launch_bounds__(256, 2) void mp_fused_kernel(MpFusedArgs a) {
  constexpr int KF = E * MF;
  constexpr int LDA = KF + 4;
  constexpr int NT = KF / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                                              // [64][LDA]
  int32_t* s_nl = reinterpret_cast<int32_t*>(sA + MTM * LDA);    // [64*K]      (fixed-K only)
  float* s_e = reinterpret_cast<float*>(s_nl) + (RAGGED ? 0 : MTM * a.K);   // [64*K*E]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ms = wave & 1, rt = wave >> 1;

  // this wave's weight slab as MFMA A-fragments, resident for the whole launch
  float wf[KF / 2];
  {
    const float4* p = reinterpret_cast<const float4*>(a.Wfrag) + (ms * NT) * 64 + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 v = p[t * 64];
      wf[4 * t + 0] = v.x; wf[4 * t + 1] = v.y; wf[4 * t + 2] = v.z; wf[4 * t + 3] = v.w;
    }
  }
  const float4* src4 = reinterpret_cast<const float4*>(a.src);
  const int64_t ntiles = (a.N + MTM - 1) / MTM;

  // contiguous chunk of tiles per workgroup: the tiles of one molecule share their gathered rows,
  // so they should hit the same CU / XCD L2 back to back
  const int64_t tpw = (ntiles + gridDim.x - 1) / gridDim.x;
  const int64_t tile_end = (blockIdx.x + 1) * tpw < ntiles ? (blockIdx.x + 1) * tpw : ntiles;
#pragma unroll 1
  for (int64_t tile = blockIdx.x * tpw; tile < tile_end; ++tile) {
    const int64_t i0 = tile * MTM;
    const int n_at = (int)(a.N - i0 < MTM ? a.N - i0 : MTM);
    if (!RAGGED) {
      for (int t = tid; t < n_at * a.K; t += 256) s_nl[t] = a.nlist[i0 * a.K + t];
      for (int t = tid; t < n_at * a.K * E; t += 256) s_e[t] = a.e[i0 * a.K * E + t];
      __syncthreads();
    }
    // ---- gather + edge-weighted segment sum: 16 lanes per atom, 4 atoms per wave per round
    const int c = lane & (MLPA - 1);
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
      const int al = wave * 16 + round * 4 + (lane >> 4);   // atom within the tile
      float4 acc[E];
#pragma unroll
      for (int n = 0; n < E; ++n) acc[n] = f4zero();
      if (al < n_at) {
        // gathers are issued 8 at a time (8 independent 16-B loads in flight per lane) before any
        // of them is consumed: the phase is latency-bound, not bandwidth-bound
        if (!RAGGED) {
          for (int j0 = 0; j0 < a.K; j0 += 8) {
            float4 hv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int j = j0 + u < a.K ? j0 + u : a.K - 1;
              hv[u] = src4[(int64_t)s_nl[al * a.K + j] * MLPA + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (j0 + u < a.K) {
#pragma unroll
                for (int n = 0; n < E; ++n) {
                  const float ev = s_e[(al * a.K + j0 + u) * E + n];
                  acc[n].x += ev * hv[u].x; acc[n].y += ev * hv[u].y;
                  acc[n].z += ev * hv[u].z; acc[n].w += ev * hv[u].w;
                }
              }
            }
          }
        } else {
          const int p0 = a.ptr[i0 + al], p1 = a.ptr[i0 + al + 1];
          for (int q0 = p0; q0 < p1; q0 += 8) {
            int eid[8];
            float4 hv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) eid[u] = a.eids[q0 + u < p1 ? q0 + u : p1 - 1];
#pragma unroll
            for (int u = 0; u < 8; ++u) hv[u] = src4[(int64_t)(eid[u] / a.K) * MLPA + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (q0 + u < p1) {
#pragma unroll
                for (int n = 0; n < E; ++n) {
                  const float ev = a.e[(int64_t)eid[u] * E + n];
                  acc[n].x += ev * hv[u].x; acc[n].y += ev * hv[u].y;
                  acc[n].z += ev * hv[u].z; acc[n].w += ev * hv[u].w;
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int n = 0; n < E; ++n)
        *reinterpret_cast<float4*>(sA + al * LDA + n * MF + 4 * c) = acc[n];
    }
    __syncthreads();
    // ---- keep A for the weight gradient (whole rows, coalesced)
    if (a.A_save) {
      constexpr int C4 = KF / 4;
      for (int t = tid; t < n_at * C4; t += 256) {
        const int r = t / C4, c4 = t % C4;
        *reinterpret_cast<float4*>(a.A_save + (i0 + r) * KF + c4 * 4) =
            *reinterpret_cast<const float4*>(sA + r * LDA + c4 * 4);
      }
    }
    // ---- [64 x KF] x [KF x 64]: this wave -> rows 32rt.., output columns 32ms..
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* xrow = sA + (rt * 32 + l31) * LDA + 4 * half;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 x = *reinterpret_cast<const float4*>(xrow + 8 * t);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 0], x.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 1], x.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 2], x.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 3], x.w, acc, 0, 0, 0);
    }
    const int64_t row = i0 + rt * 32 + l31;
    if (row < a.N) {
      const float rs = a.rowscale ? a.rowscale[row] : 1.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = ms * 32 + 8 * q + 4 * half;
        float4 v = make_float4(acc[4 * q + 0] * rs, acc[4 * q + 1] * rs, acc[4 * q + 2] * rs,
                               acc[4 * q + 3] * rs);
        if (a.act != NG_ACT_NONE) {
          v.x = act_apply(a.act, v.x); v.y = act_apply(a.act, v.y);
          v.z = act_apply(a.act, v.z); v.w = act_apply(a.act, v.w);
        }
        const int64_t o = row * MF + m;
        if (a.S_save) *reinterpret_cast<float4*>(a.S_save + o) = v;
        if (a.resid) {
          const float4 r4 = *reinterpret_cast<const float4*>(a.resid + o);
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        *reinterpret_cast<float4*>(a.out + o) = v;
      }
    }
    __syncthreads();
  }
}

// ---- backward to the edge features ----------------------------------------------------------------
struct MpEdgeBwdArgs {
  int64_t N;
  int K;
  const float* dH;        // [N][64] upstream gradient of the layer output
  const float* S;         // [N][64] saved activation output (nullptr -> linear)
  const float* rowscale;  // inv_degree
  const float* h;         // [N][64] layer input
  const int32_t* nlist;
  const float* WfragT;    // packed mode 2
  float* dP;              // [N][64] out
  float* de;              // [N*K][E]
  int act;
  int accumulate;
};

template <int E>
__global__ __launch_bounds__(256, 2) void mp_bwd_edge_kernel(MpEdgeBwdArgs a) {
  constexpr int KF = E * MF;
  constexpr int LDA = KF + 4;
  constexpr int LDP = MF + 4;
  constexpr int SL = E;              // 32-wide output slabs per wave (KF/32 slabs, 2 row tiles, 4 waves)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sdA = smem;                         // [64][LDA]
  float* sdP = sdA + MTM * LDA;              // [64][LDP]
  float* sde = sdP + MTM * LDP;              // [64][K*E]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int rt = wave & 1, sl0 = (wave >> 1) * SL;

  float wf[SL][32];   // W^T fragments: slab o in [32(sl0+j), +32), contraction m (64 -> 8 t-steps)
#pragma unroll
  for (int j = 0; j < SL; ++j) {
    const float4* p = reinterpret_cast<const float4*>(a.WfragT) + ((sl0 + j) * 8) * 64 + lane;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 v = p[t * 64];
      wf[j][4 * t + 0] = v.x; wf[j][4 * t + 1] = v.y; wf[j][4 * t + 2] = v.z; wf[j][4 * t + 3] = v.w;
    }
  }
  const float4* h4 = reinterpret_cast<const float4*>(a.h);
  const int64_t ntiles = (a.N + MTM - 1) / MTM;
  const int KE = a.K * E;

  // contiguous chunk of tiles per workgroup: the tiles of one molecule share their gathered rows,
  // so they should hit the same CU / XCD L2 back to back
  const int64_t tpw = (ntiles + gridDim.x - 1) / gridDim.x;
  const int64_t tile_end = (blockIdx.x + 1) * tpw < ntiles ? (blockIdx.x + 1) * tpw : ntiles;
#pragma unroll 1
  for (int64_t tile = blockIdx.x * tpw; tile < tile_end; ++tile) {
    const int64_t i0 = tile * MTM;
    const int n_at = (int)(a.N - i0 < MTM ? a.N - i0 : MTM);
    // ---- dP = dH * act'(S) * v  -> LDS and global
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lin = tid + i * 256;
      const int r = lin >> 4, c4 = lin & 15;
      float4 g = f4zero();
      if (r < n_at) {
        const int64_t o = (i0 + r) * MF + c4 * 4;
        g = *reinterpret_cast<const float4*>(a.dH + o);
        if (a.S) {
          const float4 s = *reinterpret_cast<const float4*>(a.S + o);
          g.x *= act_grad_from_out(a.act, s.x); g.y *= act_grad_from_out(a.act, s.y);
          g.z *= act_grad_from_out(a.act, s.z); g.w *= act_grad_from_out(a.act, s.w);
        }
        if (a.rowscale) {
          const float v = a.rowscale[i0 + r];
          g.x *= v; g.y *= v; g.z *= v; g.w *= v;
        }
        *reinterpret_cast<float4*>(a.dP + o) = g;
      }
      *reinterpret_cast<float4*>(sdP + r * LDP + c4 * 4) = g;
    }
    __syncthreads();
    // ---- dA[row][o] = sum_m dP[row][m] Wp[o][m]   (rows 32rt.., slabs sl0..sl0+SL-1)
    {
      f32x16 acc[SL];
#pragma unroll
      for (int j = 0; j < SL; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      const float* xrow = sdP + (rt * 32 + l31) * LDP + 4 * half;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float4 x = *reinterpret_cast<const float4*>(xrow + 8 * t);
#pragma unroll
        for (int j = 0; j < SL; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][4 * t + 0], x.x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][4 * t + 1], x.y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][4 * t + 2], x.z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][4 * t + 3], x.w, acc[j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < SL; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(sdA + (rt * 32 + l31) * LDA + (sl0 + j) * 32 + 8 * q + 4 * half) =
              make_float4(acc[j][4 * q + 0], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
    }
    __syncthreads();
    // ---- de[i][j][n] = <dA[i][n][:], h[nlist[i][j]][:]>   16 lanes per atom, 4 atoms per wave per round
    const int c = lane & (MLPA - 1);
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
      const int al = wave * 16 + round * 4 + (lane >> 4);
      const bool live = al < n_at;
      float4 g[E];
#pragma unroll
      for (int n = 0; n < E; ++n) g[n] = *reinterpret_cast<const float4*>(sdA + al * LDA + n * MF + 4 * c);
      for (int j0 = 0; j0 < a.K; j0 += 8) {
        float4 hv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u < a.K ? j0 + u : a.K - 1;
          const int idx = live ? a.nlist[(i0 + al) * a.K + j] : 0;
          hv[u] = h4[(int64_t)idx * MLPA + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          float part[E];
#pragma unroll
          for (int n = 0; n < E; ++n)
            part[n] = g[n].x * hv[u].x + g[n].y * hv[u].y + g[n].z * hv[u].z + g[n].w * hv[u].w;
#pragma unroll
          for (int off = MLPA >> 1; off > 0; off >>= 1) {
#pragma unroll
            for (int n = 0; n < E; ++n) part[n] += __shfl_xor(part[n], off, 64);
          }
          if (c == 0 && j0 + u < a.K) {
#pragma unroll
            for (int n = 0; n < E; ++n) sde[al * KE + (j0 + u) * E + n] = part[n];
          }
        }
      }
    }
    __syncthreads();
    // ---- coalesced copy-out of the tile's de block
    for (int t = tid; t < n_at * KE; t += 256) {
      const int64_t o = i0 * KE + t;
      a.de[o] = a.accumulate ? a.de[o] + sde[t] : sde[t];
    }
    __syncthreads();
  }
}

// ---- host side ---------------------------------------------------------------------------------------
// selected with NG_MP_PATH=fused (default is the split path of mp_split.hip, which measures faster)
bool mp_fused_enabled(int F, int E) {
  const char* v = getenv("NG_MP_PATH");
  return v && std::string(v) == "fused" && mp_fused_supported(F, E);
}

int mp_pack(ng_ctx* ctx, hipStream_t st, int E, int mode, const float* w, float* out) {
  hipLaunchKernelGGL(mp_pack_kernel, dim3(48), dim3(256), 0, st, E, mode, w, out);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

template <bool RAGGED>
static int launch_mp_fused(ng_ctx* ctx, hipStream_t st, int E, const MpFusedArgs& a, const char* tag) {
  const int KF = E * MF;
  const size_t lds = (size_t)(MTM * (KF + 4) + (RAGGED ? 0 : MTM * a.K * (1 + E))) * 4;
  const int64_t ntiles = cdiv(a.N, MTM);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu * 2);
  ProfScope ps(ctx, st, tag);
  switch (E) {
    case 1: hipLaunchKernelGGL((mp_fused_kernel<1, RAGGED>), dim3(grid), dim3(256), lds, st, a); break;
    case 2: hipLaunchKernelGGL((mp_fused_kernel<2, RAGGED>), dim3(grid), dim3(256), lds, st, a); break;
    case 3: hipLaunchKernelGGL((mp_fused_kernel<3, RAGGED>), dim3(grid), dim3(256), lds, st, a); break;
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

int mp_fused_fwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, int residual,
                 const float* h, const int32_t* nlist, const float* e, const float* inv_degree,
                 const float* w, float* h_out, float* A_save, float* s_save) {
  const int KF = E * MF;
  float* Wfrag = (float*)workspace(ctx, (size_t)KF * MF * 4);
  if (!Wfrag) return NG_ERR_NOMEM;
  int rc = mp_pack(ctx, st, E, 0, w, Wfrag);
  if (rc) return rc;
  MpFusedArgs a{};
  a.N = N; a.K = K; a.src = h; a.nlist = nlist; a.e = e; a.Wfrag = Wfrag; a.rowscale = inv_degree;
  a.resid = residual ? h : nullptr; a.out = h_out; a.A_save = A_save; a.S_save = s_save; a.act = act;
  return launch_mp_fused<false>(ctx, st, E, a, "mp_fused_fwd");
}

int mp_fused_bwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, const float* h,
                 const int32_t* nlist, const float* e, const float* inv_degree, const float* w,
                 const float* A_save, const float* s_save, const int32_t* csc_ptr,
                 const int32_t* csc_edge, const float* dh_out, float* dh_in, float* de, int de_accum,
                 float* dw) {
  const int KF = E * MF;
  const size_t dw_scr = dense_dw_scratch_floats(ctx, N, KF, MF, false);
  // scratch: two packed weight copies, dP, dw partials
  float* ws = (float*)workspace(ctx, (size_t)(2 * KF * MF + N * MF + dw_scr) * 4);
  if (!ws) return NG_ERR_NOMEM;
  float* WfragT = ws;
  float* WfragN = ws + KF * MF;
  float* dP = WfragN + KF * MF;
  float* scr = dP + N * MF;
  int rc = mp_pack(ctx, st, E, 2, w, WfragT);
  if (rc) return rc;
  rc = mp_pack(ctx, st, E, 1, w, WfragN);
  if (rc) return rc;
  if (N == 0) return dense_dw(ctx, st, N, KF, MF, NG_ACT_NONE, A_save, dP, nullptr, nullptr, dw,
                              nullptr, 1, MF, E, scr, "mp_dw");
  {
    MpEdgeBwdArgs a{};
    a.N = N; a.K = K; a.dH = dh_out; a.S = act == NG_ACT_NONE ? nullptr : s_save;
    a.rowscale = inv_degree; a.h = h; a.nlist = nlist; a.WfragT = WfragT; a.dP = dP; a.de = de;
    a.act = act; a.accumulate = de_accum;
    const size_t lds = (size_t)(MTM * (KF + 4) + MTM * (MF + 4) + MTM * K * E) * 4;
    const int grid = (int)std::min<int64_t>(cdiv(N, MTM), (int64_t)ctx->num_cu * 2);
    ProfScope ps(ctx, st, "mp_bwd_edge");
    switch (E) {
      case 1: hipLaunchKernelGGL((mp_bwd_edge_kernel<1>), dim3(grid), dim3(256), lds, st, a); break;
      case 2: hipLaunchKernelGGL((mp_bwd_edge_kernel<2>), dim3(grid), dim3(256), lds, st, a); break;
      case 3: hipLaunchKernelGGL((mp_bwd_edge_kernel<3>), dim3(grid), dim3(256), lds, st, a); break;
    }
    NG_HIP(ctx, hipGetLastError());
  }
  // dw[l][m][n] = sum_i A[i][(n,l)] dP[i][m]
  rc = dense_dw(ctx, st, N, KF, MF, NG_ACT_NONE, A_save, dP, nullptr, nullptr, dw, nullptr, 1, MF, E,
                scr, "mp_dw");
  if (rc) return rc;
  // dh_in = dh_out + (transposed aggregation of dP) x Wq
  MpFusedArgs b{};
  b.N = N; b.K = K; b.src = dP; b.ptr = csc_ptr; b.eids = csc_edge; b.e = e; b.Wfrag = WfragN;
  b.resid = dh_out; b.out = dh_in; b.act = NG_ACT_NONE;
  return launch_mp_fused<true>(ctx, st, E, b, "mp_bwd_node");
}

}  // namespace ng
