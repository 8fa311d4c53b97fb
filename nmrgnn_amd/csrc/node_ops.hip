// Node-side kernels: embedding, neighbour aggregation (gather + edge-weighted segment sum),
// MPLayer forward/backward, output head, loss, Adam, RNG.
// Reference: nmrgnn/layers.py:26-46 (MPLayer), nmrgnn/model.py:158-169 (MPBlock), 236-274 (GNNModel),
// nmrgnn/losses.py:30-39 (NameLoss s=1).
#include <algorithm>
#include <type_traits>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "reduce.cuh"
#include "rng.cuh"

namespace ng {

constexpr int MAX_E = 8;   // edge_feature_size choices {1,2,3,8} (64 is not supported yet)
constexpr int MAX_C = 32;  // one-hot width

// element functors for small_tn_kernel
struct LoadElem {            // plain row-major [N][ld]
  const float* p;
  int ld;
  __device__ __forceinline__ float operator()(int64_t r, int c) const { return p[r * ld + c]; }
};
struct LoadHeadX {           // [g * dropout_mask | 1]
  const float* g;
  const float* mask;
  int Fh;
  __device__ __forceinline__ float operator()(int64_t r, int f) const {
    if (f == Fh) return 1.0f;
    const float x = g[r * Fh + f];
    return mask ? x * mask[r * Fh + f] : x;
  }
};
struct LoadHeadY {           // dfull[i][c] = dpeaks[i] * atoms[i][c] * std[c]
  const float* dpeaks;
  const float* atoms;
  const float* pstd;
  int C;
  __device__ __forceinline__ float operator()(int64_t r, int c) const {
    return dpeaks[r] * atoms[r * C + c] * pstd[c];
  }
};

// ------------------------------------------------------------------------------------ embedding
// h0[i][f] = sum_c atoms[i][c] * Wemb[c][f]      (model.py:262; Dense without bias)
__global__ void embed_fwd_kernel(int64_t N, int C, int F, const float* __restrict__ atoms,
                                 const float* __restrict__ Wemb, float* __restrict__ h0) {
  const int c4n = F / 4;
  const int64_t total = N * c4n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / c4n;
    const int c4 = (int)(t % c4n);
    float4 acc = f4zero();
    for (int c = 0; c < C; ++c) {
      const float a = atoms[i * C + c];
      if (a != 0.f) {
        const float4 w = *reinterpret_cast<const float4*>(Wemb + (int64_t)c * F + c4 * 4);
        acc.x += a * w.x; acc.y += a * w.y; acc.z += a * w.z; acc.w += a * w.w;
      }
    }
    *reinterpret_cast<float4*>(h0 + i * F + c4 * 4) = acc;
  }
}

// The same for rows of 16 / 32 / 64 float4 (F = 64 / 128 / 256) and C <= 16: a row is one lane group, lane j of the group
// reads atoms[i][j] once and the group passes the C values round (no per-lane re-read of the row, no branch); the lane's
// column chunk of every Wemb row stays in registers across the grid-stride loop (its chunk index never changes).
template <int C0, class Fn>
__device__ __forceinline__ void static_for16(Fn&& f) {
  if constexpr (C0 < 16) {
    f(std::integral_constant<int, C0>());
    static_for16<C0 + 1>(f);
  }
}

template <int C4N>
__global__ __launch_bounds__(256) void embed_fwd_rows_kernel(int64_t N, int C, const float* __restrict__ atoms,
                                                             const float* __restrict__ Wemb, float* __restrict__ h0) {
  constexpr int F = 4 * C4N;
  const int c4 = threadIdx.x % C4N;
  float4 w[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) w[c] = c < C ? *reinterpret_cast<const float4*>(Wemb + (int64_t)c * F + c4 * 4) : f4zero();
  const int64_t rows_per_pass = (int64_t)gridDim.x * (256 / C4N);
  // four rows per trip, their one-hot loads requested together: with one row per trip the kernel waited a memory round
  // trip per row (16 us for 131k rows of 64 features; the 33 MB it writes take 4)
  for (int64_t i0 = (int64_t)blockIdx.x * (256 / C4N) + threadIdx.x / C4N; i0 < N; i0 += 4 * rows_per_pass) {
    float mine[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * rows_per_pass;
      mine[u] = ((c4 & 15) < C && i < N) ? atoms[i * C + (c4 & 15)] : 0.f;      // every 16-lane row of the group holds the C values
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * rows_per_pass;
      float4 acc = f4zero();
      // lane c of this lane's 16-lane row, by DPP row_share (a VALU move; __shfl is an LDS-crossbar op per class and row)
      static_for16<0>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const float a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine[u]), 0x150 + c, 0xf, 0xf, false));
        acc.x = fmaf(a, w[c].x, acc.x); acc.y = fmaf(a, w[c].y, acc.y); acc.z = fmaf(a, w[c].z, acc.z); acc.w = fmaf(a, w[c].w, acc.w);
      });
      if (i < N) *reinterpret_cast<float4*>(h0 + i * F + c4 * 4) = acc;
    }
  }
}

// dWemb[c][f] = sum_i atoms[i][c] dh0[i][f]  — stage 1: partial[blk][c][f] over a row chunk
__global__ __launch_bounds__(256) void embed_bwd_kernel(int64_t N, int C, int F,
                                                        int64_t rows_per_block,
                                                        const float* __restrict__ atoms,
                                                        const float* __restrict__ dh0,
                                                        float* __restrict__ partial) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = std::min<int64_t>(r0 + rows_per_block, N);
  const int items = C * F;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int c = it / F, f = it % F;
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += atoms[r * C + c] * dh0[r * F + f];
    partial[(int64_t)blockIdx.x * items + it] = s;
  }
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, int nz, int64_t n_elem,
                                    float* __restrict__ out) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_elem;
       idx += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += partial[(int64_t)z * n_elem + idx];
    out[idx] = s;
  }
}

// ------------------------------------------------------------------------------------ aggregation
// A[i][n][l] = sum_j e[i][j][n] * h[nlist[i][j]][l]        (layers.py:33 + ij-contraction of 39-40)
// F/4 lanes per atom (float4 of features each); the block's nlist / e rows are staged in LDS with
// coalesced loads and then broadcast.
template <int E>
__global__ __launch_bounds__(256) void aggregate_kernel(int64_t N, int K, int F,
                                                        const float* __restrict__ h,
                                                        const int32_t* __restrict__ nlist,
                                                        const float* __restrict__ e,
                                                        float* __restrict__ A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int c4n = F / 4;
  const int apb = 256 / c4n;  // atoms per block
  int32_t* s_nl = reinterpret_cast<int32_t*>(smem_raw);          // [apb*K]
  float* s_e = reinterpret_cast<float*>(smem_raw) + apb * K;     // [apb*K*E]
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed; used for speed only), so give every
  // XCD a CONTIGUOUS eighth of the atoms — the tiles of one molecule then share one L2, and a gathered
  // row misses once instead of once per XCD (L2 hit rate 46 % -> 90 %+ on the bench batch)
  const int64_t i0 = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * apb;
  const int64_t n_at = std::min<int64_t>(apb, N - i0);
  for (int t = threadIdx.x; t < n_at * K; t += 256) s_nl[t] = nlist[i0 * K + t];
  for (int t = threadIdx.x; t < n_at * K * E; t += 256) s_e[t] = e[i0 * K * E + t];
  __syncthreads();
  const int a = threadIdx.x / c4n, c4 = threadIdx.x % c4n;
  if (a >= n_at) return;
  float4 acc[E];
#pragma unroll
  for (int n = 0; n < E; ++n) acc[n] = f4zero();
  const float4* h4 = reinterpret_cast<const float4*>(h);
  // 8 row gathers in flight per lane before the first is consumed (the kernel is latency-bound)
  for (int j0 = 0; j0 < K; j0 += 8) {
    float4 hv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u < K ? j0 + u : K - 1;
      hv[u] = h4[(int64_t)s_nl[a * K + j] * c4n + c4];
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
  // A is written once and read by a later kernel: non-temporal, so that it does not push the gathered h rows (which
  // every tile of the molecule re-reads) out of the XCD's L2
  typedef float nt4 __attribute__((ext_vector_type(4)));
  nt4* A4 = reinterpret_cast<nt4*>(A);
#pragma unroll
  for (int n = 0; n < E; ++n)
    __builtin_nontemporal_store(nt4{acc[n].x, acc[n].y, acc[n].z, acc[n].w}, A4 + ((i0 + a) * E + n) * c4n + c4);
}

static int aggregate(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* h,
                     const int32_t* nlist, const float* e, float* A) {
  NG_REQUIRE(ctx, F % 4 == 0 && F >= 16 && F <= 1024 && (256 % (F / 4)) == 0,
             "aggregate: F in {16,32,64,128,256,512,1024}");
  NG_REQUIRE(ctx, E >= 1 && E <= 64, "aggregate: edge_feature_size <= 64");
  if (N == 0) return NG_OK;
  if (E > MAX_E) return csr_aggregate(ctx, st, N, K, F, E, h, nullptr, nlist, e, A);
  ProfScope ps(ctx, st, "mp_aggregate");
  // the reference's default width: slab windows in LDS instead of one L2 gather per edge (mp_win.hip)
  // (only for batches of small graphs, ng_ctx_set_graph_span: the lists of a whole protein leave any window, and the
  // window kernel's global-memory fall-back is slower than the kernel below)
  if (agg_win_supported(F, E, K) && N >= 4096 && ctx->graph_span > 0 && ctx->graph_span <= agg_win_rows())
    return agg_win(ctx, st, N, K, F, E, h, nlist, e, A);
  const int apb = 256 / (F / 4);
  const size_t lds = (size_t)apb * K * (1 + E) * 4;
  const dim3 grid((unsigned)cdiv(N, apb));
#define NG_AGG(EE)                                                                              \
  case EE:                                                                                      \
    hipLaunchKernelGGL((aggregate_kernel<EE>), grid, dim3(256), lds, st, N, K, F, h, nlist, e, A); \
    break;
  switch (E) {
    NG_AGG(1) NG_AGG(2) NG_AGG(3) NG_AGG(4) NG_AGG(5) NG_AGG(6) NG_AGG(7) NG_AGG(8)
  }
#undef NG_AGG
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

int mp_aggregate_padded(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* h, const int32_t* nlist,
                        const float* e, float* A) {
  return aggregate(ctx, st, N, K, F, E, h, nlist, e, A);
}

// w[l][m][n] (reference layout, n fastest) -> Wp[k = n*F + l][m]
__global__ void repack_w_kernel(int F, int E, const float* __restrict__ w, float* __restrict__ Wp) {
  const int total = F * F * E;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx / F, m = idx % F;
    const int n = k / F, l = k % F;
    Wp[idx] = w[((int64_t)l * F + m) * E + n];
  }
}

int mp_repack_w(ng_ctx* ctx, hipStream_t st, int F, int E, const float* w, float* Wp) {
  const int total = F * F * E;
  hipLaunchKernelGGL(repack_w_kernel, dim3(std::min(cdiv(total, 256), (int64_t)1024)), dim3(256), 0,
                     st, F, E, w, Wp);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

// The GEMM form Wp[n F + l][m] of an MPLayer weight for a product over M rows: the context's cached copy (rebuilt behind
// ng_adam_step together with the piece image of the product — gemm_h2_prepack_mp — so that a training step packs nothing), or
// `scratch` filled here when the image cache is off.  trans: 0 = the update product, 1 = the dA product.
int mp_plain_weights(ng_ctx* ctx, hipStream_t st, int64_t M, int F, int E, const float* w, float* scratch, int trans, const float** Wp) {
  bool have = false;
  float* Wc = (float*)cached_image(ctx, w, 1, (size_t)E * F * F * 4, &have);
  if (!Wc) {
    *Wp = scratch;
    return mp_repack_w(ctx, st, F, E, w, scratch);
  }
  *Wp = Wc;
  if (!have) {
    PackJob j;
    j.kind = PK_MP_PLAIN; j.i0 = F; j.i1 = E; j.src[0] = w; j.dst[0] = Wc;
    j.blocks = (int)std::min<int64_t>(cdiv((int64_t)E * F * F, 256), 1024);
    const int rc = pack_launch(ctx, st, j);
    if (rc) return rc;
    cache_set_job(ctx, w, 1, j);
  }
  return gemm_h2_prepack_mp(ctx, st, M, F, E, w, Wc, trans);
}

// ------------------------------------------------------------------------------------ head
// peaks[i] = sum_c atoms[i,c] * ((g*mask)[i,:] @ Wout[:,c] + bout[c]) * std[c] + atoms[i,c]*avg[c]
__global__ __launch_bounds__(256) void head_fwd_kernel(int64_t N, int Fh, int C,
                                                       const float* __restrict__ g,
                                                       const float* __restrict__ mask,
                                                       const float* __restrict__ Wout,
                                                       const float* __restrict__ bout,
                                                       const float* __restrict__ atoms,
                                                       const float* __restrict__ pstd,
                                                       const float* __restrict__ pavg,
                                                       float* __restrict__ peaks) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sW = reinterpret_cast<float*>(smem_raw);  // [Fh*C]
  for (int t = threadIdx.x; t < Fh * C; t += 256) sW[t] = Wout[t];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float full[MAX_C];
#pragma unroll
  for (int c = 0; c < MAX_C; ++c) full[c] = (c < C) ? bout[c] : 0.f;
  for (int f = 0; f < Fh; ++f) {
    float x = g[i * Fh + f];
    if (mask) x *= mask[i * Fh + f];
#pragma unroll
    for (int c = 0; c < MAX_C; ++c)
      if (c < C) full[c] += x * sW[f * C + c];
  }
  float p = 0.f;
#pragma unroll
  for (int c = 0; c < MAX_C; ++c)
    if (c < C) {
      const float a = atoms[i * C + c];
      p += full[c] * a * pstd[c] + a * pavg[c];
    }
  peaks[i] = p;
}

// dg[i][f] = mask[i][f] * sum_c dfull[i][c] Wout[f][c],  dfull[i][c] = dpeaks[i]*atoms[i][c]*std[c]
__global__ __launch_bounds__(256) void head_bwd_dg_kernel(int64_t N, int Fh, int C,
                                                          const float* __restrict__ mask,
                                                          const float* __restrict__ Wout,
                                                          const float* __restrict__ atoms,
                                                          const float* __restrict__ pstd,
                                                          const float* __restrict__ dpeaks,
                                                          float* __restrict__ dg) {
  const int64_t total = N * Fh;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / Fh;
    const int f = (int)(t % Fh);
    const float dp = dpeaks[i];
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dp * atoms[i * C + c] * pstd[c] * Wout[f * C + c];
    if (mask) s *= mask[t];
    dg[t] = s;
  }
}

// partial[blk][f*C + c] = sum_{rows} gd[r][f] dfull[r][c] ; partial[blk][Fh*C + c] = sum dfull[r][c]
__global__ __launch_bounds__(256) void head_bwd_dw_kernel(int64_t N, int Fh, int C,
                                                          int64_t rows_per_block,
                                                          const float* __restrict__ g,
                                                          const float* __restrict__ mask,
                                                          const float* __restrict__ atoms,
                                                          const float* __restrict__ pstd,
                                                          const float* __restrict__ dpeaks,
                                                          float* __restrict__ partial) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = std::min<int64_t>(r0 + rows_per_block, N);
  const int items = Fh * C + C;
  for (int it = threadIdx.x; it < items; it += 256) {
    float s = 0.f;
    if (it < Fh * C) {
      const int f = it / C, c = it % C;
      const float sc = pstd[c];
      for (int64_t r = r0; r < r1; ++r) {
        float x = g[r * Fh + f];
        if (mask) x *= mask[r * Fh + f];
        s += x * dpeaks[r] * atoms[r * C + c] * sc;
      }
    } else {
      const int c = it - Fh * C;
      const float sc = pstd[c];
      for (int64_t r = r0; r < r1; ++r) s += dpeaks[r] * atoms[r * C + c] * sc;
    }
    partial[(int64_t)blockIdx.x * items + it] = s;
  }
}

// ------------------------------------------------------------------------------------ AMPLayer
// nmrgnn/layers.py:81-100 (attention variant, forward only — the reference model never uses it):
//   q_i = h_i wq ; key_ij = e_ij wk ; qdot_ij = v_i <key_ij, q_i> ; b_i: = softmax_j(qdot_i:)
//   out_i = act( (sum_j b_ij h[nl_ij]) wv )            (values = h[nl] wv, sum pulled in front)
// One wave per atom: q and u = wk q by wave reductions, qdot on lanes j < K, softmax by DPP/shuffle,
// then the weighted gather.  The F x F product runs in the tall GEMM afterwards.
__global__ __launch_bounds__(256) void amp_attend_kernel(int64_t N, int K, int F, int E,
                                                         const float* __restrict__ h,
                                                         const int32_t* __restrict__ nlist,
                                                         const float* __restrict__ e,
                                                         const float* __restrict__ inv,
                                                         const float* __restrict__ wq,
                                                         const float* __restrict__ wk,
                                                         float* __restrict__ agg) {
  __shared__ float sq[4][64], su[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wv;
  if (i >= N) return;
  for (int n = 0; n < E; ++n) {                     // q[n] = sum_l h[i,l] wq[l,n]
    float p = 0.f;
    for (int l = lane; l < F; l += 64) p += h[i * F + l] * wq[l * E + n];
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    if (lane == 0) sq[wv][n] = p;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < E) {                                   // u[n] = sum_k wk[n,k] q[k]
    float p = 0.f;
    for (int k = 0; k < E; ++k) p += wk[lane * E + k] * sq[wv][k];
    su[wv][lane] = p;
  }
  __builtin_amdgcn_wave_barrier();
  float qd = -INFINITY;
  int nb = 0;
  if (lane < K) {
    const float* ep = e + (i * K + lane) * E;
    float p = 0.f;
    for (int n = 0; n < E; ++n) p += ep[n] * su[wv][n];
    qd = inv[i] * p;
    nb = nlist[i * K + lane];
  }
  float mx = qd;
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float ex = lane < K ? __expf(qd - mx) : 0.f;
  float sm = ex;
  for (int off = 32; off > 0; off >>= 1) sm += __shfl_xor(sm, off, 64);
  const float b = ex / sm;
  for (int l0 = 0; l0 < F; l0 += 64) {
    const int l = l0 + lane;
    float acc = 0.f;
    for (int j = 0; j < K; ++j) {
      const float bj = __shfl(b, j, 64);
      const int tj = __shfl(nb, j, 64);
      if (l < F) acc += bj * h[(int64_t)tj * F + l];
    }
    if (l < F) agg[i * F + l] = acc;
  }
}

// Backward of the attention aggregation.  With q = h_i wq, u = wk q, s_j = inv_i <e_ij, u>, b = softmax(s),
// agg_i = sum_j b_j h[nl_ij] and dA = d loss / d agg_i:
//   db_j = <dA, h[nl_ij]>,  ds_j = b_j (db_j - sum_k b_k db_k),  de_ij = inv_i ds_j u,
//   du = inv_i sum_j ds_j e_ij,  dq = wk^T du,  dwk = sum_i du q^T,  dwq = sum_i h_i dq^T,
//   dh_t = sum_{(i,j): nl_ij = t} b_ij dA_i  +  dq_t wq^T.
// Pass 1 (one wave per atom) recomputes q, u, b and leaves b [N,K], q, du, dq [N,E] in the workspace and de in
// place; pass 2 pulls dh over the incoming-slot lists (no atomics: the order of the sum is the list order);
// pass 3 forms the two weight gradients from per-block partial sums reduced in a fixed order.
__global__ __launch_bounds__(256) void amp_attend_bwd_atom_kernel(
    int64_t N, int K, int F, int E, const float* __restrict__ h, const int32_t* __restrict__ nlist,
    const float* __restrict__ e, const float* __restrict__ inv, const float* __restrict__ wq,
    const float* __restrict__ wk, const float* __restrict__ dagg, float* __restrict__ bsave,
    float* __restrict__ qbuf, float* __restrict__ dubuf, float* __restrict__ dqbuf, float* __restrict__ de) {
  __shared__ float sq[4][64], su[4][64], sdu[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wv;
  if (i >= N) return;
  for (int n = 0; n < E; ++n) {
    float p = 0.f;
    for (int l = lane; l < F; l += 64) p += h[i * F + l] * wq[l * E + n];
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    if (lane == 0) sq[wv][n] = p;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < E) {
    float p = 0.f;
    for (int k = 0; k < E; ++k) p += wk[lane * E + k] * sq[wv][k];
    su[wv][lane] = p;
  }
  __builtin_amdgcn_wave_barrier();
  const float iv = inv[i];
  float qd = -INFINITY;
  int nb = 0;
  if (lane < K) {
    const float* ep = e + (i * K + lane) * E;
    float p = 0.f;
    for (int n = 0; n < E; ++n) p += ep[n] * su[wv][n];
    qd = iv * p;
    nb = nlist[i * K + lane];
  }
  float mx = qd;
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  const float ex = lane < K ? __expf(qd - mx) : 0.f;
  float sm = ex;
  for (int off = 32; off > 0; off >>= 1) sm += __shfl_xor(sm, off, 64);
  const float b = ex / sm;
  float db = 0.f;
  for (int j = 0; j < K; ++j) {
    const int tj = __shfl(nb, j, 64);
    float p = 0.f;
    for (int l = lane; l < F; l += 64) p += dagg[i * F + l] * h[(int64_t)tj * F + l];
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    if (lane == j) db = p;
  }
  float t = b * db;
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  const float ds = b * (db - t);
  if (lane < K) {
    bsave[i * K + lane] = b;
    float* dp = de + (i * K + lane) * E;
    for (int n = 0; n < E; ++n) dp[n] = iv * ds * su[wv][n];
  }
  for (int n = 0; n < E; ++n) {
    float p = lane < K ? ds * e[(i * K + lane) * E + n] : 0.f;
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    if (lane == 0) sdu[wv][n] = iv * p;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < E) {
    float p = 0.f;
    for (int n = 0; n < E; ++n) p += wk[n * E + lane] * sdu[wv][n];
    qbuf[i * E + lane] = sq[wv][lane];
    dubuf[i * E + lane] = sdu[wv][lane];
    dqbuf[i * E + lane] = p;
  }
}

__global__ __launch_bounds__(256) void amp_attend_bwd_pull_kernel(
    int64_t N, int K, int F, int E, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_slot,
    const float* __restrict__ bsave, const float* __restrict__ dagg, const float* __restrict__ dqbuf,
    const float* __restrict__ wq, float* __restrict__ dh) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= N) return;
  const int a = in_ptr[t], z = in_ptr[t + 1];
  for (int l = lane; l < F; l += 64) {
    float acc = 0.f;
    for (int n = 0; n < E; ++n) acc += dqbuf[t * E + n] * wq[l * E + n];
    for (int c = a; c < z; ++c) {
      const int s = in_slot[c];
      acc += bsave[s] * dagg[(int64_t)(s / K) * F + l];
    }
    dh[t * F + l] = acc;
  }
}

// partial[b][r*C + c] = sum over the block's atoms of X[i,r] * Y[i,c]
__global__ __launch_bounds__(256) void amp_outer_partial_kernel(int64_t N, int R, int C, int64_t rows_per_block,
                                                                const float* __restrict__ X,
                                                                const float* __restrict__ Y,
                                                                float* __restrict__ partial) {
  const int64_t i0 = (int64_t)blockIdx.x * rows_per_block, i1 = min(N, i0 + rows_per_block);
  for (int idx = threadIdx.x; idx < R * C; idx += 256) {
    const int r = idx / C, c = idx % C;
    float acc = 0.f;
    for (int64_t i = i0; i < i1; ++i) acc += X[i * R + r] * Y[i * C + c];
    partial[(int64_t)blockIdx.x * R * C + idx] = acc;
  }
}

// ------------------------------------------------------------------------------------ loss
// one wave per graph: lg = sum w (y-p)^2 / sum w ; dpred = -2 w (y-p) / (sum w * G)
__global__ __launch_bounds__(256) void loss_graph_kernel(int G, const int32_t* __restrict__ gptr,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ pred,
                                                         float* __restrict__ per_graph,
                                                         float* __restrict__ dpred) {
  const int lane = threadIdx.x & 63;
  const int gidx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gidx >= G) return;
  const int a = gptr[gidx], b = gptr[gidx + 1];
  float sw = 0.f, sl = 0.f;
  for (int i = a + lane; i < b; i += 64) {
    const float d = y[i] - pred[i];
    sw += w[i];
    sl += w[i] * d * d;
  }
  for (int off = 32; off > 0; off >>= 1) {
    sw += __shfl_xor(sw, off, 64);
    sl += __shfl_xor(sl, off, 64);
  }
  const float inv = (sw != 0.f) ? 1.0f / sw : 0.f;
  if (lane == 0) per_graph[gidx] = sl * inv;
  const float sc = -2.0f * inv / (float)G;
  for (int i = a + lane; i < b; i += 64) dpred[i] = sc * w[i] * (y[i] - pred[i]);
}

// general NameLoss (losses.py:4-15,30-39): per graph  s*l2 + (1-s)*(1-r), r = weighted Pearson
// correlation in the reference's moment form; the handful of per-graph sums is kept in double so
// the xm2 - xm^2 cancellation does not eat the fp32 mantissa.  One wave per graph, three passes.
__device__ __forceinline__ double wave_sum_d(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void loss_name_kernel(int G, const int32_t* __restrict__ gptr,
                                                        const float* __restrict__ y,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ pred, float s,
                                                        float* __restrict__ per_graph,
                                                        float* __restrict__ dpred) {
  const int lane = threadIdx.x & 63;
  const int gidx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gidx >= G) return;
  const int a = gptr[gidx], b = gptr[gidx + 1];
  double m = 0, sx = 0, sy = 0, sxx = 0, syy = 0, sl = 0;
  for (int i = a + lane; i < b; i += 64) {
    const double wi = w[i], xi = pred[i], yi = y[i], d = yi - xi;
    m += wi; sx += wi * xi; sy += wi * yi; sxx += wi * xi * xi; syy += wi * yi * yi; sl += wi * d * d;
  }
  m = wave_sum_d(m); sx = wave_sum_d(sx); sy = wave_sum_d(sy);
  sxx = wave_sum_d(sxx); syy = wave_sum_d(syy); sl = wave_sum_d(sl);
  const double invm = m != 0 ? 1.0 / m : 0.0;
  const double xm = sx * invm, ym = sy * invm;
  const double vx = sxx * invm - xm * xm, vy = syy * invm - ym * ym;
  double cov = 0, syc = 0;
  for (int i = a + lane; i < b; i += 64) {
    const double wi = w[i];
    cov += wi * ((double)pred[i] - xm) * ((double)y[i] - ym);
    syc += wi * ((double)y[i] - ym);
  }
  cov = wave_sum_d(cov); syc = wave_sum_d(syc);
  const double prod = vx * vy;
  const bool inside = prod >= 0.0 && prod <= 1e32;               // clip_by_value passes gradient here
  const double root = sqrt(fmin(fmax(prod, 0.0), 1e32));
  const double den = m * root;
  const double r = den != 0 ? cov / den : 0.0;
  const double l2 = sl * invm;
  if (lane == 0) per_graph[gidx] = (float)(s * l2 + (1.0 - s) * (1.0 - r));
  // d r / d x_i = w_i[(y_i - ym) - syc/m]/den - cov/den^2 * m * vy * w_i (x_i - xm) / (m root)
  const double c1 = den != 0 ? 1.0 / den : 0.0;
  const double c2 = (den != 0 && inside && root != 0) ? cov / (den * den) * vy / root : 0.0;
  const double invG = 1.0 / (double)G;
  for (int i = a + lane; i < b; i += 64) {
    const double wi = w[i], xi = pred[i], yi = y[i];
    const double dl2 = -2.0 * wi * (yi - xi) * invm;
    const double dr = wi * ((yi - ym) - syc * invm) * c1 - c2 * wi * (xi - xm);
    dpred[i] = (float)((s * dl2 - (1.0 - s) * dr) * invG);
  }
}

__global__ __launch_bounds__(256) void loss_final_kernel(int G, const float* __restrict__ per_graph,
                                                         float* __restrict__ loss_out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < G; i += 256) s += per_graph[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = red[0] / (float)G;
}

// ------------------------------------------------------------------------------------ Adam
__global__ void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, float lr_t, float b1,
                            float b2, float eps, float gscale, const uint64_t* __restrict__ staged) {
  if (staged) lr_t = reinterpret_cast<const float*>(staged)[2];      // replayed step: the bias-corrected rate of THIS replay
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

// ------------------------------------------------------------------------------------ RNG
__global__ void randn_kernel(uint64_t seed, uint64_t offset, float* __restrict__ out, int64_t n) {
  const int64_t n4 = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4;
       q += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4x32(seed, offset + (uint64_t)q, r);
    float z[4];
    const float r0 = sqrtf(-2.0f * logf(u01(r[0]))), t0 = 6.28318530717958648f * u01(r[1]);
    const float r1 = sqrtf(-2.0f * logf(u01(r[2]))), t1 = 6.28318530717958648f * u01(r[3]);
    z[0] = r0 * cosf(t0); z[1] = r0 * sinf(t0); z[2] = r1 * cosf(t1); z[3] = r1 * sinf(t1);
    for (int k = 0; k < 4; ++k)
      if (q * 4 + k < n) out[q * 4 + k] = z[k];
  }
}

__global__ void dropout_mask_kernel(uint64_t seed, uint64_t offset, float keep,
                                    float* __restrict__ out, int64_t n) {
  const int64_t n4 = (n + 3) / 4;
  const float inv = 1.0f / keep;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4;
       q += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4x32(seed, offset + (uint64_t)q, r);
    for (int k = 0; k < 4; ++k)
      if (q * 4 + k < n) out[q * 4 + k] = (u01(r[k]) <= keep) ? inv : 0.f;
  }
}

// out = x + alpha * y   (GaussianNoise: d_eff = d + sigma * xi, model.py:253)
__global__ void add_scaled_kernel(int64_t n, const float* __restrict__ x,
                                  const float* __restrict__ y, float alpha,
                                  float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = x[i] + alpha * y[i];
}

// out = x + alpha * xi with the xi of randn_kernel(seed, offset) — GaussianNoise in one launch (model.py:253); same
// expressions as randn_kernel followed by add_scaled_kernel, so the bits are those of the two-launch form
__global__ void add_noise_kernel(uint64_t seed, uint64_t offset, int64_t n, const float* __restrict__ x, float alpha,
                                 float* __restrict__ out, const uint64_t* __restrict__ staged) {
  if (staged) seed = staged[0];      // replayed step: the seed of THIS replay (ng_replay_stage)
  const int64_t n4 = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4;
       q += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4x32(seed, offset + (uint64_t)q, r);
    float z[4];
    const float r0 = sqrtf(-2.0f * logf(u01(r[0]))), t0 = 6.28318530717958648f * u01(r[1]);
    const float r1 = sqrtf(-2.0f * logf(u01(r[2]))), t1 = 6.28318530717958648f * u01(r[3]);
    z[0] = r0 * cosf(t0); z[1] = r0 * sinf(t0); z[2] = r1 * cosf(t1); z[3] = r1 * sinf(t1);
    for (int k = 0; k < 4; ++k)
      if (q * 4 + k < n) out[q * 4 + k] = x[q * 4 + k] + alpha * z[k];
  }
}

// GaussianNoise over the live-edge view (round 4): out_c[pos[g]] = x[g] + alpha * xi_g for every live slot g (pos[g] >= 0),
// xi_g the draw add_noise_kernel gives slot g (same counters, same expressions: same bits), or y[g] when the caller
// supplies the draws.  Dead slots get nothing: the compacted edge kernels never see them.
__global__ void add_noise_live_kernel(uint64_t seed, uint64_t offset, int64_t n, const float* __restrict__ x,
                                      const float* __restrict__ y, float alpha, const int32_t* __restrict__ pos,
                                      float* __restrict__ out_c, const uint64_t* __restrict__ staged) {
  if (staged) seed = staged[0];
  const int64_t n4 = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4;
       q += (int64_t)gridDim.x * blockDim.x) {
    float z[4];
    if (y) {
      for (int k = 0; k < 4; ++k) z[k] = q * 4 + k < n ? y[q * 4 + k] : 0.f;
    } else {
      uint32_t r[4];
      philox4x32(seed, offset + (uint64_t)q, r);
      const float r0 = sqrtf(-2.0f * logf(u01(r[0]))), t0 = 6.28318530717958648f * u01(r[1]);
      const float r1 = sqrtf(-2.0f * logf(u01(r[2]))), t1 = 6.28318530717958648f * u01(r[3]);
      z[0] = r0 * cosf(t0); z[1] = r0 * sinf(t0); z[2] = r1 * cosf(t1); z[3] = r1 * sinf(t1);
    }
    for (int k = 0; k < 4; ++k)
      if (q * 4 + k < n) {
        const int32_t p = pos[q * 4 + k];
        if (p >= 0) out_c[p] = x[q * 4 + k] + alpha * z[k];
      }
  }
}

static inline dim3 ew_grid(int64_t work_items) {
  return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(work_items, 256), 256 * 8)));
}

}  // namespace ng

using namespace ng;

// ===================================================================================== C ABI
extern "C" int ng_randn(ng_ctx* ctx, void* stream, uint64_t seed, uint64_t offset, float* out,
                        int64_t n) {
  if (!ctx) return NG_ERR_INVALID;
  if (n == 0) return NG_OK;
  hipLaunchKernelGGL(randn_kernel, ew_grid((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, seed,
                     offset, out, n);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_dropout_mask(ng_ctx* ctx, void* stream, uint64_t seed, uint64_t offset, float keep,
                               float* out, int64_t n) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, keep > 0.f && keep <= 1.f, "dropout keep probability in (0,1]");
  if (n == 0) return NG_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, ew_grid((n + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     seed, offset, keep, out, n);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_add_noise(ng_ctx* ctx, void* stream, uint64_t seed, uint64_t offset, int64_t n, const float* x,
                            float alpha, float* out) {
  if (!ctx) return NG_ERR_INVALID;
  if (n == 0) return NG_OK;
  hipLaunchKernelGGL(add_noise_kernel, ew_grid((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, seed, offset, n, x,
                     alpha, out, ng::replay_state(ctx));
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_add_noise_live(ng_ctx* ctx, void* stream, uint64_t seed, uint64_t offset, int64_t n, const float* x,
                                 const float* y, float alpha, const int32_t* pos, float* out_c) {
  if (!ctx) return NG_ERR_INVALID;
  if (n == 0) return NG_OK;
  NG_REQUIRE(ctx, x && pos && out_c, "add_noise_live: arguments");
  hipLaunchKernelGGL(add_noise_live_kernel, ew_grid((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, seed, offset, n, x, y,
                     alpha, pos, out_c, ng::replay_state(ctx));
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_add_scaled(ng_ctx* ctx, void* stream, int64_t n, const float* x, const float* y,
                             float alpha, float* out) {
  if (!ctx) return NG_ERR_INVALID;
  if (n == 0) return NG_OK;
  hipLaunchKernelGGL(add_scaled_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, n, x, y,
                     alpha, out);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_embed_fwd(ng_ctx* ctx, void* stream, int64_t N, int C, int F, const float* atoms,
                            const float* Wemb, float* h0) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, F % 4 == 0, "embed: F % 4");
  if (N == 0) return NG_OK;
  ProfScope ps(ctx, (hipStream_t)stream, "embed_fwd");
  const dim3 grid = ew_grid(N * (F / 4));
  if (C <= 16 && F == 64 && !sw().head_generic)
    hipLaunchKernelGGL((embed_fwd_rows_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, N, C, atoms, Wemb, h0);
  else if (C <= 16 && F == 128 && !sw().head_generic)
    hipLaunchKernelGGL((embed_fwd_rows_kernel<32>), grid, dim3(256), 0, (hipStream_t)stream, N, C, atoms, Wemb, h0);
  else if (C <= 16 && F == 256 && !sw().head_generic)
    hipLaunchKernelGGL((embed_fwd_rows_kernel<64>), grid, dim3(256), 0, (hipStream_t)stream, N, C, atoms, Wemb, h0);
  else
    hipLaunchKernelGGL(embed_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, N, C, F, atoms, Wemb, h0);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_embed_bwd(ng_ctx* ctx, void* stream, int64_t N, int C, int F, const float* atoms,
                            const float* dh0, float* dWemb) {
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int64_t items = (int64_t)C * F;
  if (N == 0) {
    NG_HIP(ctx, hipMemsetAsync(dWemb, 0, items * 4, st));
    return NG_OK;
  }
  // molecule-sized call inside a deferred-reduction window: the sum is a job of the batch launch, no launch of its own (the caller
  // keeps atoms / dh0 alive until the flush, as it does for every deferred partial)
  if (defer_outer_job(ctx, st, atoms, dh0, N, C, F, dWemb)) return NG_OK;
  if (embed_bwd_fast_supported(F, C)) return embed_bwd_fast(ctx, st, N, C, F, atoms, dh0, dWemb);
  // dWemb[c][f] = sum_i atoms[i][c] dh0[i][f]: small transposed product, rows staged through LDS
  const int64_t rows = 512;
  const int64_t nb = cdiv(N, rows);
  float* partial = (float*)workspace(ctx, nb * items * 4);
  if (!partial) return NG_ERR_NOMEM;
  ProfScope ps(ctx, st, "embed_bwd");
  LoadElem fx{atoms, C};
  LoadElem fy{dh0, F};
  const dim3 grid((unsigned)nb, (unsigned)cdiv(items, 256 * SMALL_TN_ITEMS));
  hipLaunchKernelGGL((small_tn_kernel<LoadElem, LoadElem>), grid, dim3(256),
                     (size_t)64 * (C + F) * 4, st, N, C, F, rows, fx, fy, partial);
  launch_reduce_z(st, partial, (int)nb, items, dWemb);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_mp_aggregate(ng_ctx* ctx, void* stream, int64_t N, int K, int F, int E,
                               const float* h, const int32_t* nlist, const float* e, float* A) {
  if (!ctx) return NG_ERR_INVALID;
  return aggregate(ctx, (hipStream_t)stream, N, K, F, E, h, nlist, e, A);
}

extern "C" int ng_mp_layer_fwd(ng_ctx* ctx, void* stream, int64_t N, int K, int F, int E, int act,
                               int residual, const float* h, const int32_t* nlist, const float* e,
                               const float* inv_degree, const float* w, float* h_out, float* A_save,
                               float* s_save) {
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  NG_REQUIRE(ctx, (E * F) % 8 == 0, "mp_layer: (E*F) % 8");
  // F == 64: window-resident kernel (mp_win.hip); NG_MP_PATH=layered opts out
  if (N > 0 && mp_win_enabled(F, E, K)) {
    // the window kernel never materialises the aggregate; a caller that asks for it gets a separate pass
    if (A_save) {
      const int rc = aggregate(ctx, st, N, K, F, E, h, nlist, e, A_save);
      if (rc) return rc;
    }
    return mp_win_fwd(ctx, st, N, K, E, act, residual, h, nlist, e, inv_degree, w, h_out, s_save);
  }
  if (E > MAX_E)   // edge_feature_size = 64 (model.py:23): feature-chunked kernels of mp_csr.hip, fixed stride K
    return mp_generic_fwd(ctx, st, N, K, F, E, act, residual, h, nullptr, nlist, e, inv_degree, w, h_out, A_save, s_save);
  // the reference's default width: gather-GEMM, the aggregate never reaches HBM (gemm_h2.hip: mp_gg_kernel)
  // (a caller that keeps the aggregate for the backward's dw = A^T dP gets it as a by-product of the producer waves)
  if (mp_gg_supported(N, F, E, K) || mp_gw_infer_ok(ctx, N, K, F, E, false, A_save != nullptr))
    return mp_gg_fwd(ctx, st, N, K, F, E, act, residual, h, nullptr, nlist, e, inv_degree, w, h_out, s_save, A_save);
  const int64_t KF = (int64_t)E * F;
  // scratch: Wp [KF*F] (+ A [N*KF] when the caller does not keep it)
  const size_t need = (size_t)(KF * F + (A_save ? 0 : N * KF)) * 4;
  float* ws = (float*)workspace(ctx, need);
  if (!ws) return NG_ERR_NOMEM;
  const float* Wp = nullptr;
  float* A = A_save ? A_save : ws + KF * F;
  int rc = mp_plain_weights(ctx, st, N, F, E, w, ws, 0, &Wp);
  if (rc) return rc;
  rc = aggregate(ctx, st, N, K, F, E, h, nlist, e, A);
  if (rc) return rc;
  // P = inv * (A @ Wp);  h_out = act(P) + h
  return dense_fwd(ctx, st, N, (int)KF, F, act, A, Wp, nullptr, inv_degree, residual ? h : nullptr,
                   h_out, s_save,
                   "mp_update_fwd");
}

extern "C" int ng_mp_layer_wants_aggregate(int F, int E, int K) {
  if (mp_win_bwd_enabled(F, E, K)) return 0;      // dw comes from h^T B (incoming-edge aggregate of dP)
  return 1;
}

extern "C" int ng_mp_edge_records(ng_ctx* ctx, void* stream, int64_t N, int K, int E, const int32_t* csc_ptr,
                                  const int32_t* csc_edge, const float* e, float* rec) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, E >= 1 && E <= 3, "edge records hold at most 3 edge features");
  return mp_win_records(ctx, (hipStream_t)stream, N, K, E, csc_ptr, csc_edge, e, rec);
}

extern "C" int ng_mp_layer_bwd_rec(ng_ctx* ctx, void* stream, int64_t N, int K, int F, int E, int act,
                                   const float* h, const int32_t* nlist, const float* e,
                                   const float* inv_degree, const float* w, const float* A_save,
                                   const float* s_save, const int32_t* csc_ptr, const int32_t* csc_edge,
                                   const float* dh_out, float* dh_in, float* de, int de_accum,
                                   float* dw, const float* csc_rec);

extern "C" int ng_mp_layer_bwd(ng_ctx* ctx, void* stream, int64_t N, int K, int F, int E, int act,
                               const float* h, const int32_t* nlist, const float* e,
                               const float* inv_degree, const float* w, const float* A_save,
                               const float* s_save, const int32_t* csc_ptr, const int32_t* csc_edge,
                               const float* dh_out, float* dh_in, float* de, int de_accum,
                               float* dw) {
  return ng_mp_layer_bwd_rec(ctx, stream, N, K, F, E, act, h, nlist, e, inv_degree, w, A_save, s_save, csc_ptr,
                             csc_edge, dh_out, dh_in, de, de_accum, dw, nullptr);
}

extern "C" int ng_mp_layer_bwd_rec(ng_ctx* ctx, void* stream, int64_t N, int K, int F, int E, int act,
                                   const float* h, const int32_t* nlist, const float* e,
                                   const float* inv_degree, const float* w, const float* A_save,
                                   const float* s_save, const int32_t* csc_ptr, const int32_t* csc_edge,
                                   const float* dh_out, float* dh_in, float* de, int de_accum,
                                   float* dw, const float* csc_rec) {
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  NG_REQUIRE(ctx, act == NG_ACT_NONE || s_save, "mp_layer_bwd: s_save required for an activation");
  NG_REQUIRE(ctx, F % 4 == 0 && F >= 16 && F <= 256 && (256 % (F / 4)) == 0,
             "mp_layer_bwd: F in {16,32,64,128,256}");
  NG_REQUIRE(ctx, E >= 1 && E <= 64, "mp_layer_bwd: edge_feature_size <= 64");
  if (N > 0 && mp_win_bwd_enabled(F, E, K))
    return mp_win_bwd(ctx, st, N, K, E, act, h, nlist, e, inv_degree, w, s_save, csc_ptr, csc_edge, dh_out, dh_in, de,
                      de_accum, dw, csc_rec);
  // generic path (any F / E / K): the kernels of mp_csr.hip with the fixed stride K in place of row_ptr
  return mp_generic_bwd(ctx, st, N, K, F, E, act, h, nullptr, nlist, nullptr, e, inv_degree, w, A_save, s_save, csc_ptr,
                        csc_edge, dh_out, dh_in, de, de_accum, dw, E <= 3 ? csc_rec : nullptr);
}

extern "C" int ng_head_fwd(ng_ctx* ctx, void* stream, int64_t N, int Fh, int C, const float* g,
                           const float* drop_mask, const float* Wout, const float* bout,
                           const float* atoms, const float* peak_std, const float* peak_avg,
                           float* peaks) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, C >= 1 && C <= MAX_C, "head: number of elements <= 32");
  if (N == 0) return NG_OK;
  if (head_fwd_fast_supported(Fh, C))
    return head_fwd_fast(ctx, (hipStream_t)stream, N, Fh, C, g, drop_mask, Wout, bout, atoms, peak_std,
                         peak_avg, peaks);
  ProfScope ps(ctx, (hipStream_t)stream, "head_fwd");
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), (size_t)Fh * C * 4,
                     (hipStream_t)stream, N, Fh, C, g, drop_mask, Wout, bout, atoms, peak_std,
                     peak_avg, peaks);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_head_fwd_dropout(ng_ctx* ctx, void* stream, int64_t N, int Fh, int C, const float* g, uint64_t seed,
                                   uint64_t offset, float keep, float* mask_out, const float* Wout, const float* bout,
                                   const float* atoms, const float* peak_std, const float* peak_avg, float* peaks) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, C >= 1 && C <= MAX_C, "head: number of elements <= 32");
  NG_REQUIRE(ctx, keep > 0.f && keep <= 1.f, "dropout keep probability in (0,1]");
  NG_REQUIRE(ctx, mask_out, "head_fwd_dropout: mask_out [N,Fh] required (the backward reads it)");
  if (N == 0) return NG_OK;
  if (head_fwd_fast_supported(Fh, C) && Fh % 4 == 0)
    return head_fwd_fast(ctx, (hipStream_t)stream, N, Fh, C, g, nullptr, Wout, bout, atoms, peak_std, peak_avg, peaks,
                         seed, offset, keep, mask_out);
  // other shapes: the draw and the head as two launches
  NG_REQUIRE(ctx, !ctx->replay_armed, "head_fwd_dropout: this head shape cannot be replayed (seed is a launch argument)");
  const int rc = ng_dropout_mask(ctx, stream, seed, offset, keep, mask_out, N * Fh);
  if (rc) return rc;
  return ng_head_fwd(ctx, stream, N, Fh, C, g, mask_out, Wout, bout, atoms, peak_std, peak_avg, peaks);
}

// head forward + NameLoss(s = 1) + head backward as one launch (head_ops.hip: head_loss_kernel)
extern "C" int ng_head_loss_blocks(ng_ctx* ctx, int G, int Fh, int C, int64_t max_graph_atoms) {
  if (!ctx) return 0;
  const int gpw = head_loss_graphs_per_wg(ctx, G, Fh, C, max_graph_atoms);
  return gpw ? (int)cdiv(G, gpw) : 0;
}

extern "C" int ng_head_loss_bwd(ng_ctx* ctx, void* stream, int64_t N, int G, int Fh, int C, int64_t max_graph_atoms,
                                const float* g, uint64_t seed, uint64_t offset, float keep, float* mask_out,
                                const float* Wout, const float* bout, const float* atoms, const float* peak_std,
                                const float* peak_avg, const int32_t* graph_ptr, const float* y, const float* w,
                                float grad_weight, float* peaks, float* dg, float* partial) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, C >= 1 && C <= MAX_C, "head: number of elements <= 32");
  NG_REQUIRE(ctx, keep > 0.f && keep <= 1.f, "dropout keep probability in (0,1]");
  NG_REQUIRE(ctx, N >= 1 && G >= 1, "head_loss_bwd: at least one graph and one atom");
  NG_REQUIRE(ctx, g && Wout && bout && atoms && peak_std && peak_avg && graph_ptr && y && w && peaks && dg && partial,
             "head_loss_bwd: arguments");
  const int gpw = head_loss_graphs_per_wg(ctx, G, Fh, C, max_graph_atoms);
  if (!gpw) { ctx->err = "head_loss_bwd: shape not supported (ng_head_loss_blocks == 0)"; return NG_ERR_UNSUPPORTED; }
  return head_loss_launch(ctx, (hipStream_t)stream, N, G, Fh, C, gpw, g, seed, offset, keep, keep < 1.f, mask_out, Wout, bout,
                          atoms, peak_std, peak_avg, graph_ptr, y, w, grad_weight, peaks, dg, partial);
}

extern "C" int ng_head_loss_reduce(ng_ctx* ctx, void* stream, const float* partial, int nb, int Fh, int C, float* dWout,
                                   float* dbout, float* loss_out) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, partial && nb >= 1 && dWout && dbout && loss_out && C >= 1 && C <= MAX_C, "head_loss_reduce: arguments");
  return head_loss_reduce(ctx, (hipStream_t)stream, partial, nb, Fh, C, dWout, dbout, loss_out);
}

extern "C" int ng_head_bwd(ng_ctx* ctx, void* stream, int64_t N, int Fh, int C, const float* g,
                           const float* drop_mask, const float* Wout, const float* atoms,
                           const float* peak_std, const float* dpeaks, float* dg, float* dWout,
                           float* dbout) {
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  NG_REQUIRE(ctx, C >= 1 && C <= MAX_C, "head: number of elements <= 32");
  const int64_t items = (int64_t)Fh * C + C;
  if (N == 0) {
    NG_HIP(ctx, hipMemsetAsync(dWout, 0, (size_t)Fh * C * 4, st));
    NG_HIP(ctx, hipMemsetAsync(dbout, 0, (size_t)C * 4, st));
    return NG_OK;
  }
  if (head_fast_supported(Fh, C))
    return head_bwd_fast(ctx, st, N, Fh, C, g, drop_mask, Wout, atoms, peak_std, dpeaks, dg, dWout, dbout);
  const int64_t rows = 512;
  const int64_t nb = cdiv(N, rows);
  float* ws = (float*)workspace(ctx, (size_t)(nb * items + items) * 4);
  if (!ws) return NG_ERR_NOMEM;
  float* partial = ws;
  float* summed = ws + nb * items;
  ProfScope ps(ctx, st, "head_bwd");
  hipLaunchKernelGGL(head_bwd_dg_kernel, ew_grid(N * Fh), dim3(256), 0, st, N, Fh, C, drop_mask,
                     Wout, atoms, peak_std, dpeaks, dg);
  // [dWout ; dbout][f][c] = sum_i [g*mask | 1][i][f] * dfull[i][c]
  LoadHeadX fx{g, drop_mask, Fh};
  LoadHeadY fy{dpeaks, atoms, peak_std, C};
  const dim3 grid((unsigned)nb, (unsigned)cdiv(items, 256 * SMALL_TN_ITEMS));
  hipLaunchKernelGGL((small_tn_kernel<LoadHeadX, LoadHeadY>), grid, dim3(256),
                     (size_t)64 * (Fh + 1 + C) * 4, st, N, Fh + 1, C, rows, fx, fy, partial);
  launch_reduce_z(st, partial, (int)nb, items, summed);
  NG_HIP(ctx, hipMemcpyAsync(dWout, summed, (size_t)Fh * C * 4, hipMemcpyDeviceToDevice, st));
  NG_HIP(ctx, hipMemcpyAsync(dbout, summed + (size_t)Fh * C, (size_t)C * 4, hipMemcpyDeviceToDevice,
                             st));
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_amp_attend(ng_ctx* ctx, void* stream, int64_t N, int K, int F, int E,
                             const float* h, const int32_t* nlist, const float* e,
                             const float* inv_degree, const float* wq, const float* wk, float* agg) {
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  NG_REQUIRE(ctx, K >= 1 && K <= 64, "amp: neighbour count must be in [1,64]");
  NG_REQUIRE(ctx, E >= 1 && E <= 64, "amp: edge feature size must be in [1,64]");
  NG_REQUIRE(ctx, F >= 1, "amp: bad feature size");
  if (N == 0) return NG_OK;
  ProfScope ps(ctx, st, "amp_attend");
  hipLaunchKernelGGL(amp_attend_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, st, N, K, F, E, h,
                     nlist, e, inv_degree, wq, wk, agg);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_amp_attend_bwd(ng_ctx* ctx, void* stream, int64_t N, int K, int F, int E, const float* h,
                                 const int32_t* nlist, const float* e, const float* inv_degree,
                                 const float* wq, const float* wk, const int32_t* in_ptr,
                                 const int32_t* in_slot, const float* dagg, float* dh, float* de, float* dwq,
                                 float* dwk) {
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  NG_REQUIRE(ctx, K >= 1 && K <= 64, "amp: neighbour count must be in [1,64]");
  NG_REQUIRE(ctx, E >= 1 && E <= 64, "amp: edge feature size must be in [1,64]");
  NG_REQUIRE(ctx, F >= 1, "amp: bad feature size");
  if (N == 0) {
    NG_HIP(ctx, hipMemsetAsync(dwq, 0, (size_t)F * E * 4, st));
    NG_HIP(ctx, hipMemsetAsync(dwk, 0, (size_t)E * E * 4, st));
    return NG_OK;
  }
  const int64_t rows_per_block = 256;
  const int64_t nb = cdiv(N, rows_per_block);
  const size_t n_small = (size_t)N * E, n_b = (size_t)N * K;
  const size_t part = (size_t)nb * ((size_t)F * E + (size_t)E * E);
  float* ws = (float*)workspace(ctx, (n_b + 3 * n_small + part) * 4);
  if (!ws) return NG_ERR_NOMEM;
  float *bsave = ws, *qbuf = bsave + n_b, *dubuf = qbuf + n_small, *dqbuf = dubuf + n_small;
  float *pq = dqbuf + n_small, *pk = pq + (size_t)nb * F * E;
  ProfScope ps(ctx, st, "amp_attend_bwd");
  hipLaunchKernelGGL(amp_attend_bwd_atom_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, st, N, K, F, E, h,
                     nlist, e, inv_degree, wq, wk, dagg, bsave, qbuf, dubuf, dqbuf, de);
  hipLaunchKernelGGL(amp_attend_bwd_pull_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, st, N, K, F, E,
                     in_ptr, in_slot, bsave, dagg, dqbuf, wq, dh);
  hipLaunchKernelGGL(amp_outer_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, N, F, E, rows_per_block, h,
                     dqbuf, pq);
  hipLaunchKernelGGL(amp_outer_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, N, E, E, rows_per_block,
                     dubuf, qbuf, pk);
  launch_reduce_z(st, pq, (int)nb, (int64_t)F * E, dwq);
  launch_reduce_z(st, pk, (int)nb, (int64_t)E * E, dwk);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_loss_l2(ng_ctx* ctx, void* stream, int64_t N, int G, const int32_t* graph_ptr,
                          const float* y, const float* w, const float* pred, float* loss_out,
                          float* dpred) {
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  NG_REQUIRE(ctx, G >= 1, "loss: at least one graph");
  (void)N;
  float* per_graph = (float*)workspace(ctx, (size_t)G * 4);
  if (!per_graph) return NG_ERR_NOMEM;
  ProfScope ps(ctx, st, "loss_l2");
  hipLaunchKernelGGL(loss_graph_kernel, dim3((unsigned)cdiv(G, 4)), dim3(256), 0, st, G, graph_ptr,
                     y, w, pred, per_graph, dpred);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, G, per_graph, loss_out);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_loss_name(ng_ctx* ctx, void* stream, int64_t N, int G, const int32_t* graph_ptr,
                            const float* y, const float* w, const float* pred, float s,
                            float* loss_out, float* dpred) {
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  NG_REQUIRE(ctx, G >= 1, "loss: at least one graph");
  NG_REQUIRE(ctx, s >= 0.f && s <= 1.f, "loss: balance s must lie in [0,1]");
  (void)N;
  float* per_graph = (float*)workspace(ctx, (size_t)G * 4);
  if (!per_graph) return NG_ERR_NOMEM;
  ProfScope ps(ctx, st, "loss_name");
  hipLaunchKernelGGL(loss_name_kernel, dim3((unsigned)cdiv(G, 4)), dim3(256), 0, st, G, graph_ptr,
                     y, w, pred, s, per_graph, dpred);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, G, per_graph, loss_out);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_adam_step(ng_ctx* ctx, void* stream, int64_t n, float* p, const float* g, float* m,
                            float* v, float lr, float beta1, float beta2, float eps, int64_t step,
                            float grad_scale) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, step >= 1, "adam: step counts from 1");
  ctx->wver++;                       // packed weight images of a frozen-weight cache are stale from here on
  if (n == 0) return NG_OK;
  const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)step)) /
                      (1.0 - std::pow((double)beta1, (double)step));
  {
    ProfScope ps(ctx, (hipStream_t)stream, "adam");
    hipLaunchKernelGGL(adam_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v,
                       (float)lr_t, beta1, beta2, eps, grad_scale, ng::replay_state(ctx));
    NG_HIP(ctx, hipGetLastError());
  }
  // every registered weight image fed by the block just updated: ONE launch here instead of one in front of each consumer
  // of the next step (repack.hip; a no-op unless the image cache is on, ng_weights_frozen)
  return repack_all(ctx, (hipStream_t)stream, p, p + n);
}
