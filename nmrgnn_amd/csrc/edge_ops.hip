// Edge path: mask + RBFExpansion + EdgeFCBlock (forward and backward).
// Reference: nmrgnn/model.py:251-261 (mask, noise, rbf, edge MLP, mask), nmrgnn/layers.py:137-140
// (RBF), nmrgnn/model.py:111-138 (EdgeFCBlock).
//
// Two implementations sit behind ng_edge_mlp_fwd / ng_edge_mlp_bwd:
//   * the layered path in this file: RBF tile materialised, one MFMA GEMM launch per Dense layer
//     (any H % 16 == 0, any Le >= 2);
//   * the fused persistent path in edge_fused.hip (H == 128): RBF generated in LDS, all hidden
//     layers chained on-chip, only e[N,K,E] (and the saved activations when training) leave the CU.
#include <algorithm>

#include "mfma_gemm.cuh"
#include "edge_fused.h"

namespace ng {

constexpr int MAX_E = 8;

// X0[e][k] = (d_src[e] > 0) * exp(-(d_eff[e]-centers[k])^2 / gap)
__global__ void rbf_kernel(int64_t n_edges, int H, const float* __restrict__ d_src,
                           const float* __restrict__ d_eff, const float* __restrict__ centers,
                           float neg_inv_gap, float* __restrict__ X0) {
  const int c4n = H / 4;
  const int64_t total = n_edges * c4n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t eidx = t / c4n;
    const int c4 = (int)(t % c4n);
    float4 v = f4zero();
    if (d_src[eidx] > 0.f) {
      const float d = d_eff[eidx];
      const float4 mu = *reinterpret_cast<const float4*>(centers + c4 * 4);
      const float a = d - mu.x, b = d - mu.y, c = d - mu.z, dd = d - mu.w;
      v.x = __expf(a * a * neg_inv_gap);
      v.y = __expf(b * b * neg_inv_gap);
      v.z = __expf(c * c * neg_inv_gap);
      v.w = __expf(dd * dd * neg_inv_gap);
    }
    *reinterpret_cast<float4*>(X0 + eidx * H + c4 * 4) = v;
  }
}

// last (linear) edge layer with tiny output width E:  e[r][n] = m_r * (Z[r][:] @ W[:, n] + b[n])
// 4 lanes per row, each covering a quarter of every 16-float chunk; shuffle-reduce over the 4.
template <int E>
__global__ __launch_bounds__(256) void edge_out_fwd_kernel(int64_t n_edges, int H,
                                                           const float* __restrict__ Z,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ b,
                                                           const float* __restrict__ d_src,
                                                           float* __restrict__ e_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sW = reinterpret_cast<float*>(smem_raw);  // [H*E]
  for (int t = threadIdx.x; t < H * E; t += 256) sW[t] = W[t];
  __syncthreads();
  const int q = threadIdx.x & 3;
  const int64_t r = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
  const bool live = r < n_edges;
  const int64_t rr = live ? r : 0;
  float acc[E];
#pragma unroll
  for (int n = 0; n < E; ++n) acc[n] = 0.f;
  for (int k0 = 0; k0 < H; k0 += 16) {
    const float4 x = *reinterpret_cast<const float4*>(Z + rr * H + k0 + q * 4);
    const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int n = 0; n < E; ++n) acc[n] += xs[s] * sW[(k0 + q * 4 + s) * E + n];
  }
#pragma unroll
  for (int n = 0; n < E; ++n) {
    acc[n] += __shfl_xor(acc[n], 1, 64);
    acc[n] += __shfl_xor(acc[n], 2, 64);
  }
  if (live && q == 0) {
    const float m = d_src[r] > 0.f ? 1.f : 0.f;
#pragma unroll
    for (int n = 0; n < E; ++n) e_out[r * E + n] = m * (acc[n] + b[n]);
  }
}

// backward of the last layer:
//   dE = m * de ;  dZ[r][k] = sum_n dE[r][n] W[k][n]
//   partial[blk][k*E+n] = sum_{r in chunk} Z[r][k] dE[r][n] ; partial[blk][H*E+n] = sum dE[r][n]
template <int E>
__global__ __launch_bounds__(256) void edge_out_bwd_kernel(int64_t n_edges, int H,
                                                           int64_t rows_per_block,
                                                           const float* __restrict__ Z,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ d_src,
                                                           const float* __restrict__ de,
                                                           float* __restrict__ dZ,
                                                           float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sW = reinterpret_cast<float*>(smem_raw);   // [H*E]
  float* sdE = sW + H * E;                          // [64*E]
  for (int t = threadIdx.x; t < H * E; t += 256) sW[t] = W[t];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = std::min<int64_t>(r0 + rows_per_block, n_edges);
  // each thread owns columns k = tid, tid+256, ... (H <= 512 -> at most 2)
  float accw[2][E];
  float accb[E];
#pragma unroll
  for (int n = 0; n < E; ++n) { accw[0][n] = 0.f; accw[1][n] = 0.f; accb[n] = 0.f; }
  for (int64_t rb = r0; rb < r1; rb += 64) {
    const int nr = (int)std::min<int64_t>(64, r1 - rb);
    __syncthreads();
    for (int t = threadIdx.x; t < nr * E; t += 256) {
      const int64_t r = rb + t / E;
      sdE[t] = d_src[r] > 0.f ? de[rb * E + t] : 0.f;
    }
    __syncthreads();
    // dZ tile: thread -> (row, k4)
    const int c4n = H / 4;
    for (int t = threadIdx.x; t < nr * c4n; t += 256) {
      const int rr = t / c4n, c4 = t % c4n;
      float4 v = f4zero();
#pragma unroll
      for (int n = 0; n < E; ++n) {
        const float g = sdE[rr * E + n];
        v.x += g * sW[(c4 * 4 + 0) * E + n];
        v.y += g * sW[(c4 * 4 + 1) * E + n];
        v.z += g * sW[(c4 * 4 + 2) * E + n];
        v.w += g * sW[(c4 * 4 + 3) * E + n];
      }
      *reinterpret_cast<float4*>(dZ + (rb + rr) * H + c4 * 4) = v;
    }
    // weight-gradient accumulation
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int k = threadIdx.x + c * 256;
      if (k < H) {
        for (int rr = 0; rr < nr; ++rr) {
          const float z = Z[(rb + rr) * H + k];
#pragma unroll
          for (int n = 0; n < E; ++n) accw[c][n] += z * sdE[rr * E + n];
        }
      }
    }
    if (threadIdx.x < E) {
      for (int rr = 0; rr < nr; ++rr) accb[0] += sdE[rr * E + threadIdx.x];
    }
  }
  float* out = partial + (int64_t)blockIdx.x * (H * E + E);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int k = threadIdx.x + c * 256;
    if (k < H) {
#pragma unroll
      for (int n = 0; n < E; ++n) out[k * E + n] = accw[c][n];
    }
  }
  if (threadIdx.x < E) out[H * E + threadIdx.x] = accb[0];
}

__global__ void sum_partials2_kernel(const float* __restrict__ partial, int nz, int64_t n_elem,
                                     float* __restrict__ out) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_elem;
       idx += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += partial[(int64_t)z * n_elem + idx];
    out[idx] = s;
  }
}

static inline dim3 ew_grid(int64_t work_items) {
  return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(work_items, 256), 256 * 8)));
}

int edge_out_fwd(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int H, int E, const float* Z,
                 const float* W, const float* b, const float* d_src, float* e_out) {
  ProfScope ps(ctx, st, "edge_out_fwd");
  const dim3 grid((unsigned)cdiv(n_edges, 64));
  const size_t lds = (size_t)H * E * 4;
#define NG_EO(EE)                                                                                 \
  case EE:                                                                                        \
    hipLaunchKernelGGL((edge_out_fwd_kernel<EE>), grid, dim3(256), lds, st, n_edges, H, Z, W, b,  \
                       d_src, e_out);                                                             \
    break;
  switch (E) { NG_EO(1) NG_EO(2) NG_EO(3) NG_EO(4) NG_EO(5) NG_EO(6) NG_EO(7) NG_EO(8) }
#undef NG_EO
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

// scratch floats needed by edge_out_bwd
static int64_t edge_out_bwd_blocks(int64_t n_edges) {
  return std::max<int64_t>(1, std::min<int64_t>(cdiv(n_edges, 2048), 2048));
}

int edge_out_bwd(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int H, int E, const float* Z,
                 const float* W, const float* d_src, const float* de, float* dZ, float* dW,
                 float* db, float* scratch) {
  const int64_t nb = edge_out_bwd_blocks(n_edges);
  const int64_t rows = cdiv(cdiv(n_edges, nb), 64) * 64;
  const int64_t items = (int64_t)H * E + E;
  float* partial = scratch;
  float* summed = scratch + nb * items;
  ProfScope ps(ctx, st, "edge_out_bwd");
  const size_t lds = (size_t)(H * E + 64 * E) * 4;
#define NG_EB(EE)                                                                                  \
  case EE:                                                                                         \
    hipLaunchKernelGGL((edge_out_bwd_kernel<EE>), dim3((unsigned)nb), dim3(256), lds, st, n_edges, \
                       H, rows, Z, W, d_src, de, dZ, partial);                                     \
    break;
  switch (E) { NG_EB(1) NG_EB(2) NG_EB(3) NG_EB(4) NG_EB(5) NG_EB(6) NG_EB(7) NG_EB(8) }
#undef NG_EB
  hipLaunchKernelGGL(sum_partials2_kernel, ew_grid(items), dim3(256), 0, st, partial, (int)nb, items,
                     summed);
  NG_HIP(ctx, hipMemcpyAsync(dW, summed, (size_t)H * E * 4, hipMemcpyDeviceToDevice, st));
  NG_HIP(ctx, hipMemcpyAsync(db, summed + (size_t)H * E, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

// e *= (d_src > 0)   (model.py:261) for the wide output layer, whose GEMM cannot mask rows that carry a bias
__global__ void mask_rows_kernel(int64_t n_edges, int E, const float* __restrict__ d_src, const float* __restrict__ x,
                                 float* __restrict__ out) {
  const int64_t total = n_edges * E;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
    out[t] = d_src[t / E] > 0.f ? x[t] : 0.f;
}

int edge_mlp_fwd_layered(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int H, int E, int Le, int act,
                         const float* d_src, const float* d_eff, const float* centers, float gap,
                         const float* const* W, const float* const* b, float* e_out,
                         float* z_save) {
  // scratch: X0 [n,H] + (inference) two ping-pong activation buffers
  const int64_t tile = n_edges * H;
  const size_t need = (size_t)(z_save ? tile : 3 * tile) * 4;
  float* ws = (float*)workspace(ctx, need);
  if (!ws) return NG_ERR_NOMEM;
  float* X0 = ws;
  {
    ProfScope ps(ctx, st, "rbf");
    hipLaunchKernelGGL(rbf_kernel, ew_grid(n_edges * (H / 4)), dim3(256), 0, st, n_edges, H, d_src,
                       d_eff, centers, (float)(-1.0 / (double)gap), X0);
    NG_HIP(ctx, hipGetLastError());
  }
  const float* x = X0;
  for (int t = 0; t < Le - 1; ++t) {
    float* y = z_save ? z_save + (int64_t)t * tile : ws + (int64_t)(1 + (t & 1)) * tile;
    int rc = dense_fwd(ctx, st, n_edges, H, H, act, x, W[t], b[t], nullptr, nullptr, y,
                       nullptr, "edge_dense_fwd");
    if (rc) return rc;
    x = y;
  }
  if (E > MAX_E) {   // edge_feature_size = 64 (model.py:23): plain GEMM, then the row mask
    int rc = dense_fwd(ctx, st, n_edges, H, E, NG_ACT_NONE, x, W[Le - 1], b[Le - 1], nullptr, nullptr, e_out, nullptr,
                       "edge_out_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(mask_rows_kernel, ew_grid(n_edges * E), dim3(256), 0, st, n_edges, E, d_src, e_out, e_out);
    NG_HIP(ctx, hipGetLastError());
    return NG_OK;
  }
  return edge_out_fwd(ctx, st, n_edges, H, E, x, W[Le - 1], b[Le - 1], d_src, e_out);
}

int edge_mlp_bwd_layered(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int H, int E, int Le, int act,
                         const float* d_src, const float* d_eff, const float* centers, float gap,
                         const float* const* W, const float* z_save, const float* de,
                         float* const* dW, float* const* db) {
  const int64_t tile = n_edges * H;
  const int64_t nb = edge_out_bwd_blocks(n_edges);
  const bool wide = E > MAX_E;
  const size_t out_scr = wide ? dense_dw_scratch_floats(ctx, n_edges, H, E, true) + (size_t)n_edges * E
                              : (size_t)(nb + 1) * ((size_t)H * E + E);
  const size_t dw_scr = dense_dw_scratch_floats(ctx, n_edges, H, H, true);
  const size_t scr = std::max(out_scr, dw_scr);
  // scratch: X0, dZ ping, dZ pong, reduction scratch
  float* ws = (float*)workspace(ctx, (size_t)(3 * tile + scr) * 4);
  if (!ws) return NG_ERR_NOMEM;
  float* X0 = ws;
  float* dz0 = ws + tile;
  float* dz1 = ws + 2 * tile;
  float* scratch = ws + 3 * tile;
  const float* z_last = z_save + (int64_t)(Le - 2) * tile;
  int rc;
  if (wide) {   // dE = m * de ; dWo = Z^T dE, dbo = colsum dE ; dZ = dE Wo^T   (plain GEMMs)
    float* dE = scratch + dense_dw_scratch_floats(ctx, n_edges, H, E, true);
    hipLaunchKernelGGL(mask_rows_kernel, ew_grid(n_edges * E), dim3(256), 0, st, n_edges, E, d_src, de, dE);
    NG_HIP(ctx, hipGetLastError());
    rc = dense_dw(ctx, st, n_edges, H, E, NG_ACT_NONE, z_last, dE, nullptr, nullptr, dW[Le - 1], db[Le - 1], 0, 0, 0,
                  scratch, "edge_out_bwd");
    if (rc) return rc;
    rc = dense_dx(ctx, st, n_edges, H, E, NG_ACT_NONE, dE, nullptr, nullptr, W[Le - 1], nullptr, dz0, "edge_out_bwd");
  } else {
    rc = edge_out_bwd(ctx, st, n_edges, H, E, z_last, W[Le - 1], d_src, de, dz0, dW[Le - 1],
                      db[Le - 1], scratch);
  }
  if (rc) return rc;
  float* dz = dz0;
  float* dz_next = dz1;
  for (int t = Le - 2; t >= 0; --t) {
    const float* s_t = z_save + (int64_t)t * tile;  // softplus output of layer t
    const float* x_in;
    if (t > 0) {
      x_in = z_save + (int64_t)(t - 1) * tile;
    } else {
      ProfScope ps(ctx, st, "rbf");
      hipLaunchKernelGGL(rbf_kernel, ew_grid(n_edges * (H / 4)), dim3(256), 0, st, n_edges, H,
                         d_src, d_eff, centers, (float)(-1.0 / (double)gap), X0);
      NG_HIP(ctx, hipGetLastError());
      x_in = X0;
    }
    rc = dense_dw(ctx, st, n_edges, H, H, act, x_in, dz, s_t, nullptr, dW[t], db[t], 0, 0,
                  0, scratch, "edge_dense_dw");
    if (rc) return rc;
    if (t > 0) {
      rc = dense_dx(ctx, st, n_edges, H, H, act, dz, s_t, nullptr, W[t], nullptr, dz_next,
                    "edge_dense_dx");
      if (rc) return rc;
      std::swap(dz, dz_next);
    }
  }
  return NG_OK;
}

// NG_EDGE_PATH=layered forces the one-launch-per-layer path (A/B measurements, tests)
static bool force_layered() {
  return sw().edge_layered;
}

}  // namespace ng

using namespace ng;

extern "C" int ng_rbf_expand(ng_ctx* ctx, void* stream, int64_t n, int H, const float* d_src,
                             const float* d_eff, const float* centers, float gap, float* out) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, H % 4 == 0 && gap > 0.f, "rbf_expand: H % 4 == 0, gap > 0");
  if (n == 0) return NG_OK;
  ProfScope ps(ctx, (hipStream_t)stream, "rbf");
  hipLaunchKernelGGL(rbf_kernel, ew_grid(n * (H / 4)), dim3(256), 0, (hipStream_t)stream, n, H, d_src,
                     d_eff, centers, (float)(-1.0 / (double)gap), out);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

// the fused kernels (edge_fused*.hip, edge_*_h2.hip) hard-wire softplus hidden layers and edge_feature_size <= 8;
// fc_activation = relu (model.py:35-36) and edge_feature_size = 64 (model.py:23) run the layered path
static bool use_fused(int H, int E, int Le, int act) {
  return act == NG_ACT_SOFTPLUS && edge_fused_supported(H, E, Le) && !force_layered();
}

static int check_edge_shape(ng_ctx* ctx, int H, int E, int Le, int act) {
  NG_REQUIRE(ctx, H % 16 == 0 && H <= 512, "edge_mlp: edge_hidden_size % 16 == 0, <= 512");
  NG_REQUIRE(ctx, E >= 1 && (E <= MAX_E || (E % 4 == 0 && E <= 256)), "edge_mlp: edge_feature_size <= 8, or a multiple of 4 up to 256");
  NG_REQUIRE(ctx, Le >= 2, "edge_mlp: edge_fc_layers >= 2");
  NG_REQUIRE(ctx, act == NG_ACT_SOFTPLUS || act == NG_ACT_RELU || act == NG_ACT_TANH || act == NG_ACT_NONE,
             "edge_mlp: unknown activation code");
  return NG_OK;
}

extern "C" int ng_edge_mlp_fwd(ng_ctx* ctx, void* stream, int64_t n_edges, int H, int E, int Le, int act,
                               const float* d_src, const float* d_eff, const float* centers,
                               float gap, const float* const* W, const float* const* b,
                               float* e_out, float* z_save) {
  if (!ctx) return NG_ERR_INVALID;
  if (int rc = check_edge_shape(ctx, H, E, Le, act)) return rc;
  NG_REQUIRE(ctx, gap > 0.f, "edge_mlp: rbf gap > 0");
  if (n_edges == 0) return NG_OK;
  if (use_fused(H, E, Le, act))
    return edge_fused_fwd(ctx, (hipStream_t)stream, n_edges, E, d_src, d_eff, centers, gap, W, b,
                          e_out, z_save);
  return edge_mlp_fwd_layered(ctx, (hipStream_t)stream, n_edges, H, E, Le, act, d_src, d_eff, centers,
                              gap, W, b, e_out, z_save);
}

// ---- the same calls over the live-edge view of a padded list (ng_build_live_edges) --------------------------------
extern "C" int ng_edge_live_supported(int H, int E, int Le, int act) { return use_fused(H, E, Le, act) ? 1 : 0; }

static int check_live(ng_ctx* ctx, int64_t n_slots, int H, int E, int Le, int act, const int32_t* perm, const int32_t* n_live) {
  NG_REQUIRE(ctx, perm && n_live, "edge_mlp live view: perm and n_live required");
  NG_REQUIRE(ctx, use_fused(H, E, Le, act), "edge_mlp live view: only the fused edge path has it (ng_edge_live_supported)");
  // slots are addressed through 32-bit byte offsets of the de array
  NG_REQUIRE(ctx, n_slots > 0 && n_slots * (int64_t)E * 4 < ((int64_t)1 << 31), "edge_mlp live view: n_slots * E * 4 must stay below 2^31");
  return NG_OK;
}

extern "C" int ng_edge_mlp_fwd_live(ng_ctx* ctx, void* stream, int64_t n_slots, int H, int E, int Le, int act,
                                    const float* d_src_c, const float* d_eff_c, const int32_t* perm, const int32_t* n_live,
                                    const float* centers, float gap, const float* const* W, const float* const* b,
                                    float* e_out, float* z_save) {
  if (!ctx) return NG_ERR_INVALID;
  if (int rc = check_edge_shape(ctx, H, E, Le, act)) return rc;
  NG_REQUIRE(ctx, gap > 0.f, "edge_mlp: rbf gap > 0");
  if (n_slots == 0) return NG_OK;
  if (int rc = check_live(ctx, n_slots, H, E, Le, act, perm, n_live)) return rc;
  return edge_fused_fwd(ctx, (hipStream_t)stream, n_slots, E, d_src_c, d_eff_c, centers, gap, W, b, e_out, z_save,
                        LiveEdges{perm, n_live});
}

extern "C" int ng_edge_mlp_bwd_live(ng_ctx* ctx, void* stream, int64_t n_slots, int H, int E, int Le, int act,
                                    const float* d_src_c, const float* d_eff_c, const int32_t* perm, const int32_t* n_live,
                                    const float* centers, float gap, const float* const* W, const float* z_save,
                                    const float* de, float* const* dW, float* const* db, int tape_layout) {
  if (!ctx) return NG_ERR_INVALID;
  if (int rc = check_edge_shape(ctx, H, E, Le, act)) return rc;
  NG_REQUIRE(ctx, z_save, "edge_mlp_bwd: saved activations required");
  NG_REQUIRE(ctx, n_slots > 0, "edge_mlp_bwd live view: n_slots > 0");
  if (int rc = check_live(ctx, n_slots, H, E, Le, act, perm, n_live)) return rc;
  return edge_fused_bwd(ctx, (hipStream_t)stream, n_slots, E, d_src_c, d_eff_c, centers, gap, W, z_save, de, dW, db,
                        tape_layout, LiveEdges{perm, n_live});
}

extern "C" int ng_edge_tape_layout(int H, int E, int Le, int act, int64_t n_edges) {
  return use_fused(H, E, Le, act) && edge_tape_blocked(E, n_edges) ? 1 : 0;
}

extern "C" int ng_edge_mlp_bwd_tape(ng_ctx* ctx, void* stream, int64_t n_edges, int H, int E, int Le, int act,
                                    const float* d_src, const float* d_eff, const float* centers,
                                    float gap, const float* const* W, const float* z_save,
                                    const float* de, float* const* dW, float* const* db, int tape_layout) {
  if (!ctx) return NG_ERR_INVALID;
  if (int rc = check_edge_shape(ctx, H, E, Le, act)) return rc;
  NG_REQUIRE(ctx, z_save, "edge_mlp_bwd: saved activations required");
  hipStream_t st = (hipStream_t)stream;
  if (n_edges == 0) {
    for (int t = 0; t < Le; ++t) {
      const int kout = t < Le - 1 ? H : E;
      NG_HIP(ctx, hipMemsetAsync(dW[t], 0, (size_t)H * kout * 4, st));
      NG_HIP(ctx, hipMemsetAsync(db[t], 0, (size_t)kout * 4, st));
    }
    return NG_OK;
  }
  if (use_fused(H, E, Le, act))
    return edge_fused_bwd(ctx, st, n_edges, E, d_src, d_eff, centers, gap, W, z_save, de, dW, db, tape_layout);
  NG_REQUIRE(ctx, tape_layout != 1, "edge_mlp_bwd: a blocked tape needs the fused edge path");
  return edge_mlp_bwd_layered(ctx, st, n_edges, H, E, Le, act, d_src, d_eff, centers, gap, W, z_save, de,
                              dW, db);
}

extern "C" int ng_edge_mlp_bwd(ng_ctx* ctx, void* stream, int64_t n_edges, int H, int E, int Le, int act,
                               const float* d_src, const float* d_eff, const float* centers,
                               float gap, const float* const* W, const float* z_save,
                               const float* de, float* const* dW, float* const* db) {
  return ng_edge_mlp_bwd_tape(ctx, stream, n_edges, H, E, Le, act, d_src, d_eff, centers, gap, W, z_save, de, dW, db, -1);
}
