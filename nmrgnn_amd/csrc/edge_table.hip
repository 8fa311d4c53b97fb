// Edge path through a table of the edge function (round 5, OPT-IN: Engine(edge_table=True) / NG_EDGE_TABLE=1).
//
// mask + RBFExpansion + EdgeFCBlock (nmrgnn/model.py:251-261, layers.py:137-140, model.py:132-138) is a function of ONE
// scalar per edge: e_ij = m_ij * f_W(d_ij), f_W: R -> R^E smooth (Gaussians of width sqrt(gap / 2) = 0.028 through softplus
// layers).  The fused edge kernels evaluate f_W 2.1 million times per step of the bench batch — 91 % of the forward flops.
// Here f_W is evaluated with THE SAME KERNELS on T = 4096 equidistant points covering the step's distances and every edge
// interpolates (four-point cubic Lagrange): with h = (hi - lo) / (T - 3) ~ 1.5e-4 the interpolation error is
// ~ (h / 0.028)^4 |f| < 1e-9 |f|, below the fp32 rounding of e itself.  The backward is the exact adjoint: every edge adds
// w_k * m * de_ij to the four table points it read, and the unchanged fused backward runs on the table — weight gradients of
// the interpolated function, equal to the per-edge ones to the same 1e-9.
//
// Not what the headline numbers are measured on: the reference evaluates the MLP per edge, and the judged kernels are the
// per-edge ones (bench.py forces them for `value` and prints this path beside it).  It is the Engine's DEFAULT since round 6,
// behind an a-posteriori guard: the edge function is ALSO evaluated at the T midpoints between the table points (same launch:
// 2T rows), ng_edge_table_check compares it there — where the error of the cubic interpolant of a cell peaks — with what the
// table gives, and above the bound (or when a call's distances leave a cached table's range) the SAME call is answered by the
// per-edge kernels: they run over `gate[1]` rows (the live-edge row count when the guard is up, ZERO otherwise — a device
// scalar, like n_live itself), and the interpolation / the table's backward skip themselves.  No host round trip either way.
//
// Determinism: the scatter accumulates in 64-bit fixed point (LDS atomics per workgroup, integer sums over the workgroups) —
// integer addition is associative, so the bits do not depend on the order the edges arrive in.
#include <algorithm>
#include "ng_internal.h"

namespace ng {

constexpr int ET_BLOCK = 256;
constexpr int ET_EMAX = 8;         // edge features a table carries (model.py:23 offers 1, 2, 3, 8, 64: the last one stays per edge); E > 4 with T <= 2048 (LDS)
constexpr int ET_WIDE = 1024;      // the per-edge passes that keep the table in LDS: one workgroup per CU (one table copy), sixteen waves (round 6: 256 threads and
                                   // one copy per workgroup of four waves left a CU with four waves walking 32 dependent round trips each: scatter 57 us)

// lo, hi of d_eff over the live slots (d_src > 0) and max |de| (de may be null): out = {lo, hi, maxabs}.  Two stages.
// pos (nullable): d_eff is compacted, slot i's distance is d_eff[pos[i]] (ng_build_live_edges)
__global__ __launch_bounds__(ET_BLOCK) void et_range_kernel(int64_t n, int E, const float* __restrict__ d_src,
                                                          const float* __restrict__ d_eff, const int32_t* __restrict__ pos,
                                                          const float* __restrict__ de, float* __restrict__ part) {
  __shared__ float s[3][ET_BLOCK / 64];
  float lo = 3.0e38f, hi = -3.0e38f, mx = 0.f;
  // four slots per trip, every load of the trip requested before the first is used (dead slots read slot 0's distance / their own
  // gradient row and are masked out of the min / max: one slot per trip was a chain of two or three dependent round trips, 15.8 us
  // for the 33 MB of the gradient pass)
  const int64_t stride = (int64_t)gridDim.x * ET_BLOCK;
  for (int64_t i0 = (int64_t)blockIdx.x * ET_BLOCK + threadIdx.x; i0 < n; i0 += 4 * stride) {
    float ds[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ds[u] = d_src[std::min<int64_t>(i0 + u * stride, n - 1)];
    int32_t pi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = std::min<int64_t>(i0 + u * stride, n - 1);
      pi[u] = (d_eff && pos) ? pos[i] : (int32_t)i;
    }
    float dv[4], g[4][ET_EMAX];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = std::min<int64_t>(i0 + u * stride, n - 1);
      const bool live = ds[u] > 0.f && i0 + u * stride < n;
      dv[u] = d_eff ? d_eff[live ? pi[u] : 0] : 0.f;
#pragma unroll
      for (int c = 0; c < ET_EMAX; ++c) g[u][c] = (de && c < E) ? de[i * E + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool live = ds[u] > 0.f && i0 + u * stride < n;
      if (live) {
        if (d_eff) { lo = fminf(lo, dv[u]); hi = fmaxf(hi, dv[u]); }
#pragma unroll
        for (int c = 0; c < ET_EMAX; ++c) mx = fmaxf(mx, fabsf(g[u][c]));
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = lo; s[1][threadIdx.x >> 6] = hi; s[2][threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < ET_BLOCK / 64; ++w) { lo = fminf(lo, s[0][w]); hi = fmaxf(hi, s[1][w]); mx = fmaxf(mx, s[2][w]); }
    part[3 * blockIdx.x] = lo; part[3 * blockIdx.x + 1] = hi; part[3 * blockIdx.x + 2] = mx;
  }
}
// stage 2 (one block): combine; which of the three outputs are written is chosen by the caller (range pass / de pass)
__global__ __launch_bounds__(ET_BLOCK) void et_range_final_kernel(int nb, const float* __restrict__ part, float* __restrict__ out,
                                                                int write_range, int write_max, float pad) {
  __shared__ float s[3][ET_BLOCK / 64];
  float lo = 3.0e38f, hi = -3.0e38f, mx = 0.f;
  for (int i = threadIdx.x; i < nb; i += ET_BLOCK) { lo = fminf(lo, part[3 * i]); hi = fmaxf(hi, part[3 * i + 1]); mx = fmaxf(mx, part[3 * i + 2]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = lo; s[1][threadIdx.x >> 6] = hi; s[2][threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < ET_BLOCK / 64; ++w) { lo = fminf(lo, s[0][w]); hi = fmaxf(hi, s[1][w]); mx = fmaxf(mx, s[2][w]); }
    if (write_range) {
      if (!(hi >= lo)) { lo = 0.f; hi = 1.f; }            // no live edge at all
      if (!(hi - lo > 1e-6f)) hi = lo + 1e-6f;            // all distances equal: a table of (almost) one point
      const float w = (hi - lo) * pad;                    // a table kept over calls covers more than the call that built it
      out[0] = lo - w; out[1] = hi + w;
    }
    if (write_max) out[2] = mx;
  }
}

// grid geometry from the range: point t sits at lo + (t - 1) h, h = (hi - lo) / (T - 3): every d in [lo, hi] has its
// four-point stencil i-1 .. i+2 inside [0, T-1]
__device__ __forceinline__ void et_geom(const float* __restrict__ range, int T, float& lo, float& inv_h, float& h) {
  lo = range[0];
  h = (range[1] - lo) / (float)(T - 3);
  inv_h = 1.0f / h;
}
// rows [0, T): the table points; rows [T, 2T) (mid != 0): the midpoints x_t + h / 2 the guard compares at; perm: identity
__global__ __launch_bounds__(ET_BLOCK) void et_points_kernel(int T, int mid, const float* __restrict__ range, float* __restrict__ d_tab,
                                                           float* __restrict__ ones, int32_t* __restrict__ perm) {
  const int t = blockIdx.x * ET_BLOCK + threadIdx.x;
  if (t >= (mid ? 2 * T : T)) return;
  float lo, inv_h, h;
  et_geom(range, T, lo, inv_h, h);
  d_tab[t] = t < T ? fmaf((float)(t - 1), h, lo) : fmaf((float)(t - T - 1) + 0.5f, h, lo);
  ones[t] = 1.0f;
  if (perm) perm[t] = t;
}

// The guard.  e_all[2T][E]: the edge function at the table points and at the midpoints.  err = max over the interior
// midpoints and the components of |cubic interpolant(table) - value|, scale = max |table value|.
// gate[0] = bad (err > tol * scale, a value not finite, or — cover != null — the call's distances [cover[0], cover[1]] not inside
// the table's [range[0], range[1]]);  gate[1] = bad ? *n_live : -1 (rows of the per-edge kernels; negative = the launch returns
// at once: forward leaves e_out alone, backward's gradients are zero);  gate[2] = bad ? 0 : rows
// (rows of the table's backward);  gate[4], gate[5] = err, scale as float bits.  prev (nullable): a gate decided earlier (the
// table's own check, when only the range is checked per call): bad stays bad.
constexpr int ET_CHECK_THREADS = 1024;
__global__ __launch_bounds__(ET_CHECK_THREADS) void et_check_kernel(int T, int E, const float* __restrict__ e_all, float tol,
                                                                  const float* __restrict__ range, const float* __restrict__ cover,
                                                                  const int32_t* __restrict__ n_live, int rows,
                                                                  const int32_t* __restrict__ prev, int32_t* __restrict__ gate) {
  extern __shared__ float et_c[];      // e_all staged: one coalesced pass over the 2 T E floats instead of strided dependent reads
  __shared__ float s[2][ET_CHECK_THREADS / 64];
  float err = 0.f, sc = 0.f;
  bool nf = false;
  if (e_all) {
    const int tot = 2 * T * E;
    for (int i = threadIdx.x * 4; i < tot; i += ET_CHECK_THREADS * 4) {
      if (i + 4 <= tot) *reinterpret_cast<float4*>(et_c + i) = *reinterpret_cast<const float4*>(e_all + i);
      else for (int k = i; k < tot; ++k) et_c[k] = e_all[k];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += ET_CHECK_THREADS) {
      for (int c = 0; c < E; ++c) {
        const float v = et_c[t * E + c];
        nf |= !(fabsf(v) < 3.0e38f);
        sc = fmaxf(sc, fabsf(v));
      }
      if (t >= 1 && t + 2 < T) {      // midpoint between table points t and t + 1: stencil t-1 .. t+2, weights -1/16 9/16 9/16 -1/16
        for (int c = 0; c < E; ++c) {
          const float it = 0.5625f * (et_c[t * E + c] + et_c[(t + 1) * E + c]) - 0.0625f * (et_c[(t - 1) * E + c] + et_c[(t + 2) * E + c]);
          const float m = et_c[(T + t) * E + c];
          nf |= !(fabsf(m) < 3.0e38f);
          err = fmaxf(err, fabsf(it - m));
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { err = fmaxf(err, __shfl_xor(err, o)); sc = fmaxf(sc, __shfl_xor(sc, o)); }
  const bool anynf = __syncthreads_or(nf ? 1 : 0) != 0;
  if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = err; s[1][threadIdx.x >> 6] = sc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < ET_CHECK_THREADS / 64; ++w) { err = fmaxf(err, s[0][w]); sc = fmaxf(sc, s[1][w]); }
    bool bad = anynf || !(err <= tol * sc + 1e-30f);
    if (!e_all) bad = false;
    if (cover) bad = bad || !(cover[0] >= range[0] && cover[1] <= range[1]);
    if (prev) bad = bad || prev[0] != 0;
    gate[0] = bad ? 1 : 0;
    gate[1] = bad ? (n_live ? *n_live : 0) : -1;      // (negative: the per-edge launches skip themselves entirely)
    gate[2] = bad ? 0 : rows;
    gate[3] = 0;
    if (e_all) { gate[4] = __builtin_bit_cast(int32_t, err); gate[5] = __builtin_bit_cast(int32_t, sc); }
  }
}

// stencil of distance d: first point i0 = floor(u) - 1 (clamped so that i0 .. i0 + 3 exist) and the Lagrange weights
__device__ __forceinline__ void et_stencil(float d, float lo, float inv_h, int T, int& i0, float (&w)[4]) {
  const float u = (d - lo) * inv_h + 1.0f;
  int i = (int)floorf(u);
  i = min(max(i, 1), T - 3);
  const float f = u - (float)i;              // in [0, 1) up to rounding at the ends
  i0 = i - 1;
  const float fm1 = f - 1.0f, fm2 = f - 2.0f, fp1 = f + 1.0f;
  w[0] = -f * fm1 * fm2 * (1.0f / 6.0f);
  w[1] = fp1 * fm1 * fm2 * 0.5f;
  w[2] = -fp1 * f * fm2 * 0.5f;
  w[3] = fp1 * f * fm1 * (1.0f / 6.0f);
}

// e[i][c] = m_i * sum_k w_k e_tab[i0 + k][c]; the table (T x E floats) sits in LDS
template <int EC>
__global__ __launch_bounds__(ET_WIDE) void et_interp_kernel(int64_t n, int T, const float* __restrict__ d_src,
                                                           const float* __restrict__ d_eff, const int32_t* __restrict__ pos,
                                                           const float* __restrict__ range, const float* __restrict__ e_tab,
                                                           const int32_t* __restrict__ gate, float* __restrict__ e_out) {
  extern __shared__ float et_s[];
  if (gate && gate[0] != 0) return;      // the guard is up: the per-edge kernels have written e
  for (int t = threadIdx.x; t < T * EC; t += ET_WIDE) et_s[t] = e_tab[t];
  __syncthreads();
  float lo, inv_h, h;
  et_geom(range, T, lo, inv_h, h);
  for (int64_t i = (int64_t)blockIdx.x * ET_WIDE + threadIdx.x; i < n; i += (int64_t)gridDim.x * ET_WIDE) {
    float r[EC];
#pragma unroll
    for (int c = 0; c < EC; ++c) r[c] = 0.f;
    if (d_src[i] > 0.f) {
      int i0;
      float w[4];
      et_stencil(d_eff[pos ? pos[i] : i], lo, inv_h, T, i0, w);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < EC; ++c) r[c] = fmaf(w[k], et_s[(i0 + k) * EC + c], r[c]);
    }
#pragma unroll
    for (int c = 0; c < EC; ++c) e_out[i * EC + c] = r[c];
  }
}

// adjoint: table[i0 + k][c] += w_k * m_i * de[i][c], in 64-bit fixed point (scale 2^sh from max |de|: |sum| < 2^62 for up to
// 2^21 terms of size <= 2^40 each), per workgroup in LDS, partial tables to memory
template <int EC>
__global__ __launch_bounds__(ET_WIDE) void et_scatter_kernel(int64_t n, int T, const float* __restrict__ d_src,
                                                            const float* __restrict__ d_eff, const int32_t* __restrict__ pos,
                                                            const float* __restrict__ range, const float* __restrict__ de,
                                                            long long* __restrict__ part) {
  extern __shared__ long long et_q[];
  for (int t = threadIdx.x; t < T * EC; t += ET_WIDE) et_q[t] = 0;
  __syncthreads();
  float lo, inv_h, h;
  et_geom(range, T, lo, inv_h, h);
  const float mx = range[2];
  int ex = 0;
  if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &ex);      // mx <= 2^ex
  const float scale = ldexpf(1.0f, 38 - ex);                // |w de| * scale < 2^39 (|w| < 1.5)
  for (int64_t i = (int64_t)blockIdx.x * ET_WIDE + threadIdx.x; i < n; i += (int64_t)gridDim.x * ET_WIDE) {
    if (d_src[i] > 0.f) {
      int i0;
      float w[4];
      et_stencil(d_eff[pos ? pos[i] : i], lo, inv_h, T, i0, w);
      float g[EC];
#pragma unroll
      for (int c = 0; c < EC; ++c) g[c] = de[i * EC + c] * scale;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < EC; ++c) {
          const long long q = __float2ll_rn(w[k] * g[c]);
          atomicAdd(reinterpret_cast<unsigned long long*>(&et_q[(i0 + k) * EC + c]), (unsigned long long)q);
        }
    }
  }
  __syncthreads();
  long long* p = part + (size_t)blockIdx.x * T * EC;
  for (int t = threadIdx.x; t < T * EC; t += ET_WIDE) p[t] = et_q[t];
}
// 64 table entries per workgroup, four groups of 64 threads each summing a quarter of the workgroups' partial tables (eight
// loads in flight; one thread walking all 256 partials of its entry was a chain of dependent round trips: 62 us).  Integer
// sums: any order gives the same bits.
// entries [TE, TE_out): rows of the table's backward that carry no gradient (the midpoints of the guard): zeros
__global__ __launch_bounds__(ET_BLOCK) void et_scatter_final_kernel(int nb, int TE, int TE_out, const long long* __restrict__ part,
                                                                  const float* __restrict__ range, float* __restrict__ de_tab) {
  __shared__ long long red[ET_BLOCK];
  const int tt = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tt;
  long long s = 0;
  if (t < TE) {
    int b = g;
    for (; b + 28 < nb; b += 32) {
      long long v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 4 * u) * TE + t];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nb; b += 4) s += part[(size_t)b * TE + t];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && t < TE) {
    s = red[tt] + red[64 + tt] + red[128 + tt] + red[192 + tt];
    const float mx = range[2];
    int ex = 0;
    if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &ex);
    de_tab[t] = (float)((double)s * ldexp(1.0, ex - 38));
  }
  if (g == 1 && t >= TE && t < TE_out) de_tab[t] = 0.f;
}
static_assert(ET_BLOCK == 256, "et_scatter_final_kernel: four groups of 64");

static int et_blocks(ng_ctx* ctx, int64_t n) { return (int)std::min<int64_t>(cdiv(n, ET_BLOCK), (int64_t)ctx->num_cu * 4); }

}  // namespace ng

using namespace ng;

extern "C" int ng_edge_table_range(ng_ctx* ctx, void* stream, int64_t n, int E, const float* d_src, const float* d_eff,
                                   const int32_t* pos, const float* de, float pad, float* range) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, d_src && range && (d_eff || de) && pad >= 0.f && pad <= 4.f && (!de || (E >= 1 && E <= ET_EMAX)), "edge_table_range: arguments (E <= 8)");
  hipStream_t st = (hipStream_t)stream;
  const int nb = std::max(1, et_blocks(ctx, n));
  float* part = (float*)aux_workspace(ctx, (size_t)nb * 3 * 4);
  if (!part) return NG_ERR_NOMEM;
  ProfScope ps(ctx, st, "edge_table_range");
  hipLaunchKernelGGL(et_range_kernel, dim3(nb), dim3(ET_BLOCK), 0, st, n, E, d_src, d_eff, pos, de, part);
  hipLaunchKernelGGL(et_range_final_kernel, dim3(1), dim3(ET_BLOCK), 0, st, nb, part, range, d_eff ? 1 : 0, de ? 1 : 0, pad);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_edge_table_points(ng_ctx* ctx, void* stream, int T, int midpoints, const float* range, float* d_tab, float* ones,
                                    int32_t* perm) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, T >= 8 && range && d_tab && ones, "edge_table_points: arguments");
  const int rows = midpoints ? 2 * T : T;
  hipLaunchKernelGGL(et_points_kernel, dim3((unsigned)cdiv(rows, ET_BLOCK)), dim3(ET_BLOCK), 0, (hipStream_t)stream, T, midpoints, range,
                     d_tab, ones, perm);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_edge_table_check(ng_ctx* ctx, void* stream, int T, int E, const float* e_all, float tol, const float* range,
                                   const float* cover, const int32_t* n_live, int rows, const int32_t* prev, int32_t* gate) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, T >= 8 && E >= 1 && E <= ET_EMAX && gate && (e_all || cover) && (!cover || range) && (size_t)2 * T * E * 4 <= 128 * 1024,
             "edge_table_check: arguments (table + midpoints <= 128 KB)");
  ProfScope ps(ctx, (hipStream_t)stream, "edge_table_check");
  hipLaunchKernelGGL(et_check_kernel, dim3(1), dim3(ET_CHECK_THREADS), e_all ? (size_t)2 * T * E * 4 : 0, (hipStream_t)stream, T, E, e_all,
                     tol, range, cover, n_live, rows, prev, gate);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_edge_table_interp(ng_ctx* ctx, void* stream, int64_t n, int E, int T, const float* d_src, const float* d_eff,
                                    const int32_t* pos, const float* range, const float* e_tab, const int32_t* gate, float* e_out) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, E >= 1 && E <= ET_EMAX && T >= 8 && (size_t)T * E * 4 <= 64 * 1024, "edge_table_interp: E <= 8, table <= 64 KB");
  if (n == 0) return NG_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nb = std::max(1, (int)std::min<int64_t>(cdiv(n, ET_WIDE), (int64_t)ctx->num_cu));
  const size_t lds = (size_t)T * E * 4;
  ProfScope ps(ctx, st, "edge_table_interp");
  switch (E) {
#define NG_ETI(EC) case EC: hipLaunchKernelGGL((et_interp_kernel<EC>), dim3(nb), dim3(ET_WIDE), lds, st, n, T, d_src, d_eff, pos, range, e_tab, gate, e_out); break;
    NG_ETI(1) NG_ETI(2) NG_ETI(3) NG_ETI(4) NG_ETI(5) NG_ETI(6) NG_ETI(7) NG_ETI(8)
#undef NG_ETI
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_edge_table_scatter(ng_ctx* ctx, void* stream, int64_t n, int E, int T, int rows_out, const float* d_src,
                                     const float* d_eff, const int32_t* pos, float* range, const float* de, float* de_tab) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, E >= 1 && E <= ET_EMAX && T >= 8 && (size_t)T * E * 8 <= 128 * 1024 && rows_out >= T,
             "edge_table_scatter: E <= 8, table <= 128 KB, rows_out >= T");
  hipStream_t st = (hipStream_t)stream;
  // max |de| over the live slots -> range[2]
  if (int rc = ng_edge_table_range(ctx, stream, n, E, d_src, nullptr, nullptr, de, 0.f, range)) return rc;
  const int nb = std::max(1, (int)std::min<int64_t>(cdiv(n, (int64_t)ET_WIDE * 2), (int64_t)ctx->num_cu));
  long long* part = (long long*)workspace(ctx, (size_t)nb * T * E * 8);
  if (!part) return NG_ERR_NOMEM;
  const size_t lds = (size_t)T * E * 8;
  ProfScope ps(ctx, st, "edge_table_scatter");
  switch (E) {
#define NG_ETS(EC) case EC: hipLaunchKernelGGL((et_scatter_kernel<EC>), dim3(nb), dim3(ET_WIDE), lds, st, n, T, d_src, d_eff, pos, range, de, part); break;
    NG_ETS(1) NG_ETS(2) NG_ETS(3) NG_ETS(4) NG_ETS(5) NG_ETS(6) NG_ETS(7) NG_ETS(8)
#undef NG_ETS
  }
  hipLaunchKernelGGL(et_scatter_final_kernel, dim3((unsigned)cdiv(rows_out * E, 64)), dim3(ET_BLOCK), 0, st, nb, T * E, rows_out * E, part,
                     range, de_tab);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}
