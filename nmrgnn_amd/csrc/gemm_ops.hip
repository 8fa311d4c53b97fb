// Dense (fully-connected) forward / backward on the fp32 MFMA tile GEMM.
// Replaces keras Dense as used by nmrgnn/model.py:111-138 (EdgeFCBlock), 179-196 (FCBlock) and the
// [N,F*E]x[F*E,F] contraction of nmrgnn/layers.py:39-40 (MPLayer, aggregate-then-GEMM order).
#include <algorithm>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "reduce.cuh"

namespace ng {

// ---------------------------------------------------------------- epilogues
struct EpiDense {
  float* Y;
  float* S;
  const float* bias;
  const float* rowscale;
  const float* R;
  int N;
  int act;
  __device__ __forceinline__ void operator()(int64_t m, int n, float4 v, int) const {
    if (rowscale) {
      const float s = rowscale[m];
      v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    }
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (act != NG_ACT_NONE) {
      v.x = act_apply(act, v.x); v.y = act_apply(act, v.y);
      v.z = act_apply(act, v.z); v.w = act_apply(act, v.w);
    }
    const int64_t o = m * N + n;
    if (S) *reinterpret_cast<float4*>(S + o) = v;
    if (R) {
      const float4 r = *reinterpret_cast<const float4*>(R + o);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    *reinterpret_cast<float4*>(Y + o) = v;
  }
};

struct EpiAdd {
  float* out;
  const float* add;
  int N;
  __device__ __forceinline__ void operator()(int64_t m, int n, float4 v, int) const {
    const int64_t o = m * N + n;
    if (add) {
      const float4 r = *reinterpret_cast<const float4*>(add + o);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    *reinterpret_cast<float4*>(out + o) = v;
  }
};

struct EpiPartial {
  float* partial;  // [nz][Mo][No]
  int64_t Mo;
  int No;
  __device__ __forceinline__ void operator()(int64_t m, int n, float4 v, int z) const {
    *reinterpret_cast<float4*>(partial + ((int64_t)z * Mo + m) * No + n) = v;
  }
};

// ---------------------------------------------------------------- partial reductions
// out[map(idx)] = sum_z partial[z][idx]
__global__ void reduce_partials_kernel(const float* __restrict__ partial, int nz, int64_t n_elem,
                                       float* __restrict__ out, int w_map, int F, int E,
                                       int Nout) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_elem;
       idx += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += partial[(int64_t)z * n_elem + idx];
    int64_t o = idx;
    if (w_map == 1) {  // idx = k*Nout + m, k = ne*F + l  ->  (l*F + m)*E + ne
      const int k = (int)(idx / Nout), m = (int)(idx % Nout);
      const int ne = k / F, l = k % F;
      o = ((int64_t)l * F + m) * E + ne;
    }
    out[o] = s;
  }
}

// column sums of dP (bias gradient), stage 1: partial[blk][n]
template <class LoadP>
__global__ __launch_bounds__(256) void colsum_kernel(int64_t M, int N, int64_t rows_per_block,
                                                     LoadP lp, float* __restrict__ partial) {
  __shared__ float4 red[256];
  const int c4n = N / 4;
  const int rl_n = 256 / c4n;  // row lanes
  const int tid = threadIdx.x;
  const int c4 = tid % c4n, rl = tid / c4n;
  float4 s = f4zero();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  if (rl < rl_n) {
    for (int64_t r = r0 + rl; r < r1; r += rl_n) {
      const float4 v = lp(r, c4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[tid] = s;
  __syncthreads();
  if (tid < c4n) {
    float4 t = red[tid];
    for (int j = 1; j < rl_n; ++j) {
      const float4 v = red[tid + j * c4n];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * N + tid * 4) = t;
  }
}

// ---------------------------------------------------------------- launch helpers
template <int BM, int BN, int WM, int WN, bool QKC, bool PKC, class LQ, class LP, class EP>
static void launch_gemm(hipStream_t st, int64_t M, int N, int64_t K, int64_t k_chunk, int nz,
                        LQ lq, LP lp, EP ep, RangeGuard guard = RangeGuard{nullptr, 0}) {
  dim3 grid((unsigned)cdiv(M, BM), (unsigned)cdiv(N, BN), (unsigned)nz);
  if (guard.word) {      // range fallback: a small grid that walks the tiles only if the guard was raised (mfma_gemm.cuh)
    const uint64_t ntile = (uint64_t)grid.x * grid.y * grid.z;
    hipLaunchKernelGGL((gemm_kernel<BM, BN, 32, WM, WN, QKC, PKC, LQ, LP, EP>), dim3((unsigned)std::min<uint64_t>(ntile, 512)), dim3(256), 0,
                       st, M, N, K, k_chunk, lq, lp, ep, guard.word, guard.epoch, grid.x, grid.y, grid.z);
    return;
  }
  hipLaunchKernelGGL((gemm_kernel<BM, BN, 32, WM, WN, QKC, PKC, LQ, LP, EP>), grid, dim3(256), 0,
                     st, M, N, K, k_chunk, lq, lp, ep, guard.word, guard.epoch, 0u, 0u, 0u);
}

int dense_fwd(ng_ctx* ctx, hipStream_t st, int64_t M, int Kin, int Nout, int act, const float* X,
              const float* W, const float* b, const float* rowscale, const float* R, float* Y,
              float* S, const char* tag) {
  NG_REQUIRE(ctx, Kin % 8 == 0 && Nout % 4 == 0, "dense_fwd: Kin%8, Nout%4");
  if (M == 0) return NG_OK;
  // Split-operand GEMM first where the shape allows; the f32-input GEMM below then runs as its range fallback: its
  // workgroups return at once unless the first kernel raised the guard (an operand beyond the fp16 range).  Not when
  // the output aliases an input (the first kernel has already overwritten it).
  RangeGuard guard{nullptr, 0};
  if (gemm_h2_fwd_ok(M, Kin, Nout) && Y != X && Y != R && S != X) {
    guard = range_guard_begin(ctx);
    if (!guard.word) return NG_ERR_NOMEM;
    const int rc = gemm_h2_fwd(ctx, st, M, Kin, Nout, act, X, W, b, rowscale, R, Y, S, tag, guard);
    if (rc) return rc;
  }
  ProfScope ps(ctx, st, guard.word ? "gemm_range_fallback" : tag);
  LoadPlain lq{X, M, Kin, Kin};
  LoadPlain lp{W, Kin, Nout, Nout};
  EpiDense ep{Y, S, b, rowscale, R, Nout, act};
  // molecule-sized calls (one 2770-atom frame at the default width: 44 tiles of 128 x 128 on 256 CUs, 75 us): 64 x 64 tiles
  if (M * (int64_t)Nout <= (int64_t)128 * 128 * ctx->num_cu)
    launch_gemm<64, 64, 2, 2, true, false>(st, M, Nout, Kin, Kin, 1, lq, lp, ep, guard);
  else if (Nout > 64)
    launch_gemm<128, 128, 2, 2, true, false>(st, M, Nout, Kin, Kin, 1, lq, lp, ep, guard);
  else
    launch_gemm<128, 64, 4, 1, true, false>(st, M, Nout, Kin, Kin, 1, lq, lp, ep, guard);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

int dense_dx(ng_ctx* ctx, hipStream_t st, int64_t M, int Kin, int Nout, int act, const float* dY,
             const float* S, const float* rowscale, const float* W, const float* add, float* dX,
             const char* tag, const float* gscale) {
  NG_REQUIRE(ctx, Nout % 8 == 0 && Kin % 4 == 0, "dense_dx: Nout%8, Kin%4");
  if (M == 0) return NG_OK;
  RangeGuard guard{nullptr, 0};
  if (gemm_h2_fwd_ok(M, Nout, Kin) && dX != dY && dX != add) {      // (see dense_fwd)
    guard = range_guard_begin(ctx);
    if (!guard.word) return NG_ERR_NOMEM;
    const int rc = gemm_h2_dx(ctx, st, M, Kin, Nout, act, dY, S, rowscale, W, add, dX, gscale, tag, guard);
    if (rc) return rc;
  }
  ProfScope ps(ctx, st, guard.word ? "gemm_range_fallback" : tag);
  LoadGradAct lq{dY, act == NG_ACT_NONE ? nullptr : S, rowscale, M, Nout, act};
  LoadPlain lp{W, Kin, Nout, Nout};  // [k_out][n]: K-contiguous along the contraction n
  EpiAdd ep{dX, add, Kin};
  if (Kin > 64)
    launch_gemm<128, 128, 2, 2, true, true>(st, M, Kin, Nout, Nout, 1, lq, lp, ep, guard);
  else
    launch_gemm<128, 64, 4, 1, true, true>(st, M, Kin, Nout, Nout, 1, lq, lp, ep, guard);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

bool dense_grad_uses_h2(int64_t M, int Kin, int Nout) {
  return M > 0 && gemm_h2_fwd_ok(M, Nout, Kin) && gemm_h2_dw_ok(M, Kin, Nout);
}

struct DwPlan {
  int64_t nz, k_chunk, cs_blocks, cs_rows;
  bool big_n;
};

static DwPlan dw_plan(ng_ctx* ctx, int64_t M, int Kin, int Nout, bool has_db) {
  DwPlan p;
  p.big_n = Nout > 64;
  const bool t256 = gemm_h2_dw8_ok(M, Kin, Nout);       // 256 x 256 tiles, one workgroup per CU
  const int BMo = t256 ? 256 : 128, BNo = t256 ? 256 : (p.big_n ? 128 : 64);
  const int64_t tiles = cdiv(Kin, BMo) * cdiv(Nout, BNo);
  // split the contraction (rows) so that ~2 workgroups per CU (one for the 256-tiles) are in flight
  // (256-tiles: ONE 8-wave workgroup fits a CU, so the grid must not exceed the CU count — rounding 256 / 3 tiles UP gave
  // 258 workgroups, two of which ran alone in a second round and doubled the kernel's time)
  int64_t nz = t256 ? std::max<int64_t>(ctx->num_cu / tiles, 1) : cdiv((int64_t)2 * ctx->num_cu, tiles);
  // several 256-tiles per row chunk: a multiple of 8 chunks lets the kernel put the tiles of a chunk on one XCD (gemm_h2.hip)
  if (t256 && tiles > 1 && nz >= 16) nz = nz / 8 * 8;
  const int64_t max_z = std::max<int64_t>(cdiv(M, 32), 1);
  nz = std::max<int64_t>(std::min(nz, max_z), 1);
  p.k_chunk = std::max<int64_t>(cdiv(cdiv(M, nz), 32) * 32, 32);
  p.nz = std::max<int64_t>(cdiv(M, p.k_chunk), 1);
  // keep the multiple of 8 (a trailing chunk may then be empty: it writes a zero partial)
  if (t256 && tiles > 1 && nz % 8 == 0 && p.nz <= nz) p.nz = nz;
  p.cs_blocks = has_db ? std::min<int64_t>(std::max<int64_t>(cdiv(M, 512), 1), 2048) : 0;
  p.cs_rows = has_db ? cdiv(M, p.cs_blocks) : 0;
  return p;
}

size_t dense_dw_scratch_floats(ng_ctx* ctx, int64_t M, int Kin, int Nout, bool has_db) {
  const DwPlan p = dw_plan(ctx, M, Kin, Nout, has_db);
  return (size_t)(p.nz * (int64_t)Kin * Nout + p.cs_blocks * Nout);
}

int dense_dw(ng_ctx* ctx, hipStream_t st, int64_t M, int Kin, int Nout, int act, const float* X,
             const float* dY, const float* S, const float* rowscale, float* dW, float* db,
             int w_map, int F, int E, float* scratch, const char* tag, const float* gscale) {
  if (act == NG_ACT_NONE) S = nullptr;
  NG_REQUIRE(ctx, Kin % 4 == 0 && Nout % 4 == 0, "dense_dw: Kin%4, Nout%4");
  const int64_t n_elem = (int64_t)Kin * Nout;
  if (M == 0) {
    NG_HIP(ctx, hipMemsetAsync(dW, 0, n_elem * sizeof(float), st));
    if (db) NG_HIP(ctx, hipMemsetAsync(db, 0, Nout * sizeof(float), st));
    return NG_OK;
  }
  const DwPlan p = dw_plan(ctx, M, Kin, Nout, db != nullptr);
  float* partial = scratch;
  float* cs_partial = scratch + p.nz * n_elem;
  // deferred reductions (reduce.cuh): the split-K partials wait in the reduction arena for the flush
  if (float* dp = deferred_partials(ctx, (size_t)p.nz * n_elem)) partial = dp;
  RangeGuard guard{nullptr, 0};
  if (gemm_h2_dw_ok(M, Kin, Nout)) {
    guard = range_guard_begin(ctx);
    if (!guard.word) return NG_ERR_NOMEM;
    int rc = gemm_h2_dw(ctx, st, M, Kin, Nout, act, X, dY, S, rowscale, partial, (int)p.nz, p.k_chunk, gscale, tag, guard);
    if (rc) return rc;
  }
  {   // the f32-input GEMM: the product itself, or the range fallback of the split-operand one (same partial layout)
    ProfScope ps(ctx, st, guard.word ? "gemm_range_fallback" : tag);
    LoadPlain lq{X, M, Kin, Kin};
    LoadGradAct lp{dY, S, rowscale, M, Nout, act};
    EpiPartial ep{partial, Kin, Nout};
    if (p.big_n)
      launch_gemm<128, 128, 2, 2, false, false>(st, Kin, Nout, M, p.k_chunk, (int)p.nz, lq, lp, ep, guard);
    else
      launch_gemm<128, 64, 4, 1, false, false>(st, Kin, Nout, M, p.k_chunk, (int)p.nz, lq, lp, ep, guard);
    NG_HIP(ctx, hipGetLastError());
  }
  {
    ProfScope ps(ctx, st, "reduce_partials");
    const int rc = reduce_or_defer(ctx, st, partial, (int)p.nz, n_elem, dW, w_map, F, E, Nout);
    if (rc) return rc;
  }
  if (db) {
    NG_REQUIRE(ctx, Nout / 4 <= 256, "dense_dw: Nout <= 1024 for the bias gradient");
    ProfScope ps(ctx, st, "bias_grad");
    LoadGradAct lp{dY, S, rowscale, M, Nout, act};
    hipLaunchKernelGGL((colsum_kernel<LoadGradAct>), dim3((unsigned)p.cs_blocks), dim3(256), 0, st,
                       M, Nout, p.cs_rows, lp, cs_partial);
    launch_reduce_z(st, cs_partial, (int)p.cs_blocks, (int64_t)Nout, db);
    NG_HIP(ctx, hipGetLastError());
  }
  return NG_OK;
}

}  // namespace ng

namespace ng {
// shapes the register-resident tall-skinny kernels cover (tall_gemm.hip); NG_DENSE_PATH=generic disables
bool tall_dense_ok(int k_in, int n_out) {
  if (sw().dense_generic) return false;
  if (k_in % 4 || n_out % 4 || k_in > 192 || n_out > 192) return false;
  const int kpad = (k_in + 63) / 64 * 64, npad = (n_out + 63) / 64 * 64;
  return tall_gemm_supported(kpad, npad);
}
}  // namespace ng

// ------------------------------------------------------------------- C ABI
extern "C" int ng_dense_fwd(ng_ctx* ctx, void* stream, int64_t M, int Kin, int Nout, int act,
                            int residual, const float* X, const float* W, const float* b, float* Y,
                            float* s_save) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, !residual || Kin == Nout, "ng_dense_fwd: residual needs Kin == Nout");
  hipStream_t st = (hipStream_t)stream;
  if (ng::tall_dense_ok(Kin, Nout)) {   // 64-feature model: weights resident in registers
    const int kpad = (Kin + 63) / 64 * 64, npad = (Nout + 63) / 64 * 64;
    float* Wfrag = (float*)ng::workspace(ctx, (size_t)kpad * npad * 4);
    if (!Wfrag) return NG_ERR_NOMEM;
    int rc = ng::tall_pack(ctx, st, Kin, Nout, kpad, npad, Nout, 1, W, Wfrag);
    if (rc) return rc;
    ng::TallArgs a{};
    a.N = M; a.X = X; a.ldx = Kin; a.k_valid = Kin; a.Wfrag = Wfrag; a.bias = b; a.act = act;
    a.S_save = s_save; a.resid = residual ? X : nullptr; a.out = Y; a.ldo = Nout; a.n_valid = Nout;
    return ng::tall_gemm(ctx, st, kpad, npad, a, false, "dense_fwd");
  }
  return ng::dense_fwd(ctx, st, M, Kin, Nout, act, X, W, b, nullptr, residual ? X : nullptr, Y, s_save);
}

extern "C" int ng_dense_bwd(ng_ctx* ctx, void* stream, int64_t M, int Kin, int Nout, int act,
                            int residual, const float* X, const float* W, const float* s_save,
                            const float* dY, float* dX, float* dW, float* db) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, !residual || Kin == Nout, "ng_dense_bwd: residual needs Kin == Nout");
  NG_REQUIRE(ctx, act == NG_ACT_NONE || s_save, "ng_dense_bwd: activation backward needs s_save");
  const float* S = s_save;
  hipStream_t st = (hipStream_t)stream;
  if (dX && ng::tall_dense_ok(Nout, Kin) && Nout <= 128) {
    // dX[m][k] = (dY) + sum_n dP[m][n] W[k][n]: contraction over n, W(kk = n, o = k) = w[k*Nout + n]
    const int kpad = (Nout + 63) / 64 * 64, npad = (Kin + 63) / 64 * 64;
    float* Wfrag = (float*)ng::workspace(ctx, (size_t)kpad * npad * 4);
    if (!Wfrag) return NG_ERR_NOMEM;
    int rc = ng::tall_pack(ctx, st, Nout, Kin, kpad, npad, 1, Nout, W, Wfrag);
    if (rc) return rc;
    ng::TallArgs a{};
    a.N = M; a.X = dY; a.ldx = Nout; a.k_valid = Nout;
    a.S_in = act == NG_ACT_NONE ? nullptr : S; a.act_in = act;
    a.Wfrag = Wfrag; a.act = NG_ACT_NONE; a.resid = residual ? dY : nullptr; a.out = dX; a.ldo = Kin;
    a.n_valid = Kin;
    rc = ng::tall_gemm(ctx, st, kpad, npad, a, true, "dense_dx");
    if (rc) return rc;
  } else if (dX) {
    int rc = ng::dense_dx(ctx, st, M, Kin, Nout, act, dY, S, nullptr, W, residual ? dY : nullptr, dX);
    if (rc) return rc;
  }
  if (M > 0 && ng::tall_tn_supported(Kin, Nout)) {   // 64-feature model: persistent register accumulators
    float* scr = (float*)ng::workspace(ctx, ng::tall_tn_scratch_floats(ctx, Kin) * sizeof(float));
    if (!scr) return NG_ERR_NOMEM;
    return ng::tall_tn(ctx, st, M, X, Kin, Kin, dY, Nout, Nout, act == NG_ACT_NONE ? nullptr : S, act, dW,
                       db, 0, 0, 0, scr, "dense_dw");
  }
  float* scratch = (float*)ng::workspace(
      ctx, ng::dense_dw_scratch_floats(ctx, M, Kin, Nout, db != nullptr) * sizeof(float));
  if (!scratch) return NG_ERR_NOMEM;
  return ng::dense_dw(ctx, st, M, Kin, Nout, act, X, dY, S, nullptr, dW, db, 0, 0, 0, scratch);
}
