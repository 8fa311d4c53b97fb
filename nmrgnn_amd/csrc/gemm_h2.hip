// Generic dense products on the fp16 matrix pipe with two-piece split fp32 operands (h2_common.cuh: x = h + l,
// three piece products per multiply, fp32 accumulate):
//   forward   Y = act(rowscale * (X W) + b) (+ R),   X [M][K], W [K][N] row-major fp32,   K % 32 == 0, N % 128 == 0
//   dX = dP W^T (+ add),  dW = X^T dP   with  dP = dY * act'(S) * rowscale  formed by the operand loaders.
// They serve dense_fwd / dense_dx / dense_dw for the matrix-bound shapes of the reference's default width
// (atom_feature_size = 256: the MPLayer update [N, 768] x [768, 256] of nmrgnn/layers.py:39-40, its dA / dw products and
// the FCBlock layers of model.py:191-196).  NG_GEMM_MATH=fp32 opts out (f32-input MFMA tile GEMM, gemm_ops.hip).
// Until late round 2 these kernels used the exact three-piece bf16 split (six products); two fp16 pieces halve the
// matrix instructions and the split work per element at the same float64 error (tests/test_gpu_gemm_h2.py).
//
// Ranges (see h2_common.cuh).  Weight pieces are taken from 2^8 W.  Activation operands (X) are split unscaled.
// Gradient operands (dP) are multiplied by a power of two S before the split, S = 2^(14 - e) with 2^e >= max|dY| max|rs|
// (gh_absmax_kernel + gh_scale_kernel, one pass over dY per gradient tensor: gemm_grad_scale; dense_dx and dense_dw of
// the same dP share it), so pieces cannot overflow and keep their bits whatever the loss scaling is; epilogues
// multiply by 2^-8 / S.
//
// 256 threads = 4 waves, tile 128 rows x 256 columns (128 when N % 256 != 0), two workgroups per CU; per 32-wide k-step:
//   W pieces: fragment-ordered image packed once per call, 32 / 16 KB per (column tile, k-step) copied to LDS by LDS-DMA;
//   X pieces: each thread loads 16 consecutive floats of one row (prefetched one step ahead), splits them and writes
//             the two piece planes [128][32 + 8] fp16;
//   wave (n-half, m-half): 4 x 2 (2 x 2) blocks of 32 x 32, v_mfma_f32_32x32x16_f16, A = W^T pieces (rows n), B = X pieces
//   (columns m), so a lane ends with 4 consecutive n of one row m: 16-byte stores.
#include <algorithm>
#include <cstdlib>
#include <string>

#include "edge_fused.h"     // NG_LDS_BARRIER
#include "mfma_gemm.cuh"    // act_apply
#include "ng_internal.h"
#include "h2_common.cuh"
#include "pack_bodies.cuh"
#include "reduce.cuh"

namespace ng {

constexpr int GX_BM = 128, GX_BN = 128, GX_BK = 32;
constexpr int GX_XROW = 80;                       // bytes per row of an X piece plane (64 + 16: conflict-free b128 rows)
constexpr int GX_XPLANE = GX_BM * GX_XROW;        // 10,240
// NBW = 32-column blocks per wave (2 -> 128-column tiles, 4 -> 256-column tiles: half the barriers and X splits per MFMA)
constexpr int gx_wchunk(int nbw) { return 2 * nbw * 2 * 2 * 1024; }   // [n-block][k-step of 16][piece][1 KB]
constexpr int gx_lds(int nbw) { return 2 * GX_XPLANE + gx_wchunk(nbw); } // 36,864 / 53,248: two workgroups per CU
constexpr float GX_WSCALE = 256.0f, GX_WINV = 1.0f / 256.0f;      // weight pieces are taken from 2^8 W

// weight image: pack_bodies.cuh gx_img (PK_GX), [(ct * KT + kt)][nb][ks][p][lane][8 fp16], pieces of 2^8 W
static_assert(GX_BK == pk::GXP_BK && GX_WSCALE == pk::GXP_WSCALE, "pack_bodies.cuh");
static int gx_image_kind(int trans, int nbw) { return 3 + 16 * trans + 32 * nbw; }       // cache kind of the image
static PackJob gx_pack_job(int K, int N, int mode, int nbw, const float* W, void* img) {
  PackJob j;
  j.kind = PK_GX; j.i0 = K; j.i1 = N | (mode << 20) | (nbw << 24); j.src[0] = W; j.dst[0] = img;
  j.blocks = (int)std::min<int64_t>(cdiv((int64_t)(N / 32) * (K / 16) * 64, PKB), 1024);
  return j;
}

struct GxArgs {
  int64_t M;
  int K, N;
  const float* X;
  const char* Wimg;
  const float* bias;
  const float* rowscale;
  const float* R;
  float* Y;
  float* S;
  int act;
  // GRAD prologue: X := X * act'(Sin) * rs_in[m]   (dP = dY * act'(S) * rowscale of dense_dx)
  const float* Sin;
  const float* rs_in;
  int act_in;
  const float* gscale;   // GRAD: {S, 1/S} of the gradient operand (device memory); nullptr: 1
  RangeGuard guard;      // raised when an accumulator comes out non-finite (an operand beyond the fp16 range)
};

// ---- power-of-two scale of a gradient operand (header: Ranges).  The context's small scratch (64 KB, never moved)
// holds [GH_MAXBLOCKS block maxima of |dY| | GH_RSBLOCKS maxima of |rs| | {S, 1/S}]; everything is ordered on the
// caller's stream.  Producers of a gradient tensor (mp_dp_kernel, fc_dp_kernel) write the block maxima as a by-product
// (gemm_grad_blockmax / gemm_grad_scale_from_blocks); gemm_grad_scale is the stand-alone pass for everything else.
constexpr int GH_MAXBLOCKS = 8192, GH_RSBLOCKS = 256;
static_assert((GH_MAXBLOCKS + GH_RSBLOCKS + 2) * 4 <= NG_SMALL_BYTES, "small scratch layout");

__global__ __launch_bounds__(256) void gh_absmax_kernel(const float4* __restrict__ x, int64_t n4, float* __restrict__ blockmax) {
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {          // four independent loads in flight
    const float4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
    m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                       fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
    m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))),
                       fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
  }
  for (; i < n4; i += stride) {
    const float4 v = x[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  block_max_store(m, blockmax);
}

__global__ __launch_bounds__(256) void gh_rsmax_kernel(const float* __restrict__ rs, int64_t n, float* __restrict__ blockmax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(rs[i]));
  block_max_store(m, blockmax);
}

// S = 2^(14 - e), 2^e > max|dY| * max|rs|: |S dP| < 2^14 leaves a factor 4 for activation derivatives above 1
// (gelu, selu, swish: <= 1.2).  max is exact and order-free: the scale does not depend on the launch geometry.
__global__ __launch_bounds__(256) void gh_scale_kernel(const float* __restrict__ blockmax, int nblocks,
                                                       const float* __restrict__ rsmax, int nrs, float* __restrict__ scale) {
  __shared__ float red[2][256];
  const int t = threadIdx.x;
  float m = 0.f, r = 0.f;
  for (int i = t; i < nblocks; i += 256) m = fmaxf(m, blockmax[i]);
  for (int i = t; i < nrs; i += 256) r = fmaxf(r, rsmax[i]);
  red[0][t] = m; red[1][t] = r;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) { red[0][t] = fmaxf(red[0][t], red[0][t + s]); red[1][t] = fmaxf(red[1][t], red[1][t + s]); }
    __syncthreads();
  }
  if (t == 0) {
    const float bound = red[0][0] * (nrs > 0 ? red[1][0] : 1.0f);
    int ex = 0;
    if (bound > 0.f && bound < 3.0e38f) {      // zero, inf or NaN gradients keep S = 1 (and propagate)
      int eb;
      (void)frexpf(bound, &eb);
      ex = 14 - eb;
      ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
    }
    scale[0] = ldexpf(1.0f, ex);
    scale[1] = ldexpf(1.0f, -ex);
  }
}

float* gemm_grad_blockmax(ng_ctx* ctx, int* capacity) {
  if (capacity) *capacity = GH_MAXBLOCKS;
  return reinterpret_cast<float*>(small_scratch(ctx));
}

int gemm_grad_scale_from_blocks(ng_ctx* ctx, hipStream_t st, int nblocks, const float** scale_out) {
  float* blockmax = reinterpret_cast<float*>(small_scratch(ctx));
  if (!blockmax) return NG_ERR_NOMEM;
  NG_REQUIRE(ctx, nblocks >= 0 && nblocks <= GH_MAXBLOCKS, "grad scale: block count");
  float* scale = blockmax + GH_MAXBLOCKS + GH_RSBLOCKS;
  hipLaunchKernelGGL(gh_scale_kernel, dim3(1), dim3(256), 0, st, blockmax, nblocks, nullptr, 0, scale);
  NG_HIP(ctx, hipGetLastError());
  *scale_out = scale;
  return NG_OK;
}

int gemm_grad_scale(ng_ctx* ctx, hipStream_t st, const float* dY, int64_t M, int N, const float* rowscale,
                    const float** scale_out) {
  float* blockmax = reinterpret_cast<float*>(small_scratch(ctx));   // fixed address: callers keep *scale_out
  if (!blockmax) return NG_ERR_NOMEM;
  float* rsmax = blockmax + GH_MAXBLOCKS;
  float* scale = rsmax + GH_RSBLOCKS;
  const int64_t n4 = M * N / 4;                      // N % 4 == 0 (dense_dx / dense_dw contracts)
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(2048, cdiv(n4, 256 * 4)));
  const int nr = rowscale ? (int)std::max<int64_t>(1, std::min<int64_t>(GH_RSBLOCKS, cdiv(M, 1024))) : 0;
  ProfScope ps(ctx, st, "grad_scale");
  hipLaunchKernelGGL(gh_absmax_kernel, dim3(nb), dim3(256), 0, st, reinterpret_cast<const float4*>(dY), n4, blockmax);
  if (rowscale) hipLaunchKernelGGL(gh_rsmax_kernel, dim3(nr), dim3(256), 0, st, rowscale, M, rsmax);
  hipLaunchKernelGGL(gh_scale_kernel, dim3(1), dim3(256), 0, st, blockmax, nb, rsmax, nr, scale);
  NG_HIP(ctx, hipGetLastError());
  *scale_out = scale;
  return NG_OK;
}

// Epilogue of one 32-row block for the kernels below: the lane holds, for ONE row m, NJ x 4 groups of 4 consecutive
// columns (acc[j][4q..4q+3] -> column n0 + 32 j + 8 q + (0..3)).  All loads (bias, residual) are issued BEFORE the first
// store and the activation branch is taken once per block: written element by element, the compiler put a
// `s_waitcnt vmcnt(0)` — which also drains every store in flight — behind each bias and each residual load, 64 full
// memory round trips per workgroup; that serial tail, not the matrix pipe, was half of the kernel's time at the
// reference's default width (tools/f256_ab.py with the loop body switched off: 0.14-0.18 of 0.33-0.36 ms).
template <int NJ>
__device__ __forceinline__ void gx_epilogue_block(const GxArgs& a, const f32x16 (&acc)[NJ], int64_t m, int n0, float rs) {
  float4 v[NJ][4], r[NJ][4];
  const int64_t o = m * a.N + n0;
  {   // range guard (ng_internal.h): inf - inf of an out-of-range piece arrives here as NaN
    float chk = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) chk += fabsf(acc[j][t]);
    range_guard_raise(a.guard, not_finite(chk * rs));
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (a.R) r[j][q] = *reinterpret_cast<const float4*>(a.R + o + 32 * j + 8 * q);
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias) b = *reinterpret_cast<const float4*>(a.bias + n0 + 32 * j + 8 * q);
      v[j][q] = make_float4(fmaf(acc[j][4 * q + 0], rs, b.x), fmaf(acc[j][4 * q + 1], rs, b.y),
                            fmaf(acc[j][4 * q + 2], rs, b.z), fmaf(acc[j][4 * q + 3], rs, b.w));
    }
  if (a.act == NG_ACT_SOFTPLUS) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[j][q].x = act_apply(NG_ACT_SOFTPLUS, v[j][q].x); v[j][q].y = act_apply(NG_ACT_SOFTPLUS, v[j][q].y);
        v[j][q].z = act_apply(NG_ACT_SOFTPLUS, v[j][q].z); v[j][q].w = act_apply(NG_ACT_SOFTPLUS, v[j][q].w);
      }
  } else if (a.act != NG_ACT_NONE) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[j][q].x = act_apply(a.act, v[j][q].x); v[j][q].y = act_apply(a.act, v[j][q].y);
        v[j][q].z = act_apply(a.act, v[j][q].z); v[j][q].w = act_apply(a.act, v[j][q].w);
      }
  }
  // plain stores: the next kernel reads these outputs out of L2 / the Infinity Cache (non-temporal stores measured
  // 1.2-1.3x slower end to end at the default width)
  if (a.S) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(a.S + o + 32 * j + 8 * q) = v[j][q];
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 y = v[j][q];
      if (a.R) { y.x += r[j][q].x; y.y += r[j][q].y; y.z += r[j][q].z; y.w += r[j][q].w; }
      *reinterpret_cast<float4*>(a.Y + o + 32 * j + 8 * q) = y;
    }
}

template <bool GRAD, int NBW>
__global__ __launch_bounds__(256, 2) void gemm_h2_fwd_kernel(GxArgs a) {
  constexpr int WCHUNK = gx_wchunk(NBW), BN = 64 * NBW;
  extern __shared__ __attribute__((aligned(16))) char smem_gx[];
  char* sX = smem_gx;                       // [2][128][80 B]
  char* sW = smem_gx + 2 * GX_XPLANE;       // [2 NBW][2][2][1 KB]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nh = wave & 1, mh = wave >> 1;
  // column tiles of a row tile on one XCD, consecutive in time (see gemm_h2_fwdr_kernel)
  int bx = blockIdx.x, by = blockIdx.y;
  if (gridDim.y > 1 && (gridDim.x & 7) == 0) {
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    const int r = L & 7, q = L >> 3;
    by = q % gridDim.y;
    bx = (q / gridDim.y) * 8 + r;
  }
  const int64_t m0 = (int64_t)bx * GX_BM;
  const int ct = by;
  const int KT = a.K / GX_BK;

  // this thread's slice of the X tile: row tid >> 1, 16 floats at column 16 (tid & 1) of the k-step
  const int xr = tid >> 1, xh = tid & 1;
  const float* xp = a.X + std::min<int64_t>(m0 + xr, a.M - 1) * a.K + 16 * xh;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(a.Wimg), 0, (unsigned)((int64_t)(a.N / BN) * KT * WCHUNK), 0x00020000);

  const float* sp = GRAD && a.Sin ? a.Sin + std::min<int64_t>(m0 + xr, a.M - 1) * a.K + 16 * xh : nullptr;
  const float gS = GRAD && a.gscale ? a.gscale[0] : 1.0f;          // power of two (gemm_grad_scale)
  const float oscale = GX_WINV * (GRAD && a.gscale ? a.gscale[1] : 1.0f);
  const float rsi = gS * (GRAD && a.rs_in ? a.rs_in[std::min<int64_t>(m0 + xr, a.M - 1)] : 1.0f);
  float4 xv[4], sv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xv[i] = *reinterpret_cast<const float4*>(xp + 4 * i);
    if (GRAD && sp) sv[i] = *reinterpret_cast<const float4*>(sp + 4 * i);
  }

  f32x16 acc[NBW][2];
#pragma unroll
  for (int j = 0; j < NBW; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

#pragma unroll 1
  for (int kt = 0; kt < KT; ++kt) {
    NG_LDS_BARRIER();                       // the previous step's fragment reads are done
    // W pieces of this step: one-KB wave copies, 2 NBW per wave
#pragma unroll
    for (int c = 0; c < 2 * NBW; ++c) {
      const int kb = wave + 4 * c;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(sW + kb * 1024), 16,
                                               lane * 16, (ct * KT + kt) * WCHUNK + kb * 1024, 0, 0);
    }
    // X pieces: split the 16 prefetched values, two 16-B stores per plane
    {
      float v[16] = {xv[0].x, xv[0].y, xv[0].z, xv[0].w, xv[1].x, xv[1].y, xv[1].z, xv[1].w,
                     xv[2].x, xv[2].y, xv[2].z, xv[2].w, xv[3].x, xv[3].y, xv[3].z, xv[3].w};
      if (GRAD) {
        if (sp) {
          const float sg[16] = {sv[0].x, sv[0].y, sv[0].z, sv[0].w, sv[1].x, sv[1].y, sv[1].z, sv[1].w,
                                sv[2].x, sv[2].y, sv[2].z, sv[2].w, sv[3].x, sv[3].y, sv[3].z, sv[3].w};
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] *= act_grad_from_out(a.act_in, sg[j]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] *= rsi;
      }
      unsigned h[8], l[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) split2_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
      char* d = sX + xr * GX_XROW + 32 * xh;
      *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
      *reinterpret_cast<u32x4*>(d + 16) = u32x4{h[4], h[5], h[6], h[7]};
      *reinterpret_cast<u32x4*>(d + GX_XPLANE) = u32x4{l[0], l[1], l[2], l[3]};
      *reinterpret_cast<u32x4*>(d + GX_XPLANE + 16) = u32x4{l[4], l[5], l[6], l[7]};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the W copies has landed
    NG_LDS_BARRIER();
    // next step's X slice, requested behind the wait so that its HBM latency hides under this step's MFMAs
    // (clamped k: the last prefetch re-reads the last step)
    {
      const int64_t koff = (int64_t)GX_BK * std::min(kt + 1, KT - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xv[i] = *reinterpret_cast<const float4*>(xp + koff + 4 * i);
        if (GRAD && sp) sv[i] = *reinterpret_cast<const float4*>(sp + koff + 4 * i);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 wa[NBW][2], xb[2][2];
#pragma unroll
      for (int j = 0; j < NBW; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          wa[j][p] = *reinterpret_cast<const u32x4*>(sW + (((NBW * nh + j) * 2 + ks) * 2 + p) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          xb[i][p] = *reinterpret_cast<const u32x4*>(sX + p * GX_XPLANE + (32 * (2 * mh + i) + l31) * GX_XROW +
                                                     (16 * ks + 8 * half) * 2);
#pragma unroll
      for (int j = 0; j < NBW; j += 2)
#pragma unroll
        for (int i = 0; i < 2; ++i) mma3_2a(wa[j], wa[j + 1], xb[i], acc[j][i], acc[j + 1][i]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // epilogue: lane holds, for row m = m0 + 32 (2 mh + i) + l31, columns n = BN ct + 32 (NBW nh + j) + 8 q + 4 half + (0..3)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m0 + 32 * (2 * mh + i) + l31;
    if (m >= a.M) continue;
    f32x16 blk[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) blk[j] = acc[j][i];
    gx_epilogue_block<NBW>(a, blk, m, BN * ct + 32 * NBW * nh + 4 * half, oscale * (a.rowscale ? a.rowscale[m] : 1.0f));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------------------------
// Variant for N % 256 == 0 with the WEIGHT fragments loaded straight into registers.
// Why: the kernel above copies 48 KB of W pieces per (128-row tile, k-step) into LDS by LDS-DMA; that path delivers
// ~25 GB/s per CU (6.4 TB/s chip-wide: MI355X guide, "ldsdma-fill"), and with every row tile re-streaming the whole
// image (1.2 GB per [131k x 768] x [768 x 256] product) the copies, not the matrix pipe, set the pace: 34-37 % busy.
// Here wave w owns the 64 output columns [64 w, 64 w + 64) of the 256-column tile for ALL 128 rows, so its W^T
// fragments are private to it: buffer_load_dwordx4 from the fragment-ordered image in L2 to VGPRs, requested one
// 16-wide k-half ahead (48 MFMAs = 1.5k cycles of cover), no LDS, no DMA.  Only the X piece planes go through LDS, in a
// two-stage ring (2 x 30 KB): ONE barrier per 32-wide k-step, X(kt+1) is split while step kt multiplies and X(kt+2)
// is on its way from HBM.  256 threads, two workgroups per CU.
constexpr int G4_LDS = 2 * 2 * GX_XPLANE;                 // 40,960

template <bool GRAD>
__global__ __launch_bounds__(256, 2) void gemm_h2_fwdr_kernel(GxArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_g4[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int nq = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroup -> (row tile, column tile).  The column tiles of a row tile read the same X rows; dealt to the XCDs by linear
  // id, the plain (x, y) order puts them rounds apart and on different L2s.  With a multiple of 8 row tiles, ids r, r + 8,
  // r + 16, ... (one XCD, consecutive in time) take the column tiles of row tile 8 q + r.
  int bx = blockIdx.x, by = blockIdx.y;
  if (gridDim.y > 1 && (gridDim.x & 7) == 0) {
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    const int r = L & 7, q = L >> 3;
    by = q % gridDim.y;
    bx = (q / gridDim.y) * 8 + r;
  }
  const int64_t m0 = (int64_t)bx * GX_BM;
  const int ct = by;
  const int KT = a.K / GX_BK;
  constexpr int WCHUNK = gx_wchunk(4);

  // this thread's slice of an X tile: row tid >> 1, 16 floats at column 16 (tid & 1) of the k-step
  const int xr = tid >> 1, xh = tid & 1;
#ifdef GX_ABL_XL2      // ablation: every workgroup reads the same 1024 rows of X (L2 hits instead of HBM) — results are wrong
  const int64_t xrow = std::min<int64_t>(m0 + xr, a.M - 1) & 1023;
#else
  const int64_t xrow = std::min<int64_t>(m0 + xr, a.M - 1);
#endif
  const float* xp = a.X + xrow * a.K + 16 * xh;
  const float* sp = GRAD && a.Sin ? a.Sin + xrow * a.K + 16 * xh : nullptr;
  const float gS = GRAD && a.gscale ? a.gscale[0] : 1.0f;          // power of two (gemm_grad_scale)
  const float oscale = GX_WINV * (GRAD && a.gscale ? a.gscale[1] : 1.0f);
  const float rsi = gS * (GRAD && a.rs_in ? a.rs_in[xrow] : 1.0f);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(a.Wimg), 0, (unsigned)((int64_t)(a.N / 256) * KT * WCHUNK), 0x00020000);

  auto kstep = [&](int i) { return std::min(i, KT - 1); };
  float4 xv[4], sv[4];
  auto x_request = [&](int i) {            // clamped: the last requests re-read the last step
    const int64_t koff = (int64_t)GX_BK * kstep(i);
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      // plain loads: a dY row is re-read by every column tile, non-temporal loads measured 1.4x slower
      xv[i4] = *reinterpret_cast<const float4*>(xp + koff + 4 * i4);
      if (GRAD && sp) sv[i4] = *reinterpret_cast<const float4*>(sp + koff + 4 * i4);
    }
  };
  auto x_fill = [&](char* sX) {
    float v[16] = {xv[0].x, xv[0].y, xv[0].z, xv[0].w, xv[1].x, xv[1].y, xv[1].z, xv[1].w,
                   xv[2].x, xv[2].y, xv[2].z, xv[2].w, xv[3].x, xv[3].y, xv[3].z, xv[3].w};
    if (GRAD) {
      if (sp) {
        const float sg[16] = {sv[0].x, sv[0].y, sv[0].z, sv[0].w, sv[1].x, sv[1].y, sv[1].z, sv[1].w,
                              sv[2].x, sv[2].y, sv[2].z, sv[2].w, sv[3].x, sv[3].y, sv[3].z, sv[3].w};
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] *= act_grad_from_out(a.act_in, sg[j]);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] *= rsi;
    }
    unsigned h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split2_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
    char* d = sX + xr * GX_XROW + 32 * xh;
    *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(d + 16) = u32x4{h[4], h[5], h[6], h[7]};
    *reinterpret_cast<u32x4*>(d + GX_XPLANE) = u32x4{l[0], l[1], l[2], l[3]};
    *reinterpret_cast<u32x4*>(d + GX_XPLANE + 16) = u32x4{l[4], l[5], l[6], l[7]};
  };
  // W^T fragments of the k-half (kt, ks) for this wave's two 32-column blocks
  auto w_request = [&](u32x4 (&wa)[2][2], int i, int ks) {
#ifdef GX_ABL_WL1      // ablation: every request reads the fragments of k-tile 0 (L1 hits instead of L2) — results are wrong
    const int ktc = 0;
    (void)i;
#else
    const int ktc = kstep(i);
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const auto raw = __builtin_amdgcn_raw_buffer_load_b128(
            wrs, lane * 16, (ct * KT + ktc) * WCHUNK + (((2 * nq + j) * 2 + ks) * 2 + p) * 1024, 0);
        wa[j][p] = __builtin_bit_cast(u32x4, raw);
      }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  auto multiply = [&](const char* sX, const u32x4 (&wa)[2][2], int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 xb[2];
#pragma unroll
      for (int p = 0; p < 2; ++p)
        xb[p] = *reinterpret_cast<const u32x4*>(sX + p * GX_XPLANE + (32 * i + l31) * GX_XROW + (16 * ks + 8 * half) * 2);
      mma3_2a(wa[0], wa[1], xb, acc[0][i], acc[1][i]);
    }
  };

  u32x4 w0[2][2], w1[2][2];
  x_request(0);
  w_request(w0, 0, 0);
  x_fill(smem_g4);
  x_request(1);
  NG_LDS_BARRIER();
#pragma unroll 1
  for (int kt = 0; kt < KT; ++kt) {
    const char* cur = smem_g4 + (kt & 1) * (2 * GX_XPLANE);
    char* nxt = smem_g4 + ((kt + 1) & 1) * (2 * GX_XPLANE);
    w_request(w1, kt, 1);
    if (kt + 1 < KT) { x_fill(nxt); x_request(kt + 2); }
    multiply(cur, w0, 0);
    __builtin_amdgcn_sched_barrier(0);
    w_request(w0, kt + 1, 0);
    multiply(cur, w1, 1);
    __builtin_amdgcn_sched_barrier(0);
    NG_LDS_BARRIER();
  }

#ifdef GX_ABL_NOEPI    // ablation: the products are formed, nothing is written
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[j][i][r]));
  return;
#endif
  // epilogue: lane holds, for row m = m0 + 32 i + l31, columns n = 256 ct + 32 (2 nq + j) + 8 q + 4 half + (0..3)
  float rsv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rsv[i] = oscale * (a.rowscale ? a.rowscale[std::min<int64_t>(m0 + 32 * i + l31, a.M - 1)] : 1.0f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + 32 * i + l31;
    if (m >= a.M) continue;
    const f32x16 blk[2] = {acc[0][i], acc[1][i]};
    gx_epilogue_block<2>(a, blk, m, 256 * ct + 64 * nq + 4 * half, rsv[i]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// SHORT operands (one molecule per call: M = 2770 rows give the 128-row tiling 22 workgroups on 256 CUs).  32-row
// tiles, 256 columns per workgroup as above (wave w: columns [64 w, 64 w + 64), W^T fragments from L2 into registers),
// but everything is sized for latency instead of reuse: a 12-MFMA k-half covers 0.2 us, an L2 round trip takes three
// times that and a first touch of X more, so the fragments run THREE k-halves ahead (ring of four) and X moves in
// 128-wide k-tiles (one barrier and one prefetch per 96 MFMAs of a wave), split into a two-stage LDS ring.
constexpr int GS_ROW = 272;                      // bytes per row of an X piece plane: 128 fp16 + 16 (b128 rows conflict-free)
constexpr int GS_PLANE = 32 * GS_ROW;            // 8,704
constexpr int GS_LDS = 2 * 2 * GS_PLANE;         // 34,816

template <bool GRAD>
__global__ __launch_bounds__(256, 2) void gemm_h2_short_kernel(GxArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_gs[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int nq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t m0 = (int64_t)blockIdx.x * 32;
  const int ct = blockIdx.y;
  const int KT = a.K / GX_BK, NT = a.K / 128;
  constexpr int WCHUNK = gx_wchunk(4);

  // this thread's slice of an X tile: row tid >> 3, 16 floats at column 16 (tid & 7) of the 128-wide k-tile
  const int xr = tid >> 3, xc = tid & 7;
  const int64_t xrow = std::min<int64_t>(m0 + xr, a.M - 1);
  const float* xp = a.X + xrow * a.K + 16 * xc;
  const float* sp = GRAD && a.Sin ? a.Sin + xrow * a.K + 16 * xc : nullptr;
  const float gS = GRAD && a.gscale ? a.gscale[0] : 1.0f;          // power of two (gemm_grad_scale)
  const float oscale = GX_WINV * (GRAD && a.gscale ? a.gscale[1] : 1.0f);
  const float rsi = gS * (GRAD && a.rs_in ? a.rs_in[xrow] : 1.0f);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(a.Wimg), 0, (unsigned)((int64_t)(a.N / 256) * KT * WCHUNK), 0x00020000);

  float4 xv[4], sv[4];
  auto x_request = [&](int t) {            // clamped: the last request re-reads the last tile
    const int64_t koff = (int64_t)128 * std::min(t, NT - 1);
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      xv[i4] = *reinterpret_cast<const float4*>(xp + koff + 4 * i4);
      if (GRAD && sp) sv[i4] = *reinterpret_cast<const float4*>(sp + koff + 4 * i4);
    }
  };
  auto x_fill = [&](char* sX) {
    float v[16] = {xv[0].x, xv[0].y, xv[0].z, xv[0].w, xv[1].x, xv[1].y, xv[1].z, xv[1].w,
                   xv[2].x, xv[2].y, xv[2].z, xv[2].w, xv[3].x, xv[3].y, xv[3].z, xv[3].w};
    if (GRAD) {
      if (sp) {
        const float sg[16] = {sv[0].x, sv[0].y, sv[0].z, sv[0].w, sv[1].x, sv[1].y, sv[1].z, sv[1].w,
                              sv[2].x, sv[2].y, sv[2].z, sv[2].w, sv[3].x, sv[3].y, sv[3].z, sv[3].w};
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] *= act_grad_from_out(a.act_in, sg[j]);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] *= rsi;
    }
    unsigned h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split2_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
    char* d = sX + xr * GS_ROW + 32 * xc;
    *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(d + 16) = u32x4{h[4], h[5], h[6], h[7]};
    *reinterpret_cast<u32x4*>(d + GS_PLANE) = u32x4{l[0], l[1], l[2], l[3]};
    *reinterpret_cast<u32x4*>(d + GS_PLANE + 16) = u32x4{l[4], l[5], l[6], l[7]};
  };
  // W^T fragments of k-half q (= 32-wide step q >> 1, half q & 1) for this wave's two 32-column blocks; requests
  // past the end re-read the last half (never multiplied)
  auto w_request = [&](u32x4 (&wa)[2][2], int q) {
    const int qc = std::min(q, 2 * KT - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const auto raw = __builtin_amdgcn_raw_buffer_load_b128(
            wrs, lane * 16, (ct * KT + (qc >> 1)) * WCHUNK + (((2 * nq + j) * 2 + (qc & 1)) * 2 + p) * 1024, 0);
        wa[j][p] = __builtin_bit_cast(u32x4, raw);
      }
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  u32x4 w[4][2][2];
  x_request(0);
  w_request(w[0], 0); w_request(w[1], 1); w_request(w[2], 2);
  x_fill(smem_gs);
  x_request(1);
  NG_LDS_BARRIER();
#pragma unroll 1
  for (int t = 0; t < NT; ++t) {
    const char* cur = smem_gs + (t & 1) * (2 * GS_PLANE);
    char* nxt = smem_gs + ((t + 1) & 1) * (2 * GS_PLANE);
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) {
      w_request(w[(hh + 3) & 3], 8 * t + hh + 3);
      if (hh == 0 && t + 1 < NT) { x_fill(nxt); x_request(t + 2); }
      u32x4 xb[2];
#pragma unroll
      for (int p = 0; p < 2; ++p)
        xb[p] = *reinterpret_cast<const u32x4*>(cur + p * GS_PLANE + l31 * GS_ROW + (16 * hh + 8 * half) * 2);
      mma3_2a(w[hh & 3][0], w[hh & 3][1], xb, acc[0], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    NG_LDS_BARRIER();
  }

  // epilogue: lane holds, for row m = m0 + l31, columns n = 256 ct + 32 (2 nq + j) + 8 q + 4 half + (0..3)
  const int64_t m = m0 + l31;
  if (m < a.M) gx_epilogue_block<2>(a, acc, m, 256 * ct + 64 * nq + 4 * half, oscale * (a.rowscale ? a.rowscale[m] : 1.0f));
}

// the short kernel: N % 256 (its column tiling), K % 128 (its k-tiles)
static bool gx_short_ok(int K, int N) { return N % 256 == 0 && K % 128 == 0; }

bool gemm_h2_fwd_ok(int64_t M, int K, int N) {
  if (sw().gemm_math_fp32) return false;
  const bool tall = M >= 4096, short_op = M >= 256 && gx_short_ok(K, N);
  // (N < 2^20: the pack job carries N in 20 bits of one int — gx_pack_job)
  return K % GX_BK == 0 && N % GX_BN == 0 && K >= 64 && (tall || short_op) && (int64_t)N * K * 4 < ((int64_t)1 << 31) && N < (1 << 20);
}

static int gx_launch(ng_ctx* ctx, hipStream_t st, GxArgs& a, const float* W, int trans, bool grad, const char* tag) {
  const int nbw = a.N % 256 == 0 ? 4 : 2;          // 256-column tiles when N allows
  const int BN = 64 * nbw;
  const size_t img_bytes = (size_t)a.N * a.K * 4;  // two fp16 pieces per weight
  bool have = false;
  char* img = (char*)cached_image(ctx, W, gx_image_kind(trans, nbw), img_bytes, &have);
  const bool cached = img != nullptr;
  if (!img) img = (char*)aux_workspace(ctx, img_bytes);   // callers hold pointers into the main workspace
  if (!img) return NG_ERR_NOMEM;
  if (!have) {
    // (an image whose source is a weight tensor itself is rebuilt behind ng_adam_step from here on; one registered by
    // gemm_h2_prepack_mp arrives valid and never comes here)
    const PackJob j = gx_pack_job(a.K, a.N, trans, nbw, W, img);
    const int rc = pack_launch(ctx, st, j);
    if (rc) return rc;
    if (cached) cache_set_job(ctx, W, gx_image_kind(trans, nbw), j);
  }
  a.Wimg = img;
  ProfScope ps(ctx, st, tag);
  const dim3 grid((unsigned)cdiv(a.M, GX_BM), (unsigned)(a.N / BN));
  if (gx_short_ok(a.K, a.N) && cdiv(a.M, GX_BM) * (a.N / BN) * 2 <= ctx->num_cu) {
    // fewer 128-row tiles than half the CUs
    const dim3 grid1((unsigned)cdiv(a.M, 32), (unsigned)(a.N / 256));
    if (grad) hipLaunchKernelGGL((gemm_h2_short_kernel<true>), grid1, dim3(256), GS_LDS, st, a);
    else hipLaunchKernelGGL((gemm_h2_short_kernel<false>), grid1, dim3(256), GS_LDS, st, a);
  } else if (nbw == 4) {
    if (grad) hipLaunchKernelGGL((gemm_h2_fwdr_kernel<true>), grid, dim3(256), G4_LDS, st, a);
    else hipLaunchKernelGGL((gemm_h2_fwdr_kernel<false>), grid, dim3(256), G4_LDS, st, a);
  } else {
    if (grad) hipLaunchKernelGGL((gemm_h2_fwd_kernel<true, 2>), grid, dim3(256), gx_lds(2), st, a);
    else hipLaunchKernelGGL((gemm_h2_fwd_kernel<false, 2>), grid, dim3(256), gx_lds(2), st, a);
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

int gemm_h2_fwd(ng_ctx* ctx, hipStream_t st, int64_t M, int K, int N, int act, const float* X, const float* W,
                const float* b, const float* rowscale, const float* R, float* Y, float* S, const char* tag, RangeGuard guard) {
  GxArgs a{};
  a.guard = guard;
  a.M = M; a.K = K; a.N = N; a.X = X; a.bias = b; a.rowscale = rowscale; a.R = R; a.Y = Y; a.S = S; a.act = act;
  return gx_launch(ctx, st, a, W, 0, false, tag);
}

// dX[m][k] = (add ? add[m][k] : 0) + sum_n dP[m][n] W[k][n],  dP = dY * act'(S) * rowscale   (dense_dx's contract);
// here the contraction runs over Nout and the output has Kin columns
int gemm_h2_dx(ng_ctx* ctx, hipStream_t st, int64_t M, int Kin, int Nout, int act, const float* dY, const float* S,
               const float* rowscale, const float* W, const float* add, float* dX, const float* gscale, const char* tag,
               RangeGuard guard) {
  GxArgs a{};
  a.guard = guard;
  a.M = M; a.K = Nout; a.N = Kin; a.X = dY; a.R = add; a.Y = dX; a.act = NG_ACT_NONE;
  a.Sin = act == NG_ACT_NONE ? nullptr : S; a.rs_in = rowscale; a.act_in = act;
  if (!gscale) {
    int rc = gemm_grad_scale(ctx, st, dY, M, Nout, rowscale, &gscale);
    if (rc) return rc;
  }
  a.gscale = gscale;
  return gx_launch(ctx, st, a, W, 1, true, tag);
}

// The images of an MPLayer weight in its GEMM form Wp[n F + l][m] (`Wp`: the cached plain copy, the key the products will look
// their image up by) for the update product (trans = 0: [M][E F] x [E F][F]) or the dA product (trans = 1: [M][F] x Wp^T),
// packed from `w` itself and registered with `w` as their source: ng_adam_step's one pack launch refreshes them together with
// the plain copy, and gx_launch finds them valid.  No-op when the image cache is off or the product will not take this path.
int gemm_h2_prepack_mp(ng_ctx* ctx, hipStream_t st, int64_t M, int F, int E, const float* w, const float* Wp, int trans) {
  const int K = trans ? F : E * F, N = trans ? E * F : F;
  if (!ctx->wcache || !gemm_h2_fwd_ok(M, K, N)) return NG_OK;
  const int nbw = N % 256 == 0 ? 4 : 2;
  bool have = false;
  char* img = (char*)cached_image(ctx, Wp, gx_image_kind(trans, nbw), (size_t)N * K * 4, &have);
  if (!img || have) return NG_OK;
  const PackJob j = gx_pack_job(K, N, 2 + trans, nbw, w, img);
  const int rc = pack_launch(ctx, st, j);
  if (rc) return rc;
  cache_set_job(ctx, Wp, gx_image_kind(trans, nbw), j);
  return NG_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// dW[k][n] = sum_m X[m][k] dP[m][n]   (dP = dY * act'(S) * rowscale), the contraction runs over the ROWS of both
// operands: both live in LDS as fp16 piece images [32 rows][128 + 8] and are read as COLUMNS with ds_read_b64_tr_b16
// (rows stored permuted so that the four rows of a read sit 16 banks apart — see edge_bwd_h2.hip).  256 threads, output
// tile 128 (k) x 128 (n), 32 rows per step, the rows split over blockIdx.z into partials [z][K][N] that the caller
// reduces (dense_dw).  MFMA: A = dP columns (rows n of D), B = X columns (columns k of D): a lane ends with 4
// consecutive n of one k.
typedef short gx_s16x4 __attribute__((ext_vector_type(4)));
constexpr int GT_ROWB = 272;                    // image row stride (bytes)
constexpr int GT_PLANE = 32 * GT_ROWB;          // 8,704
constexpr int GT_LDS = 2 * 2 * GT_PLANE;        // 34,816

struct GtArgs {
  int64_t M, rows_per_z;
  int K, N;                // X [M][K], dY/S [M][N]
  const float* X;
  const float* dY;
  const float* S;          // may be nullptr
  const float* rowscale;   // may be nullptr
  int act;
  float* partial;          // [nz][K][N]
  const float* gscale;     // {S, 1/S} of dP (device memory)
  RangeGuard guard;        // raised when an accumulator comes out non-finite
};

__device__ __forceinline__ u32x4 gt_tr_frag(const char* p) {
  const gx_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gx_s16x4*)p);
  const gx_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gx_s16x4*)(p + GT_ROWB));
  const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
  return u32x4{a[0], a[1], b[0], b[1]};
}

__device__ __forceinline__ void gt_store16(char* img, int prow, int col0, const float (&v)[16]) {
  unsigned h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split2_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
  char* d = img + prow * GT_ROWB + col0 * 2;
  *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<u32x4*>(d + 16) = u32x4{h[4], h[5], h[6], h[7]};
  *reinterpret_cast<u32x4*>(d + GT_PLANE) = u32x4{l[0], l[1], l[2], l[3]};
  *reinterpret_cast<u32x4*>(d + GT_PLANE + 16) = u32x4{l[4], l[5], l[6], l[7]};
}

__global__ __launch_bounds__(256, 2) void gemm_h2_dw_kernel(GtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_gt[];
  char* sXi = smem_gt;                    // X image  [2][32][272 B]
  char* sPi = smem_gt + 2 * GT_PLANE;     // dP image
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nh = wave & 1, kh = wave >> 1;
  const int k0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  const int64_t r0 = (int64_t)blockIdx.z * a.rows_per_z;
  const int64_t r1 = std::min<int64_t>(r0 + a.rows_per_z, a.M);

  const float gS = a.gscale[0], gI = a.gscale[1];       // dP is split as S * dP, the partial is written as acc / S
  // loader: thread -> row tid >> 3 of the 32-row step, 16 columns at 16 (tid & 7)
  const int lr = tid >> 3, lc = 16 * (tid & 7);
  const int e16 = lr & 15;
  const int prow = 16 * (lr >> 4) + 4 * (e16 & 3) + (e16 >> 2);   // see hx_prow_g in edge_bwd_h2.hip
  float xv[16], pv[16];
  auto load = [&](int64_t row) {
    const bool ok = row < r1;
    const int64_t rc = ok ? row : a.M - 1;
    const float* xp = a.X + rc * a.K + k0 + lc;
    const float* dp = a.dY + rc * a.N + n0 + lc;
    const float rs = ok ? gS * (a.rowscale ? a.rowscale[rc] : 1.0f) : 0.0f;     // rows past the end contribute zero
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 x = *reinterpret_cast<const float4*>(xp + 4 * i);
      float4 d = *reinterpret_cast<const float4*>(dp + 4 * i);
      if (a.S) {
        const float4 s = *reinterpret_cast<const float4*>(a.S + rc * a.N + n0 + lc + 4 * i);
        d.x *= act_grad_from_out(a.act, s.x); d.y *= act_grad_from_out(a.act, s.y);
        d.z *= act_grad_from_out(a.act, s.z); d.w *= act_grad_from_out(a.act, s.w);
      }
      xv[4 * i + 0] = x.x; xv[4 * i + 1] = x.y; xv[4 * i + 2] = x.z; xv[4 * i + 3] = x.w;
      pv[4 * i + 0] = d.x * rs; pv[4 * i + 1] = d.y * rs; pv[4 * i + 2] = d.z * rs; pv[4 * i + 3] = d.w * rs;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // transposing-read lane offsets: 16-lane group g reads [4 rows][16 columns]; lane i supplies row (i>>2), columns 4(i&3)..
  const int g = lane >> 4, li = lane & 15;
  const int lane_off = (4 * (li >> 2) + 2 * (g >> 1)) * GT_ROWB + (16 * (g & 1) + 4 * (li & 3)) * 2;

  load(r0 + lr);
#pragma unroll 1
  for (int64_t rb = r0; rb < r1; rb += 32) {
    NG_LDS_BARRIER();
    gt_store16(sXi, prow, lc, xv);
    gt_store16(sPi, prow, lc, pv);
    NG_LDS_BARRIER();
    load(rb + 32 + lr);                   // next step (rows past r1 load row M-1 and are zeroed)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 pa[2][2], xb[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          pa[j][p] = gt_tr_frag(sPi + p * GT_PLANE + 16 * ks * GT_ROWB + lane_off + 64 * (2 * nh + j));
          xb[j][p] = gt_tr_frag(sXi + p * GT_PLANE + 16 * ks * GT_ROWB + lane_off + 64 * (2 * kh + j));
        }
#pragma unroll
      for (int i = 0; i < 2; ++i) mma3_2a(pa[0], pa[1], xb[i], acc[0][i], acc[1][i]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D rows = n (from dP), D cols = k (from X): lane holds k = k0 + 32 (2 kh + i) + l31, n = n0 + 32 (2 nh + j) + 8q + 4 half + (0..3)
  float* part = a.partial + (int64_t)blockIdx.z * a.K * a.N;
  {
    float chk = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) chk += fabsf(acc[j][i][t]);
    range_guard_raise(a.guard, not_finite(chk * gI));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = k0 + 32 * (2 * kh + i) + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + 32 * (2 * nh + j) + 8 * q + 4 * half;
        *reinterpret_cast<float4*>(part + (int64_t)k * a.N + n) =
            make_float4(gI * acc[j][i][4 * q + 0], gI * acc[j][i][4 * q + 1], gI * acc[j][i][4 * q + 2], gI * acc[j][i][4 * q + 3]);
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// dW for Kin % 256 == 0 and Nout % 256 == 0: output tile 256 (k) x 256 (n), 512 threads = 8 waves, one workgroup per
// CU.  Same images / transposing reads as above at twice the tile edge: per 32-row step a wave issues 96 MFMAs for the
// same 32 values split and 2 barriers (the 128 x 128 kernel: 48), and every operand row is read by half as many
// workgroups (the aggregate A [N, 768] of the MPLayer weight gradient once instead of twice, dP 3 x instead of 6 x).
constexpr int GT8_ROWB = 528;                    // image row stride (bytes): 256 fp16 + 16, 132 dwords = 4 mod 64 banks
constexpr int GT8_PLANE = 32 * GT8_ROWB;         // 16,896
constexpr int GT8_LDS = 2 * 2 * 2 * GT8_PLANE;   // 135,168: two (X image, dP image) pairs

__device__ __forceinline__ u32x4 gt8_tr_frag(const char* p) {
  const gx_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gx_s16x4*)p);
  const gx_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gx_s16x4*)(p + GT8_ROWB));
  const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
  return u32x4{a[0], a[1], b[0], b[1]};
}

__device__ __forceinline__ void gt8_store16(char* img, int prow, int col0, const float (&v)[16]) {
  unsigned h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split2_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
  char* d = img + prow * GT8_ROWB + col0 * 2;
  *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<u32x4*>(d + 16) = u32x4{h[4], h[5], h[6], h[7]};
  *reinterpret_cast<u32x4*>(d + GT8_PLANE) = u32x4{l[0], l[1], l[2], l[3]};
  *reinterpret_cast<u32x4*>(d + GT8_PLANE + 16) = u32x4{l[4], l[5], l[6], l[7]};
}

__global__ __launch_bounds__(512, 1) void gemm_h2_dw8_kernel(GtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_gt8[];
  // two image pairs (the fp16 pieces freed the room): step s multiplies pair s & 1 while the rows of step s + 1 are split
  // into the other pair — ONE barrier per 32-row step — and two register sets keep the rows of steps s + 1 and s + 2 in
  // flight (a step is 48 MFMAs = 0.75 us since the move to two pieces, less than an HBM round trip under load)
  char* const img0 = smem_gt8;             // X image [2][32][528 B] | dP image
  char* const img1 = smem_gt8 + 4 * GT8_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nh = wave & 1, kh = wave >> 1;           // n-blocks 4 nh .. 4 nh + 3, k-blocks 2 kh, 2 kh + 1
  // Workgroup -> (tile, row chunk z).  The tiles of one row chunk read the SAME rows of dY (and, across n-tiles, of X);
  // workgroups are dealt to the 8 XCDs round-robin by their linear id, so with the plain (x, y, z) order the three k-tiles
  // of a chunk sit on three different L2s and dY comes out of HBM three times.  When the chunk count is a multiple of 8
  // (dw_plan arranges that) ids r, r + 8, r + 16, ... — one XCD — take the tiles of chunk 8 q + r.
  int tx = blockIdx.x, ty = blockIdx.y, zc = blockIdx.z;
  if ((gridDim.z & 7) == 0) {
    const int tiles = gridDim.x * gridDim.y;
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int r = L & 7, q = L >> 3;
    const int t = q % tiles;
    zc = (q / tiles) * 8 + r;
    tx = t % gridDim.x; ty = t / gridDim.x;
  }
  const int k0 = tx * 256, n0 = ty * 256;
  const int64_t r0 = (int64_t)zc * a.rows_per_z;
  const int64_t r1 = std::min<int64_t>(r0 + a.rows_per_z, a.M);

  const float gS = a.gscale[0], gI = a.gscale[1];       // dP is split as S * dP, the partial is written as acc / S
  // loader: thread -> row tid >> 4 of the 32-row step, 16 columns at 16 (tid & 15)
  const int lr = tid >> 4, lc = 16 * (tid & 15);
  const int e16 = lr & 15;
  const int prow = 16 * (lr >> 4) + 4 * (e16 & 3) + (e16 >> 2);   // see hx_prow_g in edge_bwd_h2.hip
  float xa[16], pa_[16], xb_[16], pb_[16];
  auto load = [&](float (&xv)[16], float (&pv)[16], int64_t row) {
    const bool ok = row < r1;
    const int64_t rc = ok ? row : a.M - 1;
    const float* xp = a.X + rc * a.K + k0 + lc;
    const float* dp = a.dY + rc * a.N + n0 + lc;
    const float rs = ok ? gS * (a.rowscale ? a.rowscale[rc] : 1.0f) : 0.0f;     // rows past the end contribute zero
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 x = *reinterpret_cast<const float4*>(xp + 4 * i);
      float4 d = *reinterpret_cast<const float4*>(dp + 4 * i);
      if (a.S) {
        const float4 s = *reinterpret_cast<const float4*>(a.S + rc * a.N + n0 + lc + 4 * i);
        d.x *= act_grad_from_out(a.act, s.x); d.y *= act_grad_from_out(a.act, s.y);
        d.z *= act_grad_from_out(a.act, s.z); d.w *= act_grad_from_out(a.act, s.w);
      }
      xv[4 * i + 0] = x.x; xv[4 * i + 1] = x.y; xv[4 * i + 2] = x.z; xv[4 * i + 3] = x.w;
      pv[4 * i + 0] = d.x * rs; pv[4 * i + 1] = d.y * rs; pv[4 * i + 2] = d.z * rs; pv[4 * i + 3] = d.w * rs;
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  const int g = lane >> 4, li = lane & 15;
  const int lane_off = (4 * (li >> 2) + 2 * (g >> 1)) * GT8_ROWB + (16 * (g & 1) + 4 * (li & 3)) * 2;

  auto multiply = [&](const char* sXi, const char* sPi) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 pa[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          pa[j][p] = gt8_tr_frag(sPi + p * GT8_PLANE + 16 * ks * GT8_ROWB + lane_off + 64 * (4 * nh + j));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u32x4 xb[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          xb[p] = gt8_tr_frag(sXi + p * GT8_PLANE + 16 * ks * GT8_ROWB + lane_off + 64 * (2 * kh + i));
        mma3_2a(pa[0], pa[1], xb, acc[0][i], acc[1][i]);
        mma3_2a(pa[2], pa[3], xb, acc[2][i], acc[3][i]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  load(xa, pa_, r0 + lr);
  load(xb_, pb_, r0 + 32 + lr);
  gt8_store16(img0, prow, lc, xa);
  gt8_store16(img0 + 2 * GT8_PLANE, prow, lc, pa_);
  load(xa, pa_, r0 + 64 + lr);          // rows past r1 load row M-1 and are zeroed
  NG_LDS_BARRIER();
#pragma unroll 1
  for (int64_t rb = r0; rb < r1; rb += 64) {
    gt8_store16(img1, prow, lc, xb_);                       // rows rb + 32 (img1 was last read before the previous barrier)
    gt8_store16(img1 + 2 * GT8_PLANE, prow, lc, pb_);
    load(xb_, pb_, rb + 96 + lr);
    multiply(img0, img0 + 2 * GT8_PLANE);                   // rows rb
    NG_LDS_BARRIER();
    if (rb + 32 < r1) {                                     // uniform over the workgroup
      gt8_store16(img0, prow, lc, xa);                      // rows rb + 64
      gt8_store16(img0 + 2 * GT8_PLANE, prow, lc, pa_);
      load(xa, pa_, rb + 128 + lr);
      multiply(img1, img1 + 2 * GT8_PLANE);                 // rows rb + 32
      NG_LDS_BARRIER();
    }
  }
  // D rows = n (from dP), D cols = k (from X): lane holds k = k0 + 32 (2 kh + i) + l31, n = n0 + 32 (4 nh + j) + 8q + 4 half + (0..3)
  float* part = a.partial + (int64_t)zc * a.K * a.N;
  {
    float chk = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) chk += fabsf(acc[j][i][t]);
    range_guard_raise(a.guard, not_finite(chk * gI));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = k0 + 32 * (2 * kh + i) + l31;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + 32 * (4 * nh + j) + 8 * q + 4 * half;
        *reinterpret_cast<float4*>(part + (int64_t)k * a.N + n) =
            make_float4(gI * acc[j][i][4 * q + 0], gI * acc[j][i][4 * q + 1], gI * acc[j][i][4 * q + 2], gI * acc[j][i][4 * q + 3]);
      }
  }
}

bool gemm_h2_dw8_ok(int64_t M, int Kin, int Nout) {
  return !sw().gemm_math_fp32 && Kin % 256 == 0 && Nout % 256 == 0 && M >= 4096;
}

bool gemm_h2_dw_ok(int64_t M, int Kin, int Nout) {
  if (sw().gemm_math_fp32) return false;
  return Kin % 128 == 0 && Nout % 128 == 0 && M >= 4096;
}

// partial[z][Kin][Nout], z < nz, rows [z * rows_per_z, ...): same partial layout as the f32-input dense_dw GEMM
int gemm_h2_dw(ng_ctx* ctx, hipStream_t st, int64_t M, int Kin, int Nout, int act, const float* X, const float* dY,
               const float* S, const float* rowscale, float* partial, int nz, int64_t rows_per_z, const float* gscale,
               const char* tag, RangeGuard guard) {
  if (!gscale) {
    int rc = gemm_grad_scale(ctx, st, dY, M, Nout, rowscale, &gscale);
    if (rc) return rc;
  }
  GtArgs a;
  a.guard = guard;
  a.gscale = gscale;
  a.M = M; a.rows_per_z = rows_per_z; a.K = Kin; a.N = Nout; a.X = X; a.dY = dY; a.S = S; a.rowscale = rowscale;
  a.act = act; a.partial = partial;
  ProfScope ps(ctx, st, tag);
  if (gemm_h2_dw8_ok(M, Kin, Nout))
    hipLaunchKernelGGL(gemm_h2_dw8_kernel, dim3((unsigned)(Kin / 256), (unsigned)(Nout / 256), (unsigned)nz), dim3(512),
                       GT8_LDS, st, a);
  else
    hipLaunchKernelGGL(gemm_h2_dw_kernel, dim3((unsigned)(Kin / 128), (unsigned)(Nout / 128), (unsigned)nz), dim3(256), GT_LDS,
                       st, a);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

// ==================================================================================================================
// Gather-GEMM (round 4): the MPLayer update and the node-side pull of its backward at the reference's default width
// (atom_feature_size = 256, nmrgnn/model.py:22) WITHOUT the aggregate in HBM.  nmrgnn/layers.py:26-46:
//     forward   h' = act(v * sum_n A_n W_n) + h,     A[i][(n, l)]  = sum_j   e[i,j,n] h[nlist[i,j]][l]
//     backward  dh = dH + sum_n B_n Wn_n,            B[t][(n, m)]  = sum_in  e[i,j,n] dP[i][m]     (edges (i,j) with nlist[i,j] = t)
// Until round 3 both ran as aggregate -> [N, 768] in HBM (402 MB written, then read) -> GEMM, and the backward pull read
// 3 KB of dA per incoming edge through L2.  The kernel of this path is the window gather-GEMM of mp_gw.cuh (round 5: 256-row
// tiles, one wave per SIMD, the gathered operand formed in registers out of an LDS window).  Round 4's producer / consumer
// kernel (four producer waves walking the lists, four consumer waves multiplying; 375-450 us per launch against 300-330) was
// removed in round 6 — DESIGN_HISTORY.md has its measurements.  The k-step order (slab-major, n inside) is the weight
// image's (pack_bodies.cuh: PK_GG), built from w[l][m][n] directly — no Wp copy.
// Ranges.  Forward operands are activations (split unscaled; an |A| >= 65504 raises the guard).  The pull's operand is a
// gradient: the sums are multiplied by S * 2^-x before the split, S the power of two of dP (gemm_grad_scale) and 2^x >=
// max_n sum_in |e_n| of the row, so no piece can overflow; the epilogue multiplies the row by 2^x / S.  A raised guard is
// answered by mp_gg_repair_kernel: the same rows in plain fp32 on the vector ALU (slow, exact, an empty launch otherwise).
enum { GG_PADDED = 0, GG_CSR = 1, GG_REC = 2 };

struct GgArgs {
  int64_t M;               // rows (atoms)
  int F, Kpad;             // width of the gathered operand (= output width); slots per row of the padded form
  const float* G;          // gathered operand [M][F]: h (forward) / dP (pull)
  const int32_t* idx;      // GG_PADDED / GG_CSR: source row of an entry
  const float* ew;         //                     its E weights, [entries][E]
  const int32_t* ptr;      // GG_CSR / GG_REC: row extents [M + 1]
  const float4* rec;       // GG_REC: {source (int bits), e_0, e_1, e_2} per entry
  const char* Wimg;        // PK_GG image
  const float* w;          // the layer's weight in the reference layout (repair kernel)
  int mode;                // 0 forward, 1 pull (PK_GG)
  const float* rowscale;   // [M] or nullptr
  const float* R;          // residual / base [M][F] or nullptr
  float* Y;                // [M][F]
  float* S;                // activation output [M][F] or nullptr
  int act;
  float* A_out;            // forward, training: the aggregate [M][E*F] as a by-product (the backward's dw = A^T dP reads it), or nullptr
  const float* gscale;     // GRAD: {S, 1/S} of the gathered operand
  int no_window;           // mp_gw_kernel: take the memory path for every tile (NG_MP_GW=nowin)
  RangeGuard guard;
};

template <int LK, int E>
__device__ __forceinline__ float4 gg_entry(const GgArgs& a, int64_t t) {
  if (LK == GG_REC) return a.rec[t];
  float4 r;
  r.x = __builtin_bit_cast(float, a.idx[t]);
  r.y = a.ew[t * E];
  r.z = E > 1 ? a.ew[t * E + 1] : 0.f;
  r.w = E > 2 ? a.ew[t * E + 2] : 0.f;
  return r;
}

// The rows of the same call in plain fp32 (one workgroup per row at a time, thread = column), executed only when the
// kernel above raised the guard.  F == blockDim.x == 256.
template <int LK, int E>
__global__ __launch_bounds__(256) void mp_gg_repair_kernel(GgArgs a) {
  if (!range_guard_raised(a.guard)) return;
  __shared__ float sA[3 * 256];
  const int c = threadIdx.x, F = a.F;
  for (int64_t row = blockIdx.x; row < a.M; row += gridDim.x) {
    int64_t p0, p1;
    if (LK == GG_PADDED) { p0 = row * a.Kpad; p1 = p0 + a.Kpad; }
    else { p0 = a.ptr[row]; p1 = a.ptr[row + 1]; }
    float s[3] = {0.f, 0.f, 0.f};
    for (int64_t t = p0; t < p1; ++t) {
      const float4 e4 = gg_entry<LK, E>(a, t);
      const float g = a.G[(int64_t)__builtin_bit_cast(int, e4.x) * F + c];
      s[0] = fmaf(e4.y, g, s[0]); s[1] = fmaf(e4.z, g, s[1]); s[2] = fmaf(e4.w, g, s[2]);
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < E; ++n) sA[n * 256 + c] = s[n];
    __syncthreads();
    float acc = 0.f;
    for (int n = 0; n < E; ++n)
      for (int k = 0; k < F; ++k) {
        const float wv = a.mode == 0 ? a.w[((int64_t)k * F + c) * E + n] : a.w[((int64_t)c * F + k) * E + n];
        acc = fmaf(sA[n * 256 + k], wv, acc);
      }
    float v = acc * (a.rowscale ? a.rowscale[row] : 1.0f);
    v = act_apply(a.act, v);
    if (a.S) a.S[row * F + c] = v;
    a.Y[row * F + c] = v + (a.R ? a.R[row * F + c] : 0.f);
  }
}

#include "mp_gw.cuh"

// what the window gather-GEMM takes (everything else keeps the aggregate -> HBM -> GEMM path: mp_gg_supported)
static bool gw_shape_ok(int64_t M, int F, int E, bool has_ptr, int Kpad) {
  return E == 3 && F % 32 == 0 && M * (int64_t)F * 4 < ((int64_t)1 << 31) && (has_ptr || Kpad <= 16);
}

// DEFAULT for the forward of a call that does not keep the aggregate (inference: `A_save == nullptr`) on a batch of small
// graphs (ng_ctx_set_graph_span: every row's sources lie inside the 320-row window): the window gather-GEMM (mp_gw.cuh).
// Measured on MI355X at the bench batch (profiles/r05e_gw_ab.txt): 0.30-0.33 ms per layer against 0.134 + 0.227 for
// aggregate -> HBM -> GEMM.  Training calls (the aggregate is written for dw = A^T dP: its row-per-lane stores cost the
// kernel more than the round trip it saves) and the backward's pull (pull_win_kernel wins) keep their kernels unless
// NG_MP_GG=1 asks for this one.
bool mp_gw_infer_ok(ng_ctx* ctx, int64_t N, int K, int F, int E, bool csr, bool keeps_aggregate) {
  return !sw().gemm_math_fp32 && !sw().mp_layered && !keeps_aggregate && F == 256 && E == 3 && (csr || K <= 16) &&
         N >= sw().mp_gg_min_rows && N * (int64_t)F * 4 < ((int64_t)1 << 31) && ctx->graph_span > 0 && ctx->graph_span <= 256;
}

bool mp_gg_supported(int64_t N, int F, int E, int Kpad) {      // Kpad: slots per row of a padded list, 0 for CSR / record lists
  // OPT-IN (NG_MP_GG=1): measured on MI355X the kernel does not beat aggregate -> HBM -> GEMM yet (375-450 us per launch
  // against 134 + 240; the producer waves, four per CU, cannot keep enough gathers in flight — DESIGN section 4), so the
  // default stays the two-kernel path.  Below NG_MP_GG_MIN_ROWS rows (default 8192 = 64 tiles) a call has too few
  // 128-row tiles for 256 CUs either way.  Both are read with the other switches (ng_reload_env).
  return sw().mp_gg_on && !sw().gemm_math_fp32 && !sw().mp_layered && F == 256 && N >= sw().mp_gg_min_rows &&
         gw_shape_ok(N, F, E, Kpad == 0, Kpad);
}

// the PK_GG image of w for `mode` (cached / refreshed behind Adam when the image cache is on, else in the aux scratch)
static char* gg_image(ng_ctx* ctx, hipStream_t st, int F, int E, int mode, const float* w, int* rc) {
  const size_t bytes = (size_t)(F / 32) * E * gx_wchunk(4);
  bool have = false;
  char* img = (char*)cached_image(ctx, w, 15 + mode, bytes, &have);
  const bool cached = img != nullptr;
  if (!img) img = (char*)aux_workspace(ctx, bytes);
  *rc = NG_OK;
  if (!img) { *rc = NG_ERR_NOMEM; return nullptr; }
  if (!have) {
    PackJob j;
    j.kind = PK_GG; j.i0 = E; j.i1 = F | (mode << 16); j.src[0] = w; j.dst[0] = img;
    j.blocks = (int)cdiv((int64_t)(F / 32) * E * (F / 32) * 2 * 64, 256);
    *rc = pack_launch(ctx, st, j);
    if (*rc == NG_OK && cached) cache_set_job(ctx, w, 15 + mode, j);
  }
  return img;
}

template <int LK, bool GRAD>
static int gg_launch(ng_ctx* ctx, hipStream_t st, GgArgs& a, int E, const char* tag) {
  a.guard = range_guard_begin(ctx);
  if (!a.guard.word) return NG_ERR_NOMEM;
  if (!gw_shape_ok(a.M, a.F, E, a.ptr != nullptr, a.Kpad)) return fail(ctx, NG_ERR_INVALID, "gather-GEMM: shape outside the window form");
  {
    ProfScope ps(ctx, st, tag);
    a.no_window = sw().mp_gw_nowin ? 1 : 0;
    const unsigned wgrid = (unsigned)cdiv(a.M, GW_BM);
    if (LK == GG_PADDED && a.Kpad <= 16)
      hipLaunchKernelGGL((mp_gw_kernel<LK, 3, GRAD, 6>), dim3(wgrid), dim3(GW_THREADS), GW_LDS, st, a);
    else
      hipLaunchKernelGGL((mp_gw_kernel<LK, 3, GRAD, 8>), dim3(wgrid), dim3(GW_THREADS), GW_LDS, st, a);
#ifdef GW_STAMP
    if (getenv("NG_GW_STAMP")) {
      unsigned long long h[64];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(gw_stamps), sizeof(h));
      fprintf(stderr, "gw %s: stage %llu setup %llu dma0 %llu gather0 %llu loop %llu epi %llu | steps", tag, h[1] - h[0], h[2] - h[1], h[3] - h[2],
              h[4] - h[3], h[5] - h[4], h[6] - h[5]);
      for (int i = 8; i < 23; ++i) fprintf(stderr, " %llu", h[i + 1] - h[i]);
      fprintf(stderr, "\n");
    }
#endif
    NG_HIP(ctx, hipGetLastError());
  }
  ProfScope ps(ctx, st, "mp_gg_range_fallback");
  const unsigned rgrid = (unsigned)std::min<int64_t>(a.M, (int64_t)ctx->num_cu * 8);
  switch (E) {
    case 1: hipLaunchKernelGGL((mp_gg_repair_kernel<LK, 1>), dim3(rgrid), dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((mp_gg_repair_kernel<LK, 2>), dim3(rgrid), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((mp_gg_repair_kernel<LK, 3>), dim3(rgrid), dim3(256), 0, st, a); break;
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

// h_out = act(inv_degree * (A Wp)) + (residual ? h : 0),  A the neighbour aggregate of h over (row_ptr | K) lists
int mp_gg_fwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, int act, int residual, const float* h,
              const int32_t* row_ptr, const int32_t* col, const float* e, const float* inv_degree, const float* w,
              float* h_out, float* s_save, float* A_out) {
  int rc = NG_OK;
  GgArgs a{};
  a.A_out = A_out;
  a.Wimg = gg_image(ctx, st, F, E, 0, w, &rc);
  if (rc) return rc;
  a.M = N; a.F = F; a.Kpad = K; a.G = h; a.idx = col; a.ew = e; a.ptr = row_ptr; a.w = w; a.mode = 0;
  a.rowscale = inv_degree; a.R = residual ? h : nullptr; a.Y = h_out; a.S = s_save; a.act = act;
  return row_ptr ? gg_launch<GG_CSR, false>(ctx, st, a, E, "mp_gg_fwd") : gg_launch<GG_PADDED, false>(ctx, st, a, E, "mp_gg_fwd");
}

// dh_in = dh_out + sum_n B_n Wn_n,  B the aggregate of dP over the incoming-edge records {source, e_0..e_2} (csc order)
int mp_gg_pull(ng_ctx* ctx, hipStream_t st, int64_t N, int F, int E, const float* dP, const int32_t* csc_ptr,
               const float* csc_rec, const float* w, const float* dh_out, float* dh_in, const float* gscale) {
  int rc = NG_OK;
  GgArgs a{};
  a.Wimg = gg_image(ctx, st, F, E, 1, w, &rc);
  if (rc) return rc;
  if (!gscale) {
    rc = gemm_grad_scale(ctx, st, dP, N, F, nullptr, &gscale);
    if (rc) return rc;
  }
  a.M = N; a.F = F; a.Kpad = 0; a.G = dP; a.ptr = csc_ptr; a.rec = reinterpret_cast<const float4*>(csc_rec); a.w = w; a.mode = 1;
  a.R = dh_out; a.Y = dh_in; a.act = NG_ACT_NONE; a.gscale = gscale;
  return gg_launch<GG_REC, true>(ctx, st, a, E, "mp_gg_pull");
}

}  // namespace ng
