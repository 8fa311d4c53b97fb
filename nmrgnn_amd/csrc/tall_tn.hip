// Weight gradients of the 64-feature node path as a persistent transposed product
//     dW[ka][kb] = sum_rows A[row][ka] * Bp[row][kb]        KA in {64, 128, 192}, KB = 64
// (MPLayer: A = aggregated features [N, 64E], Bp = dP;  FC Dense: A = layer input, Bp = dY * act'(S)).
// The [KA x 64] accumulator lives in registers (KA/64 tiles of 32x32 per wave), 64-row tiles of A and
// Bp stream through LDS one tile ahead (register prefetch across an LDS-only barrier), each workgroup
// writes ONE partial and reduce_z_kernel sums them (deterministic).  The bias gradient (column sums of
// Bp) is accumulated from the very fragments the MFMA consumes.
// Replaces the generic split-K tile GEMM on these shapes (49 TF -> see DESIGN.md).
#include <algorithm>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "reduce.cuh"

namespace ng {

constexpr int TN_TM = 64;
constexpr int TN_KB = 64;
constexpr int TN_LDB = TN_KB + 4;

struct TnArgs {
  int64_t N;
  const float* A;       // [N][lda], columns >= ka_valid read as 0
  int lda, ka_valid;
  const float* B;       // [N][ldb], columns >= kb_valid read as 0
  int ldb, kb_valid;
  const float* S_in;    // prologue: Bp = B * act'(S_in) (same layout as B) or nullptr
  int act_in;
  float* partial;       // [grid][stride]:  dW[KA][64] | db[64]
  int stride;
};

template <int KA, bool PRO>
__global__ __launch_bounds__(256, 2) void tall_tn_kernel(TnArgs a) {
  constexpr int LDA = KA + 4;
  constexpr int SA = KA / 64;        // 32-wide ka slabs per wave
  constexpr int CA4 = KA / 4, CB4 = TN_KB / 4;
  constexpr int NA = TN_TM * CA4 / 256, NB = TN_TM * CB4 / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                  // [64][LDA]
  float* sB = smem + TN_TM * LDA;    // [64][TN_LDB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int kbs = wave & 1, ka0 = (wave >> 1) * SA;

  f32x16 acc[SA];
#pragma unroll
  for (int j = 0; j < SA; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float csum = 0.f;

  const int64_t ntiles = (a.N + TN_TM - 1) / TN_TM;
  float4 va[NA], vb[NB], vs[PRO ? NB : 1];
  auto fetch = [&](int64_t tile) {
    const int64_t i0 = tile * TN_TM;
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int t = tid + u * 256;
      const int r = t / CA4, c4 = t % CA4;
      va[u] = (i0 + r < a.N && c4 * 4 < a.ka_valid)
                  ? *reinterpret_cast<const float4*>(a.A + (i0 + r) * a.lda + c4 * 4)
                  : f4zero();
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int t = tid + u * 256;
      const int r = t / CB4, c4 = t % CB4;
      const bool ok = i0 + r < a.N && c4 * 4 < a.kb_valid;
      vb[u] = ok ? *reinterpret_cast<const float4*>(a.B + (i0 + r) * a.ldb + c4 * 4) : f4zero();
      if (PRO) vs[u] = (ok && a.S_in) ? *reinterpret_cast<const float4*>(a.S_in + (i0 + r) * a.ldb + c4 * 4)
                                      : f4zero();
    }
  };
  if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x);

#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int t = tid + u * 256;
      *reinterpret_cast<float4*>(sA + (t / CA4) * LDA + (t % CA4) * 4) = va[u];
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int t = tid + u * 256;
      float4 x = vb[u];
      if (PRO && a.S_in) {
        x.x *= act_grad_from_out(a.act_in, vs[u].x); x.y *= act_grad_from_out(a.act_in, vs[u].y);
        x.z *= act_grad_from_out(a.act_in, vs[u].z); x.w *= act_grad_from_out(a.act_in, vs[u].w);
      }
      *reinterpret_cast<float4*>(sB + (t / CB4) * TN_LDB + (t % CB4) * 4) = x;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
    // D[i = kb][j = ka] += sum_rows Bp[row][kb] * A[row][ka];  rows 8t + 4*half + s
    const float* pb = sB + (4 * half) * TN_LDB + kbs * 32 + l31;
    const float* pa = sA + (4 * half) * LDA + ka0 * 32 + l31;
#pragma unroll 2
    for (int t = 0; t < 8; ++t) {
      float fb[4], fa[SA][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        fb[s] = pb[(8 * t + s) * TN_LDB];
#pragma unroll
        for (int j = 0; j < SA; ++j) fa[j][s] = pa[(8 * t + s) * LDA + 32 * j];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < SA; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[s], fa[j][s], acc[j], 0, 0, 0);
        csum += fb[s];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  float* part = a.partial + (int64_t)blockIdx.x * a.stride;
#pragma unroll
  for (int j = 0; j < SA; ++j) {
    const int ka = (ka0 + j) * 32 + l31;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kb = kbs * 32 + 8 * q + 4 * half;
      *reinterpret_cast<float4*>(part + ka * TN_KB + kb) =
          make_float4(acc[j][4 * q + 0], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
    }
  }
  // bias gradient: the waves of ka-group 0 hold, per lane-half, the column sums of Bp
  __syncthreads();
  float* red = smem;   // [64][2]
  if (ka0 == 0) red[(kbs * 32 + l31) * 2 + half] = csum;
  __syncthreads();
  if (tid < TN_KB) part[KA * TN_KB + tid] = red[2 * tid] + red[2 * tid + 1];
}

bool tall_tn_supported(int ka, int kb) {
  if (sw().dense_generic) return false;
  const int kap = (ka + 63) / 64 * 64;
  return ka % 4 == 0 && kb % 4 == 0 && kb <= 64 && (kap == 64 || kap == 128 || kap == 192);
}

// dW (layout [ka_valid][kb_valid], or the MPLayer map when w_map == 1) and optional db[kb_valid]
size_t tall_tn_scratch_floats(ng_ctx* ctx, int ka_valid) {
  const int kap = (ka_valid + 63) / 64 * 64;
  const size_t stride = (size_t)kap * TN_KB + TN_KB;
  return ((size_t)ctx->num_cu * 2 + 1) * stride;
}

int tall_tn(ng_ctx* ctx, hipStream_t st, int64_t N, const float* A, int lda, int ka_valid, const float* B,
            int ldb, int kb_valid, const float* S_in, int act_in, float* dW, float* db, int w_map, int F,
            int E, float* scratch, const char* tag) {
  const int kap = (ka_valid + 63) / 64 * 64;
  const int64_t ntiles = std::max<int64_t>(cdiv(N, TN_TM), 1);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu * 2);
  const int stride = kap * TN_KB + TN_KB;
  // scratch (caller-provided, tall_tn_scratch_floats): partials + a dense [kap][64] reduction target
  float* ws = scratch;
  float* partial = ws;
  float* dense = ws + (size_t)grid * stride;
  TnArgs a{};
  a.N = N; a.A = A; a.lda = lda; a.ka_valid = ka_valid; a.B = B; a.ldb = ldb; a.kb_valid = kb_valid;
  a.S_in = S_in; a.act_in = act_in; a.partial = partial; a.stride = stride;
  {
    ProfScope ps(ctx, st, tag);
    const bool pro = S_in != nullptr;
#define NG_TN(KA)                                                                                      \
  {                                                                                                    \
    const size_t lds = (size_t)TN_TM * (KA + 4 + TN_LDB) * 4;                                          \
    if (pro) hipLaunchKernelGGL((tall_tn_kernel<KA, true>), dim3(grid), dim3(256), lds, st, a);        \
    else hipLaunchKernelGGL((tall_tn_kernel<KA, false>), dim3(grid), dim3(256), lds, st, a);           \
  }
    if (kap == 64) NG_TN(64) else if (kap == 128) NG_TN(128) else NG_TN(192)
#undef NG_TN
    NG_HIP(ctx, hipGetLastError());
  }
  ProfScope ps(ctx, st, "reduce_partials");
  if (w_map == 1 || w_map == 2) {
    // MPLayer: idx = k*64 + c;  w_map 1: k = ne*F + l, c = m;  w_map 2: k = ne*F + m, c = l
    //          ->  dw[(l*F + m)*E + ne]   (kb_valid == 64 == F)
    launch_reduce_z(st, partial, grid, (int64_t)ka_valid * TN_KB, dW, w_map, F, E, TN_KB, stride);
  } else if (kb_valid == TN_KB && ka_valid == kap) {
    launch_reduce_z(st, partial, grid, (int64_t)stride, dense, 0);
    NG_HIP(ctx, hipMemcpyAsync(dW, dense, (size_t)ka_valid * TN_KB * 4, hipMemcpyDeviceToDevice, st));
    if (db) NG_HIP(ctx, hipMemcpyAsync(db, dense + kap * TN_KB, (size_t)kb_valid * 4, hipMemcpyDeviceToDevice, st));
  } else {
    launch_reduce_z(st, partial, grid, (int64_t)stride, dense, 0);
    NG_HIP(ctx, hipMemcpy2DAsync(dW, (size_t)kb_valid * 4, dense, (size_t)TN_KB * 4, (size_t)kb_valid * 4,
                                 ka_valid, hipMemcpyDeviceToDevice, st));
    if (db) NG_HIP(ctx, hipMemcpyAsync(db, dense + kap * TN_KB, (size_t)kb_valid * 4, hipMemcpyDeviceToDevice, st));
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

}  // namespace ng
