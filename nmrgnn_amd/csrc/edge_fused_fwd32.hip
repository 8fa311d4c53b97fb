// Fused persistent edge forward, 32-edge tiles / four workgroups per CU variant.
// Same math and data flow as edge_fused.hip (which see): RBF -> 3 x softplus Dense -> output layer,
// activations ping-pong in LDS, weights as register-resident MFMA A-fragments reloaded per layer.
// Difference: a workgroup owns 32 edges at a time (one 32x32 accumulator per wave, 40 KB of LDS,
// <= 128 VGPRs), so FOUR workgroups = 16 waves share a CU and their MFMA chains, softplus epilogues,
// LDS traffic and weight reloads overlap 4-deep instead of 2-deep.  The price is twice the weight
// reload traffic from L2 (64 KB per layer per 32 edges).  Selected with NG_EDGE_FWD=tm32.
#include <algorithm>

#include "mfma_gemm.cuh"
#include "edge_fused.h"

namespace ng {

constexpr int F32_TM = 32;

struct EdgeFwd32Args {
  int64_t n_edges;
  const float* d_src;
  const float* d_eff;
  const float* centers;
  float neg_inv_gap;
  const float* Wpk;       // [3][4][16][64][4]
  const float* bh[3];
  const float* Wo;        // [128][E]
  const float* bo;        // [E]
  float* e_out;           // [n_edges][E]
  float* z_save;          // [3][n_edges][128] or nullptr
};

__device__ __forceinline__ float softplus6(float x) {
  const float t = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
  return fmaf(0.6931471805599453f, __builtin_amdgcn_logf(1.0f + t), fmaxf(x, 0.0f));
}

__device__ __forceinline__ void load_wfrag32(float (&wf)[64], const float* __restrict__ Wpk, int layer,
                                             int wave, int lane) {
  const float4* p = reinterpret_cast<const float4*>(Wpk) + ((layer * 4 + wave) * 16) * 64 + lane;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 v = p[t * 64];
    wf[4 * t + 0] = v.x; wf[4 * t + 1] = v.y; wf[4 * t + 2] = v.z; wf[4 * t + 3] = v.w;
  }
}

__device__ __forceinline__ void hidden_layer32(float (&wf)[64], const float* __restrict__ Xin,
                                               float* __restrict__ Xout, const float* __restrict__ bias,
                                               int wave, int lane, const float* __restrict__ Wpk,
                                               int next_layer) {
  const int half = lane >> 5, l31 = lane & 31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* x0 = Xin + l31 * FLD + half * 4;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(x0 + 8 * t);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 0], a.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 1], a.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 2], a.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 3], a.w, acc, 0, 0, 0);
  }
  load_wfrag32(wf, Wpk, next_layer, wave, lane);   // latency hides under the epilogue + other workgroups
  const int ncol = 32 * wave + 4 * half;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + ncol + 8 * q);
    float4 v;
    v.x = softplus6(acc[4 * q + 0] + bv.x); v.y = softplus6(acc[4 * q + 1] + bv.y);
    v.z = softplus6(acc[4 * q + 2] + bv.z); v.w = softplus6(acc[4 * q + 3] + bv.w);
    *reinterpret_cast<float4*>(Xout + l31 * FLD + ncol + 8 * q) = v;
  }
}

__device__ __forceinline__ void save_tile32(const float* __restrict__ X, float* __restrict__ dst,
                                            int64_t row0, int64_t n_rows, int wave, int lane) {
  const int col = (lane & 31) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8 * wave + 2 * i + (lane >> 5);
    if (row0 + r < n_rows)
      *reinterpret_cast<float4*>(dst + (row0 + r) * FH + col) =
          *reinterpret_cast<const float4*>(X + r * FLD + col);
  }
}

template <int E, bool SAVE>
__global__ __launch_bounds__(256, 4) void edge_fused_fwd32_kernel(EdgeFwd32Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X0 = smem;                         // [32][132]
  float* X1 = smem + F32_TM * FLD;          // [32][132]
  float* sWo = X1 + F32_TM * FLD;           // [128*E]
  float* sMask = sWo + FH * FMAX_E;         // [32]
  float* sCen = sMask + F32_TM;             // [128]
  float* sBias = sCen + FH;                 // [3][128]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int t = tid; t < FH * E; t += 256) sWo[t] = a.Wo[t];
  if (tid < FH) {
    sCen[tid] = a.centers[tid];
    sBias[tid] = a.bh[0][tid];
    sBias[FH + tid] = a.bh[1][tid];
    sBias[2 * FH + tid] = a.bh[2][tid];
  }
  __syncthreads();

  const int64_t ntiles = (a.n_edges + F32_TM - 1) / F32_TM;
  float wf[64];
  load_wfrag32(wf, a.Wpk, 0, wave, lane);

#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * F32_TM;
    {   // RBF tile: thread -> (row = tid & 31, eighth = tid >> 5): 16 centres
      const int r = tid & 31, ei = tid >> 5;
      const int64_t gr = row0 + r;
      float ds = 0.f, de = 0.f;
      if (gr < a.n_edges) { ds = a.d_src[gr]; de = a.d_eff[gr]; }
      const float m = ds > 0.f ? 1.f : 0.f;
      if (ei == 0) sMask[r] = m;
      float* dst = X0 + r * FLD + 16 * ei;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 mu = *reinterpret_cast<const float4*>(sCen + 16 * ei + 4 * i);
        const float u0 = de - mu.x, u1 = de - mu.y, u2 = de - mu.z, u3 = de - mu.w;
        float4 v;
        v.x = m * __expf(u0 * u0 * a.neg_inv_gap);
        v.y = m * __expf(u1 * u1 * a.neg_inv_gap);
        v.z = m * __expf(u2 * u2 * a.neg_inv_gap);
        v.w = m * __expf(u3 * u3 * a.neg_inv_gap);
        *reinterpret_cast<float4*>(dst + 4 * i) = v;
      }
    }
    NG_LDS_BARRIER();
    hidden_layer32(wf, X0, X1, sBias, wave, lane, a.Wpk, 1);
    NG_LDS_BARRIER();
    if (SAVE) save_tile32(X1, a.z_save, row0, a.n_edges, wave, lane);
    hidden_layer32(wf, X1, X0, sBias + FH, wave, lane, a.Wpk, 2);
    NG_LDS_BARRIER();
    if (SAVE) save_tile32(X0, a.z_save + a.n_edges * FH, row0, a.n_edges, wave, lane);
    hidden_layer32(wf, X0, X1, sBias + 2 * FH, wave, lane, a.Wpk, 0);
    NG_LDS_BARRIER();
    if (SAVE) save_tile32(X1, a.z_save + 2 * a.n_edges * FH, row0, a.n_edges, wave, lane);
    {   // output layer: wave w -> rows 8w..8w+7, 8 lanes per row (k = 32i + 4*(lane&7) + s)
      const int r = 8 * wave + (lane >> 3), qq = lane & 7;
      float acc[E];
#pragma unroll
      for (int n = 0; n < E; ++n) acc[n] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 32 * i + 4 * qq;
        const float4 x = *reinterpret_cast<const float4*>(X1 + r * FLD + k);
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int n = 0; n < E; ++n) acc[n] += xs[s] * sWo[(k + s) * E + n];
      }
#pragma unroll
      for (int n = 0; n < E; ++n) {
        acc[n] += __shfl_xor(acc[n], 1, 64);
        acc[n] += __shfl_xor(acc[n], 2, 64);
        acc[n] += __shfl_xor(acc[n], 4, 64);
      }
      const int64_t gr = row0 + r;
      if (qq == 0 && gr < a.n_edges) {
        const float m = sMask[r];
#pragma unroll
        for (int n = 0; n < E; ++n) a.e_out[gr * E + n] = m * (acc[n] + a.bo[n]);
      }
    }
    NG_LDS_BARRIER();
  }
}

int edge_fused_fwd32(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src,
                     const float* d_eff, const float* centers, float gap, const float* const* W,
                     const float* const* b, float* e_out, float* z_save) {
  const size_t pk_floats = (size_t)3 * FH * FH;
  float* Wpk = (float*)workspace(ctx, pk_floats * 4);
  if (!Wpk) return NG_ERR_NOMEM;
  int rc = edge_fused_pack(ctx, st, W, Wpk, nullptr);
  if (rc) return rc;
  EdgeFwd32Args a;
  a.n_edges = n_edges; a.d_src = d_src; a.d_eff = d_eff; a.centers = centers;
  a.neg_inv_gap = (float)(-1.0 / (double)gap);
  a.Wpk = Wpk;
  a.bh[0] = b[0]; a.bh[1] = b[1]; a.bh[2] = b[2];
  a.Wo = W[3]; a.bo = b[3];
  a.e_out = e_out; a.z_save = z_save;
  const int64_t ntiles = cdiv(n_edges, F32_TM);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu * 4);
  const size_t lds = (size_t)(2 * F32_TM * FLD + FH * FMAX_E + F32_TM + 4 * FH) * 4;
  ProfScope ps(ctx, st, "edge_fused_fwd");
#define NG_FW32(EE)                                                                                 \
  case EE:                                                                                          \
    if (z_save)                                                                                     \
      hipLaunchKernelGGL((edge_fused_fwd32_kernel<EE, true>), dim3(grid), dim3(256), lds, st, a);   \
    else                                                                                            \
      hipLaunchKernelGGL((edge_fused_fwd32_kernel<EE, false>), dim3(grid), dim3(256), lds, st, a);  \
    break;
  switch (E) { NG_FW32(1) NG_FW32(2) NG_FW32(3) NG_FW32(4) NG_FW32(5) NG_FW32(6) NG_FW32(7) NG_FW32(8) }
#undef NG_FW32
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

}  // namespace ng
