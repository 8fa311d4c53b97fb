// Fused persistent backward of the edge path, second design: ONE wave per SIMD with the whole
// 512-register file (edge_hidden_size == 128, edge_fc_layers == 4).  Same math and phase structure as
// edge_fused_bwd.hip (see its header); what changes is the mapping:
//
//   * 256 threads = 4 waves, one workgroup per CU, __launch_bounds__(256, 1): the three 128x128
//     weight-gradient accumulators are 192 accumulator registers per wave (2x2 tiles of 32x32 per layer),
//     the dZ accumulators 32 more; the remaining arch VGPRs hold operand prefetch and the next layer's
//     weight slab (64 VGPRs), so nothing spills (the 8-wave kernel sits on a 256-register cliff).
//   * tiles arrive by LDS-DMA (global_load_lds_dwordx4) into four LDS buffers one phase ahead — no
//     staging registers, 4 barriers per tile instead of 7.
//   * dW GEMM: 4 MFMAs per operand step (2 B + 2 A fragments read once, one step ahead);
//     dZ GEMM: 8 MFMAs per ds_read_b128 pair, A = W^T fragments resident in registers for the phase.
#include <algorithm>

#include "mfma_gemm.cuh"
#include "edge_fused.h"

namespace ng {

constexpr int B2_THREADS = 256;

struct EdgeBwd2Args {
  int64_t n_edges;
  const float* d_src;
  const float* d_eff;
  const float* centers;
  float neg_inv_gap;
  const float* WpkT;    // [3][4][16][64][4]: W_l[k = 32w + (lane&31)][n = 8t + 4*(lane>>5) + s]
  const float* Wo;      // [128][E]
  const float* z_save;  // [3][n_edges][128]
  const float* de;      // [n_edges][E]
  float* partial;       // [grid][part_stride]:  dW[3][128*128] | db[3][128] | dWo[128*E] | dbo[E]
  int part_stride;
};

// padded LDS tile (64 rows x 528 B) filled by 33 one-KiB LDS-DMA chunks; see edge_fused_bwd.hip notes:
// pad slots and rows past the end read some valid address (never consumed / multiplied by dE = 0)
__device__ __forceinline__ void dma_tile4(const float* __restrict__ src, int64_t row0, int64_t n_rows,
                                          float* __restrict__ buf, int wave, int lane) {
  for (int c = wave; c < 33; c += 4) {
    const int o = c * 1024 + lane * 16;
    const int row = o / (FLD * 4);
    int cb = o - row * (FLD * 4);
    if (cb >= FH * 4) cb = 0;
    int64_t gr = row0 + row;
    if (gr >= n_rows) gr = n_rows - 1;
    const float* g = src + gr * FH + (cb >> 2);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(buf + c * 256), 16, 0, 0);
  }
}

#define NG_DMA_BARRIER2()                                 \
  do {                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
    NG_LDS_BARRIER();                                     \
  } while (0)

// acc[kj][nj] (+)= sum_rows G[row][32(ns0+nj) + i] * Zp[row][32(ks0+kj) + j]   (D rows i = n, cols j = k)
// csum[nj] += this lane-half's column sums of G (bias gradient).
// ROLLED loop over 8 chunks of 4 steps with an explicit one-chunk-ahead operand prefetch: a fully
// unrolled body lets hipcc hoist every ds_read of the phase to the top and blow the register file.
__device__ __forceinline__ void dw_gemm4(f32x16 (&acc)[2][2], float (&csum)[2],
                                         const float* __restrict__ Zp, const float* __restrict__ G,
                                         int ks0, int ns0, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const float* zp = Zp + (4 * half) * FLD + ks0 * 32 + l31;
  const float* g0 = G + (4 * half) * FLD + ns0 * 32 + l31;
  float cb0[4], cb1[4], ca0[4], ca1[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    cb0[s] = zp[s * FLD]; cb1[s] = zp[s * FLD + 32]; ca0[s] = g0[s * FLD]; ca1[s] = g0[s * FLD + 32];
  }
#pragma unroll 1
  for (int t = 0; t < 8; ++t) {
    const int tn = t < 7 ? t + 1 : 7;
    float nb0[4], nb1[4], na0[4], na1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int off = (8 * tn + s) * FLD;
      nb0[s] = zp[off]; nb1[s] = zp[off + 32]; na0[s] = g0[off]; na1[s] = g0[off + 32];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0[s], cb0[s], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1[s], cb0[s], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0[s], cb1[s], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1[s], cb1[s], acc[1][1], 0, 0, 0);
      csum[0] += ca0[s];
      csum[1] += ca1[s];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) { cb0[s] = nb0[s]; cb1[s] = nb1[s]; ca0[s] = na0[s]; ca1[s] = na1[s]; }
  }
}

// dZ[row][k] = sum_n G[row][n] W[k][n] for k-slab `wave`, both 32-row tiles; Gout = dZ * (1 - exp(-Z)).
// W^T fragments stream from the fragment-ordered copy in L2 two iterations ahead; G one ahead (rolled).
__device__ __forceinline__ void dz_gemm2(const float* __restrict__ WpkT, int layer,
                                         const float* __restrict__ G, const float* __restrict__ Z,
                                         float* __restrict__ Gout, int wave, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const float4* wp = reinterpret_cast<const float4*>(WpkT) + ((layer * 4 + wave) * 16) * 64 + lane;
  const float* g0 = G + l31 * FLD + 4 * half;
  const float* g1 = g0 + 32 * FLD;
  float4 wc = wp[0], wn = wp[64];
  float4 x0 = *reinterpret_cast<const float4*>(g0);
  float4 x1 = *reinterpret_cast<const float4*>(g1);
#pragma unroll 1
  for (int t = 0; t < 16; ++t) {
    const int t1 = t < 15 ? t + 1 : 15, t2 = t < 14 ? t + 2 : 15;
    const float4 wnn = wp[t2 * 64];
    const float4 n0 = *reinterpret_cast<const float4*>(g0 + 8 * t1);
    const float4 n1 = *reinterpret_cast<const float4*>(g1 + 8 * t1);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.x, x0.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.x, x1.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.y, x0.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.y, x1.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.z, x0.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.z, x1.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.w, x0.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.w, x1.w, acc1, 0, 0, 0);
    wc = wn; wn = wnn; x0 = n0; x1 = n1;
  }
  // lane holds dZ[row = l31 (+32)][k = 32*wave + 8q + 4*half + (0..3)]
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = wave * 32 + 8 * q + 4 * half;
    const float4 z0 = *reinterpret_cast<const float4*>(Z + l31 * FLD + k);
    const float4 z1 = *reinterpret_cast<const float4*>(Z + (32 + l31) * FLD + k);
    float4 o0, o1;
    o0.x = acc0[4 * q + 0] * (1.0f - __expf(-z0.x)); o0.y = acc0[4 * q + 1] * (1.0f - __expf(-z0.y));
    o0.z = acc0[4 * q + 2] * (1.0f - __expf(-z0.z)); o0.w = acc0[4 * q + 3] * (1.0f - __expf(-z0.w));
    o1.x = acc1[4 * q + 0] * (1.0f - __expf(-z1.x)); o1.y = acc1[4 * q + 1] * (1.0f - __expf(-z1.y));
    o1.z = acc1[4 * q + 2] * (1.0f - __expf(-z1.z)); o1.w = acc1[4 * q + 3] * (1.0f - __expf(-z1.w));
    *reinterpret_cast<float4*>(Gout + l31 * FLD + k) = o0;
    *reinterpret_cast<float4*>(Gout + (32 + l31) * FLD + k) = o1;
  }
}

template <int E>
__global__ __launch_bounds__(B2_THREADS, 1) void edge_fused_bwd2_kernel(EdgeBwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // four tile buffers; roles per tile: P0: Z3 -> G2,  P1: Z2 -> G1,  P2: G3 -> R,  P3: Z1
  float* P0 = smem;
  float* P1 = P0 + FTM * FLD;
  float* P2 = P1 + FTM * FLD;
  float* P3 = P2 + FTM * FLD;
  float* sWo = P3 + FTM * FLD;           // [128*E]
  float* sCen = sWo + FH * FMAX_E;       // [128]
  float* sSc = sCen + FH;                // 2 x { dE[64*8], d_eff[64], mask[64] }
  constexpr int SC = FTM * FMAX_E + 2 * FTM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks0 = 2 * (wave >> 1), ns0 = 2 * (wave & 1);   // dW: 2x2 tiles
  const int c4t = tid & 31, rg = tid >> 5;                 // G3 / R tiles: 4 columns x rows rg + 8i
  const int cn = tid & 127, rh = tid >> 7;                 // dWo: column cn x rows 32rh..32rh+31

  for (int t = tid; t < FH * E; t += B2_THREADS) sWo[t] = a.Wo[t];
  if (tid < FH) sCen[tid] = a.centers[tid];

  f32x16 accW[3][2][2];
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[l][i][j][r] = 0.f;
  float csum[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  float accWo[E], accbo[E];
#pragma unroll
  for (int n = 0; n < E; ++n) { accWo[n] = 0.f; accbo[n] = 0.f; }

  const int64_t ntiles = (a.n_edges + FTM - 1) / FTM;
  const float* Z1g = a.z_save;
  const float* Z2g = a.z_save + a.n_edges * FH;
  const float* Z3g = a.z_save + 2 * a.n_edges * FH;

  float pf_de = 0.f, pf_dn = 0.f, pf_ds = 0.f;
  auto fetch_scalars = [&](int64_t row0) {
    pf_de = 0.f; pf_dn = 0.f; pf_ds = 0.f;
    if (tid < FTM * E) {
      const int64_t gr = row0 + tid / E;
      if (gr < a.n_edges && a.d_src[gr] > 0.f) pf_de = a.de[row0 * E + tid];
    }
    if (tid < FTM) {
      const int64_t gr = row0 + tid;
      if (gr < a.n_edges) { pf_ds = a.d_src[gr]; pf_dn = a.d_eff[gr]; }
    }
  };
  auto put_scalars = [&](float* sb) {
    if (tid < FTM * E) sb[tid] = pf_de;
    if (tid < FTM) {
      sb[FTM * FMAX_E + tid] = pf_dn;
      sb[FTM * FMAX_E + FTM + tid] = pf_ds > 0.f ? 1.f : 0.f;
    }
  };
  int par = 0;
  if ((int64_t)blockIdx.x < ntiles) {
    const int64_t r0 = (int64_t)blockIdx.x * FTM;
    dma_tile4(Z3g, r0, a.n_edges, P0, wave, lane);
    dma_tile4(Z2g, r0, a.n_edges, P1, wave, lane);
    fetch_scalars(r0);
    put_scalars(sSc);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * FTM;
    const float* sdE = sSc + par * SC;
    const float* sD = sdE + FTM * FMAX_E;
    const float* sM = sD + FTM;
    // ------------------------------------------------------------------ phase A   [VALU]
    dma_tile4(Z1g, row0, a.n_edges, P3, wave, lane);   // needed in phase C
    {   // G3 = (dE Wo^T) * s'(Z3) -> P2
      float wo[4][E];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < E; ++n) wo[j][n] = sWo[(4 * c4t + j) * E + n];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = rg + 8 * i;
        const float4 z = *reinterpret_cast<const float4*>(P0 + row * FLD + c4t * 4);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < E; ++n) {
          const float d = sdE[row * E + n];
#pragma unroll
          for (int j = 0; j < 4; ++j) g[j] += d * wo[j][n];
        }
        float4 o;
        o.x = g[0] * (1.0f - __expf(-z.x));
        o.y = g[1] * (1.0f - __expf(-z.y));
        o.z = g[2] * (1.0f - __expf(-z.z));
        o.w = g[3] * (1.0f - __expf(-z.w));
        *reinterpret_cast<float4*>(P2 + row * FLD + c4t * 4) = o;
      }
    }
    {   // dWo[cn][n] += sum_rows Z3[row][cn] dE[row][n] ; dbo[n] += sum_rows dE[row][n]  (rows 32rh..)
#pragma unroll
      for (int i0 = 0; i0 < 32; i0 += 8) {
        float zc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) zc[u] = P0[(32 * rh + i0 + u) * FLD + cn];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int n = 0; n < E; ++n) {
            const float dn = sdE[(32 * rh + i0 + u) * E + n];
            accWo[n] += zc[u] * dn;
            if (cn == 0) accbo[n] += dn;
          }
      }
    }
    NG_LDS_BARRIER();
    // ------------------------------------------------------------------ phase B (layer 3)
    if (tile + gridDim.x < ntiles) fetch_scalars((tile + gridDim.x) * FTM);   // lands long before phase D
    dw_gemm4(accW[2], csum[2], P1, P2, ks0, ns0, lane);
    dz_gemm2(a.WpkT, 2, P2, P1, P0, wave, lane);       // G2 -> P0
    NG_DMA_BARRIER2();                                  // Z1 has landed in P3
    // ------------------------------------------------------------------ phase C (layer 2)
    {   // R = m * rbf(d_eff) -> P2   (G3 is dead)
      const float4 mu4 = *reinterpret_cast<const float4*>(sCen + 4 * c4t);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = rg + 8 * i;
        const float d = sD[row], m = sM[row];
        const float u0 = d - mu4.x, u1 = d - mu4.y, u2 = d - mu4.z, u3 = d - mu4.w;
        float4 o;
        o.x = m * __expf(u0 * u0 * a.neg_inv_gap);
        o.y = m * __expf(u1 * u1 * a.neg_inv_gap);
        o.z = m * __expf(u2 * u2 * a.neg_inv_gap);
        o.w = m * __expf(u3 * u3 * a.neg_inv_gap);
        *reinterpret_cast<float4*>(P2 + row * FLD + c4t * 4) = o;
      }
    }
    dw_gemm4(accW[1], csum[1], P3, P0, ks0, ns0, lane);
    dz_gemm2(a.WpkT, 1, P0, P3, P1, wave, lane);       // G1 -> P1
    NG_LDS_BARRIER();
    // ------------------------------------------------------------------ phase D (layer 1)
    if (tile + gridDim.x < ntiles) {
      const int64_t rn = (tile + gridDim.x) * FTM;
      dma_tile4(Z3g, rn, a.n_edges, P0, wave, lane);
      dma_tile4(Z2g, rn, a.n_edges, P3, wave, lane);
      put_scalars(sSc + (par ^ 1) * SC);
    }
    dw_gemm4(accW[0], csum[0], P2, P1, ks0, ns0, lane);
    NG_DMA_BARRIER2();
    { float* tswap = P1; P1 = P3; P3 = tswap; }
    par ^= 1;
  }
  __syncthreads();

  // ---------------------------------------------------------------------- write this WG's partial
  float* part = a.partial + (int64_t)blockIdx.x * a.part_stride;
  {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int kj = 0; kj < 2; ++kj)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int k = (ks0 + kj) * 32 + l31;
            const int n = (ns0 + nj) * 32 + 8 * q + 4 * half;
            *reinterpret_cast<float4*>(part + l * FH * FH + k * FH + n) =
                make_float4(accW[l][kj][nj][4 * q + 0], accW[l][kj][nj][4 * q + 1],
                            accW[l][kj][nj][4 * q + 2], accW[l][kj][nj][4 * q + 3]);
          }
  }
  float* dbred = P0;                   // [3][128][2]  bias gradients: two lane-halves per column
  float* wored = P0 + 3 * FH * 2;      // [2][128*E]   dWo partials of the two row halves
  float* bored = wored + 2 * FH * E;   // [2][E]
  if (ks0 == 0) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int j = 0; j < 2; ++j) dbred[(l * FH + (ns0 + j) * 32 + l31) * 2 + half] = csum[l][j];
  }
#pragma unroll
  for (int n = 0; n < E; ++n) {
    wored[rh * FH * E + cn * E + n] = accWo[n];
    if (cn == 0) bored[rh * E + n] = accbo[n];
  }
  __syncthreads();
  for (int t = tid; t < 3 * FH; t += B2_THREADS) part[3 * FH * FH + t] = dbred[2 * t] + dbred[2 * t + 1];
  for (int t = tid; t < FH * E; t += B2_THREADS)
    part[3 * FH * FH + 3 * FH + t] = wored[t] + wored[FH * E + t];
  if (tid < E) part[3 * FH * FH + 3 * FH + FH * E + tid] = bored[tid] + bored[E + tid];
}

int edge_fused_bwd2_launch(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src,
                           const float* d_eff, const float* centers, float gap, const float* WpkT,
                           const float* Wo, const float* z_save, const float* de, float* partial,
                           int part_stride, int grid) {
  EdgeBwd2Args a;
  a.n_edges = n_edges; a.d_src = d_src; a.d_eff = d_eff; a.centers = centers;
  a.neg_inv_gap = (float)(-1.0 / (double)gap);
  a.WpkT = WpkT; a.Wo = Wo; a.z_save = z_save; a.de = de;
  a.partial = partial; a.part_stride = part_stride;
  const size_t lds = (size_t)(4 * FTM * FLD + FH * FMAX_E + FH + 2 * (FTM * FMAX_E + 2 * FTM)) * 4;
  ProfScope ps(ctx, st, "edge_fused_bwd");
#define NG_BW2(EE)                                                                                  \
  case EE:                                                                                          \
    hipLaunchKernelGGL((edge_fused_bwd2_kernel<EE>), dim3(grid), dim3(B2_THREADS), lds, st, a);     \
    break;
  switch (E) { NG_BW2(1) NG_BW2(2) NG_BW2(3) NG_BW2(4) NG_BW2(5) NG_BW2(6) NG_BW2(7) NG_BW2(8) }
#undef NG_BW2
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

}  // namespace ng
