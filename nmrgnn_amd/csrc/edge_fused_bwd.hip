// Fused persistent backward of the edge path (edge_hidden_size == 128, edge_fc_layers == 4).
// Backward of nmrgnn/model.py:251-261 (mask, RBF, EdgeFCBlock, mask); math in SURVEY App. B.
//
// Layers:  R --W1,b1--> Z1 --W2,b2--> Z2 --W3,b3--> Z3 --Wo,bo--> e   (Zt = softplus(.), e masked)
// Given de[n_edges,E] and the saved Z1..Z3, per 64-edge tile, all on-chip:
//   A:  dE = m*de ;  G3 = (dE Wo^T) * s'(Z3)            [VALU]   dWo += Z3^T dE, dbo += sum dE
//   B:  dW3 += Z2^T G3 [MFMA] ; db3 += colsum G3 ; dZ2 = G3 W3^T [MFMA] ; G2 = dZ2 * s'(Z2)
//   C:  dW2 += Z1^T G2 [MFMA] ; db2 += colsum G2 ; dZ1 = G2 W2^T [MFMA] ; G1 = dZ1 * s'(Z1)
//   D:  R = m*rbf(d) recomputed ; dW1 += R^T G1 [MFMA] ; db1 += colsum G1
// with s'(Z) = 1 - exp(-Z) (sigmoid of the pre-activation, recovered from the softplus output).
// The three 128x128 weight-gradient accumulators stay in registers for the whole launch (96 VGPRs
// per wave: 8 waves x 2 tiles of 32x32 x 3 layers); every workgroup writes ONE partial at the end
// and a small second kernel sums the partials (deterministic, no atomics).
//
// Mapping.  512 threads = 8 waves, one workgroup per CU (persistent), 3 LDS tiles [64][132] f32.
//   dW GEMM (contraction over the tile's 64 rows): wave w -> k-slab w>>1, n-slabs 2(w&1), 2(w&1)+1;
//     both operands are read K-strided (conflict-free ds_read_b32, natural [row][col] tiles).
//   dZ GEMM (contraction over n): wave w -> k-slab w&3, row-tile w>>2; A fragments = W rows streamed
//     from a fragment-ordered copy in L2 in 4-step chunks (double-buffered), B = G via ds_read_b128.
#include <algorithm>

#include "mfma_gemm.cuh"
#include "edge_fused.h"

namespace ng {

constexpr int BW_THREADS = 512;

struct EdgeBwdArgs {
  int64_t n_edges;
  const float* d_src;
  const float* d_eff;
  const float* centers;
  float neg_inv_gap;
  const float* WpkT;    // [3][4][16][64][4]: W_l[k = 32w + (lane&31)][n = 8t + 4*(lane>>5) + s]
  const float* Wo;      // [128][E]
  const float* z_save;  // [3][n_edges][128]
  const float* de;      // [n_edges][E]
  float* partial;       // [grid][part_stride]
  int part_stride;
  int tape_blocked;     // z_save layout (edge_fused.h), 1 = blocked inside full 32-edge groups
  int zero_rows;        // fallback of a multi-segment split-operand launch: partial rows grid .. zero_rows-1 are cleared
  RangeGuard guard;     // word != nullptr: run only if the guard carries this epoch (fallback of edge_bwd_h2)
  // live-edge view (ng_internal.h: LiveEdges): rows = compacted live slots (n_edges = slot count, rows = *n_live), d_eff
  // and the tape compacted, de[perm[row]] in the caller's slot layout; d_src is not read
  const int32_t* perm;
  const int32_t* n_live;
  int64_t z_layer_stride;
};

// partial layout (floats): dW[3][128*128] | db[3][128] | dWo[128*E] | dbo[E]
__host__ __device__ inline int bwd_part_floats(int E) { return 3 * FH * FH + 3 * FH + FH * E + E; }

__device__ __forceinline__ int64_t bw_tape_offset(int64_t gr, int col, int64_t n_rows, int blocked) {
  const int64_t g = gr >> 5;
  if (blocked && (g + 1) * 32 <= n_rows) {
    const int bo = col >> 5, q = (col >> 3) & 3, hf = (col >> 2) & 1, r = (int)(gr & 31);
    return g * 4096 + ((bo * 4 + q) * 64 + hf * 32 + r) * 4;
  }
  return gr * FH + col;
}

__device__ __forceinline__ void tile_to_regs(float4 (&pz)[4], const float* __restrict__ src,
                                             int64_t row0, int64_t n_rows, int tid, int blocked) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lin = tid + i * BW_THREADS;
    const int row = lin >> 5, c4 = lin & 31;
    // streamed once: non-temporal, so the 3.2 GB of saved activations do not evict the weight fragments
    // that every tile re-reads from L2
    typedef float nt4 __attribute__((ext_vector_type(4)));
    if (row0 + row < n_rows) {
      const nt4 v = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(src + bw_tape_offset(row0 + row, c4 * 4, n_rows, blocked)));
      pz[i] = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      pz[i] = f4zero();
    }
  }
}
__device__ __forceinline__ void regs_to_lds(const float4 (&pz)[4], float* __restrict__ buf, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lin = tid + i * BW_THREADS;
    const int row = lin >> 5, c4 = lin & 31;
    *reinterpret_cast<float4*>(buf + row * FLD + c4 * 4) = pz[i];
  }
}

// acc[j][n][k] += sum_rows G[row][n] * Zp[row][k]     (D rows i = n, cols j = k)
__device__ __forceinline__ void dw_gemm(f32x16 (&acc)[2], const float* __restrict__ Zp,
                                        const float* __restrict__ G, int kslab, int nsl0, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const float* zp = Zp + (4 * half) * FLD + kslab * 32 + l31;
  const float* g0 = G + (4 * half) * FLD + nsl0 * 32 + l31;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int off = (8 * t + s) * FLD;
      const float b = zp[off];
      const float a0 = g0[off];
      const float a1 = g0[off + 32];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
    }
  }
}

// bias gradient: column cn over the 16 rows of quarter rq
__device__ __forceinline__ void colsum16(float& acc, const float* __restrict__ G, int cn, int rq) {
#pragma unroll 4
  for (int r = 16 * rq; r < 16 * rq + 16; ++r) acc += G[r * FLD + cn];
}

// dZ[row][k] = sum_n G[row][n] W[k][n]  for k-slab zk, row-tile zrt;  then
// Gout[row][k] = dZ * (1 - exp(-Z[row][k]))
__device__ __forceinline__ void dz_gemm_epilogue(const float* __restrict__ G,
                                                 const float* __restrict__ Z,
                                                 float* __restrict__ Gout,
                                                 const float* __restrict__ WpkT, int layer, int zk,
                                                 int zrt, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // W^T fragments stream from L2 through BUFFER loads: scalar resource + scalar chunk offset + one VGPR lane
  // offset.  With flat/global addressing the compiler kept a 64-bit per-lane pointer per chunk, hoisted all
  // of them out of the tile loop, spilled them, and every reload carried a vmcnt(0) that serialised the
  // fragment stream behind its own latency.
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(WpkT), 0, 3 * FH * FH * 4, 0x00020000);
  const int wvo = lane * 16;
  const int wso = ((layer * 4 + zk) * 16) * 64 * 16;
  auto wload = [&](int i) {
    typedef float f32x4w __attribute__((ext_vector_type(4)));
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, wso + i * 1024, 0);
    const f32x4w v = __builtin_bit_cast(f32x4w, raw);
    return make_float4(v[0], v[1], v[2], v[3]);
  };
  const float* g = G + (zrt * 32 + l31) * FLD + 4 * half;
  float4 wc[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wc[0][i] = wload(i);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wc[(c + 1) & 1][i] = wload(4 * (c + 1) + i);
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int t = 4 * c + tt;
      const float4 x = *reinterpret_cast<const float4*>(g + 8 * t);
      const float4 w = wc[c & 1][tt];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, x.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, x.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, x.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, x.w, acc, 0, 0, 0);
    }
  }
  // lane holds dZ[row = 32*zrt + l31][k = 32*zk + 8q + 4*half + (0..3)]
  const int row = zrt * 32 + l31;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = zk * 32 + 8 * q + 4 * half;
    const float4 z = *reinterpret_cast<const float4*>(Z + row * FLD + k);
    float4 o;
    o.x = acc[4 * q + 0] * (1.0f - __expf(-z.x));
    o.y = acc[4 * q + 1] * (1.0f - __expf(-z.y));
    o.z = acc[4 * q + 2] * (1.0f - __expf(-z.z));
    o.w = acc[4 * q + 3] * (1.0f - __expf(-z.w));
    *reinterpret_cast<float4*>(Gout + row * FLD + k) = o;
  }
}

template <int E>
__global__ __launch_bounds__(BW_THREADS, 2) void edge_fused_bwd_kernel(EdgeBwdArgs a) {
  if (a.guard.word && !range_guard_raised(a.guard)) return;      // fallback launch: nothing went out of range
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* bufA = smem;                    // Z3 -> G2 -> R
  float* bufB = bufA + FTM * FLD;        // Z2 -> G1
  float* bufC = bufB + FTM * FLD;        // G3 -> Z1
  float* sWo = bufC + FTM * FLD;         // [128*E]
  float* sdE = sWo + FH * FMAX_E;        // [64*E] masked upstream gradient
  float* sD = sdE + FTM * FMAX_E;        // [64] d_eff
  float* sM = sD + FTM;                  // [64] mask
  float* sCen = sM + FTM;                // [128]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kslab = wave >> 1, nsl0 = 2 * (wave & 1);   // dW tiles
  const int zk = wave & 3, zrt = wave >> 2;             // dZ tile
  const int cn = tid & 127, rq = tid >> 7;              // column-sum / dWo ownership

  for (int t = tid; t < FH * E; t += BW_THREADS) sWo[t] = a.Wo[t];
  if (tid < FH) sCen[tid] = a.centers[tid];

  f32x16 accW[3][2];
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accW[l][j][r] = 0.f;
  float accb[3] = {0.f, 0.f, 0.f};
  float accWo[E];
#pragma unroll
  for (int n = 0; n < E; ++n) accWo[n] = 0.f;
  float accbo = 0.f;

  if (a.perm && *a.n_live < 0) return;      // (negative row count: edge_bwd_h2.hip)
  const int64_t n_edges = a.perm ? (int64_t)*a.n_live : a.n_edges;
  const int64_t ntiles = (n_edges + FTM - 1) / FTM;
  const float* Z1g = a.z_save;
  const float* Z2g = a.z_save + a.z_layer_stride;
  const float* Z3g = a.z_save + 2 * a.z_layer_stride;
  __syncthreads();

  // per-tile inputs are fetched one tile ahead (during phase D of the previous tile) so that phase A
  // never waits on HBM: Z3, Z2 tiles (32 VGPRs) and this thread's d / de scalars
  float4 pzA[4], pzB[4];
  float pf_ds = 0.f, pf_dn = 0.f, pf_de = 0.f, pf_dm = 0.f;
  auto prefetch = [&](int64_t row0) {
    tile_to_regs(pzA, Z3g, row0, n_edges, tid, a.tape_blocked);
    pf_ds = 0.f; pf_dn = 0.f; pf_de = 0.f; pf_dm = 0.f;
    if (tid < FTM) {
      const int64_t gr = row0 + tid;
      if (gr < n_edges) { pf_ds = a.perm ? 1.f : a.d_src[gr]; pf_dn = a.d_eff[gr]; }
    }
    if (tid < FTM * E) {
      // mask source and gradient are requested TOGETHER (clamped index, masked at use): reading de only
      // after d_src had arrived put a full HBM latency (5-8k cycles per tile) into this prefetch
      const int64_t gr = std::min<int64_t>(row0 + tid / E, n_edges - 1);
      if (a.perm) {
        // live view: every row is live; its gradient sits at the row's SLOT (one dependent load per tile — this kernel
        // is the strict-fp32 / range-fallback form, the split-operand kernel requests the slot two tiles ahead)
        pf_dm = row0 + tid / E < n_edges ? 1.f : 0.f;
        pf_de = a.de[(int64_t)a.perm[gr] * E + tid % E];
      } else {
        pf_dm = row0 + tid / E < n_edges ? a.d_src[gr] : 0.f;
        pf_de = a.de[std::min<int64_t>(row0 * E + tid, n_edges * E - 1)];
      }
    }
  };
  if ((int64_t)blockIdx.x < ntiles) prefetch((int64_t)blockIdx.x * FTM);

#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * FTM;
    // ------------------------------------------------------------------ phase A
    if (tid < FTM) {
      sD[tid] = pf_dn;
      sM[tid] = pf_ds > 0.f ? 1.f : 0.f;
    }
    if (tid < FTM * E) sdE[tid] = pf_dm > 0.f ? pf_de : 0.f;
    regs_to_lds(pzA, bufA, tid);
    // Z2 is requested here and lands in bufB at the end of this phase (~6k cycles of cover): holding it in
    // registers across the previous tile's last GEMM cost 16 VGPRs at the kernel's pressure peak
    tile_to_regs(pzB, Z2g, row0, n_edges, tid, a.tape_blocked);
    NG_LDS_BARRIER();
    tile_to_regs(pzA, Z1g, row0, n_edges, tid, a.tape_blocked);   // lands in bufC at the end of phase B
    // G3 = (dE Wo^T) * s'(Z3) -> bufC
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lin = tid + i * BW_THREADS;
      const int row = lin >> 5, c4 = lin & 31;
      const float4 z = *reinterpret_cast<const float4*>(bufA + row * FLD + c4 * 4);
      float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n = 0; n < E; ++n) {
        const float d = sdE[row * E + n];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] += d * sWo[(c4 * 4 + j) * E + n];
      }
      float4 o;
      o.x = g[0] * (1.0f - __expf(-z.x));
      o.y = g[1] * (1.0f - __expf(-z.y));
      o.z = g[2] * (1.0f - __expf(-z.z));
      o.w = g[3] * (1.0f - __expf(-z.w));
      *reinterpret_cast<float4*>(bufC + row * FLD + c4 * 4) = o;
    }
    // dWo[k][n] += sum_rows Z3[row][k] dE[row][n]   (thread: k = cn, rows 16rq..16rq+15)
#pragma unroll 4
    for (int r = 16 * rq; r < 16 * rq + 16; ++r) {
      const float z = bufA[r * FLD + cn];
#pragma unroll
      for (int n = 0; n < E; ++n) accWo[n] += z * sdE[r * E + n];
      if (cn < E) accbo += sdE[r * E + cn];
    }
    regs_to_lds(pzB, bufB, tid);    // Z2
    NG_LDS_BARRIER();
    // ------------------------------------------------------------------ phase B (layer 3)
    dw_gemm(accW[2], bufB, bufC, kslab, nsl0, lane);
    colsum16(accb[2], bufC, cn, rq);
    dz_gemm_epilogue(bufC, bufB, bufA, a.WpkT, 2, zk, zrt, lane);
    NG_LDS_BARRIER();
    regs_to_lds(pzA, bufC, tid);    // Z1
    NG_LDS_BARRIER();
    // ------------------------------------------------------------------ phase C (layer 2)
    dw_gemm(accW[1], bufC, bufA, kslab, nsl0, lane);
    colsum16(accb[1], bufA, cn, rq);
    dz_gemm_epilogue(bufA, bufC, bufB, a.WpkT, 1, zk, zrt, lane);
    NG_LDS_BARRIER();
    // R = m * rbf(d_eff) -> bufA
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lin = tid + i * BW_THREADS;
      const int row = lin >> 5, c4 = lin & 31;
      const float d = sD[row], m = sM[row];
      const float4 mu = *reinterpret_cast<const float4*>(sCen + c4 * 4);
      const float u0 = d - mu.x, u1 = d - mu.y, u2 = d - mu.z, u3 = d - mu.w;
      float4 o;
      o.x = m * __expf(u0 * u0 * a.neg_inv_gap);
      o.y = m * __expf(u1 * u1 * a.neg_inv_gap);
      o.z = m * __expf(u2 * u2 * a.neg_inv_gap);
      o.w = m * __expf(u3 * u3 * a.neg_inv_gap);
      *reinterpret_cast<float4*>(bufA + row * FLD + c4 * 4) = o;
    }
    NG_LDS_BARRIER();
    // ------------------------------------------------------------------ phase D (layer 1)
    if (tile + gridDim.x < ntiles) prefetch((tile + gridDim.x) * FTM);
    dw_gemm(accW[0], bufA, bufB, kslab, nsl0, lane);
    colsum16(accb[0], bufB, cn, rq);
    NG_LDS_BARRIER();
  }

  // ---------------------------------------------------------------------- write this WG's partial
  float* part = a.partial + (int64_t)blockIdx.x * a.part_stride;
  {
    const int half = lane >> 5, l31 = lane & 31;
    const int k = kslab * 32 + l31;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = (nsl0 + j) * 32 + 8 * q + 4 * half;
          *reinterpret_cast<float4*>(part + l * FH * FH + k * FH + n) =
              make_float4(accW[l][j][4 * q + 0], accW[l][j][4 * q + 1], accW[l][j][4 * q + 2],
                          accW[l][j][4 * q + 3]);
        }
  }
  // cross-row-quarter reduction of the column sums and dWo through LDS (bufA is free now)
  float* red = bufA;   // [4][3*128 + 128*E]
  const int red_stride = 3 * FH + FH * E + E;
#pragma unroll
  for (int l = 0; l < 3; ++l) red[rq * red_stride + l * FH + cn] = accb[l];
#pragma unroll
  for (int n = 0; n < E; ++n) red[rq * red_stride + 3 * FH + cn * E + n] = accWo[n];
  if (cn < E) red[rq * red_stride + 3 * FH + FH * E + cn] = accbo;
  __syncthreads();
  for (int t = tid; t < red_stride; t += BW_THREADS)
    part[3 * FH * FH + t] = red[t] + red[red_stride + t] + red[2 * red_stride + t] + red[3 * red_stride + t];
  // rows the (multi-segment) split-operand launch filled beyond this kernel's grid
  for (int64_t rowz = (int64_t)gridDim.x + blockIdx.x; rowz < a.zero_rows; rowz += gridDim.x)
    for (int t = tid; t < a.part_stride; t += BW_THREADS) a.partial[rowz * a.part_stride + t] = 0.f;
}

struct BwdOut {
  float* dW[3];
  float* db[3];
  float* dWo;
  float* dbo;
};

// out[...] = sum_wg partial[wg][idx], scattered to the individual gradient tensors.  One 1024-thread
// block per 64 consecutive elements; the 16 waves split the workgroup partials, so every lane has
// n_wg/16 independent coalesced loads in flight (the one-thread-per-element form took 66 us).
// n_live (nullable) negative: the kernels in front wrote no partials (their launch had nothing to do): the gradients are zero.
__global__ __launch_bounds__(1024) void edge_bwd_reduce_kernel(const float* __restrict__ partial, int n_wg,
                                                               int stride, int E, BwdOut o, const int32_t* __restrict__ n_live) {
  __shared__ float red[16][64];
  const int total = bwd_part_floats(E);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (n_live && *n_live < 0) n_wg = 0;
  if (idx < total) {
    // eight loads in flight per lane, added in workgroup order (same sum as the plain loop)
    const float* p = partial + idx;
    int w = wv;
    for (; w + 112 < n_wg; w += 128) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(w + 16 * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; w < n_wg; w += 16) s += p[(int64_t)w * stride];
  }
  red[wv][lane] = s;
  __syncthreads();
  if (wv != 0 || idx >= total) return;
  s = red[0][lane];
#pragma unroll
  for (int j = 1; j < 16; ++j) s += red[j][lane];
  int r = idx;
  if (r < 3 * FH * FH) { o.dW[r / (FH * FH)][r % (FH * FH)] = s; return; }
  r -= 3 * FH * FH;
  if (r < 3 * FH) { o.db[r / FH][r % FH] = s; return; }
  r -= 3 * FH;
  if (r < FH * E) { o.dWo[r] = s; return; }
  r -= FH * E;
  o.dbo[r] = s;
}

int edge_fused_bwd(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src,
                   const float* d_eff, const float* centers, float gap, const float* const* W,
                   const float* z_save, const float* de, float* const* dW, float* const* db, int tape_layout,
                   LiveEdges live) {
  const int64_t ntiles = cdiv(n_edges, FTM);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu);
  const int stride = (bwd_part_floats(E) + 3) / 4 * 4;
  const size_t pk_floats = (size_t)3 * FH * FH;
  const int nseg = edge_bwd_h2_segments(n_edges);      // launches of the split-operand kernel (1 below 8.4 M edges)
  // W^T fragments of the f32-input kernel (the strict-fp32 path, or the range fallback of the split-operand kernel): a
  // cached image while the weights are frozen / refreshed behind Adam (pack_bodies.cuh), else packed into the scratch
  bool haveT = false;
  float* WpkT = (float*)cached_image(ctx, W[0], 12, pk_floats * 4, &haveT);
  const bool cachedT = WpkT != nullptr;
  float* ws = (float*)workspace(ctx, (pk_floats + (size_t)nseg * grid * stride) * 4 + edge_bwd_h2_ws_bytes());
  if (!ws) return NG_ERR_NOMEM;
  if (!cachedT) WpkT = ws;
  float* partial = ws + pk_floats;
  if (!haveT) {
    const PackJob j = edge_fused_pack_job(W, nullptr, WpkT);
    if (int rc = pack_launch(ctx, st, j)) return rc;
    if (cachedT) cache_set_job(ctx, W[0], 12, j);
  }
  EdgeBwdArgs a;
  a.n_edges = n_edges; a.d_src = d_src; a.d_eff = d_eff; a.centers = centers;
  a.neg_inv_gap = (float)(-1.0 / (double)gap);
  a.WpkT = WpkT; a.Wo = W[3]; a.z_save = z_save; a.de = de;
  a.partial = partial; a.part_stride = stride;
  a.tape_blocked = 0; a.zero_rows = 0; a.guard = RangeGuard{nullptr, 0};
  a.perm = live.perm; a.n_live = live.n_live; a.z_layer_stride = n_edges * FH;
  const size_t lds = (size_t)(3 * FTM * FLD + FH * FMAX_E + FTM * FMAX_E + 2 * FTM + FH) * 4;
  // default: split-operand kernel on the 16-bit matrix pipe (edge_bwd_h2.hip); NG_EDGE_MATH=fp32 (both directions) or
  // NG_EDGE_BWD_MATH=fp32 (this one only) select the f32-input MFMA kernel below
  // tape_layout: what the forward that wrote z_save reported (ng_edge_tape_layout), -1 = decide as the forward would
  // now.  A blocked tape can only be read by the split-operand kernel; that kernel reads row-major tapes as well.
  const bool blocked = tape_layout < 0 ? edge_tape_blocked(E, n_edges) : tape_layout == 1;
  if (blocked && !edge_bwd_h2_supported(E, n_edges)) return fail(ctx, NG_ERR_INVALID, "edge_mlp_bwd: blocked tape for an unsupported shape");
  int n_part = grid;
  const bool split = blocked || edge_tape_blocked(E, n_edges);
  if (split) {
    const RangeGuard guard = range_guard_begin(ctx);
    if (!guard.word) return NG_ERR_NOMEM;
    int rc3 = edge_bwd_h2_launch(ctx, st, n_edges, E, d_src, d_eff, centers, gap, W, z_save, de,
                     (char*)(partial + (size_t)nseg * grid * stride), partial, stride, grid, blocked ? 1 : 0, guard, live);
    if (rc3) return rc3;
    n_part = nseg * grid;
    // range fallback: the f32-input kernel below, executed only if the split-operand kernel raised the guard; it
    // rewrites the same partial rows (and clears the rows of further segments)
    a.tape_blocked = blocked ? 1 : 0; a.zero_rows = n_part; a.guard = guard;
  }
  {
    ProfScope ps(ctx, st, split ? "edge_bwd_range_fallback" : "edge_fused_bwd");
#define NG_BW(EE)                                                                                   \
  case EE:                                                                                          \
    hipLaunchKernelGGL((edge_fused_bwd_kernel<EE>), dim3(grid), dim3(BW_THREADS), lds, st, a);      \
    break;
    switch (E) { NG_BW(1) NG_BW(2) NG_BW(3) NG_BW(4) NG_BW(5) NG_BW(6) NG_BW(7) NG_BW(8) }
#undef NG_BW
    NG_HIP(ctx, hipGetLastError());
  }
  {
    ProfScope ps(ctx, st, "edge_bwd_reduce");
    BwdOut o;
    for (int l = 0; l < 3; ++l) { o.dW[l] = dW[l]; o.db[l] = db[l]; }
    o.dWo = dW[3]; o.dbo = db[3];
    hipLaunchKernelGGL(edge_bwd_reduce_kernel, dim3((unsigned)cdiv(bwd_part_floats(E), 64)), dim3(1024), 0, st,
                       partial, n_part, stride, E, o, live.n_live);
    NG_HIP(ctx, hipGetLastError());
  }
  return NG_OK;
}

}  // namespace ng
