// Molecule-sized inference (one protein frame per call — the reference's own use, nmrgnn/main.py:236-245), round 3.
//
// At N ~ 3k atoms every kernel of the layered path does 1-5 us of work behind ~6 us of dependent-launch latency, and the
// forward is a chain of ~24 launches (tools/graph_frame.py).  The lever is launch COUNT.  This file fuses the row-local
// tail of GNNModel.call — the FC block (model.py:191-196: three residual Dense layers F -> F, one Dense F -> F/2,
// softplus) and the head (model.py:268-273: Dense F/2 -> C, per-element de-standardisation, one-hot select) — into ONE
// launch: a workgroup owns 32 atom rows from the MP block's output to the chemical shift, activations never leave
// the CU.  (Nine launches of the layered path: four GEMMs, their four range-fallback launches, the head kernel.)
//
// Arithmetic = gemm_h2_short_kernel's: X as two fp16 pieces in LDS, W^T fragments (pieces of 2^8 W, the images
// the PK_GX pack job builds and the frozen-weight cache keeps) streamed from L2 three k-halves ahead, three piece products
// per multiply on v_mfma_f32_32x32x16_f16, fp32 accumulate; wave w owns output columns [64 w, 64 w + 64).
// The residual input of a layer IS the previous epilogue's output and stays in registers in the accumulator layout.
//
// Range (ng_internal.h: RangeGuard rationale): a row whose feature leaves the fp16 range of a piece (|x| >= 65504), or
// any weight doing so, arrives in the accumulators as NaN.  Such rows are recomputed inside the kernel, cooperatively
// by the whole workgroup, in plain fp32 from the fp32 residual registers and the original weights (one row at a time:
// rare and slow by design) — no second launch.
#include <algorithm>
#include <string>

#include "edge_fused.h"      // NG_LDS_BARRIER
#include "h2_common.cuh"
#include "mfma_gemm.cuh"

namespace ng {

constexpr int FFW = 256;                       // feature width handled here (the reference's default atom_feature_size)
constexpr int FF_ROWS = 32;                    // atom rows per workgroup
constexpr int FF_XROW = FFW * 2 + 16;          // bytes per row of an X piece plane (b128 rows conflict-free)
constexpr int FF_XPLANE = FF_ROWS * FF_XROW;   // 16,896
constexpr int FF_WCHUNK = 2 * 4 * 2 * 2 * 1024; // gx_wchunk(4): [n-block 8][k-step of 16: 2][piece 2][1 KB]
constexpr float FF_WSCALE = 256.0f, FF_WINV = 1.0f / 256.0f;
constexpr int FF_GLD = FFW / 2 + 4;            // row stride (floats) of the g tile [32][128]
constexpr int FF_MAXC = 16;
#ifndef FF_MAX_TILES
#define FF_MAX_TILES 512          // batches up to 16384 atoms (above: the per-layer GEMMs have enough rows to fill the chip)
#endif
#ifndef FF_RING
#define FF_RING 16
#endif
#ifndef FF_NO_SPLIT
#define FF_NO_SPLIT 0
#endif
#ifndef FF_NB
#define FF_NB 8
#endif                    // element columns the head handles

// image of a [K][n_valid] weight matrix in the layout of pk::gx_img (pack_bodies.cuh) for 256-column tiles, columns
// n_valid .. 255 zero:  lane (row n = 32 nb + (l&31), k-slot t) = piece_p(2^8 W[k = 32 kt + 16 ks + 8 (l>>5) + t][n])
__global__ void ff_pack_kernel(int K, int n_valid, const float* __restrict__ W, unsigned* __restrict__ img) {
  const int KT = K / 32;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // (kt, nb, ks, lane)
  if (idx >= KT * 8 * 2 * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) & 1, nb = (idx >> 7) & 7, kt = idx >> 10;
  const int n = 32 * nb + (lane & 31), k0 = 32 * kt + 16 * ks + 8 * (lane >> 5);
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = n < n_valid ? FF_WSCALE * W[(int64_t)(k0 + 2 * j) * n_valid + n] : 0.f;
    const float b = n < n_valid ? FF_WSCALE * W[(int64_t)(k0 + 2 * j + 1) * n_valid + n] : 0.f;
    split2_pair(a, b, h[j], l[j]);
  }
  unsigned* dst = img + (size_t)kt * (FF_WCHUNK / 4) + ((nb * 2 + ks) * 2) * 256 + lane * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) { dst[j] = h[j]; dst[256 + j] = l[j]; }
}

struct FcHeadArgs {
  int64_t N;
  int C;
  const float* x;            // [N][256] FC block input (MP block output)
  const char* img[4];        // packed weight images of the four FC layers (the last padded to 256 columns)
  const float* W[4];         // the original weights (range fallback)
  const float* b[4];
  const float* Wout;         // [128][C]
  const float* bout;         // [C]
  const float* atoms;        // [N][C]
  const float* pstd;
  const float* pavg;
  float* peaks;              // [N]
};

// acc[j] += X[32 rows][K] (pieces in LDS planes) x W-image column block cb0 + j (32 columns each), j < NJ;  K = 32 KT
template <int KT, int XROW, int NJ>
__device__ __forceinline__ void ff_gemm_n(f32x16 (&acc)[NJ], const char* __restrict__ sX, const char* __restrict__ img, int cb0,
                                          int lane) {
  constexpr int XPLANE = FF_ROWS * XROW;
  const int half = lane >> 5, l31 = lane & 31;
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img), 0, (unsigned)(KT * FF_WCHUNK), 0x00020000);
  auto w_request = [&](u32x4 (&wa)[NJ][2], int q) {       // k-half q = (32-wide step q >> 1, half q & 1)
    const int qc = q;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const auto raw = __builtin_amdgcn_raw_buffer_load_b128(
            wrs, lane * 16, (qc >> 1) * FF_WCHUNK + (((cb0 + j) * 2 + (qc & 1)) * 2 + p) * 1024, 0);
        wa[j][p] = __builtin_bit_cast(u32x4, raw);
      }
  };
  // One wave per SIMD and one workgroup per CU: nothing covers the L2 round trip of a fragment (~0.6-1 us = 3-5 k-halves
  // of MFMAs) but the wave's own requests in flight.  The kernel has 512 registers per lane to itself, so the ring is
  // FF_RING k-halves deep (64 registers per 4): with 16, a 256-deep contraction has every fragment requested up front.
  u32x4 w[FF_RING][NJ][2];
#pragma unroll
  for (int q = 0; q < FF_RING - 1; ++q) w_request(w[q], q);
#pragma unroll
  for (int q = 0; q < 2 * KT; ++q) {
    if (q + FF_RING - 1 < 2 * KT) w_request(w[(q + FF_RING - 1) % FF_RING], q + FF_RING - 1);
    u32x4 xb[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
      xb[p] = *reinterpret_cast<const u32x4*>(sX + p * XPLANE + l31 * XROW + (16 * q + 8 * half) * 2);
    if (NJ == 2) mma3_2a(w[q % FF_RING][0], w[q % FF_RING][NJ - 1], xb, acc[0], acc[NJ - 1]);
    else acc[0] = mma3(w[q % FF_RING][0], xb, acc[0]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the 64 columns 64 nq .. of a wave (two blocks): the form the FC chain uses
template <int KT, int XROW = FF_XROW>
__device__ __forceinline__ void ff_gemm(f32x16 (&acc)[2], const char* __restrict__ sX, const char* __restrict__ img, int nq,
                                        int lane) {
  ff_gemm_n<KT, XROW, 2>(acc, sX, img, 2 * nq, lane);
}

// this lane's 32 values (row l31; columns 64 nq + 32 j + 8 q + 4 half + r in v[j][4 q + r]) -> the X piece planes
__device__ __forceinline__ void ff_store_x(char* __restrict__ sX, const float (&v)[2][16], int nq, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned h0, l0, h1, l1;
      split2_pair(v[j][4 * q + 0], v[j][4 * q + 1], h0, l0);
      split2_pair(v[j][4 * q + 2], v[j][4 * q + 3], h1, l1);
      char* p = sX + l31 * FF_XROW + (64 * nq + 32 * j + 8 * q + 4 * half) * 2;
      *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(p + FF_XPLANE) = u32x2{l0, l1};
    }
}

// Rows whose accumulators are non-finite, recomputed by the whole workgroup in plain fp32 (header: Range).
//   xres: the layer's fp32 input in the accumulator layout (this lane: row l31), W: [256][n_valid] row-major.
// On return acc holds (x W)[row][col] / FF_WINV for the repaired rows, untouched elsewhere.
__device__ __forceinline__ void ff_repair_rows(f32x16 (&acc)[2], const float (&xres)[2][16], const float* __restrict__ W, int n_valid,
                                            float* __restrict__ sRow, unsigned* __restrict__ sMask, int tid) {
  const int lane = tid & 63, nq = tid >> 6, half = lane >> 5, l31 = lane & 31;
  float chk = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < 16; ++t) chk += fabsf(acc[j][t]);
  const bool bad = not_finite(chk);
  if (tid == 0) *sMask = 0u;
  __syncthreads();
  if (bad) atomicOr(sMask, 1u << l31);
  __syncthreads();
  const unsigned mask = *sMask;
  if (mask == 0u) return;                      // the common case: two barriers, nothing else
  float* sX = sRow;                            // [256] the row's fp32 input
  float* sY = sRow + FFW;                      // [256] its outputs
  for (int r = 0; r < 32; ++r) {
    if (!((mask >> r) & 1u)) continue;
    if (l31 == r) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) sX[64 * nq + 32 * j + 8 * (t >> 2) + 4 * half + (t & 3)] = xres[j][t];
    }
    __syncthreads();
    {
      float s = 0.f;
      if (tid < n_valid)
        for (int k = 0; k < FFW; ++k) s = fmaf(sX[k], W[(int64_t)k * n_valid + tid], s);
      sY[tid] = s;
    }
    __syncthreads();
    if (l31 == r) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[j][t] = sY[64 * nq + 32 * j + 8 * (t >> 2) + 4 * half + (t & 3)] * FF_WSCALE;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256, 1) void fc_head_short_kernel(FcHeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_ff[];
  char* sX = smem_ff;                                                    // [2 pieces][32][FF_XROW]
  float* sG = reinterpret_cast<float*>(smem_ff + 2 * FF_XPLANE);         // [32][FF_GLD] last FC layer's output
  float* sRow = sG + FF_ROWS * FF_GLD;                                   // [512] range-repair row buffers
  unsigned* sMask = reinterpret_cast<unsigned*>(sRow + 2 * FFW);
  float* sWout = reinterpret_cast<float*>(sMask + 4);                    // [128][C] | bout[C] | std[C] | avg[C]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int nq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t m0 = (int64_t)blockIdx.x * FF_ROWS;
  const int64_t m = std::min<int64_t>(m0 + l31, a.N - 1);                // clamped: rows past the end compute and drop

  // this lane's slice of the input row, in the accumulator layout; also the first X operand
  float xr[2][16];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(a.x + m * FFW + 64 * nq + 32 * j + 8 * q + 4 * half);
      xr[j][4 * q + 0] = v.x; xr[j][4 * q + 1] = v.y; xr[j][4 * q + 2] = v.z; xr[j][4 * q + 3] = v.w;
    }
  ff_store_x(sX, xr, nq, lane);
  // the head's operands: in LDS from the start (read from global inside the head's loop they cost a memory round trip
  // per feature: 16 us of a 45-us kernel)
  for (int t = tid; t < (FFW / 2) * a.C; t += 256) sWout[t] = a.Wout[t];
  if (tid < a.C) {
    sWout[(FFW / 2) * a.C + tid] = a.bout[tid];
    sWout[(FFW / 2) * a.C + a.C + tid] = a.pstd[tid];
    sWout[(FFW / 2) * a.C + 2 * a.C + tid] = a.pavg[tid];
  }
  NG_LDS_BARRIER();

  // a real loop: fully unrolled, the four layers are ~60 KB of straight-line code that every workgroup executes exactly
  // once — the kernel then runs at the speed of instruction fetch (44 us for ~10 us of work)
#pragma unroll 1
  for (int l = 0; l < 4; ++l) {
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // this lane's bias values, requested in front of the GEMM (the last layer has 128 columns: clamped, unused beyond)
    float4 bias[2][4];
    {
      const int ncol = l < 3 ? FFW : FFW / 2;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          bias[j][q] = *reinterpret_cast<const float4*>(a.b[l] + std::min(64 * nq + 32 * j + 8 * q + 4 * half, ncol - 4));
    }
#ifndef FF_SKIP_GEMM
    ff_gemm<FFW / 32>(acc, sX, a.img[l], nq, lane);
#endif
    ff_repair_rows(acc, xr, a.W[l], l < 3 ? FFW : FFW / 2, sRow, sMask, tid);     // (also the barrier behind the X reads)
    if (l < 3) {
      // x <- softplus(x W + b) + x, kept in registers and re-split into the planes for the next layer
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bv = bias[j][q];
          const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
#ifdef FF_SKIP_ACT
          for (int r = 0; r < 4; ++r) xr[j][4 * q + r] += fmaf(acc[j][4 * q + r], FF_WINV, bb[r]);
#else
          for (int r = 0; r < 4; ++r) xr[j][4 * q + r] += softplus_f(fmaf(acc[j][4 * q + r], FF_WINV, bb[r]));
#endif
        }
      ff_store_x(sX, xr, nq, lane);
      NG_LDS_BARRIER();
    } else if (nq < 2) {
      // g = softplus(x W + b): 128 columns, held by waves 0 and 1 (the image's upper 128 columns are zero)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = 64 * nq + 32 * j + 8 * q + 4 * half;
          const float4 bv = bias[j][q];
          *reinterpret_cast<float4*>(sG + l31 * FF_GLD + c0) =
              make_float4(softplus_f(fmaf(acc[j][4 * q + 0], FF_WINV, bv.x)), softplus_f(fmaf(acc[j][4 * q + 1], FF_WINV, bv.y)),
                          softplus_f(fmaf(acc[j][4 * q + 2], FF_WINV, bv.z)), softplus_f(fmaf(acc[j][4 * q + 3], FF_WINV, bv.w)));
        }
    }
  }
  __syncthreads();
  // head: full = g Wout + bout; peaks = sum_c atoms[c] (full[c] std[c] + avg[c])     (8 lanes per row, 16 features each)
  {
    const int r = tid >> 3, part = tid & 7;
    const int64_t row = m0 + r;
    float full[FF_MAXC];
#pragma unroll
    for (int c = 0; c < FF_MAXC; ++c) full[c] = 0.f;
#pragma unroll 4
    for (int f = 16 * part; f < 16 * part + 16; ++f) {
      const float g = sG[r * FF_GLD + f];
#pragma unroll
      for (int c = 0; c < FF_MAXC; ++c)
        if (c < a.C) full[c] = fmaf(g, sWout[f * a.C + c], full[c]);
    }
#pragma unroll
    for (int c = 0; c < FF_MAXC; ++c) {
      float v = full[c];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
      full[c] = v;
    }
    if (part == 0 && row < a.N) {
      float pk = 0.f;
#pragma unroll
      for (int c = 0; c < FF_MAXC; ++c)
        if (c < a.C) {
          const float* hp = sWout + (FFW / 2) * a.C;
          pk = fmaf(a.atoms[row * a.C + c], fmaf(full[c] + hp[c], hp[a.C + c], hp[2 * a.C + c]), pk);
        }
      a.peaks[row] = pk;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// One MPLayer (nmrgnn/layers.py:26-46) for a molecule-sized graph in ONE launch: the neighbour aggregate
// A[i, n*F + l] = sum_j e[i,j,n] h[nlist[i,j], l] of a workgroup's 32 atoms goes straight into the X piece planes in LDS
// (it never exists in HBM), then h_out = act(inv_degree * A Wp) (+ h) with Wp[k = n*F + l][m] = w[l][m][n] streamed as
// fp16-piece fragments from L2.  Replaces aggregate + GEMM + range-fallback launch of the layered path.
struct MpShortArgs {
  int64_t N;
  int K, act, residual;
  const float* h;            // [N][256]
  const int32_t* row_ptr;    // CSR form: [N+1] (then nlist = col [nnz], e [nnz][E], K unused); nullptr: padded lists
  const int32_t* nlist;      // [N][K]
  const float* e;            // [N][K][E]
  const float* inv_degree;   // [N]
  const char* img;           // packed image of Wp (ff_pack_mp_kernel)
  const float* w;            // [256][256][E] the layer's weight (range fallback)
  float* h_out;              // [N][256]
};

// image of Wp[k = n*256 + l][m] = w[l][m][n] in the layout of ff_pack_kernel
__global__ void ff_pack_mp_kernel(int E, const float* __restrict__ w, unsigned* __restrict__ img) {
  const int KT = E * FFW / 32;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // (kt, nb, ks, lane)
  if (idx >= KT * 8 * 2 * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) & 1, nb = (idx >> 7) & 7, kt = idx >> 10;
  const int m = 32 * nb + (lane & 31), k0 = 32 * kt + 16 * ks + 8 * (lane >> 5);
  auto wp = [&](int k) { return FF_WSCALE * w[((int64_t)(k % FFW) * FFW + m) * E + k / FFW]; };
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split2_pair(wp(k0 + 2 * j), wp(k0 + 2 * j + 1), h[j], l[j]);
  unsigned* dst = img + (size_t)kt * (FF_WCHUNK / 4) + ((nb * 2 + ks) * 2) * 256 + lane * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) { dst[j] = h[j]; dst[256 + j] = l[j]; }
}

// NS = 2: TWO workgroups per 32-atom tile (blockIdx.y), each with 128 of the 256 output columns — wave nq owns ONE block of 32
// columns.  Both gather the tile's aggregate (the gather is a chain of L2 round trips, not bandwidth), each streams half of
// the weight image: for a single molecule (2770 atoms = 87 tiles on 256 CUs) the image stream per workgroup set the pace.
template <int E, bool CSR, int NS>
__global__ __launch_bounds__(256, 1) void mp_layer_short_kernel(MpShortArgs a) {
  constexpr int NJ = 2 / NS;                     // 32-column blocks per wave
  constexpr int KF = E * FFW, XROW = KF * 2 + 16, XPLANE = FF_ROWS * XROW;
  extern __shared__ __attribute__((aligned(16))) char smem_mp[];
  char* sX = smem_mp;                                                    // [2][32][XROW]
  float* sRow = reinterpret_cast<float*>(smem_mp + 2 * XPLANE);          // [KF + 256] range-repair buffers
  unsigned* sMask = reinterpret_cast<unsigned*>(sRow + KF + FFW);
  int32_t* s_nl = reinterpret_cast<int32_t*>(sMask + 4);                 // [32][K]
  float* s_e = reinterpret_cast<float*>(s_nl + FF_ROWS * a.K);           // [32][K][E]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int nq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb0 = NS == 1 ? 2 * nq : 4 * (int)blockIdx.y + nq;      // first 32-column block of this wave
  const int K = a.K;
  const int64_t m0 = (int64_t)blockIdx.x * FF_ROWS;
  const int n_at = (int)std::min<int64_t>(FF_ROWS, a.N - m0);
  const int64_t m = std::min<int64_t>(m0 + l31, a.N - 1);

  // lists of the tile (rows past the end: neighbour 0 with weight 0); the CSR form reads its entries from global memory
  if (!CSR) {
    for (int t = tid; t < FF_ROWS * K; t += 256) s_nl[t] = t < n_at * K ? a.nlist[m0 * K + t] : 0;
    for (int t = tid; t < FF_ROWS * K * E; t += 256) s_e[t] = t < n_at * K * E ? a.e[m0 * K * E + t] : 0.f;
  }
  // this lane's slice of its own row in the accumulator layout: the residual term (requested early)
  float xr[NJ][16];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(a.h + m * FFW + 32 * (cb0 + j) + 8 * q + 4 * half);
      xr[j][4 * q + 0] = v.x; xr[j][4 * q + 1] = v.y; xr[j][4 * q + 2] = v.z; xr[j][4 * q + 3] = v.w;
    }
  const float rs = a.inv_degree[m];
  __syncthreads();

  // ---- aggregate: 8 threads per atom row, 32 feature columns each, neighbours in entry order (fmaf chain)
  {
    const int r = tid >> 3, cc = tid & 7;
    float acc[E][32];
#pragma unroll
    for (int n = 0; n < E; ++n)
#pragma unroll
      for (int c = 0; c < 32; ++c) acc[n][c] = 0.f;
    const float4* h4 = reinterpret_cast<const float4*>(a.h);
    // entry range of this row: [p0, p1) of the flat lists (padded form: r*K .. r*K + K of the staged tile)
    int p0 = r * K, p1 = r * K + K;
    if (CSR) {
      p0 = r < n_at ? a.row_ptr[m0 + r] : 0;
      p1 = r < n_at ? a.row_ptr[m0 + r + 1] : 0;
    }
    const int32_t* nl = CSR ? a.nlist : s_nl;
    const float* ee = CSR ? a.e : s_e;
    // FF_NB neighbour rows (8 x 16 B each) requested before the first is consumed: the phase is a chain of L2 round trips
    // with one wave per SIMD, and the kernel has the registers (512 per lane) to keep 56-64 loads in flight
#ifdef FF_SKIP_AGG
    for (int j0 = p0; j0 < p0; j0 += FF_NB) {
#else
    for (int j0 = p0; j0 < p1; j0 += FF_NB) {
#endif
      float4 hv[FF_NB][8];
#pragma unroll
      for (int u = 0; u < FF_NB; ++u) {
        const int j = j0 + u < p1 ? j0 + u : p1 - 1;
        // lane cc takes the float4 columns cc, cc + 8, ..: the eight lanes of a row read 128 contiguous bytes per load
        // instruction (contiguous 32-column slices per lane made every instruction touch 64 different cache lines)
        const int64_t base = (int64_t)nl[j] * (FFW / 4) + cc;
#pragma unroll
        for (int v = 0; v < 8; ++v) hv[u][v] = h4[base + 8 * v];
      }
#pragma unroll
      for (int u = 0; u < FF_NB; ++u) {
        if (j0 + u < p1) {
#pragma unroll
          for (int n = 0; n < E; ++n) {
            const float ev = ee[(int64_t)(j0 + u) * E + n];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
              acc[n][4 * v + 0] = fmaf(ev, hv[u][v].x, acc[n][4 * v + 0]);
              acc[n][4 * v + 1] = fmaf(ev, hv[u][v].y, acc[n][4 * v + 1]);
              acc[n][4 * v + 2] = fmaf(ev, hv[u][v].z, acc[n][4 * v + 2]);
              acc[n][4 * v + 3] = fmaf(ev, hv[u][v].w, acc[n][4 * v + 3]);
            }
          }
        }
      }
    }
    // two fp16 pieces of the row slice -> the X planes (acc[n][4 v + i] is k = n*256 + 32 v + 4 cc + i)
#pragma unroll
    for (int n = 0; n < E; ++n)
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        unsigned h0, l0, h1, l1;
        split2_pair(acc[n][4 * v + 0], acc[n][4 * v + 1], h0, l0);
        split2_pair(acc[n][4 * v + 2], acc[n][4 * v + 3], h1, l1);
        char* dst = sX + r * XROW + (n * FFW + 32 * v + 4 * cc) * 2;
        *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(dst + XPLANE) = u32x2{l0, l1};
      }
  }
  NG_LDS_BARRIER();

  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#ifndef FF_SKIP_GEMM
  ff_gemm_n<KF / 32, XROW, NJ>(acc, sX, a.img, cb0, lane);
#endif

  // ---- range repair (file header): rows with non-finite accumulators, recomputed by the whole workgroup in fp32
  {
    float chk = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) chk += fabsf(acc[j][t]);
    if (tid == 0) *sMask = 0u;
    __syncthreads();
    if (not_finite(chk)) atomicOr(sMask, 1u << l31);
    __syncthreads();
    const unsigned mask = *sMask;
    if (mask != 0u) {
      float* sA = sRow;             // [KF] the row's aggregate
      float* sY = sRow + KF;        // [256] its pre-activations
      const int32_t* nl = CSR ? a.nlist : s_nl;
      const float* ee = CSR ? a.e : s_e;
      for (int r = 0; r < FF_ROWS; ++r) {
        if (!((mask >> r) & 1u)) continue;
        const int q0 = CSR ? a.row_ptr[m0 + r] : r * K, q1 = CSR ? a.row_ptr[m0 + r + 1] : r * K + K;   // (bad rows are < n_at)
        for (int n = 0; n < E; ++n) {
          float s = 0.f;
          for (int j = q0; j < q1; ++j) s = fmaf(ee[(int64_t)j * E + n], a.h[(int64_t)nl[j] * FFW + tid], s);
          sA[n * FFW + tid] = s;
        }
        __syncthreads();
        {
          float s = 0.f;
          for (int k = 0; k < KF; ++k) s = fmaf(sA[k], a.w[((int64_t)(k % FFW) * FFW + tid) * E + k / FFW], s);
          sY[tid] = s;
        }
        __syncthreads();
        if (l31 == r) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[j][t] = sY[32 * (cb0 + j) + 8 * (t >> 2) + 4 * half + (t & 3)] * FF_WSCALE;
        }
        __syncthreads();
      }
    }
  }

  // ---- epilogue: h_out = act(inv_degree * A Wp) (+ h)
  if (m0 + l31 < a.N) {
    const float sc = rs * FF_WINV;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 y;
        y.x = act_apply(a.act, acc[j][4 * q + 0] * sc); y.y = act_apply(a.act, acc[j][4 * q + 1] * sc);
        y.z = act_apply(a.act, acc[j][4 * q + 2] * sc); y.w = act_apply(a.act, acc[j][4 * q + 3] * sc);
        if (a.residual) { y.x += xr[j][4 * q + 0]; y.y += xr[j][4 * q + 1]; y.z += xr[j][4 * q + 2]; y.w += xr[j][4 * q + 3]; }
        *reinterpret_cast<float4*>(a.h_out + m * FFW + 32 * (cb0 + j) + 8 * q + 4 * half) = y;
      }
  }
}

bool mp_layer_short_supported(int64_t N, int K, int F, int E) {
  if (sw().mp_layered || sw().gemm_math_fp32) return false;       // the any-shape / strict-fp32 paths were asked for
  return F == FFW && E >= 1 && E <= 3 && K >= 1 && K <= 32 && N > 0 && N <= (int64_t)FF_ROWS * FF_MAX_TILES;
}

bool fc_head_short_supported(int64_t N, int F, int L, int C, int act) {
  if (sw().fc_layered || sw().gemm_math_fp32 || sw().head_generic) return false;
  return F == FFW && L == 4 && C <= FF_MAXC && act == NG_ACT_SOFTPLUS && N > 0 && N <= (int64_t)FF_ROWS * FF_MAX_TILES;
}

}  // namespace ng

using namespace ng;

extern "C" int ng_fc_head_fwd(ng_ctx* ctx, void* stream, int64_t N, int F, int L, int C, int act, const float* x,
                              const float* const* W, const float* const* b, const float* Wout, const float* bout,
                              const float* atoms, const float* pstd, const float* pavg, float* peaks) {
  if (!ctx) return NG_ERR_INVALID;
  if (!fc_head_short_supported(N, F, L, C, act)) return NG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard dg(ctx->device);
  FcHeadArgs a;
  a.N = N; a.C = C; a.x = x; a.Wout = Wout; a.bout = bout; a.atoms = atoms; a.pstd = pstd; a.pavg = pavg; a.peaks = peaks;
  const size_t img_bytes = (size_t)(FFW / 32) * FF_WCHUNK;
  char* scratch = nullptr;
  for (int l = 0; l < 4; ++l) {
    a.W[l] = W[l]; a.b[l] = b[l];
    bool have = false;
    char* img = (char*)cached_image(ctx, W[l], 9, img_bytes, &have);
    if (!img) {                     // weights not frozen: images into the aux scratch, packed on every call
      if (!scratch) scratch = (char*)aux_workspace(ctx, 4 * img_bytes);
      if (!scratch) return NG_ERR_NOMEM;
      img = scratch + l * img_bytes;
    }
    if (!have) {
      hipLaunchKernelGGL(ff_pack_kernel, dim3((FFW / 32) * 8 * 2 * 64 / 256), dim3(256), 0, st, FFW, l < 3 ? FFW : FFW / 2, W[l],
                         (unsigned*)img);
      NG_HIP(ctx, hipGetLastError());
    }
    a.img[l] = img;
  }
  ProfScope ps(ctx, st, "fc_head_short");
  const size_t lds = (size_t)2 * FF_XPLANE + (size_t)(FF_ROWS * FF_GLD + 2 * FFW + 4 + (FFW / 2 + 3) * FF_MAXC) * 4;
  hipLaunchKernelGGL(fc_head_short_kernel, dim3((unsigned)cdiv(N, FF_ROWS)), dim3(256), lds, st, a);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

static int mp_layer_short_launch(ng_ctx* ctx, void* stream, int64_t N, int K, int F, int E, int act, int residual,
                                 const float* h, const int32_t* row_ptr, const int32_t* nlist, const float* e,
                                 const float* inv_degree, const float* w, float* h_out) {
  if (!ctx) return NG_ERR_INVALID;
  if (!mp_layer_short_supported(N, K, F, E)) return NG_ERR_UNSUPPORTED;
  NG_REQUIRE(ctx, h != h_out, "mp_layer_fwd_short: in-place update not supported (other atoms still gather the input)");
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard dg(ctx->device);
  const int KF = E * FFW;
  const size_t img_bytes = (size_t)(KF / 32) * FF_WCHUNK;
  bool have = false;
  char* img = (char*)cached_image(ctx, w, 10, img_bytes, &have);
  if (!img) img = (char*)aux_workspace(ctx, img_bytes);
  if (!img) return NG_ERR_NOMEM;
  if (!have) {
    hipLaunchKernelGGL(ff_pack_mp_kernel, dim3((KF / 32) * 8 * 2 * 64 / 256), dim3(256), 0, st, E, w, (unsigned*)img);
    NG_HIP(ctx, hipGetLastError());
  }
  MpShortArgs a;
  a.N = N; a.K = K; a.act = act; a.residual = residual; a.h = h; a.row_ptr = row_ptr; a.nlist = nlist; a.e = e; a.inv_degree = inv_degree;
  a.img = img; a.w = w; a.h_out = h_out;
  const size_t lds = (size_t)2 * FF_ROWS * (KF * 2 + 16) + (size_t)(KF + FFW + 4) * 4 + (row_ptr ? 0 : (size_t)FF_ROWS * K * (1 + E) * 4);
  ProfScope ps(ctx, st, "mp_layer_short");
  // two workgroups per tile (128 output columns each) while that still fits the chip in one round
  const int64_t tiles = cdiv(N, FF_ROWS);
  const bool split = 2 * tiles <= ctx->num_cu && !FF_NO_SPLIT;
  const dim3 grid((unsigned)tiles, split ? 2u : 1u);
#define NG_MPS(EE, CC)                                                                                               \
  if (split) hipLaunchKernelGGL((mp_layer_short_kernel<EE, CC, 2>), grid, dim3(256), lds, st, a);                     \
  else hipLaunchKernelGGL((mp_layer_short_kernel<EE, CC, 1>), grid, dim3(256), lds, st, a);
  switch (E) {
    case 1: if (row_ptr) { NG_MPS(1, true) } else { NG_MPS(1, false) } break;
    case 2: if (row_ptr) { NG_MPS(2, true) } else { NG_MPS(2, false) } break;
    default: if (row_ptr) { NG_MPS(3, true) } else { NG_MPS(3, false) } break;
  }
#undef NG_MPS
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_mp_layer_fwd_short(ng_ctx* ctx, void* stream, int64_t N, int K, int F, int E, int act, int residual,
                                     const float* h, const int32_t* nlist, const float* e, const float* inv_degree,
                                     const float* w, float* h_out) {
  return mp_layer_short_launch(ctx, stream, N, K, F, E, act, residual, h, nullptr, nlist, e, inv_degree, w, h_out);
}

// the same over CSR lists (row_ptr [N+1], col [nnz], e [nnz][E]): any degree
extern "C" int ng_mp_layer_fwd_short_csr(ng_ctx* ctx, void* stream, int64_t N, int F, int E, int act, int residual,
                                         const float* h, const int32_t* row_ptr, const int32_t* col, const float* e,
                                         const float* inv_degree, const float* w, float* h_out) {
  if (ctx && !row_ptr) return fail(ctx, NG_ERR_INVALID, "mp_layer_fwd_short_csr: row_ptr required");
  return mp_layer_short_launch(ctx, stream, N, 1, F, E, act, residual, h, row_ptr, col, e, inv_degree, w, h_out);
}

/* 1 when the fused molecule-sized kernels take this shape under the path switches in force (NG_MP_PATH / NG_FC_PATH =
 * layered, NG_GEMM_MATH = fp32 and NG_HEAD_PATH = generic select the per-operation kernels) */
extern "C" int ng_mp_layer_short_ok(int64_t N, int K, int F, int E) { return mp_layer_short_supported(N, K, F, E) ? 1 : 0; }
extern "C" int ng_fc_head_ok(int64_t N, int F, int L, int C, int act) { return fc_head_short_supported(N, F, L, C, act) ? 1 : 0; }
