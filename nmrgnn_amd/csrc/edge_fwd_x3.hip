// Fused edge forward on the bf16 matrix pipe with fp32-equivalent arithmetic (default; NG_EDGE_MATH=fp32 opts out).
// Reference: nmrgnn/model.py:251-261 + nmrgnn/layers.py:137-140 + nmrgnn/model.py:132-138 (same math as
// edge_fused.hip).
//
// Why.  On gfx950 the f32-input MFMA runs at the fp32 VECTOR rate and shares the VALU's issue: PMC shows
// SQ_VALU_MFMA_COEXEC_CYCLES = 0 for edge_fused_fwd/bwd and a 72 % busy matrix pipe whose idle share is the
// kernels' VALU + LDS issue time (profiles/pmc_mfma.json).  The bf16 MFMA is a separate pipe, 16x faster per flop.  Every fp32 operand x is split exactly into three bf16 pieces x = h + m + l (8 mantissa bits
// each; h = rne(x), m = rne(x-h), l = rne(x-h-m); the subtractions are exact) and a product a*b is formed from
// the six piece products whose weight is >= 2^-16 of |a||b|:  hh + hm + mh + hl + lh + mm, each exact in the
// fp32 accumulator.  What is dropped (ml + lm + ll and the split residuals) is <= 4 * 2^-24 |a||b| — the size of
// ONE fp32 rounding of the product — so the result carries fp32-level error, at 16/6 = 2.7x the fp32 matrix rate.
//
// Mapping.  512 threads = 8 waves, one persistent workgroup per CU, 256 edges per tile; wave w owns edges
// [32w, 32w+32) through ALL layers, so activations never leave registers:
//   v_mfma_f32_32x32x16_bf16  D[feature][edge] += A[feature][k] * B[k][edge]
//   A = weight pieces, read from LDS (ds_read_b128 of a lane-linear fragment image);
//   B = activation pieces in VGPRs: lane (edge = l&31, k-slots 8*(l>>5)..+7);
//   D: lane (edge = l&31) holds features (r&3) + 8*(r>>2) + 4*(l>>5) of the 32-feature block, r = 0..15.
// The contraction index of the NEXT layer is permuted so that a lane's own 16 outputs of a block are exactly its
// k-slots of two k-steps (step s takes registers 8s..8s+7): softplus -> split -> pack turns the accumulators
// into the next B operand in place, with no LDS round trip, no transposition and no barrier for activations.
// The weight image is packed once per call with the same permutation (x3_pack_kernel).
//
// Weights reach LDS by LDS-DMA in 48-KB chunks (half a layer: 2 output blocks x 4 input blocks x 2 k-steps x
// 3 pieces x 1 KB fragments), 7 chunks per tile (3 layers x 2 + output layer), through a ring of two slots:
// while chunk c is consumed, c+1 lands; one barrier per chunk.  L2 -> CU traffic is
// 336 KB per 256 edges (the fp32 kernel: 192 KB per 64).
#include <algorithm>
#include <string>

#include "edge_fused.h"
#include "x3_common.cuh"

namespace ng {

constexpr int X3_TM = 256;
constexpr int X3_CHUNK = 48 * 1024;
constexpr int X3_NCHUNK = 7;
constexpr int X3_RING = 2 * X3_CHUNK;
constexpr int X3_TLD = 36;                 // row stride (floats) of a wave's 32 x 32 transposition tile
constexpr int X3_TBYTES = 8 * 32 * X3_TLD * 4;

// feature (within a 32-block) held in k-slot t (0..7) of k-step s by lane half hf  ==  accumulator register 8s+t
__host__ __device__ inline int x3_feat(int t, int s, int hf) { return (t & 3) + 16 * s + 8 * (t >> 2) + 4 * hf; }

// Weight image: chunk c (0..5): layer c>>1, output blocks 2*(c&1) + {0,1};  chunk 6: output layer (rows >= E zero).
//   fragment ((bo_l*4 + bi)*2 + s)*3 + p, 1 KB each, lane-linear 16 B per lane:
//   lane (row i = l&31, k-slot t) = piece_p( W[k = 32 bi + x3_feat(t, s, l>>5)][n = 32 bo + i] )
__global__ void x3_pack_kernel(const float* __restrict__ W0, const float* __restrict__ W1,
                               const float* __restrict__ W2, const float* __restrict__ Wo, int E,
                               unsigned* __restrict__ img) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (chunk, bo_l, bi, s, lane)
  if (idx >= X3_NCHUNK * 16 * 64) return;
  const int lane = idx & 63, s = (idx >> 6) & 1, bi = (idx >> 7) & 3, bo_l = (idx >> 9) & 1, c = idx >> 10;
  const int i = lane & 31, hf = lane >> 5;
  float v[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = 32 * bi + x3_feat(t, s, hf);
    if (c < 6) {
      const float* W = (c >> 1) == 0 ? W0 : ((c >> 1) == 1 ? W1 : W2);
      v[t] = W[k * FH + 32 * (2 * (c & 1) + bo_l) + i];
    } else {
      v[t] = (bo_l == 0 && i < E) ? Wo[k * E + i] : 0.f;
    }
  }
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split3_pair(v[2 * j], v[2 * j + 1], h[j], m[j], l[j]);
  const int frag = ((bo_l * 4 + bi) * 2 + s) * 3;
  unsigned* dst = img + (size_t)c * (X3_CHUNK / 4) + (size_t)frag * 256 + lane * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) { dst[j] = h[j]; dst[256 + j] = m[j]; dst[512 + j] = l[j]; }
}

struct EdgeX3Args {
  int64_t n_edges;
  const float* d_src;
  const float* d_eff;
  const float* centers;
  float neg_inv_gap_log2e;
  const char* img;        // [7][48 KB]
  const float* bh[3];
  const float* bo;
  int E;
  float* e_out;
  float* z_save;          // [3][n_edges][128] or nullptr
  float* dummy;           // 128 floats
};

// one chunk = 48 wave-instructions of 1 KB; wave w moves KB w, w+8, ..  Buffer form: scalar resource + scalar
// offset + ONE lane-offset VGPR (per-lane 64-bit addresses for 6 x 7 positions get hoisted and spilled)
__device__ __forceinline__ void x3_dma_chunk(__amdgpu_buffer_rsrc_t rsrc, int cid, char* slot, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int kb = wave + 8 * j;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(slot + kb * 1024), 16,
                                         lane * 16, cid * X3_CHUNK + kb * 1024, 0, 0);
  }
}

__device__ __forceinline__ f32x16 x3_bias(const float* __restrict__ sb, int bo, int hf);

// six piece products (smallest terms first) for two output blocks that share the B fragments; the two
// accumulator chains alternate so that consecutive MFMAs are independent.  (Accumulating all small products of a
// chunk before the leading ones — two passes over the weight fragments — halves the maximum error and costs 8 %;
// the single pass already carries less error than the f32-input MFMA chain: tests/test_gpu_edge_x3.py.)
#define X3_MM(pa, pb)                          \
  acc0 = mfma_bf16(a0[pa], b[pb], acc0);       \
  acc1 = mfma_bf16(a1[pa], b[pb], acc1);
__device__ __forceinline__ void mma6x2(const u32x4 (&a0)[3], const u32x4 (&a1)[3], const u32x4 (&b)[3], f32x16& acc0,
                                       f32x16& acc1) {
  X3_MM(2, 0) X3_MM(0, 2) X3_MM(1, 1) X3_MM(1, 0) X3_MM(0, 1) X3_MM(0, 0)
}

__device__ __forceinline__ void x3_load_a(const u32x4* fr, int step, u32x4 (&a0)[3], u32x4 (&a1)[3]) {
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    a0[p] = fr[((0 * 8 + step) * 3 + p) * 64];
    a1[p] = fr[((1 * 8 + step) * 3 + p) * 64];
  }
}

// half a hidden layer: two output blocks from the chunk in `slot`; 8 steps (bi, s), the weight fragments of
// step i+1 are requested before the 12 MFMAs of step i (explicit two-deep pipeline: left to itself the compiler
// hoists every ds_read of the chunk and spills).  The accumulators start at the bias.
__device__ __forceinline__ float x3_softplus(float x);

// PRE: while this chunk multiplies, the softplus of the two blocks the PREVIOUS chunk finished (pre0, pre1) runs in
// its shadow, four elements per step — VALU work taken out of the lock-stepped layer epilogue
template <bool PRE>
__device__ __forceinline__ void x3_hidden_chunk(const char* slot, const u32x4 (&bf)[4][2][3], f32x16& acc0,
                                                f32x16& acc1, const float* __restrict__ sb, int bo0, int lane,
                                                f32x16& pre0, f32x16& pre1) {
  const u32x4* fr = reinterpret_cast<const u32x4*>(slot) + lane;
  u32x4 a0[2][3], a1[2][3];
  x3_load_a(fr, 0, a0[0], a1[0]);
  acc0 = x3_bias(sb, bo0, lane >> 5);
  acc1 = x3_bias(sb, bo0 + 1, lane >> 5);
#pragma unroll
  for (int step = 0; step < 8; ++step) {
    if (step < 7) x3_load_a(fr, step + 1, a0[(step + 1) & 1], a1[(step + 1) & 1]);
    mma6x2(a0[step & 1], a1[step & 1], bf[step >> 1][step & 1], acc0, acc1);
    if (PRE) {
      pre0[2 * step] = x3_softplus(pre0[2 * step]); pre0[2 * step + 1] = x3_softplus(pre0[2 * step + 1]);
      pre1[2 * step] = x3_softplus(pre1[2 * step]); pre1[2 * step + 1] = x3_softplus(pre1[2 * step + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ float x3_softplus(float x) {
  const float t = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
  return fmaf(0.6931471805599453f, __builtin_amdgcn_logf(1.0f + t), fmaxf(x, 0.0f));
}

__device__ __forceinline__ f32x16 x3_bias(const float* __restrict__ sb, int bo, int hf) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 bv = *reinterpret_cast<const float4*>(sb + 32 * bo + 8 * q + 4 * hf);
    acc[4 * q + 0] = bv.x; acc[4 * q + 1] = bv.y; acc[4 * q + 2] = bv.z; acc[4 * q + 3] = bv.w;
  }
  return acc;
}

// layer epilogue for one output block: softplus, optional save, split into the next layer's B fragments.
// Saved rows go through a wave-private LDS tile: a lane holds 16-B pieces of 32 different rows (512-B stride in
// memory: measured 1.7 TB/s), after the transposition 8 lanes write one 128-B line of a row.
// SAVE: 0 none; 1 row-major tape through the LDS transposition; 2 blocked tape (edge_fused.h: edge_tape_blocked):
// inside a full 32-edge group the element (edge l31, feature 32 bo + 8 q + 4 hf + j) lives at
// ((bo*4 + q)*64 + lane)*4 + j — exactly this kernel's accumulator layout and the backward kernel's load layout, so
// every wave store (and load) is one contiguous KB and needs no transposition.  zp / qstride: this lane's first store
// address for the block and the distance between its four stores (the last, partial group stays row-major).
template <int SAVE, bool SP>
__device__ __forceinline__ void x3_epilogue(const f32x16& acc, u32x4 (&bfo)[2][3], float* __restrict__ tb,
                                            float* __restrict__ zblk, float* __restrict__ dummy, int rows_left,
                                            int lane, float* __restrict__ zp, int qstride) {
  typedef float nt4 __attribute__((ext_vector_type(4)));
  const int hf = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float z0 = SP ? x3_softplus(acc[4 * q + 0]) : acc[4 * q + 0], z1 = SP ? x3_softplus(acc[4 * q + 1]) : acc[4 * q + 1];
    const float z2 = SP ? x3_softplus(acc[4 * q + 2]) : acc[4 * q + 2], z3 = SP ? x3_softplus(acc[4 * q + 3]) : acc[4 * q + 3];
    if (SAVE == 1) *reinterpret_cast<float4*>(tb + l31 * X3_TLD + 8 * q + 4 * hf) = make_float4(z0, z1, z2, z3);
    if (SAVE == 2) __builtin_nontemporal_store(nt4{z0, z1, z2, z3}, reinterpret_cast<nt4*>(zp + q * qstride));
    // registers 4q..4q+3  ->  k-step s = q>>1, k-slots t = 4(q&1)..+3  ->  dwords 2(q&1), 2(q&1)+1
    const int s = q >> 1, j = 2 * (q & 1);
    unsigned h, m, l;
    split3_pair(z0, z1, h, m, l);
    bfo[s][0][j] = h; bfo[s][1][j] = m; bfo[s][2][j] = l;
    split3_pair(z2, z3, h, m, l);
    bfo[s][0][j + 1] = h; bfo[s][1][j + 1] = m; bfo[s][2][j + 1] = l;
  }
  if (SAVE == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (lane >> 3) + 8 * i, c = 4 * (lane & 7);
      const float4 v = *reinterpret_cast<const float4*>(tb + r * X3_TLD + c);
      float* d = r < rows_left ? zblk + (int64_t)r * FH + c : dummy + c;
      __builtin_nontemporal_store(nt4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt4*>(d));
    }
  }
}

#define X3_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

template <int SAVE>
__global__ __launch_bounds__(512, 1) void edge_fwd_x3_kernel(EdgeX3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_x3[];
  char* ring = smem_x3;
  float* sT = reinterpret_cast<float*>(smem_x3 + X3_RING);     // [8 waves][32][36] store transposition
  float* sCen = reinterpret_cast<float*>(smem_x3 + X3_RING + X3_TBYTES);   // [128]
  float* sBias = sCen + FH;                                     // [3][128]
  float* sBo = sBias + 3 * FH;                                  // [32]

  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < FH) {
    sCen[tid] = a.centers[tid];
    sBias[tid] = a.bh[0][tid];
    sBias[FH + tid] = a.bh[1][tid];
    sBias[2 * FH + tid] = a.bh[2][tid];
  }
  if (tid < 32) sBo[tid] = tid < a.E ? a.bo[tid] : 0.f;

  const int64_t ntiles = (a.n_edges + X3_TM - 1) / X3_TM;
  float ds_n, de_n;
  {
    const int64_t g0 = std::min<int64_t>((int64_t)blockIdx.x * X3_TM + 32 * wave + l31, a.n_edges - 1);
    ds_n = a.d_src[g0]; de_n = a.d_eff[g0];
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.img, 0, X3_NCHUNK * X3_CHUNK, 0x00020000);
  // ring state: chunk id (0..6) and slot (0..2) of the NEXT chunk to request / to consume
  int req_c = 0, req_s = 0, use_s = 0;
  x3_dma_chunk(rsrc, req_c, ring + req_s * X3_CHUNK, wave, lane);
  req_c = 1; req_s = 1;
  X3_WAIT_DMA();   // chunk 0 landed (this wave's share)

#define X3_STEP_BEGIN()                                              \
  NG_LDS_BARRIER();                                                  \
  x3_dma_chunk(rsrc, req_c, ring + req_s * X3_CHUNK, wave, lane);   \
  req_c = req_c == X3_NCHUNK - 1 ? 0 : req_c + 1;                    \
  req_s ^= 1;
#define X3_STEP_END()                                                \
  X3_WAIT_DMA();                                                     \
  use_s ^= 1;

  u32x4 bf[4][2][3];
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t gr = tile * X3_TM + 32 * wave + l31;
    const bool valid = gr < a.n_edges;
    const float ds = valid ? ds_n : 0.f;
    const float mask = ds > 0.f ? 1.f : 0.f;
    // ---- RBF straight into B fragments; masked edges: d = 1e19 -> exp2(-inf) = exact 0
    {
      const float dm = ds > 0.f ? de_n : 1.0e19f;
      const float c2 = a.neg_inv_gap_log2e;
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int tq = 0; tq < 2; ++tq) {
            const float4 mu = *reinterpret_cast<const float4*>(sCen + 32 * bi + 16 * s + 8 * tq + 4 * hf);
            float u0 = dm - mu.x, u1 = dm - mu.y, u2 = dm - mu.z, u3 = dm - mu.w;
            u0 = __builtin_amdgcn_exp2f(u0 * u0 * c2); u1 = __builtin_amdgcn_exp2f(u1 * u1 * c2);
            u2 = __builtin_amdgcn_exp2f(u2 * u2 * c2); u3 = __builtin_amdgcn_exp2f(u3 * u3 * c2);
            unsigned h, m, l;
            split3_pair(u0, u1, h, m, l);
            bf[bi][s][0][2 * tq] = h; bf[bi][s][1][2 * tq] = m; bf[bi][s][2][2 * tq] = l;
            split3_pair(u2, u3, h, m, l);
            bf[bi][s][0][2 * tq + 1] = h; bf[bi][s][1][2 * tq + 1] = m; bf[bi][s][2][2 * tq + 1] = l;
          }
    }
    {   // distances of this workgroup's next tile
      const int64_t gn = std::min<int64_t>((tile + gridDim.x) * X3_TM + 32 * wave + l31, a.n_edges - 1);
      ds_n = a.d_src[gn]; de_n = a.d_eff[gn];
    }
    // ---- three hidden layers, two chunks each
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
      f32x16 acc[4];
      X3_STEP_BEGIN();
      x3_hidden_chunk<false>(ring + use_s * X3_CHUNK, bf, acc[0], acc[1], sBias + layer * FH, 0, lane, acc[2], acc[3]);
      X3_STEP_END();
      X3_STEP_BEGIN();
      x3_hidden_chunk<true>(ring + use_s * X3_CHUNK, bf, acc[2], acc[3], sBias + layer * FH, 2, lane, acc[0], acc[1]);
      X3_STEP_END();
      // rows of this wave: tile*256 + 32*wave + (0..31); rows_left <= 0 when the wave lies past the end
      const int64_t wrow0 = tile * X3_TM + 32 * wave;
      const int rows_left = (int)std::min<int64_t>(32, a.n_edges - wrow0);
      float* zl = SAVE ? a.z_save + (int64_t)layer * a.n_edges * FH : nullptr;
      float* zw = SAVE ? zl + wrow0 * FH : nullptr;
      const bool full = rows_left >= 32;
      float* zp = nullptr;
      int bstride = 0, qstride = 0;
      if (SAVE == 2) {
        zp = full ? zl + (tile * 8 + wave) * 4096 + lane * 4
                  : (l31 < rows_left ? zl + (wrow0 + l31) * FH + 4 * hf : a.dummy + 4 * hf);
        bstride = full ? 1024 : 32;
        qstride = full ? 256 : 8;
      }
      float* tb = sT + wave * (32 * X3_TLD);
      x3_epilogue<SAVE, false>(acc[0], bf[0], tb, zw, a.dummy, rows_left, lane, zp, qstride);
      x3_epilogue<SAVE, false>(acc[1], bf[1], tb, zw + 32, a.dummy, rows_left, lane, zp + bstride, qstride);
      x3_epilogue<SAVE, true>(acc[2], bf[2], tb, zw + 64, a.dummy, rows_left, lane, zp + 2 * bstride, qstride);
      x3_epilogue<SAVE, true>(acc[3], bf[3], tb, zw + 96, a.dummy, rows_left, lane, zp + 3 * bstride, qstride);
    }
    // ---- output layer: rows 0..E-1 of one 32-row block
    {
      f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
      X3_STEP_BEGIN();
      const u32x4* fr = reinterpret_cast<const u32x4*>(ring + use_s * X3_CHUNK) + lane;
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
        u32x4 a0[3], a1[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a0[p] = fr[((2 * bi + 0) * 3 + p) * 64];
          a1[p] = fr[((2 * bi + 1) * 3 + p) * 64];
        }
        {
          const u32x4 (&b)[3] = bf[bi][0];
          acc0 = mfma_bf16(a0[2], b[0], acc0); acc0 = mfma_bf16(a0[0], b[2], acc0); acc0 = mfma_bf16(a0[1], b[1], acc0);
          acc0 = mfma_bf16(a0[1], b[0], acc0); acc0 = mfma_bf16(a0[0], b[1], acc0);
          acc2 = mfma_bf16(a0[0], b[0], acc2);
        }
        {
          const u32x4 (&b)[3] = bf[bi][1];
          acc1 = mfma_bf16(a1[2], b[0], acc1); acc1 = mfma_bf16(a1[0], b[2], acc1); acc1 = mfma_bf16(a1[1], b[1], acc1);
          acc1 = mfma_bf16(a1[1], b[0], acc1); acc1 = mfma_bf16(a1[0], b[1], acc1);
          acc3 = mfma_bf16(a1[0], b[0], acc3);
        }
      }
      const f32x16 acc = (acc0 + acc1) + (acc2 + acc3);   // small products | leading products
      X3_STEP_END();
      if (valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ne = r + 4 * hf;
          if (ne < a.E) a.e_out[gr * a.E + ne] = mask * (acc[r] + sBo[ne]);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

// default for the edge forward; NG_EDGE_MATH=fp32 selects the f32-input MFMA kernels of edge_fused.hip
bool edge_x3_enabled() {
  return !sw().edge_math_fp32;
}

int edge_x3_fwd(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src, const float* d_eff,
                const float* centers, float gap, const float* const* W, const float* const* b, float* e_out,
                float* z_save) {
  const size_t img_bytes = (size_t)X3_NCHUNK * X3_CHUNK;
  bool have = false;
  char* img = (char*)cached_image(ctx, W[0], 2, img_bytes + FH * 4, &have);
  if (!img) img = (char*)workspace(ctx, img_bytes + FH * 4);
  if (!img) return NG_ERR_NOMEM;
  if (!have) {
    hipLaunchKernelGGL(x3_pack_kernel, dim3(cdiv(X3_NCHUNK * 16 * 64, 256)), dim3(256), 0, st, W[0], W[1], W[2], W[3], E,
                       (unsigned*)img);
    NG_HIP(ctx, hipGetLastError());
  }
  EdgeX3Args a;
  a.n_edges = n_edges; a.d_src = d_src; a.d_eff = d_eff; a.centers = centers;
  a.neg_inv_gap_log2e = (float)(-1.4426950408889634 / (double)gap);
  a.img = img;
  a.bh[0] = b[0]; a.bh[1] = b[1]; a.bh[2] = b[2];
  a.bo = b[3]; a.E = E; a.e_out = e_out; a.z_save = z_save;
  a.dummy = (float*)(img + img_bytes);
  const int64_t ntiles = cdiv(n_edges, X3_TM);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu);
  const size_t lds = X3_RING + X3_TBYTES + (size_t)(FH + 3 * FH + 32) * 4;
  ProfScope ps(ctx, st, "edge_fwd_x3");
  if (z_save && edge_tape_blocked(E, n_edges)) hipLaunchKernelGGL(edge_fwd_x3_kernel<2>, dim3(grid), dim3(512), lds, st, a);
  else if (z_save) hipLaunchKernelGGL(edge_fwd_x3_kernel<1>, dim3(grid), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(edge_fwd_x3_kernel<0>, dim3(grid), dim3(512), lds, st, a);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

}  // namespace ng
