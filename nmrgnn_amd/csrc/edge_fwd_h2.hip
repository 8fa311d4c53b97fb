// Fused edge forward on the fp16 matrix pipe with fp32-grade arithmetic: every fp32 operand split into TWO fp16
// pieces, three piece products per multiply (h2_common.cuh).  Default; NG_EDGE_MATH=fp32 selects the f32-input MFMA
// kernel (edge_fused.hip).  Until late round 2 this kernel used the exact three-piece bf16 split (six products).
// Reference: nmrgnn/model.py:251-261 + nmrgnn/layers.py:137-140 + nmrgnn/model.py:132-138.
//
// Why.  Matrix time and VALU time ADD on a gfx950 SIMD, and the three-piece kernel sits on that sum:
// 616 MFMAs + ~3200 VALU instructions per wave and 256-edge tile.  Two fp16 pieces carry 22-24 significand bits —
// fp32's 24 — so three products (lh + hl + hh) reach fp32-level error with HALF the matrix instructions, and a split
// costs one conversion back + one subtraction + one packed conversion instead of two of each.
//
// Mapping.  512 threads = 8 waves, one persistent workgroup per CU, 256 edges per tile; wave w owns edges [32w, 32w+32) through ALL layers:
//   v_mfma_f32_32x32x16_f16  D[feature][edge] += A[feature][k] * B[k][edge]
//   A = weight pieces from LDS (ds_read_b128 of a lane-linear fragment image), B = activation pieces in VGPRs.
// The contraction index of the next layer is permuted (h2_feat) so that a lane's 16 outputs of a block are its own
// k-slots of two k-steps: softplus -> split -> pack turns accumulators into the next B operand in registers.
//
// Weight scale.  W pieces are taken from 2^WS * W (WS = 8, exact): the l piece of a typical weight (|W| ~ 0.1) would
// otherwise be an fp16 subnormal (2^-12 |W| < 2^-14) with only a few significant bits.  Accumulators then hold
// 2^WS * (pre-activation); the softplus reads them through scaled constants (one extra v_mul per element).
// Activations stay unscaled: RBF values and softplus outputs are O(1), their small elements carry an ABSOLUTE error
// <= 2^-25 (subnormal l pieces are honoured by the MFMA), below the fp32 rounding of the O(1) partners they are
// summed with.  Softplus outputs >= 65504 would overflow the h piece (inf -> NaN output, loud).
//
// Weights reach LDS by LDS-DMA in 32-KB chunks (half a layer: 2 output blocks x 4 input blocks x 2 k-steps x
// 2 pieces x 1 KB fragments); the ring has two slots of a WHOLE layer (two chunks, 128 KB in all): four barriers per
// tile.  (The row-major-tape form, which also needs the transposition tile, keeps half-layer slots: seven.)
#include <algorithm>
#include <string>

#include "edge_fused.h"
#include "h2_common.cuh"
#include "pack_bodies.cuh"

namespace ng {

constexpr int H2_TM = 256;
constexpr int H2_CHUNK = 32 * 1024;
constexpr int H2_NCHUNK = 7;
constexpr int H2_RING = 2 * H2_CHUNK;
constexpr int H2_TLD = 36;                 // row stride (floats) of a wave's 32 x 32 transposition tile
constexpr int H2_TBYTES = 8 * 32 * H2_TLD * 4;
constexpr int H2_WS = 8;                   // log2 of the weight scale
constexpr float H2_WSCALE = (float)(1 << H2_WS), H2_WINV = 1.0f / (float)(1 << H2_WS);

// feature (within a 32-block) held in k-slot t (0..7) of k-step s by lane half hf  ==  accumulator register 8s+t
__host__ __device__ inline int h2_feat(int t, int s, int hf) { return (t & 3) + 16 * s + 8 * (t >> 2) + 4 * hf; }

// Weight image (packed by pack_bodies.cuh: PK_EDGE_H2): chunk c (0..5): layer c>>1, output blocks 2*(c&1) + {0,1};  chunk 6:
// output layer (rows >= E zero).
//   fragment ((bo_l*4 + bi)*2 + s)*2 + p, 1 KB each, lane-linear 16 B per lane:
//   lane (row i = l&31, k-slot t) = piece_p( 2^WS W[k = 32 bi + h2_feat(t, s, l>>5)][n = 32 bo + i] )
static_assert(H2_NCHUNK == pk::H2_NCHUNKd && H2_CHUNK == pk::H2_CHUNKd && H2_WSCALE == pk::WSCALE && FH == pk::FHd, "pack_bodies.cuh");

struct EdgeH2Args {
  int64_t n_edges;
  const float* d_src;
  const float* d_eff;
  const float* centers;
  float neg_inv_gap_log2e;
  const char* img;        // [7][32 KB]
  const float* bh[3];
  const float* bo;
  int E;
  float* e_out;
  float* z_save;          // [3][n_edges][128] or nullptr
  float* dummy;           // 128 floats
  RangeGuard guard;       // raised when an output comes out non-finite (an operand left the fp16 range)
  // live-edge view (ng_internal.h: LiveEdges): rows = compacted live slots, row r writes e_out[perm[r]]; n_edges above is
  // then the SLOT count (upper bound of the rows, and the length of perm), the row count is *n_live
  const int32_t* perm;
  const int32_t* n_live;
  int64_t z_layer_stride; // floats between the layers of the tape
};

// one chunk = 32 wave-instructions of 1 KB; wave w moves KB w, w+8, w+16, w+24 (scalar resource + scalar offset + one
// lane-offset VGPR)
__device__ __forceinline__ void h2_dma_chunk(dma_i4 rsrc, int cid, char* slot, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kb = wave + 8 * j;
    lds_dma16(rsrc, slot + kb * 1024, lane * 16, cid * H2_CHUNK + kb * 1024);
  }
}

// softplus of x = 2^-WS y, read from the scaled accumulator y
__device__ __forceinline__ float h2_softplus(float y) {
  const float t = __builtin_amdgcn_exp2f((-1.4426950408889634f * H2_WINV) * fabsf(y));
  return fmaf(0.6931471805599453f, __builtin_amdgcn_logf(1.0f + t), fmaxf(y * H2_WINV, 0.0f));
}

// initial accumulator = scaled bias (sb holds 2^WS b)
__device__ __forceinline__ f32x16 h2_bias(const float* __restrict__ sb, int bo, int hf) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 bv = *reinterpret_cast<const float4*>(sb + 32 * bo + 8 * q + 4 * hf);
    acc[4 * q + 0] = bv.x; acc[4 * q + 1] = bv.y; acc[4 * q + 2] = bv.z; acc[4 * q + 3] = bv.w;
  }
  return acc;
}

__device__ __forceinline__ void h2_load_a(const u32x4* fr, int step, u32x4 (&a0)[2], u32x4 (&a1)[2]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    a0[p] = fr[((0 * 8 + step) * 2 + p) * 64];
    a1[p] = fr[((1 * 8 + step) * 2 + p) * 64];
  }
}

// half a hidden layer: two output blocks from the chunk in `slot`; 8 steps (bi, s), the weight fragments of step i+1
// are requested before the 6 MFMAs of step i.  PRE: the softplus of the two blocks the PREVIOUS chunk finished
// (pre0, pre1) is spread over the steps, four elements per step.
template <bool PRE>
__device__ __forceinline__ void h2_hidden_chunk(const char* slot, const u32x4 (&bf)[4][2][2], f32x16& acc0,
                                                f32x16& acc1, const float* __restrict__ sb, int bo0, int lane,
                                                f32x16& pre0, f32x16& pre1) {
  const u32x4* fr = reinterpret_cast<const u32x4*>(slot) + lane;
  u32x4 a0[2][2], a1[2][2];
  h2_load_a(fr, 0, a0[0], a1[0]);
  acc0 = h2_bias(sb, bo0, lane >> 5);
  acc1 = h2_bias(sb, bo0 + 1, lane >> 5);
#pragma unroll
  for (int step = 0; step < 8; ++step) {
    if (step < 7) h2_load_a(fr, step + 1, a0[(step + 1) & 1], a1[(step + 1) & 1]);
    mma3_2a(a0[step & 1], a1[step & 1], bf[step >> 1][step & 1], acc0, acc1);
    if (PRE) {
      pre0[2 * step] = h2_softplus(pre0[2 * step]); pre0[2 * step + 1] = h2_softplus(pre0[2 * step + 1]);
      pre1[2 * step] = h2_softplus(pre1[2 * step]); pre1[2 * step + 1] = h2_softplus(pre1[2 * step + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the blocked tape's row stores (tools: -DH2_TAPE_PLAIN for the A/B of plain against non-temporal stores)
template <class V>
__device__ __forceinline__ void h2_tape_store(V v, V* p) {
#ifdef H2_TAPE_PLAIN
  *p = v;
#else
  __builtin_nontemporal_store(v, p);
#endif
}

// layer epilogue for one output block: softplus (SP: still to do), optional save, split into the next layer's B
// fragments.  SAVE: 0 none; 1 row-major tape through a wave-private LDS transposition; 2 blocked tape
// (edge_fused.h: edge_tape_blocked) — the accumulator layout itself, one contiguous KB per wave store.
template <int SAVE, bool SP>
__device__ __forceinline__ void h2_epilogue(const f32x16& acc, u32x4 (&bfo)[2][2], float* __restrict__ tb,
                                            float* __restrict__ zblk, float* __restrict__ dummy, int rows_left,
                                            int lane, float* __restrict__ zp, int qstride) {
  typedef float nt4 __attribute__((ext_vector_type(4)));
  const int hf = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float z0 = SP ? h2_softplus(acc[4 * q + 0]) : acc[4 * q + 0], z1 = SP ? h2_softplus(acc[4 * q + 1]) : acc[4 * q + 1];
    const float z2 = SP ? h2_softplus(acc[4 * q + 2]) : acc[4 * q + 2], z3 = SP ? h2_softplus(acc[4 * q + 3]) : acc[4 * q + 3];
    if (SAVE == 1) *reinterpret_cast<float4*>(tb + l31 * H2_TLD + 8 * q + 4 * hf) = make_float4(z0, z1, z2, z3);
    if (SAVE == 2) h2_tape_store(nt4{z0, z1, z2, z3}, reinterpret_cast<nt4*>(zp + q * qstride));
    // registers 4q..4q+3  ->  k-step s = q>>1, k-slots t = 4(q&1)..+3  ->  dwords 2(q&1), 2(q&1)+1
    const int s = q >> 1, j = 2 * (q & 1);
    unsigned h, l;
    split2_pair(z0, z1, h, l);
    bfo[s][0][j] = h; bfo[s][1][j] = l;
    split2_pair(z2, z3, h, l);
    bfo[s][0][j + 1] = h; bfo[s][1][j + 1] = l;
  }
  if (SAVE == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (lane >> 3) + 8 * i, c = 4 * (lane & 7);
      const float4 v = *reinterpret_cast<const float4*>(tb + r * H2_TLD + c);
      float* d = r < rows_left ? zblk + (int64_t)r * FH + c : dummy + c;
      __builtin_nontemporal_store(nt4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt4*>(d));
    }
  }
}

// The same half layer with the caller's elementwise work handed in per step (fill(step), compile-time step): one wave
// has to put its own VALU work into the shadow of its own MFMAs — up to five plain VALU instructions issue per
// 32x32x16 MFMA for free (tools/ubench/mfma_fill.hip), while VALU work outside an MFMA sequence is not overlapped by the
// partner wave either (the two waves of a SIMD run in lockstep between the layer barriers).  The order inside a step is
// the compiler's: explicit sched_group_barrier patterns (one MFMA, then 4-10 VALU, six times) measured 2-3 % slower.
template <class F>
__device__ __forceinline__ void h2_hidden_chunk_f(const char* slot, const u32x4 (&bf)[4][2][2], f32x16& acc0, f32x16& acc1,
                                                  const float* __restrict__ sb, int bo0, int lane, F&& fill) {
  const u32x4* fr = reinterpret_cast<const u32x4*>(slot) + lane;
  u32x4 a0[2][2], a1[2][2];
  h2_load_a(fr, 0, a0[0], a1[0]);
  acc0 = h2_bias(sb, bo0, lane >> 5);
  acc1 = h2_bias(sb, bo0 + 1, lane >> 5);
#pragma unroll
  for (int step = 0; step < 8; ++step) {
    if (step < 7) h2_load_a(fr, step + 1, a0[(step + 1) & 1], a1[(step + 1) & 1]);
    mma3_2a(a0[step & 1], a1[step & 1], bf[step >> 1][step & 1], acc0, acc1);
    fill(step);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// one quarter (registers 4q .. 4q+3) of h2_epilogue for the tape forms without a transposition tile (SAVE 0 / 2)
template <int SAVE, bool SP>
__device__ __forceinline__ void h2_epilogue_q(const f32x16& acc, u32x4 (&bfo)[2][2], int q, float* __restrict__ zp, int qstride) {
  typedef float nt4 __attribute__((ext_vector_type(4)));
  const float z0 = SP ? h2_softplus(acc[4 * q + 0]) : acc[4 * q + 0], z1 = SP ? h2_softplus(acc[4 * q + 1]) : acc[4 * q + 1];
  const float z2 = SP ? h2_softplus(acc[4 * q + 2]) : acc[4 * q + 2], z3 = SP ? h2_softplus(acc[4 * q + 3]) : acc[4 * q + 3];
  if (SAVE == 2) h2_tape_store(nt4{z0, z1, z2, z3}, reinterpret_cast<nt4*>(zp + q * qstride));
  const int s = q >> 1, j = 2 * (q & 1);
  unsigned h, l;
  split2_pair(z0, z1, h, l);
  bfo[s][0][j] = h; bfo[s][1][j] = l;
  split2_pair(z2, z3, h, l);
  bfo[s][0][j + 1] = h; bfo[s][1][j + 1] = l;
}

#define H2_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// -DH2_STAMP (tools/variants): cycle stamps of one wave of one workgroup through its third tile, printed by the launcher
#ifdef H2_STAMP
__device__ unsigned long long h2_stamps[2][64];
#define H2_T(i) do { if (blockIdx.x == 100 && lane == 0 && (wave == 0 || wave == 5) && tile == blockIdx.x + 2 * (int64_t)gridDim.x) h2_stamps[wave == 0 ? 0 : 1][i] = __builtin_readcyclecounter(); } while (0)
#else
#define H2_T(i) do { } while (0)
#endif

template <int SAVE>
__global__ __launch_bounds__(512, 1) void edge_fwd_h2_kernel(EdgeH2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_h2[];
  // WL (every form but the row-major tape, which needs the transposition tile): ring slots hold a WHOLE layer (64 KB = two
  // consecutive half-layer chunks of the image): four barriers and DMA waits per tile instead of seven, and a layer's
  // weights have the whole previous layer to arrive
  constexpr bool WL = SAVE != 1;
  constexpr int SLOT = WL ? 2 * H2_CHUNK : H2_CHUNK;
  char* ring = smem_h2;
  float* sT = reinterpret_cast<float*>(smem_h2 + 2 * SLOT);    // [8 waves][32][36] store transposition (SAVE == 1)
  float* sCen = reinterpret_cast<float*>(smem_h2 + 2 * SLOT + (SAVE == 1 ? H2_TBYTES : 0));   // [128]
  float* sBias = sCen + FH;                                     // [3][128], scaled
  float* sBo = sBias + 3 * FH;                                  // [32]

  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < FH) {
    sCen[tid] = a.centers[tid];
    sBias[tid] = H2_WSCALE * a.bh[0][tid];
    sBias[FH + tid] = H2_WSCALE * a.bh[1][tid];
    sBias[2 * FH + tid] = H2_WSCALE * a.bh[2][tid];
  }
  if (tid < 32) sBo[tid] = tid < a.E ? a.bo[tid] : 0.f;

  // rows of this launch: every slot, or the compacted live ones (device scalar: no host round trip per batch).  A NEGATIVE row
  // count: this launch has nothing to do and e_out belongs to somebody else (the edge-function table's guard is down,
  // edge_table.hip): not even the dead slots are written
  if (a.perm && __builtin_amdgcn_readfirstlane(*a.n_live) < 0) return;
  const int64_t n_edges = a.perm ? (int64_t)__builtin_amdgcn_readfirstlane(*a.n_live) : a.n_edges;
  const int64_t ntiles = (n_edges + H2_TM - 1) / H2_TM;
  float ds_n, de_n;
  int pg_n = 0;           // slot of this lane's row (live view), requested a tile ahead like the distances
  {
    const int64_t g0 = std::max<int64_t>(std::min<int64_t>((int64_t)blockIdx.x * H2_TM + 32 * wave + l31, n_edges - 1), 0);
    ds_n = a.d_src[g0]; de_n = a.d_eff[g0];
    if (a.perm) pg_n = a.perm[g0];
  }
  if (a.perm) {
    // the dead slots (perm[n_live ..]) carry e == 0 (model.py:261: the mask multiplies the MLP output); nobody else writes them
    for (int64_t i = n_edges + (int64_t)blockIdx.x * 512 + tid; i < a.n_edges; i += (int64_t)gridDim.x * 512) {
      const int64_t g = a.perm[i];
      for (int c = 0; c < a.E; ++c) a.e_out[g * a.E + c] = 0.f;
    }
  }
  __syncthreads();
  const dma_i4 rsrc = dma_rsrc(a.img, (unsigned)(H2_NCHUNK * H2_CHUNK));
  // ring state: chunk id (0..6) and slot of the NEXT chunk to request / to consume
  int req_c = 0, req_s = 0, use_s = 0;
  h2_dma_chunk(rsrc, req_c, ring + req_s * SLOT, wave, lane);
  if (WL) h2_dma_chunk(rsrc, 1, ring + req_s * SLOT + H2_CHUNK, wave, lane);
  req_c = WL ? 2 : 1; req_s = 1;
  H2_WAIT_DMA();   // chunk 0 (WL: layer 0) landed (this wave's share)

  // WL: req_c counts half-layer chunks in steps of two (0, 2, 4 = layers, 6 = output layer, one half-chunk)
#define H2_STEP_BEGIN()                                              \
  NG_LDS_BARRIER();                                                  \
  h2_dma_chunk(rsrc, req_c, ring + req_s * SLOT, wave, lane);       \
  if (WL && req_c < H2_NCHUNK - 1) h2_dma_chunk(rsrc, req_c + 1, ring + req_s * SLOT + H2_CHUNK, wave, lane); \
  req_c = WL ? (req_c >= H2_NCHUNK - 1 ? 0 : req_c + 2) : (req_c == H2_NCHUNK - 1 ? 0 : req_c + 1); \
  req_s ^= 1;
#define H2_STEP_END()                                                \
  H2_WAIT_DMA();                                                     \
  use_s ^= 1;

  u32x4 bf[4][2][2];
  bool bad = false;
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t gr = tile * H2_TM + 32 * wave + l31;
    const bool valid = gr < n_edges;
    const int gout = a.perm ? pg_n : (int)gr;      // slot this row's e goes to (slot counts fit 31 bits)
    const float ds = valid ? ds_n : 0.f;
    const float mask = ds > 0.f ? 1.f : 0.f;
    const float dm = ds > 0.f ? de_n : 1.0e19f;      // masked edges: d = 1e19 -> exp2(-inf) = exact 0
    const float c2 = a.neg_inv_gap_log2e;
    // one RBF unit = the four dwords' worth of B fragment bf[bi][s][.][2 tq .. 2 tq + 1] (four centres of this lane)
    auto rbf_unit = [&](int bi, int s, int tq) {
      // (opaque offset: the centres are constants of the launch — left visible, the sixteen loads were hoisted out of the
      // tile loop and their 64 registers spilled in the inference form)
      int off = 32 * bi + 16 * s + 8 * tq + 4 * hf;
      asm volatile("" : "+v"(off));
      const float4 mu = *reinterpret_cast<const float4*>(sCen + off);
      float u0 = dm - mu.x, u1 = dm - mu.y, u2 = dm - mu.z, u3 = dm - mu.w;
      u0 = __builtin_amdgcn_exp2f(u0 * u0 * c2); u1 = __builtin_amdgcn_exp2f(u1 * u1 * c2);
      u2 = __builtin_amdgcn_exp2f(u2 * u2 * c2); u3 = __builtin_amdgcn_exp2f(u3 * u3 * c2);
      unsigned h, l;
      split2_pair(u0, u1, h, l);
      bf[bi][s][0][2 * tq] = h; bf[bi][s][1][2 * tq] = l;
      split2_pair(u2, u3, h, l);
      bf[bi][s][0][2 * tq + 1] = h; bf[bi][s][1][2 * tq + 1] = l;
    };
    // rows of this wave: tile*256 + 32*wave + (0..31); rows_left <= 0 when the wave lies past the end
    const int64_t wrow0 = tile * H2_TM + 32 * wave;
    const int rows_left = (int)std::min<int64_t>(32, n_edges - wrow0);
    const bool full = rows_left >= 32;
    if constexpr (SAVE != 1) {
      // ---- pipelined schedule (blocked tape: the training forward; the inference form spilled 56 B/lane with it).  The first chunk of a layer is 48 MFMAs with no
      // elementwise work of its own: it takes the RBF of the blocks it has not reached yet (layer 0) or the epilogue of
      // the previous layer's blocks 2 and 3 (their pieces are first needed at its step 4 / 6); the second chunk carries
      // the softplus of blocks 0 and 1 as before.
      f32x16 acc[4];
      const int bstride = full ? 1024 : 32, qstride = full ? 256 : 8;
      // tape position of this lane in layer 0; layer l adds l * n_edges * 128 floats (rows past the end go to the dummy row)
      const bool todummy = !full && l31 >= rows_left;
      float* zp0 = nullptr;
      if (SAVE == 2)
        zp0 = full ? a.z_save + (tile * 8 + wave) * 4096 + lane * 4 : a.z_save + (wrow0 + l31) * FH + 4 * hf;
      const int64_t lstride = a.z_layer_stride;
      auto zpl = [&](int layer) -> float* { return todummy ? a.dummy + 4 * hf : zp0 + layer * lstride; };
      // B fragments of the first two steps up front, the rest inside layer 0's first chunk
      H2_T(0);
      rbf_unit(0, 0, 0); rbf_unit(0, 0, 1); rbf_unit(0, 1, 0); rbf_unit(0, 1, 1);
      {   // distances of this workgroup's next tile
        const int64_t gn = std::min<int64_t>((tile + gridDim.x) * H2_TM + 32 * wave + l31, n_edges - 1);
        ds_n = a.d_src[gn]; de_n = a.d_eff[gn];
        if (a.perm) pg_n = a.perm[gn];
      }
#pragma unroll
      for (int layer = 0; layer < 3; ++layer) {
        H2_T(1 + 5 * layer);
        H2_STEP_BEGIN();
        H2_T(2 + 5 * layer);
        if (layer == 0) {
          h2_hidden_chunk_f(ring + use_s * SLOT, bf, acc[0], acc[1], sBias, 0, lane, [&](int step) {
            if (step < 6) { rbf_unit((step + 2) >> 1, (step + 2) & 1, 0); rbf_unit((step + 2) >> 1, (step + 2) & 1, 1); }
          });
        } else {
          float* zp = zpl(layer - 1);        // the previous layer's blocks 2 / 3: tape rows and next-layer pieces
          h2_hidden_chunk_f(ring + use_s * SLOT, bf, acc[0], acc[1], sBias + layer * FH, 0, lane, [&](int step) {
            if (step < 4) h2_epilogue_q<SAVE, true>(acc[2], bf[2], step, zp + 2 * bstride, qstride);
            if (step >= 2 && step < 6) h2_epilogue_q<SAVE, true>(acc[3], bf[3], step - 2, zp + 3 * bstride, qstride);
          });
        }
        H2_T(3 + 5 * layer);
        // blocks 2 / 3 of THIS layer (the previous layer's acc[2] / acc[3] were consumed inside the chunk above)
        h2_hidden_chunk_f(ring + use_s * SLOT + H2_CHUNK, bf, acc[2], acc[3], sBias + layer * FH, 2, lane, [&](int step) {
          acc[0][2 * step] = h2_softplus(acc[0][2 * step]); acc[0][2 * step + 1] = h2_softplus(acc[0][2 * step + 1]);
          acc[1][2 * step] = h2_softplus(acc[1][2 * step]); acc[1][2 * step + 1] = h2_softplus(acc[1][2 * step + 1]);
        });
        H2_T(4 + 5 * layer);
        H2_STEP_END();
        H2_T(5 + 5 * layer);
        float* zp = zpl(layer);
#pragma unroll
        for (int q = 0; q < 4; ++q) h2_epilogue_q<SAVE, false>(acc[0], bf[0], q, zp, qstride);
#pragma unroll
        for (int q = 0; q < 4; ++q) h2_epilogue_q<SAVE, false>(acc[1], bf[1], q, zp + bstride, qstride);
      }
      H2_T(16);
      {   // the last hidden layer's blocks 2 / 3 have no MFMAs left to hide under (the output layer needs all pieces)
        float* zp = zpl(2);
#pragma unroll
        for (int q = 0; q < 4; ++q) h2_epilogue_q<SAVE, true>(acc[2], bf[2], q, zp + 2 * bstride, qstride);
#pragma unroll
        for (int q = 0; q < 4; ++q) h2_epilogue_q<SAVE, true>(acc[3], bf[3], q, zp + 3 * bstride, qstride);
      }
    } else {
    // ---- RBF straight into B fragments
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int s = 0; s < 2; ++s) { rbf_unit(bi, s, 0); rbf_unit(bi, s, 1); }
    {   // distances of this workgroup's next tile
      const int64_t gn = std::min<int64_t>((tile + gridDim.x) * H2_TM + 32 * wave + l31, n_edges - 1);
      ds_n = a.d_src[gn]; de_n = a.d_eff[gn];
      if (a.perm) pg_n = a.perm[gn];
    }
    // ---- three hidden layers, two chunks each
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
      f32x16 acc[4];
      H2_STEP_BEGIN();
      h2_hidden_chunk<false>(ring + use_s * SLOT, bf, acc[0], acc[1], sBias + layer * FH, 0, lane, acc[2], acc[3]);
      if (!WL) {
        H2_STEP_END();
        H2_STEP_BEGIN();
      }
      h2_hidden_chunk<true>(ring + use_s * SLOT + (WL ? H2_CHUNK : 0), bf, acc[2], acc[3], sBias + layer * FH, 2, lane, acc[0], acc[1]);
      H2_STEP_END();
      float* zl = SAVE ? a.z_save + (int64_t)layer * a.z_layer_stride : nullptr;
      float* zw = SAVE ? zl + wrow0 * FH : nullptr;
      float* tb = sT + wave * (32 * H2_TLD);
      h2_epilogue<SAVE, false>(acc[0], bf[0], tb, zw, a.dummy, rows_left, lane, nullptr, 0);
      h2_epilogue<SAVE, false>(acc[1], bf[1], tb, zw + 32, a.dummy, rows_left, lane, nullptr, 0);
      h2_epilogue<SAVE, true>(acc[2], bf[2], tb, zw + 64, a.dummy, rows_left, lane, nullptr, 0);
      h2_epilogue<SAVE, true>(acc[3], bf[3], tb, zw + 96, a.dummy, rows_left, lane, nullptr, 0);
    }
    }
    // ---- output layer: rows 0..E-1 of one 32-row block
    H2_T(17);
    {
      f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
      H2_STEP_BEGIN();
      const u32x4* fr = reinterpret_cast<const u32x4*>(ring + use_s * SLOT) + lane;
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
        u32x4 a0[2], a1[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          a0[p] = fr[((2 * bi + 0) * 2 + p) * 64];
          a1[p] = fr[((2 * bi + 1) * 2 + p) * 64];
        }
        {
          const u32x4 (&b)[2] = bf[bi][0];
          acc0 = mfma_f16(a0[1], b[0], acc0); acc0 = mfma_f16(a0[0], b[1], acc0);
          acc2 = mfma_f16(a0[0], b[0], acc2);
        }
        {
          const u32x4 (&b)[2] = bf[bi][1];
          acc1 = mfma_f16(a1[1], b[0], acc1); acc1 = mfma_f16(a1[0], b[1], acc1);
          acc3 = mfma_f16(a1[0], b[0], acc3);
        }
      }
      const f32x16 acc = (acc0 + acc1) + (acc2 + acc3);   // small products | leading products
      H2_T(18);
      H2_STEP_END();
      H2_T(19);
      if (valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int ne = r + 4 * hf;
          asm volatile("" : "+v"(ne));      // (not a tile-loop invariant to be kept in — and spilled from — a register)
          if (ne < a.E) {
            const float v = mask * fmaf(acc[r], H2_WINV, sBo[ne]);
            bad |= not_finite(v);
            a.e_out[(int64_t)gout * a.E + ne] = v;
          }
        }
      }
    }
    H2_T(20);
  }
  range_guard_raise(a.guard, bad);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

// default for the edge path; NG_EDGE_MATH=fp32 selects the f32-input MFMA kernels of edge_fused.hip / edge_fused_bwd.hip
bool edge_split_enabled() { return !sw().edge_math_fp32; }

int edge_h2_fwd(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src, const float* d_eff,
                const float* centers, float gap, const float* const* W, const float* const* b, float* e_out,
                float* z_save, LiveEdges live) {
  const size_t img_bytes = (size_t)H2_NCHUNK * H2_CHUNK;
  bool have = false;
  char* img = (char*)cached_image(ctx, W[0], 6, img_bytes + FH * 4, &have);
  const bool cached = img != nullptr;
  if (!img) img = (char*)workspace(ctx, img_bytes + FH * 4);
  if (!img) return NG_ERR_NOMEM;
  if (!have) {
    PackJob j;
    j.kind = PK_EDGE_H2; j.blocks = (int)cdiv(H2_NCHUNK * 16 * 64, PKB); j.i0 = E;
    j.src[0] = W[0]; j.src[1] = W[1]; j.src[2] = W[2]; j.src[3] = W[3]; j.dst[0] = img;
    if (int rc = pack_launch(ctx, st, j)) return rc;
    if (cached) cache_set_job(ctx, W[0], 6, j);
  }
  EdgeH2Args a;
  a.n_edges = n_edges; a.d_src = d_src; a.d_eff = d_eff; a.centers = centers;
  a.neg_inv_gap_log2e = (float)(-1.4426950408889634 / (double)gap);
  a.img = img;
  a.bh[0] = b[0]; a.bh[1] = b[1]; a.bh[2] = b[2];
  a.bo = b[3]; a.E = E; a.e_out = e_out; a.z_save = z_save;
  a.perm = live.perm; a.n_live = live.n_live; a.z_layer_stride = n_edges * FH;
  a.dummy = (float*)(img + img_bytes);
  a.guard = range_guard_begin(ctx);
  if (!a.guard.word) return NG_ERR_NOMEM;
  const int64_t ntiles = cdiv(n_edges, H2_TM);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu);
  const size_t misc = (size_t)(FH + 3 * FH + 32) * 4;
  ProfScope ps(ctx, st, "edge_fwd_h2");
  if (z_save && edge_tape_blocked(E, n_edges))
    hipLaunchKernelGGL(edge_fwd_h2_kernel<2>, dim3(grid), dim3(512), 2 * H2_RING + misc, st, a);
  else if (z_save)
    hipLaunchKernelGGL(edge_fwd_h2_kernel<1>, dim3(grid), dim3(512), H2_RING + H2_TBYTES + misc, st, a);
  else
    hipLaunchKernelGGL(edge_fwd_h2_kernel<0>, dim3(grid), dim3(512), 2 * H2_RING + misc, st, a);
  NG_HIP(ctx, hipGetLastError());
#ifdef H2_STAMP
  {
    static int calls = 0;
    if (++calls % 20 == 0 && ntiles > 4 * (int64_t)grid) {
      unsigned long long hb[2][64];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpyFromSymbol(hb, HIP_SYMBOL(h2_stamps), sizeof(hb));
      static const char* nm[21] = {"tile start", "L0 rbf up front", "L0 barrier+request", "L0 chunk A", "L0 chunk B", "L0 dma wait", "L1 epilogue 0/1",
                                   "L1 barrier+request", "L1 chunk A", "L1 chunk B", "L1 dma wait", "L2 epilogue 0/1", "L2 barrier+request", "L2 chunk A",
                                   "L2 chunk B", "L2 dma wait", "epilogue 0/1", "epilogue 2/3", "out layer (barrier, request, 24 MFMA)", "out dma wait", "e stores"};
      for (int w = 0; w < 2; ++w) {
        fprintf(stderr, "H2 stamps, %s form, wave %d:", z_save ? "tape" : "no-tape", w ? 5 : 0);
        for (int i = 1; i <= 20; ++i) fprintf(stderr, "  %s %lld", nm[i], (long long)(hb[w][i] - hb[w][i - 1]));
        fprintf(stderr, "  | tile %lld\n", (long long)(hb[w][20] - hb[w][0]));
      }
    }
  }
#endif
  // the same call on f32-input MFMA, executed only if the kernel above raised the guard (operands beyond the fp16 range)
  return edge_fused_fwd_f32(ctx, st, n_edges, E, d_src, d_eff, centers, gap, W, b, e_out, z_save,
                            z_save && edge_tape_blocked(E, n_edges), &a.guard, live);
}

}  // namespace ng
