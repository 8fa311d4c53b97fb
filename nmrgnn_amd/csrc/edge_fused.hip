// Fused persistent edge kernels for edge_hidden_size == 128 (the bundled model's and the bench's H).
// Reference: nmrgnn/model.py:251-261 + nmrgnn/layers.py:137-140 + nmrgnn/model.py:132-138.
//
// forward, per 64-edge tile, entirely on-chip:
//   d -> mask, RBF (128 exps per edge, written straight into an LDS tile)            [VALU]
//   3 x { Z = softplus(X W + b) }   X, Z: LDS tiles [64][128] (ping-pong)            [MFMA fp32]
//   e  = mask * (Z3 Wo + bo)        E = 1..8 outputs per edge                        [VALU]
// Only e[n_edges,E] leaves the CU (plus the three hidden activations when training, copied out of
// LDS as whole 512-B rows).  The [n_edges,128] RBF tensor TensorFlow materialises (1 GB at the
// bench shape) never exists.
//
// Mapping.  256 threads = 4 waves; wave w owns output columns [32w, 32w+32) of every hidden layer.
// The layer's weight slab for those columns lives in 64 VGPRs as ready-made MFMA A-fragments
// (v_mfma_f32_32x32x2_f32: A = W^T fragment, B = activation fragment read from LDS with one
// ds_read_b128 per 4 MFMA steps), loaded from a pre-packed, fully coalesced copy of W that sits in
// L2; the next layer's slab is prefetched into a second 64-VGPR set while the current layer's 128
// MFMAs run.  <= 256 VGPRs and 68 KB LDS per workgroup -> two workgroups (8 waves) per CU, so one
// workgroup's softplus epilogue / LDS traffic overlaps the other's MFMA chain.
#include <algorithm>

#include "mfma_gemm.cuh"
#include "edge_fused.h"
#include "pack_bodies.cuh"

namespace ng {


// softplus for the epilogue in 6 VALU instructions: max(x,0) + ln2 * log2(1 + 2^(-|x| log2 e)).
// v_exp_f32 / v_log_f32 are ~1 ulp; forming 1+t rounds at 6e-8 ABSOLUTE, which is the rounding level
// of the O(1) activations this feeds (parity tests hold the 1e-4 budget on the final shifts).
__device__ __forceinline__ float softplus_fast(float x) {
  const float t = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
  return fmaf(0.6931471805599453f, __builtin_amdgcn_logf(1.0f + t), fmaxf(x, 0.0f));
}

// Wpk[layer][w][t][lane][s] = W[layer][k = 8t + 4*(lane>>5) + s][n = 32w + (lane&31)]   (forward)
// WpkT[layer][w][t][lane][s] = W[layer][k = 32w + (lane&31)][n = 8t + 4*(lane>>5) + s]  (dX = dP W^T)
// (packed by pack_bodies.cuh: PK_EDGE_F32)
static_assert(FH == pk::FHd, "pack_bodies.cuh");

struct EdgeFwdArgs {
  int64_t n_edges;
  const float* d_src;
  const float* d_eff;
  const float* centers;
  float neg_inv_gap;
  float neg_inv_gap_log2e;   // -log2(e)/gap
  const float* Wpk;       // [3][4][16][64][4]
  const float* bh[3];     // hidden biases
  const float* Wo;        // [128][E]
  const float* bo;        // [E]
  float* e_out;           // [n_edges][E]
  float* z_save;          // [3][n_edges][128] or nullptr
  float* dummy;           // 128 floats: where rows past the end store
  int tape_blocked;       // z_save layout (edge_fused.h: edge_tape_blocked): 1 = the split-operand kernels' blocked form
  RangeGuard guard;       // word != nullptr: run only if the guard carries this epoch (fallback of edge_fwd_h2)
  // live-edge view (ng_internal.h: LiveEdges): rows = compacted live slots, n_edges = slot count, row count = *n_live
  const int32_t* perm;
  const int32_t* n_live;
  int64_t z_layer_stride; // floats between the layers of the tape
};

__device__ __forceinline__ void load_wfrag(float (&wf)[64], const float* __restrict__ Wpk, int layer,
                                           int wave, int lane) {
  const float4* p = reinterpret_cast<const float4*>(Wpk) + ((layer * 4 + wave) * 16) * 64 + lane;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 v = p[t * 64];
    wf[4 * t + 0] = v.x; wf[4 * t + 1] = v.y; wf[4 * t + 2] = v.z; wf[4 * t + 3] = v.w;
  }
}

// one hidden layer for this wave's 32-column slab: Xout[:, slab] = softplus(Xin W[:, slab] + b).
// Rows 0-31 and 32-63 are two accumulator chains; after them the slab registers are dead: the NEXT layer's
// slab is loaded into them, its L2 latency hiding under the softplus epilogue.
// fp32 MFMA and VALU share the SIMD's issue (tools/ubench: no overlap, not even across the two waves of a
// SIMD), so every epilogue instruction is matrix time lost.  The bias therefore enters as the INITIAL
// accumulator value (no add per element), and the non-transcendental half of the softplus runs on the
// packed-fp32 ALU:  softplus(x) = max(x,0) + ln2 * log2(1 + 2^(-|x| log2 e))
//   per pair: 2 v_mul(|x|) + 2 v_exp + 1 v_pk_add + 2 v_log + 2 v_max + 1 v_pk_fma
typedef float f32x2e __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void softplus2(float x0, float x1, float& y0, float& y1) {
  f32x2e t = {__builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x0)),
              __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x1))};
  t = t + f32x2e{1.0f, 1.0f};
  const f32x2e l = {__builtin_amdgcn_logf(t[0]), __builtin_amdgcn_logf(t[1])};
  const f32x2e m = {fmaxf(x0, 0.0f), fmaxf(x1, 0.0f)};
  const f32x2e r = __builtin_elementwise_fma(l, f32x2e{0.6931471805599453f, 0.6931471805599453f}, m);
  y0 = r[0]; y1 = r[1];
}

__device__ __forceinline__ void epilogue_q(const f32x16& acc, int q, float* __restrict__ dst) {
  float4 v;
  softplus2(acc[4 * q + 0], acc[4 * q + 1], v.x, v.y);
  softplus2(acc[4 * q + 2], acc[4 * q + 3], v.z, v.w);
  *reinterpret_cast<float4*>(dst + 8 * q) = v;
}

__device__ __forceinline__ void hidden_layer(float (&wf)[64], const float* __restrict__ Xin,
                                             float* __restrict__ Xout,
                                             const float* __restrict__ bias, int wave, int rbase, int lane,
                                             const float* __restrict__ Wpk, int next_layer) {
  // wave = column slab (0..3); rbase = first of this wave's 64 rows in the tile
  const int half = lane >> 5, l31 = lane & 31;
  const float* x0 = Xin + (rbase + l31) * FLD + half * 4;
  const float* x1 = x0 + 32 * FLD;
  // lane holds, for rows l31 and 32+l31, columns 32*wave + 8q + 4*half + (0..3)
  const int ncol = 32 * wave + 4 * half;
  float* o0 = Xout + (rbase + l31) * FLD + ncol;
  float* o1 = o0 + 32 * FLD;
  f32x16 acc0, acc1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {          // accumulators start at the bias of their output column
    const float4 bv = *reinterpret_cast<const float4*>(bias + ncol + 8 * q);
    acc0[4 * q + 0] = bv.x; acc0[4 * q + 1] = bv.y; acc0[4 * q + 2] = bv.z; acc0[4 * q + 3] = bv.w;
    acc1[4 * q + 0] = bv.x; acc1[4 * q + 1] = bv.y; acc1[4 * q + 2] = bv.z; acc1[4 * q + 3] = bv.w;
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(x0 + 8 * t);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 0], a.x, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 1], a.y, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 2], a.z, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 3], a.w, acc0, 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 b = *reinterpret_cast<const float4*>(x1 + 8 * t);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 0], b.x, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 1], b.y, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 2], b.z, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[4 * t + 3], b.w, acc1, 0, 0, 0);
  }
  load_wfrag(wf, Wpk, next_layer, wave, lane);
  // both epilogues after the chains: fp32 MFMA and VALU do not overlap, and a VALU group placed between two
  // dependent MFMAs costs more than the same group behind the chain (measured 1.826 -> 1.808 ms)
#pragma unroll
  for (int q = 0; q < 4; ++q) epilogue_q(acc0, q, o0);
#pragma unroll
  for (int q = 0; q < 4; ++q) epilogue_q(acc1, q, o1);
}

// copy a finished [64][128] LDS tile to global as whole rows (wave w: rows 16w .. 16w+15)
// Rows past the end go to a dummy row instead of being skipped: with a fixed number of stores per tile the
// compiler's vmcnt counts for the next layer's weight slab stay exact (otherwise the last waits of the
// MFMA chain also wait for these stores to be acknowledged).
// float offset of (edge gr, features col..col+3) in a tape layer: row-major, or — inside every FULL group of 32 edges of
// a blocked tape — the split-operand kernels' register order (edge_fused.h)
__device__ __forceinline__ int64_t tape_offset(int64_t gr, int col, int64_t n_rows, int blocked) {
  const int64_t g = gr >> 5;
  if (blocked && (g + 1) * 32 <= n_rows) {
    const int bo = col >> 5, q = (col >> 3) & 3, hf = (col >> 2) & 1, r = (int)(gr & 31);
    return g * 4096 + ((bo * 4 + q) * 64 + hf * 32 + r) * 4;
  }
  return gr * FH + col;
}

__device__ __forceinline__ void save_tile(const float* __restrict__ X, float* __restrict__ dst,
                                          float* __restrict__ dummy, int64_t row0, int64_t n_rows, int wave,
                                          int lane, int blocked) {
  const int col = (lane & 31) * 4;
  typedef float nt4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = 16 * wave + 2 * i + (lane >> 5);
    const float4 v = *reinterpret_cast<const float4*>(X + r * FLD + col);
    float* d = row0 + r < n_rows ? dst + tape_offset(row0 + r, col, n_rows, blocked) : dummy + col;
    __builtin_nontemporal_store(nt4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt4*>(d));
  }
}

// TM = rows per tile = 16 per wave.  TM = 64: 256 threads, two workgroups per CU (their phases drift apart,
// which hides latency but not VALU work: beside an fp32-MFMA stream the partner wave issues nothing).
// TM = 128: 512 threads, ONE workgroup per CU whose two waves per SIMD run the SAME phase: in the VALU
// phases (softplus, RBF, output layer, z_save copies) two waves issue twice as fast as one.
template <int E, bool SAVE, int TM>
__global__ __launch_bounds__(TM * 4, TM == 64 ? 2 : 1) void edge_fused_fwd_kernel(EdgeFwdArgs a) {
  constexpr int NTHR = TM * 4;
  if (a.guard.word && !range_guard_raised(a.guard)) return;      // fallback launch: nothing went out of range
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X0 = smem;                         // [TM][132]
  float* X1 = smem + TM * FLD;              // [TM][132]
  float* sWo = X1 + TM * FLD;               // [128*E]
  float* sMask = sWo + FH * FMAX_E;         // [TM]
  float* sCen = sMask + TM;                 // [128] RBF centres
  float* sBias = sCen + FH;                 // [3][128] hidden biases (LDS: keeps them out of VGPRs)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = wave & 3, rbase = 64 * (wave >> 2);
  for (int t = tid; t < FH * E; t += NTHR) sWo[t] = a.Wo[t];
  if (tid < FH) {
    sCen[tid] = a.centers[tid];
    sBias[tid] = a.bh[0][tid];
    sBias[FH + tid] = a.bh[1][tid];
    sBias[2 * FH + tid] = a.bh[2][tid];
  }
  __syncthreads();

  if (a.perm && *a.n_live < 0) return;      // (a negative row count: nothing to do, e_out is somebody else's — edge_fwd_h2.hip)
  const int64_t n_edges = a.perm ? (int64_t)*a.n_live : a.n_edges;
  const int64_t ntiles = (n_edges + TM - 1) / TM;
  if (a.perm) {     // dead slots carry e == 0
    for (int64_t i = n_edges + (int64_t)blockIdx.x * NTHR + tid; i < a.n_edges; i += (int64_t)gridDim.x * NTHR) {
      const int64_t g = a.perm[i];
      for (int n = 0; n < E; ++n) a.e_out[g * E + n] = 0.f;
    }
  }
  float wf[64];
  load_wfrag(wf, a.Wpk, 0, slab, lane);

  // distances of the NEXT tile are requested one tile ahead (clamped index, no conditional VMEM): read at
  // the point of use they exposed a full HBM latency at the head of every tile
  float ds_n, de_n;
  {
    const int64_t g0 = std::max<int64_t>(std::min<int64_t>((int64_t)blockIdx.x * TM + (tid & (TM - 1)), n_edges - 1), 0);
    ds_n = a.d_src[g0]; de_n = a.d_eff[g0];
  }
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * TM;
    // ---- RBF tile: thread -> (row = tid % TM, quarter = tid / TM): 32 centres each
    {
      const int r = tid & (TM - 1), qt = tid / TM;
      const int64_t gr = row0 + r;
      const float ds = gr < n_edges ? ds_n : 0.f;
      const float de = de_n;
      const float m = ds > 0.f ? 1.f : 0.f;
      if (qt == 0) sMask[r] = m;
      // masked edges: a distance of 1e19 makes every (d - mu)^2 * c overflow to -inf and exp2 return an
      // exact 0 — the mask costs nothing per element.  c2 = -log2(e)/gap folds __expf's scaling.
      const float dm = ds > 0.f ? de : 1.0e19f;
      const f32x2e d2 = {dm, dm};
      const f32x2e c2 = {a.neg_inv_gap_log2e, a.neg_inv_gap_log2e};
      float* dst = X0 + r * FLD + 32 * qt;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 mu = *reinterpret_cast<const float4*>(sCen + 32 * qt + 4 * i);
        f32x2e u0 = d2 - f32x2e{mu.x, mu.y};
        f32x2e u1 = d2 - f32x2e{mu.z, mu.w};
        u0 = (u0 * u0) * c2;
        u1 = (u1 * u1) * c2;
        float4 v;
        v.x = __builtin_amdgcn_exp2f(u0[0]); v.y = __builtin_amdgcn_exp2f(u0[1]);
        v.z = __builtin_amdgcn_exp2f(u1[0]); v.w = __builtin_amdgcn_exp2f(u1[1]);
        *reinterpret_cast<float4*>(dst + 4 * i) = v;
      }
    }
    NG_LDS_BARRIER();
    // ---- hidden layer 0: X0 -> X1
    hidden_layer(wf, X0, X1, sBias, slab, rbase, lane, a.Wpk, 1);
    {   // issued here they are younger than layer 1's weight slab and a full layer older than layer 2's
      const int64_t gn = std::min<int64_t>((tile + gridDim.x) * TM + (tid & (TM - 1)), n_edges - 1);
      ds_n = a.d_src[gn]; de_n = a.d_eff[gn];
    }
    NG_LDS_BARRIER();
    if (SAVE) save_tile(X1, a.z_save, a.dummy, row0, n_edges, wave, lane, a.tape_blocked);
    // ---- hidden layer 1: X1 -> X0
    hidden_layer(wf, X1, X0, sBias + FH, slab, rbase, lane, a.Wpk, 2);
    NG_LDS_BARRIER();
    if (SAVE) save_tile(X0, a.z_save + a.z_layer_stride, a.dummy, row0, n_edges, wave, lane, a.tape_blocked);
    // ---- hidden layer 2: X0 -> X1   (reloads layer 0's slab for the next tile)
    hidden_layer(wf, X0, X1, sBias + 2 * FH, slab, rbase, lane, a.Wpk, 0);
    NG_LDS_BARRIER();
    if (SAVE) save_tile(X1, a.z_save + 2 * a.z_layer_stride, a.dummy, row0, n_edges, wave, lane, a.tape_blocked);
    // ---- output layer: wave w -> rows 16w..16w+15, 4 lanes per row (k = 16i + 4*(lane&3) + s)
    {
      const int r = 16 * wave + (lane >> 2), qq = lane & 3;
      float acc[E];
#pragma unroll
      for (int n = 0; n < E; ++n) acc[n] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 16 * i + 4 * qq;
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
      }
      const int64_t gr = row0 + r;
      if (qq == 0 && gr < n_edges) {
        const float m = sMask[r];
        const int64_t go = a.perm ? (int64_t)a.perm[gr] : gr;     // (a dependent load: this kernel is the strict-fp32 / fallback form)
#pragma unroll
        for (int n = 0; n < E; ++n) a.e_out[go * E + n] = m * (acc[n] + a.bo[n]);
      }
    }
    // (the barrier after the next tile's RBF phase orders these X1 / sMask reads before they are
    //  overwritten: X1 is next written in layer 0's epilogue, sMask in the RBF phase — see below)
    NG_LDS_BARRIER();
  }
}

PackJob edge_fused_pack_job(const float* const* W, float* Wpk, float* WpkT) {
  PackJob j;
  j.kind = PK_EDGE_F32; j.blocks = 48;
  j.src[0] = W[0]; j.src[1] = W[1]; j.src[2] = W[2]; j.dst[0] = Wpk; j.dst[1] = WpkT;
  return j;
}

int edge_fused_pack(ng_ctx* ctx, hipStream_t st, const float* const* W, float* Wpk, float* WpkT) {
  return pack_launch(ctx, st, edge_fused_pack_job(W, Wpk, WpkT));
}

bool edge_fused_supported(int H, int E, int Le) { return H == FH && Le == 4 && E >= 1 && E <= FMAX_E; }

int edge_fused_fwd(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src,
                   const float* d_eff, const float* centers, float gap, const float* const* W,
                   const float* const* b, float* e_out, float* z_save, LiveEdges live) {
  if (edge_split_enabled()) return edge_h2_fwd(ctx, st, n_edges, E, d_src, d_eff, centers, gap, W, b, e_out, z_save, live);
  return edge_fused_fwd_f32(ctx, st, n_edges, E, d_src, d_eff, centers, gap, W, b, e_out, z_save, false, nullptr, live);
}

// f32-input MFMA forward.  guard != nullptr: the range fallback of edge_h2_fwd — same launches, but the kernel's
// workgroups return at once unless the split-operand kernel raised the guard; the tape is then written in the layout
// that kernel would have used.  Its fragment copy lives in the AUX scratch: the main workspace may hold the
// split-operand image the first kernel is still reading.
int edge_fused_fwd_f32(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src,
                       const float* d_eff, const float* centers, float gap, const float* const* W,
                       const float* const* b, float* e_out, float* z_save, bool tape_blocked, const RangeGuard* guard,
                       LiveEdges live) {
  // scratch: fragment-ordered copy of the three hidden weight matrices
  const size_t pk_floats = (size_t)3 * FH * FH;
  // (frozen weights: the fragment copy is kept like every other packed image — one launch less per call)
  bool have = false;
  float* Wpk = (float*)cached_image(ctx, W[0], 11, (pk_floats + FH) * 4, &have);
  const bool cached = Wpk != nullptr;
  if (!Wpk) Wpk = (float*)(guard ? aux_workspace(ctx, (pk_floats + FH) * 4) : workspace(ctx, (pk_floats + FH) * 4));
  if (!Wpk) return NG_ERR_NOMEM;
  if (!have) {
    const PackJob j = edge_fused_pack_job(W, Wpk, nullptr);
    if (int rc = pack_launch(ctx, st, j)) return rc;
    if (cached) cache_set_job(ctx, W[0], 11, j);
  }
  EdgeFwdArgs a;
  a.tape_blocked = tape_blocked ? 1 : 0;
  a.guard = guard ? *guard : RangeGuard{nullptr, 0};
  a.n_edges = n_edges; a.d_src = d_src; a.d_eff = d_eff; a.centers = centers;
  a.neg_inv_gap = (float)(-1.0 / (double)gap);
  a.neg_inv_gap_log2e = (float)(-1.4426950408889634 / (double)gap);
  a.Wpk = Wpk;
  a.bh[0] = b[0]; a.bh[1] = b[1]; a.bh[2] = b[2];
  a.Wo = W[3]; a.bo = b[3];
  a.e_out = e_out; a.z_save = z_save; a.dummy = Wpk + pk_floats;
  a.perm = live.perm; a.n_live = live.n_live; a.z_layer_stride = n_edges * FH;
  // 64-edge tiles, two 256-thread workgroups per CU
  const int TMr = 64;
  const int64_t ntiles = cdiv(n_edges, TMr);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu * 2);
  const size_t lds = (size_t)(2 * TMr * FLD + FH * FMAX_E + TMr + 4 * FH) * 4;
  ProfScope ps(ctx, st, guard ? "edge_fwd_range_fallback" : "edge_fused_fwd");
#define NG_FW1(EE, SV) hipLaunchKernelGGL((edge_fused_fwd_kernel<EE, SV, 64>), dim3(grid), dim3(256), lds, st, a);
#define NG_FW(EE)                                                                                  \
  case EE:                                                                                         \
    if (z_save) { NG_FW1(EE, true) } else { NG_FW1(EE, false) }                                    \
    break;
  switch (E) { NG_FW(1) NG_FW(2) NG_FW(3) NG_FW(4) NG_FW(5) NG_FW(6) NG_FW(7) NG_FW(8) }
#undef NG_FW
#undef NG_FW1
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

}  // namespace ng
