// Shared pieces of the split-operand edge backward kernels: edge_bwd_h2.hip (eight waves, two per SIMD) and
// edge_bwd_rs.hip (sixteen waves, role-split).  Image geometry, tape loads, the two GEMM bodies.
#pragma once
#include <algorithm>
#include <string>
#include <type_traits>
#include <cstdio>

#include "edge_fused.h"
#include "h2_common.cuh"
#include "pack_bodies.cuh"

namespace ng {

typedef short s16x4 __attribute__((ext_vector_type(4)));
#define HX_LDS(T) __attribute__((address_space(3))) T

constexpr int HX_THREADS = 512;
// Image geometry.  G images are read as rows (ds_read_b128: stride must be a multiple of 16 B; 272 B = 68 dwords
// puts 16 consecutive rows on disjoint 4-bank slots) and as columns (transposing reads); the Z image only as
// columns (stride 264 B: the 8-B piece writes of 32 rows then cost the minimum of 2 LDS cycles).  A transposing
// read fetches 4 consecutive edges x 32 B per 16-lane group, and the LDS serves 32 lanes (two groups, the two 32-B
// column blocks of a slab) per clock over 64 banks: the ROWS are stored permuted so that the 4 edges of a quad sit 16
// banks apart (hx_prow_*: 4 rows of 272 B, 8 rows of 264 B) and the two groups interleave in 8-bank runs.  Measured
// (tools/ubench/trbank2.hip, ns per wave read with 8 waves reading): 9.8 this way, 14.8 with the quad 8 banks apart
// (the G images until late round 2), 27.3 unpermuted.
constexpr int HX_ROWG = 272, HX_ROWZ = 264;
constexpr int HX_PIECE_G = FTM * HX_ROWG, HX_PIECE_Z = FTM * HX_ROWZ;
constexpr int HX_IMG_G = 2 * HX_PIECE_G;     // 34,816 B
constexpr int HX_IMG_Z = 2 * HX_PIECE_Z;     // 33,792 B
constexpr int HX_STG = 132;                  // fp32 staging row stride (floats)
constexpr int HX_MISC_FLOATS = FH * 4 + FTM * 4 + FH;   // sWo4 | sdE | sCen
constexpr int HX_IMGS = 2 * HX_IMG_Z + 2 * HX_IMG_G;     // two Z-type images (the fp16 pieces freed the room)
#ifdef HX_STAMP
constexpr int HX_LDS_BYTES = HX_IMGS + HX_MISC_FLOATS * 4 + 1024;
#else
constexpr int HX_LDS_BYTES = HX_IMGS + HX_MISC_FLOATS * 4;
#endif             // 140,544 of 163,840
constexpr int HX_WS = 8;                     // log2 of the W^T scale
constexpr float HX_WSCALE = (float)(1 << HX_WS), HX_WINV = 1.0f / (float)(1 << HX_WS);

// physical row of edge e (0..63) in the Z image / in a G image
__device__ __forceinline__ int hx_prow_z(int e) {
  const int hi = e >> 4, a = (e >> 2) & 3, b = e & 3;
  return 2 * (a + 4 * b) + (hi & 1) + 32 * (hi >> 1);
}
__device__ __forceinline__ int hx_prow_g(int e) {
  const int hi = e >> 4, a = (e >> 2) & 3, b = e & 3;
  return 16 * hi + 4 * b + a;
}

struct EdgeBwdH2Args {
  int64_t n_edges;
  const float* d_src;
  const float* d_eff;
  const float* centers;
  float neg_inv_gap_log2e;
  const char* wt_img;   // [2 layers (W2, W3)][4 k-slabs][8 k-steps][2 pieces][1 KB], pieces of 2^8 W
  const float* blockmax;   // per-block max |de| (hx_absmax_kernel); with the row-sum bounds behind wt_img the kernel forms {S, 1/S}
  int n_blockmax;          // < 0: small call, no hx_absmax launch — blockmax IS de and every workgroup takes max |de| over its -n_blockmax values itself
  const float* Wo;      // [128][E]
  const float* z_save;  // [3][z_layer_stride / 128 edges][128], first edge of THIS launch's segment
  int64_t z_layer_stride;  // floats between the layers of the tape (= total edges * 128; a launch covers one segment)
  const float* de;      // [n_edges][E]
  float* partial;       // [grid][part_stride], layout of edge_fused_bwd.hip
  int part_stride;
  int E;
  int tape_blocked;     // z_save layout: 1 = blocked inside full 32-edge groups (edge_fused.h), 0 = row-major
  unsigned long long* stamps;
  RangeGuard guard;     // raised when a partial comes out non-finite (an operand left the fp16 range)
  // live-edge view (ng_internal.h: LiveEdges; kernel template LIVE): the rows of this launch are the compacted live
  // slots [row_base, row_base + n_edges) clipped to *n_live; d_eff and the tape are compacted (this segment's part), `de`
  // is the caller's full [n_slots][E] array, row r reads de[perm[r]]; d_src is not read (every row is live)
  const int32_t* perm;     // first row of THIS segment
  const int32_t* n_live;
  int64_t row_base;
};

// W^T fragments of the dZ GEMMs (packed by pack_bodies.cuh: PK_EDGE_WT): lane (row k = 32 zk + (l&31), k-slot t) =
// piece_p( 2^8 W[k][n = 16 ks + 8 (l>>5) + t] )
static_assert(HX_WSCALE == pk::WSCALE && FH == pk::FHd, "pack_bodies.cuh");

// 16 values of one row (columns col0 + 8q + j, v[4q + j]) -> the two piece planes of an image
template <int ROWB>
__device__ __forceinline__ void hx_img_write(char* __restrict__ img, int row, int col0, const float (&v)[16]) {
  constexpr int HX_ROWB = ROWB, HX_PIECE = FTM * ROWB;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned h0, l0, h1, l1;
    split2_pair(v[4 * q + 0], v[4 * q + 1], h0, l0);
    split2_pair(v[4 * q + 2], v[4 * q + 3], h1, l1);
    char* p = img + row * HX_ROWB + (col0 + 8 * q) * 2;
    *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(p + HX_PIECE) = u32x2{l0, l1};
  }
}

// one quarter of hx_img_write: columns col0 + 8q .. + 3 of one row
template <int ROWB>
__device__ __forceinline__ void hx_img_write_q(char* __restrict__ img, int row, int col0, int q, float v0, float v1,
                                               float v2, float v3) {
  unsigned h0, l0, h1, l1;
  split2_pair(v0, v1, h0, l0);
  split2_pair(v2, v3, h1, l1);
  char* p = img + row * ROWB + (col0 + 8 * q) * 2;
  *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
  *reinterpret_cast<u32x2*>(p + FTM * ROWB) = u32x2{l0, l1};
}

// one MFMA operand (8 consecutive edges of this lane's column) = two transposing reads of 4 edges each; `step` is
// the byte distance between the two quads' first rows
__device__ __forceinline__ u32x4 hx_tr_frag(const char* p, int step) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((HX_LDS(s16x4)*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((HX_LDS(s16x4)*)(p + step));
  const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
  return u32x4{a[0], a[1], b[0], b[1]};
}

struct HxDwFrags { u32x4 b[2], a0[2], a1[2]; };

// operands of k-step ks (edges 16 ks .. 16 ks + 15): physical rows per hx_prow_z / hx_prow_g
__device__ __forceinline__ void hx_dw_load(HxDwFrags& f, const char* zb, const char* g0, int ks) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f.b[p] = hx_tr_frag(zb + p * HX_PIECE_Z + ((ks & 1) + 32 * (ks >> 1)) * HX_ROWZ, 2 * HX_ROWZ);
    f.a0[p] = hx_tr_frag(g0 + p * HX_PIECE_G + 16 * ks * HX_ROWG, HX_ROWG);
    f.a1[p] = hx_tr_frag(g0 + 64 + p * HX_PIECE_G + 16 * ks * HX_ROWG, HX_ROWG);
  }
}

// acc[j][n][k] += sum_edges G[e][n] Zin[e][k]   (D rows n = G columns of slab nsl0 + j, D cols k = Zin columns of kslab)
// Bias gradient on the matrix pipe: db[n] = sum_e G[e][n] = G^T x ones.  In the step ks == kslab (the four waves
// that share an n-slab pair split the tile's edges) the G fragments are multiplied once more by a B operand that
// is 1.0 in five columns (5c .. 5c+4, c = 2 layer + j) and 0 elsewhere: ONE accumulator collects all six
// (layer, n-slab) column sums in disjoint column groups — 4 MFMAs per layer instead of 80 DPP adds + an LDS update.
__device__ __forceinline__ void hx_dw_gemm(f32x16 (&acc)[2], f32x16& accB, int cbase, const char* __restrict__ imgZ,
                                           const char* __restrict__ imgG, int kslab, int nsl0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  // the 16-lane group reads a [4 edges][16 columns] block: lane i supplies edge (i>>2) of the quad, columns 4(i&3)..+3
  const char* zb = imgZ + (4 * (g >> 1) + 8 * (i >> 2)) * HX_ROWZ + (16 * (g & 1) + 4 * (i & 3)) * 2 + 64 * kslab;
  const char* g0 = imgG + (4 * (i >> 2) + 2 * (g >> 1)) * HX_ROWG + (16 * (g & 1) + 4 * (i & 3)) * 2 + 64 * nsl0;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    HxDwFrags c;       // single-buffered: the kernel has no registers for a second set; the partner wave covers the LDS latency
    hx_dw_load(c, zb, g0, ks);
    mma3_2a(c.a0, c.a1, c.b, acc[0], acc[1]);
    if (ks == kslab) {
      // the ones operand is rebuilt here from the lane id (the asm keeps the compiler from hoisting it out of the tile
      // loop, where it became a spilled invariant whose reload carried a vmcnt(0) into the middle of the prefetches)
      int lv = lane;
      asm volatile("" : "+v"(lv));
      const unsigned grp = (unsigned)((lv & 31) / 5);
      const unsigned o0 = grp == (unsigned)cbase ? 0x3C003C00u : 0u, o1 = grp == (unsigned)(cbase + 1) ? 0x3C003C00u : 0u;
      const u32x4 ones0 = {o0, o0, o0, o0}, ones1 = {o1, o1, o1, o1};
#pragma unroll
      for (int p = 1; p >= 0; --p) { accB = mfma_f16(c.a0[p], ones0, accB); accB = mfma_f16(c.a1[p], ones1, accB); }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void hx_wload(u32x4 (&w)[2], __amdgpu_buffer_rsrc_t wrs, int wvo, int wso, int ks) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, wso + (ks * 2 + p) * 1024, 0);
    w[p] = __builtin_bit_cast(u32x4, raw);
  }
}

// dZ[e][k] = sum_n G[e][n] W[k][n]  for k-slab zk, edge rows 32 zrt..; D rows = k, D cols = edges.
// Returned lane layout: edge 32 zrt + (l&31), columns 32 zk + 8q + 4 (l>>5) + j  in register 4q + j.
// w0 holds the W^T fragments of step 0 (requested by the caller before the preceding GEMM); steps ks+1 .. ks+3 are
// in flight while step ks multiplies.
// fill(ks): elementwise work of the caller that does not depend on this product, handed in per k-step so that it
// issues in the shadow of the step's three MFMAs (an image build after the GEMM is VALU / LDS-store time nothing overlaps:
// the partner wave of the SIMD runs its own copy of the same code).
template <class F>
__device__ __forceinline__ void hx_dz_gemm(float (&out)[16], const char* __restrict__ imgG, int prow_g,
                                           __amdgpu_buffer_rsrc_t wrs, const u32x4 (&w0)[2], int L, int zk, int lane,
                                           F&& fill) {
  const int half = lane >> 5;
  const char* gb = imgG + prow_g * HX_ROWG + 16 * half;
  const int wvo = lane * 16;
  int wso = ((L * 4 + zk) * 8) * 2 * 1024;
  // opaque to the optimizer: otherwise the 2 x 24 fragment offsets (wso + const) are hoisted out of the tile loop as
  // scalar invariants, spilled into VGPR lanes and fetched back with v_readlane + s_nop 4 in front of every load
  asm volatile("" : "+s"(wso));
  f32x16 acc0;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
  u32x4 wa[4][2], b[2][2];     // W^T fragments three steps ahead (L2 latency is ~6 steps of 3 MFMAs), G rows one step ahead
#pragma unroll
  for (int p = 0; p < 2; ++p) { wa[0][p] = w0[p]; b[0][p] = *reinterpret_cast<const u32x4*>(gb + p * HX_PIECE_G); }
  hx_wload(wa[1], wrs, wvo, wso, 1);
  hx_wload(wa[2], wrs, wvo, wso, 2);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    if (ks < 5) hx_wload(wa[(ks + 3) & 3], wrs, wvo, wso, ks + 3);
    if (ks < 7) {
#pragma unroll
      for (int p = 0; p < 2; ++p) b[(ks + 1) & 1][p] = *reinterpret_cast<const u32x4*>(gb + 32 * (ks + 1) + p * HX_PIECE_G);
    }
    acc0 = mma3(wa[ks & 3], b[ks & 1], acc0);
    fill(ks);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) out[r] = acc0[r] * HX_WINV;     // the W^T pieces carry 2^8
}

// this lane's 16 values of a saved activation row (clamped row: rows past the end multiply a zero gradient).
// Buffer loads: scalar resource + one 32-bit lane offset — per-lane 64-bit pointers for five arrays got spilled and
// every reload put a vmcnt(0) into the middle of the prefetch.
// Tape layout (edge_fused.h: edge_tape_blocked): inside a FULL 32-edge group the block this wave needs is stored in
// exactly this register layout — four contiguous 1-KB wave loads; the last partial group is row-major (16-B pieces
// of 32 rows, 32 B apart).  voff: this lane's byte offset, qbytes: distance between its four loads (wave-uniform).
__device__ __forceinline__ void hx_load_z(float (&z)[16], __amdgpu_buffer_rsrc_t rs, int voff, int qbytes) {
  typedef float f32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, q * qbytes, 0);
    const f32x4v v = __builtin_bit_cast(f32x4v, raw);
    z[4 * q + 0] = v[0]; z[4 * q + 1] = v[1]; z[4 * q + 2] = v[2]; z[4 * q + 3] = v[3];
  }
}

// x * s'(.) with s' = 1 - exp(-z) recovered from the softplus OUTPUT z:  x - x * 2^(-z log2 e)   (mul, exp, fma)
__device__ __forceinline__ float hx_sprime(float x, float z) {
  return fmaf(-x, __builtin_amdgcn_exp2f(-1.4426950408889634f * z), x);
}

// the same for two values.  NOT packed: tools/ubench/mfma_fill.hip (round 3) shows a v_pk_{add,mul,fma}_f32 that issues while
// the SIMD's matrix pipe is busy — this wave's MFMAs or the partner wave's — costs ~20 cycles, a plain VALU op ~0.5 (up to
// five of them hide in every 32-cycle MFMA slot).  The file is built with -fno-slp-vectorize for the same reason.
__device__ __forceinline__ void hx_sprime2(float& x0, float& x1, float z0, float z1) {
  x0 = hx_sprime(x0, z0);
  x1 = hx_sprime(x1, z1);
}

// the sixteen-wave role-split kernel (edge_bwd_rs.hip): same arguments, same partial layout

}  // namespace ng
