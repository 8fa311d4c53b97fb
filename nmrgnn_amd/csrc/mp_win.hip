// Window-resident fused MPLayer kernels for atom_feature_size == 64, edge_feature_size <= 3.
// Reference: nmrgnn/layers.py:26-46 (MPLayer.call) + residual of nmrgnn/model.py:165-167.
//
//   h'[i,:] = act( v_i * sum_{n,l} A[i,n,l] W[l,:,n] ) + h[i,:],   A[i,n,l] = sum_j e[i,j,n] h[nlist[i,j], l]
//
// Molecule batches have index-local neighbour lists (a neighbour of atom i lies in i's own graph,
// a few hundred rows around i).  One persistent 512-thread workgroup per CU walks a contiguous run of
// 32-atom tiles and keeps a WINDOW of 288 consecutive h rows (72 KB) in LDS, so every gathered row is
// read from HBM/L2 once per run instead of once per referencing edge, and the [N, E*64] aggregate never
// touches HBM: it goes from the gather straight into an LDS tile that the matrix cores consume.
//
// Per tile, all eight waves run the same two phases, separated by barriers:
//   gather   16 lanes per atom, 4 atoms per wave: list reads from LDS, one ds_read_b128 per neighbour
//            from the window, v_pk_fma_f32 accumulation, aggregate written to the LDS tile;
//            the lists of the NEXT tile go registers -> LDS here and the ones after that are requested
//            from global memory (one full tile of latency cover).
//   matrix   [32 x E*64] x [E*64 x 64] on v_mfma_f32_16x16x4_f32 — wave w owns output columns
//            16(w&3).. of atoms 16(w>>2).., weight slab resident in registers — then the epilogue
//            (inv_degree, activation, residual) from the accumulators.
// Measured on gfx950 (tools/ubench): an fp32 MFMA stream and VALU work do NOT overlap on a SIMD, neither
// across its two waves nor inside one wave — fp32 MFMA runs at the vector rate on shared issue — so there
// is nothing to gain from giving waves different roles; what matters is the VALU instruction count
// (packed FMAs, vector list traffic) and that the partner wave covers LDS latency.
//
// When a tile's lists leave the window it is restaged, centred on the referenced range; a tile whose
// range is wider than the window (whole proteins with long-range contacts, or raw zero-padded lists)
// gathers from global memory instead — same kernel, workgroup-uniform branch.  Hosts that want the
// window path point padded slots at the atom itself (their edge features are exactly 0).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "edge_fused.h"   // NG_LDS_BARRIER
#include "h2_common.cuh"
#include "pack_bodies.cuh"

namespace ng {

constexpr int WF = 64;          // feature width
constexpr int WTA = 32;         // atoms per tile
constexpr int WROWS = 288;      // window rows
constexpr int WC4 = WF / 4;     // float4 per row = lanes per atom
constexpr int WTHREADS = 512;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- weight images (packed by pack_bodies.cuh) ----
// f32 fragments for v_mfma_f32_16x16x4_f32 (A operand: row i = lane & 15, k = lane >> 4):
//   out[((ct*NT + T)*64 + lane)*4 + u] = Wsrc(k = 16T + 4(lane>>4) + u, o = 16ct + (lane&15))
//   mode 0 (forward):       Wsrc(k = n*64 + l, o = m) = w[l][m][n]
//   mode 1 (back to nodes): Wsrc(k = n*64 + m, o = l) = w[l][m][n]
//   mode 2 (dA = dP Wp^T):  Wsrc(k = m, o = n*64 + l) = w[l][m][n]
// fp16 piece fragments for v_mfma_f32_16x16x32_f16 (two-piece operands, h2_common.cuh), the same three index maps:
//   out[(((ct*NT2 + T)*2 + p)*64 + lane)*4 + j] = fp16 pair (t = 2j, 2j+1) of piece p of 2^8 Wsrc(k = 32T + 8(lane>>4) + t,
//   o = 16ct + (lane&15))
// A weight whose pieces (of 2^8 w) leave the fp16 range raises the call's guard and stores the image version into the
// image's flag word: the window kernels then take their fp32-input body (RangeGuard, ng_internal.h).
static_assert(WF == pk::WFd, "pack_bodies.cuh");

int mpw_pack(ng_ctx* ctx, hipStream_t st, int E, int mode, const float* w, float* out) {
  PackJob j;
  j.kind = PK_MPW_F32; j.blocks = 24; j.i0 = E; j.i1 = mode; j.src[0] = w; j.dst[0] = out;
  return pack_launch(ctx, st, j);
}

// every backward image of a layer in one launch: T / N piece fragments (+ the two f32 images of a guarded call)
PackJob mpw_bwd_job(int E, const float* w, float* outT, float* outN, float* f32T, float* f32N, unsigned* flag, RangeGuard guard) {
  PackJob j;
  j.kind = PK_MPW_BWD; j.blocks = f32T ? 96 : 48; j.i0 = E; j.src[0] = w;
  j.dst[0] = outT; j.dst[1] = outN; j.dst[2] = f32T; j.dst[3] = f32N; j.flag = flag; j.guard = guard;
  return j;
}

int mpw_pack2(ng_ctx* ctx, hipStream_t st, int E, const float* w, int mode_a, float* out_a, int mode_b, float* out_b) {
  if (int rc = mpw_pack(ctx, st, E, mode_a, w, out_a)) return rc;
  return mpw_pack(ctx, st, E, mode_b, w, out_b);
}

// ---- shared device pieces ---------------------------------------------------------------------------

// min over the 64 lanes, valid in lane 63: row_shr 1,2,4,8 inside rows of 16, then row_bcast 15 / 31
__device__ __forceinline__ int wave_min_i32(int v) {
  const int big = 0x7fffffff;
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x118, 0xf, 0xf, false));   // row_shr:8
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1,3
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2,3
  return v;
}

// acc += w * h for one float4 of features, as two v_pk_fma_f32
__device__ __forceinline__ void pk_axpy(f32x2& lo, f32x2& hi, float w, const float4& h) {
  const f32x2 ww = {w, w};
  lo = __builtin_elementwise_fma(ww, f32x2{h.x, h.y}, lo);
  hi = __builtin_elementwise_fma(ww, f32x2{h.z, h.w}, hi);
}

// Neighbour lists of one 32-atom tile in flight between global memory and LDS.  K % 4 == 0: whole
// 16-byte vectors (32K/4 int4 of indices, 32KE/4 float4 of edge features); otherwise scalars.
template <int E, bool K4>
struct WinLists {
  int4 nl4;
  float4 e4[2];
  int nl1[2];
  float e1[2][E];

  // Every thread issues the same number of loads (indices clamped into the arrays, out-of-range slots
  // zeroed afterwards): with no conditional VMEM in the tile loop the compiler's s_waitcnt counts stay
  // exact and nothing waits on the output stores still in flight.
  __device__ __forceinline__ void issue(const int32_t* __restrict__ nlist, const float* __restrict__ e,
                                        int64_t t, int K, int64_t N, int tid) {
    const int per_tile = WTA * K;
    const int64_t base = t * per_tile, lim = N * K;
    if (K4) {
      const int64_t nb = base / 4, nlim = lim / 4;              // int4 units
      const int64_t eb = base * E / 4, elim = lim * E / 4;      // float4 units
      const int64_t qn = nb + tid;
      const int4 z4 = make_int4(0, 0, 0, 0);
      const int4 v = reinterpret_cast<const int4*>(nlist)[qn < nlim ? qn : nlim - 1];
      nl4 = qn < nlim ? v : z4;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t qe = eb + tid + WTHREADS * u;
        const float4 w = reinterpret_cast<const float4*>(e)[qe < elim ? qe : elim - 1];
        e4[u] = qe < elim ? w : f4zero();
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t q = base + tid + WTHREADS * u;
        const bool ok = q < lim;
        const int64_t qc = ok ? q : lim - 1;
        const int v = nlist[qc];
        nl1[u] = ok ? v : 0;
#pragma unroll
        for (int n = 0; n < E; ++n) {
          const float w = e[qc * E + n];
          e1[u][n] = ok ? w : 0.f;
        }
      }
    }
  }

  // registers -> LDS; the row range over ALL slots goes to ctl[wave] / ctl[8 + wave]
  __device__ __forceinline__ void commit(int32_t* __restrict__ s_nl, float* __restrict__ s_e,
                                         int* __restrict__ ctl, int K, int tid, int wave, int lane) {
    const int per_tile = WTA * K;
    int lo = 0x7fffffff, hi = -1;
    if (K4) {
      const int nv = per_tile / 4, ev = per_tile * E / 4;
      if (tid < nv) {
        reinterpret_cast<int4*>(s_nl)[tid] = nl4;
        lo = min(min(nl4.x, nl4.y), min(nl4.z, nl4.w));
        hi = max(max(nl4.x, nl4.y), max(nl4.z, nl4.w));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + WTHREADS * u < ev) reinterpret_cast<float4*>(s_e)[tid + WTHREADS * u] = e4[u];
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = tid + WTHREADS * u;
        if (q < per_tile) {
          s_nl[q] = nl1[u];
#pragma unroll
          for (int n = 0; n < E; ++n) s_e[q * E + n] = e1[u][n];
          lo = min(lo, nl1[u]); hi = max(hi, nl1[u]);
        }
      }
    }
    lo = wave_min_i32(lo);
    hi = -wave_min_i32(-hi);
    if (lane == 63) { ctl[wave] = lo; ctl[8 + wave] = hi; }
  }
};

// every thread takes the same decision from the eight partial ranges; returns true when the window has
// to be restaged at the (updated) wlo.  mode: 0 = gather from the window, 1 = gather from global memory
__device__ __forceinline__ bool win_decide(const int* __restrict__ ctl, int& wlo, int& mode) {
  int lo = ctl[0], hi = ctl[8];
#pragma unroll
  for (int i = 1; i < 8; ++i) { lo = min(lo, ctl[i]); hi = max(hi, ctl[8 + i]); }
  mode = 0;
  if (hi < lo) return false;                                  // empty tile
  if (lo >= wlo && hi < wlo + WROWS) return false;            // window hit
  if (hi - lo + 1 > WROWS) { mode = 1; return false; }        // too wide
  wlo = max(0, lo - (WROWS - (hi - lo + 1)) / 2);
  return true;
}

__device__ __forceinline__ void win_stage(float4* __restrict__ win4, const float4* __restrict__ src4,
                                          int wlo, int64_t N, int tid) {
  float4 v[9];
#pragma unroll
  for (int u = 0; u < 9; ++u) {
    const int idx = tid + WTHREADS * u;
    const int64_t row = (int64_t)wlo + (idx >> 4);
    v[u] = row < N ? src4[row * WC4 + (idx & 15)] : f4zero();
  }
#pragma unroll
  for (int u = 0; u < 9; ++u) win4[tid + WTHREADS * u] = v[u];
}
static_assert(WROWS * WC4 == 9 * WTHREADS, "window staging assumes 9 float4 per thread");

// ---- aggregate tile in LDS -------------------------------------------------------------------------------
// H2 = false: fp32 rows [32][E*64 + 4].  H2 = true: two fp16 piece planes [2][32][E*64 + 8] (h2_common.cuh): the gather
// splits its sums where they are formed and the matrix phase runs on v_mfma_f32_16x16x32_f16 (18 instead of 48 MFMAs
// per wave and tile at E = 3); aggregates are O(1-10) activations and are split unscaled.
template <int E>
struct WinTile {
  static constexpr int KF = E * WF;
  static constexpr int LD = KF + 4;                    // fp32 row stride (floats)
  static constexpr int ROWB = (KF + 8) * 2;            // fp16 plane row stride (bytes): 16 rows on disjoint 4-bank groups
  static constexpr int PLANE = WTA * ROWB;
  static constexpr int BYTES_F32 = WTA * LD * 4, BYTES_H2 = 2 * PLANE;
};

template <int S>
__device__ __forceinline__ int ror_i(int v) {
  if (S == 0) return v;
  return __builtin_amdgcn_update_dpp(0, v, 0x120 + (S & 15), 0xf, 0xf, false);
}
template <int S>
__device__ __forceinline__ float ror_f(float v) {
  return __builtin_bit_cast(float, ror_i<S>(__builtin_bit_cast(int, v)));
}

// H2 rows carry a power-of-two scale when their largest entry reaches 2^15 (an aggregate of features is a forward quantity
// of any size — the reference's MPLayer is plain fp32): the 16 lanes of the row's DPP row hold all of it, the inverse goes
// to rs[atom] for the epilogue.  Every ordinary row has scale 1 and the same bits as without.
template <int E, bool H2>
__device__ __forceinline__ void tile_put(float* __restrict__ tb, int al, int c, f32x2 (&lo)[E], f32x2 (&hi)[E],
                                         float* __restrict__ rs) {
  if (H2) {
    float m = 0.f;
#pragma unroll
    for (int n = 0; n < E; ++n)
      m = fmaxf(fmaxf(m, fmaxf(fabsf(lo[n][0]), fabsf(lo[n][1]))), fmaxf(fabsf(hi[n][0]), fabsf(hi[n][1])));
    float rsv = 1.0f;
    if (__builtin_amdgcn_ballot_w64(m >= 32768.0f) != 0) {      // wave-uniform and never taken for ordinary activations
      m = fmaxf(m, ror_f<8>(m)); m = fmaxf(m, ror_f<4>(m)); m = fmaxf(m, ror_f<2>(m)); m = fmaxf(m, ror_f<1>(m));
      const int ef = (__builtin_bit_cast(int, m) >> 23) & 255;
      const bool big = ef >= 127 + 15 && ef != 255;
      const float S = big ? __builtin_bit_cast(float, (268 - ef) << 23) : 1.0f;       // 2^(14 - e): |S x| < 2^15
      rsv = big ? __builtin_bit_cast(float, (ef - 14) << 23) : 1.0f;
      const f32x2 S2 = {S, S};
#pragma unroll
      for (int n = 0; n < E; ++n) { lo[n] *= S2; hi[n] *= S2; }
    }
    if (c == 0) rs[al] = rsv;
    char* p = reinterpret_cast<char*>(tb) + al * WinTile<E>::ROWB + 8 * c;
#pragma unroll
    for (int n = 0; n < E; ++n) {
      unsigned h0, l0, h1, l1;
      split2_pair(lo[n][0], lo[n][1], h0, l0);
      split2_pair(hi[n][0], hi[n][1], h1, l1);
      *reinterpret_cast<u32x2*>(p + n * (WF * 2)) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(p + n * (WF * 2) + WinTile<E>::PLANE) = u32x2{l0, l1};
    }
  } else {
#pragma unroll
    for (int n = 0; n < E; ++n)
      *reinterpret_cast<float4*>(tb + al * WinTile<E>::LD + n * WF + 4 * c) = make_float4(lo[n][0], lo[n][1], hi[n][0], hi[n][1]);
  }
}

// ---- rotation gather (K <= 16, window mode) ------------------------------------------------------------
// Lane c of an atom's 16-lane row owns neighbour slot c: ONE index and E weights per lane instead of
// every lane reading the whole list (which cost as much LDS bandwidth as the row gather itself).  In
// step s the lane uses the slot of lane (c + s) mod 16, fetched over the DPP network (row_ror:s) —
// each lane walks the neighbours in its own rotated order, the sum is the same.  All sixteen lanes of
// a row read the SAME bank group (4c..4c+3) of sixteen DIFFERENT window rows: still conflict-free.
// four rotation steps: row reads and FMAs are separate so that the reads of the NEXT four steps can be
// issued before the FMAs of the current four (the LDS latency is otherwise exposed: both waves of a SIMD
// run this phase in lockstep and wait at the same time)
template <int S0>
__device__ __forceinline__ void rot_load4(const char* __restrict__ wbytes, int roff, float4 (&h)[4]) {
  h[0] = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 0>(roff));
  h[1] = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 1>(roff));
  h[2] = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 2>(roff));
  h[3] = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 3>(roff));
}
template <int E, int S0>
__device__ __forceinline__ void rot_fma4(const float4 (&h)[4], const float (&w)[E], f32x2 (&lo)[E], f32x2 (&hi)[E]) {
#pragma unroll
  for (int n = 0; n < E; ++n) pk_axpy(lo[n], hi[n], ror_f<S0 + 0>(w[n]), h[0]);
#pragma unroll
  for (int n = 0; n < E; ++n) pk_axpy(lo[n], hi[n], ror_f<S0 + 1>(w[n]), h[1]);
#pragma unroll
  for (int n = 0; n < E; ++n) pk_axpy(lo[n], hi[n], ror_f<S0 + 2>(w[n]), h[2]);
#pragma unroll
  for (int n = 0; n < E; ++n) pk_axpy(lo[n], hi[n], ror_f<S0 + 3>(w[n]), h[3]);
}

template <int E, bool H2>
__device__ __forceinline__ void win_gather_rot(int K, int wave, int lane, int wlo,
                                               const int32_t* __restrict__ nl, const float* __restrict__ ee,
                                               float* __restrict__ tb, const float4* __restrict__ win4,
                                               float* __restrict__ rs) {
  const int c = lane & 15;
  const int al = wave * 4 + (lane >> 4);
  const int slot = al * K + (c < K ? c : 0);
  const int idx = nl[slot];
  float w[E];
#pragma unroll
  for (int n = 0; n < E; ++n) w[n] = c < K ? ee[slot * E + n] : 0.f;
  // byte offset of (this lane's neighbour row, column chunk 0); the chunk offset 16c is lane-local
  const int roff = min(max(idx - wlo, 0), WROWS - 1) * (WF * 4);
  const char* wbytes = reinterpret_cast<const char*>(win4) + 16 * c;
  f32x2 lo[E], hi[E];
#pragma unroll
  for (int n = 0; n < E; ++n) { lo[n] = f32x2{0.f, 0.f}; hi[n] = f32x2{0.f, 0.f}; }
  float4 ha[4], hb[4];
  rot_load4<0>(wbytes, roff, ha);
  rot_load4<4>(wbytes, roff, hb);
  __builtin_amdgcn_sched_barrier(0);
  rot_fma4<E, 0>(ha, w, lo, hi);
  rot_load4<8>(wbytes, roff, ha);
  __builtin_amdgcn_sched_barrier(0);
  rot_fma4<E, 4>(hb, w, lo, hi);
  rot_load4<12>(wbytes, roff, hb);
  __builtin_amdgcn_sched_barrier(0);
  rot_fma4<E, 8>(ha, w, lo, hi);
  rot_fma4<E, 12>(hb, w, lo, hi);
  tile_put<E, H2>(tb, al, c, lo, hi, rs);
}

// gather + edge-weighted sum of one 32-atom tile: 16 lanes per atom, 4 atoms per wave, 8 waves.
// MODE 0 reads rows from the LDS window, MODE 1 from global memory; a compile-time constant because
// the window instance must not contain global loads (the compiler's vmcnt(0) in front of their use
// would also wait for the list prefetch that is deliberately left in flight).
template <int E, bool K4, int MODE, bool H2>
__device__ __forceinline__ void win_gather(int K, int wave, int lane, int wlo,
                                           const int32_t* __restrict__ nl, const float* __restrict__ ee,
                                           float* __restrict__ tb, const float4* __restrict__ win4,
                                           const float4* __restrict__ src4, float* __restrict__ rs) {
  const int c = lane & 15;
  const int al = wave * 4 + (lane >> 4);
  const float4* wbase = win4 + c;
  f32x2 lo[E], hi[E];
#pragma unroll
  for (int n = 0; n < E; ++n) { lo[n] = f32x2{0.f, 0.f}; hi[n] = f32x2{0.f, 0.f}; }
  if (K4) {
    // 16 neighbours per round: all list reads, then all row reads, then the FMAs
    const int4* nl4 = reinterpret_cast<const int4*>(nl + al * K);
    const float4* e4 = reinterpret_cast<const float4*>(ee + al * K * E);
    const int ng = K / 4;
#pragma unroll 1
    for (int g0 = 0; g0 < ng; g0 += 4) {
      int4 r4[4];
      float4 ev4[4][E];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gq = g0 + q < ng ? g0 + q : ng - 1;
        r4[q] = nl4[gq];
#pragma unroll
        for (int n = 0; n < E; ++n) ev4[q][n] = e4[gq * E + n];
      }
      float4 hv[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr[4] = {r4[q].x, r4[q].y, r4[q].z, r4[q].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (MODE == 0) {
            const int r = min(max(rr[u] - wlo, 0), WROWS - 1);   // padded slots stay inside the window
            hv[4 * q + u] = wbase[r * WC4];
          } else {
            hv[4 * q + u] = src4[(int64_t)rr[u] * WC4 + c];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (g0 + q < ng) {
          float ev[4 * E];
#pragma unroll
          for (int n = 0; n < E; ++n) {
            ev[4 * n + 0] = ev4[q][n].x; ev[4 * n + 1] = ev4[q][n].y;
            ev[4 * n + 2] = ev4[q][n].z; ev[4 * n + 3] = ev4[q][n].w;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int n = 0; n < E; ++n) pk_axpy(lo[n], hi[n], ev[u * E + n], hv[4 * q + u]);
        }
      }
    }
  } else {
    for (int j = 0; j < K; ++j) {
      const int rj = nl[al * K + j];
      float4 hv;
      if (MODE == 0) {
        const int r = min(max(rj - wlo, 0), WROWS - 1);
        hv = wbase[r * WC4];
      } else {
        hv = src4[(int64_t)rj * WC4 + c];
      }
#pragma unroll
      for (int n = 0; n < E; ++n) pk_axpy(lo[n], hi[n], ee[(al * K + j) * E + n], hv);
    }
  }
  tile_put<E, H2>(tb, al, c, lo, hi, rs);
}

// The global-memory variant is kept out of line: inlined next to the window variant it makes the
// compiler put vmcnt waits (for registers its loads may target) into the window gather, which then
// stalls on the list prefetch in flight.
template <int E, bool K4, bool H2>
__device__ __noinline__ void win_gather_global(int K, int wave, int lane, const int32_t* nl, const float* ee,
                                               float* tb, const float4* src4, float* rs) {
  win_gather<E, K4, 1, H2>(K, wave, lane, 0, nl, ee, tb, nullptr, src4, rs);
}

// ---- forward ------------------------------------------------------------------------------------------
struct MpWinFwdArgs {
  int64_t N;
  int K;
  int64_t ntiles;
  int tiles_per_wg;
  const float* h;          // [N][64]
  const int32_t* nlist;    // [N][K]
  const float* e;          // [N*K][E]
  const float* Wfrag;      // mpw_pack mode 0
  const float* rowscale;   // [N] or nullptr
  int residual;
  float* out;              // [N][64]
  float* S_save;           // [N][64] or nullptr
  int act;
  float* dummy;            // 64 floats: where the lanes of rows >= N store
  RangeGuard guard;        // word == nullptr: unguarded (NG_GEMM_MATH=fp32)
  const unsigned* wflag;   // flag word of a weight image kept over calls, or nullptr: == wflag_ver when its weights left the piece range
  unsigned wflag_ver;
  const float* Wfrag32;    // fp32 fragments (mpw_pack mode 0) a guarded call switches to when its weights leave the range
#ifdef MPW_STAMP
  unsigned long long* stamps;
#endif
};
#ifdef MPW_STAMP
#define MPW_T(k) do { if (a.stamps && blockIdx.x == 3 && lane == 0 && t - T0 >= 2 && t - T0 < 6) a.stamps[((t - T0 - 2) * 8 + wave) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define MPW_T(k) do {} while (0)
#endif

template <int E, bool K4, bool H2>
__device__ __forceinline__ void mp_win_fwd_body(const MpWinFwdArgs& a) {
  constexpr int KF = E * WF;
  constexpr int LD = KF + 4;
  constexpr int NT = KF / 16, NT2 = KF / 32;
  constexpr int TILE_FLOATS = (H2 ? WinTile<E>::BYTES_H2 : WinTile<E>::BYTES_F32) / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* win = smem;                                                   // [WROWS][64]
  float* tile = win + WROWS * WF;                                      // [32][LD] fp32, or two fp16 piece planes (WinTile)
  int32_t* s_nl = reinterpret_cast<int32_t*>(tile + TILE_FLOATS);      // [2][32*K]
  float* s_e = reinterpret_cast<float*>(s_nl + 2 * WTA * a.K);         // [2][32*K*E]
  int* ctl = reinterpret_cast<int*>(s_e + 2 * WTA * a.K * E);          // [2][16]
  float* s_rs = reinterpret_cast<float*>(ctl + 32);                    // [32]  inverse row scale of the piece tile (H2)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;
  const int per_tile = WTA * K;

  const int64_t T0 = (int64_t)blockIdx.x * a.tiles_per_wg;
  const int64_t T1 = std::min<int64_t>(T0 + a.tiles_per_wg, a.ntiles);
  if (T0 >= T1) return;

  const float4* src4 = reinterpret_cast<const float4*>(a.h);
  float4* win4 = reinterpret_cast<float4*>(win);
  // zero the window once: clamped reads of padded slots must hit finite values
  for (int t = tid; t < WROWS * WC4; t += WTHREADS) win4[t] = f4zero();

  // this wave's weight slab (output columns 16ct..) resident in registers for the whole launch
  const int ct = wave & 3, hh = wave >> 2;
  float wf[H2 ? 1 : KF / 4];
  u32x4 wh[H2 ? NT2 : 1], wl[H2 ? NT2 : 1];        // fp16 pieces of 2^8 W: A operands of v_mfma_f32_16x16x32_f16
  if (H2) {
    const u32x4* p = reinterpret_cast<const u32x4*>(a.Wfrag) + (size_t)(ct * NT2) * 2 * 64 + lane;
#pragma unroll
    for (int T = 0; T < NT2; ++T) { wh[T] = p[(2 * T) * 64]; wl[T] = p[(2 * T + 1) * 64]; }
#pragma unroll
    for (int T = 0; T < NT2; ++T)
#pragma unroll
      for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(wh[T][j])); asm volatile("" : "+v"(wl[T][j])); }
  } else {
    const float4* p = reinterpret_cast<const float4*>(a.Wfrag) + (ct * NT) * 64 + lane;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const float4 v = p[T * 64];
      wf[4 * T + 0] = v.x; wf[4 * T + 1] = v.y; wf[4 * T + 2] = v.z; wf[4 * T + 3] = v.w;
    }
    // "use" the slab once here: otherwise the first MFMA of every tile carries a conservative vmcnt
    // wait that also drains the previous tile's output stores
#pragma unroll
    for (int i = 0; i < KF / 4; ++i) asm volatile("" : "+v"(wf[i]));
  }

  WinLists<E, K4> lists;
  int wlo = -(1 << 30);       // no window yet: the first tile stages one
  int mode = 0;
  lists.issue(a.nlist, a.e, T0, K, a.N, tid);
  lists.commit(s_nl + (T0 & 1) * per_tile, s_e + (T0 & 1) * per_tile * E, ctl + (T0 & 1) * 16, K, tid, wave, lane);
  lists.issue(a.nlist, a.e, T0 + 1 < T1 ? T0 + 1 : T0, K, a.N, tid);
  NG_LDS_BARRIER();
  if (win_decide(ctl + (T0 & 1) * 16, wlo, mode)) {
    win_stage(win4, src4, wlo, a.N, tid);
  }
  NG_LDS_BARRIER();

  const int a16 = lane & 15, g = lane >> 4;
  const int col = 16 * ct + 4 * g;
  const float resf = a.residual ? 1.f : 0.f;
#pragma unroll 1
  for (int64_t t = T0; t < T1; ++t) {
    // ---- phase 1: lists of t+1 into LDS, request lists of t+2 and the epilogue operands, gather tile t
    const int64_t row = t * WTA + 16 * hh + a16;          // this lane's atom in the matrix phase
    const bool live = row < a.N;
    const int64_t rowc = live ? row : a.N - 1;
    MPW_T(0);
    if (t + 1 < T1)
      lists.commit(s_nl + ((t + 1) & 1) * per_tile, s_e + ((t + 1) & 1) * per_tile * E,
                   ctl + ((t + 1) & 1) * 16, K, tid, wave, lane);
    lists.issue(a.nlist, a.e, t + 2 < T1 ? t + 2 : t, K, a.N, tid);
    const float rs = a.rowscale[rowc];
    const float4 re = *reinterpret_cast<const float4*>(a.h + rowc * WF + col);
    MPW_T(1);
    {
      const int32_t* nl = s_nl + (t & 1) * per_tile;
      const float* ee = s_e + (t & 1) * per_tile * E;
      if (mode == 0 && K <= 16) win_gather_rot<E, H2>(K, wave, lane, wlo, nl, ee, tile, win4, s_rs);
      else if (mode == 0) win_gather<E, K4, 0, H2>(K, wave, lane, wlo, nl, ee, tile, win4, src4, s_rs);
      else win_gather_global<E, K4, H2>(K, wave, lane, nl, ee, tile, src4, s_rs);
    }
    MPW_T(2);
    NG_LDS_BARRIER();
    MPW_T(3);
    // ---- phase 2: tile x weights on the matrix cores, epilogue
    {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      float rsx = rs;
      if (H2) {
        // B operand: lane (atom a16, k-slots 8g..8g+7 of the 32-wide step) = 16 B of each piece plane; reads run two
        // steps ahead.  acc0 collects the small products (l h, h l), acc1 the leading ones; the pieces carry 2^8.
        const char* xrow = reinterpret_cast<const char*>(tile) + (16 * hh + a16) * WinTile<E>::ROWB + 16 * g;
        u32x4 xh[NT2], xl[NT2];
#pragma unroll
        for (int T = 0; T < 2 && T < NT2; ++T) {
          xh[T] = *reinterpret_cast<const u32x4*>(xrow + 64 * T);
          xl[T] = *reinterpret_cast<const u32x4*>(xrow + 64 * T + WinTile<E>::PLANE);
        }
#pragma unroll
        for (int T = 0; T < NT2; ++T) {
          if (T + 2 < NT2) {
            xh[T + 2] = *reinterpret_cast<const u32x4*>(xrow + 64 * (T + 2));
            xl[T + 2] = *reinterpret_cast<const u32x4*>(xrow + 64 * (T + 2) + WinTile<E>::PLANE);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[T]), __builtin_bit_cast(f16x8, xh[T]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[T]), __builtin_bit_cast(f16x8, xh[T]), acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[T]), __builtin_bit_cast(f16x8, xl[T]), acc0, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        rsx = rs * (1.0f / 256.0f) * s_rs[16 * hh + a16];
      } else {
        const float* xrow = tile + (16 * hh + a16) * LD + 4 * g;
        // operand reads run two k-steps ahead of the MFMAs that consume them (pinned: left alone the
        // scheduler hoists all of them to the top and the matrix pipe idles behind the LDS)
        float4 x[NT];
        x[0] = *reinterpret_cast<const float4*>(xrow);
        x[1] = *reinterpret_cast<const float4*>(xrow + 16);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          if (T + 2 < NT) x[T + 2] = *reinterpret_cast<const float4*>(xrow + 16 * (T + 2));
          __builtin_amdgcn_sched_barrier(0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 0], x[T].x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 1], x[T].y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 2], x[T].z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 3], x[T].w, acc1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      MPW_T(4);
      float4 v = make_float4((acc0[0] + acc1[0]) * rsx, (acc0[1] + acc1[1]) * rsx, (acc0[2] + acc1[2]) * rsx,
                             (acc0[3] + acc1[3]) * rsx);
      if (a.act == NG_ACT_SOFTPLUS) {
        v.x = softplus_f(v.x); v.y = softplus_f(v.y); v.z = softplus_f(v.z); v.w = softplus_f(v.w);
      } else if (a.act != NG_ACT_NONE) {
        v.x = act_apply(a.act, v.x); v.y = act_apply(a.act, v.y);
        v.z = act_apply(a.act, v.z); v.w = act_apply(a.act, v.w);
      }
      // rows past the end store into a dummy row: the store count per tile stays fixed
      const int64_t o = row * WF + col;
      // both results first, then both stores: a wait on `re` between them would also wait for the
      // first store to be acknowledged (vmcnt retires in order)
      const float4 vo = make_float4(v.x + resf * re.x, v.y + resf * re.y, v.z + resf * re.z, v.w + resf * re.w);
      asm volatile("" :: "v"(vo.x), "v"(vo.y), "v"(vo.z), "v"(vo.w));
      if (a.S_save) *reinterpret_cast<float4*>(live ? a.S_save + o : a.dummy + col) = v;
      *reinterpret_cast<float4*>(live ? a.out + o : a.dummy + col) = vo;
    }
    MPW_T(5);
    bool restage = false;
    if (t + 1 < T1) restage = win_decide(ctl + ((t + 1) & 1) * 16, wlo, mode);
    NG_LDS_BARRIER();
    MPW_T(6);
    if (restage) {                       // uniform over the workgroup
      win_stage(win4, src4, wlo, a.N, tid);
      NG_LDS_BARRIER();
    }
  }
}

// Range guard (ng_internal.h).  The aggregate rows of the piece form carry their own scale (tile_put), so the only operand
// that can leave the fp16 range is a weight (|2^8 w| >= 65504) — known at the kernel's first instruction: the pack launch
// of this call raised the guard, or the image kept over calls (ng_weights_frozen) has its flag word set.  Both bodies live
// in the one kernel; the choice is uniform over the launch and costs no second launch.
template <int E, bool K4, bool H2>
__global__ __launch_bounds__(WTHREADS, 1) void mp_win_fwd_kernel(MpWinFwdArgs a) {
  if (H2 && a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag))) {
    a.Wfrag = a.Wfrag32;
    mp_win_fwd_body<E, K4, false>(a);
  } else {
    mp_win_fwd_body<E, K4, H2>(a);
  }
}

// ---- window-resident neighbour aggregation for the reference's default width (F % 128 == 0, e.g. 256) -------------
//   A[i][n][:] = sum_j e[i][j][n] * h[nlist[i][j]][:]          (nmrgnn/layers.py:39-40, first contraction)
// At F = 256 the generic kernel (node_ops.hip: aggregate_kernel) pulls every gathered row out of L2 once per
// referencing edge — 2.1 GB per launch at the bench batch, ~14 TB/s of L2 traffic, 190 us.  Here a persistent
// 512-thread workgroup walks its run of 32-atom tiles once per 128-column SLAB of h with a 272-row window of that slab
// in LDS (136 KB), so a row leaves L2 once per run and slab; the gathers hit LDS.  Same skeleton as the F = 64 kernels
// (lists one tile ahead through registers, restage when a tile's range leaves the window, global-memory gather when
// the range is wider than the window).  Neighbours are summed in ENTRY order with fused multiply-adds, exactly like
// the generic and the CSR kernels: the three agree bit for bit (tests/test_gpu_csr.py).
constexpr int AW_ROWS = 272;          // window rows
constexpr int AW_SLAB = 128;          // floats per slab row
constexpr int AW_C4 = AW_SLAB / 4;    // 32 float4 per slab row: lane c of an atom's 16 lanes owns chunks c and c + 16

struct AggWinArgs {
  int64_t N;
  int K, F;
  int64_t ntiles;
  int tiles_per_wg;
  const float* h;          // [N][F]
  const int32_t* nlist;    // [N][K]
  const float* e;          // [N*K][E]
  float* A;                // [N][E][F]
};

__device__ __forceinline__ bool aw_decide(const int* __restrict__ ctl, int& wlo, int& mode) {
  int lo = ctl[0], hi = ctl[8];
#pragma unroll
  for (int i = 1; i < 8; ++i) { lo = min(lo, ctl[i]); hi = max(hi, ctl[8 + i]); }
  mode = 0;
  if (hi < lo) return false;                                  // empty tile
  if (lo >= wlo && hi < wlo + AW_ROWS) return false;          // window hit
  if (hi - lo + 1 > AW_ROWS) { mode = 1; return false; }      // too wide: gather from global memory
  wlo = max(0, lo - (AW_ROWS - (hi - lo + 1)) / 2);
  return true;
}

__device__ __forceinline__ void aw_stage(float4* __restrict__ win4, const float4* __restrict__ src4, int f4n, int slab,
                                         int wlo, int64_t N, int tid) {
  float4 v[17];
#pragma unroll
  for (int u = 0; u < 17; ++u) {
    const int idx = tid + WTHREADS * u;
    const int64_t row = (int64_t)wlo + (idx >> 5);
    v[u] = row < N ? src4[row * f4n + AW_C4 * slab + (idx & 31)] : f4zero();
  }
#pragma unroll
  for (int u = 0; u < 17; ++u) win4[tid + WTHREADS * u] = v[u];
}
static_assert(AW_ROWS * AW_C4 == 17 * WTHREADS, "slab window staging assumes 17 float4 per thread");

// one 32-atom tile: 16 lanes per atom, each lane two float4 chunks (c, c + 16) of the slab; eight neighbours per round
template <int E, int MODE>
__device__ __forceinline__ void aw_gather(int K, int wave, int lane, int wlo, const int32_t* __restrict__ nl,
                                          const float* __restrict__ ee, const float4* __restrict__ win4,
                                          const float4* __restrict__ src4, int f4n, int slab, float* __restrict__ Arow) {
  const int c = lane & 15;
  const int al = wave * 4 + (lane >> 4);
  f32x2 lo[2][E], hi[2][E];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int n = 0; n < E; ++n) { lo[k][n] = f32x2{0.f, 0.f}; hi[k][n] = f32x2{0.f, 0.f}; }
  const int4* nl4 = reinterpret_cast<const int4*>(nl + al * K);
  const float4* e4 = reinterpret_cast<const float4*>(ee + al * K * E);
  const int ng = K / 4;
#pragma unroll 1
  for (int g0 = 0; g0 < ng; g0 += 2) {
    int4 r4[2];
    float4 ev4[2][E];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int gq = g0 + q < ng ? g0 + q : ng - 1;
      r4[q] = nl4[gq];
#pragma unroll
      for (int n = 0; n < E; ++n) ev4[q][n] = e4[gq * E + n];
    }
    float4 hv[8][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int rr[4] = {r4[q].x, r4[q].y, r4[q].z, r4[q].w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (MODE == 0) {
          const int r = min(max(rr[u] - wlo, 0), AW_ROWS - 1);     // padded slots stay inside the window (weight 0)
          hv[4 * q + u][0] = win4[r * AW_C4 + c];
          hv[4 * q + u][1] = win4[r * AW_C4 + c + 16];
        } else {
          hv[4 * q + u][0] = src4[(int64_t)rr[u] * f4n + AW_C4 * slab + c];
          hv[4 * q + u][1] = src4[(int64_t)rr[u] * f4n + AW_C4 * slab + c + 16];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (g0 + q < ng) {
        float ev[4 * E];
#pragma unroll
        for (int n = 0; n < E; ++n) {
          ev[4 * n + 0] = ev4[q][n].x; ev[4 * n + 1] = ev4[q][n].y;
          ev[4 * n + 2] = ev4[q][n].z; ev[4 * n + 3] = ev4[q][n].w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int n = 0; n < E; ++n) {
            pk_axpy(lo[0][n], hi[0][n], ev[u * E + n], hv[4 * q + u][0]);
            pk_axpy(lo[1][n], hi[1][n], ev[u * E + n], hv[4 * q + u][1]);
          }
      }
    }
  }
  // Arow: this atom's A[i][0][slab columns] (nullptr for rows past the end); written once, read by a later kernel
  if (Arow) {
    typedef float nt4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int n = 0; n < E; ++n) {
      nt4* d = reinterpret_cast<nt4*>(Arow) + n * f4n + c;
      __builtin_nontemporal_store(nt4{lo[0][n][0], lo[0][n][1], hi[0][n][0], hi[0][n][1]}, d);
      __builtin_nontemporal_store(nt4{lo[1][n][0], lo[1][n][1], hi[1][n][0], hi[1][n][1]}, d + 16);
    }
  }
}

template <int E>
__device__ __noinline__ void aw_gather_global(int K, int wave, int lane, const int32_t* nl, const float* ee,
                                              const float4* src4, int f4n, int slab, float* Arow) {
  aw_gather<E, 1>(K, wave, lane, 0, nl, ee, nullptr, src4, f4n, slab, Arow);
}

template <int E>
__global__ __launch_bounds__(WTHREADS, 1) void agg_win_kernel(AggWinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* win = smem;                                                        // [AW_ROWS][AW_SLAB]
  int32_t* s_nl = reinterpret_cast<int32_t*>(win + AW_ROWS * AW_SLAB);      // [2][32*K]
  float* s_e = reinterpret_cast<float*>(s_nl + 2 * WTA * a.K);              // [2][32*K*E]
  int* ctl = reinterpret_cast<int*>(s_e + 2 * WTA * a.K * E);               // [2][16]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, per_tile = WTA * K, f4n = a.F / 4;
  const int64_t T0 = (int64_t)blockIdx.x * a.tiles_per_wg;
  const int64_t T1 = std::min<int64_t>(T0 + a.tiles_per_wg, a.ntiles);
  if (T0 >= T1) return;
  const float4* src4 = reinterpret_cast<const float4*>(a.h);
  float4* win4 = reinterpret_cast<float4*>(win);
  const int al = wave * 4 + (lane >> 4);

#pragma unroll 1
  for (int slab = 0; slab < a.F / AW_SLAB; ++slab) {
    for (int t = tid; t < AW_ROWS * AW_C4; t += WTHREADS) win4[t] = f4zero();   // clamped reads must hit finite values
    WinLists<E, true> lists;
    int wlo = -(1 << 30), mode = 0;
    lists.issue(a.nlist, a.e, T0, K, a.N, tid);
    lists.commit(s_nl + (T0 & 1) * per_tile, s_e + (T0 & 1) * per_tile * E, ctl + (T0 & 1) * 16, K, tid, wave, lane);
    lists.issue(a.nlist, a.e, T0 + 1 < T1 ? T0 + 1 : T0, K, a.N, tid);
    NG_LDS_BARRIER();
    if (aw_decide(ctl + (T0 & 1) * 16, wlo, mode)) aw_stage(win4, src4, f4n, slab, wlo, a.N, tid);
    NG_LDS_BARRIER();
#pragma unroll 1
    for (int64_t t = T0; t < T1; ++t) {
      if (t + 1 < T1)
        lists.commit(s_nl + ((t + 1) & 1) * per_tile, s_e + ((t + 1) & 1) * per_tile * E, ctl + ((t + 1) & 1) * 16, K, tid,
                     wave, lane);
      lists.issue(a.nlist, a.e, t + 2 < T1 ? t + 2 : t, K, a.N, tid);
      const int32_t* nl = s_nl + (t & 1) * per_tile;
      const float* ee = s_e + (t & 1) * per_tile * E;
      const int64_t row = t * WTA + al;
      float* Arow = row < a.N ? a.A + (row * E) * a.F + AW_SLAB * slab : nullptr;
      if (mode == 0) aw_gather<E, 0>(K, wave, lane, wlo, nl, ee, win4, src4, f4n, slab, Arow);
      else aw_gather_global<E>(K, wave, lane, nl, ee, src4, f4n, slab, Arow);
      NG_LDS_BARRIER();      // lists of t+1 (and their range) are in LDS; every gather of tile t is done
      if (t + 1 < T1 && aw_decide(ctl + ((t + 1) & 1) * 16, wlo, mode)) {      // uniform over the workgroup
        aw_stage(win4, src4, f4n, slab, wlo, a.N, tid);
        NG_LDS_BARRIER();
      }
    }
    NG_LDS_BARRIER();        // the next slab zeroes and restages the window
  }
}

// ---- window-resident edge gradient for the default width ---------------------------------------------------------------
//   de[i][j][n] (+)= sum_l dA[i][n][l] * h[nlist[i][j]][l]          (generic form: csr_edge_grad_wide_kernel)
// Same slab windows of h as the aggregation above (two passes at F = 256); per atom the 16 lanes hold the slab's part of
// their OWN dA row in registers (one contiguous read per pass), walk the neighbours with the rotation scheme of the F = 64
// kernels — lane c owns slot c, in step s it multiplies ITS two chunks with the row of slot (c + s) and sends the partial
// back to the owner over DPP — and the owner adds the pass's sum to de (slab 0: += the caller's de when accumulating).
struct EGradWinArgs {
  int64_t N;
  int K, F;
  int64_t ntiles;
  int tiles_per_wg;
  const float* h;          // [N][F]
  const int32_t* nlist;    // [N][K]
  const float* dA;         // [N][E][F]
  float* de;               // [N][K][E]
  int accumulate;
};

template <int E, int S, int MODE>
__device__ __forceinline__ void eg_step(const char* __restrict__ wb, const float4* __restrict__ src4, int f4n, int slab, int c,
                                        int roff, int gidx, const float4 (&da)[2][E], float (&out)[E]) {
  float4 h0, h1;
  if (MODE == 0) {
    const char* p = wb + ror_i<S>(roff);
    h0 = *reinterpret_cast<const float4*>(p);
    h1 = *reinterpret_cast<const float4*>(p + 256);
  } else {
    const int64_t r = ror_i<S>(gidx);
    h0 = src4[r * f4n + AW_C4 * slab + c];
    h1 = src4[r * f4n + AW_C4 * slab + c + 16];
  }
#pragma unroll
  for (int n = 0; n < E; ++n) {
    float p = da[0][n].x * h0.x;
    p = fmaf(da[0][n].y, h0.y, p); p = fmaf(da[0][n].z, h0.z, p); p = fmaf(da[0][n].w, h0.w, p);
    p = fmaf(da[1][n].x, h1.x, p); p = fmaf(da[1][n].y, h1.y, p); p = fmaf(da[1][n].z, h1.z, p); p = fmaf(da[1][n].w, h1.w, p);
    out[n] += ror_f<(16 - S) & 15>(p);      // slot (c + S) was processed here: back to its owner lane
  }
}

template <int E, int MODE>
__device__ __forceinline__ void eg_tile(int K, int wave, int lane, int wlo, const int32_t* __restrict__ nl,
                                        const float4* __restrict__ win4, const float4* __restrict__ src4, int f4n, int slab,
                                        const float4 (&da)[2][E], float (&out)[E]) {
  const int c = lane & 15;
  const int al = wave * 4 + (lane >> 4);
  const int idx = nl[al * K + (c < K ? c : 0)];
  const int roff = min(max(idx - wlo, 0), AW_ROWS - 1) * (AW_SLAB * 4);
  const char* wb = reinterpret_cast<const char*>(win4) + 16 * c;
#pragma unroll
  for (int n = 0; n < E; ++n) out[n] = 0.f;
  eg_step<E, 0, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 1, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 2, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 3, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 4, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 5, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 6, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 7, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 8, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 9, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 10, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 11, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 12, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 13, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 14, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
  eg_step<E, 15, MODE>(wb, src4, f4n, slab, c, roff, idx, da, out);
}

// operands and results of the out-of-line variant travel in registers (vector arguments, a vector back): arrays passed
// by pointer, and structs of vectors passed by value, lived on the stack
template <int E>
__device__ __noinline__ f32x4 eg_tile_global(int K, const int32_t* nl, const float4* src4, int f4n, int slab,
                                                 f32x4 r0, f32x4 r1, f32x4 r2, f32x4 r3, f32x4 r4, f32x4 r5) {
  // (arguments beyond 32 dwords go over the stack: wave and lane are taken from the thread id here)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const f32x4 rows[6] = {r0, r1, r2, r3, r4, r5};      // [chunk][n], E entries per chunk
  float4 da[2][E];
  float out[E];
#pragma unroll
  for (int n = 0; n < E; ++n) {
    da[0][n] = make_float4(rows[n][0], rows[n][1], rows[n][2], rows[n][3]);
    da[1][n] = make_float4(rows[E + n][0], rows[E + n][1], rows[E + n][2], rows[E + n][3]);
  }
  eg_tile<E, 1>(K, wave, lane, 0, nl, nullptr, src4, f4n, slab, da, out);
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < E; ++n) r[n] = out[n];
  return r;
}

template <int E>
__global__ __launch_bounds__(WTHREADS, 1) void egrad_win_kernel(EGradWinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* win = smem;                                                        // [AW_ROWS][AW_SLAB]
  int32_t* s_nl = reinterpret_cast<int32_t*>(win + AW_ROWS * AW_SLAB);      // [2][32*K]
  int* ctl = reinterpret_cast<int*>(s_nl + 2 * WTA * a.K);                  // [2][16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, per_tile = WTA * K, f4n = a.F / 4;
  const int64_t T0 = (int64_t)blockIdx.x * a.tiles_per_wg;
  const int64_t T1 = std::min<int64_t>(T0 + a.tiles_per_wg, a.ntiles);
  if (T0 >= T1) return;
  const float4* src4 = reinterpret_cast<const float4*>(a.h);
  const float4* dA4 = reinterpret_cast<const float4*>(a.dA);
  float4* win4 = reinterpret_cast<float4*>(win);
  const int c = lane & 15, al = wave * 4 + (lane >> 4);

  // lists of one tile in flight: indices only (threads < 8 K int4)
  int4 p_nl;
  auto issue = [&](int64_t t) {
    const int64_t nb = t * per_tile / 4, nlim = a.N * K / 4;
    const int64_t qn = nb + tid;
    const int4 v = reinterpret_cast<const int4*>(a.nlist)[qn < nlim ? qn : nlim - 1];
    p_nl = (qn < nlim && tid < per_tile / 4) ? v : make_int4(0, 0, 0, 0);
  };
  auto commit = [&](int64_t t) {
    int lo = 0x7fffffff, hi = -1;
    if (tid < per_tile / 4) {
      reinterpret_cast<int4*>(s_nl + (t & 1) * per_tile)[tid] = p_nl;
      lo = min(min(p_nl.x, p_nl.y), min(p_nl.z, p_nl.w));
      hi = max(max(p_nl.x, p_nl.y), max(p_nl.z, p_nl.w));
    }
    lo = wave_min_i32(lo);
    hi = -wave_min_i32(-hi);
    if (lane == 63) { ctl[(t & 1) * 16 + wave] = lo; ctl[(t & 1) * 16 + 8 + wave] = hi; }
  };

#pragma unroll 1
  for (int slab = 0; slab < a.F / AW_SLAB; ++slab) {
    for (int t = tid; t < AW_ROWS * AW_C4; t += WTHREADS) win4[t] = f4zero();
    int wlo = -(1 << 30), mode = 0;
    issue(T0);
    commit(T0);
    issue(T0 + 1 < T1 ? T0 + 1 : T0);
    NG_LDS_BARRIER();
    if (aw_decide(ctl + (T0 & 1) * 16, wlo, mode)) aw_stage(win4, src4, f4n, slab, wlo, a.N, tid);
    NG_LDS_BARRIER();
#pragma unroll 1
    for (int64_t t = T0; t < T1; ++t) {
      const int64_t row = t * WTA + al;
      const int64_t rc = row < a.N ? row : a.N - 1;
      // this lane's two chunks of its own dA row (slab part), and the old value of its (atom, slot) gradient
      float4 da[2][E];
#pragma unroll
      for (int n = 0; n < E; ++n) {
        da[0][n] = dA4[(rc * E + n) * f4n + AW_C4 * slab + c];
        da[1][n] = dA4[(rc * E + n) * f4n + AW_C4 * slab + c + 16];
      }
      const bool owner = row < a.N && c < K;
      float old[E];
#pragma unroll
      for (int n = 0; n < E; ++n) old[n] = (owner && (slab > 0 || a.accumulate)) ? a.de[(row * K + c) * E + n] : 0.f;
      if (t + 1 < T1) commit(t + 1);
      issue(t + 2 < T1 ? t + 2 : t);
      float out[E];
      const int32_t* nl = s_nl + (t & 1) * per_tile;
      if (mode == 0) {
        eg_tile<E, 0>(K, wave, lane, wlo, nl, win4, src4, f4n, slab, da, out);
      } else {
        f32x4 rows[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) rows[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < E; ++n) {
          rows[n] = f32x4{da[0][n].x, da[0][n].y, da[0][n].z, da[0][n].w};
          rows[E + n] = f32x4{da[1][n].x, da[1][n].y, da[1][n].z, da[1][n].w};
        }
        const f32x4 r = eg_tile_global<E>(K, nl, src4, f4n, slab, rows[0], rows[1], rows[2], rows[3], rows[4], rows[5]);
#pragma unroll
        for (int n = 0; n < E; ++n) out[n] = r[n];
      }
      if (owner) {
#pragma unroll
        for (int n = 0; n < E; ++n) a.de[(row * K + c) * E + n] = old[n] + out[n];
      }
      NG_LDS_BARRIER();
      if (t + 1 < T1 && aw_decide(ctl + ((t + 1) & 1) * 16, wlo, mode)) {
        aw_stage(win4, src4, f4n, slab, wlo, a.N, tid);
        NG_LDS_BARRIER();
      }
    }
    NG_LDS_BARRIER();
  }
}

int egrad_win(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* h, const int32_t* nlist,
              const float* dA, float* de, int accumulate) {
  EGradWinArgs a{};
  a.N = N; a.K = K; a.F = F; a.ntiles = cdiv(N, WTA);
  const int64_t per = win_tiles_per_wg(a.ntiles, ctx->num_cu);
  a.tiles_per_wg = (int)per;
  a.h = h; a.nlist = nlist; a.dA = dA; a.de = de; a.accumulate = accumulate;
  const int grid = (int)cdiv(a.ntiles, per);
  const size_t lds = (size_t)(AW_ROWS * AW_SLAB + 2 * WTA * K + 32) * 4;
  switch (E) {
    case 1: hipLaunchKernelGGL((egrad_win_kernel<1>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 2: hipLaunchKernelGGL((egrad_win_kernel<2>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 3: hipLaunchKernelGGL((egrad_win_kernel<3>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

int agg_win_rows() { return AW_ROWS; }

bool agg_win_supported(int F, int E, int K) {
  return F % AW_SLAB == 0 && F <= 1024 && E >= 1 && E <= 3 && K % 4 == 0 && K >= 4 && K <= 16 && !sw().mp_layered;
}

int agg_win(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* h, const int32_t* nlist,
            const float* e, float* A) {
  AggWinArgs a{};
  a.N = N; a.K = K; a.F = F; a.ntiles = cdiv(N, WTA);
  const int64_t per = win_tiles_per_wg(a.ntiles, ctx->num_cu);
  a.tiles_per_wg = (int)per;
  a.h = h; a.nlist = nlist; a.e = e; a.A = A;
  const int grid = (int)cdiv(a.ntiles, per);
  const size_t lds = (size_t)(AW_ROWS * AW_SLAB + 2 * WTA * K * (1 + E) + 32) * 4;
  switch (E) {
    case 1: hipLaunchKernelGGL((agg_win_kernel<1>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 2: hipLaunchKernelGGL((agg_win_kernel<2>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 3: hipLaunchKernelGGL((agg_win_kernel<3>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

size_t mp_win_lds_bytes(int K, int E) {
  // the larger of the two tile forms (fp32 rows, fp16 piece planes): the budget check is the same for both
  const int tile_bytes = std::max(WTA * (E * WF + 4) * 4, 2 * WTA * (E * WF + 8) * 2);
  return (size_t)(WROWS * WF + 2 * WTA * K * (1 + E) + 32 + 32) * 4 + tile_bytes;
}

// matrix phase of the forward window kernel on the fp16 pipe with two-piece operands unless NG_GEMM_MATH=fp32
static bool mp_win_h2() { return !sw().gemm_math_fp32; }

bool mp_win_supported(int F, int E, int K) {
  return F == WF && E >= 1 && E <= 3 && K >= 1 && K <= 32 && mp_win_lds_bytes(K, E) <= 160 * 1024;
}

bool mp_win_enabled(int F, int E, int K) {
  return mp_win_supported(F, E, K) && !sw().mp_layered;      // default forward path for F == 64
}

int mp_win_fwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, int residual, const float* h,
               const int32_t* nlist, const float* e, const float* inv_degree, const float* w, float* h_out,
               float* s_save) {
  if (N == 0) return NG_OK;
  const int KF = E * WF;
  const bool h2 = mp_win_h2();
  // guarded call: the pack launch checks the weights against the piece range and the kernel takes its fp32-input body
  // (second image, same launch) when they do not fit
  const bool guarded = h2;
  RangeGuard guard{nullptr, 0};
  if (guarded) {
    guard = range_guard_begin(ctx);
    if (!guard.word) return NG_ERR_NOMEM;
  }
  // image: KF*64 floats (fp32 fragments / two fp16 pieces) + 64 floats of dummy row + the flag word; a guarded call keeps
  // the f32 image of the same weights behind it (one cached buffer, one pack job)
  const size_t img_floats = (size_t)KF * WF + 64 + 16;
  const size_t tot_floats = guarded ? 2 * img_floats : img_floats;
  bool have = false;
  float* Wfrag = (float*)cached_image(ctx, w, h2 ? 7 : 4, tot_floats * 4, &have);
  const bool cached = Wfrag != nullptr;
  if (!cached) {
    Wfrag = (float*)workspace(ctx, tot_floats * 4);
    if (!Wfrag) return NG_ERR_NOMEM;
    have = false;
  }
  float* Wf32 = guarded ? Wfrag + img_floats : nullptr;
  unsigned* wflag = cached && guarded ? reinterpret_cast<unsigned*>(Wfrag + KF * WF + 64) : nullptr;
  if (!have) {
    PackJob j;
    if (h2) {
      j.kind = PK_MPW_FWD; j.blocks = guarded ? 48 : 24; j.i0 = E; j.src[0] = w; j.dst[0] = Wfrag; j.dst[1] = Wf32;
      j.flag = wflag; j.guard = guard;
    } else {
      j.kind = PK_MPW_F32; j.blocks = 24; j.i0 = E; j.i1 = 0; j.src[0] = w; j.dst[0] = Wfrag;
    }
    if (int rc = pack_launch(ctx, st, j)) return rc;
    if (cached) cache_set_job(ctx, w, h2 ? 7 : 4, j);
  }
  if (h2 && sw().mp_w16 && mp_wave_wanted(ctx, N, E, K))
    return mp_wave_launch(ctx, st, N, K, act, residual, h, nlist, e, inv_degree, Wfrag, Wf32, wflag, guard, h_out, s_save);
  if (h2 && sw().mp_w16 && mp_win16_supported(E, K))
    return mp_win16_launch(ctx, st, N, K, E, act, residual, h, nlist, e, inv_degree, Wfrag, Wf32, wflag, guard, h_out, s_save);
  MpWinFwdArgs a{};
  a.N = N; a.K = K; a.ntiles = cdiv(N, WTA);
  // contiguous runs of tiles per workgroup, a multiple of 8 tiles (256 atoms) so that runs start on
  // molecule boundaries for the common 256-atom padding
  const int64_t per = win_tiles_per_wg(a.ntiles, ctx->num_cu);
  a.tiles_per_wg = (int)per;
  a.h = h; a.nlist = nlist; a.e = e; a.Wfrag = Wfrag; a.rowscale = inv_degree; a.residual = residual;
  a.out = h_out; a.S_save = s_save; a.act = act; a.dummy = Wfrag + KF * WF;
  const int grid = (int)cdiv(a.ntiles, per);
  const size_t lds = mp_win_lds_bytes(K, E);
  a.guard = guard; a.wflag = wflag; a.wflag_ver = pack_flag_version(ctx);
#define CALL(EE, HH)                                                                                      \
  if (K % 4 == 0)                                                                                         \
    hipLaunchKernelGGL((mp_win_fwd_kernel<EE, true, HH>), dim3(grid), dim3(WTHREADS), lds, st, a);        \
  else                                                                                                    \
    hipLaunchKernelGGL((mp_win_fwd_kernel<EE, false, HH>), dim3(grid), dim3(WTHREADS), lds, st, a);
#define CALLE(HH)                                                                                         \
  switch (E) {                                                                                            \
    case 1: { CALL(1, HH) } break;                                                                        \
    case 2: { CALL(2, HH) } break;                                                                        \
    case 3: { CALL(3, HH) } break;                                                                        \
  }
  a.Wfrag32 = Wf32;
#ifdef MPW_STAMP
  static unsigned long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { (void)hipMalloc(&dbg, 4 * 8 * 8 * 8); (void)hipMemset(dbg, 0, 4 * 8 * 8 * 8); }
  a.stamps = dbg;
#endif
  {
    ProfScope ps(ctx, st, "mp_win_fwd");
    if (h2) { CALLE(true) } else { CALLE(false) }
    NG_HIP(ctx, hipGetLastError());
  }
#undef CALLE
#undef CALL
#ifdef MPW_STAMP
  if (++calls == 40) {
    unsigned long long h[4 * 8 * 8];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    for (int tt = 0; tt < 4; ++tt)
      for (int w = 0; w < 8; w += 1) {
        const unsigned long long* p = h + (tt * 8 + w) * 8;
        fprintf(stderr, "MPW tile %d wave %d: commit/issue %5lld  gather %5lld  barrier %5lld  mfma %5lld  epilogue %5lld  end-barrier %5lld | total %6lld\n",
                tt, w, (long long)(p[1] - p[0]), (long long)(p[2] - p[1]), (long long)(p[3] - p[2]), (long long)(p[4] - p[3]),
                (long long)(p[5] - p[4]), (long long)(p[6] - p[5]), (long long)(p[6] - p[0]));
      }
  }
#endif
  return NG_OK;
}

}  // namespace ng
