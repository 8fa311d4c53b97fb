// Wave-autonomous edge-side backward window kernel of the MPLayer (atom_feature_size 64, E == 3, K in {4, 8, 12, 16}).
// Reference: the backward of nmrgnn/layers.py:26-46:  dP = dH * act'(S) * v ;  dA = dP Wp^T ;  de (+)= <dA, h[nlist]>.
//
// The design of mp_wave.hip (see there) applied to mp_win16_bwd.hip's work: a wave owns 16 atoms from the upstream gradient to
// the stores, the only workgroup events are the window changes.  Per micro-tile:
//   * lane (kg, atom) loads its 8 + 8 entries of dH and S of its atom (the two 32-wide k-steps of the contraction over m),
//     forms dP, writes it (the node-side kernel reads it) and splits it, scaled by the row's power of two, into the B operand
//     of v_mfma_f32_16x16x32_f16 — in registers;
//   * dA^T[(n, l)][atom] = Wp^T[(n, l)][m] dP^T[m][atom]: twelve 16-row tiles, the weight fragments (the image of mp_win16_bwd.hip,
//     pack_bodies.cuh: mpw_h2<2>) out of LDS as the A operand.  The result lands as dA[atom][n][16 lt + 4 kg .. + 3] in lane
//     (kg, atom): 48 registers, never in LDS;
//   * the edge gradient: per list entry the lane reads the four 16-byte chunks lt = 0..3 of the source row at 4 lt + kg out of
//     the (swizzled) window and adds 48 products to the entry's three partial dots; per quad of entries the four lanes of an
//     atom sum their partials (two exchange rounds) and lane kg == quad keeps them: entries 4 kg .. 4 kg + 3, twelve
//     consecutive floats of de.
// The matrix interval of a group's first micro-tile needs no window: it runs while the window travels.
// de, dP agree with mp_win16_bwd.hip to rounding (another summation order inside a dot).
#include <algorithm>
#include <cstdio>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "edge_fused.h"   // NG_LDS_BARRIER
#include "h2_common.cuh"
#include "mp_win16_common.cuh"
#include "mp_wave_common.cuh"

namespace ng {
namespace wvb {

using namespace wv;

constexpr int STRIP_BYTES = 1024;          // neighbour indices of a micro-tile, [piece][atom] x 16 B
constexpr int LDS_BYTES = WIN_BYTES + WIMG_BYTES + NWV * STRIP_BYTES;
constexpr int NCT = E * WF / 16;           // 16-row tiles of dA^T

struct Args {
  int64_t N;
  int K;
  int64_t atoms_per_wg;
  const float* dH;         // [N][64] upstream gradient of the layer output
  const float* S;          // [N][64] saved activation output, or nullptr (linear)
  const float* rowscale;   // [N]
  const float* h;          // [N][64] layer input (gathered)
  const int32_t* nlist;    // [N][K]
  const float* WfragT;     // piece fragments (mpw_h2<2>)
  const float* WfragT32;   // fp32 fragments (mpw_f32 mode 2)
  float* dP;               // [N][64] out
  float* de;               // [N*K][3] out (+= when accumulate)
  int act;
  int accumulate;
  RangeGuard guard;
  const unsigned* wflag;
};

// the edge-gradient dots of one micro-tile.  Per quad of list entries: this lane's partial dots over its 16 features (12 values), summed
// over the four lanes of the atom (two exchange rounds: every lane then holds the complete dots), kept by lane kg == quad — which
// thus ends up with entries 4 kg .. 4 kg + 3, twelve consecutive floats of de.  (All sixteen entries' partials at once and ONE
// transposing exchange at the end is fewer exchanges — 36 instead of 96 — and 36 more live registers: with the next micro-tile's
// prefetched rows beside them the kernel spilled 248 bytes per lane.)
template <bool GLOBAL>
__device__ __forceinline__ void edge_dots(const char* __restrict__ strip, const char* __restrict__ win, const float* __restrict__ h,
                                          int nq, int at, int kg, int wlo, const f32x4 (&dA)[NCT], float (&r4)[4][E]) {
  const char* rec = strip + (at << 4);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int n = 0; n < E; ++n) r4[s][n] = 0.f;
#pragma unroll 1
  for (int q = 0; q < nq; ++q) {
    const i32x4 ci = *reinterpret_cast<const i32x4*>(rec + (q << 8));
    float part[4][E];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 hr[4];
      if (!GLOBAL) {
        const int R = min(max(ci[s] - wlo, 0), WROWS - 1);
        const int sw = (R & 15) << 4;
        const char* row = win + (R << 8);
#pragma unroll
        for (int lt = 0; lt < 4; ++lt) hr[lt] = *reinterpret_cast<const f32x4*>(row + ((((4 * lt + kg) << 4)) ^ sw));
      } else {
        const f32x4* row = reinterpret_cast<const f32x4*>(h + (int64_t)ci[s] * WF);
#pragma unroll
        for (int lt = 0; lt < 4; ++lt) hr[lt] = row[4 * lt + kg];
      }
#pragma unroll
      for (int n = 0; n < E; ++n) {
        float d[4];      // four independent chains of four products (one chain of sixteen waits for itself)
#pragma unroll
        for (int lt = 0; lt < 4; ++lt) {
          const f32x4 a = dA[4 * n + lt];
          d[lt] = a[0] * hr[lt][0];
          d[lt] = __builtin_fmaf(a[1], hr[lt][1], d[lt]);
          d[lt] = __builtin_fmaf(a[2], hr[lt][2], d[lt]);
          d[lt] = __builtin_fmaf(a[3], hr[lt][3], d[lt]);
        }
        part[s][n] = (d[0] + d[1]) + (d[2] + d[3]);
      }
    }
    const bool mine = kg == q;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int n = 0; n < E; ++n) {
        float v = part[s][n];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        r4[s][n] = mine ? v : r4[s][n];
      }
  }
}

template <bool H2>
__device__ __forceinline__ void body(const Args& a) {
  extern __shared__ __attribute__((aligned(16))) char smem_wvb[];
  char* win = smem_wvb;
  char* wimg = win + WIN_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* strip = wimg + WIMG_BYTES + wave * STRIP_BYTES;
  const int at = lane & 15, kg = lane >> 4;
  const int K = a.K, nq = K >> 2;
  const int64_t A0 = (int64_t)blockIdx.x * a.atoms_per_wg;
  const int64_t A1 = std::min<int64_t>(A0 + a.atoms_per_wg, a.N);
  if (A0 >= A1) return;

  if (H2) wimg_dma(wimg, a.WfragT, wave, lane);
  int64_t have = A0 + (int64_t)wave * MT;
  if (have < A1) nlist_dma(a.nlist, K, a.N, strip, have, lane);

  f32x4 dA[NCT];
  // upstream gradient, activation output and row scale of a micro-tile, requested a micro-tile ahead (while the dots of the one
  // before run): this lane's entries m = 32 u + 8 kg + 4 v .. + 3 at index 2 u + v
  f32x4 pg[4], psv[4];
  float prs = 0.f;
  int64_t fetched = -1;
  auto fetch = [&](int64_t row0) __attribute__((always_inline)) {
    const int64_t row = row0 + at;
    const int64_t rowc = row < a.N ? row : a.N - 1;
    prs = a.rowscale[rowc];
    const f32x4* pd = reinterpret_cast<const f32x4*>(a.dH + rowc * WF + 8 * kg);
    pg[0] = pd[0]; pg[1] = pd[1]; pg[2] = pd[8]; pg[3] = pd[9];
    if (a.S) {
      const f32x4* ps = reinterpret_cast<const f32x4*>(a.S + rowc * WF + 8 * kg);
      psv[0] = ps[0]; psv[1] = ps[1]; psv[2] = ps[8]; psv[3] = ps[9];
    }
  };
  // matrix interval of micro-tile row0 (fetched): dP rows (written out), their piece operands, dA^T over the two k-steps
  auto matrix = [&](int64_t row0) __attribute__((always_inline)) {
    const int64_t row = row0 + at;
    const bool live = row < a.N;
    const float rs = prs;
    f32x4 g[4];
    {
#pragma unroll
      for (int v = 0; v < 4; ++v) g[v] = pg[v];
      if (a.S) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int j = 0; j < 4; ++j) g[v][j] *= act_grad_from_out(a.act, psv[v][j]);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) g[v] *= rs;
      if (!live) {
#pragma unroll
        for (int v = 0; v < 4; ++v) g[v] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (live) {
      f32x4* po = reinterpret_cast<f32x4*>(a.dP + row * WF + 8 * kg);
      po[0] = g[0]; po[1] = g[1]; po[8] = g[2]; po[9] = g[3];
    }
    if (H2) {
      // the row's power of two (mp_win16_bwd.hip: commit): pieces of S g with max |S g| below 2^13, dA scaled back by 2^-8 / S
      float m = 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(g[v][j]));
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      const int ef = (__builtin_bit_cast(int, m) >> 23) & 255;
      const int sb = (ef == 0 || ef == 255) ? 127 : min(267 - ef, 253);
      const float Sc = __builtin_bit_cast(float, sb << 23);
      const float osc = __builtin_bit_cast(float, (254 - sb) << 23) * (1.0f / 256.0f);
      u32x4 xh[2], xl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned hp, lp;
          const f32x4 q = g[2 * u + (j >> 1)];
          split2_pair(Sc * q[2 * (j & 1)], Sc * q[2 * (j & 1) + 1], hp, lp);
          xh[u][j] = hp; xl[u][j] = lp;
        }
      const char* wl_base = wimg + (lane << 4);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          const u32x4 wh = *reinterpret_cast<const u32x4*>(wl_base + ((ct * 2 + T) * 2) * 1024);
          const u32x4 wl = *reinterpret_cast<const u32x4*>(wl_base + ((ct * 2 + T) * 2 + 1) * 1024);
          // (the order of mp_win16_bwd.hip: per 32-wide step the two small products, then the leading one)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl), __builtin_bit_cast(f16x8, xh[T]), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xl[T]), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xh[T]), acc, 0, 0, 0);
        }
        dA[ct] = acc * osc;
      }
    } else {
      // weights beyond the fp16 piece range: f32-input MFMA, one contraction index of the lane per instruction, the weight out
      // of the fp32 fragment image (correct, not fast; never run in practice)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) dA[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      // image index of W(k = 32 u + 8 kg + t, o = 16 ct + at): ((ct 4 + (k >> 4)) 64 + ((k & 15) >> 2) 16 + at) 4 + (k & 3) = a lane part
      // + a compile-time part (one address register instead of 192)
      const float* wb = a.WfragT32 + (((kg >> 1) * 64 + (kg & 1) * 32 + at) << 2);
      // a rolled loop over the lane's sixteen contraction indices (unrolled, the 192 weight loads were hoisted and spilled): the
      // operand value by a select chain
#pragma unroll 1
      for (int ut = 0; ut < 16; ++ut) {
        const int u = ut >> 3, t = ut & 7;
        float x = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int j = 0; j < 4; ++j) x = (4 * v + j == ut) ? g[v][j] : x;       // (g[2 u + (t >> 2)][t & 3]: index 4 (2 u + (t >> 2)) + (t & 3) = ut)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const float wv = wb[(((ct * 4 + 2 * u) * 64 + (t >> 2) * 16) << 2) + (t & 3)];
          dA[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, x, dA[ct], 0, 0, 0);
        }
      }
    }
  };

#pragma unroll 1
  for (int64_t g0 = A0; g0 < A1; g0 += GROUP) {
    const int64_t wlo64 = std::max<int64_t>(0, std::min<int64_t>(g0 - (WROWS - GROUP) / 2, a.N - WROWS));
    const int wlo = (int)wlo64;
    // ---- workgroup event: every wave is through with the old window (the barrier at the end of the group before); the new one
    // travels while the first micro-tile's matrix interval — which needs no window — runs
    win_dma(win, a.h, wlo64, a.N, wave, lane);
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
      const int64_t row0 = g0 + (int64_t)(wave + NWV * i) * MT;
      if (row0 >= A1) {
        if (i == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); NG_LDS_BARRIER(); }
        break;
      }
#ifndef WV_NOPRIO
      if (i == 1 && wave >= NWV / 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
#endif
      if (have != row0) { nlist_dma(a.nlist, K, a.N, strip, row0, lane); have = row0; }
      if (fetched != row0) fetch(row0);
      matrix(row0);
      // the indices of this micro-tile (requested a micro-tile ago) and, for the group's first one, the window
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (i == 0) NG_LDS_BARRIER();
      // the next micro-tile's rows and this one's old edge gradient travel beside the dots
      const int64_t nx = row0 + (int64_t)NWV * MT;
      if (nx < A1) { fetch(nx); fetched = nx; }
      const int64_t row = row0 + at;
      const bool wr = row < a.N && 4 * kg < K;
      f32x4 old[3];
      if (a.accumulate) {
        const f32x4* pe = reinterpret_cast<const f32x4*>(a.de + ((wr ? row : 0) * K + (wr ? 4 * kg : 0)) * E);
        old[0] = pe[0]; old[1] = pe[1]; old[2] = pe[2];
      }
      const bool inwin = sources_in_window(strip, lane, nq, row0, a.N, wlo);
      float r4[4][E];
      if (inwin) edge_dots<false>(strip, win, a.h, nq, at, kg, wlo, dA, r4);
      else edge_dots<true>(strip, win, a.h, nq, at, kg, wlo, dA, r4);
      // ---- the strip is free: the indices of this wave's next micro-tile
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (nx < A1) { nlist_dma(a.nlist, K, a.N, strip, nx, lane); have = nx; }
      if (wr) {
        f32x4* pe = reinterpret_cast<f32x4*>(a.de + (row * K + 4 * kg) * E);
        f32x4 o0 = {r4[0][0], r4[0][1], r4[0][2], r4[1][0]}, o1 = {r4[1][1], r4[1][2], r4[2][0], r4[2][1]},
              o2 = {r4[2][2], r4[3][0], r4[3][1], r4[3][2]};
        if (a.accumulate) { o0 += old[0]; o1 += old[1]; o2 += old[2]; }
        pe[0] = o0; pe[1] = o1; pe[2] = o2;
      }
    }
    // every wave is through with this window
    if (g0 + GROUP < A1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); NG_LDS_BARRIER(); }
  }
}

__global__ __launch_bounds__(WTHREADS) void mp_wave_bwd_edge_kernel(Args a) {
  if (a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag))) body<false>(a);
  else body<true>(a);
}

}  // namespace wvb

int mp_wave_bwd_edge_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int act, const float* h, const int32_t* nlist,
                            const float* inv_degree, const float* WfragT, const float* s_save, const float* dh_out, float* dP,
                            float* de, int de_accum, RangeGuard guard, const float* WfragT32, const unsigned* wflag) {
  using namespace wvb;
  Args a{};
  a.N = N; a.K = K;
  a.atoms_per_wg = win16_tiles_per_wg(cdiv(N, 64), ctx->num_cu) * 64;
  a.dH = dh_out; a.S = act == NG_ACT_NONE ? nullptr : s_save; a.rowscale = inv_degree; a.h = h;
  a.nlist = nlist; a.WfragT = WfragT; a.WfragT32 = WfragT32; a.dP = dP; a.de = de; a.act = act;
  a.accumulate = de_accum; a.guard = guard; a.wflag = wflag;
  const int grid = (int)cdiv(N, a.atoms_per_wg);
  ProfScope ps(ctx, st, "mp_win_bwd_edge");
  hipLaunchKernelGGL(mp_wave_bwd_edge_kernel, dim3(grid), dim3(WTHREADS), LDS_BYTES, st, a);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

}  // namespace ng
