// Sixteen-wave form of the edge-side backward window kernel of mp_win_bwd.hip (atom_feature_size 64, E <= 3, K <= 16).
// Reference: the backward of nmrgnn/layers.py:26-46:  dP = dH * act'(S) * v ;  dA = dP Wp^T ;  de (+)= <dA, h[nlist]>.
//
// Same reasoning as mp_win16.hip: the eight-wave kernel waits (two waves per SIMD, 220 VGPRs), it does not issue.  Here a
// workgroup has 1024 threads and walks 64-atom tiles with four waves per SIMD at 128 VGPRs:
//   * matrix interval: dA[64][E*64] = dP[64][64] Wp^T is 4 row tiles x 4E column tiles.  Wave w < 4E takes column tile w
//     for all four row tiles — ONE column tile's fragments in registers (16 VGPRs instead of 48), 24 MFMAs; the waves
//     beyond 4E sit the interval out (the matrix pipe is 5 % busy in this kernel: nothing is lost);
//   * neighbour indices loaded by the lane that uses them (lane = (atom, slot) of the rotation walk), no list staging;
//   * dP piece planes single-buffered (the matrix interval of a tile is over before the next tile's rows are committed);
//   * window by LDS-DMA, requested at the top of the tile and awaited in front of the barrier behind the matrix interval.
// LDS: window 72 KB + dA tile 49 KB + dP planes 18 KB.  de comes out bit for bit as from the eight-wave kernel (same dots,
// same order); dA — and with it de — differs only where the fp32 body is taken (weights beyond the piece range: fragments
// from the image per step).
#include <algorithm>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "edge_fused.h"   // NG_LDS_BARRIER
#include "h2_common.cuh"
#include "mp_win16_common.cuh"

namespace ng {
namespace w16b {

using namespace w16c;

constexpr int PROWB = (WF + 8) * 2, PPLANE = WTA * PROWB;      // dP piece planes: 144 B per row
constexpr int SDP_LD = 68;                                     // fp32 dP rows (fp32 body)
constexpr int DP_BYTES = 2 * PPLANE;                           // >= WTA * SDP_LD * 4
static_assert(DP_BYTES >= WTA * SDP_LD * 4, "the dP slot holds either form");

struct Args {
  int64_t N;
  int K;
  int64_t ntiles;
  int tiles_per_wg;
  const float* dH;         // [N][64] upstream gradient of the layer output
  const float* S;          // [N][64] saved activation output, or nullptr (linear)
  const float* rowscale;   // [N]
  const float* h;          // [N][64] layer input (gathered)
  const int32_t* nlist;    // [N][K]
  const float* WfragT;     // piece fragments (mpw_h2<2>)
  const float* WfragT32;   // fp32 fragments (mpw_f32 mode 2)
  float* dP;               // [N][64] out
  float* de;               // [N*K][E] out (+= when accumulate)
  float* dummy;            // >= 64 floats
  int act;
  int accumulate;
  RangeGuard guard;
  const unsigned* wflag;
  unsigned wflag_ver;
};

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// one rotation step of the edge-gradient dot (mp_win_bwd.hip: edge_step): this lane's chunk of dA[i][n][:] against the row
// of the slot that the rotation brings here; the partial goes back to the accumulator of that slot's lane
template <int E, int S, bool GLOBAL>
__device__ __forceinline__ void edge_step(const char* __restrict__ wbytes, const float4* __restrict__ src4, int c, int roff,
                                          int gidx, const float4 (&da)[E], float (&out)[E]) {
  float4 hrow;
  if (!GLOBAL) hrow = *reinterpret_cast<const float4*>(wbytes + ror_i<S>(roff));
  else hrow = src4[(int64_t)ror_i<S>(gidx) * WC4 + c];
#pragma unroll
  for (int n = 0; n < E; ++n) {
    const float p = dot4(da[n], hrow);
    out[n] += ror_f<(16 - S) & 15>(p);
  }
}
template <int E, bool GLOBAL>
__device__ __forceinline__ void edge_dot(int lane, int al, int wlo, int idx, const float* __restrict__ tb, int ld,
                                         const float4* __restrict__ win4, const float4* __restrict__ src4, float (&out)[E]) {
  const int c = lane & 15;
  const int roff = min(max(idx - wlo, 0), WROWS - 1) * (WF * 4);
  const char* wbytes = reinterpret_cast<const char*>(win4) + 16 * c;
  float4 da[E];
#pragma unroll
  for (int n = 0; n < E; ++n) {
    da[n] = *reinterpret_cast<const float4*>(tb + al * ld + n * WF + 4 * c);
    out[n] = 0.f;
  }
  edge_step<E, 0, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 1, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 2, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 3, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 4, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 5, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 6, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 7, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 8, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 9, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 10, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 11, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 12, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 13, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 14, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 15, GLOBAL>(wbytes, src4, c, roff, idx, da, out);
}
template <int E>
struct EdgeDots { float v[E]; };
// out of line: its global loads must not put vmcnt waits into the window path
template <int E>
__device__ __noinline__ EdgeDots<E> edge_dot_global(int lane, int al, int idx, const float* tb, int ld, const float4* src4) {
  float out[E];
  edge_dot<E, true>(lane, al, 0, idx, tb, ld, nullptr, src4, out);
  EdgeDots<E> r;
#pragma unroll
  for (int n = 0; n < E; ++n) r.v[n] = out[n];
  return r;
}

template <int E, bool H2, int ACT = -1>      // ACT: compile-time activation (softplus) or -1 = from the arguments (mp_wave.hip: body)
__device__ __forceinline__ void body(const Args& a) {
  const int act_ = ACT >= 0 ? ACT : a.act;
  constexpr int KF = E * WF;
  constexpr int LD = KF + 4;
  constexpr int NCT = KF / 16;              // column tiles of dA: one per matrix wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* win = smem;                                                   // [WROWS][64]
  float* tile = win + WROWS * WF;                                      // [64][LD]   dA
  char* planes = reinterpret_cast<char*>(tile + WTA * LD);             // dP: two fp16 planes [64][72], or fp32 [64][68]
  float* s_inv = reinterpret_cast<float*>(planes + DP_BYTES);          // [64]  2^-8 / S per dP row
  int* ctl = reinterpret_cast<int*>(s_inv + WTA);                      // [2][2 NW]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;
  const int64_t T0 = (int64_t)blockIdx.x * a.tiles_per_wg;
  const int64_t T1 = std::min<int64_t>(T0 + a.tiles_per_wg, a.ntiles);
  if (T0 >= T1) return;

  const float4* src4 = reinterpret_cast<const float4*>(a.h);
  float4* win4 = reinterpret_cast<float4*>(win);
  for (int t = tid; t < WROWS * WC4; t += WTHREADS) win4[t] = f4zero();

  // matrix role: column tile `wave` of dA (waves >= NCT have none)
  const bool mx = wave < NCT;
  const int a16 = lane & 15, g4 = lane >> 4;
  u32x4 wh[2], wl[2];
  if (H2 && mx) {
    const u32x4* p = reinterpret_cast<const u32x4*>(a.WfragT) + (size_t)(wave * 2) * 2 * 64 + lane;
#pragma unroll
    for (int Ts = 0; Ts < 2; ++Ts) { wh[Ts] = p[(2 * Ts) * 64]; wl[Ts] = p[(2 * Ts + 1) * 64]; }
#pragma unroll
    for (int Ts = 0; Ts < 2; ++Ts)
#pragma unroll
      for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(wh[Ts][j])); asm volatile("" : "+v"(wl[Ts][j])); }
  }

  // vector roles.  edge_dot: lane = (atom al, slot c).  commit: thread = (row prow, float4 column pc) — the same split of tid
  const int al = wave * 4 + (lane >> 4);
  const int prow = tid >> 4, pc = tid & 15;
  // per-tile inputs in flight
  int p_idx;
  float4 p_dh, p_s;
  float p_rs;
  float p_de[E];
  auto issue = [&](int64_t t) {
    const int64_t row = t * WTA + prow;
    const int64_t rc = row < a.N ? row : a.N - 1;
    p_idx = a.nlist[rc * K + (pc < K ? pc : 0)];
    p_dh = *reinterpret_cast<const float4*>(a.dH + rc * WF + 4 * pc);
    p_s = a.S ? *reinterpret_cast<const float4*>(a.S + rc * WF + 4 * pc) : f4zero();
    p_rs = a.rowscale[rc];
    if (row >= a.N) p_dh = f4zero();
  };
  auto issue_de = [&](int64_t t) {       // old edge gradient of (atom, slot)
    const int64_t row = t * WTA + prow;
    const int64_t rc = row < a.N ? row : a.N - 1;
    const int sc = pc < K ? pc : 0;
#pragma unroll
    for (int n = 0; n < E; ++n) p_de[n] = a.accumulate ? a.de[(rc * K + sc) * E + n] : 0.f;
  };
  // dP rows of tile t into the planes (and HBM), the tile's neighbour range into ctl
  auto commit = [&](int64_t t) {
    float4 g = p_dh;
    if (act_ != NG_ACT_NONE) {
      g.x *= act_grad_from_out(act_, p_s.x); g.y *= act_grad_from_out(act_, p_s.y);
      g.z *= act_grad_from_out(act_, p_s.z); g.w *= act_grad_from_out(act_, p_s.w);
    }
    g.x *= p_rs; g.y *= p_rs; g.z *= p_rs; g.w *= p_rs;
    if (H2) {
      float m = fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w)));
      m = fmaxf(m, ror_f<8>(m)); m = fmaxf(m, ror_f<4>(m)); m = fmaxf(m, ror_f<2>(m)); m = fmaxf(m, ror_f<1>(m));
      const int ef = (__builtin_bit_cast(int, m) >> 23) & 255;
      const int sb = (ef == 0 || ef == 255) ? 127 : min(267 - ef, 253);
      const float S = __builtin_bit_cast(float, sb << 23);
      if (pc == 0) s_inv[prow] = __builtin_bit_cast(float, (254 - sb) << 23) * (1.0f / 256.0f);
      unsigned h0, l0, h1, l1;
      split2_pair(S * g.x, S * g.y, h0, l0); split2_pair(S * g.z, S * g.w, h1, l1);
      char* q = planes + prow * PROWB + 8 * pc;
      *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(q + PPLANE) = u32x2{l0, l1};
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(planes) + prow * SDP_LD + 4 * pc) = g;
    }
    const int64_t row = t * WTA + prow;
    *reinterpret_cast<float4*>(row < a.N ? a.dP + row * WF + 4 * pc : a.dummy + 4 * pc) = g;
    const bool live = pc < K && row < a.N;
    int lo = live ? p_idx : 0x7fffffff, hi = live ? p_idx : -1;
    lo = wave_min_i32(lo);
    hi = -wave_min_i32(-hi);
    if (lane == 63) { ctl[(t & 1) * (2 * NW) + wave] = lo; ctl[(t & 1) * (2 * NW) + NW + wave] = hi; }
  };

  int wlo = -(1 << 30), mode = 0;
  issue(T0);
  commit(T0);
  int idx_cur = p_idx;
  issue(T0 + 1 < T1 ? T0 + 1 : T0);
  issue_de(T0);
  NG_LDS_BARRIER();

#pragma unroll 1
  for (int64_t t = T0; t < T1; ++t) {
    // (no wave reads the window between the last barrier and the one behind the matrix interval)
    const bool staged = win_decide(ctl + (t & 1) * (2 * NW), wlo, mode);
    if (staged) win_dma(win, a.h, wlo, a.N, wave, lane);
    // ---- matrix interval: dA tile = dP tile x Wp^T, column tile `wave`, row tile after row tile
    if (mx) {
      if (H2) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int r = 16 * rt + a16;
          const char* xr = planes + r * PROWB + 16 * g4;
          const u32x4 xh0 = *reinterpret_cast<const u32x4*>(xr), xl0 = *reinterpret_cast<const u32x4*>(xr + PPLANE);
          const u32x4 xh1 = *reinterpret_cast<const u32x4*>(xr + 64), xl1 = *reinterpret_cast<const u32x4*>(xr + 64 + PPLANE);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          // (the order of the eight-wave kernel: per 32-wide step the two small products, then the leading one)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[0]), __builtin_bit_cast(f16x8, xh0), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[0]), __builtin_bit_cast(f16x8, xl0), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[0]), __builtin_bit_cast(f16x8, xh0), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[1]), __builtin_bit_cast(f16x8, xh1), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[1]), __builtin_bit_cast(f16x8, xl1), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[1]), __builtin_bit_cast(f16x8, xh1), acc, 0, 0, 0);
          const float osc = s_inv[r];
          *reinterpret_cast<float4*>(tile + r * LD + 16 * wave + 4 * g4) = make_float4(acc[0] * osc, acc[1] * osc, acc[2] * osc, acc[3] * osc);
        }
      } else {
        // fp32 fragments from the image per step (weights beyond the piece range: correct, not fast)
        const float4* p32 = reinterpret_cast<const float4*>(a.WfragT32) + (size_t)(wave * 4) * 64 + lane;
#pragma unroll 1
        for (int rt = 0; rt < 4; ++rt) {
          const int r = 16 * rt + a16;
          const float* xrow = reinterpret_cast<const float*>(planes) + r * SDP_LD + 4 * g4;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int T = 0; T < 4; ++T) {
            const float4 wv = p32[T * 64];
            const float4 x = *reinterpret_cast<const float4*>(xrow + 16 * T);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, x.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, x.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, x.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, x.w, acc, 0, 0, 0);
          }
          *reinterpret_cast<float4*>(tile + r * LD + 16 * wave + 4 * g4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
      }
    }
    if (staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NG_LDS_BARRIER();
    // ---- vector interval: de of tile t, then the dP rows / range of tile t+1, requests for t+2
    {
      float out[E];
      if (mode == 0) edge_dot<E, false>(lane, al, wlo, idx_cur, tile, LD, win4, src4, out);
      else {
        const EdgeDots<E> r = edge_dot_global<E>(lane, al, idx_cur, tile, LD, src4);
#pragma unroll
        for (int n = 0; n < E; ++n) out[n] = r.v[n];
      }
      const int64_t row = t * WTA + prow;
      const bool live = row < a.N && pc < K;
      float* dst = live ? a.de + (row * K + pc) * E : a.dummy;
#pragma unroll
      for (int n = 0; n < E; ++n) dst[n] = out[n] + p_de[n];
    }
    if (t + 1 < T1) commit(t + 1);
    idx_cur = p_idx;
    issue(t + 2 < T1 ? t + 2 : t);
    issue_de(t + 1 < T1 ? t + 1 : t);
    NG_LDS_BARRIER();
  }
}

template <int E>
__global__ __launch_bounds__(WTHREADS) void mp_win16_bwd_edge_kernel(Args a) {
  if (a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag))) body<E, false>(a);
  else if (a.act == NG_ACT_SOFTPLUS) body<E, true, NG_ACT_SOFTPLUS>(a);
  else body<E, true>(a);
}

}  // namespace w16b

static size_t mp_win16_bwd_edge_lds(int E) {
  return (size_t)(w16b::WROWS * w16b::WF + w16b::WTA * (E * w16b::WF + 4) + w16b::WTA + 4 * w16b::NW) * 4 + w16b::DP_BYTES;
}

bool mp_win16_bwd_edge_supported(int E, int K) { return E >= 1 && E <= 3 && K >= 1 && K <= 16; }

int mp_win16_bwd_edge_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, const float* h, const int32_t* nlist,
                             const float* inv_degree, const float* WfragT, const float* s_save, const float* dh_out, float* dP,
                             float* de, int de_accum, float* dummy, RangeGuard guard, const float* WfragT32, const unsigned* wflag,
                             unsigned wflag_ver) {
  using namespace w16b;
  Args a{};
  a.N = N; a.K = K; a.ntiles = cdiv(N, WTA);
  // contiguous runs of tiles per workgroup: multiples of 4 (256 atoms) when the batch is large enough (ng_internal.h)
  const int64_t per = win16_tiles_per_wg(a.ntiles, ctx->num_cu);
  a.tiles_per_wg = (int)per;
  a.dH = dh_out; a.S = act == NG_ACT_NONE ? nullptr : s_save; a.rowscale = inv_degree; a.h = h;
  a.nlist = nlist; a.WfragT = WfragT; a.WfragT32 = WfragT32; a.dP = dP; a.de = de; a.dummy = dummy; a.act = act;
  a.accumulate = de_accum; a.guard = guard; a.wflag = wflag; a.wflag_ver = wflag_ver;
  const int grid = (int)cdiv(a.ntiles, per);
  const size_t lds = mp_win16_bwd_edge_lds(E);
  ProfScope ps(ctx, st, "mp_win_bwd_edge");
  switch (E) {
    case 1: hipLaunchKernelGGL((mp_win16_bwd_edge_kernel<1>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 2: hipLaunchKernelGGL((mp_win16_bwd_edge_kernel<2>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 3: hipLaunchKernelGGL((mp_win16_bwd_edge_kernel<3>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

}  // namespace ng
