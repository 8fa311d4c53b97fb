// Sixteen-wave form of the wave-autonomous forward window kernel (mp_wave.hip): four waves per SIMD at <= 128 registers.
// Reference: nmrgnn/layers.py:26-46 (MPLayer.call) + residual of nmrgnn/model.py:165-167.
//
// Why: cycle stamps of the eight-wave kernel (profiles/r06a_mp_wave.txt) show a micro-tile at ~14 k cycles against ~5.7 k of
// instruction issue per wave: with two waves per SIMD one partner covers the other's LDS round trips (the gather's reads pay a
// ~3x bank-conflict factor: sixteen random rows per ds_read_b128 group), the matrix interval's operand reads and the
// transcendental-heavy epilogue — and a wave cannot cover its own.  Here a workgroup has sixteen waves and a group of 256 atoms
// is ONE micro-tile per wave:
//   gather    all sixteen waves walk their lists out of the same window: LDS-bound (~50 LDS cycles per wave and entry);
//   barrier   the window is free: the next group's window and lists are requested;
//   finish    piece split, out^T by MFMA one column tile at a time (8 accumulator registers instead of 32; the tile's activation,
//             residual and stores follow its last MFMA at once), four waves per SIMD filling each other's MFMA / LDS / v_exp gaps;
//   barrier   window and lists have landed.
// What had to go for 128 registers: the list strips (64 KB for sixteen waves) became a two-slot ring of list QUADS per wave
// (2 KB: a quad of a micro-tile = 256 B of neighbour indices + 768 B of edge features, requested two quads ahead); the next
// quad's records are not held in registers; the window check is one ballot per quad.
// Sums, order and rounding are those of mp_wave.hip except for the matrix interval's two accumulators, which are kept.
#include <algorithm>
#include <cstdio>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "edge_fused.h"   // NG_LDS_BARRIER
#include "h2_common.cuh"
#include "mp_win16_common.cuh"
#include "mp_wave_common.cuh"

namespace ng {
namespace wv16 {

using wv::f32x4;
using wv::i32x4;
constexpr int WF = wv::WF, WROWS = wv::WROWS, MT = wv::MT, E = wv::E, NT2 = wv::NT2;
constexpr int WIN_BYTES = wv::WIN_BYTES, WIMG_BYTES = wv::WIMG_BYTES;
constexpr int NWV = 16;
constexpr int WTHREADS = NWV * 64;
constexpr int GROUP = NWV * MT;             // 256 atoms: one micro-tile per wave
constexpr int QUAD_BYTES = 1024;           // [256 B neighbour indices][3 x 256 B edge features], each [16-byte piece][atom]
constexpr int STRIP_BYTES = 2 * QUAD_BYTES;
constexpr int LDS_BYTES = WIN_BYTES + WIMG_BYTES + NWV * STRIP_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct Args {
  int64_t N;
  int K;
  int64_t atoms_per_wg;      // a multiple of GROUP unless the batch is small
  const float* h;            // [N][64]
  const int32_t* nlist;      // [N][K]
  const float* e;            // [N*K][3]
  const float* Wfrag;        // piece fragments (pack_bodies.cuh: mpw_h2<0>)
  const float* Wfrag32;      // fp32 fragments (mpw_f32 mode 0)
  const float* rowscale;     // [N]
  int residual;
  float* out;                // [N][64]
  float* S_save;             // [N][64] or nullptr
  int act;
  RangeGuard guard;
  const unsigned* wflag;
#ifdef WV_STAMP
  unsigned long long* stamps;
#endif
};
#ifdef WV_STAMP
#define W16T(k) do { if (a.stamps && blockIdx.x == 3 && lane == 0) a.stamps[((int)((g0 - A0) / GROUP) * NWV + wave) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define W16T(k) do {} while (0)
#endif

__device__ __forceinline__ void win_dma16(char* win, const float* src, int64_t wlo, int64_t N, int wave, int lane) {
  asm volatile("" : "+v"(lane));      // (per-lane offsets recomputed at every call: hoisted out of the group loop they were spilled)
  const int64_t rows = std::min<int64_t>(N - wlo, WROWS);
  const dma_i4 rs = dma_rsrc(src + wlo * WF, (unsigned)(rows * (WF * 4)));
#pragma unroll
  for (int j = 0; j < (WIN_BYTES / 1024 + NWV - 1) / NWV; ++j) {
    const int kb = wave + NWV * j;
    if (kb < WIN_BYTES / 1024) {
      const int R = 4 * kb + (lane >> 4);
      lds_dma16(rs, win + kb * 1024, R * (WF * 4) + (((lane & 15) ^ (R & 15)) << 4), 0);
    }
  }
}
__device__ __forceinline__ void wimg_dma16(char* wimg, const float* img, int wave, int lane) {
  const dma_i4 rw = dma_rsrc(img, WIMG_BYTES);
#pragma unroll
  for (int j = 0; j < WIMG_BYTES / 1024 / NWV; ++j) {
    const int kb = wave + NWV * j;
    lds_dma16(rw, wimg + kb * 1024, lane * 16, kb * 1024);
  }
}

// quad q (entries 4 q .. 4 q + 3) of the lists of micro-tile row0 .. row0+15 into a ring slot: lanes 0-15 the neighbour indices
// of their atom, lanes 16-63 the three 16-byte pieces of its twelve edge features; rows past N read as zeros
__device__ __forceinline__ void quad_dma(const Args& a, char* slot, int64_t row0, int q, int lane) {
  asm volatile("" : "+v"(lane));
  const int K = a.K, at = lane & 15, pp = lane >> 4;
  const int rows = (int)std::min<int64_t>(MT, a.N - row0);
  if (pp == 0) {
    const dma_i4 rn = dma_rsrc(a.nlist + row0 * K, (unsigned)(rows * K * 4));
    lds_dma16(rn, slot, at * K * 4 + q * 16, 0);
  } else {
    const dma_i4 re = dma_rsrc(a.e + row0 * K * E, (unsigned)(rows * K * E * 4));
    // (the copy lands at slot + 16 lane: lanes 16 .. 63 fill bytes 256 .. 1023)
    lds_dma16(re, slot, at * K * E * 4 + (3 * q + pp - 1) * 16, 0);
  }
}

// entries of one quad: acc[n][j] += e[entry][n] * h[source][feature(j)],  j = 8 u + t  <->  feature 32 u + 8 kg + t
template <bool GLOBAL>
__device__ __forceinline__ void gather_quad(const char* __restrict__ slot, const char* __restrict__ win, const float* __restrict__ h,
                                            int at, int kg, int wlo, float (&acc)[E][16]) {
  const int kc0 = (2 * kg) << 4;
  const char* rec = slot + (at << 4);
  const i32x4 ci = *reinterpret_cast<const i32x4*>(rec);
  float ef[12];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(rec + 256 + (i << 8));
    ef[4 * i] = v[0]; ef[4 * i + 1] = v[1]; ef[4 * i + 2] = v[2]; ef[4 * i + 3] = v[3];
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    f32x4 hr[4];
    if (!GLOBAL) {
      const int R = min(max(ci[s] - wlo, 0), WROWS - 1);
      const int sw = (R & 15) << 4;
      const char* row = win + (R << 8);
      hr[0] = *reinterpret_cast<const f32x4*>(row + (kc0 ^ sw));
      hr[1] = *reinterpret_cast<const f32x4*>(row + ((kc0 ^ 16) ^ sw));
      hr[2] = *reinterpret_cast<const f32x4*>(row + ((kc0 ^ 128) ^ sw));
      hr[3] = *reinterpret_cast<const f32x4*>(row + ((kc0 ^ 144) ^ sw));
    } else {
      const f32x4* row = reinterpret_cast<const f32x4*>(h + (int64_t)ci[s] * WF);
      hr[0] = row[2 * kg]; hr[1] = row[2 * kg + 1]; hr[2] = row[8 + 2 * kg]; hr[3] = row[9 + 2 * kg];
    }
#pragma unroll
    for (int n = 0; n < E; ++n) {
      const float w = ef[3 * s + n];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[n][4 * r + j] = __builtin_fmaf(w, hr[r][j], acc[n][4 * r + j]);
    }
  }
}

template <bool H2>
__device__ __forceinline__ void body(const Args& a) {
  extern __shared__ __attribute__((aligned(16))) char smem_wv16[];
  char* win = smem_wv16;
  char* wimg = win + WIN_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* strip = wimg + WIMG_BYTES + wave * STRIP_BYTES;
  const int at = lane & 15, kg = lane >> 4;
  const int K = a.K, nq = K >> 2;
  const int64_t A0 = (int64_t)blockIdx.x * a.atoms_per_wg;
  const int64_t A1 = std::min<int64_t>(A0 + a.atoms_per_wg, a.N);
  if (A0 >= A1) return;
  const float resf = a.residual ? 1.f : 0.f;

  if (H2) wimg_dma16(wimg, a.Wfrag, wave, lane);
  {
    const int64_t wlo0 = std::max<int64_t>(0, std::min<int64_t>(A0 - (WROWS - GROUP) / 2, a.N - WROWS));
    win_dma16(win, a.h, wlo0, a.N, wave, lane);
    const int64_t r0 = A0 + (int64_t)wave * MT;
    if (r0 < A1) {
      quad_dma(a, strip, r0, 0, lane);
      if (nq > 1) quad_dma(a, strip + QUAD_BYTES, r0, 1, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NG_LDS_BARRIER();
  }

#pragma unroll 1
  for (int64_t g0 = A0; g0 < A1; g0 += GROUP) {
    const int wlo = (int)std::max<int64_t>(0, std::min<int64_t>(g0 - (WROWS - GROUP) / 2, a.N - WROWS));
    const int64_t row0 = g0 + (int64_t)wave * MT;
    const bool mine = row0 < A1;
    const bool more = g0 + GROUP < A1;
    float acc[E][16];
    f32x4 rew[4];
    bool ownwin = false;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) rew[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    W16T(0);
    if (mine) {
#pragma unroll
      for (int n = 0; n < E; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;
      // ---- gather: the quads out of the two-slot ring; quad q + 2 is requested when quad q has been read
#pragma unroll 1
      for (int q = 0; q < nq; ++q) {
        char* slot = strip + (q & 1) * QUAD_BYTES;
        if (q >= 2) {      // (quads 0 and 1 landed before the group's barrier)
          if (q + 1 < nq) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // do the quad's sources lie in the window?  one ballot (rows past N hold zeros: masked)
        bool outside = false;
        if (lane < 16 && row0 + lane < a.N) {
          const i32x4 v = *reinterpret_cast<const i32x4*>(slot + (lane << 4));
#pragma unroll
          for (int j = 0; j < 4; ++j) outside |= v[j] < wlo || v[j] >= wlo + WROWS;
        }
        const bool inwin = __builtin_amdgcn_ballot_w64(outside) == 0;
        if (inwin) gather_quad<false>(slot, win, a.h, at, kg, wlo, acc);
        else gather_quad<true>(slot, win, a.h, at, kg, wlo, acc);
        if (q + 2 < nq) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot has been read
          quad_dma(a, slot, row0, q + 2, lane);
        }
      }
      // the residual rows come out of the window when the micro-tile's own rows are inside (they are, for a window placed around
      // the group); read now: behind the barrier the window is the next group's
      ownwin = a.residual && row0 >= wlo && row0 + MT <= (int64_t)wlo + WROWS;
      if (ownwin) {
        const int R = (int)(row0 - wlo) + at;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) rew[ct] = *reinterpret_cast<const f32x4*>(win + wv::win_off(R, 4 * ct + kg));
      }
    }
    // ---- every wave is through with this window (and its strip): the next group's window and first two quads travel beside
    // the matrix interval, activation and stores of this group
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W16T(1);
    if (more) {
      NG_LDS_BARRIER();
      W16T(2);
      const int64_t g1 = g0 + GROUP;
      win_dma16(win, a.h, std::max<int64_t>(0, std::min<int64_t>(g1 - (WROWS - GROUP) / 2, a.N - WROWS)), a.N, wave, lane);
      const int64_t r1 = g1 + (int64_t)wave * MT;
      if (r1 < A1) {
        quad_dma(a, strip, r1, 0, lane);
        if (nq > 1) quad_dma(a, strip + QUAD_BYTES, r1, 1, lane);
      }
    }
    if (mine) {
      const int64_t row = row0 + at;
      const bool live = row < a.N;
      const int64_t rowc = live ? row : a.N - 1;
      const float rs = a.rowscale[rowc];
      float rsx = rs;
      u32x4 xh[NT2], xl[NT2];
      if (H2) {
        // a row that reaches 2^15 is scaled by a power of two (never taken for ordinary activations)
        float m = 0.f;
#pragma unroll
        for (int n = 0; n < E; ++n)
#pragma unroll
          for (int j = 0; j < 16; ++j) m = fmaxf(m, fabsf(acc[n][j]));
        float rsv = 1.0f;
        if (__builtin_amdgcn_ballot_w64(m >= 32768.0f) != 0) {
          m = fmaxf(m, __shfl_xor(m, 16));
          m = fmaxf(m, __shfl_xor(m, 32));
          const int ef = (__builtin_bit_cast(int, m) >> 23) & 255;
          const bool big = ef >= 127 + 15 && ef != 255;
          const float S = big ? __builtin_bit_cast(float, (268 - ef) << 23) : 1.0f;
          rsv = big ? __builtin_bit_cast(float, (ef - 14) << 23) : 1.0f;
#pragma unroll
          for (int n = 0; n < E; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[n][j] *= S;
        }
#pragma unroll
        for (int n = 0; n < E; ++n)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              unsigned hp, lp;
              split2_pair(acc[n][8 * u + 2 * j], acc[n][8 * u + 2 * j + 1], hp, lp);
              xh[2 * n + u][j] = hp; xl[2 * n + u][j] = lp;
            }
        rsx = rs * (1.0f / 256.0f) * rsv;
      }
      W16T(3);
      const char* wl_base = wimg + (lane << 4);
      const float* wb32 = a.Wfrag32 + (((kg >> 1) * 64 + (kg & 1) * 32 + at) << 2);
      // ---- one column tile at a time: its six k-steps, then its activation, residual and stores (lane (atom, kg) holds
      // out[atom][16 ct + 4 kg .. + 3])
      // (a rolled loop: unrolled, the scheduler hoists the 48 operand reads of all four tiles and the allocator spills)
#pragma unroll 1
      for (int ct = 0; ct < 4; ++ct) {
        // the residual chunk of this column tile: out of the window (read before the barrier), or requested from memory here
        f32x4 re4 = ct == 0 ? rew[0] : (ct == 1 ? rew[1] : (ct == 2 ? rew[2] : rew[3]));
        if (a.residual && !ownwin) re4 = *reinterpret_cast<const f32x4*>(a.h + rowc * WF + 16 * ct + 4 * kg);
        if (!a.residual) re4 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 o;
        if (H2) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int T = 0; T < NT2; ++T) {
            const u32x4 wh = *reinterpret_cast<const u32x4*>(wl_base + ((ct * NT2 + T) * 2) * 1024);
            const u32x4 wl = *reinterpret_cast<const u32x4*>(wl_base + ((ct * NT2 + T) * 2 + 1) * 1024);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl), __builtin_bit_cast(f16x8, xh[T]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xh[T]), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xl[T]), acc0, 0, 0, 0);
          }
          o = acc0 + acc1;
        } else {
          // weights beyond the fp16 piece range: f32-input MFMA, one contraction index of the lane per instruction, the weight
          // out of the fp32 fragment image (a rolled loop: correct, not fast; never run in practice).  Image index of
          // W(k, o = 16 ct + at), k = 64 n + 32 u + 8 kg + t: ((ct 12 + (k >> 4)) 64 + ((k & 15) >> 2) 16 + at) 4 + (k & 3)
          o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
          for (int nj = 0; nj < E * 16; ++nj) {
            const int n = nj >> 4, j = nj & 15, u = j >> 3, t = j & 7;
            float x = 0.f;
#pragma unroll
            for (int nn = 0; nn < E; ++nn)
#pragma unroll
              for (int jj = 0; jj < 16; ++jj) x = (16 * nn + jj == nj) ? acc[nn][jj] : x;
            const float wv = wb32[(((ct * 12 + 4 * n + 2 * u) * 64 + (t >> 2) * 16) << 2) + (t & 3)];
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, x, o, 0, 0, 0);
          }
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = o[j] * rsx;
          if (a.act == NG_ACT_SOFTPLUS) v[j] = softplus_f(v[j]);
          else if (a.act != NG_ACT_NONE) v[j] = act_apply(a.act, v[j]);
        }
        if (live) {
          typedef float nt4 __attribute__((ext_vector_type(4)));
          const f32x4 r = re4;
          *reinterpret_cast<f32x4*>(a.out + row * WF + 16 * ct + 4 * kg) =
              f32x4{v[0] + resf * r[0], v[1] + resf * r[1], v[2] + resf * r[2], v[3] + resf * r[3]};
          // (the activation copy is read a millisecond later by the backward: past the caches)
          if (a.S_save) __builtin_nontemporal_store(nt4{v[0], v[1], v[2], v[3]}, reinterpret_cast<nt4*>(a.S_save + row * WF + 16 * ct + 4 * kg));
        }
      }
    }
    W16T(4);
    if (more) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      W16T(5);
      NG_LDS_BARRIER();
      W16T(6);
    }
  }
}

__global__ __launch_bounds__(WTHREADS) void mp_wave16_fwd_kernel(Args a) {
  if (a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag))) body<false>(a);
  else body<true>(a);
}

}  // namespace wv16

int mp_wave16_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int act, int residual, const float* h, const int32_t* nlist,
                     const float* e, const float* inv_degree, const float* Wfrag, const float* Wf32, const unsigned* wflag,
                     RangeGuard guard, float* h_out, float* s_save) {
  using namespace wv16;
  Args a{};
  a.N = N; a.K = K;
  a.atoms_per_wg = win16_tiles_per_wg(cdiv(N, 64), ctx->num_cu) * 64;
  a.h = h; a.nlist = nlist; a.e = e; a.Wfrag = Wfrag; a.Wfrag32 = Wf32; a.rowscale = inv_degree; a.residual = residual;
  a.out = h_out; a.S_save = s_save; a.act = act;
  a.guard = guard; a.wflag = wflag;
  const int grid = (int)cdiv(N, a.atoms_per_wg);
#ifdef WV_STAMP
  static unsigned long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { (void)hipMalloc(&dbg, 2 * NWV * 8 * 8); (void)hipMemset(dbg, 0, 2 * NWV * 8 * 8); }
  a.stamps = dbg;
#endif
  ProfScope ps(ctx, st, "mp_win_fwd");
  hipLaunchKernelGGL(mp_wave16_fwd_kernel, dim3(grid), dim3(WTHREADS), LDS_BYTES, st, a);
  NG_HIP(ctx, hipGetLastError());
#ifdef WV_STAMP
  if (++calls == 40) {
    unsigned long long hb[2 * NWV * 8];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hb, dbg, sizeof(hb), hipMemcpyDeviceToHost);
    const unsigned long long t0 = hb[0];
    for (int g = 0; g < 2; ++g)
      for (int w = 0; w < NWV; ++w) {
        const unsigned long long* p = hb + (g * NWV + w) * 8;
        fprintf(stderr, "W16 group %d wave %2d: start %6lld gather %5lld bar1 %5lld dma+split %5lld mfma+act+stores %5lld wait %5lld bar2 %5lld\n", g, w,
                (long long)(p[0] - t0), (long long)(p[1] - p[0]), (long long)(p[2] - p[1]), (long long)(p[3] - p[2]), (long long)(p[4] - p[3]),
                (long long)(p[5] - p[4]), (long long)(p[6] - p[5]));
      }
  }
#endif
  return NG_OK;
}

}  // namespace ng
