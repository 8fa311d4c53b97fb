// Sixteen-wave form of the forward window kernel of mp_win.hip (atom_feature_size 64, E <= 3, padded lists K <= 16).
// Reference: nmrgnn/layers.py:26-46 (MPLayer.call) + residual of nmrgnn/model.py:165-167.
//
// Why: the PMC pass of the eight-wave kernel shows its SIMDs 46 % VALU-busy and 6 % matrix-busy at two waves per SIMD —
// the phases wait on LDS round trips and barriers, they do not issue.  The eight-wave text with 1024 threads and 64-atom tiles
// was bit-identical and took 116 us against 60: at the 128 VGPRs four waves per SIMD leave, the compiler put the wave's
// twelve weight fragments (48 registers at E = 3) in scratch and reloaded them every tile.  This kernel is built for 128:
//   * matrix phase split over the contraction: wave = (column tile ct, row-tile pair rp, k-half kh) holds the fragments of
//     HALF the k-steps (24 registers), forms the partial products of its two 16 x 16 blocks, hands the partial of one block
//     to its partner wave through 16 bytes per lane of LDS and finishes the other block (one more barrier per tile);
//   * lists loaded by the lane that uses them: lane (atom, slot) of the rotation gather reads nlist[atom][slot] and its E
//     weights straight into registers two tiles ahead — no list staging in LDS (32 KB at this tile size), no commit
//     writes, no list reads in the gather;
//   * four row reads of the gather in flight instead of eight (four waves per SIMD cover the LDS latency);
//   * the body for weights beyond the fp16 piece range takes its fp32 fragments from the image per k-step (never run in
//     practice; correct, not fast).
// LDS: window 72 KB + two piece planes 50 KB + exchange tile 16 KB.  Per atom the gather sums in the order of the eight-wave
// kernel; the matrix sums meet as (k-half 0) + (k-half 1), so results agree with it to rounding, not bit for bit.
#include <algorithm>
#include <cstdio>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "edge_fused.h"   // NG_LDS_BARRIER
#include "h2_common.cuh"
#include "mp_win16_common.cuh"

namespace ng {
namespace w16 {

using namespace w16c;

struct Args {
  int64_t N;
  int K;
  int64_t ntiles;
  int tiles_per_wg;
  const float* h;          // [N][64]
  const int32_t* nlist;    // [N][K]
  const float* e;          // [N*K][E]
  const float* Wfrag;      // piece fragments (pack_bodies.cuh: mpw_h2<0>)
  const float* Wfrag32;    // fp32 fragments (mpw_f32 mode 0)
  const float* rowscale;   // [N]
  int residual;
  float* out;              // [N][64]
  float* S_save;           // [N][64] or nullptr
  int act;
  float* dummy;            // 64 floats: where the lanes of rows >= N store
  RangeGuard guard;
  const unsigned* wflag;
  unsigned wflag_ver;
#ifdef W16_STAMP
  unsigned long long* stamps;
#endif
};
#ifdef W16_STAMP
#define W16_T(k) do { if (a.stamps && blockIdx.x == 3 && lane == 0 && t - T0 >= 2 && t - T0 < 6) a.stamps[((t - T0 - 2) * 16 + wave) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define W16_T(k) do {} while (0)
#endif

template <int E>
struct Tile {
  static constexpr int KF = E * WF;
  static constexpr int LD = KF + 4;                    // fp32 row stride (floats)
  static constexpr int ROWB = (KF + 8) * 2;            // fp16 plane row stride (bytes)
  static constexpr int PLANE = WTA * ROWB;
  static constexpr int BYTES = (2 * PLANE > WTA * LD * 4) ? 2 * PLANE : WTA * LD * 4;
};

// as mp_win.hip: tile_put (piece planes with a power-of-two row scale when a row reaches 2^15, or fp32 rows)
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
      const float S = big ? __builtin_bit_cast(float, (268 - ef) << 23) : 1.0f;
      rsv = big ? __builtin_bit_cast(float, (ef - 14) << 23) : 1.0f;
      const f32x2 S2 = {S, S};
#pragma unroll
      for (int n = 0; n < E; ++n) { lo[n] *= S2; hi[n] *= S2; }
    }
    if (c == 0) rs[al] = rsv;
    char* p = reinterpret_cast<char*>(tb) + al * Tile<E>::ROWB + 8 * c;
#pragma unroll
    for (int n = 0; n < E; ++n) {
      unsigned h0, l0, h1, l1;
      split2_pair(lo[n][0], lo[n][1], h0, l0);
      split2_pair(hi[n][0], hi[n][1], h1, l1);
      *reinterpret_cast<u32x2*>(p + n * (WF * 2)) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(p + n * (WF * 2) + Tile<E>::PLANE) = u32x2{l0, l1};
    }
  } else {
#pragma unroll
    for (int n = 0; n < E; ++n)
      *reinterpret_cast<float4*>(tb + al * Tile<E>::LD + n * WF + 4 * c) = make_float4(lo[n][0], lo[n][1], hi[n][0], hi[n][1]);
  }
}

// one lane's list entry: neighbour index and E edge weights of (atom, slot)
template <int E>
struct Slot { int idx; float w[E]; };

template <int E>
__device__ __forceinline__ Slot<E> slot_load(const int32_t* __restrict__ nlist, const float* __restrict__ e, int64_t row, int K,
                                             int c, int64_t N) {
  Slot<E> s;
  const int64_t rc = row < N ? row : N - 1;
  const int64_t q = rc * K + (c < K ? c : 0);
  s.idx = nlist[q];
  if (E == 3) {
    struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
    const F3 v = *reinterpret_cast<const F3*>(e + q * 3);
    s.w[0] = v.x; s.w[E > 1 ? 1 : 0] = v.y; s.w[E > 2 ? 2 : 0] = v.z;
  } else {
#pragma unroll
    for (int n = 0; n < E; ++n) s.w[n] = e[q * E + n];
  }
  return s;
}
// lanes without a slot (c >= K) and rows past the end: weight 0, and out of the window's row range
template <int E>
__device__ __forceinline__ void slot_mask(Slot<E>& s, bool live) {
  if (!live) {
#pragma unroll
    for (int n = 0; n < E; ++n) s.w[n] = 0.f;
  }
}

template <int S0>
__device__ __forceinline__ void rot_load4(const char* __restrict__ wbytes, int roff, float4 (&h)[4]) {
  h[0] = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 0>(roff));
  h[1] = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 1>(roff));
  h[2] = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 2>(roff));
  h[3] = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 3>(roff));
}
template <int S0>
__device__ __forceinline__ void rot_load4_global(const float4* __restrict__ src4, int c, int idx, float4 (&h)[4]) {
  h[0] = src4[(int64_t)ror_i<S0 + 0>(idx) * WC4 + c];
  h[1] = src4[(int64_t)ror_i<S0 + 1>(idx) * WC4 + c];
  h[2] = src4[(int64_t)ror_i<S0 + 2>(idx) * WC4 + c];
  h[3] = src4[(int64_t)ror_i<S0 + 3>(idx) * WC4 + c];
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

// rotation gather of one atom row (mp_win.hip: win_gather_rot) with the lane's slot in registers; GLOBAL: rows from HBM / L2
template <int E, bool H2, bool GLOBAL>
__device__ __forceinline__ void gather(int lane, int al, int wlo, int idx, const float (&w)[E], float* __restrict__ tb,
                                       const float4* __restrict__ win4, const float4* __restrict__ src4, float* __restrict__ rs) {
  const int c = lane & 15;
  const int roff = min(max(idx - wlo, 0), WROWS - 1) * (WF * 4);
  const char* wbytes = reinterpret_cast<const char*>(win4) + 16 * c;
  f32x2 lo[E], hi[E];
#pragma unroll
  for (int n = 0; n < E; ++n) { lo[n] = f32x2{0.f, 0.f}; hi[n] = f32x2{0.f, 0.f}; }
  float4 ha[4];
#define NG_W16_STEP(S0)                                                      \
  if (GLOBAL) rot_load4_global<S0>(src4, c, idx, ha);                        \
  else rot_load4<S0>(wbytes, roff, ha);                                      \
  __builtin_amdgcn_sched_barrier(0);                                         \
  rot_fma4<E, S0>(ha, w, lo, hi);                                            \
  __builtin_amdgcn_sched_barrier(0);
#ifdef W16_ABL_NOGATHER      // timing experiment: one rotation group instead of four
  NG_W16_STEP(0)
#else
  NG_W16_STEP(0) NG_W16_STEP(4) NG_W16_STEP(8) NG_W16_STEP(12)
#endif
#undef NG_W16_STEP
  tile_put<E, H2>(tb, al, c, lo, hi, rs);
}

// kept out of line: inlined next to the window variant its global loads make the compiler put vmcnt waits into the window path
template <int E, bool H2>
__device__ __noinline__ void gather_global(int lane, int al, int idx, float w0, float w1, float w2, float* tb, const float4* src4,
                                           float* rs) {
  float w[E];
  w[0] = w0;
  if (E > 1) w[E > 1 ? 1 : 0] = w1;
  if (E > 2) w[E > 2 ? 2 : 0] = w2;
  gather<E, H2, true>(lane, al, 0, idx, w, tb, nullptr, src4, rs);
}

template <int E, bool H2>
__device__ __forceinline__ void body(const Args& a) {
  constexpr int KF = E * WF;
  constexpr int NT2 = KF / 32, NTH = NT2 / 2;          // 32-wide k-steps in all, per k-half
  constexpr int NT = KF / 16, NT_H = NT / 2;           // fp32 body: 16-wide fragment groups (four 4-wide MFMA steps each)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* win = smem;                                                            // [WROWS][64]
  float* tile = win + WROWS * WF;                                               // piece planes / fp32 rows
  float4* xch = reinterpret_cast<float4*>(reinterpret_cast<char*>(tile) + Tile<E>::BYTES);   // [16 blocks][64 lanes]
  int* ctl = reinterpret_cast<int*>(xch + 16 * 64);                             // [2][2 NW]
  float* s_rs = reinterpret_cast<float*>(ctl + 4 * NW);                         // [WTA]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;
  const int64_t T0 = (int64_t)blockIdx.x * a.tiles_per_wg;
  const int64_t T1 = std::min<int64_t>(T0 + a.tiles_per_wg, a.ntiles);
  if (T0 >= T1) return;

  const float4* src4 = reinterpret_cast<const float4*>(a.h);
  float4* win4 = reinterpret_cast<float4*>(win);
  // matrix role: column tile, row-tile pair, k-half; the block this wave finishes is row tile 2 rp + kh
  const int ct = wave & 3, rp = (wave >> 2) & 1, kh = wave >> 3;
  const int a16 = lane & 15, g = lane >> 4;
  u32x4 wh[H2 ? NTH : 1], wl[H2 ? NTH : 1];
  if (H2) {
    const u32x4* p = reinterpret_cast<const u32x4*>(a.Wfrag) + (size_t)(ct * NT2 + kh * NTH) * 2 * 64 + lane;
#pragma unroll
    for (int T = 0; T < NTH; ++T) { wh[T] = p[(2 * T) * 64]; wl[T] = p[(2 * T + 1) * 64]; }
#pragma unroll
    for (int T = 0; T < NTH; ++T)
#pragma unroll
      for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(wh[T][j])); asm volatile("" : "+v"(wl[T][j])); }
  }
  // gather role: atom al of the tile, slot c
  const int c = lane & 15, al = wave * 4 + (lane >> 4);
  auto range_of = [&](const Slot<E>& s, int64_t t, int* ctl_t) {
    const bool live = c < K && t * WTA + al < a.N;
    int lo = live ? s.idx : 0x7fffffff, hi = live ? s.idx : -1;
    lo = wave_min_i32(lo);
    hi = -wave_min_i32(-hi);
    if (lane == 63) { ctl_t[wave] = lo; ctl_t[NW + wave] = hi; }
  };

  Slot<E> cur = slot_load<E>(a.nlist, a.e, T0 * WTA + al, K, c, a.N);
  slot_mask(cur, c < K && T0 * WTA + al < a.N);
  Slot<E> nxt = slot_load<E>(a.nlist, a.e, (T0 + 1 < T1 ? T0 + 1 : T0) * WTA + al, K, c, a.N);
  range_of(cur, T0, ctl + (T0 & 1) * (2 * NW));
  int wlo = -(1 << 30), mode = 0;
  NG_LDS_BARRIER();
  if (win_decide(ctl + (T0 & 1) * (2 * NW), wlo, mode)) {
    win_dma(win, a.h, wlo, a.N, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  NG_LDS_BARRIER();

  const int rt_own = 2 * rp + kh;
  const int col = 16 * ct + 4 * g;
  const float resf = a.residual ? 1.f : 0.f;
#pragma unroll 1
  for (int64_t t = T0; t < T1; ++t) {
    // ---- lists of t+2 requested, range of t+1 published, epilogue operands requested, tile t gathered
    const int64_t tn = t + 1 < T1 ? t + 1 : t, tnn = t + 2 < T1 ? t + 2 : t;
    W16_T(0);
    Slot<E> nn = slot_load<E>(a.nlist, a.e, tnn * WTA + al, K, c, a.N);
    slot_mask(nxt, c < K && tn * WTA + al < a.N);
    if (t + 1 < T1) range_of(nxt, t + 1, ctl + ((t + 1) & 1) * (2 * NW));
    const int64_t row = t * WTA + 16 * rt_own + a16;          // this lane's atom in the epilogue
    const bool live = row < a.N;
    const int64_t rowc = live ? row : a.N - 1;
    const float rs = a.rowscale[rowc];
    const float4 re = *reinterpret_cast<const float4*>(a.h + rowc * WF + col);
    W16_T(1);
    if (mode == 0) gather<E, H2, false>(lane, al, wlo, cur.idx, cur.w, tile, win4, src4, s_rs);
    else gather_global<E, H2>(lane, al, cur.idx, cur.w[0], cur.w[E > 1 ? 1 : 0], cur.w[E > 2 ? 2 : 0], tile, src4, s_rs);
    W16_T(2);
    NG_LDS_BARRIER();
    W16_T(3);
    // the next tile's window, when it needs one: requested NOW (every gather of this tile is done, its range was published
    // before the barrier) as LDS-DMA and awaited in front of the tile's last barrier: the HBM / L2 round trip runs beside
    // the matrix interval and the epilogue.
    bool restage = false;
    if (t + 1 < T1) restage = win_decide(ctl + ((t + 1) & 1) * (2 * NW), wlo, mode);
    if (restage) win_dma(win, a.h, wlo, a.N, wave, lane);
    // ---- matrix interval: this wave's k-half of its two blocks; the partial of the block it does not finish goes to LDS
    f32x4 part[2];
    if (H2) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const char* xrow = reinterpret_cast<const char*>(tile) + (16 * (2 * rp + hh) + a16) * Tile<E>::ROWB + 16 * g + 64 * (kh * NTH);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // the row tile's operand reads first (NTH pairs), then its MFMAs: a read per step left the LDS latency exposed NTH times
        u32x4 xh[NTH], xl[NTH];
#pragma unroll
        for (int T = 0; T < NTH; ++T) {
          xh[T] = *reinterpret_cast<const u32x4*>(xrow + 64 * T);
          xl[T] = *reinterpret_cast<const u32x4*>(xrow + 64 * T + Tile<E>::PLANE);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int T = 0; T < NTH; ++T) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[T]), __builtin_bit_cast(f16x8, xh[T]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[T]), __builtin_bit_cast(f16x8, xh[T]), acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[T]), __builtin_bit_cast(f16x8, xl[T]), acc0, 0, 0, 0);
        }
        part[hh] = acc0 + acc1;
      }
    } else {
      // fp32 fragments from the image per step (weights beyond the piece range: correct, not fast)
      const float4* p32 = reinterpret_cast<const float4*>(a.Wfrag32) + (size_t)(ct * NT + kh * NT_H) * 64 + lane;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float* xrow = tile + (16 * (2 * rp + hh) + a16) * Tile<E>::LD + 4 * g + 16 * (kh * NT_H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int T = 0; T < NT_H; ++T) {
          const float4 wv = p32[T * 64];
          const float4 x = *reinterpret_cast<const float4*>(xrow + 16 * T);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, x.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, x.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, x.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, x.w, acc, 0, 0, 0);
        }
        part[hh] = acc;
      }
    }
    {
      const f32x4 o = kh ? part[0] : part[1];      // the block the partner finishes: row tile 2 rp + (1 - kh)
      xch[((2 * rp + (1 - kh)) * 4 + ct) * 64 + lane] = make_float4(o[0], o[1], o[2], o[3]);
    }
    W16_T(4);
#ifndef W16_ABL_NOXBAR       // timing experiment: without the exchange barrier (wrong results)
    NG_LDS_BARRIER();
#endif
    W16_T(5);
    {
      const float4 q = xch[(rt_own * 4 + ct) * 64 + lane];
      const f32x4 mine = kh ? part[1] : part[0];
      // (k-half 0) + (k-half 1), whichever of the two this wave computed itself
      const f32x4 k0 = kh ? f32x4{q.x, q.y, q.z, q.w} : mine, k1 = kh ? mine : f32x4{q.x, q.y, q.z, q.w};
      const float rsx = H2 ? rs * (1.0f / 256.0f) * s_rs[16 * rt_own + a16] : rs;
      float4 v = make_float4((k0[0] + k1[0]) * rsx, (k0[1] + k1[1]) * rsx, (k0[2] + k1[2]) * rsx, (k0[3] + k1[3]) * rsx);
#ifdef W16_ABL_NOACT
      if (a.N < 0) {
#else
      if (a.act == NG_ACT_SOFTPLUS) {
#endif
        v.x = softplus_f(v.x); v.y = softplus_f(v.y); v.z = softplus_f(v.z); v.w = softplus_f(v.w);
      } else if (a.act != NG_ACT_NONE) {
        v.x = act_apply(a.act, v.x); v.y = act_apply(a.act, v.y);
        v.z = act_apply(a.act, v.z); v.w = act_apply(a.act, v.w);
      }
      const int64_t o = row * WF + col;
      const float4 vo = make_float4(v.x + resf * re.x, v.y + resf * re.y, v.z + resf * re.z, v.w + resf * re.w);
      asm volatile("" :: "v"(vo.x), "v"(vo.y), "v"(vo.z), "v"(vo.w));
      if (a.S_save) *reinterpret_cast<float4*>(live ? a.S_save + o : a.dummy + col) = v;
      *reinterpret_cast<float4*>(live ? a.out + o : a.dummy + col) = vo;
    }
    cur = nxt;
    nxt = nn;
    W16_T(6);
    if (restage) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the next gather writes the tile planes and s_rs, which this tile's matrix interval and epilogue read; a restaged
    // window must be complete: one barrier for both
    NG_LDS_BARRIER();
    W16_T(7);
  }
}

template <int E>
__global__ __launch_bounds__(WTHREADS) void mp_win16_fwd_kernel(Args a) {
  if (a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag))) body<E, false>(a);
  else body<E, true>(a);
}

}  // namespace w16

size_t mp_win16_lds_bytes(int E) {
  const size_t tile = E == 1 ? w16::Tile<1>::BYTES : (E == 2 ? w16::Tile<2>::BYTES : w16::Tile<3>::BYTES);
  return (size_t)w16::WROWS * w16::WF * 4 + tile + 16 * 64 * 16 + (4 * w16::NW + w16::WTA) * 4;
}

bool mp_win16_supported(int E, int K) { return E >= 1 && E <= 3 && K >= 1 && K <= 16; }

// launch of the 16-wave forward on the images mp_win_fwd has prepared (same fragments, same flag word)
int mp_win16_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, int residual, const float* h,
                    const int32_t* nlist, const float* e, const float* inv_degree, const float* Wfrag, const float* Wf32,
                    const unsigned* wflag, RangeGuard guard, float* h_out, float* s_save) {
  using namespace w16;
  Args a{};
  a.N = N; a.K = K; a.ntiles = cdiv(N, WTA);
  // contiguous runs of tiles per workgroup: multiples of 4 (256 atoms) when the batch is large enough (ng_internal.h)
  const int64_t per = win16_tiles_per_wg(a.ntiles, ctx->num_cu);
  a.tiles_per_wg = (int)per;
  a.h = h; a.nlist = nlist; a.e = e; a.Wfrag = Wfrag; a.Wfrag32 = Wf32; a.rowscale = inv_degree; a.residual = residual;
  a.out = h_out; a.S_save = s_save; a.act = act; a.dummy = const_cast<float*>(Wfrag) + (size_t)E * WF * WF;
  a.guard = guard; a.wflag = wflag; a.wflag_ver = pack_flag_version(ctx);
  const int grid = (int)cdiv(a.ntiles, per);
  const size_t lds = mp_win16_lds_bytes(E);
#ifdef W16_STAMP
  static unsigned long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { (void)hipMalloc(&dbg, 4 * 16 * 8 * 8); (void)hipMemset(dbg, 0, 4 * 16 * 8 * 8); }
  a.stamps = dbg;
#endif
  ProfScope ps(ctx, st, "mp_win_fwd");
  switch (E) {
    case 1: hipLaunchKernelGGL((mp_win16_fwd_kernel<1>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 2: hipLaunchKernelGGL((mp_win16_fwd_kernel<2>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 3: hipLaunchKernelGGL((mp_win16_fwd_kernel<3>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
  }
  NG_HIP(ctx, hipGetLastError());
#ifdef W16_STAMP
  if (++calls == 40) {
    unsigned long long hbuf[4 * 16 * 8];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hbuf, dbg, sizeof(hbuf), hipMemcpyDeviceToHost);
    for (int tt = 0; tt < 4; ++tt)
      for (int w = 0; w < 16; w += (tt == 0 ? 1 : 4)) {
        const unsigned long long* p = hbuf + (tt * 16 + w) * 8;
        fprintf(stderr, "W16 tile %d wave %2d: loads %5lld  gather %5lld  bar1 %5lld  mfma %5lld  bar1b %5lld  epilogue %5lld  end-bar %5lld | total %6lld\n",
                tt, w, (long long)(p[1] - p[0]), (long long)(p[2] - p[1]), (long long)(p[3] - p[2]), (long long)(p[4] - p[3]),
                (long long)(p[5] - p[4]), (long long)(p[6] - p[5]), (long long)(p[7] - p[6]), (long long)(p[7] - p[0]));
      }
  }
#endif
  return NG_OK;
}

}  // namespace ng
