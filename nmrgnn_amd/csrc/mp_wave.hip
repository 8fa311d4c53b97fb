// Wave-autonomous forward window kernel of the MPLayer (atom_feature_size 64, E == 3, padded lists with K in {4, 8, 12, 16}).
// Reference: nmrgnn/layers.py:26-46 (MPLayer.call) + residual of nmrgnn/model.py:165-167.
//
// Why (round 6): cycle stamps of the sixteen-wave kernel (mp_win16.hip) show its tile bound by VALU ISSUE — per SIMD the
// four waves' gathers run one after the other (1.9 k cycles for the oldest wave, 5 k until the youngest is through), and of a
// wave's ~1250 gather cycles only 384 are the FMAs: in the rotation layout a lane owns 4 features of an atom, so every
// 16-byte row read is followed by 6 packed FMAs and preceded by four DPP moves, an address and the read itself.  Here
//   * a wave owns 16 atoms from the lists to the stores, no workgroup barrier in between: lane (kg, atom) owns 16 features of
//     its atom — exactly the 8 + 8 contraction indices the B operand of v_mfma_f32_16x16x32_f16 wants from lane (kg, n) in
//     the two 32-wide k-steps of an edge feature — so per list entry it reads four 16-byte pieces of the source row and
//     issues 48 FMAs, and the sums are split into the MFMA operand IN REGISTERS (the aggregate never touches LDS);
//   * the weight fragments (the same image as mp_win16.hip, pack_bodies.cuh: mpw_h2<0>) sit in LDS, 48 KB, read as the A operand;
//   * a wave's lists arrive by LDS-DMA in a private 4-KB strip laid out [16-byte piece][atom] (conflict-free reads, the four
//     lanes of an atom read the same record), requested while the previous micro-tile is in its matrix interval;
//   * the window holds 288 source rows with the 16-byte chunks of row R stored at chunk ^ (R & 15): the 16 lanes of a
//     ds_read_b128 group read 16 different rows at the same chunk, unswizzled they would all hit the same four banks;
//   * waves drift apart: one wave's FMAs run beside another's matrix interval and a third's stores.
// A workgroup walks its atoms in groups of 256 (16 micro-tiles, two per wave); the window of a group is placed around the
// group's own rows (molecule batches: the graph) without looking at the lists, and a micro-tile whose sources do not all
// lie inside takes them from memory instead (same sums, same order).  Only the window change is a workgroup event.
// Per atom the entries are added in list order and the k-steps in order: results agree with mp_win16.hip to rounding.
#include <algorithm>
#include <cstdio>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "edge_fused.h"   // NG_LDS_BARRIER
#include "h2_common.cuh"
#include "mp_win16_common.cuh"
#include "mp_wave_common.cuh"

namespace ng {
namespace wv {

constexpr int STRIP_BYTES = 4096;          // [1 KB neighbour indices][3 KB edge features], each [piece][atom] x 16 B
constexpr int LDS_BYTES = WIN_BYTES + WIMG_BYTES + NWV * STRIP_BYTES;

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
  unsigned wflag_ver;
#ifdef WV_STAMP
  unsigned long long* stamps;
#endif
};
#ifdef WV_STAMP
// [group 0..1][micro-tile 0..1][wave][slot]
#define WV_T(k) do { if (a.stamps && blockIdx.x == 3 && lane == 0) a.stamps[(((int)((g0 - A0) / GROUP) * 2 + i) * NWV + wave) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#define WV_G(k) do { if (a.stamps && blockIdx.x == 3 && lane == 0) a.stamps[4 * NWV * 8 + ((int)((g0 - A0) / GROUP) * NWV + wave) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
#define WV_GN(k) do { if (a.stamps && blockIdx.x == 3 && lane == 0 && g0 == A0) a.stamps[4 * NWV * 8 + (NWV + wave) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define WV_T(k) do {} while (0)
#define WV_G(k) do {} while (0)
#define WV_GN(k) do {} while (0)
#endif

// the lists of micro-tile row0 .. row0+15 into the wave's strip; rows past N and pieces past K read as zeros
__device__ __forceinline__ void lists_dma(const Args& a, char* strip, int64_t row0, int lane) {
  const int K = a.K, at = lane & 15, pp = lane >> 4;
  const int rows = (int)std::min<int64_t>(MT, a.N - row0);
  nlist_dma(a.nlist, K, a.N, strip, row0, lane);
  const dma_i4 re = dma_rsrc(a.e + row0 * K * E, (unsigned)(rows * K * E * 4));
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int p = 4 * t + pp;
    lds_dma16(re, strip + 1024 + t * 1024, 4 * p < 3 * K ? at * K * E * 4 + p * 16 : VOFF_NONE, 0);
  }
}

// sums of one micro-tile: acc[n][j], j = 8 u + t  <->  feature 32 u + 8 kg + t.  GLOBAL: source rows from memory.
template <bool GLOBAL>
__device__ __forceinline__ void gather(const char* __restrict__ strip, const char* __restrict__ win, const float* __restrict__ h,
                                       int nq, int at, int kg, int wlo, float (&acc)[E][16]) {
  const int kc0 = (2 * kg) << 4;
  const char* rec = strip + (at << 4);
  i32x4 idx4 = *reinterpret_cast<const i32x4*>(rec);
  float4 ep[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) ep[i] = *reinterpret_cast<const float4*>(rec + 1024 + (i << 8));
#pragma unroll 1
  for (int q = 0; q < nq; ++q) {
    const i32x4 ci = idx4;
    const float ef[12] = {ep[0].x, ep[0].y, ep[0].z, ep[0].w, ep[1].x, ep[1].y, ep[1].z, ep[1].w, ep[2].x, ep[2].y, ep[2].z, ep[2].w};
    // the next quad's records, requested now and pinned at the END of the iteration (left alone the compiler sinks the reads to
    // the top of the next iteration, in front of their first use; pinned right here it waits for them at once)
    const int qn = q + 1 < nq ? q + 1 : q;
    i32x4 nidx = *reinterpret_cast<const i32x4*>(rec + (qn << 8));
    float4 nep[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) nep[i] = *reinterpret_cast<const float4*>(rec + 1024 + ((3 * qn + i) << 8));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float4 hr[4];
      if (!GLOBAL) {
#ifdef WV_ABL_SAMEROW      // timing experiment (wrong results): the sixteen lanes of a read group take sixteen consecutive rows — no bank conflicts
        const int R = (at + 16 * s + (ci[s] & 0)) & 255;
#else
        const int R = min(max(ci[s] - wlo, 0), WROWS - 1);
#endif
        const int sw = (R & 15) << 4;
        const char* row = win + (R << 8);
        hr[0] = *reinterpret_cast<const float4*>(row + (kc0 ^ sw));
        hr[1] = *reinterpret_cast<const float4*>(row + ((kc0 ^ 16) ^ sw));
        hr[2] = *reinterpret_cast<const float4*>(row + ((kc0 ^ 128) ^ sw));
        hr[3] = *reinterpret_cast<const float4*>(row + ((kc0 ^ 144) ^ sw));
      } else {
        const float4* row = reinterpret_cast<const float4*>(h + (int64_t)ci[s] * WF);
        hr[0] = row[2 * kg]; hr[1] = row[2 * kg + 1]; hr[2] = row[8 + 2 * kg]; hr[3] = row[9 + 2 * kg];
      }
#pragma unroll
      for (int n = 0; n < E; ++n) {
        const float w = ef[3 * s + n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[n][4 * r + 0] = __builtin_fmaf(w, hr[r].x, acc[n][4 * r + 0]);
          acc[n][4 * r + 1] = __builtin_fmaf(w, hr[r].y, acc[n][4 * r + 1]);
          acc[n][4 * r + 2] = __builtin_fmaf(w, hr[r].z, acc[n][4 * r + 2]);
          acc[n][4 * r + 3] = __builtin_fmaf(w, hr[r].w, acc[n][4 * r + 3]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(nidx[i]));
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(nep[i].x), "+v"(nep[i].y), "+v"(nep[i].z), "+v"(nep[i].w));
    idx4 = nidx;
#pragma unroll
    for (int i = 0; i < 3; ++i) ep[i] = nep[i];
  }
}

template <bool H2>
__device__ __forceinline__ void body(const Args& a) {
  extern __shared__ __attribute__((aligned(16))) char smem_wv[];
  char* win = smem_wv;
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
#ifdef WV_STAMP
  if (a.stamps && tid == 0) a.stamps[1024 + blockIdx.x] = wall_clock64();
#endif

  if (H2) wimg_dma(wimg, a.Wfrag, wave, lane);
  int64_t have = A0 + (int64_t)wave * MT;          // the micro-tile whose lists the strip holds (or is receiving)
  if (have < A1) lists_dma(a, strip, have, lane);

  // matrix interval + epilogue of a gathered micro-tile: sums -> piece operands in place -> out^T over the six k-steps ->
  // activation, residual, stores.  `re` = the residual rows if they were taken out of the window already (FROMWIN)
  float acc[E][16];
  f32x4 rew[4];
  auto finish = [&](int64_t row0, bool fromwin, bool wait_lists, int g_i, int64_t g0) __attribute__((always_inline)) {
    const int i = g_i;
    (void)i; (void)g0;
    const int64_t row = row0 + at;
    const bool live = row < a.N;
    const int64_t rowc = live ? row : a.N - 1;
    const float rs = a.rowscale[rowc];
    f32x4 o[4];
    float rsx;
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
      u32x4 xh[NT2], xl[NT2];
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
      WV_T(3);
      // matrix interval: weights as the A operand out of LDS; per k-step the products of the four column tiles interleaved
      // (consecutive MFMAs on different accumulators)
      f32x4 acc0[4], acc1[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) { acc0[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      const char* wl_base = wimg + (lane << 4);
#pragma unroll
      for (int T = 0; T < NT2; ++T) {
        u32x4 wh[4], wl[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          wh[ct] = *reinterpret_cast<const u32x4*>(wl_base + ((ct * NT2 + T) * 2) * 1024);
          wl[ct] = *reinterpret_cast<const u32x4*>(wl_base + ((ct * NT2 + T) * 2 + 1) * 1024);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc0[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[ct]), __builtin_bit_cast(f16x8, xh[T]), acc0[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[ct]), __builtin_bit_cast(f16x8, xh[T]), acc1[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc0[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[ct]), __builtin_bit_cast(f16x8, xl[T]), acc0[ct], 0, 0, 0);
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) o[ct] = acc0[ct] + acc1[ct];
      rsx = rs * (1.0f / 256.0f) * rsv;
    } else {
      // weights beyond the fp16 piece range: f32-input MFMA, one contraction index of the lane per instruction, the weight
      // from the fp32 fragment image (correct, not fast; never run in practice)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n = 0; n < E; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int k = 64 * n + 32 * (j >> 3) + 8 * kg + (j & 7);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const float wv = a.Wfrag32[((ct * (E * WF / 16) + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + at) * 4 + (k & 3)];
            o[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, acc[n][j], o[ct], 0, 0, 0);
          }
        }
      rsx = rs;
    }
    WV_T(4);
    // epilogue: lane (atom, kg) holds out[atom][16 ct + 4 kg .. + 3]
    f32x4 re[4];
    if (a.residual && !fromwin) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) re[ct] = *reinterpret_cast<const f32x4*>(a.h + rowc * WF + 16 * ct + 4 * kg);
    } else {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) re[ct] = rew[ct];
    }
    float4 v[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      v[ct] = make_float4(o[ct][0] * rsx, o[ct][1] * rsx, o[ct][2] * rsx, o[ct][3] * rsx);
      if (a.act == NG_ACT_SOFTPLUS) {
        v[ct].x = softplus_f(v[ct].x); v[ct].y = softplus_f(v[ct].y); v[ct].z = softplus_f(v[ct].z); v[ct].w = softplus_f(v[ct].w);
      } else if (a.act != NG_ACT_NONE) {
        v[ct].x = act_apply(a.act, v[ct].x); v[ct].y = act_apply(a.act, v[ct].y);
        v[ct].z = act_apply(a.act, v[ct].z); v[ct].w = act_apply(a.act, v[ct].w);
      }
    }
    WV_T(5);
    // the prefetched lists have landed long ago; waiting for them HERE (in front of the stores) keeps the stores' own
    // completion out of the next micro-tile's first wait.  (Not for the micro-tile finished beside a window in flight.)
    if (wait_lists) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WV_T(6);
    if (live) {
      typedef float nt4 __attribute__((ext_vector_type(4)));
      float* po = a.out + row * WF + 4 * kg;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        *reinterpret_cast<float4*>(po + 16 * ct) = make_float4(v[ct].x + resf * re[ct][0], v[ct].y + resf * re[ct][1],
                                                               v[ct].z + resf * re[ct][2], v[ct].w + resf * re[ct][3]);
      if (a.S_save) {
        // the activation copy is read a millisecond later by the backward: past the caches
        float* ps = a.S_save + row * WF + 4 * kg;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          __builtin_nontemporal_store(nt4{v[ct].x, v[ct].y, v[ct].z, v[ct].w}, reinterpret_cast<nt4*>(ps + 16 * ct));
      }
    }
    WV_T(7);
  };

#pragma unroll 1
  for (int64_t g0 = A0; g0 < A1; g0 += GROUP) {
    // ---- workgroup event: the group's window
    const int64_t wlo64 = std::max<int64_t>(0, std::min<int64_t>(g0 - (WROWS - GROUP) / 2, a.N - WROWS));
    const int wlo = (int)wlo64;      // (window kernels are dispatched for N < 2^31)
    const bool more = g0 + GROUP < A1;
    if (g0 == A0) {
      WV_G(0); WV_G(1);
      win_dma(win, a.h, wlo64, a.N, wave, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WV_G(2);
      NG_LDS_BARRIER();
      WV_G(3);
    }
    int64_t drow0 = -1;              // the micro-tile finished behind the group's barrier
    bool dwin = false;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) rew[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
      const int64_t row0 = g0 + (int64_t)(wave + NWV * i) * MT;
      if (row0 >= A1) break;
#ifndef WV_NOPRIO
      // the two waves of a SIMD: the older one wins the issue arbitration; in its second micro-tile the younger one is given
      // priority, so that both reach the group's barrier together
      if (i == 1 && wave >= NWV / 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
#endif
      if (have != row0) {            // (not reached with the prefetch below; kept so that the strip is right by construction)
        lists_dma(a, strip, row0, lane);
        have = row0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      WV_T(0);
      // ---- do the micro-tile's sources lie in the window?
      const bool inwin = sources_in_window(strip, lane, nq, row0, a.N, wlo);
#pragma unroll
      for (int n = 0; n < E; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;
      WV_T(1);
      if (inwin) gather<false>(strip, win, a.h, nq, at, kg, wlo, acc);
      else gather<true>(strip, win, a.h, nq, at, kg, wlo, acc);
      WV_T(2);
      // ---- the strip is free: the lists of this wave's next micro-tile travel beside the matrix interval
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {
        const int64_t nx = row0 + (int64_t)NWV * MT;
        if (nx < A1) { lists_dma(a, strip, nx, lane); have = nx; }
      }
      // the residual rows come out of the window when the micro-tile's own rows are inside (they are, for a window placed
      // around the group)
      const bool ownwin = a.residual && row0 >= wlo && row0 + MT <= (int64_t)wlo + WROWS;
      if (ownwin) {
        const int R = (int)(row0 - wlo) + at;
        const char* rp = win + (R << 8);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) rew[ct] = *reinterpret_cast<const f32x4*>(rp + (((4 * ct + kg) ^ (R & 15)) << 4));
      }
      const bool last_here = i == 1 || row0 + (int64_t)NWV * MT >= std::min<int64_t>(g0 + GROUP, A1);
      if (more && last_here) {       // finished beside the next group's window load
        drow0 = row0;
        dwin = ownwin;
        break;
      }
      finish(row0, ownwin, true, i, g0);
    }
    if (more) {
      // every wave is through with this window; the next one is requested, and the last micro-tile's matrix interval,
      // activation and stores run while it travels
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WV_GN(0);
      NG_LDS_BARRIER();
      WV_GN(1);
      const int64_t g1 = g0 + GROUP;
      const int64_t nlo = std::max<int64_t>(0, std::min<int64_t>(g1 - (WROWS - GROUP) / 2, a.N - WROWS));
      win_dma(win, a.h, nlo, a.N, wave, lane);
      if (drow0 >= 0) {
        const int i = 1;
        (void)i;
        finish(drow0, dwin, false, 1, g0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WV_GN(2);
      NG_LDS_BARRIER();
      WV_GN(3);
    }
  }
#ifdef WV_STAMP
  if (a.stamps && lane == 0) a.stamps[2048 + blockIdx.x * NWV + wave] = wall_clock64();
#endif
}

__global__ __launch_bounds__(WTHREADS) void mp_wave_fwd_kernel(Args a) {
  if (a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag))) body<false>(a);
  else body<true>(a);
}

}  // namespace wv

bool mp_wave_supported(int E, int K) { return E == 3 && K >= 4 && K <= 16 && K % 4 == 0; }

// NG_MP_WAVE=1: every supported call; =0: none; default: batches of small graphs (ng_ctx_set_graph_span: every graph fits the
// window a group places around its own rows — a protein's neighbours lie all over its frame, its micro-tiles would take their
// sources from memory and mp_win16.hip's per-tile window choice serves it better) that fill the chip (half a group per CU and more)
bool mp_wave_wanted(const ng_ctx* ctx, int64_t N, int E, int K) {
  if (!mp_wave_supported(E, K) || N >= (int64_t(1) << 31)) return false;
  if (sw().mp_wave == 0) return false;
  if (sw().mp_wave == 1) return true;
  return N >= (int64_t)ctx->num_cu * (wv::GROUP / 2) && ctx->graph_span > 0 && ctx->graph_span <= wv::WROWS - (wv::WROWS - wv::GROUP) / 2;
}

// launch on the images mp_win_fwd has prepared (same fragments, same flag word as mp_win16.hip)
int mp_wave_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int act, int residual, const float* h, const int32_t* nlist,
                   const float* e, const float* inv_degree, const float* Wfrag, const float* Wf32, const unsigned* wflag,
                   RangeGuard guard, float* h_out, float* s_save) {
  using namespace wv;
  Args a{};
  a.N = N; a.K = K;
  // contiguous runs per workgroup: whole groups when the batch is large enough, 64-atom steps below (ng_internal.h)
  a.atoms_per_wg = win16_tiles_per_wg(cdiv(N, 64), ctx->num_cu) * 64;
  a.h = h; a.nlist = nlist; a.e = e; a.Wfrag = Wfrag; a.Wfrag32 = Wf32; a.rowscale = inv_degree; a.residual = residual;
  a.out = h_out; a.S_save = s_save; a.act = act;
  a.guard = guard; a.wflag = wflag; a.wflag_ver = pack_flag_version(ctx);
  const int grid = (int)cdiv(N, a.atoms_per_wg);
#ifdef WV_STAMP
  static unsigned long long* dbg = nullptr;
  static int calls = 0;
  constexpr int NST = 4 * NWV * 8 + 2 * NWV * 4;
  if (!dbg) { (void)hipMalloc(&dbg, 8192 * 8); (void)hipMemset(dbg, 0, 8192 * 8); }
  a.stamps = dbg;
#endif
  ProfScope ps(ctx, st, "mp_win_fwd");
  hipLaunchKernelGGL(mp_wave_fwd_kernel, dim3(grid), dim3(WTHREADS), LDS_BYTES, st, a);
  NG_HIP(ctx, hipGetLastError());
#ifdef WV_STAMP
  if (++calls == 40) {
    {
      static unsigned long long wb[8192];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(wb, dbg, sizeof(wb), hipMemcpyDeviceToHost);
      unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
      for (int b = 0; b < grid && b < 512; ++b) {
        s0 = std::min(s0, wb[1024 + b]); s1 = std::max(s1, wb[1024 + b]);
        for (int w = 0; w < NWV; ++w) { e0 = std::min(e0, wb[2048 + b * NWV + w]); e1 = std::max(e1, wb[2048 + b * NWV + w]); }
      }
      fprintf(stderr, "WV wall (100 MHz ticks): first start 0, last start %llu, first end %llu, last end %llu; block 3: start %llu end %llu\n",
              s1 - s0, e0 - s0, e1 - s0, wb[1024 + 3] - s0, wb[2048 + 3 * NWV] - s0);
      int hist[16] = {0};
      for (int b = 0; b < grid && b < 512; ++b) { unsigned long long m = 0; for (int w = 0; w < NWV; ++w) m = std::max(m, wb[2048 + b * NWV + w]); const int k = (int)((m - s0) / 500); hist[k < 15 ? k : 15]++; }
      fprintf(stderr, "WV end-time histogram (5 us bins):");
      for (int k = 0; k < 16; ++k) fprintf(stderr, " %d", hist[k]);
      fprintf(stderr, "\n");
    }
    unsigned long long hb[NST];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hb, dbg, sizeof(hb), hipMemcpyDeviceToHost);
    const unsigned long long t0 = hb[4 * NWV * 8 + 0];
    for (int g = 0; g < 2; ++g)
      for (int w = 0; w < NWV; ++w) {
        const unsigned long long* q = hb + 4 * NWV * 8 + (g * NWV + w) * 4;
        fprintf(stderr, "WV group %d wave %d: arrive %6lld  bar %5lld  window %5lld  bar %5lld\n", g, w, (long long)(q[0] - t0),
                (long long)(q[1] - q[0]), (long long)(q[2] - q[1]), (long long)(q[3] - q[2]));
        for (int i = 0; i < 2; ++i) {
          const unsigned long long* p = hb + ((g * 2 + i) * NWV + w) * 8;
          fprintf(stderr, "   mt %d: start %6lld  range %4lld  gather %5lld  dma+scale %4lld  split %4lld  mfma %5lld  act %5lld  wait %4lld  stores %4lld | %6lld\n",
                  i, (long long)(p[0] - t0), (long long)(p[1] - p[0]), (long long)(p[2] - p[1]), (long long)(p[3] - p[2]), 0LL,
                  (long long)(p[4] - p[3]), (long long)(p[5] - p[4]), (long long)(p[6] - p[5]), (long long)(p[7] - p[6]), (long long)(p[7] - p[0]));
        }
      }
  }
#endif
  return NG_OK;
}

}  // namespace ng
