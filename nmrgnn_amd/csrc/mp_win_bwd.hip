// Window-resident fused MPLayer BACKWARD kernels (atom_feature_size == 64, edge_feature_size <= 3,
// K % 4 == 0, K <= 16); companions of mp_win.hip.  Math: SURVEY App. B.
//
//   dP  = dH' * act'(S) * v                                  (activation gradient, inv_degree folded in)
//   edge kernel:  dA[i,(n,l)] = sum_m dP[i,m] W[l,m,n]  (MFMA, stays in LDS)
//                 de[i,j,n]   = sum_l dA[i,(n,l)] h[nlist[i,j], l]        (window gather + dot)
//   node kernel:  B[t,(n,m)]  = sum_{p -> t} e[p,n] dP[src(p), m]         (incoming edges, window gather)
//                 dh[t,l]     = dH'[t,l] + sum_{n,m} B[t,(n,m)] W[l,m,n]  (MFMA)
//                 dw[l,m,n]  += sum_t h[t,l] B[t,(n,m)]                   (MFMA, accumulators in registers)
// Neither dA nor B ever reaches HBM (the split path writes and re-reads both, 768 B per atom each).
//
// Same skeleton as the forward kernel: one persistent 512-thread workgroup per CU walks a contiguous run
// of 32-atom tiles, keeps a 288-row window of the gathered tensor in LDS (restaged when the tile's lists
// leave it, global-memory gather when the range is wider than the window), prefetches the next tile's
// inputs through registers one tile ahead, and alternates a VALU interval and an MFMA interval (fp32
// MFMA and VALU do not overlap on a SIMD: tools/ubench).  The gathers use the rotation scheme of
// mp_win.hip: lane c of an atom's 16-lane row owns list slot c and walks the others over row_ror DPP.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "edge_fused.h"   // NG_LDS_BARRIER
#include "reduce.cuh"
#include "h2_common.cuh"
#include "pack_bodies.cuh"

namespace ng {

namespace {

constexpr int WF = 64;
constexpr int WTA = 32;
constexpr int WROWS = 288;
constexpr int WC4 = WF / 4;
constexpr int WTHREADS = 512;
constexpr int SDP_LD = 68;      // dP / h tile row stride (== 4 mod 16: conflict-free K-strided reads)
constexpr int SDP_SLOT = 2304;  // floats per dP tile slot of the edge kernel: [32][68] fp32, or two fp16 piece planes [32][72]

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short gs16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wave_min_i32(int v) {
  const int big = 0x7fffffff;
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x111, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x112, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x114, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x118, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x142, 0xa, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x143, 0xc, 0xf, false));
  return v;
}

template <int S>
__device__ __forceinline__ int ror_i(int v) {
  if (S == 0) return v;
  return __builtin_amdgcn_update_dpp(0, v, 0x120 + (S & 15), 0xf, 0xf, false);
}
template <int S>
__device__ __forceinline__ float ror_f(float v) {
  return __builtin_bit_cast(float, ror_i<S>(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ bool win_decide(const int* __restrict__ ctl, int& wlo, int& mode) {
  int lo = ctl[0], hi = ctl[8];
#pragma unroll
  for (int i = 1; i < 8; ++i) { lo = min(lo, ctl[i]); hi = max(hi, ctl[8 + i]); }
  mode = 0;
  if (hi < lo) return false;
  if (lo >= wlo && hi < wlo + WROWS) return false;
  if (hi - lo + 1 > WROWS) { mode = 1; return false; }
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

// The window by LDS-DMA (round 4): its 288 rows are one contiguous 72-KB block of the source array — 72 wave-instructions
// of 1 KB, nine per wave, straight into LDS with no registers in between, so the request can be made as soon as no wave
// reads the old window any more and awaited (s_waitcnt vmcnt(0)) in front of the barrier before the next gather: the
// HBM / L2 round trip runs beside the matrix interval.  Rows past the end read as zeros (buffer bounds).  32-bit byte
// offsets: up to 16.7 M rows, beyond that the register staging above.
__device__ __forceinline__ void win_dma(float* __restrict__ win, dma_i4 rs, int wlo, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < WROWS * WF * 4 / 1024 / 8; ++j) {
    const int kb = wave + 8 * j;
    lds_dma16(rs, reinterpret_cast<char*>(win) + kb * 1024, lane * 16, wlo * (WF * 4) + kb * 1024);
  }
}
static_assert(WROWS * WF * 4 == 72 * 1024 && WTHREADS == 512, "win_dma: nine 1-KB instructions for each of eight waves");

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// ---- edge kernel ---------------------------------------------------------------------------------------
struct MpWinEdgeArgs {
  int64_t N;
  int K;
  int64_t ntiles;
  int tiles_per_wg;
  const float* dH;         // [N][64] upstream gradient of the layer output
  const float* S;          // [N][64] saved activation output, or nullptr (linear)
  const float* rowscale;   // [N]
  const float* h;          // [N][64] layer input (gathered)
  const int32_t* nlist;    // [N][K]
  const float* WfragT;     // mpw_pack mode 2
  float* dP;               // [N][64] out
  float* de;               // [N*K][E] out (+= when accumulate)
  float* dummy;            // >= 64 floats
  int act;
  int accumulate;
  RangeGuard guard;        // word == nullptr: unguarded
  const float* WfragT32;   // fp32 fragments (mpw_pack mode 2) for a guarded call whose weights leave the piece range
  const unsigned* wflag;   // flag word of an image kept over calls (== wflag_ver: its weights left the piece range), or nullptr
  unsigned wflag_ver;
};

// one rotation step of the edge-gradient dot: this lane's chunk of dA[i][n][:] against the row of the slot
// that the rotation brings here; the partial lands in the accumulator of THAT slot's lane afterwards
template <int E, int S, int MODE>
__device__ __forceinline__ void edge_step(const char* __restrict__ wbytes, const float4* __restrict__ src4,
                                          int c, int roff, int gidx, const float4 (&da)[E], float (&out)[E]) {
  float4 hrow;
  if (MODE == 0) hrow = *reinterpret_cast<const float4*>(wbytes + ror_i<S>(roff));
  else hrow = src4[(int64_t)ror_i<S>(gidx) * WC4 + c];
#pragma unroll
  for (int n = 0; n < E; ++n) {
    const float p = dot4(da[n], hrow);
    // slot (c + S) was processed here; rotate the partial back to its owner: out_j = sum_S ror_{16-S}(p_S)
    out[n] += ror_f<(16 - S) & 15>(p);
  }
}

template <int E, int MODE>
__device__ __forceinline__ void edge_dot(int K, int wave, int lane, int wlo, const int32_t* __restrict__ nl,
                                         const float* __restrict__ tb, int ld, const float4* __restrict__ win4,
                                         const float4* __restrict__ src4, float (&out)[E]) {
  const int c = lane & 15;
  const int al = wave * 4 + (lane >> 4);
  const int idx = nl[al * K + (c < K ? c : 0)];
  const int roff = min(max(idx - wlo, 0), WROWS - 1) * (WF * 4);
  const char* wbytes = reinterpret_cast<const char*>(win4) + 16 * c;
  float4 da[E];
#pragma unroll
  for (int n = 0; n < E; ++n) {
    da[n] = *reinterpret_cast<const float4*>(tb + al * ld + n * WF + 4 * c);
    out[n] = 0.f;
  }
  edge_step<E, 0, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 1, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 2, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 3, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 4, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 5, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 6, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 7, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 8, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 9, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 10, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 11, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 12, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 13, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 14, MODE>(wbytes, src4, c, roff, idx, da, out);
  edge_step<E, 15, MODE>(wbytes, src4, c, roff, idx, da, out);
}

// kept out of line: inlined next to the window variant its global loads make the compiler put vmcnt
// waits into the window path (see mp_win.hip)
template <int E>
struct EdgeDots { float v[E]; };      // returned in registers: an array passed out by pointer lived on the stack

template <int E>
__device__ __noinline__ EdgeDots<E> edge_dot_global(int K, int wave, int lane, const int32_t* nl, const float* tb,
                                                    int ld, const float4* src4) {
  float out[E];
  edge_dot<E, 1>(K, wave, lane, 0, nl, tb, ld, nullptr, src4, out);
  EdgeDots<E> r;
#pragma unroll
  for (int n = 0; n < E; ++n) r.v[n] = out[n];
  return r;
}

// H2: the dA product on the fp16 pipe with two-piece operands (h2_common.cuh).  dP is a gradient of arbitrary
// magnitude, and dA = dP Wp^T is independent per atom row: every row gets its OWN power-of-two scale, taken from the
// row's max |dP| where the row is formed (commit: 16 lanes of a DPP row hold it) and kept beside the row in LDS
// (columns 64 / 65 of the padded dP tile: S and 1/S); the lane that multiplies a row splits S * dP in registers and
// multiplies its outputs by 2^-8 / S.  Scaling dH by 2^k moves every S by 2^-k: bit-identical scaled results.
template <int E, bool H2>
__device__ __forceinline__ void mp_win_bwd_edge_body(const MpWinEdgeArgs& a) {
  constexpr int KF = E * WF;
  constexpr int LD = KF + 4;
  constexpr int NCT = KF / 16 / 4;          // column tiles per wave (4 waves per atom half)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* win = smem;                                                   // [WROWS][64]
  float* tile = win + WROWS * WF;                                      // [32][LD]   dA
  float* sdp = tile + WTA * LD;                                        // [2][32][SDP_LD]   (piece form: [2][2 planes][32][72] fp16)
  int32_t* s_nl = reinterpret_cast<int32_t*>(sdp + 2 * SDP_SLOT);      // [2][32*K]
  int* ctl = reinterpret_cast<int*>(s_nl + 2 * WTA * a.K);             // [2][16]
  float* s_inv = reinterpret_cast<float*>(ctl + 32);                   // [2][32]  2^-8 / S per dP row (piece form)
  constexpr int PROWB = (WF + 8) * 2, PPLANE = WTA * PROWB;            // dP piece planes: 144 B per row
  static_assert(2 * PPLANE <= SDP_SLOT * 4 && WTA * SDP_LD <= SDP_SLOT, "one tile's dP slot holds either form");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;
  const int per_tile = WTA * K;
  const int64_t T0 = (int64_t)blockIdx.x * a.tiles_per_wg;
  const int64_t T1 = std::min<int64_t>(T0 + a.tiles_per_wg, a.ntiles);
  if (T0 >= T1) return;

  const float4* src4 = reinterpret_cast<const float4*>(a.h);
  float4* win4 = reinterpret_cast<float4*>(win);
  for (int t = tid; t < WROWS * WC4; t += WTHREADS) win4[t] = f4zero();
  const bool dma_ok = a.N * (int64_t)(WF * 4) < ((int64_t)1 << 32);
  const dma_i4 hrsrc = dma_rsrc(a.h, dma_ok ? (unsigned)(a.N * (WF * 4)) : 0u);

  // weight fragments: this wave's NCT column tiles of dA (o = 16*ct + ...), contraction over m (4 k-steps)
  const int hh = wave >> 2, ct0 = (wave & 3) * NCT;
  float wf[H2 ? 1 : NCT][16];
  u32x4 wh[H2 ? NCT : 1][2], wl[H2 ? NCT : 1][2];      // fp16 pieces of 2^8 Wp^T, two 32-wide k-steps per column tile
  if (H2) {
#pragma unroll
    for (int u = 0; u < NCT; ++u) {
      const u32x4* p = reinterpret_cast<const u32x4*>(a.WfragT) + (size_t)((ct0 + u) * 2) * 2 * 64 + lane;
#pragma unroll
      for (int Ts = 0; Ts < 2; ++Ts) { wh[u][Ts] = p[(2 * Ts) * 64]; wl[u][Ts] = p[(2 * Ts + 1) * 64]; }
#pragma unroll
      for (int Ts = 0; Ts < 2; ++Ts)
#pragma unroll
        for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(wh[u][Ts][j])); asm volatile("" : "+v"(wl[u][Ts][j])); }
    }
  } else {
#pragma unroll
    for (int u = 0; u < NCT; ++u) {
      const float4* p = reinterpret_cast<const float4*>(a.WfragT) + ((ct0 + u) * 4) * 64 + lane;
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        const float4 v = p[T * 64];
        wf[u][4 * T + 0] = v.x; wf[u][4 * T + 1] = v.y; wf[u][4 * T + 2] = v.z; wf[u][4 * T + 3] = v.w;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(wf[u][i]));
    }
  }

  // per-tile inputs in flight: neighbour indices (threads < 8K), this thread's float4 of dH and S
  // (row = tid >> 4, column chunk = tid & 15), the row's inv_degree, the old de of its (atom, slot)
  int4 p_nl;
  float4 p_dh, p_s;
  float p_rs;
  float p_de[E];
  const int prow = tid >> 4, pc = tid & 15;
  auto issue = [&](int64_t t) {
    const int64_t nb = t * per_tile / 4, nlim = a.N * K / 4;
    const int64_t qn = nb + tid;
    const int4 v = reinterpret_cast<const int4*>(a.nlist)[qn < nlim ? qn : nlim - 1];
    p_nl = (qn < nlim && tid < per_tile / 4) ? v : make_int4(0, 0, 0, 0);
    const int64_t row = t * WTA + prow;
    const int64_t rc = row < a.N ? row : a.N - 1;
    p_dh = *reinterpret_cast<const float4*>(a.dH + rc * WF + 4 * pc);
    p_s = a.S ? *reinterpret_cast<const float4*>(a.S + rc * WF + 4 * pc) : f4zero();
    p_rs = a.rowscale[rc];
    if (row >= a.N) p_dh = f4zero();
  };
  auto issue_de = [&](int64_t t) {       // old edge gradient of (atom tid >> 4, slot tid & 15)
    const int64_t row = t * WTA + prow;
    const int64_t rc = row < a.N ? row : a.N - 1;
    const int sc = pc < K ? pc : 0;
#pragma unroll
    for (int n = 0; n < E; ++n) p_de[n] = a.accumulate ? a.de[(rc * K + sc) * E + n] : 0.f;
  };
  auto commit = [&](int64_t t) {
    int32_t* nl = s_nl + (t & 1) * per_tile;
    float* dp = sdp + (t & 1) * SDP_SLOT;
    int lo = 0x7fffffff, hi = -1;
    if (tid < per_tile / 4) {
      reinterpret_cast<int4*>(nl)[tid] = p_nl;
      lo = min(min(p_nl.x, p_nl.y), min(p_nl.z, p_nl.w));
      hi = max(max(p_nl.x, p_nl.y), max(p_nl.z, p_nl.w));
    }
    float4 g = p_dh;
    if (a.act != NG_ACT_NONE) {
      g.x *= act_grad_from_out(a.act, p_s.x); g.y *= act_grad_from_out(a.act, p_s.y);
      g.z *= act_grad_from_out(a.act, p_s.z); g.w *= act_grad_from_out(a.act, p_s.w);
    }
    g.x *= p_rs; g.y *= p_rs; g.z *= p_rs; g.w *= p_rs;
    if (H2) {
      // row max over the 16 lanes of this DPP row -> S = 2^(14 - e), 2^e > max (biased exponent arithmetic; an all-zero
      // or non-finite row keeps S = 1).  The row is split HERE, once, into the two fp16 planes the matrix interval reads
      // (round 4; until then every one of the four waves that multiply a row split it again from an fp32 copy: 50 of a
      // wave's ~500 instructions per tile)
      float m = fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w)));
      m = fmaxf(m, ror_f<8>(m)); m = fmaxf(m, ror_f<4>(m)); m = fmaxf(m, ror_f<2>(m)); m = fmaxf(m, ror_f<1>(m));
      const int ef = (__builtin_bit_cast(int, m) >> 23) & 255;
      const int sb = (ef == 0 || ef == 255) ? 127 : min(267 - ef, 253);
      const float S = __builtin_bit_cast(float, sb << 23);
      if (pc == 0) s_inv[(t & 1) * WTA + prow] = __builtin_bit_cast(float, (254 - sb) << 23) * (1.0f / 256.0f);
      unsigned h0, l0, h1, l1;
      split2_pair(S * g.x, S * g.y, h0, l0); split2_pair(S * g.z, S * g.w, h1, l1);
      char* q = reinterpret_cast<char*>(dp) + prow * PROWB + 8 * pc;
      *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(q + PPLANE) = u32x2{l0, l1};
    } else {
      *reinterpret_cast<float4*>(dp + prow * SDP_LD + 4 * pc) = g;
    }
    const int64_t row = t * WTA + prow;
    *reinterpret_cast<float4*>(row < a.N ? a.dP + row * WF + 4 * pc : a.dummy + 4 * pc) = g;
    lo = wave_min_i32(lo);
    hi = -wave_min_i32(-hi);
    if (lane == 63) { ctl[(t & 1) * 16 + wave] = lo; ctl[(t & 1) * 16 + 8 + wave] = hi; }
  };

  int wlo = -(1 << 30), mode = 0;
  issue(T0);
  commit(T0);
  issue(T0 + 1 < T1 ? T0 + 1 : T0);
  issue_de(T0);
  NG_LDS_BARRIER();

  const int a16 = lane & 15, g4 = lane >> 4;
#pragma unroll 1
  for (int64_t t = T0; t < T1; ++t) {
    // (no wave reads the window between the last barrier and the one behind the matrix interval)
    const bool staged = win_decide(ctl + (t & 1) * 16, wlo, mode);
    if (staged) {
      if (dma_ok) win_dma(win, hrsrc, wlo, wave, lane);
      else win_stage(win4, src4, wlo, a.N, tid);
    }
    // ---- matrix interval: dA tile = dP tile x Wp^T
    {
      f32x4 acc[NCT];
#pragma unroll
      for (int u = 0; u < NCT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      float oscale = 1.0f;
      if (H2) {
        // B operand of step Ts: this lane's row (atom a16), k-slots 32 Ts + 8 g4 + (0..7), scaled and split here
        const char* xr = reinterpret_cast<const char*>(sdp + (t & 1) * SDP_SLOT) + (16 * hh + a16) * PROWB + 16 * g4;
        oscale = s_inv[(t & 1) * WTA + 16 * hh + a16];
        u32x4 xh[2], xl[2];
#pragma unroll
        for (int Ts = 0; Ts < 2; ++Ts) {
          xh[Ts] = *reinterpret_cast<const u32x4*>(xr + 64 * Ts);
          xl[Ts] = *reinterpret_cast<const u32x4*>(xr + 64 * Ts + PPLANE);
        }
#pragma unroll
        for (int Ts = 0; Ts < 2; ++Ts) {
#pragma unroll
          for (int u = 0; u < NCT; ++u)
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[u][Ts]), __builtin_bit_cast(f16x8, xh[Ts]), acc[u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < NCT; ++u)
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[u][Ts]), __builtin_bit_cast(f16x8, xl[Ts]), acc[u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < NCT; ++u)
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[u][Ts]), __builtin_bit_cast(f16x8, xh[Ts]), acc[u], 0, 0, 0);
        }
      } else {
      const float* xrow = sdp + (t & 1) * SDP_SLOT + (16 * hh + a16) * SDP_LD + 4 * g4;
      float4 x[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) x[T] = *reinterpret_cast<const float4*>(xrow + 16 * T);
#pragma unroll
      for (int T = 0; T < 4; ++T) {
#pragma unroll
        for (int u = 0; u < NCT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][4 * T + 0], x[T].x, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NCT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][4 * T + 1], x[T].y, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NCT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][4 * T + 2], x[T].z, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NCT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][4 * T + 3], x[T].w, acc[u], 0, 0, 0);
      }
      }
      // lane holds dA[atom 16hh + a16][o = 16(ct0+u) + 4*g4 + (0..3)]
#pragma unroll
      for (int u = 0; u < NCT; ++u)
        *reinterpret_cast<float4*>(tile + (16 * hh + a16) * LD + 16 * (ct0 + u) + 4 * g4) =
            make_float4(acc[u][0] * oscale, acc[u][1] * oscale, acc[u][2] * oscale, acc[u][3] * oscale);
    }
    if (staged && dma_ok) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NG_LDS_BARRIER();
    // ---- vector interval: de of tile t, then dP / lists of tile t+1 into LDS, requests for t+2
    {
      float out[E];
      const int32_t* nl = s_nl + (t & 1) * per_tile;
      if (mode == 0) edge_dot<E, 0>(K, wave, lane, wlo, nl, tile, LD, win4, src4, out);
      else {
        const EdgeDots<E> r = edge_dot_global<E>(K, wave, lane, nl, tile, LD, src4);
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
    issue(t + 2 < T1 ? t + 2 : t);
    issue_de(t + 1 < T1 ? t + 1 : t);
    NG_LDS_BARRIER();
  }
}

// Range guard (ng_internal.h).  The gradient operand of the piece form carries its own per-row scale, so the only thing
// that can leave the fp16 range here is a weight (|2^8 w| >= 65504) — known when the kernel starts: the pack launch in
// front raised the guard.  Both forms live in the one kernel and the launch picks, uniformly, at its first instruction:
// no second launch on the timeline (an empty one costs ~4.6 us on this part, profiles/r03b).
template <int E, bool H2>
__global__ __launch_bounds__(WTHREADS, 1) void mp_win_bwd_edge_kernel(MpWinEdgeArgs a) {
  if (H2 && a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag))) {
    a.WfragT = a.WfragT32;
    mp_win_bwd_edge_body<E, false>(a);
  } else {
    mp_win_bwd_edge_body<E, H2>(a);
  }
}

// ---- node kernel ---------------------------------------------------------------------------------------
// Incoming-edge records in CSC order, 16 bytes each: { source atom (int bits), e[p][0..2] } — built once
// per backward pass (the edge features are the same for every layer).
constexpr int NREC_CAP = 768;           // records staged per 32-atom tile (mean 16 * 32 = 512); 1024 until the fp16 piece planes of B needed the room

struct MpWinNodeArgs {
  int64_t N;
  int64_t ntiles;
  int tiles_per_wg;
  const float* dP;          // [N][64]  (gathered)
  const float* dH;          // [N][64]  residual term of dh
  const float* h;           // [N][64]  layer input (for dw)
  const int32_t* csc_ptr;   // [N+1]
  const float4* rec;        // [nnz] records
  const float* WfragN;      // mpw_pack mode 1
  float* dh;                // [N][64] out
  float* partial;           // [grid][64*E*64] dw partials, layout [(n,m)][l]
  float* dummy;
  RangeGuard guard;         // word == nullptr: unguarded
  const float* WfragN32;    // fp32 fragments (mpw_pack mode 1) for a guarded call whose weights leave the piece range
  const unsigned* wflag;    // as in MpWinEdgeArgs
  unsigned wflag_ver;
};

__device__ __forceinline__ void pk_axpy(f32x2& lo, f32x2& hi, float w, const float4& h) {
  const f32x2 ww = {w, w};
  lo = __builtin_elementwise_fma(ww, f32x2{h.x, h.y}, lo);
  hi = __builtin_elementwise_fma(ww, f32x2{h.z, h.w}, hi);
}

template <int E, int S0, int MODE>
__device__ __forceinline__ void node_steps4(const char* __restrict__ wbytes, const float4* __restrict__ src4,
                                            int c, int roff, int gidx, const float (&w)[E], f32x2 (&lo)[E],
                                            f32x2 (&hi)[E]) {
  float4 h0, h1, h2, h3;
  if (MODE == 0) {
    h0 = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 0>(roff));
    h1 = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 1>(roff));
    h2 = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 2>(roff));
    h3 = *reinterpret_cast<const float4*>(wbytes + ror_i<S0 + 3>(roff));
  } else {
    h0 = src4[(int64_t)ror_i<S0 + 0>(gidx) * WC4 + c];
    h1 = src4[(int64_t)ror_i<S0 + 1>(gidx) * WC4 + c];
    h2 = src4[(int64_t)ror_i<S0 + 2>(gidx) * WC4 + c];
    h3 = src4[(int64_t)ror_i<S0 + 3>(gidx) * WC4 + c];
  }
#pragma unroll
  for (int n = 0; n < E; ++n) pk_axpy(lo[n], hi[n], ror_f<S0 + 0>(w[n]), h0);
#pragma unroll
  for (int n = 0; n < E; ++n) pk_axpy(lo[n], hi[n], ror_f<S0 + 1>(w[n]), h1);
#pragma unroll
  for (int n = 0; n < E; ++n) pk_axpy(lo[n], hi[n], ror_f<S0 + 2>(w[n]), h2);
#pragma unroll
  for (int n = 0; n < E; ++n) pk_axpy(lo[n], hi[n], ror_f<S0 + 3>(w[n]), h3);
}

// B[t][(n,m)] for the tile's 32 target atoms: 16 lanes per atom, the atom's incoming records taken 16 at a
// time (lane c owns record 16*round + c), rotation walk as in the forward gather
// pl (H2 form of the dh product): besides the fp32 tile (the dw product contracts over atoms and needs one scale for
// all rows) the row goes into two fp16 piece planes [2][32][E*64 + 8], scaled by a power of two taken from the row's
// own max |B| (the 16 lanes of a DPP row hold the whole row); rs[atom] receives 2^-8 / S for the epilogue.
template <int E, int MODE>
__device__ __forceinline__ void node_gather(int wave, int lane, int wlo, const int* __restrict__ s_ptr,
                                            const float4* __restrict__ recs, int rec_base,
                                            float* __restrict__ tb, int ld, const float4* __restrict__ win4,
                                            const float4* __restrict__ src4, char* __restrict__ pl, float* __restrict__ rs,
                                            int* __restrict__ sbv, int* __restrict__ wmin) {
  const int c = lane & 15;
  const int al = wave * 4 + (lane >> 4);
  const int p0 = s_ptr[al], cnt = s_ptr[al + 1] - p0;
  // wave-uniform number of rounds: the largest record count among this wave's four atoms
  int mx = cnt;
  mx = max(mx, __builtin_amdgcn_update_dpp(0, mx, 0x142, 0xa, 0xf, false));   // row_bcast15 (rows 1,3)
  mx = max(mx, __builtin_amdgcn_update_dpp(0, mx, 0x143, 0xc, 0xf, false));   // row_bcast31 (rows 2,3)
  const int rounds = (__builtin_amdgcn_readlane(mx, 63) + 15) >> 4;
  const char* wbytes = reinterpret_cast<const char*>(win4) + 16 * c;
  f32x2 lo[E], hi[E];
#pragma unroll
  for (int n = 0; n < E; ++n) { lo[n] = f32x2{0.f, 0.f}; hi[n] = f32x2{0.f, 0.f}; }
  const int mxw = __builtin_amdgcn_readlane(mx, 63);
  (void)rounds;
  {   // the first 16 records of every atom: rotation walk (lane c owns record c, 16 steps)
    const int q = c;
    float4 rc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < cnt) rc = recs[p0 - rec_base + q];
    const int src = q < cnt ? __builtin_bit_cast(int, rc.x) : wlo;
    float w[E];
    w[0] = rc.y;
    if (E > 1) w[1] = rc.z;
    if (E > 2) w[2] = rc.w;
    const int roff = min(max(src - wlo, 0), WROWS - 1) * (WF * 4);
    node_steps4<E, 0, MODE>(wbytes, src4, c, roff, src, w, lo, hi);
    node_steps4<E, 4, MODE>(wbytes, src4, c, roff, src, w, lo, hi);
    node_steps4<E, 8, MODE>(wbytes, src4, c, roff, src, w, lo, hi);
    node_steps4<E, 12, MODE>(wbytes, src4, c, roff, src, w, lo, hi);
  }
  // Records beyond the sixteenth (round 4).  In-degrees scatter around K: with the bench's lists 83 % of the waves have an atom
  // with more than 16 incoming edges, and a second 16-step rotation round for those few records nearly doubled the gather.  The
  // tail is walked RECORD by record instead — all 16 lanes of an atom read the same record from the staged list (an LDS
  // broadcast) and their 16 bytes of its source row — for as many steps as the wave's largest in-degree needs (a few, not
  // 16).  Same summation order per atom as before (record 16, 17, ... after 0 .. 15), so the bits do not move.
  // (four records per trip, their reads issued together: one record per trip left the two dependent LDS latencies of a
  // step exposed — two waves per SIMD cannot cover them — and cost as much as the round it replaced)
#pragma unroll 1
  for (int q0 = 16; q0 < mxw; q0 += 4) {
    float4 rc[4];
    float4 h[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q0 + u < cnt) rc[u] = recs[p0 - rec_base + q0 + u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int src = q0 + u < cnt ? __builtin_bit_cast(int, rc[u].x) : wlo;
      if (MODE == 0) h[u] = *reinterpret_cast<const float4*>(wbytes + min(max(src - wlo, 0), WROWS - 1) * (WF * 4));
      else h[u] = src4[(int64_t)src * WC4 + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pk_axpy(lo[0], hi[0], rc[u].y, h[u]);
      if (E > 1) pk_axpy(lo[1], hi[1], rc[u].z, h[u]);
      if (E > 2) pk_axpy(lo[2], hi[2], rc[u].w, h[u]);
    }
  }
#pragma unroll
  for (int n = 0; n < E; ++n)
    *reinterpret_cast<float4*>(tb + al * ld + n * WF + 4 * c) = make_float4(lo[n][0], lo[n][1], hi[n][0], hi[n][1]);
  if (pl) {
    float m = 0.f;
#pragma unroll
    for (int n = 0; n < E; ++n) m = fmaxf(fmaxf(m, fmaxf(fabsf(lo[n][0]), fabsf(lo[n][1]))), fmaxf(fabsf(hi[n][0]), fabsf(hi[n][1])));
    m = fmaxf(m, ror_f<8>(m)); m = fmaxf(m, ror_f<4>(m)); m = fmaxf(m, ror_f<2>(m)); m = fmaxf(m, ror_f<1>(m));
    const int ef = (__builtin_bit_cast(int, m) >> 23) & 255;
    const int sb = (ef == 0 || ef == 255) ? 127 : min(267 - ef, 253);      // S = 2^(14 - e), 2^e > max; zero / non-finite rows: 1
    const float S = __builtin_bit_cast(float, sb << 23);
    if (c == 0) { rs[al] = __builtin_bit_cast(float, (254 - sb) << 23) * (1.0f / 256.0f); sbv[al] = sb; }
    // the tile's reference scale for the dw product: the smallest S (= the largest row) among rows that are not all
    // zero; this wave's four atoms here, the eight waves' minima meet after the barrier
    const int wm = wave_min_i32(ef == 0 ? 253 : sb);
    if (lane == 63) wmin[wave] = wm;
    constexpr int ROWB = (E * WF + 8) * 2, PLANE = WTA * ROWB;
    char* p = pl + al * ROWB + 8 * c;
#pragma unroll
    for (int n = 0; n < E; ++n) {
      unsigned h0, l0, h1, l1;
      split2_pair(S * lo[n][0], S * lo[n][1], h0, l0);
      split2_pair(S * hi[n][0], S * hi[n][1], h1, l1);
      *reinterpret_cast<u32x2*>(p + n * (WF * 2)) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(p + n * (WF * 2) + PLANE) = u32x2{l0, l1};
    }
  }
}

template <int E>
__device__ __noinline__ void node_gather_global(int wave, int lane, const int* s_ptr, const float4* recs,
                                                int rec_base, float* tb, int ld, const float4* src4, char* pl, float* rs,
                                                int* sbv, int* wmin) {
  node_gather<E, 1>(wave, lane, 0, s_ptr, recs, rec_base, tb, ld, nullptr, src4, pl, rs, sbv, wmin);
}

#ifdef WN_STAMP
__device__ unsigned long long wn_stamps[64];
#define WN_T(i) do { if (blockIdx.x == 100 && threadIdx.x == 0 && t == T0 + 2) wn_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define WN_T(i) do { } while (0)
#endif
template <int E, bool H2>
__device__ __forceinline__ void mp_win_bwd_node_body(const MpWinNodeArgs& a) {
  constexpr int KF = E * WF;
  constexpr int LD = KF + 4;
  constexpr int NT = KF / 16, NT2 = KF / 32;
  constexpr int ROWB = (KF + 8) * 2, PLANE = WTA * ROWB;     // fp16 piece planes of B (H2)
  constexpr int NCT = NT / 2;              // dw column tiles per wave: (lt = w & 3) x (NT/2 tiles of half w >> 2)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* win = smem;                                                   // [WROWS][64]   dP rows
  float* tile = win + WROWS * WF;                                      // [32][LD]      B
  float* htile = tile + WTA * LD;                                      // [32][SDP_LD]  h rows of the tile
  float4* s_rec = reinterpret_cast<float4*>(htile + WTA * SDP_LD);     // [2][NREC_CAP]
  int* s_ptr = reinterpret_cast<int*>(s_rec + 2 * NREC_CAP);           // [2][36]
  int* ctl = s_ptr + 2 * 36;                                           // [2][16]
  float* s_rs = reinterpret_cast<float*>(ctl + 32);                    // [32]  2^-8 / S per atom row (H2)
  int* s_sb = reinterpret_cast<int*>(s_rs + 32);                       // [32]  biased exponent of S per atom row (H2)
  int* s_wmin = s_sb + 32;                                             // [8]   per-wave minimum of it over non-zero rows
  char* planes = reinterpret_cast<char*>(s_wmin + 8);                  // [2][32][ROWB] (H2)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t T0 = (int64_t)blockIdx.x * a.tiles_per_wg;
  const int64_t T1 = std::min<int64_t>(T0 + a.tiles_per_wg, a.ntiles);
  float* part = a.partial + (int64_t)blockIdx.x * (KF * WF);
  char* const pl = H2 ? planes : nullptr;

  const float4* src4 = reinterpret_cast<const float4*>(a.dP);
  float4* win4 = reinterpret_cast<float4*>(win);
  for (int t = tid; t < WROWS * WC4; t += WTHREADS) win4[t] = f4zero();
  const bool dma_ok = a.N * (int64_t)(WF * 4) < ((int64_t)1 << 32);
  const dma_i4 psrc = dma_rsrc(a.dP, dma_ok ? (unsigned)(a.N * (WF * 4)) : 0u);

  const int ct = wave & 3, hh = wave >> 2;      // dh: column tile, atom half;  dw: l-tile ct, column half hh
  float wf[H2 ? 1 : KF / 4];
  u32x4 wh[H2 ? NT2 : 1], wl[H2 ? NT2 : 1];      // fp16 pieces of 2^8 Wn: A operands of v_mfma_f32_16x16x32_f16
  if (H2) {
    const u32x4* p = reinterpret_cast<const u32x4*>(a.WfragN) + (size_t)(ct * NT2) * 2 * 64 + lane;
#pragma unroll
    for (int T = 0; T < NT2; ++T) { wh[T] = p[(2 * T) * 64]; wl[T] = p[(2 * T + 1) * 64]; }
#pragma unroll
    for (int T = 0; T < NT2; ++T)
#pragma unroll
      for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(wh[T][j])); asm volatile("" : "+v"(wl[T][j])); }
  } else {
    const float4* p = reinterpret_cast<const float4*>(a.WfragN) + (ct * NT) * 64 + lane;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const float4 v = p[T * 64];
      wf[4 * T + 0] = v.x; wf[4 * T + 1] = v.y; wf[4 * T + 2] = v.z; wf[4 * T + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < KF / 4; ++i) asm volatile("" : "+v"(wf[i]));
  }
  f32x4 accW[NCT];
#pragma unroll
  for (int u = 0; u < NCT; ++u) accW[u] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (T0 < T1) {
    // per-tile inputs in flight
    f32x4 p_rec0, p_rec1;      // native vectors: as HIP float4 (a union type) the two stayed in a 32-byte stack slot
    float4 p_h, p_dH;
    int p_ptr;
    const int a16 = lane & 15, g4 = lane >> 4;
    const int col = 16 * ct + 4 * g4;
    const int64_t nnz = a.csc_ptr[a.N];
    // The record range of a tile starts at csc_ptr[32 t]; that offset is itself a global load, so it is
    // fetched one issue ahead (base_next) — a dependent load inside issue() would expose its latency.
    int p_base, p_end;
    int base_next = a.csc_ptr[std::min<int64_t>(T0 * WTA, a.N)];
    auto issue = [&](int64_t t) {          // records + pointers of tile t (calls walk consecutive tiles)
      const int64_t r0 = t * WTA;
      const int64_t rp = std::min<int64_t>(r0 + tid, a.N);
      p_ptr = a.csc_ptr[tid <= WTA ? rp : a.N];
      p_end = a.csc_ptr[std::min<int64_t>(r0 + WTA, a.N)];
      p_base = base_next;
      const int64_t q0 = (int64_t)p_base + tid, q1 = q0 + WTHREADS;
      p_rec0 = *reinterpret_cast<const f32x4*>(a.rec + (q0 < nnz ? q0 : (nnz > 0 ? nnz - 1 : 0)));
      p_rec1 = *reinterpret_cast<const f32x4*>(a.rec + (q1 < nnz ? q1 : (nnz > 0 ? nnz - 1 : 0)));
      base_next = a.csc_ptr[std::min<int64_t>(r0 + WTA, a.N)];
    };
    auto commit = [&](int64_t t) {
      float4* rc = s_rec + (t & 1) * NREC_CAP;
      int* pp = s_ptr + (t & 1) * 36;
      *reinterpret_cast<f32x4*>(rc + tid) = p_rec0;
      if (tid + WTHREADS < NREC_CAP) *reinterpret_cast<f32x4*>(rc + tid + WTHREADS) = p_rec1;
      if (tid <= WTA) pp[tid] = p_ptr;
      // row range over the tile's OWN records only (the staging area also holds the head of later tiles)
      int lo = 0x7fffffff, hi = -1;
      const int s0 = __builtin_bit_cast(int, p_rec0[0]), s1 = __builtin_bit_cast(int, p_rec1[0]);
      if (p_base + tid < p_end) { lo = s0; hi = s0; }
      if (p_base + tid + WTHREADS < p_end) { lo = min(lo, s1); hi = max(hi, s1); }
      lo = wave_min_i32(lo);
      hi = -wave_min_i32(-hi);
      if (lane == 63) { ctl[(t & 1) * 16 + wave] = lo; ctl[(t & 1) * 16 + 8 + wave] = hi; }
    };
    auto issue_rows = [&](int64_t t) {     // h rows of tile t (dw operand) and this lane's dH chunk (epilogue)
      const int64_t rh = t * WTA + (tid >> 4);
      p_h = rh < a.N ? reinterpret_cast<const float4*>(a.h)[rh * WC4 + (tid & 15)] : f4zero();
      const int64_t rd = t * WTA + 16 * hh + a16;
      p_dH = *reinterpret_cast<const float4*>(a.dH + (rd < a.N ? rd : a.N - 1) * WF + col);
    };

    int wlo = -(1 << 30), mode = 0;
    issue(T0);
    issue_rows(T0);
    commit(T0);
    issue(T0 + 1 < T1 ? T0 + 1 : T0);
    NG_LDS_BARRIER();
    if (win_decide(ctl + (T0 & 1) * 16, wlo, mode)) win_stage(win4, src4, wlo, a.N, tid);
    NG_LDS_BARRIER();

#pragma unroll 1
    for (int64_t t = T0; t < T1; ++t) {
      // ---- vector interval: records of t+1 and h rows of t into LDS FIRST, then the B tile of t, then the
      // requests for t+2.  Nothing freshly requested may be live across the gather: its out-of-line
      // global-memory variant is a call, and registers live across a call are saved to scratch — with a
      // vmcnt wait on the loads that fill them (measured: the prefetch latency came back every tile).
      WN_T(0);
      *reinterpret_cast<float4*>(htile + (tid >> 4) * SDP_LD + 4 * (tid & 15)) = p_h;
      const float4 dHc = p_dH;
      if (t + 1 < T1) commit(t + 1);
      WN_T(1);
      {
        const int* pp = s_ptr + (t & 1) * 36;
        const float4* rc = s_rec + (t & 1) * NREC_CAP;
        const int rec_base = pp[0];
        const bool fits = pp[WTA] - rec_base <= NREC_CAP;
        // a tile with more records than the staging area reads them from global memory instead (separate
        // call sites: the fast path must see an LDS pointer, not a generic one)
        if (mode == 0 && fits) node_gather<E, 0>(wave, lane, wlo, pp, rc, rec_base, tile, LD, win4, src4, pl, s_rs, s_sb, s_wmin);
        else if (fits) node_gather_global<E>(wave, lane, pp, rc, rec_base, tile, LD, src4, pl, s_rs, s_sb, s_wmin);
        else node_gather_global<E>(wave, lane, pp, a.rec + rec_base, rec_base, tile, LD, src4, pl, s_rs, s_sb, s_wmin);
      }
      WN_T(2);
      issue(t + 2 < T1 ? t + 2 : t);
      issue_rows(t + 1 < T1 ? t + 1 : t);
      WN_T(3);
      NG_LDS_BARRIER();
      WN_T(4);
      // the next tile's window, when it needs one: every gather of this tile is done and the range of t + 1 was published
      // before the barrier; nothing reads the window until the gather behind the interval's last barrier
      bool restage = false;
      if (t + 1 < T1) restage = win_decide(ctl + ((t + 1) & 1) * 16, wlo, mode);
      if (restage) {
        if (dma_ok) win_dma(win, psrc, wlo, wave, lane);
        else win_stage(win4, src4, wlo, a.N, tid);
      }
      // ---- matrix interval: dh = dH + B Wn ;  dw += h^T B
      {
        // operand reads run two k-steps ahead of their MFMAs (pinned): holding all NT of them kept 48 VGPRs
        // live and pushed freshly requested prefetch registers into scratch — behind a vmcnt wait
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float osc = 1.0f;
        if (H2) {
          // B operand: lane (atom a16, k-slots 8 g4 .. +7 of the 32-wide step) = 16 B of each piece plane, two steps
          // ahead; acc0 collects the small products, acc1 the leading ones
          const char* xrow = planes + (16 * hh + a16) * ROWB + 16 * g4;
          osc = s_rs[16 * hh + a16];
          u32x4 xh[NT2], xl[NT2];
#pragma unroll
          for (int T = 0; T < 2 && T < NT2; ++T) {
            xh[T] = *reinterpret_cast<const u32x4*>(xrow + 64 * T);
            xl[T] = *reinterpret_cast<const u32x4*>(xrow + 64 * T + PLANE);
          }
#pragma unroll
          for (int T = 0; T < NT2; ++T) {
            if (T + 2 < NT2) {
              xh[T + 2] = *reinterpret_cast<const u32x4*>(xrow + 64 * (T + 2));
              xl[T + 2] = *reinterpret_cast<const u32x4*>(xrow + 64 * (T + 2) + PLANE);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[T]), __builtin_bit_cast(f16x8, xh[T]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[T]), __builtin_bit_cast(f16x8, xh[T]), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[T]), __builtin_bit_cast(f16x8, xl[T]), acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          const float* xrow = tile + (16 * hh + a16) * LD + 4 * g4;
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
        WN_T(5);
        const int64_t row = t * WTA + 16 * hh + a16;
        const float4 v = make_float4(fmaf(acc0[0] + acc1[0], osc, dHc.x), fmaf(acc0[1] + acc1[1], osc, dHc.y),
                                     fmaf(acc0[2] + acc1[2], osc, dHc.z), fmaf(acc0[3] + acc1[3], osc, dHc.w));
        *reinterpret_cast<float4*>(row < a.N ? a.dh + row * WF + col : a.dummy + col) = v;
      }
      if (H2) {
        // dw on the fp16 pipe.  The contraction runs over the tile's 32 atoms (ONE 32-deep MFMA step), whose B rows carry
        // different power-of-two scales S_a in the piece planes: the h operand takes the inverse, h'[a] = h[a] * S_ref / S_a
        // with S_ref the smallest S of the tile (its largest row; ratio <= 1, a small row's h' may underflow — its
        // term is below the fp32 rounding of the large rows' terms), so that h'[a] (S_a B[a]) = S_ref h[a] B[a]; the tile's
        // product goes into a fresh accumulator and is added to the running sums times 1 / S_ref.
        int sbref = s_wmin[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) sbref = min(sbref, s_wmin[i]);
        const float inv_ref = __builtin_bit_cast(float, (254 - sbref) << 23);
        float hv[8];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
          const int at = 8 * g4 + tt;
          const int sba = s_sb[at];
          // all-zero rows keep S = 1 (sba may lie below the reference): their B pieces are exact zeros, any finite h' does
          const float ratio = sba < sbref ? 1.0f : __builtin_bit_cast(float, max(sbref - sba + 127, 0) << 23);
          hv[tt] = htile[at * SDP_LD + 16 * ct + a16] * ratio;
        }
        // the layer input is a forward quantity of any size (the reference's MPLayer is plain fp32): row l of h'^T (this
        // lane's l = 16 ct + a16 over the tile's 32 atoms: 8 here, the rest in the lanes 16 / 32 / 48 further on) gets a
        // power-of-two scale when its largest entry reaches 2^15, and the output rows take the inverse.  1 (bit-neutral)
        // for every ordinary input.
        float hm = fmaxf(fmaxf(fmaxf(fabsf(hv[0]), fabsf(hv[1])), fmaxf(fabsf(hv[2]), fabsf(hv[3]))),
                         fmaxf(fmaxf(fabsf(hv[4]), fabsf(hv[5])), fmaxf(fabsf(hv[6]), fabsf(hv[7]))));
        f32x4 osc4 = {inv_ref, inv_ref, inv_ref, inv_ref};
        if (__builtin_amdgcn_ballot_w64(hm >= 32768.0f) != 0) {      // wave-uniform, never taken for ordinary activations
          hm = fmaxf(hm, __shfl_xor(hm, 16));
          hm = fmaxf(hm, __shfl_xor(hm, 32));
          const int hef = (__builtin_bit_cast(int, hm) >> 23) & 255;
          const bool hbig = hef >= 127 + 15 && hef != 255;
          const float hs = hbig ? __builtin_bit_cast(float, (268 - hef) << 23) : 1.0f;
          const float hsi = hbig ? __builtin_bit_cast(float, (hef - 14) << 23) : 1.0f;
#pragma unroll
          for (int tt = 0; tt < 8; ++tt) hv[tt] *= hs;
          // accumulator element r of this lane is output row l = 16 ct + 4 g4 + r: its inverse scale sits in lane 4 g4 + r
#pragma unroll
          for (int r = 0; r < 4; ++r) osc4[r] = __shfl(hsi, 4 * g4 + r) * inv_ref;
        }
        unsigned h0, l0, h1, l1, h2, l2, h3, l3;
        split2_pair(hv[0], hv[1], h0, l0); split2_pair(hv[2], hv[3], h1, l1);
        split2_pair(hv[4], hv[5], h2, l2); split2_pair(hv[6], hv[7], h3, l3);
        const u32x4 ah = {h0, h1, h2, h3}, al = {l0, l1, l2, l3};
        // B operand: this lane's column of the piece planes for atoms 8 g4 .. 8 g4 + 7: two transposing reads of four atoms
        const char* bp = planes + (8 * g4 + (a16 >> 2)) * ROWB + (16 * (NCT * hh) + 4 * (a16 & 3)) * 2;
#pragma unroll
        for (int u = 0; u < NCT; ++u) {
          const char* q0 = bp + 32 * u;
          const gs16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)q0);
          const gs16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(q0 + 4 * ROWB));
          const gs16x4 w0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(q0 + PLANE));
          const gs16x4 w1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(q0 + PLANE + 4 * ROWB));
          const u32x2 a0 = __builtin_bit_cast(u32x2, v0), a1 = __builtin_bit_cast(u32x2, v1);
          const u32x2 c0 = __builtin_bit_cast(u32x2, w0), c1 = __builtin_bit_cast(u32x2, w1);
          const u32x4 bh = {a0[0], a0[1], a1[0], a1[1]}, bl = {c0[0], c0[1], c1[0], c1[1]};
          f32x4 at = {0.f, 0.f, 0.f, 0.f};
          at = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, bh), at, 0, 0, 0);
          at = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bl), at, 0, 0, 0);
          at = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), at, 0, 0, 0);
          accW[u] += at * osc4;
        }
      } else {
        // D[i = l][j = (n,m)] += sum_atoms h[atom][l] B[atom][(n,m)]: l-tile ct, column tiles NCT*hh + u
        const float* ha = htile + 16 * ct + a16;
        const float* bb = tile + 16 * (NCT * hh) + a16;
#pragma unroll
        for (int T = 0; T < 8; ++T) {
          const int ro = (T & 3) + 4 * g4 + 16 * (T >> 2);
          const float av = ha[ro * SDP_LD];
#pragma unroll
          for (int u = 0; u < NCT; ++u)
            accW[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bb[ro * LD + 16 * u], accW[u], 0, 0, 0);
        }
      }
      WN_T(6);
      if (restage && dma_ok) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      NG_LDS_BARRIER();
      WN_T(7);
    }
  }
  // ---- dw partial of this workgroup, layout [(n,m)][l]: lane holds l = 16ct + 4(lane>>4) + r, column = 16(NCT*hh+u) + (lane&15)
  {
    const int a16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int u = 0; u < NCT; ++u) {
      const int cidx = 16 * (NCT * hh + u) + a16;
      *reinterpret_cast<float4*>(part + cidx * WF + 16 * ct + 4 * g4) =
          make_float4(accW[u][0], accW[u][1], accW[u][2], accW[u][3]);
    }
  }
}

// Range guard: as in the edge kernel — the piece operands B (per-row scale) and h' (per-row scale, above) are range-safe by
// construction, a weight image out of range is known at the first instruction and selects the fp32-input body.
template <int E, bool H2>
__global__ __launch_bounds__(WTHREADS, 1) void mp_win_bwd_node_kernel(MpWinNodeArgs a) {
  if (H2 && a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag))) {
    a.WfragN = a.WfragN32;
    mp_win_bwd_node_body<E, false>(a);
  } else {
    mp_win_bwd_node_body<E, H2>(a);
  }
}

// (row_of != nullptr: CSR lists — the source of entry eid is row_of[eid] instead of eid / K)
__global__ void mp_records_kernel(int64_t N, int K, int E, const int32_t* __restrict__ csc_ptr,
                                  const int32_t* __restrict__ csc_edge, const float* __restrict__ e,
                                  float4* __restrict__ rec, const int32_t* __restrict__ row_of) {
  const int64_t nnz = csc_ptr[N];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // four entries per trip: the four edge ids are requested together, then the twelve gathers (one entry per trip ran
  // two dependent memory round trips per entry: 36 us for 2.1 M entries)
  for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p0 < nnz; p0 += 4 * stride) {
    int eid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) eid[u] = csc_edge[std::min<int64_t>(p0 + u * stride, nnz - 1)];
    float4 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      r[u].x = __builtin_bit_cast(float, row_of ? row_of[eid[u]] : eid[u] / K);
      if (E == 3) {
        // one 12-byte load per entry (global_load_dwordx3): three scattered dword loads fetched the entry's sector three times
        struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
        const F3 v = *reinterpret_cast<const F3*>(e + (int64_t)eid[u] * 3);
        r[u].y = v.x; r[u].z = v.y; r[u].w = v.z;
      } else {
        r[u].y = e[(int64_t)eid[u] * E];
        r[u].z = E > 1 ? e[(int64_t)eid[u] * E + 1] : 0.f;
        r[u].w = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p0 + u * stride < nnz) rec[p0 + u * stride] = r[u];
  }
}

size_t node_lds_bytes(int E, bool h2) {
  return (size_t)(WROWS * WF + WTA * (E * WF + 4) + WTA * SDP_LD) * 4 + (size_t)2 * NREC_CAP * 16 + (2 * 36 + 32 + 32 + 32 + 8) * 4 +
         (h2 ? (size_t)2 * WTA * (E * WF + 8) * 2 : 0);
}

size_t edge_lds_bytes(int K, int E) {
  return (size_t)(WROWS * WF + WTA * (E * WF + 4) + 2 * SDP_SLOT + 2 * WTA * K + 32 + 2 * WTA) * 4;
}

}  // namespace

// the dA product of the edge kernel and the dh product of the node kernel on the fp16 pipe with two-piece operands
// unless NG_GEMM_MATH=fp32
static bool mp_win_bwd_h2() { return !sw().gemm_math_fp32; }

int mp_win_records(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, const int32_t* csc_ptr,
                   const int32_t* csc_edge, const float* e, float* rec, const int32_t* row_of, int64_t n_entries) {
  if (N == 0) return NG_OK;
  ProfScope ps(ctx, st, "mp_records");
  const int grid = (int)std::min<int64_t>(cdiv(std::max<int64_t>(row_of ? n_entries : N * K, 1), 256), (int64_t)ctx->num_cu * 8);
  hipLaunchKernelGGL(mp_records_kernel, dim3(grid), dim3(256), 0, st, N, K, E, csc_ptr, csc_edge, e,
                     reinterpret_cast<float4*>(rec), row_of);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

size_t mp_win_node_scratch_floats(ng_ctx* ctx, int E) { return (size_t)(ctx->num_cu + 1) * E * WF * WF; }

// dh_in = dh_out + B Wn,  dw = h^T B  with B the incoming-edge aggregate of dP (never materialised).
// guard.word != nullptr: WfragN is the piece image, WfragN32 the fp32 one the kernel switches to when the guard is up.
int mp_win_bwd_node(ng_ctx* ctx, hipStream_t st, int64_t N, int E, const float* h, const float* dP,
                    const int32_t* csc_ptr, const float* rec, const float* WfragN, const float* dh_out,
                    float* dh_in, float* dw, float* scratch, float* dummy, RangeGuard guard, const float* WfragN32,
                    const unsigned* wflag, unsigned wflag_ver) {
  MpWinNodeArgs a{};
  a.N = N; a.ntiles = cdiv(N, WTA);
  const int64_t per = win_tiles_per_wg(a.ntiles, ctx->num_cu);
  a.tiles_per_wg = (int)per;
  a.dP = dP; a.dH = dh_out; a.h = h; a.csc_ptr = csc_ptr; a.rec = reinterpret_cast<const float4*>(rec);
  a.WfragN = WfragN; a.dh = dh_in; a.partial = scratch; a.dummy = dummy; a.guard = guard;
  const int grid = (int)cdiv(a.ntiles, per);
  const bool h2 = mp_win_bwd_h2();
#define NODE(HH)                                                                                                        \
  switch (E) {                                                                                                          \
    case 1: hipLaunchKernelGGL((mp_win_bwd_node_kernel<1, HH>), dim3(grid), dim3(WTHREADS), node_lds_bytes(E, HH), st, a); break; \
    case 2: hipLaunchKernelGGL((mp_win_bwd_node_kernel<2, HH>), dim3(grid), dim3(WTHREADS), node_lds_bytes(E, HH), st, a); break; \
    case 3: hipLaunchKernelGGL((mp_win_bwd_node_kernel<3, HH>), dim3(grid), dim3(WTHREADS), node_lds_bytes(E, HH), st, a); break; \
  }
  a.WfragN32 = WfragN32; a.wflag = wflag; a.wflag_ver = wflag_ver;
  {
    ProfScope ps(ctx, st, "mp_win_bwd_node");
    if (h2) { NODE(true) } else { NODE(false) }
    NG_HIP(ctx, hipGetLastError());
#ifdef WN_STAMP
    if (getenv("NG_WN_STAMP")) {
      unsigned long long hs[64];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(wn_stamps), sizeof(hs));
      fprintf(stderr, "wn tiles/wg %d: store+commit %llu gather %llu issue %llu barrier %llu dh %llu dw %llu barrier %llu\n", a.tiles_per_wg,
              hs[1] - hs[0], hs[2] - hs[1], hs[3] - hs[2], hs[4] - hs[3], hs[5] - hs[4], hs[6] - hs[5], hs[7] - hs[6]);
    }
#endif
  }
#undef NODE
  ProfScope ps(ctx, st, "reduce_partials");
  // partial idx = (n*64 + m)*64 + l  ->  dw[(l*64 + m)*E + n]
  return reduce_or_defer(ctx, st, scratch, grid, (int64_t)E * WF * WF, dw, 2, WF, E, WF, (int64_t)E * WF * WF);
}

bool mp_win_bwd_supported(int F, int E, int K) {
  return F == WF && E >= 1 && E <= 3 && K % 4 == 0 && K >= 4 && K <= 16;
}

int mp_win_bwd_edge(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, const float* h,
                    const int32_t* nlist, const float* inv_degree, const float* WfragT, const float* s_save,
                    const float* dh_out, float* dP, float* de, int de_accum, float* dummy, RangeGuard guard,
                    const float* WfragT32, const unsigned* wflag, unsigned wflag_ver) {
  if (mp_win_bwd_h2() && guard.word && sw().mp_w16 && mp_win16_bwd_edge_supported(E, K))
    return mp_win16_bwd_edge_launch(ctx, st, N, K, E, act, h, nlist, inv_degree, WfragT, s_save, dh_out, dP, de, de_accum, dummy, guard,
                                    WfragT32, wflag, wflag_ver);
  MpWinEdgeArgs a{};
  a.N = N; a.K = K; a.ntiles = cdiv(N, WTA);
  const int64_t per = win_tiles_per_wg(a.ntiles, ctx->num_cu);
  a.tiles_per_wg = (int)per;
  a.dH = dh_out; a.S = act == NG_ACT_NONE ? nullptr : s_save; a.rowscale = inv_degree; a.h = h;
  a.nlist = nlist; a.WfragT = WfragT; a.dP = dP; a.de = de; a.dummy = dummy; a.act = act;
  a.accumulate = de_accum; a.guard = guard;
  const int grid = (int)cdiv(a.ntiles, per);
  const size_t lds = edge_lds_bytes(K, E);
  const bool h2 = mp_win_bwd_h2();
#define EDGE(HH)                                                                                                  \
  switch (E) {                                                                                                    \
    case 1: hipLaunchKernelGGL((mp_win_bwd_edge_kernel<1, HH>), dim3(grid), dim3(WTHREADS), lds, st, a); break;   \
    case 2: hipLaunchKernelGGL((mp_win_bwd_edge_kernel<2, HH>), dim3(grid), dim3(WTHREADS), lds, st, a); break;   \
    case 3: hipLaunchKernelGGL((mp_win_bwd_edge_kernel<3, HH>), dim3(grid), dim3(WTHREADS), lds, st, a); break;   \
  }
  a.WfragT32 = WfragT32; a.wflag = wflag; a.wflag_ver = wflag_ver;
  {
    ProfScope ps(ctx, st, "mp_win_bwd_edge");
    if (h2) { EDGE(true) } else { EDGE(false) }
    NG_HIP(ctx, hipGetLastError());
  }
#undef EDGE
  return NG_OK;
}

// MPLayer backward for atom_feature_size == 64 on the window-resident kernels:
//   edge kernel   dP = dH * act'(S) * v ;  dA = dP Wp^T (LDS) ;  de (+)= <dA, h[nlist]>
//   node kernel   B = incoming-edge aggregate of dP (records) ;  dh = dH + B Wn ;  dw = h^T B
bool mp_win_bwd_enabled(int F, int E, int K) { return mp_win_bwd_supported(F, E, K) && !sw().mp_layered; }

int mp_win_bwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, const float* h, const int32_t* nlist,
               const float* e, const float* inv_degree, const float* w, const float* s_save, const int32_t* csc_ptr,
               const int32_t* csc_edge, const float* dh_out, float* dh_in, float* de, int de_accum, float* dw,
               const float* csc_rec) {
  const int KF = E * WF;
  const size_t dw_scr = mp_win_node_scratch_floats(ctx, E);
  const bool h2 = mp_win_bwd_h2();
  // guarded call (RangeGuard, ng_internal.h): the pack launch raises the guard for weights out of the piece range and
  // both kernels then run their fp32-input bodies
  const bool guarded = h2;
  RangeGuard guard{nullptr, 0};
  if (guarded) {
    guard = range_guard_begin(ctx);
    if (!guard.word) return NG_ERR_NOMEM;
  }
  // four packed weight images (piece T, N; fp32 T, N) + the flag word: a cached image while the weights are frozen /
  // refreshed behind Adam (pack_bodies.cuh), else the head of the scratch.
  // scratch: [images] | dP [N,64] | records [N*K,4] (when the caller has none) | dw partials
  const size_t img_floats = (size_t)4 * KF * WF + 16;
  bool have = false;
  float* imgs = (float*)cached_image(ctx, w, h2 ? 8 : 14, img_floats * 4, &have);
  const bool cached = imgs != nullptr;
  const size_t rec_floats = csc_rec ? 0 : (size_t)N * K * 4;
  float* ws = (float*)workspace(ctx, (size_t)((cached ? 0 : img_floats) + N * WF + rec_floats + dw_scr + 64) * 4);
  if (!ws) return NG_ERR_NOMEM;
  if (!cached) { imgs = ws; ws += img_floats; have = false; }
  float* WfragT = imgs;
  float* WfragN = WfragT + KF * WF;
  float* WfragT32 = WfragN + KF * WF;
  float* WfragN32 = WfragT32 + KF * WF;
  unsigned* wflag = cached && guarded ? reinterpret_cast<unsigned*>(WfragN32 + KF * WF) : nullptr;
  float* dP = ws;
  float* rec = dP + N * WF;
  float* scr = rec + rec_floats;
  float* dummy = scr + dw_scr;
  // deferred reductions (reduce.cuh): the dw partials of this layer stay in the reduction arena until the flush
  if (float* dscr = deferred_partials(ctx, dw_scr)) scr = dscr;
  int rc = NG_OK;
  if (!have) {
    if (h2) {
      const PackJob j = mpw_bwd_job(E, w, WfragT, WfragN, guarded ? WfragT32 : nullptr, guarded ? WfragN32 : nullptr, wflag, guard);
      rc = pack_launch(ctx, st, j);
      if (rc == NG_OK && cached) cache_set_job(ctx, w, 8, j);
    } else {
      rc = mpw_pack2(ctx, st, E, w, 2, WfragT, 1, WfragN);       // (strict-fp32 mode: two small launches, not registered)
      if (cached) have = false;
    }
  }
  if (rc) return rc;
  const unsigned wver = pack_flag_version(ctx);
  rc = mp_win_bwd_edge(ctx, st, N, K, E, act, h, nlist, inv_degree, WfragT, s_save, dh_out, dP, de, de_accum, dummy, guard,
                       WfragT32, wflag, wver);
  if (rc) return rc;
  if (!csc_rec) {
    rc = mp_win_records(ctx, st, N, K, E, csc_ptr, csc_edge, e, rec);
    if (rc) return rc;
    csc_rec = rec;
  }
  return mp_win_bwd_node(ctx, st, N, E, h, dP, csc_ptr, csc_rec, WfragN, dh_out, dh_in, dw, scr, dummy, guard, WfragN32, wflag, wver);
}

}  // namespace ng
