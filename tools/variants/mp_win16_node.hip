// Sixteen-wave form of the node-side backward window kernel of mp_win_bwd.hip (atom_feature_size 64, E <= 3).
// Reference: the backward of nmrgnn/layers.py:26-46, node side:
//   B[j][n][:] = sum over incoming edges (i, k -> j) of e[i,k,n] dP[i][:] ;  dh = dH + B Wn ;  dw += h^T B.
//
// Same reasoning as mp_win16.hip / mp_win16_bwd.hip (the eight-wave kernel waits at two waves per SIMD and 220 VGPRs):
// 1024 threads, 64-atom tiles, four waves per SIMD at 128 VGPRs.
//   * incoming-edge records loaded by the lane that uses them: lane (atom, c) holds records c and 16 + c of its atom
//     (requested a tile ahead, their CSC offsets two tiles ahead); the first sixteen are walked by rotation (DPP), the tail
//     record by record with the record broadcast over the atom's sixteen lanes by ds_bpermute — no record staging in LDS
//     (49 KB at this tile size).  Atoms with more than 32 incoming edges fetch their further records inside the gather
//     and send their tile to the global-memory gather (the range of those sources is not known when the window is chosen);
//   * dh = dH + B Wn split over the contraction between partner waves (24 weight registers), as in mp_win16.hip;
//   * dw += h^T B with BOTH operands from transposing reads of fp16 piece planes: the tile's h rows are split once where
//     they are staged (a power-of-two row scale only for rows reaching 2^15), the B rows carry their own scales from the
//     gather; what makes the rows of one 32-deep step commensurable — a power of two <= 1 per row — is applied to the B
//     pieces as packed fp16 multiplies (exact, or an underflow of a term far below the sum's rounding).  Twelve accumulator
//     registers per wave (4 l-tiles x 4 column groups over 16 waves) instead of twenty-four;
//   * window (dP rows) by LDS-DMA beside the matrix interval.
// LDS: window 72 KB + B planes 50 KB + h planes 18 KB + exchange tile 16 KB.  B comes out bit for bit as in the eight-wave
// kernel; dh and dw agree with it to rounding (other summation order / scaling point).
#include <algorithm>
#include <cstdio>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "edge_fused.h"   // NG_LDS_BARRIER
#include "h2_common.cuh"
#include "mp_win16_common.cuh"

namespace ng {
namespace w16n {

using namespace w16c;

constexpr int HROWB = (WF + 8) * 2, HPLANE = WTA * HROWB;      // h piece planes: 144 B per row
constexpr int SDP_LD = 68;                                     // fp32 h rows (fp32 body)

typedef short gs16x4 __attribute__((ext_vector_type(4)));

struct Args {
  int64_t N;
  int64_t ntiles;
  int tiles_per_wg;
  const float* dP;          // [N][64] gathered rows
  const float* dH;          // [N][64] upstream gradient of the layer output (residual term of dh)
  const float* h;           // [N][64] layer input (dw operand)
  const int32_t* csc_ptr;   // [N+1]
  const float4* rec;        // [nnz] records { source atom (int bits), e0, e1, e2 }
  const float* WfragN;      // piece fragments (mpw_h2<1>)
  const float* WfragN32;    // fp32 fragments (mpw_f32 mode 1)
  float* dh;                // [N][64] out
  float* partial;           // [grid][64*E*64] dw partials, layout [(n,m)][l]
  float* dummy;
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
  static constexpr int LD = KF + 4;                    // fp32 row stride (floats), fp32 body
  static constexpr int ROWB = (KF + 8) * 2;            // fp16 plane row stride (bytes)
  static constexpr int PLANE = WTA * ROWB;
  static constexpr int BYTES = (2 * PLANE > WTA * LD * 4) ? 2 * PLANE : WTA * LD * 4;
};
constexpr int H_BYTES = (2 * HPLANE > WTA * SDP_LD * 4) ? 2 * HPLANE : WTA * SDP_LD * 4;

// four rotation steps of the first sixteen records (mp_win_bwd.hip: node_steps4)
template <int E, int S0, bool GLOBAL>
__device__ __forceinline__ void rot4(const char* __restrict__ wbytes, const float4* __restrict__ src4, int c, int roff, int gidx,
                                     const float (&w)[E], f32x2 (&lo)[E], f32x2 (&hi)[E]) {
  float4 h0, h1, h2, h3;
  if (!GLOBAL) {
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

// B row of atom al (this lane: float4 chunk c of every edge feature n), then its piece planes / fp32 row.
// rec0 / rec1: records c and 16 + c of the atom (zero where the atom has none).
template <int E, bool H2, bool GLOBAL>
__device__ __forceinline__ void gather(int wave, int lane, int al, int wlo, int p0, int cnt, const float4& rec0_raw, const float4& rec1_raw,
                                       const float4* __restrict__ recs, float* __restrict__ tb, const float4* __restrict__ win4,
                                       const float4* __restrict__ src4, float* __restrict__ rs, int* __restrict__ sbv, int* __restrict__ wmin) {
  const int c = lane & 15;
  const float4 rec0 = c < cnt ? rec0_raw : f4zero();
  const float4 rec1 = 16 + c < cnt ? rec1_raw : f4zero();
  int mx = cnt;
  mx = max(mx, __builtin_amdgcn_update_dpp(0, mx, 0x142, 0xa, 0xf, false));   // row_bcast15 (rows 1,3)
  mx = max(mx, __builtin_amdgcn_update_dpp(0, mx, 0x143, 0xc, 0xf, false));   // row_bcast31 (rows 2,3)
  const int mxw = __builtin_amdgcn_readlane(mx, 63);
  const char* wbytes = reinterpret_cast<const char*>(win4) + 16 * c;
  f32x2 lo[E], hi[E];
#pragma unroll
  for (int n = 0; n < E; ++n) { lo[n] = f32x2{0.f, 0.f}; hi[n] = f32x2{0.f, 0.f}; }
  {   // the first 16 records: rotation walk (lane c owns record c)
    const int src = c < cnt ? __builtin_bit_cast(int, rec0.x) : (GLOBAL ? 0 : wlo);
    float w[E];
    w[0] = rec0.y;
    if (E > 1) w[E > 1 ? 1 : 0] = rec0.z;
    if (E > 2) w[E > 2 ? 2 : 0] = rec0.w;
    const int roff = min(max(src - wlo, 0), WROWS - 1) * (WF * 4);
    rot4<E, 0, GLOBAL>(wbytes, src4, c, roff, src, w, lo, hi);
    rot4<E, 4, GLOBAL>(wbytes, src4, c, roff, src, w, lo, hi);
    rot4<E, 8, GLOBAL>(wbytes, src4, c, roff, src, w, lo, hi);
    rot4<E, 12, GLOBAL>(wbytes, src4, c, roff, src, w, lo, hi);
  }
  // records beyond the sixteenth, in order, four per trip: the record of slot q sits in lane q of the atom's row and reaches
  // the row's sixteen lanes by ds_bpermute
#pragma unroll 1
  for (int base = 16; base < mxw; base += 16) {
    float4 rr = rec1;
    if (base > 16) {      // (hubs: more than 32 incoming edges)
      rr = make_float4(0.f, 0.f, 0.f, 0.f);
      if (base + c < cnt) rr = recs[(int64_t)p0 + base + c];
    }
    const int nq = min(16, mxw - base);
#pragma unroll 1
    for (int q0 = 0; q0 < nq; q0 += 2) {      // (two per trip: four rows in flight here cost spills elsewhere — 112 against 102 us)
      int sv[2];
      float wv[2][E];
      float4 hv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int sl = ((lane & 48) + ((q0 + u) & 15)) << 2;
        sv[u] = __builtin_amdgcn_ds_bpermute(sl, __builtin_bit_cast(int, rr.x));
        wv[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sl, __builtin_bit_cast(int, rr.y)));
        if (E > 1) wv[u][E > 1 ? 1 : 0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sl, __builtin_bit_cast(int, rr.z)));
        if (E > 2) wv[u][E > 2 ? 2 : 0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sl, __builtin_bit_cast(int, rr.w)));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        // (a slot the atom does not have carries the zero record: weight 0, source row 0 — any finite row does)
        if (!GLOBAL) hv[u] = *reinterpret_cast<const float4*>(wbytes + min(max(sv[u] - wlo, 0), WROWS - 1) * (WF * 4));
        else hv[u] = src4[(int64_t)sv[u] * WC4 + c];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        pk_axpy(lo[0], hi[0], wv[u][0], hv[u]);
        if (E > 1) pk_axpy(lo[E > 1 ? 1 : 0], hi[E > 1 ? 1 : 0], wv[u][E > 1 ? 1 : 0], hv[u]);
        if (E > 2) pk_axpy(lo[E > 2 ? 2 : 0], hi[E > 2 ? 2 : 0], wv[u][E > 2 ? 2 : 0], hv[u]);
      }
    }
  }
  if (H2) {
    // the row into two fp16 piece planes, scaled by a power of two taken from its own max |B| (the 16 lanes of the DPP row
    // hold the whole row); rs[atom] = 2^-8 / S for the dh epilogue, sbv[atom] the exponent for the dw product
    float m = 0.f;
#pragma unroll
    for (int n = 0; n < E; ++n) m = fmaxf(fmaxf(m, fmaxf(fabsf(lo[n][0]), fabsf(lo[n][1]))), fmaxf(fabsf(hi[n][0]), fabsf(hi[n][1])));
    m = fmaxf(m, ror_f<8>(m)); m = fmaxf(m, ror_f<4>(m)); m = fmaxf(m, ror_f<2>(m)); m = fmaxf(m, ror_f<1>(m));
    const int ef = (__builtin_bit_cast(int, m) >> 23) & 255;
    const int sb = (ef == 0 || ef == 255) ? 127 : min(267 - ef, 253);      // S = 2^(14 - e), 2^e > max; zero / non-finite rows: 1
    const float S = __builtin_bit_cast(float, sb << 23);
    if (c == 0) { rs[al] = __builtin_bit_cast(float, (254 - sb) << 23) * (1.0f / 256.0f); sbv[al] = ef == 0 ? 253 : sb; }
    const int wm = wave_min_i32(ef == 0 ? 253 : sb);
    if (lane == 63) wmin[wave] = wm;
    char* p = reinterpret_cast<char*>(tb) + al * Tile<E>::ROWB + 8 * c;
#pragma unroll
    for (int n = 0; n < E; ++n) {
      unsigned h0, l0, h1, l1;
      split2_pair(S * lo[n][0], S * lo[n][1], h0, l0);
      split2_pair(S * hi[n][0], S * hi[n][1], h1, l1);
      *reinterpret_cast<u32x2*>(p + n * (WF * 2)) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(p + n * (WF * 2) + Tile<E>::PLANE) = u32x2{l0, l1};
    }
  } else {
#pragma unroll
    for (int n = 0; n < E; ++n)
      *reinterpret_cast<float4*>(tb + al * Tile<E>::LD + n * WF + 4 * c) = make_float4(lo[n][0], lo[n][1], hi[n][0], hi[n][1]);
  }
}

// out of line: inlined next to the window variant its global loads put vmcnt waits into the window path
template <int E, bool H2>
__device__ __noinline__ void gather_global(int wave, int lane, int al, int p0, int cnt, float4 rec0, float4 rec1, const float4* recs,
                                           float* tb, const float4* src4, float* rs, int* sbv, int* wmin) {
  gather<E, H2, true>(wave, lane, al, 0, p0, cnt, rec0, rec1, recs, tb, nullptr, src4, rs, sbv, wmin);
}

template <int E, bool H2>
__device__ __forceinline__ void body(const Args& a) {
  constexpr int KF = E * WF;
  constexpr int NT2 = KF / 32, NTH = NT2 / 2;          // dh: 32-wide k-steps in all, per k-half
  constexpr int NT = KF / 16, NT_H = NT / 2;           // fp32 body
  constexpr int LD = Tile<E>::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* win = smem;                                                            // [WROWS][64]   dP rows
  float* tile = win + WROWS * WF;                                               // B: piece planes / fp32 rows
  char* hpl = reinterpret_cast<char*>(tile) + Tile<E>::BYTES;                   // h of the tile: piece planes / fp32 rows [64][68]
  float4* xch = reinterpret_cast<float4*>(hpl + H_BYTES);                       // [16 blocks][64 lanes]
  float* s_rs = reinterpret_cast<float*>(xch + 16 * 64);                        // [64] 2^-8 / S per B row
  int* s_sb = reinterpret_cast<int*>(s_rs + WTA);                               // [64] exponent of S per B row (253: all-zero row)
  int* s_hb = s_sb + WTA;                                                       // [64] exponent of the h row's scale
  int* s_wmin = s_hb + WTA;                                                     // [16] per-wave minimum of s_sb
  int* s_hmin = s_wmin + NW;                                                    // [16] per-wave minimum of s_hb
  int* ctl = s_hmin + NW;                                                       // [2][2 NW]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t T0 = (int64_t)blockIdx.x * a.tiles_per_wg;
  const int64_t T1 = std::min<int64_t>(T0 + a.tiles_per_wg, a.ntiles);
  float* part = a.partial + (int64_t)blockIdx.x * (KF * WF);

  const float4* src4 = reinterpret_cast<const float4*>(a.dP);
  float4* win4 = reinterpret_cast<float4*>(win);
  for (int t = tid; t < WROWS * WC4; t += WTHREADS) win4[t] = f4zero();
  const int a16 = lane & 15, g4 = lane >> 4;
  // dh role: column tile, row-tile pair, k-half; this wave finishes row tile 2 rp + kh
  const int ct = wave & 3, rp = (wave >> 2) & 1, kh = wave >> 3;
  const int rt_own = 2 * rp + kh;
  u32x4 wh[H2 ? NTH : 1], wl[H2 ? NTH : 1];
  if (H2) {
    const u32x4* p = reinterpret_cast<const u32x4*>(a.WfragN) + (size_t)(ct * NT2 + kh * NTH) * 2 * 64 + lane;
#pragma unroll
    for (int T = 0; T < NTH; ++T) { wh[T] = p[(2 * T) * 64]; wl[T] = p[(2 * T + 1) * 64]; }
#pragma unroll
    for (int T = 0; T < NTH; ++T)
#pragma unroll
      for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(wh[T][j])); asm volatile("" : "+v"(wl[T][j])); }
  }
  // dw role: l-tile lt, column tiles E cg .. E cg + E - 1
  const int lt = wave & 3, cg = wave >> 2;
  f32x4 accW[E];
#pragma unroll
  for (int u = 0; u < E; ++u) accW[u] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (T0 < T1) {
    // gather role: atom al of the tile, lane c of its sixteen; staging role: row prow (== al), float4 column pc (== c)
    const int c = lane & 15, al = wave * 4 + (lane >> 4);
    const int nnz = __builtin_amdgcn_readfirstlane(a.csc_ptr[a.N]);      // (uniform: a scalar register)
    // in flight per tile: CSC offsets (two tiles ahead), records c and 16 + c (one tile ahead), h and dH chunks
    // (addresses: a per-tile base that is uniform over the workgroup — scalar registers — plus a 32-bit lane offset; 64-bit
    // per-lane pointers to the six arrays were hoisted out of the tile loop and then spilled: at 128 VGPRs they do not fit)
    auto load_ptr = [&](int64_t t, int& p0, int& cnt) {
      const int nvalid = (int)std::min<int64_t>(a.N - t * WTA, WTA);          // rows of the tile that exist (>= 1)
      const int32_t* cp = a.csc_ptr + t * WTA;
      p0 = cp[min(al, nvalid)];
      cnt = cp[min(al + 1, nvalid)] - p0;
    };
    auto load_rec = [&](int p0, int cnt, float4& r0, float4& r1) {
      const int last = max(nnz - 1, 0);
      const float4 v0 = a.rec[min(p0 + c, last)];
      const float4 v1 = a.rec[min(p0 + c + 16, last)];
      (void)cnt;
      r0 = v0;      // raw: the masks (c < cnt, 16 + c < cnt) are applied where the records are used — a select here
      r1 = v1;      // would wait for the loads at once, a full HBM round trip at the end of every tile
    };
    // range of the sources of one tile's records -> ctl (an atom with more than 32 records: too wide by decree)
    auto range_of = [&](int cnt, const float4& r0, const float4& r1, int* ctl_t) {
      int lo = 0x7fffffff, hi = -1;
      if (c < cnt) { lo = hi = __builtin_bit_cast(int, r0.x); }
      if (16 + c < cnt) { const int s1 = __builtin_bit_cast(int, r1.x); lo = min(lo, s1); hi = max(hi, s1); }
      if (cnt > 32) { lo = 0; hi = 0x3fffffff; }
      lo = wave_min_i32(lo);
      hi = -wave_min_i32(-hi);
      if (lane == 63) { ctl_t[wave] = lo; ctl_t[NW + wave] = hi; }
    };
    float4 p_h, p_dH;
    auto issue_h = [&](int64_t t) {        // h chunk (row al, column c) of tile t
      const int nvalid = (int)std::min<int64_t>(a.N - t * WTA, WTA);
      const float* hb = a.h + t * (WTA * WF);
      const float4 v = *reinterpret_cast<const float4*>(hb + min(al, nvalid - 1) * WF + 4 * c);
      p_h = al < nvalid ? v : f4zero();
    };
    auto issue_dH = [&](int64_t t) {       // dH chunk of this lane's dh block
      const int nvalid = (int)std::min<int64_t>(a.N - t * WTA, WTA);
      const float* db = a.dH + t * (WTA * WF);
      p_dH = *reinterpret_cast<const float4*>(db + min(16 * rt_own + a16, nvalid - 1) * WF + 16 * ct + 4 * g4);
    };

    // one record set in flight: the records of tile t + 1 are requested into the registers of tile t's as soon as its
    // gather is done (their CSC offsets arrived a tile earlier); offsets: current, next, the one after
    int c_p0, c_cnt, n_p0, n_cnt, nn_p0, nn_cnt;
    float4 r0, r1;
    load_ptr(T0, c_p0, c_cnt);
    load_ptr(T0 + 1 < T1 ? T0 + 1 : T0, n_p0, n_cnt);
    load_rec(c_p0, c_cnt, r0, r1);
    issue_h(T0);
    range_of(c_cnt, r0, r1, ctl + (T0 & 1) * (2 * NW));
    int wlo = -(1 << 30), mode = 0;
    NG_LDS_BARRIER();
    if (win_decide(ctl + (T0 & 1) * (2 * NW), wlo, mode)) {
      win_dma(win, a.dP, wlo, a.N, wave, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    NG_LDS_BARRIER();

#pragma unroll 1
    for (int64_t t = T0; t < T1; ++t) {
      // ---- vector interval: h rows of t into LDS, range of t+1 published, offsets of t+2 requested, B rows of t gathered
      W16_T(0);
      if (H2) {
        // h row (16 lanes = one row) into two fp16 planes; a power-of-two row scale only when the row reaches 2^15
        float m = fmaxf(fmaxf(fabsf(p_h.x), fabsf(p_h.y)), fmaxf(fabsf(p_h.z), fabsf(p_h.w)));
        int hb = 127, hmw = 127;
        float4 hs = p_h;
        if (__builtin_amdgcn_ballot_w64(m >= 32768.0f) != 0) {      // wave-uniform, never taken for ordinary activations
          m = fmaxf(m, ror_f<8>(m)); m = fmaxf(m, ror_f<4>(m)); m = fmaxf(m, ror_f<2>(m)); m = fmaxf(m, ror_f<1>(m));
          const int hef = (__builtin_bit_cast(int, m) >> 23) & 255;
          const bool big = hef >= 127 + 15 && hef != 255;
          hb = big ? 268 - hef : 127;
          const float s = __builtin_bit_cast(float, hb << 23);
          hs = make_float4(p_h.x * s, p_h.y * s, p_h.z * s, p_h.w * s);
          hmw = wave_min_i32(hb);      // (valid in lane 63)
        }
        if (c == 0) s_hb[al] = hb;
        if (lane == 63) s_hmin[wave] = hmw;
        unsigned h0, l0, h1, l1;
        split2_pair(hs.x, hs.y, h0, l0); split2_pair(hs.z, hs.w, h1, l1);
        char* q = hpl + al * HROWB + 8 * c;
        *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(q + HPLANE) = u32x2{l0, l1};
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(hpl) + al * SDP_LD + 4 * c) = p_h;
      }
      issue_h(t + 1 < T1 ? t + 1 : t);      // (a whole tile ahead of its use)
      load_ptr(t + 2 < T1 ? t + 2 : t, nn_p0, nn_cnt);
      W16_T(1);
      if (mode == 0) gather<E, H2, false>(wave, lane, al, wlo, c_p0, c_cnt, r0, r1, a.rec, tile, win4, src4, s_rs, s_sb, s_wmin);
      else gather_global<E, H2>(wave, lane, al, c_p0, c_cnt, r0, r1, a.rec, tile, src4, s_rs, s_sb, s_wmin);
      load_rec(n_p0, n_cnt, r0, r1);      // records of t + 1
      W16_T(2);
      NG_LDS_BARRIER();
      W16_T(3);
      issue_dH(t);      // (for this tile's dh epilogue, two barriers away)
      // ---- dw += h^T B: l-tile lt, column tiles E cg + u; the contraction over the tile's atoms in two 32-deep steps
      if (H2) {
        int sbref = s_wmin[0], hbref = s_hmin[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) { sbref = min(sbref, s_wmin[i]); hbref = min(hbref, s_hmin[i]); }
        // reference exponent of the step's common scale: <= every row's (sb + hb); 1 / (S_ref Sh_ref) for the sums
        const int eref = sbref + hbref;      // biased by 2 * 127
        const float inv_ref = __builtin_bit_cast(float, (254 - sbref) << 23) * __builtin_bit_cast(float, (254 - hbref) << 23);
#pragma unroll 1
        for (int step = 0; step < 2; ++step) {
          // per row of the step a power of two 2^(eref - sb_r - hb_r) <= 1 as an fp16 pair for the rows (2j, 2j + 1) of this
          // lane's eight k-slots (all-zero B rows carry sb = 253: their factor underflows to 0, their pieces are zeros anyway)
          unsigned rpk[4];
          {
            const int r0 = 32 * step + 8 * g4;
            const int4 sa = *reinterpret_cast<const int4*>(s_sb + r0), sc = *reinterpret_cast<const int4*>(s_sb + r0 + 4);
            const int4 ha = *reinterpret_cast<const int4*>(s_hb + r0), hc = *reinterpret_cast<const int4*>(s_hb + r0 + 4);
            const int ex[8] = {sa.x + ha.x, sa.y + ha.y, sa.z + ha.z, sa.w + ha.w, sc.x + hc.x, sc.y + hc.y, sc.z + hc.z, sc.w + hc.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int e0 = max(eref - ex[2 * j] + 15, 0), e1 = max(eref - ex[2 * j + 1] + 15, 0);      // fp16 exponent field (0: zero)
              rpk[j] = (unsigned)(e0 << 10) | ((unsigned)(e1 << 10) << 16);
            }
          }
          // A operand: h pieces of l-tile lt over the step's rows 8 g4 .. + 7: two transposing reads of four rows per plane
          const char* hp = hpl + (32 * step + 8 * g4 + (a16 >> 2)) * HROWB + (16 * lt + 4 * (a16 & 3)) * 2;
          u32x4 ah, al4;
          {
            const gs16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)hp);
            const gs16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(hp + 4 * HROWB));
            const gs16x4 w0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(hp + HPLANE));
            const gs16x4 w1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(hp + HPLANE + 4 * HROWB));
            const u32x2 a0 = __builtin_bit_cast(u32x2, v0), a1 = __builtin_bit_cast(u32x2, v1);
            const u32x2 c0 = __builtin_bit_cast(u32x2, w0), c1 = __builtin_bit_cast(u32x2, w1);
            ah = u32x4{a0[0], a0[1], a1[0], a1[1]}; al4 = u32x4{c0[0], c0[1], c1[0], c1[1]};
          }
          const char* bp = reinterpret_cast<const char*>(tile) + (32 * step + 8 * g4 + (a16 >> 2)) * Tile<E>::ROWB + (16 * (E * cg) + 4 * (a16 & 3)) * 2;
          // the step's B reads for all E column tiles first, then the scaling and the products: a tile at a time left every
          // transposing read's latency exposed
          gs16x4 bv[E][4];
#pragma unroll
          for (int u = 0; u < E; ++u) {
            const char* q0 = bp + 32 * u;
            bv[u][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)q0);
            bv[u][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(q0 + 4 * Tile<E>::ROWB));
            bv[u][2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(q0 + Tile<E>::PLANE));
            bv[u][3] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(q0 + Tile<E>::PLANE + 4 * Tile<E>::ROWB));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < E; ++u) {
            const u32x2 a0 = __builtin_bit_cast(u32x2, bv[u][0]), a1 = __builtin_bit_cast(u32x2, bv[u][1]);
            const u32x2 c0 = __builtin_bit_cast(u32x2, bv[u][2]), c1 = __builtin_bit_cast(u32x2, bv[u][3]);
            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
            auto sc = [&](unsigned v, unsigned r) {
              return __builtin_bit_cast(unsigned, (h2v)(__builtin_bit_cast(h2v, v) * __builtin_bit_cast(h2v, r)));
            };
            const u32x4 bh = {sc(a0[0], rpk[0]), sc(a0[1], rpk[1]), sc(a1[0], rpk[2]), sc(a1[1], rpk[3])};
            const u32x4 bl = {sc(c0[0], rpk[0]), sc(c0[1], rpk[1]), sc(c1[0], rpk[2]), sc(c1[1], rpk[3])};
            f32x4 at = {0.f, 0.f, 0.f, 0.f};
            at = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al4), __builtin_bit_cast(f16x8, bh), at, 0, 0, 0);
            at = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bl), at, 0, 0, 0);
            at = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), at, 0, 0, 0);
            accW[u] += at * inv_ref;
          }
        }
      } else {
        // D[i = l][j = (n,m)] += sum_atoms h[atom][l] B[atom][(n,m)] with f32-input MFMAs (weights beyond the piece range)
        const float* ha = reinterpret_cast<const float*>(hpl) + 16 * lt + a16;
        const float* bb = tile + 16 * (E * cg) + a16;
#pragma unroll 4
        for (int T = 0; T < 16; ++T) {
          const int ro = (T & 3) + 4 * g4 + 16 * (T >> 2);
          const float av = ha[ro * SDP_LD];
#pragma unroll
          for (int u = 0; u < E; ++u)
            accW[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bb[ro * LD + 16 * u], accW[u], 0, 0, 0);
        }
      }
      // the range of the next tile's sources: its records were requested at the end of the previous tile and have had this
      // tile's gather and matrix interval to arrive
      W16_T(4);
      // ---- matrix interval, first part: this wave's k-half of its two dh blocks; the partial it does not finish goes to LDS
      f32x4 pd[2];
      if (H2) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const char* xrow = reinterpret_cast<const char*>(tile) + (16 * (2 * rp + hh) + a16) * Tile<E>::ROWB + 16 * g4 + 64 * (kh * NTH);
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
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
          pd[hh] = acc0 + acc1;
        }
      } else {
        const float4* p32 = reinterpret_cast<const float4*>(a.WfragN32) + (size_t)(ct * NT + kh * NT_H) * 64 + lane;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const float* xrow = tile + (16 * (2 * rp + hh) + a16) * LD + 4 * g4 + 16 * (kh * NT_H);
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
          pd[hh] = acc;
        }
      }
      {
        const f32x4 o = kh ? pd[0] : pd[1];
        xch[((2 * rp + (1 - kh)) * 4 + ct) * 64 + lane] = make_float4(o[0], o[1], o[2], o[3]);
      }
      W16_T(5);
      if (t + 1 < T1) range_of(n_cnt, r0, r1, ctl + ((t + 1) & 1) * (2 * NW));
      NG_LDS_BARRIER();
      W16_T(6);
      // the next tile's window, when it needs one (no wave reads the window before the gather behind the tile's last barrier)
      bool restage = false;
      if (t + 1 < T1) restage = win_decide(ctl + ((t + 1) & 1) * (2 * NW), wlo, mode);
      if (restage) win_dma(win, a.dP, wlo, a.N, wave, lane);
      // ---- dh of this wave's own block: (k-half 0) + (k-half 1), the row's 2^-8 / S, the residual term
      {
        const float4 q = xch[(rt_own * 4 + ct) * 64 + lane];
        const f32x4 mine = kh ? pd[1] : pd[0];
        const f32x4 k0 = kh ? f32x4{q.x, q.y, q.z, q.w} : mine, k1 = kh ? mine : f32x4{q.x, q.y, q.z, q.w};
        const float osc = H2 ? s_rs[16 * rt_own + a16] : 1.0f;
        const int nvalid = (int)std::min<int64_t>(a.N - t * WTA, WTA);
        const float4 v = make_float4(fmaf(k0[0] + k1[0], osc, p_dH.x), fmaf(k0[1] + k1[1], osc, p_dH.y),
                                     fmaf(k0[2] + k1[2], osc, p_dH.z), fmaf(k0[3] + k1[3], osc, p_dH.w));
        float* ob = a.dh + t * (WTA * WF);
        const int rl = 16 * rt_own + a16;
        *reinterpret_cast<float4*>(rl < nvalid ? ob + rl * WF + 16 * ct + 4 * g4 : a.dummy + 16 * ct + 4 * g4) = v;
      }
      c_p0 = n_p0; c_cnt = n_cnt;
      n_p0 = nn_p0; n_cnt = nn_cnt;
      if (restage) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      NG_LDS_BARRIER();
      W16_T(7);
    }
  }
  // ---- dw partial of this workgroup, layout [(n,m)][l]: lane holds l = 16 lt + 4 g4 + r, column = 16 (E cg + u) + a16
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const int cidx = 16 * (E * cg + u) + a16;
    *reinterpret_cast<float4*>(part + cidx * WF + 16 * lt + 4 * g4) = make_float4(accW[u][0], accW[u][1], accW[u][2], accW[u][3]);
  }
}

template <int E>
__global__ __launch_bounds__(WTHREADS) void mp_win16_bwd_node_kernel(Args a) {
  if (a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag))) body<E, false>(a);
  else body<E, true>(a);
}

}  // namespace w16n

static size_t mp_win16_node_lds(int E) {
  const size_t tile = E == 1 ? w16n::Tile<1>::BYTES : (E == 2 ? w16n::Tile<2>::BYTES : w16n::Tile<3>::BYTES);
  return (size_t)w16n::WROWS * w16n::WF * 4 + tile + w16n::H_BYTES + 16 * 64 * 16 + (3 * w16n::WTA + 2 * w16n::NW + 4 * w16n::NW) * 4;
}

bool mp_win16_bwd_node_supported(int E) { return E >= 1 && E <= 3; }

// the caller (mp_win_bwd_node) reduces the partials: one [E*64*64] block per workgroup, grid <= num_cu
int mp_win16_bwd_node_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int E, const float* h, const float* dP, const int32_t* csc_ptr,
                             const float* rec, const float* WfragN, const float* dh_out, float* dh_in, float* scratch, float* dummy,
                             RangeGuard guard, const float* WfragN32, const unsigned* wflag, unsigned wflag_ver, int* grid_out) {
  using namespace w16n;
  Args a{};
  a.N = N; a.ntiles = cdiv(N, WTA);
  // contiguous runs of tiles per workgroup: multiples of 4 (256 atoms) when the batch is large enough (ng_internal.h)
  const int64_t per = win16_tiles_per_wg(a.ntiles, ctx->num_cu);
  a.tiles_per_wg = (int)per;
  a.dP = dP; a.dH = dh_out; a.h = h; a.csc_ptr = csc_ptr; a.rec = reinterpret_cast<const float4*>(rec);
  a.WfragN = WfragN; a.WfragN32 = WfragN32; a.dh = dh_in; a.partial = scratch; a.dummy = dummy;
  a.guard = guard; a.wflag = wflag; a.wflag_ver = wflag_ver;
  const int grid = (int)cdiv(a.ntiles, per);
  *grid_out = grid;
  const size_t lds = mp_win16_node_lds(E);
#ifdef W16_STAMP
  static unsigned long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { (void)hipMalloc(&dbg, 4 * 16 * 8 * 8); (void)hipMemset(dbg, 0, 4 * 16 * 8 * 8); }
  a.stamps = dbg;
#endif
  ProfScope ps(ctx, st, "mp_win_bwd_node");
  switch (E) {
    case 1: hipLaunchKernelGGL((mp_win16_bwd_node_kernel<1>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 2: hipLaunchKernelGGL((mp_win16_bwd_node_kernel<2>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 3: hipLaunchKernelGGL((mp_win16_bwd_node_kernel<3>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
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
        fprintf(stderr, "W16N tile %d wave %2d: stage %5lld  gather %5lld  bar1 %5lld  dh %5lld  dw %5lld  range+bar2 %5lld  tail+bar3 %5lld | total %6lld\n",
                tt, w, (long long)(p[1] - p[0]), (long long)(p[2] - p[1]), (long long)(p[3] - p[2]), (long long)(p[4] - p[3]),
                (long long)(p[5] - p[4]), (long long)(p[6] - p[5]), (long long)(p[7] - p[6]), (long long)(p[7] - p[0]));
      }
  }
#endif
  return NG_OK;
}

}  // namespace ng
