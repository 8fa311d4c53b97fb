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

namespace ng {

namespace {

constexpr int WF = 64;
constexpr int WTA = 32;
constexpr int WROWS = 288;
constexpr int WC4 = WF / 4;
constexpr int WTHREADS = 512;
constexpr int SDP_LD = 68;      // dP / h tile row stride (== 4 mod 16: conflict-free K-strided reads)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
__device__ __noinline__ void edge_dot_global(int K, int wave, int lane, const int32_t* nl, const float* tb,
                                             int ld, const float4* src4, float* out3) {
  float out[E];
  edge_dot<E, 1>(K, wave, lane, 0, nl, tb, ld, nullptr, src4, out);
#pragma unroll
  for (int n = 0; n < E; ++n) out3[n] = out[n];
}

template <int E>
__global__ __launch_bounds__(WTHREADS, 1) void mp_win_bwd_edge_kernel(MpWinEdgeArgs a) {
  constexpr int KF = E * WF;
  constexpr int LD = KF + 4;
  constexpr int NCT = KF / 16 / 4;          // column tiles per wave (4 waves per atom half)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* win = smem;                                                   // [WROWS][64]
  float* tile = win + WROWS * WF;                                      // [32][LD]   dA
  float* sdp = tile + WTA * LD;                                        // [2][32][SDP_LD]
  int32_t* s_nl = reinterpret_cast<int32_t*>(sdp + 2 * WTA * SDP_LD);  // [2][32*K]
  int* ctl = reinterpret_cast<int*>(s_nl + 2 * WTA * a.K);             // [2][16]

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

  // weight fragments: this wave's NCT column tiles of dA (o = 16*ct + ...), contraction over m (4 k-steps)
  const int hh = wave >> 2, ct0 = (wave & 3) * NCT;
  float wf[NCT][16];
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
    float* dp = sdp + (t & 1) * WTA * SDP_LD;
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
    *reinterpret_cast<float4*>(dp + prow * SDP_LD + 4 * pc) = g;
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
    if (win_decide(ctl + (t & 1) * 16, wlo, mode)) win_stage(win4, src4, wlo, a.N, tid);
    // ---- matrix interval: dA tile = dP tile x Wp^T
    {
      const float* xrow = sdp + (t & 1) * WTA * SDP_LD + (16 * hh + a16) * SDP_LD + 4 * g4;
      float4 x[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) x[T] = *reinterpret_cast<const float4*>(xrow + 16 * T);
      f32x4 acc[NCT];
#pragma unroll
      for (int u = 0; u < NCT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
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
      // lane holds dA[atom 16hh + a16][o = 16(ct0+u) + 4*g4 + (0..3)]
#pragma unroll
      for (int u = 0; u < NCT; ++u)
        *reinterpret_cast<float4*>(tile + (16 * hh + a16) * LD + 16 * (ct0 + u) + 4 * g4) =
            make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
    }
    NG_LDS_BARRIER();
    // ---- vector interval: de of tile t, then dP / lists of tile t+1 into LDS, requests for t+2
    {
      float out[E];
      const int32_t* nl = s_nl + (t & 1) * per_tile;
      if (mode == 0) edge_dot<E, 0>(K, wave, lane, wlo, nl, tile, LD, win4, src4, out);
      else edge_dot_global<E>(K, wave, lane, nl, tile, LD, src4, out);
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

size_t edge_lds_bytes(int K, int E) {
  return (size_t)(WROWS * WF + WTA * (E * WF + 4) + 2 * WTA * SDP_LD + 2 * WTA * K + 32) * 4;
}

}  // namespace

bool mp_win_bwd_supported(int F, int E, int K) {
  return F == WF && E >= 1 && E <= 3 && K % 4 == 0 && K >= 4 && K <= 16;
}

int mp_win_bwd_edge(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, const float* h,
                    const int32_t* nlist, const float* inv_degree, const float* WfragT, const float* s_save,
                    const float* dh_out, float* dP, float* de, int de_accum, float* dummy) {
  MpWinEdgeArgs a{};
  a.N = N; a.K = K; a.ntiles = cdiv(N, WTA);
  int64_t per = cdiv(a.ntiles, ctx->num_cu);
  per = cdiv(per, 8) * 8;
  a.tiles_per_wg = (int)per;
  a.dH = dh_out; a.S = act == NG_ACT_NONE ? nullptr : s_save; a.rowscale = inv_degree; a.h = h;
  a.nlist = nlist; a.WfragT = WfragT; a.dP = dP; a.de = de; a.dummy = dummy; a.act = act;
  a.accumulate = de_accum;
  const int grid = (int)cdiv(a.ntiles, per);
  const size_t lds = edge_lds_bytes(K, E);
  ProfScope ps(ctx, st, "mp_win_bwd_edge");
  switch (E) {
    case 1: hipLaunchKernelGGL((mp_win_bwd_edge_kernel<1>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 2: hipLaunchKernelGGL((mp_win_bwd_edge_kernel<2>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
    case 3: hipLaunchKernelGGL((mp_win_bwd_edge_kernel<3>), dim3(grid), dim3(WTHREADS), lds, st, a); break;
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

}  // namespace ng
