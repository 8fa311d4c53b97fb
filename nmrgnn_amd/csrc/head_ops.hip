// Bandwidth-shaped versions of the small node-side kernels around the hot path: embedding weight
// gradient, output head forward / backward.  Reference: nmrgnn/model.py:239-243,262,266-273.
//
// All of them stream [N, F] activations once (whole 16-byte vectors, a row's lanes side by side) and
// keep the C-wide (number of elements <= 32) one-hot / standardisation arithmetic in registers:
//   head   peaks_i = <x_i, u_i> + v_i,   u_i[f] = sum_c a_ic std_c Wout[f][c],  v_i = sum_c a_ic (std_c b_c + avg_c)
//          dg_i[f] = mask * dpeaks_i * u_i[f]
//          dWout[f][c] = sum_i x_i[f] dpeaks_i a_ic std_c,   dbout[c] = sum_i dpeaks_i a_ic std_c
//   embed  dWemb[c][f] = sum_i a_ic dh0_i[f]
// `a` is read as general floats (the reference feeds one-hot rows but does not require it).
// Weight-gradient sums are two-stage and deterministic: per-workgroup partials, then reduce_z.
#include <algorithm>
#include <cstdlib>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "reduce.cuh"
#include "rng.cuh"

namespace ng {

constexpr int HC_MAX = 32;     // one-hot width limit

// quad / row reductions on the DPP network
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over aligned groups of 8 lanes (valid in every lane of the group)
__device__ __forceinline__ float sum8(float v) {
  v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);      // row_half_mirror: lane i <-> 7 - i within each half row
  return v;
}

// ---- head forward: LPR lanes per row, one float4 of the Fh features each --------------------------------
// draw: the keep-mask is drawn here (the values of dropout_mask_kernel(seed, offset, keep) over the [N][Fh] elements: a
// lane's four columns are one Philox counter) and written to mask_out for the backward, instead of being read
struct HeadDraw { uint64_t seed, offset; float keep; float* mask_out; const uint64_t* staged; };      // staged: the seed of a replayed step (ng_replay_stage)

template <int LPR, bool DRAW>
__global__ __launch_bounds__(256) void head_fwd_fast_kernel(int64_t N, int C, const float* __restrict__ g,
                                                            const float* __restrict__ mask,
                                                            const float* __restrict__ Wout,
                                                            const float* __restrict__ bout,
                                                            const float* __restrict__ atoms,
                                                            const float* __restrict__ pstd,
                                                            const float* __restrict__ pavg,
                                                            float* __restrict__ peaks, HeadDraw dr) {
  constexpr int Fh = LPR * 4;
  __shared__ __attribute__((aligned(16))) float sWs[Fh * HC_MAX];     // [c][f]: std_c * Wout[f][c] (a lane's four f are one 16-byte read)
  __shared__ float sV[HC_MAX];           // std_c * b_c + avg_c
  for (int t = threadIdx.x; t < Fh * C; t += 256) sWs[(t % C) * Fh + t / C] = Wout[t] * pstd[t % C];
  if (threadIdx.x < C) sV[threadIdx.x] = pstd[threadIdx.x] * bout[threadIdx.x] + pavg[threadIdx.x];
  __syncthreads();
  const int q = threadIdx.x % LPR;
  const int64_t rows_per_pass = (int64_t)gridDim.x * (256 / LPR);
  for (int64_t i = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR; i < N; i += rows_per_pass) {
    float4 x = *reinterpret_cast<const float4*>(g + i * Fh + 4 * q);
    if (DRAW) {
      uint32_t r[4];
      philox4x32(dr.staged ? dr.staged[0] : dr.seed, dr.offset + (uint64_t)(i * LPR + q), r);
      const float inv = 1.0f / dr.keep;
      float4 m = make_float4(u01(r[0]) <= dr.keep ? inv : 0.f, u01(r[1]) <= dr.keep ? inv : 0.f,
                             u01(r[2]) <= dr.keep ? inv : 0.f, u01(r[3]) <= dr.keep ? inv : 0.f);
      *reinterpret_cast<float4*>(dr.mask_out + i * Fh + 4 * q) = m;
      // opaque from here on, like a loaded mask: folded into the select the products below were contracted differently
      // from the mask-reading form (peaks 1 ulp apart)
      asm volatile("" : "+v"(m.x), "+v"(m.y), "+v"(m.z), "+v"(m.w));
      x.x *= m.x; x.y *= m.y; x.z *= m.z; x.w *= m.w;
    } else if (mask) {
      const float4 m = *reinterpret_cast<const float4*>(mask + i * Fh + 4 * q);
      x.x *= m.x; x.y *= m.y; x.z *= m.z; x.w *= m.w;
    }
    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f, v = 0.f;
    for (int c = 0; c < C; ++c) {
      const float a = atoms[i * C + c];
      const float4 w4 = *reinterpret_cast<const float4*>(sWs + c * Fh + 4 * q);
      u0 += a * w4.x; u1 += a * w4.y;
      u2 += a * w4.z; u3 += a * w4.w;
      v += a * sV[c];
    }
    // explicit fused chain: left to the compiler the two instantiations contracted this sum differently (1 ulp apart)
    float p = fmaf(x.w, u3, fmaf(x.z, u2, fmaf(x.y, u1, x.x * u0)));
    p = sum8(p);
    if (LPR >= 16) p += __shfl_xor(p, 8, 64);
    if (LPR == 32) p += __shfl_xor(p, 16, 64);
    if (q == 0) peaks[i] = p + v;
  }
}

// ---- head backward: dg + per-workgroup partials of [dWout ; dbout] ---------------------------------------
// thread = (row lane r = tid / LPR, column lane q); accumulators acc[c][4] for its four f's, plus db[c] on q == 0
template <int LPR, int CM>
__global__ __launch_bounds__(256) void head_bwd_fast_kernel(int64_t N, int C, int64_t rows_per_block,
                                                            const float* __restrict__ g,
                                                            const float* __restrict__ mask,
                                                            const float* __restrict__ Wout,
                                                            const float* __restrict__ atoms,
                                                            const float* __restrict__ pstd,
                                                            const float* __restrict__ dpeaks,
                                                            float* __restrict__ dg, float* __restrict__ partial) {
  constexpr int Fh = LPR * 4;
  constexpr int RL = 256 / LPR;          // row lanes
  __shared__ __attribute__((aligned(16))) float sWs[Fh * HC_MAX];     // [c][f], as in the forward
  __shared__ float sStd[HC_MAX];
  extern __shared__ __attribute__((aligned(16))) float red[];     // [RL][Fh*C + C] for the final sum
  for (int t = threadIdx.x; t < Fh * C; t += 256) sWs[(t % C) * Fh + t / C] = Wout[t] * pstd[t % C];
  if (threadIdx.x < C) sStd[threadIdx.x] = pstd[threadIdx.x];
  __syncthreads();
  const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
  float acc[CM][4];
  float db[CM];
#pragma unroll
  for (int c = 0; c < CM; ++c) { acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f; db[c] = 0.f; }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = std::min<int64_t>(r0 + rows_per_block, N);
  // two rows per trip: both rows' loads (features, mask, dpeaks, one-hot row) are requested before either row is
  // worked on (one row per trip ran a memory round trip per row); the sums keep their row order
  auto row_body = [&](int64_t i, float4 x, const float4& m, float dp, const float (&av)[CM]) {
    x.x *= m.x; x.y *= m.y; x.z *= m.z; x.w *= m.w;
    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) {
      if (c < C) {
        const float a = av[c];
        const float4 w4 = *reinterpret_cast<const float4*>(sWs + c * Fh + 4 * q);
        u0 += a * w4.x; u1 += a * w4.y;
        u2 += a * w4.z; u3 += a * w4.w;
        const float d = dp * a * sStd[c];            // dfull[i][c]
        acc[c][0] += x.x * d; acc[c][1] += x.y * d; acc[c][2] += x.z * d; acc[c][3] += x.w * d;
        db[c] += d;
      }
    }
    *reinterpret_cast<float4*>(dg + i * Fh + 4 * q) = make_float4(m.x * dp * u0, m.y * dp * u1, m.z * dp * u2,
                                                                  m.w * dp * u3);
  };
  for (int64_t i = r0 + r; i < r1; i += 2 * RL) {
    const int64_t i2 = i + RL;
    const bool two = i2 < r1;
    const int64_t j2 = two ? i2 : i;
    const float4 xa = *reinterpret_cast<const float4*>(g + i * Fh + 4 * q);
    const float4 xb = *reinterpret_cast<const float4*>(g + j2 * Fh + 4 * q);
    float4 ma = make_float4(1.f, 1.f, 1.f, 1.f), mb = ma;
    if (mask) {
      ma = *reinterpret_cast<const float4*>(mask + i * Fh + 4 * q);
      mb = *reinterpret_cast<const float4*>(mask + j2 * Fh + 4 * q);
    }
    const float dpa = dpeaks[i], dpb = dpeaks[j2];
    float aa[CM], ab[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
      aa[c] = c < C ? atoms[i * C + c] : 0.f;
      ab[c] = c < C ? atoms[j2 * C + c] : 0.f;
    }
    row_body(i, xa, ma, dpa, aa);
    if (two) row_body(i2, xb, mb, dpb, ab);
  }
  // sum over the row lanes through LDS, one partial per workgroup: layout [f*C + c] then [Fh*C + c]
  const int items = Fh * C + C;
#pragma unroll
  for (int c = 0; c < CM; ++c) {
    if (c < C) {
#pragma unroll
      for (int s = 0; s < 4; ++s) red[r * items + (4 * q + s) * C + c] = acc[c][s];
      if (q == 0) red[r * items + Fh * C + c] = db[c];
    }
  }
  __syncthreads();
  for (int it = threadIdx.x; it < items; it += 256) {
    float s = 0.f;
    for (int rr = 0; rr < RL; ++rr) s += red[rr * items + it];
    partial[(int64_t)blockIdx.x * items + it] = s;
  }
}

// ---- head forward + per-graph L2 loss + head backward in ONE launch (round 6; nmrgnn/model.py:266-273, losses.py:30-39 with s = 1)
// The loss of a graph needs the graph's peaks and nothing else, and the head's backward needs dloss/dpeaks and nothing else: a
// workgroup that owns whole graphs can run the three steps back to back with the peaks / their gradient in LDS.  Row arithmetic is
// that of head_fwd_fast_kernel / loss_graph_kernel / head_bwd_fast_kernel (peaks, per-graph losses, dpeaks and dg carry the same
// bits; the weight-gradient partials are cut at graph boundaries instead of row counts, so dWout / dbout differ in the last bits).
// The keep-mask is drawn twice from the same Philox counters instead of being stored and read back.  The mean over graphs is one
// more column of the workgroup's partial row (its graphs' losses / G), summed over the workgroups by the second stage that the
// weight-gradient partials need anyway (ng_head_loss_reduce): the loss is complete when that has run.  (A first form let the
// workgroup that finishes last take the mean: the two agent-scope fences per workgroup wrote back the XCD's L2 — 16.8 MB of
// freshly stored dg — 512 times: 123 us at the bench shape.)
struct HeadLossArgs {
  int64_t N;
  int G, C, gpw;                 // graphs per workgroup
  const float *g, *Wout, *bout, *atoms, *pstd, *pavg;
  const int32_t* gptr;
  const float *y, *w;
  float gweight;                 // shard weight of the rank's loss gradient (parallel.shard_grad_weight); 1 = none
  float* peaks;
  float* dg;
  float* partial;                // [grid][Fh*C + C + 1]: [dWout ; dbout ; sum of the workgroup's graph losses / G]
  HeadDraw dr;                   // mask_out may be nullptr (nobody reads the mask after this launch)
};
constexpr int HL_ROWS = 256;     // rows (atoms) a workgroup can own — a graph of the reference's training set; HL_CM: one-hot width limit
constexpr int HL_CM = 16, HL_NT = 512;

// Every input of the three steps is requested in the prologue — the rows' features stay in registers (RPT float4 per thread), the
// one-hot rows, labels and weights go to LDS — so the launch is ONE memory round trip of loads long.  (A first form walked the rows
// in trips of 32, eight dependent trips per phase: 84 us at the bench shape against 44 for the three separate launches; a loop form
// for longer graphs lost to the separate launches too and is gone: such shapes report "not supported".)  137-204 registers at 512
// threads: one workgroup per CU, so the launch pays only while the workgroups fit the chip in one round (head_loss_graphs_per_wg).
template <int LPR, bool DRAW>
__global__ __launch_bounds__(HL_NT) void head_loss_kernel(HeadLossArgs a) {
  constexpr int Fh = LPR * 4, CM = HL_CM, NT = HL_NT;
  constexpr int RL = NT / LPR;          // rows per trip
  constexpr int NW = NT / 64;
  constexpr int RPT = HL_ROWS / RL;
  __shared__ __attribute__((aligned(16))) float sWs[Fh * HL_CM];
  __shared__ float sV[HL_CM], sStd[HL_CM];
  __shared__ float s_pk[HL_ROWS], s_y[HL_ROWS], s_w[HL_ROWS];
  __shared__ float s_red[NW];
  extern __shared__ __attribute__((aligned(16))) float red[];     // [NW][Fh*C + C], then the one-hot rows [HL_ROWS][C]
  const int C = a.C;
  const int items = Fh * C + C;
  float* s_at = red + NW * items;
  const float* __restrict__ gin = a.g;
  float* __restrict__ peaks = a.peaks;
  float* __restrict__ dgo = a.dg;
  const int g0 = blockIdx.x * a.gpw, g1 = min(a.G, g0 + a.gpw);
  const int64_t r0 = a.gptr[g0], r1 = a.gptr[g1];
  int nrows = (int)(r1 - r0);            // <= HL_ROWS by the host's choice of gpw from max_graph_atoms ...
  const bool overlong = nrows > HL_ROWS;       // ... unless the caller's max_graph_atoms understated graph_ptr: stay inside the LDS
  if (overlong) nrows = HL_ROWS;               // arrays and poison the loss (the rows beyond are left unwritten)
  const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
  float4 xr[RPT];
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    const int rr = r + u * RL;
    xr[u] = *reinterpret_cast<const float4*>(gin + (r0 + (rr < nrows ? rr : 0)) * Fh + 4 * q);
  }
  for (int t = threadIdx.x; t < nrows * C; t += NT) s_at[t] = a.atoms[r0 * C + t];
  for (int t = threadIdx.x; t < nrows; t += NT) { s_y[t] = a.y[r0 + t]; s_w[t] = a.w[r0 + t]; }
  for (int t = threadIdx.x; t < Fh * C; t += NT) sWs[(t % C) * Fh + t / C] = a.Wout[t] * a.pstd[t % C];
  if (threadIdx.x < C) {
    sV[threadIdx.x] = a.pstd[threadIdx.x] * a.bout[threadIdx.x] + a.pavg[threadIdx.x];
    sStd[threadIdx.x] = a.pstd[threadIdx.x];
  }
  __syncthreads();
  const float inv_keep = 1.0f / a.dr.keep;
  auto draw = [&](int64_t i) {
    float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
    if (DRAW) {
      uint32_t rr[4];
      philox4x32(a.dr.staged ? a.dr.staged[0] : a.dr.seed, a.dr.offset + (uint64_t)(i * LPR + q), rr);
      m = make_float4(u01(rr[0]) <= a.dr.keep ? inv_keep : 0.f, u01(rr[1]) <= a.dr.keep ? inv_keep : 0.f,
                      u01(rr[2]) <= a.dr.keep ? inv_keep : 0.f, u01(rr[3]) <= a.dr.keep ? inv_keep : 0.f);
      asm volatile("" : "+v"(m.x), "+v"(m.y), "+v"(m.z), "+v"(m.w));      // opaque, as in head_fwd_fast_kernel
    }
    return m;
  };
  // ---- peaks
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    const int rr = r + u * RL;
    if (rr < nrows) {
      const int64_t i = r0 + rr;
      const float* av = s_at + rr * C;
      float4 x = xr[u];
      if (DRAW) {
        const float4 m = draw(i);
        if (a.dr.mask_out) *reinterpret_cast<float4*>(a.dr.mask_out + i * Fh + 4 * q) = m;
        x.x *= m.x; x.y *= m.y; x.z *= m.z; x.w *= m.w;
      }
      float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f, v = 0.f;
      for (int c = 0; c < C; ++c) {
        const float aa = av[c];
        const float4 w4 = *reinterpret_cast<const float4*>(sWs + c * Fh + 4 * q);
        u0 += aa * w4.x; u1 += aa * w4.y;
        u2 += aa * w4.z; u3 += aa * w4.w;
        v += aa * sV[c];
      }
      float p = fmaf(x.w, u3, fmaf(x.z, u2, fmaf(x.y, u1, x.x * u0)));
      p = sum8(p);
      if (LPR >= 16) p += __shfl_xor(p, 8, 64);
      if (LPR == 32) p += __shfl_xor(p, 16, 64);
      if (q == 0) { peaks[i] = p + v; s_pk[rr] = p + v; }
    }
  }
  __syncthreads();
  // ---- loss of each graph and dloss / dpeaks: one wave per graph, loss_graph_kernel's order
  {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float wl = 0.f;
    for (int gi = g0 + wv; gi < g1; gi += NW) {
      const int ga = min(a.gptr[gi] - (int)r0, nrows), gb = min(a.gptr[gi + 1] - (int)r0, nrows);
      float sw = 0.f, sl = 0.f;
      for (int i = ga + lane; i < gb; i += 64) {
        const float d = s_y[i] - s_pk[i];
        sw += s_w[i];
        sl += s_w[i] * d * d;
      }
      for (int off = 32; off > 0; off >>= 1) {
        sw += __shfl_xor(sw, off, 64);
        sl += __shfl_xor(sl, off, 64);
      }
      const float inv = (sw != 0.f) ? 1.0f / sw : 0.f;
      wl += sl * inv;
      const float sc = -2.0f * inv / (float)a.G;
      for (int i = ga + lane; i < gb; i += 64) {
        float dp = sc * s_w[i] * (s_y[i] - s_pk[i]);
        if (a.gweight != 1.0f) dp *= a.gweight;
        s_pk[i] = dp;
      }
    }
    if (lane == 0) s_red[wv] = wl;
  }
  __syncthreads();
  // ---- dg and the workgroup's partial of [dWout ; dbout]
  float acc[CM][4];
  float db[CM];
#pragma unroll
  for (int c = 0; c < CM; ++c) { acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f; db[c] = 0.f; }
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    const int rr = r + u * RL;
    if (rr < nrows) {
      const int64_t i = r0 + rr;
      const float* av = s_at + rr * C;
      const float4 m = draw(i);
      const float dp = s_pk[rr];
      float4 x = xr[u];
      x.x *= m.x; x.y *= m.y; x.z *= m.z; x.w *= m.w;
      float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
#pragma unroll
      for (int c = 0; c < CM; ++c) {
        if (c < C) {
          const float aa = av[c];
          const float4 w4 = *reinterpret_cast<const float4*>(sWs + c * Fh + 4 * q);
          u0 += aa * w4.x; u1 += aa * w4.y;
          u2 += aa * w4.z; u3 += aa * w4.w;
          const float d = dp * aa * sStd[c];
          acc[c][0] += x.x * d; acc[c][1] += x.y * d; acc[c][2] += x.z * d; acc[c][3] += x.w * d;
          db[c] += d;
        }
      }
      *reinterpret_cast<float4*>(dgo + i * Fh + 4 * q) = make_float4(m.x * dp * u0, m.y * dp * u1, m.z * dp * u2, m.w * dp * u3);
    }
  }
  // sum over the wave's row lanes in registers (lanes q, q + LPR, ... hold the same columns), then over the waves through LDS
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < CM; ++c) {
    if (c < C) {
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[c][s] += __shfl_xor(acc[c][s], off, 64);
        db[c] += __shfl_xor(db[c], off, 64);
      }
      if (lane < LPR) {
#pragma unroll
        for (int s = 0; s < 4; ++s) red[wv * items + (4 * q + s) * C + c] = acc[c][s];
        if (q == 0) red[wv * items + Fh * C + c] = db[c];
      }
    }
  }
  __syncthreads();
  for (int it = threadIdx.x; it < items; it += NT) {
    float s = 0.f;
    for (int w = 0; w < NW; ++w) s += red[w * items + it];
    a.partial[(int64_t)blockIdx.x * (items + 1) + it] = s;
  }
  if (threadIdx.x == 0) {      // s_red: written before the barrier behind the loss
    float s = 0.f;
    for (int w = 0; w < NW; ++w) s += s_red[w];
    a.partial[(int64_t)blockIdx.x * (items + 1) + items] = overlong ? __builtin_nanf("") : s / (float)a.G;
  }
}

// ---- embedding weight gradient --------------------------------------------------------------------------
// thread = (row lane, float4 column); acc[c] float4 per thread
template <int CM>
__global__ __launch_bounds__(256) void embed_bwd_fast_kernel(int64_t N, int C, int F, int64_t rows_per_block,
                                                             const float* __restrict__ atoms,
                                                             const float* __restrict__ dh0,
                                                             float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];     // [RL][C*F]
  const int c4n = F / 4;
  const int RL = 256 / c4n;
  const int q = threadIdx.x % c4n, r = threadIdx.x / c4n;
  float4 acc[CM];
#pragma unroll
  for (int c = 0; c < CM; ++c) acc[c] = f4zero();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = std::min<int64_t>(r0 + rows_per_block, N);
  // two rows per trip, both rows' loads requested before either is used (the sums keep their row order)
  for (int64_t i = r0 + r; i < r1; i += 2 * RL) {
    const int64_t i2 = i + RL;
    const bool two = i2 < r1;
    const int64_t j2 = two ? i2 : i;
    const float4 d = *reinterpret_cast<const float4*>(dh0 + i * F + 4 * q);
    const float4 d2 = *reinterpret_cast<const float4*>(dh0 + j2 * F + 4 * q);
    float a1[CM], a2[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
      a1[c] = c < C ? atoms[i * C + c] : 0.f;
      a2[c] = (c < C && two) ? atoms[j2 * C + c] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) { acc[c].x += a1[c] * d.x; acc[c].y += a1[c] * d.y; acc[c].z += a1[c] * d.z; acc[c].w += a1[c] * d.w; }
    if (two) {
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) { acc[c].x += a2[c] * d2.x; acc[c].y += a2[c] * d2.y; acc[c].z += a2[c] * d2.z; acc[c].w += a2[c] * d2.w; }
    }
  }
  const int items = C * F;
#pragma unroll
  for (int c = 0; c < CM; ++c)
    if (c < C) *reinterpret_cast<float4*>(red + r * items + c * F + 4 * q) = acc[c];
  __syncthreads();
  for (int it = threadIdx.x; it < items; it += 256) {
    float s = 0.f;
    for (int rr = 0; rr < RL; ++rr) s += red[rr * items + it];
    partial[(int64_t)blockIdx.x * items + it] = s;
  }
}

// ---- host side --------------------------------------------------------------------------------------------
static bool fast_enabled() {
  return !sw().head_generic;
}

bool head_fast_supported(int Fh, int C) {
  if (!(fast_enabled() && (Fh == 32 || Fh == 64 || Fh == 128) && C >= 1 && C <= HC_MAX)) return false;
  return (size_t)(256 / (Fh / 4)) * (Fh * C + C) * 4 <= 96 * 1024;     // LDS of the backward's final sum
}

bool head_fwd_fast_supported(int Fh, int C) {
  return fast_enabled() && (Fh == 32 || Fh == 64 || Fh == 128) && C >= 1 && C <= HC_MAX;
}

int head_fwd_fast(ng_ctx* ctx, hipStream_t st, int64_t N, int Fh, int C, const float* g, const float* mask,
                  const float* Wout, const float* bout, const float* atoms, const float* pstd,
                  const float* pavg, float* peaks, uint64_t seed, uint64_t offset, float keep, float* mask_out) {
  const int lpr = Fh / 4;
  const int64_t rpb = 256 / lpr;
  const int grid = (int)std::min<int64_t>(cdiv(N, rpb), (int64_t)ctx->num_cu * 8);
  const HeadDraw dr{seed, offset, keep, mask_out, mask_out ? replay_state(ctx) : nullptr};
  ProfScope ps(ctx, st, "head_fwd");
#define NG_HF(L)                                                                                                   \
  do {                                                                                                             \
    if (mask_out)                                                                                                  \
      hipLaunchKernelGGL((head_fwd_fast_kernel<L, true>), dim3(grid), dim3(256), 0, st, N, C, g, mask, Wout, bout, \
                         atoms, pstd, pavg, peaks, dr);                                                            \
    else                                                                                                           \
      hipLaunchKernelGGL((head_fwd_fast_kernel<L, false>), dim3(grid), dim3(256), 0, st, N, C, g, mask, Wout, bout, \
                         atoms, pstd, pavg, peaks, dr);                                                            \
  } while (0)
  if (lpr == 8) NG_HF(8);
  else if (lpr == 32) NG_HF(32);     // the reference's default width: fc output 128 (the one-thread-per-atom kernel took 91 us for 2770 atoms)
  else NG_HF(16);
#undef NG_HF
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

int head_bwd_fast(ng_ctx* ctx, hipStream_t st, int64_t N, int Fh, int C, const float* g, const float* mask,
                  const float* Wout, const float* atoms, const float* pstd, const float* dpeaks, float* dg,
                  float* dWout, float* dbout) {
  const int lpr = Fh / 4, rl = 256 / lpr;
  const int items = Fh * C + C;
  const int grid = (int)std::min<int64_t>(cdiv(N, rl), (int64_t)ctx->num_cu * 2);
  const int64_t rows = cdiv(cdiv(N, grid), rl) * rl;
  const int nb = (int)cdiv(N, rows);
  float* ws = (float*)workspace(ctx, (size_t)(nb + 1) * items * 4);
  if (!ws) return NG_ERR_NOMEM;
  float* partial = ws;
  float* summed = ws + (size_t)nb * items;
  if (float* dp = deferred_partials(ctx, (size_t)nb * items)) partial = dp;
  const size_t lds = (size_t)rl * items * 4;
  ProfScope ps(ctx, st, "head_bwd");
#define NG_HB(L, CM)                                                                                      \
  hipLaunchKernelGGL((head_bwd_fast_kernel<L, CM>), dim3(nb), dim3(256), lds, st, N, C, rows, g, mask, Wout, \
                     atoms, pstd, dpeaks, dg, partial)
  if (lpr == 8) { if (C <= 16) NG_HB(8, 16); else NG_HB(8, 32); }
  else if (lpr == 32) { if (C <= 16) NG_HB(32, 16); else NG_HB(32, 32); }     // fc output 128: the default width
  else { if (C <= 16) NG_HB(16, 16); else NG_HB(16, 32); }
#undef NG_HB
  ReduceSegs sg{};
  sg.n = 2;
  sg.begin[0] = 0; sg.len[0] = Fh * C; sg.dst[0] = dWout;
  sg.begin[1] = Fh * C; sg.len[1] = C; sg.dst[1] = dbout;
  (void)summed;
  NG_HIP(ctx, hipGetLastError());
  return reduce_seg_or_defer(ctx, st, partial, nb, items, items, sg);
}

// graphs per workgroup of the fused head + loss launch; 0 = this shape does not take it (a graph longer than HL_ROWS atoms, more
// than HL_CM elements, or more workgroups than the chip holds at once — two per CU at Fh = 32 (128 registers), one otherwise: a
// second round costs what the fusion saves)
int head_loss_graphs_per_wg(ng_ctx* ctx, int G, int Fh, int C, int64_t max_graph_atoms) {
  if (!head_fast_supported(Fh, C) || C > HL_CM || G < 1 || max_graph_atoms < 1 || max_graph_atoms > HL_ROWS) return 0;
  const int gpw = (int)cdiv(G, (int64_t)ctx->num_cu * (Fh == 32 ? 2 : 1));
  if ((int64_t)gpw * max_graph_atoms > HL_ROWS) return 0;
  return gpw;
}

int head_loss_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int G, int Fh, int C, int gpw, const float* g, uint64_t seed,
                     uint64_t offset, float keep, bool draw, float* mask_out, const float* Wout, const float* bout,
                     const float* atoms, const float* pstd, const float* pavg, const int32_t* gptr, const float* y,
                     const float* w, float gweight, float* peaks, float* dg, float* partial) {
  const int lpr = Fh / 4;
  const int items = Fh * C + C;
  const int grid = (int)cdiv(G, gpw);
  HeadLossArgs a{};
  a.N = N; a.G = G; a.C = C; a.gpw = gpw;
  a.g = g; a.Wout = Wout; a.bout = bout; a.atoms = atoms; a.pstd = pstd; a.pavg = pavg;
  a.gptr = gptr; a.y = y; a.w = w; a.gweight = gweight;
  a.peaks = peaks; a.dg = dg; a.partial = partial;
  a.dr = HeadDraw{seed, offset, keep, mask_out, draw ? replay_state(ctx) : nullptr};
  const size_t lds = ((size_t)(HL_NT / 64) * items + (size_t)HL_ROWS * C) * 4;
  ProfScope ps(ctx, st, "head_loss");
#define NG_HL(L)                                                                                               \
  do {                                                                                                         \
    if (draw) hipLaunchKernelGGL((head_loss_kernel<L, true>), dim3(grid), dim3(HL_NT), lds, st, a);            \
    else hipLaunchKernelGGL((head_loss_kernel<L, false>), dim3(grid), dim3(HL_NT), lds, st, a);                \
  } while (0)
  if (lpr == 8) NG_HL(8);
  else if (lpr == 32) NG_HL(32);
  else NG_HL(16);
#undef NG_HL
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

// second stage of the fused launch's weight-gradient partials: queued with the backward's other reductions while they are
// deferred (the caller keeps `partial` alive until the flush), launched at once otherwise
int head_loss_reduce(ng_ctx* ctx, hipStream_t st, const float* partial, int nb, int Fh, int C, float* dWout, float* dbout,
                     float* loss_out) {
  const int items = Fh * C + C;
  ReduceSegs sg{};
  sg.n = 3;
  sg.begin[0] = 0; sg.len[0] = Fh * C; sg.dst[0] = dWout;
  sg.begin[1] = Fh * C; sg.len[1] = C; sg.dst[1] = dbout;
  sg.begin[2] = items; sg.len[2] = 1; sg.dst[2] = loss_out;
  return reduce_seg_or_defer(ctx, st, partial, nb, items + 1, items + 1, sg, /*caller_owned=*/true);
}

bool embed_bwd_fast_supported(int F, int C) {
  // LDS for the final sum: (256 / (F/4)) * C * F floats
  return fast_enabled() && F % 4 == 0 && F >= 16 && F <= 256 && 256 % (F / 4) == 0 && C >= 1 && C <= HC_MAX &&
         (size_t)(256 / (F / 4)) * C * F * 4 <= 64 * 1024;
}

int embed_bwd_fast(ng_ctx* ctx, hipStream_t st, int64_t N, int C, int F, const float* atoms, const float* dh0,
                   float* dWemb) {
  const int rl = 256 / (F / 4);
  const int items = C * F;
  const int grid = (int)std::min<int64_t>(cdiv(N, rl), (int64_t)ctx->num_cu * 4);      // four workgroups per CU (their LDS allows it): 19.6 -> 15 us at F = 64, 55 -> 40 at F = 256; the head backward is fastest at two
  const int64_t rows = cdiv(cdiv(N, grid), rl) * rl;
  const int nb = (int)cdiv(N, rows);
  float* partial = deferred_partials(ctx, (size_t)nb * items);
  if (!partial) partial = (float*)workspace(ctx, (size_t)nb * items * 4);
  if (!partial) return NG_ERR_NOMEM;
  const size_t lds = (size_t)rl * items * 4;
  ProfScope ps(ctx, st, "embed_bwd");
  if (C <= 16)
    hipLaunchKernelGGL((embed_bwd_fast_kernel<16>), dim3(nb), dim3(256), lds, st, N, C, F, rows, atoms, dh0, partial);
  else
    hipLaunchKernelGGL((embed_bwd_fast_kernel<32>), dim3(nb), dim3(256), lds, st, N, C, F, rows, atoms, dh0, partial);
  NG_HIP(ctx, hipGetLastError());
  return reduce_or_defer(ctx, st, partial, nb, items, dWemb);
}

}  // namespace ng
