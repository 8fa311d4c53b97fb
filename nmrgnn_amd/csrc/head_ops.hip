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
