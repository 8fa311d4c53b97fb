// fp32 MFMA tile GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak).
//
//   C[m][n] = sum_k Q(m,k) * P(n,k)
//
// Q is the "row" operand (activations), P the "column" operand (weights).  Each operand lives in
// LDS either K-contiguous ([idx][k], fragment = one ds_read_b128 per 4 MFMA steps) or K-strided
// ([k][idx], fragment = four conflict-free ds_read_b32), so that NN / NT / TN products all read
// their natural row-major global layouts without any transpose:
//     Y  = X  W      : Q = X  [m][k] (KC)      P = W  [k][n] (KS)
//     dX = dP W^T    : Q = dP [m][n] (KC)      P = W  [k][n] (KC, contraction over n)
//     dW = X^T dP    : Q = X  [r][k] (KS)      P = dP [r][n] (KS, contraction over rows r)
//
// MFMA mapping: P feeds the A operand (output rows i = n), Q feeds the B operand (output cols
// j = m), so a lane ends up with 4 consecutive n for one m: a float4 of a row-major C row.
// The k order inside an 8-wide block is permuted (lane-half h handles k = 8t+4h+s at step s);
// both operands use the same permutation so the product is unchanged.
#pragma once
#include <hip/hip_runtime.h>

namespace ng {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- activations ---------------------------------------------------------------------------
// softplus(x) = log(1+exp(x)) = max(x,0) + log1p(exp(-|x|))   (keras 'softplus')
// v_exp / v_log hardware transcendentals; below t = 2^-11 the series t - t^2/2 keeps the relative
// accuracy where 1+t would round (both arms computed: a v_cndmask, no branch).  |error| < 2e-7.
// The logarithm is v_log_f32 (log2) times ln 2, not __logf: its argument lies in [1, 2], so the denormal-input scaling
// and the two-word ln 2 product of the library expansion (14 instructions around the v_log) buy nothing against the
// 6e-8 rounding of 1 + t; in fc_fwd the activation was 22 of 71 us (-DFC_ABL_NOACT), in mp_win_fwd ~1.1 k of 6.7 k cycles.
__device__ __forceinline__ float softplus_f(float x) {
  const float t = __expf(-fabsf(x));
  const float l_big = 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + t);
  const float l_small = t * (1.0f - 0.5f * t);
  return fmaxf(x, 0.0f) + (t > 4.8828125e-4f ? l_big : l_small);
}
// d softplus / dx expressed through the saved OUTPUT s = softplus(x):  sigmoid(x) = 1 - exp(-s)
__device__ __forceinline__ float softplus_grad_from_out(float s) { return 1.0f - __expf(-s); }

// keras activations selectable through hypers 'mp_activation' / 'fc_activation' (model.py:33-36).
// codes: NG_ACT_NONE 0, NG_ACT_SOFTPLUS 1, NG_ACT_RELU 2, NG_ACT_TANH 3
__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case 1: return softplus_f(x);
    case 2: return fmaxf(x, 0.0f);
    case 3: return tanhf(x);
    default: return x;
  }
}
// derivative w.r.t. the pre-activation, recovered from the activation OUTPUT s
__device__ __forceinline__ float act_grad_from_out(int act, float s) {
  switch (act) {
    case 1: return softplus_grad_from_out(s);
    case 2: return s > 0.0f ? 1.0f : 0.0f;
    case 3: return 1.0f - s * s;
    default: return 1.0f;
  }
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// XCD-aware workgroup -> tile remap (bijective for any grid size).  The dispatcher is observed to
// place workgroup b on XCD b % 8; this gives XCD x the contiguous tile range [x*n/8, (x+1)*n/8) so
// that neighbouring tiles (which share gathered rows) share one L2.  Placement is used for speed only.
__device__ __forceinline__ unsigned xcd_tile(unsigned b, unsigned n) {
  const unsigned q = n / 8, r = n % 8, x = b % 8, k = b / 8;
  const unsigned base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + k;
}

// ---- generic operand loaders -----------------------------------------------------------------
// A loader returns the float4 at (r, c..c+3) of a row-major [R][C] array, zero outside.
struct LoadPlain {
  const float* p;
  int64_t R;
  int C;   // columns (multiple of 4)
  int ld;  // leading dimension
  __device__ __forceinline__ float4 operator()(int64_t r, int c) const {
    if (r < R && c < C) return *reinterpret_cast<const float4*>(p + r * ld + c);
    return f4zero();
  }
};

// dP = dY * act'(S) * rowscale   (S = saved activation output; nullptr -> linear)
struct LoadGradAct {
  const float* dY;
  const float* S;         // may be nullptr
  const float* rowscale;  // may be nullptr
  int64_t R;
  int C;
  int act;
  __device__ __forceinline__ float4 operator()(int64_t r, int c) const {
    if (r < R && c < C) {
      float4 g = *reinterpret_cast<const float4*>(dY + r * C + c);
      if (S) {
        float4 s = *reinterpret_cast<const float4*>(S + r * C + c);
        g.x *= act_grad_from_out(act, s.x);
        g.y *= act_grad_from_out(act, s.y);
        g.z *= act_grad_from_out(act, s.z);
        g.w *= act_grad_from_out(act, s.w);
      }
      if (rowscale) {
        float v = rowscale[r];
        g.x *= v; g.y *= v; g.z *= v; g.w *= v;
      }
      return g;
    }
    return f4zero();
  }
};

// ---- the tile kernel ---------------------------------------------------------------------------
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool Q_KC, bool P_KC>
struct GemmTile {
  static_assert(WAVES_M * WAVES_N == 4, "256-thread workgroups");
  static constexpr int TM = BM / (32 * WAVES_M);
  static constexpr int TN = BN / (32 * WAVES_N);
  static constexpr int LDQ = (Q_KC ? BK : BM) + 4;
  static constexpr int LDP = (P_KC ? BK : BN) + 4;
  static constexpr int ROWS_Q = Q_KC ? BM : BK;
  static constexpr int ROWS_P = P_KC ? BN : BK;
  static constexpr int NQ = BM * BK / 4 / 256;  // float4 per thread per k-tile
  static constexpr int NP = BN * BK / 4 / 256;
  static constexpr int LDS_FLOATS = ROWS_Q * LDQ + ROWS_P * LDP;
  static_assert(NQ >= 1 && NP >= 1, "tile too small for 256 threads");
};

// Epi: void operator()(int64_t m, int n, float4 v, int z)
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool Q_KC, bool P_KC, class LoadQ,
          class LoadP, class Epi>
__global__ __launch_bounds__(256) void gemm_kernel(int64_t M, int N, int64_t K, int64_t k_chunk,
                                                   LoadQ lq, LoadP lp, Epi epi, unsigned* guard_word, unsigned guard_epoch,
                                                   unsigned tgx, unsigned tgy, unsigned tgz) {
  // range fallback of a split-operand GEMM (ng_internal.h: RangeGuard): run only if that kernel raised the guard.  Launched
  // with a SMALL one-dimensional grid whose workgroups walk the (tgx, tgy, tgz) tile grid — a fallback that does not run
  // should cost the dispatch of a few hundred workgroups, not of the product's thousands (6.4 us per launch, 24 per step at
  // the default width).  tgx == 0: the ordinary launch, one workgroup per tile.
  if (guard_word && __hip_atomic_load(guard_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != guard_epoch) return;
  const uint64_t ntile = tgx ? (uint64_t)tgx * tgy * tgz : 0;
  uint64_t tile = blockIdx.x;
  if (tgx && tile >= ntile) return;
#pragma unroll 1
  for (;;) {
  const unsigned bix = tgx ? (unsigned)(tile % tgx) : blockIdx.x;
  const unsigned biy = tgx ? (unsigned)((tile / tgx) % tgy) : blockIdx.y;
  const unsigned biz = tgx ? (unsigned)(tile / ((uint64_t)tgx * tgy)) : blockIdx.z;
  using T = GemmTile<BM, BN, BK, WAVES_M, WAVES_N, Q_KC, P_KC>;
  __shared__ __attribute__((aligned(16))) float smem[T::LDS_FLOATS];
  float* sQ = smem;
  float* sP = smem + T::ROWS_Q * T::LDQ;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  const int64_t m0 = (int64_t)bix * BM;
  const int n0 = biy * BN;
  const int64_t k_begin = (int64_t)biz * k_chunk;
  int64_t k_end = k_begin + k_chunk;
  if (k_end > K) k_end = K;

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 rq[T::NQ], rp[T::NP];

  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < T::NQ; ++i) {
      const int lin = tid + i * 256;
      if constexpr (Q_KC) {
        const int row = lin / (BK / 4), c4 = lin % (BK / 4);
        rq[i] = lq(m0 + row, (int)(k0 + c4 * 4));
      } else {
        const int row = lin / (BM / 4), c4 = lin % (BM / 4);
        rq[i] = lq(k0 + row, (int)(m0 + c4 * 4));
      }
    }
#pragma unroll
    for (int i = 0; i < T::NP; ++i) {
      const int lin = tid + i * 256;
      if constexpr (P_KC) {
        const int row = lin / (BK / 4), c4 = lin % (BK / 4);
        rp[i] = lp((int64_t)n0 + row, (int)(k0 + c4 * 4));
      } else {
        const int row = lin / (BN / 4), c4 = lin % (BN / 4);
        rp[i] = lp(k0 + row, n0 + c4 * 4);
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < T::NQ; ++i) {
      const int lin = tid + i * 256;
      constexpr int C4 = (Q_KC ? BK : BM) / 4;
      const int row = lin / C4, c4 = lin % C4;
      *reinterpret_cast<float4*>(sQ + row * T::LDQ + c4 * 4) = rq[i];
    }
#pragma unroll
    for (int i = 0; i < T::NP; ++i) {
      const int lin = tid + i * 256;
      constexpr int C4 = (P_KC ? BK : BN) / 4;
      const int row = lin / C4, c4 = lin % C4;
      *reinterpret_cast<float4*>(sP + row * T::LDP + c4 * 4) = rp[i];
    }
  };

  if (k_begin < k_end) fetch(k_begin);
  for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
    stash();
    __syncthreads();
    if (k0 + BK < k_end) fetch(k0 + BK);
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      float pf[T::TN][4], qf[T::TM][4];
#pragma unroll
      for (int j = 0; j < T::TN; ++j) {
        const int idx = wn * (T::TN * 32) + j * 32 + l31;
        if constexpr (P_KC) {
          const float4 v = *reinterpret_cast<const float4*>(sP + idx * T::LDP + kb * 8 + half * 4);
          pf[j][0] = v.x; pf[j][1] = v.y; pf[j][2] = v.z; pf[j][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) pf[j][s] = sP[(kb * 8 + half * 4 + s) * T::LDP + idx];
        }
      }
#pragma unroll
      for (int i = 0; i < T::TM; ++i) {
        const int idx = wm * (T::TM * 32) + i * 32 + l31;
        if constexpr (Q_KC) {
          const float4 v = *reinterpret_cast<const float4*>(sQ + idx * T::LDQ + kb * 8 + half * 4);
          qf[i][0] = v.x; qf[i][1] = v.y; qf[i][2] = v.z; qf[i][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) qf[i][s] = sQ[(kb * 8 + half * 4 + s) * T::LDQ + idx];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
          for (int j = 0; j < T::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[j][s], qf[i][s], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: lane holds C[m = col l31][n = 8q + 4*half + (0..3)] per 32x32 tile
#pragma unroll
  for (int i = 0; i < T::TM; ++i) {
    const int64_t m = m0 + wm * (T::TM * 32) + i * 32 + l31;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * (T::TN * 32) + j * 32 + q * 8 + half * 4;
        if (m < M && n < N) {
          epi(m, n, make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2],
                                acc[i][j][4 * q + 3]),
              (int)biz);
        }
      }
    }
  }
  if (!tgx) break;
  tile += gridDim.x;
  if (tile >= ntile) break;
  __syncthreads();        // the next tile's operands overwrite this one's in LDS
  }
}

}  // namespace ng
