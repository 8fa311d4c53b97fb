// Tall-skinny dense products for the 64-feature node path:  Y[N, NOUT] = epi( X[N, KIN] . W )
// with N ~ 1e5 rows and KIN, NOUT in {64, 128, 192}.  The generic tile GEMM (mfma_gemm.cuh) re-stages
// the (tiny) weight matrix through LDS for every 128-row tile and reaches ~25 % of the MFMA peak on
// these shapes; here each wave keeps its weight slab as MFMA A-fragments in registers for the whole
// launch (<= 96 VGPRs), a workgroup streams 64-row X tiles through one LDS buffer, and three
// workgroups per CU overlap each other's loads, MFMAs and epilogues.
//
// Used for (reference lines in the callers):
//   MPLayer update     P = v * (A . Wp)            KIN = 64E, NOUT = 64     (+ activation, residual)
//   MPLayer dA         dA = dP . Wp^T              KIN = 64,  NOUT = 64E    (prologue forms dP)
//   MPLayer dh         dh = dH + B . Wq            KIN = 64E, NOUT = 64
//   FCBlock Dense      Y = act(X . W + b) (+ X)    KIN = 64,  NOUT = 64 / 32
//   FCBlock dX         dX = (dY) + dP . W^T        KIN = 64 / 32, NOUT = 64 (prologue forms dP)
#include <algorithm>

#include "mfma_gemm.cuh"
#include "ng_internal.h"

namespace ng {

constexpr int TG_TM = 64;

// Wfrag[slab][t][lane][s] = W(k = 8t + 4(lane>>5) + s, o = 32 slab + (lane&31)),  W(k,o) = w[k*sk + o*so]
// (o >= n_out or k >= k_in -> 0: lets NOUT = 32 / KIN = 32 run on the 64-wide instantiation)
__global__ void tall_pack_kernel(int k_in, int n_out, int kpad, int npad, int sk, int so,
                                 const float* __restrict__ w, float* __restrict__ out) {
  const int total = kpad * npad;
  const int nt = kpad / 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int r = idx;
    const int s = r & 3; r >>= 2;
    const int lane = r & 63; r >>= 6;
    const int t = r % nt, slab = r / nt;
    const int k = 8 * t + 4 * (lane >> 5) + s, o = 32 * slab + (lane & 31);
    out[idx] = (k < k_in && o < n_out) ? w[(int64_t)k * sk + (int64_t)o * so] : 0.f;
  }
}

int tall_pack(ng_ctx* ctx, hipStream_t st, int k_in, int n_out, int kpad, int npad, int sk, int so,
              const float* w, float* out) {
  hipLaunchKernelGGL(tall_pack_kernel, dim3(48), dim3(256), 0, st, k_in, n_out, kpad, npad, sk, so, w,
                     out);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

template <int KIN, int NOUT, bool PRO>
__global__ __launch_bounds__(256, 2) void tall_gemm_kernel(TallArgs a) {
  constexpr int LDX = KIN + 4;
  constexpr int NT = KIN / 8;
  constexpr int S = NOUT / 64;          // 32-wide output slabs per wave
  constexpr int LDY = NOUT + 4;
  extern __shared__ __attribute__((aligned(16))) float sX[];   // [64][LDX] | sY [64][LDY]
  float* sY = sX + TG_TM * LDX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int rt = wave & 1, sl0 = (wave >> 1) * S;

  float wf[S][KIN / 2];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const float4* p = reinterpret_cast<const float4*>(a.Wfrag) + ((sl0 + j) * NT) * 64 + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 v = p[t * 64];
      wf[j][4 * t + 0] = v.x; wf[j][4 * t + 1] = v.y; wf[j][4 * t + 2] = v.z; wf[j][4 * t + 3] = v.w;
    }
  }
  const int64_t ntiles = (a.N + TG_TM - 1) / TG_TM;
  constexpr int C4 = KIN / 4;

  // X tiles are fetched ONE TILE AHEAD into registers: the loads of tile t+1 are issued right after
  // tile t has been written to LDS and stay in flight (LDS-only barrier) while tile t's MFMAs run.
  constexpr int NLD = TG_TM * C4 / 256;
  float4 v[NLD], sv[PRO ? NLD : 1];
  auto fetch = [&](int64_t tile) {
    const int64_t i0 = tile * TG_TM;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int t = tid + u * 256;
      const int r = t / C4, c4 = t % C4;
      const bool ok = i0 + r < a.N && c4 * 4 < a.k_valid;
      v[u] = ok ? *reinterpret_cast<const float4*>(a.X + (i0 + r) * a.ldx + c4 * 4) : f4zero();
      if (PRO) {
        sv[u] = (ok && a.S_in) ? *reinterpret_cast<const float4*>(a.S_in + (i0 + r) * a.ldx + c4 * 4)
                               : f4zero();
      }
    }
  };
  if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x);

#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t i0 = tile * TG_TM;
    // ---- X tile -> LDS (prologue: dP = dH * act'(S) * rowscale, also written out when asked)
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int t = tid + u * 256;
      const int r = t / C4, c4 = t % C4;
      float4 x = v[u];
      if (PRO) {
        const bool ok = i0 + r < a.N && c4 * 4 < a.k_valid;
        if (ok) {
          if (a.S_in) {
            x.x *= act_grad_from_out(a.act_in, sv[u].x); x.y *= act_grad_from_out(a.act_in, sv[u].y);
            x.z *= act_grad_from_out(a.act_in, sv[u].z); x.w *= act_grad_from_out(a.act_in, sv[u].w);
          }
          if (a.rs_in) {
            const float sc = a.rs_in[i0 + r];
            x.x *= sc; x.y *= sc; x.z *= sc; x.w *= sc;
          }
          if (a.dP_out) *reinterpret_cast<float4*>(a.dP_out + (i0 + r) * a.ldx + c4 * 4) = x;
        }
      }
      *reinterpret_cast<float4*>(sX + r * LDX + c4 * 4) = x;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
    // ---- rows 32rt.., output slabs sl0 .. sl0+S-1
    f32x16 acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* xrow = sX + (rt * 32 + l31) * LDX + 4 * half;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 x = *reinterpret_cast<const float4*>(xrow + 8 * t);
#pragma unroll
      for (int j = 0; j < S; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][4 * t + 0], x.x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][4 * t + 1], x.y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][4 * t + 2], x.z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][4 * t + 3], x.w, acc[j], 0, 0, 0);
      }
    }
    // ---- epilogue part 1 (registers): rowscale, bias, activation -> LDS tile sY
    {
      const int lr = rt * 32 + l31;
      const int64_t row = i0 + lr;
      const float rs = (a.rowscale && row < a.N) ? a.rowscale[row] : 1.0f;
#pragma unroll
      for (int j = 0; j < S; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = (sl0 + j) * 32 + 8 * q + 4 * half;
          float4 v = make_float4(acc[j][4 * q + 0] * rs, acc[j][4 * q + 1] * rs, acc[j][4 * q + 2] * rs,
                                 acc[j][4 * q + 3] * rs);
          if (a.bias && n < a.n_valid) {
            const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
          }
          if (a.act != NG_ACT_NONE) {
            v.x = act_apply(a.act, v.x); v.y = act_apply(a.act, v.y);
            v.z = act_apply(a.act, v.z); v.w = act_apply(a.act, v.w);
          }
          *reinterpret_cast<float4*>(sY + lr * LDY + n) = v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- epilogue part 2: whole rows leave the CU coalesced (activation copy, residual, output)
    {
      constexpr int O4 = NOUT / 4;
      for (int t = tid; t < TG_TM * O4; t += 256) {
        const int r = t / O4, c4 = t % O4;
        if (i0 + r < a.N && c4 * 4 < a.n_valid) {
          float4 v = *reinterpret_cast<const float4*>(sY + r * LDY + c4 * 4);
          const int64_t o = (i0 + r) * a.ldo + c4 * 4;
          if (a.S_save) *reinterpret_cast<float4*>(a.S_save + o) = v;
          if (a.resid) {
            const float4 r4 = *reinterpret_cast<const float4*>(a.resid + o);
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
          }
          *reinterpret_cast<float4*>(a.out + o) = v;
        }
      }
    }
    // (sX is rewritten only after the next loop-top barrier sequence; sY after the next MFMA phase)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
}

// (KIN = 192 with the gradient prologue is not instantiated: its 48-float slab beside the prologue operands spilled 332 B per
// lane; ng_dense_bwd sends that shape — a Dense with 129..192 outputs and <= 64 inputs — to the generic GEMM)
template <int KIN, int NOUT>
static void launch_tall(hipStream_t st, int grid, const TallArgs& a, bool pro) {
  const size_t lds = (size_t)TG_TM * (KIN + 4 + NOUT + 4) * 4;
  if constexpr (KIN < 192) {
    if (pro) {
      hipLaunchKernelGGL((tall_gemm_kernel<KIN, NOUT, true>), dim3(grid), dim3(256), lds, st, a);
      return;
    }
  }
  if (!pro)
    hipLaunchKernelGGL((tall_gemm_kernel<KIN, NOUT, false>), dim3(grid), dim3(256), lds, st, a);
}

bool tall_gemm_supported(int kpad, int npad) {
  return (kpad == 64 || kpad == 128 || kpad == 192) && (npad == 64 || npad == 128 || npad == 192) &&
         (kpad == 64 || npad == 64);
}

int tall_gemm(ng_ctx* ctx, hipStream_t st, int kpad, int npad, const TallArgs& a, bool prologue,
              const char* tag) {
  if (a.N == 0) return NG_OK;
  NG_REQUIRE(ctx, tall_gemm_supported(kpad, npad), "tall_gemm: unsupported shape");
  NG_REQUIRE(ctx, !(prologue && kpad == 192), "tall_gemm: the gradient prologue is not built for a 192-wide contraction");
  const int64_t ntiles = cdiv(a.N, TG_TM);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu * 2);
  ProfScope ps(ctx, st, tag);
  const int key = kpad * 1000 + npad;
  switch (key) {
    case 64064: launch_tall<64, 64>(st, grid, a, prologue); break;
    case 128064: launch_tall<128, 64>(st, grid, a, prologue); break;
    case 192064: launch_tall<192, 64>(st, grid, a, prologue); break;
    case 64128: launch_tall<64, 128>(st, grid, a, prologue); break;
    case 64192: launch_tall<64, 192>(st, grid, a, prologue); break;
    default: return fail(ctx, NG_ERR_UNSUPPORTED, "tall_gemm shape");
  }
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

}  // namespace ng
