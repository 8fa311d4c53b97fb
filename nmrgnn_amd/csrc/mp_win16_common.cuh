// Shared pieces of the sixteen-wave window kernels (mp_win16.hip forward, mp_win16_bwd.hip edge-side backward,
// mp_win16_node.hip node-side backward): tile / workgroup constants, DPP helpers, the window decision and its LDS-DMA staging.
#pragma once
#include <algorithm>

#include <hip/hip_runtime.h>

namespace ng {
namespace w16c {

constexpr int WF = 64;          // feature width
constexpr int WTA = 64;         // atoms per tile
constexpr int WROWS = 288;      // window rows
constexpr int WC4 = WF / 4;     // float4 per row = lanes per atom
constexpr int WTHREADS = 1024;
constexpr int NW = WTHREADS / 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// min over the 64 lanes, valid in lane 63: row_shr 1,2,4,8 inside rows of 16, then row_bcast 15 / 31
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
// row_ror:S inside the 16 lanes of a DPP row
template <int S>
__device__ __forceinline__ int ror_i(int v) {
  if (S == 0) return v;
  return __builtin_amdgcn_update_dpp(0, v, 0x120 + (S & 15), 0xf, 0xf, false);
}
template <int S>
__device__ __forceinline__ float ror_f(float v) {
  return __builtin_bit_cast(float, ror_i<S>(__builtin_bit_cast(int, v)));
}
// acc += w * h for one float4 of features, as two v_pk_fma_f32
__device__ __forceinline__ void pk_axpy(f32x2& lo, f32x2& hi, float w, const float4& h) {
  const f32x2 ww = {w, w};
  lo = __builtin_elementwise_fma(ww, f32x2{h.x, h.y}, lo);
  hi = __builtin_elementwise_fma(ww, f32x2{h.z, h.w}, hi);
}

// every thread takes the same decision from the sixteen partial ranges (ctl[wave], ctl[NW + wave]); returns true when the
// window has to be restaged at the (updated) wlo.  mode: 0 = gather from the window, 1 = gather from global memory
__device__ __forceinline__ bool win_decide(const int* __restrict__ ctl, int& wlo, int& mode) {
  int lo = ctl[0], hi = ctl[NW];
#pragma unroll
  for (int i = 1; i < NW; ++i) { lo = min(lo, ctl[i]); hi = max(hi, ctl[NW + i]); }
  mode = 0;
  if (hi < lo) return false;                                  // empty tile
  if (lo >= wlo && hi < wlo + WROWS) return false;            // window hit
  if (hi - lo + 1 > WROWS || hi - lo < 0) { mode = 1; return false; }      // too wide: gather from global memory
  wlo = max(0, lo - (WROWS - (hi - lo + 1)) / 2);
  return true;
}

// The window by LDS-DMA: its 288 rows are one contiguous 72-KB block of the source array — 72 wave-instructions of 1 KB
// straight into LDS, no registers in between.  The buffer is the block itself (base = row wlo, clipped at the array's end:
// rows past it read as zeros), so there is no 32-bit limit on the array and no register-staged second path (whose five
// per-lane 64-bit addresses, hoisted out of the tile loop, were spilled and reloaded every tile).
__device__ __forceinline__ void win_dma(float* __restrict__ win, const float* __restrict__ src, int wlo_v, int64_t N, int wave, int lane) {
  const int wlo = __builtin_amdgcn_readfirstlane(wlo_v);
  const int64_t rows = std::min<int64_t>(N - wlo, WROWS);
  const dma_i4 rs = dma_rsrc(src + (int64_t)wlo * WF, (unsigned)(rows * (WF * 4)));
#pragma unroll
  for (int j = 0; j < (WROWS * WF * 4 / 1024 + NW - 1) / NW; ++j) {
    const int kb = wave + NW * j;
    if (kb < WROWS * WF * 4 / 1024) lds_dma16(rs, reinterpret_cast<char*>(win) + kb * 1024, lane * 16, kb * 1024);
  }
}

}  // namespace w16c
}  // namespace ng
