// Shared pieces of the wave-autonomous window kernels (mp_wave.hip forward, mp_wave_bwd.hip edge-side backward): geometry, the
// swizzled window and its LDS-DMA staging, the micro-tile's source-range check.  See mp_wave.hip for the design.
#pragma once
#include <algorithm>

#include "h2_common.cuh"
#include "mp_win16_common.cuh"

namespace ng {
namespace wv {

using w16c::f32x4;

constexpr int WF = 64;
constexpr int WROWS = 288;                 // window rows
constexpr int MT = 16;                     // atoms per micro-tile
constexpr int NWV = 8;                     // waves per workgroup
constexpr int WTHREADS = NWV * 64;
constexpr int GROUP = 2 * NWV * MT;        // atoms per window group: two micro-tiles per wave
constexpr int E = 3;
constexpr int NT2 = E * WF / 32;           // 32-wide k-steps of the forward contraction
constexpr int WIN_BYTES = WROWS * WF * 4;
constexpr int WIMG_BYTES = E * WF * WF * 2 * 2;      // a piece image of w[l][m][n]: 12,288 x two fp16 pieces = 48 KB
constexpr int VOFF_NONE = 0x7ffffff0;      // beyond every buffer: the lane's piece reads as zeros

typedef int i32x4 __attribute__((ext_vector_type(4)));

// the window: rows wlo .. wlo+287 of src, chunk c of row R at 16-byte position 16 R + (c ^ (R & 15))
__device__ __forceinline__ void win_dma(char* win, const float* src, int64_t wlo, int64_t N, int wave, int lane) {
  const int64_t rows = std::min<int64_t>(N - wlo, WROWS);
  const dma_i4 rs = dma_rsrc(src + wlo * WF, (unsigned)(rows * (WF * 4)));
#pragma unroll
  for (int j = 0; j < WIN_BYTES / 1024 / NWV; ++j) {
    const int kb = wave + NWV * j;
    const int R = 4 * kb + (lane >> 4);
    lds_dma16(rs, win + kb * 1024, R * (WF * 4) + (((lane & 15) ^ (R & 15)) << 4), 0);
  }
}
// 16-byte chunk c of window row R
__device__ __forceinline__ int win_off(int R, int c) { return (R << 8) + ((c ^ (R & 15)) << 4); }

// a 48-KB weight piece image, fragment order, into LDS
__device__ __forceinline__ void wimg_dma(char* wimg, const float* img, int wave, int lane) {
  const dma_i4 rw = dma_rsrc(img, WIMG_BYTES);
#pragma unroll
  for (int j = 0; j < WIMG_BYTES / 1024 / NWV; ++j) {
    const int kb = wave + NWV * j;
    lds_dma16(rw, wimg + kb * 1024, lane * 16, kb * 1024);
  }
}

// the neighbour indices of micro-tile row0 .. row0+15 into 1 KB of the wave's strip, laid out [16-byte piece][atom]; rows past N
// and pieces past K read as zeros
__device__ __forceinline__ void nlist_dma(const int32_t* nlist, int K, int64_t N, char* strip, int64_t row0, int lane) {
  const int at = lane & 15, pp = lane >> 4;
  const int rows = (int)std::min<int64_t>(MT, N - row0);
  const dma_i4 rn = dma_rsrc(nlist + row0 * K, (unsigned)(rows * K * 4));
  lds_dma16(rn, strip, pp < (K >> 2) ? at * K * 4 + pp * 16 : VOFF_NONE, 0);
}

// do the micro-tile's sources (its staged neighbour indices) lie in the window [wlo, wlo + WROWS)?  wave-uniform
__device__ __forceinline__ bool sources_in_window(const char* strip, int lane, int nq, int64_t row0, int64_t N, int wlo) {
  const i32x4 mine = *reinterpret_cast<const i32x4*>(strip + (lane << 4));
  const bool valid = (lane >> 4) < nq && row0 + (lane & 15) < N;
  int lo = std::min(std::min(mine[0], mine[1]), std::min(mine[2], mine[3]));
  int hi = std::max(std::max(mine[0], mine[1]), std::max(mine[2], mine[3]));
  lo = valid ? lo : 0x7fffffff;
  hi = valid ? hi : -1;
  lo = __builtin_amdgcn_readlane(w16c::wave_min_i32(lo), 63);
  hi = -__builtin_amdgcn_readlane(w16c::wave_min_i32(-hi), 63);
  return hi < lo || (lo >= wlo && hi < wlo + WROWS);
}

}  // namespace wv
}  // namespace ng
