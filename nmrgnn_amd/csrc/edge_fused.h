// Shared constants of the fused persistent edge kernels (edge_fused.hip, edge_fused_bwd.hip).
#pragma once
#include "ng_internal.h"

namespace ng {

constexpr int FH = 128;       // hidden width handled by the fused path
constexpr int FTM = 64;       // edges per tile
constexpr int FLD = FH + 4;   // LDS row stride (floats): 16-B slots shift by one per row
constexpr int FMAX_E = 8;

// Wpk / WpkT: fragment-ordered copies of the three hidden weight matrices (see edge_fused.hip)
int edge_fused_pack(ng_ctx* ctx, hipStream_t st, const float* const* W, float* Wpk, float* WpkT);
PackJob edge_fused_pack_job(const float* const* W, float* Wpk, float* WpkT);

}  // namespace ng

namespace ng {
int edge_fused_bwd(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src,
                   const float* d_eff, const float* centers, float gap, const float* const* W,
                   const float* z_save, const float* de, float* const* dW, float* const* db, int tape_layout = -1,
                   LiveEdges live = LiveEdges());
}  // namespace ng

// Workgroup barrier that orders LDS traffic only.  hipcc's __syncthreads() also drains vmcnt(0), which
// would make every barrier wait for the register-destination prefetches (next tile's activations,
// next layer's weight slab) and the activation stores that are deliberately left in flight across
// phases; the compiler still inserts the vmcnt wait in front of the first USE of a loaded register.
#define NG_LDS_BARRIER()                                   \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("" ::: "memory");                         \
  } while (0)

namespace ng {
// Edge path on the fp16 matrix pipe with two-piece split fp32 operands (edge_fwd_h2.hip, edge_bwd_h2.hip): the default;
// NG_EDGE_MATH=fp32 selects the f32-input MFMA kernels (edge_fused.hip, edge_fused_bwd.hip)
bool edge_split_enabled();
int edge_h2_fwd(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src, const float* d_eff,
                const float* centers, float gap, const float* const* W, const float* const* b, float* e_out,
                float* z_save, LiveEdges live = LiveEdges());
bool edge_bwd_h2_supported(int E, int64_t n_edges);
int edge_bwd_h2_segments(int64_t n_edges);
// Layout of the saved-activation tape z_save[Le-1][n_edges][128] between the edge forward and backward:
// false: row-major.  true (both directions run the split-operand kernels): inside every FULL group of 32 consecutive
// edges the 32 x 128 block is stored in the kernels' register layout, float index ((bo*4 + q)*64 + hf*32 + r)*4 + j for
// edge r, feature 32 bo + 8 q + 4 hf + j; a last partial group stays row-major.  Same footprint either way.
bool edge_tape_blocked(int E, int64_t n_edges);
size_t edge_bwd_h2_ws_bytes();
// wt_img: edge_bwd_h2_ws_bytes() of scratch; partial: [edge_bwd_h2_segments(n_edges) * grid][part_stride] in the
// layout of edge_fused_bwd.hip
int edge_bwd_h2_launch(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src, const float* d_eff,
                       const float* centers, float gap, const float* const* W, const float* z_save, const float* de,
                       char* wt_img, float* partial, int part_stride, int grid, int tape_blocked, RangeGuard guard,
                       LiveEdges live = LiveEdges());
}  // namespace ng
