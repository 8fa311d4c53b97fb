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

}  // namespace ng

namespace ng {
int edge_fused_bwd(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src,
                   const float* d_eff, const float* centers, float gap, const float* const* W,
                   const float* z_save, const float* de, float* const* dW, float* const* db);
}  // namespace ng
