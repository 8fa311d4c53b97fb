// Internal C++ entry points shared between the .hip translation units.
#pragma once
#include <algorithm>

#include "ng_common.h"

namespace ng {

// ---- operand-range guard of the fp16-piece ("h2") kernels (round 3) ------------------------------------------------
// Two fp16 pieces hold |x| < 65504; the reference's Dense / einsum are plain fp32 (nmrgnn/model.py:132-138,
// layers.py:39-40) and return finite numbers far beyond that.  An operand outside the range turns into inf - inf = NaN
// in the piece products, so every h2 kernel checks its OUTPUTS for non-finite values and, if it sees one, stores the
// call's epoch into a per-context device word; the entry point then launches the f32-input MFMA kernel of the same
// contract with the same (word, epoch): its workgroups return at once unless the word carries their epoch, in which
// case they recompute the whole call.  No host synchronisation, no flag to clear (epochs are unique per call, calls on
// one stream are ordered), ~2 us of empty launch per guarded call.  A genuinely non-finite input takes the fp32 path as
// well and comes out non-finite there too.
// (struct RangeGuard {word, epoch} itself lives in ng_common.h: the context's image cache stores pack jobs that carry one)
RangeGuard range_guard_begin(ng_ctx* ctx);
// kernels: raise when `bad` (any lane), test at the top of the fallback
__device__ __forceinline__ void range_guard_raise(RangeGuard g, bool bad) {
  if (bad) __hip_atomic_store(g.word, g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool range_guard_raised(RangeGuard g) {
  return __hip_atomic_load(g.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g.epoch;
}
// flag word pair of a cached weight image (pack_bodies.cuh: pack_job_block): raised while the image was built from weights
// outside the fp16 piece range
__device__ __forceinline__ bool wimage_flag_raised(const unsigned* wflag) { return wflag && wflag[0] == wflag[1]; }
__device__ __forceinline__ bool not_finite(float v) { return !(fabsf(v) < INFINITY); }

// ---- live-edge view of a padded edge list (round 4: ng_build_live_edges / ng_edge_mlp_*_live) -----------------------
// A padded slot (edges == 0) yields e == 0 and contributes nothing to any gradient (nmrgnn/model.py:251,257,261), yet
// the fused edge kernels ran all four layers on it.  With a view the kernels walk the n_live COMPACTED rows only: row r
// is slot perm[r]; d_src / d_eff are the compacted arrays (every row live), the tape is compacted too (layer stride =
// the slot count, the host-side upper bound), e_out / de keep the caller's [n_slots][E] layout and are reached through
// perm.  n_live is a device scalar: no host synchronisation per batch.  perm == nullptr: no view, every slot is a row.
struct LiveEdges {
  const int32_t* perm = nullptr;     // [n_slots]: the live slots in ascending order, then the dead ones
  const int32_t* n_live = nullptr;   // device scalar, number of live slots
};

// Y = act(rowscale[m] * (X @ W) + b) (+ R);  S (optional) receives the activation output.
int dense_fwd(ng_ctx* ctx, hipStream_t st, int64_t M, int Kin, int Nout, int act, const float* X,
              const float* W, const float* b, const float* rowscale, const float* R, float* Y,
              float* S, const char* tag = "dense_fwd");
// dX[m][k] = (add ? add[m][k] : 0) + sum_n dP[m][n] W[k][n];  dP = dY * act'(S) * rowscale
int dense_dx(ng_ctx* ctx, hipStream_t st, int64_t M, int Kin, int Nout, int act, const float* dY,
             const float* S, const float* rowscale, const float* W, const float* add, float* dX,
             const char* tag = "dense_dx", const float* gscale = nullptr);
// dW[k][n] = sum_m X[m][k] dP[m][n]  (+ db[n] = sum_m dP[m][n] when db != nullptr)
// w_map = 0: dW stored [Kin][Nout];  w_map = 1: MPLayer layout, k = n_e*F + l -> dw[l][m][n_e]
int dense_dw(ng_ctx* ctx, hipStream_t st, int64_t M, int Kin, int Nout, int act, const float* X,
             const float* dY, const float* S, const float* rowscale, float* dW, float* db,
             int w_map, int F, int E, float* scratch, const char* tag = "dense_dw", const float* gscale = nullptr);
size_t dense_dw_scratch_floats(ng_ctx* ctx, int64_t M, int Kin, int Nout, bool has_db);

// split-operand GEMMs on the fp16 matrix pipe (gemm_h2.hip: two fp16 pieces per fp32 operand); same contracts as
// dense_fwd / dense_dx / dense_dw.  Gradient operands are split as S * dP with a power of two S (gemm_grad_scale: one
// pass over dY); gscale = nullptr makes the call compute it, callers that feed the same dP to dx and dw compute it once.
bool gemm_h2_fwd_ok(int64_t M, int K, int N);
int gemm_h2_fwd(ng_ctx* ctx, hipStream_t st, int64_t M, int K, int N, int act, const float* X, const float* W,
                const float* b, const float* rowscale, const float* R, float* Y, float* S, const char* tag, RangeGuard guard);
// gather-GEMM forms of the MPLayer update and of the backward's node-side pull (gemm_h2.hip): no aggregate in HBM
bool mp_gg_supported(int64_t N, int F, int E, int Kpad);
bool mp_gw_infer_ok(ng_ctx* ctx, int64_t N, int K, int F, int E, bool csr, bool keeps_aggregate);   // gemm_h2.hip / mp_gw.cuh
int mp_gg_fwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, int act, int residual, const float* h,
              const int32_t* row_ptr, const int32_t* col, const float* e, const float* inv_degree, const float* w,
              float* h_out, float* s_save, float* A_out = nullptr);
int mp_gg_pull(ng_ctx* ctx, hipStream_t st, int64_t N, int F, int E, const float* dP, const int32_t* csc_ptr,
               const float* csc_rec, const float* w, const float* dh_out, float* dh_in, const float* gscale);
bool gemm_h2_dw_ok(int64_t M, int Kin, int Nout);
bool gemm_h2_dw8_ok(int64_t M, int Kin, int Nout);      // 256 x 256 output tiles, one 8-wave workgroup per CU
int gemm_h2_dw(ng_ctx* ctx, hipStream_t st, int64_t M, int Kin, int Nout, int act, const float* X, const float* dY,
               const float* S, const float* rowscale, float* partial, int nz, int64_t rows_per_z, const float* gscale,
               const char* tag, RangeGuard guard);
// images of an MPLayer weight for the generic GEMMs, packed from w and refreshed behind the weight update (gemm_h2.hip)
int gemm_h2_prepack_mp(ng_ctx* ctx, hipStream_t st, int64_t M, int F, int E, const float* w, const float* Wp, int trans);
int gemm_h2_dx(ng_ctx* ctx, hipStream_t st, int64_t M, int Kin, int Nout, int act, const float* dY, const float* S,
               const float* rowscale, const float* W, const float* add, float* dX, const float* gscale, const char* tag,
               RangeGuard guard);
// {S, 1/S} for the gradient tensor dY [M][N] (and its row scale); *scale_out stays valid until the next call
int gemm_grad_scale(ng_ctx* ctx, hipStream_t st, const float* dY, int64_t M, int N, const float* rowscale,
                    const float** scale_out);
// for kernels that PRODUCE a gradient tensor: write max|dP| of every block to blockmax[blockIdx.x] (<= capacity blocks,
// reduce.cuh: block_max_store), then gemm_grad_scale_from_blocks — no extra pass over the tensor
float* gemm_grad_blockmax(ng_ctx* ctx, int* capacity);
int gemm_grad_scale_from_blocks(ng_ctx* ctx, hipStream_t st, int nblocks, const float** scale_out);
// true when dense_dx / dense_dw of this shape run on gemm_h2 (i.e. a shared gscale is worth computing)
bool dense_grad_uses_h2(int64_t M, int Kin, int Nout);

// fused persistent edge path (edge_fused.hip), edge_hidden_size == 128, edge_fc_layers == 4
bool edge_fused_supported(int H, int E, int Le);
int edge_fused_fwd(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src,
                   const float* d_eff, const float* centers, float gap, const float* const* W,
                   const float* const* b, float* e_out, float* z_save, LiveEdges live = LiveEdges());
int edge_fused_fwd_f32(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src,
                       const float* d_eff, const float* centers, float gap, const float* const* W,
                       const float* const* b, float* e_out, float* z_save, bool tape_blocked, const RangeGuard* guard,
                       LiveEdges live = LiveEdges());

// elementwise helpers (node_ops.hip)
int mp_plain_weights(ng_ctx* ctx, hipStream_t st, int64_t M, int F, int E, const float* w, float* scratch, int trans, const float** Wp);
int mp_repack_w(ng_ctx* ctx, hipStream_t st, int F, int E, const float* w, float* Wp);

// generic MPLayer over CSR lists, or padded lists when row_ptr == nullptr (row i = [i*K, (i+1)*K)); any
// edge_feature_size <= 64 (mp_csr.hip)
int mp_generic_fwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, int act, int residual, const float* h,
                   const int32_t* row_ptr, const int32_t* col, const float* e, const float* inv_degree, const float* w,
                   float* h_out, float* A_save, float* s_save);
int mp_generic_bwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, int act, const float* h,
                   const int32_t* row_ptr, const int32_t* col, const int32_t* row_of, const float* e,
                   const float* inv_degree, const float* w, const float* A_save, const float* s_save,
                   const int32_t* csc_ptr, const int32_t* csc_edge, const float* dh_out, float* dh_in, float* de,
                   int de_accum, float* dw, const float* csc_rec = nullptr, int64_t nnz = 0);
int csr_aggregate(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* h, const int32_t* row_ptr,
                  const int32_t* col, const float* e, float* A);

}  // namespace ng

namespace ng {
// tall-skinny dense products with register-resident weights (tall_gemm.hip)
struct TallArgs {
  int64_t N;
  const float* X;        // [N][ldx]; columns >= k_valid read as 0
  int ldx, k_valid;
  const float* S_in;     // prologue (dP = X * act'(S_in) * rs_in), same layout as X
  const float* rs_in;
  int act_in;
  float* dP_out;         // optional copy of the prologue result
  const float* Wfrag;    // packed by tall_pack / mp_pack
  const float* bias;     // [n_valid] or nullptr
  const float* rowscale; // [N] or nullptr
  int act;
  float* S_save;         // activation output (before the residual) or nullptr
  const float* resid;    // [N][ldo] or nullptr
  float* out;            // [N][ldo]; columns >= n_valid are not written
  int ldo, n_valid;
};
bool tall_gemm_supported(int kpad, int npad);
int tall_pack(ng_ctx* ctx, hipStream_t st, int k_in, int n_out, int kpad, int npad, int sk, int so,
              const float* w, float* out);
int tall_gemm(ng_ctx* ctx, hipStream_t st, int kpad, int npad, const TallArgs& a, bool prologue,
              const char* tag);

// persistent transposed product for weight gradients (tall_tn.hip): dW[ka][kb] = sum_rows A[row][ka] Bp[row][kb]
bool tall_tn_supported(int ka, int kb);
size_t tall_tn_scratch_floats(ng_ctx* ctx, int ka_valid);
int tall_tn(ng_ctx* ctx, hipStream_t st, int64_t N, const float* A, int lda, int ka_valid, const float* B,
            int ldb, int kb_valid, const float* S_in, int act_in, float* dW, float* db, int w_map, int F,
            int E, float* scratch, const char* tag);

// K nearest neighbours through a cell grid, for large frames (knn_cells.hip); the same lists as the brute-force kernels
bool knn_cells_supported(int G, int n, int K);
int knn_cells(ng_ctx* ctx, hipStream_t st, int G, int n, int K, float scale, const float* pos, int32_t* nlist, float* edges,
              float* inv_degree);

// window-resident fused MPLayer kernels (mp_win.hip): atom_feature_size == 64, edge_feature_size <= 3
bool mp_win_supported(int F, int E, int K);
bool mp_win_enabled(int F, int E, int K);
int mpw_pack(ng_ctx* ctx, hipStream_t st, int E, int mode, const float* w, float* out);
int mpw_pack2(ng_ctx* ctx, hipStream_t st, int E, const float* w, int mode_a, float* out_a, int mode_b, float* out_b);
// Contiguous runs of 32-atom tiles per persistent workgroup of the window kernels: a multiple of 8 tiles (256 atoms) so
// that runs start on molecule boundaries for the common 256-atom padding — unless that leaves more than a tenth of the
// CUs without a run (N not a multiple of 256 * num_cu), then the coarsest of 4 / 2 / 1 that does not.
inline int64_t win_tiles_per_wg(int64_t ntiles, int num_cu) {
  const int64_t base = std::max<int64_t>(cdiv(ntiles, num_cu), 1);
  const int64_t want = std::min<int64_t>(num_cu, ntiles);
  for (int align = 8; align > 1; align >>= 1) {
    const int64_t per = cdiv(base, align) * align;
    if (cdiv(ntiles, per) * 10 >= want * 9) return per;
  }
  return base;
}

// the same for the 64-atom tiles of the sixteen-wave kernels: runs of 4 tiles (256 atoms) unless that idles CUs
inline int64_t win16_tiles_per_wg(int64_t ntiles, int num_cu) {
  const int64_t base = std::max<int64_t>(cdiv(ntiles, num_cu), 1);
  const int64_t want = std::min<int64_t>(num_cu, ntiles);
  for (int align = 4; align > 1; align >>= 1) {
    const int64_t per = cdiv(base, align) * align;
    if (cdiv(ntiles, per) * 10 >= want * 9) return per;
  }
  return base;
}

// window-resident neighbour aggregation for F % 128 == 0 (mp_win.hip); padded lists with K % 4 == 0, K <= 16, E <= 3
bool agg_win_supported(int F, int E, int K);
int agg_win_rows();
// the edge gradient de (+)= <dA, h[nlist]> with the same slab windows (same conditions as agg_win)
int egrad_win(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* h, const int32_t* nlist,
              const float* dA, float* de, int accumulate);
int agg_win(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* h, const int32_t* nlist,
            const float* e, float* A);
// backward images, the dA image as fp16 piece fragments (mp_win.hip)
// f32T / f32N (optional): the fp32 fragment images of the same launch, for the guarded fallback kernels
PackJob mpw_bwd_job(int E, const float* w, float* outT, float* outN, float* f32T, float* f32N, unsigned* flag, RangeGuard guard);
// 16-wave form of the edge-side backward window kernel (mp_win16_bwd.hip), launched by mp_win_bwd_edge on its images
bool mp_win16_bwd_edge_supported(int E, int K);
int mp_win16_bwd_edge_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, const float* h, const int32_t* nlist,
                             const float* inv_degree, const float* WfragT, const float* s_save, const float* dh_out, float* dP,
                             float* de, int de_accum, float* dummy, RangeGuard guard, const float* WfragT32, const unsigned* wflag,
                             unsigned wflag_ver);
// 16-wave form of the forward window kernel (mp_win16.hip; default; NG_MP_W16=0: the eight-wave kernels), launched by mp_win_fwd on its images
bool mp_win16_supported(int E, int K);
int mp_win16_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, int residual, const float* h,
                    const int32_t* nlist, const float* e, const float* inv_degree, const float* Wfrag, const float* Wf32,
                    const unsigned* wflag, RangeGuard guard, float* h_out, float* s_save);
// wave-autonomous form of the forward window kernel (mp_wave.hip; round 6; default for E == 3 on batches that fill the chip,
// NG_MP_WAVE=1 / 0: always / never), launched by mp_win_fwd on the same images
bool mp_wave_wanted(const ng_ctx* ctx, int64_t N, int E, int K);
int mp_wave_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int act, int residual, const float* h, const int32_t* nlist,
                   const float* e, const float* inv_degree, const float* Wfrag, const float* Wf32, const unsigned* wflag,
                   RangeGuard guard, float* h_out, float* s_save);
int mp_win_fwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, int residual, const float* h,
               const int32_t* nlist, const float* e, const float* inv_degree, const float* w, float* h_out,
               float* s_save);

// window-resident backward kernels (mp_win_bwd.hip)
bool mp_win_bwd_supported(int F, int E, int K);
int mp_win_bwd_edge(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, const float* h,
                    const int32_t* nlist, const float* inv_degree, const float* WfragT, const float* s_save,
                    const float* dh_out, float* dP, float* de, int de_accum, float* dummy, RangeGuard guard,
                    const float* WfragT32, const unsigned* wflag = nullptr, unsigned wflag_ver = 0);

int mp_win_records(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, const int32_t* csc_ptr,
                   const int32_t* csc_edge, const float* e, float* rec, const int32_t* row_of = nullptr, int64_t n_entries = 0);
// neighbour aggregate over padded lists with the best kernel for the shape (node_ops.hip)
int mp_aggregate_padded(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int F, int E, const float* h, const int32_t* nlist,
                        const float* e, float* A);
size_t mp_win_node_scratch_floats(ng_ctx* ctx, int E);
int mp_win_bwd_node(ng_ctx* ctx, hipStream_t st, int64_t N, int E, const float* h, const float* dP,
                    const int32_t* csc_ptr, const float* rec, const float* WfragN, const float* dh_out,
                    float* dh_in, float* dw, float* scratch, float* dummy, RangeGuard guard, const float* WfragN32,
                    const unsigned* wflag = nullptr, unsigned wflag_ver = 0);

// window-resident MPLayer backward, both kernels (mp_win_bwd.hip)
bool mp_win_bwd_enabled(int F, int E, int K);
int mp_win_bwd(ng_ctx* ctx, hipStream_t st, int64_t N, int K, int E, int act, const float* h, const int32_t* nlist,
               const float* e, const float* inv_degree, const float* w, const float* s_save, const int32_t* csc_ptr,
               const int32_t* csc_edge, const float* dh_out, float* dh_in, float* de, int de_accum, float* dw,
               const float* csc_rec);

// bandwidth-shaped head / embedding kernels (head_ops.hip); NG_HEAD_PATH=generic selects the old ones
bool head_fast_supported(int Fh, int C);
int head_loss_graphs_per_wg(ng_ctx* ctx, int G, int Fh, int C, int64_t max_graph_atoms);
int head_loss_launch(ng_ctx* ctx, hipStream_t st, int64_t N, int G, int Fh, int C, int gpw, const float* g, uint64_t seed,
                     uint64_t offset, float keep, bool draw, float* mask_out, const float* Wout, const float* bout,
                     const float* atoms, const float* pstd, const float* pavg, const int32_t* gptr, const float* y,
                     const float* w, float gweight, float* peaks, float* dg, float* partial);
int head_loss_reduce(ng_ctx* ctx, hipStream_t st, const float* partial, int nb, int Fh, int C, float* dWout, float* dbout,
                     float* loss_out);
bool head_fwd_fast_supported(int Fh, int C);
int head_fwd_fast(ng_ctx* ctx, hipStream_t st, int64_t N, int Fh, int C, const float* g, const float* mask,
                  const float* Wout, const float* bout, const float* atoms, const float* pstd,
                  const float* pavg, float* peaks,
                  uint64_t seed = 0, uint64_t offset = 0, float keep = 1.f, float* mask_out = nullptr);
int head_bwd_fast(ng_ctx* ctx, hipStream_t st, int64_t N, int Fh, int C, const float* g, const float* mask,
                  const float* Wout, const float* atoms, const float* pstd, const float* dpeaks, float* dg,
                  float* dWout, float* dbout);
bool embed_bwd_fast_supported(int F, int C);
int embed_bwd_fast(ng_ctx* ctx, hipStream_t st, int64_t N, int C, int F, const float* atoms, const float* dh0,
                   float* dWemb);

}  // namespace ng
