/* nmrgnn_hip.h — C ABI of libnmrgnn_hip.so: the MI355X (gfx950) engine for nmrgnn's
 * message-passing hot path.
 *
 * The reference (ur-whitelab/nmrgnn v0.7) is pure Python/TensorFlow and has NO plugin /
 * FFI / custom-op interface; this ABI is therefore defined here, one entry point per stage
 * of GNNModel.call (nmrgnn/model.py:245-274), and each declaration cites the reference
 * lines it replaces.  INTEGRATION.md shows the reference-side (ctypes) binding.
 *
 * Conventions
 *   - every pointer named *_dev (and every float* / int32_t* tensor argument) is a DEVICE pointer owned
 *     by the caller (torch allocations in the Python host); the library allocates only the
 *     opaque ng_ctx (scratch workspace, profiling events).
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream
 *     and never synchronises the device.
 *   - return 0 = NG_OK, negative = error; message via ng_last_error(ctx).  Nothing throws.
 *   - all tensors are fp32 row-major; index tensors are int32.
 *   - a ctx is not thread-safe: one per host thread / GPU.
 */
#ifndef NMRGNN_HIP_H
#define NMRGNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NG_ABI_VERSION 9

enum {
  NG_OK = 0,
  NG_ERR_INVALID = -1, /* bad shape / argument */
  NG_ERR_HIP = -2,     /* HIP runtime error */
  NG_ERR_NOMEM = -3,
  NG_ERR_UNSUPPORTED = -4
};

/* keras activation names of hypers mp_activation / fc_activation (nmrgnn/model.py:33-36) */
enum { NG_ACT_NONE = 0, NG_ACT_SOFTPLUS = 1, NG_ACT_RELU = 2, NG_ACT_TANH = 3 };

typedef struct ng_ctx ng_ctx;

int ng_abi_version(void);
int ng_ctx_create(int device, ng_ctx** out);
void ng_ctx_destroy(ng_ctx* ctx);
const char* ng_last_error(ng_ctx* ctx);
/* The NG_* path switches (NG_EDGE_MATH, NG_GEMM_MATH, NG_MP_PATH, ... — listed in csrc/ng_common.h) are parsed from
 * the environment once per process; ng_reload_env() parses them again (tests / A-B tools that flip one in-process). */
int ng_reload_env(void);
/* Inference with constant weights: between ng_weights_frozen(ctx, owner != 0) and ng_weights_frozen(ctx, 0) the library
 * keeps its packed weight images (MFMA fragment orders, fp16-piece images) across calls instead of re-packing them on
 * every call.  The images are keyed by the weight tensors' addresses, so `owner` names the model they belong to: a
 * call with a different non-zero owner discards what the cache holds (another model's freed weights may have had the
 * same addresses).  A change of a weight tensor of the current owner must be announced with ng_weights_changed(ctx)
 * (ng_adam_step does so itself).  Off (owner 0) is the default; calls made while it is off never touch the cache.
 * Contract while frozen: every weight pointer handed to an entry point keeps its contents until ng_weights_changed /
 * the end of the window (a temporary tensor whose address is later reused for other weights must be announced the
 * same way).  Weights the library repacks into its own scratch inside a call (the MPLayer backward's Wp) are never
 * cached, so backward entry points may run inside a frozen window. */
int ng_weights_frozen(ng_ctx* ctx, int owner);
int ng_weights_changed(ng_ctx* ctx);
/* Deferred weight-gradient sums (ABI 5).  The backward entry points end in a second-stage reduction of per-workgroup
 * partials into the gradient tensor (deterministic, fixed order): seven launches of 5-12 us in a training step's
 * backward (head, FC block, four MPLayers, embedding; nmrgnn/model.py:262-273 differentiated).  Between
 * ng_defer_reductions(ctx, stream, 1) and the matching ng_defer_reductions(ctx, stream, 0) the entry points keep their
 * partials in a context-owned arena and only QUEUE the reduction; ng_flush_reductions(ctx, stream) runs everything
 * queued so far in one launch on `stream` (switching deferral off flushes too).  A gradient tensor is defined only
 * after the flush that follows its entry point; the bits are those of the eager form.  Off by default.
 * ABI 9: inside such a window ng_embed_bwd of a molecule-sized call (N <= 2048) launches nothing of its own — its sum
 * is a job of the flush launch that reads `atoms` and `dh0` THEN: both must stay valid and unchanged until the flush
 * (the same holds for the caller-owned `partial` of ng_head_loss_reduce). */
int ng_defer_reductions(ng_ctx* ctx, void* stream, int on);
int ng_flush_reductions(ng_ctx* ctx, void* stream);
/* Hint about the batch the following calls work on: the largest number of atoms of one member graph (the reference
 * concatenates molecules with offset neighbour indices, nmrgnn/library.py:106-117, so a neighbour index lies within its
 * own graph).  0 = unknown (default).  With 0 < span <= 272 the default-width neighbour aggregation (F % 128 == 0) keeps
 * slab windows of the gathered rows in LDS; larger or unknown spans (whole proteins) take the L2-gather kernel.  With
 * 0 < span <= 256, F = 256, E = 3 and no aggregate kept (a_save == NULL: inference) ng_mp_layer_fwd(_csr) runs the window
 * gather-GEMM (csrc/mp_gw.cuh: the aggregate never reaches HBM); the backward's scatter-sum takes its LDS row-block form for
 * span <= 288.  Every kernel checks per tile that its window really holds the tile's sources and reads the others from
 * memory: results do not depend on the hint, a wrong hint only costs time. */
int ng_ctx_set_graph_span(ng_ctx* ctx, int64_t max_graph_atoms);
/* pre-size the scratch workspace (so that later calls never hipMalloc, e.g. under graph capture) */
int ng_ctx_reserve(ng_ctx* ctx, uint64_t bytes);

/* ---- graph replay of small calls (ABI 7; ng_replay_token / ng_replay_commit: ABI 8) -----------------------------------------------------------------------------
 * The reference trains on ONE graph per step (nmrgnn/library.py:88-89, nmrgnn/main.py:74-80) and predicts one structure per
 * call (main.py:236-245): ~33 / ~8 launches of a few microseconds each, bound by the host's launch rate.  Every entry point of
 * this library is asynchronous on the stream it is given and allocates nothing once its scratch is sized (a warm-up call, or
 * ng_ctx_reserve), so a chain of calls can be CAPTURED on that stream (hipStreamBeginCapture, torch.cuda.CUDAGraph) and
 * replayed as one graph launch per shape.  Three launch arguments of a training step change from step to step and would be
 * frozen into the captured nodes: the seed of ng_add_noise(_live) and ng_head_fwd_dropout, and ng_adam_step's bias-corrected
 * rate.  While a context is ARMED those launches ignore these arguments and read the values ng_replay_stage last wrote to
 * a device block of the context:
 *   ng_replay_arm    on: arm before capturing (and leave armed while replaying: the captured nodes hold the block's address);
 *                    off: eager calls take their arguments again.
 *   ng_replay_stage  ONE eager launch per replayed step, before the graph launch: stores `seed` and Adam's rate for `step`
 *                    (the expression of ng_adam_step: same bits), clears the operand-range guard word, and copies up to 8
 *                    buffers (whole 32-bit words, device to device) — the step's inputs into the static buffers the captured
 *                    chain reads.  Inference replays need it only for the copies (seed / step are then ignored: pass step 1).
 * A replayed TRAINING step changes the weights on the device while none of the library's host code runs: ng_adam_step's
 * bookkeeping of the packed-weight-image cache (the weight version, which images its launch rebuilt) happened once, at capture.
 *   ng_replay_token   after the capture: identifies the set of cached images the captured ng_adam_step rebuilds (0: none).
 *   ng_replay_commit  after EVERY replay of a captured training step: advances the context's weight version as one
 *                     ng_adam_step does and marks exactly the images of `token` as current; every other cached image —
 *                     built by an eager call of another shape, by a big validation batch, by another captured step — is
 *                     rebuilt at its next use instead of being served with the weights of N steps ago.  An unknown token
 *                     marks nothing.  Inference replays (no weight change) do not call it.
 * nmrgnn_amd/replay.py holds the Python side (TrainStepReplay, ForwardReplay). */
int ng_replay_arm(ng_ctx* ctx, int on);
int ng_replay_stage(ng_ctx* ctx, void* stream, uint64_t seed, float lr, float beta1, float beta2, int64_t step, int n_copies,
                    const void* const* src, void* const* dst, const uint64_t* bytes);
int ng_replay_token(ng_ctx* ctx, uint64_t* token);
int ng_replay_commit(ng_ctx* ctx, uint64_t token);

/* per-kernel hipEvent bracketing for bench.py's roofline leg */
int ng_prof_enable(ng_ctx* ctx, int on);
int ng_prof_reset(ng_ctx* ctx);
/* synchronises the recorded events; returns number of distinct kernel names (<= cap).
 * names[i] points to a static string; total_ms[i] / count[i] aggregate launches. */
int ng_prof_read(ng_ctx* ctx, int cap, const char** names, double* total_ms, int64_t* count);

/* ---- RNG (explicit draws so that tests can feed the same numbers to the oracle) ------------ */
/* xi ~ N(0,1): the draw inside keras GaussianNoise, nmrgnn/model.py:213,253 */
int ng_randn(ng_ctx*, void* stream, uint64_t seed, uint64_t offset, float* out, int64_t n);
/* keras Dropout keep-mask, nmrgnn/model.py:216-219,266-267: out = 1/keep with prob keep, else 0 */
int ng_dropout_mask(ng_ctx*, void* stream, uint64_t seed, uint64_t offset, float keep, float* out,
                    int64_t n);

/* out = x + alpha*xi with the xi of ng_randn(seed, offset): GaussianNoise (nmrgnn/model.py:213,253) in one launch
 * instead of ng_randn + ng_add_scaled; the bits are those of the two-call form */
int ng_add_noise(ng_ctx*, void* stream, uint64_t seed, uint64_t offset, int64_t n, const float* x, float alpha,
                 float* out);
/* out = x + alpha*y : applies the GaussianNoise draw, d_eff = d + sigma*xi (nmrgnn/model.py:253) */
int ng_add_scaled(ng_ctx*, void* stream, int64_t n, const float* x, const float* y, float alpha,
                  float* out);

/* RBFExpansion alone, nmrgnn/layers.py:137-140: out[n,H] = (d_src>0) * exp(-(d_eff-centers)^2/gap)
 * (pass d_src = d_eff for the unmasked layer on positive distances) */
int ng_rbf_expand(ng_ctx*, void* stream, int64_t n, int H, const float* d_src, const float* d_eff,
                  const float* centers, float gap, float* out);

/* ---- edge path: mask + RBFExpansion + EdgeFCBlock ------------------------------------------
 * replaces nmrgnn/model.py:251-261, nmrgnn/layers.py:137-140, nmrgnn/model.py:132-138.
 *   d_src  [n_edges]  raw distances (mask = d_src > 0)
 *   d_eff  [n_edges]  distances fed to the RBF (= d_src, or d_src + sigma*xi when training)
 *   centers[H], gap   RBF grid (layers.py:126-129)
 *   W[t] [in,out], b[t] [out]  host arrays (length Le) of device pointers, Keras Dense layout
 *   act    activation of the hidden layers = hypers fc_activation (model.py:35-36,123): NG_ACT_SOFTPLUS runs the fused
 *          kernels (H = 128, Le = 4, E <= 8); other codes, other shapes and E up to 256 (E % 4 == 0) the layered path
 *   e_out  [n_edges,E]
 *   z_save [Le-1, n_edges, H] softplus outputs of the hidden layers (NULL for inference): the tape handed to
 *          ng_edge_mlp_bwd; its element order inside a layer is given by ng_edge_tape_layout()
 */
int ng_edge_mlp_fwd(ng_ctx*, void* stream, int64_t n_edges, int H, int E, int Le, int act,
                    const float* d_src, const float* d_eff, const float* centers, float gap,
                    const float* const* W, const float* const* b, float* e_out, float* z_save);
/* Element order of a z_save layer for this shape (same footprint either way; depends on NG_EDGE_* switches, so ask
 * in the process that runs the kernels):
 *   0: row-major [n_edges][H]
 *   1: inside every FULL group of 32 consecutive edges the 32 x 128 block is stored in the kernels' register layout:
 *      edge r (0..31), feature 32*bo + 8*q + 4*hf + j  ->  float ((bo*4 + q)*64 + hf*32 + r)*4 + j of the group's
 *      4096; a last partial group is row-major.  (Both kernels then move whole contiguous KBs per wave.) */
int ng_edge_tape_layout(int H, int E, int Le, int act, int64_t n_edges);
/* de [n_edges,E] upstream gradient; writes dW[t], db[t] (overwrites).
 * ng_edge_mlp_bwd_tape: tape_layout = the value ng_edge_tape_layout() returned when the forward wrote z_save (the
 * layout then no longer depends on the switches in force at backward time); ng_edge_mlp_bwd == tape_layout -1 (ask again). */
int ng_edge_mlp_bwd(ng_ctx*, void* stream, int64_t n_edges, int H, int E, int Le, int act,
                    const float* d_src, const float* d_eff, const float* centers, float gap,
                    const float* const* W, const float* z_save, const float* de,
                    float* const* dW, float* const* db);
int ng_edge_mlp_bwd_tape(ng_ctx*, void* stream, int64_t n_edges, int H, int E, int Le, int act,
                         const float* d_src, const float* d_eff, const float* centers, float gap,
                         const float* const* W, const float* z_save, const float* de,
                         float* const* dW, float* const* db, int tape_layout);

/* ---- the edge path over the LIVE slots of a padded list (ABI 6) -----------------------------------------------
 * A padded slot (edges == 0) yields e == 0 and contributes to no gradient: the mask multiplies the RBF and the MLP
 * output (nmrgnn/model.py:251,257,261).  The fused edge kernels can therefore skip such slots entirely: results are the
 * ones of ng_edge_mlp_fwd / _bwd (e bit for bit; weight gradients up to the summation order over edges).
 *   ng_build_live_edges: a stable partition of the n_slots = N*K slots, once per batch, no host synchronisation:
 *     perm  [n_slots]  the live slots (edges > 0) in ascending order, then the dead ones
 *     pos   [n_slots]  row of slot g in the compacted order, -1 for a dead slot
 *     d_c   [n_slots]  d_c[r] = edges[perm[r]] for r < n_live: the compacted distances
 *     n_live[1]        device scalar
 *   ng_add_noise_live: GaussianNoise into the compacted order, out_c[pos[g]] = x[g] + alpha * xi_g with the xi_g of
 *     ng_add_noise(seed, offset) (y == NULL) or xi_g = y[g] (explicit draws); dead slots are skipped.
 *   ng_edge_mlp_fwd_live / _bwd_live: d_src_c / d_eff_c are the COMPACTED distances (n_live rows; d_eff_c = d_src_c
 *     for inference); e_out and de keep the caller's [n_slots, E] layout (dead slots of e_out are written 0);
 *     z_save [Le-1, n_slots, H]: layer stride as for n_slots rows, the first n_live rows of every layer used.
 *   ng_edge_live_supported: 1 when this shape runs the fused edge path (the only one with a live view).
 *   *n_live < 0 (ABI 8): the launch has nothing to do and returns at once — _fwd_live leaves e_out (dead slots included) and
 *     z_save untouched, _bwd_live writes dW = db = 0.  What the edge-function table's guard hands the per-edge launches when it
 *     is down (ng_edge_table_check: gate[1]). */
int ng_build_live_edges(ng_ctx*, void* stream, int64_t n_slots, const float* edges, int32_t* perm, int32_t* pos,
                        float* d_c, int32_t* n_live);
/* ng_build_incoming_lists (padded form) and ng_build_live_edges as ONE call (ABI 9): the same outputs; one launch when
 * ng_graph_lists_one_launch(N, K) != 0 (molecule-sized calls, the reference's own granularity: nmrgnn/library.py:88-89) */
int ng_build_graph_lists(ng_ctx*, void* stream, int64_t N, int K, const int32_t* nlist, const float* edges, int32_t* nlist_c,
                         int32_t* csc_ptr, int32_t* csc_edge, int32_t* perm, int32_t* pos, float* d_c, int32_t* n_live);
int ng_graph_lists_one_launch(int64_t N, int K);
int ng_add_noise_live(ng_ctx*, void* stream, uint64_t seed, uint64_t offset, int64_t n, const float* x, const float* y,
                      float alpha, const int32_t* pos, float* out_c);
int ng_edge_live_supported(int H, int E, int Le, int act);
int ng_edge_mlp_fwd_live(ng_ctx*, void* stream, int64_t n_slots, int H, int E, int Le, int act,
                         const float* d_src_c, const float* d_eff_c, const int32_t* perm, const int32_t* n_live,
                         const float* centers, float gap, const float* const* W, const float* const* b,
                         float* e_out, float* z_save);
int ng_edge_mlp_bwd_live(ng_ctx*, void* stream, int64_t n_slots, int H, int E, int Le, int act,
                         const float* d_src_c, const float* d_eff_c, const int32_t* perm, const int32_t* n_live,
                         const float* centers, float gap, const float* const* W, const float* z_save,
                         const float* de, float* const* dW, float* const* db, int tape_layout);

/* ---- node path ----------------------------------------------------------------------------- */
/* embed_layer, nmrgnn/model.py:241,262: h0 = atoms[N,C] @ Wemb[C,F] */
int ng_embed_fwd(ng_ctx*, void* stream, int64_t N, int C, int F, const float* atoms,
                 const float* Wemb, float* h0);
int ng_embed_bwd(ng_ctx*, void* stream, int64_t N, int C, int F, const float* atoms,
                 const float* dh0, float* dWemb);

/* MPLayer aggregation, nmrgnn/layers.py:33 + the (i,j)-contraction of layers.py:39-40:
 *   A[i,n,l] = sum_j e[i,j,n] * h[nlist[i,j], l]        A is [N,E,F] */
int ng_mp_aggregate(ng_ctx*, void* stream, int64_t N, int K, int F, int E, const float* h,
                    const int32_t* nlist, const float* e, float* A);

/* MPLayer + residual, nmrgnn/layers.py:26-46 and nmrgnn/model.py:165-167:
 *   P = inv_degree * einsum('ijn,ijl,lmn->im', e, h[nlist], w);  h_out = act(P) (+ h if residual:
 *   MPBlock, model.py:167; residual = 0 gives the bare MPLayer of layers.py:26-46)
 *   w is the reference layout [F,F,E].  A_save [N,E,F] and s_save [N,F] (= act(P)) are written
 *   when non-NULL (training). */
int ng_mp_layer_fwd(ng_ctx*, void* stream, int64_t N, int K, int F, int E, int act, int residual,
                    const float* h, const int32_t* nlist, const float* e, const float* inv_degree,
                    const float* w, float* h_out, float* A_save, float* s_save);
/* 1 when the backward pass of the kernel path selected for (F, E, K) reads the forward aggregate
 * (pass a [N,E,F] buffer as A_save to ng_mp_layer_fwd and hand it to ng_mp_layer_bwd), 0 when it does
 * not (pass NULL to both: the weight gradient is formed from h and the incoming-edge aggregate of dP).
 * A_save == NULL is always accepted by ng_mp_layer_bwd; paths that need the aggregate rebuild it. */
int ng_mp_layer_wants_aggregate(int F, int E, int K);

/* backward of the above.  csc_ptr[N+1], csc_edge[nnz]: incoming-edge lists (edge id = i*K+j
 * grouped by target nlist[i,j]); dh_out [N,F] upstream; writes dh_in (overwrite),
 * de (accumulate if de_accum else overwrite), dw [F,F,E] (overwrite). */
int ng_mp_layer_bwd(ng_ctx*, void* stream, int64_t N, int K, int F, int E, int act,
                    const float* h, const int32_t* nlist, const float* e, const float* inv_degree,
                    const float* w, const float* A_save, const float* s_save,
                    const int32_t* csc_ptr, const int32_t* csc_edge, const float* dh_out,
                    float* dh_in, float* de, int de_accum, float* dw);

/* Incoming-edge records for ng_mp_layer_bwd_rec: rec[p] = { source atom of CSC entry p (int bits),
 * e[edge p][0..2] }, 16 bytes per entry, nnz = csc_ptr[N] entries (buffer: 4*N*K floats).  The edge
 * features are the same for every MPLayer of a backward pass, so the records are built once and shared. */
int ng_mp_edge_records(ng_ctx*, void* stream, int64_t N, int K, int E, const int32_t* csc_ptr,
                       const int32_t* csc_edge, const float* e, float* rec);
/* ng_mp_layer_bwd with the records supplied (csc_rec may be NULL: they are then rebuilt per call).  The lists of
 * ng_build_incoming_lists hold every target's entries in ascending order of the SOURCE atom; the default-width scatter-sum
 * stages source rows block by block in that order.  Caller-built lists in another order give the same sums — entries behind
 * the staged block are read from memory — at a lower rate. */
int ng_mp_layer_bwd_rec(ng_ctx*, void* stream, int64_t N, int K, int F, int E, int act,
                    const float* h, const int32_t* nlist, const float* e, const float* inv_degree,
                    const float* w, const float* A_save, const float* s_save,
                    const int32_t* csc_ptr, const int32_t* csc_edge, const float* dh_out,
                    float* dh_in, float* de, int de_accum, float* dw, const float* csc_rec);

/* ---- CSR (variable-degree) form of the neighbour lists, SURVEY 8(b) ------------------------------------------
 * The reference's padded [N,K] tuple (nmrgnn/library.py:106-117) with the edges == 0 slots dropped — they contribute
 * exactly 0 through the edge mask (nmrgnn/model.py:251,261):
 *   row_ptr [N+1]  row i owns the entries p in [row_ptr[i], row_ptr[i+1])
 *   col     [nnz]  neighbour atom (batch-global index)          dist [nnz] distance
 * The edge path runs on the flat list: ng_edge_mlp_fwd(n_edges = nnz, d_src = dist, ...) -> e [nnz,E].
 * Same contract as ng_mp_aggregate / ng_mp_layer_fwd / ng_mp_layer_bwd (layers.py:26-46, model.py:165-167); on lists
 * that differ only by dropped zero-weight slots the results equal the padded generic path (NG_MP_PATH=layered) bit for
 * bit.  row_of [nnz] = row of entry p;  csc_ptr [N+1] / csc_edge [nnz] = entries grouped by col (ascending p inside a
 * group), the incoming-edge lists of the deterministic backward scatter. */
int ng_mp_aggregate_csr(ng_ctx*, void* stream, int64_t N, int F, int E, const float* h,
                        const int32_t* row_ptr, const int32_t* col, const float* e, float* A);
int ng_mp_layer_fwd_csr(ng_ctx*, void* stream, int64_t N, int64_t nnz, int F, int E, int act, int residual,
                        const float* h, const int32_t* row_ptr, const int32_t* col, const float* e,
                        const float* inv_degree, const float* w, float* h_out, float* A_save, float* s_save);
int ng_mp_layer_bwd_csr(ng_ctx*, void* stream, int64_t N, int64_t nnz, int F, int E, int act, const float* h,
                        const int32_t* row_ptr, const int32_t* col, const int32_t* row_of, const float* e,
                        const float* inv_degree, const float* w, const float* A_save, const float* s_save,
                        const int32_t* csc_ptr, const int32_t* csc_edge, const float* dh_out, float* dh_in,
                        float* de, int de_accum, float* dw);

/* ---- molecule-sized inference: FC block + head in one launch ------------------------------------------------------
 * nmrgnn/model.py:191-196 (FCBlock: L-1 residual softplus Dense F -> F, one softplus Dense F -> F/2) followed by
 * model.py:268-273 (out Dense F/2 -> C, full * std + avg, one-hot select) for one protein-sized graph (the reference's
 * eval-struct loop, main.py:236-245): peaks[N] from the MP block's output x[N][F], activations never leave the CU.
 * Inference only (no dropout, no tape).  Supported: F == 256, L == 4, act softplus, C <= 16, N <= 16384; anything else
 * returns NG_ERR_UNSUPPORTED and the caller uses ng_fc_block_fwd + ng_head_fwd.  Rows with a feature beyond the fp16
 * range of the split-operand products are recomputed in plain fp32 inside the kernel. */
/* One MPLayer forward (nmrgnn/layers.py:26-46; the call of ng_mp_layer_fwd without saved tensors) for a molecule-sized
 * graph in one launch: the neighbour aggregate of a workgroup's 32 atoms is formed in LDS and multiplied at once.
 * Supported: F == 256, E <= 3, K <= 32, N <= 16384, h != h_out; otherwise NG_ERR_UNSUPPORTED (use ng_mp_layer_fwd). */
int ng_mp_layer_fwd_short(ng_ctx*, void* stream, int64_t N, int K, int F, int E, int act, int residual, const float* h,
                          const int32_t* nlist, const float* e, const float* inv_degree, const float* w, float* h_out);
int ng_mp_layer_fwd_short_csr(ng_ctx*, void* stream, int64_t N, int F, int E, int act, int residual, const float* h,
                              const int32_t* row_ptr, const int32_t* col, const float* e, const float* inv_degree,
                              const float* w, float* h_out);      /* the same over CSR lists, any degree */
int ng_mp_layer_short_ok(int64_t N, int K, int F, int E);        /* 1: ng_mp_layer_fwd_short takes this shape now */
int ng_fc_head_ok(int64_t N, int F, int L, int C, int act);      /* 1: ng_fc_head_fwd takes this shape now */
int ng_fc_head_fwd(ng_ctx*, void* stream, int64_t N, int F, int L, int C, int act, const float* x,
                   const float* const* W, const float* const* b, const float* Wout, const float* bout,
                   const float* atoms, const float* pstd, const float* pavg, float* peaks);

/* ---- per-batch graph preprocessing: the incoming-edge lists of the deterministic backward scatter --------------
 * The reference hands the model a NEW graph tuple every step (nmrgnn/library.py:88-89, main.py:79); its backward is
 * TensorFlow's unsorted scatter-add behind tf.gather (layers.py:33).  The engine's backward pulls over incoming edges
 * instead, which needs, per batch, the transposed lists
 *   csc_ptr [N+1], csc_edge [<= n_entries]: the entries (eid = i*K + j for padded lists with K > 0 and slots
 *   edges[eid] == 0 dropped; eid = CSR entry index when K == 0, edges may be NULL = every entry live) whose neighbour
 *   is atom t, for t = 0..N-1, ascending eid inside a target (= a stable sort by target), csc_ptr[N] = number of
 *   live entries; and, for padded lists, nlist_c [N*K] (may be NULL): nlist with padded slots replaced by the atom's
 *   own index (their weight is exactly 0), the list the MP kernels gather through.
 * A deterministic counting sort in HIP; scratch comes from the context.  All pointers are device pointers. */
int ng_build_incoming_lists(ng_ctx*, void* stream, int64_t N, int K, int64_t n_entries, const int32_t* nlist,
                            const float* edges, int32_t* nlist_c, int32_t* csc_ptr, int32_t* csc_edge);
size_t ng_incoming_lists_scratch_bytes(int64_t N, int64_t n_entries);

/* Distance-cutoff graph builder (BASELINE configs[4], "variable degree"; the counterpart of ng_knn_graph for the CSR
 * form, in front of nmrgnn/library.py:106-117): every OTHER atom of the same frame closer than `cutoff` (Angstrom).
 *   ng_cutoff_count: deg [G*n] neighbours per atom  ->  the caller's exclusive prefix sum is row_ptr [G*n+1]
 *   ng_cutoff_fill : col [nnz] batch-global (frame*n + j), ascending j inside a row; dist [nnz] = distance*scale;
 *                    inv_degree [G*n] = 1/#(local neighbour index > 0), 0 when none (library.py:115-116) */
int ng_cutoff_count(ng_ctx*, void* stream, int G, int n, float cutoff, const float* pos, int32_t* deg);
int ng_cutoff_fill(ng_ctx*, void* stream, int G, int n, float cutoff, float scale, const float* pos,
                   const int32_t* row_ptr, int32_t* col, float* dist, float* inv_degree);
/* ng_cutoff_fill that also writes row_of[nnz], the row of every entry (NULL: not written) */
int ng_cutoff_fill_rows(ng_ctx*, void* stream, int G, int n, float cutoff, float scale, const float* pos,
                        const int32_t* row_ptr, int32_t* col, float* dist, float* inv_degree, int32_t* row_of);
/* out[0..n] = exclusive prefix sums of in[0..n-1], out[n] = total (row_ptr from ng_cutoff_count's degrees) */
int ng_exclusive_scan_i32(ng_ctx*, void* stream, int64_t n, const int32_t* in, int32_t* out);

/* ---- graph front end: K nearest neighbours per atom, per frame --------------------------------
 * Replaces the neighbour search behind nmrgnn.universe2graph (nmrgnn/library.py:106-117, external
 * nmrdata.parse_universe) and the per-frame graph construction of eval-struct (main.py:236-243).
 *   pos [G][n][3] (Angstrom) -> nlist [G*n][K] batch-global indices (frame*n + j), self excluded,
 *   ascending distance, ties -> lower index; edges [G*n][K] = distance*scale; unused slots (0, 0.0);
 *   inv_degree [G*n] = 1/#(local neighbour index > 0), 0 when none (library.py:115-116).  K <= 64. */
int ng_knn_graph(ng_ctx*, void* stream, int G, int n, int K, float scale, const float* pos,
                 int32_t* nlist, float* edges, float* inv_degree);

/* AMPLayer attention aggregation, nmrgnn/layers.py:89-96 (the layer is exported by the reference package but not
 * used by its model):
 *   b[i,:] = softmax_j( inv[i] * <e[i,j,:] @ wk, h[i,:] @ wq> ),  agg[i,:] = sum_j b[i,j] * h[nlist[i,j],:]
 * The layer output is act(agg @ wv) = ng_dense_fwd(agg, wv, bias 0).  K <= 64, E <= 64. */
int ng_amp_attend(ng_ctx*, void* stream, int64_t N, int K, int F, int E, const float* h,
                  const int32_t* nlist, const float* e, const float* inv_degree, const float* wq,
                  const float* wk, float* agg);
/* Backward of ng_amp_attend (the gradient TensorFlow derives for nmrgnn/layers.py:89-96): from dagg = d loss/d agg
 * it writes dh [N,F] (both uses of the node features: the gathered values and the query), de [N,K,E], dwq [F,E] and
 * dwk [E,E].  in_ptr [N+1] / in_slot [N*K] list, for every atom t, the slots i*K+j of nlist that hold t (ALL slots:
 * the reference's softmax runs over padded slots too, layers.py:94); the sums follow the list order, no atomics.
 * The wv product in front is ng_dense_bwd. */
int ng_amp_attend_bwd(ng_ctx*, void* stream, int64_t N, int K, int F, int E, const float* h,
                      const int32_t* nlist, const float* e, const float* inv_degree, const float* wq,
                      const float* wk, const int32_t* in_ptr, const int32_t* in_slot, const float* dagg,
                      float* dh, float* de, float* dwq, float* dwk);

/* FCBlock, nmrgnn/model.py:179-196, all layers in one call:
 *   x_{l+1} = act(x_l @ W[l] + b[l]) + x_l   for l < L-1  (F -> F),   g = act(x_{L-1} @ W[L-1] + b[L-1])  (F -> F/2)
 * y[l] receives x_{l+1} (l = 0 .. L-2; the tape of the backward pass), g the block output [N,F/2].
 * W / b / y are host arrays of device pointers.  F == 64 runs one fused kernel; other sizes loop over the
 * per-layer kernels. */
int ng_fc_block_fwd(ng_ctx*, void* stream, int64_t N, int F, int L, int act, const float* x,
                    const float* const* W, const float* const* b, float* const* y, float* g);

/* backward of ng_fc_block_fwd.  x[l] = layer inputs (x[0] = block input, x[l+1] = y[l] of the forward call),
 * g = block output, dg its gradient; writes dx (gradient w.r.t. x[0]) and dW[l], db[l] (overwrites).  The
 * activation outputs are rebuilt as x[l+1] - x[l].  scratch: ng_fc_block_scratch_floats(N, F, L) floats. */
/* floats of scratch ng_fc_block_bwd needs for this shape (0 when the fused kernel handles it) */
int64_t ng_fc_block_scratch_floats(int64_t N, int F, int L);
int ng_fc_block_bwd(ng_ctx*, void* stream, int64_t N, int F, int L, int act, const float* const* x,
                    const float* g, const float* const* W, const float* dg, float* dx, float* const* dW,
                    float* const* db, float* scratch);

/* keras Dense (+ residual), nmrgnn/model.py:191-196:  Y = act(X@W + b) (+ X if residual)
 *   s_save [M,Nout] = act(X@W+b) written when non-NULL */
int ng_dense_fwd(ng_ctx*, void* stream, int64_t M, int Kin, int Nout, int act, int residual,
                 const float* X, const float* W, const float* b, float* Y, float* s_save);
/* dX = (residual ? dY : 0) + dP @ W^T,  dW = X^T dP,  db = colsum dP,
 * dP = dY * act'(.) recovered from s_save (softplus: 1-exp(-s)).  dX may be NULL. */
int ng_dense_bwd(ng_ctx*, void* stream, int64_t M, int Kin, int Nout, int act, int residual,
                 const float* X, const float* W, const float* s_save, const float* dY, float* dX,
                 float* dW, float* db);

/* dropout + out_layer + de-standardisation, nmrgnn/model.py:266-273:
 *   peaks[i] = sum_c atoms[i,c]*((g*mask)[i,:]@Wout[:,c] + bout[c])*std[c] + atoms[i,c]*avg[c] */
int ng_head_fwd(ng_ctx*, void* stream, int64_t N, int Fh, int C, const float* g,
                const float* drop_mask, const float* Wout, const float* bout, const float* atoms,
                const float* peak_std, const float* peak_avg, float* peaks);
/* the same with the keras Dropout mask (model.py:216-219,266-267) drawn inside the launch: mask_out[N,Fh] receives the
 * values of ng_dropout_mask(seed, offset, keep) over the N*Fh elements (the backward reads them) */
int ng_head_fwd_dropout(ng_ctx*, void* stream, int64_t N, int Fh, int C, const float* g, uint64_t seed, uint64_t offset,
                        float keep, float* mask_out, const float* Wout, const float* bout, const float* atoms,
                        const float* peak_std, const float* peak_avg, float* peaks);
int ng_head_bwd(ng_ctx*, void* stream, int64_t N, int Fh, int C, const float* g,
                const float* drop_mask, const float* Wout, const float* atoms,
                const float* peak_std, const float* dpeaks, float* dg, float* dWout, float* dbout);

/* Head forward + NameLoss with s = 1 (ng_loss_l2) + head backward in ONE launch (ABI 9).  Replaces, for a training step whose
 * loss is the L2 NameLoss, the chain ng_head_fwd_dropout -> ng_loss_l2 -> ng_head_bwd (nmrgnn/model.py:266-273,
 * nmrgnn/losses.py:30-39): a workgroup owns whole graphs, so the graph's loss and dloss/dpeaks never leave the chip.
 * peaks and dg carry the bits of the three-call chain; dWout / dbout and the mean over graphs are summed in a different order
 * (last bits).  The loss is COMPLETE ONLY AFTER ng_head_loss_reduce (its first stage is a column of `partial`).
 *   ng_head_loss_blocks  number of workgroups = rows of `partial` the call writes (0: shape not supported — a graph longer
 *                        than 256 atoms, more than 16 elements, more graphs than the chip takes in one round of workgroups,
 *                        or a head shape ng_head_bwd's fast form does not take; use the three-call chain)
 *   keep < 1: the keras Dropout mask of ng_dropout_mask(seed, offset, keep) is applied (mask_out[N,Fh] optional: may be NULL);
 *   keep == 1: no dropout.  grad_weight multiplies dloss/dpeaks (uneven data-parallel shards), 1 = none.
 *   partial [blocks][Fh*C + C + 1]: first-stage sums of [dWout ; dbout ; loss]; ng_head_loss_reduce finishes them (queued behind
 *   ng_defer_reductions like every other second stage: the caller keeps `partial` alive until the flush). */
int ng_head_loss_blocks(ng_ctx*, int G, int Fh, int C, int64_t max_graph_atoms);
int ng_head_loss_bwd(ng_ctx*, void* stream, int64_t N, int G, int Fh, int C, int64_t max_graph_atoms, const float* g,
                     uint64_t seed, uint64_t offset, float keep, float* mask_out, const float* Wout, const float* bout,
                     const float* atoms, const float* peak_std, const float* peak_avg, const int32_t* graph_ptr,
                     const float* y, const float* w, float grad_weight, float* peaks, float* dg, float* partial);
int ng_head_loss_reduce(ng_ctx*, void* stream, const float* partial, int blocks, int Fh, int C, float* dWout, float* dbout,
                        float* loss_out);

/* ---- edge path through a table of the edge function (round 5; the Engine's default since round 6: csrc/edge_table.hip) ----
 * mask + RBFExpansion + EdgeFCBlock (nmrgnn/model.py:251-261) maps ONE scalar per edge to e[E]: e_ij = m_ij f_W(d_ij).  The
 * caller evaluates f_W with ng_edge_mlp_fwd(_live) on T equidistant points that cover the step's distances and interpolates
 * every edge (four-point cubic Lagrange, error ~ (h / 0.028)^4 < 1e-9 relative at T = 4096); the backward scatters de onto the
 * table (the exact adjoint, 64-bit fixed-point sums: order-free) and runs ng_edge_mlp_bwd on the table's rows.
 * GUARD (ABI 8).  The same launch also evaluates f_W at the T midpoints (rows T .. 2T-1); ng_edge_table_check compares the
 * interpolant with it there and decides ON THE DEVICE: gate[0] != 0 means "answer this call per edge".  The caller launches
 * ng_edge_mlp_fwd_live / _bwd_live with n_live = &gate[1] (the live row count when the guard is up, -1 = "skip" otherwise) and the
 * table's backward with n_live = &gate[2] (0 when the guard is up), so exactly one of the two paths does work; interp skips
 * itself on gate[0].  No host synchronisation.
 *   d_src  [n]     raw distances in SLOT order (mask = d_src > 0)
 *   d_eff, pos     the distances fed to the RBF: slot order (pos == NULL) or compacted, slot i at d_eff[pos[i]]
 *                  (ng_build_live_edges / ng_add_noise_live)
 *   ng_edge_table_range    range[0..1] = min / max of d_eff over the live slots, widened by pad * (max - min) on both sides
 *                          (pad > 0: a table kept over calls), when d_eff != NULL; range[2] = max |de| over the live slots
 *                          when de != NULL.  range: 3 device floats.
 *   ng_edge_table_points   d_tab[t] = lo + (t - 1) h, h = (hi - lo) / (T - 3), t < T;  midpoints != 0: d_tab[T + t] =
 *                          lo + (t - 1/2) h too;  ones[.] = 1 (the table's d_src: all live);  perm[.] = identity (or NULL)
 *   ng_edge_table_check    e_all [2T,E] (or NULL: only the range is checked): err = max |interpolant - f_W| over the interior
 *                          midpoints, scale = max |f_W| over the table; bad = err > tol * scale, or a value not finite, or
 *                          cover != NULL and [cover[0], cover[1]] not inside [range[0], range[1]], or prev != NULL and
 *                          prev[0] != 0.  gate (8 device int32): {bad, bad ? *n_live : -1, bad ? 0 : rows, 0, err, scale (float
 *                          bits), -, -}.
 *   ng_edge_table_interp   e_out[i][c] = m_i sum_k w_k(d_i) e_tab[i0 + k][c]   (E <= 8, T * E <= 16384: T = 2048 for E > 4); skipped on gate[0]
 *   ng_edge_table_scatter  de_tab[t][c] = sum_i m_i w_k(d_i) de[i][c] over the stencils that contain t, rows T .. rows_out-1
 *                          of de_tab zeroed (the midpoint rows of the table's backward); writes range[2] */
int ng_edge_table_range(ng_ctx*, void* stream, int64_t n, int E, const float* d_src, const float* d_eff, const int32_t* pos,
                        const float* de, float pad, float* range);
int ng_edge_table_points(ng_ctx*, void* stream, int T, int midpoints, const float* range, float* d_tab, float* ones, int32_t* perm);
int ng_edge_table_check(ng_ctx*, void* stream, int T, int E, const float* e_all, float tol, const float* range, const float* cover,
                        const int32_t* n_live, int rows, const int32_t* prev, int32_t* gate);
int ng_edge_table_interp(ng_ctx*, void* stream, int64_t n, int E, int T, const float* d_src, const float* d_eff,
                         const int32_t* pos, const float* range, const float* e_tab, const int32_t* gate, float* e_out);
int ng_edge_table_scatter(ng_ctx*, void* stream, int64_t n, int E, int T, int rows_out, const float* d_src, const float* d_eff,
                          const int32_t* pos, float* range, const float* de, float* de_tab);

/* ---- training: NameLoss (s = 1), nmrgnn/losses.py:30-39, batched over graphs -----------------
 *   loss = mean_g  sum_{i in g} w_i (y_i - pred_i)^2 / sum_{i in g} w_i   (divide_no_nan)
 *   dpred written for backward; loss_out is one device float. */
int ng_loss_l2(ng_ctx*, void* stream, int64_t N, int G, const int32_t* graph_ptr, const float* y,
               const float* w, const float* pred, float* loss_out, float* dpred);

/* NameLoss with balance s in [0,1], nmrgnn/losses.py:4-15,30-39, batched over graphs:
 *   loss = mean_g [ s*l2_g + (1-s)*(1 - r_g) ],  r = cov/(m*sqrt(clip(var_x*var_y,0,1e32))) with the
 *   weighted moments of corr_coeff (divide_no_nan); s = 1 equals ng_loss_l2.  dpred as above. */
int ng_loss_name(ng_ctx*, void* stream, int64_t N, int G, const int32_t* graph_ptr, const float* y,
                 const float* w, const float* pred, float s, float* loss_out, float* dpred);

/* Keras Adam (nmrgnn/model.py:44-45): one fused pass over the flat parameter buffer.
 *   g is first multiplied by grad_scale (1/world_size after a summing all-reduce). */
int ng_adam_step(ng_ctx*, void* stream, int64_t n, float* p, const float* g, float* m, float* v,
                 float lr, float beta1, float beta2, float eps, int64_t step, float grad_scale);

/* Gradient exchange for a caller without torch.distributed (SURVEY §8(b) suggested export, §8(e): graph-parallel data
 * parallelism needs ONE all-reduce(sum, fp32) over the flat gradient bucket per step; the reference, nmrgnn/main.py:74-80,
 * is single-device).  RCCL is bound at first use (dlopen of librccl.so): NG_ERR_UNSUPPORTED where it is absent.
 *   ng_comm_unique_id   rank 0 only: 128 opaque bytes the caller hands to every rank (MPI, a file, a socket)
 *   ng_comm_init        every rank, collectively: binds the context's GPU to a communicator of `world` ranks; world == 1
 *                       needs no id and no RCCL.  One communicator per context.
 *   ng_allreduce_grads  in-place sum of flat_grad[n] over the ranks on `stream`, asynchronous; identity in a world of one.
 *                       Divide by the world size in ng_adam_step (grad_scale).
 *   ng_comm_world       the world size of the context (1 without a communicator) */
#define NG_COMM_ID_BYTES 128
int ng_comm_unique_id(void* id_out);
int ng_comm_init(ng_ctx*, int rank, int world, const void* id);
int ng_comm_destroy(ng_ctx*);
int ng_comm_world(ng_ctx*);
int ng_allreduce_grads(ng_ctx*, void* stream, float* flat_grad, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* NMRGNN_HIP_H */
