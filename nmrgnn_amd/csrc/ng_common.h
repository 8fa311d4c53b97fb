// Shared host-side definitions for libnmrgnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <map>
#include <utility>
#include <vector>

#include "../../include/nmrgnn_hip.h"

namespace ng {
// operand-range guard of the fp16-piece kernels (ng_internal.h has the protocol and the device helpers)
struct RangeGuard {
  unsigned* word;     // device word of the context (small scratch)
  unsigned epoch;     // this call's number, never 0
};

// One weight image to (re)build — see pack_bodies.cuh for the kinds and their arguments
enum PackKind : int {
  PK_NONE = 0,
  PK_EDGE_H2 = 1,    // edge forward fp16-piece image: src W0..W3, i0 = E, dst[0] = img [7][32 KB]
  PK_EDGE_WT = 2,    // edge backward W^T piece fragments: src[0] = W2, src[1] = W3, dst[0] = img [64 KB + 16 B]; src[2] = Wo, i0 = E: + the
                     // row-sum bounds {nW2, nW3, nWo} behind the fragments (blocks = 17)
  PK_EDGE_F32 = 3,   // f32 fragments of the three hidden edge layers: src W0..W2, dst[0] = Wpk, dst[1] = WpkT (either may be null)
  PK_MPW_FWD = 4,    // window forward: src[0] = w, i0 = E, dst[0] = piece image, dst[1] = f32 image (blocks 48) or none (24)
  PK_MPW_BWD = 5,    // window backward: dst[0] = T pieces, dst[1] = N pieces, dst[2] / dst[3] = f32 T / N (blocks 96) or none (48)
  PK_MPW_F32 = 6,    // one f32 fragment image: i1 = mode, dst[0]
  PK_FC = 7,         // FC block: src[0..L-1] = W_l, i0 = L, dst[0] = Wf, dst[1] = Wb
  PK_GG = 8,         // gather-GEMM image of an MPLayer weight (gemm_h2.hip): src[0] = w, i0 = E, i1 = F | mode << 16, dst[0] = img
  PK_GX = 9,         // piece image of a GEMM weight operand (gemm_h2.hip: gx_launch): src[0] = W (or an MPLayer weight w), i0 = K,
                     // i1 = N | mode << 20 | nbw << 24, dst[0] = img; mode: pack_bodies.cuh gx_img
  PK_MP_PLAIN = 10,  // MPLayer weight w[l][m][n] as the plain GEMM operand Wp[n F + l][m]: src[0] = w, i0 = F, i1 = E, dst[0] = Wp
};
struct PackJob {
  int kind = PK_NONE;
  int blocks = 0;     // of 256 threads
  int i0 = 0, i1 = 0;
  const float* src[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  void* dst[4] = {nullptr, nullptr, nullptr, nullptr};
  // weights out of the fp16 piece range (|2^8 w| >= 65504): the packer stores the image VERSION into *flag (a consumer
  // compares with the version it was handed: nothing ever has to be cleared) and raises the per-call guard if there is one
  unsigned* flag = nullptr;
  RangeGuard guard = {nullptr, 0};
};
}  // namespace ng

struct ng_prof_rec {
  const char* name;
  hipEvent_t start, stop;
};

struct ng_ctx {
  int device = 0;
  std::string err;
  // growable scratch (split-K partials, repacked weights, aggregated tiles)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // second, independent scratch for leaf kernels whose callers already hold pointers into `ws`
  void* aux = nullptr;
  size_t aux_bytes = 0;
  // NG_SMALL_BYTES that never move (device scalars handed from one launch to later ones: gradient scales of the fp16 GEMMs)
  void* small = nullptr;
  // optional per-kernel hipEvent bracketing
  bool prof = false;
  std::vector<ng_prof_rec> recs;
  std::vector<hipEvent_t> pool;
  // cached device properties
  int num_cu = 256;
  // largest member graph of the current batch (ng_ctx_set_graph_span), 0 = unknown
  int64_t graph_span = 0;
  // packed weight images kept across calls while the caller declares the weights frozen (ng_weights_frozen):
  // inference repacks nothing.  key = (source pointer, image kind); an entry is valid while its version equals wver,
  // which ng_weights_changed / ng_adam_step / (un)freezing bump.
  // `job` (has_job): how to rebuild the image; ng_adam_step re-runs every registered job of the weights it has just
  // updated in ONE launch (repack.hip) instead of leaving the images to be rebuilt one launch each at their next use
  struct WImage { void* buf = nullptr; size_t bytes = 0; uint64_t ver = 0; bool has_job = false; ng::PackJob job; };
  bool wcache = false;
  int wowner = 0;
  uint64_t wver = 1;
  std::map<std::pair<const void*, int>, WImage> wimg;
  // device copy of the registered jobs (rebuilt when the registry changes)
  void* wjobs_dev = nullptr;
  size_t wjobs_cap = 0;
  bool wjobs_dirty = true;
  int wjobs_n = 0, wjobs_blocks = 0;
  uint64_t wjobs_hash = 0;      // of the job set the device copy holds
  std::vector<char> wjobs_host;  // staging of the upload (stays alive while an asynchronous copy may read it)
  // operand-range guard of the fp16-piece kernels (ng_internal.h: RangeGuard): call counter behind the epochs
  uint32_t range_epoch = 0;
  // gradient exchange for a C-ABI caller (comm.hip): RCCL communicator of this rank, nullptr = none / world of one
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  // deferred second-stage reductions (ng_defer_reductions; reduce.cuh): queue + partial arena, owned by capi.hip
  bool defer_reduce = false;
  void* rq = nullptr;
  // graph replay of a training step (ng_replay_arm / ng_replay_stage): while armed, the launches whose arguments change from
  // step to step (noise / dropout seed, Adam's bias-corrected rate) read them from the staged device block instead
  bool replay_armed = false;
  // job tables of repack launches recorded while armed: one per job set, never rewritten or freed before ng_ctx_destroy — a
  // captured launch reads its table at every replay, whatever other engines of the device have registered since
  std::map<uint64_t, void*> wjobs_private;
  // the images each private table rebuilds (map nodes of wimg: stable addresses) and the table of the last armed repack launch:
  // ng_replay_token / ng_replay_commit re-stamp exactly these after a replayed step (round-5 advisor finding: a replay changes
  // the weights but runs no host bookkeeping, so every OTHER cached image kept reading as valid with weights of N steps ago)
  std::map<uint64_t, std::vector<WImage*>> wjobs_private_sel;
  uint64_t replay_token = 0;
};

namespace ng {

// Process-wide path switches, parsed from the environment ONCE (first use) instead of a getenv + string compare
// on every launch; ng_reload_env() re-reads them (tests and A/B tools flip variables inside one process).
//   NG_EDGE_MATH=fp32      edge MLP forward+backward on f32-input MFMA (default: two-piece fp16 split operands)
//   NG_EDGE_BWD_MATH=fp32  only the edge backward on f32-input MFMA
//   NG_GEMM_MATH=fp32      generic GEMMs on f32-input MFMA (default: split operands where the shape allows)
//   NG_EDGE_PATH=layered   one launch per edge-MLP layer (any H / Le)
//   NG_MP_PATH=layered     aggregate -> A[N,E*F] -> GEMM for every width (default: window kernels at F == 64)
//   NG_FC_PATH=layered     one launch per FC layer
//   NG_DENSE_PATH=generic  no register-resident tall-skinny kernels
//   NG_HEAD_PATH=generic   the first head / embedding-gradient kernels
//   NG_KNN=serial / lanes  one lane / 8-16 lanes per query atom in the kNN graph kernel (default: one wave per query)
//   NG_KNN=cells / brute   the cell-grid neighbour search for every frame size / for none (default: frames >= 16384 atoms)
//   NG_MP_GG=1             default-width MPLayer as the window gather-GEMM (mp_gw.cuh) for every eligible call, training included (default: inference of molecule batches only)
//   NG_MP_GG_MIN_ROWS=n    smallest call (rows) that takes the gather-GEMM (default 8192)
//   NG_MP_GW=nowin         the window gather-GEMM reads every tile's sources from memory (test: same bits as the window)
//   NG_MP_W16=0            the eight-wave window kernels at F == 64 (default: sixteen waves, mp_win16*.hip)
//   NG_MP_WAVE=1 / 0       the wave-autonomous forward window kernel (mp_wave.hip) for every supported call / none (default: batches that fill the chip)
//   NG_REDUCE=narrow       second-stage reductions 64 elements per block at every size (test of the wide form's bits)
struct Switches {
  bool edge_math_fp32 = false, edge_bwd_math_fp32 = false, gemm_math_fp32 = false;
  bool edge_layered = false, mp_layered = false, fc_layered = false;
  bool dense_generic = false, head_generic = false, knn_serial = false, knn_cells = false, knn_brute = false;
  bool mp_gg_on = false;             // NG_MP_GG=1
  bool mp_gw_nowin = false;          // NG_MP_GW=nowin: the window form reads every tile's sources from memory (tests: same bits as the window)
  int mp_wave = -1;                  // NG_MP_WAVE=1 / 0: the wave-autonomous forward window kernel (mp_wave.hip) for every supported call / for none (default: batches that fill the chip)
  bool mp_w16 = true;                // NG_MP_W16=0: the eight-wave forward window kernel instead of the 16-wave one (mp_win16.hip)
  bool reduce_narrow = false;        // NG_REDUCE=narrow: second-stage reductions 64 elements per block at every size (reduce.cuh)
  bool knn_lanes = false;            // NG_KNN=lanes
  int64_t mp_gg_min_rows = 8192;     // NG_MP_GG_MIN_ROWS
};
const Switches& sw();

// hipSetDevice(dev) for the lifetime of the object, previous device restored afterwards
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

inline int fail(ng_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define NG_HIP(ctx, expr)                                                              \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return ng::fail(ctx, NG_ERR_HIP,                                                 \
                      std::string(#expr) + ": " + hipGetErrorString(_e));              \
  } while (0)

#define NG_REQUIRE(ctx, cond, msg)                                                     \
  do {                                                                                 \
    if (!(cond)) return ng::fail(ctx, NG_ERR_INVALID, std::string(msg) + " [" #cond "]"); \
  } while (0)

// scratch: returns nullptr on failure (error string set)
void* workspace(ng_ctx* ctx, size_t bytes);
void* aux_workspace(ng_ctx* ctx, size_t bytes);
constexpr size_t NG_SMALL_BYTES = 64 * 1024;
void* small_scratch(ng_ctx* ctx);      // NG_SMALL_BYTES, allocated once per context, address stable until ng_ctx_destroy
// staged per-step state of a replayed training step: {u64 seed; f32 lr_t; ...} at a fixed place of the small scratch;
// nullptr unless the context is armed (ng_replay_arm)
constexpr size_t NG_REPLAY_STATE_OFFSET = NG_SMALL_BYTES - 256;
const uint64_t* replay_state(ng_ctx* ctx);
// Frozen-weight image cache.  Returns nullptr when the cache is off (pack into scratch as before); otherwise a
// persistent buffer of `bytes` for (src, kind) with *valid = true when it already holds the image of the current
// weights (skip the pack launch).  kinds: 1 MPLayer Wp, 3 GEMM fp16-piece image, 4 window fragments, 5 FC fragments, 6 edge fp16-piece image
void* cached_image(ng_ctx* ctx, const void* src, int kind, size_t bytes, bool* valid);
// after packing a cached image: remember how (pack_bodies.cuh), so that ng_adam_step can refresh it
void cache_set_job(ng_ctx* ctx, const void* src, int kind, const PackJob& job);
// one image now (first use / cache off); the version a flag word is compared with
int pack_launch(ng_ctx* ctx, hipStream_t st, const PackJob& job);
unsigned pack_flag_version(const ng_ctx* ctx);
// every registered image whose sources all lie in [lo, hi): one launch on `st`; the images then carry the current version
int repack_all(ng_ctx* ctx, hipStream_t st, const void* lo, const void* hi);

// RAII-less profiling bracket: call begin before the launch(es), end after.
struct ProfScope {
  ng_ctx* ctx;
  hipStream_t stream;
  hipEvent_t stop = nullptr;
  ProfScope(ng_ctx* c, hipStream_t s, const char* name);
  ~ProfScope();
};

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace ng
