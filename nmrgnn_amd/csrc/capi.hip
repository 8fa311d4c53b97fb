// Context, error reporting, scratch workspace and per-kernel hipEvent profiling.
#include <cmath>
#include <algorithm>
#include <cstring>
#include <map>
#include <new>

#include "ng_common.h"
#include "ng_internal.h"
#include "reduce.cuh"

namespace ng {

static Switches g_sw;
static bool g_sw_loaded = false;

static bool env_is(const char* name, const char* value) {
  const char* v = getenv(name);
  return v && !strcmp(v, value);
}

static void load_switches() {
  Switches s;
  s.edge_math_fp32 = env_is("NG_EDGE_MATH", "fp32");
  s.edge_bwd_math_fp32 = env_is("NG_EDGE_BWD_MATH", "fp32");
  s.gemm_math_fp32 = env_is("NG_GEMM_MATH", "fp32");
  s.edge_layered = env_is("NG_EDGE_PATH", "layered");
  s.mp_layered = env_is("NG_MP_PATH", "layered");
  s.fc_layered = env_is("NG_FC_PATH", "layered");
  s.dense_generic = env_is("NG_DENSE_PATH", "generic");
  s.head_generic = env_is("NG_HEAD_PATH", "generic");
  s.knn_serial = env_is("NG_KNN", "serial");
  s.knn_lanes = env_is("NG_KNN", "lanes");
  s.knn_cells = env_is("NG_KNN", "cells");
  s.knn_brute = env_is("NG_KNN", "brute");
  s.mp_gg_on = env_is("NG_MP_GG", "1");
  s.mp_gw_nowin = env_is("NG_MP_GW", "nowin");
  s.mp_w16 = !env_is("NG_MP_W16", "0");
  s.mp_wave = env_is("NG_MP_WAVE", "1") ? 1 : (env_is("NG_MP_WAVE", "0") ? 0 : -1);
  s.reduce_narrow = env_is("NG_REDUCE", "narrow");
  if (const char* v = getenv("NG_MP_GG_MIN_ROWS")) { const long long r = atoll(v); if (r >= 1) s.mp_gg_min_rows = r; }
  g_sw = s;
  g_sw_loaded = true;
}

const Switches& sw() {
  if (!g_sw_loaded) load_switches();
  return g_sw;
}

// the guard word lives at the END of the small scratch (the gradient-scale slots grow from its start: gemm_h2.hip)
RangeGuard range_guard_begin(ng_ctx* ctx) {
  RangeGuard g;
  char* s = (char*)small_scratch(ctx);
  g.word = s ? reinterpret_cast<unsigned*>(s + NG_SMALL_BYTES - 64) : nullptr;
  if (++ctx->range_epoch == 0) ++ctx->range_epoch;
  g.epoch = ctx->range_epoch;
  return g;
}

const uint64_t* replay_state(ng_ctx* ctx) {
  if (!ctx->replay_armed) return nullptr;
  char* s = (char*)small_scratch(ctx);
  return s ? reinterpret_cast<const uint64_t*>(s + NG_REPLAY_STATE_OFFSET) : nullptr;
}

void* workspace(ng_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return ctx->ws;
  DeviceGuard dg(ctx->device);     // the scratch belongs to the context's GPU, whatever device is current
  // grow (1.5x headroom); synchronises the device because older launches may still use the block
  size_t want = bytes + bytes / 2;
  if (ctx->ws) {
    (void)hipDeviceSynchronize();
    (void)hipFree(ctx->ws);
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    ctx->err = std::string("workspace hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e);
    return nullptr;
  }
  ctx->ws = p;
  ctx->ws_bytes = want;
  return p;
}

void* aux_workspace(ng_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->aux_bytes) return ctx->aux;
  DeviceGuard dg(ctx->device);
  const size_t want = bytes + bytes / 2;
  if (ctx->aux) {
    (void)hipDeviceSynchronize();
    (void)hipFree(ctx->aux);
    ctx->aux = nullptr;
    ctx->aux_bytes = 0;
  }
  void* p = nullptr;
  const hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    ctx->err = std::string("aux workspace hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e);
    return nullptr;
  }
  ctx->aux = p;
  ctx->aux_bytes = want;
  return p;
}

void* small_scratch(ng_ctx* ctx) {
  if (ctx->small) return ctx->small;
  DeviceGuard dg(ctx->device);
  void* p = nullptr;
  const hipError_t e = hipMalloc(&p, NG_SMALL_BYTES);
  if (e != hipSuccess) {
    ctx->err = std::string("small scratch hipMalloc: ") + hipGetErrorString(e);
    return nullptr;
  }
  (void)hipMemset(p, 0, NG_SMALL_BYTES);      // the range-guard word must not start out equal to an epoch
  ctx->small = p;
  return p;
}

void* cached_image(ng_ctx* ctx, const void* src, int kind, size_t bytes, bool* valid) {
  *valid = false;
  if (!ctx->wcache) return nullptr;
  // A source inside the context's own scratch is a temporary (a weight matrix repacked for THIS call: the same address
  // holds another layer's weights on the next call) — never a cache key.  Round-2 advisor finding: with the cache keyed
  // by such an address a frozen window around a multi-layer backward multiplied with the previous layer's image.
  auto inside = [&](const void* base, size_t n) {
    return base && (const char*)src >= (const char*)base && (const char*)src < (const char*)base + n;
  };
  if (inside(ctx->ws, ctx->ws_bytes) || inside(ctx->aux, ctx->aux_bytes)) return nullptr;
  ng_ctx::WImage& w = ctx->wimg[std::make_pair(src, kind)];
  if (w.bytes < bytes) {
    DeviceGuard dg(ctx->device);
    if (w.buf) { (void)hipDeviceSynchronize(); (void)hipFree(w.buf); w.buf = nullptr; w.bytes = 0; }
    if (hipMalloc(&w.buf, bytes) != hipSuccess) { w.buf = nullptr; return nullptr; }
    (void)hipMemset(w.buf, 0, bytes);      // (flag words inside an image start out equal to no version)
    w.bytes = bytes;
    w.ver = 0;
    w.has_job = false;                     // a registered job points into the old buffer
    ctx->wjobs_dirty = true;
  }
  *valid = w.ver == ctx->wver;
  w.ver = ctx->wver;
  return w.buf;
}

// ---- deferred second-stage reductions (reduce.cuh) ------------------------------------------------------------------
constexpr int RB_MAX_JOBS = 24;
struct ReduceBatch {
  int njobs;
  unsigned block0[RB_MAX_JOBS + 1];      // first block of job j; block0[njobs] = grid
  ReduceJob job[RB_MAX_JOBS];
};

// The job of a block is picked with compile-time indices (scalar selects over the kernel arguments): a run-time index
// into the argument struct made the compiler copy the struct to scratch per thread — 0.45 ms for seven jobs.
static __global__ __launch_bounds__(1024) void reduce_batch_kernel(ReduceBatch b) {
  __shared__ __attribute__((aligned(16))) float red[16 * 4][64];      // [16][64] floats, or float4 (wide jobs: reduce.cuh)
  ReduceJob job = b.job[0];
  unsigned b0 = 0;
#pragma unroll
  for (int k = 1; k < RB_MAX_JOBS; ++k)
    if (k < b.njobs && blockIdx.x >= b.block0[k]) { job = b.job[k]; b0 = b.block0[k]; }
  reduce_job_block(job, nullptr, blockIdx.x - b0, red);
}

struct ReduceQueue {
  ReduceBatch batch{};
  // partial arena: chunks live until the flush; more than one chunk = the arena was too small this round, the flush
  // replaces them by one chunk of the total size
  struct Chunk { char* p; size_t bytes, used; };
  std::vector<Chunk> chunks;
};

static ReduceQueue* rqueue(ng_ctx* ctx) {
  if (!ctx->rq) ctx->rq = new (std::nothrow) ReduceQueue();
  return (ReduceQueue*)ctx->rq;
}

float* deferred_partials(ng_ctx* ctx, size_t floats) {
  if (!ctx->defer_reduce) return nullptr;
  ReduceQueue* q = rqueue(ctx);
  if (!q) return nullptr;
  const size_t bytes = (floats * 4 + 255) / 256 * 256;
  if (!q->chunks.empty()) {
    ReduceQueue::Chunk& c = q->chunks.back();
    if (c.used + bytes <= c.bytes) { char* r = c.p + c.used; c.used += bytes; return (float*)r; }
  }
  DeviceGuard dg(ctx->device);
  const size_t want = std::max(bytes * 2, (size_t)32 << 20);
  void* p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) return nullptr;      // the caller falls back to its own scratch + eager reduction
  q->chunks.push_back({(char*)p, want, bytes});
  return (float*)p;
}

static bool in_arena(ng_ctx* ctx, const float* p) {
  ReduceQueue* q = (ReduceQueue*)ctx->rq;
  if (!q) return false;
  for (auto& c : q->chunks)
    if ((const char*)p >= c.p && (const char*)p < c.p + c.bytes) return true;
  return false;
}

static int run_queue(ng_ctx* ctx, hipStream_t st, ReduceQueue* q);

int flush_reductions(ng_ctx* ctx, hipStream_t st) {
  ReduceQueue* q = (ReduceQueue*)ctx->rq;
  if (!q) return NG_OK;
  {
    const int rc = run_queue(ctx, st, q);
    if (rc) return rc;
  }
  if (q->chunks.size() > 1) {
    DeviceGuard dg(ctx->device);
    size_t total = 0;
    (void)hipDeviceSynchronize();
    for (auto& c : q->chunks) { total += c.bytes; (void)hipFree(c.p); }
    q->chunks.clear();
    void* p = nullptr;
    if (hipMalloc(&p, total) == hipSuccess) q->chunks.push_back({(char*)p, total, 0});
  } else if (!q->chunks.empty()) {
    q->chunks[0].used = 0;
  }
  return NG_OK;
}

static int run_queue(ng_ctx* ctx, hipStream_t st, ReduceQueue* q) {
  if (q->batch.njobs == 0) return NG_OK;
  ProfScope ps(ctx, st, "reduce_partials");
  hipLaunchKernelGGL(reduce_batch_kernel, dim3(q->batch.block0[q->batch.njobs]), dim3(1024), 0, st, q->batch);
  NG_HIP(ctx, hipGetLastError());
  q->batch.njobs = 0;
  return NG_OK;
}

static int queue_job(ng_ctx* ctx, hipStream_t st, const ReduceJob& j) {
  ReduceQueue* q = rqueue(ctx);
  if (q->batch.njobs == RB_MAX_JOBS) {
    // queue full: run what is queued (the arena keeps its contents: only the job list is emptied here)
    const int rc = run_queue(ctx, st, q);
    if (rc) return rc;
  }
  ReduceBatch& b = q->batch;
  const int k = b.njobs++;
  if (k == 0) b.block0[0] = 0;
  b.job[k] = j;
  b.block0[k + 1] = b.block0[k] + reduce_job_blocks(j);
  return NG_OK;
}

int reduce_or_defer(ng_ctx* ctx, hipStream_t st, const float* partial, int nz, int64_t n_elem, float* out, int w_map, int F,
                    int E, int Nout, int64_t z_stride) {
  if (ctx->defer_reduce && in_arena(ctx, partial))
    return queue_job(ctx, st, ReduceJob{partial, out, n_elem, z_stride ? z_stride : n_elem, nz, w_map, F, E, Nout, sw().reduce_narrow ? 1 : 0});
  launch_reduce_z(st, partial, nz, n_elem, out, w_map, F, E, Nout, z_stride);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

bool defer_outer_job(ng_ctx* ctx, hipStream_t st, const float* X, const float* Y, int64_t n, int C, int F, float* out) {
  if (!ctx->defer_reduce || n > OUTER_JOB_MAX_ROWS || !rqueue(ctx)) return false;
  ReduceJob j{Y, out, (int64_t)C * F, (int64_t)F, (int)n, 3, F, C, 1, 1};
  j.aux = X;
  return queue_job(ctx, st, j) == NG_OK;
}

int reduce_seg_or_defer(ng_ctx* ctx, hipStream_t st, const float* partial, int nz, int64_t n_elem, int64_t z_stride,
                        const ReduceSegs& sg, bool caller_owned) {
  if (ctx->defer_reduce && (caller_owned || in_arena(ctx, partial))) {
    // a segment is a plain job of its own on the segment's slice of the partial rows (an element's sum does not
    // depend on which block holds it: the bits are those of reduce_z_seg_kernel)
    (void)n_elem;
    for (int k = 0; k < sg.n; ++k) {
      const int rc = queue_job(ctx, st, ReduceJob{partial + sg.begin[k], sg.dst[k], (int64_t)sg.len[k], z_stride, nz, 0, 0, 0, 1, sw().reduce_narrow ? 1 : 0});
      if (rc) return rc;
    }
    return NG_OK;
  }
  launch_reduce_z_seg(st, partial, nz, n_elem, z_stride, sg);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

ProfScope::ProfScope(ng_ctx* c, hipStream_t s, const char* name) : ctx(c), stream(s) {
  if (!ctx || !ctx->prof) return;
  hipEvent_t a = nullptr, b = nullptr;
  if (ctx->pool.size() >= 2) {
    a = ctx->pool.back(); ctx->pool.pop_back();
    b = ctx->pool.back(); ctx->pool.pop_back();
  } else {
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  }
  (void)hipEventRecord(a, stream);
  ctx->recs.push_back({name, a, b});
  stop = b;
}

ProfScope::~ProfScope() {
  if (stop) (void)hipEventRecord(stop, stream);
}

}  // namespace ng

extern "C" int ng_abi_version(void) { return NG_ABI_VERSION; }

extern "C" int ng_weights_frozen(ng_ctx* ctx, int owner) {
  if (!ctx) return NG_ERR_INVALID;
  if (owner != 0 && owner != ctx->wowner) {   // another model takes the cache over: nothing in it is its own
    ctx->wver++;
    ctx->wowner = owner;
    for (auto& kv : ctx->wimg) kv.second.has_job = false;     // (its weights may be gone: nothing of it is refreshed)
    ctx->wjobs_dirty = true;
  }
  ctx->wcache = owner != 0;
  return NG_OK;
}

extern "C" int ng_ctx_set_graph_span(ng_ctx* ctx, int64_t max_graph_atoms) {
  if (!ctx) return NG_ERR_INVALID;
  ctx->graph_span = max_graph_atoms > 0 ? max_graph_atoms : 0;
  return NG_OK;
}

extern "C" int ng_weights_changed(ng_ctx* ctx) {
  if (!ctx) return NG_ERR_INVALID;
  ctx->wver++;
  return NG_OK;
}

extern "C" int ng_defer_reductions(ng_ctx* ctx, void* stream, int on) {
  if (!ctx) return NG_ERR_INVALID;
  if (!on && ctx->defer_reduce) {
    const int rc = ng::flush_reductions(ctx, (hipStream_t)stream);
    if (rc) return rc;
  }
  ctx->defer_reduce = on != 0;
  return NG_OK;
}

extern "C" int ng_flush_reductions(ng_ctx* ctx, void* stream) {
  if (!ctx) return NG_ERR_INVALID;
  return ng::flush_reductions(ctx, (hipStream_t)stream);
}

// ---- graph replay of small training steps (round 5) --------------------------------------------------------------------
// One 256-atom graph per step is ~33 launches of a few microseconds: the step is bound by the host's launch rate.  The chain
// can be captured once per shape (torch.cuda.CUDAGraph on the stream the caller passes, or hipStreamBeginCapture) and replayed
// — if (a) nothing inside allocates or synchronises (scratch is sized by a warm-up step, ng_ctx_reserve) and (b) no kernel
// argument changes from step to step.  Three do: the seed of GaussianNoise and Dropout and Adam's bias-corrected rate.  While
// a context is armed those launches read them from a device block that ONE eager launch per step (ng_replay_stage) fills —
// together with the step's inputs, copied into the static buffers the captured chain reads, and the range-guard word reset.
struct StageArgs {
  const uint32_t* src[8];
  uint32_t* dst[8];
  uint32_t words[8];
  int n;
  uint64_t seed;
  float lr_t;
  uint64_t* state;
  unsigned* guard_word;
};
__global__ __launch_bounds__(256) void replay_stage_kernel(StageArgs a) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.state[0] = a.seed;
    reinterpret_cast<float*>(a.state)[2] = a.lr_t;
    *a.guard_word = 0;
  }
  for (int c = 0; c < a.n; ++c)
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.words[c]; i += gridDim.x * 256) a.dst[c][i] = a.src[c][i];
}

extern "C" int ng_replay_arm(ng_ctx* ctx, int on) {
  if (!ctx) return NG_ERR_INVALID;
  if (on && !ng::small_scratch(ctx)) return NG_ERR_NOMEM;
  ctx->replay_armed = on != 0;
  return NG_OK;
}

extern "C" int ng_replay_stage(ng_ctx* ctx, void* stream, uint64_t seed, float lr, float beta1, float beta2, int64_t step,
                               int n_copies, const void* const* src, void* const* dst, const uint64_t* bytes) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, n_copies >= 0 && n_copies <= 8, "replay_stage: at most 8 copies");
  NG_REQUIRE(ctx, step >= 1, "replay_stage: step counts from 1");
  char* sm = (char*)ng::small_scratch(ctx);
  if (!sm) return NG_ERR_NOMEM;
  StageArgs a{};
  uint64_t most = 0;
  for (int c = 0; c < n_copies; ++c) {
    NG_REQUIRE(ctx, src[c] && dst[c] && bytes[c] % 4 == 0 && bytes[c] < ((uint64_t)1 << 32), "replay_stage: copies of whole 32-bit words");
    a.src[c] = (const uint32_t*)src[c]; a.dst[c] = (uint32_t*)dst[c]; a.words[c] = (uint32_t)(bytes[c] / 4);
    most = std::max<uint64_t>(most, bytes[c] / 4);
  }
  a.n = n_copies; a.seed = seed;
  // the expression of ng_adam_step: the same bits reach the kernel
  a.lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)step)) / (1.0 - std::pow((double)beta1, (double)step)));
  a.state = reinterpret_cast<uint64_t*>(sm + ng::NG_REPLAY_STATE_OFFSET);
  a.guard_word = reinterpret_cast<unsigned*>(sm + ng::NG_SMALL_BYTES - 64);
  const int grid = (int)std::min<uint64_t>(256, std::max<uint64_t>(1, (most + 255) / 256));
  hipLaunchKernelGGL(replay_stage_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

extern "C" int ng_replay_token(ng_ctx* ctx, uint64_t* token) {
  if (!ctx || !token) return NG_ERR_INVALID;
  *token = ctx->replay_token;
  return NG_OK;
}

// host bookkeeping of ONE replayed ng_adam_step: the weights moved, the captured repack launch rebuilt the images of `token`
extern "C" int ng_replay_commit(ng_ctx* ctx, uint64_t token) {
  if (!ctx) return NG_ERR_INVALID;
  ctx->wver++;
  auto it = ctx->wjobs_private_sel.find(token);
  if (token != 0 && it != ctx->wjobs_private_sel.end())
    for (ng_ctx::WImage* w : it->second)
      if (w->buf && w->has_job) w->ver = ctx->wver;
  return NG_OK;
}

extern "C" int ng_reload_env(void) {
  ng::load_switches();
  return NG_OK;
}

extern "C" int ng_ctx_create(int device, ng_ctx** out) {
  if (!out) return NG_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return NG_ERR_HIP;
  ng::DeviceGuard dg(device);      // the caller's current device is left as it was
  (void)ng::sw();
  ng_ctx* ctx = new (std::nothrow) ng_ctx();
  if (!ctx) return NG_ERR_NOMEM;
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
  *out = ctx;
  return NG_OK;
}

extern "C" int ng_comm_destroy(ng_ctx* ctx);

extern "C" void ng_ctx_destroy(ng_ctx* ctx) {
  if (!ctx) return;
  (void)ng_comm_destroy(ctx);
  ng::DeviceGuard dg(ctx->device);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->aux) (void)hipFree(ctx->aux);
  if (ctx->small) (void)hipFree(ctx->small);
  for (auto& kv : ctx->wimg)
    if (kv.second.buf) (void)hipFree(kv.second.buf);
  if (ctx->wjobs_dev) (void)hipFree(ctx->wjobs_dev);
  for (auto& kv : ctx->wjobs_private) (void)hipFree(kv.second);
  for (auto& r : ctx->recs) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  for (auto e : ctx->pool) (void)hipEventDestroy(e);
  if (ctx->rq) {
    ng::ReduceQueue* q = (ng::ReduceQueue*)ctx->rq;
    for (auto& c : q->chunks) (void)hipFree(c.p);
    delete q;
  }
  delete ctx;
}

extern "C" const char* ng_last_error(ng_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" int ng_ctx_reserve(ng_ctx* ctx, uint64_t bytes) {
  if (!ctx) return NG_ERR_INVALID;
  return ng::workspace(ctx, (size_t)bytes) ? NG_OK : NG_ERR_NOMEM;
}

extern "C" int ng_prof_enable(ng_ctx* ctx, int on) {
  if (!ctx) return NG_ERR_INVALID;
  ctx->prof = on != 0;
  return NG_OK;
}

extern "C" int ng_prof_reset(ng_ctx* ctx) {
  if (!ctx) return NG_ERR_INVALID;
  for (auto& r : ctx->recs) {
    ctx->pool.push_back(r.start);
    ctx->pool.push_back(r.stop);
  }
  ctx->recs.clear();
  return NG_OK;
}

extern "C" int ng_prof_read(ng_ctx* ctx, int cap, const char** names, double* total_ms,
                            int64_t* count) {
  if (!ctx) return NG_ERR_INVALID;
  std::vector<const char*> order;
  std::map<std::string, int> index;
  std::vector<double> tot;
  std::vector<int64_t> cnt;
  for (auto& r : ctx->recs) {
    if (hipEventSynchronize(r.stop) != hipSuccess) return ng::fail(ctx, NG_ERR_HIP, "prof sync");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) continue;
    auto it = index.find(r.name);
    int k;
    if (it == index.end()) {
      k = (int)order.size();
      index[r.name] = k;
      order.push_back(r.name);
      tot.push_back(0.0);
      cnt.push_back(0);
    } else {
      k = it->second;
    }
    tot[k] += ms;
    cnt[k] += 1;
  }
  int n = (int)order.size();
  if (n > cap) n = cap;
  for (int i = 0; i < n; ++i) {
    names[i] = order[i];
    total_ms[i] = tot[i];
    count[i] = cnt[i];
  }
  return n;
}
