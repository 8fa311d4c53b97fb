// One kernel for every weight image of the step (round 4; pack_bodies.cuh has the bodies and the why).
//   pack_launch : ONE image now — first use of a cached image, or the cache is off (the job travels as a kernel argument)
//   repack_all  : every job registered in the context's image cache whose source weights lie inside the block Adam has
//                 just updated, in ONE launch behind the Adam kernel; the images then carry the new weight version and no
//                 consumer of the next step packs anything.  The job table lives in device memory and is uploaded only
//                 when the registry changes (a table in the kernel arguments indexed at run time is copied to scratch
//                 per thread: capi.hip, reduce_batch_kernel).
#include <algorithm>
#include <cstring>
#include <vector>

#include "pack_bodies.cuh"

namespace ng {

__global__ __launch_bounds__(PKB) void pack_one_kernel(PackJob j, unsigned ver) {
  pack_job_block(j, (int)blockIdx.x, (int)threadIdx.x, ver);
}

// block0[k] = first block of job k (ascending), block0[njobs] = grid
__global__ __launch_bounds__(PKB) void repack_all_kernel(const PackJob* __restrict__ jobs, const unsigned* __restrict__ block0, int njobs,
                                                         unsigned ver) {
  int k = 0;
  for (int t = 1; t < njobs; ++t) k = blockIdx.x >= block0[t] ? t : k;      // uniform: scalar loads
  const PackJob j = jobs[k];
  pack_job_block(j, (int)(blockIdx.x - block0[k]), (int)threadIdx.x, ver);
}

unsigned pack_flag_version(const ng_ctx* ctx) { return (unsigned)ctx->wver | 0x80000000u; }

int pack_launch(ng_ctx* ctx, hipStream_t st, const PackJob& job) {
  if (job.kind == PK_NONE || job.blocks <= 0) return fail(ctx, NG_ERR_INVALID, "pack_launch: empty job");
  hipLaunchKernelGGL(pack_one_kernel, dim3(job.blocks), dim3(PKB), 0, st, job, pack_flag_version(ctx));
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

void cache_set_job(ng_ctx* ctx, const void* src, int kind, const PackJob& job) {
  auto it = ctx->wimg.find(std::make_pair(src, kind));
  if (it == ctx->wimg.end()) return;
  it->second.job = job;
  it->second.job.guard = RangeGuard{nullptr, 0};      // a per-call guard means nothing to a later refresh
  it->second.has_job = true;
  ctx->wjobs_dirty = true;
}

int repack_all(ng_ctx* ctx, hipStream_t st, const void* lo, const void* hi) {
  if (!ctx->wcache || ctx->wimg.empty()) return NG_OK;
  auto inside = [&](const void* p) { return p == nullptr || ((const char*)p >= (const char*)lo && (const char*)p < (const char*)hi); };
  // the jobs to run: registered, and fed ONLY by weights of the updated block (an image with a source outside it may be
  // waiting for another ng_adam_step call: it stays invalid and is rebuilt at its next use, as before)
  std::vector<ng_ctx::WImage*> sel;
  for (auto& kv : ctx->wimg) {
    ng_ctx::WImage& w = kv.second;
    if (!w.has_job || !w.buf) continue;
    bool ok = true, any = false;
    for (const float* s : w.job.src) { ok = ok && inside(s); any = any || s != nullptr; }
    if (ok && any) sel.push_back(&w);
  }
  if (sel.empty()) return NG_OK;
  DeviceGuard dg(ctx->device);
  const int n = (int)sel.size();
  const size_t jobs_bytes = ((size_t)n * sizeof(PackJob) + 255) / 256 * 256;
  const size_t need = jobs_bytes + (size_t)(n + 1) * sizeof(unsigned);
  uint64_t hash = 1469598103934665603ull;      // which images, in which order, with which block counts
  for (ng_ctx::WImage* w : sel) {
    hash = (hash ^ (uint64_t)(uintptr_t)w->buf) * 1099511628211ull;
    hash = (hash ^ (uint64_t)w->job.blocks) * 1099511628211ull;
  }
  auto fill_host = [&](std::vector<char>& host, unsigned* blocks_out) {
    host.assign(need, 0);
    unsigned* b0 = reinterpret_cast<unsigned*>(host.data() + jobs_bytes);
    unsigned blocks = 0;
    for (int k = 0; k < n; ++k) {
      memcpy(host.data() + (size_t)k * sizeof(PackJob), &sel[k]->job, sizeof(PackJob));
      b0[k] = blocks;
      blocks += (unsigned)sel[k]->job.blocks;
    }
    b0[n] = blocks;
    *blocks_out = blocks;
  };
  const void* table = nullptr;
  int table_blocks = 0;
  if (ctx->replay_armed) {
    // a launch that may be captured (ng_replay_arm): its own immutable table.  The shared one below is rewritten whenever
    // another engine of the device updates its weights — a replayed step then rebuilt the OTHER engine's images (round 5:
    // two trainers on one device diverged at the second replayed step).  Uploaded by the warm-up steps, before any capture.
    hash = (hash ^ (uint64_t)n) * 1099511628211ull;
    for (ng_ctx::WImage* w : sel) {      // the whole job (sources, destinations, parameters) defines the table
      const unsigned char* jb = reinterpret_cast<const unsigned char*>(&w->job);
      for (size_t i = 0; i < sizeof(PackJob); ++i) hash = (hash ^ jb[i]) * 1099511628211ull;
    }
    unsigned blocks = 0;
    std::vector<char> host;
    fill_host(host, &blocks);
    auto it = ctx->wjobs_private.find(hash);
    if (it == ctx->wjobs_private.end()) {
      void* p = nullptr;
      if (hipMalloc(&p, need) != hipSuccess) return fail(ctx, NG_ERR_NOMEM, "repack_all: private job table");
      NG_HIP(ctx, hipMemcpy(p, host.data(), need, hipMemcpyHostToDevice));      // synchronous: `host` is a local
      it = ctx->wjobs_private.emplace(hash, p).first;
    }
    table = it->second;
    table_blocks = (int)blocks;
    ctx->wjobs_private_sel[hash] = sel;
    ctx->replay_token = hash;
  } else {
  if (ctx->wjobs_dirty || ctx->wjobs_n != n || ctx->wjobs_hash != hash || ctx->wjobs_cap < need) {
    if (ctx->wjobs_cap < need) {
      if (ctx->wjobs_dev) { (void)hipStreamSynchronize(st); (void)hipFree(ctx->wjobs_dev); ctx->wjobs_dev = nullptr; ctx->wjobs_cap = 0; }
      if (hipMalloc(&ctx->wjobs_dev, need * 2) != hipSuccess) return fail(ctx, NG_ERR_NOMEM, "repack_all: job table");
      ctx->wjobs_cap = need * 2;
    }
    unsigned blocks = 0;
    fill_host(ctx->wjobs_host, &blocks);
    NG_HIP(ctx, hipMemcpyAsync(ctx->wjobs_dev, ctx->wjobs_host.data(), need, hipMemcpyHostToDevice, st));
    ctx->wjobs_n = n;
    ctx->wjobs_blocks = (int)blocks;
    ctx->wjobs_dirty = false;
    ctx->wjobs_hash = hash;
  }
  table = ctx->wjobs_dev;
  table_blocks = ctx->wjobs_blocks;
  }
  {
    ProfScope ps(ctx, st, "repack_all");
    const size_t jb = ((size_t)n * sizeof(PackJob) + 255) / 256 * 256;
    hipLaunchKernelGGL(repack_all_kernel, dim3(table_blocks), dim3(PKB), 0, st, (const PackJob*)table,
                       (const unsigned*)((const char*)table + jb), n, pack_flag_version(ctx));
    NG_HIP(ctx, hipGetLastError());
  }
  for (ng_ctx::WImage* w : sel) w->ver = ctx->wver;
  return NG_OK;
}

}  // namespace ng
