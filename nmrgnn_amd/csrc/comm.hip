// Gradient exchange behind the C ABI (SURVEY §8(b) suggested export `ng_allreduce_grads`, §8(e): training needs ONE
// all-reduce(sum, fp32) over the flat gradient bucket per step; the reference itself is single-device).  The Python host
// side keeps using torch.distributed ("nccl" = RCCL on ROCm: parallel.py); these entry points are for a caller that has no
// torch — they bind RCCL at first use (dlopen) so that the library loads on machines without it.
//   rank 0:  ng_comm_unique_id(id)  -> the caller hands the 128 bytes to every rank (MPI, a file, a socket)
//   all:     ng_comm_init(ctx, rank, world, id);  per step: ng_allreduce_grads(ctx, stream, grad, n);  ng_comm_destroy(ctx)
#include <dlfcn.h>
#include <glob.h>
#include <link.h>

#include <cstdlib>
#include <cstring>
#include <string>

#include "ng_common.h"
#include "ng_internal.h"

namespace {

struct RcclId { char internal[128]; };           // rccl.h: ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
static_assert(sizeof(RcclId) == NG_COMM_ID_BYTES, "unique-id size");
typedef void* RcclComm;
typedef int (*GetUniqueIdFn)(RcclId*);
typedef int (*CommInitRankFn)(RcclComm*, int, RcclId, int);
typedef int (*CommDestroyFn)(RcclComm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, RcclComm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
constexpr int RCCL_FLOAT32 = 7, RCCL_SUM = 0;    // rccl.h: ncclFloat32, ncclSum

struct Rccl {
  void* lib = nullptr;
  GetUniqueIdFn get_id = nullptr;
  CommInitRankFn init_rank = nullptr;
  CommDestroyFn destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  GetErrorStringFn err = nullptr;
  bool tried = false;
};
Rccl g_rccl;

bool rccl_load() {
  if (g_rccl.tried) return g_rccl.lib != nullptr;
  g_rccl.tried = true;
  // Where RCCL may live, in this order: NG_RCCL_PATH (a file, or a directory holding librccl.so[.1]); a copy some other
  // library of the process has already mapped (torch brings its own: using a second copy beside it would be two RCCLs in one
  // process); the loader's search path; ROCm's default prefix; the copy a torch wheel ships in torch/lib — found without
  // importing torch, next to a mapped libc10_hip.so / libtorch_hip.so or under the usual site-packages roots (a torch-free
  // caller on a box whose only RCCL is the wheel's).
  void* lib = nullptr;
  auto try_open = [&](const std::string& path, int extra) {
    if (!lib && !path.empty()) lib = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL | extra);
  };
  if (const char* envp = getenv("NG_RCCL_PATH")) {
    const std::string e(envp);
    try_open(e, 0);
    try_open(e + "/librccl.so.1", 0);
    try_open(e + "/librccl.so", 0);
  }
  for (const char* n : {"librccl.so.1", "librccl.so"}) try_open(n, RTLD_NOLOAD);
  if (!lib) {      // next to torch's HIP libraries, if they are mapped
    struct Probe { std::string dir; } probe;
    dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* data) -> int {
      const std::string name = info->dlpi_name ? info->dlpi_name : "";
      for (const char* key : {"/libc10_hip.so", "/libtorch_hip.so", "/libtorch.so"}) {
        const size_t pos = name.rfind(key);
        if (pos != std::string::npos) { static_cast<Probe*>(data)->dir = name.substr(0, pos); return 1; }
      }
      return 0;
    }, &probe);
    if (!probe.dir.empty()) try_open(probe.dir + "/librccl.so", 0);
  }
  for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) try_open(n, 0);
  if (!lib) {
    glob_t g;
    for (const char* pat : {"/usr/local/lib/python3*/dist-packages/torch/lib/librccl.so", "/usr/lib/python3*/site-packages/torch/lib/librccl.so",
                            "/usr/lib/python3/dist-packages/torch/lib/librccl.so", "/opt/conda/lib/python3*/site-packages/torch/lib/librccl.so"}) {
      if (lib) break;
      if (glob(pat, 0, nullptr, &g) == 0) {
        for (size_t i = 0; i < g.gl_pathc && !lib; ++i) try_open(g.gl_pathv[i], 0);
        globfree(&g);
      }
    }
  }
  if (!lib) return false;
  Rccl r;
  r.get_id = (GetUniqueIdFn)dlsym(lib, "ncclGetUniqueId");
  r.init_rank = (CommInitRankFn)dlsym(lib, "ncclCommInitRank");
  r.destroy = (CommDestroyFn)dlsym(lib, "ncclCommDestroy");
  r.all_reduce = (AllReduceFn)dlsym(lib, "ncclAllReduce");
  r.err = (GetErrorStringFn)dlsym(lib, "ncclGetErrorString");
  if (!r.get_id || !r.init_rank || !r.destroy || !r.all_reduce) { dlclose(lib); return false; }
  r.lib = lib; r.tried = true;
  g_rccl = r;
  return true;
}

int rccl_fail(ng_ctx* ctx, const char* what, int rc) {
  return ng::fail(ctx, NG_ERR_HIP, std::string(what) + ": " + (g_rccl.err ? g_rccl.err(rc) : "RCCL error ") + " (" + std::to_string(rc) + ")");
}

}  // namespace

extern "C" int ng_comm_unique_id(void* id_out) {
  if (!id_out) return NG_ERR_INVALID;
  if (!rccl_load()) return NG_ERR_UNSUPPORTED;
  RcclId id;
  if (g_rccl.get_id(&id) != 0) return NG_ERR_HIP;
  memcpy(id_out, &id, sizeof(id));
  return NG_OK;
}

extern "C" int ng_comm_init(ng_ctx* ctx, int rank, int world, const void* id) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "comm_init: 0 <= rank < world");
  NG_REQUIRE(ctx, !ctx->comm, "comm_init: this context already has a communicator");
  ctx->comm_rank = rank; ctx->comm_world = world;
  if (world == 1 && !id) return NG_OK;            // nothing to exchange with: ng_allreduce_grads is the identity
  NG_REQUIRE(ctx, id != nullptr, "comm_init: unique id of rank 0 required");   // (a world of one WITH an id runs on RCCL too)
  if (!rccl_load()) return ng::fail(ctx, NG_ERR_UNSUPPORTED, "comm_init: librccl.so not found");
  ng::DeviceGuard dg(ctx->device);                // the communicator binds to the context's GPU
  RcclId uid;
  memcpy(&uid, id, sizeof(uid));
  RcclComm c = nullptr;
  const int rc = g_rccl.init_rank(&c, world, uid, rank);
  if (rc != 0) { ctx->comm_world = 1; ctx->comm_rank = 0; return rccl_fail(ctx, "ncclCommInitRank", rc); }
  ctx->comm = c;
  return NG_OK;
}

extern "C" int ng_comm_destroy(ng_ctx* ctx) {
  if (!ctx) return NG_ERR_INVALID;
  if (ctx->comm) {
    ng::DeviceGuard dg(ctx->device);
    (void)g_rccl.destroy((RcclComm)ctx->comm);
    ctx->comm = nullptr;
  }
  ctx->comm_world = 1; ctx->comm_rank = 0;
  return NG_OK;
}

extern "C" int ng_comm_world(ng_ctx* ctx) { return ctx ? ctx->comm_world : 0; }

extern "C" int ng_allreduce_grads(ng_ctx* ctx, void* stream, float* flat_grad, int64_t n) {
  if (!ctx) return NG_ERR_INVALID;
  NG_REQUIRE(ctx, n >= 0 && (flat_grad || n == 0), "allreduce_grads: buffer");
  if ((ctx->comm_world == 1 && !ctx->comm) || n == 0) return NG_OK;
  NG_REQUIRE(ctx, ctx->comm != nullptr, "allreduce_grads: ng_comm_init first");
  ng::DeviceGuard dg(ctx->device);
  const int rc = g_rccl.all_reduce(flat_grad, flat_grad, (size_t)n, RCCL_FLOAT32, RCCL_SUM, (RcclComm)ctx->comm, (hipStream_t)stream);
  if (rc != 0) return rccl_fail(ctx, "ncclAllReduce", rc);
  return NG_OK;
}
