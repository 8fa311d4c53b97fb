// Shared host-side definitions for libnmrgnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/nmrgnn_hip.h"

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
  // optional per-kernel hipEvent bracketing
  bool prof = false;
  std::vector<ng_prof_rec> recs;
  std::vector<hipEvent_t> pool;
  // cached device properties
  int num_cu = 256;
};

namespace ng {

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
