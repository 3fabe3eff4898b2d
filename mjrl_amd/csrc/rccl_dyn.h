// rccl_dyn.h -- RCCL bound at run time (dlopen), so that libmjx.so has no link-time dependency on a particular
// librccl and a single-GPU user never loads it.  Only what the update path needs: one communicator per context
// (one process per GPU), sum all-reduces of fp32 / fp64 buffers on the launch stream, grouped calls.
//
// The multi-rank placement follows SURVEY 8e: every sample sum (gradient, Fisher-vector product, surrogate / KL sums)
// is formed locally with the GLOBAL sample count and all-reduced -- d floats once after K1 and once per CG iteration
// (22.8 KB at BASELINE configs[1]: latency-bound), a few doubles after K1 / K3.  The reference has no counterpart
// (its only parallelism is the sampler pool, mjrl/samplers/core.py:189-210).
#pragma once
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include <string>

namespace mjx {

struct RcclApi {
  // the subset of rccl.h this library calls (types restated: ncclUniqueId is 128 opaque bytes, passed BY VALUE)
  struct UniqueId { char internal[128]; };
  typedef void* Comm;
  enum { kSum = 0, kFloat32 = 7, kFloat64 = 8 };
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* handle = nullptr;
  std::string path, error;

  // the librccl this process already mapped (PyTorch-ROCm ships and loads its own), else the ROCm one
  static std::string mapped_rccl() {
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return "";
    char line[1024];
    std::string hit;
    while (fgets(line, sizeof line, f)) {
      const char* p = strstr(line, "librccl");
      if (!p) continue;
      const char* s = strchr(line, '/');
      if (!s) continue;
      hit.assign(s);
      while (!hit.empty() && (hit.back() == '\n' || hit.back() == ' ')) hit.pop_back();
      break;
    }
    fclose(f);
    return hit;
  }

  bool load() {
    if (handle) return true;
    if (const char* off = getenv("MJX_RCCL_DISABLE")) {      // (tests: the fall-back chain of engine._native_comm without a loadable RCCL)
      if (off[0] == '1') { error = "RCCL binding disabled (MJX_RCCL_DISABLE=1)"; return false; }
    }
    const char* env = getenv("MJX_RCCL_LIB");
    std::string cands[4] = {env ? env : "", mapped_rccl(), "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const std::string& c : cands) {
      if (c.empty()) continue;
      handle = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (handle) { path = c; break; }
    }
    if (!handle) { error = std::string("cannot dlopen librccl: ") + (dlerror() ? dlerror() : "?"); return false; }
    auto sym = [&](const char* n) { void* p = dlsym(handle, n); if (!p) error = std::string("librccl lacks ") + n; return p; };
    GetUniqueId = (int (*)(UniqueId*))sym("ncclGetUniqueId");
    CommInitRank = (int (*)(Comm*, int, UniqueId, int))sym("ncclCommInitRank");
    CommDestroy = (int (*)(Comm))sym("ncclCommDestroy");
    AllReduce = (int (*)(const void*, void*, size_t, int, int, Comm, void*))sym("ncclAllReduce");
    GroupStart = (int (*)())sym("ncclGroupStart");
    GroupEnd = (int (*)())sym("ncclGroupEnd");
    GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) {
      dlclose(handle); handle = nullptr;
      return false;
    }
    return true;
  }
};

inline RcclApi& rccl() {
  static RcclApi api;
  return api;
}

}  // namespace mjx
