// agp_comm.h -- the collective behind agp_comm_* (include/agp_hip.h): sum all-reduce of device buffers across the
// ranks of a multi-GPU run, one process per GPU.
//
// Two transports:
//   * RCCL (the real one: rings over xGMI).  librccl is NOT a link-time dependency of libagp_hip.so: it is resolved with
//     dlopen at agp_comm_init, preferring an instance the process already holds (a Julia host with AMDGPU.jl, or torch,
//     ships its own librccl.so.1 -- two RCCL instances in one process would each bring their own HIP runtime state), then
//     $AGP_RCCL_PATH, then the loader's search path, then /opt/rocm/lib.
//   * a host-supplied callback (agp_comm_init_callback): MPI, a gloo group, shared memory between two processes that
//     share one GPU -- whatever the host has.  The library calls it with the device pointer and the stream to order on.
// All collectives of this path are in-place sums (the three exchange points of SURVEY.md section 8e: LogisticSoftMax
// sum_k gamma_k, the multi-output (mean_f, var_f) exchange, the batch statistics; plus the tied-Z hyper-gradient).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only; the symbols are bound at run time

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/agp_hip.h"

namespace agp {

struct RcclApi {
  void* handle = nullptr;
  std::string where;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};

// returns nullptr and fills `why` when no usable librccl is found
inline RcclApi* rccl_api(std::string& why) {
  static RcclApi api;
  static bool tried = false;
  static std::string err;
  if (!tried) {
    tried = true;
    std::vector<std::pair<std::string, int>> cand;
    cand.push_back({"librccl.so.1", RTLD_NOW | RTLD_NOLOAD});  // already in the process (torch / AMDGPU.jl)
    cand.push_back({"librccl.so", RTLD_NOW | RTLD_NOLOAD});
    if (const char* p = getenv("AGP_RCCL_PATH")) cand.push_back({p, RTLD_NOW});
    cand.push_back({"librccl.so.1", RTLD_NOW});
    cand.push_back({"/opt/rocm/lib/librccl.so.1", RTLD_NOW});
    for (auto& c : cand) {
      void* h = dlopen(c.first.c_str(), c.second);
      if (!h) continue;
      api.handle = h;
      api.where = c.first + ((c.second & RTLD_NOLOAD) ? " (already loaded)" : "");
      break;
    }
    if (!api.handle) {
      err = "librccl not found (set AGP_RCCL_PATH)";
    } else {
#define AGP_RCCL_SYM(field, name)                                      \
  api.field = (decltype(api.field))dlsym(api.handle, name);            \
  if (!api.field && err.empty()) err = std::string("librccl lacks ") + name
      AGP_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
      AGP_RCCL_SYM(CommInitRank, "ncclCommInitRank");
      AGP_RCCL_SYM(CommDestroy, "ncclCommDestroy");
      AGP_RCCL_SYM(AllReduce, "ncclAllReduce");
      AGP_RCCL_SYM(GetErrorString, "ncclGetErrorString");
      AGP_RCCL_SYM(GetVersion, "ncclGetVersion");
#undef AGP_RCCL_SYM
    }
  }
  if (!err.empty()) {
    why = err;
    return nullptr;
  }
  return &api;
}

}  // namespace agp

struct agp_comm {
  agp_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  int kind = 0;  // 0 RCCL, 1 host callback
  ncclComm_t nccl = nullptr;
  agp_allreduce_fn fn = nullptr;
  void* user = nullptr;
  // accounting for bench.py: bytes reduced and (optionally) HIP-event time of the collectives since the last read
  int64_t n_calls = 0, bytes = 0;
  bool timing = false;
  int timing_every = 1;      // every n-th collective is bracketed by events (two records cost the stream ~20 us)
  int64_t n_timed = 0;       // collectives bracketed since the last agp_comm_stats
  bool timing_now = false;
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  // AGP_SPLIT_OVERLAP: the collective's own stream (highest priority) and the two events that tie it to the ctx's stream --
  // `ev_in`: what the collective reduces is complete; `ev_out`: its last group is done (only waited for by a flush)
  hipStream_t side = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
};
