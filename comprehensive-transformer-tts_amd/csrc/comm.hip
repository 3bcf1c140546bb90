// comm.hip - the data-parallel collective behind the C ABI (include/ctts.h "Gradient all-reduce").
//
// Replaces what `DistributedDataParallel(model, device_ids=[rank])` does for the reference after backward (train.py:29-35,58,112):
// average the gradients of all ranks.  Here the gradients are ONE flat fp32 arena cut into a few buckets by backward stage, so the
// whole exchange is a handful of large in-place all-reduces (ncclAvg) over RCCL / xGMI on the stream the caller names - stream-ordered
// and graph-capturable like every other entry point.
//
// RCCL is bound at RUN time (dlopen of librccl.so.1, which resolves to the copy already in the process when PyTorch-ROCm loaded one):
// libctts_hip.so has no link-time dependency on it, single-GPU users never load it, and a host that is not Python (the C / C++ side
// of this ABI) needs nothing but these four calls to run data parallel.  Host code only - no kernel in this file.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <mutex>

#include "../../include/ctts.h"
#include "ctts_common.h"

namespace {

typedef int rccl_result_t;                         // ncclResult_t (0 = ncclSuccess)
struct rccl_unique_id { char internal[CTTS_COMM_ID_BYTES]; };      // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
enum { kFloat32 = 7, kAvg = 4 };                    // ncclFloat32, ncclAvg (rccl.h of ROCm 7.x; checked against the header by tests/test_abi_cpu.py)

struct Rccl {
  void* handle = nullptr;
  rccl_result_t (*GetUniqueId)(rccl_unique_id*) = nullptr;
  rccl_result_t (*CommInitRank)(void**, int, rccl_unique_id, int) = nullptr;
  rccl_result_t (*CommDestroy)(void*) = nullptr;
  rccl_result_t (*AllReduce)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
  const char* (*GetErrorString)(rccl_result_t) = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) return;
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.handle, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.handle, "ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.handle, "ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(g_rccl.handle, "ncclAllReduce");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.handle, "ncclGetErrorString");
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce;
}

bool rccl_ready(const char* who) {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.ok) {
    ctts_set_error("%s: RCCL is not available (dlopen librccl.so.1: %s)", who, g_rccl.handle ? "symbols missing" : dlerror());
    return false;
  }
  return true;
}

int fail(const char* who, rccl_result_t r) {
  ctts_set_error("%s: RCCL error %d (%s)", who, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return -1;
}

}  // namespace

extern "C" int ctts_comm_unique_id(void* id_out) {
  if (!id_out) { ctts_set_error("ctts_comm_unique_id: null output"); return -1; }
  if (!rccl_ready("ctts_comm_unique_id")) return -1;
  rccl_unique_id id;
  rccl_result_t r = g_rccl.GetUniqueId(&id);
  if (r != 0) return fail("ctts_comm_unique_id", r);
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int ctts_comm_create(void** comm_out, int32_t nranks, int32_t rank, const void* id) {
  if (!comm_out || !id || nranks < 1 || rank < 0 || rank >= nranks) {
    ctts_set_error("ctts_comm_create: bad arguments (nranks %d, rank %d)", (int)nranks, (int)rank);
    return -1;
  }
  if (!rccl_ready("ctts_comm_create")) return -1;
  rccl_unique_id uid;
  memcpy(&uid, id, sizeof(uid));
  void* comm = nullptr;
  rccl_result_t r = g_rccl.CommInitRank(&comm, nranks, uid, rank);      // binds the calling thread's CURRENT HIP device to this rank
  if (r != 0) return fail("ctts_comm_create", r);
  *comm_out = comm;
  return 0;
}

extern "C" int ctts_comm_destroy(void* comm) {
  if (!comm) return 0;
  if (!rccl_ready("ctts_comm_destroy")) return -1;
  rccl_result_t r = g_rccl.CommDestroy(comm);
  return r == 0 ? 0 : fail("ctts_comm_destroy", r);
}

extern "C" int ctts_allreduce_mean(float* buf, int64_t n, void* comm, void* stream) {
  if (n == 0) return 0;
  if (!buf || !comm || n < 0) { ctts_set_error("ctts_allreduce_mean: bad arguments"); return -1; }
  if (!rccl_ready("ctts_allreduce_mean")) return -1;
  rccl_result_t r = g_rccl.AllReduce(buf, buf, (size_t)n, kFloat32, kAvg, comm, stream);
  return r == 0 ? 0 : fail("ctts_allreduce_mean", r);
}
