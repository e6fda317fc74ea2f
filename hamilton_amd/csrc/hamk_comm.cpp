// hamk_comm.cpp -- the path's ONE collective for hosts that run one PROCESS per GPU: the final all-gather of a sharded ensemble
// over RCCL (xGMI), through the C ABI alone ("RCCL over xGMI only for the final gather").
// hamk_gather_batch (hamk_api.cpp) serves a single process that drives every GPU of the node with peer copies; a Haskell or C host
// started once per GPU (mpirun, torchrun-like launchers) has no peer pointers to hand over -- it needs a communicator.  bench.py
// reaches the same collective through torch.distributed; this file is what a host WITHOUT torch binds.
// RCCL is loaded on first use (dlopen): libhamk.so keeps no link-time dependency on it, and a box without RCCL only loses these four
// entry points (HAMK_ERR_UNSUPPORTED with the loader's message).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "hamk_host.h"      // (include/hamk.h under default visibility: hamk_internal.h)

using hamk_host::fail;

namespace {

// the part of RCCL's C interface used here (rccl.h: ncclResult_t is an int enum with ncclSuccess = 0, ncclDouble = 8, the unique id 128 opaque bytes)
struct RcclId { char internal[HAMK_COMM_ID_BYTES]; };
typedef void* RcclComm;
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;      // why it could not be loaded
};
constexpr int kRcclDouble = 8;

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
      r.why += std::string(r.why.empty() ? "" : "; ") + dlerror();
    }
    if (!r.lib) return;
    auto sym = [&](const char* s) -> void* {
      void* p = dlsym(r.lib, s);
      if (!p) r.why += std::string(r.why.empty() ? "" : "; ") + "missing symbol " + s;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Broadcast || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
      dlclose(r.lib);
      r.lib = nullptr;
    }
  });
  return r;
}

int need_rccl(Rccl*& out) {
  Rccl& r = rccl();
  if (!r.lib) return fail(HAMK_ERR_UNSUPPORTED, "RCCL is not available: " + r.why);
  out = &r;
  return HAMK_OK;
}

#define RCCL_TRY(r, expr)                                                                                      \
  do {                                                                                                         \
    const int e_ = (expr);                                                                                     \
    if (e_ != 0) return fail(HAMK_ERR_HIP, std::string(#expr) + ": " + (r)->GetErrorString(e_));                \
  } while (0)

}  // namespace

struct hamk_comm {
  RcclComm comm = nullptr;
  int world = 0, rank = 0, device = 0;
};

extern "C" {

int hamk_comm_unique_id(void* id) {
  if (!id) return fail(HAMK_ERR_INVALID, "hamk_comm_unique_id: null id");
  Rccl* r = nullptr;
  TRY(need_rccl(r));
  RcclId u;
  std::memset(&u, 0, sizeof u);
  RCCL_TRY(r, r->GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return HAMK_OK;
}

int hamk_comm_create(const void* id, int32_t world, int32_t rank, hamk_comm** out) {
  if (!out) return fail(HAMK_ERR_INVALID, "hamk_comm_create: out is null");
  *out = nullptr;
  if (!id) return fail(HAMK_ERR_INVALID, "hamk_comm_create: null id (hamk_comm_unique_id on one rank, the 128 bytes to the others)");
  if (world <= 0 || rank < 0 || rank >= world) return fail(HAMK_ERR_INVALID, "hamk_comm_create: need 0 <= rank < world");
  Rccl* r = nullptr;
  TRY(need_rccl(r));
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  RcclId u;
  std::memcpy(&u, id, sizeof u);
  RcclComm c = nullptr;
  RCCL_TRY(r, r->CommInitRank(&c, world, u, rank));
  hamk_comm* h = new hamk_comm;
  h->comm = c; h->world = world; h->rank = rank; h->device = dev;
  *out = h;
  return HAMK_OK;
}

int hamk_comm_allgather_batch(hamk_comm* comm, int32_t n, const int64_t* B_parts, const double* part, double* out) {
  if (!comm) return fail(HAMK_ERR_INVALID, "hamk_comm_allgather_batch: null communicator");
  if (n <= 0 || !B_parts) return fail(HAMK_ERR_INVALID, "hamk_comm_allgather_batch: bad n / null B_parts");
  Rccl* r = nullptr;
  TRY(need_rccl(r));
  int64_t total = 0;
  bool equal = true;
  std::vector<int64_t> at((size_t)comm->world);
  for (int g = 0; g < comm->world; ++g) {
    if (B_parts[g] < 0) return fail(HAMK_ERR_INVALID, "hamk_comm_allgather_batch: negative shard size");
    at[(size_t)g] = total;
    total += B_parts[g];
    equal = equal && B_parts[g] == B_parts[0];
  }
  if (hamk_host::test_env("HAMK_COMM_FORCE_BCAST")) equal = false;      // (tests: the ragged path on a one-rank communicator)
  if (total == 0) return HAMK_OK;
  const int64_t Bme = B_parts[comm->rank];
  if (!out || (Bme > 0 && !part)) return fail(HAMK_ERR_INVALID, "hamk_comm_allgather_batch: null part / out");
  int here = 0;
  HIP_TRY(hipGetDevice(&here));
  if (here != comm->device) return fail(HAMK_ERR_INVALID, "hamk_comm_allgather_batch: the communicator was created on device " + std::to_string(comm->device) +
                                                          ", the calling thread's current device is " + std::to_string(here));
  // Row j of every shard to columns [at_g, at_g + B_g) of row j of the output: with equal shards ONE all-gather per row (rank g's block lands
  // at out + j total + g B, which is where it belongs), all n of them fused in one group -- one launch of RCCL's kernel.  Ragged shards
  // (a last rank with fewer trajectories): one broadcast per (rank, row), fused the same way.
  RCCL_TRY(r, r->GroupStart());
  int rc = 0;
  for (int j = 0; j < n && rc == 0; ++j) {
    if (equal) {
      rc = r->AllGather(part + (int64_t)j * Bme, out + (int64_t)j * total, (size_t)Bme, kRcclDouble, comm->comm, nullptr);
    } else {
      for (int g = 0; g < comm->world && rc == 0; ++g) {
        if (B_parts[g] == 0) continue;
        double* dst = out + (int64_t)j * total + at[(size_t)g];
        const double* src = (g == comm->rank) ? part + (int64_t)j * Bme : dst;
        rc = r->Broadcast(src, dst, (size_t)B_parts[g], kRcclDouble, g, comm->comm, nullptr);
      }
    }
  }
  const int rc_end = r->GroupEnd();
  if (rc != 0) return fail(HAMK_ERR_HIP, std::string("hamk_comm_allgather_batch: ") + r->GetErrorString(rc));
  if (rc_end != 0) return fail(HAMK_ERR_HIP, std::string("hamk_comm_allgather_batch: ncclGroupEnd: ") + r->GetErrorString(rc_end));
  HIP_TRY(hipStreamSynchronize(nullptr));
  return HAMK_OK;
}

int hamk_comm_destroy(hamk_comm* comm) {
  if (!comm) return HAMK_OK;
  Rccl& r = rccl();
  int rc = 0;
  if (r.lib && comm->comm) rc = r.CommDestroy(comm->comm);
  delete comm;
  if (rc != 0) return fail(HAMK_ERR_HIP, std::string("ncclCommDestroy: ") + r.GetErrorString(rc));
  return HAMK_OK;
}

}  // extern "C"
