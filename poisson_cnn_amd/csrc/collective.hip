// The data-parallel collective behind the C-ABI: an in-place sum all-reduce of a flat fp32 buffer over RCCL (xGMI), issued on the handle's
// stream so that it orders with the kernels that produced the gradients.  It replaces what tf.distribute.MirroredStrategy does inside
// `with dist_strategy.scope()` (train/hpnn_legacy_train.py:37): one all-reduce of the gradient bucket per optimizer step.
// RCCL is bound at run time (dlopen of librccl.so.1 on the first pcnn_comm_* call): libpcnn.so itself keeps no link-time dependency on it,
// single-GPU users never load it, and a missing library is an ordinary error return.
// Rendezvous is the caller's business, as in NCCL: rank 0 calls pcnn_comm_unique_id and ships the 128 bytes to the other ranks over any
// channel it has (a file, MPI, torch.distributed's store ...); every rank then calls pcnn_comm_init with its rank.
#include "pcnn_internal.h"
#include <stdlib.h>
#include <dlfcn.h>
#include <string.h>

namespace {

// the part of the NCCL API used here, restated (rccl.h: ncclUniqueId is 128 bytes, ncclFloat = 7, ncclSum = 0, ncclSuccess = 0)
struct UniqueId { char internal[PCNN_UNIQUE_ID_BYTES]; };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*AllReduceFn)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream);
typedef int (*BroadcastFn)(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t stream);
typedef int (*CommDestroyFn)(void* comm);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
  void* lib = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  BroadcastFn broadcast = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
  std::string why;
};

Rccl& rccl() {
  static Rccl r;
  if (r.lib || !r.why.empty()) return r;
  // PCNN_RCCL_LIBRARY names the one library to try (deployments with RCCL outside the loader path; tests force a failure with it)
  const char* forced = getenv("PCNN_RCCL_LIBRARY");
  std::string first_error;
  for (const char* name : {forced ? forced : "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.lib) break;
    const char* e = dlerror();               // ONE call per failure: dlerror() returns the message once and clears it
    if (first_error.empty()) first_error = e ? e : "unknown dlopen error";
    if (forced) break;
  }
  if (!r.lib) { r.why = std::string("cannot load ") + (forced ? forced : "librccl.so.1") + ": " + first_error; return r; }
  r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.lib, "ncclGetUniqueId"));
  r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.lib, "ncclCommInitRank"));
  r.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(r.lib, "ncclAllReduce"));
  r.broadcast = reinterpret_cast<BroadcastFn>(dlsym(r.lib, "ncclBroadcast"));
  r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.lib, "ncclCommDestroy"));
  r.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(r.lib, "ncclGetErrorString"));
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.broadcast || !r.comm_destroy) {
    r.why = "librccl.so.1 lacks an expected nccl* symbol";
    dlclose(r.lib); r.lib = nullptr;
  }
  return r;
}

const char* estr(Rccl& r, int rc) { return r.error_string ? r.error_string(rc) : "rccl error"; }

}  // namespace

// Host-only: 0 when RCCL can be bound, otherwise 1 with the loader's reason in `why` (always NUL-terminated).  Needs no handle and no GPU.
extern "C" int pcnn_collective_available(char* why, size_t why_bytes) {
  Rccl& r = rccl();
  if (why && why_bytes) snprintf(why, why_bytes, "%s", r.lib ? "" : r.why.c_str());
  return r.lib ? 0 : 1;
}

void pcnn_comm_release(pcnn_handle_s* h) {
  if (h->comm) { Rccl& r = rccl(); if (r.lib) (void)r.comm_destroy(h->comm); h->comm = nullptr; h->comm_size = 0; }
}

extern "C" int pcnn_comm_unique_id(pcnn_handle h, void* id_out) {
  PCNN_REQUIRE(h, h && id_out, "pcnn_comm_unique_id: null argument");
  Rccl& r = rccl();
  PCNN_REQUIRE(h, r.lib, "pcnn_comm_unique_id: %s", r.why.c_str());
  UniqueId id;
  memset(&id, 0, sizeof(id));
  const int rc = r.get_unique_id(&id);
  PCNN_REQUIRE(h, rc == 0, "pcnn_comm_unique_id: %s", estr(r, rc));
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int pcnn_comm_init(pcnn_handle h, const void* id, int rank, int world_size) {
  PCNN_REQUIRE(h, h && id, "pcnn_comm_init: null argument");
  PCNN_REQUIRE(h, world_size >= 1 && rank >= 0 && rank < world_size, "pcnn_comm_init: rank %d outside [0, %d)", rank, world_size);
  PCNN_REQUIRE(h, !h->comm, "pcnn_comm_init: this handle already has a communicator (pcnn_comm_destroy it first)");
  Rccl& r = rccl();
  PCNN_REQUIRE(h, r.lib, "pcnn_comm_init: %s", r.why.c_str());
  PCNN_REQUIRE(h, hipSetDevice(h->device) == hipSuccess, "pcnn_comm_init: cannot select device %d", h->device);
  UniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  void* comm = nullptr;
  const int rc = r.comm_init_rank(&comm, world_size, uid, rank);
  PCNN_REQUIRE(h, rc == 0 && comm, "pcnn_comm_init: ncclCommInitRank: %s", estr(r, rc));
  h->comm = comm; h->comm_rank = rank; h->comm_size = world_size;
  return 0;
}

extern "C" int pcnn_comm_destroy(pcnn_handle h) {
  PCNN_REQUIRE(h, h, "pcnn_comm_destroy: null handle");
  if (h->comm) { (void)hipStreamSynchronize(h->stream); pcnn_comm_release(h); }
  return 0;
}

extern "C" int pcnn_allreduce(pcnn_handle h, float* buf, size_t count) {
  PCNN_REQUIRE(h, h && (buf || count == 0), "pcnn_allreduce: null argument");
  PCNN_REQUIRE(h, h->comm, "pcnn_allreduce: no communicator on this handle (call pcnn_comm_init)");
  if (count == 0) return 0;
  Rccl& r = rccl();
  const int rc = r.all_reduce(buf, buf, count, /*ncclFloat*/ 7, /*ncclSum*/ 0, h->comm, h->stream);
  PCNN_REQUIRE(h, rc == 0, "pcnn_allreduce: ncclAllReduce: %s", estr(r, rc));
  return 0;
}

extern "C" int pcnn_broadcast(pcnn_handle h, float* buf, size_t count, int root) {
  PCNN_REQUIRE(h, h && (buf || count == 0), "pcnn_broadcast: null argument");
  PCNN_REQUIRE(h, h->comm, "pcnn_broadcast: no communicator on this handle (call pcnn_comm_init)");
  PCNN_REQUIRE(h, root >= 0 && root < h->comm_size, "pcnn_broadcast: root %d outside [0, %d)", root, h->comm_size);
  if (count == 0) return 0;
  Rccl& r = rccl();
  const int rc = r.broadcast(buf, buf, count, 7, root, h->comm, h->stream);
  PCNN_REQUIRE(h, rc == 0, "pcnn_broadcast: ncclBroadcast: %s", estr(r, rc));
  return 0;
}
