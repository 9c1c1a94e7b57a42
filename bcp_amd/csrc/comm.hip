// bcp_amd/csrc/comm.hip -- the ONE exchange of the data-parallel BCP step: all-reduce (sum) of the flat fp32
// gradient buffer over RCCL / xGMI, behind the C ABI (include/bcp_hip.h: bcp_comm_*, bcp_allreduce_f32).
//
// The reference has no multi-GPU code for LA / ACDC (its only precedent is nn.DataParallel in pancreas/dataloaders.py:14);
// this is the exchange a DDP-style run of LA_BCP_train.py:265-267 (loss.backward(); optimizer.step()) needs between the two.
//
// librccl.so is opened lazily with dlopen on the first bcp_comm_* call: single-GPU users never load it, and libbcp_hip.so has
// no link-time dependency on it.  One communicator per process (one process per GPU); the caller orders the collective with
// its own streams / events -- nothing here synchronises the host.
#include "common.h"
#include "../../include/bcp_hip.h"
#include <cstring>
#include <dlfcn.h>

namespace {

typedef int ncclResult_t_;                       // ncclSuccess == 0
typedef struct { char internal[128]; } ncclUniqueId_;
typedef void* ncclComm_t_;
enum { kNcclFloat = 7, kNcclSum = 0 };           // rccl.h: ncclFloat32 = 7, ncclSum = 0

struct Rccl {
  void* lib = nullptr;
  ncclResult_t_ (*GetUniqueId)(ncclUniqueId_*) = nullptr;
  ncclResult_t_ (*CommInitRank)(ncclComm_t_*, int, ncclUniqueId_, int) = nullptr;
  ncclResult_t_ (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
  ncclResult_t_ (*CommDestroy)(ncclComm_t_) = nullptr;
  ncclResult_t_ (*CommCount)(const ncclComm_t_, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t_) = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
      r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
      r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
      if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) { dlclose(r.lib); r.lib = nullptr; }
    }
  }
  return r.lib ? &r : nullptr;
}

int fail(const char* what, ncclResult_t_ rc) {
  Rccl* r = rccl();
  bcp::set_error("%s: RCCL error %d (%s)", what, rc, (r && r->GetErrorString) ? r->GetErrorString(rc) : "?");
  return BCP_ELAUNCH;
}

}  // namespace

#define BCP_NEED_RCCL(name)                                                                        \
  Rccl* R = rccl();                                                                                \
  if (!R) { bcp::set_error(name ": librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing"); return BCP_EUNSUP; }

extern "C" int bcp_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" int bcp_comm_unique_id(void* id128) {
  BCP_REQUIRE(id128, "bcp_comm_unique_id: null pointer");
  BCP_NEED_RCCL("bcp_comm_unique_id")
  ncclUniqueId_ id;
  const ncclResult_t_ rc = R->GetUniqueId(&id);
  if (rc) return fail("ncclGetUniqueId", rc);
  memcpy(id128, id.internal, 128);
  return BCP_OK;
}

extern "C" int bcp_comm_init_rank(void** comm, int world, int rank, const void* id128) {
  BCP_REQUIRE(comm && id128 && world >= 1 && rank >= 0 && rank < world, "bcp_comm_init_rank: bad arguments");
  BCP_NEED_RCCL("bcp_comm_init_rank")
  ncclUniqueId_ id;
  memcpy(id.internal, id128, 128);
  ncclComm_t_ c = nullptr;
  const ncclResult_t_ rc = R->CommInitRank(&c, world, id, rank);
  if (rc) return fail("ncclCommInitRank", rc);
  *comm = c;
  return BCP_OK;
}

extern "C" int bcp_comm_count(void* comm, int* world) {
  BCP_REQUIRE(comm && world, "bcp_comm_count: null pointer");
  BCP_NEED_RCCL("bcp_comm_count")
  if (!R->CommCount) { bcp::set_error("bcp_comm_count: ncclCommCount missing"); return BCP_EUNSUP; }
  const ncclResult_t_ rc = R->CommCount(comm, world);
  if (rc) return fail("ncclCommCount", rc);
  return BCP_OK;
}

extern "C" int bcp_allreduce_f32(void* comm, float* buf, long long n, void* stream) {
  BCP_REQUIRE(comm && buf && n > 0, "bcp_allreduce_f32: bad arguments");
  BCP_NEED_RCCL("bcp_allreduce_f32")
  const ncclResult_t_ rc = R->AllReduce(buf, buf, (size_t)n, kNcclFloat, kNcclSum, comm, (hipStream_t)stream);
  if (rc) return fail("ncclAllReduce", rc);
  return BCP_OK;
}

extern "C" int bcp_comm_destroy(void* comm) {
  if (!comm) return BCP_OK;
  BCP_NEED_RCCL("bcp_comm_destroy")
  const ncclResult_t_ rc = R->CommDestroy(comm);
  if (rc) return fail("ncclCommDestroy", rc);
  return BCP_OK;
}
