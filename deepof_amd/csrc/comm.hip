// Data-parallel exchange behind the C ABI (include/deepof_hip.h, "data-parallel exchange"): a thin, link-free binding of
// RCCL.  The reference wraps its models in DistributedDataParallel (/root/reference/deepof/clustering/training.py:1087-1096,
// 1321-1330, 1567-1576), which all-reduces bucketed gradients and broadcasts the parameters once; here the gradient is one
// contiguous buffer (86 KB - 1 MB), so the whole exchange is ONE ncclAllReduce on the step's stream, latency-bound on the
// xGMI links, and the initial broadcast one ncclBroadcast.  librccl is dlopen'ed on first use.
#include <dlfcn.h>

#include <cstring>

#include <mutex>

#include "dof_rt.h"
#include "launchers.h"

namespace {

// the slice of rccl.h this file needs (types are ABI-stable since NCCL 2.x)
struct RcclUniqueId { char internal[DOF_COMM_ID_BYTES]; };
typedef void* RcclComm;
constexpr int kRcclFloat32 = 7, kRcclSum = 0;
struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*CommAbort)(RcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

const Rccl* rccl() {
#if DOF_HAS_DEVICE_RUNTIME
  std::call_once(g_once, [] {
    // the soname first: inside a PyTorch process this is the copy torch.distributed already loaded
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) return;
    void* h = g_rccl.handle;
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(h, "ncclCommAbort"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(dlsym(h, "ncclBroadcast"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Broadcast;
  });
#endif
  if (!g_rccl.ok) {
    dof_set_error("dof_comm: librccl is not available in this process (dlopen / dlsym failed); no host fallback exists");
    return nullptr;
  }
  return &g_rccl;
}

int rccl_fail(const Rccl* r, int rc, const char* what) {
  dof_set_error("%s failed: RCCL error %d (%s)", what, rc, r->GetErrorString ? r->GetErrorString(rc) : "?");
  return DOF_ERR_LAUNCH;
}

}  // namespace

struct DofComm {
  RcclComm comm;
  int rank, world;
};

extern "C" int dof_comm_unique_id(void* id_out) {
  if (!id_out) {
    dof_set_error("dof_comm_unique_id: null argument");
    return DOF_ERR_ARG;
  }
  const Rccl* r = rccl();
  if (!r) return DOF_ERR_UNSUPPORTED;
  RcclUniqueId id;
  const int rc = r->GetUniqueId(&id);
  if (rc != 0) return rccl_fail(r, rc, "ncclGetUniqueId");
  memcpy(id_out, id.internal, DOF_COMM_ID_BYTES);
  return DOF_OK;
}

extern "C" int dof_comm_create(const void* id, int32_t rank, int32_t world, DofComm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) {
    dof_set_error("dof_comm_create: bad arguments (rank %d of %d)", (int)rank, (int)world);
    return DOF_ERR_ARG;
  }
  const Rccl* r = rccl();
  if (!r) return DOF_ERR_UNSUPPORTED;
  RcclUniqueId uid;
  memcpy(uid.internal, id, DOF_COMM_ID_BYTES);
  RcclComm c = nullptr;
  const int rc = r->CommInitRank(&c, world, uid, rank);
  if (rc != 0) return rccl_fail(r, rc, "ncclCommInitRank");
  *out = new DofComm{c, rank, world};
  return DOF_OK;
}

extern "C" int dof_comm_destroy(DofComm* comm) {
  if (!comm) return DOF_OK;
  const Rccl* r = rccl();
  int rc = 0;
  if (r) rc = r->CommDestroy(comm->comm);
  delete comm;
  return (r && rc != 0) ? rccl_fail(r, rc, "ncclCommDestroy") : DOF_OK;
}

extern "C" int dof_comm_abort(DofComm* comm) {
  if (!comm) return DOF_OK;
  const Rccl* r = rccl();
  int rc = 0;
  // without ncclCommAbort the communicator is leaked: ncclCommDestroy would block behind the collective being abandoned
  if (r && r->CommAbort) rc = r->CommAbort(comm->comm);
  delete comm;
  return (r && rc != 0) ? rccl_fail(r, rc, "ncclCommAbort") : DOF_OK;
}

extern "C" int dof_flat_allreduce(DofComm* comm, float* buf, int64_t n, void* stream) {
  if (!comm || !buf || n < 0) {
    dof_set_error("dof_flat_allreduce: bad arguments");
    return DOF_ERR_ARG;
  }
  const Rccl* r = rccl();
  if (!r) return DOF_ERR_UNSUPPORTED;
  if (n == 0) return DOF_OK;
  const int rc = r->AllReduce(buf, buf, (size_t)n, kRcclFloat32, kRcclSum, comm->comm, (hipStream_t)stream);
  return rc != 0 ? rccl_fail(r, rc, "ncclAllReduce") : DOF_OK;
}

extern "C" int dof_comm_broadcast(DofComm* comm, float* buf, int64_t n, int32_t root, void* stream) {
  if (!comm || !buf || n < 0 || root < 0 || root >= comm->world) {
    dof_set_error("dof_comm_broadcast: bad arguments");
    return DOF_ERR_ARG;
  }
  const Rccl* r = rccl();
  if (!r) return DOF_ERR_UNSUPPORTED;
  if (n == 0) return DOF_OK;
  const int rc = r->Broadcast(buf, buf, (size_t)n, kRcclFloat32, root, comm->comm, (hipStream_t)stream);
  return rc != 0 ? rccl_fail(r, rc, "ncclBroadcast") : DOF_OK;
}
