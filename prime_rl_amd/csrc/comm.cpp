// comm.cpp — pm_comm_*: thin RCCL wrappers behind the C ABI (SURVEY.md §8(b) minimum set).
//
// One communicator = one ncclComm_t created from a 128-byte unique id that the caller distributes out of band (the Python side
// uses a torch TCPStore, the same rendezvous shape as the reference's StatelessProcessGroup + PyNcclCommunicator for the weight
// push: src/prime_rl/transports/weights/nccl.py:128-136, inference/vllm/worker/nccl.py:63-76).  Every collective is ENQUEUED on the
// caller's hipStream_t and returns; ordering against compute is the caller's business (events), exactly like the kernels.
//
// librccl is resolved at first use with dlopen: a process that already carries an RCCL (PyTorch-ROCm bundles one with the same
// SONAME) keeps using that single copy — two RCCL runtimes in one process would each own half of the xGMI bootstrap state.
#include "common.hip.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <mutex>

namespace {

struct Api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
};

Api g_api;
std::once_flag g_once;
bool g_ok = false;

template <typename F>
bool sym(F& f, const char* name) {
  f = reinterpret_cast<F>(dlsym(g_api.handle, name));
  return f != nullptr;
}

void load_once() {
  // 1) the copy already mapped into this process (PyTorch's), 2) the default search path, 3) the ROCm install
  const char* tries[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  g_api.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!g_api.handle) g_api.handle = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  for (int i = 0; !g_api.handle && i < 3; ++i) g_api.handle = dlopen(tries[i], RTLD_NOW | RTLD_GLOBAL);
  if (!g_api.handle) return;
  g_ok = sym(g_api.GetUniqueId, "ncclGetUniqueId") && sym(g_api.CommInitRank, "ncclCommInitRank") && sym(g_api.CommDestroy, "ncclCommDestroy") &&
         sym(g_api.CommCount, "ncclCommCount") && sym(g_api.CommUserRank, "ncclCommUserRank") && sym(g_api.AllReduce, "ncclAllReduce") &&
         sym(g_api.Broadcast, "ncclBroadcast") && sym(g_api.ReduceScatter, "ncclReduceScatter") && sym(g_api.AllGather, "ncclAllGather") &&
         sym(g_api.AllToAll, "ncclAllToAll") && sym(g_api.Send, "ncclSend") && sym(g_api.Recv, "ncclRecv") && sym(g_api.GroupStart, "ncclGroupStart") &&
         sym(g_api.GroupEnd, "ncclGroupEnd") && sym(g_api.GetErrorString, "ncclGetErrorString") && sym(g_api.GetVersion, "ncclGetVersion");
}

bool ready(const char* who) {
  std::call_once(g_once, load_once);
  if (!g_ok) pm::set_error("%s: librccl could not be loaded (%s)", who, g_api.handle ? "missing symbol" : "dlopen failed");
  return g_ok;
}

int fail(const char* who, ncclResult_t r) {
  pm::set_error("%s: RCCL error %d (%s)", who, (int)r, g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
  return PM_ERR_LAUNCH;
}

bool dtype_of(int pm_dtype, ncclDataType_t* out) {
  switch (pm_dtype) {
    case PM_DTYPE_F32: *out = ncclFloat32; return true;
    case PM_DTYPE_BF16: *out = ncclBfloat16; return true;
    case PM_DTYPE_I64: *out = ncclInt64; return true;
    case PM_DTYPE_U8: *out = ncclUint8; return true;
    default: return false;
  }
}

#define PM_COMM_PROLOGUE(who)                                                                   \
  if (!ready(who)) return PM_ERR_UNSUPPORTED;                                                   \
  if (!comm) { pm::set_error("%s: null communicator", who); return PM_ERR_INVALID_ARG; }        \
  ncclDataType_t dt;                                                                            \
  if (!dtype_of(dtype, &dt)) { pm::set_error("%s: unsupported dtype %d", who, dtype); return PM_ERR_INVALID_ARG; }

}  // namespace

extern "C" int pm_comm_available(void) { return ready("pm_comm_available") ? 1 : 0; }

extern "C" int pm_comm_rccl_version(void) {
  if (!ready("pm_comm_rccl_version")) return PM_ERR_UNSUPPORTED;
  int v = 0;
  return g_api.GetVersion(&v) == ncclSuccess ? v : PM_ERR_LAUNCH;
}

extern "C" int pm_comm_unique_id(void* id_host_128) {
  if (!ready("pm_comm_unique_id")) return PM_ERR_UNSUPPORTED;
  if (!id_host_128) return PM_ERR_INVALID_ARG;
  ncclUniqueId id;
  ncclResult_t r = g_api.GetUniqueId(&id);
  if (r != ncclSuccess) return fail("pm_comm_unique_id", r);
  memcpy(id_host_128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return PM_OK;
}

extern "C" int pm_comm_init(void** comm_out, const void* id_host_128, int32_t n_ranks, int32_t rank) {
  if (!ready("pm_comm_init")) return PM_ERR_UNSUPPORTED;
  if (!comm_out || !id_host_128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) { pm::set_error("pm_comm_init: bad arguments"); return PM_ERR_INVALID_ARG; }
  ncclUniqueId id;
  memcpy(id.internal, id_host_128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  ncclResult_t r = g_api.CommInitRank(&c, n_ranks, id, rank);  // uses the calling thread's current HIP device
  if (r != ncclSuccess) return fail("pm_comm_init", r);
  *comm_out = c;
  return PM_OK;
}

extern "C" int pm_comm_destroy(void* comm) {
  if (!ready("pm_comm_destroy")) return PM_ERR_UNSUPPORTED;
  if (!comm) return PM_OK;
  ncclResult_t r = g_api.CommDestroy((ncclComm_t)comm);
  return r == ncclSuccess ? PM_OK : fail("pm_comm_destroy", r);
}

extern "C" int pm_comm_rank_size(void* comm, int32_t* rank_out, int32_t* size_out) {
  if (!ready("pm_comm_rank_size")) return PM_ERR_UNSUPPORTED;
  if (!comm) return PM_ERR_INVALID_ARG;
  int r = 0, n = 0;
  if (g_api.CommUserRank((ncclComm_t)comm, &r) != ncclSuccess || g_api.CommCount((ncclComm_t)comm, &n) != ncclSuccess) return PM_ERR_LAUNCH;
  if (rank_out) *rank_out = r;
  if (size_out) *size_out = n;
  return PM_OK;
}

extern "C" int pm_comm_all_reduce(void* comm, const void* send, void* recv, int64_t count, int dtype, void* stream) {
  PM_COMM_PROLOGUE("pm_comm_all_reduce");
  ncclResult_t r = g_api.AllReduce(send, recv, (size_t)count, dt, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  return r == ncclSuccess ? PM_OK : fail("pm_comm_all_reduce", r);
}

extern "C" int pm_comm_broadcast(void* comm, const void* send, void* recv, int64_t count, int dtype, int32_t root, void* stream) {
  PM_COMM_PROLOGUE("pm_comm_broadcast");
  ncclResult_t r = g_api.Broadcast(send, recv, (size_t)count, dt, root, (ncclComm_t)comm, (hipStream_t)stream);
  return r == ncclSuccess ? PM_OK : fail("pm_comm_broadcast", r);
}

extern "C" int pm_comm_reduce_scatter(void* comm, const void* send, void* recv, int64_t recv_count, int dtype, void* stream) {
  PM_COMM_PROLOGUE("pm_comm_reduce_scatter");
  ncclResult_t r = g_api.ReduceScatter(send, recv, (size_t)recv_count, dt, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  return r == ncclSuccess ? PM_OK : fail("pm_comm_reduce_scatter", r);
}

extern "C" int pm_comm_all_gather(void* comm, const void* send, void* recv, int64_t send_count, int dtype, void* stream) {
  PM_COMM_PROLOGUE("pm_comm_all_gather");
  ncclResult_t r = g_api.AllGather(send, recv, (size_t)send_count, dt, (ncclComm_t)comm, (hipStream_t)stream);
  return r == ncclSuccess ? PM_OK : fail("pm_comm_all_gather", r);
}

extern "C" int pm_comm_all_to_all(void* comm, const void* send, void* recv, int64_t count_per_peer, int dtype, void* stream) {
  PM_COMM_PROLOGUE("pm_comm_all_to_all");
  ncclResult_t r = g_api.AllToAll(send, recv, (size_t)count_per_peer, dt, (ncclComm_t)comm, (hipStream_t)stream);
  return r == ncclSuccess ? PM_OK : fail("pm_comm_all_to_all", r);
}

extern "C" int pm_comm_send(void* comm, const void* send, int64_t count, int dtype, int32_t peer, void* stream) {
  PM_COMM_PROLOGUE("pm_comm_send");
  ncclResult_t r = g_api.Send(send, (size_t)count, dt, peer, (ncclComm_t)comm, (hipStream_t)stream);
  return r == ncclSuccess ? PM_OK : fail("pm_comm_send", r);
}

extern "C" int pm_comm_recv(void* comm, void* recv, int64_t count, int dtype, int32_t peer, void* stream) {
  PM_COMM_PROLOGUE("pm_comm_recv");
  ncclResult_t r = g_api.Recv(recv, (size_t)count, dt, peer, (ncclComm_t)comm, (hipStream_t)stream);
  return r == ncclSuccess ? PM_OK : fail("pm_comm_recv", r);
}

extern "C" int pm_comm_group_start(void) {
  if (!ready("pm_comm_group_start")) return PM_ERR_UNSUPPORTED;
  ncclResult_t r = g_api.GroupStart();
  return r == ncclSuccess ? PM_OK : fail("pm_comm_group_start", r);
}

extern "C" int pm_comm_group_end(void) {
  if (!ready("pm_comm_group_end")) return PM_ERR_UNSUPPORTED;
  ncclResult_t r = g_api.GroupEnd();
  return r == ncclSuccess ? PM_OK : fail("pm_comm_group_end", r);
}
