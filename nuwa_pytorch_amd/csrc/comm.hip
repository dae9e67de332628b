// Gradient exchange of the data-parallel step straight on RCCL (amdnuwa_comm_init / allreduce / destroy).
//
// The reference trains through plain DDP-less single-GPU code (train_nuwa.py); its multi-GPU story is torch's.  Here one process
// per GPU owns one communicator; the flat fp32 gradient buckets of nuwa_pytorch_amd/distributed.py go through ncclAllReduce with
// ncclAvg (no separate division pass), or reduce-scatter + all-gather, on the reducer's private HIP stream.
//
// librccl is opened at run time (dlopen): the library has no link-time dependency on it, builds without the RCCL headers (the few
// declarations it needs are restated under __has_include below), loads on hosts without RCCL, and inside a
// torch process picks up the librccl.so.1 torch already mapped -- one RCCL per process.  xGMI is point to point (7 links per GPU),
// ring collectives are bound per link: the caller sizes its buckets for that, this file adds no policy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// ROCm install without the RCCL development headers: the handful of declarations this file uses, as rccl.h (NCCL 2.x ABI) has them.
// librccl itself is only ever opened at run time; without it every amdnuwa_comm_* entry point reports AMDNUWA_ERR_UNSUPPORTED.
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8, ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
#endif
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include "common.h"
#include "../../include/amdnuwa.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
thread_local char g_comm_err[256] = "";

void open_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) return;
    bool all = true;
    auto sym = [&](const char* name) { void* p = dlsym(g_rccl.handle, name); all = all && p; return p; };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.ReduceScatter = (decltype(g_rccl.ReduceScatter))sym("ncclReduceScatter");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.ok = all;
}

const Rccl* rccl() {
    std::call_once(g_rccl_once, open_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

int fail(const Rccl* r, ncclResult_t e, const char* what) {
    snprintf(g_comm_err, sizeof g_comm_err, "%s: %s", what, r && r->GetErrorString ? r->GetErrorString(e) : "rccl error");
    return AMDNUWA_ERR_COMM;
}

}  // namespace

struct amdnuwa_comm {
    ncclComm_t comm;
    int rank, world, device;
};

extern "C" const char* amdnuwa_comm_last_error(void) { return g_comm_err; }

extern "C" int amdnuwa_comm_available(void) { return rccl() != nullptr; }

extern "C" int amdnuwa_comm_unique_id(void* id_out, size_t bytes) {
    if (!id_out || bytes < AMDNUWA_COMM_ID_BYTES) return AMDNUWA_ERR_ARG;
    const Rccl* r = rccl();
    if (!r) return AMDNUWA_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == AMDNUWA_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return fail(r, e, "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof id);
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_comm_init(amdnuwa_comm** out, const void* id, size_t id_bytes, int rank, int world, int device) {
    if (!out || !id || id_bytes < AMDNUWA_COMM_ID_BYTES || world <= 0 || rank < 0 || rank >= world || device < 0) return AMDNUWA_ERR_ARG;
    *out = nullptr;
    const Rccl* r = rccl();
    if (!r) return AMDNUWA_ERR_UNSUPPORTED;
    int prev = -1;
    (void)hipGetDevice(&prev);                                   // the caller's current device is restored below
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return (int)he;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t c;
    const ncclResult_t e = r->CommInitRank(&c, world, uid, rank);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != ncclSuccess) return fail(r, e, "ncclCommInitRank");
    *out = new amdnuwa_comm{c, rank, world, device};
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_comm_rank(const amdnuwa_comm* c) { return c ? c->rank : AMDNUWA_ERR_ARG; }
extern "C" int amdnuwa_comm_world(const amdnuwa_comm* c) { return c ? c->world : AMDNUWA_ERR_ARG; }

extern "C" int amdnuwa_comm_allreduce(amdnuwa_comm* c, float* buf, size_t count, int average, hipStream_t stream) {
    if (!c || (!buf && count)) return AMDNUWA_ERR_ARG;
    if (!count) return AMDNUWA_OK;
    const Rccl* r = rccl();
    const ncclResult_t e = r->AllReduce(buf, buf, count, ncclFloat32, average ? ncclAvg : ncclSum, c->comm, stream);
    return e == ncclSuccess ? AMDNUWA_OK : fail(r, e, "ncclAllReduce");
}

// reduce-scatter + all-gather over a store of world * shard elements (the 'rs_ag' exchange: every link carries 1 / world of the
// bucket twice instead of the ring all-reduce's 2 (world - 1) / world -- the same bytes, but a rank may run its optimiser shard
// between the two halves); in place, rank r's shard at buf + r * shard
extern "C" int amdnuwa_comm_reduce_scatter_allgather(amdnuwa_comm* c, float* buf, size_t shard, int average, hipStream_t stream) {
    if (!c || (!buf && shard)) return AMDNUWA_ERR_ARG;
    if (!shard) return AMDNUWA_OK;
    const Rccl* r = rccl();
    float* mine = buf + (size_t)c->rank * shard;
    ncclResult_t e = r->ReduceScatter(buf, mine, shard, ncclFloat32, average ? ncclAvg : ncclSum, c->comm, stream);
    if (e != ncclSuccess) return fail(r, e, "ncclReduceScatter");
    e = r->AllGather(mine, buf, shard, ncclFloat32, c->comm, stream);
    return e == ncclSuccess ? AMDNUWA_OK : fail(r, e, "ncclAllGather");
}

extern "C" int amdnuwa_comm_broadcast(amdnuwa_comm* c, void* buf, size_t bytes, int root, hipStream_t stream) {
    if (!c || (!buf && bytes) || root < 0 || root >= c->world) return AMDNUWA_ERR_ARG;
    if (!bytes) return AMDNUWA_OK;
    const Rccl* r = rccl();
    const ncclResult_t e = r->Broadcast(buf, buf, bytes, ncclUint8, root, c->comm, stream);
    return e == ncclSuccess ? AMDNUWA_OK : fail(r, e, "ncclBroadcast");
}

extern "C" int amdnuwa_comm_destroy(amdnuwa_comm* c) {
    if (!c) return AMDNUWA_OK;
    const Rccl* r = rccl();
    const ncclResult_t e = r ? r->CommDestroy(c->comm) : ncclSuccess;
    delete c;
    return e == ncclSuccess ? AMDNUWA_OK : fail(r, e, "ncclCommDestroy");
}
