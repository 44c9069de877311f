// comm.hip - data-parallel gradient exchange: one RCCL communicator per process (one process per GPU), owned by the
// library so the host VM can all-reduce its gradient slab in-order on its own stream with no Python in the loop.
// RCCL is resolved at run time (dlopen by SONAME: inside a torch process that is the librccl torch already loaded, so
// there is exactly one RCCL in the process); libt4hip.so itself has no link-time dependency on it.
// Reference: none - the reference is single-GPU; sharding contract in SURVEY.md 8(e).
#include "t4k_common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

using namespace t4k;

namespace {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
} R;

int load_rccl() {
    if (R.so) return T4K_OK;
    const char *names[] = { getenv("T4K_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char *n : names) { if (n && *n) { R.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (R.so) break; } }
    if (!R.so) return fail(T4K_ERR_UNSUPPORTED, "RCCL not found (%s)", dlerror());
    R.GetUniqueId    = (decltype(R.GetUniqueId))dlsym(R.so, "ncclGetUniqueId");
    R.CommInitRank   = (decltype(R.CommInitRank))dlsym(R.so, "ncclCommInitRank");
    R.AllReduce      = (decltype(R.AllReduce))dlsym(R.so, "ncclAllReduce");
    R.CommDestroy    = (decltype(R.CommDestroy))dlsym(R.so, "ncclCommDestroy");
    R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.so, "ncclGetErrorString");
    if (!R.GetUniqueId || !R.CommInitRank || !R.AllReduce || !R.CommDestroy) { R.so = nullptr; return fail(T4K_ERR_UNSUPPORTED, "RCCL symbols missing"); }
    return T4K_OK;
}
int nccl_fail(ncclResult_t r, const char *what) {
    return fail(T4K_ERR_HIP, "%s: %s", what, R.GetErrorString ? R.GetErrorString(r) : "rccl error");
}

} // namespace

extern "C" {

int t4k_comm_unique_id(void *id128) {
    T4K_REQUIRE_INIT();
    if (!id128) return fail(T4K_ERR_ARG, "t4k_comm_unique_id: null");
    int rc = load_rccl(); if (rc) return rc;
    ncclUniqueId id; ncclResult_t r = R.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId");
    memcpy(id128, &id, sizeof(id));
    return T4K_OK;
}
int t4k_comm_init(const void *id128, int rank, int world) {
    T4K_REQUIRE_INIT();
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(T4K_ERR_ARG, "t4k_comm_init: bad argument");
    int rc = load_rccl(); if (rc) return rc;
    if (R.comm) { R.CommDestroy(R.comm); R.comm = nullptr; }
    ncclUniqueId id; memcpy(&id, id128, sizeof(id));
    T4K_HIP(hipSetDevice(st().device));
    ncclResult_t r = R.CommInitRank(&R.comm, world, id, rank);
    if (r != ncclSuccess) { R.comm = nullptr; return nccl_fail(r, "ncclCommInitRank"); }
    R.rank = rank; R.world = world;
    st().shard_rank = rank; st().shard_world = world;     // dropout masks are keyed by the sample's place in the whole batch from now on
    return T4K_OK;
}
// world / rank of the data-parallel job: the RCCL communicator's, or the one-shot peer exchange's (xchg.hip) when only that is connected
int t4k_comm_world(void) { return R.comm ? R.world : t4k_xchg_world(); }
int t4k_comm_rank(void)  { return R.comm ? R.rank : t4k_xchg_rank(); }
int t4k_allreduce_sum(float *buf, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!R.comm && xchg().connected) {                          // no RCCL: the same sum over the peer windows (rank order, deterministic)
        if (n <= 0) return T4K_OK;
        if (!buf) return fail(T4K_ERR_ARG, "t4k_allreduce_sum: null");
        return xchg().world > 1 ? xchg_allreduce(buf, n, S(s)) : T4K_OK;
    }
    if (!R.comm) return fail(T4K_ERR_UNSUPPORTED, "t4k_allreduce_sum: no communicator (t4k_comm_init)");
    if (n <= 0) return T4K_OK;
    if (!buf) return fail(T4K_ERR_ARG, "t4k_allreduce_sum: null");
    ncclResult_t r = R.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, R.comm, S(s));
    if (r != ncclSuccess) return nccl_fail(r, "ncclAllReduce");
    return T4K_OK;
}
int t4k_comm_sync_batchnorm(int on) { st().bn_sync = on != 0; return T4K_OK; }
int t4k_comm_destroy(void) {
    if (R.comm && R.CommDestroy) { (void)hipDeviceSynchronize(); R.CommDestroy(R.comm); }
    R.comm = nullptr; R.world = 0; R.rank = 0;
    st().shard_rank = 0; st().shard_world = 1; st().bn_sync = false;
    return T4K_OK;
}

} // extern "C"
