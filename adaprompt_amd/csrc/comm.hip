// adap_allreduce_bucket: the data-parallel gradient exchange as a C entry (SURVEY.md 8b "minimum surface":
// allreduce_bucket(ptr, count, dtype, comm, stream)) -- RCCL's ncclAllReduce on a communicator this library creates itself
// (ncclCommInitRank from a unique id the caller passes between its processes by any means), so a host that is not PyTorch can
// drive the exchange.  The reference's exchange is Lightning DDP's bucketed NCCL all-reduce of the trainable gradients after
// every micro-batch backward (main.py:829 strategy="ddp"; no no_sync anywhere in the tree).  RCCL is resolved at run time
// (dlopen of librccl.so.1 -- the copy the process already has, e.g. PyTorch's, when there is one), not linked: a process that
// never exchanges gradients does not need it.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return ADAP_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return adap_set_error(ADAP_ERR_UNSUPPORTED, "librccl.so not found: %s", dlerror());
    Rccl r;
    r.h = h;
    *(void**)&r.GetUniqueId = dlsym(h, "ncclGetUniqueId");
    *(void**)&r.CommInitRank = dlsym(h, "ncclCommInitRank");
    *(void**)&r.AllReduce = dlsym(h, "ncclAllReduce");
    *(void**)&r.CommDestroy = dlsym(h, "ncclCommDestroy");
    *(void**)&r.GetErrorString = dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString)
        return adap_set_error(ADAP_ERR_UNSUPPORTED, "librccl.so lacks an expected symbol");
    g_rccl = r;
    return ADAP_OK;
}

int check(ncclResult_t rc, const char* what) {
    if (rc == ncclSuccess) return ADAP_OK;
    return adap_set_error(ADAP_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString(rc));
}

}  // namespace

extern "C" int adap_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int adap_comm_unique_id(void* id_out) {
    ADAP_REQUIRE(id_out, ADAP_ERR_SHAPE, "comm_unique_id: null pointer");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    if ((rc = check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId"))) return rc;
    memcpy(id_out, &id, NCCL_UNIQUE_ID_BYTES);
    return ADAP_OK;
}

extern "C" int adap_comm_init(void** comm_out, const void* id, int nranks, int rank) {
    ADAP_REQUIRE(comm_out && id && nranks >= 1 && rank >= 0 && rank < nranks, ADAP_ERR_SHAPE, "comm_init: bad arguments");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    if ((rc = check(g_rccl.CommInitRank(&comm, nranks, uid, rank), "ncclCommInitRank"))) return rc;
    *comm_out = (void*)comm;
    return ADAP_OK;
}

extern "C" int adap_allreduce_bucket(void* comm, void* buf, long count, int dtype, int average, void* stream) {
    ADAP_REQUIRE(comm && buf && count >= 0, ADAP_ERR_SHAPE, "allreduce_bucket: bad arguments");
    ADAP_REQUIRE(dtype == 0 || dtype == 1, ADAP_ERR_UNSUPPORTED, "allreduce_bucket: dtype %d (0 = f32, 1 = bf16)", dtype);
    int rc = load_rccl();
    if (rc) return rc;
    if (count == 0) return ADAP_OK;
    return check(g_rccl.AllReduce(buf, buf, (size_t)count, dtype == 0 ? ncclFloat32 : ncclBfloat16, average ? ncclAvg : ncclSum,
                                  (ncclComm_t)comm, (hipStream_t)stream),
                 "ncclAllReduce");
}

extern "C" int adap_comm_destroy(void* comm) {
    if (!comm) return ADAP_OK;
    int rc = load_rccl();
    if (rc) return rc;
    return check(g_rccl.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
}
