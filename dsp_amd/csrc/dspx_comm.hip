// Optional exchange step of the sharded filter: an all-gather of result shards over RCCL (xGMI between the GPUs of
// a node), one process per GPU.  The filter itself needs no collective; this exists for callers that want the
// whole result on every GPU.  RCCL is bound at run time -- the copy the process already holds (PyTorch ships its
// own) or the system's -- so that libdspx.so carries no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: every call goes through the pointers below

#include <cstring>
#include <mutex>
#include <string>

#include "dspx.h"
#include "dspx_common.h"

namespace {

using dspx::fail;

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;  // why not, as the loader said it when it failed
};

const Rccl& rccl() {
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that is already loaded first (a second RCCL in one process would bring a second set of kernels)
        void* h = nullptr;
        for (const char* name : {"librccl.so", "librccl.so.1"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (h) break;
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) {
                const char* e = dlerror();  // (reading it clears it: once, right after the failure)
                api.why = e ? e : "dlopen failed";
            }
        }
        if (!h) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
        if (!api.ok) {
            const char* e = dlerror();
            api.why = e ? e : "a symbol of the RCCL API is missing";
        }
    });
    return api;
}

int need_rccl() {
    return rccl().ok ? DSPX_OK : fail(DSPX_ERR_NO_DEVICE, "RCCL (librccl.so) could not be loaded: %s", rccl().why.c_str());
}

#define DSPX_RCCL(call)                                                                              \
    do {                                                                                             \
        const ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) return fail(DSPX_ERR_HIP, "%s: %s", #call, rccl().GetErrorString(r_)); \
    } while (0)

}  // namespace

struct dspx_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
};

extern "C" {

static_assert(DSPX_COMM_ID_BYTES == sizeof(ncclUniqueId), "the id travels as an opaque byte string");

int dspx_comm_unique_id(void* id_bytes) {
    if (!id_bytes) return fail(DSPX_ERR_INVALID, "id_bytes is NULL");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    DSPX_RCCL(rccl().GetUniqueId(&id));
    std::memcpy(id_bytes, &id, sizeof id);
    return DSPX_OK;
}

int dspx_comm_create(const void* id_bytes, int nranks, int rank, dspx_comm** comm) {
    if (!id_bytes || !comm) return fail(DSPX_ERR_INVALID, "NULL pointer");
    *comm = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(DSPX_ERR_INVALID, "rank %d of %d", rank, nranks);
    if (int rc = need_rccl()) return rc;
    return dspx::guarded([&] {
        dspx_comm* c = new dspx_comm;
        c->nranks = nranks, c->rank = rank;
        ncclUniqueId id;
        std::memcpy(&id, id_bytes, sizeof id);
        if (hipGetDevice(&c->device) != hipSuccess) {
            delete c;
            return fail(DSPX_ERR_NO_DEVICE, "no current HIP device");
        }
        const ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, id, rank);
        if (r != ncclSuccess) {
            delete c;
            return fail(DSPX_ERR_HIP, "ncclCommInitRank: %s", rccl().GetErrorString(r));
        }
        *comm = c;
        return (int)DSPX_OK;
    });
}

int dspx_comm_ranks(const dspx_comm* comm) { return comm ? comm->nranks : 0; }

int dspx_allgather_f64_dev(dspx_comm* comm, const double* d_send, double* d_recv, size_t count, void* stream) {
    if (!comm) return fail(DSPX_ERR_INVALID, "comm is NULL");
    if (count && (!d_send || !d_recv)) return fail(DSPX_ERR_INVALID, "NULL buffer");
    if (count == 0) return DSPX_OK;
    // the communicator, its buffers and the stream belong to the device it was created on
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != comm->device)
        return fail(DSPX_ERR_INVALID, "the communicator was created on device %d, the current device is %d", comm->device, cur);
    DSPX_RCCL(rccl().AllGather(d_send, d_recv, count, ncclDouble, comm->comm, static_cast<hipStream_t>(stream)));
    return DSPX_OK;
}

void dspx_comm_destroy(dspx_comm* comm) {
    if (!comm) return;
    if (comm->comm && rccl().ok) {
        // destroyed with its own device current, whatever the caller's is (restored afterwards)
        int cur = -1;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        if (have && cur != comm->device) (void)hipSetDevice(comm->device);
        (void)rccl().CommDestroy(comm->comm);
        if (have && cur != comm->device) (void)hipSetDevice(cur);
    }
    delete comm;
}

}  // extern "C"
