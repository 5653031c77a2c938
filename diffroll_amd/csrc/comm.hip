// RCCL behind the C-ABI: the one collective of the path - the all-gather of the finished rolls over xGMI
//  - for callers that have no torch.distributed.  librccl is looked up at run time (dlopen): the
// engine library has no link-time dependency on it, and a process that already carries an RCCL (PyTorch loads its
// own copy) shares that one.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/diffroll_amd.h"

namespace {

constexpr int kUniqueIdBytes = 128;                 // NCCL_UNIQUE_ID_BYTES (rccl.h)
struct UniqueId { char internal[kUniqueIdBytes]; };  // ncclUniqueId: passed BY VALUE to ncclCommInitRank
typedef void* Comm;                                  // ncclComm_t
constexpr int kFloat32 = 7;                          // ncclFloat32
constexpr int kInt32 = 2;                            // ncclInt32

struct Rccl {
    void* lib = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
    Rccl() {
        // an RCCL already in the process (torch's) first, then the ROCm installation's
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (lib) break;
        }
        if (!lib)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (lib) break;
            }
        if (!lib) { why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        GetVersion = reinterpret_cast<int (*)(int*)>(sym("ncclGetVersion"));
        GetUniqueId = reinterpret_cast<int (*)(UniqueId*)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<int (*)(Comm*, int, UniqueId, int)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<int (*)(Comm)>(sym("ncclCommDestroy"));
        AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, Comm, hipStream_t)>(sym("ncclAllGather"));
        GetErrorString = reinterpret_cast<const char* (*)(int)>(sym("ncclGetErrorString"));
        if (!GetVersion || !GetUniqueId || !CommInitRank || !CommDestroy || !AllGather || !GetErrorString) {
            why = "librccl lacks an expected symbol";
            lib = nullptr;
        }
    }
};
Rccl& rccl() { static Rccl r; return r; }

thread_local std::string g_comm_error;
int cfail(int code, const std::string& msg) { g_comm_error = msg; return code; }
std::string nccl_err(const char* what, int rc) {
    return std::string(what) + " failed: " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
}

}  // namespace

struct dr_comm {
    Comm comm = nullptr;
    int n_ranks = 0, rank = 0, device = 0;
    // validity of a gather is decided collectively: [0] = this rank's status word, [1 .. n_ranks] = everybody's (device),
    // and a pinned host mirror the verdict is read from
    int32_t* d_status = nullptr;
    int32_t* h_status = nullptr;
};

extern "C" {

const char* dr_comm_last_error(void) { return g_comm_error.c_str(); }

int dr_comm_unique_id(char* id_out) {
    if (!id_out) return cfail(DR_EINVAL, "null argument");
    if (!rccl().lib) return cfail(DR_ESTATE, rccl().why);
    UniqueId id;
    int rc = rccl().GetUniqueId(&id);
    if (rc) return cfail(DR_EHIP, nccl_err("ncclGetUniqueId", rc));
    memcpy(id_out, id.internal, kUniqueIdBytes);
    return DR_OK;
}

int dr_comm_create(dr_comm** out, const char* id, int n_ranks, int rank, int device) {
    if (!out || !id) return cfail(DR_EINVAL, "null argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return cfail(DR_EINVAL, "bad rank / world size");
    if (!rccl().lib) return cfail(DR_ESTATE, rccl().why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return cfail(DR_EINVAL, "device " + std::to_string(device) + " out of range");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return cfail(DR_EHIP, "hipSetDevice failed");
    UniqueId uid;
    memcpy(uid.internal, id, kUniqueIdBytes);
    Comm c = nullptr;
    int rc = rccl().CommInitRank(&c, n_ranks, uid, rank);       // collective: every rank of the job calls it
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (rc) return cfail(DR_EHIP, nccl_err("ncclCommInitRank", rc));
    dr_comm* h = new dr_comm();
    h->comm = c; h->n_ranks = n_ranks; h->rank = rank; h->device = device;
    {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        void *d = nullptr, *hp = nullptr;
        const bool ok = hipMalloc(&d, (size_t)(n_ranks + 1) * sizeof(int32_t)) == hipSuccess &&
                        hipHostMalloc(&hp, (size_t)(n_ranks + 1) * sizeof(int32_t), hipHostMallocDefault) == hipSuccess;
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
        if (!ok) {
            if (d) (void)hipFree(d);
            (void)rccl().CommDestroy(c);
            delete h;
            return cfail(DR_ENOMEM, "allocating the gather status words failed");
        }
        h->d_status = (int32_t*)d; h->h_status = (int32_t*)hp;
    }
    *out = h;
    return DR_OK;
}

void dr_comm_destroy(dr_comm* c) {
    if (!c) return;
    if (c->comm && rccl().lib) (void)rccl().CommDestroy(c->comm);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->h_status) (void)hipHostFree(c->h_status);
    delete c;
}

int dr_comm_info(const dr_comm* c, int* n_ranks, int* rank, int* rccl_version) {
    if (c) {
        if (n_ranks) *n_ranks = c->n_ranks;
        if (rank) *rank = c->rank;
    } else if (n_ranks || rank) {
        return cfail(DR_EINVAL, "null communicator");
    }
    if (rccl_version) {
        if (!rccl().lib) return cfail(DR_ESTATE, rccl().why);
        int rc = rccl().GetVersion(rccl_version);
        if (rc) return cfail(DR_EHIP, nccl_err("ncclGetVersion", rc));
    }
    return DR_OK;
}

int dr_gather(dr_engine* e, dr_comm* comm, const float* d_shard, float* d_full, int B_local, int T, void* stream) {
    if (!comm || !d_shard || !d_full) return cfail(DR_EINVAL, "null argument");
    if (B_local < 0 || T <= 0) return cfail(DR_EINVAL, "bad shape");
    if (B_local == 0) return DR_OK;
    // The one piece of engine state that matters here: a roll produced by a fused launch that timed out is invalid
    // (include/diffroll_amd.h: dr_finish).  A time-out is a PER-RANK event and the gather is collective: a rank that
    // returned before the collective would leave its peers blocked in ncclAllGather for ever.  So this rank still takes
    // part - and tells everybody: behind the rolls every rank gathers one status word (0 valid, 1 invalid) on the same
    // communicator and stream, and EVERY rank answers DR_ETIMEOUT when any word is set.  (Until round 5 only the rank
    // that timed out knew; its peers got the garbage shard in d_full together with DR_OK.)  `e` may be NULL.
    const bool invalid = e && dr_pending_timeout(e, stream) != DR_OK;
    hipStream_t st = (hipStream_t)stream;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != comm->device && hipSetDevice(comm->device) != hipSuccess) return cfail(DR_EHIP, "hipSetDevice failed");
    auto back = [&]() { if (prev >= 0 && prev != comm->device) (void)hipSetDevice(prev); };
    int rc = rccl().AllGather(d_shard, d_full, (size_t)B_local * T * 88, kFloat32, comm->comm, st);
    if (rc) { back(); return cfail(DR_EHIP, nccl_err("ncclAllGather", rc)); }
    const int n = comm->n_ranks;
    comm->h_status[0] = invalid ? 1 : 0;
    hipError_t he = hipMemcpyAsync(comm->d_status, comm->h_status, sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) {
        rc = rccl().AllGather(comm->d_status, comm->d_status + 1, 1, kInt32, comm->comm, st);
        if (rc) { back(); return cfail(DR_EHIP, nccl_err("ncclAllGather (status words)", rc)); }
        he = hipMemcpyAsync(comm->h_status + 1, comm->d_status + 1, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    back();
    if (he != hipSuccess) return cfail(DR_EHIP, std::string("dr_gather: ") + hipGetErrorString(he));
    int bad = -1, n_bad = 0;
    for (int r = 0; r < n; ++r)
        if (comm->h_status[1 + r]) { if (bad < 0) bad = r; ++n_bad; }
    if (n_bad == 0) return DR_OK;
    return cfail(DR_ETIMEOUT, "dr_gather: the shard of rank " + std::to_string(bad) + (n_bad > 1 ? " (and " + std::to_string(n_bad - 1) + " more)" : "") +
                              " came out of a fused launch that timed out: d_full is invalid on every rank - that rank recomputes "
                              "(dr_finish heals it), then all ranks gather again" + (invalid ? std::string("; this rank: ") + dr_last_error(e) : std::string()));
}

}  // extern "C"
