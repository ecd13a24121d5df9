// collective.hip -- the ONE exchange step of the hot path for hosts that are not PyTorch: an in-place float32 sum
// all-reduce of the shared gradients (d_volume, d_tf) over RCCL / xGMI. Thin shim: RCCL is dlopen()ed on first use, so
// the library has no link-time dependency on it (a process that already holds an RCCL -- PyTorch's -- gets that one),
// and a single-GPU host never loads it. The Python package does not call these: it uses torch.distributed, whose "nccl"
// backend is the same RCCL (differender_amd/distributed.py). No counterpart in the reference (single device).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/differender_hip.h"

namespace {

// the subset of rccl.h this shim needs (stable NCCL ABI)
typedef int ncclResult_t;
typedef void *ncclComm_t;
enum { ncclFloat32 = 7, ncclSum = 0 };
struct ncclUniqueId { char internal[128]; };

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.handle) break;
        }
        if (!x.handle) return x;
#define DR_SYM(field, sym) x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.handle, sym))
        DR_SYM(GetUniqueId, "ncclGetUniqueId"); DR_SYM(CommInitRank, "ncclCommInitRank"); DR_SYM(CommInitAll, "ncclCommInitAll");
        DR_SYM(AllReduce, "ncclAllReduce"); DR_SYM(CommDestroy, "ncclCommDestroy");
        DR_SYM(GroupStart, "ncclGroupStart"); DR_SYM(GroupEnd, "ncclGroupEnd");
#undef DR_SYM
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommInitAll && x.AllReduce && x.CommDestroy && x.GroupStart && x.GroupEnd;
        return x;
    }();
    return r;
}

int rc_of(ncclResult_t r) { return r == 0 ? 0 : DR_ECOLLECTIVE; }

}  // namespace

extern "C" {

int dr_comm_unique_id(void *id128) {
    if (!id128) return DR_EINVAL;
    if (!rccl().ok) return DR_EUNSUPPORTED;
    return rc_of(rccl().GetUniqueId(static_cast<ncclUniqueId *>(id128)));
}

int dr_comm_init_rank(void **comm, int n_ranks, const void *id128, int rank) {
    if (!comm || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return DR_EINVAL;
    if (!rccl().ok) return DR_EUNSUPPORTED;
    ncclUniqueId id = *static_cast<const ncclUniqueId *>(id128);
    return rc_of(rccl().CommInitRank(reinterpret_cast<ncclComm_t *>(comm), n_ranks, id, rank));
}

int dr_comm_init_all(void **comms, int n_devices, const int *devices) {
    if (!comms || n_devices < 1) return DR_EINVAL;
    if (!rccl().ok) return DR_EUNSUPPORTED;
    return rc_of(rccl().CommInitAll(reinterpret_cast<ncclComm_t *>(comms), n_devices, devices));
}

int dr_allreduce_f32(void *comm, float *buf, size_t n, void *stream) {
    if (!comm || (!buf && n)) return DR_EINVAL;
    if (n == 0) return 0;
    if (!rccl().ok) return DR_EUNSUPPORTED;
    return rc_of(rccl().AllReduce(buf, buf, n, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
}

int dr_allreduce_gradients_f32(void *comm, float *d_vol, size_t n_vol, float *d_tf, size_t n_tf, void *stream) {
    if (!comm) return DR_EINVAL;
    if (!rccl().ok) return DR_EUNSUPPORTED;
    // one group: the 4 KiB d_tf message rides with the d_volume one instead of paying its own launch latency
    int rc = rc_of(rccl().GroupStart());
    if (rc) return rc;
    if (d_vol && n_vol) rc = rc_of(rccl().AllReduce(d_vol, d_vol, n_vol, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
    if (!rc && d_tf && n_tf) rc = rc_of(rccl().AllReduce(d_tf, d_tf, n_tf, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
    const int rc2 = rc_of(rccl().GroupEnd());
    return rc ? rc : rc2;
}

int dr_comm_destroy(void *comm) {
    if (!comm) return DR_EINVAL;
    if (!rccl().ok) return DR_EUNSUPPORTED;
    return rc_of(rccl().CommDestroy(static_cast<ncclComm_t>(comm)));
}

}  // extern "C"
