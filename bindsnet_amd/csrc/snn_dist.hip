// snn_dist.hip -- the multi-GPU entry points of the C ABI: one process per GPU, collectives
// over RCCL (xGMI).  A caller that is not Python / torch.distributed -- the boundary this header exists for -- gets:
//
//   snn_dist_unique_id   rank 0 creates the 128-byte rendezvous id and hands it to the other ranks by its own means
//   snn_dist_init        every rank joins (ncclCommInitRank on the CURRENT HIP device)
//   snn_dist_allreduce_dw      the north-star schedule: sum the per-input weight / threshold DELTAS over ranks, in
//                              place, one flat f32 buffer (Nin*N + N floats: 1.25 MB at cfg2) -- ring time ~ 2 (G-1)/G *
//                              bytes / 153 GB/s per xGMI link, i.e. tens of microseconds per input
//   snn_dist_allgather_step    the exact per-timestep mode: gather every rank's packed spike / trace factors of one
//                              step ([bytes_per_rank] each, rank order) so that each GPU applies the full-batch update
//   snn_dist_destroy
//
// RCCL is opened at first use (dlopen of librccl.so.1 -- the copy PyTorch already loaded, or /opt/rocm/lib's), so
// libsnnhip.so itself has no link-time dependency on it and single-GPU users never touch it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/snnhip.h"
#include "snn_common.hpp"

namespace {

typedef void *ncclComm_t_;
typedef struct { char internal[128]; } ncclUniqueId_;
enum { kNcclSum = 0, kNcclUint8 = 1, kNcclFloat32 = 7 };          // rccl.h: ncclRedOp_t / ncclDataType_t

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId_ *) = nullptr;
    int (*CommInitRank)(ncclComm_t_ *, int, ncclUniqueId_, int) = nullptr;
    int (*CommDestroy)(ncclComm_t_) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t_, hipStream_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[3] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (r.handle) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
            r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
            r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
            r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather;
        }
    }
    return r;
}

}  // namespace

struct snn_dist { ncclComm_t_ comm; int rank, world; };

extern "C" int snn_dist_unique_id(void *h_id128) {
    if (!h_id128) return SNN_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.ok) return SNN_ERR_UNSUPPORTED;
    ncclUniqueId_ id;
    if (r.GetUniqueId(&id) != 0) return SNN_ERR_LAUNCH;
    memcpy(h_id128, id.internal, 128);
    return SNN_OK;
}

extern "C" int snn_dist_init(int rank, int world, const void *h_id128, snn_dist **out) {
    if (!h_id128 || !out || world < 1 || rank < 0 || rank >= world) return SNN_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.ok) return SNN_ERR_UNSUPPORTED;
    ncclUniqueId_ id;
    memcpy(id.internal, h_id128, 128);
    snn_dist *d = new snn_dist{nullptr, rank, world};
    if (r.CommInitRank(&d->comm, world, id, rank) != 0) { delete d; return SNN_ERR_LAUNCH; }
    *out = d;
    return SNN_OK;
}

extern "C" int snn_dist_world(const snn_dist *d, int *h_rank, int *h_world) {
    if (!d) return SNN_ERR_INVALID;
    if (h_rank) *h_rank = d->rank;
    if (h_world) *h_world = d->world;
    return SNN_OK;
}

extern "C" int snn_dist_allreduce_dw(snn_dist *d, float *buf, long long count, snn_stream_t stream) {
    if (!d || !buf || count <= 0) return SNN_ERR_INVALID;
    return rccl().AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, d->comm, (hipStream_t)stream) == 0 ? SNN_OK : SNN_ERR_LAUNCH;
}

extern "C" int snn_dist_allgather_step(snn_dist *d, const void *send, void *recv, long long bytes_per_rank, snn_stream_t stream) {
    if (!d || !send || !recv || bytes_per_rank <= 0) return SNN_ERR_INVALID;
    return rccl().AllGather(send, recv, (size_t)bytes_per_rank, kNcclUint8, d->comm, (hipStream_t)stream) == 0 ? SNN_OK : SNN_ERR_LAUNCH;
}

extern "C" int snn_dist_destroy(snn_dist *d) {
    if (!d) return SNN_ERR_INVALID;
    const int rc = rccl().CommDestroy(d->comm);
    delete d;
    return rc == 0 ? SNN_OK : SNN_ERR_LAUNCH;
}
