// rccl_comm.hpp -- RCCL behind the C-ABI (north_star: "an RCCL gather of the resulting SE(3)
// poses over xGMI", "a node-wide 6x6 reduction").
//
// One communicator per process (or per host thread) and GPU.  librccl is resolved at the first slam3d_comm_* call:
// a copy the process already holds (torch ships one with the same SONAME) is reused, otherwise librccl.so.1 is
// loaded -- the shared library itself has no link-time dependency on RCCL, so single-GPU hosts need none.
// Every collective is enqueued on a HIP stream in program order with the kernels that produce / consume its buffer.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/slam3d_icp.h"

namespace s3d {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void *lib = nullptr;
    std::string err;
    bool ok = false;
};

// resolved once per process, thread-safe: the C++ front end creates its communicators from one host thread per GPU
// (function-local static initialised by a lambda: the language guarantees a single, completed initialisation)
inline RcclApi &rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        const char *names[] = { "librccl.so.1", "librccl.so" };
        for (const char *n : names) {                                  // a copy that is already mapped (e.g. torch's)
            a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) {
            const char *paths[] = { "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" };
            for (const char *n : paths) {
                a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (a.lib) break;
            }
        }
        if (!a.lib) {
            const char *why = dlerror();                               // read ONCE: the call clears the pending error
            a.err = std::string("librccl not loadable: ") + (why ? why : "?");
            return a;
        }
        auto sym = [&](const char *name) { void *p = dlsym(a.lib, name); if (!p && a.err.empty()) a.err = std::string("missing symbol ") + name; return p; };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(a.lib, "ncclCommAbort"));      // optional
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.ok = a.err.empty();
        return a;
    }();
    return api;
}

} // namespace s3d

static_assert(sizeof(ncclUniqueId) == SLAM3D_COMM_ID_BYTES, "slam3d_comm id size");
static_assert(sizeof(slam3d_pose_record) == 160, "pose record is 160 bytes (SURVEY.md 8(e))");

struct slam3d_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;                  // the gather's own stream: overlaps the caller's kernels
    // pose gather: two pipelined slots of device + pinned staging
    static constexpr int NSLOT = 2;
    slam3d_pose_record *d_send[NSLOT] = { nullptr, nullptr }, *d_recv[NSLOT] = { nullptr, nullptr };
    slam3d_pose_record *h_send[NSLOT] = { nullptr, nullptr }, *h_recv[NSLOT] = { nullptr, nullptr };
    hipEvent_t done[NSLOT] = { nullptr, nullptr };
    int cap = 0;                                    // records per rank the staging holds
    int n_pending = 0, head = 0, pending_n[NSLOT] = { 0, 0 };
    std::string err;
};

#define S3D_NCCLCHK(c, call)                                                                                       \
    do {                                                                                                          \
        ncclResult_t r__ = (call);                                                                                \
        if (r__ != ncclSuccess) {                                                                                 \
            char buf__[384];                                                                                      \
            snprintf(buf__, sizeof buf__, "%s failed: %s", #call, s3d::rccl().GetErrorString ? s3d::rccl().GetErrorString(r__) : "?"); \
            if (c) (c)->err = buf__;                                                                              \
            return SLAM3D_E_COMM;                                                                                 \
        }                                                                                                         \
    } while (0)
#define S3D_COMM_HIPCHK(c, call)                                                                                   \
    do {                                                                                                          \
        hipError_t e__ = (call);                                                                                  \
        if (e__ != hipSuccess) {                                                                                  \
            char buf__[384];                                                                                      \
            snprintf(buf__, sizeof buf__, "%s failed: %s", #call, hipGetErrorString(e__));                        \
            if (c) (c)->err = buf__;                                                                              \
            return SLAM3D_E_HIP;                                                                                  \
        }                                                                                                         \
    } while (0)

extern "C" void slam3d_shard_range(int32_t n, int32_t world, int32_t rank, int32_t *begin, int32_t *end)
{
    if (world <= 0 || rank < 0 || rank >= world || n < 0) { if (begin) *begin = 0; if (end) *end = 0; return; }
    const int base = n / world, rem = n % world;
    const int b = rank * base + (rank < rem ? rank : rem);
    if (begin) *begin = b;
    if (end) *end = b + base + (rank < rem ? 1 : 0);
}

extern "C" int slam3d_comm_get_unique_id(void *id)
{
    if (!id) return SLAM3D_E_INVALID;
    s3d::RcclApi &api = s3d::rccl();
    if (!api.ok) return SLAM3D_E_COMM;
    ncclUniqueId u;
    if (api.GetUniqueId(&u) != ncclSuccess) return SLAM3D_E_COMM;
    memcpy(id, &u, sizeof u);
    return SLAM3D_OK;
}

extern "C" void slam3d_comm_destroy(slam3d_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < slam3d_comm::NSLOT; ++k) {
        if (c->d_send[k]) (void)hipFree(c->d_send[k]);
        if (c->d_recv[k]) (void)hipFree(c->d_recv[k]);
        if (c->h_send[k]) (void)hipHostFree(c->h_send[k]);
        if (c->h_recv[k]) (void)hipHostFree(c->h_recv[k]);
        if (c->done[k]) (void)hipEventDestroy(c->done[k]);
    }
    if (c->comm && s3d::rccl().ok) (void)s3d::rccl().CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int slam3d_comm_init(const void *id, int32_t rank, int32_t world, int32_t device, slam3d_comm **out)
{
    if (!id || !out || world <= 0 || rank < 0 || rank >= world) return SLAM3D_E_INVALID;
    *out = nullptr;
    s3d::RcclApi &api = s3d::rccl();
    if (!api.ok) return SLAM3D_E_COMM;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SLAM3D_E_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return SLAM3D_E_NODEVICE;
    auto *c = new slam3d_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    if (api.CommInitRank(&c->comm, world, u, rank) != ncclSuccess) { delete c; return SLAM3D_E_COMM; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { slam3d_comm_destroy(c); return SLAM3D_E_HIP; }
    for (int k = 0; k < slam3d_comm::NSLOT; ++k)
        if (hipEventCreateWithFlags(&c->done[k], hipEventDisableTiming) != hipSuccess) { slam3d_comm_destroy(c); return SLAM3D_E_HIP; }
    *out = c;
    return SLAM3D_OK;
}

extern "C" int slam3d_comm_rank(const slam3d_comm *c) { return c ? c->rank : 0; }
extern "C" int slam3d_comm_world(const slam3d_comm *c) { return c ? c->world : 1; }
extern "C" const char *slam3d_comm_last_error(const slam3d_comm *c)
{
    if (c) return c->err.c_str();
    return s3d::rccl().err.c_str();
}

extern "C" void slam3d_pose_record_from_result(const slam3d_icp_result *r, slam3d_pose_record *rec)
{
    if (!r || !rec) return;
    memcpy(rec->T, r->T, sizeof rec->T);
    rec->norm = r->norm; rec->inliers = r->inliers; rec->status = r->status; rec->rmse = r->rmse; rec->_pad = 0.0;
}

static int comm_reserve(slam3d_comm *c, int n_local)
{
    if (n_local <= c->cap) return SLAM3D_OK;
    if (c->n_pending) return SLAM3D_E_STATE;                       // a gather in flight still uses the staging
    for (int k = 0; k < slam3d_comm::NSLOT; ++k) {
        if (c->d_send[k]) (void)hipFree(c->d_send[k]);
        if (c->d_recv[k]) (void)hipFree(c->d_recv[k]);
        if (c->h_send[k]) (void)hipHostFree(c->h_send[k]);
        if (c->h_recv[k]) (void)hipHostFree(c->h_recv[k]);
        c->d_send[k] = c->d_recv[k] = c->h_send[k] = c->h_recv[k] = nullptr;
        const size_t one = sizeof(slam3d_pose_record) * (size_t)n_local, all = one * (size_t)c->world;
        S3D_COMM_HIPCHK(c, hipMalloc((void **)&c->d_send[k], one));
        S3D_COMM_HIPCHK(c, hipMalloc((void **)&c->d_recv[k], all));
        S3D_COMM_HIPCHK(c, hipHostMalloc((void **)&c->h_send[k], one, hipHostMallocDefault));
        S3D_COMM_HIPCHK(c, hipHostMalloc((void **)&c->h_recv[k], all, hipHostMallocDefault));
    }
    c->cap = n_local;
    return SLAM3D_OK;
}

extern "C" int slam3d_pose_gather_submit(slam3d_comm *c, const slam3d_pose_record *local, int32_t n_local)
{
    if (!c || !local || n_local <= 0) return SLAM3D_E_INVALID;
    if (c->n_pending >= slam3d_comm::NSLOT) return SLAM3D_E_STATE;
    if (!c->comm) { c->err = "communicator was aborted after a local failure (slam3d_icp_dense_run); create a new one"; return SLAM3D_E_COMM; }
    S3D_COMM_HIPCHK(c, hipSetDevice(c->device));
    const int rc = comm_reserve(c, n_local);
    if (rc) return rc;
    const int k = (c->head + c->n_pending) % slam3d_comm::NSLOT;
    const size_t one = sizeof(slam3d_pose_record) * (size_t)n_local;
    memcpy(c->h_send[k], local, one);
    S3D_COMM_HIPCHK(c, hipMemcpyAsync(c->d_send[k], c->h_send[k], one, hipMemcpyHostToDevice, c->stream));
    S3D_NCCLCHK(c, s3d::rccl().AllGather(c->d_send[k], c->d_recv[k], one, ncclUint8, c->comm, c->stream));
    S3D_COMM_HIPCHK(c, hipMemcpyAsync(c->h_recv[k], c->d_recv[k], one * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    S3D_COMM_HIPCHK(c, hipEventRecord(c->done[k], c->stream));
    c->pending_n[k] = n_local;
    c->n_pending += 1;
    return SLAM3D_OK;
}

extern "C" int slam3d_pose_gather_collect(slam3d_comm *c, slam3d_pose_record *all)
{
    if (!c || !all) return SLAM3D_E_INVALID;
    if (c->n_pending <= 0) return SLAM3D_E_STATE;
    S3D_COMM_HIPCHK(c, hipSetDevice(c->device));
    const int k = c->head;
    S3D_COMM_HIPCHK(c, hipEventSynchronize(c->done[k]));
    memcpy(all, c->h_recv[k], sizeof(slam3d_pose_record) * (size_t)c->pending_n[k] * (size_t)c->world);
    c->head = (c->head + 1) % slam3d_comm::NSLOT;
    c->n_pending -= 1;
    return SLAM3D_OK;
}

extern "C" int slam3d_pose_gather(slam3d_comm *c, const slam3d_pose_record *local, int32_t n_local, slam3d_pose_record *all)
{
    const int rc = slam3d_pose_gather_submit(c, local, n_local);
    return rc ? rc : slam3d_pose_gather_collect(c, all);
}
