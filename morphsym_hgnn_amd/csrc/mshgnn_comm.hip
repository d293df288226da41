// mshgnn_comm.hip -- the data-parallel path's ONE collective, enqueued on the step's own HIP stream through the C-ABI.
//
// replicated weights, windows sharded over one process per GPU, one mean all-reduce of the flat fp32 gradient per step
// (what Lightning-DDP does for the reference, gnnLightning.py:1396-1400).  torch.distributed puts that all-reduce on ITS OWN stream behind an
// event hand-over and a Python call (+16 us per step on a 1-rank group, profiles/r04g_*); here the caller's stream gets
// `ncclAllReduce(grad, grad, n, ncclFloat32, ncclAvg)` directly, so step + exchange is one stream-ordered (and graph-capturable) sequence.
//
// RCCL is bound at RUN time (dlopen): libmshgnn.so has no link-time dependency on it, single-GPU users never load it, and a process that
// already carries torch's bundled librccl.so gets THAT instance (RTLD_NOLOAD first) instead of a second copy of the library.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>

#include "mshgnn_device.hpp"      // set_err / g_err: the thread-local text behind mshgnn_last_error, shared by every translation unit

namespace {
// the slice of rccl.h this file needs (/opt/rocm/include/rccl/rccl.h: ncclUniqueId 128 bytes, ncclFloat32 = 7, ncclBfloat16 = 9, ncclSum = 0, ncclAvg = 4)
struct NcclId { char internal[128]; };
using NcclComm = void*;
using fn_get_id = int (*)(NcclId*);
using fn_init = int (*)(NcclComm*, int, NcclId, int);
using fn_destroy = int (*)(NcclComm);
using fn_allreduce = int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
using fn_errstr = const char* (*)(int);
using fn_group = int (*)();

struct Rccl {
    void* h = nullptr;
    fn_get_id get_id = nullptr; fn_init init = nullptr; fn_destroy destroy = nullptr; fn_allreduce allreduce = nullptr; fn_errstr errstr = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    std::string path;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl(const char* path) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return MSHGNN_OK;
    const char* cands[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* c : cands) if (c && *c && (h = dlopen(c, RTLD_NOW | RTLD_NOLOAD))) { g_rccl.path = c; break; }      // an instance the process already has
    if (!h) for (const char* c : cands) if (c && *c && (h = dlopen(c, RTLD_NOW | RTLD_LOCAL))) { g_rccl.path = c; break; }
    if (!h) return set_err(MSHGNN_EUNSUPPORTED, std::string("mshgnn_comm: librccl.so not found (") + (dlerror() ? dlerror() : "") + ")");
    Rccl r; r.h = h; r.path = g_rccl.path;
    r.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId"); r.init = (fn_init)dlsym(h, "ncclCommInitRank"); r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    r.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce"); r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    r.group_start = (fn_group)dlsym(h, "ncclGroupStart"); r.group_end = (fn_group)dlsym(h, "ncclGroupEnd");
    if (!r.get_id || !r.init || !r.destroy || !r.allreduce) return set_err(MSHGNN_EUNSUPPORTED, "mshgnn_comm: " + r.path + " lacks the NCCL entry points");
    g_rccl = r;
    return MSHGNN_OK;
}
int nccl_fail(int rc, const char* what) {
    return set_err(MSHGNN_EHIP, std::string("mshgnn_comm: ") + what + " failed: " + (g_rccl.errstr ? g_rccl.errstr(rc) : "?") + " (" + std::to_string(rc) + ")");
}
}  // namespace

struct mshgnn_comm { NcclComm comm; int nranks, rank; };

extern "C" int mshgnn_comm_unique_id(const char* rccl_path, void* id_out) {
    if (!id_out) return set_err(MSHGNN_EINVAL, "mshgnn_comm_unique_id: null id_out");
    if (int rc = load_rccl(rccl_path)) return rc;
    NcclId id;
    if (int rc = g_rccl.get_id(&id)) return nccl_fail(rc, "ncclGetUniqueId");
    std::memcpy(id_out, &id, sizeof(id));
    return MSHGNN_OK;
}

extern "C" int mshgnn_comm_create(const char* rccl_path, const void* id, int nranks, int rank, mshgnn_comm** out) {
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return set_err(MSHGNN_EINVAL, "mshgnn_comm_create: bad arguments");
    if (int rc = load_rccl(rccl_path)) return rc;
    NcclId nid; std::memcpy(&nid, id, sizeof(nid));
    NcclComm c = nullptr;
    if (int rc = g_rccl.init(&c, nranks, nid, rank)) return nccl_fail(rc, "ncclCommInitRank");      // (the caller has made its device current)
    *out = new mshgnn_comm{c, nranks, rank};
    return MSHGNN_OK;
}

extern "C" void mshgnn_comm_destroy(mshgnn_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.destroy) g_rccl.destroy(c->comm);
    delete c;
}

extern "C" int mshgnn_comm_allreduce_mean(mshgnn_comm* c, float* buf, int64_t n, void* stream) {
    if (!c || !buf || n < 0) return set_err(MSHGNN_EINVAL, "mshgnn_comm_allreduce_mean: bad arguments");
    if (n == 0) return MSHGNN_OK;
    if (int rc = g_rccl.allreduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclAvg*/ 4, c->comm, (hipStream_t)stream)) return nccl_fail(rc, "ncclAllReduce");
    return MSHGNN_OK;
}

extern "C" int mshgnn_comm_allreduce_sum(mshgnn_comm* c, float* buf, int64_t n, void* stream) {
    if (!c || !buf || n < 0) return set_err(MSHGNN_EINVAL, "mshgnn_comm_allreduce_sum: bad arguments");
    if (n == 0) return MSHGNN_OK;
    if (int rc = g_rccl.allreduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, (hipStream_t)stream)) return nccl_fail(rc, "ncclAllReduce");
    return MSHGNN_OK;
}
