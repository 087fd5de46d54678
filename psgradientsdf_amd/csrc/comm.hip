// comm.hip -- the exchange layer of the z-slab multi-GPU path: one communicator per context, every
// collective enqueued on the context's HIP stream by the C++ host itself (loop.hip / engine.hip), never by a host program.
//
//   transport "rccl" : RCCL over xGMI.  librccl is bound at run time (dlopen) the first time a communicator is asked for, so a
//                      single-GPU process never loads it; inside a process that already maps RCCL (e.g. PyTorch) the same copy is reused.
//   transport "ext"  : the caller supplies the two primitives (psgsdf_comm_ops): a seam for hosts that already own a
//                      communicator (MPI, a test harness).  tests/ use it to run two ranks on ONE device, which RCCL refuses.
//
// What moves (all latency-bound, <= 1 MiB): all-reduce of the per-frame light / pose rows, of the 7 sums of a PCG pass, of the
// iteration's folded scalars; halo rows of `blk`, the PCG records and `dist` with the two z-neighbours (and, once per band, of the static
// stencil-direction bits; before a 2x refinement, of albedo and gradient).
#include "engine_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enum values only: every function is resolved with dlsym

namespace psge {

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) { api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.handle) break; }
    if (!api.handle) { api.error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"); return &api; }
    auto sym = [&](const char* n) { void* p = dlsym(api.handle, n); if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + n; return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!api.error.empty()) { dlclose(api.handle); api.handle = nullptr; }
    return &api;
}

}  // namespace

struct Comm {
    int rank = 0, n = 1;
    ncclComm_t nccl = nullptr;         // transport "rccl"
    psgsdf_comm_ops ext{};             // transport "ext" (nccl == nullptr)
    bool is_ext = false;
};

#define NCCLCHK(c, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return fail(c, PSGSDF_ERR_COMM, "%s: %s", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "rccl error"); } while (0)

int comm_unique_id(uint8_t id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "psgsdf_comm_unique_id hands out 128 bytes");
    RcclApi* r = rccl();
    if (!r->handle) return PSGSDF_ERR_COMM;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return PSGSDF_ERR_COMM;
    memcpy(id, &u, 128);
    return PSGSDF_OK;
}

void comm_destroy(psgsdf_ctx* c) {
    if (!c->comm) return;
    if (c->comm->nccl && rccl()->handle) { hipStreamSynchronize(c->stream); rccl()->CommDestroy(c->comm->nccl); }
    delete c->comm; c->comm = nullptr;
}

int comm_create_rccl(psgsdf_ctx* c, const uint8_t id[128], int rank, int n) {
    RcclApi* r = rccl();
    if (!r->handle) return fail(c, PSGSDF_ERR_COMM, "%s", r->error.c_str());
    comm_destroy(c);
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId u; memcpy(&u, id, 128);
    Comm* cm = new Comm(); cm->rank = rank; cm->n = n;
    ncclResult_t rc = r->CommInitRank(&cm->nccl, n, u, rank);
    if (rc != ncclSuccess) { delete cm; return fail(c, PSGSDF_ERR_COMM, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, n, c->device, r->GetErrorString(rc)); }
    c->comm = cm;
    return PSGSDF_OK;
}
int comm_create_ext(psgsdf_ctx* c, const psgsdf_comm_ops* ops, int rank, int n) {
    if (!ops || !ops->allreduce_f64 || !ops->sendrecv) return fail(c, PSGSDF_ERR_ARG, "comm_init_ext: both primitives are required");
    comm_destroy(c);
    Comm* cm = new Comm(); cm->rank = rank; cm->n = n; cm->ext = *ops; cm->is_ext = true;
    c->comm = cm;
    return PSGSDF_OK;
}

static int need_comm(psgsdf_ctx* c) {
    if (c->comm) return 0;
    return fail(c, PSGSDF_ERR_COMM, "rank %d of %d has no communicator: call psgsdf_comm_init with the id from psgsdf_comm_unique_id (or psgsdf_comm_init_ext)", c->rank, c->n_ranks);
}

// in-place sum over the ranks of n doubles at device pointer buf, ordered on the context's stream
int comm_allreduce(psgsdf_ctx* c, double* buf, int n) {
    if (!slab_mode(c) || n <= 0) return 0;
    int rc = need_comm(c); if (rc) return rc;
    c->n_collectives++;
    if (c->comm->is_ext) { if (c->comm->ext.allreduce_f64(c->comm->ext.user, buf, n, c->stream)) return fail(c, PSGSDF_ERR_COMM, "ext allreduce failed"); return 0; }
    NCCLCHK(c, rccl()->AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, c->comm->nccl, c->stream));
    return 0;
}

// halo rows of `planes` planes of `width` 4-byte words per row (plane stride Spad rows) with the two z-neighbours:
// what this slab needs of them ([row0-need_lo,row0) and [row1,row1+need_hi)) against what they need of it (give_lo / give_hi rows)
int comm_halo(psgsdf_ctx* c, void* base, int planes, int width) {
    if (c->n_ranks == 1 || !c->halo_active) return 0;
    int rc = need_comm(c); if (rc) return rc;
    const size_t rowb = 4 * (size_t)width, planeb = rowb * (size_t)c->band.Spad;
    std::vector<psgsdf_comm_xfer> sends, recvs;
    for (int p = 0; p < planes; ++p) {
        char* pl = (char*)base + planeb * p;
        if (c->give[0]) sends.push_back({pl + rowb * c->row0, rowb * c->give[0], c->rank - 1});
        if (c->need[0]) recvs.push_back({pl + rowb * (c->row0 - c->need[0]), rowb * c->need[0], c->rank - 1});
        if (c->give[1]) sends.push_back({pl + rowb * (c->row1 - c->give[1]), rowb * c->give[1], c->rank + 1});
        if (c->need[1]) recvs.push_back({pl + rowb * c->row1, rowb * c->need[1], c->rank + 1});
    }
    if (sends.empty() && recvs.empty()) return 0;
    c->n_collectives++;
    if (c->comm->is_ext) {
        if (c->comm->ext.sendrecv(c->comm->ext.user, sends.data(), (int)sends.size(), recvs.data(), (int)recvs.size(), c->stream)) return fail(c, PSGSDF_ERR_COMM, "ext sendrecv failed");
        return 0;
    }
    RcclApi* r = rccl();
    NCCLCHK(c, r->GroupStart());
    // (an error inside the bracket must not leave the group open on this thread: later collectives would be queued into it or hang)
    ncclResult_t first = ncclSuccess;
    for (auto& x : sends) { if (first != ncclSuccess) break; first = r->Send(x.ptr_dev, x.bytes, ncclChar, x.peer, c->comm->nccl, c->stream); }
    for (auto& x : recvs) { if (first != ncclSuccess) break; first = r->Recv(x.ptr_dev, x.bytes, ncclChar, x.peer, c->comm->nccl, c->stream); }
    const ncclResult_t end = r->GroupEnd();
    if (first != ncclSuccess) return fail(c, PSGSDF_ERR_COMM, "halo exchange (ncclSend / ncclRecv): %s", r->GetErrorString ? r->GetErrorString(first) : "rccl error");
    NCCLCHK(c, end);
    return 0;
}

}  // namespace psge
