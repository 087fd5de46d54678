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
#include <unistd.h>
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


// ---- cross-rank persistent solve: IPC mappings ---------------------------------------------------------------------------------------
// Every rank exports (a) its small mailbox region and (b) its band arena (the neighbours write the records of their cut-side rows into its halo rows);
// the handles and a few numbers travel through ONE all-reduce in which every rank fills its own slice (bytes as doubles, zeros elsewhere), so the
// exchange works over RCCL and over a caller-supplied transport alike.  All ranks agree on the outcome with a second all-reduce: either every rank
// runs the cross-rank persistent solve or every rank stays on the per-pass kernels.
void xr_release(psgsdf_ctx* c) {
    for (void* p : c->xr_opened) hipIpcCloseMemHandle(p);
    c->xr_opened.clear(); c->xr_peer.clear(); c->band_peer[0] = c->band_peer[1] = nullptr; c->xr_ready = false; c->xr_args = XrArgs{};
}
int xr_setup(psgsdf_ctx* c, const std::vector<double>& info) {
    xr_release(c);
    const int R = c->n_ranks, me = c->rank;
    if (R <= 1 || !c->comm || !c->xr_enable || !c->pcg_persist || !c->pcg_fuse_asm || R > kXrMaxRanks) return 0;
    if (!c->xr) { HIPCHK(c, hipMalloc(&c->xr, sizeof(double) * kXrDoubles)); HIPCHK(c, hipMemsetAsync(c->xr, 0, sizeof(double) * kXrDoubles, c->stream)); }
    constexpr int kSlice = 64 + 64 + 8;      // two IPC handles (64 bytes each, one double per byte) + {rec[0] offset, rec[1] offset, pid, ok}
    std::vector<double> buf((size_t)R * kSlice, 0.0);
    hipIpcMemHandle_t hx{}, hb{};
    int Gs, Rs;
    bool ok = cgf_solve_shape(c, &Gs, &Rs, true)      // (this slab fits the persistent kernel at all)
        && hipIpcGetMemHandle(&hx, c->xr) == hipSuccess && hipIpcGetMemHandle(&hb, c->band_mem) == hipSuccess;
    (void)hipGetLastError();
    double* mine = buf.data() + (size_t)me * kSlice;
    for (int i = 0; i < 64; ++i) { mine[i] = (double)((unsigned char*)&hx)[i]; mine[64 + i] = (double)((unsigned char*)&hb)[i]; }
    mine[128] = (double)((char*)c->band.rec[0] - (char*)c->band_mem); mine[129] = (double)((char*)c->band.rec[1] - (char*)c->band_mem);
    mine[130] = (double)getpid(); mine[131] = ok ? 1.0 : 0.0;
    double* d_buf = nullptr;
    HIPCHK(c, hipMalloc(&d_buf, sizeof(double) * buf.size()));
    int rc = 0;
    if (hipMemcpyAsync(d_buf, buf.data(), sizeof(double) * buf.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "cross-rank set-up: upload");
    if (!rc) rc = comm_allreduce(c, d_buf, (int)buf.size());
    if (!rc && (hipMemcpyAsync(buf.data(), d_buf, sizeof(double) * buf.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) rc = fail(c, PSGSDF_ERR_DEVICE, "cross-rank set-up: download");
    if (rc) { hipFree(d_buf); return rc; }
    // map every rank's region, and the two neighbours' band arenas
    c->xr_peer.assign(R, nullptr); c->xr_peer[me] = c->xr;
    auto open = [&](const double* bytes) -> void* {
        hipIpcMemHandle_t h; for (int i = 0; i < 64; ++i) ((unsigned char*)&h)[i] = (unsigned char)bytes[i];
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        c->xr_opened.push_back(p);
        return p;
    };
    for (int r = 0; r < R && ok; ++r) {
        const double* sl = buf.data() + (size_t)r * kSlice;
        if (sl[131] != 1.0) { ok = false; break; }
        if (r == me) continue;
        if (sl[130] == (double)getpid()) { ok = false; break; }      // two ranks in ONE process: no IPC between them (the per-pass path serves that case)
        c->xr_peer[r] = (double*)open(sl);
        if (!c->xr_peer[r]) { ok = false; break; }
        if (r == me - 1 || r == me + 1) { c->band_peer[r == me - 1 ? 0 : 1] = open(sl + 64); if (!c->band_peer[r == me - 1 ? 0 : 1]) { ok = false; break; } }
    }
    // the launch shapes of the neighbours (a function of their own row count and the CU count, the same on every rank): how many of their
    // workgroups own rows this slab holds as halo = how many tags its cut-side workgroups wait for
    auto shape_of = [&](int own, int* G, int* per) {
        const int cap = std::min(c->num_cu, kSolveMaxBlocksHost);
        int g = std::min(cap, (own + kSolveThreadsHost - 1) / kSolveThreadsHost); g = std::max(8, g / 8 * 8); if (g > cap) g = cap;
        *G = g; *per = ((own + g - 1) / g + 63) / 64 * 64;
    };
    XrArgs x{}; x.rank = me; x.n_ranks = R;
    x.need_lo = c->need[0]; x.need_hi = c->need[1]; x.give_lo = c->give[0]; x.give_hi = c->give[1];
    if (ok) {
        for (int r = 0; r < R; ++r) x.region[r] = c->xr_peer[r];
        if (me > 0) {
            const double* sl = buf.data() + (size_t)(me - 1) * kSlice;
            const int own_p = (int)info[3 * (me - 1) + 2], need_lo_p = (int)info[3 * (me - 1)];
            // its upper halo rows start at its row1 = need_lo_p + own_p (local band order: lower halo, own rows, upper halo)
            for (int q = 0; q < 2; ++q) x.lo_rec[q] = (float4*)((char*)c->band_peer[0] + (size_t)sl[128 + q]) + (need_lo_p + own_p);
            int G, per; shape_of(own_p, &G, &per);
            const int first = std::max(0, own_p - c->need[0]) / per, last = (own_p - 1) / per;      // its workgroups that own its last need_lo(me) rows
            x.wait_lo = c->need[0] > 0 ? last - first + 1 : 0;
        }
        if (me < R - 1) {
            const double* sl = buf.data() + (size_t)(me + 1) * kSlice;
            const int own_p = (int)info[3 * (me + 1) + 2];
            for (int q = 0; q < 2; ++q) x.hi_rec[q] = (float4*)((char*)c->band_peer[1] + (size_t)sl[128 + q]);      // its lower halo rows are its first rows
            int G, per; shape_of(own_p, &G, &per);
            x.wait_hi = c->need[1] > 0 ? std::min(own_p, c->need[1]) > 0 ? (std::min(own_p, c->need[1]) - 1) / per + 1 : 0 : 0;
        }
        if (x.wait_lo > kXrPeerTags || x.wait_hi > kXrPeerTags) ok = false;
        // (this rank's own cut-side workgroups must fit the neighbours' tag slots too: the same formula on their side)
        int G, per; shape_of(c->row1 - c->row0, &G, &per);
        if (c->give[0] > 0 && (c->give[0] - 1) / per + 1 > kXrPeerTags) ok = false;
        if (c->give[1] > 0 && (c->row1 - c->row0 - 1) / per - std::max(0, c->row1 - c->row0 - c->give[1]) / per + 1 > kXrPeerTags) ok = false;
    }
    // agreement: every rank or none
    double flag = ok ? 1.0 : 0.0;
    rc = 0;
    if (hipMemcpyAsync(d_buf, &flag, sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "cross-rank set-up: flag");
    if (!rc) rc = comm_allreduce(c, d_buf, 1);
    if (!rc && (hipMemcpyAsync(&flag, d_buf, sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) rc = fail(c, PSGSDF_ERR_DEVICE, "cross-rank set-up: flag");
    hipFree(d_buf);
    if (rc) return rc;
    if (flag != (double)R) { xr_release(c); return 0; }
    c->xr_args = x; c->xr_ready = true;
    return 0;
}

}  // namespace psge
