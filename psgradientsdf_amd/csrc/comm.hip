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

#include <atomic>
#include <chrono>
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
// Round 4: where the ranks' mailbox regions and halo stagings are mapped (xr_setup) the exchange is two small kernels on the context's stream instead
// of an RCCL send / recv group: k_halo_push stores the cut-side rows into the neighbours' stagings (system-scope write-through stores, drained, then a
// flag carrying the exchange's number in the neighbour's region), k_halo_pull waits for the two neighbours' flags and copies the staged rows into
// the halo rows of the array.  Two stagings alternate, so a push never overwrites rows the neighbour has yet to pull (its pull of exchange n sits
// in front of its push of n + 1, which this rank's pull of n + 1 -- in front of its push of n + 2 -- waits for).
namespace {
struct HaloSide { const unsigned* src; unsigned* dst; int rows; double* flag; };
struct HaloArgs { HaloSide s[2]; int planes, width; long long plane_words; double tag; double* abort_flag; double* host_late; int spin_max; };
__global__ void __launch_bounds__(1024) k_halo_push(HaloArgs h) {
    const HaloSide& sd = h.s[blockIdx.x];
    if (!sd.flag) return;                      // (no neighbour on this side)
    // a side with 0 rows still FLAGS: the neighbour's pull of exchange n + 1 is what orders its push of n + 2 behind this rank's pull of n, and
    // that only holds if every pull waits for every existing neighbour -- whether or not rows travel in that direction (need[] is the
    // band-row count of the adjacent plane and can be 0 on one side of a cut only)
    const int n = sd.rows * h.width;
    for (int p = 0; p < h.planes; ++p)
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned v = sd.src[(size_t)p * h.plane_words + i];
            asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(sd.dst + (size_t)p * n + i), "v"(v) : "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(sd.flag, h.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(1024) k_halo_pull(HaloArgs h) {
    const HaloSide& sd = h.s[blockIdx.x];      // (src: the own staging, dst: the array's halo rows, flag: in the own region)
    if (!sd.flag) return;                      // (no neighbour on this side; with one, the wait happens even for 0 rows: see k_halo_push)
    __shared__ int s_late;
    if (threadIdx.x == 0) {
        int spins = 0; s_late = 0;
        while (__hip_atomic_load(sd.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != h.tag) { __builtin_amdgcn_s_sleep(2); if (++spins > h.spin_max) { s_late = 1; break; } }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        if (s_late && h.host_late) __hip_atomic_store(h.host_late, h.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (host-mapped: engine.hip deliver_first turns it into PSGSDF_ERR_DEVICE)
        if (s_late) __hip_atomic_store(h.abort_flag, h.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (the "late" word of the halo pulls, NOT the solve's abort flag: a late pull hands on NaN rows, it does not disarm later solves -- ADVICE r04)
    }
    __syncthreads();
    const int n = sd.rows * h.width;
    for (int p = 0; p < h.planes; ++p)
        for (int i = threadIdx.x; i < n; i += blockDim.x)      // (a neighbour that never delivers: NaN rows -- the energies turn NaN and the host reports PSGSDF_ERR_DEVICE)
            sd.dst[(size_t)p * h.plane_words + i] = s_late ? 0x7fc00000u : __hip_atomic_load(sd.src + (size_t)p * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
int comm_halo(psgsdf_ctx* c, void* base, int planes, int width) {
    if (c->n_ranks == 1 || !c->halo_active) return 0;
    int rc = need_comm(c); if (rc) return rc;
    if (c->hx_ready && c->xr_ready && planes * width <= kHxWords) {
        const long long ep = ++c->hx_epoch;
        const int par = (int)(ep & 1);
        const size_t pw = (size_t)width * c->band.Spad;
        HaloArgs push{}, pull{};
        push.planes = pull.planes = planes; push.width = pull.width = width; push.plane_words = pull.plane_words = (long long)pw; push.tag = pull.tag = (double)ep;
        pull.abort_flag = push.abort_flag = c->xr + kXrLate + 2;
        pull.host_late = push.host_late = c->mbox_dev ? c->mbox_dev + c->mbox_n + 1 : nullptr; pull.spin_max = push.spin_max = c->xwait_spins;
        unsigned* arr = (unsigned*)base;
        for (int sd = 0; sd < 2; ++sd) {
            const int nb = sd == 0 ? c->rank - 1 : c->rank + 1;
            // what I give: my first give[0] rows to the lower neighbour (its upper side), my last give[1] rows to the upper neighbour (its lower side)
            if (nb < 0 || nb >= c->n_ranks) continue;
            if (c->give[sd] && c->hx_peer[sd]) {
                push.s[sd].src = arr + (size_t)width * (sd == 0 ? c->row0 : c->row1 - c->give[1]);
                push.s[sd].dst = (unsigned*)(c->hx_peer[sd] + (size_t)par * c->hx_peer_par[sd] + c->hx_peer_off[sd]);
                push.s[sd].rows = c->give[sd];
            }
            push.s[sd].flag = c->xr_peer[nb] + c->hx_flag_off + par * 2 + (1 - sd);      // (the neighbour's side seen from ITS end)
            if (c->need[sd]) {
                pull.s[sd].src = (const unsigned*)((char*)c->hx_mem + (size_t)par * c->hx_par + (sd == 0 ? 0 : sizeof(unsigned) * kHxWords * (size_t)c->need[0]));
                pull.s[sd].dst = arr + (size_t)width * (sd == 0 ? c->row0 - c->need[0] : c->row1);
                pull.s[sd].rows = c->need[sd];
            }
            pull.s[sd].flag = c->xr + c->hx_flag_off + par * 2 + sd;
        }
        if (!(c->fault_halo > 0 && ep == c->fault_halo))      // PSGSDF_FAULT_HALO=n: this rank's n-th exchange pushes nothing (the neighbours' bounded waits: tests)
            hipLaunchKernelGGL(k_halo_push, dim3(2), dim3(1024), 0, c->stream, push);
        hipLaunchKernelGGL(k_halo_pull, dim3(2), dim3(1024), 0, c->stream, pull);
        c->n_halo_pushes++;
        return 0;
    }
    const size_t rowb = 4 * (size_t)width, planeb = rowb * (size_t)c->band.Spad;
    std::vector<psgsdf_comm_xfer> sends, recvs;
    for (int p = 0; p < planes; ++p) {
        char* pl = (char*)base + planeb * p;
        if (c->give[0]) sends.push_back({pl + rowb * c->row0, rowb * c->give[0], c->rank - 1});
        if (c->need[0]) recvs.push_back({pl + rowb * (c->row0 - c->need[0]), rowb * c->need[0], c->rank - 1});
        if (c->give[1]) sends.push_back({pl + rowb * (c->row1 - c->give[1]), rowb * c->give[1], c->rank + 1});
        if (c->need[1]) recvs.push_back({pl + rowb * c->row1, rowb * c->need[1], c->rank + 1});
    }
    return comm_xfer(c, sends, recvs);
}
// grouped point-to-point transfers of device memory, ordered on the context's stream (halo rows with the z-neighbours; whole planes between any two
// ranks when the slabs are re-cut, api.hip psgsdf_rebalance_slabs); matched in list order per peer
int comm_xfer(psgsdf_ctx* c, const std::vector<psgsdf_comm_xfer>& sends, const std::vector<psgsdf_comm_xfer>& recvs) {
    if (sends.empty() && recvs.empty()) return 0;
    int rc = need_comm(c); if (rc) return rc;
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
    if (first != ncclSuccess) return fail(c, PSGSDF_ERR_COMM, "point-to-point exchange (ncclSend / ncclRecv): %s", r->GetErrorString ? r->GetErrorString(first) : "rccl error");
    NCCLCHK(c, end);
    return 0;
}


// ---- cross-rank persistent solve: memory kinds, the hand-off probe, IPC mappings -------------------------------------------------------
// Memory another device writes (or this device polls) WHILE a kernel runs must not be plain hipMalloc memory: HIP only promises visibility of
// another device's writes to coarse-grained memory at kernel boundaries.  Two kinds are used:
//   polled words (the mailbox region: tags, rank granules, abort flag; KB-sized)   -> hipDeviceMallocUncached (every access goes to memory)
//   the two PCG record planes (their halo rows are written by the neighbours)      -> hipDeviceMallocFinegrained (kind 1) or uncached (kind 2)
// Which of the two carries the hand-offs between the REAL neighbours is not assumed but probed once per context (xr_probe): the very store /
// load / fence forms of pcg.hip k_cgf_solve<.., MR> over the real mappings, the reader holding stale copies of the payload in its caches first.
// On one GPU (the tests) every kind passes; what the first multi-GPU run finds is reported in psgsdf_debug_sync_stats / the bench line.
int xr_alloc(psgsdf_ctx* c, void** p, size_t bytes, bool polled) {
    *p = nullptr;
    const int kind = c->xr_mem_kind;
    if (c->n_ranks <= 1 || kind <= 0 || kind == 3) { HIPCHK(c, hipMalloc(p, bytes)); return 0; }      // single rank (or the cross-rank solve is off): nobody else touches it  (3: PSGSDF_XR_MEM=coarse, round 3's allocation, for comparison)
    const unsigned flag = (polled || kind == 2) ? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
    HIPCHK(c, hipExtMallocWithFlags(p, bytes, flag));
    return 0;
}

namespace {
typedef float v4f_probe_t __attribute__((ext_vector_type(4)));
constexpr int kProbeRecs = 2048, kProbeRounds = 48, kProbeFlagDoubles = 64;
// (the instruction of device_common.h store8_system / pcg.hip store8_sys, spelled out: this file does not include the kernels' header)
__device__ __forceinline__ void probe_store8(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
// Two roles, one workgroup each.  OWNER (towards the lower neighbour): pull the payload into this device's caches, wait for the neighbour's
// round flag (system-scope relaxed loads), ONE system-scope acquire fence, read the payload with plain loads and count records that do not
// carry the round, answer.  PEER (towards the upper neighbour): write-through stores into its payload -- 16-byte records in even rounds (the classic
// solve), 8-byte words in odd rounds (the pipelined solve, the frame rows, the scalar folds) --, drain, flag, wait for the answer.
__global__ void __launch_bounds__(256) k_xr_probe(double* myF, const float4* myP, double* loF, double* hiF, float4* hiP, int has_lo, int has_hi, double* out) {
    __shared__ int s_to; __shared__ unsigned long long s_stale;
    if (threadIdx.x == 0) { s_to = 0; s_stale = 0; }
    __syncthreads();
    if (blockIdx.x == 0 && has_lo) {
        unsigned long long stale = 0; float sink = 0.f;
        for (int r = 1; r <= kProbeRounds; ++r) {
            for (int i = threadIdx.x; i < kProbeRecs; i += blockDim.x) sink += myP[i].x + (float)((const double*)myP)[2 * i + 1];      // (stale copies of both forms in this device's caches)
            __syncthreads();
            if ((r & 3) == 3) {
                // every fourth round: the pipelined solve's SELF-VALIDATING hand-off (pcg.hip k_cgp_solve<.., MR, TM>) -- no flag wait, no fence:
                // every word is polled with system-scope relaxed loads until it carries the round's 2-bit tag, as a halo row's m is.  The previous round of
                // this kind (r - 4) left words with the SAME tag in the same place, and this device holds cached copies of them (the loop above): a word
                // that validates by its tag but is not this round's is exactly the silent failure a two-valued tag would allow -- counted as stale.
                for (int i = threadIdx.x; i < kProbeRecs; i += blockDim.x) {
                    const double* src = (const double*)myP + 2 * i;
                    int spins = 0; long long w;
                    do { w = (long long)__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); if (((w >> 12) & 3) == 3) break; __builtin_amdgcn_s_sleep(1); } while (++spins <= (1 << 21));
                    if (spins > (1 << 21)) s_to = 1;
                    else if (w != (((long long)(r >> 2) << 14) | (3ll << 12) | (long long)i)) stale++;
                }
            }
            if (threadIdx.x == 0) {
                int spins = 0;
                while (__hip_atomic_load(myF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (double)r) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 21)) { s_to = 1; break; } }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            }
            __syncthreads();
            if (s_to) break;
            if ((r & 3) == 3) { }
            else if (r & 1) { const double* pd = (const double*)myP; for (int i = threadIdx.x; i < kProbeRecs; i += blockDim.x) { if (pd[2 * i] != (double)r || pd[2 * i + 1] != (double)(r + i)) stale++; } }      // odd rounds: the pipelined solve's form (8-byte records, plain 8-byte loads)
            else for (int i = threadIdx.x; i < kProbeRecs; i += blockDim.x) { const float4 v = myP[i]; if (v.x != (float)r || v.w != (float)(r + i)) stale++; }
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(loF + 8, (double)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        atomicAdd(&s_stale, stale);
        __syncthreads();
        if (threadIdx.x == 0) { out[0] = (double)s_stale; out[1] = (double)s_to; out[3] = (double)sink; }
    }
    if (blockIdx.x == 1 && has_hi) {
        for (int r = 1; r <= kProbeRounds; ++r) {
            // even rounds: k_cgf_solve's hand-off (16-byte sc0 sc1 records); odd rounds: k_cgp_solve's, the frame rows' and the scalar folds' (8-byte
            // sc0 sc1 words: device_common.h store8_system / pcg.hip store8_sys) -- the default solve is the pipelined one
            if ((r & 3) == 3) for (int i = threadIdx.x; i < kProbeRecs; i += blockDim.x) probe_store8((double*)hiP + 2 * i, (double)(((long long)(r >> 2) << 14) | (3ll << 12) | (long long)i));      // tagged words: (round / 4) << 14 | tag 3 << 12 | index
            else if (r & 1) for (int i = threadIdx.x; i < kProbeRecs; i += blockDim.x) { probe_store8((double*)hiP + 2 * i, (double)r); probe_store8((double*)hiP + 2 * i + 1, (double)(r + i)); }
            else for (int i = threadIdx.x; i < kProbeRecs; i += blockDim.x) {
                const v4f_probe_t d = {(float)r, 0.f, 0.f, (float)(r + i)};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(hiP + i), "v"(d) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_store(hiF, (double)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                int spins = 0;
                while (__hip_atomic_load(myF + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (double)r) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 21)) { s_to = 1; break; } }
            }
            __syncthreads();
            if (s_to) break;
        }
        if (threadIdx.x == 0) out[2] = (double)s_to;
    }
}

}  // namespace
// all-reduce (sum over the ranks) of a host vector of doubles through a scratch device buffer
int host_allreduce(psgsdf_ctx* c, std::vector<double>& buf, const char* what) {
    double* d = nullptr;
    HIPCHK(c, hipMalloc(&d, sizeof(double) * buf.size()));
    int rc = 0;
    if (hipMemcpyAsync(d, buf.data(), sizeof(double) * buf.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "%s: upload", what);
    if (!rc) rc = comm_allreduce(c, d, (int)buf.size());
    if (!rc && (hipMemcpyAsync(buf.data(), d, sizeof(double) * buf.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) rc = fail(c, PSGSDF_ERR_DEVICE, "%s: download", what);
    hipFree(d);
    return rc;
}
namespace {
void handle_to_doubles(const hipIpcMemHandle_t& h, double* out) { for (int i = 0; i < 64; ++i) out[i] = (double)((const unsigned char*)&h)[i]; }
void* open_handle(const double* bytes) {
    hipIpcMemHandle_t h; for (int i = 0; i < 64; ++i) ((unsigned char*)&h)[i] = (unsigned char)bytes[i];
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
}  // namespace

// Collective, once per context (the first band of a multi-rank context): choose the memory kind for the cross-rank hand-offs by trying them
// between the real neighbours.  Leaves c->xr_mem_kind = 1 / 2 (passed on EVERY rank) or 0 (none did: the cross-rank solve stays off).
int xr_probe(psgsdf_ctx* c) {
    const int R = c->n_ranks, me = c->rank;
    if (c->xr_mem_kind >= 0 || R <= 1 || !c->comm) return 0;
    c->xr_mem_kind = 0;
    int want = 0;                    // PSGSDF_XR_MEM=fine|uncached|coarse pins the kind (coarse: plain hipMalloc, round 3's allocation -- for comparison only)
    if (const char* e = getenv("PSGSDF_XR_MEM")) want = !strcmp(e, "fine") ? 1 : !strcmp(e, "uncached") ? 2 : !strcmp(e, "coarse") ? 3 : 0;
    constexpr int kSlice = 64 + 64 + 2;
    for (int kind = 1; kind <= 2; ++kind) {
        if (want == 1 || want == 2) { if (kind != want) continue; }
        double* F = nullptr; float4* P = nullptr; double* out = nullptr;
        hipIpcMemHandle_t hF{}, hP{};
        bool ok = c->xr_enable
            && hipExtMallocWithFlags((void**)&F, sizeof(double) * kProbeFlagDoubles, hipDeviceMallocUncached) == hipSuccess
            && (want == 3 ? hipMalloc((void**)&P, sizeof(float4) * kProbeRecs) : hipExtMallocWithFlags((void**)&P, sizeof(float4) * kProbeRecs, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached)) == hipSuccess
            && hipMalloc((void**)&out, sizeof(double) * 4) == hipSuccess
            && hipMemsetAsync(F, 0, sizeof(double) * kProbeFlagDoubles, c->stream) == hipSuccess && hipMemsetAsync(P, 0, sizeof(float4) * kProbeRecs, c->stream) == hipSuccess
            && hipMemsetAsync(out, 0, sizeof(double) * 4, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess
            && hipIpcGetMemHandle(&hF, F) == hipSuccess && hipIpcGetMemHandle(&hP, P) == hipSuccess;
        (void)hipGetLastError();
        std::vector<double> buf((size_t)R * kSlice, 0.0);
        double* mine = buf.data() + (size_t)me * kSlice;
        handle_to_doubles(hF, mine); handle_to_doubles(hP, mine + 64);
        mine[128] = (double)getpid(); mine[129] = ok ? 1.0 : 0.0;
        int rc = host_allreduce(c, buf, "cross-rank probe");
        std::vector<void*> opened;
        double *loF = nullptr, *hiF = nullptr; float4* hiP = nullptr;
        if (!rc) {
            for (int r = 0; r < R; ++r) if (buf[(size_t)r * kSlice + 129] != 1.0 || (r != me && buf[(size_t)r * kSlice + 128] == (double)getpid())) ok = false;      // (two ranks in ONE process: no IPC between them)
            if (ok && me > 0) { loF = (double*)open_handle(buf.data() + (size_t)(me - 1) * kSlice); if (loF) opened.push_back(loF); else ok = false; }
            if (ok && me < R - 1) {
                hiF = (double*)open_handle(buf.data() + (size_t)(me + 1) * kSlice); if (hiF) opened.push_back(hiF); else ok = false;
                hiP = ok ? (float4*)open_handle(buf.data() + (size_t)(me + 1) * kSlice + 64) : nullptr; if (hiP) opened.push_back(hiP); else ok = false;
            }
            // everybody mapped everything?  (this all-reduce also lines the ranks up in front of the kernels that wait for each other)
            std::vector<double> agree(1, ok ? 1.0 : 0.0);
            rc = host_allreduce(c, agree, "cross-rank probe");
            ok = !rc && agree[0] == (double)R;
        }
        std::vector<double> res(3, 0.0);
        if (!rc && ok) {
            hipLaunchKernelGGL(k_xr_probe, dim3(2), dim3(256), 0, c->stream, F, (const float4*)P, loF, hiF, hiP, me > 0 ? 1 : 0, me < R - 1 ? 1 : 0, out);
            double o[4] = {0, 0, 0, 0};
            if (hipMemcpyAsync(o, out, sizeof(o), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "cross-rank probe: kernel");
            res[0] = o[0]; res[1] = o[1] + o[2]; res[2] = 1.0;
            c->xr_probe_local[kind][0] = (long long)o[0]; c->xr_probe_local[kind][1] = (long long)o[1]; c->xr_probe_local[kind][2] = (long long)o[2]; c->xr_probe_local[kind][3] = 1;      // this rank's own view: stale records it read from its lower neighbour, waits that expired towards the lower / upper neighbour
            if (!rc) rc = host_allreduce(c, res, "cross-rank probe");
            c->xr_probe_stale += (long long)res[0]; c->xr_probe_timeouts += (long long)res[1];
        }
        // close before anybody frees: mappings first, then a barrier, then the allocations
        for (void* p : opened) hipIpcCloseMemHandle(p);
        if (!rc) { std::vector<double> bar(1, 1.0); rc = host_allreduce(c, bar, "cross-rank probe"); }
        hipFree(F); hipFree(P); hipFree(out);
        (void)hipGetLastError();
        if (rc) return rc;
        if (ok && res[2] == (double)R && res[0] == 0.0 && res[1] == 0.0) { c->xr_mem_kind = want == 3 ? 3 : kind; break; }
        if (ok && me == 0) fprintf(stderr, "psgsdf: cross-rank hand-off probe with %s record memory: %lld stale records, %lld timed-out waits over %d ranks\n",
                                   kind == 1 ? "fine-grained" : "uncached", (long long)res[0], (long long)res[1], R);
        if (!ok) break;      // (no IPC between these ranks at all: another kind will not help)
    }
    if (c->xr_mem_kind == 0 && me == 0 && c->xr_enable) fprintf(stderr, "psgsdf: no memory kind carried the in-kernel hand-offs between the %d ranks: the distance solve uses the per-pass kernels + one all-reduce per pass\n", R);
    return 0;
}

// Every rank exports (a) its small mailbox region and (b) its two record planes (the neighbours write the records of their cut-side rows into its halo rows);
// the handles and a few numbers travel through ONE all-reduce in which every rank fills its own slice (bytes as doubles, zeros elsewhere), so the
// exchange works over RCCL and over a caller-supplied transport alike.  All ranks agree on the outcome with a second all-reduce: either every rank
// runs the cross-rank persistent solve or every rank stays on the per-pass kernels.  EVERY rank of a multi-rank context takes part in both
// all-reduces, whatever its own knobs or state say (a rank that cannot contributes ok = 0): nobody is left waiting in a collective.
namespace {
// the nonce words of up to 32 mapped regions, read the way the exchange kernels read and write them (system scope, through the mappings)
struct NonceArgs { const double* src[32]; int n; };
__global__ void k_xr_nonces(NonceArgs a, double* out) { if ((int)threadIdx.x < a.n) out[threadIdx.x] = a.src[threadIdx.x] ? __hip_atomic_load(a.src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0; }
__global__ void k_xr_closed(double* slot, double serial) { __hip_atomic_store(slot, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
}  // namespace
// Closing this rank's mappings of the peers' memory.  Every peer whose region is mapped is TOLD so first: one system-scope store of the set-up's
// serial number into slot [closed_off + my rank] of ITS region (a one-thread kernel on the context's stream -- the same kind of store every
// in-kernel exchange uses), drained before the mappings go.  The owner of a region reads these slots before it frees what it exported
// (xr_quiesce): the release protocol needs no collective.
void xr_release(psgsdf_ctx* c) {
    bool told = false;
    if (c->xr_closed_off > 0 && c->stream)
        for (size_t r = 0; r < c->xr_peer.size(); ++r)
            if ((int)r != c->rank && c->xr_peer[r]) { hipLaunchKernelGGL(k_xr_closed, dim3(1), dim3(1), 0, c->stream, c->xr_peer[r] + c->xr_closed_off + c->rank, (double)c->xr_serial); told = true; }
    if (told) { (void)hipStreamSynchronize(c->stream); (void)hipGetLastError(); }
    for (void* p : c->xr_opened) hipIpcCloseMemHandle(p);
    c->xr_opened.clear(); c->xr_peer.clear(); c->band_peer[0] = c->band_peer[1] = nullptr; c->xr_ready = false; c->xr_args = XrArgs{};
    c->hx_ready = false; c->hx_peer[0] = c->hx_peer[1] = nullptr;
}
// Before rec_mem, the halo staging or the mailbox region are freed (band rebuild, destroy) every rank that mapped them has to have closed its
// mappings: freeing memory an importer still maps is undefined in HIP.  NOT a collective (a barrier here hung every
// healthy rank inside psgsdf_destroy as soon as one rank had failed, had already gone, or destroyed its contexts in another order): this rank
// closes its own mappings (telling their owners, xr_release) and then watches the "closed" slots of ITS region until every rank that opened it
// at the last set-up (xr_openers, agreed there) has written that set-up's serial number -- for at most `timeout_s` seconds.
//   returns 0 : nobody maps this rank's memory any more (or nobody ever did)
//           1 : a peer never reported: the caller must NOT free xr / rec_mem / hx_mem (psgsdf_destroy leaks them, logged once; build_band fails)
int xr_quiesce(psgsdf_ctx* c, double timeout_s) {
    xr_release(c);
    const unsigned long long openers = c->xr_openers;
    const bool was_mapped = c->xr_mapped;
    c->xr_mapped = false; c->xr_openers = 0;
    if (!was_mapped || !openers || !c->xr || c->xr_closed_off <= 0 || c->n_ranks <= 1) return 0;
    const int R = c->n_ranks;
    std::vector<double> slots(R, 0.0);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        if (hipMemcpy(slots.data(), c->xr + c->xr_closed_off, sizeof(double) * R, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 1; }
        bool all = true;
        for (int r = 0; r < R && all; ++r) if (r != c->rank && ((openers >> r) & 1ull) && slots[r] != (double)c->xr_serial) all = false;
        if (all) return 0;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return 1;
        usleep(200);
    }
}
int xr_setup(psgsdf_ctx* c, const std::vector<double>& info) {
    xr_release(c);
    const int R = c->n_ranks, me = c->rank;
    if (R <= 1 || !c->comm) return 0;
    if (R > kXrMaxRanks) return 0;                                 // (the same on every rank)
    constexpr int kSlice = 64 + 64 + 8 + 64;      // two IPC handles (64 bytes each, one double per byte) + {rec[0] offset, rec[1] offset, pid, ok} + the halo staging's handle
    std::vector<double> buf((size_t)R * kSlice, 0.0);
    hipIpcMemHandle_t hx{}, hb{}, hh{};
    c->hx_ready = false; c->hx_peer[0] = c->hx_peer[1] = nullptr;
    int Gs, Rs;
    bool ok = c->xr_enable && c->pcg_persist && c->pcg_fuse_asm && c->xr_mem_kind > 0 && c->rec_mem;
    // the region: the solve's fixed words + the frame rows' exchange area (engine.h XfTable) for this many ranks and keyframes.  (Every peer closed
    // its mapping of the old region in xr_quiesce at the top of build_band: it may be replaced here.)
    const size_t frows = (size_t)2 * R * std::max(c->F, 1);
    const size_t want = (size_t)kXrDoubles + frows * kFrameRow + frows + (size_t)2 * R * 8 + (size_t)2 * R + 8 + (size_t)R;      // (+ 4 flags of the halo pushes, comm_halo; + one "closed" slot per rank, xr_release / xr_quiesce)
    // halo staging: what the two neighbours push of their cut-side rows (comm_halo): [2 parities][lower side: need_lo rows | upper side: need_hi rows] x kHxWords words
    const size_t hx_par = sizeof(unsigned) * kHxWords * ((size_t)c->need[0] + c->need[1]);
    if (c->hx_mem) { hipFree(c->hx_mem); c->hx_mem = nullptr; }
    if (ok && c->xh_enable && hx_par) { if (xr_alloc(c, &c->hx_mem, 2 * hx_par, false)) { (void)hipGetLastError(); c->hx_mem = nullptr; } }
    const bool hx_ok = ok && c->hx_mem && hipIpcGetMemHandle(&hh, c->hx_mem) == hipSuccess;
    if (ok && c->xr && c->xr_doubles < want) { hipFree(c->xr); c->xr = nullptr; c->xr_doubles = 0; }
    if (ok && !c->xr) {
        if (xr_alloc(c, (void**)&c->xr, sizeof(double) * want, true) || hipMemsetAsync(c->xr, 0, sizeof(double) * want, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) ok = false;
        else c->xr_doubles = want;
    }
    ok = ok && cgf_solve_shape(c, &Gs, &Rs, true)      // (this slab fits the persistent kernel at all)
        && hipIpcGetMemHandle(&hx, c->xr) == hipSuccess && hipIpcGetMemHandle(&hb, c->rec_mem) == hipSuccess;
    (void)hipGetLastError();
    // The region is zeroed HERE -- every peer closed its mapping of it before this band was built (xr_quiesce), nobody writes into it until the
    // agreement at the end of this function -- and then stamped with a nonce no other region of this run carries (pid, allocation counter): the
    // peers read it back THROUGH THEIR MAPPING before they rely on it.  (Round 5: with eight processes creating and destroying contexts in turn,
    // a rank was seen to sweep into a region nobody polled -- every peer timed out waiting for ITS flags while it saw all of theirs.  A mapping
    // that does not show the owner's nonce is not the owner's memory.)
    static std::atomic<unsigned> nonce_counter{0};      // (contexts of one process may live on different host threads)
    const double nonce = (double)(getpid() % 1000000) * 4096.0 + (double)((nonce_counter.fetch_add(1) + 1u) % 4096u) + 0.5;
    if (c->xr) {
        if (hipMemsetAsync(c->xr, 0, sizeof(double) * c->xr_doubles, c->stream) != hipSuccess
            || hipMemcpyAsync(c->xr + kXrNonce, &nonce, sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess
            || hipStreamSynchronize(c->stream) != hipSuccess) ok = false;
        (void)hipGetLastError();
    }
    double* mine = buf.data() + (size_t)me * kSlice;
    handle_to_doubles(hx, mine); handle_to_doubles(hb, mine + 64);
    mine[128] = (double)((char*)c->band.rec[0] - (char*)c->rec_mem); mine[129] = (double)((char*)c->band.rec[1] - (char*)c->rec_mem);
    mine[130] = (double)getpid(); mine[131] = ok ? 1.0 : 0.0; mine[133] = nonce;
    mine[132] = (hx_ok || (ok && c->xh_enable && !hx_par)) ? 1.0 : 0.0;      // (a rank that needs no halo rows has nothing to stage and is fine with the pushes)
    if (hx_ok) handle_to_doubles(hh, mine + 136);
    c->xr_serial += 1;               // (the same on every rank: set-ups are collective)
    c->xr_closed_off = (long long)want - R;
    int rc = host_allreduce(c, buf, "cross-rank set-up");
    if (rc) return rc;
    c->xr_mapped = true;             // (handles are out: a peer may map them from here on)
    c->xr_openers = ~0ull;           // (until the agreement below says who did)
    auto open = [&](const double* bytes) -> void* { void* p = open_handle(bytes); if (p) c->xr_opened.push_back(p); return p; };
    // map every rank's region, and the two neighbours' band arenas
    c->xr_peer.assign(R, nullptr); c->xr_peer[me] = c->xr;
    for (int r = 0; r < R && ok; ++r) {
        const double* sl = buf.data() + (size_t)r * kSlice;
        if (sl[131] != 1.0) { ok = false; break; }
        if (r == me) continue;
        if (sl[130] == (double)getpid()) { ok = false; break; }      // two ranks in ONE process: no IPC between them (the per-pass path serves that case)
        c->xr_peer[r] = (double*)open(sl);
        if (!c->xr_peer[r]) { ok = false; break; }
        if (r == me - 1 || r == me + 1) { c->band_peer[r == me - 1 ? 0 : 1] = open(sl + 64); if (!c->band_peer[r == me - 1 ? 0 : 1]) { ok = false; break; } }
    }
    // is every mapping the owner's LIVE region?  (its nonce, read through the mapping by a kernel)
    if (ok) {
        NonceArgs na{}; na.n = R;
        for (int r = 0; r < R; ++r) na.src[r] = c->xr_peer[r] ? c->xr_peer[r] + kXrNonce : nullptr;
        double got[32] = {0};
        hipLaunchKernelGGL(k_xr_nonces, dim3(1), dim3(32), 0, c->stream, na, c->mg_scal);
        if (hipMemcpyAsync(got, c->mg_scal, sizeof(double) * R, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); ok = false; }
        for (int r = 0; r < R && ok; ++r) {
            const double want_nonce = buf[(size_t)r * kSlice + 133];
            if (got[r] != want_nonce) {
                fprintf(stderr, "psgsdf: rank %d: the mapping of rank %d's exchange region does not show that rank's memory (nonce %.1f, expected %.1f): in-kernel exchanges off for this band\n", me, r, got[r], want_nonce);
                c->xr_stale_maps++; ok = false;
            }
        }
    }
    // the neighbours' halo stagings (halo pushes instead of RCCL send / recv: all ranks or none)
    bool hx_all = ok;
    for (int r = 0; r < R && hx_all; ++r) if (buf[(size_t)r * kSlice + 132] != 1.0) hx_all = false;
    if (me > 0 && c->give[0] != (int)info[3 * (me - 1) + 1]) hx_all = false;      // (what I push has to be exactly what the neighbour stages)
    if (me < R - 1 && c->give[1] != (int)info[3 * (me + 1)]) hx_all = false;
    if (hx_all) {
        for (int sd = 0; sd < 2 && hx_all; ++sd) {
            const int nb = sd == 0 ? me - 1 : me + 1;
            if (nb < 0 || nb >= R || c->give[sd] == 0) continue;
            c->hx_peer[sd] = (char*)open(buf.data() + (size_t)nb * kSlice + 136);
            if (!c->hx_peer[sd]) { hx_all = false; break; }
            const size_t nlo = (size_t)info[3 * nb], nhi = (size_t)info[3 * nb + 1];
            c->hx_peer_par[sd] = sizeof(unsigned) * kHxWords * (nlo + nhi);
            // my first rows are the LOWER neighbour's upper halo, my last rows the UPPER neighbour's lower halo
            c->hx_peer_off[sd] = sd == 0 ? sizeof(unsigned) * kHxWords * nlo : 0;
        }
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
            for (int q = 0; q < 2; ++q) { x.lo_base[q] = (float4*)((char*)c->band_peer[0] + (size_t)sl[128 + q]); x.lo_rec[q] = x.lo_base[q] + (need_lo_p + own_p); }
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
    // agreement: every rank or none.  (Every rank's region was zeroed above, before its handle went out: no word of an earlier band or a raised
    // abort flag survives into the solves of this one.)
    if (ok) {      // the frame rows' exchange table (the flags carry the exchange's number, which only grows: the zeroed region matches none)
        XfTable t{}; t.n_ranks = R; t.rank = me; t.F = std::max(c->F, 1); t.spin_max = c->xwait_spins; t.pay = kXrDoubles; t.flg = (long long)kXrDoubles + (long long)frows * kFrameRow; t.spay = t.flg + (long long)frows; t.sflg = t.spay + (long long)2 * R * 8;
        for (int r = 0; r < R; ++r) t.region[r] = c->xr_peer[r];
        if (!c->xf_table && hipMalloc(&c->xf_table, sizeof(XfTable)) != hipSuccess) ok = false;
        if (ok && hipMemcpyAsync(c->xf_table, &t, sizeof(t), hipMemcpyHostToDevice, c->stream) != hipSuccess) ok = false;
        if (ok && hipStreamSynchronize(c->stream) != hipSuccess) ok = false;
    }
    // ... and who holds a mapping of whose memory: entry 2 + r collects one bit per rank that opened rank r's region (R <= 32: exact in a double)
    std::vector<double> agree(2 + (size_t)R, 0.0); agree[0] = ok ? 1.0 : 0.0; agree[1] = (ok && hx_all) ? 1.0 : 0.0;
    for (int r = 0; r < R && r < (int)c->xr_peer.size(); ++r) if (r != me && c->xr_peer[r]) agree[2 + r] = (double)(1ull << me);
    if ((rc = host_allreduce(c, agree, "cross-rank set-up"))) return rc;
    c->xr_openers = (unsigned long long)agree[2 + me];
    if (agree[0] != (double)R) { xr_release(c); return 0; }      // (xr_release tells the owners: their xr_quiesce will not wait for this rank)
    c->xr_args = x; c->xr_ready = true;
    c->hx_ready = agree[1] == (double)R;
    c->hx_par = hx_par; c->hx_flag_off = (long long)want - 8 - R;
    return 0;
}

}  // namespace psge
