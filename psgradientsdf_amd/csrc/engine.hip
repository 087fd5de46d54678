// engine.hip — host orchestration of the HIP engine and the C ABI of include/psgsdf.h.
// One context = one HIP device + one stream.  The kernels live in band / sweeps / dist / pcg / albedo_reg / frontend .hip.
#include "engine.h"
#include "../../include/psgsdf.h"

#include <math.h>
#include <float.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <vector>

using namespace psg;

namespace {
constexpr int kMgScal = 64;   // doubles in the folded-scalar exchange buffer

struct KTime { double ms = 0; int64_t n = 0; };

}  // namespace

struct psgsdf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    psgsdf_settings set{};
    float reg_n = 0, reg_l = 0;
    GridP grid{};
    float shift[3]{};
    Cam cam{};
    // dense
    DenseView dense{};
    uint64_t* vis_seq = nullptr; int wpv_seq = 0;
    int* block_sums = nullptr; int* d_total = nullptr;
    bool have_volume = false;
    // frames
    int F = 0;
    int* frame_idx = nullptr;
    float* img = nullptr;
    FrameP* frames = nullptr;            // device
    std::vector<FrameP> frames_h;        // host mirror of the initial records
    float* led_light = nullptr;          // device [3]
    bool have_frames = false;
    // band
    void* band_mem = nullptr; size_t band_bytes = 0;
    void* obs_mem = nullptr;
    float* stage = nullptr; size_t stage_px = 0;   // device staging of one RGB-D frame (integrate_frame)
    // front end: FALS cache (9 float planes), box-filter scratch, tracker partials
    float* ncache = nullptr; double* ntmp = nullptr; float* nout = nullptr; float* ndepth = nullptr; int ncache_w = 0, ncache_h = 0;
    double* track_part = nullptr; double* track_host = nullptr;
    Band band{};
    bool inited = false;
    // accumulators
    double* acc_frame = nullptr; size_t acc_frame_n = 0;
    double* part = nullptr; int PB = 0;  // [SC_COUNT][PB] per-workgroup partials
    double* pcg_sc = nullptr; int pcg_cap = 4096;
    double* pcg_part = nullptr;          // [2][3][kPcgMaxBlocks]
    int last_cg_iters = 0;
    bool want_counts = true;             // read back the accepted-update counts (debug statistic of the reference)
    double* host_buf = nullptr; size_t host_buf_n = 0;   // pinned readback
    // deferred read-backs: small fold kernels write into a host-mapped pinned mailbox (no D2H copies), consumed at the
    // next host sync
    double* mbox = nullptr; double* mbox_dev = nullptr; size_t mbox_n = 0, mbox_used = 0;
    std::vector<std::function<void()>> deferred;
    // cached energies
    double en_sum = 0, el_sum = 0;       // sums over the band from the last k_derive
    // row partition (multi-rank): this context owns band rows [row0, row1); halo = widest column reach
    int rank = 0, n_ranks = 1;
    int row0 = 0, row1 = 0, halo = 0;
    double* mg_scal = nullptr;           // [kMgScal] folded local sums the host program all-reduces (phase results land at mg_fold_base)
    double* mg_ext = nullptr;            // [8] PCG: local sums of a pass out, globally reduced sums in
    double* mg_hist = nullptr;           // [pcg_cap + 2] PCG: what kernel k published (|b|^2, then |r|^2 after pass k-1)
    int mg_fold_base = 0;
    float reg_r = 0.f;                   // "reg albedo" (never normalised, PsOptimizer.cpp:279)
    void* areg_mem = nullptr; AlbedoReg ar{};   // planes of the albedo regulariser, allocated with the band when reg_r != 0
    double er_sum = 0;                   // sum over the band of sum_c ||grad rho_c|| at the last evaluation
    FoldReq pending_fold{};              // scalar fold waiting for the next kernel (read_parts_deferred / take_fold)
    bool fold_in_next = true;            // PSGSDF_FOLD_IN_NEXT=0: always a k_sum_parts launch
    double* frame_e_slot = nullptr;      // mailbox slot the next per-frame solve writes its sweep's energy sums to
    bool pcg_poll = true;                // PCG stop test by watching the mapped mailbox (PSGSDF_PCG_POLL=0: drain the stream instead)
    int need[2] = {0, 0}; int* d_need = nullptr;   // halo rows needed below row0 / from row1 up
    int* mg_slots = nullptr;             // [8] device copy of slot ids for k_sum_parts
    bool own_stream = true;
    // profiling
    bool profiling = false;
    std::map<std::string, KTime> ktimes;
    std::vector<const char*> kt_names;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // asynchronous watch of ONE kernel name: event pairs recorded on the launch stream, resolved on query
    std::string watch;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> watch_pool;
    size_t watch_used = 0; int watch_every = 1; size_t watch_seen = 0;
    char err[512] = {0};
};

namespace {

int flush(psgsdf_ctx* c);
int fail(psgsdf_ctx* c, int code, const char* fmt, ...) {
    if (c) { va_list ap; va_start(ap, fmt); vsnprintf(c->err, sizeof(c->err), fmt, ap); va_end(ap); }
    return code;
}
#define HIPCHK(c, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(c, PSGSDF_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)

template <class Fn> void timed(psgsdf_ctx* c, const char* name, Fn&& fn) {
    if (!c->profiling) {
        if (!c->watch.empty() && c->watch == name && (c->watch_seen++ % c->watch_every) == 0) {   // a sample of the launches: the event pair costs ~3 us of stream time
            if (c->watch_used == c->watch_pool.size()) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); c->watch_pool.emplace_back(a, b);
            }
            auto& pr = c->watch_pool[c->watch_used++];
            hipEventRecord(pr.first, c->stream);
            fn();
            hipEventRecord(pr.second, c->stream);
            return;
        }
        fn(); return;
    }
    hipEventRecord(c->ev0, c->stream);
    fn();
    hipEventRecord(c->ev1, c->stream);
    hipEventSynchronize(c->ev1);
    float ms = 0; hipEventElapsedTime(&ms, c->ev0, c->ev1);
    KTime& k = c->ktimes[name]; k.ms += ms; k.n += 1;
}

SweepArgs make_args(psgsdf_ctx* c, int laplacian_reg) {
    SweepArgs a{};
    a.b = c->band; a.frames = c->frames; a.img = c->img; a.F = c->F; a.cam = c->cam; a.grid = c->grid;
    a.img32 = (size_t)c->F * c->cam.W * c->cam.H * 12 < ((size_t)1 << 32);
    a.rob.loss = c->set.loss; a.rob.lambda = c->set.lambda; a.rob.lambda_sq = c->set.lambda * c->set.lambda; a.rob.inv_lambda = 1.0f / c->set.lambda;
    a.acc.frame = c->acc_frame; a.acc.part = c->part; a.acc.PB = c->PB;
    a.fold.n = 0; a.gate = nullptr;
    a.ar = c->ar; a.ar.weight = c->reg_r;
    a.model = c->set.model; a.quirks = c->set.ref_quirks;
    a.reg_n = c->reg_n; a.reg_l = c->reg_l;
    a.normal_reg = c->reg_n != 0.0f; a.laplacian_reg = laplacian_reg;
    a.damping = c->set.damping;
    a.row0 = c->row0; a.row1 = c->row1;
    a.ext = nullptr;
    return a;
}

inline int band_blocks(const psgsdf_ctx* c) { return (c->row1 - c->row0 + kBlock - 1) / kBlock; }
int read_parts_deferred(psgsdf_ctx* c, const int* slots, int n, std::function<void(const double*)> consume);
int read_frame_energy_deferred(psgsdf_ctx* c, int col_e, std::function<void(double, double)> consume);
void materialize_fold(psgsdf_ctx* c);
// blocking variants
int read_parts(psgsdf_ctx* c, const int* slots, int n, double* out) {
    int rc = read_parts_deferred(c, slots, n, [out, n](const double* v) { for (int i = 0; i < n; ++i) out[i] = v[i]; });
    return rc ? rc : flush(c);
}
int read_frame_energy(psgsdf_ctx* c, int col_e, double* E, double* nobs) {
    int rc = read_frame_energy_deferred(c, col_e, [E, nobs](double e, double n) { *E = e; *nobs = n; });
    return rc ? rc : flush(c);
}
// synchronise the stream once and run every deferred consumer in submission order
int flush(psgsdf_ctx* c) {
    materialize_fold(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto& f : c->deferred) f();
    c->deferred.clear(); c->mbox_used = 0;
    return 0;
}
// sum of per-workgroup partials of `slots`, delivered to `consume(sums)` at the next flush (no host sync here).
// The fold itself is left pending: the next kernel that can take it (take_fold) does it in its first workgroup;
// anything else that needs the value first (flush, a kernel that writes those slots) launches k_sum_parts.
void materialize_fold(psgsdf_ctx* c) {
    if (!c->pending_fold.n) return;
    SlotList sl; sl.n = c->pending_fold.n; for (int i = 0; i < sl.n; ++i) sl.id[i] = c->pending_fold.id[i];
    launch_sum_parts(c->part, c->PB, c->pending_fold.nblk, sl, c->pending_fold.out, c->stream);
    c->pending_fold.n = 0;
}
// hand the pending fold to a kernel about to be launched with `a`; `writes` = bit mask of the partial slots that kernel writes
void take_fold(psgsdf_ctx* c, SweepArgs& a, unsigned writes) {
    a.fold.n = 0;
    if (!c->pending_fold.n) return;
    for (int i = 0; i < c->pending_fold.n; ++i) if (writes & (1u << c->pending_fold.id[i])) { materialize_fold(c); return; }
    a.fold = c->pending_fold; c->pending_fold.n = 0;
}
int read_parts_deferred(psgsdf_ctx* c, const int* slots, int n, std::function<void(const double*)> consume) {
    materialize_fold(c);
    if (c->mbox_used + (size_t)n > c->mbox_n) { int rc = flush(c); if (rc) return rc; }
    const size_t off = c->mbox_used; c->mbox_used += n;
    if (n <= 4 && c->fold_in_next) {
        c->pending_fold.n = n; for (int i = 0; i < n; ++i) c->pending_fold.id[i] = slots[i];
        c->pending_fold.nblk = band_blocks(c); c->pending_fold.out = c->mbox_dev + off;
    } else {
        SlotList sl; sl.n = n; for (int i = 0; i < n; ++i) sl.id[i] = slots[i];
        launch_sum_parts(c->part, c->PB, band_blocks(c), sl, c->mbox_dev + off, c->stream);
    }
    const double* src = c->mbox + off;
    c->deferred.push_back([src, consume] { consume(src); });
    return 0;
}
int read_frame_energy_deferred(psgsdf_ctx* c, int col_e, std::function<void(double, double)> consume) {
    if (c->mbox_used + 2 > c->mbox_n) { int rc = flush(c); if (rc) return rc; }
    const size_t off = c->mbox_used; c->mbox_used += 2;
    launch_frame_cols(c->acc_frame, c->F, col_e, c->mbox_dev + off, c->stream);
    const double* src = c->mbox + off;
    c->deferred.push_back([src, consume] { consume(src[0], src[1]); });
    return 0;
}
// deferred variant without a kernel of its own: the solve kernel that follows the sweep writes the two sums to *dev_slot
int reserve_frame_energy_deferred(psgsdf_ctx* c, std::function<void(double, double)> consume, double** dev_slot) {
    if (c->mbox_used + 2 > c->mbox_n) { int rc = flush(c); if (rc) return rc; }
    const size_t off = c->mbox_used; c->mbox_used += 2;
    *dev_slot = c->mbox_dev + off;
    const double* src = c->mbox + off;
    c->deferred.push_back([src, consume] { consume(src[0], src[1]); });
    return 0;
}
int ensure_host_buf(psgsdf_ctx* c, size_t n) {
    if (n <= c->host_buf_n) return 0;
    if (c->host_buf) hipHostFree(c->host_buf);
    c->host_buf = nullptr; c->host_buf_n = 0;
    HIPCHK(c, hipHostMalloc(&c->host_buf, sizeof(double) * n));
    c->host_buf_n = n;
    return 0;
}

void free_dense(psgsdf_ctx* c) {
    hipFree(c->dense.dist); for (int a = 0; a < 3; ++a) { hipFree(c->dense.g[a]); hipFree(c->dense.rho[a]); }
    hipFree(c->dense.weight); hipFree(c->dense.vis); hipFree(c->dense.row_of); hipFree(c->block_sums);
    c->dense = DenseView{}; c->block_sums = nullptr;
}
int alloc_dense(psgsdf_ctx* c, DenseView& d, long long nvox, int KW, bool with_rowof) {
    HIPCHK(c, hipMalloc(&d.dist, sizeof(float) * nvox));
    for (int a = 0; a < 3; ++a) { HIPCHK(c, hipMalloc(&d.g[a], sizeof(float) * nvox)); HIPCHK(c, hipMalloc(&d.rho[a], sizeof(float) * nvox)); }
    HIPCHK(c, hipMalloc(&d.weight, sizeof(float) * nvox));
    d.KW = KW;
    if (KW > 0) HIPCHK(c, hipMalloc(&d.vis, sizeof(uint64_t) * nvox * KW));
    if (with_rowof) HIPCHK(c, hipMalloc(&d.row_of, sizeof(int) * nvox));
    return 0;
}

// (re)build the band from the dense grid: flags -> scan -> compact planes -> neighbour tables
int build_band(psgsdf_ctx* c) {
    const long long nvox = c->grid.nvox;
    const int KW = c->dense.KW;
    if (!c->block_sums) HIPCHK(c, hipMalloc(&c->block_sums, sizeof(int) * ((nvox + 1023) / 1024 + 1)));
    timed(c, "band_flags", [&] { launch_band_flags(c->dense.dist, c->dense.vis, KW, c->grid.vs, nvox, c->dense.row_of, c->stream); });
    timed(c, "band_scan", [&] { launch_band_scan(c->dense.row_of, nvox, c->block_sums, c->d_total, c->stream); });
    int S = 0;
    HIPCHK(c, hipMemcpyAsync(&S, c->d_total, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int Spad = ((S + kBlock - 1) / kBlock) * kBlock + kBlock;
    // planes (4-byte units per row): see Band
    const size_t n4 = 1 + 1 + 3 + 3 + 6 + 6 + kNQ + 12 + 6 + 14 + kNQ + 4 + 8 + 9 + 1;
    const size_t bytes = (n4 * 4 + (size_t)KW * 8) * Spad + 256;
    if (c->band_mem) { hipFree(c->band_mem); c->band_mem = nullptr; }
    HIPCHK(c, hipMalloc(&c->band_mem, bytes));
    HIPCHK(c, hipMemsetAsync(c->band_mem, 0, bytes, c->stream));
    c->band_bytes = bytes;
    char* p = (char*)c->band_mem;
    auto take = [&](size_t elems, size_t esz) { void* r = p; p += elems * esz * Spad; return r; };
    Band& b = c->band;
    b.S = S; b.Spad = Spad; b.KW = KW;
    b.vis = (uint64_t*)take(KW, 8);
    b.rec[0] = (float4*)take(4, 4); b.rec[1] = (float4*)take(4, 4);
    b.lin = (int*)take(1, 4);
    b.dist = (float*)take(1, 4);
    for (int a = 0; a < 3; ++a) b.g[a] = (float*)take(1, 4);
    for (int a = 0; a < 3; ++a) b.rho[a] = (float*)take(1, 4);
    b.nb = (int*)take(6, 4); b.nbd = (float*)take(6, 4); b.col = (int*)take(kNQ, 4); b.colp = (unsigned*)take(9, 4); b.dirb = (int*)take(1, 4);
    for (int a = 0; a < 3; ++a) b.xs[a] = (float*)take(1, 4);
    for (int a = 0; a < 3; ++a) b.gn[a] = (float*)take(1, 4);
    for (int a = 0; a < 3; ++a) b.gfd[a] = (float*)take(1, 4);
    for (int a = 0; a < 3; ++a) b.nfd[a] = (float*)take(1, 4);
    b.aH = (float*)take(3, 4); b.ab = (float*)take(3, 4);
    b.blk = (float*)take(14, 4); b.H = (float*)take(kNQ, 4);
    b.hx = (int*)take(1, 4);
    b.rhs = (float*)take(1, 4); b.x = (float*)take(1, 4); b.t = (float*)take(1, 4);
    HIPCHK(c, hipMemsetAsync(c->d_total, 0, sizeof(int), c->stream));
    timed(c, "band_fill", [&] { launch_band_fill(c->dense, c->grid, b, c->d_total, c->stream); });
    {   // 16-bit column deltas are usable if the widest reach of any ELL column fits (PSGSDF_PCG_COL16=0 forces the 32-bit table)
        int reach = 0;
        HIPCHK(c, hipMemcpyAsync(&reach, c->d_total, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        b.col16 = reach <= 32767;
        if (const char* e = getenv("PSGSDF_PCG_COL16")) if (atoi(e) == 0) b.col16 = 0;
    }
    if (c->areg_mem) { hipFree(c->areg_mem); c->areg_mem = nullptr; c->ar = AlbedoReg{}; }
    if (c->reg_r != 0.f) {   // "reg albedo": stencil tables + matrix-free CG vectors over the 3S unknowns
        if (c->n_ranks > 1) return fail(c, PSGSDF_ERR_UNSUPPORTED, "reg albedo is single-rank only");
        const size_t planes = 3 + 1 + 9 + 12 + 3 + 8 * 3;
        HIPCHK(c, hipMalloc(&c->areg_mem, planes * 4 * (size_t)Spad));
        HIPCHK(c, hipMemsetAsync(c->areg_mem, 0, planes * 4 * (size_t)Spad, c->stream));
        char* q = (char*)c->areg_mem;
        auto tk = [&](size_t n) { void* r = q; q += n * 4 * (size_t)Spad; return r; };
        AlbedoReg& ar = c->ar;
        ar.anb = (int*)tk(3); ar.back = (int*)tk(1); ar.anrho = (float*)tk(9); ar.J = (float*)tk(12); ar.res = (float*)tk(3);
        ar.rhs = (float*)tk(3); ar.diag = (float*)tk(3); ar.diag0 = (float*)tk(3); ar.x = (float*)tk(3); ar.r = (float*)tk(3); ar.p = (float*)tk(3); ar.q = (float*)tk(3); ar.t = (float*)tk(3);
        SweepArgs at{}; at.b = b; at.ar = ar;
        launch_areg_tables(c->dense, c->grid, at, c->stream);
    }
    // row partition: equal band count per rank = z-slabs (the band is sorted by linear index, z slowest)
    {
        const int C = (S + c->n_ranks - 1) / c->n_ranks;
        c->row0 = std::min(S, c->rank * C); c->row1 = std::min(S, c->row0 + C);
        c->halo = 0; c->need[0] = c->need[1] = 0;
        if (c->n_ranks > 1 && c->row1 <= c->row0) return fail(c, PSGSDF_ERR_UNSUPPORTED, "rank %d of %d would own no band rows (band of %d): use fewer ranks", c->rank, c->n_ranks, S);
        if (c->n_ranks > 1 && S > 0) {
            HIPCHK(c, hipMemsetAsync(c->d_need, 0, 2 * sizeof(int), c->stream));
            launch_reach(b, c->row0, c->row1, c->d_need, c->stream);
            HIPCHK(c, hipMemcpyAsync(c->need, c->d_need, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->halo = std::max(c->need[0], c->need[1]);
            if (c->halo > C) return fail(c, PSGSDF_ERR_UNSUPPORTED, "slab of %d rows is thinner than the stencil reach %d: use fewer ranks", C, c->halo);
        }
    }
    // per-frame observation lists of the owned rows (counts -> host prefix -> fill)
    {
        const int nch = (c->row1 - c->row0 + kObsChunk - 1) / kObsChunk, F = c->F;
        if (c->obs_mem) { hipFree(c->obs_mem); c->obs_mem = nullptr; }
        b.obs_ptr = nullptr; b.obs_rows = nullptr; b.obs_max = 0;
        if (nch > 0 && F > 0) {
            int* d_counts = nullptr;
            HIPCHK(c, hipMalloc(&d_counts, sizeof(int) * (size_t)nch * F));
            launch_obs_count(b, F, c->row0, c->row1, d_counts, c->stream);
            std::vector<int> cnt((size_t)nch * F), off((size_t)nch * F), ptr(F + 1);
            HIPCHK(c, hipMemcpyAsync(cnt.data(), d_counts, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            int run = 0, mx = 0;
            for (int f = 0; f < F; ++f) { ptr[f] = run; for (int k = 0; k < nch; ++k) { off[(size_t)f * nch + k] = run; run += cnt[(size_t)f * nch + k]; } mx = std::max(mx, run - ptr[f]); }
            ptr[F] = run;
            HIPCHK(c, hipMalloc(&c->obs_mem, sizeof(int) * ((size_t)run + F + 2)));
            b.obs_ptr = (int*)c->obs_mem; b.obs_rows = b.obs_ptr + (F + 1); b.obs_max = mx;
            HIPCHK(c, hipMemcpyAsync(b.obs_ptr, ptr.data(), sizeof(int) * (F + 1), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(d_counts, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice, c->stream));
            launch_obs_fill(b, F, c->row0, c->row1, d_counts, c->stream);
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipFree(d_counts);
        }
    }
    if (c->part) { hipFree(c->part); c->part = nullptr; }
    c->PB = Spad / kBlock + 1;
    HIPCHK(c, hipMalloc(&c->part, sizeof(double) * SC_COUNT * c->PB));
    HIPCHK(c, hipMemsetAsync(c->part, 0, sizeof(double) * SC_COUNT * c->PB, c->stream));
    {
        const size_t need = 4096;
        if (need > c->mbox_n) {
            if (c->mbox) hipHostFree(c->mbox);
            c->mbox = nullptr; c->mbox_n = 0;
            HIPCHK(c, hipHostMalloc(&c->mbox, sizeof(double) * need, hipHostMallocMapped));
            HIPCHK(c, hipHostGetDevicePointer((void**)&c->mbox_dev, c->mbox, 0));
            c->mbox_n = need;
        }
        c->mbox_used = 0; c->deferred.clear();
    }
    return ensure_host_buf(c, (size_t)SC_COUNT * c->PB + (size_t)c->F * kFrameRow + 4096 + 64);
}

int derive(psgsdf_ctx* c, int update_grad) {
    SweepArgs a = make_args(c, 0);
    timed(c, "derive", [&] { launch_derive(a, update_grad, c->stream); });
    const int slots[2] = {SC_EN, SC_EL}; double s[2];
    int rc = read_parts(c, slots, 2, s); if (rc) return rc;
    c->en_sum = s[0]; c->el_sum = s[1];
    return 0;
}

inline double band_mean(const psgsdf_ctx* c, double sum) { return c->band.S ? sum / (double)c->band.S : 0.0; }
inline float total_energy(const psgsdf_ctx* c, float E, float E_n, float E_l, float E_r = 0.f) { return E + c->reg_n * E_n + c->reg_l * E_l + c->reg_r * E_r; }   // OptimizerAux.cpp:261

int ps_energy(psgsdf_ctx* c, double* E, int64_t* nobs) {
    SweepArgs a = make_args(c, 0);
    take_fold(c, a, (1u << SC_ENERGY) | (1u << SC_NOBS) | (1u << SC_AUX0) | (1u << SC_AUX1) | (1u << SC_AUX2));
    timed(c, "energy", [&] { launch_energy(a, c->stream); });
    const int slots[2] = {SC_ENERGY, SC_NOBS}; double s[2];
    int rc = read_parts(c, slots, 2, s); if (rc) return rc;
    *E = band_mean(c, s[0]); if (nobs) *nobs = (int64_t)s[1];
    return 0;
}

// Launch shape of the fused PCG pass.  The pass is a chain of memory round trips per workgroup (coefficients + column
// indices -> two batches of record gathers, the second overlapping the reduction), so what matters is how many rows have
// their loads in flight at once.  One row per thread at 114 VGPRs keeps 4 waves per SIMD resident (1024 workgroups); the
// reduction of the previous pass's partials costs every workgroup G x 7 doubles, which caps G at 768.  Measured on the
// 256^3 band (1317 row-blocks): 659 workgroups x 2 trips 17.3 us, 768 x 2 trips 18.1 us, 512 x 3 trips 18.8 us
// (tools/pcg_ablate.py, profiles/r01_notes.md).
static void cgf_shape(int nblk, int* G, int* rows) {
    nblk = std::max(1, nblk);
    int r = 1, cap = kCgfMaxBlocks;
    if (const char* e = getenv("PSGSDF_PCG_ROWS")) { int v = atoi(e); if (v >= 1 && v <= 2) r = v; }       // tuning knobs
    if (const char* e = getenv("PSGSDF_PCG_BLOCKS")) { int v = atoi(e); if (v > 0 && v <= kCgfMaxBlocks) cap = v; }
    const int per = (nblk + r - 1) / r;                 // workgroups if every thread took r rows once
    const int trips = (per + cap - 1) / cap;
    *G = (per + trips - 1) / trips; *rows = r;
}

// Fused PCG (pcg.hip: k_cgf_pass): kernel k finishes pass k-1 and runs pass k, so a chunk of n kernels tells the host
// about the passes up to k0+n-2; the kernel that detects convergence (or hits the cap) is also the one that finalises x.
// `tail(gate)`, if given, enqueues what follows a finished solve (distance update + regrad) right behind every chunk of passes,
// gated on the device-side 'solve finished' flag: when the chunk converges -- the normal case -- the GPU runs it without
// waiting for the host to notice; when it does not, the gated kernels do nothing and the tail is enqueued again behind the
// next chunk.  *tail_ran tells the caller whether the enqueued tail is the one that took effect.
int pcg_solve(psgsdf_ctx* c, const SweepArgs& a, int* iters_out, int* success_out, double* err_out,
              const std::function<void(const double*)>& tail = nullptr, bool gate_on_converged = true, bool* tail_ran = nullptr) {
    const int S = c->band.S;
    if (tail_ran) *tail_ran = false;
    if (a.row1 <= a.row0) { *iters_out = 0; *success_out = 1; *err_out = 0; c->last_cg_iters = 0; return 0; }   // empty band: b = 0, x = 0, Success
    int cap = c->set.cg_max_it > 0 ? c->set.cg_max_it : 2 * S;
    if (cap > c->pcg_cap) cap = c->pcg_cap;
    int G, rows;
    cgf_shape(band_blocks(c), &G, &rows);
    timed(c, "pcg_init", [&] { launch_cgf_init(a, c->pcg_sc, c->pcg_part, G, c->stream); });
    // first chunk sized from the previous solve (the count is stable between Gauss-Newton iterations)
    int chunk = std::min(64, std::max(4, c->last_cg_iters + 2));
    int k = 0, iters = -1;            // k = next kernel index; kernels 0..cap exist (kernel cap only finalises)
    float rhsN = 0, rn2_last = 0, threshold = 0;
    while (true) {
        const int n = std::min(chunk, cap + 1 - k);
        if (c->mbox_used + (size_t)n > c->mbox_n) { int rc = flush(c); if (rc) return rc; }
        const size_t off = c->mbox_used; c->mbox_used += n;
        volatile double* st = c->mbox + off;
        for (int q = 0; q < n; ++q) st[q] = NAN;       // "not published yet" (kernel q of the chunk overwrites its slot)
        for (int q = 0; q < n; ++q)
            timed(c, "pcg_pass", [&] { launch_cgf_pass(a, c->pcg_sc, c->pcg_part, G, rows, k + q, cap, c->mbox_dev + off + q, c->stream); });
        if (tail && c->pcg_poll && !c->profiling) { tail(c->pcg_sc + (gate_on_converged ? 2 : 1)); if (tail_ran) *tail_ran = true; }
        // Watch the mapped slots instead of waiting for the stream to drain: the kernel that detects convergence publishes
        // at its START, so the host learns the outcome while that kernel and the surplus (no-op) kernels of the chunk are
        // still running, and enqueues the rest of the iteration behind them without a bubble.
        bool drained = !c->pcg_poll;
        if (drained) { int rc = flush(c); if (rc) return rc; }
        for (int q = 0; q < n && iters < 0; ++q) {
            const int kk = k + q;
            while (!drained && std::isnan(st[q])) {
                if (hipStreamQuery(c->stream) == hipSuccess) drained = true;   // nothing left that could publish
            }
            const double v = st[q];
            if (std::isnan(v)) return fail(c, PSGSDF_ERR_DEVICE, "PCG kernel %d published nothing", kk);
            if (kk == 0) {
                rhsN = (float)v;
                if (rhsN == 0.f) { iters = 0; break; }
                threshold = fmaxf(FLT_EPSILON * FLT_EPSILON * rhsN, FLT_MIN);
                rn2_last = rhsN;
                continue;
            }
            rn2_last = (float)v;                           // |r|^2 after pass kk-1
            if (rn2_last < threshold) iters = kk - 1;      // Eigen breaks before ++i
            else if (kk == cap) iters = cap;
        }
        if (!drained && c->pending_fold.n) { int rc = flush(c); if (rc) return rc; drained = true; }   // (cannot happen: assemble took it)
        if (!drained) {   // every deferred read-back enqueued before the chunk has landed (in-order stream): deliver them
            for (auto& f : c->deferred) f();
            c->deferred.clear(); c->mbox_used = 0;
        }
        if (rhsN == 0.f) { *iters_out = 0; *success_out = 1; *err_out = 0; c->last_cg_iters = 0; return 0; }
        if (iters >= 0) break;
        k += n;
        chunk = 4;
    }
    double err = sqrt((double)rn2_last / (double)rhsN);
    *iters_out = iters; *err_out = err; *success_out = err <= (double)FLT_EPSILON;
    c->last_cg_iters = iters;
    return 0;
}

// "reg albedo": mean over the band of sum_c ||grad rho_c|| (Optimizer.cpp:122-136); also refreshes the Jacobian planes
int albedo_reg_energy(psgsdf_ctx* c, double* Er) {
    SweepArgs a = make_args(c, 0);
    launch_areg_build(a, c->stream);
    const int slots[1] = {SC_AUX0}; double s[1];
    int rc = read_parts(c, slots, 1, s); if (rc) return rc;
    c->er_sum = s[0]; *Er = band_mean(c, s[0]);
    return 0;
}
// optimizeAlbedoAll with the regulariser (PsOptimizer.cpp:85-121): Eigen ConjugateGradient over the 3S unknowns on
// H = H_d + reg_rho Jr^T Jr applied matrix-free (albedo_reg.hip).  Host-driven, two read-backs per CG iteration: no shipped
// configuration enables this term.  The step is left in ar.x.
int albedo_reg_solve(psgsdf_ctx* c, const SweepArgs& a, int* iters_out, int* ok_out, double* err_out) {
    const AlbedoReg& ar = a.ar;
    launch_areg_build(a, c->stream);
    launch_areg_system(a, c->stream);
    launch_areg_cg_init(a, c->stream);
    const int two[2] = {SC_AUX0, SC_AUX1}, one[1] = {SC_AUX0}; double s[2];
    int rc = read_parts(c, two, 2, s); if (rc) return rc;
    const float rhsN = (float)s[0];
    *iters_out = 0; *ok_out = 1; *err_out = 0;
    if (rhsN == 0.f) return 0;
    const float thr = fmaxf(FLT_EPSILON * FLT_EPSILON * rhsN, FLT_MIN);
    float res2 = rhsN, absNew = (float)s[1];
    const int maxIters = c->set.cg_max_it > 0 ? c->set.cg_max_it : 6 * c->band.S;
    int i = 0;
    if (res2 >= thr) {
        while (i < maxIters) {
            launch_areg_jx(a, ar.p, ar.t, c->stream);
            launch_areg_jt(a, ar.p, ar.t, ar.q, c->stream);
            if ((rc = read_parts(c, one, 1, s))) return rc;
            const float alpha = absNew / (float)s[0];
            launch_areg_cg_update(a, alpha, c->stream);
            if ((rc = read_parts(c, two, 2, s))) return rc;
            res2 = (float)s[0];
            if (res2 < thr) break;
            const float absOld = absNew; absNew = (float)s[1];
            launch_areg_cg_dir(a, absNew / absOld, c->stream);
            ++i;
        }
    }
    *iters_out = i; *err_out = sqrt((double)res2 / (double)rhsN); *ok_out = *err_out <= (double)FLT_EPSILON;
    return 0;
}

// A sub-step in two halves so that the alternation loop can look at the energy of the state a sweep started from
// (= the energy AFTER the previous block, PsOptimizer.cpp:311,323,338,354) before anything is modified:
//   step_begin : the sweep (normal equations + PS energy of the input state)          -> st->e_in, st->n_obs
//   step_finish: solve + update (albedo apply / light, pose solves / distance PCG + apply + regrad)
int step_begin(psgsdf_ctx* c, int block, int laplacian_reg, psgsdf_step_stats* st, std::function<void(double, double)> deferred_consumer = nullptr) {
    memset(st, 0, sizeof(*st));
    st->block = block;
    SweepArgs a = make_args(c, laplacian_reg);
    const bool led = c->set.model == PSGSDF_LED;
    double e_sum = 0, nobs = 0;
    int rc;
    switch (block) {
        case PSGSDF_ALBEDO: case PSGSDF_DIST: {
            if (block == PSGSDF_ALBEDO) { take_fold(c, a, (1u << SC_ENERGY) | (1u << SC_NOBS)); timed(c, "sweep_albedo", [&] { launch_sweep_albedo(a, c->stream); }); }
            else timed(c, "sweep_dist", [&] { launch_sweep_dist(a, c->stream); });
            const int slots[2] = {SC_ENERGY, SC_NOBS}; double s[2];
            if (deferred_consumer) return read_parts_deferred(c, slots, 2, [deferred_consumer](const double* v) { deferred_consumer(v[0], v[1]); });
            if ((rc = read_parts(c, slots, 2, s))) return rc;
            e_sum = s[0]; nobs = s[1];
            break;
        }
        case PSGSDF_LIGHT: case PSGSDF_POSE: {
            int col;   // the frame accumulator is all-zero here: whoever consumed it last cleared it (sweeps.hip: frame_rows_finish)
            if (block == PSGSDF_LIGHT) {
                take_fold(c, a, 0u);
                timed(c, "sweep_light", [&] { launch_sweep_light(a, c->stream); });
                const int n = led ? 3 : (c->set.model == PSGSDF_SH2 ? 9 : 4), nh = led ? 3 : n * (n + 1) / 2;
                col = nh + n;
            } else { take_fold(c, a, 0u); timed(c, "sweep_pose", [&] { launch_sweep_pose(a, c->stream); }); col = 27; }
            if (deferred_consumer) return reserve_frame_energy_deferred(c, deferred_consumer, &c->frame_e_slot);   // filled by the solve kernel
            if ((rc = read_frame_energy(c, col, &e_sum, &nobs))) return rc;
            break;
        }
        default: return fail(c, PSGSDF_ERR_ARG, "unknown block %d", block);
    }
    st->e_in = band_mean(c, e_sum);
    st->n_obs = (int64_t)nobs;
    return 0;
}
int step_finish(psgsdf_ctx* c, int block, int laplacian_reg, psgsdf_step_stats* st, bool defer_reg_sums = false) {
    SweepArgs a = make_args(c, laplacian_reg);
    const bool led = c->set.model == PSGSDF_LED;
    int rc;
    switch (block) {
        case PSGSDF_ALBEDO: {
            const int slots[1] = {SC_ACCEPT}; double s[1];
            if (c->reg_r != 0.f) {
                int iters = 0, ok = 1; double err = 0;
                if ((rc = albedo_reg_solve(c, a, &iters, &ok, &err))) return rc;
                const int apply = (led || ok) ? 1 : 0;      // PsOptimizer.cpp:117-119 (only on success) / LedOptimizer.cpp:195 (always)
                if (apply) timed(c, "apply_albedo", [&] { launch_apply_albedo_delta(a, c->ar.x, c->stream); });
                if (apply && c->want_counts) { if ((rc = read_parts(c, slots, 1, s))) return rc; st->n_accepted = (int64_t)s[0]; }
                st->cg_iters = iters; st->cg_converged = ok; st->cg_error = err; st->applied = apply;
                break;
            }
            take_fold(c, a, 1u << SC_ACCEPT);
            timed(c, "apply_albedo", [&] { launch_apply_albedo(a, c->stream); });
            if (c->want_counts) { if ((rc = read_parts(c, slots, 1, s))) return rc; st->n_accepted = (int64_t)s[0]; }
            st->cg_iters = 1; st->cg_converged = 1; st->applied = 1;
            break;
        }
        case PSGSDF_LIGHT:
            timed(c, "solve_light", [&] { launch_solve_light(a, c->frames, c->led_light, c->frame_e_slot, c->stream); });
            c->frame_e_slot = nullptr;
            st->cg_converged = 1; st->applied = 1; st->n_accepted = led ? 1 : c->F;
            break;
        case PSGSDF_POSE:
            timed(c, "solve_pose", [&] { launch_solve_pose(a, c->frames, c->frame_e_slot, c->stream); });
            c->frame_e_slot = nullptr;
            st->cg_converged = 1; st->applied = 1; st->n_accepted = c->F;
            break;
        case PSGSDF_DIST: {
            take_fold(c, a, 0u);
            timed(c, "assemble", [&] { launch_assemble(a, c->stream); });
            int iters = 0, ok = 1; double err = 0;
            const bool only_on_success = !led && c->set.ref_quirks;   // PsOptimizer.cpp:168-170 (B8): SH skips the update unless the solve reports Success
            bool tail_ran = false;
            auto tail = [&](const double* gate) {                     // distance update + regrad, gated on the device-side outcome of the solve
                SweepArgs ag = a; ag.fold.n = 0; ag.gate = gate;
                timed(c, "apply_dist", [&] { launch_apply_dist(ag, c->stream); });
                SweepArgs a2 = make_args(c, 0); a2.gate = gate;
                timed(c, "derive", [&] { launch_derive(a2, 1, c->stream); });
            };
            if ((rc = pcg_solve(c, a, &iters, &ok, &err, tail, only_on_success, &tail_ran))) return rc;
            int apply = 1;
            if (only_on_success && !ok) apply = 0;
            if (apply) {
                if (!tail_ran) tail(nullptr);
                // regrad + Eikonal / Laplacian sums; one read-back for the accepted count and the two sums
                const int slots[3] = {SC_ACCEPT, SC_EN, SC_EL}; double s[3];
                if (defer_reg_sums) { if ((rc = read_parts_deferred(c, slots, 3, [c](const double* v) { c->en_sum = v[1]; c->el_sum = v[2]; }))) return rc; }
                else { if ((rc = read_parts(c, slots, 3, s))) return rc; st->n_accepted = (int64_t)s[0]; c->en_sum = s[1]; c->el_sum = s[2]; }
            }
            st->cg_iters = iters; st->cg_converged = ok; st->cg_error = err; st->applied = apply;
            break;
        }
        default: return fail(c, PSGSDF_ERR_ARG, "unknown block %d", block);
    }
    return 0;
}
int do_step(psgsdf_ctx* c, int block, int laplacian_reg, psgsdf_step_stats* st) {
    psgsdf_step_stats tmp; if (!st) st = &tmp;
    int rc = step_begin(c, block, laplacian_reg, st); if (rc) return rc;
    return step_finish(c, block, laplacian_reg, st);
}

// The alternation loop (PsOptimizer.cpp:303-425 / LedOptimizer.cpp:343-475).  The reference evaluates getPSEnergy after
// every block; here the energy after block k is the `e_in` of block k+1's sweep, and the energy that closes iteration i
// is the `e_in` of the FIRST sweep of iteration i+1, which runs before anything of iteration i+1 is applied -- so the
// convergence / divergence exit still leaves exactly the state the reference would leave.  Only the last iteration
// (and the one that triggers the 2x refinement) needs a stand-alone energy sweep.
int do_upsample(psgsdf_ctx* c);
struct LoopState { float E, E_n, E_l, E_prev; int laplacian_reg; float E_r; };

// closes record `rec` of an iteration with the PS energy E that followed its last block
void close_iteration(psgsdf_ctx* c, LoopState& L, psgsdf_iter_stats* rec, int pending_slot, float E, bool early_exit_semantics) {
    L.E = E;
    if (pending_slot >= 0) rec->e_after[pending_slot] = (double)E;
    rec->e_n = L.E_n; rec->e_l = L.E_l; rec->e_r = L.E_r;
    rec->e_total = (double)total_energy(c, L.E, L.E_n, L.E_l, L.E_r);
    rec->reg_weight_n = c->reg_n; rec->reg_weight_l = c->reg_l;
    float Et = (float)rec->e_total;
    rec->rel_diff = (double)(fabsf(L.E_prev - Et) / L.E_prev);
    rec->converged = rec->rel_diff < (double)c->set.conv_threshold;
    rec->diverged = early_exit_semantics ? (!rec->converged && (L.E_prev < Et)) : (L.E_prev < Et);
}

// Runs iterations [first, ...) until max_iters or (if stop_early) convergence / divergence.  `on_iter`, upsampling and
// the Laplacian schedule only apply when `full` (psgsdf_optimize).
int run_loop(psgsdf_ctx* c, int flags, LoopState& L, int max_iters, bool full, psgsdf_iter_stats* stats, int stats_cap, int* n_done, int* result,
             psgsdf_iter_cb on_iter, void* user) {
    const bool led = c->set.model == PSGSDF_LED;
    const int order[4] = {led ? PSGSDF_LIGHT : PSGSDF_ALBEDO, led ? PSGSDF_ALBEDO : PSGSDF_LIGHT, PSGSDF_DIST, PSGSDF_POSE};
    psgsdf_iter_stats rec; memset(&rec, 0, sizeof(rec));
    psgsdf_iter_stats prev; int prev_slot = -1; bool have_prev = false;   // iteration waiting for its closing energy
    struct CountsOff { psgsdf_ctx* c; bool old; CountsOff(psgsdf_ctx* c_) : c(c_), old(c_->want_counts) { c->want_counts = false; } ~CountsOff() { c->want_counts = old; } } counts_off(c);
    int done = 0, iter = 0; if (result) *result = 0;
    bool stop = false;
    auto finalize = [&](psgsdf_iter_stats& r, int it) -> int {   // everything that happens after E_total(it) is known
        const bool term = full && (r.converged || r.diverged);
        float E_last = (float)r.e_total;
        if (full && !term && it == 5 && c->set.upsample) {   // PsOptimizer.cpp:386-409
            if (c->reg_l == 0.0f) c->reg_l = 1.0f;
            L.laplacian_reg = 1;
            int rc = do_upsample(c); if (rc) return rc;
            L.E_l = (float)band_mean(c, c->el_sum);
            c->reg_l *= L.E / L.E_l;
            E_last = total_energy(c, L.E, L.E_n, L.E_l, L.E_r);
            r.upsampled = 1;
        }
        if (full && !term && c->set.upsample && (led ? it == 15 : it > 15)) c->reg_l = 0.0f;   // PsOptimizer.cpp:411-413 / LedOptimizer.cpp:461-463
        L.E_prev = E_last;
        if (stats && done < stats_cap) stats[done] = r;
        done++;
        if (term) { if (result && r.converged) *result = 1; stop = true; return 0; }
        if (full && on_iter && on_iter(user, it + 1, &r)) stop = true;
        return 0;
    };
    // per-iteration values that arrive through deferred read-backs (stable addresses: two alternating slots)
    struct Late { double e_in[4]; int blk_of[4]; int n; bool dist_ran; int cg_iters; bool alb_reg; float e_r; } late[2];
    int li = 0;
    auto apply_late = [&](psgsdf_iter_stats& r, const Late& lt, int first_slot_pending) {
        // e_in of sweep q is the energy AFTER the block that ran before it in the same iteration
        int pend = first_slot_pending;
        for (int q = 0; q < lt.n; ++q) {
            if (pend >= 0 && q > 0) { r.e_after[pend] = (double)(float)band_mean(c, lt.e_in[q]); }   // deferred values are raw sums
            pend = lt.blk_of[q] == PSGSDF_ALBEDO ? 0 : lt.blk_of[q] == PSGSDF_LIGHT ? 1 : lt.blk_of[q] == PSGSDF_DIST ? 2 : 3;
        }
        if (lt.alb_reg) L.E_r = lt.e_r;
        if (lt.dist_ran) {
            r.cg_iters = lt.cg_iters;
            if (c->reg_n != 0.f) L.E_n = (float)band_mean(c, c->en_sum);
            if (L.laplacian_reg) L.E_l = (float)band_mean(c, c->el_sum);
        }
    };
    Late* prev_late = nullptr;
    double* prev_close = nullptr;   // where the lazily delivered closing energy of `prev` will appear
    while (iter < max_iters && !stop) {
        memset(&rec, 0, sizeof(rec));
        for (int q = 0; q < 4; ++q) rec.e_after[q] = NAN;
        Late& lt = late[li]; lt.n = 0; lt.dist_ran = false; lt.cg_iters = 0; lt.alb_reg = false; lt.e_r = 0.f;
        int pending = -1;
        for (int q = 0; q < 4 && !stop; ++q) {
            const int blk = order[q];
            if (!(flags & blk)) continue;
            psgsdf_step_stats st;
            const int qi = lt.n;
            lt.blk_of[qi] = blk; lt.e_in[qi] = NAN; lt.n++;
            if (have_prev && full) {   // synchronous: this sweep's input energy closes the previous iteration (stop decision)
                int rc = step_begin(c, blk, L.laplacian_reg, &st); if (rc) return rc;   // (flushes every deferred read of the previous iteration)
                lt.e_in[qi] = st.e_in;
                apply_late(prev, *prev_late, -1);
                close_iteration(c, L, &prev, prev_slot, (float)st.e_in, full);
                have_prev = false;
                if ((rc = finalize(prev, iter - 1))) return rc;
                if (stop) {        // converged / diverged / aborted: nothing of this iteration has been applied
                    if (blk == PSGSDF_LIGHT || blk == PSGSDF_POSE) launch_zero_f64(c->acc_frame, (int)c->acc_frame_n, c->stream);   // the sweep's rows stay unconsumed
                    break;
                }
            } else {
                // no stop decision pending (psgsdf_iterate never exits early): even the closing energy of the previous
                // iteration is delivered lazily, at the next host sync (the PCG status read of this iteration)
                double* slot_e = &lt.e_in[qi];
                int rc = step_begin(c, blk, L.laplacian_reg, &st, [slot_e](double e_sum, double) { *slot_e = e_sum; }); if (rc) return rc;
                if (have_prev && prev_close == nullptr) prev_close = slot_e;
            }
            int rc = step_finish(c, blk, L.laplacian_reg, &st, true); if (rc) return rc;
            if (blk == PSGSDF_ALBEDO && c->reg_r != 0.f) { double er; if ((rc = albedo_reg_energy(c, &er))) return rc; lt.alb_reg = true; lt.e_r = (float)er; }   // PsOptimizer.cpp:312 (enters L when the record closes)
            if (blk == PSGSDF_DIST) { lt.dist_ran = true; lt.cg_iters = st.cg_iters; }
            if (have_prev && prev_close && !std::isnan(*prev_close)) {   // the lazy closing energy has arrived
                apply_late(prev, *prev_late, -1);
                close_iteration(c, L, &prev, prev_slot, (float)band_mean(c, *prev_close), full);
                have_prev = false; prev_close = nullptr;
                if ((rc = finalize(prev, iter - 1))) return rc;
            }
            pending = blk == PSGSDF_ALBEDO ? 0 : blk == PSGSDF_LIGHT ? 1 : blk == PSGSDF_DIST ? 2 : 3;
        }
        if (stop) break;
        if (have_prev && prev_close) {   // still open (no host sync happened during this iteration): force one
            int rc = flush(c); if (rc) return rc;
            apply_late(prev, *prev_late, -1);
            close_iteration(c, L, &prev, prev_slot, (float)band_mean(c, *prev_close), full);
            have_prev = false; prev_close = nullptr;
            if ((rc = finalize(prev, iter - 1))) return rc;
        }
        // deferred e_in values are raw sums (not yet divided by S) except the synchronous first one: normalise on use
        const bool last = iter + 1 >= max_iters;
        const bool refine_next = full && c->set.upsample && iter == 5;
        if (pending >= 0 && !last && !refine_next) { prev = rec; prev_slot = pending; have_prev = true; prev_late = &lt; li ^= 1; }
        else {
            double e; int rc = ps_energy(c, &e, nullptr); if (rc) return rc;   // flushes the deferred reads of this iteration
            apply_late(rec, lt, -1);
            close_iteration(c, L, &rec, pending, (float)e, full);
            if ((rc = finalize(rec, iter))) return rc;
        }
        ++iter;
    }
    if (have_prev && !stop) {   // loop ended by max_iters while an iteration was still open (cannot happen: `last` closes it)
        double e; int rc = ps_energy(c, &e, nullptr); if (rc) return rc;
        apply_late(prev, *prev_late, -1);
        close_iteration(c, L, &prev, prev_slot, (float)e, full);
        if ((rc = finalize(prev, iter - 1))) return rc;
    }
    if (n_done) *n_done = done;
    return 0;
}

int do_upsample(psgsdf_ctx* c) {
    // bring the dense grid up to date, refine, rebuild the band
    timed(c, "band_scatter", [&] { launch_band_scatter(c->dense, c->band, c->stream); });
    DenseView nd{};
    const long long nn = 8 * c->grid.nvox;
    int rc = alloc_dense(c, nd, nn, c->dense.KW, true); if (rc) return rc;
    timed(c, "upsample", [&] { launch_upsample(c->dense, nd, c->grid, c->stream); });
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_dense(c);
    c->dense = nd;
    if (c->vis_seq) { hipFree(c->vis_seq); c->vis_seq = nullptr; }
    GridP& g = c->grid;
    g.vs *= 0.5f;
    for (int a = 0; a < 3; ++a) g.dim[a] *= 2;
    for (int a = 0; a < 3; ++a) g.origin[a] = c->shift[a] - (float)(0.5 * (double)g.vs) * (float)g.dim[a] - (float)(0.5 * (double)g.vs) * 1.0f;   // VoxelGrid.h:143-149
    g.nvox = nn;
    g.vs_inv = (float)(1.0 / (double)g.vs);
    if ((rc = build_band(c))) return rc;
    return derive(c, 0);
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" {

const char* psgsdf_version(void) { return "psgsdf-hip gfx950 r1"; }
const char* psgsdf_last_error(const psgsdf_ctx* c) { return c ? c->err : "null context"; }

int psgsdf_create(const psgsdf_grid_desc* grid, const float K[9], const psgsdf_settings* settings, int device, psgsdf_ctx** out) {
    if (!grid || !K || !settings || !out) return PSGSDF_ERR_ARG;
    if (settings->model < 0 || settings->model > 2) return PSGSDF_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PSGSDF_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PSGSDF_ERR_DEVICE;
    psgsdf_ctx* c = new psgsdf_ctx();
    c->device = device;
    if (const char* e = getenv("PSGSDF_PCG_POLL")) c->pcg_poll = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_FOLD_IN_NEXT")) c->fold_in_next = atoi(e) != 0;
    c->set = *settings; c->reg_n = settings->reg_weight_n; c->reg_l = settings->reg_weight_l; c->reg_r = settings->reg_weight_rho;
    GridP& g = c->grid;
    for (int a = 0; a < 3; ++a) { g.dim[a] = grid->dim[a]; c->shift[a] = grid->shift[a]; }
    g.nvox = (long long)g.dim[0] * g.dim[1] * g.dim[2];
    g.vs = grid->voxel_size; g.vs_inv = 1.f / g.vs; g.T = grid->truncation;
    for (int a = 0; a < 3; ++a) g.origin[a] = c->shift[a] - (float)(0.5 * (double)g.vs) * (float)g.dim[a];   // VoxelGrid.h:130
    c->cam.fx = K[0]; c->cam.fy = K[4]; c->cam.cx = K[2]; c->cam.cy = K[5];
    bool ok = hipStreamCreate(&c->stream) == hipSuccess
        && hipMalloc(&c->pcg_sc, sizeof(double) * (16 + 8 * (size_t)kPcgMaxBlocks)) == hipSuccess   // fs[0..1] + stage stamps of the timing hook
        && hipMalloc(&c->pcg_part, sizeof(double) * 14 * kPcgMaxBlocks) == hipSuccess
        && hipMalloc(&c->mg_scal, sizeof(double) * kMgScal) == hipSuccess && hipMalloc(&c->mg_ext, sizeof(double) * 8) == hipSuccess
        && hipMalloc(&c->mg_hist, sizeof(double) * ((size_t)c->pcg_cap + 2)) == hipSuccess && hipMalloc(&c->d_need, 2 * sizeof(int)) == hipSuccess
        && hipMalloc(&c->mg_slots, sizeof(int) * 8) == hipSuccess
        && hipMalloc(&c->d_total, sizeof(int)) == hipSuccess
        && hipMalloc(&c->led_light, sizeof(float) * 3) == hipSuccess
        && hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess;
    if (!ok) { delete c; return PSGSDF_ERR_DEVICE; }
    *out = c;
    return PSGSDF_OK;
}

void psgsdf_destroy(psgsdf_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    free_dense(c);
    hipFree(c->vis_seq); hipFree(c->frame_idx); hipFree(c->img); hipFree(c->frames); hipFree(c->led_light);
    hipFree(c->band_mem); hipFree(c->obs_mem); hipFree(c->stage);
    hipFree(c->ncache); hipFree(c->ntmp); hipFree(c->nout); hipFree(c->ndepth); hipFree(c->track_part); if (c->track_host) hipHostFree(c->track_host); hipFree(c->acc_frame); hipFree(c->part); hipFree(c->pcg_sc); hipFree(c->pcg_part); hipFree(c->d_total);
    if (c->host_buf) hipHostFree(c->host_buf);
    if (c->mbox) hipHostFree(c->mbox);
    if (c->ev0) hipEventDestroy(c->ev0); if (c->ev1) hipEventDestroy(c->ev1);
    for (auto& pr : c->watch_pool) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    hipFree(c->areg_mem); hipFree(c->mg_scal); hipFree(c->mg_ext); hipFree(c->mg_hist); hipFree(c->d_need); hipFree(c->mg_slots);
    if (c->stream && c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

int psgsdf_upload_volume(psgsdf_ctx* c, const float* dist, const float* grad_xyz, const float* weight, const float* rgb, const uint64_t* vis_words, int words_per_voxel) {
    if (!c || !dist || !grad_xyz || !weight || !rgb || !vis_words || words_per_voxel < 1) return fail(c, PSGSDF_ERR_ARG, "upload_volume: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    const long long n = c->grid.nvox;
    free_dense(c);
    if (c->vis_seq) { hipFree(c->vis_seq); c->vis_seq = nullptr; }
    int rc = alloc_dense(c, c->dense, n, 0, true); if (rc) return rc;
    HIPCHK(c, hipMalloc(&c->vis_seq, sizeof(uint64_t) * n * words_per_voxel));
    c->wpv_seq = words_per_voxel;
    HIPCHK(c, hipMemcpyAsync(c->dense.dist, dist, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    for (int a = 0; a < 3; ++a) {
        HIPCHK(c, hipMemcpyAsync(c->dense.g[a], grad_xyz + (size_t)a * n, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->dense.rho[a], rgb + (size_t)a * n, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->dense.weight, weight, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->vis_seq, vis_words, sizeof(uint64_t) * n * words_per_voxel, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_volume = true; c->inited = false;
    return PSGSDF_OK;
}

int psgsdf_volume_init(psgsdf_ctx* c, int max_frames) {
    if (!c || max_frames < 1) return fail(c, PSGSDF_ERR_ARG, "volume_init: max_frames");
    HIPCHK(c, hipSetDevice(c->device));
    const long long n = c->grid.nvox;
    free_dense(c);
    if (c->vis_seq) { hipFree(c->vis_seq); c->vis_seq = nullptr; }
    int rc = alloc_dense(c, c->dense, n, 0, true); if (rc) return rc;
    c->wpv_seq = (max_frames + 63) / 64;
    HIPCHK(c, hipMalloc(&c->vis_seq, sizeof(uint64_t) * n * c->wpv_seq));
    launch_fill_f32(c->dense.dist, c->grid.T, n, c->stream);
    for (int a = 0; a < 3; ++a) { HIPCHK(c, hipMemsetAsync(c->dense.g[a], 0, sizeof(float) * n, c->stream)); HIPCHK(c, hipMemsetAsync(c->dense.rho[a], 0, sizeof(float) * n, c->stream)); }
    HIPCHK(c, hipMemsetAsync(c->dense.weight, 0, sizeof(float) * n, c->stream));
    HIPCHK(c, hipMemsetAsync(c->vis_seq, 0, sizeof(uint64_t) * n * c->wpv_seq, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_volume = true; c->inited = false;
    return PSGSDF_OK;
}

int psgsdf_integrate_frame(psgsdf_ctx* c, const float* rgb, const float* depth, const float* normals_xyz, int width, int height, const float pose[16], int counter, float z_min, float z_max) {
    if (!c || !c->have_volume || !c->vis_seq) return fail(c, PSGSDF_ERR_STATE, "integrate_frame: volume_init or upload_volume first");
    if (!rgb || !depth || !normals_xyz || !pose || width < 2 || height < 2 || counter < 0 || counter >= 64 * c->wpv_seq) return fail(c, PSGSDF_ERR_ARG, "integrate_frame: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npx = (size_t)width * height;
    if (c->stage_px < npx) {
        hipFree(c->stage); c->stage = nullptr; c->stage_px = 0;
        HIPCHK(c, hipMalloc(&c->stage, sizeof(float) * npx * 7));
        c->stage_px = npx;
    }
    float* d_rgb = c->stage; float* d_depth = c->stage + 3 * npx; float* d_nrm = c->stage + 4 * npx;
    HIPCHK(c, hipMemcpyAsync(d_rgb, rgb, sizeof(float) * 3 * npx, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_depth, depth, sizeof(float) * npx, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_nrm, normals_xyz, sizeof(float) * 3 * npx, hipMemcpyHostToDevice, c->stream));
    Cam cam = c->cam; cam.W = width; cam.H = height;
    FrameP fp{};
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) fp.R[i * 3 + j] = pose[i * 4 + j]; fp.t[i] = pose[i * 4 + 3]; }
    timed(c, "integrate_frame", [&] { launch_integrate(c->dense, c->vis_seq, c->wpv_seq, c->grid, cam, fp, d_rgb, d_depth, d_nrm, counter, z_min, z_max, c->stream); });
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->inited = false;
    return PSGSDF_OK;
}

int psgsdf_set_keyframes(psgsdf_ctx* c, int n_frames, const int32_t* frame_idx, const float* rgb_images, int width, int height, const float* poses) {
    if (!c || n_frames <= 0 || !frame_idx || !rgb_images || !poses || width <= 1 || height <= 1) return fail(c, PSGSDF_ERR_ARG, "set_keyframes: bad argument");
    if (n_frames > kMaxFramesLds) return fail(c, PSGSDF_ERR_UNSUPPORTED, "at most %d keyframes", kMaxFramesLds);
    HIPCHK(c, hipSetDevice(c->device));
    hipFree(c->frame_idx); hipFree(c->img); hipFree(c->frames); hipFree(c->acc_frame);
    c->frame_idx = nullptr; c->img = nullptr; c->frames = nullptr; c->acc_frame = nullptr;
    c->F = n_frames; c->cam.W = width; c->cam.H = height;
    const size_t npx = (size_t)n_frames * width * height * 3;
    HIPCHK(c, hipMalloc(&c->frame_idx, sizeof(int) * n_frames));
    HIPCHK(c, hipMalloc(&c->img, sizeof(float) * npx));
    HIPCHK(c, hipMalloc(&c->frames, sizeof(FrameP) * n_frames));
    c->acc_frame_n = (size_t)n_frames * 64;
    HIPCHK(c, hipMalloc(&c->acc_frame, sizeof(double) * c->acc_frame_n));
    HIPCHK(c, hipMemsetAsync(c->acc_frame, 0, sizeof(double) * c->acc_frame_n, c->stream));   // invariant: zero outside [sweep, solve]
    HIPCHK(c, hipMemcpyAsync(c->frame_idx, frame_idx, sizeof(int) * n_frames, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->img, rgb_images, sizeof(float) * npx, hipMemcpyHostToDevice, c->stream));
    c->frames_h.assign(n_frames, FrameP{});
    for (int f = 0; f < n_frames; ++f) {
        const float* P = poses + 16 * f;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) c->frames_h[f].R[i * 3 + j] = P[i * 4 + j]; c->frames_h[f].t[i] = P[i * 4 + 3]; }
    }
    HIPCHK(c, hipMemcpyAsync(c->frames, c->frames_h.data(), sizeof(FrameP) * n_frames, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_frames = true; c->inited = false;
    return PSGSDF_OK;
}

int psgsdf_init(psgsdf_ctx* c) {
    if (!c || !c->have_volume || !c->have_frames) return fail(c, PSGSDF_ERR_STATE, "init: upload_volume and set_keyframes first");
    if (!c->vis_seq) return fail(c, PSGSDF_ERR_STATE, "init: volume was refined; upload it again");
    HIPCHK(c, hipSetDevice(c->device));
    const int KW = (c->F + 63) / 64;
    if (c->dense.vis) { hipFree(c->dense.vis); c->dense.vis = nullptr; }
    HIPCHK(c, hipMalloc(&c->dense.vis, sizeof(uint64_t) * c->grid.nvox * KW));
    c->dense.KW = KW;
    timed(c, "select_vis", [&] { launch_select_vis(c->vis_seq, c->wpv_seq, c->dense.vis, KW, c->frame_idx, c->F, c->grid.nvox, c->stream); });
    int rc = build_band(c); if (rc) return rc;
    // light initialisation: PsOptimizer.cpp:30-37 l = SH(R*(0,0,-1)), l[0] = 0.02 ; LED: ones, then intensity ratio
    const bool led = c->set.model == PSGSDF_LED;
    for (int f = 0; f < c->F; ++f) {
        FrameP& fp = c->frames_h[f];
        for (int i = 0; i < 9; ++i) fp.l[i] = 0.f;
        if (led) { fp.l[0] = fp.l[1] = fp.l[2] = 1.0f; continue; }
        float n[3];
        for (int i = 0; i < 3; ++i) n[i] = (fp.R[i * 3 + 0] * 0.0f + fp.R[i * 3 + 1] * 0.0f) + fp.R[i * 3 + 2] * -1.0f;
        fp.l[0] = 0.02f; fp.l[1] = n[0]; fp.l[2] = n[1]; fp.l[3] = n[2];
        if (c->set.model == PSGSDF_SH2) { fp.l[4] = n[0] * n[1]; fp.l[5] = n[0] * n[2]; fp.l[6] = n[1] * n[2]; fp.l[7] = n[0] * n[0] - n[1] * n[1]; fp.l[8] = n[0] * n[0] - n[2] * n[2]; }
    }
    HIPCHK(c, hipMemcpyAsync(c->frames, c->frames_h.data(), sizeof(FrameP) * c->F, hipMemcpyHostToDevice, c->stream));
    if ((rc = derive(c, 0))) return rc;
    if (led && c->n_ranks == 1) {   // computeLightIntensive, LedOptimizer.cpp:76-112 (multi-rank: phases MG_LED_SUMS / MG_LED_SET)
        SweepArgs a = make_args(c, 0);
        timed(c, "led_light_init", [&] { launch_led_light_init(a, c->stream); });
        const int slots[6] = {SC_AUX0, SC_AUX1, SC_AUX2, SC_EN, SC_EL, SC_ACCEPT}; double s[6];
        if ((rc = read_parts(c, slots, 6, s))) return rc;
        float L[3] = {(float)s[0] / (float)s[3], (float)s[1] / (float)s[4], (float)s[2] / (float)s[5]};
        std::vector<FrameP> fr(c->F);
        HIPCHK(c, hipMemcpy(fr.data(), c->frames, sizeof(FrameP) * c->F, hipMemcpyDeviceToHost));
        for (int f = 0; f < c->F; ++f) for (int ch = 0; ch < 3; ++ch) fr[f].l[ch] = L[ch];
        HIPCHK(c, hipMemcpy(c->frames, fr.data(), sizeof(FrameP) * c->F, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->led_light, L, sizeof(L), hipMemcpyHostToDevice));
        if ((rc = derive(c, 0))) return rc;   // restores the cached Eikonal / Laplacian sums
    }
    c->inited = true;
    return PSGSDF_OK;
}

int psgsdf_init_albedo(psgsdf_ctx* c) {
    if (!c || !c->inited) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    SweepArgs a = make_args(c, 0);
    timed(c, "init_albedo", [&] { launch_init_albedo(a, c->stream); });
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PSGSDF_OK;
}

int psgsdf_energy(psgsdf_ctx* c, double out[4]) {
    if (!c || !c->inited || !out) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    double E; int rc = ps_energy(c, &E, nullptr); if (rc) return rc;
    out[0] = E; out[1] = band_mean(c, c->en_sum); out[2] = band_mean(c, c->el_sum);
    double er = 0; if (c->reg_r != 0.f && (rc = albedo_reg_energy(c, &er))) return rc;
    out[3] = (double)total_energy(c, (float)out[0], c->reg_n != 0.f ? (float)out[1] : 0.f, c->reg_l != 0.f ? (float)out[2] : 0.f, (float)er);
    return PSGSDF_OK;
}

int psgsdf_normalize_weights(psgsdf_ctx* c, double* e_total) {
    if (!c || !c->inited) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    double e; int rc = ps_energy(c, &e, nullptr); if (rc) return rc;
    float E = (float)e, E_n = 0, E_l = 0;
    if (c->reg_n != 0.f) { E_n = (float)band_mean(c, c->en_sum); c->reg_n *= E / E_n; }   // PsOptimizer.cpp:275-278
    if (c->reg_l != 0.f) { E_l = (float)band_mean(c, c->el_sum); c->reg_l *= E / E_l; }   // PsOptimizer.cpp:281-284
    double er = 0; if (c->reg_r != 0.f && (rc = albedo_reg_energy(c, &er))) return rc;   // reg_rho is not normalised (PsOptimizer.cpp:279)
    if (e_total) *e_total = (double)total_energy(c, E, E_n, E_l, (float)er);
    return PSGSDF_OK;
}

int psgsdf_step(psgsdf_ctx* c, int block, psgsdf_step_stats* stats) {
    if (!c || !c->inited) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    return do_step(c, block, c->reg_l != 0.f, stats);
}

int psgsdf_iterate(psgsdf_ctx* c, int flags, int n_iters, psgsdf_iter_stats* stats) {
    if (!c || !c->inited) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    double e; int rc = ps_energy(c, &e, nullptr); if (rc) return rc;
    LoopState L{};
    L.E = (float)e;
    L.E_n = c->reg_n != 0.f ? (float)band_mean(c, c->en_sum) : 0.f; L.E_l = c->reg_l != 0.f ? (float)band_mean(c, c->el_sum) : 0.f;
    if (c->reg_r != 0.f) { double er; if ((rc = albedo_reg_energy(c, &er))) return rc; L.E_r = (float)er; }
    L.E_prev = total_energy(c, L.E, L.E_n, L.E_l, L.E_r);
    L.laplacian_reg = c->reg_l != 0.f;
    int done = 0;
    return run_loop(c, flags, L, n_iters, false, stats, stats ? n_iters : 0, &done, nullptr, nullptr, nullptr);
}

int psgsdf_optimize(psgsdf_ctx* c, int flags, psgsdf_iter_stats* stats, int stats_cap, int* n_done, int* result, psgsdf_iter_cb on_iter, void* user) {
    if (!c || !c->inited) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    LoopState L{};
    L.laplacian_reg = c->reg_l != 0.f;
    int rc = psgsdf_init_albedo(c); if (rc) return rc;
    double e; if ((rc = ps_energy(c, &e, nullptr))) return rc;
    L.E = (float)e;
    if (c->reg_n != 0.f) { L.E_n = (float)band_mean(c, c->en_sum); c->reg_n *= L.E / L.E_n; }                                             // PsOptimizer.cpp:275-278
    if (L.laplacian_reg) { L.E_l = (float)band_mean(c, c->el_sum); c->reg_l *= L.E / L.E_l; if (c->set.upsample) L.laplacian_reg = 0; }   // :281-285
    if (c->reg_r != 0.f) { double er; if ((rc = albedo_reg_energy(c, &er))) return rc; L.E_r = (float)er; }   // PsOptimizer.cpp:279
    L.E_prev = total_energy(c, L.E, L.E_n, L.E_l, L.E_r);
    return run_loop(c, flags, L, c->set.max_it, true, stats, stats_cap, n_done, result, on_iter, user);
}

int psgsdf_upsample2x(psgsdf_ctx* c) {
    if (!c || !c->inited) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    return do_upsample(c);
}

int psgsdf_get_info(psgsdf_ctx* c, psgsdf_info* info) {
    if (!c || !info) return PSGSDF_ERR_ARG;
    for (int a = 0; a < 3; ++a) { info->dim[a] = c->grid.dim[a]; info->origin[a] = c->grid.origin[a]; }
    info->voxel_size = c->grid.vs; info->n_frames = c->F; info->n_band = c->inited ? c->band.S : 0;
    info->light_stride = c->set.model == PSGSDF_LED ? 3 : (c->set.model == PSGSDF_SH2 ? 9 : 4);
    info->vis_words = c->dense.KW; info->reg_weight_n = c->reg_n; info->reg_weight_l = c->reg_l;
    return PSGSDF_OK;
}

int psgsdf_download_volume(psgsdf_ctx* c, float* dist, float* grad_xyz, float* weight, float* rgb, uint64_t* vis_words) {
    if (!c || !c->have_volume) return fail(c, PSGSDF_ERR_STATE, "no volume");
    HIPCHK(c, hipSetDevice(c->device));
    const long long n = c->grid.nvox;
    if (c->inited) launch_band_scatter(c->dense, c->band, c->stream);
    if (dist) HIPCHK(c, hipMemcpyAsync(dist, c->dense.dist, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    for (int a = 0; a < 3; ++a) {
        if (grad_xyz) HIPCHK(c, hipMemcpyAsync(grad_xyz + (size_t)a * n, c->dense.g[a], sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
        if (rgb) HIPCHK(c, hipMemcpyAsync(rgb + (size_t)a * n, c->dense.rho[a], sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    }
    if (weight) HIPCHK(c, hipMemcpyAsync(weight, c->dense.weight, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    if (vis_words) {
        if (!c->dense.vis) return fail(c, PSGSDF_ERR_STATE, "visibility not selected yet");
        HIPCHK(c, hipMemcpyAsync(vis_words, c->dense.vis, sizeof(uint64_t) * n * c->dense.KW, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PSGSDF_OK;
}

int psgsdf_download_vis_seq(psgsdf_ctx* c, uint64_t* out) {
    if (!c || !c->vis_seq || !out) return PSGSDF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(out, c->vis_seq, sizeof(uint64_t) * c->grid.nvox * c->wpv_seq, hipMemcpyDeviceToHost));
    return c->wpv_seq;
}

int psgsdf_download_band(psgsdf_ctx* c, int32_t* lin_idx) {
    if (!c || !c->inited || !lin_idx) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(lin_idx, c->band.lin, sizeof(int) * c->band.S, hipMemcpyDeviceToHost));
    return PSGSDF_OK;
}

int psgsdf_download_poses(psgsdf_ctx* c, float* poses) {
    if (!c || !c->have_frames || !poses) return fail(c, PSGSDF_ERR_STATE, "no keyframes");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<FrameP> fr(c->F);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(fr.data(), c->frames, sizeof(FrameP) * c->F, hipMemcpyDeviceToHost));
    for (int f = 0; f < c->F; ++f) {
        float* P = poses + 16 * f;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P[i * 4 + j] = fr[f].R[i * 3 + j]; P[i * 4 + 3] = fr[f].t[i]; }
        P[12] = P[13] = P[14] = 0.f; P[15] = 1.f;
    }
    return PSGSDF_OK;
}

int psgsdf_download_light(psgsdf_ctx* c, float* light) {
    if (!c || !c->inited || !light) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<FrameP> fr(c->F);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(fr.data(), c->frames, sizeof(FrameP) * c->F, hipMemcpyDeviceToHost));
    if (c->set.model == PSGSDF_LED) { for (int ch = 0; ch < 3; ++ch) light[ch] = fr[0].l[ch]; return PSGSDF_OK; }
    const int nb = c->set.model == PSGSDF_SH2 ? 9 : 4;
    for (int f = 0; f < c->F; ++f) for (int i = 0; i < nb; ++i) light[(size_t)f * nb + i] = fr[f].l[i];
    return PSGSDF_OK;
}

int psgsdf_upload_light(psgsdf_ctx* c, const float* light) {
    if (!c || !c->inited || !light) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<FrameP> fr(c->F);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(fr.data(), c->frames, sizeof(FrameP) * c->F, hipMemcpyDeviceToHost));
    const bool led = c->set.model == PSGSDF_LED;
    const int nb = led ? 3 : (c->set.model == PSGSDF_SH2 ? 9 : 4);
    for (int f = 0; f < c->F; ++f) for (int i = 0; i < nb; ++i) fr[f].l[i] = led ? light[i] : light[(size_t)f * nb + i];
    HIPCHK(c, hipMemcpy(c->frames, fr.data(), sizeof(FrameP) * c->F, hipMemcpyHostToDevice));
    if (led) HIPCHK(c, hipMemcpy(c->led_light, light, sizeof(float) * 3, hipMemcpyHostToDevice));
    return PSGSDF_OK;
}

}  // extern "C"

// ---- front end: FALS normals and depth tracker -------------------------------------------
namespace {
inline int reflect101_h(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; } return i; }
void box_filter_h(const std::vector<double>& src, std::vector<double>& dst, int W, int H, int r) {
    std::vector<double> tmp((size_t)W * H);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) { double s = 0; for (int k = -r; k <= r; ++k) s += src[(size_t)y * W + reflect101_h(x + k, W)]; tmp[(size_t)y * W + x] = s; }
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) { double s = 0; for (int k = -r; k <= r; ++k) s += tmp[(size_t)reflect101_h(y + k, H) * W + x]; dst[(size_t)y * W + x] = s; }
}
// NormalEstimator::cache (NormalEstimator.h:52-125), once per image size: double on the host, 9 float planes on the device
int normals_cache(psgsdf_ctx* c, int W, int H) {
    if (c->ncache && c->ncache_w == W && c->ncache_h == H) return 0;
    const size_t n = (size_t)W * H;
    hipFree(c->ncache); hipFree(c->ntmp); hipFree(c->nout); hipFree(c->ndepth); c->ncache = nullptr; c->ntmp = nullptr; c->nout = nullptr; c->ndepth = nullptr;
    std::vector<double> a[6], M[6]; for (int i = 0; i < 6; ++i) { a[i].resize(n); M[i].resize(n); }
    std::vector<float> out(9 * n);
    const double fx_inv = 1. / (double)c->cam.fx, fy_inv = 1. / (double)c->cam.fy, cx = (double)c->cam.cx, cy = (double)c->cam.cy;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        size_t p = (size_t)y * W + x;
        double x0 = fx_inv * ((double)x - cx), y0 = fy_inv * ((double)y - cy), nsi = 1. / (1. + x0 * x0 + y0 * y0);
        a[0][p] = x0 * x0 * nsi; a[1][p] = x0 * y0 * nsi; a[2][p] = x0 * nsi; a[3][p] = y0 * y0 * nsi; a[4][p] = y0 * nsi; a[5][p] = nsi;
        out[p] = (float)(x0 * nsi); out[n + p] = (float)(y0 * nsi); out[2 * n + p] = (float)nsi;
    }
    for (int i = 0; i < 6; ++i) box_filter_h(a[i], M[i], W, H, 5);
    for (size_t p = 0; p < n; ++p) {
        double M11 = M[0][p], M12 = M[1][p], M13 = M[2][p], M22 = M[3][p], M23 = M[4][p], M33 = M[5][p];
        double det = M11 * (M22 * M33) + 2 * M12 * (M23 * M13) - (M13 * (M13 * M22) + M12 * (M12 * M33) + M23 * (M23 * M11));
        double di = 1. / det;
        out[3 * n + p] = (float)(di * (M22 * M33 - M23 * M23)); out[4 * n + p] = (float)(di * (M13 * M23 - M12 * M33));
        out[5 * n + p] = (float)(di * (M12 * M23 - M13 * M22)); out[6 * n + p] = (float)(di * (M11 * M33 - M13 * M13));
        out[7 * n + p] = (float)(di * (M12 * M13 - M11 * M23)); out[8 * n + p] = (float)(di * (M11 * M22 - M12 * M12));
    }
    HIPCHK(c, hipMalloc(&c->ncache, sizeof(float) * 9 * n)); HIPCHK(c, hipMalloc(&c->ntmp, sizeof(double) * 3 * n));
    HIPCHK(c, hipMalloc(&c->nout, sizeof(float) * 3 * n)); HIPCHK(c, hipMalloc(&c->ndepth, sizeof(float) * n));
    HIPCHK(c, hipMemcpy(c->ncache, out.data(), sizeof(float) * 9 * n, hipMemcpyHostToDevice));
    c->ncache_w = W; c->ncache_h = H;
    return 0;
}
}  // namespace

extern "C" int psgsdf_estimate_normals(psgsdf_ctx* c, const float* depth, int width, int height, float* normals_xyz) {
    if (!c || !depth || !normals_xyz || width < 2 || height < 2) return fail(c, PSGSDF_ERR_ARG, "estimate_normals: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = normals_cache(c, width, height); if (rc) return rc;
    const size_t n = (size_t)width * height;
    HIPCHK(c, hipMemcpyAsync(c->ndepth, depth, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    timed(c, "normals", [&] { launch_normals(c->ndepth, c->ncache, width, height, 5, c->ntmp, c->nout, c->stream); });
    HIPCHK(c, hipMemcpyAsync(normals_xyz, c->nout, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PSGSDF_OK;
}

// RigidPointOptimizer::optimize_sampled (RigidPointOptimizer.cpp:12-79), sampling = 1: frame-to-model tracking on the dense volume
extern "C" int psgsdf_track(psgsdf_ctx* c, const float* depth, int width, int height, float pose[16], float z_min, float z_max,
                            int num_iterations, float conv_threshold, float damping, int* iters_out, int* converged) {
    if (!c || !c->have_volume || !depth || !pose) return fail(c, PSGSDF_ERR_STATE, "track: volume first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)width * height;
    int rc = normals_cache(c, width, height); if (rc) return rc;   // (allocates the depth staging buffer)
    const int nblk = 256;
    if (!c->track_part) { HIPCHK(c, hipMalloc(&c->track_part, sizeof(double) * nblk * 29)); HIPCHK(c, hipHostMalloc(&c->track_host, sizeof(double) * nblk * 29)); }
    HIPCHK(c, hipMemcpyAsync(c->ndepth, depth, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    Cam cam = c->cam; cam.W = width; cam.H = height;
    if (converged) *converged = 0;
    int k = 0;
    for (; k < num_iterations; ++k) {
        FrameP fp{};
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) fp.R[i * 3 + j] = pose[i * 4 + j]; fp.t[i] = pose[i * 4 + 3]; }
        timed(c, "track", [&] { launch_track(c->dense, c->grid, cam, fp, c->ndepth, z_min, z_max, c->track_part, nblk, c->stream); });
        HIPCHK(c, hipMemcpyAsync(c->track_host, c->track_part, sizeof(double) * nblk * 29, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        double acc[29] = {0};
        for (int b = 0; b < nblk; ++b) for (int q = 0; q < 29; ++q) acc[q] += c->track_host[(size_t)b * 29 + q];
        if (acc[28] == 0) break;
        // xi = damping * H.llt().solve(g): 6x6 LDL^T in double on the host
        double Hd[36], gd[6], L[36] = {0}, D[6], y[6], xd[6]; int q = 0;
        for (int i = 0; i < 6; ++i) { for (int j = i; j < 6; ++j) { Hd[i * 6 + j] = (double)(float)acc[q]; Hd[j * 6 + i] = (double)(float)acc[q]; ++q; } gd[i] = (double)(float)acc[21 + i]; }
        double scale = 0; for (int i = 0; i < 6; ++i) scale = std::max(scale, fabs(Hd[i * 6 + i])); const double tiny = scale * 1e-12;
        for (int j = 0; j < 6; ++j) { double dd = Hd[j * 6 + j]; for (int m = 0; m < j; ++m) dd -= L[j * 6 + m] * L[j * 6 + m] * D[m]; D[j] = dd; L[j * 6 + j] = 1;
            for (int i = j + 1; i < 6; ++i) { double s = Hd[i * 6 + j]; for (int m = 0; m < j; ++m) s -= L[i * 6 + m] * L[j * 6 + m] * D[m]; L[i * 6 + j] = dd > tiny ? s / dd : 0; } }
        for (int i = 0; i < 6; ++i) { double s = gd[i]; for (int m = 0; m < i; ++m) s -= L[i * 6 + m] * y[m]; y[i] = s; }
        for (int i = 0; i < 6; ++i) y[i] = D[i] > tiny ? y[i] / D[i] : 0;
        for (int i = 5; i >= 0; --i) { double s = y[i]; for (int m = i + 1; m < 6; ++m) s -= L[m * 6 + i] * xd[m]; xd[i] = s; }
        float xi[6], n2 = 0; for (int i = 0; i < 6; ++i) { xi[i] = damping * (float)xd[i]; n2 += xi[i] * xi[i]; }
        if (n2 < conv_threshold * conv_threshold) { if (converged) *converged = 1; break; }
        // pose = SE3::exp(-xi) * pose
        float w[3] = {-xi[3], -xi[4], -xi[5]}, u[3] = {-xi[0], -xi[1], -xi[2]};
        float th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
        float Rm[9];
        {   // SO3::exp via quaternion (same form as the device so3_exp)
            float imag, real;
            if (th2 < 1e-10f) { float t4 = th2 * th2; imag = 0.5f - (1.0f / 48.0f) * th2 + (1.0f / 3840.0f) * t4; real = 1.0f - (1.0f / 8.0f) * th2 + (1.0f / 384.0f) * t4; }
            else { float th = sqrtf(th2), half = 0.5f * th; imag = sinf(half) / th; real = cosf(half); }
            float qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
            float tx = 2 * qx, ty = 2 * qy, tz = 2 * qz, twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
            Rm[0] = 1 - (tyy + tzz); Rm[1] = txy - twz; Rm[2] = txz + twy; Rm[3] = txy + twz; Rm[4] = 1 - (txx + tzz); Rm[5] = tyz - twx; Rm[6] = txz - twy; Rm[7] = tyz + twx; Rm[8] = 1 - (txx + tyy);
        }
        float Om[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, Om2[9], V[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Om2[i * 3 + j] = (Om[i * 3] * Om[j] + Om[i * 3 + 1] * Om[3 + j]) + Om[i * 3 + 2] * Om[6 + j];
        if (th2 < 1e-10f) { for (int i = 0; i < 9; ++i) V[i] = Rm[i]; }
        else { float th = sqrtf(th2), a = (1.f - cosf(th)) / th2, bq = (th - sinf(th)) / (th2 * th); for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.f : 0.f) + a * Om[i] + bq * Om2[i]; }
        float E[16] = {0}, P[16];
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) E[i * 4 + j] = Rm[i * 3 + j]; E[i * 4 + 3] = (V[i * 3] * u[0] + V[i * 3 + 1] * u[1]) + V[i * 3 + 2] * u[2]; }
        E[15] = 1;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float s = 0; for (int m = 0; m < 4; ++m) s += E[i * 4 + m] * pose[m * 4 + j]; P[i * 4 + j] = s; }
        memcpy(pose, P, sizeof(P));
    }
    if (iters_out) *iters_out = k;
    return PSGSDF_OK;
}

extern "C" {
// ---- multi-rank (z-slab) phase API ---------------------------------------------------------
// One process per GPU.  Every rank holds the whole band but owns rows [row0,row1) (equal band count = z-slabs);
// the host program (psgradientsdf_amd/distributed.py) runs the phases below and performs the exchanges between them
// with torch.distributed (RCCL on the GPUs): all-reduce of the frame accumulators / folded scalars / PCG scalars,
// halo exchange of contiguous row ranges of `blk`, `{z,p}` and `dist`.  DESIGN.md §7.
int psgsdf_comm_unique_id(uint8_t id[128]) { (void)id; return PSGSDF_ERR_UNSUPPORTED; }   // collectives live in the host program
int psgsdf_comm_init(psgsdf_ctx* c, const uint8_t id[128], int rank, int n_ranks) {
    (void)id;
    if (!c || rank < 0 || n_ranks < 1 || rank >= n_ranks) return PSGSDF_ERR_ARG;
    c->rank = rank; c->n_ranks = n_ranks; c->inited = false;
    return PSGSDF_OK;
}
int psgsdf_set_stream(psgsdf_ctx* c, void* hip_stream) {
    if (!c) return PSGSDF_ERR_ARG;
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->stream && c->own_stream) hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream; c->own_stream = false;
    return PSGSDF_OK;
}
int psgsdf_mg_info(psgsdf_ctx* c, int32_t out[10]) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    out[0] = c->band.S; out[1] = c->band.Spad; out[2] = c->row0; out[3] = c->row1; out[4] = c->halo; out[5] = c->F; out[6] = c->rank; out[7] = c->n_ranks; out[8] = c->need[0]; out[9] = c->need[1];
    return PSGSDF_OK;
}
int psgsdf_mg_buffer(psgsdf_ctx* c, int which, void** ptr, int64_t* count) {
    if (!c || !c->inited || !ptr || !count) return PSGSDF_ERR_STATE;
    const int64_t Sp = c->band.Spad;
    switch (which) {
        case PSGSDF_MG_BUF_FRAME_ACC: *ptr = c->acc_frame; *count = (int64_t)c->F * kFrameRow; break;   /* f64 */
        case PSGSDF_MG_BUF_SCAL: *ptr = c->mg_scal; *count = kMgScal; break;                            /* f64 */
        case PSGSDF_MG_BUF_PCG: *ptr = c->mg_ext; *count = 8; break;                                    /* f64 */
        case PSGSDF_MG_BUF_DIST: *ptr = c->band.dist; *count = Sp; break;                               /* f32 */
        case PSGSDF_MG_BUF_BLK: *ptr = c->band.blk; *count = 14 * Sp; break;                            /* f32, 14 planes */
        case PSGSDF_MG_BUF_REC0: *ptr = c->band.rec[0]; *count = 4 * Sp; break;                         /* f32 x 4 per row */
        case PSGSDF_MG_BUF_REC1: *ptr = c->band.rec[1]; *count = 4 * Sp; break;
        case PSGSDF_MG_BUF_RHO: *ptr = c->band.rho[0]; *count = 3 * Sp; break;                          /* f32, 3 planes */
        case PSGSDF_MG_BUF_GRAD: *ptr = c->band.g[0]; *count = 3 * Sp; break;                           /* f32, 3 planes */
        default: return PSGSDF_ERR_ARG;
    }
    return PSGSDF_OK;
}
static int mg_fold(psgsdf_ctx* c, std::initializer_list<int> slots) {
    SlotList sl; sl.n = 0; for (int s_ : slots) sl.id[sl.n++] = s_;
    if (c->mg_fold_base < 0 || c->mg_fold_base + sl.n > kMgScal) return fail(c, PSGSDF_ERR_ARG, "fold base %d out of range", c->mg_fold_base);
    if (c->row1 <= c->row0) {   // this rank owns no rows: its kernels were not launched, its contribution to every sum is 0
        HIPCHK(c, hipMemsetAsync(c->mg_scal + c->mg_fold_base, 0, sizeof(double) * sl.n, c->stream));
        return 0;
    }
    launch_sum_parts(c->part, c->PB, band_blocks(c), sl, c->mg_scal + c->mg_fold_base, c->stream);
    return 0;
}
static int mg_pcg_cap(psgsdf_ctx* c) { int cap = c->set.cg_max_it > 0 ? c->set.cg_max_it : 2 * c->band.S; return std::min(cap, c->pcg_cap); }
int psgsdf_mg_phase(psgsdf_ctx* c, int phase, int arg) {
    if (!c || !c->inited) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    SweepArgs a = make_args(c, arg);
    int rc = 0;
    switch (phase) {
        case PSGSDF_MG_ENERGY: timed(c, "energy", [&] { launch_energy(a, c->stream); }); return mg_fold(c, {SC_ENERGY, SC_NOBS});
        case PSGSDF_MG_INIT_ALBEDO: launch_init_albedo(a, c->stream); return 0;
        case PSGSDF_MG_LED_SUMS: launch_led_light_init(a, c->stream); return mg_fold(c, {SC_AUX0, SC_AUX1, SC_AUX2, SC_EN, SC_EL, SC_ACCEPT});
        case PSGSDF_MG_LED_SET: {   // mg_scal holds the all-reduced sums
            double s_[6]; HIPCHK(c, hipMemcpyAsync(s_, c->mg_scal, sizeof(s_), hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
            float L[3] = {(float)s_[0] / (float)s_[3], (float)s_[1] / (float)s_[4], (float)s_[2] / (float)s_[5]};
            return psgsdf_upload_light(c, L);
        }
        case PSGSDF_MG_SWEEP_ALBEDO: timed(c, "sweep_albedo", [&] { launch_sweep_albedo(a, c->stream); }); return mg_fold(c, {SC_ENERGY, SC_NOBS});
        case PSGSDF_MG_APPLY_ALBEDO: timed(c, "apply_albedo", [&] { launch_apply_albedo(a, c->stream); }); return mg_fold(c, {SC_ACCEPT});
        case PSGSDF_MG_SWEEP_LIGHT: HIPCHK(c, hipMemsetAsync(c->acc_frame, 0, sizeof(double) * c->acc_frame_n, c->stream)); timed(c, "sweep_light", [&] { launch_sweep_light(a, c->stream); }); return 0;
        case PSGSDF_MG_SOLVE_LIGHT: timed(c, "solve_light", [&] { launch_solve_light(a, c->frames, c->led_light, nullptr, c->stream); }); return 0;
        case PSGSDF_MG_SWEEP_POSE: HIPCHK(c, hipMemsetAsync(c->acc_frame, 0, sizeof(double) * c->acc_frame_n, c->stream)); timed(c, "sweep_pose", [&] { launch_sweep_pose(a, c->stream); }); return 0;
        case PSGSDF_MG_SOLVE_POSE: timed(c, "solve_pose", [&] { launch_solve_pose(a, c->frames, nullptr, c->stream); }); return 0;
        case PSGSDF_MG_SWEEP_DIST: timed(c, "sweep_dist", [&] { launch_sweep_dist(a, c->stream); }); return mg_fold(c, {SC_ENERGY, SC_NOBS});
        case PSGSDF_MG_ASSEMBLE: timed(c, "assemble", [&] { launch_assemble(a, c->stream); }); return 0;
        case PSGSDF_MG_PCG_INIT: {
            int G, rows; cgf_shape(band_blocks(c), &G, &rows);
            timed(c, "pcg_init", [&] { launch_cgf_init(a, c->pcg_sc, c->pcg_part, G, c->stream); });
            if (c->row1 <= c->row0) HIPCHK(c, hipMemsetAsync(c->mg_ext, 0, sizeof(double) * 8, c->stream));   // no rows: contributes 0
            else launch_cgf_sum(c->pcg_part, G, -1, c->mg_ext, c->stream);      // local |b|^2 -> ext[0]
            return 0;
        }
        case PSGSDF_MG_PCG_PASS: {   // arg = kernel index k: finishes pass k-1, runs pass k; ext holds the all-reduced sums of pass k-1
            if (arg < 0 || arg > mg_pcg_cap(c)) return fail(c, PSGSDF_ERR_ARG, "PCG kernel index %d out of range", arg);
            int G, rows; cgf_shape(band_blocks(c), &G, &rows);
            a.ext = c->mg_ext; a.laplacian_reg = 0;
            timed(c, "pcg_pass", [&] { launch_cgf_pass(a, c->pcg_sc, c->pcg_part, G, rows, arg, mg_pcg_cap(c), c->mg_hist + arg, c->stream); });
            if (c->row1 <= c->row0) HIPCHK(c, hipMemsetAsync(c->mg_ext, 0, sizeof(double) * 8, c->stream));
            else launch_cgf_sum(c->pcg_part, G, arg, c->mg_ext, c->stream);     // local sums of pass k -> ext[0..6]
            return 0;
        }
        case PSGSDF_MG_APPLY_DIST: timed(c, "apply_dist", [&] { launch_apply_dist(a, c->stream); }); return mg_fold(c, {SC_ACCEPT});
        case PSGSDF_MG_DERIVE: a.laplacian_reg = 0; timed(c, "derive", [&] { launch_derive(a, arg, c->stream); }); return mg_fold(c, {SC_EN, SC_EL});
        default: return fail(c, PSGSDF_ERR_ARG, "unknown phase %d", phase);
    }
    return rc;
}
// after the PCG kernels [k0, k0+n) have been enqueued: did the solve stop?  iters = -1 while it is still running.
// Kernel k publishes |b|^2 (k = 0) or |r|^2 after pass k-1; every rank sees the same (all-reduced) values.
int psgsdf_mg_pcg_status(psgsdf_ctx* c, int k0, int n, int32_t* iters, double* err) {
    if (!c || !c->inited || !iters || !err || k0 < 0 || n < 1 || n > 64) return PSGSDF_ERR_ARG;
    const int cap = mg_pcg_cap(c);
    HIPCHK(c, hipMemcpyAsync(c->host_buf, c->mg_hist, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->host_buf + 1, c->mg_hist + k0, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const float rhsN = (float)c->host_buf[0];
    *iters = -1; *err = 0;
    if (rhsN == 0.f) { *iters = 0; c->last_cg_iters = 0; return PSGSDF_OK; }
    const float thr = fmaxf(FLT_EPSILON * FLT_EPSILON * rhsN, FLT_MIN);
    float rn2 = rhsN;
    for (int q = 0; q < n && *iters < 0; ++q) {
        const int kk = k0 + q;
        if (kk == 0) continue;
        rn2 = (float)c->host_buf[1 + q];
        if (rn2 < thr) *iters = kk - 1; else if (kk == cap) *iters = cap;
    }
    *err = sqrt((double)rn2 / (double)rhsN);
    if (*iters >= 0) c->last_cg_iters = *iters;
    return PSGSDF_OK;
}
// where (offset in doubles) the scalars folded by the following phases land in the SCAL buffer: the host program gives every
// phase of an iteration its own slots and all-reduces / reads the buffer ONCE per iteration
int psgsdf_mg_fold_base(psgsdf_ctx* c, int base) { if (!c || base < 0 || base >= kMgScal) return PSGSDF_ERR_ARG; c->mg_fold_base = base; return PSGSDF_OK; }
// all-reduced Eikonal / Laplacian energy sums (host values) -> the context's energy bookkeeping
int psgsdf_mg_set_reg_sums(psgsdf_ctx* c, double en_sum, double el_sum) { if (!c) return PSGSDF_ERR_ARG; c->en_sum = en_sum; c->el_sum = el_sum; return PSGSDF_OK; }
// the engine's band planes ARE the exchange planes: nothing to pack (the CPU oracle keeps a dense grid and needs these)
int psgsdf_mg_pack_state(psgsdf_ctx* c) { return c ? PSGSDF_OK : PSGSDF_ERR_ARG; }
int psgsdf_mg_unpack_state(psgsdf_ctx* c) { return c ? PSGSDF_OK : PSGSDF_ERR_ARG; }
// effective regulariser weights are host state: the host program sets them after the global normalisation
int psgsdf_mg_set_weights(psgsdf_ctx* c, float reg_n, float reg_l) { if (!c) return PSGSDF_ERR_ARG; c->reg_n = reg_n; c->reg_l = reg_l; return PSGSDF_OK; }

// ---- measurement / test hooks -------------------------------------------------------------
int psgsdf_set_profiling(psgsdf_ctx* c, int enabled) { if (!c) return PSGSDF_ERR_ARG; c->profiling = enabled != 0; return PSGSDF_OK; }
int psgsdf_reset_kernel_times(psgsdf_ctx* c) { if (!c) return PSGSDF_ERR_ARG; c->ktimes.clear(); c->watch_used = 0; return PSGSDF_OK; }
int psgsdf_watch_kernel(psgsdf_ctx* c, const char* name) {
    if (!c) return PSGSDF_ERR_ARG;
    hipStreamSynchronize(c->stream);
    // "name" or "name/N": HIP events around every N-th launch of that kernel (default every launch)
    std::string w = name ? name : ""; int every = 1;
    const size_t sl = w.find('/');
    if (sl != std::string::npos) { every = std::max(1, atoi(w.c_str() + sl + 1)); w.resize(sl); }
    c->watch = w; c->watch_every = every; c->watch_seen = 0; c->watch_used = 0;
    return PSGSDF_OK;
}
int psgsdf_kernel_times(psgsdf_ctx* c, const char** names, double* ms, int64_t* launches, int cap) {
    if (!c) return 0;
    if (c->watch_used) {   // resolve the asynchronous event pairs of the watched kernel
        hipStreamSynchronize(c->stream);
        KTime& k = c->ktimes[c->watch];
        for (size_t i = 0; i < c->watch_used; ++i) { float t = 0; if (hipEventElapsedTime(&t, c->watch_pool[i].first, c->watch_pool[i].second) == hipSuccess) { k.ms += t; k.n += 1; } }
        c->watch_used = 0;
    }
    int n = 0;
    for (auto& kv : c->ktimes) { if (n >= cap) break; names[n] = kv.first.c_str(); ms[n] = kv.second.ms; launches[n] = kv.second.n; ++n; }
    return n;
}

int psgsdf_debug_dist_system(psgsdf_ctx* c, float* diag, float* rhs, const float* x, float* y) {
    if (!c || !c->inited) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    SweepArgs a = make_args(c, c->reg_l != 0.f);
    launch_sweep_dist(a, c->stream);
    launch_assemble(a, c->stream);
    const int S = c->band.S;
    if (diag) HIPCHK(c, hipMemcpyAsync(diag, c->band.H, sizeof(float) * S, hipMemcpyDeviceToHost, c->stream));
    if (rhs) HIPCHK(c, hipMemcpyAsync(rhs, c->band.rhs, sizeof(float) * S, hipMemcpyDeviceToHost, c->stream));
    if (x && y) {
        HIPCHK(c, hipMemcpyAsync(c->band.x, x, sizeof(float) * S, hipMemcpyHostToDevice, c->stream));
        launch_matvec(a, c->band.x, c->band.t, c->stream);
        HIPCHK(c, hipMemcpyAsync(y, c->band.t, sizeof(float) * S, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PSGSDF_OK;
}

// timing ablations of the PCG pass (tools/pcg_ablate.py): `reps` launches of k_cgf_pass with the given grid and ablation
// bits on the current (already assembled) distance system; results of the solve are garbage afterwards
int psgsdf_debug_time_pcg_pass(psgsdf_ctx* c, int blocks, int rows, int ablate, int reps, double* avg_ms, long long* stamps) {
    if (!c || !c->inited || !avg_ms) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    SweepArgs a = make_args(c, c->reg_l != 0.f);
    launch_sweep_dist(a, c->stream);
    launch_assemble(a, c->stream);
    int G, rdef; cgf_shape(band_blocks(c), &G, &rdef);
    if (blocks > 0) G = std::min(blocks, kCgfMaxBlocks);
    if (rows <= 0) rows = rdef;
    launch_cgf_init(a, c->pcg_sc, c->pcg_part, G, c->stream);
    hipEvent_t e0, e1; HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    for (int q = 0; q < 3; ++q) launch_cgf_pass(a, c->pcg_sc, c->pcg_part, G, rows, q, 1 << 30, c->mbox_dev, c->stream, ablate | 16);
    HIPCHK(c, hipEventRecord(e0, c->stream));
    for (int q = 0; q < reps; ++q) launch_cgf_pass(a, c->pcg_sc, c->pcg_part, G, rows, 3 + q, 1 << 30, c->mbox_dev, c->stream, ablate | 16);
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *avg_ms = (double)ms / reps;
    if (stamps) {   // [G][8] wall-clock ticks (100 MHz) of the LAST launch, taken with ablate | 1024
        HIPCHK(c, hipMemcpy(stamps, c->pcg_sc + 16, sizeof(long long) * 8 * G, hipMemcpyDeviceToHost));
    }
    return PSGSDF_OK;
}

// how many rows of the assembled distance system carry any of the 6 "rare" ELL columns, and how many 64-row groups
// (wavefronts of a one-row-per-thread launch) contain such a row
int psgsdf_debug_rare_rows(psgsdf_ctx* c, int64_t* rows, int64_t* waves) {
    if (!c || !c->inited || !rows || !waves) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    SweepArgs a = make_args(c, c->reg_l != 0.f);
    launch_sweep_dist(a, c->stream);
    launch_assemble(a, c->stream);
    std::vector<int> hx(c->band.S);
    HIPCHK(c, hipMemcpyAsync(hx.data(), c->band.hx, sizeof(int) * hx.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *rows = 0; *waves = 0;
    for (size_t i = 0; i < hx.size(); i += 64) { bool any = false; for (size_t k = i; k < std::min(hx.size(), i + 64); ++k) if (hx[k]) { ++*rows; any = true; } *waves += any; }
    return PSGSDF_OK;
}

int psgsdf_debug_frame_system(psgsdf_ctx* c, int block, double* H, double* b) {
    if (!c || !c->inited || !H || !b) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->acc_frame, 0, sizeof(double) * c->acc_frame_n, c->stream));
    SweepArgs a = make_args(c, 0);
    const bool led = c->set.model == PSGSDF_LED;
    int n, nb, nh;
    if (block == PSGSDF_LIGHT) { launch_sweep_light(a, c->stream); n = led ? 3 : (c->set.model == PSGSDF_SH2 ? 9 : 4); nb = led ? 1 : c->F; nh = led ? 3 : n * (n + 1) / 2; }
    else if (block == PSGSDF_POSE) { launch_sweep_pose(a, c->stream); n = 6; nb = c->F; nh = 21; }
    else return fail(c, PSGSDF_ERR_ARG, "block must be LIGHT or POSE");
    std::vector<double> acc(c->acc_frame_n);
    HIPCHK(c, hipMemcpyAsync(acc.data(), c->acc_frame, sizeof(double) * c->acc_frame_n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->acc_frame, 0, sizeof(double) * c->acc_frame_n, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (led && block == PSGSDF_LIGHT) {   // one global system: sum the per-frame rows
        for (int i = 0; i < 9; ++i) H[i] = 0;
        for (int i = 0; i < 3; ++i) { b[i] = 0; for (int f = 0; f < c->F; ++f) { H[i * 3 + i] += acc[(size_t)f * kFrameRow + i]; b[i] += acc[(size_t)f * kFrameRow + 3 + i]; } }
        return PSGSDF_OK;
    }
    for (int k = 0; k < nb; ++k) {
        const double* A = acc.data() + (size_t)k * kFrameRow;
        double* Hk = H + (size_t)k * n * n; double* bk = b + (size_t)k * n;
        for (int i = 0; i < n * n; ++i) Hk[i] = 0;
        int q = 0;
        for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { Hk[i * n + j] = A[q]; Hk[j * n + i] = A[q]; ++q; }
        for (int i = 0; i < n; ++i) bk[i] = A[nh + i];
    }
    return PSGSDF_OK;
}

int psgsdf_debug_albedo_system(psgsdf_ctx* c, float* H, float* b) {
    if (!c || !c->inited || !H || !b) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    SweepArgs a = make_args(c, 0);
    launch_sweep_albedo(a, c->stream);
    const int S = c->band.S, Sp = c->band.Spad;
    std::vector<float> h(3 * (size_t)Sp), bb(3 * (size_t)Sp);
    HIPCHK(c, hipMemcpyAsync(h.data(), c->band.aH, sizeof(float) * 3 * Sp, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(bb.data(), c->band.ab, sizeof(float) * 3 * Sp, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int j = 0; j < S; ++j) for (int ch = 0; ch < 3; ++ch) { H[3 * j + ch] = h[(size_t)ch * Sp + j]; b[3 * j + ch] = bb[(size_t)ch * Sp + j]; }
    return PSGSDF_OK;
}

}  // extern "C"
