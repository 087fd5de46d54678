// engine.hip -- context plumbing of the HIP engine: launch timing, deferred read-backs through the host-mapped mailbox,
// scalar folds, dense / band allocation and band construction, derived quantities, PS energy.  One context = one HIP device +
// one stream.  The kernels live in band / sweeps / dist / pcg / albedo_reg / frontend .hip; the C ABI in api*.hip.
#include "engine_internal.h"
#include <chrono>
#include <sched.h>

namespace psge {

int wait_mapped(psgsdf_ctx* c, const std::function<bool()>& ready, const char* what) {
    using clk = std::chrono::steady_clock;
    static const double limit_s = [] { const char* e = getenv("PSGSDF_WAIT_TIMEOUT_S"); double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 120.0; }();
    clk::time_point t0{}, next_query{};
    bool timing = false;
    for (unsigned spin = 1;; ++spin) {
        if (ready()) return 0;
        if (spin & 0x3ffu) continue;                       // look at the clock every 1024 reads (~0.1 ms)
        const clk::time_point now = clk::now();
        if (!timing) { timing = true; t0 = now; next_query = now + std::chrono::milliseconds(20); continue; }
        if (now - t0 > std::chrono::milliseconds(2)) sched_yield();   // long wait: let the runtime's own threads (and a rank sharing the cores) run
        if (now < next_query) continue;
        next_query = now + std::chrono::milliseconds(20);
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return ready() ? 0 : 1;        // drained: whatever was going to publish has
        if (q != hipErrorNotReady) return fail(c, PSGSDF_ERR_DEVICE, "%s: stream error while waiting: %s", what, hipGetErrorString(q));
        if (std::chrono::duration<double>(now - t0).count() > limit_s)
            return fail(c, PSGSDF_ERR_DEVICE, "%s: nothing published within %.0f s (PSGSDF_WAIT_TIMEOUT_S)", what, limit_s);
    }
}

int fail(psgsdf_ctx* c, int code, const char* fmt, ...) {
    if (c) { va_list ap; va_start(ap, fmt); vsnprintf(c->err, sizeof(c->err), fmt, ap); va_end(ap); }
    return code;
}
SweepArgs make_args(psgsdf_ctx* c, int laplacian_reg) {
    SweepArgs a{};
    a.b = c->band; a.frames = c->frames; a.F = c->F; a.cam = c->cam; a.grid = c->grid;
    a.im.f32 = c->img; a.im.u8 = c->img8; a.im.scale = c->img_scale;
    a.im.idx32 = (size_t)c->F * c->cam.W * c->cam.H * 12 < ((size_t)1 << 32) && (size_t)c->F * c->cam.H < ((size_t)1 << 24) && 3 * (size_t)c->cam.W < ((size_t)1 << 24);   // (24-bit index multiplies, device_common.h sample_cell)
    a.rob.loss = c->set.loss; a.rob.lambda = c->set.lambda; a.rob.lambda_sq = c->set.lambda * c->set.lambda; a.rob.inv_lambda = 1.0f / c->set.lambda;
    a.acc.frame = c->acc_frame; a.acc.part = c->part; a.acc.PB = c->PB;
    a.acc.fpart = c->frame_part; a.acc.fcap = c->frame_cap; a.acc.fdone = c->frame_done;
    a.fold.n = 0; a.gate = nullptr; a.fuse_apply = 0;
    a.xcd_map = c->xcd_map; a.xf = nullptr; a.xf_epoch = 0; a.fm_led_light = nullptr; a.vm_order = c->vm_order;
    a.fm_solve = 0; a.fm_frames = nullptr; a.fm_undo = nullptr; a.fm_e_out = nullptr; a.fm_e_key = 0;
    a.pcg_part = c->pcg_part; a.pcg_fs = c->pcg_sc; a.pcg_fuse_init = 0; a.pcg_init_blocks = 0; a.pcg_gran = nullptr; a.pcg_gran_n = 0; a.pcg_asm = 0; a.pcg_epoch = 0; a.pcg_pipe = c->pcg_pipeline ? (c->pcg_tagm ? (c->n_ranks <= 1 ? 2 : (c->pcg_tagm_mr ? 3 : 1)) : 1) : 0; a.pcg_apply = 0; a.pcg_xcd_local = (c->pcg_xcd_local ? 1 : 0) | (c->pcg_prefetch ? 2 : 0) | (c->pcg_ablate << 3);
    a.ar = c->ar; a.ar.weight = c->reg_r;
    a.model = c->set.model; a.quirks = c->set.ref_quirks;
    a.reg_n = c->reg_n; a.reg_l = c->reg_l;
    a.normal_reg = c->reg_n != 0.0f; a.laplacian_reg = laplacian_reg;
    a.damping = c->set.damping;
    a.row0 = c->row0; a.row1 = c->row1;
    a.ext = nullptr;
    return a;
}

// blocking variants
int read_parts(psgsdf_ctx* c, const int* slots, int n, double* out) {
    int rc = read_parts_deferred(c, slots, n, [out, n](const double* v) { for (int i = 0; i < n; ++i) out[i] = v[i]; });
    return rc ? rc : flush(c);
}
int read_frame_energy(psgsdf_ctx* c, int col_e, double* E, double* nobs) {
    int rc = read_frame_energy_deferred(c, col_e, [E, nobs](double e, double n) { *E = e; *nobs = n; });
    return rc ? rc : flush(c);
}
// Every deferred read-back is taken from the mailbox only when its values match their check words (device_common.h mbox_put): the marker of
// flush() and the status word of the persistent solve say that the producing kernels have RUN, not that their words have reached host
// memory -- a read-back that is still on its way (profiles/r03_notes.md section 1: the round-2 flake) is waited for, and counted.
bool readback_landed(const Deferred& d) {
    const volatile double* v = d.src;
    for (int i = 0; i < d.n; ++i) if ((dbits(v[i]) ^ dbits(v[d.n + i])) != d.key + (unsigned long long)i) return false;
    return true;
}
int deliver_first(psgsdf_ctx* c, size_t count, bool told_landed) {
    count = std::min(count, c->deferred.size());
    for (size_t e = 0; e < count; ++e) {
        Deferred& d = c->deferred[e];
        if (d.key && c->mbox_check) {
            c->mbox_checked++;
            if (!readback_landed(d)) {
                if (told_landed) c->mbox_late++;      // (a speculation window WAITS here for its closing energy: not an event)
                const Deferred* dp = &d;
                const int w = wait_mapped(c, [dp] { return readback_landed(*dp); }, "read-back");
                if (w < 0) return w;
                if (w == 1) { HIPCHK(c, hipStreamSynchronize(c->stream)); if (!readback_landed(d)) return fail(c, PSGSDF_ERR_DEVICE, "a scalar read-back never arrived (the kernel that owed it did not run)"); }
            }
        }
        d.consume(d.src);
    }
    c->deferred.erase(c->deferred.begin(), c->deferred.begin() + (long)count);
    if (c->deferred.empty()) c->mbox_used = 0;      // (slots are handed out again only when nothing is in flight)
    // a halo pull whose bounded wait expired hands on NaN rows -- which the distance solve may swallow (a CG on NaN sums never reports Success and the
    // reference's B8 rule then simply skips the update): the pull says so in a host-mapped word of its own, and the host looks at it here
    if (c->mbox && c->mbox_n && ((volatile double*)c->mbox)[c->mbox_n + 1] != 0.0) {
        const double ep = ((volatile double*)c->mbox)[c->mbox_n + 1];
        ((volatile double*)c->mbox)[c->mbox_n + 1] = 0.0;
        return fail(c, PSGSDF_ERR_DEVICE, "rank %d of %d: halo pull %.0f gave up -- a neighbour's halo rows never arrived within the bounded wait (2^%d polls); the halo rows were filled with NaN", c->rank, c->n_ranks, ep, (int)log2((double)c->xwait_spins));
    }
    if (c->xf_timeout) {
        c->xf_timeout = false;
        // which bounded wait expired is on record in this rank's region (kXrLate: written by the kernel whose wait expired); none: the NaN is the state's own
        double late[4] = {0, 0, 0, 0};
        if (c->xr) { (void)hipStreamSynchronize(c->stream); if (hipMemcpy(late, c->xr + kXrLate, sizeof(late), hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError(); }
        if (late[0] == 0 && late[1] == 0 && late[2] == 0)
            return fail(c, PSGSDF_ERR_DEVICE, "rank %d of %d: NaN came back from an exchange between the ranks' kernels, and no bounded wait of this rank expired: a peer handed on NaN (its wait expired, or the state itself is NaN)", c->rank, c->n_ranks);
        return fail(c, PSGSDF_ERR_DEVICE, "rank %d of %d: NaN came back from an exchange between the ranks' kernels -- a peer's contribution never arrived within the bounded wait (2^%d polls): frame rows of exchange %.0f (missing: rank %d's row of frame %d), scalar fold %.0f, halo pull %.0f (0 = not that one)",
                    c->rank, c->n_ranks, (int)log2((double)c->xwait_spins), late[0], (int)late[3] / 1000, (int)late[3] % 1000, late[1], late[2]);
    }
    return 0;
}
int deliver(psgsdf_ctx* c) { return deliver_first(c, c->deferred.size(), true); }
int mbox_reserve(psgsdf_ctx* c, int n, size_t* off, unsigned long long* key) {
    if (c->mbox_used + 2 * (size_t)n > c->mbox_n) { int rc = flush(c); if (rc) return rc; }
    *off = c->mbox_used; c->mbox_used += 2 * (size_t)n;
    *key = (++c->mbox_serial << 8) | 0x80u;      // never 0; keys of different reservations differ in every index they use (n <= 64)
    return 0;
}
// wait for every deferred read-back and run its consumer, in submission order
int flush(psgsdf_ctx* c) {
    materialize_fold(c);
    { int rc = mg_commit(c); if (rc) return rc; }
    // With check words on every read-back the host needs no marker kernel behind them (round 2: a marker + its two kernel boundaries per
    // flush): it watches the LAST read-back's words, then validates all of them in deliver().  Without read-backs in flight (or with
    // PSGSDF_MBOX_CHECK=0 / PSGSDF_PCG_POLL=0 / profiling) the stream is drained the ordinary way.
    bool synced = false;
    if (c->pcg_poll && !c->profiling && c->mbox && c->mbox_check && !c->deferred.empty() && c->deferred.back().key) {
        const Deferred& d = c->deferred.back();
        const volatile double* v = d.src; const int n = d.n; const unsigned long long key = d.key;
        const int w = wait_mapped(c, [v, n, key] { for (int i = 0; i < n; ++i) if ((dbits(v[i]) ^ dbits(v[n + i])) != key + (unsigned long long)i) return false; return true; }, "flush");
        if (w < 0) return w;
        synced = w == 0;
    } else if (c->pcg_poll && !c->profiling && c->mbox) {
        // a marker kernel + re-reading its mapped slot instead of hipStreamSynchronize: the runtime call itself is slower and makes the
        // next dispatch wait 5.8 us behind a system-scope fence (profiles/r01_notes.md, step p)
        const double seq = (c->flush_seq += 1.0);
        launch_marker(c->mbox_dev + c->mbox_n, seq, c->stream);
        volatile double* m = c->mbox + c->mbox_n;
        const int w = wait_mapped(c, [m, seq] { return *m == seq; }, "flush");
        if (w < 0) return w;
        synced = w == 0;
    }
    if (!synced) HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());                            // asynchronous launch failures surface here, not never
    return deliver(c);
}
// sum of per-workgroup partials of `slots`, delivered to `consume(sums)` at the next flush (no host sync here).
// The fold itself is left pending: the next kernel that can take it (take_fold) does it in its first workgroup;
// anything else that needs the value first (flush, a kernel that writes those slots) launches k_sum_parts.
void fold_by_kernel(psgsdf_ctx* c, FoldReq& f) {
    if (!f.n) return;
    SlotList sl; sl.n = f.n; for (int i = 0; i < sl.n; ++i) sl.id[i] = f.id[i];
    launch_sum_parts(c->part, c->PB, f.nblk, sl, f.out, f.key, c->stream, f.xf, f.xf_epoch);
    f.n = 0;
}
void materialize_fold(psgsdf_ctx* c) { fold_by_kernel(c, c->pending_fold); }
// Multi-rank: the scalar read-backs staged in the mailbox shadow since the last commit are summed over the ranks in ONE all-reduce (the
// contiguous range that spans them; slots in between belong to values that are global already -- frame-row sums, PCG status -- and are
// not copied) and land in their mailbox slots.  Every rank runs the same control flow on the same global values, so the collectives
// match up across ranks by construction.  Called wherever the host is about to wait on the mailbox (flush, pcg_solve).
int mg_commit(psgsdf_ctx* c) {
    if (!slab_mode(c) || c->mg_segs.empty()) return 0;
    materialize_fold(c);
    unsigned lo = ~0u, hi = 0;
    for (const MgSeg& g : c->mg_segs) { lo = std::min(lo, g.off); hi = std::max(hi, g.off + g.n); }
    int rc = comm_allreduce(c, c->mbox_shadow + lo, (int)(hi - lo)); if (rc) return rc;
    for (size_t i = 0; i < c->mg_segs.size(); i += 16) {
        CopySegs cs{}; cs.n = (int)std::min<size_t>(16, c->mg_segs.size() - i);
        for (int q = 0; q < cs.n; ++q) { cs.off[q] = c->mg_segs[i + q].off; cs.len[q] = c->mg_segs[i + q].n; cs.key[q] = c->mg_segs[i + q].key; }
        launch_copy_segs(c->mbox_shadow, c->mbox_dev, cs, c->stream);
    }
    c->mg_segs.clear();
    return 0;
}
// This context owns the global z-planes [z0, z1) and holds them plus one halo plane on each inner side: the LOCAL grid every dense kernel
// works on.  World coordinates stay those of the whole volume (GridP.origin is global, GridP.koff shifts the local plane index).
int set_local_grid(psgsdf_ctx* c, int z0, int z1) {
    if (z0 < 0 || z1 > c->gdim[2] || z1 <= z0) return fail(c, PSGSDF_ERR_ARG, "slab planes [%d, %d) of %d", z0, z1, c->gdim[2]);
    c->z0 = z0; c->z1 = z1;
    c->zlo = std::max(0, z0 - 1); c->zhi = std::min(c->gdim[2], z1 + 1);
    GridP& g = c->grid;
    g.dim[0] = c->gdim[0]; g.dim[1] = c->gdim[1]; g.dim[2] = c->zhi - c->zlo;
    g.nvox = (long long)g.dim[0] * g.dim[1] * g.dim[2];
    g.koff = c->zlo;
    return 0;
}
// hand the pending fold to a kernel about to be launched with `a`; `writes` = bit mask of the partial slots that kernel writes
void take_fold(psgsdf_ctx* c, SweepArgs& a, unsigned writes) {
    a.fold.n = 0;
    if (!c->pending_fold.n) return;
    if (c->row1 <= c->row0) { materialize_fold(c); return; }      // empty band: the sweeps are not launched at all, the read-back is still owed
    for (int i = 0; i < c->pending_fold.n; ++i) if (writes & (1u << c->pending_fold.id[i])) { materialize_fold(c); return; }
    a.fold = c->pending_fold; c->pending_fold.n = 0;
}
int read_parts_deferred(psgsdf_ctx* c, const int* slots, int n, std::function<void(const double*)> consume) {
    materialize_fold(c);
    size_t off; unsigned long long key;
    { int rc = mbox_reserve(c, n, &off, &key); if (rc) return rc; }
    // multi-rank: this slab's sums go to the device shadow of the mailbox; mg_commit all-reduces them and fills the mailbox slots (values + check words)
    // multi-rank, where the ranks' mailbox regions are mapped (round 4): the folding thread itself exchanges the slab's sums with the other ranks
    // (device_common.h fold_exchange) and writes the GLOBAL sums straight into the mailbox, check words and all -- as a single-rank context does
    const bool xs = xs_active(c) && n <= 8;
    const bool staged = slab_mode(c) && !xs;
    double* dst = (staged ? c->mbox_shadow : c->mbox_dev) + off;
    const unsigned long long dkey = staged ? 0ull : key;
    if (staged) c->mg_segs.push_back({(unsigned)off, (unsigned)n, key});
    const XfTable* xf = xs ? c->xf_table : nullptr;
    const long long ep = xs ? ++c->xs_epoch : 0;
    if (n <= 4 && c->fold_in_next) {
        c->pending_fold.n = n; for (int i = 0; i < n; ++i) c->pending_fold.id[i] = slots[i];
        c->pending_fold.nblk = band_blocks(c); c->pending_fold.out = dst; c->pending_fold.key = dkey; c->pending_fold.xf = xf; c->pending_fold.xf_epoch = ep;
    } else {
        SlotList sl; sl.n = n; for (int i = 0; i < n; ++i) sl.id[i] = slots[i];
        launch_sum_parts(c->part, c->PB, band_blocks(c), sl, dst, dkey, c->stream, xf, ep);
    }
    if (xs) c->deferred.push_back({c->mbox + off, n, key, [c, consume](const double* v) { if (std::isnan(v[0])) c->xf_timeout = true; consume(v); }});      // (NaN: a rank's sums never arrived, fold_exchange)
    else c->deferred.push_back({c->mbox + off, n, key, std::move(consume)});
    return 0;
}
int read_frame_energy_deferred(psgsdf_ctx* c, int col_e, std::function<void(double, double)> consume) {
    size_t off; unsigned long long key;
    { int rc = mbox_reserve(c, 2, &off, &key); if (rc) return rc; }
    launch_frame_cols(c->acc_frame, c->F, col_e, c->mbox_dev + off, key, c->stream);
    c->deferred.push_back({c->mbox + off, 2, key, [consume](const double* v) { consume(v[0], v[1]); }});
    return 0;
}
// deferred variant without a kernel of its own: the solve kernel that follows the sweep writes the two sums to *dev_slot
int reserve_frame_energy_deferred(psgsdf_ctx* c, std::function<void(double, double)> consume, double** dev_slot, unsigned long long* key_out) {
    size_t off; unsigned long long key;
    { int rc = mbox_reserve(c, 2, &off, &key); if (rc) return rc; }
    *dev_slot = c->mbox_dev + off; *key_out = key;
    c->deferred.push_back({c->mbox + off, 2, key, [consume](const double* v) { consume(v[0], v[1]); }});
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
    if (!c->deferred.empty() || c->pending_fold.n) { int rc = flush(c); if (rc) return rc; }   // (read-backs of the band that is about to be replaced)
    // multi-rank: the neighbours map this rank's record planes (cross-rank persistent solve); every rank closes its mappings and tells the
    // owners, and nothing is freed before every rank that mapped THIS rank's planes has said so (comm.hip xr_quiesce: flags, no collective).  The first band of a multi-rank context also chooses the memory kind of those planes (comm.hip xr_probe).
    if (xr_quiesce(c, 120.0)) return fail(c, PSGSDF_ERR_COMM, "band rebuild: a rank never closed its mappings of this rank's record planes (lost or failed peer?)");
    if (c->n_ranks > 1 && c->comm) { int rc = xr_probe(c); if (rc) return rc; }
    c->persist_off = false;          // a persistent solve that gave up did so on the previous band's launch shape / mappings: this band tries again
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
    const size_t n4 = 1 + 1 + 3 + 3 + 6 + 6 + kNQ + 12 + 6 + 14 + kNQ + 4 + 9 + 1 + 3;   // (xs, gn, nfd live in the packed planes vp: 12 instead of 9 words; the two record planes are an allocation of their own)
    const size_t bytes = (n4 * 4 + (size_t)KW * 8) * Spad + 256;
    if (c->band_mem) { hipFree(c->band_mem); c->band_mem = nullptr; }
    if (c->rec_mem) { hipFree(c->rec_mem); c->rec_mem = nullptr; }
    HIPCHK(c, hipMalloc(&c->band_mem, bytes));
    HIPCHK(c, hipMemsetAsync(c->band_mem, 0, bytes, c->stream));
    { int rc = xr_alloc(c, &c->rec_mem, 2 * sizeof(float4) * (size_t)Spad, false); if (rc) return rc; }      // (fine-grained / uncached on a multi-rank context: the neighbours' kernels write its halo rows)
    HIPCHK(c, hipMemsetAsync(c->rec_mem, 0, 2 * sizeof(float4) * (size_t)Spad, c->stream));
    c->band_bytes = bytes;
    char* p = (char*)c->band_mem;
    auto take = [&](size_t elems, size_t esz) { void* r = p; p += elems * esz * Spad; return r; };
    Band& b = c->band;
    b.S = S; b.Spad = Spad; b.KW = KW;
    b.vis = (uint64_t*)take(KW, 8);
    b.rec[0] = (float4*)c->rec_mem; b.rec[1] = b.rec[0] + Spad;
    b.lin = (int*)take(1, 4);
    b.dist = (float*)take(1, 4);
    for (int a = 0; a < 3; ++a) b.g[a] = (float*)take(1, 4);
    for (int a = 0; a < 3; ++a) b.rho[a] = (float*)take(1, 4);
    b.nb = (int*)take(6, 4); b.nbd = (float*)take(6, 4); b.col = (int*)take(kNQ, 4); b.colp = (unsigned*)take(9, 4); b.dirb = (int*)take(1, 4);
    for (int a = 0; a < 3; ++a) b.gfd[a] = (float*)take(1, 4);
    for (int a = 0; a < 3; ++a) b.vp[a] = (float4*)take(4, 4);
    b.aH = (float*)take(3, 4); b.ab = (float*)take(3, 4);
    b.blk = (float*)take(14, 4); b.H = (float*)take(kNQ, 4);
    b.hx = (int*)take(1, 4);
    b.rhs = (float*)take(1, 4); b.x = (float*)take(1, 4); b.t = (float*)take(1, 4);
    if ((size_t)(p - (char*)c->band_mem) > bytes) return fail(c, PSGSDF_ERR_DEVICE, "band arena overrun: %zu > %zu bytes", (size_t)(p - (char*)c->band_mem), bytes);
    HIPCHK(c, hipMemsetAsync(c->d_total, 0, sizeof(int), c->stream));
    timed(c, "band_fill", [&] { launch_band_fill(c->dense, c->grid, b, c->d_total, c->stream); });
    {   // 16-bit column deltas are usable if the widest reach of any ELL column fits (PSGSDF_PCG_COL16=0 forces the 32-bit table)
        int reach = 0;
        HIPCHK(c, hipMemcpyAsync(&reach, c->d_total, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        b.col16 = reach <= 32767; b.reach = reach;
        if (const char* e = getenv("PSGSDF_PCG_COL16")) if (atoi(e) == 0) b.col16 = 0;
    }
    // row partition.  The band is sorted by linear index, z slowest, so the rows of the OWN planes [z0, z1) are one contiguous range
    // [row0, row1); the rows before it are the lower halo plane, the rows behind it the upper one (multi-rank only).
    {
        c->row0 = 0; c->row1 = S;
        c->halo = 0; c->need[0] = c->need[1] = 0; c->give[0] = c->give[1] = 0; c->halo_active = false;
        if (c->n_ranks > 1 && !c->comm) return fail(c, PSGSDF_ERR_COMM, "rank %d of %d has no communicator (psgsdf_comm_init)", c->rank, c->n_ranks);
        if (3 * c->n_ranks > kMgScal) return fail(c, PSGSDF_ERR_UNSUPPORTED, "at most %d ranks", kMgScal / 3);      // (the set-up exchange below carries 3 doubles per rank)
        if (c->n_ranks > 1) {
            const long long plane = (long long)c->grid.dim[0] * c->grid.dim[1];
            int rr[2] = {0, S};
            if (S > 0) {
                launch_lower_bound(b.lin, S, (int)((c->z0 - c->zlo) * plane), c->d_need, c->stream);
                launch_lower_bound(b.lin, S, (int)((c->z1 - c->zlo) * plane), c->d_need + 1, c->stream);
                HIPCHK(c, hipMemcpyAsync(rr, c->d_need, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
            }
            c->row0 = rr[0]; c->row1 = rr[1];
            c->need[0] = c->row0; c->need[1] = S - c->row1; c->halo = std::max(c->need[0], c->need[1]);
            // {need_lo, need_hi, own rows} of every rank in one tiny all-reduce: what a slab SENDS is its neighbours' need
            std::vector<double> info(3 * (size_t)c->n_ranks, 0.0);
            info[3 * c->rank] = c->need[0]; info[3 * c->rank + 1] = c->need[1]; info[3 * c->rank + 2] = c->row1 - c->row0;
            HIPCHK(c, hipMemcpyAsync(c->mg_scal, info.data(), sizeof(double) * info.size(), hipMemcpyHostToDevice, c->stream));
            int rcc = comm_allreduce(c, c->mg_scal, (int)info.size()); if (rcc) return rcc;
            HIPCHK(c, hipMemcpyAsync(info.data(), c->mg_scal, sizeof(double) * info.size(), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const int own = c->row1 - c->row0;
            if (c->rank > 0) c->give[0] = std::min((int)info[3 * (c->rank - 1) + 1], own);               // the lower neighbour's upper halo plane = my first own plane
            if (c->rank < c->n_ranks - 1) c->give[1] = std::min((int)info[3 * (c->rank + 1)], own);       // the upper neighbour's lower halo plane = my last own plane
            c->S_global = 0;
            for (int r = 0; r < c->n_ranks; ++r) {
                if (info[3 * r] != 0.0 || info[3 * r + 1] != 0.0) c->halo_active = true;                 // (global: every rank takes part in an exchange or none does)
                if (info[3 * r + 2] == 0.0) return fail(c, PSGSDF_ERR_UNSUPPORTED, "rank %d of %d owns no band rows: use fewer ranks", r, c->n_ranks);
                c->S_global += (long long)info[3 * r + 2];
            }
            // the stencil-direction bits of the halo rows are static and depend on the plane BEYOND the halo: take them from their owner
            int hrc = comm_halo(c, b.dirb, 1, 1); if (hrc) return hrc;
            if ((hrc = xr_setup(c, info))) return hrc;      // cross-rank persistent solve: map the neighbours' halo rows and every rank's mailbox region
        } else c->S_global = S;
    }
    if (c->areg_mem) { hipFree(c->areg_mem); c->areg_mem = nullptr; c->ar = AlbedoReg{}; }
    if (c->reg_r != 0.f) {   // "reg albedo": stencil tables + matrix-free CG vectors over the 3S unknowns
        const size_t planes = 3 + 1 + 9 + 12 + 3 + 8 * 3 + 1;      // (+ 1: the device-driven CG's scalars)
        HIPCHK(c, hipMalloc(&c->areg_mem, planes * 4 * (size_t)Spad));
        HIPCHK(c, hipMemsetAsync(c->areg_mem, 0, planes * 4 * (size_t)Spad, c->stream));
        char* q = (char*)c->areg_mem;
        auto tk = [&](size_t n) { void* r = q; q += n * 4 * (size_t)Spad; return r; };
        AlbedoReg& ar = c->ar;
        ar.anb = (int*)tk(3); ar.back = (int*)tk(1); ar.anrho = (float*)tk(9); ar.J = (float*)tk(12); ar.res = (float*)tk(3);
        ar.rhs = (float*)tk(3); ar.diag = (float*)tk(3); ar.diag0 = (float*)tk(3); ar.x = (float*)tk(3); ar.r = (float*)tk(3); ar.p = (float*)tk(3); ar.q = (float*)tk(3); ar.t = (float*)tk(3); ar.cgs = (double*)tk(1);
        SweepArgs at{}; at.b = b; at.ar = ar;
        launch_areg_tables(c->dense, c->grid, at, c->stream);
        // which side a halo row's stencil takes depends on the plane beyond the halo: its owner knows
        if (int hrc = comm_halo(c, ar.back, 1, 1)) return hrc;
    }
    // per-frame observation lists of the owned rows (counts -> host prefix -> fill)
    {
        const int nch = (c->row1 - c->row0 + kObsChunk - 1) / kObsChunk, F = c->F;
        if (c->obs_mem) { hipFree(c->obs_mem); c->obs_mem = nullptr; }
        b.obs_ptr = nullptr; b.obs_rows = nullptr; b.obs_max = 0; b.obs_ptr_total = 0;
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
            b.obs_ptr = (int*)c->obs_mem; b.obs_rows = b.obs_ptr + (F + 1); b.obs_max = mx; b.obs_ptr_total = run;
            HIPCHK(c, hipMemcpyAsync(b.obs_ptr, ptr.data(), sizeof(int) * (F + 1), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(d_counts, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice, c->stream));
            launch_obs_fill(b, F, c->row0, c->row1, d_counts, c->stream);
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipFree(d_counts);
        }
    }
    {   // dispatch order of the voxel-major distance sweep: blocks sorted by their work, heaviest first (the sweep needs 1.3 generations of resident
        // workgroups: what is dispatched last should be short).  Logical ids carry rows and partial-sum slots: results do not depend on the order.
        if (c->vm_order) { hipFree(c->vm_order); c->vm_order = nullptr; }
        const int nb = (c->row1 - c->row0 + kBlock - 1) / kBlock;
        if (nb > 1) {
            int* d_work = nullptr;
            HIPCHK(c, hipMalloc(&d_work, sizeof(int) * nb));
            launch_block_work(b, c->row0, c->row1, d_work, c->stream);
            std::vector<int> work(nb), order(nb);
            HIPCHK(c, hipMemcpyAsync(work.data(), d_work, sizeof(int) * nb, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (int i = 0; i < nb; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return work[x] > work[y]; });
            HIPCHK(c, hipMalloc(&c->vm_order, sizeof(int) * nb));
            HIPCHK(c, hipMemcpyAsync(c->vm_order, order.data(), sizeof(int) * nb, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipFree(d_work);
        }
    }
    // a slab without a single observation launches no frame-major sweep and could not deliver its rows to the other ranks from inside one (loop.hip
    // step_begin: the in-sweep exchange): every rank has to know
    c->any_empty_slab = false;
    if (c->n_ranks > 1 && c->comm) {
        std::vector<double> e(1, (c->row1 <= c->row0 || b.obs_max <= 0) ? 1.0 : 0.0);
        if (int hrc = host_allreduce(c, e, "empty slabs")) return hrc;
        c->any_empty_slab = e[0] != 0.0;
    }
    {   // partial rows of the frame-major sweeps: one per workgroup and frame; a sweep uses at most ceil(obs_max / (256 * 4)) workgroups per frame
        if (c->frame_part) { hipFree(c->frame_part); c->frame_part = nullptr; }
        if (c->frame_done) { hipFree(c->frame_done); c->frame_done = nullptr; }
        c->frame_cap = (b.obs_max + kBlock * 4 - 1) / (kBlock * 4) + 1;
        const int F = std::max(c->F, 1);
        HIPCHK(c, hipMalloc(&c->frame_part, sizeof(double) * (size_t)F * c->frame_cap * kFrameRow));
        HIPCHK(c, hipMalloc(&c->frame_done, sizeof(int) * (F + 1)));      // (+ 1: the sweep's finished-frames counter, frame_rows_publish fm_solve)
        HIPCHK(c, hipMemsetAsync(c->frame_done, 0, sizeof(int) * (F + 1), c->stream));
    }
    if (c->part) { hipFree(c->part); c->part = nullptr; }
    c->PB = Spad / kBlock + 1;
    HIPCHK(c, hipMalloc(&c->part, sizeof(double) * SC_COUNT * c->PB));
    HIPCHK(c, hipMemsetAsync(c->part, 0, sizeof(double) * SC_COUNT * c->PB, c->stream));
    {
        const size_t need = 4096;
        if (need > c->mbox_alloc) {
            if (c->mbox) hipHostFree(c->mbox);
            c->mbox = nullptr; c->mbox_n = 0; c->mbox_alloc = 0;
            HIPCHK(c, hipHostMalloc(&c->mbox, sizeof(double) * need, hipHostMallocMapped));
            HIPCHK(c, hipHostGetDevicePointer((void**)&c->mbox_dev, c->mbox, 0));
            c->mbox_alloc = need; c->mbox_n = need - 2;      // [mbox_n] = flush marker, [mbox_n + 1] = the "a halo pull gave up" word (comm.hip k_halo_pull)
            memset(c->mbox, 0, sizeof(double) * need); c->flush_seq = 0;
        }
        if (slab_mode(c) && !c->mbox_shadow) {
            HIPCHK(c, hipMalloc(&c->mbox_shadow, sizeof(double) * c->mbox_alloc));
            HIPCHK(c, hipMemsetAsync(c->mbox_shadow, 0, sizeof(double) * c->mbox_alloc, c->stream));
        }
        c->mbox_used = 0; c->deferred.clear(); c->mg_segs.clear();
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


int ps_energy(psgsdf_ctx* c, double* E, int64_t* nobs) {
    SweepArgs a = make_args(c, 0);
    take_fold(c, a, (1u << SC_ENERGY) | (1u << SC_NOBS) | (1u << SC_AUX0) | (1u << SC_AUX1) | (1u << SC_AUX2));
    timed(c, "energy", [&] { launch_energy(a, c->stream); });
    const int slots[2] = {SC_ENERGY, SC_NOBS}; double s[2];
    int rc = read_parts(c, slots, 2, s); if (rc) return rc;
    *E = band_mean(c, s[0]); if (nobs) *nobs = (int64_t)s[1];
    return 0;
}

}  // namespace psge
