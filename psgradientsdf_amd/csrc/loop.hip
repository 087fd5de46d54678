// loop.hip -- the solves and the alternation loop: launch shape and host driver of the fused PCG, the regularised albedo solve,
// the four sub-steps (sweep + solve + update), the control flow of psgsdf_iterate / psgsdf_optimize, the 2x refinement.
#include "engine_internal.h"
#include <climits>

namespace psge {

// Launch shape of the fused PCG pass.  The pass is a chain of memory round trips per workgroup (coefficients + column
// indices -> two batches of record gathers, the second overlapping the reduction), so what matters is how many rows have
// their loads in flight at once.  One row per thread at ≤128 VGPRs keeps 4 waves per SIMD resident (1024 workgroups); the
// reduction of the previous pass's partials costs every workgroup G x 7 doubles, which caps G at 768.  Measured on the
// 256^3 band (1317 row-blocks): 659 workgroups x 2 trips 14.4 us, 512 x 3 trips 15.6 us (17.3 / 18.8 us before the DPP reductions)
// (tools/pcg_ablate.py, profiles/r01_notes.md).
constexpr int kCgfSumsHost = 7;   // = kCgfSums (device_common.h): P, B, C, D, E, Z, R of a pass
void cgf_shape(int nblk, int* G, int* rows) {
    nblk = std::max(1, nblk);
    int r = 1, cap = kCgfMaxBlocks;
    if (const char* e = getenv("PSGSDF_PCG_ROWS")) { int v = atoi(e); if (v >= 1 && v <= 2) r = v; }       // tuning knobs
    if (const char* e = getenv("PSGSDF_PCG_BLOCKS")) { int v = atoi(e); if (v > 0 && v <= kCgfMaxBlocks) cap = v; }
    const int per = (nblk + r - 1) / r;                 // workgroups if every thread took r rows once
    const int trips = (per + cap - 1) / cap;
    *G = (per + trips - 1) / trips; *rows = r;
}

// Shape of the persistent solve (pcg.hip: k_cgf_solve): G workgroups of 512 threads, at most one per CU (all co-resident), the band dealt
// evenly to them (<= 4 rows per thread).  One rank only -- a slab needs the other slabs' sums every pass,
// which is the per-pass kernels' all-reduce -- and only with 16-bit column deltas and the assembly kernel's fused initialisation.
bool cgf_solve_shape(psgsdf_ctx* c, int* G, int* rows_per_wg, bool any_ranks) {
    // (multi-rank: only with the cross-rank mappings of comm.hip xr_setup in place -- all ranks or none, agreed there)
    if (!c->pcg_persist || c->persist_off || (c->n_ranks > 1 && !c->xr_ready && !any_ranks) || !c->band.col16 || !c->fuse_pcg_init || c->num_cu <= 0 || band_blocks(c) > kPcgMaxBlocks) return false;
    const int n = c->row1 - c->row0, cap = std::min(c->num_cu, kSolveMaxBlocksHost);
    if (n <= 0) return false;
    // as many workgroups as CUs (in multiples of 8: every XCD owns a contiguous range of rows) unless the band is so small that a workgroup
    // would get less than one row per thread; the band is dealt evenly, in multiples of 64 rows
    int g = std::min(cap, (n + kSolveThreadsHost - 1) / kSolveThreadsHost);
    g = std::max(8, g / 8 * 8);
    if (g > cap) g = cap;
    const int per = ((n + g - 1) / g + 63) / 64 * 64;
    const int r = (per + kSolveThreadsHost - 1) / kSolveThreadsHost;
    if (r > kSolveMaxRowsHost || cgf_solve_max_blocks(r) < 1) return false;
    *G = g; *rows_per_wg = per;
    return true;
}

// Fused PCG (pcg.hip: k_cgf_pass): kernel k finishes pass k-1 and runs pass k, so a chunk of n kernels tells the host
// about the passes up to k0+n-2; the kernel that detects convergence (or hits the cap) is also the one that finalises x.
// `tail(gate)`, if given, enqueues what follows a finished solve (distance update + regrad) right behind every chunk of passes,
// gated on the device-side 'solve finished' flag: when the chunk converges -- the normal case -- the GPU runs it without
// waiting for the host to notice; when it does not, the gated kernels do nothing and the tail is enqueued again behind the
// next chunk.  *tail_ran tells the caller whether the enqueued tail is the one that took effect.
int pcg_solve(psgsdf_ctx* c, SweepArgs& a, int* iters_out, int* success_out, double* err_out,
              const std::function<void(const double*)>& tail, bool gate_on_converged, bool* tail_ran) {
    const int S = c->band.S;
    if (tail_ran) *tail_ran = false;
    if (a.row1 <= a.row0) { *iters_out = 0; *success_out = 1; *err_out = 0; c->last_cg_iters = 0; return 0; }   // empty band: b = 0, x = 0, Success
    int cap = c->set.cg_max_it > 0 ? c->set.cg_max_it : 2 * S;
    if (cap > c->pcg_cap) cap = c->pcg_cap;
    int G, rows;
    if (a.pcg_gran && cgf_solve_shape(c, &G, &rows)) {
        // ---- the whole solve as one persistent kernel: nothing for the host to decide until it is over, so the distance update and
        // the regrad are enqueued right behind it (gated on the device-side outcome) and the host only picks up the statistics
        // the fold of the distance sweep's sums that this kernel would do in its prologue has to be on the stream BEFORE anything that consumes
        // its result: a flush (mailbox full) validates and delivers it, mg_commit (a communicator, also a one-rank one) all-reduces and copies it
        SweepArgs as = a;
        auto fold_now = [&] { fold_by_kernel(c, as.fold); };
        if (slab_mode(c) && !xs_active(c)) fold_now();      // (with the in-kernel exchange of the folds the solve's own prologue does it, as on one rank)
        if (c->mbox_used + (size_t)kSolveMbSlots > c->mbox_n) { fold_now(); int rc = flush(c); if (rc) return rc; }
        { int rc = mg_commit(c); if (rc) return rc; }      // (a one-rank communicator: the read-backs staged so far are delivered below)
        const size_t off = c->mbox_used; c->mbox_used += kSolveMbSlots;
        volatile double* st = c->mbox + off;
        const unsigned long long key = (++c->mbox_serial << 8) | 0x80u;
        st[3] = NAN; st[4] = 0.0;
        const int inject = (c->fault_solve > 0 && ++c->solves_seen == c->fault_solve) ? -7 : 0;      // PSGSDF_FAULT_SOLVE=n: one workgroup of the n-th solve stops publishing (tests the fallback below)
        const XrArgs* xr = (c->n_ranks > 1 && c->xr_ready && as.pcg_asm) ? &c->xr_args : nullptr;
        if (xr) {
            // the solve's epoch (the same on every rank: all ranks run the same solves) goes into every cross-rank tag, so nothing in the mailbox
            // regions is ever cleared between solves.  When the 14-bit epoch wraps, every region is cleared once behind an all-rank barrier
            // (a word that has sat unrewritten for 16 384 solves must not match again).
            c->xr_solves++;
            c->xr_args.epoch = (unsigned)(c->xr_solves & (long long)kXrEpochMask);
            if (c->xr_args.epoch == 0) {
                HIPCHK(c, hipMemsetAsync(c->xr, 0, sizeof(double) * kXrDoubles, c->stream));
                HIPCHK(c, hipMemsetAsync(c->mg_ext, 0, sizeof(double), c->stream));
                int rc = comm_allreduce(c, c->mg_ext, 1); if (rc) return rc;      // (a rank's kernel writes into another's region only behind this: every memset is ordered before its owner's contribution)
            }
        }
        as.pcg_epoch = ++c->pcg_solve_serial;
        timed(c, "pcg_solve", [&] { launch_cgf_solve(as, c->pcg_sc, c->pcg_gran, G, rows, cap, c->mbox_dev + off, key, inject, c->stream, xr); });
        if (tail && !c->profiling) { tail(c->pcg_sc + (gate_on_converged ? 2 : 1)); if (tail_ran) *tail_ran = true; }
        // the four status words are taken only together with their check word (engine.h FoldReq)
        const bool chk = c->mbox_check;
        auto landed = [st, key, chk] {
            const double s3 = st[3];
            if (std::isnan(s3)) return false;
            return !chk || (dbits(st[0]) ^ dbits(st[1]) ^ dbits(st[2]) ^ dbits(s3) ^ dbits(st[4])) == key;
        };
        const int w = wait_mapped(c, landed, "pcg_solve");
        if (w < 0) return w;
        if (!landed()) HIPCHK(c, hipStreamSynchronize(c->stream));      // ("drained" came from a stream query: drain for certain before calling it a failure)
        if (!landed()) return fail(c, PSGSDF_ERR_DEVICE, "the PCG kernel published nothing");
        if (w == 0 && !c->pending_fold.n) { int rc = deliver(c); if (rc) return rc; }   // everything enqueued before the solve has run: its read-backs are validated and taken
        if (st[3] != 1.0) {
            // The persistent kernel needs all its workgroups co-resident (no cooperative launch) and gave up waiting for some of them: something
            // else holds CUs of this device (another process, a CU mask).  Nothing has been applied (its epilogue only acts on status 1 and it
            // leaves both gates closed, so the tail enqueued behind it did nothing): redo this solve with the per-pass kernels, which make
            // no residency assumption, and keep this context on them until the next band is built.  NOT the same bits: the per-pass kernels run
            // Eigen's classic recurrences in float, the persistent kernel the pipelined ones in double (deviation 3) -- the two agree to
            // rounding (tests/test_knobs_gpu.py, tests/test_edge_gpu.py::test_persistent_solve_falls_back...: 2e-6 in the energies, 5e-5 voxel), so a
            // run that fell back is reproducible only given the same fall-back; psgsdf_debug_sync_stats out[2] / the bench line's `degraded` say so.
            // Multi-rank: the decision is the same on every rank without asking -- a rank that gives up in pass j never publishes the sums of
            // pass j, so no rank can obtain them and finish; a rank CAN only reach its last pass when every rank has published everything that
            // pass needs (records before sums), and the last pass itself waits for nothing.  The abort flag only shortens the others' waits.
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (!c->persist_fallbacks++) fprintf(stderr, "psgsdf: the persistent distance solve gave up waiting (status %g, %d workgroups%s): %s  Continuing with the per-pass kernels.\n", (double)st[3], G, xr ? ", cross-rank" : "",
                                                 xr ? "a neighbour rank's hand-off did not arrive in time." : "is another process holding CUs of this device?");
            c->persist_off = true;
            a.pcg_asm = 0; a.pcg_apply = 0; a.pcg_gran = nullptr; a.pcg_gran_n = 0; a.fold.n = 0;      // (the sweep's sums were folded by the kernel's prologue / by k_assemble)
            timed(c, "assemble", [&] { launch_assemble(a, c->stream); });      // H, rhs, x = 0, the initial records (the attempt overwrote them) and |b|^2 in memory, status words reset
            if (tail_ran) *tail_ran = false;
            return pcg_solve(c, a, iters_out, success_out, err_out, tail, gate_on_converged, tail_ran);
        }
        const float rhsN = (float)st[2];
        if (rhsN == 0.f) { *iters_out = 0; *success_out = 1; *err_out = 0; c->last_cg_iters = 0; return 0; }
        const double err = sqrt((double)(float)st[1] / (double)rhsN);
        *iters_out = (int)st[0]; *err_out = err; *success_out = err <= (double)FLT_EPSILON;
        c->last_cg_iters = *iters_out;
        return 0;
    }
    cgf_shape(band_blocks(c), &G, &rows);
    if (!a.pcg_fuse_init) timed(c, "pcg_init", [&] { launch_cgf_init(a, c->pcg_sc, c->pcg_part, G, c->stream); });
    // Multi-rank (z-slabs): the same fused kernel, reading the globally reduced sums of the previous pass from `ext` instead of its own
    // partials.  Per pass: halo rows of the records it gathers from, the pass, a 1-workgroup fold of its partials, ONE all-reduce of 7 doubles.
    // Every rank publishes the same values to its mailbox, so all ranks take the same decisions below.
    const bool mr = slab_mode(c);
    SweepArgs ap = a;
    if (mr) {
        int rc = mg_commit(c); if (rc) return rc;      // the read-backs staged so far are delivered when the first slot of this solve is seen
        launch_cgf_sum(c->pcg_part, a.pcg_fuse_init ? a.pcg_init_blocks : G, -1, c->mg_ext, c->stream);      // local |b|^2
        if ((rc = comm_allreduce(c, c->mg_ext, 1))) return rc;
        ap.ext = c->mg_ext;
    }
    // first chunk sized from the previous solve (the count is stable between Gauss-Newton iterations)
    int chunk = std::min(64, std::max(4, c->last_cg_iters + 2));
    int k = 0, iters = -1;            // k = next kernel index; kernels 0..cap exist (kernel cap only finalises)
    float rhsN = 0, rn2_last = 0, threshold = 0;
    while (true) {
        const int n = std::min(chunk, cap + 1 - k);
        if (c->mbox_used + (size_t)n > c->mbox_n) { int rc = flush(c); if (rc) return rc; }
        const size_t off = c->mbox_used; c->mbox_used += n;
        volatile double* st = c->mbox + off;
        for (int q = 0; q < n; ++q) st[q] = -1.0;      // "not published yet" (kernel q of the chunk overwrites its slot with a squared norm: >= 0, or NaN)
        for (int q = 0; q < n; ++q) {
            if (mr) { int rc = comm_halo(c, c->band.rec[(k + q + 1) & 1], 1, 4); if (rc) return rc; }
            timed(c, "pcg_pass", [&] { launch_cgf_pass(ap, c->pcg_sc, c->pcg_part, G, rows, k + q, cap, c->mbox_dev + off + q, c->stream); });
            if (mr) {
                launch_cgf_sum(c->pcg_part, G, k + q, c->mg_ext, c->stream);
                int rc = comm_allreduce(c, c->mg_ext, kCgfSumsHost); if (rc) return rc;
            }
        }
        if (tail && c->pcg_poll && !c->profiling) { tail(c->pcg_sc + (gate_on_converged ? 2 : 1)); if (tail_ran) *tail_ran = true; }
        // Watch the mapped slots instead of waiting for the stream to drain: the kernel that detects convergence publishes
        // at its START, so the host learns the outcome while that kernel and the surplus (no-op) kernels of the chunk are
        // still running, and enqueues the rest of the iteration behind them without a bubble.
        bool drained = !c->pcg_poll;
        if (drained) { int rc = flush(c); if (rc) return rc; }
        for (int q = 0; q < n && iters < 0; ++q) {
            const int kk = k + q;
            if (!drained && st[q] == -1.0) {
                const int w = wait_mapped(c, [st, q] { return st[q] != -1.0; }, "pcg_solve");
                if (w < 0) return w;
                drained = w == 1;                              // nothing left that could publish
            }
            double v = st[q];
            if (v == -1.0) {      // "drained" came from a stream query: before calling it a failure, drain for certain and look again
                HIPCHK(c, hipStreamSynchronize(c->stream));
                v = st[q];
            }
            if (v == -1.0) return fail(c, PSGSDF_ERR_DEVICE, "PCG kernel %d published nothing", kk);
            if (std::isnan(v)) return fail(c, PSGSDF_ERR_DEVICE, "PCG kernel %d published NaN: the distance system (or the sums another rank contributed) contains NaN", kk);
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
        if (!drained) { int rc = deliver(c); if (rc) return rc; }   // every deferred read-back enqueued before the chunk has been produced (in-order stream): validate and take them
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
    int rc = comm_halo(c, c->band.rho[0], 3, 1); if (rc) return rc;      // multi-rank: the stencils of the rows at a cut read the neighbour slab's albedo
    launch_areg_build(a, c->stream);
    const int slots[1] = {SC_AUX0}; double s[1];
    if ((rc = read_parts(c, slots, 1, s))) return rc;
    c->er_sum = s[0]; *Er = band_mean(c, s[0]);
    return 0;
}
// optimizeAlbedoAll with the regulariser (PsOptimizer.cpp:85-121): Eigen ConjugateGradient over the 3S unknowns on
// H = H_d + reg_rho Jr^T Jr applied matrix-free (albedo_reg.hip).  Host-driven, two read-backs per CG iteration: no shipped
// configuration enables this term.  The step is left in ar.x.
int albedo_reg_solve(psgsdf_ctx* c, const SweepArgs& a, int* iters_out, int* ok_out, double* err_out) {
    const AlbedoReg& ar = a.ar;
    // multi-rank: Jr^T reads the Jacobian rows, residuals and CG vectors of the rows across a cut -> halo exchanges (no-ops on one rank);
    // the sums go through read_parts, i.e. one all-reduce per read-back
    int rc = comm_halo(c, c->band.rho[0], 3, 1); if (rc) return rc;
    launch_areg_build(a, c->stream);
    if ((rc = comm_halo(c, ar.J, 15, 1))) return rc;          // J (12 planes) and res (3 planes) are adjacent
    launch_areg_system(a, c->stream);
    launch_areg_cg_init(a, c->stream);
    const int two[2] = {SC_AUX0, SC_AUX1}, one[1] = {SC_AUX0}; double s[2];
    if ((rc = read_parts(c, two, 2, s))) return rc;
    const float rhsN = (float)s[0];
    *iters_out = 0; *ok_out = 1; *err_out = 0;
    if (rhsN == 0.f) return 0;
    const float thr = fmaxf(FLT_EPSILON * FLT_EPSILON * rhsN, FLT_MIN);
    float res2 = rhsN, absNew = (float)s[1];
    const int maxIters = c->set.cg_max_it > 0 ? c->set.cg_max_it : (int)std::min<long long>(6 * c->S_global, INT_MAX);
    int i = 0;
    if (res2 >= thr && c->areg_device && !slab_mode(c)) {
        // ---- device-driven (round 6, single rank): chunks of iterations enqueued back to back, alpha / beta / the stop test derived on the device from the
        // kernels' own partial sums (albedo_reg.hip); ONE look at the five scalars per chunk instead of two read-backs per iteration.  The first chunk is
        // sized from the previous solve (the count is stable between Gauss-Newton iterations); a converged solve makes the rest of its chunk no-ops.
        const double init[5] = {0.0, 0.0, (double)rhsN, s[1], 0.0};
        HIPCHK(c, hipMemcpyAsync(ar.cgs, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));      // (`init` lives on this stack frame)
        const int nblk = band_blocks(c);
        int chunk = std::min(64, std::max(4, c->areg_last_iters + 2));
        double got[5] = {0, 0, 0, 0, 0};
        while (i < maxIters) {
            const int n = std::min(chunk, maxIters - i);
            for (int q = 0; q < n; ++q) launch_areg_cg_iteration(a, i + q, nblk, thr, c->stream);
            HIPCHK(c, hipMemcpyAsync(got, ar.cgs, sizeof(got), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            res2 = (float)got[2];
            if (got[0] != 0.0) { i = (int)got[1]; break; }
            i += n; chunk = 4;
        }
        if (!(res2 == res2)) return fail(c, PSGSDF_ERR_DEVICE, "the regularised albedo solve produced NaN");
        c->areg_last_iters = i;
    } else if (res2 >= thr) {
        while (i < maxIters) {
            if ((rc = comm_halo(c, ar.p, 3, 1))) return rc;
            launch_areg_jx(a, ar.p, ar.t, c->stream);
            if ((rc = comm_halo(c, ar.t, 3, 1))) return rc;
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
int step_begin(psgsdf_ctx* c, int block, int laplacian_reg, psgsdf_step_stats* st, std::function<void(double, double)> deferred_consumer, bool may_apply) {
    memset(st, 0, sizeof(*st));
    st->block = block;
    SweepArgs a = make_args(c, laplacian_reg);
    const bool led = c->set.model == PSGSDF_LED;
    double e_sum = 0, nobs = 0;
    int rc;
    switch (block) {
        case PSGSDF_ALBEDO: case PSGSDF_DIST: {
            if (block == PSGSDF_ALBEDO) {
                // without the albedo regulariser the system is diagonal: the sweep applies the update itself (may_apply = false: the caller
                // has a stop decision pending on this sweep's input energy and nothing may be modified yet)
                c->albedo_applied = may_apply && c->reg_r == 0.f && c->fuse_albedo;
                a.fuse_apply = c->albedo_applied ? (c->spec_undo ? 2 : 1) : 0;      // (2: speculative, the old albedo is kept for an undo)
                if (c->albedo_applied && c->spec_undo) c->spec_albedo_saved = true;
                take_fold(c, a, (1u << SC_ENERGY) | (1u << SC_NOBS) | (c->albedo_applied ? (1u << SC_ACCEPT) : 0u));
                timed(c, "sweep_albedo", [&] { launch_sweep_albedo(a, c->stream); });
            }
            else {
                materialize_fold(c);
                { int Gs, Rs; if (c->pcg_fuse_asm && c->fuse_pcg_init && band_blocks(c) <= kPcgMaxBlocks && cgf_solve_shape(c, &Gs, &Rs)) { a.pcg_gran = c->pcg_gran; a.pcg_gran_n = 2 * kSolveGranPlanes * kSolveMaxBlocksHost; a.pcg_asm = 1; } }   // the sweep clears the LOCAL tags of the persistent solve behind it (the cross-rank words carry the solve's epoch and are never cleared)
                timed(c, "sweep_dist", [&] { launch_sweep_dist(a, c->stream); });
            }   // (LED: the fused albedo sweep's sums are still pending and this sweep writes the same slots)
            const int slots[2] = {SC_ENERGY, SC_NOBS}; double s[2];
            if (deferred_consumer) return read_parts_deferred(c, slots, 2, [deferred_consumer](const double* v) { deferred_consumer(v[0], v[1]); });
            if ((rc = read_parts(c, slots, 2, s))) return rc;
            e_sum = s[0]; nobs = s[1];
            break;
        }
        case PSGSDF_LIGHT: case PSGSDF_POSE: {
            int col, launched = 0;
            // The per-frame solve in the sweep's own epilogue (sweeps.hip frame_rows_publish: the last workgroup of a frame solves it, the last frame
            // sums the energy columns into the mailbox): no k_solve_light / k_solve_pose launch.  Only where step_finish is certain to follow (the
            // deferred path: no stop decision hangs on this sweep's input energy), on one rank (a slab all-reduces the rows first), and not for the
            // LED light, which is ONE vector over all frames.
            // Multi-rank (round 4): the slabs' rows of a frame meet inside the sweep (engine.h XfTable: through the ranks' mailbox regions, where the
            // cross-rank persistent solve is set up), so the same epilogue serves -- no RCCL all-reduce of the rows, no solve launch.  A one-rank
            // communicator has nothing to exchange.  (An empty slab still has to deliver its -- zero -- rows: it keeps the all-reduce path, on every rank.)
            const bool xf = c->n_ranks > 1 && c->xf_enable && c->xr_ready && c->xf_table && !c->any_empty_slab;
            const bool fuse = deferred_consumer && c->fm_solve && c->frame_solve == 0 && (!slab_mode(c) || c->n_ranks == 1 || xf) && !c->profiling && (block == PSGSDF_POSE || !led || c->fm_solve_led) && c->row1 > c->row0 && c->band.obs_max > 0;
            double* fm_slot = nullptr; unsigned long long fm_key = 0;
            if (fuse) {      // (the mailbox slot first: reserving may flush, and a flush must not find the pending fold already handed to `a`)
                if ((rc = reserve_frame_energy_deferred(c, xf ? std::function<void(double, double)>([c, deferred_consumer](double e, double n) { if (std::isnan(e)) c->xf_timeout = true; deferred_consumer(e, n); }) : deferred_consumer, &fm_slot, &fm_key))) return rc;
                a.fm_solve = 1; a.fm_frames = c->frames; a.fm_led_light = c->led_light; a.fm_e_out = fm_slot; a.fm_e_key = fm_key;
                if (xf) { a.xf = c->xf_table; a.xf_epoch = ++c->xf_epoch; }
                a.fm_undo = (block == PSGSDF_LIGHT && c->spec_undo) ? (float*)c->frames_undo : nullptr;
            }
            if (block == PSGSDF_LIGHT) {
                take_fold(c, a, 0u);
                timed(c, "sweep_light", [&] { launched = launch_sweep_light(a, c->stream); });
                const int n = led ? 3 : (c->set.model == PSGSDF_SH2 ? 9 : 4), nh = led ? 3 : n * (n + 1) / 2;
                col = nh + n;
            } else { take_fold(c, a, 0u); timed(c, "sweep_pose", [&] { launched = launch_sweep_pose(a, c->stream); }); col = 27; }
            if (fuse) {
                if (launched) { c->fm_solved = true; return 0; }
                c->frame_e_slot = fm_slot; c->frame_e_key = fm_key;      // (nothing was launched: the solve kernel of step_finish fills the slot)
            }
            if (!launched) {      // no observations at all: empty rows, not the previous sweep's -- and the fold the sweep was to take is still owed
                HIPCHK(c, hipMemsetAsync(c->acc_frame, 0, sizeof(double) * c->acc_frame_n, c->stream));
                fold_by_kernel(c, a.fold);
            }
            if (fuse) return 0;
            // multi-rank: every slab has summed its own observations into the per-frame rows; after the all-reduce every rank holds the
            // global normal equations and solves all F (tiny) systems itself
            if ((rc = comm_allreduce(c, c->acc_frame, c->F * kFrameRow))) return rc;
            if (deferred_consumer) return reserve_frame_energy_deferred(c, deferred_consumer, &c->frame_e_slot, &c->frame_e_key);   // filled by the solve kernel
            if ((rc = read_frame_energy(c, col, &e_sum, &nobs))) return rc;
            break;
        }
        default: return fail(c, PSGSDF_ERR_ARG, "unknown block %d", block);
    }
    st->e_in = band_mean(c, e_sum);
    st->n_obs = (int64_t)nobs;
    return 0;
}
// the eigen frame solve's {iterations, error, Success, applied} (synchronous: psgsdf_step and the debug paths only -- the alternation loop never waits for them)
int read_frame_solver_stats(psgsdf_ctx* c, int kind, psgsdf_step_stats* st) {
    HIPCHK(c, hipMemcpyAsync(c->fs_last[kind], c->fs_stats + 4 * kind, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (st) { st->cg_iters = (int)c->fs_last[kind][0]; st->cg_error = c->fs_last[kind][1]; st->cg_converged = (int)c->fs_last[kind][2]; st->applied = (int)c->fs_last[kind][3]; }
    return 0;
}
int step_finish(psgsdf_ctx* c, int block, int laplacian_reg, psgsdf_step_stats* st, bool defer_reg_sums) {
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
            if (!c->albedo_applied) {
                take_fold(c, a, 1u << SC_ACCEPT);
                timed(c, "apply_albedo", [&] { launch_apply_albedo(a, c->stream); });
            }
            c->albedo_applied = false;
            if (c->want_counts) { if ((rc = read_parts(c, slots, 1, s))) return rc; st->n_accepted = (int64_t)s[0]; }
            st->cg_iters = 1; st->cg_converged = 1; st->applied = 1;
            break;
        }
        case PSGSDF_LIGHT:
            // (a speculative light update keeps the coefficients it overwrites: the solve kernel copies them to frames_undo first)
            if (c->fm_solved) c->fm_solved = false;      // (the sweep's last workgroups solved their frames: step_begin)
            else if (c->frame_solve == 1) {             // the reference's global float Jacobi-PCG over all frames' blocks (frame_solve.hip)
                if (!frames_eigen_fits(c->set.model, c->F)) return fail(c, PSGSDF_ERR_UNSUPPORTED, "PSGSDF_FRAME_SOLVE=eigen holds at most 2048 unknowns in its one workgroup (%d keyframes)", c->F);
                timed(c, "solve_light_eigen", [&] { launch_frames_eigen_light(a, c->frames, c->led_light, c->frame_e_slot, c->frame_e_key, c->spec_undo ? (float*)c->frames_undo : nullptr, c->fs_stats, c->stream); });
            }
            else timed(c, "solve_light", [&] { launch_solve_light(a, c->frames, c->led_light, c->frame_e_slot, c->frame_e_key, c->spec_undo ? (float*)c->frames_undo : nullptr, c->stream); });
            if (c->spec_undo) c->spec_light_saved = true;
            c->frame_e_slot = nullptr; c->frame_e_key = 0;
            st->cg_converged = 1; st->applied = 1; st->n_accepted = led ? 1 : c->F;
            if (c->frame_solve == 1 && c->want_counts) { if ((rc = read_frame_solver_stats(c, 0, st))) return rc; }
            break;
        case PSGSDF_POSE:
            if (c->fm_solved) c->fm_solved = false;
            else if (c->frame_solve == 1) {
                if (!frames_eigen_fits(c->set.model, c->F)) return fail(c, PSGSDF_ERR_UNSUPPORTED, "PSGSDF_FRAME_SOLVE=eigen holds at most 2048 unknowns in its one workgroup (%d keyframes)", c->F);
                timed(c, "solve_pose_eigen", [&] { launch_frames_eigen_pose(a, c->frames, c->frame_e_slot, c->frame_e_key, c->fs_stats + 4, c->stream); });
            }
            else timed(c, "solve_pose", [&] { launch_solve_pose(a, c->frames, c->frame_e_slot, c->frame_e_key, c->stream); });
            c->frame_e_slot = nullptr; c->frame_e_key = 0;
            st->cg_converged = 1; st->applied = 1; st->n_accepted = c->F;
            if (c->frame_solve == 1 && c->want_counts) { if ((rc = read_frame_solver_stats(c, 1, st))) return rc; st->n_accepted = st->applied ? c->F : 0; }
            break;
        case PSGSDF_DIST: {
            take_fold(c, a, 0u);
            if (c->fuse_pcg_init && band_blocks(c) <= kPcgMaxBlocks) { a.pcg_fuse_init = 1; a.pcg_init_blocks = band_blocks(c); }   // the assembly kernel initialises the PCG
            bool apply_in_solve = false;      // (the accepted count goes to the first G entries of a partial slot that holds one entry per 256 rows)
            { int Gs, Rs; if (a.pcg_fuse_init && cgf_solve_shape(c, &Gs, &Rs)) { a.pcg_gran = c->pcg_gran; a.pcg_gran_n = 2 * kSolveGranPlanes * kSolveMaxBlocksHost; a.pcg_asm = c->pcg_fuse_asm ? 1 : 0; apply_in_solve = a.pcg_asm && c->pcg_fuse_apply && Gs <= band_blocks(c); } }   // the assembly kernel also clears the persistent solve's tags
            if ((rc = comm_halo(c, c->band.blk, 14, 1))) return rc;   // multi-rank: rows of H next to a cut take contributions from the neighbour slab's voxel blocks
            if (!a.pcg_asm) { timed(c, "assemble", [&] { launch_assemble(a, c->stream); }); a.fold.n = 0; }      // (pcg_asm: the persistent solve assembles its rows itself; the distance sweep cleared its tags)
            int iters = 0, ok = 1; double err = 0;
            const bool only_on_success = !led && c->set.ref_quirks;   // PsOptimizer.cpp:168-170 (B8): SH skips the update unless the solve reports Success
            if (apply_in_solve) a.pcg_apply = only_on_success ? 2 : 1;      // persistent solve: the update is its epilogue
            bool tail_ran = false; int tail_rc = 0;
            auto tail = [&](const double* gate) {                     // distance update + regrad, gated on the device-side outcome of the solve
                SweepArgs ag = a; ag.fold.n = 0; ag.gate = gate;
                if (!a.pcg_apply) timed(c, "apply_dist", [&] { launch_apply_dist(ag, c->stream); });      // (pcg_apply: the persistent solve has done it)
                if (int hrc = comm_halo(c, c->band.dist, 1, 1)) tail_rc = hrc;   // multi-rank: the regrad's stencils read the neighbour slab's new distances (harmless when the gate is closed: nothing changed)
                SweepArgs a2 = make_args(c, 0); a2.gate = gate;
                timed(c, "derive", [&] { launch_derive(a2, 1, c->stream); });
            };
            if ((rc = pcg_solve(c, a, &iters, &ok, &err, tail, only_on_success, &tail_ran))) return rc;
            if (tail_rc) return tail_rc;
            int apply = 1;
            if (only_on_success && !ok) apply = 0;
            if (apply) {
                if (!tail_ran) { tail(nullptr); if (tail_rc) return tail_rc; }
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
        if (full && c->observer && c->observer(c->observer_user, it + 1, &r)) stop = true;      // (passive: the state may be ahead of the record)
        if (!stop && full && on_iter && ((it + 1) % c->on_iter_period == 0 || r.upsampled) && on_iter(user, it + 1, &r)) stop = true;      // (the exact-state callback, when it is due: psgsdf_set_on_iter_period)
        return 0;
    };
    // per-iteration values that arrive through deferred read-backs (stable addresses: two alternating slots)
    struct Late { double e_in[4]; int blk_of[4]; int n; bool dist_ran; int cg_iters; bool alb_reg; float e_r; } late[2];
    int li = 0;
    auto apply_late = [&](psgsdf_iter_stats& r, const Late& lt, int first_slot_pending) {
        r.e_n_in = L.E_n; r.e_l_in = L.E_l;      // (records are closed in order: L still holds what was in force before this iteration's distance block)
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
    // ---- speculative start of an iteration (psgsdf_optimize without a per-iteration callback).  The stop decision on
    // iteration i needs the PS energy of its final state, which the FIRST sweep of iteration i+1 computes; waiting for it there leaves the GPU idle
    // for the round trip (read-back -> host decision -> next launch) and forces the albedo update into a kernel of its own.  Instead the blocks
    // whose updates can be undone cheaply -- albedo (the old values go to a spare plane) and light (F small records) -- are enqueued right away,
    // updates included, and the decision is taken when the energy arrives, at the latest before the distance block.  If the loop ends there
    // (once per optimisation) the saved albedo / light are put back: the state left behind is exactly the reference's.  A per-iteration
    // callback (on_iter) must see the state of the iteration it is told about, so with a callback every iteration is closed first, as before.
    // Multi-rank (round 4): the same window, with two rules that keep the ranks' collectives in step -- the closing energy is a sum over the slabs,
    // so the staged read-backs are committed (mg_commit: ONE all-reduce on the stream) right behind the first speculative sweep instead of at the next
    // host wait, and the window is only ever closed at the same program point on every rank (the blocking poll in front of the first block that
    // cannot be undone): no "has it landed yet?" polls, whose answer could differ from rank to rank and with it the order of the collectives.
    const bool mr_spec = slab_mode(c);
    const bool spec_base = full && c->speculate && (!mr_spec || c->speculate_mr) && c->reg_r == 0.f && !c->profiling;
    auto can_spec_at = [&](int it_closing) { return spec_base && (!on_iter || (it_closing + 1) % c->on_iter_period != 0); };      // (a due on_iter must see the state of the iteration it reports)
    c->spec_undo = false; c->spec_albedo_saved = false; c->spec_light_saved = false;
    const double* close_src = nullptr;   // mailbox address of the closing energy's read-back while a speculation window is open
    bool spec_open = false, mr_committed = false;
    auto end_window = [&] { spec_open = false; close_src = nullptr; c->spec_undo = false; c->spec_albedo_saved = false; c->spec_light_saved = false; };
    auto undo_window = [&]() -> int {   // the loop ends on iteration iter-1: take back what iteration `iter` has applied speculatively
        if (c->spec_light_saved) launch_restore_light(c->F, c->frames, c->led_light, (const float*)c->frames_undo, c->stream);
        if (c->spec_albedo_saved) { SweepArgs au = make_args(c, 0); launch_restore_albedo(au, c->stream); }
        c->spec_undos++;
        end_window();
        return flush(c);                 // (read-backs of the abandoned iteration land in its scratch record)
    };
    // the closing energy of `prev` has been delivered: close the record, take the stop decision, undo if the loop ends
    auto lazy_close = [&]() -> int {
        apply_late(prev, *prev_late, -1);
        close_iteration(c, L, &prev, prev_slot, (float)band_mean(c, *prev_close), full);
        have_prev = false; prev_close = nullptr;
        int rc = finalize(prev, iter - 1); if (rc) return rc;
        if (spec_open) { if (stop) return undo_window(); end_window(); }
        return 0;
    };
    // bring the closing energy of an open window in: non-blocking (only if it has landed) or blocking
    auto window_poll = [&](bool block) -> int {
        if (!spec_open || !have_prev || !prev_close) return 0;
        if (mr_spec) { if (!block) return 0; materialize_fold(c); int rc = mg_commit(c); if (rc) return rc; }
        if (std::isnan(*prev_close)) {
            size_t idx = c->deferred.size();
            for (size_t e = 0; e < c->deferred.size(); ++e) if (c->deferred[e].src == close_src) { idx = e; break; }
            if (idx < c->deferred.size()) {
                if (c->pending_fold.n && c->pending_fold.out == c->mbox_dev + (close_src - c->mbox)) { if (!block) return 0; materialize_fold(c); }   // (its fold is still waiting for a kernel to take it)
                if (!block && !readback_landed(c->deferred[idx])) return 0;
                int rc = deliver_first(c, idx + 1, false); if (rc) return rc;
            }
        }
        if (std::isnan(*prev_close)) return block ? fail(c, PSGSDF_ERR_DEVICE, "the closing energy of iteration %d never arrived", iter - 1) : 0;
        return lazy_close();
    };
    while (iter < max_iters && !stop) {
        memset(&rec, 0, sizeof(rec));
        for (int q = 0; q < 4; ++q) rec.e_after[q] = NAN;
        Late& lt = late[li]; lt.n = 0; lt.dist_ran = false; lt.cg_iters = 0; lt.alb_reg = false; lt.e_r = 0.f;
        int pending = -1;
        for (int q = 0; q < 4 && !stop; ++q) {
            const int blk = order[q];
            if (!(flags & blk)) continue;
            psgsdf_step_stats st;
            const bool undoable = (blk == PSGSDF_ALBEDO && !c->spec_albedo_saved && c->fuse_albedo) || (blk == PSGSDF_LIGHT && !c->spec_light_saved);
            if (spec_open && !undoable) { int rc = window_poll(true); if (rc) return rc; if (stop) break; }   // the decision is due now (it arrived long ago: no bubble)
            const int qi = lt.n;
            lt.blk_of[qi] = blk; lt.e_in[qi] = NAN; lt.n++;
            if (have_prev && full && !spec_open && !(can_spec_at(iter - 1) && undoable)) {   // synchronous: this sweep's input energy closes the previous iteration (stop decision)
                int rc = step_begin(c, blk, L.laplacian_reg, &st, nullptr, false); if (rc) return rc;   // (flushes every deferred read of the previous iteration)
                lt.e_in[qi] = st.e_in;
                apply_late(prev, *prev_late, -1);
                close_iteration(c, L, &prev, prev_slot, (float)st.e_in, full);
                have_prev = false;
                if ((rc = finalize(prev, iter - 1))) return rc;
                if (stop) {        // converged / diverged / aborted: nothing of this iteration has been applied
                    // (a frame-major sweep's rows stay unconsumed: nothing to clean up, the next sweep overwrites them)
                    break;
                }
            } else {
                // no stop decision pending (psgsdf_iterate never exits early) or a speculation window: even the closing energy of the previous
                // iteration is delivered lazily, at the next host sync (the PCG status read of this iteration) or when the window polls for it
                if (have_prev && full && !spec_open) { spec_open = true; mr_committed = false; c->spec_undo = true; c->spec_albedo_saved = c->spec_light_saved = false; c->spec_windows++; }
                double* slot_e = &lt.e_in[qi];
                int rc = step_begin(c, blk, L.laplacian_reg, &st, [slot_e](double e_sum, double) { *slot_e = e_sum; }); if (rc) return rc;
                if (have_prev && prev_close == nullptr) { prev_close = slot_e; if (spec_open && !c->deferred.empty()) close_src = c->deferred.back().src; }
            }
            int rc = step_finish(c, blk, L.laplacian_reg, &st, true); if (rc) return rc;
            if (blk == PSGSDF_ALBEDO && c->reg_r != 0.f) { double er; if ((rc = albedo_reg_energy(c, &er))) return rc; lt.alb_reg = true; lt.e_r = (float)er; }   // PsOptimizer.cpp:312 (enters L when the record closes)
            if (blk == PSGSDF_DIST) { lt.dist_ran = true; lt.cg_iters = st.cg_iters; }
            if (spec_open && mr_spec && !mr_committed && prev_close && std::isnan(*prev_close)) { mr_committed = true; materialize_fold(c); if ((rc = mg_commit(c))) return rc; }   // the closing energy starts travelling now (once per window)
            if (spec_open) { if ((rc = window_poll(false))) return rc; }
            else if (have_prev && prev_close && !std::isnan(*prev_close)) { if ((rc = lazy_close())) return rc; }   // the lazy closing energy has arrived
            pending = blk == PSGSDF_ALBEDO ? 0 : blk == PSGSDF_LIGHT ? 1 : blk == PSGSDF_DIST ? 2 : 3;
        }
        if (!stop && spec_open) { int rc = window_poll(true); if (rc) return rc; }      // (an iteration of undoable blocks only)
        if (stop) break;
        if (have_prev && prev_close) {   // still open (no host sync happened during this iteration): force one
            int rc = flush(c); if (rc) return rc;
            if ((rc = lazy_close())) return rc;
            if (stop) break;
        }
        // deferred e_in values are raw sums (not yet divided by S) except the synchronous first one: normalise on use
        const bool last = iter + 1 >= max_iters;
        const bool refine_next = full && c->set.upsample && iter == 5;
        // the Laplacian schedule (PsOptimizer.cpp:411-413 / LedOptimizer.cpp:461-463) zeroes reg_l in finalize(): the next iteration's first
        // sweep must not be enqueued with the old weight, so this iteration is closed synchronously as well
        const bool sched_next = full && c->set.upsample && c->reg_l != 0.0f && (led ? iter == 15 : iter == 16);
        if (pending >= 0 && !last && !refine_next && !sched_next) { prev = rec; prev_slot = pending; have_prev = true; prev_late = &lt; li ^= 1; }
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
    // bring the dense grid up to date, refine, rebuild the band.  Multi-rank: every slab refines the planes it holds; the halo planes need
    // their owners' albedo and gradient first (the distances are exchanged every iteration anyway), and of the refined halo planes only the
    // inner one is kept: the slab [z0, z1) becomes [2 z0, 2 z1) with one halo plane on each inner side again.
    int rc;
    if ((rc = comm_halo(c, c->band.rho[0], 3, 1))) return rc;
    if ((rc = comm_halo(c, c->band.g[0], 3, 1))) return rc;
    timed(c, "band_scatter", [&] { launch_band_scatter(c->dense, c->band, c->stream); });
    DenseView nd{};
    const long long nn = 8 * c->grid.nvox;
    if ((rc = alloc_dense(c, nd, nn, c->dense.KW, true))) return rc;
    timed(c, "upsample", [&] { launch_upsample(c->dense, nd, c->grid, c->stream); });
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_dense(c);
    if (c->vis_seq) { hipFree(c->vis_seq); c->vis_seq = nullptr; }
    const int old_zlo = c->zlo;
    for (int a = 0; a < 3; ++a) c->gdim[a] *= 2;
    c->gnvox *= 8;
    GridP& g = c->grid;
    g.vs *= 0.5f;
    for (int a = 0; a < 3; ++a) g.origin[a] = c->shift[a] - (float)(0.5 * (double)g.vs) * (float)c->gdim[a] - (float)(0.5 * (double)g.vs) * 1.0f;   // VoxelGrid.h:143-149
    g.vs_inv = (float)(1.0 / (double)g.vs);
    if ((rc = set_local_grid(c, 2 * c->z0, 2 * c->z1))) return rc;
    if (g.nvox == nn) c->dense = nd;                                  // one rank, or a slab without halo planes: the refined grid is the local grid
    else {                                                            // drop the outer child plane of each refined halo plane
        const size_t plane = (size_t)g.dim[0] * g.dim[1], skip = (size_t)(c->zlo - 2 * old_zlo) * plane, n = (size_t)g.nvox;
        DenseView td{};
        if ((rc = alloc_dense(c, td, g.nvox, nd.KW, true))) return rc;
        HIPCHK(c, hipMemcpyAsync(td.dist, nd.dist + skip, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(td.weight, nd.weight + skip, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
        for (int a = 0; a < 3; ++a) {
            HIPCHK(c, hipMemcpyAsync(td.g[a], nd.g[a] + skip, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(td.rho[a], nd.rho[a] + skip, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
        }
        HIPCHK(c, hipMemcpyAsync(td.vis, nd.vis + skip * nd.KW, sizeof(uint64_t) * n * nd.KW, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(nd.dist); hipFree(nd.weight); hipFree(nd.vis); hipFree(nd.row_of);
        for (int a = 0; a < 3; ++a) { hipFree(nd.g[a]); hipFree(nd.rho[a]); }
        c->dense = td;
    }
    if ((rc = build_band(c))) return rc;
    return derive(c, 0);
}

}  // namespace psge
