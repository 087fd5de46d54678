// api_debug.hip -- measurement and test hooks (not part of the reference seam).
#include "engine_internal.h"

using namespace psge;

extern "C" {
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

// the persistent solve run for exactly `passes` passes on the current distance system (stop rule off): total kernel time and its shape
int psgsdf_debug_time_pcg_solve(psgsdf_ctx* c, int passes, int reps, double* ms_per_launch, int32_t* shape, double* stamps /*[16] or NULL*/) {
    if (!c || !c->inited || !ms_per_launch || !shape || (passes < 1 && passes != -7) || reps < 1) return fail(c, PSGSDF_ERR_STATE, "init first");      // passes = -7: fault injection, one workgroup stops publishing (the others' bounded waits must end the kernel)
    HIPCHK(c, hipSetDevice(c->device));
    int G = 0, rows = 0;
    if (!cgf_solve_shape(c, &G, &rows)) return fail(c, PSGSDF_ERR_UNSUPPORTED, "the persistent solve does not apply to this context");
    shape[0] = G; shape[1] = rows;      // rows = rows per workgroup
    SweepArgs a = make_args(c, c->reg_l != 0.f);
    a.pcg_fuse_init = 1; a.pcg_init_blocks = band_blocks(c); a.pcg_gran = c->pcg_gran; a.pcg_gran_n = 2 * kSolveGranPlanes * kSolveMaxBlocksHost; a.pcg_asm = c->pcg_fuse_asm ? 1 : 0;
    hipEvent_t e0, e1; HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    float total = 0;
    for (int r = 0; r < reps + 1; ++r) {
        launch_sweep_dist(a, c->stream);
        launch_assemble(a, c->stream);
        HIPCHK(c, hipEventRecord(e0, c->stream));
        a.pcg_epoch = ++c->pcg_solve_serial;
        launch_cgf_solve(a, c->pcg_sc, c->pcg_gran, G, rows, 1 << 30, c->mbox_dev, 0ull, passes, c->stream);
        HIPCHK(c, hipEventRecord(e1, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) total += ms;
        if (c->mbox[3] != 1.0) { hipEventDestroy(e0); hipEventDestroy(e1); return fail(c, PSGSDF_ERR_DEVICE, "persistent solve status %g: a workgroup gave up waiting for the others", c->mbox[3]); }
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    *ms_per_launch = (double)total / reps;
    if (stamps) for (int i = 0; i < 16; ++i) stamps[i] = c->mbox[8 + i];      // pass 8 of the last launch: 7 stage stamps of the first and of the last workgroup (100 MHz ticks)
    if (const char* dump = getenv("PSGSDF_SOLVE_DUMP")) {   // per-workgroup publish / gather-done / sums-seen times (tools/pcg_solve_time.py)
        std::vector<double> h(8 * 256);      // columns: workgroup, publish(8), sums seen(9), =, XCC, gathers done(9), publish(9), neighbour tags seen(9), prefetch valid(9)
        HIPCHK(c, hipMemcpy(h.data(), c->pcg_sc + 16, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(dump, "w")) { for (int i = 0; i < G; ++i) fprintf(f, "%d %.0f %.0f %.0f %.0f %.0f %.0f %.0f %.0f\n", i, h[i], h[256 + i], h[512 + i], h[768 + i], h[1024 + i], h[1280 + i], h[1536 + i], h[1792 + i]); fclose(f); }
    }
    return PSGSDF_OK;
}

// VERDICT r04 item 4b, measured instead of argued: how much of the distance solve could an early start hide under the distance sweep?  Timing only.
//   out[0] the distance sweep alone, out[1] the solve alone (its own assembly prologue, natural stop, no update applied), out[2] the two back to back on
//   one stream (what the loop does), out[3] the solve started on a SECOND stream at the same moment as a repeat of the sweep -- it assembles from the
//   blocks of the identical sweep before, so it never waits for the one running beside it: every workgroup of the solve that finds a free CU is resident
//   from the first microsecond.  out[2] - out[3] is the most any overlap scheme (per-block "done" flags included) could win.  Milliseconds, means over reps.
int psgsdf_debug_overlap_probe(psgsdf_ctx* c, int reps, double* out) {
    if (!c || !c->inited || !out || reps < 1) return fail(c, PSGSDF_ERR_STATE, "init first");
    if (c->n_ranks > 1) return fail(c, PSGSDF_ERR_UNSUPPORTED, "overlap probe: one rank");
    HIPCHK(c, hipSetDevice(c->device));
    int G = 0, rows = 0;
    if (!cgf_solve_shape(c, &G, &rows) || !c->pcg_fuse_asm) return fail(c, PSGSDF_ERR_UNSUPPORTED, "the persistent solve with its fused assembly does not apply to this context");
    { int rc = flush(c); if (rc) return rc; }
    SweepArgs a = make_args(c, c->reg_l != 0.f);
    a.pcg_fuse_init = 1; a.pcg_init_blocks = band_blocks(c); a.pcg_gran = c->pcg_gran; a.pcg_gran_n = 2 * kSolveGranPlanes * kSolveMaxBlocksHost; a.pcg_asm = 1;
    SweepArgs quiet = a; quiet.pcg_gran = nullptr; quiet.pcg_gran_n = 0;      // (a sweep that does NOT clear the solve's tags: the one that runs beside a solve)
    hipStream_t s2; HIPCHK(c, hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e[4]; for (auto& x : e) HIPCHK(c, hipEventCreate(&x));
    double acc[4] = {0, 0, 0, 0};
    int rc = PSGSDF_OK;
    auto ms = [&](hipEvent_t x, hipEvent_t y) { float t = 0; hipEventElapsedTime(&t, x, y); return (double)t; };
    auto solve = [&](hipStream_t s) { a.pcg_epoch = ++c->pcg_solve_serial; launch_cgf_solve(a, c->pcg_sc, c->pcg_gran, G, rows, 1 << 30, c->mbox_dev, 0ull, 0, s); };
    for (int r = 0; r < reps + 1 && !rc; ++r) {
        // [0] + [1]: each alone
        hipEventRecord(e[0], c->stream); launch_sweep_dist(a, c->stream); hipEventRecord(e[1], c->stream);
        hipStreamSynchronize(c->stream);
        hipEventRecord(e[2], c->stream); solve(c->stream); hipEventRecord(e[3], c->stream);
        hipStreamSynchronize(c->stream);
        if (c->mbox[3] != 1.0) { rc = fail(c, PSGSDF_ERR_DEVICE, "overlap probe: solve status %g", c->mbox[3]); break; }
        if (r) { acc[0] += ms(e[0], e[1]); acc[1] += ms(e[2], e[3]); }
        // [2]: back to back
        hipEventRecord(e[0], c->stream); launch_sweep_dist(a, c->stream); solve(c->stream); hipEventRecord(e[1], c->stream);
        hipStreamSynchronize(c->stream);
        if (r) acc[2] += ms(e[0], e[1]);
        // [3]: side by side (the tags were cleared by the sweep of [2]; its solve consumed them, so clear them again first)
        launch_sweep_dist(a, c->stream);
        hipEventRecord(e[0], c->stream); hipStreamWaitEvent(s2, e[0], 0);
        launch_sweep_dist(quiet, c->stream); hipEventRecord(e[1], c->stream);
        solve(s2); hipEventRecord(e[2], s2);
        hipStreamSynchronize(c->stream); hipStreamSynchronize(s2);
        if (c->mbox[3] != 1.0) { rc = fail(c, PSGSDF_ERR_DEVICE, "overlap probe (side by side): solve status %g", c->mbox[3]); break; }
        if (r) acc[3] += std::max(ms(e[0], e[1]), ms(e[0], e[2]));
    }
    for (auto& x : e) hipEventDestroy(x);
    hipStreamDestroy(s2);
    for (int i = 0; i < 4; ++i) out[i] = acc[i] / reps;
    return rc;
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

int psgsdf_set_frame_solver(psgsdf_ctx* c, int mode) {
    if (!c) return PSGSDF_ERR_ARG;
    if (mode != 0 && mode != 1) return fail(c, PSGSDF_ERR_ARG, "frame solver mode %d (0 = LDL^T per frame, 1 = the reference's global float Jacobi-PCG)", mode);
    c->frame_solve = mode;
    return PSGSDF_OK;
}
int psgsdf_get_frame_solver_stats(psgsdf_ctx* c, int block, int32_t* iterations, double* error, int32_t* converged, int32_t* applied) {
    if (!c) return PSGSDF_ERR_ARG;
    if (block != PSGSDF_LIGHT && block != PSGSDF_POSE) return fail(c, PSGSDF_ERR_ARG, "block must be LIGHT or POSE");
    HIPCHK(c, hipSetDevice(c->device));
    const int kind = block == PSGSDF_POSE ? 1 : 0;
    psgsdf_step_stats st; memset(&st, 0, sizeof(st));
    if (int rc = read_frame_solver_stats(c, kind, &st)) return rc;
    if (iterations) *iterations = st.cg_iters;
    if (error) *error = st.cg_error;
    if (converged) *converged = st.cg_converged;
    if (applied) *applied = st.applied;
    return PSGSDF_OK;
}
int psgsdf_debug_frame_cg(psgsdf_ctx* c, int nb, int n, const float* H, const float* b, float* x, int max_it, int32_t* iterations, double* error, int32_t* converged) {
    if (!c || !H || !b || !x || nb <= 0) return fail(c, PSGSDF_ERR_ARG, "null argument");
    if ((n != 3 && n != 4 && n != 6 && n != 9) || (long long)nb * n > 2048) return fail(c, PSGSDF_ERR_UNSUPPORTED, "block size %d x %d blocks", n, nb);
    HIPCHK(c, hipSetDevice(c->device));
    float *dH = nullptr, *db = nullptr, *dx = nullptr; double* ds = nullptr;
    const size_t nH = (size_t)nb * n * n, nv = (size_t)nb * n;
    int rc = PSGSDF_OK;
    if (hipMalloc(&dH, sizeof(float) * nH) != hipSuccess || hipMalloc(&db, sizeof(float) * nv) != hipSuccess || hipMalloc(&dx, sizeof(float) * nv) != hipSuccess || hipMalloc(&ds, sizeof(double) * 4) != hipSuccess)
        rc = fail(c, PSGSDF_ERR_DEVICE, "out of device memory");
    double st[4] = {0, 0, 0, 0};
    if (!rc && (hipMemcpy(dH, H, sizeof(float) * nH, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(db, b, sizeof(float) * nv, hipMemcpyHostToDevice) != hipSuccess)) rc = fail(c, PSGSDF_ERR_DEVICE, "upload failed");
    if (!rc && launch_frames_eigen_raw(nb, n, dH, db, dx, ds, max_it, c->stream)) rc = fail(c, PSGSDF_ERR_UNSUPPORTED, "shape");
    if (!rc && (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(x, dx, sizeof(float) * nv, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(st, ds, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess))
        rc = fail(c, PSGSDF_ERR_DEVICE, "%s", hipGetErrorString(hipGetLastError()));
    hipFree(dH); hipFree(db); hipFree(dx); hipFree(ds);
    if (rc) return rc;
    if (iterations) *iterations = (int32_t)st[0];
    if (error) *error = st[1];
    if (converged) *converged = (int32_t)st[2];
    return PSGSDF_OK;
}
int psgsdf_debug_frame_system(psgsdf_ctx* c, int block, double* H, double* b) {
    if (!c || !c->inited || !H || !b) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    SweepArgs a = make_args(c, 0);
    const bool led = c->set.model == PSGSDF_LED;
    int n, nb, nh;
    int launched = 0;
    if (block == PSGSDF_LIGHT) { launched = launch_sweep_light(a, c->stream); n = led ? 3 : (c->set.model == PSGSDF_SH2 ? 9 : 4); nb = led ? 1 : c->F; nh = led ? 3 : n * (n + 1) / 2; }
    else if (block == PSGSDF_POSE) { launched = launch_sweep_pose(a, c->stream); n = 6; nb = c->F; nh = 21; }
    else return fail(c, PSGSDF_ERR_ARG, "block must be LIGHT or POSE");
    if (!launched) HIPCHK(c, hipMemsetAsync(c->acc_frame, 0, sizeof(double) * c->acc_frame_n, c->stream));
    std::vector<double> acc(c->acc_frame_n);
    HIPCHK(c, hipMemcpyAsync(acc.data(), c->acc_frame, sizeof(double) * c->acc_frame_n, hipMemcpyDeviceToHost, c->stream));
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

int psgsdf_debug_sync_stats(psgsdf_ctx* c, int64_t out[8]) {
    if (!c || !out) return PSGSDF_ERR_ARG;
    out[0] = c->mbox_checked; out[1] = c->mbox_late; out[2] = c->persist_fallbacks + (c->img_compacted ? 1000000LL : 0); out[3] = c->spec_windows * 1000000LL + c->spec_undos;
    out[4] = (c->xr_ready ? 1 : 0) + 10 * c->n_halo_pushes; out[5] = c->xr_solves; out[6] = c->xr_mem_kind; out[7] = c->xr_probe_stale * 1000000LL + c->xr_probe_timeouts;
    return PSGSDF_OK;
}

// Which tuning knobs are in force on this context: one JSON object -- "build", "env" (every supported PSGSDF_* variable that was set when the context
// was created, verbatim), "ignored_dev_only" (fault-injection / ablation variables that were set but are compiled out of this build) and "effective"
// (what the switches resolved to, defaults included).  Returns the length the text needs (excluding the terminator); writes at most cap - 1 characters.
int psgsdf_get_tuning(psgsdf_ctx* c, char* json, size_t cap) {
    if (!c) return PSGSDF_ERR_ARG;
    std::string o = "{\"build\": \"";
    o += psgsdf_version(); o += "\", \"env\": {";
    auto esc = [](const std::string& v) { std::string r; for (char ch : v) { if (ch == '"' || ch == '\\') r += '\\'; if ((unsigned char)ch >= 0x20) r += ch; } return r; };
    bool first = true;
    for (auto& kv : c->tuning_env) { o += (first ? "\"" : ", \"") + kv.first + "\": \"" + esc(kv.second) + "\""; first = false; }
    o += "}, \"ignored_dev_only\": {"; first = true;
    for (auto& kv : c->tuning_ignored) { o += (first ? "\"" : ", \"") + kv.first + "\": \"" + esc(kv.second) + "\""; first = false; }
    o += "}, \"effective\": {";
    char buf[1024];
    snprintf(buf, sizeof(buf), "\"pcg_poll\": %d, \"speculate\": %d, \"speculate_mr\": %d, \"fold_in_next\": %d, \"fuse_albedo\": %d, \"fuse_pcg_init\": %d, \"pcg_persist\": %d, \"pcg_pipeline\": %d, \"pcg_tagm\": %d, "
             "\"pcg_prefetch\": %d, \"pcg_fuse_asm\": %d, \"pcg_fuse_apply\": %d, \"pcg_xcd_local\": %d, \"fm_solve\": %d, \"fm_solve_led\": %d, \"frame_solve\": \"%s\", \"img_compact\": %d, \"xcd_map\": %d, "
             "\"xr\": %d, \"xf\": %d, \"xs\": %d, \"xh\": %d, \"xr_mem_kind\": %d, \"xwait_log2\": %d, \"cu_mask\": [%d, %d], \"mbox_check\": %d, \"pcg_ablate\": %d, \"fault_solve\": %d, \"fault_halo\": %lld",
             (int)c->pcg_poll, (int)c->speculate, (int)c->speculate_mr, (int)c->fold_in_next, (int)c->fuse_albedo, (int)c->fuse_pcg_init, (int)c->pcg_persist, (int)c->pcg_pipeline, (c->pcg_pipeline && c->pcg_tagm) ? (c->n_ranks > 1 ? (c->pcg_tagm_mr ? 2 : 0) : 1) : 0,
             (int)c->pcg_prefetch, (int)c->pcg_fuse_asm, (int)c->pcg_fuse_apply, (int)c->pcg_xcd_local, (int)c->fm_solve, (int)c->fm_solve_led, c->frame_solve == 1 ? "eigen" : "ldlt", (int)c->img_compact, c->xcd_map,
             (int)c->xr_enable, (int)c->xf_enable, (int)c->xs_enable, (int)c->xh_enable, c->xr_mem_kind, (int)lround(log2((double)c->xwait_spins)), c->cu_mask_lo, c->cu_mask_hi,
             (int)c->mbox_check, c->pcg_ablate, c->fault_solve, c->fault_halo);
    o += buf; o += "}";
    if (c->n_ranks > 1) {      // the hand-off probe as THIS rank saw it, per memory kind tried (comm.hip xr_probe): a first multi-GPU run reads its pairs here
        snprintf(buf, sizeof(buf), ", \"xr_probe\": {\"rank\": %d, \"n_ranks\": %d, \"kind_chosen\": %d, \"stale_mappings\": %lld, \"fine_grained\": {\"tried\": %lld, \"stale_records_from_lower\": %lld, \"expired_waits_lower\": %lld, \"expired_waits_upper\": %lld}, "
                 "\"uncached\": {\"tried\": %lld, \"stale_records_from_lower\": %lld, \"expired_waits_lower\": %lld, \"expired_waits_upper\": %lld}}",
                 c->rank, c->n_ranks, c->xr_mem_kind, c->xr_stale_maps, c->xr_probe_local[1][3], c->xr_probe_local[1][0], c->xr_probe_local[1][1], c->xr_probe_local[1][2],
                 c->xr_probe_local[2][3], c->xr_probe_local[2][0], c->xr_probe_local[2][1], c->xr_probe_local[2][2]);
        o += buf;
    }
    o += "}";
    if (json && cap) { const size_t n = std::min(cap - 1, o.size()); memcpy(json, o.data(), n); json[n] = 0; }
    return (int)o.size();
}

}  // extern "C"
