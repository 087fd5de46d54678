// api_multi_gpu.hip -- C ABI of the z-slab multi-GPU path: one context per rank, phases + exchange buffers.
#include "engine_internal.h"

using namespace psge;

extern "C" {
// ---- multi-rank (z-slab) phase API ---------------------------------------------------------
// One process per GPU.  Every rank holds the whole band but owns rows [row0,row1) (equal band count = z-slabs);
// the host program (psgradientsdf_amd/distributed.py) runs the phases below and performs the exchanges between them
// with torch.distributed (RCCL on the GPUs): all-reduce of the frame accumulators / folded scalars / the 7 sums of a PCG
// pass, halo exchange of contiguous row ranges of `blk`, the PCG records and `dist`.  DESIGN.md §7.
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
            c->mg_hist[arg] = NAN;   // "not published yet": psgsdf_mg_pcg_status watches the mapped slot
            timed(c, "pcg_pass", [&] { launch_cgf_pass(a, c->pcg_sc, c->pcg_part, G, rows, arg, mg_pcg_cap(c), c->mg_hist_dev + arg, c->stream); });
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
    // The kernels publish into host-mapped slots, so the outcome is visible while the passes enqueued behind the converged one
    // (no-ops) and their all-reduces are still draining: the host program can enqueue the rest of the iteration without a bubble.
    // A slot that is still NaN when the stream has drained belongs to a kernel that returned early (the solve had stopped).
    volatile double* hist = c->mg_hist;
    bool drained = false;
    for (int q = 0; q < n; ++q) {
        if (!drained && std::isnan(hist[k0 + q])) {
            const int w = wait_mapped(c, [hist, k0, q] { return !std::isnan(hist[k0 + q]); }, "mg_pcg_status");
            if (w < 0) return w;
            drained = w == 1;
        }
        if (std::isnan(hist[k0 + q])) break;
        if (k0 + q > 0 && (float)hist[k0 + q] < fmaxf(FLT_EPSILON * FLT_EPSILON * (float)hist[0], FLT_MIN)) break;   // decided: no need to wait for later slots
        if ((float)hist[0] == 0.f) break;
    }
    c->host_buf[0] = hist[0];
    for (int q = 0; q < n; ++q) c->host_buf[1 + q] = hist[k0 + q];
    const float rhsN = (float)c->host_buf[0];
    *iters = -1; *err = 0;
    if (rhsN == 0.f) { *iters = 0; c->last_cg_iters = 0; return PSGSDF_OK; }
    const float thr = fmaxf(FLT_EPSILON * FLT_EPSILON * rhsN, FLT_MIN);
    float rn2 = rhsN;
    for (int q = 0; q < n && *iters < 0; ++q) {
        const int kk = k0 + q;
        if (kk == 0) continue;
        if (std::isnan(c->host_buf[1 + q])) break;      // never published: an earlier kernel had already stopped the solve
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

}  // extern "C"
