// api.hip -- the C ABI of include/psgsdf.h: context, volume, keyframes, init, energies, steps, loops, downloads.
#include "engine_internal.h"

using namespace psge;

// ============================================================================================
// C ABI
// ============================================================================================
namespace psge {
int host_stage(psgsdf_ctx* c, size_t bytes, void** p) {
    if (c->pin_bytes < bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));      // (nothing may still be reading the old buffer)
        if (c->pin_stage) hipHostFree(c->pin_stage);
        c->pin_stage = nullptr; c->pin_bytes = 0;
        const size_t want = bytes + bytes / 4;
        HIPCHK(c, hipHostMalloc(&c->pin_stage, want, hipHostMallocDefault));
        c->pin_bytes = want;
    }
    *p = c->pin_stage;
    return 0;
}
}  // namespace psge

extern "C" {

#ifdef PSGSDF_DEV
const char* psgsdf_version(void) { return "psgsdf-hip gfx950 r5 dev (fault-injection and ablation knobs compiled in: libpsgsdf_dev.so, tests and tools only)"; }
#else
const char* psgsdf_version(void) { return "psgsdf-hip gfx950 r5"; }
#endif

// Every environment variable the engine reads (include/psgsdf.h lists them with their meaning).  The values are snapshotted when a context is created;
// psgsdf_get_tuning reports the snapshot and what it resolved to.  kDevKnobs exist only in the development build: they inject faults or make results
// WRONG on purpose and are compiled out of libpsgsdf.so.
static const char* const kKnobs[] = {
    "PSGSDF_PCG_POLL", "PSGSDF_SPECULATE", "PSGSDF_FOLD_IN_NEXT", "PSGSDF_FUSE_ALBEDO", "PSGSDF_FUSE_PCG_INIT", "PSGSDF_PCG_PERSIST", "PSGSDF_PCG_XCD_LOCAL",
    "PSGSDF_PCG_FUSE_ASM", "PSGSDF_PCG_FUSE_APPLY", "PSGSDF_PCG_PIPELINE", "PSGSDF_PCG_TAGM", "PSGSDF_PCG_PREFETCH", "PSGSDF_PCG_COL16", "PSGSDF_PCG_ROWS", "PSGSDF_PCG_BLOCKS",
    "PSGSDF_FM_SOLVE", "PSGSDF_FRAME_SOLVE", "PSGSDF_FM_ROWS", "PSGSDF_IMG_COMPACT", "PSGSDF_XCD_MAP", "PSGSDF_XCD_STRIPE",
    "PSGSDF_XR", "PSGSDF_XF", "PSGSDF_XS", "PSGSDF_XH", "PSGSDF_XR_MEM", "PSGSDF_XWAIT_LOG2", "PSGSDF_SPECULATE_MR", "PSGSDF_CU_MASK",
    "PSGSDF_WAIT_TIMEOUT_S", "PSGSDF_DESTROY_TIMEOUT_S", "PSGSDF_SOLVE_DUMP"};
static const char* const kDevKnobs[] = {"PSGSDF_PCG_ABLATE", "PSGSDF_FAULT_SOLVE", "PSGSDF_FAULT_HALO", "PSGSDF_MBOX_CHECK"};
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
    if (const char* e = getenv("PSGSDF_SPECULATE")) c->speculate = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_FOLD_IN_NEXT")) c->fold_in_next = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_FUSE_ALBEDO")) c->fuse_albedo = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_FUSE_PCG_INIT")) c->fuse_pcg_init = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_PCG_PERSIST")) c->pcg_persist = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_PCG_XCD_LOCAL")) c->pcg_xcd_local = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_AREG_DEVICE")) c->areg_device = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_PCG_FUSE_ASM")) c->pcg_fuse_asm = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_PCG_PIPELINE")) c->pcg_pipeline = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_PCG_TAGM")) { c->pcg_tagm = atoi(e) != 0; c->pcg_tagm_mr = atoi(e) >= 2; }
    if (const char* e = getenv("PSGSDF_PCG_PREFETCH")) c->pcg_prefetch = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_FM_SOLVE")) { c->fm_solve = atoi(e) != 0; c->fm_solve_led = atoi(e) == 1; }
    if (const char* e = getenv("PSGSDF_FRAME_SOLVE")) c->frame_solve = (!strcmp(e, "eigen") || !strcmp(e, "1")) ? 1 : 0;
    if (const char* e = getenv("PSGSDF_XF")) c->xf_enable = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_XS")) c->xs_enable = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_XH")) c->xh_enable = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_XWAIT_LOG2")) { const int l = atoi(e); if (l >= 8 && l <= 30) c->xwait_spins = 1 << l; }
    if (const char* e = getenv("PSGSDF_IMG_COMPACT")) c->img_compact = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_SPECULATE_MR")) c->speculate_mr = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_XCD_MAP")) c->xcd_map = atoi(e);
    if (const char* e = getenv("PSGSDF_XCD_STRIPE")) c->xcd_map = (c->xcd_map & 255) | (atoi(e) << 8);
#ifdef PSGSDF_DEV      // fault injection / ablation / the round-2 race: development build only (libpsgsdf_dev.so)
    if (const char* e = getenv("PSGSDF_PCG_ABLATE")) c->pcg_ablate = atoi(e) & 7;
    if (const char* e = getenv("PSGSDF_MBOX_CHECK")) c->mbox_check = atoi(e) != 0;
    if (const char* e = getenv("PSGSDF_FAULT_SOLVE")) c->fault_solve = atoi(e);
    if (const char* e = getenv("PSGSDF_FAULT_HALO")) c->fault_halo = atoll(e);
#endif
    for (const char* k : kKnobs) if (const char* e = getenv(k)) c->tuning_env.emplace_back(k, e);
    for (const char* k : kDevKnobs) if (const char* e = getenv(k)) {
#ifdef PSGSDF_DEV
        c->tuning_env.emplace_back(k, e);
#else
        c->tuning_ignored.emplace_back(k, e);      // (set, but this build does not know it: reported, never silently honoured)
#endif
    }
    if (const char* e = getenv("PSGSDF_PCG_FUSE_APPLY")) c->pcg_fuse_apply = atoi(e) != 0;
    if (hipDeviceGetAttribute(&c->num_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) c->num_cu = 0;
    c->set = *settings; c->reg_n = settings->reg_weight_n; c->reg_l = settings->reg_weight_l; c->reg_r = settings->reg_weight_rho;
    GridP& g = c->grid;
    for (int a = 0; a < 3; ++a) { g.dim[a] = grid->dim[a]; c->shift[a] = grid->shift[a]; }
    g.nvox = (long long)g.dim[0] * g.dim[1] * g.dim[2]; g.koff = 0;
    for (int a = 0; a < 3; ++a) c->gdim[a] = g.dim[a];
    c->gnvox = g.nvox; c->z0 = c->zlo = 0; c->z1 = c->zhi = g.dim[2];
    g.vs = grid->voxel_size; g.vs_inv = 1.f / g.vs; g.T = grid->truncation;
    for (int a = 0; a < 3; ++a) g.origin[a] = c->shift[a] - (float)(0.5 * (double)g.vs) * (float)g.dim[a];   // VoxelGrid.h:130
    c->cam.fx = K[0]; c->cam.fy = K[4]; c->cam.cx = K[2]; c->cam.cy = K[5];
    if (const char* e = getenv("PSGSDF_XR")) c->xr_enable = atoi(e) != 0;
    bool stream_ok;
    if (const char* e = getenv("PSGSDF_CU_MASK")) {      // "lo:hi": the context's stream runs on CUs [lo, hi) only -- two ranks sharing ONE GPU with half the CUs each (tests of the cross-rank persistent solve)
        int lo = 0, hi = 0;
        if (sscanf(e, "%d:%d", &lo, &hi) != 2 || lo < 0 || hi <= lo || hi > c->num_cu) { delete c; return PSGSDF_ERR_ARG; }
        std::vector<uint32_t> mask((size_t)(c->num_cu + 31) / 32, 0u);
        for (int i = lo; i < hi; ++i) mask[i / 32] |= 1u << (i % 32);
        stream_ok = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data()) == hipSuccess;
        c->cu_mask_lo = lo; c->cu_mask_hi = hi; c->num_cu = hi - lo;
    } else stream_ok = hipStreamCreate(&c->stream) == hipSuccess;
    bool ok = stream_ok
        && hipMalloc(&c->pcg_sc, sizeof(double) * (16 + 8 * (size_t)kPcgMaxBlocks)) == hipSuccess   // fs[0..1] + stage stamps of the timing hook
        && hipMalloc(&c->pcg_part, sizeof(double) * 14 * kPcgMaxBlocks) == hipSuccess
        && hipMalloc(&c->pcg_gran, sizeof(double) * 2 * kSolveGranPlanes * kSolveMaxBlocksHost) == hipSuccess
        && hipMalloc(&c->mg_scal, sizeof(double) * kMgScal) == hipSuccess && hipMalloc(&c->mg_ext, sizeof(double) * 8) == hipSuccess
        && hipMalloc(&c->d_need, 2 * sizeof(int)) == hipSuccess
        && hipMalloc(&c->d_total, sizeof(int)) == hipSuccess
        && hipMalloc(&c->led_light, sizeof(float) * 3) == hipSuccess
        && hipMalloc(&c->fs_stats, sizeof(double) * 8) == hipSuccess && hipMemset(c->fs_stats, 0, sizeof(double) * 8) == hipSuccess
        && hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess;
    if (!ok) { delete c; return PSGSDF_ERR_DEVICE; }
    *out = c;
    return PSGSDF_OK;
}

void psgsdf_destroy(psgsdf_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    // multi-rank: this rank closes its mappings of the others' record planes / mailbox regions (telling their owners) and frees what IT exported
    // only once every rank that mapped it has reported the same -- bounded, and no collective: a rank that failed, left early or destroys its
    // contexts in another order cannot hang this call.  If a peer never reports, the exported allocations are leaked instead of freed.
    double qt = 15.0; if (const char* e = getenv("PSGSDF_DESTROY_TIMEOUT_S")) qt = atof(e);
    if (xr_quiesce(c, qt)) {
        c->leak_exported = true;
        fprintf(stderr, "psgsdf: rank %d: a peer did not close its mappings of this rank's exchange memory within %.0f s (failed or still running?): that memory is leaked, not freed\n", c->rank, qt);
    }
    free_dense(c);
    hipFree(c->vis_seq); hipFree(c->frame_idx); hipFree(c->img); hipFree(c->img8); hipFree(c->frames); hipFree(c->frames_undo); hipFree(c->led_light); hipFree(c->fs_stats);
    hipFree(c->band_mem); if (!c->leak_exported) hipFree(c->rec_mem); hipFree(c->obs_mem); hipFree(c->stage);
    hipFree(c->ncache); hipFree(c->ntmp); hipFree(c->nout); hipFree(c->ndepth); hipFree(c->track_part); if (c->track_host) hipHostFree(c->track_host); hipFree(c->acc_frame); hipFree(c->frame_part); hipFree(c->frame_done); hipFree(c->part); hipFree(c->pcg_sc); hipFree(c->pcg_part); hipFree(c->pcg_gran); hipFree(c->d_total);
    for (void* p : c->xo_host) if (p) hipHostFree(p);
    if (c->pin_stage) hipHostFree(c->pin_stage);
    if (c->host_buf) hipHostFree(c->host_buf);
    if (c->mbox) hipHostFree(c->mbox);
    if (c->ev0) hipEventDestroy(c->ev0); if (c->ev1) hipEventDestroy(c->ev1);
    for (auto& pr : c->watch_pool) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    if (!c->leak_exported) { hipFree(c->xr); hipFree(c->hx_mem); }
    hipFree(c->xf_table); hipFree(c->vm_order);
    comm_destroy(c);
    hipFree(c->areg_mem); hipFree(c->mg_scal); hipFree(c->mg_ext); hipFree(c->mbox_shadow); hipFree(c->d_need);
    if (c->stream && c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

// Multi-rank: where to cut the volume into z-slabs of (about) equal band count.  Every rank counts the band candidates (|d| <= sqrt(3) vs and
// seen in some frame: OptimizerAux.cpp:249 before the keyframes are selected) of ITS share of the z-planes on the host, one all-reduce makes
// the per-plane histogram global, and every rank takes the same cuts from its prefix sum.
static double count_plane(const psgsdf_ctx* c, const float* d, const uint64_t* v, int wpv) {
    const long long plane = (long long)c->gdim[0] * c->gdim[1];
    const float lim = (float)(sqrt(3.0) * (double)c->grid.vs);
    long long m = 0;
    for (long long i = 0; i < plane; ++i) { if (!(fabsf(d[i]) <= lim)) continue; bool seen = false; for (int w = 0; w < wpv && !seen; ++w) seen = v[(size_t)i * wpv + w] != 0; m += seen; }
    return (double)m;
}
// cnt[nz]: this rank's contribution to the per-plane histogram (0 for planes another rank counted) -> the planes [z0, z1) this rank owns
static int cut_slabs(psgsdf_ctx* c, std::vector<double>& cnt, int* z0, int* z1) {
    const int nz = c->gdim[2], n = c->n_ranks;
    if (nz < n) return fail(c, PSGSDF_ERR_UNSUPPORTED, "%d z-planes cannot be cut into %d slabs", nz, n);
    if (n > 1) {
        double* d_cnt = nullptr;
        HIPCHK(c, hipMalloc(&d_cnt, sizeof(double) * nz));
        int rc = 0;
        if (hipMemcpyAsync(d_cnt, cnt.data(), sizeof(double) * nz, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "slab histogram upload");
        if (!rc) rc = comm_allreduce(c, d_cnt, nz);
        if (!rc && (hipMemcpyAsync(cnt.data(), d_cnt, sizeof(double) * nz, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) rc = fail(c, PSGSDF_ERR_DEVICE, "slab histogram download");
        hipFree(d_cnt);
        if (rc) return rc;
    }
    double total = 0; for (double x : cnt) total += x;
    if (total <= 0) return fail(c, PSGSDF_ERR_UNSUPPORTED, "the volume has no band candidates to partition");
    // cut r = first plane at which the running count reaches r/n of the total; every slab keeps at least one plane
    std::vector<int> cut(n + 1, 0); cut[n] = nz;
    double run = 0; int r = 1;
    for (int k = 0; k < nz && r < n; ++k) { run += cnt[k]; while (r < n && run >= total * r / n) { cut[r] = k + 1; ++r; } }
    for (int q = 1; q < n; ++q) cut[q] = std::max(cut[q], cut[q - 1] + 1);
    for (int q = n - 1; q >= 1; --q) cut[q] = std::min(cut[q], cut[q + 1] - 1);
    *z0 = cut[c->rank]; *z1 = cut[c->rank + 1];
    return 0;
}
static int choose_slab(psgsdf_ctx* c, const float* dist, const uint64_t* vis_words, int wpv, int* z0, int* z1) {
    const int nz = c->gdim[2], n = c->n_ranks;
    const long long plane = (long long)c->gdim[0] * c->gdim[1];
    std::vector<double> cnt(nz, 0.0);
    for (int k = c->rank; k < nz; k += n) cnt[k] = count_plane(c, dist + (size_t)k * plane, vis_words + (size_t)k * plane * wpv, wpv);
    return cut_slabs(c, cnt, z0, z1);
}

// the planes [zlo, zhi) = own planes [z0, z1) + one halo plane per inner side, from arrays that hold exactly those planes (`src_n` voxels per plane set)
static int upload_planes(psgsdf_ctx* c, int z0, int z1, size_t src_n, size_t src_off, const float* dist, const float* grad_xyz, const float* weight, const float* rgb, const uint64_t* vis_words, int words_per_voxel) {
    { int rc = set_local_grid(c, z0, z1); if (rc) return rc; }
    const long long n = c->grid.nvox;
    free_dense(c);
    if (c->vis_seq) { hipFree(c->vis_seq); c->vis_seq = nullptr; }
    int rc = alloc_dense(c, c->dense, n, 0, true); if (rc) return rc;
    HIPCHK(c, hipMalloc(&c->vis_seq, sizeof(uint64_t) * n * words_per_voxel));
    c->wpv_seq = words_per_voxel;
    HIPCHK(c, hipMemcpyAsync(c->dense.dist, dist + src_off, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    for (int a = 0; a < 3; ++a) {
        HIPCHK(c, hipMemcpyAsync(c->dense.g[a], grad_xyz + (size_t)a * src_n + src_off, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->dense.rho[a], rgb + (size_t)a * src_n + src_off, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->dense.weight, weight + src_off, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->vis_seq, vis_words + src_off * words_per_voxel, sizeof(uint64_t) * n * words_per_voxel, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_volume = true; c->inited = false;
    return PSGSDF_OK;
}

int psgsdf_upload_volume(psgsdf_ctx* c, const float* dist, const float* grad_xyz, const float* weight, const float* rgb, const uint64_t* vis_words, int words_per_voxel) {
    if (!c || !dist || !grad_xyz || !weight || !rgb || !vis_words || words_per_voxel < 1) return fail(c, PSGSDF_ERR_ARG, "upload_volume: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    // the caller hands over the WHOLE volume; a rank of a multi-rank run keeps its z-slab (+ one halo plane per inner side) of it on the device
    // (a host that cannot or should not hold the whole volume on every rank uses psgsdf_plan_slab + psgsdf_upload_volume_slab instead)
    int z0 = 0, z1 = c->gdim[2];
    if (c->n_ranks > 1) { int rc = choose_slab(c, dist, vis_words, words_per_voxel, &z0, &z1); if (rc) return rc; }
    const int zlo = std::max(0, z0 - 1);
    return upload_planes(c, z0, z1, (size_t)c->gnvox, (size_t)zlo * c->gdim[0] * c->gdim[1], dist, grad_xyz, weight, rgb, vis_words, words_per_voxel);
}

// ---- slab-local upload: no rank ever touches the whole volume ----------------------------------------------
int psgsdf_slab_plane_count(psgsdf_ctx* c, const float* dist_plane, const uint64_t* vis_plane, int words_per_voxel, double* count) {
    if (!c || !dist_plane || !vis_plane || !count || words_per_voxel < 1) return fail(c, PSGSDF_ERR_ARG, "slab_plane_count: null argument");
    *count = count_plane(c, dist_plane, vis_plane, words_per_voxel);
    return PSGSDF_OK;
}
int psgsdf_plan_slab(psgsdf_ctx* c, const double* plane_counts, int* z0, int* z1) {
    if (!c || !plane_counts || !z0 || !z1) return fail(c, PSGSDF_ERR_ARG, "plan_slab: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<double> cnt(plane_counts, plane_counts + c->gdim[2]);
    return cut_slabs(c, cnt, z0, z1);
}
int psgsdf_upload_volume_slab(psgsdf_ctx* c, int z0, int z1, const float* dist, const float* grad_xyz, const float* weight, const float* rgb, const uint64_t* vis_words, int words_per_voxel) {
    if (!c || !dist || !grad_xyz || !weight || !rgb || !vis_words || words_per_voxel < 1) return fail(c, PSGSDF_ERR_ARG, "upload_volume_slab: null argument");
    if (z0 < 0 || z1 > c->gdim[2] || z1 <= z0) return fail(c, PSGSDF_ERR_ARG, "upload_volume_slab: planes [%d, %d) of %d", z0, z1, c->gdim[2]);
    HIPCHK(c, hipSetDevice(c->device));
    const int zlo = std::max(0, z0 - 1), zhi = std::min(c->gdim[2], z1 + 1);
    return upload_planes(c, z0, z1, (size_t)(zhi - zlo) * c->gdim[0] * c->gdim[1], 0, dist, grad_xyz, weight, rgb, vis_words, words_per_voxel);
}

int psgsdf_volume_init(psgsdf_ctx* c, int max_frames) {
    if (!c || max_frames < 1) return fail(c, PSGSDF_ERR_ARG, "volume_init: max_frames");
    // multi-rank: fusion is slab-parallel (VolumetricGradSdf.cpp:78-134 touches every voxel independently).  The band does not exist yet, so the
    // volume is cut into slabs of equal HEIGHT; psgsdf_rebalance_slabs moves planes to slabs of equal band count once the frames are fused.
    if (c->n_ranks > 1 && c->gdim[2] < c->n_ranks) return fail(c, PSGSDF_ERR_UNSUPPORTED, "%d z-planes cannot be cut into %d slabs", c->gdim[2], c->n_ranks);
    { int rc0 = c->n_ranks > 1 ? set_local_grid(c, (int)((long long)c->rank * c->gdim[2] / c->n_ranks), (int)((long long)(c->rank + 1) * c->gdim[2] / c->n_ranks)) : set_local_grid(c, 0, c->gdim[2]); if (rc0) return rc0; }
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

// Re-cut a slab-parallel volume (psgsdf_volume_init cut it by height) into slabs of equal band-candidate count -- the partition every other entry
// point works on (psgsdf_upload_volume: choose_slab) -- and move the planes: every rank counts the candidates of the planes it owns, one all-reduce
// makes the histogram global, all ranks take the same cuts, and each plane travels from its old owner to every rank that holds it under the new
// cut (own planes + one halo plane per inner side).  Collective.
int psgsdf_rebalance_slabs(psgsdf_ctx* c) {
    if (!c || !c->have_volume || !c->vis_seq) return fail(c, PSGSDF_ERR_STATE, "rebalance_slabs: volume first");
    if (c->n_ranks <= 1) return PSGSDF_OK;
    if (!c->comm) return fail(c, PSGSDF_ERR_COMM, "rank %d of %d has no communicator (psgsdf_comm_init)", c->rank, c->n_ranks);
    HIPCHK(c, hipSetDevice(c->device));
    const int R = c->n_ranks, nz = c->gdim[2], wpv = c->wpv_seq;
    const size_t plane = (size_t)c->gdim[0] * c->gdim[1];
    // band candidates of the planes this rank owns (|d| <= sqrt(3) vs and seen by some frame: OptimizerAux.cpp:249 before the keyframes are selected)
    const int oz0 = c->z0, oz1 = c->z1, ozlo = c->zlo;
    std::vector<float> hd((size_t)(oz1 - oz0) * plane); std::vector<uint64_t> hv((size_t)(oz1 - oz0) * plane * wpv);
    HIPCHK(c, hipMemcpyAsync(hd.data(), c->dense.dist + (size_t)(oz0 - ozlo) * plane, sizeof(float) * hd.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hv.data(), c->vis_seq + (size_t)(oz0 - ozlo) * plane * wpv, sizeof(uint64_t) * hv.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<double> cnt(nz, 0.0);
    for (int k = oz0; k < oz1; ++k) cnt[k] = count_plane(c, hd.data() + (size_t)(k - oz0) * plane, hv.data() + (size_t)(k - oz0) * plane * wpv, wpv);
    int nz0 = 0, nz1 = nz;
    { int rc = cut_slabs(c, cnt, &nz0, &nz1); if (rc) return rc; }
    // every rank's old and new planes
    std::vector<double> cuts((size_t)4 * R, 0.0);
    cuts[4 * c->rank] = oz0; cuts[4 * c->rank + 1] = oz1; cuts[4 * c->rank + 2] = nz0; cuts[4 * c->rank + 3] = nz1;
    { int rc = host_allreduce(c, cuts, "rebalance_slabs"); if (rc) return rc; }
    // the new local grid and its arrays.  The new arrays are allocated BEFORE the context lets go of the old ones, and every error path frees what it
    // allocated and leaves the context as it was (a failed allocation used to leak both sets and leave have_volume = true over empty arrays)
    DenseView od = c->dense; uint64_t* ovis = c->vis_seq;
    const int nzlo = std::max(0, nz0 - 1), nzhi = std::min(nz, nz1 + 1);
    const long long n = (long long)(nzhi - nzlo) * (long long)plane;
    DenseView nd{};
    uint64_t* nvis = nullptr;
    auto free_new = [&] { hipFree(nd.dist); for (int a = 0; a < 3; ++a) { hipFree(nd.g[a]); hipFree(nd.rho[a]); } hipFree(nd.weight); hipFree(nd.vis); hipFree(nd.row_of); hipFree(nvis); (void)hipGetLastError(); };
    int rc = alloc_dense(c, nd, n, 0, true);
    if (!rc && hipMalloc(&nvis, sizeof(uint64_t) * n * wpv) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "rebalance_slabs: out of memory");
    if (rc) { free_new(); return rc; }
    { int rc2 = set_local_grid(c, nz0, nz1); if (rc2) { free_new(); set_local_grid(c, oz0, oz1); return rc2; } }
    c->dense = DenseView{}; c->vis_seq = nullptr;
    float* oarr[8] = {od.dist, od.g[0], od.g[1], od.g[2], od.rho[0], od.rho[1], od.rho[2], od.weight};
    float* narr[8] = {nd.dist, nd.g[0], nd.g[1], nd.g[2], nd.rho[0], nd.rho[1], nd.rho[2], nd.weight};
    std::vector<psgsdf_comm_xfer> sends, recvs;
    for (int peer = 0; peer < R; ++peer) {
        const int pz0 = (int)cuts[4 * peer], pz1 = (int)cuts[4 * peer + 1], pn0 = (int)cuts[4 * peer + 2], pn1 = (int)cuts[4 * peer + 3];
        const int pnlo = std::max(0, pn0 - 1), pnhi = std::min(nz, pn1 + 1);
        // what I own (old cut) and `peer` holds (new cut): I send (or copy, if I am the peer)
        const int s0 = std::max(oz0, pnlo), s1 = std::min(oz1, pnhi);
        if (s1 > s0) {
            const size_t so = (size_t)(s0 - ozlo) * plane, cntv = (size_t)(s1 - s0) * plane;
            if (peer == c->rank) {
                const size_t dn = (size_t)(s0 - nzlo) * plane;
                for (int q = 0; q < 8; ++q) if (hipMemcpyAsync(narr[q] + dn, oarr[q] + so, sizeof(float) * cntv, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "rebalance_slabs: plane copy");
                if (hipMemcpyAsync(nvis + dn * wpv, ovis + so * wpv, sizeof(uint64_t) * cntv * wpv, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "rebalance_slabs: plane copy");
            } else {
                for (int q = 0; q < 8; ++q) sends.push_back({oarr[q] + so, sizeof(float) * cntv, peer});
                sends.push_back({ovis + so * wpv, sizeof(uint64_t) * cntv * wpv, peer});
            }
        }
        // what `peer` owns (old cut) and I hold (new cut): I receive
        if (peer != c->rank) {
            const int r0 = std::max(pz0, nzlo), r1 = std::min(pz1, nzhi);
            if (r1 > r0) {
                const size_t dn = (size_t)(r0 - nzlo) * plane, cntv = (size_t)(r1 - r0) * plane;
                for (int q = 0; q < 8; ++q) recvs.push_back({narr[q] + dn, sizeof(float) * cntv, peer});
                recvs.push_back({nvis + dn * wpv, sizeof(uint64_t) * cntv * wpv, peer});
            }
        }
    }
    if (!rc) rc = comm_xfer(c, sends, recvs);      // (every rank enters the exchange unless its own local copies failed)
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "rebalance_slabs: exchange");
    for (int q = 0; q < 8; ++q) hipFree(oarr[q]);
    hipFree(od.vis); hipFree(od.row_of); hipFree(ovis); hipFree(c->block_sums); c->block_sums = nullptr;
    c->dense = nd; c->vis_seq = nvis;
    c->inited = false;
    if (rc) c->have_volume = false;      // (a failed exchange leaves planes missing: the volume must be uploaded / fused again)
    return rc;
}

int psgsdf_integrate_frame(psgsdf_ctx* c, const float* rgb, const float* depth, const float* normals_xyz, int width, int height, const float pose[16], int counter, float z_min, float z_max) {
    if (!c || !c->have_volume || !c->vis_seq) return fail(c, PSGSDF_ERR_STATE, "integrate_frame: volume_init or upload_volume first");
    if (!rgb || !depth || !pose || width < 2 || height < 2 || counter < 0) return fail(c, PSGSDF_ERR_ARG, "integrate_frame: bad argument");      // (multi-rank: every rank fuses the frame into the planes it holds, halo planes included -- no exchange)
    HIPCHK(c, hipSetDevice(c->device));
    if (counter >= 64 * c->wpv_seq) {   // the sequence is longer than volume_init was told (the reference's vector<bool> simply grows): widen the per-voxel words
        const long long n = c->grid.nvox;
        const int wnew = std::max(2 * c->wpv_seq, counter / 64 + 1);
        uint64_t* grown = nullptr;
        HIPCHK(c, hipMalloc(&grown, sizeof(uint64_t) * n * wnew));
        HIPCHK(c, hipMemsetAsync(grown, 0, sizeof(uint64_t) * n * wnew, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(grown, sizeof(uint64_t) * wnew, c->vis_seq, sizeof(uint64_t) * c->wpv_seq, sizeof(uint64_t) * c->wpv_seq, (size_t)n, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->vis_seq); c->vis_seq = grown; c->wpv_seq = wnew;
    }
    const size_t npx = (size_t)width * height;
    if (c->stage_px < npx) {
        hipFree(c->stage); c->stage = nullptr; c->stage_px = 0;
        HIPCHK(c, hipMalloc(&c->stage, sizeof(float) * npx * 7));
        c->stage_px = npx;
    }
    float* d_rgb = c->stage; float* d_depth = c->stage + 3 * npx; float* d_nrm = c->stage + 4 * npx;
    // The caller's arrays go through engine-owned PINNED memory: a DMA straight from pageable memory makes the runtime pin the caller's pages for every call --
    // 5-11 ms per 380 x 570 frame when the frames come out of fresh allocations (voxelPS's prefetching decoder), against 0.3 ms for the copy into the staging
    // buffer (profiles/r05_notes.md section 8).  One transfer: rgb | depth | normals are contiguous on both sides.
    const size_t nfl = npx * (normals_xyz ? 7 : 4);
    void* hs = nullptr;
    { int src = host_stage(c, sizeof(float) * nfl, &hs); if (src) return src; }
    memcpy(hs, rgb, sizeof(float) * 3 * npx); memcpy((float*)hs + 3 * npx, depth, sizeof(float) * npx);
    if (normals_xyz) memcpy((float*)hs + 4 * npx, normals_xyz, sizeof(float) * 3 * npx);
    HIPCHK(c, hipMemcpyAsync(d_rgb, hs, sizeof(float) * nfl, hipMemcpyHostToDevice, c->stream));
    if (!normals_xyz) {      // normals_xyz == NULL: NormalEstimator::compute on the depth map that is on the device already (what VolumetricGradSdf::update does, VolumetricGradSdf.cpp:59-61) -- no 3-plane round trip through the host
        int nrc = frontend_normals_dev(c, d_depth, width, height, d_nrm); if (nrc) return nrc;
    }
    Cam cam = c->cam; cam.W = width; cam.H = height;
    FrameP fp{};
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) fp.R[i * 3 + j] = pose[i * 4 + j]; fp.t[i] = pose[i * 4 + 3]; }
    timed(c, "integrate_frame", [&] { launch_integrate(c->dense, c->vis_seq, c->wpv_seq, c->grid, cam, fp, d_rgb, d_depth, d_nrm, counter, z_min, z_max, c->stream); });
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->inited = false;
    return PSGSDF_OK;
}

// everything of set_keyframes but the pixels; exactly one of rgb_f32 / rgb_u8 is given
static int set_keyframes_impl(psgsdf_ctx* c, int n_frames, const int32_t* frame_idx, const float* rgb_f32, const uint8_t* rgb_u8, float scale,
                              int width, int height, const float* poses, const float* const* rgb_f32_frames = nullptr) {
    if (!c || n_frames <= 0 || !frame_idx || (!rgb_f32 && !rgb_u8 && !rgb_f32_frames) || !poses || width <= 1 || height <= 1) return fail(c, PSGSDF_ERR_ARG, "set_keyframes: bad argument");
    if (rgb_f32_frames) for (int f = 0; f < n_frames; ++f) if (!rgb_f32_frames[f]) return fail(c, PSGSDF_ERR_ARG, "set_keyframes: image %d is NULL", f);
    if (n_frames > kMaxFramesLds) return fail(c, PSGSDF_ERR_UNSUPPORTED, "at most %d keyframes", kMaxFramesLds);
    if (rgb_u8 && ((size_t)n_frames * width * height >= ((size_t)1 << 30) || (size_t)n_frames * height >= ((size_t)1 << 24))) return fail(c, PSGSDF_ERR_UNSUPPORTED, "8-bit keyframes: at most 2^30 pixels and 2^24 image rows");
    HIPCHK(c, hipSetDevice(c->device));
    hipFree(c->frame_idx); hipFree(c->img); hipFree(c->img8); hipFree(c->frames); hipFree(c->frames_undo); hipFree(c->acc_frame);
    c->frame_idx = nullptr; c->img = nullptr; c->img8 = nullptr; c->frames = nullptr; c->frames_undo = nullptr; c->acc_frame = nullptr;
    c->F = n_frames; c->cam.W = width; c->cam.H = height;
    const size_t npix = (size_t)n_frames * width * height;
    HIPCHK(c, hipMalloc(&c->frame_idx, sizeof(int) * n_frames));
    HIPCHK(c, hipMalloc(&c->frames, sizeof(FrameP) * n_frames));
    HIPCHK(c, hipMalloc(&c->frames_undo, sizeof(FrameP) * n_frames + 16));
    c->acc_frame_n = (size_t)n_frames * 64;
    HIPCHK(c, hipMalloc(&c->acc_frame, sizeof(double) * c->acc_frame_n));
    HIPCHK(c, hipMemsetAsync(c->acc_frame, 0, sizeof(double) * c->acc_frame_n, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->frame_idx, frame_idx, sizeof(int) * n_frames, hipMemcpyHostToDevice, c->stream));
    if (rgb_f32 || rgb_f32_frames) {
        HIPCHK(c, hipMalloc(&c->img, sizeof(float) * npix * 3));
        const size_t per = (size_t)width * height * 3;
        if (rgb_f32) HIPCHK(c, hipMemcpyAsync(c->img, rgb_f32, sizeof(float) * npix * 3, hipMemcpyHostToDevice, c->stream));
        else for (int f = 0; f < n_frames; ++f) HIPCHK(c, hipMemcpyAsync(c->img + per * f, rgb_f32_frames[f], sizeof(float) * per, hipMemcpyHostToDevice, c->stream));      // one image per allocation (the reference's std::vector<cv::Mat>): no gather on the host
        // Keyframes that came out of 8-bit files (the reference's loader: imread + convertTo(CV_32FC3, 1.0f / 255.0f)) are kept as RGBA8 words: the
        // samplers give back the SAME floats (device_common.h unpack_rgb8), with two tap loads per observation instead of four and a third of the bytes.
        c->img_compacted = false;
        if (c->img_compact && npix < ((size_t)1 << 30) && (size_t)n_frames * height < ((size_t)1 << 24)) {
            const float scale = 1.0f / 255.0f;
            // a sample first (every 61st pixel, nothing written: data that is not 8-bit at all is recognised in microseconds), the full pass only behind a clean sample
            int* d_flags = nullptr; std::vector<int> h_flags(kTryPackFlags);
            HIPCHK(c, hipMalloc(&d_flags, sizeof(int) * kTryPackFlags));
            auto any_miss = [&](unsigned* out, size_t stride, int* bad) -> int {
                if (hipMemsetAsync(d_flags, 0, sizeof(int) * kTryPackFlags, c->stream) != hipSuccess) return fail(c, PSGSDF_ERR_DEVICE, "set_keyframes: 8-bit check");
                launch_try_pack_f32(c->img, out, npix, stride, scale, d_flags, c->stream);
                if (hipMemcpyAsync(h_flags.data(), d_flags, sizeof(int) * kTryPackFlags, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return fail(c, PSGSDF_ERR_DEVICE, "set_keyframes: 8-bit check");
                *bad = 0; for (int v : h_flags) *bad |= v;
                return 0;
            };
            int bad = 1;
            int prc = any_miss(nullptr, 61, &bad);
            if (!prc && !bad) {
                if (hipMalloc(&c->img8, sizeof(unsigned) * npix) != hipSuccess) prc = fail(c, PSGSDF_ERR_DEVICE, "set_keyframes: out of memory");
                else prc = any_miss(c->img8, 1, &bad);
            }
            hipFree(d_flags);
            if (prc) return prc;
            if (bad) { hipFree(c->img8); c->img8 = nullptr; }
            else { hipFree(c->img); c->img = nullptr; c->img_scale = scale; c->img_compacted = true; }
        }
    } else {
        uint8_t* tmp = nullptr;
        HIPCHK(c, hipMalloc(&c->img8, sizeof(unsigned) * npix));
        HIPCHK(c, hipMalloc(&tmp, npix * 3));
        HIPCHK(c, hipMemcpyAsync(tmp, rgb_u8, npix * 3, hipMemcpyHostToDevice, c->stream));
        launch_pack_rgb8(tmp, c->img8, npix, c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(tmp);
        c->img_scale = scale;
    }
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
int psgsdf_set_keyframes(psgsdf_ctx* c, int n_frames, const int32_t* frame_idx, const float* rgb_images, int width, int height, const float* poses) {
    return set_keyframes_impl(c, n_frames, frame_idx, rgb_images, nullptr, 0.f, width, height, poses);
}
int psgsdf_set_keyframes_frames(psgsdf_ctx* c, int n_frames, const int32_t* frame_idx, const float* const* rgb_images, int width, int height, const float* poses) {
    return set_keyframes_impl(c, n_frames, frame_idx, nullptr, nullptr, 0.f, width, height, poses, rgb_images);
}
int psgsdf_set_keyframes_u8(psgsdf_ctx* c, int n_frames, const int32_t* frame_idx, const uint8_t* rgb_images, float scale, int width, int height, const float* poses) {
    return set_keyframes_impl(c, n_frames, frame_idx, nullptr, rgb_images, scale, width, height, poses);
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
    if (led) {   // computeLightIntensive, LedOptimizer.cpp:76-112 (multi-rank: read_parts delivers the sums over all slabs)
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

int psgsdf_set_record_observer(psgsdf_ctx* c, psgsdf_iter_cb observer, void* user) {
    if (!c) return PSGSDF_ERR_ARG;
    c->observer = observer; c->observer_user = user;
    return PSGSDF_OK;
}

int psgsdf_set_on_iter_period(psgsdf_ctx* c, int period) {
    if (!c || period < 1) return PSGSDF_ERR_ARG;
    c->on_iter_period = period;
    return PSGSDF_OK;
}

int psgsdf_upsample2x(psgsdf_ctx* c) {
    if (!c || !c->inited) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    return do_upsample(c);
}

int psgsdf_get_info(psgsdf_ctx* c, psgsdf_info* info) {
    if (!c || !info) return PSGSDF_ERR_ARG;
    for (int a = 0; a < 3; ++a) { info->dim[a] = c->gdim[a]; info->origin[a] = c->grid.origin[a]; }     // the WHOLE volume, also on a slab
    info->voxel_size = c->grid.vs; info->n_frames = c->F; info->n_band = c->inited ? (int)c->S_global : 0;
    info->light_stride = c->set.model == PSGSDF_LED ? 3 : (c->set.model == PSGSDF_SH2 ? 9 : 4);
    info->vis_words = c->dense.KW; info->reg_weight_n = c->reg_n; info->reg_weight_l = c->reg_l;
    return PSGSDF_OK;
}

int psgsdf_download_volume(psgsdf_ctx* c, float* dist, float* grad_xyz, float* weight, float* rgb, uint64_t* vis_words) {
    if (!c || !c->have_volume) return fail(c, PSGSDF_ERR_STATE, "no volume");
    HIPCHK(c, hipSetDevice(c->device));
    // The caller's arrays are those of the WHOLE volume.  A rank of a multi-rank run writes the z-planes it OWNS, [z0, z1) (psgsdf_mg_info),
    // and leaves the rest of the arrays untouched: the slabs of all ranks tile the volume.
    const size_t plane = (size_t)c->gdim[0] * c->gdim[1];
    const size_t src = (size_t)(c->z0 - c->zlo) * plane, dst = (size_t)c->z0 * plane, n = (size_t)(c->z1 - c->z0) * plane, gn = (size_t)c->gnvox;
    if (c->inited) launch_band_scatter(c->dense, c->band, c->stream);
    if (dist) HIPCHK(c, hipMemcpyAsync(dist + dst, c->dense.dist + src, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    for (int a = 0; a < 3; ++a) {
        if (grad_xyz) HIPCHK(c, hipMemcpyAsync(grad_xyz + (size_t)a * gn + dst, c->dense.g[a] + src, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
        if (rgb) HIPCHK(c, hipMemcpyAsync(rgb + (size_t)a * gn + dst, c->dense.rho[a] + src, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    }
    if (weight) HIPCHK(c, hipMemcpyAsync(weight + dst, c->dense.weight + src, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    if (vis_words) {
        if (!c->dense.vis) return fail(c, PSGSDF_ERR_STATE, "visibility not selected yet");
        HIPCHK(c, hipMemcpyAsync(vis_words + dst * c->dense.KW, c->dense.vis + src * c->dense.KW, sizeof(uint64_t) * n * c->dense.KW, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PSGSDF_OK;
}

int psgsdf_download_vis_seq(psgsdf_ctx* c, uint64_t* out) {
    if (!c || !c->vis_seq || !out) return PSGSDF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    // (the caller's array is the whole volume's; a slab writes the planes it owns, like psgsdf_download_volume)
    const size_t plane = (size_t)c->gdim[0] * c->gdim[1] * c->wpv_seq;
    HIPCHK(c, hipMemcpy(out + (size_t)c->z0 * plane, c->vis_seq + (size_t)(c->z0 - c->zlo) * plane, sizeof(uint64_t) * (size_t)(c->z1 - c->z0) * plane, hipMemcpyDeviceToHost));
    return c->wpv_seq;
}

int psgsdf_download_band(psgsdf_ctx* c, int32_t* lin_idx) {
    if (!c || !c->inited || !lin_idx) return fail(c, PSGSDF_ERR_STATE, "init first");
    HIPCHK(c, hipSetDevice(c->device));
    // surface_points_ in the indexing of the WHOLE volume.  One rank: all n_band of them.  A slab: its own rows only (row1 - row0 of
    // psgsdf_mg_info entries, ascending; the slabs' lists concatenated in rank order are the whole band).
    const int n = c->row1 - c->row0;
    if (n <= 0) return PSGSDF_OK;
    HIPCHK(c, hipMemcpy(lin_idx, c->band.lin + c->row0, sizeof(int) * n, hipMemcpyDeviceToHost));
    const long long shift = (long long)c->zlo * c->gdim[0] * c->gdim[1];
    if (shift) for (int i = 0; i < n; ++i) lin_idx[i] = (int32_t)(lin_idx[i] + shift);
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
