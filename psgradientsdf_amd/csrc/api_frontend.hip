// api_frontend.hip -- C ABI of the state producers next to the hot path: FALS normals and the depth tracker.
#include "engine_internal.h"

using namespace psge;

// ---- front end: FALS normals and depth tracker -------------------------------------------
namespace {
// NormalEstimator::cache (NormalEstimator.h:52-125), once per image size: 9 float planes on the device
int normals_cache(psgsdf_ctx* c, int W, int H) {
    if (c->ncache && c->ncache_w == W && c->ncache_h == H) return 0;
    const size_t n = (size_t)W * H;
    hipFree(c->ncache); hipFree(c->ntmp); hipFree(c->nout); hipFree(c->ndepth); c->ncache = nullptr; c->ntmp = nullptr; c->nout = nullptr; c->ndepth = nullptr;
    // on the device (round 5; frontend.hip k_ncache_*): the same doubles in the same order as the host loop this replaces
    HIPCHK(c, hipMalloc(&c->ncache, sizeof(float) * 9 * n)); HIPCHK(c, hipMalloc(&c->ntmp, sizeof(double) * 3 * n));
    HIPCHK(c, hipMalloc(&c->nout, sizeof(float) * 3 * n)); HIPCHK(c, hipMalloc(&c->ndepth, sizeof(float) * n));
    double* work = nullptr;
    HIPCHK(c, hipMalloc(&work, sizeof(double) * 12 * n));
    timed(c, "normals_cache", [&] { launch_normals_cache(W, H, 5, 1. / (double)c->cam.fx, 1. / (double)c->cam.fy, (double)c->cam.cx, (double)c->cam.cy, work, c->ncache, c->stream); });
    const hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(work);
    if (e != hipSuccess) return fail(c, PSGSDF_ERR_DEVICE, "normals cache: %s", hipGetErrorString(e));
    c->ncache_w = W; c->ncache_h = H;
    return 0;
}
}  // namespace

// FALS normals of a depth map that is on the device already, into 3 device planes (psgsdf_integrate_frame with normals_xyz == NULL)
namespace psge {
int frontend_normals_dev(psgsdf_ctx* c, const float* d_depth, int width, int height, float* d_normals) {
    int rc = normals_cache(c, width, height); if (rc) return rc;
    timed(c, "normals", [&] { launch_normals(d_depth, c->ncache, width, height, 5, c->ntmp, d_normals, c->stream); });
    return 0;
}
}  // namespace psge

extern "C" int psgsdf_debug_normals_cache(psgsdf_ctx* c, int width, int height, float* cache9) {
    if (!c || !cache9 || width < 2 || height < 2) return fail(c, PSGSDF_ERR_ARG, "debug_normals_cache: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = normals_cache(c, width, height); if (rc) return rc;
    HIPCHK(c, hipMemcpy(cache9, c->ncache, sizeof(float) * 9 * (size_t)width * height, hipMemcpyDeviceToHost));
    return PSGSDF_OK;
}

extern "C" int psgsdf_estimate_normals(psgsdf_ctx* c, const float* depth, int width, int height, float* normals_xyz) {
    if (!c || !depth || !normals_xyz || width < 2 || height < 2) return fail(c, PSGSDF_ERR_ARG, "estimate_normals: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = normals_cache(c, width, height); if (rc) return rc;
    const size_t n = (size_t)width * height;
    { void* hs = nullptr; int src = host_stage(c, sizeof(float) * n, &hs); if (src) return src; memcpy(hs, depth, sizeof(float) * n);
      HIPCHK(c, hipMemcpyAsync(c->ndepth, hs, sizeof(float) * n, hipMemcpyHostToDevice, c->stream)); }
    timed(c, "normals", [&] { launch_normals(c->ndepth, c->ncache, width, height, 5, c->ntmp, c->nout, c->stream); });
    HIPCHK(c, hipMemcpyAsync(normals_xyz, c->nout, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PSGSDF_OK;
}

// RigidPointOptimizer::optimize_sampled (RigidPointOptimizer.cpp:12-79), sampling = 1: frame-to-model tracking on the dense volume
extern "C" int psgsdf_track(psgsdf_ctx* c, const float* depth, int width, int height, float pose[16], float z_min, float z_max,
                            int num_iterations, float conv_threshold, float damping, int* iters_out, int* converged) {
    if (!c || !c->have_volume || !depth || !pose) return fail(c, PSGSDF_ERR_STATE, "track: volume first");
    if (c->n_ranks > 1 && !c->comm) return fail(c, PSGSDF_ERR_COMM, "rank %d of %d has no communicator (psgsdf_comm_init)", c->rank, c->n_ranks);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)width * height;
    int rc = normals_cache(c, width, height); if (rc) return rc;   // (allocates the depth staging buffer)
    const int nblk = 256;
    if (!c->track_part) { HIPCHK(c, hipMalloc(&c->track_part, sizeof(double) * nblk * 29)); HIPCHK(c, hipHostMalloc(&c->track_host, sizeof(double) * nblk * 29)); }
    { void* hs = nullptr; int src = host_stage(c, sizeof(float) * n, &hs); if (src) return src; memcpy(hs, depth, sizeof(float) * n);
      HIPCHK(c, hipMemcpyAsync(c->ndepth, hs, sizeof(float) * n, hipMemcpyHostToDevice, c->stream)); }
    Cam cam = c->cam; cam.W = width; cam.H = height;
    if (converged) *converged = 0;
    int k = 0;
    for (; k < num_iterations; ++k) {
        FrameP fp{};
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) fp.R[i * 3 + j] = pose[i * 4 + j]; fp.t[i] = pose[i * 4 + 3]; }
        // (z-slabs: every rank sums the pixels whose nearest voxel lies in a plane it OWNS; the 29 sums are all-reduced below, every rank solves the same 6x6)
        timed(c, "track", [&] { launch_track(c->dense, c->grid, cam, fp, c->ndepth, z_min, z_max, c->track_part, nblk, c->gdim[2], c->z0, c->z1, c->stream); });
        HIPCHK(c, hipMemcpyAsync(c->track_host, c->track_part, sizeof(double) * nblk * 29, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        double acc[29] = {0};
        for (int b = 0; b < nblk; ++b) for (int q = 0; q < 29; ++q) acc[q] += c->track_host[(size_t)b * 29 + q];
        if (c->n_ranks > 1) { std::vector<double> all(acc, acc + 29); int arc = host_allreduce(c, all, "tracker"); if (arc) return arc; for (int q = 0; q < 29; ++q) acc[q] = all[q]; }
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
