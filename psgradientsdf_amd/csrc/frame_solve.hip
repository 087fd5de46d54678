// frame_solve.hip -- the reference's OWN solver for the light and pose blocks (PSGSDF_FRAME_SOLVE=eigen, psgsdf_set_frame_solver): ONE
// Eigen::ConjugateGradient<SparseMatrix<float>> with its default Jacobi preconditioner over the whole block-diagonal system of all frames
// (PsOptimizer.cpp:175-203 light, :207-234 pose; LedOptimizer.cpp:134-160 light, :245-275 pose), tolerance eps_f32, at most 2n passes, x0 = 0.
// The blocks are only coupled through the scalars alpha and beta (dot products over ALL frames) and the common stop test -- which is why a frame's
// step differs from its own direct solve (sweeps.hip frame_solve_*: LDL^T in double, the engine's default) by cond(block) * eps_f32.
//
// One workgroup; thread t owns the unknowns t, t + T (n <= 2 * 1024): its row of the (symmetric, float) block and x, r, p, 1/diag live in
// registers, p also in LDS for the other rows of its block.  Float recurrences in Eigen's order with contraction OFF, the matrix-vector product and
// the three dot products of a pass accumulated in double and rounded to float (oracle/psgsdf_oracle.c eigen_cg: the restatement this kernel is
// compared with; the sums here are taken wavefront by wavefront, the oracle's front to back -- 1e-16 apart before the rounding to float).
#include "device_common.h"

namespace psg {

constexpr int kEigThreads = 1024, kEigK = 2, kEigWaves = kEigThreads / 64;

struct EigRed { double a[kEigWaves]; double b[2 * kEigWaves]; double c[kEigWaves]; };

__device__ __forceinline__ double eig_sum(double v, double* red) {      // all threads; `red` must not be the array of the previous call
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ void eig_sum2(double& v0, double& v1, double* red) {
    v0 = wave_sum(v0); v1 = wave_sum(v1);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) { red[2 * w] = v0; red[2 * w + 1] = v1; }
    __syncthreads();
    double s0 = 0, s1 = 0;
    for (int i = 0; i < nw; ++i) { s0 += red[2 * i]; s1 += red[2 * i + 1]; }
    v0 = s0; v1 = s1;
}

// Eigen/src/IterativeLinearSolvers/ConjugateGradient.h conjugate_gradient(), DiagonalPreconditioner; n unknowns in blocks of NB.
// Hrow[k][j]: entry (u_k, base(u_k) + j) of the matrix; bu[k]: right-hand side; xu[k]: the solution (out).  All threads of the workgroup call.
template <int NB>
__device__ __forceinline__ void eigen_cg_blockdiag(int n, const float (&Hrow)[kEigK][NB], const float (&bu)[kEigK], float (&xu)[kEigK], int max_it,
                                                   float* sp /*[kEigK * kEigThreads]*/, EigRed& red, int& iters_out, double& err_out, int& ok_out) {
#pragma clang fp contract(off)
    const int T = blockDim.x;
    bool act[kEigK]; int base[kEigK];
    float r[kEigK], p[kEigK], inv[kEigK], z[kEigK], tmp[kEigK];
    double s = 0;
#pragma unroll
    for (int k = 0; k < kEigK; ++k) {
        const int u = (int)threadIdx.x + k * T;
        act[k] = u < n; base[k] = (u / NB) * NB;
        float d = 1.0f;
#pragma unroll
        for (int j = 0; j < NB; ++j) if (base[k] + j == u) d = Hrow[k][j];
        inv[k] = d != 0.f ? 1.0f / d : 1.0f;
        r[k] = act[k] ? bu[k] : 0.f; xu[k] = 0.f; p[k] = 0.f; z[k] = 0.f; tmp[k] = 0.f;
        s += (double)r[k] * (double)r[k];
    }
    const float tol = FLT_EPSILON;
    const int maxIters = max_it > 0 ? max_it : 2 * n;
    const float rhsNorm2 = (float)eig_sum(s, red.a);
    if (rhsNorm2 == 0.f) { iters_out = 0; err_out = 0.0; ok_out = 1; return; }
    const float threshold = fmaxf(tol * tol * rhsNorm2, FLT_MIN);
    float residualNorm2 = rhsNorm2;      // residual = rhs - A * 0
    int i = 0;
    if (residualNorm2 >= threshold) {
        s = 0;
#pragma unroll
        for (int k = 0; k < kEigK; ++k) { p[k] = inv[k] * r[k]; s += (double)r[k] * (double)p[k]; }
        float absNew = (float)eig_sum(s, red.c);
        while (i < maxIters) {
#pragma unroll
            for (int k = 0; k < kEigK; ++k) if (act[k]) sp[threadIdx.x + k * T] = p[k];
            __syncthreads();
            s = 0;
#pragma unroll
            for (int k = 0; k < kEigK; ++k) {
                double m = 0;
                if (act[k]) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) m += (double)Hrow[k][j] * (double)sp[base[k] + j];
                }
                tmp[k] = (float)m;
                s += (double)p[k] * (double)tmp[k];
            }
            const float alpha = absNew / (float)eig_sum(s, red.a);
            double rr = 0, rz = 0;
#pragma unroll
            for (int k = 0; k < kEigK; ++k) {
                xu[k] += alpha * p[k]; r[k] -= alpha * tmp[k];
                rr += (double)r[k] * (double)r[k];
                z[k] = inv[k] * r[k];
                rz += (double)r[k] * (double)z[k];
            }
            eig_sum2(rr, rz, red.b);
            residualNorm2 = (float)rr;
            if (residualNorm2 < threshold) break;
            const float absOld = absNew;
            absNew = (float)rz;
            const float beta = absNew / absOld;
#pragma unroll
            for (int k = 0; k < kEigK; ++k) p[k] = z[k] + beta * p[k];
            i++;
        }
    }
    iters_out = i;
    err_out = sqrt((double)residualNorm2 / (double)rhsNorm2);
    ok_out = err_out <= (double)tol ? 1 : 0;
}

__device__ __forceinline__ int sym_idx(int NB, int a, int b) { if (a > b) { const int t = a; a = b; b = t; } return a * NB - (a * (a - 1)) / 2 + (b - a); }

// KIND 0: light (SH: F blocks of NB x NB, no damping, always applied -- PsOptimizer.cpp:175-203; LED: ONE diagonal 3 x 3 system over all frames,
// damped, always applied -- LedOptimizer.cpp:134-160).  KIND 1: pose (F blocks of 6 x 6, damped; SH always applied, LED only when info() == Success
// -- LedOptimizer.cpp:271-273, under ref_quirks).  KIND 2: a caller-supplied block-diagonal system (psgsdf_debug_frame_cg: the known-answer tests).
// stats[0..3] = {iterations, ||r|| / ||b||, info() == Success, update applied}
template <int NB, int KIND, bool LED>
__global__ void __launch_bounds__(kEigThreads) k_frames_eigen(SweepArgs a, FrameP* frames, float* led_light, double* e_out, unsigned long long e_key, float* undo,
                                                               double* stats, int nb, const float* rawH, const float* rawb, float* rawx, int max_it) {
    __shared__ float sp[kEigK * kEigThreads];
    __shared__ EigRed red;
    __shared__ float s_dl[3];
    const int T = blockDim.x, n = nb * NB;
    if (KIND == 0 && undo) {
        for (int i = threadIdx.x; i < a.F * 9; i += T) undo[i] = frames[i / 9].l[i % 9];
        if (threadIdx.x < 3) undo[a.F * 9 + threadIdx.x] = led_light[threadIdx.x];
    }
    float Hrow[kEigK][NB], bu[kEigK], xu[kEigK];
    constexpr int NH = (KIND == 0 && LED) ? 3 : NB * (NB + 1) / 2;
#pragma unroll
    for (int k = 0; k < kEigK; ++k) {
        const int u = (int)threadIdx.x + k * T;
        bu[k] = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) Hrow[k][j] = 0.f;
        if (u >= n) continue;
        const int f = u / NB, i = u - f * NB;
        if (KIND == 2) {
#pragma unroll
            for (int j = 0; j < NB; ++j) Hrow[k][j] = rawH[((size_t)f * NB + i) * NB + j];
            bu[k] = rawb[u];
        } else if (KIND == 0 && LED) {      // sums over the frames in frame order, then float, then the damping (k_solve_light's arithmetic)
            double hs = 0, bs = 0;
            for (int ff = 0; ff < a.F; ++ff) { hs += a.acc.frame[(size_t)ff * kFrameRow + i]; bs += a.acc.frame[(size_t)ff * kFrameRow + NH + i]; }
            float h = (float)hs;
            if (a.damping != 0.0f) h += a.damping * h;
#pragma unroll
            for (int j = 0; j < NB; ++j) if (j == i) Hrow[k][j] = h;
            bu[k] = (float)bs;
        } else {
            const double* row = a.acc.frame + (size_t)f * kFrameRow;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                float v = (float)row[sym_idx(NB, i, j)];
                if (KIND == 1 && j == i && a.damping != 0.0f) v += a.damping * v;
                Hrow[k][j] = v;
            }
            bu[k] = (float)row[NH + i];
        }
    }
    int iters, ok; double err;
    eigen_cg_blockdiag<NB>(n, Hrow, bu, xu, max_it, sp, red, iters, err, ok);
    int applied = 1;
    if (KIND == 1 && LED && a.quirks && !ok) applied = 0;      // LedOptimizer.cpp:271-273
    __syncthreads();                                           // (sp is reused below; every thread has left the solver's last matvec)
    if (KIND == 2) {
#pragma unroll
        for (int k = 0; k < kEigK; ++k) { const int u = (int)threadIdx.x + k * T; if (u < n) rawx[u] = xu[k]; }
    } else if (KIND == 0 && !LED) {
#pragma unroll
        for (int k = 0; k < kEigK; ++k) { const int u = (int)threadIdx.x + k * T; if (u < n) { const int f = u / NB, i = u - f * NB; frames[f].l[i] -= xu[k]; } }
    } else if (KIND == 0) {
        if (threadIdx.x < 3) s_dl[threadIdx.x] = xu[0];
        __syncthreads();
        for (int ff = threadIdx.x; ff < a.F; ff += T) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { const float nl = frames[ff].l[ch] - s_dl[ch]; frames[ff].l[ch] = nl; if (ff == 0) led_light[ch] = nl; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < kEigK; ++k) { const int u = (int)threadIdx.x + k * T; if (u < n) sp[u] = xu[k]; }
        __syncthreads();
        if (applied) for (int ff = threadIdx.x; ff < a.F; ff += T) { float xi[6]; for (int i = 0; i < 6; ++i) xi[i] = sp[ff * 6 + i]; pose_update(frames, ff, xi); }
    }
    if (threadIdx.x == 0 && stats) { stats[0] = (double)iters; stats[1] = err; stats[2] = (double)ok; stats[3] = (double)applied; }
    if (KIND != 2 && e_out) {      // the energy / n_obs columns of the rows over the frames (sweeps.hip frame_rows_finish's order: threads striding the frames, wavefront sums in order)
        constexpr int col_e = KIND == 1 ? 27 : NH + NB;
        __syncthreads();
        double e = 0, nn = 0;
        for (int f = threadIdx.x; f < a.F; f += T) { e += a.acc.frame[(size_t)f * kFrameRow + col_e]; nn += a.acc.frame[(size_t)f * kFrameRow + col_e + 1]; }
        eig_sum2(e, nn, red.b);
        if (threadIdx.x == 0) { mbox_put(e_out, 2, 0, e, e_key); mbox_put(e_out, 2, 1, nn, e_key); mbox_commit(e_key); }
    }
}

static int eig_threads(int n) { return std::min(kEigThreads, std::max(64, (n + 63) / 64 * 64)); }
bool frames_eigen_fits(int model, int F) { const int nbm = model == 1 ? 9 : 6; return (long long)F * nbm <= (long long)kEigK * kEigThreads; }

void launch_frames_eigen_light(const SweepArgs& a, FrameP* frames, float* led_light, double* e_out, unsigned long long e_key, float* undo, double* stats, hipStream_t s) {
    if (a.F <= 0) return;
    if (a.model == 0) hipLaunchKernelGGL((k_frames_eigen<4, 0, false>), dim3(1), dim3(eig_threads(a.F * 4)), 0, s, a, frames, led_light, e_out, e_key, undo, stats, a.F, nullptr, nullptr, nullptr, 0);
    else if (a.model == 1) hipLaunchKernelGGL((k_frames_eigen<9, 0, false>), dim3(1), dim3(eig_threads(a.F * 9)), 0, s, a, frames, led_light, e_out, e_key, undo, stats, a.F, nullptr, nullptr, nullptr, 0);
    else hipLaunchKernelGGL((k_frames_eigen<3, 0, true>), dim3(1), dim3(kBlock), 0, s, a, frames, led_light, e_out, e_key, undo, stats, 1, nullptr, nullptr, nullptr, 0);
}
void launch_frames_eigen_pose(const SweepArgs& a, FrameP* frames, double* e_out, unsigned long long e_key, double* stats, hipStream_t s) {
    if (a.F <= 0) return;
    if (a.model == 2) hipLaunchKernelGGL((k_frames_eigen<6, 1, true>), dim3(1), dim3(eig_threads(a.F * 6)), 0, s, a, frames, nullptr, e_out, e_key, nullptr, stats, a.F, nullptr, nullptr, nullptr, 0);
    else hipLaunchKernelGGL((k_frames_eigen<6, 1, false>), dim3(1), dim3(eig_threads(a.F * 6)), 0, s, a, frames, nullptr, e_out, e_key, nullptr, stats, a.F, nullptr, nullptr, nullptr, 0);
}
// the solver alone on nb blocks of n x n floats (row-major, used as given): n in {3, 4, 6, 9}
int launch_frames_eigen_raw(int nb, int n, const float* H, const float* b, float* x, double* stats, int max_it, hipStream_t s) {
    SweepArgs a{};
    const dim3 g(1), bl(eig_threads(nb * n));
    if ((long long)nb * n > (long long)kEigK * kEigThreads) return -1;
    if (n == 3) hipLaunchKernelGGL((k_frames_eigen<3, 2, false>), g, bl, 0, s, a, nullptr, nullptr, nullptr, 0ull, nullptr, stats, nb, H, b, x, max_it);
    else if (n == 4) hipLaunchKernelGGL((k_frames_eigen<4, 2, false>), g, bl, 0, s, a, nullptr, nullptr, nullptr, 0ull, nullptr, stats, nb, H, b, x, max_it);
    else if (n == 6) hipLaunchKernelGGL((k_frames_eigen<6, 2, false>), g, bl, 0, s, a, nullptr, nullptr, nullptr, 0ull, nullptr, stats, nb, H, b, x, max_it);
    else if (n == 9) hipLaunchKernelGGL((k_frames_eigen<9, 2, false>), g, bl, 0, s, a, nullptr, nullptr, nullptr, 0ull, nullptr, stats, nb, H, b, x, max_it);
    else return -1;
    return 0;
}

}  // namespace psg
