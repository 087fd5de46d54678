// dist.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Gradient-SDF photometric-stereo hot path:
// the distance block: per-voxel normal-equation blocks and their ELL assembly.  No CUDA compatibility layer, no other back end.
// Shared device helpers: device_common.h; the launchers are declared in engine.h.
#include "device_common.h"

#ifndef PSG_DIST_PK
#define PSG_DIST_PK 1      // (development: 0 = the scalar form of the SH models' Jacobian rows and block accumulation in k_sweep_dist)
#endif

namespace psg {

// ------------------------------------------------------------------------------------------
// distance block: per-voxel 4x4 normal-equation blocks, ELL assembly, Jacobi-PCG
// ------------------------------------------------------------------------------------------
// Optimizer.cpp:269-284 normalJacobian(grad, direction, lag=false)
__device__ __forceinline__ void normal_jacobian(float vs_inv, const float* grad, const float* direction, float* J) {
    float n_d[3] = {-vs_inv * direction[0], -vs_inv * direction[1], -vs_inv * direction[2]};
    float N_inv = (float)(1.0 / (double)fmaxf(norm3(grad), 0.001f));
    double Nd = (double)N_inv;
    float dN = (float)((Nd * Nd * Nd) * (double)dot3(n_d, grad));
#pragma unroll
    for (int k = 0; k < 3; ++k) J[k] = N_inv * n_d[k] - dN * grad[k];
}

// distJacobian per observation PsOptimizerJa.cpp:160-289 / LedOptimizerJa.cpp:117-218, accumulated directly
// into the per-voxel block over {self, x-, y-, z-stencil neighbour}; regularisers Optimizer.cpp:196-218,477-590.
template <int MODEL, int LOSS, int IMG>
__global__ void __launch_bounds__(kBlock, 4) k_sweep_dist(SweepArgs a) {      // 4 waves per SIMD (<= 128 VGPRs): without the hint SH2 takes 132 and LED 130
    constexpr int NB = ModelTraits<MODEL>::NB;
    constexpr bool LED = ModelTraits<MODEL>::LED;
    FrameP* sf = reinterpret_cast<FrameP*>(psg_dyn_smem);   // F records, dynamic LDS
    __shared__ double red[kBlock / 64];
    load_frames(sf, a.frames, a.F);
    const Band& b = a.b;
    const int bid = vm_bid(a, 4);
    int j = a.row0 + bid * blockDim.x + threadIdx.x;
    double E = 0, nobs = 0;
    if (j < a.row1) {
        Vox v; load_vox(b, j, v);
        const float vs_inv = a.grid.vs_inv;
        float grad[3] = {b.gfd[0][j], b.gfd[1][j], b.gfd[2][j]};
        float dir[3]; bool exists[4]; exists[0] = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bool fwd = b.nb[(size_t)(2 * k) * b.Spad + j] >= 0;
            dir[k] = fwd ? 1.0f : -1.0f;
            exists[k + 1] = fwd ? true : (b.nb[(size_t)(2 * k + 1) * b.Spad + j] >= 0);
        }
        float dn[4][3];
        normal_jacobian(vs_inv, grad, dir, dn[0]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float nd[3] = {0.f, 0.f, 0.f};
            if (LED && a.quirks) nd[k] += dir[k]; else nd[k] -= dir[k];   // B6: LedOptimizerJa.cpp:157-167 vs PsOptimizerJa.cpp:200-210
            normal_jacobian(vs_inv, grad, nd, dn[k + 1]);
        }
        const float d = b.dist[j];
        float dx[4][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) dx[0][k] = -v.gn[k] - d * dn[0][k];
#pragma unroll
        for (int q = 1; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 3; ++k) dx[q][k] = -d * dn[q][k];
        float shfd[kMaxBasis];
        if (!LED) SH<NB == 3 ? 4 : NB>(v.nfd, shfd);
        obs_acc_t B[10], g[4];   // <= F terms each: float accumulation (oracle: double) differs ~1e-7 relative (PSG_STRICT & 8: double)
        obs_acc_t Ef = 0; int nobs_i = 0;
#pragma unroll
        for (int i = 0; i < 10; ++i) B[i] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = 0;
        // SH models, packed form (round 6): the four stencil slots travel as two pairs -- the rows' three basis vectors (S, T, Z below), the Jacobian rows and the
        // block's accumulators are 2-vectors, so the multiply-adds issue as v_pk_fma_f32 (two slots per instruction, the scalar factor broadcast).  The same
        // products and sums as the scalar form, entry for entry.
        constexpr bool kPk = PSG_DIST_PK && !(PSG_STRICT & 24);
        f2_t dxp[3][2], dnp[3][2];      // [component][slot pair]: {slot 2h, slot 2h + 1}
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) { dxp[k][h] = f2_t{dx[2 * h][k], dx[2 * h + 1][k]}; dnp[k][h] = f2_t{dn[2 * h][k], dn[2 * h + 1][k]}; }
        f2_t B01 = {0.f, 0.f}, B23 = {0.f, 0.f}, B1x = {0.f, 0.f}, B2x = {0.f, 0.f}, g01 = {0.f, 0.f}, g23 = {0.f, 0.f}; float B11 = 0.f, B33 = 0.f;      // B(0,0..1), B(0,2..3), B(1,2..3), B(2,2..3); B(1,1), B(3,3)
        FOR_EACH_VISIBLE_FRAME(b, j, a.F, f) {
            const FrameP& fp = frame_at(sf, f);
            Proj pr = project(v.xs, fp, a.cam);
            if (!pr.ok) continue;
            float I[3], gu[3], gv[3], ren[3];
            const ProjJ pj = project_jac(pr, a.cam);      // distJacobian projects again, with its own in-image test (PsOptimizerJa.cpp:180-190)
            sample<true, IMG>(a.im, f, a.cam, pr.m, pr.n, pj.mj, pj.nj, I, gu, gv);
            rendered<MODEL>(fp, pr, v.nfd, shfd, v.rho, ren);
            bool done_pk = false;
            if constexpr (kPk) {
                done_pk = true;
                float U[3], V[3]; pi_rows_world(pi_rows(a.cam, pr), fp.R, U, V);
                float lc[3] = {fp.l[1], fp.l[2], fp.l[3]};      // l . dSH/dn (SH2: regrouped as in the scalar form below)
                if (!LED && NB == 9) {
                    const float* nh = v.nfd;
                    lc[0] = (fp.l[1] + fp.l[4] * nh[1]) + (fp.l[5] * nh[2] + 2 * nh[0] * (fp.l[7] + fp.l[8]));
                    lc[1] = (fp.l[2] + fp.l[4] * nh[0]) + (fp.l[6] * nh[2] - 2 * nh[1] * fp.l[7]);
                    lc[2] = (fp.l[3] + fp.l[5] * nh[0]) + (fp.l[6] * nh[1] - 2 * nh[2] * fp.l[8]);
                }
                f2_t S2[2], T2[2], Z2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    S2[h] = (U[0] * dxp[0][h] + U[1] * dxp[1][h]) + U[2] * dxp[2][h];
                    T2[h] = (V[0] * dxp[0][h] + V[1] * dxp[1][h]) + V[2] * dxp[2][h];
                    if (!LED) Z2[h] = (lc[0] * dnp[0][h] + lc[1] * dnp[1][h]) + lc[2] * dnp[2][h];
                }
                if (LED) {      // the fall-off term dm_q of the scalar form below (LedOptimizerJa.cpp:157-186), two slots at a time; Z = -dm so that J_c = gu_c S + gv_c T - (rho_c l_c) Z
                    float Rp[3]; mul3(fp.R, pr.p, Rp);
                    const float pn = norm3(pr.p); const double pd = (double)pn;
                    const float radius = (float)(pd * pd * pd), p5 = (float)(pd * pd * pd * pd * pd);
                    const float nRp = dot3(v.nfd, Rp);
                    const float y5 = 1.0f / p5, y3 = 1.0f / radius;
                    auto div2 = [](f2_t a2, float b, float y) { const f2_t q0 = a2 * y; return __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, f2_t{b, b}, a2), f2_t{y, y}, q0); };      // div_by on a pair: the same bits
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f2_t d1 = (dnp[0][h] * Rp[0] + dnp[1][h] * Rp[1]) + dnp[2][h] * Rp[2];
                        const f2_t d2 = (v.nfd[0] * dxp[0][h] + v.nfd[1] * dxp[1][h]) + v.nfd[2] * dxp[2][h];
                        const f2_t d3 = (Rp[0] * dxp[0][h] + Rp[1] * dxp[1][h]) + Rp[2] * dxp[2][h];
                        const f2_t dm2 = div2(-3 * d3, p5, y5);
                        Z2[h] = -(div2(d1 + d2, radius, y3) + dm2 * nRp);
                    }
                }
                float l = 0.f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    f2_t J0 = gu[ch] * S2[0] + gv[ch] * T2[0], J1 = gu[ch] * S2[1] + gv[ch] * T2[1];
                    const float zc = LED ? v.rho[ch] * fp.l[ch] : v.rho[ch];
                    J0 = J0 - zc * Z2[0]; J1 = J1 - zc * Z2[1];
                    const float r = I[ch] - ren[ch]; float w = robust_weight<LOSS>(a.rob, r);
                    l += robust_loss<LOSS>(a.rob, r);
                    w = pj.ok ? w : 0.f;                      // the residual counts for the energy, but the row has no Jacobian
                    const f2_t jw0 = J0 * w, jw1 = J1 * w;
                    B01 += jw0.x * J0; B23 += jw0.x * J1; B1x += jw0.y * J1; B2x += jw1.x * J1;
                    B11 += jw0.y * J0.y; B33 += jw1.y * J1.y;
                    g01 += jw0 * r; g23 += jw1 * r;
                }
                Ef += (obs_acc_t)l; nobs_i += 1;
            }
            if (!done_pk) {
            float J[4][3];
#if PSG_STRICT & 16
            {   // the reference's order (PsOptimizerJa.cpp:78-100,212-289 / LedOptimizerJa.cpp:117-218; oracle dist_jacobian): G = image_grad(3x2) pi_grad(2x3) with its structural
                // zeros, then G R^T, then (G R^T) dx_q; the shading terms per channel and stencil slot
                const PiRows pi = pi_rows(a.cam, pr);
                float G[9], GRt[9];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) { G[ch * 3 + 0] = gu[ch] * pi.p00 + gv[ch] * 0.0f; G[ch * 3 + 1] = gu[ch] * 0.0f + gv[ch] * pi.p11; G[ch * 3 + 2] = gu[ch] * pi.p02 + gv[ch] * pi.p12; }
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                    for (int k = 0; k < 3; ++k) GRt[ch * 3 + k] = (G[ch * 3 + 0] * fp.R[k * 3 + 0] + G[ch * 3 + 1] * fp.R[k * 3 + 1]) + G[ch * 3 + 2] * fp.R[k * 3 + 2];
                float dI[4][3];
#pragma unroll
                for (int q = 0; q < 4; ++q) mul3(GRt, dx[q], dI[q]);
                if (!LED && NB != 9) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) { const float dr[3] = {v.rho[ch] * fp.l[1], v.rho[ch] * fp.l[2], v.rho[ch] * fp.l[3]}; J[q][ch] = dI[q][ch] - dot3(dr, dn[q]); }
                } else if (!LED) {
                    const float* nh = v.nfd;
                    const float D[3][9] = {{0, 1, 0, 0, nh[1], nh[2], 0, 2 * nh[0], 2 * nh[0]}, {0, 0, 1, 0, nh[0], 0, nh[2], -2 * nh[1], 0}, {0, 0, 0, 1, 0, nh[0], nh[1], 0, -2 * nh[2]}};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float dsh[9];
#pragma unroll
                        for (int i = 0; i < 9; ++i) dsh[i] = (D[0][i] * dn[q][0] + D[1][i] * dn[q][1]) + D[2][i] * dn[q][2];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) { float sacc = 0; for (int i = 0; i < 9; ++i) sacc += (v.rho[ch] * fp.l[i]) * dsh[i]; J[q][ch] = dI[q][ch] - sacc; }
                    }
                } else {
                    float Rp[3]; mul3(fp.R, pr.p, Rp);
                    const float pn = norm3(pr.p); const double pd = (double)pn;
                    const float radius = (float)(pd * pd * pd), p5 = (float)(pd * pd * pd * pd * pd);
                    const float nRp = dot3(v.nfd, Rp);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float dm = dot3(dn[q], Rp) + dot3(v.nfd, dx[q]);
                        float tmp3[3]; mulT3(fp.R, dx[q], tmp3);
                        const float dm2 = -3 * dot3(pr.p, tmp3) / p5;
                        dm = dm / radius + dm2 * nRp;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) J[q][ch] = dI[q][ch] + (v.rho[ch] * fp.l[ch]) * dm;
                    }
                }
            }
#else
            float U[3], V[3]; pi_rows_world(pi_rows(a.cam, pr), fp.R, U, V);
#pragma unroll
            for (int q = 0; q < 4; ++q) {                         // dI_q = image_grad pi_grad R^T dx_q, contracted from the right (device_common.h)
                const float sq = dot3(U, dx[q]), tq = dot3(V, dx[q]);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) J[q][ch] = gu[ch] * sq + gv[ch] * tq;
            }
            if (!LED) {
                // shading term rho_c * (l . dSH/dd_q): the frame's light is contracted with the (per-voxel) normal derivative ONCE per
                // stencil slot and then scaled by the three albedos, instead of forming rho_c * l per channel first (the reference's
                // order, PsOptimizerJa.cpp:225-278; same value up to the rounding of one product, a third of the instructions)
                // SH2: l . (dSH/dn dn_q) regrouped as (l . dSH/dn) . dn_q -- the frame's light meets the 3 x 9 derivative table once per observation (14
                // non-zero entries) instead of once per stencil slot; same value up to the rounding of the regrouped sums (engine deviation 7)
                float lw[3] = {0.f, 0.f, 0.f};
                if (NB == 9) {
                    const float* nh = v.nfd;
                    lw[0] = (fp.l[1] + fp.l[4] * nh[1]) + (fp.l[5] * nh[2] + 2 * nh[0] * (fp.l[7] + fp.l[8]));
                    lw[1] = (fp.l[2] + fp.l[4] * nh[0]) + (fp.l[6] * nh[2] - 2 * nh[1] * fp.l[7]);
                    lw[2] = (fp.l[3] + fp.l[5] * nh[0]) + (fp.l[6] * nh[1] - 2 * nh[2] * fp.l[8]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float sq;
                    if (NB == 4) sq = (fp.l[1] * dn[q][0] + fp.l[2] * dn[q][1]) + fp.l[3] * dn[q][2];
                    else sq = (lw[0] * dn[q][0] + lw[1] * dn[q][1]) + lw[2] * dn[q][2];
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) J[q][ch] = J[q][ch] - v.rho[ch] * sq;
                }
            } else {
                float Rp[3]; mul3(fp.R, pr.p, Rp);
                float pn = norm3(pr.p); double pd = (double)pn;
                float radius = (float)(pd * pd * pd);
                float p5 = (float)(pd * pd * pd * pd * pd);
                float nRp = dot3(v.nfd, Rp);
                const float y5 = 1.0f / p5, y3 = 1.0f / radius;      // eight quotients, two divisors (device_common.h div_by: same bits)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float dm = dot3(dn[q], Rp) + dot3(v.nfd, dx[q]);
                    // p . (R^T dx_q) regrouped as (R p) . dx_q -- R p is there already (LedOptimizerJa.cpp:176-186 forms R^T dx_q first; same value up
                    // to the rounding of the regrouped products, 9 multiply-adds less per stencil slot: engine deviation 7)
                    float dm2 = div_by(-3 * dot3(Rp, dx[q]), p5, y5);
                    dm = div_by(dm, radius, y3) + dm2 * nRp;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) J[q][ch] = J[q][ch] + (v.rho[ch] * fp.l[ch]) * dm;
                }
            }
#endif
            float l = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float r = I[ch] - ren[ch]; float w = robust_weight<LOSS>(a.rob, r);
                l += robust_loss<LOSS>(a.rob, r);
                w = pj.ok ? w : 0.f;                      // the residual counts for the energy, but the row has no Jacobian
                int q = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float jw = J[p][ch] * w;
#pragma unroll
                    for (int k = p; k < 4; ++k) B[q++] += (obs_acc_t)(jw * J[k][ch]);
                    g[p] += (obs_acc_t)(jw * r);
                }
            }
            Ef += (obs_acc_t)l; nobs_i += 1;
            }
        }
        E = (double)Ef; nobs = (double)nobs_i;
        if (kPk) {      // (B / g were zero: exact copies)
            B[0] += (obs_acc_t)B01.x; B[1] += (obs_acc_t)B01.y; B[2] += (obs_acc_t)B23.x; B[3] += (obs_acc_t)B23.y; B[4] += (obs_acc_t)B11; B[5] += (obs_acc_t)B1x.x; B[6] += (obs_acc_t)B1x.y;
            B[7] += (obs_acc_t)B2x.x; B[8] += (obs_acc_t)B2x.y; B[9] += (obs_acc_t)B33; g[0] += (obs_acc_t)g01.x; g[1] += (obs_acc_t)g01.y; g[2] += (obs_acc_t)g23.x; g[3] += (obs_acc_t)g23.y;
        }
        if (a.normal_reg) {   // Eikonal row, Optimizer.cpp:196-218 + residual :509
            float n_d[3] = {-vs_inv * dir[0], -vs_inv * dir[1], -vs_inv * dir[2]};
            float Jr[4];
            Jr[0] = dot3(grad, n_d);
#pragma unroll
            for (int k = 0; k < 3; ++k) Jr[k + 1] = grad[k] * (vs_inv * dir[k]);
            float gnrm = norm3(grad);
            if (gnrm > 0.0f) {
#pragma unroll
                for (int k = 0; k < 4; ++k) Jr[k] /= gnrm;
            }
            float res = gnrm - 1;
            int q = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int k = p; k < 4; ++k) B[q++] += (obs_acc_t)(a.reg_n * (Jr[p] * Jr[k]));
                g[p] += (obs_acc_t)(a.reg_n * (Jr[p] * res));
            }
        }
        if (a.laplacian_reg) {   // diagonal only (reference drops the off-diagonals), Optimizer.cpp:540-590
            float vs2 = vs_inv * vs_inv; float Jl = -6 * vs2; float res = laplacian(b, j, vs_inv);
            B[0] += (obs_acc_t)(a.reg_l * (Jl * Jl)); g[0] += (obs_acc_t)(a.reg_l * (Jl * res));
        }
        // columns whose stencil neighbour is outside the band are dropped (PsOptimizerJa.cpp:536-552)
        int q = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int k = p; k < 4; ++k) { b.blk[(size_t)q * b.Spad + j] = (exists[p] && exists[k]) ? (float)B[q] : 0.f; ++q; }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) b.blk[(size_t)(10 + p) * b.Spad + j] = exists[p] ? (float)g[p] : 0.f;
    }
    block_part_store(E, PART(a, SC_ENERGY), red, bid);
    block_part_store(nobs, PART(a, SC_NOBS), red, bid);
    if (a.pcg_asm && a.pcg_gran) {   // the persistent solve right behind this sweep assembles the system itself: no tag of an earlier solve may survive
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.pcg_fs[1] = 0.0; a.pcg_fs[2] = 0.0; a.pcg_fs[3] = 0.0; }
        const int gid = blockIdx.x * blockDim.x + threadIdx.x;
        for (int q = gid; q < a.pcg_gran_n; q += gridDim.x * blockDim.x) a.pcg_gran[q] = 0.0;
    }
}
void launch_sweep_dist(const SweepArgs& a, hipStream_t s) {
    if (a.row1 <= a.row0) return;
    dim3 g((a.row1 - a.row0 + kBlock - 1) / kBlock), bl(kBlock);
    PSG_LAUNCH_SWEEP(k_sweep_dist, a, false, g, bl, a.F * sizeof(FrameP), s, a);
}

// one ELL row (and, if a.pcg_fuse_init, the PCG initialisation of that row: returns r_0^2)
__device__ __forceinline__ double assemble_row(const SweepArgs& a, int i, double (*acc)[kBlock]) {
    const Band& b = a.b;
    const int tid = threadIdx.x;
    double bb = 0.0;
    // round 1: the six axis neighbours (contributor 0 is the row itself; 1,2 = x lower / upper; 3,4 = y; 5,6 = z)
    int jr[7]; jr[0] = i;
#pragma unroll
    for (int c = 1; c < 7; ++c) { const int ax = (c - 1) >> 1; const bool upper = (c - 1) & 1; jr[c] = b.nb[(size_t)(2 * ax + (upper ? 0 : 1)) * b.Spad + i]; }
    // round 2: their stencil directions
    int db[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) db[c] = b.dirb[jr[c] >= 0 ? jr[c] : i];
    bool use[7]; use[0] = true;
#pragma unroll
    for (int c = 1; c < 7; ++c) { const int ax = (c - 1) >> 1; const bool upper = (c - 1) & 1; use[c] = jr[c] >= 0 && !(upper && ((db[c] >> ax) & 1)); }   // an upper neighbour with a forward stencil does not touch row i
    // round 3: their block entries (row s of the symmetric 4x4 block + rhs entry s)
    float val[7][4], gr[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const int s = c == 0 ? 0 : ((c - 1) >> 1) + 1;
        const int jrow = use[c] ? jr[c] : i;
        gr[c] = b.blk[(size_t)(10 + s) * b.Spad + jrow];
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) val[c][bq] = b.blk[(size_t)sym4(s, bq) * b.Spad + jrow];
    }
    double rhs = 0.0;
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        if (!use[c]) continue;
        const int ax = c == 0 ? 0 : (c - 1) >> 1; const bool upper = c > 0 && ((c - 1) & 1);
        int coff[3] = {0, 0, 0};
        if (c > 0) coff[ax] = upper ? 1 : -1;
        rhs += (double)gr[c];
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            int o[3] = {coff[0], coff[1], coff[2]};
            if (bq > 0) o[bq - 1] += ((db[c] >> (bq - 1)) & 1) ? 1 : -1;
            if (val[c][bq] != 0.f) acc[q_of(o)][tid] += (double)val[c][bq];
        }
    }
    int extra = 0;
#pragma unroll
    for (int q = 0; q < kNQ; ++q) { float h = (float)acc[q][tid]; b.H[(size_t)q * b.Spad + i] = h; if (q >= kNQCommon && h != 0.f) extra = 1; }
    b.hx[i] = extra;
    b.rhs[i] = (float)rhs;
    if (a.pcg_fuse_init) {   // k_cgf_init's work for this row (pcg.hip)
        float dg = (float)acc[0][tid];
        if (a.damping != 0.0f) dg += a.damping * dg;
        const float inv = dg != 0.f ? 1.0f / dg : 1.0f;
        const float r = (float)rhs;
        b.x[i] = 0.f;
        b.rec[1][i] = make_float4(r, 0.f, 0.f, inv);
        bb = (double)r * (double)r;
    }
    return bb;
}
// H = sum_j P_j^T B_j P_j assembled row-wise into 19 fixed column offsets (ELL); a row receives
// slices from itself, from each lower neighbour (whose forward stencil points at it) and from each
// upper neighbour whose stencil was forced backward.  Accumulation in LDS (dynamic column index).
// The loads are arranged in three rounds for ALL seven possible contributors at once (neighbour rows -> their stencil
// direction bits -> their block entries); a loop over the contributors made 7 x 3 dependent round trips (30 us).
__global__ void __launch_bounds__(kBlock) k_assemble(SweepArgs a) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    __shared__ double acc[kNQ][kBlock];
    int i = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < kNQ; ++q) acc[q][tid] = 0.0;
    const double bb = i < a.row1 ? assemble_row(a, i, acc) : 0.0;
    if (a.pcg_fuse_init) {
        __shared__ double red[kBlock / 64];
        block_part_store(bb, fpart(a.pcg_part, -1, 6), red);
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.pcg_fs[1] = 0.0; a.pcg_fs[2] = 0.0; a.pcg_fs[3] = 0.0; }
        if (a.pcg_gran) {   // persistent solve: no tag of an earlier solve may survive
            const int gid = blockIdx.x * blockDim.x + threadIdx.x;
            for (int q = gid; q < a.pcg_gran_n; q += gridDim.x * blockDim.x) a.pcg_gran[q] = 0.0;
        }
    }
}
void launch_assemble(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_assemble, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}

}  // namespace psg
