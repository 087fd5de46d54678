// albedo_reg.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Gradient-SDF photometric-stereo hot path:
// the "reg albedo" term (matrix-free regularised albedo solve).  No CUDA compatibility layer, no other back end.
// Shared device helpers: device_common.h; the launchers are declared in engine.h.
#include "device_common.h"

namespace psg {

// ------------------------------------------------------------------------------------------
// "reg albedo": E_r = sum_v sum_c ||grad rho_c(v)||, Gauss-Newton term reg_rho Jr^T Jr in the albedo system
// (Optimizer.cpp:122-136 energy, :396-460 computeAlbedoGrad, :221-245 per-voxel Jacobian, :593-647 sparse Jr).
// No shipped configuration enables it, so this path is written for clarity, not speed: the 3S x 3S system is never
// assembled; Jr (<= 4 entries per row: the voxel and its three stencil neighbours) is applied matrix-free.
// Multi-rank: every kernel but the table builder covers the slab's OWN rows [row0, row1); what a row reads of a neighbouring row
// (albedo, `back`, J, res, the CG vectors p and t) is exchanged for the halo rows by the host (loop.hip, comm_halo).
//   unknown index = (row, channel), stored channel-major: plane[ch * Spad + row]
//   quirk (ref_quirks): the blue self-entry of Jr sits in the GREEN column of the same voxel (Optimizer.cpp:617)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_areg_tables(DenseView d, GridP grid, SweepArgs a) {
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= b.S) return;
    const long long lin = b.lin[j];
    const long long stride[3] = {1, grid.dim[0], (long long)grid.dim[0] * grid.dim[1]};
    int back = 0;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const int fwd = b.nb[(size_t)(2 * ax) * b.Spad + j];
        int row; long long ln;
        if (fwd >= 0) { row = fwd; ln = lin + stride[ax]; }
        else { back |= 1 << ax; row = b.nb[(size_t)(2 * ax + 1) * b.Spad + j]; ln = lin - stride[ax]; }
        if (ln < 0 || ln >= grid.nvox) { ln = lin; row = j; }       // reference reads out of bounds (UB): zero difference
        ar.anb[(size_t)ax * b.Spad + j] = row;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) ar.anrho[(size_t)(ax * 3 + ch) * b.Spad + j] = row >= 0 ? 0.f : d.rho[ch][ln];   // static: only band voxels change
    }
    ar.back[j] = back;
}
void launch_areg_tables(const DenseView& d, const GridP& g, const SweepArgs& a, hipStream_t s) {
    if (a.b.S > 0) hipLaunchKernelGGL(k_areg_tables, dim3((a.b.S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d, g, a);
}
__global__ void __launch_bounds__(kBlock) k_areg_build(SweepArgs a) {
#pragma clang fp contract(off)
    __shared__ double red[kBlock / 64];
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0;
    if (j < a.row1) {
        const float vs_inv = a.grid.vs_inv;
        const int back = ar.back[j];
        float dir[3], G[3][3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            dir[ax] = (back >> ax) & 1 ? -1.0f : 1.0f;
            const int row = ar.anb[(size_t)ax * b.Spad + j];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float rn = row >= 0 ? b.rho[ch][row] : ar.anrho[(size_t)(ax * 3 + ch) * b.Spad + j];
                G[ch][ax] = (dir[ax] * (rn - b.rho[ch][j])) * vs_inv;
            }
        }
        const float r_d[3] = {-vs_inv * dir[0], -vs_inv * dir[1], -vs_inv * dir[2]};
        float esum = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float gn = norm3(G[ch]);
            float J[4];
            J[0] = (G[ch][0] * r_d[0] + G[ch][1] * r_d[1]) + G[ch][2] * r_d[2];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) J[ax + 1] = G[ch][ax] * (vs_inv * dir[ax]);
            if (gn != 0.0f) {
#pragma unroll
                for (int q = 0; q < 4; ++q) J[q] /= gn;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) ar.J[(size_t)(q * 3 + ch) * b.Spad + j] = J[q];
            ar.res[(size_t)ch * b.Spad + j] = gn;
            esum += gn;
        }
        e = (double)esum;
    }
    block_part_store(e, PART(a, SC_AUX0), red);
}
void launch_areg_build(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_areg_build, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}
// (Jr v)(row, ch) for a vector v over the unknowns
__device__ __forceinline__ float areg_jrow(const SweepArgs& a, const float* v, int j, int ch) {
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    const int c0 = (ch == 2 && a.quirks) ? 1 : ch;
    float t = ar.J[(size_t)ch * b.Spad + j] * v[(size_t)c0 * b.Spad + j];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const int row = ar.anb[(size_t)ax * b.Spad + j];
        if (row >= 0 && row != j) t += ar.J[(size_t)((ax + 1) * 3 + ch) * b.Spad + j] * v[(size_t)ch * b.Spad + row];
    }
    return t;
}
// (Jr^T t)(w, ch): rows of Jr with an entry in column (w, ch) = the voxel's own rows (self slot) and the rows of the axis
// neighbours whose stencil points at w.  SQ: the same sum with squared entries and t = 1 (diagonal of Jr^T Jr).
template <bool SQ>
__device__ __forceinline__ float areg_jtcol(const SweepArgs& a, const float* t, int w, int ch) {
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {   // self slots that land in column ch
        const int c0 = (c == 2 && a.quirks) ? 1 : c;
        if (c0 != ch) continue;
        const float Jv = ar.J[(size_t)c * b.Spad + w];
        s += SQ ? Jv * Jv : Jv * t[(size_t)c * b.Spad + w];
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const int vm = b.nb[(size_t)(2 * ax + 1) * b.Spad + w];          // w - e_ax: points at w iff it uses its forward neighbour
        if (vm >= 0 && !((ar.back[vm] >> ax) & 1)) { const float Jv = ar.J[(size_t)((ax + 1) * 3 + ch) * b.Spad + vm]; s += SQ ? Jv * Jv : Jv * t[(size_t)ch * b.Spad + vm]; }
        const int vp = b.nb[(size_t)(2 * ax) * b.Spad + w];              // w + e_ax: points at w iff it uses its backward neighbour
        if (vp >= 0 && ((ar.back[vp] >> ax) & 1)) { const float Jv = ar.J[(size_t)((ax + 1) * 3 + ch) * b.Spad + vp]; s += SQ ? Jv * Jv : Jv * t[(size_t)ch * b.Spad + vp]; }
    }
    return s;
}
// rhs = b_d + weight Jr^T res ; diag = (1 + damping) (H_d + weight diag(Jr^T Jr))     (PsOptimizer.cpp:95-105)
__global__ void __launch_bounds__(kBlock) k_areg_system(SweepArgs a) {
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.row1) return;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const size_t u = (size_t)ch * b.Spad + j;
        ar.rhs[u] = b.ab[u] + ar.weight * areg_jtcol<false>(a, ar.res, j, ch);
        float dg = b.aH[u] + ar.weight * areg_jtcol<true>(a, nullptr, j, ch);
        ar.diag0[u] = dg;
        if (a.damping != 0.0f) dg += a.damping * dg;
        ar.diag[u] = dg;
    }
}
void launch_areg_system(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_areg_system, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}
__global__ void __launch_bounds__(kBlock) k_areg_jx(SweepArgs a, const float* __restrict__ p, float* __restrict__ t) {
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.row1) return;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) t[(size_t)ch * a.b.Spad + j] = areg_jrow(a, p, j, ch);
}
void launch_areg_jx(const SweepArgs& a, const float* p, float* t, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_areg_jx, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, p, t);
}
// q = H p with H = H_d + weight Jr^T Jr and H.diagonal() += damping * H.diagonal(); partial p.q
__global__ void __launch_bounds__(kBlock) k_areg_jt(SweepArgs a, const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ q) {
    __shared__ double red[kBlock / 64];
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double pq = 0;
    if (j < a.row1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const size_t u = (size_t)ch * b.Spad + j;
            float v = b.aH[u] * p[u] + ar.weight * areg_jtcol<false>(a, t, j, ch);
            if (a.damping != 0.0f) v += (a.damping * ar.diag0[u]) * p[u];   // H.diagonal() += damping * H.diagonal()
            q[u] = v; pq += (double)p[u] * (double)v;
        }
    }
    block_part_store(pq, PART(a, SC_AUX0), red);
}
void launch_areg_jt(const SweepArgs& a, const float* p, const float* t, float* q, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_areg_jt, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, p, t, q);
}
__global__ void __launch_bounds__(kBlock) k_areg_cg_init(SweepArgs a) {
    __shared__ double red[kBlock / 64];
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double bb = 0, rz = 0;
    if (j < a.row1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const size_t u = (size_t)ch * b.Spad + j;
            const float r = ar.rhs[u], dg = ar.diag[u];
            const float z = (dg != 0.f ? 1.0f / dg : 1.0f) * r;
            ar.x[u] = 0.f; ar.r[u] = r; ar.p[u] = z;
            bb += (double)r * (double)r; rz += (double)r * (double)z;
        }
    }
    block_part_store(bb, PART(a, SC_AUX0), red);
    block_part_store(rz, PART(a, SC_AUX1), red);
}
void launch_areg_cg_init(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_areg_cg_init, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}
__global__ void __launch_bounds__(kBlock) k_areg_cg_update(SweepArgs a, float alpha) {
    __shared__ double red[kBlock / 64];
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double rr = 0, rz = 0;
    if (j < a.row1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const size_t u = (size_t)ch * b.Spad + j;
            ar.x[u] += alpha * ar.p[u];
            const float r = ar.r[u] - alpha * ar.q[u];
            ar.r[u] = r;
            const float dg = ar.diag[u];
            const float z = (dg != 0.f ? 1.0f / dg : 1.0f) * r;
            rr += (double)r * (double)r; rz += (double)r * (double)z;
        }
    }
    block_part_store(rr, PART(a, SC_AUX0), red);
    block_part_store(rz, PART(a, SC_AUX1), red);
}
void launch_areg_cg_update(const SweepArgs& a, float alpha, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_areg_cg_update, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, alpha);
}
__global__ void __launch_bounds__(kBlock) k_areg_cg_dir(SweepArgs a, float beta) {
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.row1) return;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const size_t u = (size_t)ch * b.Spad + j;
        const float dg = ar.diag[u];
        ar.p[u] = (dg != 0.f ? 1.0f / dg : 1.0f) * ar.r[u] + beta * ar.p[u];
    }
}
void launch_areg_cg_dir(const SweepArgs& a, float beta, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_areg_cg_dir, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, beta);
}
// ---- device-driven CG (round 6; single rank).  The four kernels of an iteration as above, but alpha, beta and the stop test are derived on the device:
// every workgroup sums the previous kernel's per-workgroup partials in block_total's fixed order (the order of the host's read-backs: the same bits as the
// host-driven loop) and the chain r.z_old -> r.z_new travels through ar.cgs[3 + parity].  Once |r|^2 < threshold the flag ar.cgs[0] turns the rest of an
// enqueued chunk of iterations into no-ops; the host looks at ar.cgs once per chunk (loop.hip albedo_reg_solve).
__global__ void __launch_bounds__(kBlock) k_areg_jx_d(SweepArgs a) {
    if (a.ar.cgs[0] != 0.0) return;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.row1) return;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) a.ar.t[(size_t)ch * a.b.Spad + j] = areg_jrow(a, a.ar.p, j, ch);
}
__global__ void __launch_bounds__(kBlock) k_areg_jt_d(SweepArgs a) {
    __shared__ double red[kBlock / 64];
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    if (ar.cgs[0] != 0.0) return;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double pq = 0;
    if (j < a.row1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const size_t u = (size_t)ch * b.Spad + j;
            float v = b.aH[u] * ar.p[u] + ar.weight * areg_jtcol<false>(a, ar.t, j, ch);
            if (a.damping != 0.0f) v += (a.damping * ar.diag0[u]) * ar.p[u];
            ar.q[u] = v; pq += (double)ar.p[u] * (double)v;
        }
    }
    block_part_store(pq, PART(a, SC_AUX0), red);
}
__global__ void __launch_bounds__(kBlock) k_areg_update_d(SweepArgs a, int i, int nblk) {
    __shared__ double red[kBlock / 64];
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    if (ar.cgs[0] != 0.0) return;
    const double pq = block_total(PART(a, SC_AUX0), nblk, red);
    const float alpha = (float)ar.cgs[3 + (i & 1)] / (float)pq;      // alpha = absNew / p.dot(tmp)
    __syncthreads();      // (red is reused below)
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double rr = 0, rz = 0;
    if (j < a.row1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const size_t u = (size_t)ch * b.Spad + j;
            ar.x[u] += alpha * ar.p[u];
            const float r = ar.r[u] - alpha * ar.q[u];
            ar.r[u] = r;
            const float dg = ar.diag[u];
            const float z = (dg != 0.f ? 1.0f / dg : 1.0f) * r;
            rr += (double)r * (double)r; rz += (double)r * (double)z;
        }
    }
    block_part_store(rr, PART(a, SC_AUX1), red);
    block_part_store(rz, PART(a, SC_AUX2), red);
}
__global__ void __launch_bounds__(kBlock) k_areg_dir_d(SweepArgs a, int i, int nblk, float thr) {
    __shared__ double red[kBlock / 64];
    __shared__ double s_stop;
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    if (threadIdx.x == 0) s_stop = ar.cgs[0];      // (workgroup 0 of THIS kernel may raise the flag: every thread of a workgroup must see the same value)
    __syncthreads();
    if (s_stop != 0.0) return;
    const double rr = block_total(PART(a, SC_AUX1), nblk, red);
    __syncthreads();
    const double rz = block_total(PART(a, SC_AUX2), nblk, red);
    const float res2 = (float)rr;
    if (res2 < thr) {      // Eigen leaves the loop before ++i
        if (blockIdx.x == 0 && threadIdx.x == 0) { ar.cgs[1] = (double)i; ar.cgs[2] = rr; __threadfence(); ar.cgs[0] = 1.0; }
        return;
    }
    const float beta = (float)rz / (float)ar.cgs[3 + (i & 1)];      // absNew / absOld
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j < a.row1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const size_t u = (size_t)ch * b.Spad + j;
            const float dg = ar.diag[u];
            ar.p[u] = (dg != 0.f ? 1.0f / dg : 1.0f) * ar.r[u] + beta * ar.p[u];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { ar.cgs[3 + ((i + 1) & 1)] = rz; ar.cgs[1] = (double)(i + 1); ar.cgs[2] = rr; }
}
void launch_areg_cg_iteration(const SweepArgs& a, int i, int nblk, float thr, hipStream_t s) {
    if (a.row1 <= a.row0) return;
    const dim3 g((a.row1 - a.row0 + kBlock - 1) / kBlock), bl(kBlock);
    hipLaunchKernelGGL(k_areg_jx_d, g, bl, 0, s, a);
    hipLaunchKernelGGL(k_areg_jt_d, g, bl, 0, s, a);
    hipLaunchKernelGGL(k_areg_update_d, g, bl, 0, s, a, i, nblk);
    hipLaunchKernelGGL(k_areg_dir_d, g, bl, 0, s, a, i, nblk, thr);
}
// updateAlbedo (OptimizerAux.cpp:120-150) with a solved step instead of b / H
__global__ void __launch_bounds__(kBlock) k_apply_albedo_delta(SweepArgs a, const float* __restrict__ delta) {
    __shared__ double red[kBlock / 64];
    const Band& b = a.b;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double cnt = 0;
    if (j < a.row1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float v = b.rho[ch][j] - delta[(size_t)ch * b.Spad + j];
            if (v > 0.0f && v < 1.0f) { set_rho(b, j, ch, v); cnt += 1.0; }
        }
    }
    block_part_store(cnt, PART(a, SC_ACCEPT), red);
}
void launch_apply_albedo_delta(const SweepArgs& a, const float* delta, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_apply_albedo_delta, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, delta);
}

}  // namespace psg
