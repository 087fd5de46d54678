// extract.hip -- the writers' geometry on the device: what voxelPS dumps every third iteration used to cost a
// download of the whole dense volume (7 N^3 floats), a single-threaded marching-cubes pass and a pass over N^3 voxels on the host.  Here:
//   psgsdf_extract_mesh        Optimizer::extract_mesh (OptimizerAux.cpp:278-363) + MarchingCubes::computeIsoSurface / computeTriangles
//                              (third/mesh/MarchingCubes.cpp:314-637): crop box of |d| <= sqrt(3) vs, count -> exclusive scan -> emit, the reference's
//                              cell order (z, y, x) and the classic table's face order, non-indexed vertices
//   psgsdf_extract_pointcloud  Optimizer::save_pointcloud (OptimizerAux.cpp:456-511, band voxels) / VolumetricGradSdf::extract_pc
//                              (VolumetricGradSdf.cpp:320-376, every voxel with weight > 0): flags -> scan -> fill in ascending voxel order
//   psgsdf_extract_sdf         the cropped -dist block of Optimizer::saveSDF / VolumetricGradSdf::saveSDF (OptimizerAux.cpp:513-577)
// The arithmetic is the HOST writers' (psgradientsdf_amd/host/marching_cubes.hpp, ps_optimizer.hpp), operation for operation, with FMA contraction
// off: the files written from these arrays are byte-identical to the ones the host-side pass writes (tests/test_extract_gpu.py).
// Multi-rank contexts (z-slabs): the calls are collective and every rank returns ITS share -- the cells whose lower z-plane it owns, its own band rows /
// voxels / planes of the (global) crop box -- in the single context's order, so that the shares concatenated in rank order ARE the single context's
// arrays (tests/test_extract_gpu.py gathers the slabs' volume into one context and compares).  Two exchanges: the crop box (one all-reduce of 6 R
// numbers) and, for the mesh, the albedo of the upper neighbour's first plane (a cell reaches one plane up; the engine itself only keeps the halo
// planes' DISTANCES current).
#include "engine_internal.h"

namespace psg {
namespace {

#pragma clang fp contract(off)

__constant__ signed char kTri[256][16] = {
#include "../host/mc_tritable.inc"
};
__constant__ int kCornerD[8][3] = {{1, 1, 0}, {1, 0, 0}, {0, 0, 0}, {0, 1, 0}, {1, 1, 1}, {1, 0, 1}, {0, 0, 1}, {0, 1, 1}};      // marching_cubes.hpp kCorner (computeLutIndex :511-556)
__constant__ int kEdgeD[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

// ---- crop box: min / max voxel index over |d| <= sqrt(3) vs (crop_box, ps_optimizer.hpp; the comparison is the host's: double)
__global__ void __launch_bounds__(kBlock) k_box_part(const float* __restrict__ dist, int nx, int ny, long long nvox, int k0, double lim, int* __restrict__ part) {
    int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-(1 << 30), -(1 << 30), -(1 << 30)};
    const long long nxy = (long long)nx * ny;
    for (long long lin = blockIdx.x * (long long)blockDim.x + threadIdx.x; lin < nvox; lin += (long long)gridDim.x * blockDim.x) {
        if ((double)fabsf(dist[lin]) > lim) continue;
        const int kl = (int)(lin / nxy), rest = (int)(lin - (long long)kl * nxy), j = rest / nx, i = rest - j * nx, k = kl + k0;      // (k0: global z of the first plane looked at)
        lo[0] = min(lo[0], i); hi[0] = max(hi[0], i); lo[1] = min(lo[1], j); hi[1] = max(hi[1], j); lo[2] = min(lo[2], k); hi[2] = max(hi[2], k);
    }
    __shared__ int s[6][kBlock / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int l = lo[a], h = hi[a];
        for (int o = 32; o > 0; o >>= 1) { l = min(l, __shfl_xor(l, o, 64)); h = max(h, __shfl_xor(h, o, 64)); }
        if (lane == 0) { s[a][w] = l; s[3 + a][w] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int v = s[threadIdx.x][0];
        for (int i = 1; i < kBlock / 64; ++i) v = threadIdx.x < 3 ? min(v, s[threadIdx.x][i]) : max(v, s[threadIdx.x][i]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
__global__ void __launch_bounds__(kBlock) k_box_final(const int* __restrict__ part, int nblk, int* __restrict__ box) {
    __shared__ int s[6][kBlock];
    int v[6] = {1 << 30, 1 << 30, 1 << 30, -(1 << 30), -(1 << 30), -(1 << 30)};
    for (int b = threadIdx.x; b < nblk; b += blockDim.x)
        for (int a = 0; a < 6; ++a) v[a] = a < 3 ? min(v[a], part[b * 6 + a]) : max(v[a], part[b * 6 + a]);
    for (int a = 0; a < 6; ++a) s[a][threadIdx.x] = v[a];
    __syncthreads();
    if (threadIdx.x < 6) {
        int r = s[threadIdx.x][0];
        for (int i = 1; i < kBlock; ++i) r = threadIdx.x < 3 ? min(r, s[threadIdx.x][i]) : max(r, s[threadIdx.x][i]);
        box[threadIdx.x] = r;
    }
}

// ---- exclusive scan of int counts (1024-element tiles; the counts are replaced by their offsets, *total = their sum)
constexpr int kTile = 1024;
__global__ void __launch_bounds__(kBlock) k_cscan_tile(int* __restrict__ v, long long n, int* __restrict__ sums) {
    __shared__ int wsum[kBlock / 64];
    const long long base = (long long)blockIdx.x * kTile + threadIdx.x * 4;
    int f[4]; int loc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = (base + i < n) ? v[base + i] : 0; loc += f[i]; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < w; ++i) woff += wsum[i];
    int excl = woff + inc - loc;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (base + i < n) v[base + i] = excl; excl += f[i]; }
    if (threadIdx.x == kBlock - 1) sums[blockIdx.x] = woff + inc;
}
__global__ void __launch_bounds__(1024) k_cscan_sums(int* __restrict__ sums, int nb, int* __restrict__ total) {
    __shared__ int wsum[16]; __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int val = i < nb ? sums[i] : 0;
        int inc = val;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        const int carry = carry_s;
        if (i < nb) sums[i] = carry + woff + inc - val;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}
__global__ void __launch_bounds__(kBlock) k_cscan_add(int* __restrict__ v, long long n, const int* __restrict__ sums) {
    const long long base = (long long)blockIdx.x * kTile + threadIdx.x * 4;
    const int off = sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) if (base + i < n) v[base + i] += off;
}

// ---- marching cubes over the cropped grid
struct McGrid {
    const float* dist; const float* weight; const float* rho[3];
    int nx, ny;                 // the dense grid's row / plane strides
    int lo[3], d[3];            // crop box: first voxel and extent (global)
    int zlo;                    // global z of the context's local plane 0 (0 on one rank)
    int zc0;                    // first cell plane of this launch (a slab: the cells whose lower plane it owns)
    float voxel[3], origin[3];  // MarchingCubes ctor: size / dim, and the origin offset that is subtracted (write_mesh: -vs * lo)
    long long total;            // voxels of the cropped grid (the colour look-ups run one / two past it: B10)
};
__device__ __forceinline__ long long dense_lin(const McGrid& g, int i, int j, int k) { return (long long)(k + g.lo[2] - g.zlo) * g.nx * g.ny + (long long)(j + g.lo[1]) * g.nx + (i + g.lo[0]); }
// colour byte of channel ch at CROPPED linear index cl (write_mesh: (unsigned char)int(255 * rgb); at(): 0 past the last voxel)
__device__ __forceinline__ unsigned char colour_at(const McGrid& g, int ch, long long cl) {
    if (cl >= g.total) return 0;
    const int k = (int)(cl / ((long long)g.d[0] * g.d[1])), rest = (int)(cl - (long long)k * g.d[0] * g.d[1]), j = rest / g.d[0], i = rest - j * g.d[0];
    return (unsigned char)(int)(255 * g.rho[ch][dense_lin(g, i, j, k)]);
}
// MarchingCubes.cpp:559-579 (marching_cubes.hpp interpolate)
__device__ __forceinline__ void mc_interp(float t0, float t1, const float* v0, const float* v1, float* out) {
    const float iso = 0.0f;
    if ((double)fabsf(iso - t0) < 1e-7) { for (int a = 0; a < 3; ++a) out[a] = v0[a]; return; }
    if ((double)fabsf(iso - t1) < 1e-7) { for (int a = 0; a < 3; ++a) out[a] = v1[a]; return; }
    if ((double)fabsf(t0 - t1) < 1e-7) { for (int a = 0; a < 3; ++a) out[a] = v0[a]; return; }
    double mu = (double)((iso - t0) / (t1 - t0));
    if (mu > 1.0) mu = 1.0; else if (mu < 0) mu = 0.0;
    for (int a = 0; a < 3; ++a) out[a] = (float)((double)v0[a] + mu * (double)(v1[a] - v0[a]));
}
// one cell: its triangles (non-degenerate ones, in table order) -> count, or written at `out_v / out_c` (3 vertices per face)
template <bool EMIT>
__device__ __forceinline__ int mc_cell(const McGrid& g, int x, int y, int z, float* out_v, unsigned char* out_c) {
    long long off[8]; long long dl[8]; bool valid = true; int cs = 0;
    float t[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int i = x + kCornerD[c][0], j = y + kCornerD[c][1], k = z + kCornerD[c][2];
        off[c] = (long long)k * g.d[0] * g.d[1] + (long long)j * g.d[0] + i;
        dl[c] = dense_lin(g, i, j, k);
        if (g.weight[dl[c]] == 0.0f) valid = false;
    }
    if (!valid) return 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) { t[c] = -g.dist[dl[c]]; if (t[c] > 0.0f) cs |= 1 << c; }
    if (cs == 0 || cs == 255) return 0;
    float ep[12][3]; unsigned char ec[12][3]; unsigned have = 0;
    for (int q = 0; q < 16 && kTri[cs][q] >= 0; ++q) {
        const int e = kTri[cs][q];
        if (have & (1u << e)) continue;
        have |= 1u << e;
        const int a = kEdgeD[e][0], b = kEdgeD[e][1];
        float pa[3], pb[3];
        const int base[3] = {x, y, z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int ia = base[k] + kCornerD[a][k], ib = base[k] + kCornerD[b][k];
            pa[k] = ia * g.voxel[k] - g.origin[k]; pb[k] = ib * g.voxel[k] - g.origin[k];      // voxelToWorld, :647-651
        }
        mc_interp(t[a], t[b], pa, pb, ep[e]);
        if (EMIT) {
            // getColor :592-608 with the reversed end points of edges 2, 3, 6, 7 and the +1 / +2 index offsets of green / blue (B10)
            const bool rev = e == 2 || e == 3 || e == 6 || e == 7;
            const int c1 = rev ? b : a, c2 = rev ? a : b;
            const long long o1 = off[c1], o2 = off[c2];
            const float ca[3] = {colour_at(g, 0, o1) / 255.0f, colour_at(g, 1, o1 + 1) / 255.0f, colour_at(g, 2, o1 + 2) / 255.0f};
            const float cb[3] = {colour_at(g, 0, o2) / 255.0f, colour_at(g, 1, o2 + 1) / 255.0f, colour_at(g, 2, o2 + 2) / 255.0f};
            float cv[3]; mc_interp(t[c1], t[c2], ca, cb, cv);
            for (int k = 0; k < 3; ++k) ec[e][k] = (unsigned char)(cv[k] * 255.0f);
        }
    }
    int n = 0;
    for (int q = 0; q + 2 < 16 && kTri[cs][q] >= 0; q += 3) {
        const int e0 = kTri[cs][q], e1 = kTri[cs][q + 1], e2 = kTri[cs][q + 2];
        auto same = [&](int p, int r) { return ep[p][0] == ep[r][0] && ep[p][1] == ep[r][1] && ep[p][2] == ep[r][2]; };
        if (same(e0, e1) || same(e0, e2) || same(e1, e2)) continue;   // degenerate, :623
        if (EMIT) {
            const int es[3] = {e0, e1, e2};
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 3; ++k) { out_v[(n * 3 + v) * 3 + k] = ep[es[v]][k]; out_c[(n * 3 + v) * 3 + k] = ec[es[v]][k]; }
        }
        ++n;
    }
    return n;
}
__global__ void __launch_bounds__(kBlock) k_mc_count(McGrid g, long long ncell, int* __restrict__ cnt) {
    const long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const int cx = g.d[0] - 2, cy = g.d[1] - 2;
    const int zl = (int)(c / ((long long)cx * cy)), rest = (int)(c - (long long)zl * cx * cy), y = rest / cx, x = rest - y * cx, z = zl + g.zc0;
    cnt[c] = mc_cell<false>(g, x, y, z, nullptr, nullptr);
}
__global__ void __launch_bounds__(kBlock) k_mc_emit(McGrid g, long long ncell, const int* __restrict__ offs, int total, float* __restrict__ xyz, unsigned char* __restrict__ rgb) {
    const long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const int mine = (c + 1 < ncell ? offs[c + 1] : total) - offs[c];
    if (mine <= 0) return;
    const int cx = g.d[0] - 2, cy = g.d[1] - 2;
    const int zl = (int)(c / ((long long)cx * cy)), rest = (int)(c - (long long)zl * cx * cy), y = rest / cx, x = rest - y * cx, z = zl + g.zc0;
    mc_cell<true>(g, x, y, z, xyz + (size_t)offs[c] * 9, rgb + (size_t)offs[c] * 9);
}

// ---- point clouds
// which = 0: the band voxels (ascending) with |d| < sqrt(3) vs;  which = 1: every voxel with weight > 0 and |d| < sqrt(3) vs
__global__ void __launch_bounds__(kBlock) k_pc_flags(const float* __restrict__ dist, const float* __restrict__ weight, const int* __restrict__ band_lin, long long lin0, long long n, double lim, int* __restrict__ flag) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long lin = band_lin ? band_lin[i] : i + lin0;      // (lin0: the first voxel of the planes the context owns)
    flag[i] = ((double)fabsf(dist[lin]) < lim && (band_lin || weight[lin] > 0)) ? 1 : 0;
}
__global__ void __launch_bounds__(kBlock) k_pc_fill(DenseView d, int nx, int ny, int zlo, float vs, const int* __restrict__ band_lin, long long lin0, long long n, double lim, const int* __restrict__ offs,
                                                    float* __restrict__ pn, int* __restrict__ col) {
    const long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (q >= n) return;
    const long long lin = band_lin ? band_lin[q] : q + lin0;
    if (!((double)fabsf(d.dist[lin]) < lim && (band_lin || d.weight[lin] > 0))) return;
    const long long nxy = (long long)nx * ny;
    const int kl = (int)(lin / nxy), rest = (int)(lin - (long long)kl * nxy), j = rest / nx, i = rest - j * nx, k = kl + zlo;
    float g[3] = {d.g[0][lin], d.g[1][lin], d.g[2][lin]};
    const float z = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
    if (z > 0) { const float s = sqrtf(z); g[0] /= s; g[1] /= s; g[2] /= s; }
    const float dd = d.dist[lin];
    const size_t o = (size_t)offs[q];
    pn[o * 6 + 0] = vs * i - dd * g[0]; pn[o * 6 + 1] = vs * j - dd * g[1]; pn[o * 6 + 2] = vs * k - dd * g[2];
    pn[o * 6 + 3] = g[0]; pn[o * 6 + 4] = g[1]; pn[o * 6 + 5] = g[2];
    for (int a = 0; a < 3; ++a) col[o * 3 + a] = (int)(255 * d.rho[a][lin]);      // (printed as int, not as a byte: ps_optimizer.hpp save_pointcloud)
}
__global__ void __launch_bounds__(kBlock) k_sdf_crop(const float* __restrict__ dist, int nx, int ny, int lo0, int lo1, int lo2, int d0, int d1, long long n, float* __restrict__ out) {
    const long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int k = (int)(q / ((long long)d0 * d1)), rest = (int)(q - (long long)k * d0 * d1), j = rest / d0, i = rest - j * d0;
    out[q] = -dist[(long long)(k + lo2) * nx * ny + (long long)(j + lo1) * nx + (i + lo0)];
}

}  // namespace
}  // namespace psg

using namespace psge;

namespace {
int scan_counts(psgsdf_ctx* c, int* v, long long n, int* sums, int* total_host) {
    const int nb = (int)((n + psg::kTile - 1) / psg::kTile);
    hipLaunchKernelGGL(psg::k_cscan_tile, dim3(nb), dim3(kBlock), 0, c->stream, v, n, sums);
    hipLaunchKernelGGL(psg::k_cscan_sums, dim3(1), dim3(1024), 0, c->stream, sums, nb, c->d_total);
    hipLaunchKernelGGL(psg::k_cscan_add, dim3(nb), dim3(kBlock), 0, c->stream, v, n, (const int*)sums);
    HIPCHK(c, hipMemcpyAsync(total_host, c->d_total, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
// engine-owned pinned host buffer that lives until the next extraction on this context
int host_out(psgsdf_ctx* c, int slot, size_t bytes, void** p) {
    if (c->xo_bytes[slot] < bytes) {
        if (c->xo_host[slot]) hipHostFree(c->xo_host[slot]);
        c->xo_host[slot] = nullptr; c->xo_bytes[slot] = 0;
        HIPCHK(c, hipHostMalloc(&c->xo_host[slot], std::max<size_t>(bytes, 4096), hipHostMallocDefault));
        c->xo_bytes[slot] = std::max<size_t>(bytes, 4096);
    }
    *p = c->xo_host[slot];
    return 0;
}
// the crop box of |d| <= sqrt(3) vs; any = false if no voxel qualifies
int crop_box_dev(psgsdf_ctx* c, int lo[3], int hi[3], bool* any) {
    // (a slab looks at the planes it OWNS: [z0, z1) of the volume = local planes [z0 - zlo, z1 - zlo))
    const long long plane = (long long)c->grid.dim[0] * c->grid.dim[1], n = plane * (c->z1 - c->z0);
    const float* dist0 = c->dense.dist + plane * (c->z0 - c->zlo);
    const int nblk = (int)std::max<long long>(1, std::min<long long>((n + kBlock - 1) / kBlock, 2048));
    int* part = nullptr;
    HIPCHK(c, hipMalloc(&part, sizeof(int) * (6 * (size_t)nblk + 6)));
    const double lim = sqrt(3.0) * (double)c->grid.vs;      // std::sqrt(3) * vs: double (ps_optimizer.hpp crop_box)
    hipLaunchKernelGGL(psg::k_box_part, dim3(nblk), dim3(kBlock), 0, c->stream, dist0, c->grid.dim[0], c->grid.dim[1], n, c->z0, lim, part);
    hipLaunchKernelGGL(psg::k_box_final, dim3(1), dim3(kBlock), 0, c->stream, part, nblk, part + 6 * (size_t)nblk);
    int box[6];
    hipError_t e = hipMemcpyAsync(box, part + 6 * (size_t)nblk, sizeof(box), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(part);
    if (e != hipSuccess) return fail(c, PSGSDF_ERR_DEVICE, "crop box: %s", hipGetErrorString(e));
    if (c->n_ranks > 1) {      // min / max over the slabs: every rank's six numbers in its own slots of one sum
        std::vector<double> all((size_t)6 * c->n_ranks, 0.0);
        for (int a = 0; a < 6; ++a) all[(size_t)6 * c->rank + a] = (double)box[a];
        if (int rc = host_allreduce(c, all, "crop box")) return rc;
        for (int r = 0; r < c->n_ranks; ++r)
            for (int a = 0; a < 6; ++a) { const int v = (int)all[(size_t)6 * r + a]; box[a] = r == 0 ? v : (a < 3 ? std::min(box[a], v) : std::max(box[a], v)); }
    }
    for (int a = 0; a < 3; ++a) { lo[a] = box[a]; hi[a] = box[3 + a]; }
    *any = hi[0] >= lo[0];
    return 0;
}
int extract_ready(psgsdf_ctx* c, const char* what) {
    if (!c || !c->have_volume) return fail(c, PSGSDF_ERR_STATE, "%s: no volume", what);
    if (c->n_ranks > 1 && !c->comm) return fail(c, PSGSDF_ERR_COMM, "%s: rank %d of %d has no communicator", what, c->rank, c->n_ranks);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->deferred.empty() || c->pending_fold.n) { int rc = flush(c); if (rc) return rc; }
    if (c->inited) launch_band_scatter(c->dense, c->band, c->stream);      // the band's state back into the dense arrays (as psgsdf_download_volume does)
    return 0;
}
}  // namespace

extern "C" {

int psgsdf_extract_mesh(psgsdf_ctx* c, const float** xyz, const uint8_t** rgb, int64_t* n_vertices) {
    if (!xyz || !rgb || !n_vertices) return fail(c, PSGSDF_ERR_ARG, "extract_mesh: null argument");
    { int rc = extract_ready(c, "extract_mesh"); if (rc) return rc; }
    *xyz = nullptr; *rgb = nullptr; *n_vertices = 0;
    int lo[3], hi[3]; bool any = false;
    { int rc = crop_box_dev(c, lo, hi, &any); if (rc) return rc; }
    if (!any) return PSGSDF_OK;
    psg::McGrid g{};
    g.dist = c->dense.dist; g.weight = c->dense.weight; for (int a = 0; a < 3; ++a) g.rho[a] = c->dense.rho[a];
    g.nx = c->grid.dim[0]; g.ny = c->grid.dim[1]; g.zlo = c->zlo;
    const float vs = c->grid.vs;
    for (int a = 0; a < 3; ++a) {
        g.lo[a] = lo[a]; g.d[a] = hi[a] - lo[a] + 1;
        const float size = vs * g.d[a];                    // write_mesh: size[] = {vs * d[0], ..}, org[] = {-vs * lo[0], ..}
        g.voxel[a] = size / g.d[a];                        // MarchingCubes ctor: voxel_ = size / dim
        g.origin[a] = -vs * lo[a];
    }
    g.total = (long long)g.d[0] * g.d[1] * g.d[2];
    if (g.d[0] < 3 || g.d[1] < 3 || g.d[2] < 3) return PSGSDF_OK;      // (no cell: the loops of computeIsoSurface run to dim - 2)
    // the cells of this context: lower plane lo2 + zc in [z0, z1) (their upper plane is the halo plane of a slab with a neighbour above)
    g.zc0 = std::max(0, c->z0 - lo[2]);
    const int zc1 = std::min(g.d[2] - 2, c->z1 - lo[2]);
    if (c->n_ranks > 1) {      // the albedo of plane z1 from the rank above (the cells' upper corners and their colours)
        const size_t plane = (size_t)g.nx * g.ny;
        std::vector<psgsdf_comm_xfer> sends, recvs;
        for (int a = 0; a < 3; ++a) {
            if (c->rank > 0) sends.push_back({(void*)(c->dense.rho[a] + plane * (size_t)(c->z0 - c->zlo)), sizeof(float) * plane, c->rank - 1});
            if (c->rank + 1 < c->n_ranks) recvs.push_back({(void*)(c->dense.rho[a] + plane * (size_t)(c->z1 - c->zlo)), sizeof(float) * plane, c->rank + 1});
        }
        if (int rc = comm_xfer(c, sends, recvs)) return rc;
    }
    const long long ncell = (long long)(g.d[0] - 2) * (g.d[1] - 2) * std::max(0, zc1 - g.zc0);
    if (ncell == 0) return PSGSDF_OK;
    if (ncell >= (1ll << 31)) return fail(c, PSGSDF_ERR_UNSUPPORTED, "extract_mesh: %lld cells", ncell);
    const int nb = (int)((ncell + psg::kTile - 1) / psg::kTile);
    int* cnt = nullptr; int* sums = nullptr;
    HIPCHK(c, hipMalloc(&cnt, sizeof(int) * (size_t)ncell));
    if (hipMalloc(&sums, sizeof(int) * (size_t)(nb + 1)) != hipSuccess) { hipFree(cnt); return fail(c, PSGSDF_ERR_DEVICE, "extract_mesh: out of memory"); }
    int rc = 0, total = 0;
    float* d_xyz = nullptr; unsigned char* d_rgb = nullptr;
    timed(c, "mc_count", [&] { hipLaunchKernelGGL(psg::k_mc_count, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, g, ncell, cnt); });
    rc = scan_counts(c, cnt, ncell, sums, &total);
    if (!rc && total > 0) {
        const size_t nv = (size_t)total * 3;
        if (hipMalloc(&d_xyz, sizeof(float) * 3 * nv) != hipSuccess || hipMalloc(&d_rgb, 3 * nv) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "extract_mesh: out of memory (%d faces)", total);
        void *hx = nullptr, *hc = nullptr;
        if (!rc) rc = host_out(c, 0, sizeof(float) * 3 * nv, &hx);
        if (!rc) rc = host_out(c, 1, 3 * nv, &hc);
        if (!rc) {
            timed(c, "mc_emit", [&] { hipLaunchKernelGGL(psg::k_mc_emit, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, g, ncell, (const int*)cnt, total, d_xyz, d_rgb); });
            if (hipMemcpyAsync(hx, d_xyz, sizeof(float) * 3 * nv, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipMemcpyAsync(hc, d_rgb, 3 * nv, hipMemcpyDeviceToHost, c->stream) != hipSuccess
                || hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "extract_mesh: download");
        }
        if (!rc) { *xyz = (const float*)hx; *rgb = (const uint8_t*)hc; *n_vertices = (int64_t)nv; }
    }
    hipFree(cnt); hipFree(sums); hipFree(d_xyz); hipFree(d_rgb);
    return rc;
}

int psgsdf_extract_pointcloud(psgsdf_ctx* c, int which, const float** xyz_nxyz, const int32_t** rgb, int64_t* n_points) {
    if (!xyz_nxyz || !rgb || !n_points || which < 0 || which > 1) return fail(c, PSGSDF_ERR_ARG, "extract_pointcloud: bad argument");
    { int rc = extract_ready(c, "extract_pointcloud"); if (rc) return rc; }
    if (which == 0 && !c->inited) return fail(c, PSGSDF_ERR_STATE, "extract_pointcloud(band): psgsdf_init first");
    *xyz_nxyz = nullptr; *rgb = nullptr; *n_points = 0;
    // a slab: its own band rows [row0, row1) / the voxels of the planes it owns
    const long long plane = (long long)c->grid.dim[0] * c->grid.dim[1], lin0 = plane * (c->z0 - c->zlo);
    const long long n = which == 0 ? (long long)(c->row1 - c->row0) : plane * (c->z1 - c->z0);
    if (n <= 0) return PSGSDF_OK;
    const int* band_lin = which == 0 ? c->band.lin + c->row0 : nullptr;
    const double lim = sqrt(3.0) * (double)c->grid.vs;
    const int nb = (int)((n + psg::kTile - 1) / psg::kTile);
    int* flag = nullptr; int* sums = nullptr;
    HIPCHK(c, hipMalloc(&flag, sizeof(int) * (size_t)n));
    if (hipMalloc(&sums, sizeof(int) * (size_t)(nb + 1)) != hipSuccess) { hipFree(flag); return fail(c, PSGSDF_ERR_DEVICE, "extract_pointcloud: out of memory"); }
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(psg::k_pc_flags, dim3(grid), dim3(kBlock), 0, c->stream, c->dense.dist, c->dense.weight, band_lin, lin0, n, lim, flag);
    int total = 0;
    int rc = scan_counts(c, flag, n, sums, &total);
    float* d_pn = nullptr; int* d_col = nullptr;
    if (!rc && total > 0) {
        if (hipMalloc(&d_pn, sizeof(float) * 6 * (size_t)total) != hipSuccess || hipMalloc(&d_col, sizeof(int) * 3 * (size_t)total) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "extract_pointcloud: out of memory");
        void *hp = nullptr, *hc = nullptr;
        if (!rc) rc = host_out(c, 2, sizeof(float) * 6 * (size_t)total, &hp);
        if (!rc) rc = host_out(c, 3, sizeof(int) * 3 * (size_t)total, &hc);
        if (!rc) {
            hipLaunchKernelGGL(psg::k_pc_fill, dim3(grid), dim3(kBlock), 0, c->stream, c->dense, c->grid.dim[0], c->grid.dim[1], c->zlo, c->grid.vs, band_lin, lin0, n, lim, (const int*)flag, d_pn, d_col);
            if (hipMemcpyAsync(hp, d_pn, sizeof(float) * 6 * (size_t)total, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipMemcpyAsync(hc, d_col, sizeof(int) * 3 * (size_t)total, hipMemcpyDeviceToHost, c->stream) != hipSuccess
                || hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "extract_pointcloud: download");
        }
        if (!rc) { *xyz_nxyz = (const float*)hp; *rgb = (const int32_t*)hc; *n_points = total; }
    }
    hipFree(flag); hipFree(sums); hipFree(d_pn); hipFree(d_col);
    return rc;
}

int psgsdf_extract_sdf(psgsdf_ctx* c, int32_t lo[3], int32_t dim[3], const float** neg_dist) {
    if (!lo || !dim || !neg_dist) return fail(c, PSGSDF_ERR_ARG, "extract_sdf: null argument");
    { int rc = extract_ready(c, "extract_sdf"); if (rc) return rc; }
    *neg_dist = nullptr; for (int a = 0; a < 3; ++a) { lo[a] = 0; dim[a] = 0; }
    int l[3], h[3]; bool any = false;
    { int rc = crop_box_dev(c, l, h, &any); if (rc) return rc; }
    if (!any) return PSGSDF_OK;
    const int d0 = h[0] - l[0] + 1, d1 = h[1] - l[1] + 1, d2 = h[2] - l[2] + 1;
    for (int a = 0; a < 3; ++a) lo[a] = l[a];
    dim[0] = d0; dim[1] = d1; dim[2] = d2;
    // a slab: the planes of the box it owns, [ka, kb)
    const int ka = std::max(l[2], c->z0), kb = std::min(h[2] + 1, c->z1);
    const long long n = (long long)d0 * d1 * std::max(0, kb - ka);
    if (n == 0) return PSGSDF_OK;
    float* dv = nullptr; void* hv = nullptr;
    HIPCHK(c, hipMalloc(&dv, sizeof(float) * (size_t)n));
    int rc = host_out(c, 4, sizeof(float) * (size_t)n, &hv);
    if (!rc) {
        hipLaunchKernelGGL(psg::k_sdf_crop, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->dense.dist, c->grid.dim[0], c->grid.dim[1], l[0], l[1], ka - c->zlo, d0, d1, n, dv);
        if (hipMemcpyAsync(hv, dv, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, PSGSDF_ERR_DEVICE, "extract_sdf: download");
    }
    hipFree(dv);
    if (rc) return rc;
    *neg_dist = (const float*)hv;
    return PSGSDF_OK;
}

}  // extern "C"
