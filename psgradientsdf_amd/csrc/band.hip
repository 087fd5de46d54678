// band.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Gradient-SDF photometric-stereo hot path:
// dense-grid kernels (visibility selection, band construction, scatter, 2x refinement), per-voxel derived quantities, observation lists, scalar folds.  No CUDA compatibility layer, no other back end.
// Shared device helpers: device_common.h; the launchers are declared in engine.h.
#include "device_common.h"

namespace psg {

// ------------------------------------------------------------------------------------------
// dense-grid kernels: visibility selection, band construction, scatter, 2x refinement
// ------------------------------------------------------------------------------------------

// Optimizer.cpp:30-47 select_vis
__global__ void __launch_bounds__(kBlock) k_select_vis(const uint64_t* __restrict__ vis_seq, int wpv_seq, uint64_t* __restrict__ vis_key, int KW, const int* __restrict__ frame_idx, int F, long long nvox) {
    for (long long lin = blockIdx.x * (long long)blockDim.x + threadIdx.x; lin < nvox; lin += (long long)gridDim.x * blockDim.x) {
        if (wpv_seq <= 2) {
            // the common layouts (<= 128 integrated frames): the voxel's sequence words are loaded ONCE -- round 4 re-loaded the word for every keyframe
            // (50 loads per voxel, 546 us at 256^3 x 50) -- and a voxel no frame has seen (98 % of the grid) is done after that load
            const uint64_t s0 = vis_seq[lin * wpv_seq], s1 = wpv_seq > 1 ? vis_seq[lin * wpv_seq + 1] : 0ull;
            for (int w = 0; w < KW; ++w) {
                uint64_t out = 0;
                if (s0 | s1) {
                    const int f1 = min(F, 64 * (w + 1));
                    for (int f = 64 * w; f < f1; ++f) {
                        const int s = frame_idx[f];
                        const uint64_t word = s < 64 ? s0 : s1;
                        if (s >= 0 && s < 64 * wpv_seq && ((word >> (s & 63)) & 1ull)) out |= 1ull << (f & 63);
                    }
                }
                vis_key[lin * KW + w] = out;
            }
            continue;
        }
        for (int w = 0; w < KW; ++w) {
            uint64_t out = 0;
            int f1 = min(F, 64 * (w + 1));
            for (int f = 64 * w; f < f1; ++f) {
                int s = frame_idx[f];
                if (s >= 0 && s < 64 * wpv_seq && ((vis_seq[lin * wpv_seq + (s >> 6)] >> (s & 63)) & 1ull)) out |= 1ull << (f & 63);
            }
            vis_key[lin * KW + w] = out;
        }
    }
}
void launch_select_vis(const uint64_t* vis_seq, int wpv_seq, uint64_t* vis_key, int KW, const int* frame_idx, int F, long long nvox, hipStream_t s) {
    int grid = (int)min((nvox + kBlock - 1) / kBlock, (long long)256 * 16);
    hipLaunchKernelGGL(k_select_vis, dim3(grid), dim3(kBlock), 0, s, vis_seq, wpv_seq, vis_key, KW, frame_idx, F, nvox);
}

// OptimizerAux.cpp:237-257 getSurfaceVoxel membership test
__global__ void __launch_bounds__(kBlock) k_band_flags(const float* __restrict__ dist, const uint64_t* __restrict__ vis_key, int KW, float vs, long long nvox, int* __restrict__ flags) {
    const double thr = sqrt(3.0) * (double)vs;
    for (long long lin = blockIdx.x * (long long)blockDim.x + threadIdx.x; lin < nvox; lin += (long long)gridDim.x * blockDim.x) {
        bool seen = false;
        for (int w = 0; w < KW; ++w) seen |= vis_key[lin * KW + w] != 0;
        flags[lin] = ((double)fabsf(dist[lin]) <= thr && seen) ? 1 : 0;
    }
}
void launch_band_flags(const float* dist, const uint64_t* vis_key, int KW, float vs, long long nvox, int* flags, hipStream_t s) {
    int grid = (int)min((nvox + kBlock - 1) / kBlock, (long long)256 * 16);
    hipLaunchKernelGGL(k_band_flags, dim3(grid), dim3(kBlock), 0, s, dist, vis_key, KW, vs, nvox, flags);
}

// three-phase exclusive scan over 1024-element tiles: flag -> band row (or -1)
constexpr int kScanTile = 1024;
__global__ void __launch_bounds__(kBlock) k_scan_tile(int* __restrict__ v, long long n, int* __restrict__ sums) {
    __shared__ int wsum[kBlock / 64];
    long long base = (long long)blockIdx.x * kScanTile + threadIdx.x * 4;
    int f[4]; int loc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = (base + i < n) ? v[base + i] : 0; loc += f[i]; }
    // inclusive scan of loc across the block: wave scan + wave offsets
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < w; ++i) woff += wsum[i];
    int excl = woff + inc - loc;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (base + i < n) v[base + i] = f[i] ? excl : -1; excl += f[i]; }
    if (threadIdx.x == kBlock - 1) sums[blockIdx.x] = woff + inc;
}
__global__ void __launch_bounds__(1024) k_scan_sums(int* __restrict__ sums, int nb, int* __restrict__ total) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 1024) {
        int i = base + threadIdx.x;
        int val = i < nb ? sums[i] : 0;
        int inc = val;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        int carry = carry_s;
        if (i < nb) sums[i] = carry + woff + inc - val;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}
__global__ void __launch_bounds__(kBlock) k_scan_add(int* __restrict__ v, long long n, const int* __restrict__ sums) {
    long long base = (long long)blockIdx.x * kScanTile + threadIdx.x * 4;
    int off = sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) if (base + i < n) { int x = v[base + i]; if (x >= 0) v[base + i] = x + off; }
}
void launch_band_scan(int* v, long long nvox, int* block_sums, int* d_total, hipStream_t s) {
    int nb = (int)((nvox + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(k_scan_tile, dim3(nb), dim3(kBlock), 0, s, v, nvox, block_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, block_sums, nb, d_total);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(kBlock), 0, s, v, nvox, (const int*)block_sums);
}

// gather the compact band planes from the dense grid (one thread per dense voxel, coalesced reads)
__global__ void __launch_bounds__(kBlock) k_band_fill(DenseView d, GridP grid, Band b) {
    for (long long lin = blockIdx.x * (long long)blockDim.x + threadIdx.x; lin < grid.nvox; lin += (long long)gridDim.x * blockDim.x) {
        int j = d.row_of[lin];
        if (j < 0) continue;
        b.lin[j] = (int)lin;
        b.dist[j] = d.dist[lin];
#pragma unroll
        for (int a = 0; a < 3; ++a) { b.g[a][j] = d.g[a][lin]; set_rho(b, j, a, d.rho[a][lin]); }
        for (int w = 0; w < b.KW; ++w) b.vis[(size_t)w * b.Spad + j] = d.vis[lin * b.KW + w];
    }
}
// neighbour tables: membership by linear index exactly as Optimizer.cpp:462-474 does it
__global__ void __launch_bounds__(kBlock) k_band_nb(DenseView d, GridP grid, Band b, int* __restrict__ reach) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= b.S) return;
    long long lin = b.lin[j];
    long long stride[3] = {1, grid.dim[0], (long long)grid.dim[0] * grid.dim[1]};
    float dj = d.dist[lin];
    int fwd = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        long long ln = lin + ((q & 1) ? -stride[q >> 1] : stride[q >> 1]);
        bool in = ln >= 0 && ln < grid.nvox;
        const int r = in ? d.row_of[ln] : -1;
        b.nb[(size_t)q * b.Spad + j] = r;
        b.nbd[(size_t)q * b.Spad + j] = in ? d.dist[ln] : dj;   // reference reads out of bounds here (UB): use own value
        if (!(q & 1) && r >= 0) fwd |= 1 << (q >> 1);
    }
    b.dirb[j] = fwd;
    int dl[kNQ], far = 0;
    for (int q = 0; q < kNQ; ++q) {
        int o[3]; q_offset(q, o);
        long long ln = lin + o[0] * stride[0] + o[1] * stride[1] + o[2] * stride[2];
        int r = (ln >= 0 && ln < grid.nvox) ? d.row_of[ln] : -1;
        b.col[(size_t)q * b.Spad + j] = r >= 0 ? r : j;   // absent column: coefficient is 0, point at self so gathers stay in range
        dl[q] = r >= 0 ? r - j : 0;
        far = max(far, abs(dl[q]));
    }
    for (int w = 0; w < (kNQ - 1) / 2; ++w)
        b.colp[(size_t)w * b.Spad + j] = ((unsigned)dl[2 * w + 1] & 0xffffu) | ((unsigned)dl[2 * w + 2] << 16);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) far = max(far, __shfl_down(far, o, 64));
    if ((threadIdx.x & 63) == 0 && far > 0) atomicMax(reach, far);
}
void launch_band_fill(const DenseView& d, const GridP& grid, Band b, int* d_reach, hipStream_t s) {
    int g1 = (int)min((grid.nvox + kBlock - 1) / kBlock, (long long)256 * 16);
    hipLaunchKernelGGL(k_band_fill, dim3(g1), dim3(kBlock), 0, s, d, grid, b);
    if (b.S > 0) hipLaunchKernelGGL(k_band_nb, dim3((b.S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d, grid, b, d_reach);
}
__global__ void __launch_bounds__(kBlock) k_band_scatter(DenseView d, Band b) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= b.S) return;
    long long lin = b.lin[j];
    d.dist[lin] = b.dist[j];
#pragma unroll
    for (int a = 0; a < 3; ++a) { d.g[a][lin] = b.g[a][j]; d.rho[a][lin] = b.rho[a][j]; }
}
void launch_band_scatter(const DenseView& d, Band b, hipStream_t s) {
    if (b.S > 0) hipLaunchKernelGGL(k_band_scatter, dim3((b.S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d, b);
}

// Optimizer::subsampling, OptimizerAux.cpp:622-684 + VolumetricGradSdf.cpp:469-494; one thread per CHILD
// voxel so that the 8x larger output is written fully coalesced.
__global__ void __launch_bounds__(kBlock) k_upsample(DenseView src, DenseView dst, GridP g) {
    const long long nx = 2LL * g.dim[0], ny = 2LL * g.dim[1], nn = 8 * g.nvox;
    const float vs4 = (float)(0.25 * (double)g.vs);
    for (long long ls = blockIdx.x * (long long)blockDim.x + threadIdx.x; ls < nn; ls += (long long)gridDim.x * blockDim.x) {
        long long kz = ls / (nx * ny); long long rest = ls - kz * nx * ny; long long jy = rest / nx; long long ix = rest - jy * nx;
        long long lin = (ix >> 1) + (jy >> 1) * g.dim[0] + (kz >> 1) * (long long)g.dim[0] * g.dim[1];
        float d = src.dist[lin];
        if (d == g.T) {  // untouched children keep the defaults (OptimizerAux.cpp:625-631,652)
            dst.dist[ls] = g.T; dst.weight[ls] = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) { dst.g[a][ls] = 0.f; dst.rho[a][ls] = 0.5f; }
            for (int w = 0; w < src.KW; ++w) dst.vis[ls * src.KW + w] = 0;
            continue;
        }
        float gr[3] = {src.g[0][lin], src.g[1][lin], src.g[2][lin]}, gn[3];
        normalized3(gr, gn);
        float ax = (ix & 1) ? gn[0] : -gn[0], ay = (jy & 1) ? gn[1] : -gn[1], az = (kz & 1) ? gn[2] : -gn[2];
        dst.dist[ls] = d + vs4 * (ax + ay + az);
        dst.weight[ls] = src.weight[lin];
#pragma unroll
        for (int a = 0; a < 3; ++a) { dst.g[a][ls] = gr[a]; dst.rho[a][ls] = src.rho[a][lin]; }
        for (int w = 0; w < src.KW; ++w) dst.vis[ls * src.KW + w] = src.vis[lin * src.KW + w];
    }
}
void launch_upsample(const DenseView& src, const DenseView& dst, const GridP& g_old, hipStream_t s) {
    int grid = (int)min((8 * g_old.nvox + kBlock - 1) / kBlock, (long long)256 * 32);
    hipLaunchKernelGGL(k_upsample, dim3(grid), dim3(kBlock), 0, s, src, dst, g_old);
}
// FD gradient, optional updateGrad (OptimizerAux.cpp:152-160), surface point, and the Eikonal /
// Laplacian energies (Optimizer.cpp:86-119) in one pass over the band.
__global__ void __launch_bounds__(kBlock) k_derive(SweepArgs a, int update_grad) {
#pragma clang fp contract(off)
    __shared__ double red[kBlock / 64];
    if (a.gate && *a.gate == 0.0) return;       // launched speculatively behind a PCG chunk that did not finish the solve
    const Band& b = a.b;
    const int bid = vm_bid(a);
    int j = a.row0 + bid * blockDim.x + threadIdx.x;
    double en = 0, el = 0;
    if (j < a.row1) derive_row(a, j, update_grad, en, el);
    block_part_store(en, PART(a, SC_EN), red, bid);
    block_part_store(el, PART(a, SC_EL), red, bid);
}
void launch_derive(const SweepArgs& a, int update_grad, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_derive, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, update_grad);
}

// ------------------------------------------------------------------------------------------
// per-frame observation lists: band rows whose visibility bit f is set, ascending.  Visibility is static
// between band rebuilds, so the frame-major sweeps run over fully populated wavefronts instead of testing
// (and mostly rejecting) every (voxel, frame) pair.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_obs_count(Band b, int F, int row0, int row1, int* __restrict__ counts) {
    __shared__ int red[kBlock / 64];
    const int f = blockIdx.y, nch = gridDim.x;
    int cnt = 0;
    for (int it = 0; it < kObsChunk / kBlock; ++it) {
        int j = row0 + blockIdx.x * kObsChunk + it * kBlock + threadIdx.x;
        if (j < row1) cnt += (int)((b.vis[(size_t)(f >> 6) * b.Spad + j] >> (f & 63)) & 1ull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int i = 0; i < kBlock / 64; ++i) s += red[i]; counts[f * nch + blockIdx.x] = s; }
}
// work of a voxel-major block of kBlock rows = sum over its wavefronts of the LONGEST lane (a lane visits its visible frames one by one and the
// wavefront lasts as long as its busiest lane): what the distance sweep's dispatch order is sorted by (engine.hip build_band)
__global__ void __launch_bounds__(kBlock) k_block_work(Band b, int row0, int row1, int* __restrict__ work) {
    __shared__ int red[kBlock / 64];
    const int j = row0 + blockIdx.x * kBlock + threadIdx.x;
    int n = 0;
    if (j < row1) for (int w = 0; w < b.KW; ++w) n += __popcll(b.vis[(size_t)w * b.Spad + j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n = max(n, __shfl_down(n, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int i = 0; i < kBlock / 64; ++i) s += red[i]; work[blockIdx.x] = s; }
}
void launch_block_work(const Band& b, int row0, int row1, int* work, hipStream_t s) {
    const int nb = (row1 - row0 + kBlock - 1) / kBlock;
    if (nb > 0) hipLaunchKernelGGL(k_block_work, dim3(nb), dim3(kBlock), 0, s, b, row0, row1, work);
}
void launch_obs_count(const Band& b, int F, int row0, int row1, int* counts, hipStream_t s) {
    int nch = (row1 - row0 + kObsChunk - 1) / kObsChunk;
    if (nch > 0 && F > 0) hipLaunchKernelGGL(k_obs_count, dim3(nch, F), dim3(kBlock), 0, s, b, F, row0, row1, counts);
}
__global__ void __launch_bounds__(kBlock) k_obs_fill(Band b, int F, int row0, int row1, const int* __restrict__ offsets) {
    __shared__ int wsum[kBlock / 64];
    __shared__ int run_s;
    const int f = blockIdx.y, nch = gridDim.x;
    if (threadIdx.x == 0) run_s = offsets[f * nch + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int it = 0; it < kObsChunk / kBlock; ++it) {
        int j = row0 + blockIdx.x * kObsChunk + it * kBlock + threadIdx.x;
        bool flag = j < row1 && ((b.vis[(size_t)(f >> 6) * b.Spad + j] >> (f & 63)) & 1ull);
        unsigned long long m = __ballot(flag);
        int pre = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int i = 0; i < kBlock / 64; ++i) { if (i < w) woff += wsum[i]; tot += wsum[i]; }
        int run = run_s;
        if (flag) b.obs_rows[run + woff + pre] = j;
        __syncthreads();
        if (threadIdx.x == 0) run_s = run + tot;
        __syncthreads();
    }
}
void launch_obs_fill(const Band& b, int F, int row0, int row1, const int* offsets, hipStream_t s) {
    int nch = (row1 - row0 + kObsChunk - 1) / kObsChunk;
    if (nch > 0 && F > 0) hipLaunchKernelGGL(k_obs_fill, dim3(nch, F), dim3(kBlock), 0, s, b, F, row0, row1, offsets);
}
// halo of a row partition: how many rows below row0 / from row1 upward the ELL columns of the owned rows reach
// (contiguous ranges suffice because the band is sorted by linear index).  need[0] = rows below, need[1] = rows above.
// number of band rows whose (ascending) linear index is below `target`: where a z-plane starts in the band
__global__ void __launch_bounds__(kBlock) k_lower_bound(const int* __restrict__ lin, int S, int target, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > S) return;
    const bool below_prev = i == 0 || lin[i - 1] < target;
    const bool here = i == S || lin[i] >= target;
    if (below_prev && here) *out = i;
}
void launch_lower_bound(const int* lin, int S, int target, int* out, hipStream_t s) {
    hipLaunchKernelGGL(k_lower_bound, dim3((S + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, lin, S, target, out);
}
// fold per-workgroup partials into a few doubles.  `out` may be host-mapped pinned memory: the host then needs no
// D2H copy (each hipMemcpyAsync costs ~10 us of GPU idle around it), only the stream synchronisation it does anyway.
__global__ void __launch_bounds__(kBlock) k_sum_parts(const double* __restrict__ part, int PB, int nblk, SlotList slots, double* __restrict__ out, unsigned long long key, const XfTable* xf, long long xf_epoch) {
    __shared__ double red[kBlock / 64];
    double t[8];
    for (int s = 0; s < slots.n; ++s) { t[s] = block_total(part + (size_t)slots.id[s] * PB, nblk, red); __syncthreads(); }
    if (threadIdx.x < 64) {
        if (xf) fold_exchange(xf, xf_epoch, slots.n, t);      // multi-rank: the sums over all slabs (device_common.h; the whole first wavefront takes part)
        if (threadIdx.x == 0) {
            for (int s = 0; s < slots.n; ++s) mbox_put(out, slots.n, s, t[s], key);
            mbox_commit(key);
        }
    }
}
void launch_sum_parts(const double* part, int PB, int nblk, const SlotList& slots, double* out, unsigned long long key, hipStream_t s, const XfTable* xf, long long xf_epoch) {
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(kBlock), 0, s, part, PB, nblk, slots, out, key, xf, xf_epoch);
}
// sum of columns (col, col+1) over the F frame-accumulator rows -> out[0..1]
__global__ void __launch_bounds__(kBlock) k_frame_cols(const double* __restrict__ frame, int F, int col, double* __restrict__ out, unsigned long long key) {
    __shared__ double red[kBlock / 64];
    double a = 0, b = 0;
    for (int f = threadIdx.x; f < F; f += blockDim.x) { a += frame[(size_t)f * kFrameRow + col]; b += frame[(size_t)f * kFrameRow + col + 1]; }
    a = wave_sum(a); b = wave_sum(b);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = a;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < kBlock / 64; ++i) t += red[i]; mbox_put(out, 2, 0, t, key); }
    __syncthreads();
    if (lane == 0) red[w] = b;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < kBlock / 64; ++i) t += red[i]; mbox_put(out, 2, 1, t, key); mbox_commit(key); }
}
void launch_frame_cols(const double* frame, int F, int col, double* out, unsigned long long key, hipStream_t s) {
    hipLaunchKernelGGL(k_frame_cols, dim3(1), dim3(kBlock), 0, s, frame, F, col, out, key);
}
// flush marker: everything queued before it has completed when the host reads `v` from the mapped slot (engine.hip: flush)
__global__ void k_marker(double* p, double v) { *p = v; __threadfence_system(); }
void launch_marker(double* p, double v, hipStream_t s) { hipLaunchKernelGGL(k_marker, dim3(1), dim3(1), 0, s, p, v); }
__global__ void k_zero_f64(double* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.0; }
// multi-rank scalar read-backs: all-reduced values from the device shadow into their (host-mapped) mailbox slots
__global__ void k_copy_segs(const double* __restrict__ src, double* __restrict__ dst, CopySegs segs) {
    for (int q = 0; q < segs.n; ++q)
        for (unsigned i = threadIdx.x; i < segs.len[q]; i += blockDim.x) mbox_put(dst + segs.off[q], (int)segs.len[q], (int)i, src[segs.off[q] + i], segs.key[q]);
    __threadfence_system();
}
void launch_copy_segs(const double* src, double* dst, const CopySegs& segs, hipStream_t s) { if (segs.n > 0) hipLaunchKernelGGL(k_copy_segs, dim3(1), dim3(64), 0, s, src, dst, segs); }
void launch_zero_f64(double* p, int n, hipStream_t s) { if (n > 0) hipLaunchKernelGGL(k_zero_f64, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), dim3(256), 0, s, p, n); }

}  // namespace psg
