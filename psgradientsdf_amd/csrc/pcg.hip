// pcg.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Gradient-SDF photometric-stereo hot path:
// the fused Jacobi-PCG of the distance system and the distance update.  No CUDA compatibility layer, no other back end.
// Shared device helpers: device_common.h; the launchers are declared in engine.h.
#include "device_common.h"

namespace psg {

__device__ __forceinline__ float pcg_threshold(float rhsNorm2) { return fmaxf(FLT_EPSILON * FLT_EPSILON * rhsNorm2, FLT_MIN); }

// Jacobi-PCG with Eigen::ConjugateGradient semantics: x0 = 0, threshold = max(eps^2 |b|^2, FLT_MIN), scalar
// recurrences in float, dot products accumulated in double.  ONE kernel and ONE reduction per CG iteration:
// Pass k (t = A p_k) also reduces, over the same rows,
//     P = p.t   B = sum inv r t   C = sum inv t^2   D = sum r t   E = sum t^2   Z = r.z   R = |r|^2      (r = r_k, double)
// from which the NEXT kernel derives alpha_k = Z / P and, without ever reducing r_{k+1} = r_k - alpha t separately,
//     r_{k+1}.z_{k+1} = Z - 2 alpha B + alpha^2 C        |r_{k+1}|^2 = R - 2 alpha D + alpha^2 E
// (Z and R are re-summed from the vectors every pass, so the expansions never chain and the cancellation costs at most the
// digits of one pass's residual drop, taken from a double).  The vector updates x += alpha p, r -= alpha t, z = inv r,
// p = z + beta p are applied lazily in float exactly as the reference does them: kernel k first finishes pass k-1 for its
// own rows, and re-derives r_k, z_k, p_k of every gathered column from that column's record {r, t, p, inv} of pass k-1
// (ONE 16-byte gather per column; records double-buffered because neighbours still read the old ones).
//   fs (device doubles): [0] |b|^2   [1] done (0 = running, else stopping kernel + 1)   [2] 1 if it stopped because it converged
//   part: [2 parity][kCgfSums][kPcgMaxBlocks] per-workgroup partial sums, summed in a fixed order by every workgroup of
//         the next kernel (deterministic, no atomics)
//   mb  : slot of THIS kernel (mapped host memory on one GPU): k = 0 -> |b|^2, k > 0 -> |r|^2 after pass k-1
// Multi-rank (a.ext != nullptr): a 1-workgroup kernel folds the partials of a pass into a.ext[0..6], the host program
// all-reduces them over the ranks, and the next kernel reads the global sums from a.ext instead of the partials; the
// records of the halo rows are exchanged before each pass.
// ------------------------------------------------------------------------------------------

// n sums at once, identical in every thread of every workgroup.  All loads of a thread are issued before the first use
// (fixed trip count, predicated): ONE memory round trip however many partials there are -- a dynamic-trip loop made it three.
template <int N> struct PartLoads { double ld[kCgfMaxBlocks / kBlock][N]; };
template <int N>
__device__ __forceinline__ void block_total_issue(double* const* src, int n, PartLoads<N>& pl) {
#pragma unroll
    for (int j = 0; j < kCgfMaxBlocks / kBlock; ++j) {
        const int i = threadIdx.x + j * kBlock;
#pragma unroll
        for (int q = 0; q < N; ++q) pl.ld[j][q] = i < n ? src[q][i] : 0.0;
    }
}
template <int N>
__device__ __forceinline__ void block_total_finish(const PartLoads<N>& pl, double* red /*[8 * kBlock/64]*/, double* out) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = 0.0;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        v[q] = pl.ld[0][q];
#pragma unroll
        for (int j = 1; j < kCgfMaxBlocks / kBlock; ++j) v[q] += pl.ld[j][q];
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if constexpr (N > 2) {
        double t0, t1; wave_sum8(v, t0, t1);
        __syncthreads();
        wave_sum8_store<kBlock / 64>(t0, t1, red, w);
    } else {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = wave_sum(v[q]);
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < N; ++q) red[q * (kBlock / 64) + w] = v[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < kBlock / 64; ++i) s += red[q * (kBlock / 64) + i];
        out[q] = s;
    }
}
template <int N>
__device__ __forceinline__ void block_total_n(double* const* src, int n, double* red /*[N * kBlock/64]*/, double* out) {
    PartLoads<N> pl;
    block_total_issue<N>(src, n, pl);
    block_total_finish<N>(pl, red, out);
}
template <int N>
__device__ __forceinline__ void block_part_store_n(const double* vin, double* const* dst, double* red /*[8 * kBlock/64]*/, int slot) {
    static_assert(N > 2 && N <= 8, "reduce-scatter of up to eight sums");
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = q < N ? vin[q] : 0.0;
    const int w = threadIdx.x >> 6;
    double t0, t1; wave_sum8(v, t0, t1);
    __syncthreads();
    wave_sum8_store<kBlock / 64>(t0, t1, red, w);
    __syncthreads();
    if (threadIdx.x < N) {
        double s = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[threadIdx.x * (kBlock / 64) + i];
        dst[threadIdx.x][slot] = s;
    }
}

__global__ void __launch_bounds__(kBlock) k_cgf_init(SweepArgs a, double* fs, double* part) {
    __shared__ double red[kBlock / 64];
    const Band& b = a.b;
    double bb = 0;
    for (int i = a.row0 + blockIdx.x * blockDim.x + threadIdx.x; i < a.row1; i += gridDim.x * blockDim.x) {
        float dg = b.H[i];
        if (a.damping != 0.0f) dg += a.damping * dg;
        const float inv = dg != 0.f ? 1.0f / dg : 1.0f;
        const float r = b.rhs[i];
        b.x[i] = 0.f;
        b.rec[1][i] = make_float4(r, 0.f, 0.f, inv);      // {r_0, t_{-1} = 0, p_{-1} = 0, inv}: read by kernel 0
        bb += (double)r * (double)r;
    }
    block_part_store(bb, fpart(part, -1, 6), red);
    if (blockIdx.x == 0 && threadIdx.x == 0) { fs[1] = 0.0; fs[2] = 0.0; }
}
void launch_cgf_init(const SweepArgs& a, double* fs, double* part, int G, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_cgf_init, dim3(G), dim3(kBlock), 0, s, a, fs, part);
}
// kernel k: finishes pass k-1 (k > 0), decides convergence, then runs pass k unless k == kmax (the iteration cap).
//
// The pass is a chain of memory round trips (column indices -> 16-byte record gathers -> reduction of the previous pass's
// partials), not a bandwidth problem, so the kernel is arranged to need nothing from the reduction until the very end:
// t = A p_k with p_k[c] = inv_c (r_c - alpha t_c) + beta p_c is LINEAR in the three gathered fields,
//     t = A1 - alpha A2 + beta A3,   A1 = sum_c h_c inv_c r_c,  A2 = sum_c h_c inv_c t_c,  A3 = sum_c h_c p_c   (double),
// so the three sums are accumulated as the gathers arrive, before alpha and beta exist, and the records never have to be
// kept in registers.  (p_k of a NEIGHBOUR is therefore not rounded to float before it enters the product, unlike Eigen's
// explicit vector; the row's own r, z, p, x are updated in float exactly as the reference does.  DESIGN.md §2, deviation 3.)
// All 19 ELL columns are treated alike: at the band sizes of this path 61 % of the rows and every wavefront use the 6
// columns that only backward-forced stencils produce.
struct CgfRow { double A1, A2, A3; float4 me; float x; int i; bool live; };
// The gathers of a thread's rows are issued in two batches (10 + 9 columns): all 57 records of 3 rows at once would need
// 228 registers.  The second batch is in flight
// while the caller reduces the previous pass's partials; cgf_rows_finish folds it in afterwards.
constexpr int kCgfB1 = 10;   // columns of the first gather batch (the split is about registers, not about which columns are common)
template <int R> struct CgfPending { float h[R][kNQ - kCgfB1]; float4 o[R][kNQ - kCgfB1]; };
// The streamed loads go through buffer instructions (scalar resource + ONE 32-bit lane offset per row, the plane offset
// q * Spad in the scalar offset operand): with flat 64-bit addresses the 38 streamed loads of a row cost two address
// registers each and the kernel spilled.
// (The record gathers stay flat loads: this compiler narrows `raw.ptr.buffer.load.v4i32` to a one-dword load.)
template <int R> struct CgfStream { float h[R][kNQ]; int c[R][kNQ]; };
template <int R, bool C16>
__device__ __forceinline__ void cgf_stream_issue(const Band& b, int i0, int stride, int row1, CgfRow* w, CgfStream<R>& sl) {
    const int plane = b.Spad * 4;              // bytes of one ELL column plane
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc((void*)b.H, 0, kNQ * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rC = C16 ? __builtin_amdgcn_make_buffer_rsrc((void*)b.colp, 0, (kNQ - 1) / 2 * plane, 0x00020000)
                                          : __builtin_amdgcn_make_buffer_rsrc((void*)b.col, 0, kNQ * plane, 0x00020000);
    auto& h = sl.h; auto& c = sl.c;
    // round trip 1: everything addressed by the rows themselves, for ALL rows of the thread
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int i = i0 + u * stride;
        w[u].i = i; w[u].live = i < row1;
        const int ii = w[u].live ? i : row1 - 1;
#pragma unroll
        for (int q = 0; q < kNQ; ++q) h[u][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rH, ii * 4, q * plane, 0));
        c[u][0] = ii;
        if (C16) {   // 9 words of two 16-bit deltas
#pragma unroll
            for (int wd = 0; wd < (kNQ - 1) / 2; ++wd) {
                const int pk = (int)__builtin_amdgcn_raw_buffer_load_b32(rC, ii * 4, wd * plane, 0);
                c[u][2 * wd + 1] = ii + ((pk << 16) >> 16);
                c[u][2 * wd + 2] = ii + (pk >> 16);
            }
        } else {
#pragma unroll
            for (int q = 1; q < kNQ; ++q) c[u][q] = (int)__builtin_amdgcn_raw_buffer_load_b32(rC, ii * 4, q * plane, 0);
        }
        w[u].x = b.x[ii];
    }
}
template <int R>
__device__ __forceinline__ void cgf_gather(const float4* __restrict__ rin, float damping, CgfRow* w, CgfStream<R>& sl, CgfPending<R>& pend) {
    auto& h = sl.h; auto& c = sl.c;
    // round trip 2: the records of the first batch of columns of every row, folded into the three sums as they arrive
    float4 o[R][kCgfB1];
#pragma unroll
    for (int u = 0; u < R; ++u) {
#pragma unroll
        for (int q = 0; q < kCgfB1; ++q) o[u][q] = rin[c[u][q]];
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        if (damping != 0.0f) h[u][0] += damping * h[u][0];
        double A1 = 0, A2 = 0, A3 = 0;
#pragma unroll
        for (int q = 0; q < kCgfB1; ++q) {
            const double hq = (double)h[u][q], iv = (double)o[u][q].w;
            A1 += hq * (iv * (double)o[u][q].x); A2 += hq * (iv * (double)o[u][q].y); A3 += hq * (double)o[u][q].z;
        }
        w[u].A1 = A1; w[u].A2 = A2; w[u].A3 = A3; w[u].me = o[u][0];
    }
    __builtin_amdgcn_sched_barrier(0);        // keep the second batch behind the first one's consumption (register budget)
    // round trip 3 (overlaps the caller's reduction): the remaining columns
#pragma unroll
    for (int u = 0; u < R; ++u) {
#pragma unroll
        for (int q = kCgfB1; q < kNQ; ++q) { pend.h[u][q - kCgfB1] = h[u][q]; pend.o[u][q - kCgfB1] = rin[c[u][q]]; }
    }
}
template <int R>
__device__ __forceinline__ void cgf_rows_finish(CgfRow* w, const CgfPending<R>& pend) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
        double A1 = w[u].A1, A2 = w[u].A2, A3 = w[u].A3;
#pragma unroll
        for (int q = 0; q < kNQ - kCgfB1; ++q) {
            const double hq = (double)pend.h[u][q], iv = (double)pend.o[u][q].w;
            A1 += hq * (iv * (double)pend.o[u][q].x); A2 += hq * (iv * (double)pend.o[u][q].y); A3 += hq * (double)pend.o[u][q].z;
        }
        w[u].A1 = A1; w[u].A2 = A2; w[u].A3 = A3;
    }
}
template <int kCgfRows, int kMinWaves, bool C16>
__global__ void __launch_bounds__(kBlock, kMinWaves) k_cgf_pass(SweepArgs a, double* fs, double* part, int k, int kmax, double* mb, int ab) {   // ab: timing ablations (tools/), 0 in production
    __shared__ double red[8 * kBlock / 64];
    const Band& b = a.b;
    long long* ts = (long long*)(fs + 16) + (size_t)blockIdx.x * 8;   // ab & 1024: stage timestamps of every workgroup
#define CGF_STAMP(j) do { if ((ab & 1024) && threadIdx.x == 0) { ts[j] = clock64(); if (j == 0) ts[6] = wall_clock64(); if (j == 4) ts[7] = wall_clock64(); } } while (0)
    CGF_STAMP(0);
    const float4* __restrict__ rin = b.rec[(k + 1) & 1];
    float4* __restrict__ rout = b.rec[k & 1];
    const int stride = gridDim.x * blockDim.x;
    // XCD-aware: workgroups are dealt to the 8 XCDs round-robin; the LOGICAL id gives every XCD a contiguous range of row blocks per trip, so the records a row
    // gathers (its z neighbours sit one plane of band rows away: ~10 row blocks at 512^3) are fetched into ONE L2 instead of several (round 6: the per-pass
    // kernel's counter traffic was 1.5 x algorithmic at the 512^3 band, the surplus = the record planes fetched by three XCDs each).  Rows AND partial-sum
    // slots follow the logical id, so the sums -- added in slot order -- do not depend on the mapping.
    const int lb = (a.xcd_map & 128) ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;      // (PSGSDF_XCD_MAP bit 7)
    int i0 = a.row0 + lb * blockDim.x + threadIdx.x;
    CgfRow w[kCgfRows]; CgfPending<kCgfRows> pend; CgfStream<kCgfRows> sl;
    const double stopped = fs[1];
    cgf_stream_issue<kCgfRows, C16>(b, i0, stride, a.row1, w, sl);
    cgf_gather<kCgfRows>(rin, a.damping, w, sl, pend);

    CGF_STAMP(1);
    float alpha_prev = 0.f, beta = 0.f, rr_cur, rhsNorm2;
    if (ab & 1) { alpha_prev = 0.01f; beta = 0.5f; rr_cur = 1.f; rhsNorm2 = 1.f; cgf_rows_finish<kCgfRows>(w, pend); }
    else if (!(ab & 16) && stopped != 0.0 && stopped <= (double)k) return;   // stopped by an EARLIER kernel of this solve (kernel j writes j + 1)
    else if (k == 0) {
        double bb;
        if (a.ext) bb = a.ext[0];
        else if (a.pcg_init_blocks > 0) bb = block_total(fpart(part, -1, 6), a.pcg_init_blocks, red);   // written by the assembly kernel's grid
        else { double* src[1] = {fpart(part, -1, 6)}; block_total_n<1>(src, gridDim.x, red, &bb); }
        cgf_rows_finish<kCgfRows>(w, pend);
        rhsNorm2 = (float)bb; rr_cur = rhsNorm2;
        if (lb == 0 && threadIdx.x == 0) { fs[0] = bb; mb[0] = bb; __threadfence_system(); }   // mb may be host-mapped: the host watches it
    } else {
        double* src[kCgfSums]; double t[kCgfSums];
#pragma unroll
        for (int q = 0; q < kCgfSums; ++q) src[q] = fpart(part, k - 1, q);
        if (a.ext) {
#pragma unroll
            for (int q = 0; q < kCgfSums; ++q) t[q] = a.ext[q];
            cgf_rows_finish<kCgfRows>(w, pend);
        } else {   // the partial sums of the previous pass are requested while the second gather batch is still in flight
            PartLoads<kCgfSums> pl;
            block_total_issue<kCgfSums>(src, gridDim.x, pl);
            cgf_rows_finish<kCgfRows>(w, pend);
            block_total_finish<kCgfSums>(pl, red, t);
        }
        rhsNorm2 = (float)fs[0];
        const float rz_old = (float)t[5];
        alpha_prev = rz_old / (float)t[0];                // alpha = absNew / p.dot(tmp)
        const double al = (double)alpha_prev;
        const float rz_cur = (float)(t[5] - 2.0 * al * t[1] + al * al * t[2]);
        rr_cur = (float)(t[6] - 2.0 * al * t[3] + al * al * t[4]);
        beta = rz_cur / rz_old;                            // beta = absNew / absOld
        if (lb == 0 && threadIdx.x == 0) { mb[0] = (double)rr_cur; __threadfence_system(); }
    }
    CGF_STAMP(2);
    const bool rhs_zero = rhsNorm2 == 0.f;
    const bool stop = !(ab & 16) && (rhs_zero || k == kmax || (k > 0 && rr_cur < pcg_threshold(rhsNorm2)));
    if (stop && lb == 0 && threadIdx.x == 0) { fs[1] = (double)(k + 1); fs[2] = (rhs_zero || sqrt((double)rr_cur / (double)rhsNorm2) <= (double)FLT_EPSILON) ? 1.0 : 0.0; }   // fs[2]: Eigen's info() == Success, the host's rule (loop.hip: pcg_solve)
    double s[kCgfSums];
#pragma unroll
    for (int q = 0; q < kCgfSums; ++q) s[q] = 0;
    auto rows_out = [&](const CgfRow* wr) {
#pragma unroll
        for (int u = 0; u < kCgfRows; ++u) {
            const CgfRow& r = wr[u];
            const float4 me = r.me;
            // finish pass k-1 for the own row: x += alpha p ; residual -= alpha tmp
            if (r.live && k > 0 && !(ab & 32)) b.x[r.i] = r.x + alpha_prev * me.z;
            if (stop || !r.live) continue;
            const float r_i = me.x - alpha_prev * me.y;
            const float z_i = me.w * r_i;
            const float p_i = z_i + beta * me.z;
            const float t = (float)(r.A1 - (double)alpha_prev * r.A2 + (double)beta * r.A3);
            if (!(ab & 64)) rout[r.i] = make_float4(r_i, t, p_i, me.w);
            const double rd = (double)r_i, td = (double)t, iv = (double)me.w;
            s[0] += (double)p_i * td; s[1] += iv * rd * td; s[2] += iv * td * td; s[3] += rd * td; s[4] += td * td;
            s[5] += rd * (double)z_i; s[6] += rd * rd;
        }
    };
    rows_out(w);
    i0 += kCgfRows * stride;
    while (i0 - (int)threadIdx.x < a.row1) {          // workgroup-uniform
        cgf_stream_issue<kCgfRows, C16>(b, i0, stride, a.row1, w, sl);
        cgf_gather<kCgfRows>(rin, a.damping, w, sl, pend);
        cgf_rows_finish<kCgfRows>(w, pend);
        rows_out(w);
        i0 += kCgfRows * stride;
    }
    CGF_STAMP(3);
    if (stop || (ab & 4)) return;
    double* dst[kCgfSums];
#pragma unroll
    for (int q = 0; q < kCgfSums; ++q) dst[q] = fpart(part, k, q);
    block_part_store_n<kCgfSums>(s, dst, red, lb);
    CGF_STAMP(4);
#undef CGF_STAMP
}
void launch_cgf_pass(const SweepArgs& a, double* fs, double* part, int G, int rows, int k, int kmax, double* mb, hipStream_t s, int ablate) {
    if (a.row1 <= a.row0) return;
    // rows per thread in flight at once <-> registers <-> resident workgroups per CU (launch bound = waves per SIMD).
    // One row per thread (<= 128 VGPRs, 4 waves per SIMD) is the production shape; two rows spill at 3 waves per SIMD and are
    // kept for the timing tool only.
    if (rows >= 2) hipLaunchKernelGGL((k_cgf_pass<2, 2, false>), dim3(G), dim3(kBlock), 0, s, a, fs, part, k, kmax, mb, ablate);
    else if (a.b.col16) hipLaunchKernelGGL((k_cgf_pass<1, 4, true>), dim3(G), dim3(kBlock), 0, s, a, fs, part, k, kmax, mb, ablate);
    else hipLaunchKernelGGL((k_cgf_pass<1, 4, false>), dim3(G), dim3(kBlock), 0, s, a, fs, part, k, kmax, mb, ablate);
}

// ------------------------------------------------------------------------------------------
// The whole solve as ONE persistent kernel (single-rank contexts whose band fits the register file).
//
// A per-pass launch re-streams the 19 ELL coefficients and 9 index words of every row (152 B/row, 51 MB at the 256^3 band:
// 9.7 of the pass's 14.3 us at the ~5.3 TB/s the L2-miss path sustains; profiles/r01_notes.md) although the matrix never
// changes during a solve.  Here every thread loads the coefficients and column deltas of its R rows ONCE into registers
// (28 dwords per row; 38 MB of the chip's 128 MB of VGPRs at 256^3), keeps its own {r, t, p, x} there too, and loops over
// the passes; per pass it gathers the 18 neighbour records (16 B each, L2-resident: 5.4 MB per buffer), computes exactly what
// k_cgf_pass computes (same recurrences, same seven sums, same fixed-order reduction, same stop rule) and meets the other
// workgroups at a device-wide all-gather of the seven per-workgroup sums:
//   publish : records with 16-byte write-through (sc1) stores -> every wave s_waitcnt vmcnt(0) -> __syncthreads -> seven 8-byte
//             granules {sum with a 2-bit pass tag in its two lowest mantissa bits}, one sc1 store each: data and tag arrive
//             together, no flag, no ordering between the seven.  (Plain stores + an agent-scope release fence measured 4.8 us
//             for drain + write-back per pass; profiles/r02_notes.md.)
//   wait    : in two steps.  The gathers only need the records of the few workgroups that are this one's neighbours in band order,
//             and -- t = A1 - alpha A2 + beta A3 -- neither alpha nor beta: a few threads poll those neighbours' tags, lane 0 does an
//             agent-scope acquire, and the gathers run while the sums of the far workgroups are still on their way.  Then thread t
//             polls the seven granules of workgroup t until their tags name the pass, and the fixed-order sum over workgroups that
//             every workgroup computes identically gives alpha, beta, |r|^2 and the stop decision -- the same everywhere, so all
//             workgroups leave in the same pass.  The records double-buffer exactly as in k_cgf_pass: a workgroup writes its pass-k
//             records only after it has seen EVERY workgroup's pass k-1 sums, i.e. after every reader of the old ones is done.
// One workgroup of 512 threads per CU at most (grid <= the CU count, so all workgroups are co-resident); logical workgroup ids
// are remapped so that every XCD owns a contiguous range of rows (gathers stay in that XCD's L2).  Every wait is bounded:
// a workgroup that sees nothing for ~1 s raises the abort flag and every workgroup leaves (the host reports PSGSDF_ERR_DEVICE).
// Two mantissa bits of the double sums carry the tag: 4e-16 relative, far below the float the sums are rounded to.
// ------------------------------------------------------------------------------------------
constexpr int kSolveThreads = 512;            // one workgroup of 8 waves per CU: 2 waves per SIMD at <= 256 VGPRs (16 waves at 128 VGPRs spilled: 21 us per pass)
constexpr int kSolveMaxBlocks = 256;          // one per CU
constexpr int kSolveMaxRows = 4;              // rows per thread: R x 38 KB of coefficients in the CU's 160 KB of LDS
typedef float v4f_t __attribute__((ext_vector_type(4)));
// 16-byte write-through store (sc1): the record reaches memory without a later L2 write-back, so publishing needs no release fence
// (MI355X_MICROARCH.md "publish-large": 3.0 vs 8.2 us); the trailing s_nop keeps the assembler's hazard rules for inline VMEM
__device__ __forceinline__ void store16_sc1(float4* p, const float4& v) {
    const v4f_t d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
}
// the same towards another GPU's memory (a neighbour slab's halo rows, IPC-mapped over xGMI): write-through at SYSTEM scope
__device__ __forceinline__ void store16_sys(float4* p, const float4& v) {
    const v4f_t d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ double gran_tag(double v, unsigned tag) { return __longlong_as_double((long long)(((unsigned long long)__double_as_longlong(v) & ~3ull) | tag)); }
__device__ __forceinline__ unsigned gran_tag_of(double v) { return (unsigned)((unsigned long long)__double_as_longlong(v) & 3ull); }
// cross-rank words (multi-rank solve): the tag is 16 bits wide, (epoch of the solve << 2 | pass tag) -- 2^-36 of the value, still far below the float the
// sums are rounded to.  No word of an earlier solve can match, so the regions are never cleared between solves (engine.h XrArgs).
__device__ __forceinline__ double xr_tag(double v, unsigned tag) { return __longlong_as_double((long long)(((unsigned long long)__double_as_longlong(v) & ~0xffffull) | tag)); }
__device__ __forceinline__ unsigned xr_tag_of(double v) { return (unsigned)((unsigned long long)__double_as_longlong(v) & 0xffffull); }

// One ELL row of the distance system accumulated in REGISTERS: the contributions of assemble_row (dist.hip), in its order.  There the column
// of a contribution is a run-time index into an LDS table; here every (contributor, block column) pair has at most two possible
// columns, known at compile time, and the one the contributor's stencil direction does not select receives an exact + 0.0 -- the
// nineteen sums are bit-identical to the LDS version's.
__device__ __forceinline__ bool ell_valid(const int* o) {
    const int nz = (o[0] != 0) + (o[1] != 0) + (o[2] != 0);
    return nz <= 2 && o[0] >= -1 && o[0] <= 1 && o[1] >= -1 && o[1] <= 1 && o[2] >= -1 && o[2] <= 1;
}
__device__ __forceinline__ void assemble_row_regs(const SweepArgs& a, int i, double (&acc)[kNQ], double& rhs) {
    const Band& b = a.b;
    int jr[7]; jr[0] = i;
#pragma unroll
    for (int c = 1; c < 7; ++c) { const int ax = (c - 1) >> 1; const bool upper = (c - 1) & 1; jr[c] = b.nb[(size_t)(2 * ax + (upper ? 0 : 1)) * b.Spad + i]; }
    int db[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) db[c] = b.dirb[jr[c] >= 0 ? jr[c] : i];
    bool use[7]; use[0] = true;
#pragma unroll
    for (int c = 1; c < 7; ++c) { const int ax = (c - 1) >> 1; const bool upper = (c - 1) & 1; use[c] = jr[c] >= 0 && !(upper && ((db[c] >> ax) & 1)); }
    float val[7][4], gr[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const int sl = c == 0 ? 0 : ((c - 1) >> 1) + 1;
        const int jrow = use[c] ? jr[c] : i;
        gr[c] = b.blk[(size_t)(10 + sl) * b.Spad + jrow];
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) val[c][bq] = b.blk[(size_t)sym4(sl, bq) * b.Spad + jrow];
    }
#pragma unroll
    for (int q = 0; q < kNQ; ++q) acc[q] = 0.0;
    rhs = 0.0;
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const int ax = c == 0 ? 0 : (c - 1) >> 1; const bool upper = c > 0 && ((c - 1) & 1);
        int coff[3] = {0, 0, 0};
        if (c > 0) coff[ax] = upper ? 1 : -1;
        rhs += use[c] ? (double)gr[c] : 0.0;
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            const double v = use[c] ? (double)val[c][bq] : 0.0;
            if (bq == 0) { acc[q_of(coff)] += v; continue; }
            const bool fwd = (db[c] >> (bq - 1)) & 1;
            int oP[3] = {coff[0], coff[1], coff[2]}, oM[3] = {coff[0], coff[1], coff[2]};
            oP[bq - 1] += 1; oM[bq - 1] -= 1;
            if (ell_valid(oP)) acc[q_of(oP)] += fwd ? v : 0.0;
            if (ell_valid(oM)) acc[q_of(oM)] += fwd ? 0.0 : v;
        }
    }
}

// MR (multi-rank, z-slabs): the same kernel on every rank's slab, meeting the other ranks in two places.  (1) The records of the rows next to a cut
// are ALSO written into the neighbour's halo rows (system-scope write-through stores through the IPC mapping), followed -- once the wave's stores
// have drained -- by a tag in the neighbour's mailbox region; the neighbour's workgroups whose gathers reach across the cut wait for those tags
// next to the tags of their local neighbours.  (2) The seven sums: every workgroup first obtains the rank's sums from the local all-gather exactly
// as on one GPU; workgroup 0 then writes them as tagged granules into EVERY rank's region, and every workgroup sums the R rank granules of its
// own region in rank order -- the same bits on every rank, one more hop per pass instead of a kernel boundary, a fold kernel and an RCCL
// all-reduce (loop.hip: the per-pass path stays as the fallback and as the reference of the tests).
template <int R, bool ASM, bool MR>
__global__ void __launch_bounds__(kSolveThreads, 2) k_cgf_solve(SweepArgs a, double* fs, double* gran, int rows_per_wg, int kmax, double* mb, unsigned long long mb_key, int force_passes, XrArgs xr) {
    __shared__ double red[8 * kSolveThreads / 64];
    __shared__ int s_abort;
    __shared__ int s_foreign;      // a neighbour workgroup (in band order) runs on another XCD
    const Band& b = a.b;
    const int G = gridDim.x, tid = threadIdx.x;
    // The XCD this workgroup REALLY runs on (HW_REG_XCC_ID): if every workgroup that gathers from its rows sits on the same XCD, its records can stay
    // in that XCD's L2 (plain stores) instead of going through memory with write-through stores, which drop the line and make every reader fetch it at the
    // cross-XCD rate (MI355X_MICROARCH.md: same-XCD hand-offs 1.7x; r02 notes section 8 measured 9.5-9.6 vs 9.9-10.5 us per pass but would not rely on an
    // ASSUMED placement).  Here the neighbours tell each other where they are: the 'records are out' flag of the prologue carries 1 + the XCD id.
    const int my_xcc = (int)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
    bool xcd_local = false;
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs; logical block lb gives XCD x the contiguous blocks [x G/8, (x+1) G/8)
    const int lb = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int plane = b.Spad * 4;
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc((void*)b.H, 0, kNQ * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)b.colp, 0, (kNQ - 1) / 2 * plane, 0x00020000);
    // MR: which of this workgroup's rows the neighbours hold as halo, and whose tags it has to wait for
    const int own_n = a.row1 - a.row0;
    const int wg_first = lb * rows_per_wg, wg_last = min(own_n, wg_first + rows_per_wg) - 1;      // (relative to row0)
    const bool cut_lo = MR && xr.give_lo > 0 && wg_first < xr.give_lo && wg_first < own_n;          // owns rows of the lower neighbour's upper halo
    const bool cut_hi = MR && xr.give_hi > 0 && wg_last >= own_n - xr.give_hi && wg_first < own_n;
    const int hi_first_wg = MR ? max(0, own_n - xr.give_hi) / rows_per_wg : 0;                      // first workgroup that owns such rows
    double* const xr_me = MR ? xr.region[xr.rank] : nullptr;
    const unsigned etag0 = MR ? (xr.epoch & kXrEpochMask) << 2 : 0u;      // cross-rank tags: (epoch << 2 | pass tag)
    constexpr int kLocalSpins = MR ? (1 << 24) : (1 << 22);               // MR: a local neighbour may itself be waiting for a late RANK -- the local waits must not expire first
    // the neighbour's mailbox slots this workgroup tags: it is the (lb)-th cut-side workgroup towards the lower neighbour, the (lb - hi_first_wg)-th towards the upper one
    auto peer_tag = [&](int buf, double v) {
        if (cut_lo && lb < kXrPeerTags) __hip_atomic_store(xr.region[xr.rank - 1] + kXrPtag + (1 * 3 + buf) * kXrPeerTags + lb, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);              // (side 1 of the lower rank = tags of its UPPER neighbour)
        if (cut_hi && lb - hi_first_wg < kXrPeerTags) __hip_atomic_store(xr.region[xr.rank + 1] + kXrPtag + (0 * 3 + buf) * kXrPeerTags + (lb - hi_first_wg), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto push_record = [&](int buf, int rel, const float4& rec) {      // rel = row - row0
        if (MR && rel < xr.give_lo) store16_sys(xr.lo_rec[buf] + rel, rec);
        if (MR && rel >= own_n - xr.give_hi) store16_sys(xr.hi_rec[buf] + (rel - (own_n - xr.give_hi)), rec);
    };
    // ---- once: the rows of this thread.  The 19 coefficients of a row live in LDS ([row slot][column][thread]: conflict-free, R x 38 KB of
    // the CU's 160 KB), the 9 index words and the row's own state in registers.
    float* hs = (float*)psg_dyn_smem;
    unsigned cp[R][(kNQ - 1) / 2]; float4 me[R]; float x[R]; int row[R]; bool live[R];
    if (ASM && a.fold.n != 0 && blockIdx.x == 0) {
        // no k_assemble in front of this kernel to fold the sums the distance sweep left pending (device_common.h fold_pending): done here,
        // by the first 256 threads in that function's order (the same bits as in any 256-thread kernel)
        double ftot[4] = {0.0, 0.0, 0.0, 0.0};
        for (int sl = 0; sl < a.fold.n; ++sl) {
            const double* part = PART(a, a.fold.id[sl]);
            double v = 0;
            if (tid < kBlock) for (int i = tid; i < a.fold.nblk; i += kBlock) v += part[i];
            v = wave_sum(v);
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = v;
            __syncthreads();
            if (tid == 0) { double t = 0; for (int i = 0; i < kBlock / 64; ++i) t += red[i]; ftot[sl] = t; }
        }
        if (tid < 64) {      // (the exchange between the ranks takes the whole first wavefront; lane 0 holds the slab's sums)
            if (a.fold.xf) fold_exchange(a.fold.xf, a.fold.xf_epoch, a.fold.n, ftot);      // multi-rank: the sums over all slabs
            if (tid == 0) {
                for (int sl = 0; sl < a.fold.n; ++sl) mbox_put(a.fold.out, a.fold.n, sl, ftot[sl], a.fold.key);
                mbox_commit(a.fold.key);
            }
        }
        __syncthreads();
    }
    double bb_thread = 0.0;                     // ASM: |b|^2 of this thread's rows (summed over the device with the sums of pass 0)
#pragma unroll
    for (int u = 0; u < R; ++u) {
        // the band is dealt evenly to ALL workgroups (rows_per_wg each, a multiple of 64): the last row slot of a thread is only partly used
        const int i = a.row0 + lb * rows_per_wg + u * kSolveThreads + tid;
        live[u] = u * kSolveThreads + tid < rows_per_wg && i < a.row1; row[u] = live[u] ? i : a.row1 - 1;
        if (ASM) {
            // the row's matrix entries straight from the voxel blocks of the distance sweep (no k_assemble, no H in memory)
            double acc[kNQ], rhs;
            assemble_row_regs(a, row[u], acc, rhs);
#pragma unroll
            for (int q = 0; q < kNQ; ++q) {
                float hv = (float)acc[q];
                if (q == 0 && a.damping != 0.0f) hv += a.damping * hv;
                hs[(u * kNQ + q) * kSolveThreads + tid] = hv;
            }
            float dg = (float)acc[0];
            if (a.damping != 0.0f) dg += a.damping * dg;
            const float inv = dg != 0.f ? 1.0f / dg : 1.0f;
            const float r = (float)rhs;
            me[u] = make_float4(r, 0.f, 0.f, inv);
            if (live[u]) { store16_sc1(b.rec[1] + row[u], me[u]); push_record(1, row[u] - a.row0, me[u]); bb_thread += (double)r * (double)r; }      // what pass 0 of the neighbours gathers
        } else {
#pragma unroll
            for (int q = 0; q < kNQ; ++q) {
                float hv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rH, row[u] * 4, q * plane, 0));
                if (q == 0 && a.damping != 0.0f) hv += a.damping * hv;
                hs[(u * kNQ + q) * kSolveThreads + tid] = hv;
            }
            me[u] = b.rec[1][row[u]];            // {r_0, 0, 0, inv} written by the assembly kernel
        }
#pragma unroll
        for (int wd = 0; wd < (kNQ - 1) / 2; ++wd) cp[u][wd] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rC, row[u] * 4, wd * plane, 0);
        x[u] = 0.f;
    }
    // |b|^2: from the assembly kernel's per-workgroup partials (an earlier kernel: plain loads) -- or, ASM, not known before the sums of pass 0
    double bb = 0.0;
    if (!ASM) {
      double* src[1] = {fpart(a.pcg_part, -1, 6)};
      double v = 0.0;
      for (int i = tid; i < a.pcg_init_blocks; i += kSolveThreads) v += src[0][i];
      v = wave_sum(v);
      if ((tid & 63) == 0) red[tid >> 6] = v;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kSolveThreads / 64; ++i) bb += red[i];
      __syncthreads();
    } else {
        // the records of this workgroup's rows are on their way: drained, then the flag the neighbours' pass 0 waits for (plane 7 of buffer 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) { __hip_atomic_store(gran + (size_t)(kSolveGranPlanes + 7) * kSolveMaxBlocks + lb, (double)(1 + my_xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (MR) peer_tag(2, xr_tag(1.0, etag0 | 1u)); }
    }
    float rhsNorm2 = (float)bb;
    float thr = pcg_threshold(rhsNorm2);
    float alpha_prev = 0.f, beta = 0.f, rr_cur = rhsNorm2;
    if (!ASM && lb == 0 && tid == 0) fs[0] = bb;
    int k = 0, status = 1;                    // status 1 = finished, 2 = a wait timed out
    // The workgroups whose records this one gathers from: rows within `reach` of its own range (a handful of neighbours in band order).
    const int first = lb * rows_per_wg, last = first + rows_per_wg - 1;                    // (relative to row0)
    const int nlo = max(0, (first - b.reach) / rows_per_wg), nhi = min(G - 1, (last + b.reach) / rows_per_wg);
    // stage timestamps of pass 8 (timing hook only: force_passes > 0), wall clock at 100 MHz, written by thread 0 of two workgroups
#define SOLVE_STAMP(j) do { if (force_passes > 0 && k == 8 && tid == 0 && (lb == 0 || lb == (G * 9) / 16)) mb[8 + (lb ? 8 : 0) + (j)] = (double)wall_clock64(); } while (0)
    for (;; ++k) {
        SOLVE_STAMP(0);
        const unsigned want = (unsigned)k & 3u;            // tag of pass k-1 = ((k-1) + 1) & 3
        const double* gp = gran + (size_t)((k - 1) & 1) * kSolveGranPlanes * kSolveMaxBlocks;
        // (the packed column deltas are loop-invariant: given the chance, the compiler precomputes all 18 R gather addresses as 64-bit pairs
        // ahead of the loop and spills them -- an empty asm per word keeps the three-instruction address arithmetic inside the pass)
#pragma unroll
        for (int u = 0; u < R; ++u)
#pragma unroll
            for (int wd = 0; wd < (kNQ - 1) / 2; ++wd) asm volatile("" : "+v"(cp[u][wd]));
        if (k > 0 || ASM) {
            // ---- A: the records this workgroup gathers from are those of its NEIGHBOURS in band order: wait for their pass k-1 only (the tag
            // of a workgroup's first sum is stored after its records have drained), not for the whole device
            if (tid == 0) { s_abort = 0; if (ASM && k == 0) s_foreign = 0; }
            __syncthreads();
            if (tid <= nhi - nlo) {
                int spins = 0;
                // (ASM, pass 0: the neighbours' assembled records -- their flag in plane 7 of buffer 1)
                const double* wp = k > 0 ? gp + nlo + tid : gran + (size_t)(kSolveGranPlanes + 7) * kSolveMaxBlocks + nlo + tid;
                double seen = 0.0;
                while (k > 0 ? gran_tag_of(seen = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != want : (seen = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0.0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kLocalSpins || __hip_atomic_load(fs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) { s_abort = 1; break; }
                }
                if (ASM && k == 0 && (int)seen != 1 + my_xcc) s_foreign = 1;      // (the relation is symmetric: whoever gathers from this workgroup is in [nlo, nhi])
            }
            if (MR) {
                // rows within `reach` of a cut gather from halo rows: wait for the tags the neighbour slab's cut-side workgroups wrote into this rank's
                // region after their records had drained (threads 64.. / 128..: wavefronts 1 and 2, next to the local pollers in wavefront 0)
                const bool need_lo = xr.need_lo > 0 && wg_first < b.reach, need_hi = xr.need_hi > 0 && wg_last + b.reach >= own_n;
                const int side = tid >= 128 ? 1 : 0, j = tid - (side ? 128 : 64);      // (kXrPeerTags = 64 slots per side: threads 64..127 / 128..191)
                if (tid >= 64 && tid < 192 && j < (side ? xr.wait_hi : xr.wait_lo) && (side ? need_hi : need_lo)) {
                    int spins = 0;
                    const double* wp = xr_me + kXrPtag + (side * 3 + (k > 0 ? ((k - 1) & 1) : 2)) * kXrPeerTags + j;
                    const unsigned wantx = etag0 | (k > 0 ? want : 1u);      // (prologue flag: pass tag 1 in buffer 2)
                    while (xr_tag_of(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != wantx) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1 << 24) || __hip_atomic_load(fs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0 || __hip_atomic_load(xr_me + kXrAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0.0) { s_abort = 1; break; }
                    }
                }
            }
            __syncthreads();
            if (s_abort) { if (tid == 0) { __hip_atomic_store(fs + 3, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (MR) for (int r = 0; r < xr.n_ranks; ++r) __hip_atomic_store(xr.region[r] + kXrAbort, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } status = 2; break; }
            if (tid == 0) { if (MR) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }      // (MR: system scope -- the halo records came from another GPU)
            if (ASM && k == 0) xcd_local = a.pcg_xcd_local && !s_foreign;
            __syncthreads();
        }
        SOLVE_STAMP(1);
        // ---- B: t = A p_k is linear in the gathered fields (A1 - alpha A2 + beta A3): the gathers run BEFORE alpha and beta exist, i.e.
        // while the sums of the far workgroups are still on their way
        const float4* __restrict__ rin = b.rec[(k + 1) & 1];
        float4* __restrict__ rout = b.rec[k & 1];
        double A1[R], A2[R], A3[R];
        // the 18 neighbour records of a row in two batches of 9, software-pipelined ACROSS the rows of the thread: two batches (72 registers)
        // are in flight at any time -- batch t+2 is requested into the registers batch t has just been consumed from, so a wave's memory
        // requests never run dry while it converts and multiplies (one row at a time did: gather stage 4.1-5.1 -> 3.5-3.8 us)
        float4 ob[2][9];
        auto issue = [&](int t) {
            const int u = t >> 1, j0 = (t & 1) * 9;
#pragma unroll
            for (int j = 0; j < 9; ++j) { const int jj = j0 + j; const int pk = (int)cp[u][jj >> 1]; ob[t & 1][j] = rin[row[u] + ((jj & 1) ? (pk >> 16) : ((pk << 16) >> 16))]; }
        };
        issue(0);
        issue(1);
        double a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int t = 0; t < 2 * R; ++t) {
            const int u = t >> 1, j0 = (t & 1) * 9;
            const float* hrow = hs + (size_t)u * kNQ * kSolveThreads + tid;
            if (!(t & 1)) { const double hq = (double)hrow[0], iv = (double)me[u].w; a1 = hq * (iv * (double)me[u].x); a2 = hq * (iv * (double)me[u].y); a3 = hq * (double)me[u].z; }
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const double hq = (double)hrow[(j0 + j + 1) * kSolveThreads], iv = (double)ob[t & 1][j].w;
                a1 += hq * (iv * (double)ob[t & 1][j].x); a2 += hq * (iv * (double)ob[t & 1][j].y); a3 += hq * (double)ob[t & 1][j].z;
            }
            asm volatile("" : "+v"(a1), "+v"(a2), "+v"(a3));      // the sums exist HERE: without this the compiler sinks the consumption of the
            if (t & 1) { A1[u] = a1; A2[u] = a2; A3[u] = a3; }     // gathered records below the wait for the global sums (72 registers live across it)
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < 2 * R) issue(t + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        SOLVE_STAMP(2);
        if (force_passes > 0 && k == 9 && tid == 0) fs[16 + 1024 + lb] = (double)wall_clock64();            // gathers of pass 9 done
        if (k > 0) {
            // ---- C: the seven sums of pass k-1 of EVERY workgroup (data and tag in one granule: no fence needed for them)
            // (requesting these granules ahead of time -- in front of the gathers or behind the last batch -- measured SLOWER: the sc1 loads
            // lengthen the gather stage by 1-2 us, and the stage is a wait for the late workgroups, not a load latency; profiles/r02_notes.md)
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = 0.0;
            if (tid < G) {
                int spins = 0; bool ok = false;
                while (!ok) {
                    ok = true;
#pragma unroll
                    for (int q = 0; q < kCgfSums; ++q) { v[q] = __hip_atomic_load(gp + (size_t)q * kSolveMaxBlocks + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = ok && gran_tag_of(v[q]) == want; }
                    if (ASM && k == 1) { v[7] = __hip_atomic_load(gp + (size_t)7 * kSolveMaxBlocks + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = ok && gran_tag_of(v[7]) == want; }   // |b|^2 travels with the sums of pass 0
                    if (!ok) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > kLocalSpins || __hip_atomic_load(fs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) { s_abort = 1; break; }
                    }
                }
            }
            double t0, t1; wave_sum8(v, t0, t1);
            wave_sum8_store<kSolveThreads / 64>(t0, t1, red, tid >> 6);
            __syncthreads();
            if (s_abort) { if (tid == 0) { __hip_atomic_store(fs + 3, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (MR) for (int r = 0; r < xr.n_ranks; ++r) __hip_atomic_store(xr.region[r] + kXrAbort, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } status = 2; break; }
            double t[kCgfSums];
#pragma unroll
            for (int q = 0; q < kCgfSums; ++q) { double s_ = 0; for (int i = 0; i < kSolveThreads / 64; ++i) s_ += red[q * (kSolveThreads / 64) + i]; t[q] = s_; }
            if (ASM && k == 1) {
                bb = 0.0;
                for (int i = 0; i < kSolveThreads / 64; ++i) bb += red[7 * (kSolveThreads / 64) + i];
            }
            __syncthreads();
            if (MR) {
                // ---- C2: the sums just obtained are this RANK's (every local workgroup holds the same bits).  Workgroup 0 hands them to every rank as tagged
                // granules; every workgroup then adds the R granules of its own region in rank order (one fixed tree: the same bits on every rank).
                const int pb = (k - 1) & 1;
                if (lb == 0 && tid < 8 && (tid < kCgfSums || (ASM && k == 1))) {
                    const double mine = xr_tag(tid < kCgfSums ? t[tid] : bb, etag0 | want);
                    for (int r = 0; r < xr.n_ranks; ++r) __hip_atomic_store(xr.region[r] + kXrRankGran + (pb * 8 + tid) * kXrMaxRanks + xr.rank, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                double rv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) rv[q] = 0.0;
                if (tid < xr.n_ranks) {
                    int spins = 0; bool ok = false;
                    while (!ok) {
                        ok = true;
#pragma unroll
                        for (int q = 0; q < kCgfSums; ++q) { rv[q] = __hip_atomic_load(xr_me + kXrRankGran + (pb * 8 + q) * kXrMaxRanks + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ok = ok && xr_tag_of(rv[q]) == (etag0 | want); }
                        if (ASM && k == 1) { rv[7] = __hip_atomic_load(xr_me + kXrRankGran + (pb * 8 + 7) * kXrMaxRanks + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ok = ok && xr_tag_of(rv[7]) == (etag0 | want); }
                        if (!ok) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1 << 24) || __hip_atomic_load(fs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0 || __hip_atomic_load(xr_me + kXrAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0.0) { s_abort = 1; break; }
                        }
                    }
                }
                double r0, r1; wave_sum8(rv, r0, r1);
                wave_sum8_store<kSolveThreads / 64>(r0, r1, red, tid >> 6);
                __syncthreads();
                if (s_abort) { if (tid == 0) { __hip_atomic_store(fs + 3, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); for (int r = 0; r < xr.n_ranks; ++r) __hip_atomic_store(xr.region[r] + kXrAbort, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } status = 2; break; }
#pragma unroll
                for (int q = 0; q < kCgfSums; ++q) t[q] = red[q * (kSolveThreads / 64)];      // (all R <= 32 granules sit in wavefront 0)
                if (ASM && k == 1) bb = red[7 * (kSolveThreads / 64)];
                __syncthreads();
            }
            if (ASM && k == 1) {
                rhsNorm2 = (float)bb; thr = pcg_threshold(rhsNorm2);
                if (lb == 0 && tid == 0) fs[0] = bb;
            }
            const float rz_old = (float)t[5];
            alpha_prev = rz_old / (float)t[0];
            const double al = (double)alpha_prev;
            const float rz_cur = (float)(t[5] - 2.0 * al * t[1] + al * al * t[2]);
            rr_cur = (float)(t[6] - 2.0 * al * t[3] + al * al * t[4]);
            beta = rz_cur / rz_old;
            if (force_passes > 0 && k == 9 && tid == 0) fs[16 + 256 + lb] = (double)wall_clock64();         // ... and when it had the sums of pass 8 of all the others
            if (force_passes > 0 && k == 9 && tid == 0) fs[16 + 768 + lb] = (double)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // XCC_ID
        }
        if (force_passes > 0 && k == 9 && tid == 0) fs[16 + 512 + lb] = (double)wall_clock64();
        SOLVE_STAMP(3);
        const bool rhs_zero = (!ASM || k > 0) && rhsNorm2 == 0.f;          // (ASM: |b|^2 arrives with the sums of pass 0)
        const bool stop = force_passes > 0 ? k >= force_passes : (rhs_zero || k == kmax || (k > 0 && rr_cur < thr));
        // ---- D: finish pass k-1 for the own rows; pass k
        double s[kCgfSums];
#pragma unroll
        for (int q = 0; q < kCgfSums; ++q) s[q] = 0;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (k > 0 && !rhs_zero) x[u] = x[u] + alpha_prev * me[u].z;
            if (stop) continue;
            const float r_i = me[u].x - alpha_prev * me[u].y;
            const float z_i = me[u].w * r_i;
            const float p_i = z_i + beta * me[u].z;
            const float tt = (float)(A1[u] - (double)alpha_prev * A2[u] + (double)beta * A3[u]);
            me[u] = make_float4(r_i, tt, p_i, me[u].w);
            if (live[u]) {
                if (xcd_local) rout[row[u]] = me[u]; else store16_sc1(rout + row[u], me[u]);      // (all its readers share this XCD's L2 / through memory)
                push_record(k & 1, row[u] - a.row0, me[u]);
                const double rd = (double)r_i, td = (double)tt, iv = (double)me[u].w;
                s[0] += (double)p_i * td; s[1] += iv * rd * td; s[2] += iv * td * td; s[3] += rd * td; s[4] += td * td;
                s[5] += rd * (double)z_i; s[6] += rd * rd;
            }
        }
        if (stop) break;
        if (force_passes == -7 && k == 2 && lb == 1) { status = 2; break; }      // fault injection (psgsdf_debug_time_pcg_solve(passes = -7)): this workgroup never publishes pass 2 -- every other one must give up waiting, not hang
        SOLVE_STAMP(4);
        // ---- E: publish: the records have to be out (write-through, drained) before the seven tagged sums
        double sv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) sv[q] = q < kCgfSums ? s[q] : ((ASM && k == 0) ? bb_thread : 0.0);
        double t0, t1; wave_sum8(sv, t0, t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sum8_store<kSolveThreads / 64>(t0, t1, red, tid >> 6);
        __syncthreads();
        SOLVE_STAMP(5);
        if (tid < kCgfSums || (ASM && k == 0 && tid == 7)) {
            double tot = 0;
            for (int i = 0; i < kSolveThreads / 64; ++i) tot += red[tid * (kSolveThreads / 64) + i];
            double* gq = gran + (size_t)(k & 1) * kSolveGranPlanes * kSolveMaxBlocks + (size_t)tid * kSolveMaxBlocks + lb;
            __hip_atomic_store(gq, gran_tag(tot, (unsigned)(k + 1) & 3u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MR && tid == 64) peer_tag(k & 1, xr_tag(1.0, etag0 | ((unsigned)(k + 1) & 3u)));      // (every wave drained its stores before the barrier above: the halo records are out)
        __syncthreads();
        SOLVE_STAMP(6);
        if (force_passes > 0 && k == 8 && tid == 0) fs[16 + lb] = (double)wall_clock64();                 // timing hook: when every workgroup published pass 8 ...
    }
#undef SOLVE_STAMP
    // ---- leave: x of the own rows; the outcome for the host and for the gated kernels behind this one
#pragma unroll
    for (int u = 0; u < R; ++u) if (live[u]) b.x[row[u]] = x[u];
    if (a.pcg_apply) {
        // the distance update (k_apply_dist: updateDist's accept rule, OptimizerAux.cpp:162-188) for the own rows, under the rule the host applies
        // to the solve's outcome -- every workgroup holds the same |r|^2 and |b|^2.  The accepted count is a sum of integers: the per-workgroup
        // counts go to the first G entries of the slot, the rest of the slot (k_apply_albedo's leftovers) is cleared.
        const bool z0 = rhsNorm2 == 0.f;
        const bool ok_all = status == 1 && (z0 || sqrt((double)rr_cur / (double)rhsNorm2) <= (double)FLT_EPSILON);
        const bool apply = status == 1 && (a.pcg_apply == 1 || ok_all);
        double cnt = 0;
        if (apply) {
#pragma unroll
            for (int u = 0; u < R; ++u) if (live[u] && (double)fabsf(x[u]) < sqrt(3.0) * (double)a.grid.vs) { b.dist[row[u]] -= x[u]; cnt += 1.0; }
        }
        cnt = wave_sum(cnt);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = cnt;
        __syncthreads();
        double* slot = PART(a, SC_ACCEPT);
        if (tid == 0) { double t = 0; for (int i = 0; i < kSolveThreads / 64; ++i) t += red[i]; slot[lb] = t; }
        for (int i = G + lb * kSolveThreads + tid; i < a.acc.PB; i += G * kSolveThreads) slot[i] = 0.0;
    }
    if (lb == 0 && tid == 0) {
        const bool rhs_zero = rhsNorm2 == 0.f;
        const bool ok = status == 1 && (rhs_zero || sqrt((double)rr_cur / (double)rhsNorm2) <= (double)FLT_EPSILON);
        int iters = 0;
        if (!rhs_zero && k > 0) iters = (rr_cur < thr) ? k - 1 : k;      // Eigen leaves the loop before ++i when it detects convergence; k == kmax otherwise
        fs[1] = status == 1 ? (double)(k + 1) : 0.0; fs[2] = ok ? 1.0 : 0.0;      // a solve that gave up leaves BOTH gates closed: nothing behind it may act on it (the host re-runs the solve, loop.hip)
        const double m0 = (double)iters, m1 = (double)rr_cur, m2 = (double)rhsNorm2, m3 = (double)status;
        mb[0] = m0; mb[1] = m1; mb[2] = m2;
        __threadfence_system();
        mb[3] = m3;                            // the host watches this slot ...
        // ... and takes the four words only together with their check word (engine.h FoldReq)
        if (mb_key) reinterpret_cast<unsigned long long*>(mb)[4] = (unsigned long long)(__double_as_longlong(m0) ^ __double_as_longlong(m1) ^ __double_as_longlong(m2) ^ __double_as_longlong(m3)) ^ mb_key;
        __threadfence_system();
    }
}

// ------------------------------------------------------------------------------------------
// The persistent solve, PIPELINED (round 4): the same matrix-in-LDS kernel, the same hand-offs, but with the recurrences of
// Ghysels & Vanroose's pipelined CG so that a pass no longer waits for its own reduction.
//
// In k_cgf_solve the seven sums of pass k contain t_k = A p_k, i.e. they can only be published AFTER the gathers of pass k, and the next pass can
// only start after they have gone round the device: per pass  tags ~0.9 + gathers ~4.2 + sums ~2.7 + finish ~1.1 = 9.5 us (tools/pcg_solve_time.py),
// the 2.7 us being a pure wait for the slowest workgroup's publish to arrive.  Here a pass publishes, at its END, the next search-direction input
// m_{k+1} = M^-1 w_{k+1} AND the three sums of the NEXT pass -- gamma = r.u, delta = w.u, |r|^2 with u = M^-1 r, w = A u carried by recurrence --
// none of which involves the matrix-vector product n_{k+1} = A m_{k+1} that the next pass gathers for.  The all-gather of the sums therefore runs
// concurrently with the neighbour hand-off and the gathers, and is complete when they are:
//     per pass  tags + gathers + finish   (~6 us: profiles/r04_notes.md)
// Recurrences (Jacobi M = diag, so q = M^-1 s and m = M^-1 w need no extra vectors), Eigen's x0 = 0:
//     r_0 = b, u_0 = M^-1 b, w_0 = A u_0 (one extra gather round in front);   pass k:  n = A (M^-1 w)   ||   gamma, delta, |r|^2 over the device
//     beta = gamma / gamma_old (0 in pass 0), alpha = gamma / (delta - beta gamma / alpha_old)
//     z = n + beta z;  s = w + beta s;  p = u + beta p;  x += alpha p;  r -= alpha s;  w -= alpha z;  u = M^-1 r
// In exact arithmetic the iterates ARE Eigen::ConjugateGradient's (alpha equals r.z / p.Ap); in float they differ from it -- and from k_cgf_pass,
// which keeps the round-1 recurrences for bands that do not fit the LDS and as the fallback -- at rounding level (deviation 3;
// measured margins in profiles/r04_parity_margins.json).  Stop rule, iteration count and info() are Eigen's: |r_k|^2 of the recursively updated
// residual against max(eps^2 |b|^2, FLT_MIN), checked before update k + 1.  Gather rounds per solve: iterations + 2 (w_0, and the round that is in
// flight when the stop is detected) instead of iterations + 1.
// What travels: ONE float per row (m) instead of a 16-byte record, three tagged sums instead of seven (+ |b|^2 = the |r|^2 of pass 0).
// ------------------------------------------------------------------------------------------
constexpr int kCgpSums = 3;
constexpr int kCgpMaxRows = 4;                // row slots per thread the pipelined kernel supports
#ifndef PSG_CGP_DEPTH
#define PSG_CGP_DEPTH 2
#endif
constexpr int kCgpDepth = PSG_CGP_DEPTH;      // gather batches (of 9 doubles) in flight per thread
__device__ __forceinline__ void store8_sc1(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store8_sys(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }

// TM (round 5, single rank, PSGSDF_PCG_TAGM): SELF-VALIDATING m.  Every exchanged value carries (solve epoch << 2 | pass tag) in its four lowest mantissa bits
// (2^-48 of a double that multiplies float coefficients), is stored write-through and gathered with agent-scope loads that are re-issued until the tag is
// the pass's: no neighbour-tag wait, no acquire, no drain of the stores in front of the sums -- the hand-off of a pass is ONE trip through memory.
__device__ __forceinline__ double m_tag(double v, unsigned tag) { return __longlong_as_double((long long)(((unsigned long long)__double_as_longlong(v) & ~15ull) | tag)); }
__device__ __forceinline__ unsigned m_tag_of(double v) { return (unsigned)((unsigned long long)__double_as_longlong(v) & 15ull); }
template <int R, bool MR, bool TM = false>
__global__ void __launch_bounds__(kSolveThreads, 2) k_cgp_solve(SweepArgs a, double* fs, double* gran, int rows_per_wg, int kmax, double* mb, unsigned long long mb_key, int force_passes, XrArgs xr) {
    __shared__ double red[8 * kSolveThreads / 64];
    __shared__ int s_abort;
    // kFast (self-validating values: the production kernels): TWO workgroup barriers per pass instead of five (round 6: 7.05 -> 6.5 us per
    // pass).  What is left is one barrier per cross-wavefront sum: in front of stage C's reads of `red` and in front of stage E's reads of `red2`.  The sums
    // of C and E meet in buffers of their own, E's alternating by pass parity -- so neither the barrier between C's reads and E's writes nor the one behind
    // E's reads is needed: whoever writes a buffer again is two barriers further on than its last reader.  The 'a wait gave up' flag rotates through three
    // words (pass mod 3): thread 0 clears the word of pass k + 2 behind stage E's barrier of pass k -- its last readers passed that barrier with it, its next
    // raisers are two barriers away -- so the barrier that only published the cleared flag at the top of a pass goes too.  Every read of a pass's word lies
    // behind a barrier of that pass and every raise in front of it: all threads of the workgroup take the same decision.
    constexpr bool kFast = TM;      // (multi-rank too: its stage C2 -- the ranks' sums -- gets a third buffer and keeps the one barrier in front of its reads: three per pass instead of seven)
    __shared__ double red2[kFast ? 2 * 8 * kSolveThreads / 64 : 1];
    __shared__ double red3[(kFast && MR) ? 8 * kSolveThreads / 64 : 1];
    __shared__ int s_ab3[3];
#define CGP_ABORT (*(kFast ? &s_ab3[(k + 3) % 3] : &s_abort))
    __shared__ int s_foreign;
    const Band& b = a.b;
    const int G = gridDim.x, tid = threadIdx.x;
    const int my_xcc = (int)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
    bool xcd_local = false;
    const int lb = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int plane = b.Spad * 4;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)b.colp, 0, (kNQ - 1) / 2 * plane, 0x00020000);
    double* const recd[2] = {(double*)b.rec[0], (double*)b.rec[1]};      // m_k lives in recd[k & 1] (u_0 of the prologue in recd[1]): the first Spad doubles of a record plane
    const int own_n = a.row1 - a.row0;
    const int wg_first = lb * rows_per_wg, wg_last = min(own_n, wg_first + rows_per_wg) - 1;
    const bool cut_lo = MR && xr.give_lo > 0 && wg_first < xr.give_lo && wg_first < own_n;
    const bool cut_hi = MR && xr.give_hi > 0 && wg_last >= own_n - xr.give_hi && wg_first < own_n;
    const int hi_first_wg = MR ? max(0, own_n - xr.give_hi) / rows_per_wg : 0;
    double* const xr_me = MR ? xr.region[xr.rank] : nullptr;
    const unsigned etag0 = MR ? (xr.epoch & kXrEpochMask) << 2 : 0u;
    constexpr int kLocalSpins = MR ? (1 << 24) : (1 << 22);
    auto peer_tag = [&](int buf, double v) {
        if (cut_lo && lb < kXrPeerTags) __hip_atomic_store(xr.region[xr.rank - 1] + kXrPtag + (1 * 3 + buf) * kXrPeerTags + lb, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (cut_hi && lb - hi_first_wg < kXrPeerTags) __hip_atomic_store(xr.region[xr.rank + 1] + kXrPtag + (0 * 3 + buf) * kXrPeerTags + (lb - hi_first_wg), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    // the neighbour slab's halo row of the same band row.  (xr.lo_rec / hi_rec point at the first halo RECORD, 16 bytes per row: the same row counted in doubles)
    const int lo_row0 = MR && xr.lo_rec[0] ? (int)(xr.lo_rec[0] - xr.lo_base[0]) : 0;
    auto push_record = [&](int buf, int rel, double v) {
        if (MR && rel < xr.give_lo) store8_sys((double*)xr.lo_base[buf] + lo_row0 + rel, v);
        if (MR && rel >= own_n - xr.give_hi) store8_sys((double*)xr.hi_rec[buf] + (rel - (own_n - xr.give_hi)), v);
    };
    auto raise_abort = [&] {
        __hip_atomic_store(fs + 3, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MR) for (int r = 0; r < xr.n_ranks; ++r) __hip_atomic_store(xr.region[r] + kXrAbort, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    float* hs = (float*)psg_dyn_smem;
    const unsigned mepoch = TM ? (1u + a.pcg_epoch % 3u) << 2 : 0u;      // 1 .. 3, never the epoch of the solve before; 0 = memory no solve has written yet (the planes are zeroed when the band is built)
    unsigned cp[R][(kNQ - 1) / 2]; int row[R]; bool live[R];
    double x[R], r[R], w[R], z[R], sv[R], pv[R]; float inv[R];      // every vector of the recurrences in double: see "precision" above
    if (a.fold.n != 0 && blockIdx.x == 0) {      // the sums the distance sweep left pending (device_common.h fold_pending), in that function's order
        double ftot[4] = {0.0, 0.0, 0.0, 0.0};
        for (int sl = 0; sl < a.fold.n; ++sl) {
            const double* part = PART(a, a.fold.id[sl]);
            double v = 0;
            if (tid < kBlock) for (int i = tid; i < a.fold.nblk; i += kBlock) v += part[i];
            v = wave_sum(v);
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = v;
            __syncthreads();
            if (tid == 0) { double t = 0; for (int i = 0; i < kBlock / 64; ++i) t += red[i]; ftot[sl] = t; }
        }
        if (tid < 64) {      // (the exchange between the ranks takes the whole first wavefront; lane 0 holds the slab's sums)
            if (a.fold.xf) fold_exchange(a.fold.xf, a.fold.xf_epoch, a.fold.n, ftot);      // multi-rank: the sums over all slabs
            if (tid == 0) {
                for (int sl = 0; sl < a.fold.n; ++sl) mbox_put(a.fold.out, a.fold.n, sl, ftot[sl], a.fold.key);
                mbox_commit(a.fold.key);
            }
        }
        __syncthreads();
    }
    // ---- once: assemble the rows of this thread (coefficients -> LDS), r_0 = b, u_0 = M^-1 b out for the neighbours
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int i = a.row0 + lb * rows_per_wg + u * kSolveThreads + tid;
        live[u] = u * kSolveThreads + tid < rows_per_wg && i < a.row1; row[u] = live[u] ? i : a.row1 - 1;
        double acc[kNQ], rhs;
        assemble_row_regs(a, row[u], acc, rhs);
#pragma unroll
        for (int q = 0; q < kNQ; ++q) {
            float hv = (float)acc[q];
            if (q == 0 && a.damping != 0.0f) hv += a.damping * hv;
            hs[(u * kNQ + q) * kSolveThreads + tid] = hv;
        }
        float dg = (float)acc[0];
        if (a.damping != 0.0f) dg += a.damping * dg;
        inv[u] = dg != 0.f ? 1.0f / dg : 1.0f;
        r[u] = live[u] ? (double)(float)rhs : 0.0;      // (b is the float vector the reference solves for)
        x[u] = 0.0; z[u] = 0.0; sv[u] = 0.0; pv[u] = 0.0; w[u] = 0.0;
        if (live[u]) {
            const double u0 = (double)inv[u] * r[u];
            if (TM) { const double ut = m_tag(u0, mepoch | 0u); __hip_atomic_store(recd[1] + row[u], ut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); push_record(1, row[u] - a.row0, ut); }
            else { store8_sc1(recd[1] + row[u], u0); push_record(1, row[u] - a.row0, u0); }
        }
#pragma unroll
        for (int wd = 0; wd < (kNQ - 1) / 2; ++wd) cp[u][wd] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rC, row[u] * 4, wd * plane, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) { __hip_atomic_store(gran + (size_t)(kSolveGranPlanes + 7) * kSolveMaxBlocks + lb, (double)(1 + my_xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (MR && !TM) peer_tag(2, xr_tag(1.0, etag0 | 1u)); }
    float rhsNorm2 = 0.f, thr = 0.f, rr_cur = 0.f;
    double gamma_old = 0.0, alpha = 0.0;
    int k = -1, status = 1;                   // k = -1: the extra round w_0 = A u_0
    const int first = lb * rows_per_wg, last = first + rows_per_wg - 1;
    const int nlo = max(0, (first - b.reach) / rows_per_wg), nhi = min(G - 1, (last + b.reach) / rows_per_wg);
    const bool need_lo = MR && xr.need_lo > 0 && wg_first < b.reach, need_hi = MR && xr.need_hi > 0 && wg_last + b.reach >= own_n;
#define SOLVE_STAMP(j) do { if (force_passes > 0 && k == 8 && tid == 0 && (lb == 0 || lb == (G * 9) / 16)) mb[8 + (lb ? 8 : 0) + (j)] = (double)wall_clock64(); } while (0)
    for (;; ++k) {
        SOLVE_STAMP(0);
        const unsigned want = (unsigned)(k + 1) & 3u;      // tag of the sums published for pass k (k >= 0)
        const double* gp = gran + (size_t)(k & 1) * kSolveGranPlanes * kSolveMaxBlocks;
#pragma unroll
        for (int u = 0; u < R; ++u)
#pragma unroll
            for (int wd = 0; wd < (kNQ - 1) / 2; ++wd) asm volatile("" : "+v"(cp[u][wd]));
        // ---- A: the values this workgroup gathers are those of its neighbours in band order: their tag (the first sum of pass k, stored after
        // their m_k had drained; k = -1: the prologue's flag)
        if (!kFast || k < 0) {
            if (tid == 0) { s_abort = 0; s_ab3[0] = 0; s_ab3[1] = 0; s_ab3[2] = 0; if (k < 0) s_foreign = 0; }
            __syncthreads();
        }
        if (!TM && tid <= nhi - nlo && !((a.pcg_xcd_local >> 3) & 2)) {
            int spins = 0;
            const double* wp = k >= 0 ? gp + nlo + tid : gran + (size_t)(kSolveGranPlanes + 7) * kSolveMaxBlocks + nlo + tid;
            double seen = 0.0;
            while (k >= 0 ? gran_tag_of(seen = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != want : (seen = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0.0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kLocalSpins || __hip_atomic_load(fs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) { CGP_ABORT = 1; break; }
            }
            if (k < 0 && (int)seen != 1 + my_xcc) s_foreign = 1;
        }
        if (MR && !TM) {
            const int side = tid >= 128 ? 1 : 0, j = tid - (side ? 128 : 64);
            if (tid >= 64 && tid < 192 && j < (side ? xr.wait_hi : xr.wait_lo) && (side ? need_hi : need_lo)) {
                int spins = 0;
                const double* wp = xr_me + kXrPtag + (side * 3 + (k >= 0 ? (k & 1) : 2)) * kXrPeerTags + j;
                const unsigned wantx = etag0 | (k >= 0 ? want : 1u);
                while (xr_tag_of(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != wantx) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24) || __hip_atomic_load(fs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0 || __hip_atomic_load(xr_me + kXrAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0.0) { CGP_ABORT = 1; break; }
                }
            }
        }
        if (!TM) {
            __syncthreads();
            if (CGP_ABORT) { if (tid == 0) raise_abort(); status = 2; break; }
            if (tid == 0) { if (MR) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
            if (k < 0) xcd_local = (a.pcg_xcd_local & 1) && !s_foreign;
            __syncthreads();
        }
        SOLVE_STAMP(1);
        if (force_passes > 0 && k == 9 && tid == 0) fs[16 + 1536 + lb] = (double)wall_clock64();      // ... neighbours' tags of pass 9 seen
        // ---- B: n = A m (k = -1: w_0 = A u_0): 18 four-byte gathers per row in two batches of 9, software-pipelined across the rows of the thread
        const double* __restrict__ rin = recd[k >= 0 ? (k & 1) : 1];
        const int abl = (a.pcg_xcd_local >> 3) & 7;      // timing ablations (tools/pcg_variants.py; never set in production): 1 = every gather reads the row's own element, 2 = no neighbour-tag wait, 4 = the 8 in-plane columns are not fetched
        double nres[R];
        double ob[kCgpDepth][9];
        auto issue = [&](int t) {
            const int u = t >> 1, j0 = (t & 1) * 9;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int jj = j0 + j; const int pk = (int)cp[u][jj >> 1];
                constexpr unsigned kInPlane = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 3) | (1u << 6) | (1u << 7) | (1u << 12) | (1u << 13);      // columns without a z offset (q_offset: q = jj + 1)
                if ((abl & 4) && ((kInPlane >> jj) & 1u)) { ob[t % kCgpDepth][j] = 0.0; continue; }      // ablation 4: the in-plane columns are not gathered at all
                const int srow = row[u] + ((abl & 1) ? 0 : ((jj & 1) ? (pk >> 16) : ((pk << 16) >> 16)));
                const double* src = rin + srow;
                //  a HALO row's value is written by the neighbour RANK: gathered at system scope from the first attempt on -- the tag alone cannot tell
                // this pass's value from the one of four passes ago (a buffer's tags alternate between two values), so the load itself must not be served from a
                // copy this device cached then.  Own rows: agent scope, as on one rank.
                if (TM && MR && (srow < a.row0 || srow >= a.row1)) ob[t % kCgpDepth][j] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else ob[t % kCgpDepth][j] = TM ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
            }
        };
        // The sums of pass k were published when m_k was: by the time the last gather batch is on its way they have normally arrived, but FETCHING
        // them is a memory round trip of its own (agent-scope loads past the XCD's L2: ~1.5 us if issued only after the gathers).  So they are
        // requested right behind the last batch and checked in stage C; only a late workgroup's granules are polled for again there.
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.0;
        auto prefetch_sums = [&] {
            if (k >= 0 && (a.pcg_xcd_local & 2) && tid < G) {
#pragma unroll
                for (int q = 0; q < kCgpSums; ++q) v[q] = __hip_atomic_load(gp + (size_t)q * kSolveMaxBlocks + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
#pragma unroll
        for (int t = 0; t < kCgpDepth && t < 2 * R; ++t) issue(t);
        if (2 * R - 1 - kCgpDepth < 0) { __builtin_amdgcn_sched_barrier(0); prefetch_sums(); __builtin_amdgcn_sched_barrier(0); }
        double acc = 0;
#pragma unroll
        for (int t = 0; t < 2 * R; ++t) {
            const int u = t >> 1, j0 = (t & 1) * 9;
            const float* hrow = hs + (size_t)u * kNQ * kSolveThreads + tid;
            if (!(t & 1)) { const double mine = (double)inv[u] * (k >= 0 ? w[u] : r[u]); acc = (double)hrow[0] * mine; }      // (the row's own m: what it published)
            if (TM) {      // a value that is not this pass's yet: ask again (the producer is at most one hand-off behind)
                const unsigned wantm = mepoch | want;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    if (m_tag_of(ob[t % kCgpDepth][j]) == wantm) continue;
                    const int jj = j0 + j; const int pk = (int)cp[u][jj >> 1];
                    if ((abl & 4) && ((0x30cfu >> jj) & 1u)) continue;      // (timing ablation 4: this column was not gathered)
                    const double* src = rin + (row[u] + ((jj & 1) ? (pk >> 16) : ((pk << 16) >> 16)));
                    int spins = 0; double vv;
                    do {
                        __builtin_amdgcn_s_sleep(1);
                        vv = MR ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (a halo row is written by the neighbour RANK)
                        if (++spins > kLocalSpins || ((spins & 255) == 0 && (__hip_atomic_load(fs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0 || (MR && __hip_atomic_load(xr_me + kXrAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0.0)))) { CGP_ABORT = 1; break; }
                    } while (m_tag_of(vv) != wantm);
                    ob[t % kCgpDepth][j] = vv;
                }
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) acc += (double)hrow[(j0 + j + 1) * kSolveThreads] * ob[t % kCgpDepth][j];
            asm volatile("" : "+v"(acc));
            if (t & 1) nres[u] = acc;
            __builtin_amdgcn_sched_barrier(0);
            if (t + kCgpDepth < 2 * R) issue(t + kCgpDepth);
            if (t == 2 * R - 1 - kCgpDepth) prefetch_sums();
            __builtin_amdgcn_sched_barrier(0);
        }
        SOLVE_STAMP(2);
        if (force_passes > 0 && k == 9 && tid == 0) { fs[16 + 1024 + lb] = (double)wall_clock64(); fs[16 + 768 + lb] = (double)my_xcc; fs[16 + 1792 + lb] = (double)(gran_tag_of(v[0]) == want && gran_tag_of(v[1]) == want && gran_tag_of(v[2]) == want); }      // timing hook (PSGSDF_SOLVE_DUMP): gathers of pass 9 done; was the prefetch of this thread's granules valid?
        double beta = 0.0;
        bool stop = false;
        if (k >= 0) {
            // ---- C: the three sums for pass k of EVERY workgroup (published at the end of pass k - 1: normally all here by now)
            auto poll3 = [&](const double* base, bool prefetched) {      // this thread's three granules at base[q * kSolveMaxBlocks] until they carry the pass's tag
                int spins = 0;
                bool ok = prefetched && gran_tag_of(v[0]) == want && gran_tag_of(v[1]) == want && gran_tag_of(v[2]) == want;
                while (!ok) {
                    if (spins) __builtin_amdgcn_s_sleep(1);
                    if (++spins > kLocalSpins || __hip_atomic_load(fs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) { CGP_ABORT = 1; break; }
                    ok = true;
#pragma unroll
                    for (int q = 0; q < kCgpSums; ++q) { v[q] = __hip_atomic_load(base + (size_t)q * kSolveMaxBlocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = ok && gran_tag_of(v[q]) == want; }
                }
            };
            if (tid < G) poll3(gp + tid, (a.pcg_xcd_local & 2) != 0);
            { const double v4[4] = {v[0], v[1], v[2], 0.0}; wave_sum4_store<kSolveThreads / 64>(wave_sum4(v4), red, tid >> 6); }      // (three sums: the four-value reduce-scatter, the same bits as wave_sum8's)
            __syncthreads();
            if (CGP_ABORT) { if (tid == 0) raise_abort(); status = 2; break; }
            double t[kCgpSums];
#pragma unroll
            for (int q = 0; q < kCgpSums; ++q) { double s_ = 0; for (int i = 0; i < kSolveThreads / 64; ++i) s_ += red[q * (kSolveThreads / 64) + i]; t[q] = s_; }
            if (!kFast) __syncthreads();      // (kFast: stage E has a buffer of its own)
            if (MR) {
                // ---- C2: this RANK's sums -> every rank's region (workgroup 0), then the R rank granules of the own region in rank order
                const int pb = k & 1;
                if (lb == 0 && tid < kCgpSums) {
                    const double mine = xr_tag(t[tid], etag0 | want);
                    for (int rk = 0; rk < xr.n_ranks; ++rk) __hip_atomic_store(xr.region[rk] + kXrRankGran + (pb * 8 + tid) * kXrMaxRanks + xr.rank, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                double rv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) rv[q] = 0.0;
                if (tid < xr.n_ranks) {
                    int spins = 0; bool ok = false;
                    while (!ok) {
                        ok = true;
#pragma unroll
                        for (int q = 0; q < kCgpSums; ++q) { rv[q] = __hip_atomic_load(xr_me + kXrRankGran + (pb * 8 + q) * kXrMaxRanks + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ok = ok && xr_tag_of(rv[q]) == (etag0 | want); }
                        if (!ok) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1 << 24) || __hip_atomic_load(fs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0 || __hip_atomic_load(xr_me + kXrAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0.0) { CGP_ABORT = 1; break; }
                        }
                    }
                }
                double* const redR = kFast ? red3 : red;
                double r0, r1; wave_sum8(rv, r0, r1);
                wave_sum8_store<kSolveThreads / 64>(r0, r1, redR, tid >> 6);
                __syncthreads();
                if (CGP_ABORT) { if (tid == 0) raise_abort(); status = 2; break; }
#pragma unroll
                for (int q = 0; q < kCgpSums; ++q) t[q] = redR[q * (kSolveThreads / 64)];      // (all <= 32 rank granules sit in wavefront 0)
                if (!kFast) __syncthreads();
            }
            if (k == 0) { rhsNorm2 = (float)t[2]; thr = pcg_threshold(rhsNorm2); if (lb == 0 && tid == 0) fs[0] = t[2]; }
            rr_cur = (float)t[2];
            const bool rhs_zero = rhsNorm2 == 0.f;
            stop = force_passes > 0 ? k >= force_passes : (rhs_zero || k == kmax || (k > 0 && rr_cur < thr));
            if (!stop) {
                // alpha = gamma / (delta - beta gamma / alpha_old): the reference's alpha = r.z / p.Ap without the product that is still being gathered
                beta = k > 0 ? t[0] / gamma_old : 0.0;
                alpha = t[0] / (k > 0 ? t[1] - beta * t[0] / alpha : t[1]);
                gamma_old = t[0];
            }
        }
        SOLVE_STAMP(3);
        if (force_passes > 0 && k == 9 && tid == 0) fs[16 + 256 + lb] = fs[16 + 512 + lb] = (double)wall_clock64();      // ... the sums of pass 9 of all the others seen
        if (stop) break;
        if (force_passes == -7 && k == 2 && lb == 1) { status = 2; break; }      // fault injection: this workgroup never publishes pass 3
        // ---- D: the update (k = -1: w_0 = n), the three sums and m for the next pass
        double s[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] = 0.0;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            double ui = (double)inv[u] * r[u];
            if (k < 0) w[u] = nres[u];
            else {
                z[u] = nres[u] + beta * z[u];
                sv[u] = w[u] + beta * sv[u];
                pv[u] = ui + beta * pv[u];
                x[u] = x[u] + alpha * pv[u];
                r[u] = r[u] - alpha * sv[u];
                w[u] = w[u] - alpha * z[u];
                ui = (double)inv[u] * r[u];
            }
            const double mnext = (double)inv[u] * w[u];
            if (live[u]) {
                s[0] += r[u] * ui; s[1] += w[u] * ui; s[2] += r[u] * r[u];
                double* dst = recd[(k + 1) & 1] + row[u];
                // readers on this XCD hit the line in its L2 if it is DIRTY there (a plain store: the per-pass acquire only drops clean lines); readers
                // on another XCD need it in memory (write-through store).  A workgroup with neighbours on both sides does both.
                if (TM) { const double mt = m_tag(mnext, mepoch | ((unsigned)(k + 2) & 3u)); __hip_atomic_store(dst, mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); push_record((k + 1) & 1, row[u] - a.row0, mt); }
                else {
                    if (!xcd_local) store8_sc1(dst, mnext);
                    *dst = mnext;
                    push_record((k + 1) & 1, row[u] - a.row0, mnext);
                }
            }
        }
        SOLVE_STAMP(4);
        // ---- E: publish: m has to be out (drained) before the tagged sums
        const double s4[4] = {s[0], s[1], s[2], 0.0};
        const double tE = wave_sum4(s4);
        if (!TM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        double* const redE = kFast ? red2 + ((k + 1) & 1) * (8 * kSolveThreads / 64) : red;
        wave_sum4_store<kSolveThreads / 64>(tE, redE, tid >> 6);
        __syncthreads();
        if (TM && CGP_ABORT) { if (tid == 0) raise_abort(); status = 2; break; }      // (a gather of this pass gave up)
        if (kFast && tid == 0) s_ab3[(k + 5) % 3] = 0;      // the flag of pass k + 2: its readers (pass k - 1) are past this barrier, its raisers two barriers away
        SOLVE_STAMP(5);
        if (tid < kCgpSums) {
            double tot = 0;
            for (int i = 0; i < kSolveThreads / 64; ++i) tot += redE[tid * (kSolveThreads / 64) + i];
            double* gq = gran + (size_t)((k + 1) & 1) * kSolveGranPlanes * kSolveMaxBlocks + (size_t)tid * kSolveMaxBlocks + lb;
            __hip_atomic_store(gq, gran_tag(tot, (unsigned)(k + 2) & 3u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MR && !TM && tid == 64) peer_tag((k + 1) & 1, xr_tag(1.0, etag0 | ((unsigned)(k + 2) & 3u)));
        if (!kFast) __syncthreads();
        SOLVE_STAMP(6);
        if (force_passes > 0 && k == 8 && tid == 0) fs[16 + lb] = (double)wall_clock64();      // ... published at the end of pass 8 (m_9 and the sums of pass 9)
        if (force_passes > 0 && k == 9 && tid == 0) fs[16 + 1280 + lb] = (double)wall_clock64();  // ... and at the end of pass 9
    }
#undef SOLVE_STAMP
#undef CGP_ABORT
    // ---- leave: x of the own rows; the distance update; the outcome for the host and for the gated kernels behind this one
    float xf[R];
#pragma unroll
    for (int u = 0; u < R; ++u) { xf[u] = (float)x[u]; if (live[u]) b.x[row[u]] = xf[u]; }
    if (a.pcg_apply) {
        const bool z0 = rhsNorm2 == 0.f;
        const bool ok_all = status == 1 && (z0 || sqrt((double)rr_cur / (double)rhsNorm2) <= (double)FLT_EPSILON);
        const bool apply = status == 1 && (a.pcg_apply == 1 || ok_all);
        double cnt = 0;
        if (apply) {
#pragma unroll
            for (int u = 0; u < R; ++u) if (live[u] && (double)fabsf(xf[u]) < sqrt(3.0) * (double)a.grid.vs) { b.dist[row[u]] -= xf[u]; cnt += 1.0; }
        }
        cnt = wave_sum(cnt);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = cnt;
        __syncthreads();
        double* slot = PART(a, SC_ACCEPT);
        if (tid == 0) { double t = 0; for (int i = 0; i < kSolveThreads / 64; ++i) t += red[i]; slot[lb] = t; }
        for (int i = G + lb * kSolveThreads + tid; i < a.acc.PB; i += G * kSolveThreads) slot[i] = 0.0;
        // (Measured and rejected, profiles/r04_notes.md: the regrad -- k_derive -- in this epilogue behind one more neighbour hand-off.  Same bits, no
        // launch, but 2 648 vs 2 680 it/s: with one workgroup of eight waves per CU the dependent loads of three rows per thread take longer than the
        // 13 us the stand-alone kernel needs at full occupancy.)
    }
    if (lb == 0 && tid == 0) {
        const bool rhs_zero = rhsNorm2 == 0.f;
        const bool ok = status == 1 && (rhs_zero || sqrt((double)rr_cur / (double)rhsNorm2) <= (double)FLT_EPSILON);
        int iters = 0;
        if (!rhs_zero && k > 0) iters = (rr_cur < thr) ? k - 1 : k;
        fs[1] = status == 1 ? (double)(k + 1) : 0.0; fs[2] = ok ? 1.0 : 0.0;
        const double m0 = (double)iters, m1 = (double)rr_cur, m2 = (double)rhsNorm2, m3 = (double)status;
        mb[0] = m0; mb[1] = m1; mb[2] = m2;
        __threadfence_system();
        mb[3] = m3;
        if (mb_key) reinterpret_cast<unsigned long long*>(mb)[4] = (unsigned long long)(__double_as_longlong(m0) ^ __double_as_longlong(m1) ^ __double_as_longlong(m2) ^ __double_as_longlong(m3)) ^ mb_key;
        __threadfence_system();
    }
}
static size_t cgf_solve_lds(int rows) { return sizeof(float) * (size_t)rows * kNQ * kSolveThreads; }
static size_t cgp_solve_lds(int rows) { return cgf_solve_lds(rows); }
template <int R, bool ASM, bool MR> static int cgf_solve_prepare() {      // > 64 KB of dynamic LDS has to be asked for, once per instance
    static int per_cu = -1;
    if (per_cu >= 0) return per_cu;
    per_cu = 0;
    if (hipFuncSetAttribute((const void*)k_cgf_solve<R, ASM, MR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cgf_solve_lds(R)) != hipSuccess) return per_cu;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_cgf_solve<R, ASM, MR>, kSolveThreads, cgf_solve_lds(R)) == hipSuccess) per_cu = n;
    return per_cu;
}
template <int R, bool MR, bool TM = false> static int cgp_solve_prepare() {
    static int per_cu = -1;
    if (per_cu >= 0) return per_cu;
    per_cu = 0;
    if (hipFuncSetAttribute((const void*)k_cgp_solve<R, MR, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cgp_solve_lds(R)) != hipSuccess) return per_cu;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_cgp_solve<R, MR, TM>, kSolveThreads, cgp_solve_lds(R)) == hipSuccess) per_cu = n;
    return per_cu;
}
template <int R> static int cgf_solve_prepare_both() {
    const int classic = std::min(std::min(cgf_solve_prepare<R, false, false>(), cgf_solve_prepare<R, true, false>()), cgf_solve_prepare<R, true, true>());
    if (R > kCgpMaxRows) return classic;
    return std::min(classic, std::min(std::min(cgp_solve_prepare<R, false>(), cgp_solve_prepare<R, false, true>()), std::min(cgp_solve_prepare<R, true>(), cgp_solve_prepare<R, true, true>())));
}
int cgf_solve_max_blocks(int rows) {
    return rows == 1 ? cgf_solve_prepare_both<1>() : rows == 2 ? cgf_solve_prepare_both<2>() : rows == 3 ? cgf_solve_prepare_both<3>() : cgf_solve_prepare_both<4>();
}
template <int R>
static void launch_cgf_solve_r(const SweepArgs& a, double* fs, double* gran, int G, int rows_per_wg, int kmax, double* mb, unsigned long long mb_key, int force_passes, hipStream_t s, const XrArgs* xr) {
    XrArgs none{};
    if (a.pcg_asm && a.pcg_pipe && R <= kCgpMaxRows) {      // the pipelined recurrences (always with the fused assembly)
        if (xr && xr->n_ranks > 1 && a.pcg_pipe == 3) hipLaunchKernelGGL((k_cgp_solve<R, true, true>), dim3(G), dim3(kSolveThreads), cgp_solve_lds(R), s, a, fs, gran, rows_per_wg, kmax, mb, mb_key, force_passes, *xr);      // ... across ranks too (PSGSDF_PCG_TAGM=2)
        else if (xr && xr->n_ranks > 1) hipLaunchKernelGGL((k_cgp_solve<R, true>), dim3(G), dim3(kSolveThreads), cgp_solve_lds(R), s, a, fs, gran, rows_per_wg, kmax, mb, mb_key, force_passes, *xr);
        else if (a.pcg_pipe >= 2) { hipLaunchKernelGGL((k_cgp_solve<R, false, true>), dim3(G), dim3(kSolveThreads), cgp_solve_lds(R), s, a, fs, gran, rows_per_wg, kmax, mb, mb_key, force_passes, none); }      // self-validating m (PSGSDF_PCG_TAGM)
        else hipLaunchKernelGGL((k_cgp_solve<R, false>), dim3(G), dim3(kSolveThreads), cgp_solve_lds(R), s, a, fs, gran, rows_per_wg, kmax, mb, mb_key, force_passes, none);
        return;
    }
    if (xr && xr->n_ranks > 1 && a.pcg_asm) hipLaunchKernelGGL((k_cgf_solve<R, true, true>), dim3(G), dim3(kSolveThreads), cgf_solve_lds(R), s, a, fs, gran, rows_per_wg, kmax, mb, mb_key, force_passes, *xr);
    else if (a.pcg_asm) hipLaunchKernelGGL((k_cgf_solve<R, true, false>), dim3(G), dim3(kSolveThreads), cgf_solve_lds(R), s, a, fs, gran, rows_per_wg, kmax, mb, mb_key, force_passes, none);
    else hipLaunchKernelGGL((k_cgf_solve<R, false, false>), dim3(G), dim3(kSolveThreads), cgf_solve_lds(R), s, a, fs, gran, rows_per_wg, kmax, mb, mb_key, force_passes, none);
}
void launch_cgf_solve(const SweepArgs& a, double* fs, double* gran, int G, int rows_per_wg, int kmax, double* mb, unsigned long long mb_key, int force_passes, hipStream_t s, const XrArgs* xr) {
    const int rows = (rows_per_wg + kSolveThreads - 1) / kSolveThreads;
    if (rows == 1) launch_cgf_solve_r<1>(a, fs, gran, G, rows_per_wg, kmax, mb, mb_key, force_passes, s, xr);
    else if (rows == 2) launch_cgf_solve_r<2>(a, fs, gran, G, rows_per_wg, kmax, mb, mb_key, force_passes, s, xr);
    else if (rows == 3) launch_cgf_solve_r<3>(a, fs, gran, G, rows_per_wg, kmax, mb, mb_key, force_passes, s, xr);
    else launch_cgf_solve_r<4>(a, fs, gran, G, rows_per_wg, kmax, mb, mb_key, force_passes, s, xr);
}

// multi-rank: fold the partials of pass k (k = -1: |b|^2 of the init) into out[0..6] for the host program's all-reduce
__global__ void __launch_bounds__(kBlock) k_cgf_sum(double* part, int G, int k, double* __restrict__ out) {
    __shared__ double red[8 * kBlock / 64];
    if (k < 0) {
        double* src[1] = {fpart(part, -1, 6)}; double bb;
        block_total_n<1>(src, G, red, &bb);
        if (threadIdx.x == 0) out[0] = bb;
    } else {
        double* src[kCgfSums]; double t[kCgfSums];
#pragma unroll
        for (int q = 0; q < kCgfSums; ++q) src[q] = fpart(part, k, q);
        block_total_n<kCgfSums>(src, G, red, t);
        if (threadIdx.x < kCgfSums) out[threadIdx.x] = t[threadIdx.x];
    }
}
void launch_cgf_sum(double* part, int G, int k, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_cgf_sum, dim3(1), dim3(kBlock), 0, s, part, G, k, out);
}

// debug: y = H x without damping
__global__ void __launch_bounds__(kBlock) k_matvec(SweepArgs a, const float* x, float* y) {
    const Band& b = a.b;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.S) return;
    double acc = 0;
    for (int q = 0; q < kNQ; ++q) {
        int c = q == 0 ? i : b.col[(size_t)q * b.Spad + i];
        if (c < 0) continue;
        acc += (double)b.H[(size_t)q * b.Spad + i] * (double)x[c];
    }
    y[i] = (float)acc;
}
void launch_matvec(const SweepArgs& a, const float* x, float* y, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_matvec, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, x, y);
}
// updateDist accept rule OptimizerAux.cpp:162-188
__global__ void __launch_bounds__(kBlock) k_apply_dist(SweepArgs a) {
    __shared__ double red[kBlock / 64];
    if (a.gate && *a.gate == 0.0) return;       // launched speculatively behind a PCG chunk that did not finish the solve
    const Band& b = a.b;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double cnt = 0;
    if (j < a.row1) {
        float d = b.x[j];
        if ((double)fabsf(d) < sqrt(3.0) * (double)a.grid.vs) { b.dist[j] -= d; cnt = 1.0; }
    }
    block_part_store(cnt, PART(a, SC_ACCEPT), red);
}
void launch_apply_dist(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_apply_dist, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}

}  // namespace psg
