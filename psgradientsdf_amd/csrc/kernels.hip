// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the Gradient-SDF photometric-stereo
// hot path.  No CUDA compatibility layer, no other back end.
//
// Reference arithmetic is cited per function (paths relative to /root/reference/cpp/include/).
// Per-observation arithmetic is float32 in the reference's operation order; FMA contraction is
// off in this file so projections / bilinear weights are bit-compatible with the CPU reference
// build (baseline x86-64, no FMA).  Sums over many observations are accumulated in double.
//
// Kernel map:
//   dense grid : k_select_vis, k_band_flags, k_scan_*, k_band_fill, k_band_nb, k_band_scatter, k_upsample
//   per voxel  : k_derive (FD gradient / surface point / Eikonal+Laplacian energies), k_init_albedo,
//                k_energy, k_sweep_albedo, k_sweep_dist            (voxel-major, set-bit iteration)
//   per frame  : k_sweep_light, k_sweep_pose                        (frame-major, wave+LDS reduction)
//   solves     : k_solve_light, k_solve_pose (LDL^T per frame), k_assemble, k_cgf_init/pass (fused Jacobi-PCG)
#include "engine.h"
#include <float.h>

// Floating-point contraction: ON for the Jacobian / normal-equation algebra (FMA: fewer instructions, one rounding
// less), OFF inside the functions whose results feed discrete decisions or must match the CPU reference build bit for
// bit (baseline x86-64, no FMA): normalisation, the surface point, the projection (floor / in-image test), FD gradient.
#pragma clang fp contract(fast)

namespace psg {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ float norm3(const float* a) { return sqrtf(dot3(a, a)); }
// Eigen normalized(): z>0 ? v/sqrt(z) : v
__device__ __forceinline__ void normalized3(const float* v, float* o) {
#pragma clang fp contract(off)
    float z = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    if (z > 0.f) { float s = sqrtf(z); o[0] = v[0] / s; o[1] = v[1] / s; o[2] = v[2] / s; }
    else { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; }
}
__device__ __forceinline__ void mulT3(const float* M, const float* v, float* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = (M[0 * 3 + i] * v[0] + M[1 * 3 + i] * v[1]) + M[2 * 3 + i] * v[2];
}
__device__ __forceinline__ void mul3(const float* M, const float* v, float* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = (M[i * 3 + 0] * v[0] + M[i * 3 + 1] * v[1]) + M[i * 3 + 2] * v[2];
}
template <int NB> __device__ __forceinline__ float dotn(const float* a, const float* b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) s += a[i] * b[i];
    return s;
}
// PsOptimizerJa.cpp:17-28
template <int NB> __device__ __forceinline__ void SH(const float* n, float* sh) {
    sh[0] = 1.0f; sh[1] = n[0]; sh[2] = n[1]; sh[3] = n[2];
    if (NB == 9) { sh[4] = n[0] * n[1]; sh[5] = n[0] * n[2]; sh[6] = n[1] * n[2]; sh[7] = n[0] * n[0] - n[1] * n[1]; sh[8] = n[0] * n[0] - n[2] * n[2]; }
}
template <int MODEL> struct ModelTraits { static constexpr int NB = MODEL == 1 ? 9 : (MODEL == 0 ? 4 : 3); static constexpr bool LED = MODEL == 2; };

// Optimizer.cpp:140-161 / 164-186
__device__ __forceinline__ float robust_weight(const Robust& rb, float r) {
    switch (rb.loss) {
        case 1: { float x = r * rb.inv_lambda; return __builtin_amdgcn_rcpf(1.0f + x * x); }   // v_rcp_f32: 1 ulp
        case 3: { float x = r * rb.inv_lambda; float w = (1.0f - x * x); w = w * w; return (r * r < rb.lambda_sq) ? w : 0.0f; }
        case 2: { float w = rb.lambda * fabsf(__builtin_amdgcn_rcpf(r)); return (r * r < rb.lambda_sq) ? 1.0f : w; }
        case 4: return (r * r < rb.lambda_sq) ? 1.0f : 0.0f;
        default: return 1.0f;
    }
}
__device__ __forceinline__ float robust_loss(const Robust& rb, float r) {
    switch (rb.loss) {
        case 1: { float x = r * rb.inv_lambda; return __logf(1.0f + x * x); }
        case 3: { float x = r * rb.inv_lambda; float u = 1.0f - x * x; float v = 1.0f - u * u * u; return (r * r < rb.lambda_sq) ? v : 1.0f; }
        case 2: return (r * r < rb.lambda_sq) ? 0.5f * (r * r) : rb.lambda * (fabsf(r) - 0.5f * rb.lambda * 1.0f);
        case 4: { float x = fminf(fmaxf(r, -rb.lambda), rb.lambda); return x * x; }
        default: return r * r;
    }
}

// wavefront (64 lanes) and workgroup reductions; one device-scope atomic per workgroup
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sumf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// Workgroup reduction -> ONE plain store per workgroup into a per-block partial slot (no atomics: 1300+
// workgroups hitting one address serialise at ~12 ns each, which made trivial kernels take 30 us).
// All threads of the block must call; red is __shared__ double[kBlock/64].  The partials are summed by the
// consumer (host, or the next PCG kernel) in a fixed order, so results are run-to-run deterministic.
__device__ __forceinline__ void block_part_store(double v, double* part_slot, double* red) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
        part_slot[blockIdx.x] = s;
    }
}
// sum of n partials, identical in every thread of every block (fixed order)
__device__ __forceinline__ double block_total(const double* part, int n, double* red) {
    double v = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += part[i];
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}
#define PART(a, slot) ((a).acc.part + (size_t)(slot) * (a).acc.PB)
// a.fold: sum the partial slots the PREVIOUS kernel left behind (first workgroup only; all its threads must call).
// The calling kernel must not write the folded slots itself (engine.hip: take_fold checks).
__device__ __forceinline__ void fold_pending(const SweepArgs& a, double* red /*[kBlock/64]*/) {
    if (a.fold.n == 0 || blockIdx.x != 0 || blockIdx.y != 0) return;
    for (int s = 0; s < a.fold.n; ++s) {
        const double t = block_total(PART(a, a.fold.id[s]), a.fold.nblk, red);
        if (threadIdx.x == 0) a.fold.out[s] = t;
    }
    __syncthreads();
}

// frame records (pose, light) of all keyframes staged in dynamic LDS: F * 96 B (<= 60 KiB, F <= kMaxFramesLds)
extern __shared__ __align__(16) unsigned char psg_dyn_smem[];
__device__ __forceinline__ void load_frames(FrameP* sf, const FrameP* frames, int F) {
    const float* src = (const float*)frames; float* dst = (float*)sf;
    for (int i = threadIdx.x; i < F * (int)(sizeof(FrameP) / 4); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// projection + image sampling
// ------------------------------------------------------------------------------------------
struct Proj { float p[3]; float m, n, z_inv; bool ok; };

// OptimizerAux.cpp:207-226 (surface point precomputed in xs = x_v - d*normalized(grad))
__device__ __forceinline__ Proj project(const float* xs, const FrameP& fp, const Cam& cam) {
#pragma clang fp contract(off)
    Proj o;
    float tmp[3] = {xs[0] - fp.t[0], xs[1] - fp.t[1], xs[2] - fp.t[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) o.p[i] = (fp.R[0 * 3 + i] * tmp[0] + fp.R[1 * 3 + i] * tmp[1]) + fp.R[2 * 3 + i] * tmp[2];
    // reference: (float)(1. / point[2]) evaluated in double (OptimizerAux.cpp:219); the correctly rounded float
    // reciprocal differs from that only in double-rounding corner cases (~1e-8 of all inputs)
    const float z_inv = 1.0f / o.p[2];
    o.z_inv = z_inv;
    o.m = cam.fx * o.p[0] * z_inv + cam.cx;
    o.n = cam.fy * o.p[1] * z_inv + cam.cy;
    o.ok = (o.m >= 0.f && o.m < (float)cam.W && o.n >= 0.f && o.n < (float)cam.H);
    return o;
}

__device__ __forceinline__ const float* pix(const float* img, const Cam& cam, int row, int col) {
    row = row < 0 ? 0 : (row >= cam.H ? cam.H - 1 : row);
    col = col < 0 ? 0 : (col >= cam.W ? cam.W - 1 : col);
    return img + ((size_t)row * cam.W + col) * 3;
}

// Auxilary.h:41-61 interpolateImage + Auxilary.h:64-123 computeImageGradient from one set of taps.
// (row coordinate n_row, column coordinate m_col); gu = d/d(col), gv = d/d(row).
// `base` is wave-uniform (the image stack, or one frame of it), `frame` selects the image inside it (0 for a frame pointer).
// idx32: the whole stack is < 4 GiB, so a tap's byte offset fits 32 bits and the loads take the scalar-base form (one address
// register, no 64-bit integer multiply-adds, which issue at quarter rate and made up ~10 % of a sweep's instruction slots).
template <bool GRAD>
__device__ __forceinline__ void sample(const float* base, int frame, bool idx32, const Cam& cam, float m_col, float n_row, float* I, float* gu, float* gv) {
    const float m = n_row, n = m_col;  // names of Auxilary.h: m = row, n = column
    int x = (int)floorf(m), y = (int)floorf(n);
    const float* img = base + (size_t)frame * cam.H * cam.W * 3;   // only the rare border path below uses it
    if ((x + 1) < cam.H && (y + 1) < cam.W) {
        float a00[3], a01[3], a10[3], a11[3];
        if (idx32) {
            const unsigned e = (((unsigned)frame * (unsigned)cam.H + (unsigned)x) * (unsigned)cam.W + (unsigned)y) * 3u;
            const float* p00 = (const float*)((const char*)base + (size_t)(e << 2));
            const float* p10 = (const float*)((const char*)base + (size_t)((e + 3u * (unsigned)cam.W) << 2));
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { a00[ch] = p00[ch]; a01[ch] = p00[3 + ch]; a10[ch] = p10[ch]; a11[ch] = p10[3 + ch]; }
        } else {
            const float* p00 = img + ((size_t)x * cam.W + y) * 3;
            const float* p10 = p00 + (size_t)cam.W * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { a00[ch] = p00[ch]; a01[ch] = p00[3 + ch]; a10[ch] = p10[ch]; a11[ch] = p10[3 + ch]; }
        }
        // reference: weights partly in double (Auxilary.h:47); float weights agree to ~1e-7 relative
        const float fm = m - (float)x, fn = n - (float)y;
        const float w1 = (1.0f - fn) * fm, w2 = (1.0f - fn) * (1.0f - fm), w3 = fn * fm, w4 = fn * (1.0f - fm);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) I[ch] = ((a10[ch] * w1 + a00[ch] * w2) + a11[ch] * w3) + a01[ch] * w4;
        if (GRAD) {
            const float w01 = fm, w11 = fn, w00 = 1.0f - fm, w10 = 1.0f - fn;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                gu[ch] = w00 * (a01[ch] - a00[ch]) + w01 * (a11[ch] - a10[ch]);
                gv[ch] = w10 * (a10[ch] - a00[ch]) + w11 * (a11[ch] - a01[ch]);
            }
        }
    } else {  // last row / column: nearest sample, one-sided differences (Auxilary.h:55-57,90-121)
        const float* p = pix(img, cam, x, y);
        I[0] = p[0]; I[1] = p[1]; I[2] = p[2];
        if (GRAD) {
            float w01 = m - (float)x, w11 = n - (float)y;
            float w00 = (float)(1.0 - (double)w01), w10 = (float)(1.0 - (double)w11);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                if ((x + 1) >= cam.H) gu[ch] = pix(img, cam, x, y + 1)[ch] - pix(img, cam, x, y)[ch];
                else { float v0 = -pix(img, cam, x, y - 1)[ch] + pix(img, cam, x, y)[ch]; float v1 = -pix(img, cam, x + 1, y - 1)[ch] + pix(img, cam, x + 1, y)[ch]; gu[ch] = w00 * v0 + w01 * v1; }
                if ((x + 1) >= cam.H && (y + 1) < cam.W) { float v0 = -pix(img, cam, x - 1, y)[ch] + pix(img, cam, x, y)[ch]; float v1 = -pix(img, cam, x - 1, y + 1)[ch] + pix(img, cam, x, y + 1)[ch]; gv[ch] = w10 * v0 + w11 * v1; }
                else gv[ch] = pix(img, cam, x + 1, y)[ch] - pix(img, cam, x, y)[ch];
            }
        }
    }
}

// rendered intensity: PsOptimizerJa.cpp:30-40 (SH) / LedOptimizerJa.cpp:15-29 (LED).
// nfd = normalized FD gradient, shfd = SH(nfd) (SH models only)
template <int MODEL>
__device__ __forceinline__ void rendered(const FrameP& fp, const Proj& pr, const float* nfd, const float* shfd, const float* rho, float* out) {
    constexpr int NB = ModelTraits<MODEL>::NB;
    float irr;
    if (ModelTraits<MODEL>::LED) {
        float Rp[3]; mul3(fp.R, pr.p, Rp);
        irr = -dot3(nfd, Rp);
        float pn = norm3(pr.p); double pd = (double)pn;
        float ld = (float)(pd * pd * pd);
        irr /= ld;
        out[0] = rho[0] * fp.l[0] * irr; out[1] = rho[1] * fp.l[1] * irr; out[2] = rho[2] * fp.l[2] * irr;
    } else {
        irr = dotn<NB>(fp.l, shfd);
        out[0] = rho[0] * irr; out[1] = rho[1] * irr; out[2] = rho[2] * irr;
    }
}

// rhoJacobian: PsOptimizerJa.cpp:118-122 / LedOptimizerJa.cpp:85-99 (stored normal gn)
template <int MODEL>
__device__ __forceinline__ void rho_jac(const FrameP& fp, const Proj& pr, const float* gn, const float* shg, float* J) {
    constexpr int NB = ModelTraits<MODEL>::NB;
    if (ModelTraits<MODEL>::LED) {
        float Rp[3]; mul3(fp.R, pr.p, Rp);
        float refl = dot3(gn, Rp);
        float pn = norm3(pr.p); double pd = (double)pn;
        refl /= (float)(pd * pd * pd);
        J[0] = refl * fp.l[0]; J[1] = refl * fp.l[1]; J[2] = refl * fp.l[2];
    } else {
        float j = -dotn<NB>(fp.l, shg);
        J[0] = J[1] = J[2] = j;
    }
}

// per-voxel state in registers
struct Vox { float xs[3], gn[3], nfd[3], rho[3]; };
__device__ __forceinline__ void load_vox(const Band& b, int j, Vox& v) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { v.xs[a] = b.xs[a][j]; v.gn[a] = b.gn[a][j]; v.rho[a] = b.rho[a][j]; v.nfd[a] = b.nfd[a][j]; }
}

// ------------------------------------------------------------------------------------------
// dense-grid kernels: visibility selection, band construction, scatter, 2x refinement
// ------------------------------------------------------------------------------------------

// Optimizer.cpp:30-47 select_vis
__global__ void __launch_bounds__(kBlock) k_select_vis(const uint64_t* __restrict__ vis_seq, int wpv_seq, uint64_t* __restrict__ vis_key, int KW, const int* __restrict__ frame_idx, int F, long long nvox) {
    for (long long lin = blockIdx.x * (long long)blockDim.x + threadIdx.x; lin < nvox; lin += (long long)gridDim.x * blockDim.x) {
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

// ELL column offsets of one assembled distance row: self, 6 axis neighbours, 12 axis pairs
__host__ __device__ inline void q_offset(int q, int* o) {
    o[0] = o[1] = o[2] = 0;
    if (q == 0) return;
    if (q <= 6) { int a = (q - 1) >> 1; o[a] = ((q - 1) & 1) ? -1 : 1; return; }
    // q 7..12: mixed-sign axis pairs (the only pair columns a forward-only stencil produces), q 13..18: (+,+) / (-,-)
    int pi, sa, sb;
    if (q < kNQCommon) { pi = (q - 7) >> 1; sa = ((q - 7) & 1) ? -1 : 1; sb = -sa; }
    else { pi = (q - kNQCommon) >> 1; sa = ((q - kNQCommon) & 1) ? -1 : 1; sb = sa; }
    int a = pi == 2 ? 1 : 0, b = pi == 0 ? 1 : 2;
    o[a] = sa; o[b] = sb;
}
__device__ __forceinline__ int q_of(const int* o) {
    int nz = (o[0] != 0) + (o[1] != 0) + (o[2] != 0);
    if (nz == 0) return 0;
    if (nz == 1) { int a = o[0] ? 0 : (o[1] ? 1 : 2); return 1 + 2 * a + (o[a] < 0 ? 1 : 0); }
    int a = o[0] ? 0 : 1, b = o[2] ? 2 : 1;
    int pi = (a == 0 && b == 1) ? 0 : ((a == 0) ? 1 : 2);
    if (o[a] != o[b]) return 7 + 2 * pi + (o[a] < 0 ? 1 : 0);
    return kNQCommon + 2 * pi + (o[a] < 0 ? 1 : 0);
}

// gather the compact band planes from the dense grid (one thread per dense voxel, coalesced reads)
__global__ void __launch_bounds__(kBlock) k_band_fill(DenseView d, GridP grid, Band b) {
    for (long long lin = blockIdx.x * (long long)blockDim.x + threadIdx.x; lin < grid.nvox; lin += (long long)gridDim.x * blockDim.x) {
        int j = d.row_of[lin];
        if (j < 0) continue;
        b.lin[j] = (int)lin;
        b.dist[j] = d.dist[lin];
#pragma unroll
        for (int a = 0; a < 3; ++a) { b.g[a][j] = d.g[a][lin]; b.rho[a][j] = d.rho[a][lin]; }
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
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        long long ln = lin + ((q & 1) ? -stride[q >> 1] : stride[q >> 1]);
        bool in = ln >= 0 && ln < grid.nvox;
        b.nb[(size_t)q * b.Spad + j] = in ? d.row_of[ln] : -1;
        b.nbd[(size_t)q * b.Spad + j] = in ? d.dist[ln] : dj;   // reference reads out of bounds here (UB): use own value
    }
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
// VolumetricGradSdf::update, VolumetricGradSdf.cpp:51-138 (+ truncate / weight, Sdf.h:44-66): fuse one RGB-D frame.
// One thread per voxel of the dense grid, x fastest: 8 coalesced float planes + one visibility word are read and,
// for the voxels the frame sees, written back.  HBM-bound: 36 B read + up to 40 B written per voxel.
__global__ void __launch_bounds__(kBlock) k_integrate(DenseView d, uint64_t* __restrict__ vis_seq, int wpv_seq, GridP g, Cam cam, FrameP fp,
                                                      const float* __restrict__ rgb, const float* __restrict__ depth, const float* __restrict__ normals,
                                                      int counter, float z_min, float z_max) {
#pragma clang fp contract(off)
    const float T = g.T, inv_T = (float)(1.0 / (double)g.T);
    const double fx_inv = 1.0 / (double)cam.fx, fy_inv = 1.0 / (double)cam.fy;
    const size_t npx = (size_t)cam.W * cam.H;
    const long long nxy = (long long)g.dim[0] * g.dim[1];
    for (long long lin = blockIdx.x * (long long)blockDim.x + threadIdx.x; lin < g.nvox; lin += (long long)gridDim.x * blockDim.x) {
        int k = (int)(lin / nxy); int rest = (int)(lin - (long long)k * nxy); int j = rest / g.dim[0]; int i = rest - j * g.dim[0];
        float xv[3] = {g.origin[0] + g.vs * (float)i, g.origin[1] + g.vs * (float)j, g.origin[2] + g.vs * (float)k};
        float tmp[3] = {xv[0] - fp.t[0], xv[1] - fp.t[1], xv[2] - fp.t[2]}, p[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a] = (fp.R[0 * 3 + a] * tmp[0] + fp.R[1 * 3 + a] * tmp[1]) + fp.R[2 * 3 + a] * tmp[2];
        if (p[2] < 0.f) continue;
        const int n = (int)((double)(cam.cx + cam.fx * p[0] / p[2]) + 0.5);
        const int m = (int)((double)(cam.cy + cam.fy * p[1] / p[2]) + 0.5);
        if (n < 0 || n >= cam.W || m < 0 || m >= cam.H) continue;
        const size_t px = (size_t)m * cam.W + n;
        const float z = depth[px];
        if (z <= z_min || z >= z_max) continue;
        const float sdf = z - p[2];
        float w = 0.f;
        if (sdf >= 0.) w = 1.f; else if (sdf >= -T) w = 1.f + sdf * inv_T;
        if (w == 0) continue;
        float nrm[3] = {normals[px], normals[npx + px], normals[2 * npx + px]};
        if ((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2] < .1) continue;
        const float zi = (float)(1. / (double)p[2]);
        const float xy[3] = {zi * p[0], zi * p[1], zi * p[2]};
        const double x0 = fx_inv * ((double)n - (double)cam.cx), y0 = fy_inv * ((double)m - (double)cam.cy);
        const float n_sq_inv = (float)(1.0 / (1.0 + x0 * x0 + y0 * y0));
        const float dn = (nrm[0] * xy[0] + nrm[1] * xy[1]) + nrm[2] * xy[2];
        if (dn * dn * n_sq_inv < .25 * .25) continue;   // normal more than 75.5 deg off the viewing ray
        const float wsum = d.weight[lin] + w;
        d.weight[lin] = wsum;
        const float ts = fmaxf(-T, fminf(T, sdf));
        const float dv = d.dist[lin];
        d.dist[lin] = dv + (ts - dv) * w / wsum;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float Rn = (fp.R[a * 3 + 0] * nrm[0] + fp.R[a * 3 + 1] * nrm[1]) + fp.R[a * 3 + 2] * nrm[2];
            d.g[a][lin] -= w * Rn;
            float cv = d.rho[a][lin];
            d.rho[a][lin] = cv + (rgb[px * 3 + a] - cv) * w / wsum;
        }
        vis_seq[lin * wpv_seq + (counter >> 6)] |= 1ull << (counter & 63);
    }
}
void launch_integrate(const DenseView& d, uint64_t* vis_seq, int wpv_seq, const GridP& g, const Cam& cam, const FrameP& fp,
                      const float* rgb, const float* depth, const float* normals, int counter, float z_min, float z_max, hipStream_t s) {
    int grid = (int)min((g.nvox + kBlock - 1) / kBlock, (long long)256 * 32);
    hipLaunchKernelGGL(k_integrate, dim3(grid), dim3(kBlock), 0, s, d, vis_seq, wpv_seq, g, cam, fp, rgb, depth, normals, counter, z_min, z_max);
}

// ------------------------------------------------------------------------------------------
// front end: FALS normals (normals/NormalEstimator.h:150-176) and the depth tracker reduction
// (sdf_tracker/RigidPointOptimizer.cpp:38-60).  Image-space, once per input frame.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; } return i; }
// horizontal pass of the un-normalised (2r+1)^2 box filter over the 3 products cache_q * (1/z); double sums like OpenCV
__global__ void __launch_bounds__(kBlock) k_normals_h(const float* __restrict__ depth, const float* __restrict__ cache, int W, int H, int r, double* __restrict__ tmp) {
    const size_t n = (size_t)W * H;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        double s0 = 0, s1 = 0, s2 = 0;
        for (int k = -r; k <= r; ++k) {
            size_t q = (size_t)y * W + reflect101(x + k, W);
            float z = depth[q]; float zi = z != 0.f ? 1.0f / z : 0.f;
            s0 += (double)(cache[q] * zi); s1 += (double)(cache[n + q] * zi); s2 += (double)(cache[2 * n + q] * zi);
        }
        tmp[p] = s0; tmp[n + p] = s1; tmp[2 * n + p] = s2;
    }
}
__global__ void __launch_bounds__(kBlock) k_normals_v(const double* __restrict__ tmp, const float* __restrict__ cache, int W, int H, int r, float* __restrict__ out) {
#pragma clang fp contract(off)
    const size_t n = (size_t)W * H;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        double s0 = 0, s1 = 0, s2 = 0;
        for (int k = -r; k <= r; ++k) { size_t q = (size_t)reflect101(y + k, H) * W + x; s0 += tmp[q]; s1 += tmp[n + q]; s2 += tmp[2 * n + q]; }
        const float b1 = (float)s0, b2 = (float)s1, b3 = (float)s2;
        const float Q11 = cache[3 * n + p], Q12 = cache[4 * n + p], Q13 = cache[5 * n + p], Q22 = cache[6 * n + p], Q23 = cache[7 * n + p], Q33 = cache[8 * n + p];
        float nx = b1 * Q11 + b2 * Q12 + b3 * Q13, ny = b1 * Q12 + b2 * Q22 + b3 * Q23, nz = b1 * Q13 + b2 * Q23 + b3 * Q33;
        float nn = sqrtf(nx * nx + ny * ny + nz * nz);
        out[p] = nx / nn; out[n + p] = ny / nn; out[2 * n + p] = nz / nn;
    }
}
void launch_normals(const float* depth, const float* cache, int W, int H, int r, double* tmp, float* out, hipStream_t s) {
    int grid = (int)min(((size_t)W * H + kBlock - 1) / kBlock, (size_t)4096);
    hipLaunchKernelGGL(k_normals_h, dim3(grid), dim3(kBlock), 0, s, depth, cache, W, H, r, tmp);
    hipLaunchKernelGGL(k_normals_v, dim3(grid), dim3(kBlock), 0, s, (const double*)tmp, cache, W, H, r, out);
}
// one Gauss-Newton pass of the tracker: H (21) | g (6) | E | count per workgroup -> partial rows part[blockIdx][29]
__global__ void __launch_bounds__(kBlock) k_track(DenseView d, GridP g, Cam cam, FrameP fp, const float* __restrict__ depth, float z_min, float z_max, double* __restrict__ part) {
#pragma clang fp contract(off)
    __shared__ double lds[(kBlock / 64) * 29];
    const float fx_inv = 1.f / cam.fx, fy_inv = 1.f / cam.fy;
    float acc[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) acc[k] = 0.f;
    const int n = cam.W * cam.H;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const float z = depth[p];
        if (z <= z_min || z >= z_max) continue;
        const int y = p / cam.W, x = p - y * cam.W;
        const float x0 = ((float)x - cam.cx) * fx_inv, y0 = ((float)y - cam.cy) * fy_inv;
        float pc[3] = {x0 * z, y0 * z, z}, pw[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) pw[a] = ((fp.R[a * 3 + 0] * pc[0] + fp.R[a * 3 + 1] * pc[1]) + fp.R[a * 3 + 2] * pc[2]) + fp.t[a];
        float fi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) fi[a] = (pw[a] - g.origin[a]) / g.vs;
        if (fi[0] <= 0 || fi[1] <= 0 || fi[2] <= 0 || fi[0] >= (g.dim[0] - 1) || fi[1] >= (g.dim[1] - 1) || fi[2] >= (g.dim[2] - 1)) continue;
        const int im = (int)(fi[0] + 0.5), jm = (int)(fi[1] + 0.5), km = (int)(fi[2] + 0.5);
        const long long I = (long long)im + (long long)jm * g.dim[0] + (long long)km * g.dim[0] * g.dim[1];
        if (!(d.weight[I] > 0)) continue;
        float gr[3] = {d.g[0][I], d.g[1][I], d.g[2][I]}, gn[3]; normalized3(gr, gn);
        int idx[3] = {(int)(fi[0] + 0.5f), (int)(fi[1] + 0.5f), (int)(fi[2] + 0.5f)};
        float dv[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) dv[a] = (g.origin[a] + g.vs * (float)idx[a]) - pw[a];
        const float phi = d.dist[I] + ((gn[0] * dv[0] + gn[1] * dv[1]) + gn[2] * dv[2]);
        const float gxi[6] = {gn[0], gn[1], gn[2], pw[1] * gn[2] - pw[2] * gn[1], pw[2] * gn[0] - pw[0] * gn[2], pw[0] * gn[1] - pw[1] * gn[0]};
        int q = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int k = i; k < 6; ++k) acc[q++] += gxi[i] * gxi[k];
            acc[21 + i] += phi * gxi[i];
        }
        acc[27] += phi * phi; acc[28] += 1.0f;
    }
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 29; ++k) { double vv = wave_sum((double)acc[k]); if (lane == 0) lds[w * 29 + k] = vv; }
    __syncthreads();
    for (int k = threadIdx.x; k < 29; k += blockDim.x) { double s_ = 0; for (int i = 0; i < kBlock / 64; ++i) s_ += lds[i * 29 + k]; part[(size_t)blockIdx.x * 29 + k] = s_; }
}
void launch_track(const DenseView& d, const GridP& g, const Cam& cam, const FrameP& fp, const float* depth, float z_min, float z_max, double* part, int nblk, hipStream_t s) {
    hipLaunchKernelGGL(k_track, dim3(nblk), dim3(kBlock), 0, s, d, g, cam, fp, depth, z_min, z_max, part);
}
__global__ void k_fill_f32(float* p, float v, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_f32(float* p, float v, long long n, hipStream_t s) {
    int grid = (int)min((n + kBlock - 1) / kBlock, (long long)256 * 16);
    if (n > 0) hipLaunchKernelGGL(k_fill_f32, dim3(grid), dim3(kBlock), 0, s, p, v, n);
}

// ------------------------------------------------------------------------------------------
// per-voxel derived quantities
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float nb_dist(const Band& b, int q, int j) {
    int r = b.nb[(size_t)q * b.Spad + j];
    return r >= 0 ? b.dist[r] : b.nbd[(size_t)q * b.Spad + j];
}
// Optimizer.cpp:287-364 computeDistGrad -> (n, dir)
__device__ __forceinline__ void fd_grad(const Band& b, int j, float vs_inv, float* n, float* dir) {
#pragma clang fp contract(off)
    float d = b.dist[j];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        bool fwd = b.nb[(size_t)(2 * a) * b.Spad + j] >= 0;
        dir[a] = fwd ? 1.0f : -1.0f;
        float dn = fwd ? b.dist[b.nb[(size_t)(2 * a) * b.Spad + j]] : nb_dist(b, 2 * a + 1, j);
        n[a] = dir[a] * (dn - d);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) n[a] = n[a] * vs_inv;
}
// Optimizer.cpp:368-393 computeDistLaplacian
__device__ __forceinline__ float laplacian(const Band& b, int j, float vs_inv) {
    float d = b.dist[j];
    float dd[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { float p1 = nb_dist(b, 2 * a, j), p0 = nb_dist(b, 2 * a + 1, j); dd[a] = p1 + p0 - 2 * d; }
    return (dd[0] + dd[1] + dd[2]) * vs_inv * vs_inv;
}

// FD gradient, optional updateGrad (OptimizerAux.cpp:152-160), surface point, and the Eikonal /
// Laplacian energies (Optimizer.cpp:86-119) in one pass over the band.
__global__ void __launch_bounds__(kBlock) k_derive(SweepArgs a, int update_grad) {
#pragma clang fp contract(off)
    __shared__ double red[kBlock / 64];
    const Band& b = a.b;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double en = 0, el = 0;
    if (j < a.row1) {
        float n[3], dir[3];
        fd_grad(b, j, a.grid.vs_inv, n, dir);
        float g[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { b.gfd[k][j] = n[k]; if (update_grad) b.g[k][j] = n[k]; g[k] = update_grad ? n[k] : b.g[k][j]; }
        float gn[3]; normalized3(g, gn);
        float nn[3]; normalized3(n, nn);
#pragma unroll
        for (int k = 0; k < 3; ++k) b.nfd[k][j] = nn[k];
        long long lin = b.lin[j];
        int nxy = a.grid.dim[0] * a.grid.dim[1];
        int kz = (int)(lin / nxy); int rest = (int)(lin - (long long)kz * nxy); int jy = rest / a.grid.dim[0]; int ix = rest - jy * a.grid.dim[0];
        int idx[3] = {ix, jy, kz};
        float d = b.dist[j];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float xv = a.grid.origin[k] + a.grid.vs * (float)idx[k];   // VoxelGrid.h:38-40
            b.gn[k][j] = gn[k];
            b.xs[k][j] = xv - d * gn[k];
        }
        float e = norm3(n) - 1; en = (double)(e * e);
        float l = laplacian(b, j, a.grid.vs_inv); el = (double)(l * l);
    }
    block_part_store(en, PART(a, SC_EN), red);
    block_part_store(el, PART(a, SC_EL), red);
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
__global__ void __launch_bounds__(kBlock) k_reach(Band b, int row0, int row1, int* __restrict__ need) {
    int i = row0 + blockIdx.x * blockDim.x + threadIdx.x;
    int lo = 0, hi = 0;
    if (i < row1) for (int q = 1; q < kNQ; ++q) { int c = b.col[(size_t)q * b.Spad + i]; if (c < row0) lo = max(lo, row0 - c); if (c >= row1) hi = max(hi, c - row1 + 1); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = max(lo, __shfl_down(lo, o, 64)); hi = max(hi, __shfl_down(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { if (lo > 0) atomicMax(need, lo); if (hi > 0) atomicMax(need + 1, hi); }
}
void launch_reach(const Band& b, int row0, int row1, int* d_need, hipStream_t s) {
    if (row1 > row0) hipLaunchKernelGGL(k_reach, dim3((row1 - row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, b, row0, row1, d_need);
}
// fold per-workgroup partials into a few doubles.  `out` may be host-mapped pinned memory: the host then needs no
// D2H copy (each hipMemcpyAsync costs ~10 us of GPU idle around it), only the stream synchronisation it does anyway.
__global__ void __launch_bounds__(kBlock) k_sum_parts(const double* __restrict__ part, int PB, int nblk, SlotList slots, double* __restrict__ out) {
    __shared__ double red[kBlock / 64];
    for (int s = 0; s < slots.n; ++s) {
        double t = block_total(part + (size_t)slots.id[s] * PB, nblk, red);
        if (threadIdx.x == 0) out[s] = t;
        __syncthreads();
    }
}
void launch_sum_parts(const double* part, int PB, int nblk, const SlotList& slots, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(kBlock), 0, s, part, PB, nblk, slots, out);
}
// sum of columns (col, col+1) over the F frame-accumulator rows -> out[0..1]
__global__ void __launch_bounds__(kBlock) k_frame_cols(const double* __restrict__ frame, int F, int col, double* __restrict__ out) {
    __shared__ double red[kBlock / 64];
    double a = 0, b = 0;
    for (int f = threadIdx.x; f < F; f += blockDim.x) { a += frame[(size_t)f * kFrameRow + col]; b += frame[(size_t)f * kFrameRow + col + 1]; }
    a = wave_sum(a); b = wave_sum(b);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = a;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < kBlock / 64; ++i) t += red[i]; out[0] = t; }
    __syncthreads();
    if (lane == 0) red[w] = b;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < kBlock / 64; ++i) t += red[i]; out[1] = t; }
}
void launch_frame_cols(const double* frame, int F, int col, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_frame_cols, dim3(1), dim3(kBlock), 0, s, frame, F, col, out);
}
__global__ void k_zero_f64(double* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.0; }
void launch_zero_f64(double* p, int n, hipStream_t s) { if (n > 0) hipLaunchKernelGGL(k_zero_f64, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), dim3(256), 0, s, p, n); }

// ------------------------------------------------------------------------------------------
// voxel-major sweeps: one thread per band voxel, iterating the set bits of its visibility mask
// ------------------------------------------------------------------------------------------
#define FOR_EACH_VISIBLE_FRAME(b, j, F, f)                                                   \
    for (int _w = 0; _w < (b).KW; ++_w)                                                      \
        for (uint64_t _m = (b).vis[(size_t)_w * (b).Spad + (j)]; _m; _m &= _m - 1)            \
            if (int f = 64 * _w + __builtin_ctzll(_m); f < (F))

// Optimizer.cpp:50-81 initAlbedo
__global__ void __launch_bounds__(kBlock) k_init_albedo(SweepArgs a) {
#pragma clang fp contract(off)
    FrameP* sf = reinterpret_cast<FrameP*>(psg_dyn_smem);   // F records, dynamic LDS
    load_frames(sf, a.frames, a.F);
    const Band& b = a.b;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.row1) return;
    float xs[3] = {b.xs[0][j], b.xs[1][j], b.xs[2][j]};
    int count = 0; float rho[3] = {0, 0, 0};
    FOR_EACH_VISIBLE_FRAME(b, j, a.F, f) {
        Proj pr = project(xs, sf[f], a.cam);
        if (!pr.ok) continue;
        float I[3];
        sample<false>(a.img, f, a.img32, a.cam, pr.m, pr.n, I, nullptr, nullptr);
        rho[0] += I[0]; rho[1] += I[1]; rho[2] += I[2]; count++;
    }
    if (count) {
#pragma unroll
        for (int k = 0; k < 3; ++k) b.rho[k][j] = rho[k] / (float)count;
    }
}
void launch_init_albedo(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_init_albedo, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), a.F * sizeof(FrameP), s, a);
}

// getPSEnergy PsOptimizer.cpp:47-78 / LedOptimizer.cpp:40-71; LED_INIT: computeLightIntensive
// LedOptimizer.cpp:76-112 (sums of observed and rendered intensity)
template <int MODEL, bool LED_INIT>
__global__ void __launch_bounds__(kBlock) k_energy(SweepArgs a) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    constexpr int NB = ModelTraits<MODEL>::NB;
    FrameP* sf = reinterpret_cast<FrameP*>(psg_dyn_smem);   // F records, dynamic LDS
    __shared__ double red[kBlock / 64];
    load_frames(sf, a.frames, a.F);
    const Band& b = a.b;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double E = 0, nobs = 0, sI[3] = {0, 0, 0}, sR[3] = {0, 0, 0};
    float Ef = 0.f;
    if (j < a.row1) {
        Vox v; load_vox(b, j, v);
        float shfd[kMaxBasis];
        if (!ModelTraits<MODEL>::LED) SH<NB == 3 ? 4 : NB>(v.nfd, shfd);
        FOR_EACH_VISIBLE_FRAME(b, j, a.F, f) {
            const FrameP& fp = sf[f];
            Proj pr = project(v.xs, fp, a.cam);
            if (!pr.ok) continue;
            float I[3], ren[3];
            sample<false>(a.img, f, a.img32, a.cam, pr.m, pr.n, I, nullptr, nullptr);
            rendered<MODEL>(fp, pr, v.nfd, shfd, v.rho, ren);
            if (LED_INIT) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) { sI[ch] += (double)I[ch]; sR[ch] += (double)ren[ch]; }
            } else {
                float l = 0.f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) l += robust_loss(a.rob, I[ch] - ren[ch]);
                Ef += l; nobs += 1.0;
            }
        }
    }
    E = (double)Ef;
    if (LED_INIT) {
        // 6 sums: observed rgb into SC_AUX0.., rendered into SC_EN/SC_EL/SC_ACCEPT slots (scratch use at init only)
        block_part_store(sI[0], PART(a, SC_AUX0), red); block_part_store(sI[1], PART(a, SC_AUX1), red); block_part_store(sI[2], PART(a, SC_AUX2), red);
        block_part_store(sR[0], PART(a, SC_EN), red); block_part_store(sR[1], PART(a, SC_EL), red); block_part_store(sR[2], PART(a, SC_ACCEPT), red);
    } else {
        block_part_store(E, PART(a, SC_ENERGY), red);
        block_part_store(nobs, PART(a, SC_NOBS), red);
    }
}
void launch_energy(const SweepArgs& a, hipStream_t s) {
    if (a.row1 <= a.row0) return;
    dim3 g((a.row1 - a.row0 + kBlock - 1) / kBlock), bl(kBlock);
    if (a.model == 0) hipLaunchKernelGGL((k_energy<0, false>), g, bl, a.F * sizeof(FrameP), s, a);
    else if (a.model == 1) hipLaunchKernelGGL((k_energy<1, false>), g, bl, a.F * sizeof(FrameP), s, a);
    else hipLaunchKernelGGL((k_energy<2, false>), g, bl, a.F * sizeof(FrameP), s, a);
}
void launch_led_light_init(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL((k_energy<2, true>), dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), a.F * sizeof(FrameP), s, a);
}

// albedo normal equations (diagonal): optimizeAlbedoAll PsOptimizer.cpp:85-121 / LedOptimizer.cpp:162-196,
// albedoJacobian PsOptimizerJa.cpp:375-422, computeResidual :567-626.  Also yields the PS energy of the input state.
template <int MODEL>
__global__ void __launch_bounds__(kBlock) k_sweep_albedo(SweepArgs a) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    constexpr int NB = ModelTraits<MODEL>::NB;
    FrameP* sf = reinterpret_cast<FrameP*>(psg_dyn_smem);   // F records, dynamic LDS
    __shared__ double red[kBlock / 64];
    load_frames(sf, a.frames, a.F);
    const Band& b = a.b;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double E = 0, nobs = 0;
    if (j < a.row1) {
        Vox v; load_vox(b, j, v);
        float shfd[kMaxBasis], shg[kMaxBasis];
        if (!ModelTraits<MODEL>::LED) { SH<NB == 3 ? 4 : NB>(v.nfd, shfd); SH<NB == 3 ? 4 : NB>(v.gn, shg); }
        float Hd[3] = {0, 0, 0}, bd[3] = {0, 0, 0};
        float Ef = 0.f; int nobs_i = 0;
        FOR_EACH_VISIBLE_FRAME(b, j, a.F, f) {
            const FrameP& fp = sf[f];
            Proj pr = project(v.xs, fp, a.cam);
            if (!pr.ok) continue;
            float I[3], ren[3], J[3];
            sample<false>(a.img, f, a.img32, a.cam, pr.m, pr.n, I, nullptr, nullptr);
            rendered<MODEL>(fp, pr, v.nfd, shfd, v.rho, ren);
            rho_jac<MODEL>(fp, pr, v.gn, shg, J);
            float l = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float r = I[ch] - ren[ch]; float w = robust_weight(a.rob, r);
                float jw = J[ch] * w;
                Hd[ch] += jw * J[ch]; bd[ch] += jw * r;
                l += robust_loss(a.rob, r);
            }
            Ef += l; nobs_i += 1;
        }
        E = (double)Ef; nobs = (double)nobs_i;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { b.aH[(size_t)ch * b.Spad + j] = Hd[ch]; b.ab[(size_t)ch * b.Spad + j] = bd[ch]; }
    }
    block_part_store(E, PART(a, SC_ENERGY), red);
    block_part_store(nobs, PART(a, SC_NOBS), red);
}
void launch_sweep_albedo(const SweepArgs& a, hipStream_t s) {
    if (a.row1 <= a.row0) return;
    dim3 g((a.row1 - a.row0 + kBlock - 1) / kBlock), bl(kBlock);
    if (a.model == 0) hipLaunchKernelGGL((k_sweep_albedo<0>), g, bl, a.F * sizeof(FrameP), s, a);
    else if (a.model == 1) hipLaunchKernelGGL((k_sweep_albedo<1>), g, bl, a.F * sizeof(FrameP), s, a);
    else hipLaunchKernelGGL((k_sweep_albedo<2>), g, bl, a.F * sizeof(FrameP), s, a);
}
// delta = b / ((1+damping) H), updateAlbedo accept rule OptimizerAux.cpp:120-150
__global__ void __launch_bounds__(kBlock) k_apply_albedo(SweepArgs a) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    __shared__ double red[kBlock / 64];
    const Band& b = a.b;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    double cnt = 0;
    if (j < a.row1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float h = b.aH[(size_t)ch * b.Spad + j];
            if (a.damping != 0.0f) h += a.damping * h;
            float delta = (h != 0.f) ? b.ab[(size_t)ch * b.Spad + j] / h : 0.f;
            float v = b.rho[ch][j] - delta;
            if (v > 0.0f && v < 1.0f) { b.rho[ch][j] = v; cnt += 1.0; }
        }
    }
    block_part_store(cnt, PART(a, SC_ACCEPT), red);
}
void launch_apply_albedo(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_apply_albedo, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}

// ------------------------------------------------------------------------------------------
// "reg albedo": E_r = sum_v sum_c ||grad rho_c(v)||, Gauss-Newton term reg_rho Jr^T Jr in the albedo system
// (Optimizer.cpp:122-136 energy, :396-460 computeAlbedoGrad, :221-245 per-voxel Jacobian, :593-647 sparse Jr).
// No shipped configuration enables it, so this path is written for clarity, not speed: the 3S x 3S system is never
// assembled; Jr (<= 4 entries per row: the voxel and its three stencil neighbours) is applied matrix-free.
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
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0;
    if (j < b.S) {
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
    if (a.b.S > 0) hipLaunchKernelGGL(k_areg_build, dim3((a.b.S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
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
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= b.S) return;
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
    if (a.b.S > 0) hipLaunchKernelGGL(k_areg_system, dim3((a.b.S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}
__global__ void __launch_bounds__(kBlock) k_areg_jx(SweepArgs a, const float* __restrict__ p, float* __restrict__ t) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.b.S) return;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) t[(size_t)ch * a.b.Spad + j] = areg_jrow(a, p, j, ch);
}
void launch_areg_jx(const SweepArgs& a, const float* p, float* t, hipStream_t s) {
    if (a.b.S > 0) hipLaunchKernelGGL(k_areg_jx, dim3((a.b.S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, p, t);
}
// q = H p with H = H_d + weight Jr^T Jr and H.diagonal() += damping * H.diagonal(); partial p.q
__global__ void __launch_bounds__(kBlock) k_areg_jt(SweepArgs a, const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ q) {
    __shared__ double red[kBlock / 64];
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    double pq = 0;
    if (j < b.S) {
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
    if (a.b.S > 0) hipLaunchKernelGGL(k_areg_jt, dim3((a.b.S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, p, t, q);
}
__global__ void __launch_bounds__(kBlock) k_areg_cg_init(SweepArgs a) {
    __shared__ double red[kBlock / 64];
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    double bb = 0, rz = 0;
    if (j < b.S) {
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
    if (a.b.S > 0) hipLaunchKernelGGL(k_areg_cg_init, dim3((a.b.S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}
__global__ void __launch_bounds__(kBlock) k_areg_cg_update(SweepArgs a, float alpha) {
    __shared__ double red[kBlock / 64];
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    double rr = 0, rz = 0;
    if (j < b.S) {
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
    if (a.b.S > 0) hipLaunchKernelGGL(k_areg_cg_update, dim3((a.b.S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, alpha);
}
__global__ void __launch_bounds__(kBlock) k_areg_cg_dir(SweepArgs a, float beta) {
    const Band& b = a.b; const AlbedoReg& ar = a.ar;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= b.S) return;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const size_t u = (size_t)ch * b.Spad + j;
        const float dg = ar.diag[u];
        ar.p[u] = (dg != 0.f ? 1.0f / dg : 1.0f) * ar.r[u] + beta * ar.p[u];
    }
}
void launch_areg_cg_dir(const SweepArgs& a, float beta, hipStream_t s) {
    if (a.b.S > 0) hipLaunchKernelGGL(k_areg_cg_dir, dim3((a.b.S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, beta);
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
            if (v > 0.0f && v < 1.0f) { b.rho[ch][j] = v; cnt += 1.0; }
        }
    }
    block_part_store(cnt, PART(a, SC_ACCEPT), red);
}
void launch_apply_albedo_delta(const SweepArgs& a, const float* delta, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_apply_albedo_delta, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, delta);
}


// ------------------------------------------------------------------------------------------
// frame-major sweeps: grid = (row chunks, F); per-thread accumulation over several voxels of ONE
// frame, then wavefront shuffle reduction -> LDS -> one double atomic per value per workgroup
// ------------------------------------------------------------------------------------------
constexpr int kRowsPerThread = 16;
constexpr int kChunk = kBlock * kRowsPerThread;

// light normal equations: lightJacobian PsOptimizerJa.cpp:132-143,323-371 (per frame NBxNB),
// LED LightJacobian LedOptimizerJa.cpp:101-115,299-346 (one global diagonal 3x3)
template <int MODEL>
__global__ void __launch_bounds__(kBlock) k_sweep_light(SweepArgs a) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    constexpr int NB = ModelTraits<MODEL>::NB;
    constexpr bool LED = ModelTraits<MODEL>::LED;
    constexpr int NH = LED ? 3 : NB * (NB + 1) / 2;
    constexpr int NV = NH + NB + 2;   // + energy, n_obs
    __shared__ FrameP sfp;
    __shared__ double lds[(kBlock / 64) * NV];
    const int f = blockIdx.y;
    if (threadIdx.x < (int)(sizeof(FrameP) / 4)) ((float*)&sfp)[threadIdx.x] = ((const float*)(a.frames + f))[threadIdx.x];
    __syncthreads();
    const Band& b = a.b;
    const FrameP& fp = sfp;
    const float* img = a.img + (size_t)f * a.cam.H * a.cam.W * 3;
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
    const int beg = b.obs_ptr[f], end = b.obs_ptr[f + 1];
    // Software pipeline over the thread's observations: the row index is fetched two observations ahead and the voxel state one
    // ahead, so that an iteration waits for its image taps only (un-pipelined, index -> state -> taps were three dependent round
    // trips per observation and the sweep was latency-bound at 4-6 waves per SIMD).
    int e = beg + blockIdx.x * kChunk + threadIdx.x;
    int j_cur = e < end ? b.obs_rows[e] : -1;
    int j_nxt = (kRowsPerThread > 1 && e + kBlock < end) ? b.obs_rows[e + kBlock] : -1;
    Vox vn;
    if (j_cur >= 0) load_vox(b, j_cur, vn);
    for (int it = 0; it < kRowsPerThread && j_cur >= 0; ++it, e += kBlock) {
        const Vox v = vn;
        const int j_nn = (it + 2 < kRowsPerThread && e + 2 * kBlock < end) ? b.obs_rows[e + 2 * kBlock] : -1;
        if (j_nxt >= 0) load_vox(b, j_nxt, vn);
        j_cur = j_nxt; j_nxt = j_nn;
        Proj pr = project(v.xs, fp, a.cam);
        if (!pr.ok) continue;
        float shfd[kMaxBasis], shg[kMaxBasis];
        if (!LED) { SH<NB == 3 ? 4 : NB>(v.nfd, shfd); SH<NB == 3 ? 4 : NB>(v.gn, shg); }
        float I[3], ren[3];
        sample<false>(img, 0, true, a.cam, pr.m, pr.n, I, nullptr, nullptr);
        rendered<MODEL>(fp, pr, v.nfd, shfd, v.rho, ren);
        float refl = 0.f;
        if (LED) { float Rp[3]; mul3(fp.R, pr.p, Rp); refl = dot3(v.gn, Rp); float pn = norm3(pr.p); double pd = (double)pn; refl /= (float)(pd * pd * pd); }
        float l = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float r = I[ch] - ren[ch]; float w = robust_weight(a.rob, r);
            l += robust_loss(a.rob, r);
            if (LED) {
                float J = refl * v.rho[ch]; float jw = J * w;
                acc[ch] += jw * J; acc[NH + ch] += jw * r;
            } else {
                float J[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) J[i] = -v.rho[ch] * shg[i];
                int q = 0;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    float jw = J[i] * w;
#pragma unroll
                    for (int k = i; k < NB; ++k) acc[q++] += jw * J[k];
                    acc[NH + i] += jw * r;
                }
            }
        }
        acc[NH + NB] += l; acc[NH + NB + 1] += 1.0f;
    }
    // one atomic per value per workgroup into THIS frame's row (<= S/2048 workgroups contend per address);
    // row layout: [NH H entries | NB rhs | energy | n_obs]
    double* dst = a.acc.frame + (size_t)f * kFrameRow;
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) { double vv = wave_sum((double)acc[k]); if (lane == 0) lds[w * NV + k] = vv; }   // double: SH2 light blocks are ill-conditioned
    __syncthreads();
    for (int k = threadIdx.x; k < NV; k += blockDim.x) {
        double s = 0;
        for (int i = 0; i < kBlock / 64; ++i) s += lds[i * NV + k];
        if (s != 0.0) atomicAdd(dst + k, s);
    }
}
void launch_sweep_light(const SweepArgs& a, hipStream_t s) {
    if (a.b.S <= 0 || a.F <= 0 || a.b.obs_max <= 0) return;
    dim3 g((a.b.obs_max + kChunk - 1) / kChunk, a.F), bl(kBlock);
    if (a.model == 0) hipLaunchKernelGGL((k_sweep_light<0>), g, bl, 0, s, a);
    else if (a.model == 1) hipLaunchKernelGGL((k_sweep_light<1>), g, bl, 0, s, a);
    else hipLaunchKernelGGL((k_sweep_light<2>), g, bl, 0, s, a);
}

// G = image_grad(3x2) * pi_grad(2x3), PsOptimizerJa.cpp:78-90
__device__ __forceinline__ void image_pi_grad(const Cam& cam, const Proj& pr, const float* gu, const float* gv, float* G) {
    const float z_inv = pr.z_inv;
    float z_inv_sq = z_inv * z_inv;
    float p00 = cam.fx * z_inv, p02 = -cam.fx * pr.p[0] * z_inv_sq, p11 = cam.fy * z_inv, p12 = -cam.fy * pr.p[1] * z_inv_sq;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        G[ch * 3 + 0] = gu[ch] * p00 + gv[ch] * 0.0f;
        G[ch * 3 + 1] = gu[ch] * 0.0f + gv[ch] * p11;
        G[ch * 3 + 2] = gu[ch] * p02 + gv[ch] * p12;
    }
}

// pose normal equations: poseJacobian PsOptimizerJa.cpp:61-115,427-475 / LedOptimizerJa.cpp:32-81,351-399
template <int MODEL>
__global__ void __launch_bounds__(kBlock) k_sweep_pose(SweepArgs a) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    constexpr int NB = ModelTraits<MODEL>::NB;
    constexpr bool LED = ModelTraits<MODEL>::LED;
    constexpr int NV = 21 + 6 + 2;
    __shared__ FrameP sfp;
    __shared__ double lds[(kBlock / 64) * NV];
    const int f = blockIdx.y;
    if (threadIdx.x < (int)(sizeof(FrameP) / 4)) ((float*)&sfp)[threadIdx.x] = ((const float*)(a.frames + f))[threadIdx.x];
    __syncthreads();
    const Band& b = a.b;
    const FrameP& fp = sfp;
    const float* img = a.img + (size_t)f * a.cam.H * a.cam.W * 3;
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
    const int beg = b.obs_ptr[f], end = b.obs_ptr[f + 1];
    // Software pipeline over the thread's observations: the row index is fetched two observations ahead and the voxel state one
    // ahead, so that an iteration waits for its image taps only (un-pipelined, index -> state -> taps were three dependent round
    // trips per observation and the sweep was latency-bound at 4-6 waves per SIMD).
    int e = beg + blockIdx.x * kChunk + threadIdx.x;
    int j_cur = e < end ? b.obs_rows[e] : -1;
    int j_nxt = (kRowsPerThread > 1 && e + kBlock < end) ? b.obs_rows[e + kBlock] : -1;
    Vox vn;
    if (j_cur >= 0) load_vox(b, j_cur, vn);
    for (int it = 0; it < kRowsPerThread && j_cur >= 0; ++it, e += kBlock) {
        const Vox v = vn;
        const int j_nn = (it + 2 < kRowsPerThread && e + 2 * kBlock < end) ? b.obs_rows[e + 2 * kBlock] : -1;
        if (j_nxt >= 0) load_vox(b, j_nxt, vn);
        j_cur = j_nxt; j_nxt = j_nn;
        Proj pr = project(v.xs, fp, a.cam);
        if (!pr.ok) continue;
        float shfd[kMaxBasis];
        if (!LED) SH<NB == 3 ? 4 : NB>(v.nfd, shfd);
        float I[3], gu[3], gv[3], ren[3];
        sample<true>(img, 0, true, a.cam, pr.m, pr.n, I, gu, gv);
        rendered<MODEL>(fp, pr, v.nfd, shfd, v.rho, ren);
        float G[9]; image_pi_grad(a.cam, pr, gu, gv, G);
        float J[18];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float s = (G[ch * 3 + 0] * fp.R[k * 3 + 0] + G[ch * 3 + 1] * fp.R[k * 3 + 1]) + G[ch * 3 + 2] * fp.R[k * 3 + 2];
                J[ch * 6 + k] = -s;
            }
        const float* p = pr.p;
        float sk[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int k = 0; k < 3; ++k) J[ch * 6 + 3 + k] = (G[ch * 3 + 0] * sk[0 * 3 + k] + G[ch * 3 + 1] * sk[1 * 3 + k]) + G[ch * 3 + 2] * sk[2 * 3 + k];
        if (LED) {
            float pn = norm3(p); double pd = (double)pn; float l3 = (float)(pd * pd * pd);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float s = -(v.rho[ch] * fp.l[ch] / l3);
#pragma unroll
                for (int k = 0; k < 3; ++k) J[ch * 6 + k] += s * v.gn[k];
            }
        }
        float l = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float r = I[ch] - ren[ch]; float w = robust_weight(a.rob, r);
            l += robust_loss(a.rob, r);
            int q = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float jw = J[ch * 6 + i] * w;
#pragma unroll
                for (int k = i; k < 6; ++k) acc[q++] += jw * J[ch * 6 + k];
                acc[21 + i] += jw * r;
            }
        }
        acc[27] += l; acc[28] += 1.0f;
    }
    double* dst = a.acc.frame + (size_t)f * kFrameRow;   // [21 H | 6 rhs | energy | n_obs]
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) { double vv = wave_sum((double)acc[k]); if (lane == 0) lds[w * NV + k] = vv; }   // double: SH2 light blocks are ill-conditioned
    __syncthreads();
    for (int k = threadIdx.x; k < NV; k += blockDim.x) {
        double s = 0;
        for (int i = 0; i < kBlock / 64; ++i) s += lds[i * NV + k];
        if (s != 0.0) atomicAdd(dst + k, s);
    }
}
void launch_sweep_pose(const SweepArgs& a, hipStream_t s) {
    if (a.b.S <= 0 || a.F <= 0 || a.b.obs_max <= 0) return;
    dim3 g((a.b.obs_max + kChunk - 1) / kChunk, a.F), bl(kBlock);
    if (a.model == 0) hipLaunchKernelGGL((k_sweep_pose<0>), g, bl, 0, s, a);
    else if (a.model == 1) hipLaunchKernelGGL((k_sweep_pose<1>), g, bl, 0, s, a);
    else hipLaunchKernelGGL((k_sweep_pose<2>), g, bl, 0, s, a);
}

// ------------------------------------------------------------------------------------------
// small dense solves (one thread per frame): LDL^T in double, zero step on non-positive pivots
// ------------------------------------------------------------------------------------------
template <int N>
__device__ void solve_spd(const double* Hin, const double* bin, double* x) {
    double L[N * N], D[N], y[N];
    double scale = 0;
    for (int i = 0; i < N; ++i) scale = fmax(scale, fabs(Hin[i * N + i]));
    const double tiny = scale * 1e-12;
    for (int i = 0; i < N * N; ++i) L[i] = 0;
    for (int j = 0; j < N; ++j) {
        double d = Hin[j * N + j];
        for (int k = 0; k < j; ++k) d -= L[j * N + k] * L[j * N + k] * D[k];
        D[j] = d; L[j * N + j] = 1.0;
        for (int i = j + 1; i < N; ++i) {
            double s = Hin[i * N + j];
            for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k] * D[k];
            L[i * N + j] = (d > tiny) ? s / d : 0.0;
        }
    }
    for (int i = 0; i < N; ++i) { double s = bin[i]; for (int k = 0; k < i; ++k) s -= L[i * N + k] * y[k]; y[i] = s; }
    for (int i = 0; i < N; ++i) y[i] = (D[i] > tiny) ? y[i] / D[i] : 0.0;
    for (int i = N - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k]; x[i] = s; }
}

// The per-frame solves run as ONE workgroup (a thread takes frames tid, tid+256, ...), which lets the same kernel do
// what used to be two more launches around every frame-major sweep: sum the energy / n_obs columns of the rows
// (e_out, may be host-mapped) and clear the rows for the next sweep.  Invariant: the frame accumulator is all-zero
// outside [sweep, solve].
__device__ __forceinline__ void frame_rows_finish(const SweepArgs& a, int col_e, double* e_out, double* red) {
    __syncthreads();                                       // every thread has read the rows it solves from
    if (e_out) {
        double e = 0, n = 0;
        for (int f = threadIdx.x; f < a.F; f += blockDim.x) { e += a.acc.frame[(size_t)f * kFrameRow + col_e]; n += a.acc.frame[(size_t)f * kFrameRow + col_e + 1]; }
        e = wave_sum(e); n = wave_sum(n);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) { red[2 * w] = e; red[2 * w + 1] = n; }
        __syncthreads();
        if (threadIdx.x == 0) { double te = 0, tn = 0; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { te += red[2 * i]; tn += red[2 * i + 1]; } e_out[0] = te; e_out[1] = tn; }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < a.F * kFrameRow; i += blockDim.x) a.acc.frame[i] = 0.0;
}

// optimizeLightAll: PsOptimizer.cpp:175-203 (no damping) / LedOptimizer.cpp:134-160 (damped, one RGB vector)
template <int MODEL>
__global__ void __launch_bounds__(kBlock) k_solve_light(SweepArgs a, FrameP* frames, float* led_light, double* e_out) {
    constexpr int NB = ModelTraits<MODEL>::NB;
    constexpr bool LED = ModelTraits<MODEL>::LED;
    constexpr int NH = LED ? 3 : NB * (NB + 1) / 2;
    __shared__ double red[2 * kBlock / 64];
    if (LED) {
        // every thread sums the per-frame rows, solves the same 3 scalar equations and updates its own records
        float dl[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            double hs = 0, bs = 0;
            for (int ff = 0; ff < a.F; ++ff) { hs += a.acc.frame[(size_t)ff * kFrameRow + ch]; bs += a.acc.frame[(size_t)ff * kFrameRow + NH + ch]; }
            float h = (float)hs, bb = (float)bs;
            if (a.damping != 0.0f) h += a.damping * h;
            double Hd[1] = {(double)h}, bd[1] = {(double)bb}, xd[1];
            solve_spd<1>(Hd, bd, xd);
            dl[ch] = (float)xd[0];
        }
        for (int f = threadIdx.x; f < a.F; f += blockDim.x) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { const float nl = frames[f].l[ch] - dl[ch]; frames[f].l[ch] = nl; if (f == 0) led_light[ch] = nl; }
        }
    } else {
        for (int f = threadIdx.x; f < a.F; f += blockDim.x) {
            const double* acc = a.acc.frame + (size_t)f * kFrameRow;
            double Hd[NB * NB], bd[NB], xd[NB];
            int q = 0;
            for (int i = 0; i < NB; ++i) for (int k = i; k < NB; ++k) { double v = (double)(float)acc[q++]; Hd[i * NB + k] = v; Hd[k * NB + i] = v; }
            for (int i = 0; i < NB; ++i) bd[i] = (double)(float)acc[NH + i];
            solve_spd<NB>(Hd, bd, xd);
            for (int i = 0; i < NB; ++i) frames[f].l[i] -= (float)xd[i];
        }
    }
    frame_rows_finish(a, NH + NB, e_out, red);
}
void launch_solve_light(const SweepArgs& a, FrameP* frames, float* led_light, double* e_out, hipStream_t s) {
    if (a.F <= 0) return;
    dim3 g(1), bl(kBlock);
    if (a.model == 0) hipLaunchKernelGGL((k_solve_light<0>), g, bl, 0, s, a, frames, led_light, e_out);
    else if (a.model == 1) hipLaunchKernelGGL((k_solve_light<1>), g, bl, 0, s, a, frames, led_light, e_out);
    else hipLaunchKernelGGL((k_solve_light<2>), g, bl, 0, s, a, frames, led_light, e_out);
}

// Sophus SO3::exp(w).matrix() (quaternion exponential + Eigen toRotationMatrix)
__device__ void so3_exp(const float* w, float* R) {
    float theta_sq = dot3(w, w);
    float imag, real;
    if (theta_sq < 1e-10f) {
        float theta_po4 = theta_sq * theta_sq;
        imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_po4;
        real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_po4;
    } else {
        float theta = sqrtf(theta_sq), half = 0.5f * theta;
        imag = sinf(half) / theta; real = cosf(half);
    }
    float qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
    float tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// optimizePosesAll PsOptimizer.cpp:207-234 + updatePose OptimizerAux.cpp:190-205
__global__ void __launch_bounds__(kBlock) k_solve_pose(SweepArgs a, FrameP* frames, double* e_out) {
    __shared__ double red[2 * kBlock / 64];
    for (int f = threadIdx.x; f < a.F; f += blockDim.x) {
        const double* acc = a.acc.frame + (size_t)f * kFrameRow;
        double Hd[36], bd[6], xd[6];
        int q = 0;
        for (int i = 0; i < 6; ++i) for (int k = i; k < 6; ++k) {
            float v = (float)acc[q++];
            if (i == k && a.damping != 0.0f) v += a.damping * v;
            Hd[i * 6 + k] = (double)v; Hd[k * 6 + i] = (double)v;
        }
        for (int i = 0; i < 6; ++i) bd[i] = (double)(float)acc[21 + i];
        solve_spd<6>(Hd, bd, xd);
        float xi[6];
        for (int i = 0; i < 6; ++i) xi[i] = (float)xd[i];
        float R[9], t[3];
        for (int i = 0; i < 9; ++i) R[i] = frames[f].R[i];
        for (int i = 0; i < 3; ++i) t[i] = frames[f].t[i];
        float mw[3] = {-xi[3], -xi[4], -xi[5]}, E3[9];
        so3_exp(mw, E3);
        for (int i = 0; i < 3; ++i) {
            frames[f].t[i] = t[i] - xi[i];
            for (int k = 0; k < 3; ++k) frames[f].R[i * 3 + k] = (R[i * 3 + 0] * E3[0 * 3 + k] + R[i * 3 + 1] * E3[1 * 3 + k]) + R[i * 3 + 2] * E3[2 * 3 + k];
        }
    }
    frame_rows_finish(a, 27, e_out, red);
}
void launch_solve_pose(const SweepArgs& a, FrameP* frames, double* e_out, hipStream_t s) {
    if (a.F > 0) hipLaunchKernelGGL(k_solve_pose, dim3(1), dim3(kBlock), 0, s, a, frames, e_out);
}

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
__device__ __forceinline__ int sym4(int a, int b) {   // index into the 10 upper-triangular entries
    if (a > b) { int t = a; a = b; b = t; }
    return a * 4 - (a * (a - 1)) / 2 + (b - a);
}

// distJacobian per observation PsOptimizerJa.cpp:160-289 / LedOptimizerJa.cpp:117-218, accumulated directly
// into the per-voxel block over {self, x-, y-, z-stencil neighbour}; regularisers Optimizer.cpp:196-218,477-590.
template <int MODEL>
__global__ void __launch_bounds__(kBlock) k_sweep_dist(SweepArgs a) {
    constexpr int NB = ModelTraits<MODEL>::NB;
    constexpr bool LED = ModelTraits<MODEL>::LED;
    FrameP* sf = reinterpret_cast<FrameP*>(psg_dyn_smem);   // F records, dynamic LDS
    __shared__ double red[kBlock / 64];
    load_frames(sf, a.frames, a.F);
    const Band& b = a.b;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
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
        float Dm[3][9];
        if (NB == 9) {
            const float* nh = v.nfd;
            float D0[9] = {0, 1, 0, 0, nh[1], nh[2], 0, 2 * nh[0], 2 * nh[0]};
            float D1[9] = {0, 0, 1, 0, nh[0], 0, nh[2], -2 * nh[1], 0};
            float D2[9] = {0, 0, 0, 1, 0, nh[0], nh[1], 0, -2 * nh[2]};
#pragma unroll
            for (int i = 0; i < 9; ++i) { Dm[0][i] = D0[i]; Dm[1][i] = D1[i]; Dm[2][i] = D2[i]; }
        }
        float B[10], g[4];   // <= F terms each: float accumulation (oracle: double) differs ~1e-7 relative
        float Ef = 0.f; int nobs_i = 0;
#pragma unroll
        for (int i = 0; i < 10; ++i) B[i] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = 0;
        FOR_EACH_VISIBLE_FRAME(b, j, a.F, f) {
            const FrameP& fp = sf[f];
            Proj pr = project(v.xs, fp, a.cam);
            if (!pr.ok) continue;
            float I[3], gu[3], gv[3], ren[3];
            sample<true>(a.img, f, a.img32, a.cam, pr.m, pr.n, I, gu, gv);
            rendered<MODEL>(fp, pr, v.nfd, shfd, v.rho, ren);
            float G[9]; image_pi_grad(a.cam, pr, gu, gv, G);
            float GRt[9];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                for (int k = 0; k < 3; ++k) GRt[ch * 3 + k] = (G[ch * 3 + 0] * fp.R[k * 3 + 0] + G[ch * 3 + 1] * fp.R[k * 3 + 1]) + G[ch * 3 + 2] * fp.R[k * 3 + 2];
            float J[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) mul3(GRt, dx[q], J[q]);   // dI_q
            if (!LED) {
                if (NB == 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            float dr[3] = {v.rho[ch] * fp.l[1], v.rho[ch] * fp.l[2], v.rho[ch] * fp.l[3]};
                            J[q][ch] = J[q][ch] - dot3(dr, dn[q]);
                        }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float dsh[9];
#pragma unroll
                        for (int i = 0; i < 9; ++i) dsh[i] = (Dm[0][i] * dn[q][0] + Dm[1][i] * dn[q][1]) + Dm[2][i] * dn[q][2];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            float s = 0.f;
#pragma unroll
                            for (int i = 0; i < 9; ++i) s += (v.rho[ch] * fp.l[i]) * dsh[i];
                            J[q][ch] = J[q][ch] - s;
                        }
                    }
                }
            } else {
                float Rp[3]; mul3(fp.R, pr.p, Rp);
                float pn = norm3(pr.p); double pd = (double)pn;
                float radius = (float)(pd * pd * pd);
                float p5 = (float)(pd * pd * pd * pd * pd);
                float nRp = dot3(v.nfd, Rp);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float dm = dot3(dn[q], Rp) + dot3(v.nfd, dx[q]);
                    float tmp[3]; mulT3(fp.R, dx[q], tmp);
                    float dm2 = -3 * dot3(pr.p, tmp) / p5;
                    dm = dm / radius + dm2 * nRp;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) J[q][ch] = J[q][ch] + (v.rho[ch] * fp.l[ch]) * dm;
                }
            }
            float l = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float r = I[ch] - ren[ch]; float w = robust_weight(a.rob, r);
                l += robust_loss(a.rob, r);
                int q = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float jw = J[p][ch] * w;
#pragma unroll
                    for (int k = p; k < 4; ++k) B[q++] += jw * J[k][ch];
                    g[p] += jw * r;
                }
            }
            Ef += l; nobs_i += 1;
        }
        E = (double)Ef; nobs = (double)nobs_i;
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
                for (int k = p; k < 4; ++k) B[q++] += a.reg_n * (Jr[p] * Jr[k]);
                g[p] += a.reg_n * (Jr[p] * res);
            }
        }
        if (a.laplacian_reg) {   // diagonal only (reference drops the off-diagonals), Optimizer.cpp:540-590
            float vs2 = vs_inv * vs_inv; float Jl = -6 * vs2; float res = laplacian(b, j, vs_inv);
            B[0] += a.reg_l * (Jl * Jl); g[0] += a.reg_l * (Jl * res);
        }
        // columns whose stencil neighbour is outside the band are dropped (PsOptimizerJa.cpp:536-552)
        int q = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int k = p; k < 4; ++k) { b.blk[(size_t)q * b.Spad + j] = (exists[p] && exists[k]) ? B[q] : 0.f; ++q; }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) b.blk[(size_t)(10 + p) * b.Spad + j] = exists[p] ? g[p] : 0.f;
    }
    block_part_store(E, PART(a, SC_ENERGY), red);
    block_part_store(nobs, PART(a, SC_NOBS), red);
}
void launch_sweep_dist(const SweepArgs& a, hipStream_t s) {
    if (a.row1 <= a.row0) return;
    dim3 g((a.row1 - a.row0 + kBlock - 1) / kBlock), bl(kBlock);
    if (a.model == 0) hipLaunchKernelGGL((k_sweep_dist<0>), g, bl, a.F * sizeof(FrameP), s, a);
    else if (a.model == 1) hipLaunchKernelGGL((k_sweep_dist<1>), g, bl, a.F * sizeof(FrameP), s, a);
    else hipLaunchKernelGGL((k_sweep_dist<2>), g, bl, a.F * sizeof(FrameP), s, a);
}

// H = sum_j P_j^T B_j P_j assembled row-wise into 19 fixed column offsets (ELL); a row receives
// slices from itself, from each lower neighbour (whose forward stencil points at it) and from each
// upper neighbour whose stencil was forced backward.  Accumulation in LDS (dynamic column index).
__global__ void __launch_bounds__(kBlock) k_assemble(SweepArgs a) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    __shared__ double acc[kNQ][kBlock];
    const Band& b = a.b;
    int i = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < kNQ; ++q) acc[q][tid] = 0.0;
    if (i >= a.row1) return;
    double rhs = 0.0;
    for (int c = 0; c < 7; ++c) {
        int jrow, s; int coff[3] = {0, 0, 0};
        if (c == 0) { jrow = i; s = 0; }
        else {
            int ax = (c - 1) >> 1; bool upper = (c - 1) & 1;
            jrow = b.nb[(size_t)(2 * ax + (upper ? 0 : 1)) * b.Spad + i];
            if (jrow < 0) continue;
            if (upper && b.nb[(size_t)(2 * ax) * b.Spad + jrow] >= 0) continue;   // its stencil is forward: does not touch row i
            s = ax + 1; coff[ax] = upper ? 1 : -1;
        }
        int dirj[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) dirj[k] = b.nb[(size_t)(2 * k) * b.Spad + jrow] >= 0 ? 1 : -1;
        rhs += (double)b.blk[(size_t)(10 + s) * b.Spad + jrow];
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            int o[3] = {coff[0], coff[1], coff[2]};
            if (bq > 0) o[bq - 1] += dirj[bq - 1];
            float val = b.blk[(size_t)sym4(s, bq) * b.Spad + jrow];
            if (val != 0.f) acc[q_of(o)][tid] += (double)val;
        }
    }
    int extra = 0;
#pragma unroll
    for (int q = 0; q < kNQ; ++q) { float h = (float)acc[q][tid]; b.H[(size_t)q * b.Spad + i] = h; if (q >= kNQCommon && h != 0.f) extra = 1; }
    b.hx[i] = extra;
    b.rhs[i] = (float)rhs;
}
void launch_assemble(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_assemble, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}

__device__ __forceinline__ float pcg_threshold(float rhsNorm2) { return fmaxf(FLT_EPSILON * FLT_EPSILON * rhsNorm2, FLT_MIN); }

// ------------------------------------------------------------------------------------------
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
//   fs (device doubles): [0] |b|^2   [1] done (0 = running)
//   part: [2 parity][kCgfSums][kPcgMaxBlocks] per-workgroup partial sums, summed in a fixed order by every workgroup of
//         the next kernel (deterministic, no atomics)
//   mb  : slot of THIS kernel (mapped host memory on one GPU): k = 0 -> |b|^2, k > 0 -> |r|^2 after pass k-1
// Multi-rank (a.ext != nullptr): a 1-workgroup kernel folds the partials of a pass into a.ext[0..6], the host program
// all-reduces them over the ranks, and the next kernel reads the global sums from a.ext instead of the partials; the
// records of the halo rows are exchanged before each pass.
// ------------------------------------------------------------------------------------------
constexpr int kCgfSums = 7;
__device__ __forceinline__ double* fpart(double* part, int k, int kind) { return part + ((size_t)((k & 1) * kCgfSums + kind)) * kPcgMaxBlocks; }

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
__device__ __forceinline__ void block_total_finish(const PartLoads<N>& pl, double* red /*[N * kBlock/64]*/, double* out) {
    double v[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        v[q] = pl.ld[0][q];
#pragma unroll
        for (int j = 1; j < kCgfMaxBlocks / kBlock; ++j) v[q] += pl.ld[j][q];
        v[q] = wave_sum(v[q]);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) red[q * (kBlock / 64) + w] = v[q];
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
__device__ __forceinline__ void block_part_store_n(const double* vin, double* const* dst, double* red) {
    double v[N];
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = wave_sum(vin[q]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) red[q * (kBlock / 64) + w] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < N) {
        double s = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[threadIdx.x * (kBlock / 64) + i];
        dst[threadIdx.x][blockIdx.x] = s;
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
    if (blockIdx.x == 0 && threadIdx.x == 0) fs[1] = 0.0;
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
template <int R, bool C16>
__device__ __forceinline__ void cgf_rows(const Band& b, const float4* __restrict__ rin, int i0, int stride, int row1, float damping, CgfRow* w, CgfPending<R>& pend, int ab) {
    const int plane = b.Spad * 4;              // bytes of one ELL column plane
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc((void*)b.H, 0, kNQ * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rC = C16 ? __builtin_amdgcn_make_buffer_rsrc((void*)b.colp, 0, (kNQ - 1) / 2 * plane, 0x00020000)
                                          : __builtin_amdgcn_make_buffer_rsrc((void*)b.col, 0, kNQ * plane, 0x00020000);
    float h[R][kNQ]; int c[R][kNQ];
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
    __shared__ double red[kCgfSums * kBlock / 64];
    const Band& b = a.b;
    long long* ts = (long long*)(fs + 16) + (size_t)blockIdx.x * 8;   // ab & 1024: stage timestamps of every workgroup
#define CGF_STAMP(j) do { if ((ab & 1024) && threadIdx.x == 0) { ts[j] = clock64(); if (j == 0) ts[6] = wall_clock64(); if (j == 4) ts[7] = wall_clock64(); } } while (0)
    CGF_STAMP(0);
    const float4* __restrict__ rin = b.rec[(k + 1) & 1];
    float4* __restrict__ rout = b.rec[k & 1];
    const int stride = gridDim.x * blockDim.x;
    int i0 = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    CgfRow w[kCgfRows]; CgfPending<kCgfRows> pend;
    const double stopped = fs[1];
    cgf_rows<kCgfRows, C16>(b, rin, i0, stride, a.row1, a.damping, w, pend, ab);

    CGF_STAMP(1);
    float alpha_prev = 0.f, beta = 0.f, rr_cur, rhsNorm2;
    if (ab & 1) { alpha_prev = 0.01f; beta = 0.5f; rr_cur = 1.f; rhsNorm2 = 1.f; cgf_rows_finish<kCgfRows>(w, pend); }
    else if (!(ab & 16) && stopped != 0.0 && stopped <= (double)k) return;   // stopped by an EARLIER kernel of this solve (kernel j writes j + 1)
    else if (k == 0) {
        double bb;
        if (a.ext) bb = a.ext[0];
        else { double* src[1] = {fpart(part, -1, 6)}; block_total_n<1>(src, gridDim.x, red, &bb); }
        cgf_rows_finish<kCgfRows>(w, pend);
        rhsNorm2 = (float)bb; rr_cur = rhsNorm2;
        if (blockIdx.x == 0 && threadIdx.x == 0) { fs[0] = bb; mb[0] = bb; __threadfence_system(); }   // mb may be host-mapped: the host watches it
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
        if (blockIdx.x == 0 && threadIdx.x == 0) { mb[0] = (double)rr_cur; __threadfence_system(); }
    }
    CGF_STAMP(2);
    const bool rhs_zero = rhsNorm2 == 0.f;
    const bool stop = !(ab & 16) && (rhs_zero || k == kmax || (k > 0 && rr_cur < pcg_threshold(rhsNorm2)));
    if (stop && blockIdx.x == 0 && threadIdx.x == 0) fs[1] = (double)(k + 1);
    double s[kCgfSums];
#pragma unroll
    for (int q = 0; q < kCgfSums; ++q) s[q] = 0;
    while (true) {
#pragma unroll
        for (int u = 0; u < kCgfRows; ++u) {
            const CgfRow& r = w[u];
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
        i0 += kCgfRows * stride;
        if (i0 - (int)threadIdx.x >= a.row1) break;          // workgroup-uniform
        cgf_rows<kCgfRows, C16>(b, rin, i0, stride, a.row1, a.damping, w, pend, ab);
        cgf_rows_finish<kCgfRows>(w, pend);
    }
    CGF_STAMP(3);
    if (stop || (ab & 4)) return;
    double* dst[kCgfSums];
#pragma unroll
    for (int q = 0; q < kCgfSums; ++q) dst[q] = fpart(part, k, q);
    block_part_store_n<kCgfSums>(s, dst, red);
    CGF_STAMP(4);
#undef CGF_STAMP
}
void launch_cgf_pass(const SweepArgs& a, double* fs, double* part, int G, int rows, int k, int kmax, double* mb, hipStream_t s, int ablate) {
    if (a.row1 <= a.row0) return;
    // rows per thread in flight at once <-> registers <-> resident workgroups per CU (launch bound = waves per SIMD).
    // One row per thread (114 VGPRs, 4 waves per SIMD) is the production shape; two rows spill at 3 waves per SIMD and are
    // kept for the timing tool only.
    if (rows >= 2) hipLaunchKernelGGL((k_cgf_pass<2, 2, false>), dim3(G), dim3(kBlock), 0, s, a, fs, part, k, kmax, mb, ablate);
    else if (a.b.col16) hipLaunchKernelGGL((k_cgf_pass<1, 4, true>), dim3(G), dim3(kBlock), 0, s, a, fs, part, k, kmax, mb, ablate);
    else hipLaunchKernelGGL((k_cgf_pass<1, 4, false>), dim3(G), dim3(kBlock), 0, s, a, fs, part, k, kmax, mb, ablate);
}

// multi-rank: fold the partials of pass k (k = -1: |b|^2 of the init) into out[0..6] for the host program's all-reduce
__global__ void __launch_bounds__(kBlock) k_cgf_sum(double* part, int G, int k, double* __restrict__ out) {
    __shared__ double red[kCgfSums * kBlock / 64];
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
