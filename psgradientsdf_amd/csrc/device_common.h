// device_common.h -- device-side helpers shared by the gfx950 kernel files (band.hip, frontend.hip, sweeps.hip,
// albedo_reg.hip, dist.hip, pcg.hip): small vector algebra, robust weights, workgroup reductions, projection + image
// sampling, shading, the finite-difference stencil on the compact band, ELL column numbering.  Internal; the public
// boundary is include/psgsdf.h.
//
// Reference arithmetic is cited per function (paths relative to /root/reference/cpp/include/).  Per-observation arithmetic
// is float32 in the reference's operation order; sums over many observations are accumulated in double.
//
// Kernel map:
//   band.hip       : k_select_vis, k_band_flags, k_scan_*, k_band_fill, k_band_nb, k_band_scatter, k_upsample, k_derive, k_obs_*, k_sum_parts
//   sweeps.hip     : k_init_albedo, k_energy, k_sweep_albedo (voxel-major, set-bit iteration); k_sweep_light, k_sweep_pose
//                    (frame-major, wave + LDS reduction); k_solve_light, k_solve_pose (LDL^T per frame)
//   dist.hip       : k_sweep_dist, k_assemble (per-pass path only)
//   pcg.hip        : k_cgp_solve (the whole distance step as ONE persistent kernel: assembly into LDS, pipelined Jacobi-PCG in double, update; the
//                    default), k_cgf_solve (the same with the classic recurrences), k_cgf_init / k_cgf_pass / k_cgf_sum (one kernel per pass:
//                    bands that do not fit the LDS, multi-rank contexts without mappings, the fall-back), k_apply_dist
//   albedo_reg.hip : k_areg_*                          frontend.hip: k_integrate, k_normals_h/v, k_track, k_try_pack_f32, k_pack_rgb8
//   extract.hip    : k_box_*, k_mc_count / k_mc_emit (marching cubes), k_pc_flags / k_pc_fill (point clouds), k_sdf_crop, k_cscan_*
//   comm.hip       : k_halo_push / k_halo_pull, k_xr_probe, k_xr_nonces, k_xr_closed (multi-rank exchanges through IPC-mapped memory)
#pragma once
#include "engine.h"
#include <float.h>

// Floating-point contraction: ON for the Jacobian / normal-equation algebra (FMA: fewer instructions, one rounding
// less), OFF inside the functions whose results feed discrete decisions or must match the CPU reference build bit for
// bit (baseline x86-64, no FMA): normalisation, the surface point, the projection (floor / in-image test), FD gradient.
// PSG_STRICT (development builds only: `make strict STRICT=<mask>` -> libpsgsdf_strict<mask>.so; the product library is built with 0): every deviation of the
// device arithmetic from the reference's can be switched off, bit by bit, to measure what it contributes to the
// drift of a whole optimisation (tools/deviations.py, profiles/r06_notes.md):
//   1  no FMA contraction anywhere in the per-observation algebra (the oracle -- and the reference's baseline x86-64 build -- has none)
//   2  the robust weights / losses with IEEE divisions and logf, `r / lambda` as Optimizer.cpp:140-186 writes it (not r * (1 / lambda), v_rcp_f32, v_log_f32)
//   4  bilinear weights partly in double as Auxilary.h:47 evaluates them, 1 / z through double (OptimizerAux.cpp:219)
//   8  sums over a voxel's / a thread's observations in double (the oracle's accumulators) instead of float
//  16  the Jacobian chains in the reference's order of evaluation: image_grad x pi_grad first, then R^T, then the direction (PsOptimizerJa.cpp:78-100,
//      160-289; LedOptimizerJa.cpp:117-218), the SH2 / LED shading terms per channel and stencil slot -- not contracted from the right
// (The solver of the light / pose blocks and the recurrences of the distance solve are run-time switches: PSGSDF_FRAME_SOLVE, PSGSDF_PCG_PIPELINE / _PERSIST.)
#ifndef PSG_STRICT
#define PSG_STRICT 0
#endif
#if PSG_STRICT & 1
#pragma clang fp contract(off)
#else
#pragma clang fp contract(fast)
#endif

namespace psg {
#if PSG_STRICT & 8
typedef double obs_acc_t;      // accumulator of a sum over observations
#else
typedef float obs_acc_t;
#endif

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
typedef float f2_t __attribute__((ext_vector_type(2)));      // two floats in an aligned register pair: a * b + c on them is ONE v_pk_fma_f32 (the scalar factor of a product is broadcast)
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
// launches K<model, loss, img>: the three shading models x {Cauchy loss and the image format compiled in, both decided at run time}.
// frame_major: the kernel addresses one frame's image at a time, so float taps always fit 32-bit offsets.
#define PSG_LAUNCH_SWEEP(K, args, frame_major, grid, block, shmem, stream, ...) do { \
        const int img_ = (args).im.u8 ? 1 : (((frame_major) || (args).im.idx32) ? 0 : -1); \
        if ((args).rob.loss == 1 && img_ == 0) { \
            if ((args).model == 0) hipLaunchKernelGGL((K<0, 1, 0>), grid, block, shmem, stream, __VA_ARGS__); \
            else if ((args).model == 1) hipLaunchKernelGGL((K<1, 1, 0>), grid, block, shmem, stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((K<2, 1, 0>), grid, block, shmem, stream, __VA_ARGS__); \
        } else if ((args).rob.loss == 1 && img_ == 1) { \
            if ((args).model == 0) hipLaunchKernelGGL((K<0, 1, 1>), grid, block, shmem, stream, __VA_ARGS__); \
            else if ((args).model == 1) hipLaunchKernelGGL((K<1, 1, 1>), grid, block, shmem, stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((K<2, 1, 1>), grid, block, shmem, stream, __VA_ARGS__); \
        } else { \
            if ((args).model == 0) hipLaunchKernelGGL((K<0, -1, -1>), grid, block, shmem, stream, __VA_ARGS__); \
            else if ((args).model == 1) hipLaunchKernelGGL((K<1, -1, -1>), grid, block, shmem, stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((K<2, -1, -1>), grid, block, shmem, stream, __VA_ARGS__); \
        } } while (0)
// LOSS: the loss function as a compile-time constant (the sweeps are instantiated for Cauchy, what every shipped config uses) or -1 =
// decided at run time -- six wavefront-uniform switches per observation cost the sweeps 3-4 us each
template <int LOSS = -1>
__device__ __forceinline__ float robust_weight(const Robust& rb, float r) {
#if PSG_STRICT & 2
    switch (LOSS >= 0 ? LOSS : rb.loss) {      // Optimizer.cpp:140-161 as written
        case 1: { float x = r / rb.lambda; return 1.0f / (1.0f + x * x); }
        case 3: { float x = r / rb.lambda; float w = (1.0f - x * x); w = w * w; return (r * r < rb.lambda_sq) ? w : 0.0f; }
        case 2: { float w = rb.lambda * fabsf(1.0f / r); return (r * r < rb.lambda_sq) ? 1.0f : w; }
        case 4: return (r * r < rb.lambda_sq) ? 1.0f : 0.0f;
        default: return 1.0f;
    }
#endif
    switch (LOSS >= 0 ? LOSS : rb.loss) {
        case 1: { float x = r * rb.inv_lambda; return __builtin_amdgcn_rcpf(1.0f + x * x); }   // v_rcp_f32: 1 ulp
        case 3: { float x = r * rb.inv_lambda; float w = (1.0f - x * x); w = w * w; return (r * r < rb.lambda_sq) ? w : 0.0f; }
        case 2: { float w = rb.lambda * fabsf(__builtin_amdgcn_rcpf(r)); return (r * r < rb.lambda_sq) ? 1.0f : w; }
        case 4: return (r * r < rb.lambda_sq) ? 1.0f : 0.0f;
        default: return 1.0f;
    }
}
template <int LOSS = -1>
__device__ __forceinline__ float robust_loss(const Robust& rb, float r) {
#if PSG_STRICT & 2
    switch (LOSS >= 0 ? LOSS : rb.loss) {      // Optimizer.cpp:164-186 as written
        case 1: { float x = r / rb.lambda; return logf(1.0f + x * x); }
        case 3: { float x = r / rb.lambda; float u = 1.0f - x * x; float v = 1.0f - u * u * u; return (r * r < rb.lambda_sq) ? v : 1.0f; }
        case 2: return (r * r < rb.lambda_sq) ? 0.5f * (r * r) : rb.lambda * (fabsf(r) - 0.5f * rb.lambda * 1.0f);
        case 4: { float x = fminf(fmaxf(r, -rb.lambda), rb.lambda); return x * x; }
        default: return r * r;
    }
#endif
    switch (LOSS >= 0 ? LOSS : rb.loss) {
        case 1: { float x = r * rb.inv_lambda; return __builtin_amdgcn_logf(1.0f + x * x) * 0.693147180559945f; }   // v_log_f32 (argument >= 1: no denormal scaling) x ln 2: 2 instructions instead of logf's 13, <= 2 ulp
        case 3: { float x = r * rb.inv_lambda; float u = 1.0f - x * x; float v = 1.0f - u * u * u; return (r * r < rb.lambda_sq) ? v : 1.0f; }
        case 2: return (r * r < rb.lambda_sq) ? 0.5f * (r * r) : rb.lambda * (fabsf(r) - 0.5f * rb.lambda * 1.0f);
        case 4: { float x = fminf(fmaxf(r, -rb.lambda), rb.lambda); return x * x; }
        default: return r * r;
    }
}

// wavefront (64 lanes) and workgroup reductions; one device-scope atomic per workgroup
// fused PCG (pcg.hip): per-workgroup partial sums [2 parity][kCgfSums][kPcgMaxBlocks]; k = -1 / kind 6 holds |b|^2 of the initialisation
constexpr int kCgfSums = 7;
__device__ __forceinline__ double* fpart(double* part, int k, int kind) { return part + ((size_t)((k & 1) * kCgfSums + kind)) * kPcgMaxBlocks; }

// Cross-lane adds through DPP (register-to-register, no LDS round trip as with ds_bpermute): butterfly inside each row of 16
// lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows
// 2 and 3; lane 63 holds the total, which is broadcast through an SGPR.  Fixed order, so still run-to-run deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);     // row_mirror
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 -> rows 1, 3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 -> rows 2, 3
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// Eight wavefront sums at once as a reduce-scatter: lanes exchange HALF of their values with lane^1, then half of the rest with
// lane^2 (8 -> 4 -> 2 values per lane), the two survivors are summed over the four lanes of the row that share the low lane bits
// (row_ror 4, 8) and over the four rows (two shuffles).  66 instructions instead of 8 x 20; afterwards EVERY lane holds the
// totals of the values  4 (lane & 1) + (lane & 2) + {0, 1}  in t0, t1.
__device__ __forceinline__ void wave_sum8(const double (&v)[8], double& t0, double& t1) {
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2;
    double u[4], t[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const double keep = b0 ? v[4 + j] : v[j], give = b0 ? v[j] : v[4 + j]; u[j] = keep + dpp_get<0xB1>(give); }   // quad_perm [1,0,3,2]
#pragma unroll
    for (int j = 0; j < 2; ++j) { const double keep = b1 ? u[2 + j] : u[j], give = b1 ? u[j] : u[2 + j]; t[j] = keep + dpp_get<0x4E>(give); }   // quad_perm [2,3,0,1]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        t[j] += dpp_get<0x124>(t[j]);      // row_ror:4
        t[j] += dpp_get<0x128>(t[j]);      // row_ror:8
        t[j] += __shfl_xor(t[j], 16, 64);
        t[j] += __shfl_xor(t[j], 32, 64);
    }
    t0 = t[0]; t1 = t[1];
}
// Four wavefront sums at once: the same reduce-scatter with half the values -- and, value for value, the SAME additions in the same order as wave_sum8
// computes for its values 0..3 (pairs lane^1, lane^2, then row_ror 4 / 8 and the two cross-row shuffles): the same bits for 31 instead of 66 instructions.
// Afterwards every lane holds the total of value  2 (lane & 1) + ((lane >> 1) & 1).
__device__ __forceinline__ double wave_sum4(const double (&v)[4]) {
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2;
    double u[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const double keep = b0 ? v[2 + j] : v[j], give = b0 ? v[j] : v[2 + j]; u[j] = keep + dpp_get<0xB1>(give); }   // quad_perm [1,0,3,2]
    const double keep = b1 ? u[1] : u[0], give = b1 ? u[0] : u[1];
    double t = keep + dpp_get<0x4E>(give);                                                                                                     // quad_perm [2,3,0,1]
    t += dpp_get<0x124>(t);      // row_ror:4
    t += dpp_get<0x128>(t);      // row_ror:8
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    return t;
}
// lanes 0..3 of wavefront w file the four totals under red[value * NW + w] (the layout of wave_sum8_store)
template <int NW>
__device__ __forceinline__ void wave_sum4_store(double t, double* red, int w) {
    const int lane = threadIdx.x & 63;
    if (lane < 4) red[(2 * (lane & 1) + ((lane >> 1) & 1)) * NW + w] = t;
}
// lanes 0..3 of wavefront w file the eight totals under red[value * NW + w]
template <int NW>
__device__ __forceinline__ void wave_sum8_store(double t0, double t1, double* red, int w) {
    const int lane = threadIdx.x & 63;
    if (lane < 4) { const int base = 4 * (lane & 1) + (lane & 2); red[base * NW + w] = t0; red[(base + 1) * NW + w] = t1; }
}
// NV wavefront sums of float accumulators, in double, eight at a time -> row[0..NV) (written by lanes 0..3)
template <int NV, class T>
__device__ __forceinline__ void wave_sums_to(const T* acc, double* row) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int g = 0; g < (NV + 7) / 8; ++g) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 8 * g + q < NV ? (double)acc[8 * g + q] : 0.0;
        double t0, t1; wave_sum8(v, t0, t1);
        const int k = 8 * g + 4 * (lane & 1) + (lane & 2);
        if (lane < 4) { if (k < NV) row[k] = t0; if (k + 1 < NV) row[k + 1] = t1; }
    }
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
__device__ __forceinline__ void block_part_store(double v, double* part_slot, double* red, int bid = -1) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
        part_slot[bid >= 0 ? bid : (int)blockIdx.x] = s;
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
// Workgroups are dealt to the 8 XCDs round-robin in dispatch order (physical id mod 8).  xcd_remap turns the physical id into a LOGICAL one such that
// XCD x works on the contiguous logical range [x G/8, (x+1) G/8): neighbouring band rows (and the image regions they project to) then meet in ONE L2
// instead of in all eight.  A bijection on [0, G) for any G; the logical id replaces blockIdx everywhere (rows AND partial-sum slots), so results do
// not depend on the mapping.
__device__ __forceinline__ unsigned xcd_remap(unsigned pid, unsigned G) {
    const unsigned x = pid & 7u, s = pid >> 3, q = G >> 3, rem = G & 7u;
    return x * q + (x < rem ? x : rem) + s;
}
// frame-major grids (chunks, F): logical ids run chunk-major, so an XCD owns a range of CHUNKS (= of band rows: the observation lists ascend) of every frame
__device__ __forceinline__ void fm_ids(const SweepArgs& a, int& cx, int& f) {
    cx = blockIdx.x; f = blockIdx.y;
    if (a.xcd_map & 1) {
        const unsigned L = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
        cx = (int)(L / gridDim.y); f = (int)(L - (unsigned)cx * gridDim.y);
    }
}
// voxel-major grids: the logical workgroup id (rows bid * blockDim.x ..., partial-sum slot bid)
// striped variant: XCD x owns the stripes x, x + 8, ... of T consecutive logical ids (locality within a stripe, load spread over the whole range);
// the ids behind the last complete round of 8 stripes keep their physical order
__device__ __forceinline__ unsigned xcd_remap_striped(unsigned pid, unsigned G, unsigned T) {
    const unsigned Gc = G / (8u * T) * (8u * T);
    if (pid >= Gc) return pid;
    const unsigned x = pid & 7u, s = pid >> 3;
    return ((s / T) * 8u + x) * T + s % T;
}
__device__ __forceinline__ int vm_bid(const SweepArgs& a, int bit = 2, bool per_obs = false) {
    if (bit == 4 && (a.xcd_map & 32) && a.vm_order) return a.vm_order[blockIdx.x];      // distance sweep: heaviest blocks first (longest-processing-time order: a shorter tail)
    if (per_obs && (a.xcd_map & 64) && a.vm_order) return a.vm_order[blockIdx.x];       // (the other per-observation voxel-major kernels: albedo sweep, energy)
    if (!(a.xcd_map & bit)) return (int)blockIdx.x;
    const unsigned T = (unsigned)a.xcd_map >> 8;
    return T ? (int)xcd_remap_striped(blockIdx.x, gridDim.x, T) : (int)xcd_remap(blockIdx.x, gridDim.x);
}
#define PART(a, slot) ((a).acc.part + (size_t)(slot) * (a).acc.PB)
// Value i of an n-value read-back slot.  key != 0: the slot lives in the host-mapped mailbox and the value travels with its check word
// bits(v) ^ (key + i), n doubles further on (engine.h FoldReq): the host takes the value only when the pair matches, so a read-back can
// never be consumed before it has arrived, whatever the marker or status word the host was waiting for says.  The writing thread ends with
// mbox_commit (system-scope release: the words leave this XCD's L2 for host memory now, not at some later write-back).
__device__ __forceinline__ void mbox_put(double* out, int n, int i, double v, unsigned long long key) {
    out[i] = v;
    if (key) reinterpret_cast<unsigned long long*>(out)[n + i] = (unsigned long long)__double_as_longlong(v) ^ (key + (unsigned long long)i);
}
__device__ __forceinline__ void mbox_commit(unsigned long long key) { if (key) __threadfence_system(); }
// 8-byte write-through store at system scope: the word reaches memory (another device's view) without a later write-back
__device__ __forceinline__ void store8_system(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
// Multi-rank fold (ONE WAVEFRONT, all 64 lanes call; lane 0 holds the slab's n <= 8 sums): they go into slot [parity][rank] of EVERY rank's mailbox
// region (payload, drained, then a flag carrying the fold's number); lane r waits for rank r's flag -- R polls side by side instead of one after the
// other: a fold sits on the iteration's critical path several times per iteration and a poll of an uncached word costs ~0.7 us -- and loads rank r's
// sums; they are added in rank order, the same bits on every rank and in every lane.  A rank that never delivers (bounded wait) turns the sums into
// NaN and leaves its mark in the region's "late" word; the host reports it.
__device__ __forceinline__ void fold_exchange(const XfTable* xf, long long epoch, int n, double* t) {
    const XfTable& x = *xf;
    const int Rk = x.n_ranks, buf = (int)(epoch & 1), lane = (int)(threadIdx.x & 63);
    const double tag = (double)epoch;
    const long long slot = (long long)buf * Rk + x.rank;
    double* const mine = x.region[x.rank];
    for (int s = 0; s < n; ++s) t[s] = __shfl(t[s], 0, 64);
    if (lane == 0) {
        for (int r = 0; r < Rk; ++r) for (int s = 0; s < n; ++s) store8_system(x.region[r] + x.spay + slot * 8 + s, t[s]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int r = 0; r < Rk; ++r) __hip_atomic_store(x.region[r] + x.sflg + slot, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    bool late = false;
    if (lane < Rk) {
        int spins = 0;
        while (__hip_atomic_load(mine + x.sflg + (long long)buf * Rk + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != tag) { __builtin_amdgcn_s_sleep(2); if (++spins > x.spin_max) { late = true; break; } }
    }
    const bool any_late = __builtin_amdgcn_ballot_w64(late) != 0ull;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    for (int s = 0; s < n; ++s) {
        const double v = lane < Rk ? __hip_atomic_load(mine + x.spay + ((long long)buf * Rk + lane) * 8 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
        double tot = 0.0;
        for (int r = 0; r < Rk; ++r) tot += __shfl(v, r, 64);
        t[s] = any_late ? __builtin_nan("") : tot;
    }
    if (any_late && lane == 0) __hip_atomic_store(mine + kXrLate + 1, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (out of band: the host tells a lost peer from a NaN state, engine.hip deliver_first)
}
// a.fold: sum the partial slots the PREVIOUS kernel left behind (first workgroup only; all its threads must call).
// The calling kernel must not write the folded slots itself (engine.hip: take_fold checks).
__device__ __forceinline__ void fold_pending(const SweepArgs& a, double* red /*[kBlock/64]*/) {
    if (a.fold.n == 0 || blockIdx.x != 0 || blockIdx.y != 0) return;
    double t[4];
    for (int s = 0; s < a.fold.n; ++s) t[s] = block_total(PART(a, a.fold.id[s]), a.fold.nblk, red);
    if (threadIdx.x < 64) {      // (the exchange between the ranks takes the whole first wavefront)
        if (a.fold.xf) fold_exchange(a.fold.xf, a.fold.xf_epoch, a.fold.n, t);
        if (threadIdx.x == 0) {
            for (int s = 0; s < a.fold.n; ++s) mbox_put(a.fold.out, a.fold.n, s, t[s], a.fold.key);
            mbox_commit(a.fold.key);
        }
    }
    __syncthreads();
}

// frame records (pose, light) of all keyframes staged in dynamic LDS: F * 96 B (<= 60 KiB, F <= kMaxFramesLds)
extern __shared__ __align__(16) unsigned char psg_dyn_smem[];
// record of frame f in the LDS copy: a 24-bit multiply for the offset (v_mul_u32_u24, full rate; f * sizeof through v_mul_lo_u32 is quarter rate)
__device__ __forceinline__ const FrameP& frame_at(const FrameP* sf, int f) {
    return *reinterpret_cast<const FrameP*>(reinterpret_cast<const char*>(sf) + __umul24((unsigned)f, (unsigned)sizeof(FrameP)));
}
__device__ __forceinline__ void load_frames(FrameP* sf, const FrameP* frames, int F) {
    const float* src = (const float*)frames; float* dst = (float*)sf;
    for (int i = threadIdx.x; i < F * (int)(sizeof(FrameP) / 4); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// projection + image sampling
// ------------------------------------------------------------------------------------------
struct Proj { float p[3]; float m, n, z_inv; bool ok; };
// The reference projects a second time inside its Jacobians, with fx*px/pz instead of fx*px*(1/pz) (PsOptimizerJa.cpp:70-76,
// LedOptimizerJa.cpp:40-46): the pixel coordinate can differ by one ulp, and where that crosses a pixel boundary the image GRADIENT is taken
// from the neighbouring cell (at a silhouette that one observation moves a pose block by 1e-3 of its largest entry).  The sweeps that
// need gradients therefore carry both coordinates: (m, n) for the colour, (mj, nj) for the gradient and the Jacobian's own in-image test.
// a / b correctly rounded from y = RN(1 / b) (Markstein: q0 = RN(a y), r = a - b q0 exactly (fma), q = RN(q0 + r y)): the quotient's bits for
// three instructions instead of a division's ten, wherever one divisor serves several quotients
__device__ __forceinline__ float div_by(float a, float b, float y) { const float q0 = a * y; return __builtin_fmaf(__builtin_fmaf(-q0, b, a), y, q0); }
struct ProjJ { float mj, nj; bool ok; };
__device__ __forceinline__ ProjJ project_jac(const Proj& pr, const Cam& cam) {
#pragma clang fp contract(off)
    ProjJ o;
    // a / pz correctly rounded from the correctly rounded reciprocal project() already holds (Markstein: q0 = RN(a y), r = a - pz q0 exactly
    // (fma), q = RN(q0 + r y) is RN(a / pz) when y = RN(1 / pz)): three instructions instead of the ten of a full division, same bits
    o.mj = div_by(cam.fx * pr.p[0], pr.p[2], pr.z_inv) + cam.cx;
    o.nj = div_by(cam.fy * pr.p[1], pr.p[2], pr.z_inv) + cam.cy;
    o.ok = (o.mj >= 0.f && o.mj < (float)cam.W && o.nj >= 0.f && o.nj < (float)cam.H);
    return o;
}

// OptimizerAux.cpp:207-226 (surface point precomputed in xs = x_v - d*normalized(grad))
__device__ __forceinline__ Proj project(const float* xs, const FrameP& fp, const Cam& cam) {
#pragma clang fp contract(off)
    Proj o;
    float tmp[3] = {xs[0] - fp.t[0], xs[1] - fp.t[1], xs[2] - fp.t[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) o.p[i] = (fp.R[0 * 3 + i] * tmp[0] + fp.R[1 * 3 + i] * tmp[1]) + fp.R[2 * 3 + i] * tmp[2];
    // reference: (float)(1. / point[2]) evaluated in double (OptimizerAux.cpp:219); the correctly rounded float
    // reciprocal differs from that only in double-rounding corner cases (~1e-8 of all inputs)
#if PSG_STRICT & 4
    const float z_inv = (float)(1.0 / (double)o.p[2]);      // OptimizerAux.cpp:219
#else
    const float z_inv = 1.0f / o.p[2];
#endif
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
// Output convention (col=m_col / row=n_row are the projected pixel coordinates): I = bilinear colour, gu = d/d(col), gv = d/d(row).
template <bool GRAD>
__device__ __forceinline__ void interp_taps(const float* a00, const float* a01, const float* a10, const float* a11, float fm, float fn, float gm, float gn, float* I, float* gu, float* gv,
                                            float m = 0.f, float n = 0.f, int x = 0, int y = 0) {   // (fm, fn): fractions of the colour's coordinates (row, column), (gm, gn): of the gradient's; (m, n, x, y): the coordinates themselves (PSG_STRICT & 4)
#if PSG_STRICT & 4
    {   // Auxilary.h:41-61 as C++ evaluates it: `y + 1. - n` is double, `(m - x)` float -> three of the four weights are double products, rounded to float per tap
#pragma clang fp contract(off)
        const double w1 = ((double)y + 1.0 - (double)n) * (double)(m - (float)x);
        const double w2 = ((double)y + 1.0 - (double)n) * ((double)x + 1.0 - (double)m);
        const float w3 = (n - (float)y) * (m - (float)x);
        const double w4 = (double)(n - (float)y) * ((double)x + 1.0 - (double)m);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float t1 = (float)((double)a10[ch] * w1), t2 = (float)((double)a00[ch] * w2), t3 = a11[ch] * w3, t4 = (float)((double)a01[ch] * w4);
            I[ch] = ((t1 + t2) + t3) + t4;
        }
    }
#else
    // reference: weights partly in double (Auxilary.h:47); float weights agree to ~1e-7 relative
    const float w1 = (1.0f - fn) * fm, w2 = (1.0f - fn) * (1.0f - fm), w3 = fn * fm, w4 = fn * (1.0f - fm);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) I[ch] = ((a10[ch] * w1 + a00[ch] * w2) + a11[ch] * w3) + a01[ch] * w4;
#endif
    if (GRAD) {
        const float w01 = gm, w11 = gn, w00 = 1.0f - gm, w10 = 1.0f - gn;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            gu[ch] = w00 * (a01[ch] - a00[ch]) + w01 * (a11[ch] - a10[ch]);
            gv[ch] = w10 * (a10[ch] - a00[ch]) + w11 * (a11[ch] - a01[ch]);
        }
    }
}
// last row / column: nearest sample, one-sided differences (Auxilary.h:55-57,90-121); TEX(row, col, out[3]) clamps like pix()
template <bool GRAD, class TEX>
__device__ __forceinline__ void sample_border(const TEX& tex, const Cam& cam, int x, int y, float m, float n, float* I, float* gu, float* gv) {
    float p[3]; tex(x, y, p);
    I[0] = p[0]; I[1] = p[1]; I[2] = p[2];
    if (GRAD) {
        float w01 = m - (float)x, w11 = n - (float)y;
        float w00 = (float)(1.0 - (double)w01), w10 = (float)(1.0 - (double)w11);
        float q_y1[3], q_ym[3], q_x1ym[3], q_x1[3], q_xm[3], q_xmy1[3];
        tex(x, y + 1, q_y1); tex(x, y - 1, q_ym); tex(x + 1, y - 1, q_x1ym); tex(x + 1, y, q_x1); tex(x - 1, y, q_xm); tex(x - 1, y + 1, q_xmy1);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            if ((x + 1) >= cam.H) gu[ch] = q_y1[ch] - p[ch];
            else { float v0 = -q_ym[ch] + p[ch]; float v1 = -q_x1ym[ch] + q_x1[ch]; gu[ch] = w00 * v0 + w01 * v1; }
            if ((x + 1) >= cam.H && (y + 1) < cam.W) { float v0 = -q_xm[ch] + p[ch]; float v1 = -q_xmy1[ch] + q_y1[ch]; gv[ch] = w10 * v0 + w11 * v1; }
            else gv[ch] = q_x1[ch] - p[ch];
        }
    }
}
// a * b + c with 24-bit a and a wavefront-uniform 24-bit b: v_mad_u32_u24 (full rate).  Written as inline assembly because the compiler turns
// __umul24(a, b) + c into v_mad_u64_u32, which issues at quarter rate like v_mul_lo_u32.
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b_uniform, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(__builtin_amdgcn_readfirstlane(b_uniform)), "v"(c));
    return r;
}
// float RGB images [F][H][W][3].
// idx32: the whole stack is < 4 GiB, so a tap's byte offset fits 32 bits and the loads take the scalar-base form (one address
// register, no 64-bit integer multiply-adds, which issue at quarter rate and made up ~10 % of a sweep's instruction slots).
template <bool GRAD>
__device__ __forceinline__ void sample_cell(const float* base, int frame, bool idx32, const Cam& cam, float m_col, float n_row, float mj_col, float nj_row, float* I, float* gu, float* gv) {
    const float m = n_row, n = m_col;  // names of Auxilary.h: m = row, n = column
    int x = (int)floorf(m), y = (int)floorf(n);
    if ((x + 1) < cam.H && (y + 1) < cam.W) {
        float a00[3], a01[3], a10[3], a11[3];
        if (idx32) {
            // 24-bit multiplies (v_mad_u32_u24, full rate; 32-bit v_mul_lo_u32 / v_mad_u64_u32 issue at quarter rate and were 15 % of an
            // energy sweep's issue cycles): F * H and 3 W are < 2^24 (engine.hip: idx32), the products fit 32 bits
            const unsigned rowi = (__builtin_constant_p(frame) && frame == 0) ? (unsigned)x : mad24((unsigned)frame, (unsigned)cam.H, (unsigned)x);
            const unsigned e = mad24(rowi, 3u * (unsigned)cam.W, 3u * (unsigned)y);
            const float* p00 = (const float*)((const char*)base + (size_t)(e << 2));
            const float* p10 = (const float*)((const char*)base + (size_t)((e + 3u * (unsigned)cam.W) << 2));
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { a00[ch] = p00[ch]; a01[ch] = p00[3 + ch]; a10[ch] = p10[ch]; a11[ch] = p10[3 + ch]; }
        } else {
            const float* p00 = base + ((size_t)frame * cam.H * cam.W + (size_t)x * cam.W + y) * 3;
            const float* p10 = p00 + (size_t)cam.W * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { a00[ch] = p00[ch]; a01[ch] = p00[3 + ch]; a10[ch] = p10[ch]; a11[ch] = p10[3 + ch]; }
        }
        interp_taps<GRAD>(a00, a01, a10, a11, m - (float)x, n - (float)y, nj_row - (float)x, mj_col - (float)y, I, gu, gv, m, n, x, y);
    } else {
        const float* img = base + (size_t)frame * cam.H * cam.W * 3;   // (64-bit multiply-adds issue at quarter rate: keep them on this rare path)
        auto tex = [&](int row, int col, float* o) { const float* q = pix(img, cam, row, col); o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; };
        sample_border<GRAD>(tex, cam, x, y, nj_row, mj_col, I, gu, gv);
    }
}
// colour at (m_col, n_row), gradient at (mj_col, nj_row): one set of taps unless the two fall into different pixel cells
template <bool GRAD>
__device__ __forceinline__ void sample(const float* base, int frame, bool idx32, const Cam& cam, float m_col, float n_row, float mj_col, float nj_row, float* I, float* gu, float* gv) {
    sample_cell<GRAD>(base, frame, idx32, cam, m_col, n_row, mj_col, nj_row, I, gu, gv);
    // the Jacobian's projection landed in the neighbouring cell (~1e-5 of the observations): redo the gradient there -- unless that cell is
    // outside the image (coordinate < 0: the Jacobian's own in-image test fails, its row gets weight 0, and cell -1 must not be read)
    if (GRAD && (floorf(nj_row) != floorf(n_row) || floorf(mj_col) != floorf(m_col)) && nj_row >= 0.f && mj_col >= 0.f) {
        float Iu[3];
        sample_cell<true>(base, frame, idx32, cam, mj_col, nj_row, mj_col, nj_row, Iu, gu, gv);
    }
}
// 8-bit RGB images as the reference's loader receives them (ImageLoader.h:167-181: cv::imread, then convertTo(CV_32FC3, 1/255)),
// stored as one RGBA8 word per pixel [F][H][W]: the two taps of an image row are ONE 8-byte load instead of two 12-byte ones and a
// cache line holds 16 pixels instead of 5: ~3 % more iterations per second once the format is a template parameter of the sweeps
// (behind a run-time branch it measured no faster than the float path), a quarter of the upload, a third of the resident bytes.
// (float)byte * scale is exactly the float the reference's conversion produces.
__device__ __forceinline__ void unpack_rgb8(unsigned w, float scale, float* o) {
    // __fmul_rn: a rounded product, never contracted into a following add -- the taps must be the floats the reference holds
    o[0] = __fmul_rn((float)(w & 0xffu), scale); o[1] = __fmul_rn((float)((w >> 8) & 0xffu), scale); o[2] = __fmul_rn((float)((w >> 16) & 0xffu), scale);
}
template <bool GRAD>
__device__ __forceinline__ void sample_u8_cell(const unsigned* base, float scale, int frame, const Cam& cam, float m_col, float n_row, float mj_col, float nj_row, float* I, float* gu, float* gv) {
    const float m = n_row, n = m_col;
    int x = (int)floorf(m), y = (int)floorf(n);
    if ((x + 1) < cam.H && (y + 1) < cam.W) {
        const unsigned rowi = (__builtin_constant_p(frame) && frame == 0) ? (unsigned)x : mad24((unsigned)frame, (unsigned)cam.H, (unsigned)x);
        const unsigned e = mad24(rowi, (unsigned)cam.W, (unsigned)y);      // < 2^30 pixels, < 2^24 image rows (api.hip checks)
        const unsigned* p0 = (const unsigned*)((const char*)base + (size_t)(e << 2));
        const unsigned* p1 = (const unsigned*)((const char*)base + (size_t)((e + (unsigned)cam.W) << 2));
        const unsigned t00 = p0[0], t01 = p0[1], t10 = p1[0], t11 = p1[1];
        float a00[3], a01[3], a10[3], a11[3];
        unpack_rgb8(t00, scale, a00); unpack_rgb8(t01, scale, a01); unpack_rgb8(t10, scale, a10); unpack_rgb8(t11, scale, a11);
        interp_taps<GRAD>(a00, a01, a10, a11, m - (float)x, n - (float)y, nj_row - (float)x, mj_col - (float)y, I, gu, gv, m, n, x, y);
    } else {
        const unsigned* img = base + (size_t)frame * cam.H * cam.W;
        auto tex = [&](int row, int col, float* o) {
            row = row < 0 ? 0 : (row >= cam.H ? cam.H - 1 : row);
            col = col < 0 ? 0 : (col >= cam.W ? cam.W - 1 : col);
            unpack_rgb8(img[(size_t)row * cam.W + col], scale, o);
        };
        sample_border<GRAD>(tex, cam, x, y, nj_row, mj_col, I, gu, gv);
    }
}
template <bool GRAD>
__device__ __forceinline__ void sample_u8(const unsigned* base, float scale, int frame, const Cam& cam, float m_col, float n_row, float mj_col, float nj_row, float* I, float* gu, float* gv) {
    sample_u8_cell<GRAD>(base, scale, frame, cam, m_col, n_row, mj_col, nj_row, I, gu, gv);
    if (GRAD && (floorf(nj_row) != floorf(n_row) || floorf(mj_col) != floorf(m_col)) && nj_row >= 0.f && mj_col >= 0.f) {   // as in sample()
        float Iu[3];
        sample_u8_cell<true>(base, scale, frame, cam, mj_col, nj_row, mj_col, nj_row, Iu, gu, gv);
    }
}
// either format, chosen by the (wavefront-uniform) image source of the launch
// IMG: 0 = float RGB with 32-bit tap offsets, 1 = RGBA8 words, -1 = look at the source (two more uniform branches per observation)
// (mj_col, nj_row): where the gradient is taken (project_jac); pass (m_col, n_row) again when GRAD is off
template <bool GRAD, int IMG = -1>
__device__ __forceinline__ void sample(const ImgSrc& s, int frame, const Cam& cam, float m_col, float n_row, float mj_col, float nj_row, float* I, float* gu, float* gv) {
    if (IMG == 0) sample<GRAD>(s.f32, frame, true, cam, m_col, n_row, mj_col, nj_row, I, gu, gv);
    else if (IMG == 1) sample_u8<GRAD>(s.u8, s.scale, frame, cam, m_col, n_row, mj_col, nj_row, I, gu, gv);
    else if (s.u8) sample_u8<GRAD>(s.u8, s.scale, frame, cam, m_col, n_row, mj_col, nj_row, I, gu, gv);
    else sample<GRAD>(s.f32, frame, s.idx32, cam, m_col, n_row, mj_col, nj_row, I, gu, gv);
}
template <bool GRAD, int IMG = -1>
__device__ __forceinline__ void sample(const ImgSrc& s, int frame, const Cam& cam, float m_col, float n_row, float* I, float* gu, float* gv) {
    static_assert(!GRAD, "gradient samples carry the Jacobian's own pixel coordinates (project_jac)");
    sample<false, IMG>(s, frame, cam, m_col, n_row, m_col, n_row, I, gu, gv);
}

// The same sample in two halves (frame-major sweeps, round 6): taps_issue computes the cell and REQUESTS the four taps, taps_colour / taps_colour_grad
// weigh them.  A sweep issues observation i + 1 before it finishes observation i, so a thread has two observations' taps in flight and an iteration no
// longer starts with a wait for its own loads (k_sweep_light ran at 53 % VALU-busy with five wavefronts per SIMD: one exposed tap latency per
// observation).  Same loads, same arithmetic: the bits of sample().  Frame 0 of the launch's own image stack (frame-major kernels point the source at
// their frame); IMG 0 = float RGB with 32-bit offsets, 1 = RGBA8 words.
// Two rules keep the compiler from waiting for the loads where they are issued: (1) they are UNCONDITIONAL -- a lane whose observation is outside the
// image requests cell (0, 0) and ignores it (loads inside a divergent branch are waited for at the end of the branch, where the loaded registers are
// merged); (2) the finishing half contains NO load at all -- with one behind a rare branch, every use of the taps behind that branch waits for vmcnt(0),
// i.e. for the taps just requested for the next observation as well.  Hence the image's last row / column (nearest sample, Auxilary.h:55-57) is served
// from the four taps of the cell clamped into the image -- the nearest sample IS one of them -- and what needs taps outside the cell (the one-sided
// differences of the border, a gradient whose own projection fell into the neighbouring cell: ~1e-4 of the observations) is reported back to the caller.
template <int IMG> struct Taps {
    float a00[3], a01[3], a10[3], a11[3];      // IMG 0: the taps of cell (xc, yc)
    unsigned w[4];                             // IMG 1: the four packed words
    bool cell;                                 // the bilinear cell of the observation is inside the image
    bool bx, by;                               // border: the nearest sample is tap (bx, by) of the clamped cell
};
template <int IMG>
__device__ __forceinline__ void taps_issue(const ImgSrc& s, const Cam& cam, bool ok, float m_col, float n_row, Taps<IMG>& t) {
    static_assert(IMG == 0 || IMG == 1, "image format known at compile time");
    const int x = (int)floorf(n_row), y = (int)floorf(m_col);
    t.cell = ok && (x + 1) < cam.H && (y + 1) < cam.W;
    const int xc = ok ? max(min(x, cam.H - 2), 0) : 0, yc = ok ? max(min(y, cam.W - 2), 0) : 0;      // (ok: 0 <= x < H, 0 <= y < W; images have at least two rows and columns: api.hip)
    t.bx = x != xc; t.by = y != yc;
    if (IMG == 0) {
        const unsigned e = mad24((unsigned)xc, 3u * (unsigned)cam.W, 3u * (unsigned)yc);
        const float* p00 = (const float*)((const char*)s.f32 + (size_t)(e << 2));
        const float* p10 = (const float*)((const char*)s.f32 + (size_t)((e + 3u * (unsigned)cam.W) << 2));
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { t.a00[ch] = p00[ch]; t.a01[ch] = p00[3 + ch]; t.a10[ch] = p10[ch]; t.a11[ch] = p10[3 + ch]; }
    } else {
        const unsigned e = mad24((unsigned)xc, (unsigned)cam.W, (unsigned)yc);
        const unsigned* p0 = (const unsigned*)((const char*)s.u8 + (size_t)(e << 2));
        const unsigned* p1 = (const unsigned*)((const char*)s.u8 + (size_t)((e + (unsigned)cam.W) << 2));
        t.w[0] = p0[0]; t.w[1] = p0[1]; t.w[2] = p1[0]; t.w[3] = p1[1];
    }
}
// colour at (m_col, n_row) [+ image gradient at (mj_col, nj_row), project_jac].  GRAD: returns false -- nothing written -- when the gradient needs taps
// outside the cell; the caller evaluates such an observation with sample<true>().
template <int IMG, bool GRAD>
__device__ __forceinline__ bool taps_colour(const ImgSrc& s, const Cam& cam, const Taps<IMG>& t, float m_col, float n_row, float mj_col, float nj_row, float* I, float* gu, float* gv) {
    const float m = n_row, n = m_col;
    const float xf = floorf(m), yf = floorf(n);
    if (GRAD && (!t.cell || floorf(nj_row) != xf || floorf(mj_col) != yf)) return false;
    float c00[3], c01[3], c10[3], c11[3];      // (values, not pointers into either source: a select between pointers sends the taps through scratch memory)
    if (IMG == 1) { unpack_rgb8(t.w[0], s.scale, c00); unpack_rgb8(t.w[1], s.scale, c01); unpack_rgb8(t.w[2], s.scale, c10); unpack_rgb8(t.w[3], s.scale, c11); }
    else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { c00[ch] = t.a00[ch]; c01[ch] = t.a01[ch]; c10[ch] = t.a10[ch]; c11[ch] = t.a11[ch]; }
    }
    if (GRAD || t.cell) interp_taps<GRAD>(c00, c01, c10, c11, m - xf, n - yf, nj_row - xf, mj_col - yf, I, gu, gv);
    else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { const float lo = t.by ? c01[ch] : c00[ch], hi = t.by ? c11[ch] : c10[ch]; I[ch] = t.bx ? hi : lo; }      // sample_border: I = the pixel (x, y) itself
    }
    return true;
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
// Plane element at a band row with the byte offset computed in 32 BITS: uniform base pointer + zero-extended 32-bit offset
// selects the scalar-base form of global_load, ONE offset register shared by all the planes of a row.  (`p[j]` makes the
// compiler build a 64-bit address per plane -- two registers and a 64-bit shift-add each -- because j * 4 may not fit 32 bits.)
// Valid while a plane is < 4 GiB, i.e. Spad < 2^30 rows.
template <class T> __device__ __forceinline__ T at32(const T* p, unsigned byte_off) { return *(const T*)((const char*)p + (size_t)byte_off); }

// the packed mirror (Band::vp): every writer of rho goes through set_rho; k_derive rewrites the whole record
__device__ __forceinline__ void set_rho(const Band& b, int j, int ch, float v) { b.rho[ch][j] = v; reinterpret_cast<float*>(b.vp[ch] + j)[3] = v; }
__device__ __forceinline__ void load_vox(const Band& b, int j, Vox& v) {
    const unsigned o = (unsigned)j << 4;
    const float4 p0 = at32(b.vp[0], o), p1 = at32(b.vp[1], o), p2 = at32(b.vp[2], o);
    v.xs[0] = p0.x; v.xs[1] = p0.y; v.xs[2] = p0.z; v.rho[0] = p0.w;
    v.gn[0] = p1.x; v.gn[1] = p1.y; v.gn[2] = p1.z; v.rho[1] = p1.w;
    v.nfd[0] = p2.x; v.nfd[1] = p2.y; v.nfd[2] = p2.z; v.rho[2] = p2.w;
}

// The observations e0, e0 + kBlock, ... (< end, at most `rows` <= 64) of one thread of a frame-major sweep as a software pipeline, three stages deep
// (round 6): row index three observations ahead, voxel state two ahead, and observation k + 1 PROJECTED and its four taps REQUESTED (taps_issue) before
// `body` weighs and accumulates observation k -- an iteration used to begin with a wait for its own taps, and at 4-5 wavefronts per SIMD that latency
// was not covered (k_sweep_light: 53 % VALU-busy).  Every load is unconditional: index k is clamped to the thread's last observation (a repeat that is
// requested and never weighed).  Unrolled by two with the in-flight state in two fixed register sets (P0 / P1, vX / vY): a rotating copy at the loop's
// end would wait for the loads just issued.  `body` must not load (taps_colour's rule 2); an observation it cannot finish (returns false) is redone by
// `slow(v, pr)` behind the loop -- in the order of the observations, but after the thread's others: the sums of such a thread differ from the two-stage
// loop's in the order of their float additions (deterministically); every other thread's are the same bits.
template <int IMG> struct ObsPend { Vox v; Proj pr; Taps<IMG> ts; };
template <int IMG, class F, class G>
__device__ __forceinline__ void fm_for_each_obs(const Band& b, const FrameP& fp, const Cam& cam, const ImgSrc& img, int e0, int end, int rows, F&& body, G&& slow) {
    const int n = e0 < end ? min(rows, (end - e0 + kBlock - 1) / kBlock) : 0;
    if (n <= 0) return;
    const int eL = e0 + (n - 1) * kBlock;
    auto row_of_obs = [&](int k) { return b.obs_rows[min(e0 + k * kBlock, eL)]; };
    auto stage_a = [&](const Vox& v, ObsPend<IMG>& P) {
        P.v = v; P.pr = project(v.xs, fp, cam);
        taps_issue<IMG>(img, cam, P.pr.ok, P.pr.m, P.pr.n, P.ts);
    };
    Vox vX, vY; ObsPend<IMG> P0, P1;
    { Vox v0; load_vox(b, row_of_obs(0), v0); load_vox(b, row_of_obs(1), vX); stage_a(v0, P0); }
    int jY = row_of_obs(2);
    unsigned long long redo = 0ull;
    for (int k = 0; k < n; k += 2) {
        // [P0 = observation k in flight, vX = state of k + 1, jY = row of k + 2]
        const int jX = row_of_obs(k + 3);
        load_vox(b, jY, vY);
        stage_a(vX, P1);
        if (P0.pr.ok && !body(P0)) redo |= 1ull << k;
        // [P1 = observation k + 1 in flight, vY = state of k + 2, jX = row of k + 3]
        jY = row_of_obs(k + 4);
        load_vox(b, jX, vX);
        stage_a(vY, P0);
        if (k + 1 < n && P1.pr.ok && !body(P1)) redo |= 2ull << k;
    }
    for (; redo; redo &= redo - 1) {
        Vox v; load_vox(b, row_of_obs(__builtin_ctzll(redo)), v);
        slow(v, project(v.xs, fp, cam));
    }
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

// One band row of k_derive: FD gradient (Optimizer.cpp:287-364), optional updateGrad (OptimizerAux.cpp:152-160), the packed per-voxel record the
// sweeps read (surface point, the two normalised gradients, albedo) and the row's Eikonal / Laplacian energy terms (Optimizer.cpp:86-119).
// Shared by k_derive (band.hip) and by the epilogue of the persistent distance solve (pcg.hip: the regrad behind the solve's last hand-off).
__device__ __forceinline__ void derive_row(const SweepArgs& a, int j, int update_grad, double& en, double& el) {
#pragma clang fp contract(off)
    const Band& b = a.b;
    float n[3], dir[3];
    fd_grad(b, j, a.grid.vs_inv, n, dir);
    float g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { b.gfd[k][j] = n[k]; if (update_grad) b.g[k][j] = n[k]; g[k] = update_grad ? n[k] : b.g[k][j]; }
    float gn[3]; normalized3(g, gn);
    float nn[3]; normalized3(n, nn);
    long long lin = b.lin[j];
    int nxy = a.grid.dim[0] * a.grid.dim[1];
    int kz, rest;
    if (a.grid.nvox < (1LL << 31)) { const unsigned l = (unsigned)lin; kz = (int)(l / (unsigned)nxy); rest = (int)(l - (unsigned)kz * (unsigned)nxy); }      // (a 64-bit division is ~150 instructions, a third of this kernel's)
    else { kz = (int)(lin / nxy); rest = (int)(lin - (long long)kz * nxy); }
    int jy = rest / a.grid.dim[0]; int ix = rest - jy * a.grid.dim[0];
    int idx[3] = {ix, jy, kz};
    float d = b.dist[j];
    float xs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float xv = a.grid.origin[k] + a.grid.vs * (float)(idx[k] + (k == 2 ? a.grid.koff : 0));   // VoxelGrid.h:38-40 (global voxel index: a slab's local planes start at koff)
        xs[k] = xv - d * gn[k];
    }
    b.vp[0][j] = make_float4(xs[0], xs[1], xs[2], b.rho[0][j]);
    b.vp[1][j] = make_float4(gn[0], gn[1], gn[2], b.rho[1][j]);
    b.vp[2][j] = make_float4(nn[0], nn[1], nn[2], b.rho[2][j]);
    float e = norm3(n) - 1; en = (double)(e * e);
    float l = laplacian(b, j, a.grid.vs_inv); el = (double)(l * l);
}

// ------------------------------------------------------------------------------------------
// voxel-major sweeps: one thread per band voxel, iterating the set bits of its visibility mask
// ------------------------------------------------------------------------------------------
#define FOR_EACH_VISIBLE_FRAME(b, j, F, f)                                                   \
    for (int _w = 0; _w < (b).KW; ++_w)                                                      \
        for (uint64_t _m = (b).vis[(size_t)_w * (b).Spad + (j)]; _m; _m &= _m - 1)            \
            if (int f = 64 * _w + __builtin_ctzll(_m); f < (F))

// The chain of PsOptimizerJa.cpp:78-100 -- image_grad(3x2) * pi_grad(2x3) * R^T * d(point) -- contracted from the RIGHT: with
//   pi_grad = [a; b],  a = (fx/z, 0, -fx x/z^2),  b = (0, fy/z, -fy y/z^2),   image_grad row of channel c = (gu_c, gv_c)
// the row of channel c is gu_c (a R^T) + gv_c (b R^T): U = a R^T and V = b R^T are channel-independent 3-vectors, so a derivative
// along a direction dx is gu_c (U . dx) + gv_c (V . dx).  The reference forms the 3x3 product per channel first (left to right,
// structural zeros included); same value up to the rounding of the regrouped products, 64 instead of 87 instructions per
// observation in the distance sweep (engine deviation 7).
struct PiRows { float p00, p02, p11, p12; };
__device__ __forceinline__ PiRows pi_rows(const Cam& cam, const Proj& pr) {
    const float z_inv = pr.z_inv, z_inv_sq = z_inv * z_inv;
    PiRows o; o.p00 = cam.fx * z_inv; o.p02 = -cam.fx * pr.p[0] * z_inv_sq; o.p11 = cam.fy * z_inv; o.p12 = -cam.fy * pr.p[1] * z_inv_sq;
    return o;
}
__device__ __forceinline__ void pi_rows_world(const PiRows& pi, const float* R, float* U, float* V) {      // U = a R^T, V = b R^T
#pragma unroll
    for (int k = 0; k < 3; ++k) { U[k] = pi.p00 * R[k * 3 + 0] + pi.p02 * R[k * 3 + 2]; V[k] = pi.p11 * R[k * 3 + 1] + pi.p12 * R[k * 3 + 2]; }
}
__device__ __forceinline__ int sym4(int a, int b) {   // index into the 10 upper-triangular entries
    if (a > b) { int t = a; a = b; b = t; }
    return a * 4 - (a * (a - 1)) / 2 + (b - a);
}

// Sophus SO3::exp(w).matrix() (quaternion exponential + Eigen toRotationMatrix): OptimizerAux.cpp:103,193
__device__ inline void so3_exp(const float* w, float* R) {
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
// updatePose (OptimizerAux.cpp:190-205): t <- t - xi[0:3], R <- R exp(-xi[3:6]), on frame f's record
__device__ inline void pose_update(FrameP* frames, int f, const float* xi) {
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

}  // namespace psg
