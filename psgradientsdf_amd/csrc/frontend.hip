// frontend.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Gradient-SDF photometric-stereo hot path:
// the state producers next to the hot path: frame fusion, FALS normals, depth-tracker reduction.  No CUDA compatibility layer, no other back end.
// Shared device helpers: device_common.h; the launchers are declared in engine.h.
#include "device_common.h"

namespace psg {

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
        float xv[3] = {g.origin[0] + g.vs * (float)i, g.origin[1] + g.vs * (float)j, g.origin[2] + g.vs * (float)(k + g.koff)};      // (a z-slab's local plane k is plane k + koff of the volume)
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
// NormalEstimator::cache (NormalEstimator.h:52-125), once per image size: per pixel the ray (x0, y0, 1) / (1 + x0^2 + y0^2), the (2r+1)^2 box sums of the six
// products M and the inverse of the 3x3 matrix they form -- in double, operation for operation and in the summation order of the host code this replaces
// (round 4: 0.15-0.5 s of one core at 1139 x 1709, more than the fusion of four such frames); tests/test_frontend.py pins it to a numpy restatement bit for bit.
// work: 12 planes of n doubles (a | tmp); out: the 9 float planes of the cache
__global__ void __launch_bounds__(kBlock) k_ncache_rays(int W, int H, double fx_inv, double fy_inv, double cx, double cy, double* __restrict__ a, float* __restrict__ out) {
#pragma clang fp contract(off)
    const size_t n = (size_t)W * H;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        const double x0 = fx_inv * ((double)x - cx), y0 = fy_inv * ((double)y - cy), nsi = 1. / (1. + x0 * x0 + y0 * y0);
        a[p] = x0 * x0 * nsi; a[n + p] = x0 * y0 * nsi; a[2 * n + p] = x0 * nsi; a[3 * n + p] = y0 * y0 * nsi; a[4 * n + p] = y0 * nsi; a[5 * n + p] = nsi;
        out[p] = (float)(x0 * nsi); out[n + p] = (float)(y0 * nsi); out[2 * n + p] = (float)nsi;
    }
}
template <bool VERTICAL>
__global__ void __launch_bounds__(kBlock) k_ncache_box(const double* __restrict__ src, double* __restrict__ dst, int W, int H, int r) {
#pragma clang fp contract(off)
    const size_t n = (size_t)W * H;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        double s[6] = {0, 0, 0, 0, 0, 0};
        for (int k = -r; k <= r; ++k) {
            const size_t q = VERTICAL ? (size_t)reflect101(y + k, H) * W + x : (size_t)y * W + reflect101(x + k, W);
#pragma unroll
            for (int i = 0; i < 6; ++i) s[i] += src[i * n + q];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) dst[i * n + p] = s[i];
    }
}
__global__ void __launch_bounds__(kBlock) k_ncache_inverse(const double* __restrict__ M, size_t n, float* __restrict__ out) {
#pragma clang fp contract(off)
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const double M11 = M[p], M12 = M[n + p], M13 = M[2 * n + p], M22 = M[3 * n + p], M23 = M[4 * n + p], M33 = M[5 * n + p];
        const double det = M11 * (M22 * M33) + 2 * M12 * (M23 * M13) - (M13 * (M13 * M22) + M12 * (M12 * M33) + M23 * (M23 * M11));
        const double di = 1. / det;
        out[3 * n + p] = (float)(di * (M22 * M33 - M23 * M23)); out[4 * n + p] = (float)(di * (M13 * M23 - M12 * M33));
        out[5 * n + p] = (float)(di * (M12 * M23 - M13 * M22)); out[6 * n + p] = (float)(di * (M11 * M33 - M13 * M13));
        out[7 * n + p] = (float)(di * (M12 * M13 - M11 * M23)); out[8 * n + p] = (float)(di * (M11 * M22 - M12 * M12));
    }
}
void launch_normals_cache(int W, int H, int r, double fx_inv, double fy_inv, double cx, double cy, double* work, float* out, hipStream_t s) {
    const size_t n = (size_t)W * H;
    const int grid = (int)min((n + kBlock - 1) / kBlock, (size_t)4096);
    double* a = work; double* t = work + 6 * n;
    hipLaunchKernelGGL(k_ncache_rays, dim3(grid), dim3(kBlock), 0, s, W, H, fx_inv, fy_inv, cx, cy, a, out);
    hipLaunchKernelGGL(k_ncache_box<false>, dim3(grid), dim3(kBlock), 0, s, (const double*)a, t, W, H, r);
    hipLaunchKernelGGL(k_ncache_box<true>, dim3(grid), dim3(kBlock), 0, s, (const double*)t, a, W, H, r);
    hipLaunchKernelGGL(k_ncache_inverse, dim3(grid), dim3(kBlock), 0, s, (const double*)a, n, out);
}
// one Gauss-Newton pass of the tracker: H (21) | g (6) | E | count per workgroup -> partial rows part[blockIdx][29]
// gdimz, zown0, zown1: z-slabs -- the volume's plane count and the planes THIS context owns; a pixel is counted by the rank that owns the plane of its
// nearest voxel (the ranks' sums are all-reduced by the host), the bounds test is the whole volume's
__global__ void __launch_bounds__(kBlock) k_track(DenseView d, GridP g, Cam cam, FrameP fp, const float* __restrict__ depth, float z_min, float z_max, double* __restrict__ part, int gdimz, int zown0, int zown1) {
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
        if (fi[0] <= 0 || fi[1] <= 0 || fi[2] <= 0 || fi[0] >= (g.dim[0] - 1) || fi[1] >= (g.dim[1] - 1) || fi[2] >= (gdimz - 1)) continue;
        const int im = (int)(fi[0] + 0.5), jm = (int)(fi[1] + 0.5), km = (int)(fi[2] + 0.5);
        if (km < zown0 || km >= zown1) continue;
        const long long I = (long long)im + (long long)jm * g.dim[0] + (long long)(km - g.koff) * g.dim[0] * g.dim[1];
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
void launch_track(const DenseView& d, const GridP& g, const Cam& cam, const FrameP& fp, const float* depth, float z_min, float z_max, double* part, int nblk, int gdimz, int zown0, int zown1, hipStream_t s) {
    hipLaunchKernelGGL(k_track, dim3(nblk), dim3(kBlock), 0, s, d, g, cam, fp, depth, z_min, z_max, part, gdimz, zown0, zown1);
}
__global__ void k_fill_f32(float* p, float v, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_f32(float* p, float v, long long n, hipStream_t s) {
    int grid = (int)min((n + kBlock - 1) / kBlock, (long long)256 * 16);
    if (n > 0) hipLaunchKernelGGL(k_fill_f32, dim3(grid), dim3(kBlock), 0, s, p, v, n);
}

// 8-bit keyframes: interleaved RGB bytes -> one word per pixel (psgsdf_set_keyframes_u8)
__global__ void __launch_bounds__(kBlock) k_pack_rgb8(const uint8_t* __restrict__ rgb, unsigned* __restrict__ rgba, size_t npix) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x)
        rgba[i] = (unsigned)rgb[3 * i] | ((unsigned)rgb[3 * i + 1] << 8) | ((unsigned)rgb[3 * i + 2] << 16);
}
// float keyframes that ARE 8-bit data -- every channel equal to RN((float)b * scale) for a byte b, which is what the reference's loader produces from a
// PNG (ImageLoader.h:181 convertTo(CV_32FC3, 1.0f / 255.0f)) -- can be kept as RGBA8 words: the sampler's (float)b * scale gives back the same floats.
// *fail becomes non-zero if any channel is not such a value (the words are then not used).
// One flag for the whole image stack, and no same-address traffic on it: round 4 did one atomicOr per wavefront that saw a non-8-bit channel (240 k
// serialised atomics: 2.7 ms for 50 x 640 x 480 float keyframes -- DESIGN.md 4 "No same-address atomics"); polling the flag once per pixel instead (round
// 5's first version) hammered its memory channel just the same (143-154 us on rendered floats, more than the 50 us a full pass takes).  Now in two steps:
// a SAMPLE of the stack (every `stride`-th pixel, nothing written) decides for data that is not 8-bit at all -- rendered or filtered images fail at the
// first non-background pixel: ~9 us; (one flag word per WORKGROUP: 3 900 wavefronts storing the same 1 to ONE word still took 140 us) --, and only a sample without a miss is followed by the full pass, which streams the stack once (184 MB in, 61 MB
// out at 50 x 640 x 480: 50 us = 4.9 TB/s) and never reads the flag; a wavefront that still finds a miss stores a plain 1 and leaves.
__global__ void __launch_bounds__(kBlock) k_try_pack_f32(const float* __restrict__ rgb, unsigned* __restrict__ rgba, size_t npix, size_t stride, float scale, float inv_scale, int* __restrict__ fail) {
    const size_t n = (npix + stride - 1) / stride;
    for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
        const size_t i = q * stride;
        unsigned w = 0; bool bad = false;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float v = rgb[3 * i + ch];
            const float bf = rintf(v * inv_scale);
            const bool ok = bf >= 0.f && bf <= 255.f && __fmul_rn(bf, scale) == v;
            bad = bad || !ok;
            w |= (ok ? (unsigned)bf : 0u) << (8 * ch);
        }
        if (rgba) rgba[i] = w;
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {      // one store per wavefront that found a miss, into ITS WORKGROUP'S flag (kTryPackFlags distinct words: the host ORs them), then it leaves
            if (bad && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(bad))) fail[blockIdx.x] = 1;
            return;
        }
    }
}
// step 1: the sample (rgba = nullptr: nothing is written); step 2 (only if the sample had no miss): the full pass
void launch_try_pack_f32(const float* rgb, unsigned* rgba, size_t npix, size_t stride, float scale, int* fail, hipStream_t s) {
    if (!npix) return;
    const size_t n = (npix + stride - 1) / stride;
    hipLaunchKernelGGL(k_try_pack_f32, dim3((unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, (size_t)kTryPackFlags)), dim3(kBlock), 0, s, rgb, rgba, npix, stride, scale, 1.0f / scale, fail);
}
void launch_pack_rgb8(const uint8_t* rgb, unsigned* rgba, size_t npix, hipStream_t s) {
    if (npix) hipLaunchKernelGGL(k_pack_rgb8, dim3((unsigned)std::min<size_t>((npix + kBlock - 1) / kBlock, 65535)), dim3(kBlock), 0, s, rgb, rgba, npix);
}

}  // namespace psg
