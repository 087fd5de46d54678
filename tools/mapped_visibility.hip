// mapped_visibility.hip -- does a host that polls a MARKER written by a later kernel see the host-mapped (pinned, hipHostMallocMapped) words
// an EARLIER kernel of the same stream wrote with plain stores?  (The engine's deferred read-backs relied on exactly that: engine.hip flush().)
//
//   hipcc --offload-arch=gfx950 -O2 tools/mapped_visibility.hip -o gpurun_out/mapped_visibility && gpurun_out/mapped_visibility [iterations]
//
// Per round: kernel W (8 workgroups, one per XCD under the observed round-robin; workgroup b streams some memory, then thread 0 stores
// slot[b] = seq), then kernel M (one thread: *marker = seq; __threadfence_system()).  The host spins on the marker, then reads the eight slots
// at once and counts those that do not hold seq yet ("stale"), then drains the stream and checks again.  Variants: W's writer fences
// (system scope) after its store; W is a single workgroup; a memory-heavy kernel runs in between.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#define CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(2); } } while (0)

__global__ void k_write(double* slot, int* xcc, double seq, const float* stream, size_t n, float* sink, int fence) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += stream[i];
    if (acc == 12345.678f) sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        slot[blockIdx.x] = seq;
        if (fence) __threadfence_system();
        xcc[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID
    }
}
__global__ void k_marker(double* p, double v) { *p = v; __threadfence_system(); }
__global__ void k_load(float* buf, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] += 1.0f; }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipStream_t s; CHK(hipStreamCreate(&s));
    double* mb; CHK(hipHostMalloc(&mb, 4096, hipHostMallocMapped));
    double* mb_dev; CHK(hipHostGetDevicePointer((void**)&mb_dev, mb, 0));
    memset(mb, 0, 4096);
    int* xcc; CHK(hipHostMalloc(&xcc, 64 * sizeof(int), hipHostMallocMapped));
    int* xcc_dev; CHK(hipHostGetDevicePointer((void**)&xcc_dev, xcc, 0));
    const size_t n = 4u << 20;
    float *buf, *sink, *big; CHK(hipMalloc(&buf, n * 4)); CHK(hipMalloc(&sink, 64)); CHK(hipMalloc(&big, (size_t)64 << 20));
    CHK(hipMemset(buf, 0, n * 4)); CHK(hipMemset(big, 0, (size_t)64 << 20));
    volatile double* marker = mb + 64;
    struct Variant { const char* name; int blocks; int fence; int load; size_t stream_n; } variants[] = {
        {"8 workgroups, plain stores", 8, 0, 0, n},
        {"8 workgroups, plain stores, heavy kernel in between", 8, 0, 1, n},
        {"8 workgroups, writer fences (system)", 8, 1, 0, n},
        {"1 workgroup, plain store", 1, 0, 0, n / 8},
        {"1 workgroup, plain store, no streaming", 1, 0, 0, 0},
        {"8 workgroups, plain stores, no streaming", 8, 0, 0, 0},
    };
    double seq = 0;
    for (const Variant& v : variants) {
        long long stale[8] = {0}, stale_after_sync = 0, rounds_with_stale = 0; int xcc_seen[8] = {0};
        auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; ++it) {
            seq += 1.0;
            hipLaunchKernelGGL(k_write, dim3(v.blocks), dim3(256), 0, s, mb_dev, xcc_dev, seq, (const float*)buf, v.stream_n, sink, v.fence);
            if (v.load) hipLaunchKernelGGL(k_load, dim3(2048), dim3(256), 0, s, big, ((size_t)64 << 20) / 4);
            hipLaunchKernelGGL(k_marker, dim3(1), dim3(1), 0, s, mb_dev + 64, seq);
            while (*marker != seq) {}
            bool any = false;
            double snap[8];
            for (int b = 0; b < v.blocks; ++b) snap[b] = ((volatile double*)mb)[b];
            for (int b = 0; b < v.blocks; ++b) if (snap[b] != seq) { stale[b]++; any = true; }
            rounds_with_stale += any;
            CHK(hipStreamSynchronize(s));
            for (int b = 0; b < v.blocks; ++b) { if (((volatile double*)mb)[b] != seq) stale_after_sync++; xcc_seen[b] |= 1 << xcc[b]; }
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        printf("{\"variant\": \"%s\", \"rounds\": %d, \"rounds_with_a_stale_slot\": %lld, \"stale_per_workgroup\": [", v.name, iters, rounds_with_stale);
        for (int b = 0; b < v.blocks; ++b) printf("%lld%s", stale[b], b + 1 < v.blocks ? ", " : "");
        printf("], \"xcc_mask_per_workgroup\": [");
        for (int b = 0; b < v.blocks; ++b) printf("%d%s", xcc_seen[b], b + 1 < v.blocks ? ", " : "");
        printf("], \"stale_after_stream_sync\": %lld, \"us_per_round\": %.1f}\n", stale_after_sync, us);
        fflush(stdout);
    }
    return 0;
}
