// ipc_memkind.hip -- which device-memory kinds can carry the cross-rank persistent solve's in-kernel hand-offs?
//
//   hipcc --offload-arch=gfx950 -O2 tools/ipc_memkind.hip -o gpurun_out/ipc_memkind && gpurun_out/ipc_memkind [rounds]
//
// For each kind -- hipMalloc (coarse-grained), hipExtMallocWithFlags(hipDeviceMallocFinegrained), (..Uncached) -- the parent (the OWNER)
//   1. allocates a region {flag words, a payload of 16-byte records}, exports it with hipIpcGetMemHandle, hands the handle to a child process
//      (a PEER: device 1 when the box has two devices, else the same device);
//   2. runs ONE kernel (one workgroup) that plays the owner's side of the hand-off for `rounds` rounds: read the payload (so that stale copies
//      sit in its caches), wait for the peer's tagged flag with system-scope relaxed loads, one system-scope acquire fence, read the payload with
//      PLAIN loads, count records that do not carry the round's value, answer through a flag in the peer's direction;
//   3. the peer's kernel per round: sc0 sc1 write-through 16-byte stores of the payload, s_waitcnt vmcnt(0), system-scope relaxed store of the
//      tagged flag, wait for the answer.
// These are exactly the store / load / fence forms of pcg.hip k_cgf_solve<.., MR>.  It also times a local streaming read of 64 MiB of each kind
// (is the owner's mapping of fine-grained / uncached memory cached?).  Prints one line per kind: ipc ok?, stale records, timeouts, us per round, GB/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#include <chrono>

#define CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "[%d] %s: %s\n", (int)getpid(), #e, hipGetErrorString(_e)); exit(2); } } while (0)

constexpr int kRecs = 4096;                       // 64 KiB of records per round
constexpr int kFlagOff = 0, kAnsOff = 8, kStat = 16;   // doubles: flag (peer -> owner), answer (owner -> peer), owner's statistics
constexpr size_t kRegionBytes = 4096 + (size_t)kRecs * 16;

typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_sys(float4* p, const float4& v) {
    const v4f_t d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
}

__global__ void __launch_bounds__(256) k_owner(double* region, int rounds, double* out) {
    float4* rec = (float4*)((char*)region + 4096);
    __shared__ int s_to; __shared__ long long s_stale;
    if (threadIdx.x == 0) { s_to = 0; s_stale = 0; }
    __syncthreads();
    long long stale = 0; float sink = 0.f;
    for (int r = 1; r <= rounds; ++r) {
        for (int i = threadIdx.x; i < kRecs; i += blockDim.x) sink += rec[i].x;     // pull the old payload into this device's caches
        __syncthreads();
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(region + kFlagOff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (double)r) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 24)) { s_to = 1; break; } }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
        __syncthreads();
        if (s_to) break;
        for (int i = threadIdx.x; i < kRecs; i += blockDim.x) { const float4 v = rec[i]; if (v.x != (float)r || v.w != (float)(r + i)) stale++; }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(region + kAnsOff, (double)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    atomicAdd((unsigned long long*)&s_stale, (unsigned long long)stale);
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = (double)s_stale; out[1] = (double)s_to; out[2] = (double)sink; }
}
__global__ void __launch_bounds__(256) k_peer(double* region, int rounds, double* out) {
    float4* rec = (float4*)((char*)region + 4096);
    __shared__ int s_to;
    if (threadIdx.x == 0) s_to = 0;
    __syncthreads();
    for (int r = 1; r <= rounds; ++r) {
        for (int i = threadIdx.x; i < kRecs; i += blockDim.x) store16_sys(rec + i, make_float4((float)r, 0.f, 0.f, (float)(r + i)));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(region + kFlagOff, (double)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            int spins = 0;
            while (__hip_atomic_load(region + kAnsOff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (double)r) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 24)) { s_to = 1; break; } }
        }
        __syncthreads();
        if (s_to) break;
    }
    if (threadIdx.x == 0) out[0] = (double)s_to;
}
__global__ void k_stream(const float4* p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x;
    if (acc == 1234.5f) *sink = acc;
}

static hipError_t alloc_kind(void** p, size_t bytes, int kind) {
    if (kind == 0) return hipMalloc(p, bytes);
    return hipExtMallocWithFlags(p, bytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    const char* names[3] = {"hipMalloc (coarse)", "hipDeviceMallocFinegrained", "hipDeviceMallocUncached"};
    // all three peers are forked BEFORE this process touches HIP (the runtime does not survive a fork); each waits on its pipe for its turn
    int to_child_all[3][2], to_parent_all[3][2]; pid_t pids[3];
    for (int kind = 0; kind < 3; ++kind) {
        if (pipe(to_child_all[kind]) || pipe(to_parent_all[kind])) return 2;
        fflush(stdout);
        pids[kind] = fork();
        if (pids[kind] == 0) {
            int* to_child = to_child_all[kind]; int* to_parent = to_parent_all[kind];
            close(to_child[1]); close(to_parent[0]);
            hipIpcMemHandle_t h; char okc = 0;
            if (read(to_child[0], &okc, 1) != 1 || !okc) _exit(0);
            if (read(to_child[0], &h, sizeof(h)) != (ssize_t)sizeof(h)) _exit(3);
            int nd = 0; CHK(hipGetDeviceCount(&nd));
            CHK(hipSetDevice(nd > 1 ? 1 : 0));
            void* reg = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&reg, h, hipIpcMemLazyEnablePeerAccess);
            char opened = e == hipSuccess;
            if (write(to_parent[1], &opened, 1) != 1) _exit(3);
            if (!opened) { fprintf(stderr, "  peer: hipIpcOpenMemHandle: %s\n", hipGetErrorString(e)); _exit(0); }
            double* out; CHK(hipMalloc(&out, 64)); CHK(hipMemset(out, 0, 64));
            hipLaunchKernelGGL(k_peer, dim3(1), dim3(256), 0, 0, (double*)reg, rounds, out);
            CHK(hipDeviceSynchronize());
            double o[1]; CHK(hipMemcpy(o, out, 8, hipMemcpyDeviceToHost));
            if (o[0] != 0.0) fprintf(stderr, "  peer: timed out waiting for the owner's answer\n");
            hipIpcCloseMemHandle(reg);
            _exit(0);
        }
    }
    int ndev = 0; CHK(hipGetDeviceCount(&ndev));
    printf("devices: %d, peer on device %d, %d rounds of %d 16-byte records\n", ndev, ndev > 1 ? 1 : 0, rounds, kRecs);
    for (int kind = 0; kind < 3; ++kind) {
        int* to_child = to_child_all[kind]; int* to_parent = to_parent_all[kind]; const pid_t pid = pids[kind];
        close(to_child[0]); close(to_parent[1]);
        CHK(hipSetDevice(0));
        void* reg = nullptr;
        hipError_t e = alloc_kind(&reg, kRegionBytes, kind);
        hipIpcMemHandle_t h{};
        char okc = 0;
        if (e == hipSuccess) { CHK(hipMemset(reg, 0, kRegionBytes)); CHK(hipDeviceSynchronize()); e = hipIpcGetMemHandle(&h, reg); okc = e == hipSuccess; }
        if (!okc) { printf("%-28s alloc/export FAILED: %s\n", names[kind], hipGetErrorString(e)); (void)hipGetLastError(); }
        if (write(to_child[1], &okc, 1) != 1) return 2;
        double stale = -1, to = -1, us = 0;
        if (okc) {
            if (write(to_child[1], &h, sizeof(h)) != (ssize_t)sizeof(h)) return 2;
            char opened = 0;
            if (read(to_parent[0], &opened, 1) != 1) opened = 0;
            if (opened) {
                double* out; CHK(hipMalloc(&out, 64)); CHK(hipMemset(out, 0, 64));
                auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(k_owner, dim3(1), dim3(256), 0, 0, (double*)reg, rounds, out);
                CHK(hipDeviceSynchronize());
                us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
                double o[3]; CHK(hipMemcpy(o, out, 24, hipMemcpyDeviceToHost));
                stale = o[0]; to = o[1];
                hipFree(out);
            } else printf("%-28s peer could not open the handle\n", names[kind]);
        }
        int status = 0; waitpid(pid, &status, 0);
        close(to_child[1]); close(to_parent[0]);
        // local streaming read of 64 MiB of this kind, second pass timed (is the owner's own mapping cached? L2 + Infinity Cache hold most of it)
        double gbs = 0;
        {
            void* big = nullptr; const size_t nb = (size_t)64 << 20;
            if (alloc_kind(&big, nb, kind) == hipSuccess) {
                float* sink; CHK(hipMalloc(&sink, 64));
                CHK(hipMemset(big, 0, nb));
                hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
                for (int rep = 0; rep < 3; ++rep) {
                    CHK(hipEventRecord(a, 0));
                    hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const float4*)big, nb / 16, sink);
                    CHK(hipEventRecord(b, 0)); CHK(hipEventSynchronize(b));
                    float ms; CHK(hipEventElapsedTime(&ms, a, b)); gbs = nb / (ms * 1e-3) / 1e9;
                }
                hipFree(big); hipFree(sink);
            } else (void)hipGetLastError();
        }
        if (okc && stale >= 0) printf("%-28s ipc ok   stale records %.0f / %lld   owner timed out %.0f   %.2f us per round   local 64 MiB stream %.0f GB/s\n",
                                      names[kind], stale, (long long)rounds * kRecs, to, us, gbs);
        if (reg) hipFree(reg);
    }
    return 0;
}
