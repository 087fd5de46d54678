// sweeps.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Gradient-SDF photometric-stereo hot path:
// the photometric sweeps (initial albedo, PS energy, albedo / light / pose normal equations) and the small per-frame solves.  No CUDA compatibility layer, no other back end.
// Shared device helpers: device_common.h; the launchers are declared in engine.h.
#include "device_common.h"

#ifndef PSG_FM_PIPE
#define PSG_FM_PIPE 3      // (development: bit 0 = k_sweep_light, bit 1 = k_sweep_pose run the three-stage observation pipeline of device_common.h fm_for_each_obs)
#endif
#ifndef PSG_POSE_WAVES
#define PSG_POSE_WAVES 1
#endif

namespace psg {

// Optimizer.cpp:50-81 initAlbedo
__global__ void __launch_bounds__(kBlock) k_init_albedo(SweepArgs a) {
#pragma clang fp contract(off)
    FrameP* sf = reinterpret_cast<FrameP*>(psg_dyn_smem);   // F records, dynamic LDS
    load_frames(sf, a.frames, a.F);
    const Band& b = a.b;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.row1) return;
    const float4 p0 = b.vp[0][j];
    float xs[3] = {p0.x, p0.y, p0.z};
    int count = 0; float rho[3] = {0, 0, 0};
    FOR_EACH_VISIBLE_FRAME(b, j, a.F, f) {
        Proj pr = project(xs, frame_at(sf, f), a.cam);
        if (!pr.ok) continue;
        float I[3];
        sample<false>(a.im, f, a.cam, pr.m, pr.n, I, nullptr, nullptr);
        rho[0] += I[0]; rho[1] += I[1]; rho[2] += I[2]; count++;
    }
    if (count) {
#pragma unroll
        for (int k = 0; k < 3; ++k) set_rho(b, j, k, rho[k] / (float)count);
    }
}
void launch_init_albedo(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_init_albedo, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), a.F * sizeof(FrameP), s, a);
}

// getPSEnergy PsOptimizer.cpp:47-78 / LedOptimizer.cpp:40-71; LED_INIT: computeLightIntensive
// LedOptimizer.cpp:76-112 (sums of observed and rendered intensity)
template <int MODEL, int LOSS, int IMG, bool LED_INIT = false>
__global__ void __launch_bounds__(kBlock) k_energy(SweepArgs a) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    constexpr int NB = ModelTraits<MODEL>::NB;
    FrameP* sf = reinterpret_cast<FrameP*>(psg_dyn_smem);   // F records, dynamic LDS
    __shared__ double red[kBlock / 64];
    load_frames(sf, a.frames, a.F);
    const Band& b = a.b;
    const int bid = vm_bid(a, 2, true);
    int j = a.row0 + bid * blockDim.x + threadIdx.x;
    double E = 0, nobs = 0, sI[3] = {0, 0, 0}, sR[3] = {0, 0, 0};
    obs_acc_t Ef = 0;
    if (j < a.row1) {
        Vox v; load_vox(b, j, v);
        float shfd[kMaxBasis];
        if (!ModelTraits<MODEL>::LED) SH<NB == 3 ? 4 : NB>(v.nfd, shfd);
        FOR_EACH_VISIBLE_FRAME(b, j, a.F, f) {
            const FrameP& fp = frame_at(sf, f);
            Proj pr = project(v.xs, fp, a.cam);
            if (!pr.ok) continue;
            float I[3], ren[3];
            sample<false, IMG>(a.im, f, a.cam, pr.m, pr.n, I, nullptr, nullptr);
            rendered<MODEL>(fp, pr, v.nfd, shfd, v.rho, ren);
            if (LED_INIT) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) { sI[ch] += (double)I[ch]; sR[ch] += (double)ren[ch]; }
            } else {
                float l = 0.f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) l += robust_loss<LOSS>(a.rob, I[ch] - ren[ch]);
                Ef += (obs_acc_t)l; nobs += 1.0;
            }
        }
    }
    E = (double)Ef;
    if (LED_INIT) {
        // 6 sums: observed rgb into SC_AUX0.., rendered into SC_EN/SC_EL/SC_ACCEPT slots (scratch use at init only)
        block_part_store(sI[0], PART(a, SC_AUX0), red, bid); block_part_store(sI[1], PART(a, SC_AUX1), red, bid); block_part_store(sI[2], PART(a, SC_AUX2), red, bid);
        block_part_store(sR[0], PART(a, SC_EN), red, bid); block_part_store(sR[1], PART(a, SC_EL), red, bid); block_part_store(sR[2], PART(a, SC_ACCEPT), red, bid);
    } else {
        block_part_store(E, PART(a, SC_ENERGY), red, bid);
        block_part_store(nobs, PART(a, SC_NOBS), red, bid);
    }
}
void launch_energy(const SweepArgs& a, hipStream_t s) {
    if (a.row1 <= a.row0) return;
    dim3 g((a.row1 - a.row0 + kBlock - 1) / kBlock), bl(kBlock);
    PSG_LAUNCH_SWEEP(k_energy, a, false, g, bl, a.F * sizeof(FrameP), s, a);
}
void launch_led_light_init(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL((k_energy<2, -1, -1, true>), dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), a.F * sizeof(FrameP), s, a);
}

// albedo normal equations (diagonal): optimizeAlbedoAll PsOptimizer.cpp:85-121 / LedOptimizer.cpp:162-196,
// albedoJacobian PsOptimizerJa.cpp:375-422, computeResidual :567-626.  Also yields the PS energy of the input state.
template <int MODEL, int LOSS, int IMG>
__global__ void __launch_bounds__(kBlock) k_sweep_albedo(SweepArgs a) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    constexpr int NB = ModelTraits<MODEL>::NB;
    FrameP* sf = reinterpret_cast<FrameP*>(psg_dyn_smem);   // F records, dynamic LDS
    __shared__ double red[kBlock / 64];
    load_frames(sf, a.frames, a.F);
    const Band& b = a.b;
    const int bid = vm_bid(a, 2, true);
    int j = a.row0 + bid * blockDim.x + threadIdx.x;
    double E = 0, nobs = 0, cnt = 0;
    if (j < a.row1) {
        Vox v; load_vox(b, j, v);
        float shfd[kMaxBasis], shg[kMaxBasis];
        if (!ModelTraits<MODEL>::LED) { SH<NB == 3 ? 4 : NB>(v.nfd, shfd); SH<NB == 3 ? 4 : NB>(v.gn, shg); }
        obs_acc_t Hd[3] = {0, 0, 0}, bd[3] = {0, 0, 0};
        obs_acc_t Ef = 0; int nobs_i = 0;
        FOR_EACH_VISIBLE_FRAME(b, j, a.F, f) {
            const FrameP& fp = frame_at(sf, f);
            Proj pr = project(v.xs, fp, a.cam);
            if (!pr.ok) continue;
            float I[3], ren[3], J[3];
            sample<false, IMG>(a.im, f, a.cam, pr.m, pr.n, I, nullptr, nullptr);
            rendered<MODEL>(fp, pr, v.nfd, shfd, v.rho, ren);
            rho_jac<MODEL>(fp, pr, v.gn, shg, J);
            float l = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float r = I[ch] - ren[ch]; float w = robust_weight<LOSS>(a.rob, r);
                float jw = J[ch] * w;
                Hd[ch] += (obs_acc_t)(jw * J[ch]); bd[ch] += (obs_acc_t)(jw * r);
                l += robust_loss<LOSS>(a.rob, r);
            }
            Ef += (obs_acc_t)l; nobs_i += 1;
        }
        E = (double)Ef; nobs = (double)nobs_i;
        if (a.fuse_apply) {   // the system is diagonal and the voxel's albedo is read by this thread only: k_apply_albedo's arithmetic, here
            // fuse_apply == 2: the update is SPECULATIVE (the host has not taken the stop decision this sweep's energy feeds yet, loop.hip): the old
            // albedo goes to the (otherwise unused) aH planes, from where k_restore_albedo puts it back if the loop ends here
            if (a.fuse_apply == 2) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) b.aH[(size_t)ch * b.Spad + j] = v.rho[ch];
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float h = (float)Hd[ch];
                if (a.damping != 0.0f) h += a.damping * h;
                const float delta = (h != 0.f) ? (float)bd[ch] / h : 0.f;
                const float nv = v.rho[ch] - delta;
                if (nv > 0.0f && nv < 1.0f) { set_rho(b, j, ch, nv); cnt += 1.0; }
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { b.aH[(size_t)ch * b.Spad + j] = (float)Hd[ch]; b.ab[(size_t)ch * b.Spad + j] = (float)bd[ch]; }
        }
    }
    block_part_store(E, PART(a, SC_ENERGY), red, bid);
    block_part_store(nobs, PART(a, SC_NOBS), red, bid);
    if (a.fuse_apply) block_part_store(cnt, PART(a, SC_ACCEPT), red, bid);
}
void launch_sweep_albedo(const SweepArgs& a, hipStream_t s) {
    if (a.row1 <= a.row0) return;
    dim3 g((a.row1 - a.row0 + kBlock - 1) / kBlock), bl(kBlock);
    PSG_LAUNCH_SWEEP(k_sweep_albedo, a, false, g, bl, a.F * sizeof(FrameP), s, a);
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
            if (v > 0.0f && v < 1.0f) { set_rho(b, j, ch, v); cnt += 1.0; }
        }
    }
    block_part_store(cnt, PART(a, SC_ACCEPT), red);
}
// undo of a speculative fused albedo update (k_sweep_albedo, fuse_apply == 2): the saved albedo back into rho and its packed mirror
__global__ void __launch_bounds__(kBlock) k_restore_albedo(SweepArgs a) {
    const Band& b = a.b;
    int j = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.row1) return;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) set_rho(b, j, ch, b.aH[(size_t)ch * b.Spad + j]);
}
void launch_restore_albedo(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_restore_albedo, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}
void launch_apply_albedo(const SweepArgs& a, hipStream_t s) {
    if (a.row1 > a.row0) hipLaunchKernelGGL(k_apply_albedo, dim3((a.row1 - a.row0 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}

// ------------------------------------------------------------------------------------------
// frame-major sweeps: grid = (row chunks, F); per-thread accumulation over several voxels of ONE
// frame, then wavefront DPP reduction -> LDS -> the workgroup's partial row (plain write-through stores, NO floating-point
// atomics); the frame's last workgroup to arrive sums the partial rows in launch order (frame_rows_publish below)
// ------------------------------------------------------------------------------------------
// Observations per thread: a launch parameter.  A workgroup's life is `rows` sequential observations per lane, and these kernels
// are VALU-bound per SIMD, so the grid should fill the resident workgroup slots of the chip evenly in ONE generation; a fixed 16
// left 13 % of the pose sweep's workgroups for a second, nearly empty generation (fm_rows below).
static int fm_rows(const SweepArgs& a, int slots_per_cu) {
    if (const char* e = getenv("PSGSDF_FM_ROWS")) { int v = atoi(e); if (v >= 4 && v <= 64) return v; }   // tuning knob (>= 4: the partial-row buffer is sized for that; <= 64: fm_for_each_obs keeps one bit per observation of a thread)
    const long long total = a.b.obs_ptr_total > 0 ? a.b.obs_ptr_total : (long long)a.b.obs_max * a.F;
    const long long slots = 256LL * slots_per_cu;                       // resident workgroups on the chip
    // every frame wastes half a chunk on average: aim at ~92 % of the slots
    long long r = (total + (long long)(0.92 * slots) * kBlock - 1) / ((long long)(0.92 * slots) * kBlock);
    return (int)std::min<long long>(std::max<long long>(r, 4), 64);
}

// small dense solves and SO3::exp (defined below)
template <int N> __device__ void solve_spd(const double* Hin, const double* bin, double* x);

// One frame's light step (optimizeLightAll, PsOptimizer.cpp:175-203: no damping; SH models: NB x NB per frame) from its final row `acc`, and one
// frame's pose step (optimizePosesAll PsOptimizer.cpp:207-234 + updatePose OptimizerAux.cpp:190-205).  Shared by the one-workgroup solve kernels
// below and by the sweeps' own epilogue (fm_solve: the last workgroup of a frame to arrive solves that frame at once -- no solve launch).
// (a row element: LDS in a frame's own last workgroup, memory -- written by other threads of the workgroup a barrier ago -- in the multi-rank tail)
__device__ __forceinline__ double ldrow(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int NB>
__device__ __forceinline__ void frame_solve_light_sh(FrameP* frames, int f, const double* acc, float* undo) {
    constexpr int NH = NB * (NB + 1) / 2;
    if (undo) for (int i = 0; i < 9; ++i) undo[f * 9 + i] = frames[f].l[i];      // (a speculative update keeps what it overwrites: loop.hip run_loop)
    double Hd[NB * NB], bd[NB], xd[NB];
    int q = 0;
    for (int i = 0; i < NB; ++i) for (int k = i; k < NB; ++k) { double v = (double)(float)ldrow(acc + q++); Hd[i * NB + k] = v; Hd[k * NB + i] = v; }
    for (int i = 0; i < NB; ++i) bd[i] = (double)(float)ldrow(acc + NH + i);
    solve_spd<NB>(Hd, bd, xd);
    for (int i = 0; i < NB; ++i) frames[f].l[i] -= (float)xd[i];
}
__device__ __forceinline__ void frame_solve_pose(const SweepArgs& a, FrameP* frames, int f, const double* acc) {
    double Hd[36], bd[6], xd[6];
    int q = 0;
    for (int i = 0; i < 6; ++i) for (int k = i; k < 6; ++k) {
        float v = (float)ldrow(acc + q++);
        if (i == k && a.damping != 0.0f) v += a.damping * v;
        Hd[i * 6 + k] = (double)v; Hd[k * 6 + i] = (double)v;
    }
    for (int i = 0; i < 6; ++i) bd[i] = (double)(float)ldrow(acc + 21 + i);
    solve_spd<6>(Hd, bd, xd);
    float xi[6];
    for (int i = 0; i < 6; ++i) xi[i] = (float)xd[i];
    pose_update(frames, f, xi);
}

// What the LAST workgroup of a frame-major launch does once every frame's final row is in a.acc.frame: (multi-rank: the per-frame solves, one lane per
// frame --) the LED light vector over all frames, and the energy / n_obs sums over the frames into the mailbox.  Shared by the sweeps' own epilogue
// (single rank: the frames were solved by their own last workgroups) and by k_frame_gather (multi-rank).
template <int NV, int KIND, int MODEL>
__device__ __forceinline__ void frame_tail(const SweepArgs& a, double* lds, bool solve_all) {
    if (solve_all) {
        for (int ff = threadIdx.x; ff < a.F; ff += blockDim.x) {
            const double* row = a.acc.frame + (size_t)ff * kFrameRow;
            if (KIND == 0) { if (!ModelTraits<MODEL>::LED) frame_solve_light_sh<ModelTraits<MODEL>::NB == 3 ? 4 : ModelTraits<MODEL>::NB>(a.fm_frames, ff, row, a.fm_undo); }
            else frame_solve_pose(a, a.fm_frames, ff, row);
        }
        __syncthreads();
    }
    if (KIND == 0 && ModelTraits<MODEL>::LED) {
        // LED: ONE light vector over all frames (LedOptimizer.cpp:128-160): this workgroup is the last of the whole sweep, every frame's final row is
        // in place and no workgroup reads a frame record any more -- k_solve_light's arithmetic (sums over the frames in frame order, three
        // scalar equations, every frame's copy updated), thread 0 solving, all threads updating
        constexpr int NHL = 3;
        __shared__ float s_dl[3];
        if (a.fm_undo) {
            for (int i = threadIdx.x; i < a.F * 9; i += blockDim.x) a.fm_undo[i] = a.fm_frames[i / 9].l[i % 9];
            if (threadIdx.x < 3) a.fm_undo[a.F * 9 + threadIdx.x] = a.fm_led_light[threadIdx.x];
        }
        __shared__ double srow[6 * kMaxFramesLds];      // the six columns staged by all threads at once (summed straight from memory: 2 x F dependent loads per thread)
        for (int i = threadIdx.x; i < 6 * a.F; i += blockDim.x) { const int ff = i / 6, c6 = i % 6; srow[c6 * kMaxFramesLds + ff] = __hip_atomic_load(a.acc.frame + (size_t)ff * kFrameRow + (c6 < 3 ? c6 : NHL + c6 - 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int ch = threadIdx.x;
            double hs = 0, bs = 0;
            for (int ff = 0; ff < a.F; ++ff) { hs += srow[ch * kMaxFramesLds + ff]; bs += srow[(3 + ch) * kMaxFramesLds + ff]; }
            float h = (float)hs; const float bb = (float)bs;
            if (a.damping != 0.0f) h += a.damping * h;
            double Hd[1] = {(double)h}, bd[1] = {(double)bb}, xd[1];
            solve_spd<1>(Hd, bd, xd);
            s_dl[ch] = (float)xd[0];
        }
        __syncthreads();
        for (int ff = threadIdx.x; ff < a.F; ff += blockDim.x) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { const float nl = a.fm_frames[ff].l[ch] - s_dl[ch]; a.fm_frames[ff].l[ch] = nl; if (ff == 0) a.fm_led_light[ch] = nl; }
        }
    }
    if (a.fm_e_out) {      // energy / n_obs over the frames, in frame_rows_finish's order (one 256-thread workgroup, threads striding the frames)
        constexpr int col_e = NV - 2;
        double e = 0, n = 0;
        for (int ff = threadIdx.x; ff < a.F; ff += blockDim.x) {
            e += __hip_atomic_load(a.acc.frame + (size_t)ff * kFrameRow + col_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            n += __hip_atomic_load(a.acc.frame + (size_t)ff * kFrameRow + col_e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        e = wave_sum(e); n = wave_sum(n);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) { lds[2 * w] = e; lds[2 * w + 1] = n; }
        __syncthreads();
        if (threadIdx.x == 0) { double te = 0, tn = 0; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { te += lds[2 * i]; tn += lds[2 * i + 1]; } mbox_put(a.fm_e_out, 2, 0, te, a.fm_e_key); mbox_put(a.fm_e_out, 2, 1, tn, a.fm_e_key); mbox_commit(a.fm_e_key); }
    }
}

// Epilogue of the frame-major sweeps: every workgroup stores ITS partial row of frame f (plain stores, no floating-point atomics) and
// takes a ticket on the frame's arrival counter; the LAST workgroup of the frame to arrive sums the frame's partial rows in launch order
// into the final row -- reproducible from run to run, and done while the other frames' workgroups are still sweeping.
// fm_solve (single rank, the deferred path of the alternation loop): that last workgroup also SOLVES the frame -- KIND 0: the SH light block
// (the LED light is one vector over all frames: k_solve_light keeps it), KIND 1: the pose block -- and takes a second ticket on the sweep's
// frame counter; the last FRAME to finish sums the energy / n_obs columns of all rows in frame order into the mailbox (what the solve kernels'
// frame_rows_finish does).  The frame's record is only read by the frame's own workgroups, all of which have finished; same arithmetic, same bits.
template <int NV, int KIND, int MODEL>
__device__ __forceinline__ void frame_rows_publish(const SweepArgs& a, int cx, int f, double* lds /*[kBlock/64][NV] wavefront sums*/) {
    // Hand-off without fences (an agent-scope release fence in every workgroup's tail writes back the XCD's L2 each time: light sweep
    // 46 -> 82 us): the row goes out with write-through (sc1) stores, the wave drains them, then takes the ticket; the last arriver reads
    // the rows with sc1 loads (MI355X_MICROARCH.md: "sc1 payload -> vmcnt(0) -> sc1 flag", "sc1 loads may replace the acquire").
    __shared__ int s_last;
    double* dst = a.acc.fpart + ((size_t)f * a.acc.fcap + cx) * kFrameRow;
    if (threadIdx.x < NV) {                      // (NV <= 64: all in wavefront 0, whose lane 0 takes the ticket below)
        double s = 0;
        for (int i = 0; i < kBlock / 64; ++i) s += lds[i * NV + threadIdx.x];
        __hip_atomic_store(dst + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(a.acc.fdone + f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(a.acc.fdone + f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool solve = a.fm_solve != 0;
    if (threadIdx.x < NV) {
        const double* p = a.acc.fpart + (size_t)f * a.acc.fcap * kFrameRow + threadIdx.x;
        double s = 0;
        for (int c = 0; c < (int)gridDim.x; c += 16) {          // 16 loads in flight, added in launch order
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = c + j < (int)gridDim.x ? __hip_atomic_load(p + (size_t)(c + j) * kFrameRow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) s += v[j];
        }
        if (solve && a.xf) {
            // multi-rank: `s` is this SLAB's row of frame f.  It goes into every rank's mailbox region (engine.h XfTable) behind a flag carrying the
            // exchange's number -- and this workgroup is DONE with the frame.  The R rows of a frame are added (rank order) and the frame solved by the
            // sweep's LAST workgroup below: ONE workgroup per rank ever waits for another rank, whatever the frame count and however few workgroups
            // the rank's CUs hold.  (Round 4 let the last workgroup of EVERY frame wait for the other ranks' rows: up to F waiting workgroups that keep
            // their CU slots -- and under the XCD-contiguous ids all of an empty frame's last arrivals sit on one XCD.  Eight ranks on 32 CUs each
            // with 32 keyframes dead-locked there: each rank's sweep stuck behind its own waiters, every one of them waiting for a frame of another
            // rank's stuck sweep; 400 keyframes on eight whole GPUs would have done the same.  profiles/r05_notes.md section 1.)
            const XfTable& t = *a.xf;
            const int Rk = t.n_ranks, buf = (int)(a.xf_epoch & 1);
            const long long slot = ((long long)buf * Rk + t.rank) * t.F + f;
            for (int r = 0; r < Rk; ++r) store8_system(t.region[r] + t.pay + slot * kFrameRow + threadIdx.x, s);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (NV <= 64: the whole row sits in wavefront 0)
            if (threadIdx.x == 0) for (int r = 0; r < Rk; ++r) __hip_atomic_store(t.region[r] + t.flg + slot, (double)a.xf_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        else if (solve) { __hip_atomic_store(a.acc.frame + (size_t)f * kFrameRow + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); lds[threadIdx.x] = s; }      // (the last frame's workgroup reads the energy columns of every row)
        else a.acc.frame[(size_t)f * kFrameRow + threadIdx.x] = s;
    }
    if (!solve) return;
    if (a.xf) return;      // multi-rank: this frame's slab row is out; k_frame_gather (launched behind the sweep) forms the global rows and solves
    __syncthreads();
    if (threadIdx.x == 0) {
        if (KIND == 0) { if (!ModelTraits<MODEL>::LED) frame_solve_light_sh<ModelTraits<MODEL>::NB == 3 ? 4 : ModelTraits<MODEL>::NB>(a.fm_frames, f, lds, a.fm_undo); }
        else frame_solve_pose(a, a.fm_frames, f, lds);
    }
    if (threadIdx.x < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(a.acc.fdone + a.F, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.y - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(a.acc.fdone + a.F, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    frame_tail<NV, KIND, MODEL>(a, lds, false);
}

// Multi-rank: the slabs' rows of a frame meet here, not inside the sweep.  The sweep's workgroups publish their slab's row of every frame into all
// ranks' regions and are done (frame_rows_publish); this kernel, launched behind the sweep on the same stream, has one workgroup per kGatherFrames
// frames: (a) wait -- bounded -- for the R flags of its frames, (b) element (f, l) of the global rows = the R slabs' values added in rank order (the
// same bits on every rank), written where the solves read them; the last workgroup to finish solves every frame (one lane each), the LED light vector
// and the energy sums (frame_tail).  Waiting happens in <= F / kGatherFrames workgroups of a kernel that starts when this rank's sweep has published
// everything the other ranks could be waiting for: no rank's progress ever depends on a waiting workgroup of another rank (round 4's in-sweep waits
// dead-locked at world size 8, profiles/r05_notes.md section 1; round 5's first fix did all of this in the sweep's very last workgroup: F x NV / 256
// serial gather rounds -- 45 us at 400 keyframes).
constexpr int kGatherFrames = 16;
template <int NV, int KIND, int MODEL>
__global__ void __launch_bounds__(kBlock) k_frame_gather(SweepArgs a) {
    __shared__ double lds[(kBlock / 64) * NV > 2 * (kBlock / 64) ? (kBlock / 64) * NV : 2 * (kBlock / 64)];
    __shared__ unsigned char s_late[kGatherFrames];
    __shared__ int s_last;
    const XfTable& t = *a.xf;
    const int Rk = t.n_ranks, buf = (int)(a.xf_epoch & 1);
    const double tag = (double)a.xf_epoch;
    double* const mine = t.region[t.rank];
    const int f0 = blockIdx.x * kGatherFrames, nf = min(kGatherFrames, a.F - f0);
    if (threadIdx.x < kGatherFrames) s_late[threadIdx.x] = 0;
    __syncthreads();
    // (a) one thread per (frame, rank) flag
    for (int q = threadIdx.x; q < nf * Rk; q += blockDim.x) {
        const int ff = f0 + q / Rk, r = q - (q / Rk) * Rk;
        const double* fp = mine + t.flg + ((long long)buf * Rk + r) * t.F + ff;
        int spins = 0; bool late = false;
        while (__hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != tag) { __builtin_amdgcn_s_sleep(2); if (++spins > t.spin_max) { late = true; break; } }
        if (late) {      // (a rank that never delivers: the host sees the NaN energy and reports PSGSDF_ERR_DEVICE -- and what was missing: engine.hip deliver_first)
            s_late[ff - f0] = 1;
            __hip_atomic_store(mine + kXrLate, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mine + kXrLate + 3, (double)(r * 1000 + ff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    // (b) the global rows of this workgroup's frames
    for (int i = threadIdx.x; i < nf * NV; i += blockDim.x) {
        const int fl = i / NV, l = i - fl * NV, ff = f0 + fl;
        const double* p0 = mine + t.pay + ((long long)buf * Rk * t.F + ff) * kFrameRow + l;
        double tot = 0.0;
        for (int r0 = 0; r0 < Rk; r0 += 8) {      // eight ranks' values in flight at a time
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = r0 + j < Rk ? __hip_atomic_load(p0 + (long long)(r0 + j) * t.F * kFrameRow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) tot += v[j];
        }
        if (s_late[fl]) tot = __builtin_nan("");
        __hip_atomic_store(a.acc.frame + (size_t)ff * kFrameRow + l, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(a.acc.fdone + a.F, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(a.acc.fdone + a.F, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    frame_tail<NV, KIND, MODEL>(a, lds, true);
}
template <int KIND>
static void launch_frame_gather(const SweepArgs& a, hipStream_t s) {
    const dim3 g((a.F + kGatherFrames - 1) / kGatherFrames), bl(kBlock);
    constexpr int NVP = 21 + 6 + 2;
    if (KIND == 1) {
        if (a.model == 0) hipLaunchKernelGGL((k_frame_gather<NVP, 1, 0>), g, bl, 0, s, a);
        else if (a.model == 1) hipLaunchKernelGGL((k_frame_gather<NVP, 1, 1>), g, bl, 0, s, a);
        else hipLaunchKernelGGL((k_frame_gather<NVP, 1, 2>), g, bl, 0, s, a);
    } else {      // light rows: NH + NB + 2 (SH1: 10 + 4 + 2, SH2: 45 + 9 + 2, LED: 3 + 3 + 2)
        if (a.model == 0) hipLaunchKernelGGL((k_frame_gather<16, 0, 0>), g, bl, 0, s, a);
        else if (a.model == 1) hipLaunchKernelGGL((k_frame_gather<56, 0, 1>), g, bl, 0, s, a);
        else hipLaunchKernelGGL((k_frame_gather<8, 0, 2>), g, bl, 0, s, a);
    }
}

// light normal equations: lightJacobian PsOptimizerJa.cpp:132-143,323-371 (per frame NBxNB),
// LED LightJacobian LedOptimizerJa.cpp:101-115,299-346 (one global diagonal 3x3)
template <int MODEL, int LOSS, int IMG>
__global__ void __launch_bounds__(kBlock) k_sweep_light(SweepArgs a, int rows) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    constexpr int NB = ModelTraits<MODEL>::NB;
    constexpr bool LED = ModelTraits<MODEL>::LED;
    constexpr int NH = LED ? 3 : NB * (NB + 1) / 2;
    constexpr int NV = NH + NB + 2;   // + energy, n_obs
    __shared__ FrameP sfp;
    __shared__ double lds[(kBlock / 64) * NV];
    int cx, f; fm_ids(a, cx, f);
    if (threadIdx.x < (int)(sizeof(FrameP) / 4)) ((float*)&sfp)[threadIdx.x] = ((const float*)(a.frames + f))[threadIdx.x];
    __syncthreads();
    const Band& b = a.b;
    const FrameP& fp = sfp;
    ImgSrc img = a.im;                       // this frame's image as frame 0 of its own stack (offsets always fit 32 bits)
    if (img.u8) img.u8 += (size_t)f * a.cam.H * a.cam.W; else img.f32 += (size_t)f * a.cam.H * a.cam.W * 3;
    img.idx32 = true;
    obs_acc_t acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0;
    const int beg = b.obs_ptr[f], end = b.obs_ptr[f + 1];
    // one observation's contribution once its colour I is known (shared by the two loop forms below)
    auto accumulate = [&](const Vox& v, const Proj& pr, const float* I) {
        float shfd[kMaxBasis], shg[kMaxBasis];
        if (!LED) { SH<NB == 3 ? 4 : NB>(v.nfd, shfd); SH<NB == 3 ? 4 : NB>(v.gn, shg); }
        float ren[3];
        rendered<MODEL>(fp, pr, v.nfd, shfd, v.rho, ren);
        float refl = 0.f;
        if (LED) { float Rp[3]; mul3(fp.R, pr.p, Rp); refl = dot3(v.gn, Rp); float pn = norm3(pr.p); double pd = (double)pn; refl /= (float)(pd * pd * pd); }
        float l = 0.f, w2 = 0.f, r1 = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float r = I[ch] - ren[ch]; float w = robust_weight<LOSS>(a.rob, r);
            l += robust_loss<LOSS>(a.rob, r);
            if (LED) {
                float J = refl * v.rho[ch]; float jw = J * w;
                acc[ch] += (obs_acc_t)(jw * J); acc[NH + ch] += (obs_acc_t)(jw * r);
            } else {   // J_c = -rho_c SH(g): the three channels share the direction SH(g), so their normal equations are ONE outer product
#if PSG_STRICT & 16
                float Jc[kMaxBasis];      // the reference's order: one row per channel, J = -rho_c SH(g), H += (J w) J^T (PsOptimizerJa.cpp:132-143,323-371; oracle light_system)
#pragma unroll
                for (int i = 0; i < NB; ++i) Jc[i] = -v.rho[ch] * shg[i];
                int q = 0;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const float jw = Jc[i] * w;
#pragma unroll
                    for (int k = i; k < NB; ++k) acc[q++] += (obs_acc_t)(jw * Jc[k]);
                    acc[NH + i] += (obs_acc_t)(jw * r);
                }
#else
                w2 += w * (v.rho[ch] * v.rho[ch]);
                r1 -= w * (v.rho[ch] * r);
#endif
            }
        }
#if !(PSG_STRICT & 16)
        if (!LED) {    // H += (sum_c w_c rho_c^2) SH SH^T ; b += (-sum_c w_c rho_c r_c) SH   (lightJacobian, PsOptimizerJa.cpp:132-143,323-371)
            int q = 0;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const float sw = shg[i] * w2;
#pragma unroll
                for (int k = i; k < NB; ++k) acc[q++] += (obs_acc_t)(sw * shg[k]);
                acc[NH + i] += (obs_acc_t)(shg[i] * r1);
            }
        }
#endif
        acc[NH + NB] += (obs_acc_t)l; acc[NH + NB + 1] += 1;
    };
    const int e0 = beg + cx * (kBlock * rows) + threadIdx.x;
    if constexpr (IMG >= 0 && !(PSG_STRICT & 4) && (PSG_FM_PIPE & 1)) {
        // three-stage software pipeline over the thread's observations (device_common.h fm_for_each_obs; rows <= 64: fm_rows)
        fm_for_each_obs<IMG>(b, fp, a.cam, img, e0, end, rows, [&](const ObsPend<IMG>& P) {
            float I[3];
            taps_colour<IMG, false>(img, a.cam, P.ts, P.pr.m, P.pr.n, P.pr.m, P.pr.n, I, nullptr, nullptr);
            accumulate(P.v, P.pr, I);
            return true;
        }, [&](const Vox&, const Proj&) {});
    } else {
    // Software pipeline over the thread's observations: the row index is fetched two observations ahead and the voxel state one
    // ahead, so that an iteration waits for its image taps only (un-pipelined, index -> state -> taps were three dependent round
    // trips per observation and the sweep was latency-bound at 4-6 waves per SIMD).
    int e = e0;
    int j_cur = e < end ? b.obs_rows[e] : -1;
    int j_nxt = (rows > 1 && e + kBlock < end) ? b.obs_rows[e + kBlock] : -1;
    Vox vn;
    if (j_cur >= 0) load_vox(b, j_cur, vn);
    for (int it = 0; it < rows && j_cur >= 0; ++it, e += kBlock) {
        const Vox v = vn;
        const int j_nn = (it + 2 < rows && e + 2 * kBlock < end) ? b.obs_rows[e + 2 * kBlock] : -1;
        if (j_nxt >= 0) load_vox(b, j_nxt, vn);
        j_cur = j_nxt; j_nxt = j_nn;
        Proj pr = project(v.xs, fp, a.cam);
        if (!pr.ok) continue;
        float I[3];
        sample<false, IMG>(img, 0, a.cam, pr.m, pr.n, I, nullptr, nullptr);
        accumulate(v, pr, I);
    }
    }
    // row layout: [NH H entries | NB rhs | energy | n_obs]
    const int w = threadIdx.x >> 6;
    wave_sums_to<NV>(acc, lds + w * NV);   // double: SH2 light blocks are ill-conditioned
    __syncthreads();
    frame_rows_publish<NV, 0, MODEL>(a, cx, f, lds);
}
int launch_sweep_light(const SweepArgs& a, hipStream_t s) {
    if (a.b.S <= 0 || a.F <= 0 || a.b.obs_max <= 0) return 0;
    const int rows = fm_rows(a, a.model == 1 ? 3 : ((PSG_FM_PIPE & 1) ? 4 : 5));           // resident workgroups per CU at this kernel's register count (pipelined: 108-120 VGPRs, SH2 161-168)
    const int chunk = kBlock * rows;
    dim3 g((a.b.obs_max + chunk - 1) / chunk, a.F), bl(kBlock);
    PSG_LAUNCH_SWEEP(k_sweep_light, a, true, g, bl, 0, s, a, rows);
    if (a.xf && a.fm_solve) launch_frame_gather<0>(a, s);      // multi-rank: the slabs' rows meet, and the frames are solved, in a kernel of their own behind the sweep
    return (int)g.x;
}

// pose normal equations: poseJacobian PsOptimizerJa.cpp:61-115,427-475 / LedOptimizerJa.cpp:32-81,351-399
template <int MODEL, int LOSS, int IMG>
__global__ void __launch_bounds__(kBlock, PSG_POSE_WAVES) k_sweep_pose(SweepArgs a, int rows) {
    { __shared__ double fred[kBlock / 64]; fold_pending(a, fred); }
    constexpr int NB = ModelTraits<MODEL>::NB;
    constexpr bool LED = ModelTraits<MODEL>::LED;
    constexpr int NV = 21 + 6 + 2;
    __shared__ FrameP sfp;
    __shared__ double lds[(kBlock / 64) * NV];
    int cx, f; fm_ids(a, cx, f);
    if (threadIdx.x < (int)(sizeof(FrameP) / 4)) ((float*)&sfp)[threadIdx.x] = ((const float*)(a.frames + f))[threadIdx.x];
    __syncthreads();
    const Band& b = a.b;
    const FrameP& fp = sfp;
    ImgSrc img = a.im;                       // this frame's image as frame 0 of its own stack (offsets always fit 32 bits)
    if (img.u8) img.u8 += (size_t)f * a.cam.H * a.cam.W; else img.f32 += (size_t)f * a.cam.H * a.cam.W * 3;
    img.idx32 = true;
    obs_acc_t acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0;
    const int beg = b.obs_ptr[f], end = b.obs_ptr[f + 1];
    // one observation's contribution once its colour I and image gradient (gu, gv) are known (shared by the two loop forms below)
    auto accumulate = [&](const Vox& v, const Proj& pr, const ProjJ& pj, const float* I, const float* gu, const float* gv) {
        float shfd[kMaxBasis];
        if (!LED) SH<NB == 3 ? 4 : NB>(v.nfd, shfd);
        float ren[3];
        rendered<MODEL>(fp, pr, v.nfd, shfd, v.rho, ren);
        // J_c = image_grad_c pi_grad [-R^T | skew(p)] (PsOptimizerJa.cpp:78-100), contracted from the right (device_common.h pi_rows):
        // J_c = gu_c [-U | a skew(p)] + gv_c [-V | b skew(p)] with the channel-independent rows written out (structural zeros dropped)
        const PiRows pi = pi_rows(a.cam, pr);
        const float* p = pr.p;
        float J[18];
#if PSG_STRICT & 16
        {   // the reference's order (PsOptimizerJa.cpp:78-100 / LedOptimizerJa.cpp:48-78; oracle pose_jacobian): G = image_grad pi_grad with its structural zeros, -G R^T, G skew(p)
            float G[9];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { G[ch * 3 + 0] = gu[ch] * pi.p00 + gv[ch] * 0.0f; G[ch * 3 + 1] = gu[ch] * 0.0f + gv[ch] * pi.p11; G[ch * 3 + 2] = gu[ch] * pi.p02 + gv[ch] * pi.p12; }
            const float sk[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float sg = (G[ch * 3 + 0] * fp.R[k * 3 + 0] + G[ch * 3 + 1] * fp.R[k * 3 + 1]) + G[ch * 3 + 2] * fp.R[k * 3 + 2];
                    J[ch * 6 + k] = -sg;
                    J[ch * 6 + 3 + k] = (G[ch * 3 + 0] * sk[0 * 3 + k] + G[ch * 3 + 1] * sk[1 * 3 + k]) + G[ch * 3 + 2] * sk[2 * 3 + k];
                }
            if (LED) {
                const float pn = norm3(p); const double pd = (double)pn; const float l3 = (float)(pd * pd * pd);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float sl = -(v.rho[ch] * fp.l[ch] / l3);
#pragma unroll
                    for (int k = 0; k < 3; ++k) J[ch * 6 + k] += sl * v.gn[k];
                }
            }
        }
#else
        float U[3], V[3]; pi_rows_world(pi, fp.R, U, V);
        const float AS[3] = {-(pi.p02 * p[1]), pi.p02 * p[0] - pi.p00 * p[2], pi.p00 * p[1]};
        const float BS[3] = {pi.p11 * p[2] - pi.p12 * p[1], pi.p12 * p[0], -(pi.p11 * p[0])};
        if (!LED) {
            // SH models: J_c = gu_c A + gv_c B with the channel-independent 6-vectors A = [-U | a skew(p)], B = [-V | b skew(p)], so the three channels'
            // normal equations are TWO outer products with three scalar sums in front (round 6; the light sweep does the same with its one direction):
            //   sum_c w_c J_c J_c^T = (aa A + ab B) A^T + (ab A + bb B) B^T,   aa = sum w gu^2, ab = sum w gu gv, bb = sum w gv^2
            //   sum_c w_c r_c J_c   = ra A + rb B,                             ra = sum w gu r,  rb = sum w gv r
            // 99 instead of 135 instructions per observation; same value up to the rounding of the regrouped products (engine deviation 7; PSG_STRICT & 16:
            // the reference's row-per-channel order above)
            float A6[6], B6[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) { A6[k] = -U[k]; B6[k] = -V[k]; A6[3 + k] = AS[k]; B6[3 + k] = BS[k]; }
            float aa = 0.f, ab = 0.f, bb = 0.f, ra = 0.f, rb = 0.f, l = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float r = I[ch] - ren[ch]; float w = robust_weight<LOSS>(a.rob, r);
                l += robust_loss<LOSS>(a.rob, r);
                w = pj.ok ? w : 0.f;                      // the residual counts for the energy, but the row has no Jacobian
                const float wu = w * gu[ch], wv = w * gv[ch];
                aa += wu * gu[ch]; ab += wu * gv[ch]; bb += wv * gv[ch]; ra += wu * r; rb += wv * r;
            }
            int q = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float Pi = aa * A6[i] + ab * B6[i], Qi = ab * A6[i] + bb * B6[i];
#pragma unroll
                for (int k = i; k < 6; ++k) { acc[q] += (obs_acc_t)(Pi * A6[k]); acc[q] += (obs_acc_t)(Qi * B6[k]); ++q; }      // (two chained multiply-adds)
                acc[21 + i] += (obs_acc_t)(ra * A6[i]); acc[21 + i] += (obs_acc_t)(rb * B6[i]);
            }
            acc[27] += (obs_acc_t)l; acc[28] += 1;
            return;
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                J[ch * 6 + k] = -(gu[ch] * U[k] + gv[ch] * V[k]);
                J[ch * 6 + 3 + k] = gu[ch] * AS[k] + gv[ch] * BS[k];
            }
        {
            float pn = norm3(p); double pd = (double)pn; float l3 = (float)(pd * pd * pd);
            const float yl3 = 1.0f / l3;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float s = -div_by(v.rho[ch] * fp.l[ch], l3, yl3);
#pragma unroll
                for (int k = 0; k < 3; ++k) J[ch * 6 + k] += s * v.gn[k];
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
            for (int i = 0; i < 6; ++i) {
                float jw = J[ch * 6 + i] * w;
#pragma unroll
                for (int k = i; k < 6; ++k) acc[q++] += (obs_acc_t)(jw * J[ch * 6 + k]);
                acc[21 + i] += (obs_acc_t)(jw * r);
            }
        }
        acc[27] += (obs_acc_t)l; acc[28] += 1;
    };
    const int e0 = beg + cx * (kBlock * rows) + threadIdx.x;
    if constexpr (IMG >= 0 && !(PSG_STRICT & 4) && (PSG_FM_PIPE & 2)) {
        // three-stage software pipeline over the thread's observations (device_common.h fm_for_each_obs; rows <= 64: fm_rows); an observation whose
        // gradient needs taps outside its cell (image border, the Jacobian's projection in the neighbouring cell) is evaluated behind the loop
        fm_for_each_obs<IMG>(b, fp, a.cam, img, e0, end, rows, [&](const ObsPend<IMG>& P) {
            const ProjJ pj = project_jac(P.pr, a.cam);      // poseJacobian projects again, with its own in-image test (PsOptimizerJa.cpp:70-76)
            float I[3], gu[3], gv[3];
            if (!taps_colour<IMG, true>(img, a.cam, P.ts, P.pr.m, P.pr.n, pj.mj, pj.nj, I, gu, gv)) return false;
            accumulate(P.v, P.pr, pj, I, gu, gv);
            return true;
        }, [&](const Vox& v, const Proj& pr) {
            const ProjJ pj = project_jac(pr, a.cam);
            float I[3], gu[3], gv[3];
            sample<true, IMG>(img, 0, a.cam, pr.m, pr.n, pj.mj, pj.nj, I, gu, gv);
            accumulate(v, pr, pj, I, gu, gv);
        });
    } else {
    // Software pipeline over the thread's observations: the row index is fetched two observations ahead and the voxel state one
    // ahead, so that an iteration waits for its image taps only (un-pipelined, index -> state -> taps were three dependent round
    // trips per observation and the sweep was latency-bound at 4-6 waves per SIMD).
    int e = e0;
    int j_cur = e < end ? b.obs_rows[e] : -1;
    int j_nxt = (rows > 1 && e + kBlock < end) ? b.obs_rows[e + kBlock] : -1;
    Vox vn;
    if (j_cur >= 0) load_vox(b, j_cur, vn);
    for (int it = 0; it < rows && j_cur >= 0; ++it, e += kBlock) {
        const Vox v = vn;
        const int j_nn = (it + 2 < rows && e + 2 * kBlock < end) ? b.obs_rows[e + 2 * kBlock] : -1;
        if (j_nxt >= 0) load_vox(b, j_nxt, vn);
        j_cur = j_nxt; j_nxt = j_nn;
        Proj pr = project(v.xs, fp, a.cam);
        if (!pr.ok) continue;
        float I[3], gu[3], gv[3];
        const ProjJ pj = project_jac(pr, a.cam);      // poseJacobian projects again, with its own in-image test (PsOptimizerJa.cpp:70-76)
        sample<true, IMG>(img, 0, a.cam, pr.m, pr.n, pj.mj, pj.nj, I, gu, gv);
        accumulate(v, pr, pj, I, gu, gv);
    }
    }
    const int w = threadIdx.x >> 6;
    wave_sums_to<NV>(acc, lds + w * NV);   // row layout: [21 H | 6 rhs | energy | n_obs]
    __syncthreads();
    frame_rows_publish<NV, 1, MODEL>(a, cx, f, lds);
}
int launch_sweep_pose(const SweepArgs& a, hipStream_t s) {
    if (a.b.S <= 0 || a.F <= 0 || a.b.obs_max <= 0) return 0;
    const int rows = fm_rows(a, ((PSG_FM_PIPE & 2) && PSG_POSE_WAVES < 4) ? 3 : 4);      // (pipelined: 131-145 VGPRs)
    const int chunk = kBlock * rows;
    dim3 g((a.b.obs_max + chunk - 1) / chunk, a.F), bl(kBlock);
    PSG_LAUNCH_SWEEP(k_sweep_pose, a, true, g, bl, 0, s, a, rows);
    if (a.xf && a.fm_solve) launch_frame_gather<1>(a, s);
    return (int)g.x;
}

// ------------------------------------------------------------------------------------------
// small dense solves: LDL^T in double, zero step on non-positive pivots
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

// The per-frame solves run as ONE workgroup (a thread takes frames tid, tid+256, ...), which lets the same kernel also sum the
// energy / n_obs columns of the rows over the frames (e_out, may be host-mapped) in a fixed order.  The rows are final when it runs:
// the last workgroup of every frame of the sweep has summed that frame's partial rows (frame_rows_publish).
__device__ __forceinline__ void frame_rows_finish(const SweepArgs& a, int col_e, double* e_out, unsigned long long e_key, double* red) {
    __syncthreads();                                       // every thread has read the rows it solves from
    if (e_out) {
        double e = 0, n = 0;
        for (int f = threadIdx.x; f < a.F; f += blockDim.x) { e += a.acc.frame[(size_t)f * kFrameRow + col_e]; n += a.acc.frame[(size_t)f * kFrameRow + col_e + 1]; }
        e = wave_sum(e); n = wave_sum(n);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) { red[2 * w] = e; red[2 * w + 1] = n; }
        __syncthreads();
        if (threadIdx.x == 0) { double te = 0, tn = 0; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { te += red[2 * i]; tn += red[2 * i + 1]; } mbox_put(e_out, 2, 0, te, e_key); mbox_put(e_out, 2, 1, tn, e_key); mbox_commit(e_key); }
        __syncthreads();
    }
}

// optimizeLightAll: PsOptimizer.cpp:175-203 (no damping) / LedOptimizer.cpp:134-160 (damped, one RGB vector)
template <int MODEL>
// undo (nullable): a speculative light update (loop.hip run_loop) keeps what it overwrites -- the frames' light coefficients [F][9] and the LED light [3] behind them --
// so that k_restore_light can put them back if the loop ends on the previous iteration
__global__ void __launch_bounds__(kBlock) k_solve_light(SweepArgs a, FrameP* frames, float* led_light, double* e_out, unsigned long long e_key, float* undo) {
    if (undo) {
        for (int i = threadIdx.x; i < a.F * 9; i += blockDim.x) undo[i] = frames[i / 9].l[i % 9];
        if (threadIdx.x < 3) undo[a.F * 9 + threadIdx.x] = led_light[threadIdx.x];
        __syncthreads();
    }
    constexpr int NB = ModelTraits<MODEL>::NB;
    constexpr bool LED = ModelTraits<MODEL>::LED;
    constexpr int NH = LED ? 3 : NB * (NB + 1) / 2;
    __shared__ double red[2 * kBlock / 64];
    if (LED) {
        // every thread sums the per-frame rows (frame order), solves the same 3 scalar equations and updates its own records.  The six columns
        // are staged in LDS first: summed straight from memory, 2 x 3 x F dependent global loads per thread made this kernel 21.7 us at F = 50
        constexpr int kStage = 512;
        __shared__ double srow[6 * kStage];
        const bool staged = a.F <= kStage;
        if (staged) {
            for (int i = threadIdx.x; i < 6 * a.F; i += blockDim.x) { const int ff = i / 6, c6 = i % 6; srow[c6 * kStage + ff] = a.acc.frame[(size_t)ff * kFrameRow + (c6 < 3 ? c6 : NH + c6 - 3)]; }
            __syncthreads();
        }
        float dl[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            double hs = 0, bs = 0;
            if (staged) { for (int ff = 0; ff < a.F; ++ff) { hs += srow[ch * kStage + ff]; bs += srow[(3 + ch) * kStage + ff]; } }
            else for (int ff = 0; ff < a.F; ++ff) { hs += a.acc.frame[(size_t)ff * kFrameRow + ch]; bs += a.acc.frame[(size_t)ff * kFrameRow + NH + ch]; }
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
        for (int f = threadIdx.x; f < a.F; f += blockDim.x) frame_solve_light_sh<NB>(frames, f, a.acc.frame + (size_t)f * kFrameRow, nullptr);      // (undo: copied above)
    }
    frame_rows_finish(a, NH + NB, e_out, e_key, red);
}
__global__ void __launch_bounds__(kBlock) k_restore_light(int F, FrameP* frames, float* led_light, const float* undo) {
    for (int i = threadIdx.x; i < F * 9; i += blockDim.x) frames[i / 9].l[i % 9] = undo[i];
    if (threadIdx.x < 3) led_light[threadIdx.x] = undo[F * 9 + threadIdx.x];
}
void launch_restore_light(int F, FrameP* frames, float* led_light, const float* undo, hipStream_t s) { if (F > 0) hipLaunchKernelGGL(k_restore_light, dim3(1), dim3(kBlock), 0, s, F, frames, led_light, undo); }
void launch_solve_light(const SweepArgs& a, FrameP* frames, float* led_light, double* e_out, unsigned long long e_key, float* undo, hipStream_t s) {
    if (a.F <= 0) return;
    dim3 g(1), bl(kBlock);
    if (a.model == 0) hipLaunchKernelGGL((k_solve_light<0>), g, bl, 0, s, a, frames, led_light, e_out, e_key, undo);
    else if (a.model == 1) hipLaunchKernelGGL((k_solve_light<1>), g, bl, 0, s, a, frames, led_light, e_out, e_key, undo);
    else hipLaunchKernelGGL((k_solve_light<2>), g, bl, 0, s, a, frames, led_light, e_out, e_key, undo);
}

// optimizePosesAll PsOptimizer.cpp:207-234 + updatePose OptimizerAux.cpp:190-205
__global__ void __launch_bounds__(kBlock) k_solve_pose(SweepArgs a, FrameP* frames, double* e_out, unsigned long long e_key) {
    __shared__ double red[2 * kBlock / 64];
    for (int f = threadIdx.x; f < a.F; f += blockDim.x) frame_solve_pose(a, frames, f, a.acc.frame + (size_t)f * kFrameRow);
    frame_rows_finish(a, 27, e_out, e_key, red);
}
void launch_solve_pose(const SweepArgs& a, FrameP* frames, double* e_out, unsigned long long e_key, hipStream_t s) {
    if (a.F > 0) hipLaunchKernelGGL(k_solve_pose, dim3(1), dim3(kBlock), 0, s, a, frames, e_out, e_key);
}

}  // namespace psg
