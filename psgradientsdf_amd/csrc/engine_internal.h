// engine_internal.h -- the context and the host-side helpers shared by the engine's source files:
//   engine.hip        context plumbing: launch timing, deferred read-backs through the mapped mailbox, band construction, energies
//   loop.hip          the solves and the alternation loop (fused PCG driver, sub-steps, psgsdf_iterate / psgsdf_optimize control flow)
//   api.hip           the C ABI of include/psgsdf.h (volume, keyframes, init, steps, downloads)
//   api_frontend.hip  frame fusion, FALS normals, depth tracker        api_multi_gpu.hip  the z-slab phase API
//   api_debug.hip     measurement and test hooks
// Internal: nothing here is part of the boundary (include/psgsdf.h).
#pragma once
#include "engine.h"
#include "../../include/psgsdf.h"

#include <math.h>
#include <float.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <memory>
#include <map>
#include <string>
#include <vector>

namespace psge {
constexpr int kMgScal = 96;   // doubles in the set-up exchange buffer (3 per rank: at most 32 ranks)
struct KTime { double ms = 0; int64_t n = 0; };
struct Comm;                  // comm.hip: RCCL communicator or caller-supplied transport
struct MgSeg { unsigned off, n; unsigned long long key; };
// a scalar read-back waiting in the host-mapped mailbox: n values at src, their check words behind them (engine.h FoldReq; key 0: unchecked)
struct Deferred { const double* src; int n; unsigned long long key; std::function<void(const double*)> consume; };
}  // namespace psge
using psge::KTime;
using namespace psg;   // the layout structs of engine.h

struct psgsdf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    psgsdf_settings set{};
    float reg_n = 0, reg_l = 0;
    GridP grid{};                        // the LOCAL grid (whole volume on one rank; z-planes [zlo, zhi) of it on a slab)
    int gdim[3]{}; long long gnvox = 0;  // the whole volume
    int z0 = 0, z1 = 0;                  // global z-planes this context OWNS: [z0, z1)
    int zlo = 0, zhi = 0;                // global z-planes it HOLDS: [zlo, zhi) = own planes + one halo plane on each inner side
    long long S_global = 0;              // band voxels of the whole volume (sum of the slabs' own rows)
    float shift[3]{};
    Cam cam{};
    // dense
    DenseView dense{};
    uint64_t* vis_seq = nullptr; int wpv_seq = 0;
    int* block_sums = nullptr; int* d_total = nullptr;
    bool have_volume = false;
    // frames
    int F = 0;
    int* frame_idx = nullptr;
    float* img = nullptr;
    FrameP* frames = nullptr;            // device
    std::vector<FrameP> frames_h;        // host mirror of the initial records
    float* led_light = nullptr;          // device [3]
    bool have_frames = false;
    // band
    void* band_mem = nullptr; size_t band_bytes = 0;
    void* rec_mem = nullptr;             // the two PCG record planes (Band::rec): an allocation of their own, fine-grained on a multi-rank context (the neighbours' solve kernels write its halo rows while this rank's kernel reads them)
    void* obs_mem = nullptr;
    float* stage = nullptr; size_t stage_px = 0;   // device staging of one RGB-D frame (integrate_frame)
    void* pin_stage = nullptr; size_t pin_bytes = 0;   // pinned host staging of the caller's frame (host_stage): the DMA never has to pin the caller's pages
    // front end: FALS cache (9 float planes), box-filter scratch, tracker partials
    float* ncache = nullptr; double* ntmp = nullptr; float* nout = nullptr; float* ndepth = nullptr; int ncache_w = 0, ncache_h = 0;
    double* track_part = nullptr; double* track_host = nullptr;
    Band band{};
    bool inited = false;
    // accumulators
    double* acc_frame = nullptr; size_t acc_frame_n = 0;
    double* frame_part = nullptr; int frame_cap = 0;   // [F][frame_cap][kFrameRow] per-workgroup partial rows of the frame-major sweeps
    int* frame_done = nullptr;                         // [F] arrival counters
    double* part = nullptr; int PB = 0;  // [SC_COUNT][PB] per-workgroup partials
    double* pcg_sc = nullptr; int pcg_cap = 4096;
    double* pcg_part = nullptr;          // [2][7][kPcgMaxBlocks]
    double* pcg_gran = nullptr;          // persistent solve: [2][kSolveGranPlanes][kSolveMaxBlocksHost] tagged per-workgroup sums
    bool pcg_fuse_asm = true;            // PSGSDF_PCG_FUSE_ASM=0: k_assemble in front of the persistent solve (round-2a)
    unsigned pcg_solve_serial = 0;       // distance solves launched on this context (SweepArgs::pcg_epoch)
    bool pcg_tagm_mr = true;             // ... also between the ranks of a multi-rank context (the default, PSGSDF_PCG_TAGM=2: halo rows gathered at system scope from the first attempt on); PSGSDF_PCG_TAGM=1: on one rank only
    bool pcg_tagm = true;                // PSGSDF_PCG_TAGM (default 2): the pipelined solve with self-validating exchanged values (pcg.hip k_cgp_solve<.., TM>); 0 = round 4's ordered hand-off; whether it also runs BETWEEN ranks: pcg_tagm_mr
    int pcg_ablate = 0;                  // PSGSDF_PCG_ABLATE: timing ablations of the pipelined solve (wrong results; tools only)
    bool pcg_prefetch = true;            // PSGSDF_PCG_PREFETCH=0: pipelined solve: the sums of a pass are only fetched after its gathers (not behind the last gather batch)
    bool pcg_pipeline = true;            // PSGSDF_PCG_PIPELINE=0: the persistent solve with round 2's recurrences (k_cgf_solve: a pass waits for its own reduction)
    bool pcg_fuse_apply = true;          // PSGSDF_PCG_FUSE_APPLY=0: k_apply_dist behind it
    bool pcg_xcd_local = true;           // PSGSDF_PCG_XCD_LOCAL=0: every record through memory (write-through stores)
    bool pcg_persist = true;             // PSGSDF_PCG_PERSIST=0: always the per-pass kernels
    bool persist_off = false;            // a persistent solve gave up on this band (bounded wait expired): per-pass kernels until the next band is built (build_band re-arms)
    int num_cu = 0;
    int last_cg_iters = 0;
    bool want_counts = true;             // read back the accepted-update counts (debug statistic of the reference)
    double* host_buf = nullptr; size_t host_buf_n = 0;   // pinned readback
    // deferred read-backs: small fold kernels write into a host-mapped pinned mailbox (no D2H copies), consumed at the
    // next host sync
    double* mbox = nullptr; double* mbox_dev = nullptr; size_t mbox_n = 0, mbox_used = 0;
    size_t mbox_alloc = 0; double flush_seq = 0;    // the last slot of the allocation is the flush marker
    std::vector<psge::Deferred> deferred;
    unsigned long long mbox_serial = 0;  // key generator of the read-back slots
    bool mbox_check = true;              // PSGSDF_MBOX_CHECK=0: take read-backs on the marker's / status word's say-so (round-2 behaviour; reproduces its flake)
    // speculative start of an iteration (loop.hip run_loop): albedo / light updates applied before the stop decision of the previous iteration is
    // known keep what they overwrite, so that the loop can still end on exactly the state the reference ends on
    psgsdf_iter_cb observer = nullptr; void* observer_user = nullptr;   // psgsdf_set_record_observer
    int on_iter_period = 1;              // psgsdf_set_on_iter_period
    bool speculate = true;               // PSGSDF_SPECULATE=0: always wait for the decision first (round 2)
    bool spec_undo = false, spec_albedo_saved = false, spec_light_saved = false;
    FrameP* frames_undo = nullptr;       // device [F] + 3 floats (LED light)
    long long spec_windows = 0, spec_undos = 0;
    int fault_solve = 0, solves_seen = 0;   // PSGSDF_FAULT_SOLVE=n: fault injection into the n-th persistent solve of this context
    long long persist_fallbacks = 0;     // distance steps re-run on the per-pass kernels after the persistent solve gave up (loop.hip)
    long long mbox_checked = 0, mbox_late = 0;   // read-backs validated / of those: not complete yet when the host was told everything had landed
    // cached energies
    double en_sum = 0, el_sum = 0;       // sums over the band from the last k_derive
    // row partition (multi-rank): this context owns band rows [row0, row1); halo = widest column reach
    int rank = 0, n_ranks = 1;
    int row0 = 0, row1 = 0, halo = 0;
    psge::Comm* comm = nullptr;          // transport of the multi-rank exchanges (comm.hip); null on a single-rank context
    // cross-rank persistent solve (comm.hip xr_setup, pcg.hip k_cgf_solve<.., MR>): this rank's mailbox region, every rank's region and the two
    // neighbours' band arenas mapped through IPC handles; rebuilt with every band
    bool xr_enable = true;               // PSGSDF_XR=0: multi-rank contexts always use the per-pass kernels + RCCL all-reduce (round 2)
    bool xr_ready = false; long long xr_solves = 0;
    double* xr = nullptr;
    int xr_mem_kind = -1;                // memory kind of xr / rec_mem chosen by xr_probe: 1 fine-grained records + uncached region, 2 both uncached, 0 none passed (cross-rank solve off); -1 not probed yet
    long long xr_probe_stale = 0, xr_probe_timeouts = 0;   // what the probe saw (all ranks, all kinds tried)
    long long xr_probe_local[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // [memory kind 1 / 2][stale records read from the lower neighbour, expired waits towards the lower, towards the upper neighbour, tried]: this rank's own view (psgsdf_get_tuning)
    bool xr_mapped = false;              // peers may hold IPC mappings of xr / rec_mem: they have to be closed everywhere before either is freed (xr_quiesce)
    std::shared_ptr<void> comm_keep;     // what a built-in caller-side transport (psgsdf_comm_init_sockets) needs for the life of the context
    unsigned long long xr_openers = 0;   // bit r: rank r opened this rank's region at the last set-up (agreed there); xr_quiesce waits for exactly those
    long long xr_serial = 0, xr_closed_off = 0;   // number of the last set-up (the same on every rank) and where the R "closed" slots of a region sit
    void* xo_host[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; size_t xo_bytes[5] = {0, 0, 0, 0, 0};   // pinned host results of the extraction calls (extract.hip): mesh xyz / rgb, point cloud xyz+n / rgb, sdf block; valid until the next extraction
    long long xr_stale_maps = 0;         // mappings that did not show their owner's nonce (xr_setup)
    bool leak_exported = false;          // a peer never reported its mappings closed: xr / rec_mem / hx_mem are never freed by this context
    std::vector<double*> xr_peer;        // [n_ranks] (own entry = xr)
    void* band_peer[2] = {nullptr, nullptr};   // lower / upper neighbour's band arena
    std::vector<void*> xr_opened;        // IPC mappings to close
    XrArgs xr_args{};
    int cu_mask_lo = -1, cu_mask_hi = -1;      // PSGSDF_CU_MASK=lo:hi: the context's stream only uses CUs [lo, hi) (two ranks sharing one GPU)
    double* mg_scal = nullptr;           // [kMgScal] set-up exchange (what each slab needs of its neighbours)
    double* mg_ext = nullptr;            // [8] PCG: local sums of a pass out, globally reduced sums in
    // multi-rank: deferred scalar read-backs are folded into a DEVICE shadow of the mailbox; before the host waits on the mailbox the
    // segments written since the last commit are all-reduced in ONE collective and copied to their mailbox slots (engine.hip: mg_commit)
    double* mbox_shadow = nullptr;
    std::vector<psge::MgSeg> mg_segs;
    int give[2] = {0, 0};                // rows the lower / upper neighbour needs of this slab
    bool halo_active = false;            // any stencil of any slab crosses a cut
    long long n_collectives = 0;
    float reg_r = 0.f;                   // "reg albedo" (never normalised, PsOptimizer.cpp:279)
    void* areg_mem = nullptr; AlbedoReg ar{};   // planes of the albedo regulariser, allocated with the band when reg_r != 0
    bool areg_device = true; int areg_last_iters = 0;      // PSGSDF_AREG_DEVICE=0: the regularised albedo solve driven by the host (two read-backs per CG iteration: what multi-rank contexts run)
    double er_sum = 0;                   // sum over the band of sum_c ||grad rho_c|| at the last evaluation
    FoldReq pending_fold{};              // scalar fold waiting for the next kernel (read_parts_deferred / take_fold)
    bool fuse_pcg_init = true;           // PSGSDF_FUSE_PCG_INIT=0: separate k_cgf_init launch
    bool fuse_albedo = true;             // PSGSDF_FUSE_ALBEDO=0: separate k_apply_albedo launch
    bool fold_in_next = true;            // PSGSDF_FOLD_IN_NEXT=0: always a k_sum_parts launch
    int xwait_spins = 1 << 24;           // polls of a wait inside a kernel for a peer's flag (PSGSDF_XWAIT_LOG2)
    long long fault_halo = 0;            // PSGSDF_FAULT_HALO
    bool xh_enable = true;               // PSGSDF_XH=0: halo rows through RCCL send / recv (round 3)
    void* hx_mem = nullptr;              // halo staging the two neighbours push into (fine-grained, IPC-exported): [2 parities][lower | upper side]
    char* hx_peer[2] = {nullptr, nullptr};   // the neighbours' stagings (lower, upper), mapped
    size_t hx_peer_par[2] = {0, 0}, hx_peer_off[2] = {0, 0}, hx_par = 0;
    long long hx_flag_off = 0, hx_epoch = 0, n_halo_pushes = 0;
    bool hx_ready = false;
    bool xs_enable = true;               // PSGSDF_XS=0: multi-rank scalar read-backs staged in the mailbox shadow and all-reduced over RCCL (round 3)
    long long xs_epoch = 0;              // scalar folds exchanged so far (the same on every rank)
    bool xf_enable = true;               // PSGSDF_XF=0: multi-rank frame rows through an RCCL all-reduce + solve kernels (round 3)
    XfTable* xf_table = nullptr;         // device copy of the exchange table (valid while xr_ready)
    long long xf_epoch = 0;              // exchanges so far (the same on every rank)
    size_t xr_doubles = 0;               // size of this rank's mailbox region
    bool any_empty_slab = false;         // some rank's slab has no observation at all (agreed at band build)
    bool xf_timeout = false;             // a frame row came back NaN: a rank's row never arrived
    bool img_compact = true;             // PSGSDF_IMG_COMPACT=0: float keyframes stay float even when every value is (float)byte / 255
    bool img_compacted = false;          // the float keyframes of this context are held as RGBA8 words
    bool speculate_mr = true;            // PSGSDF_SPECULATE_MR=0: no speculative start of the next iteration on multi-rank contexts (round 3's loop)
    int* vm_order = nullptr;             // distance sweep: physical workgroup -> logical block, heaviest first (build_band)
    int xcd_map = 163;                   // PSGSDF_XCD_MAP: logical workgroup ids -- bit 0 frame-major sweeps XCD-contiguous, bit 1 k_sweep_albedo / k_energy / k_derive XCD-contiguous, bit 2 k_sweep_dist XCD-contiguous (off: 68 -> 76 us), bit 5 (32) k_sweep_dist heaviest block first (70 -> 62 us), bit 6 (64) albedo / energy heaviest first (no gain), bit 7 (128) the per-pass distance solve k_cgf_pass XCD-contiguous (round 6); PSGSDF_XCD_STRIPE=T: stripes of T ids instead of eighths
    bool fm_solve = true;                // PSGSDF_FM_SOLVE=0: k_solve_light / k_solve_pose as kernels of their own behind the frame-major sweeps
    bool fm_solve_led = true;            // ... also the LED light vector (by the sweep's very last workgroup); PSGSDF_FM_SOLVE=2 keeps k_solve_light for it
    bool fm_solved = false;              // the sweep just launched solves its frames itself (step_begin -> step_finish)
    int frame_solve = 0;                 // PSGSDF_FRAME_SOLVE / psgsdf_set_frame_solver: 0 = LDL^T per frame in double (default), 1 = the reference's solver: ONE Eigen-style float Jacobi-PCG over the block-diagonal system of all frames (frame_solve.hip)
    double* fs_stats = nullptr;          // device [2][4]: {iterations, ||r||/||b||, Success, applied} of the last eigen light / pose solve
    double fs_last[2][4] = {{0, 0, 1, 1}, {0, 0, 1, 1}};   // host copy, refreshed by the synchronous steps (psgsdf_step, psgsdf_get_frame_solver_stats)
    bool albedo_applied = false;         // the last albedo sweep already applied its update (step_begin -> step_finish)
    unsigned* img8 = nullptr; float img_scale = 0.f;   // keyframes uploaded as 8-bit RGB (psgsdf_set_keyframes_u8): RGBA8 words, c->img stays null
    double* frame_e_slot = nullptr; unsigned long long frame_e_key = 0;   // mailbox slot (and its key) the next per-frame solve writes its sweep's energy sums to
    bool pcg_poll = true;                // PCG stop test by watching the mapped mailbox (PSGSDF_PCG_POLL=0: drain the stream instead)
    int need[2] = {0, 0}; int* d_need = nullptr;   // halo rows needed below row0 / from row1 up
    bool own_stream = true;
    // profiling
    bool profiling = false;
    std::map<std::string, KTime> ktimes;
    std::vector<const char*> kt_names;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // asynchronous watch of ONE kernel name: event pairs recorded on the launch stream, resolved on query
    std::string watch;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> watch_pool;
    size_t watch_used = 0; int watch_every = 1; size_t watch_seen = 0;
    std::vector<std::pair<std::string, std::string>> tuning_env, tuning_ignored;   // environment knobs as they stood when the context was created / dev-only ones this build ignores (psgsdf_get_tuning)
    char err[512] = {0};
};

namespace psge {
using namespace psg;

int fail(psgsdf_ctx* c, int code, const char* fmt, ...);
inline unsigned long long dbits(double v) { unsigned long long u; memcpy(&u, &v, 8); return u; }
#define HIPCHK(c, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(c, PSGSDF_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)

template <class Fn> void timed(psgsdf_ctx* c, const char* name, Fn&& fn) {
    if (!c->profiling) {
        if (!c->watch.empty() && c->watch == name && (c->watch_seen++ % c->watch_every) == 0) {   // a sample of the launches: the event pair costs ~3 us of stream time
            if (c->watch_used == c->watch_pool.size()) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); c->watch_pool.emplace_back(a, b);
            }
            auto& pr = c->watch_pool[c->watch_used++];
            hipEventRecord(pr.first, c->stream);
            fn();
            hipEventRecord(pr.second, c->stream);
            return;
        }
        fn(); return;
    }
    hipEventRecord(c->ev0, c->stream);
    fn();
    hipEventRecord(c->ev1, c->stream);
    hipEventSynchronize(c->ev1);
    float ms = 0; hipEventElapsedTime(&ms, c->ev0, c->ev1);
    KTime& k = c->ktimes[name]; k.ms += ms; k.n += 1;
}

// Bounded wait on a host-mapped slot a kernel publishes to (flush marker, PCG status): re-read the slot (~0.1 us per read;
// a hipStreamQuery costs the NEXT dispatch 5.8 us, profiles/r01_notes.md step p) and ask the runtime only every ~20 ms.
//   returns 0  : ready() became true
//           1  : the stream drained and ready() is still false (nothing is left that could publish)
//          <0  : the stream reported an error, or the wall-clock bound expired -> PSGSDF_ERR_DEVICE (never spins forever)
int wait_mapped(psgsdf_ctx* c, const std::function<bool()>& ready, const char* what);

// ---- comm.hip (all no-ops on a single-rank context; PSGSDF_ERR_COMM if a multi-rank context has no communicator)
int comm_unique_id(uint8_t id[128]);
int comm_create_rccl(psgsdf_ctx* c, const uint8_t id[128], int rank, int n);
int comm_create_ext(psgsdf_ctx* c, const psgsdf_comm_ops* ops, int rank, int n);
void comm_destroy(psgsdf_ctx* c);
int comm_allreduce(psgsdf_ctx* c, double* buf, int n);                 // in-place sum of n doubles (device), on the context's stream
int comm_halo(psgsdf_ctx* c, void* base, int planes, int width);       // halo rows of `planes` band planes of `width` 4-byte words per row
int comm_xfer(psgsdf_ctx* c, const std::vector<psgsdf_comm_xfer>& sends, const std::vector<psgsdf_comm_xfer>& recvs);   // grouped send / recv of device memory with any ranks
int host_allreduce(psgsdf_ctx* c, std::vector<double>& buf, const char* what);   // comm.hip: sum over the ranks of a HOST vector (set-up exchanges, the tracker's 6x6 system); synchronous
int xr_setup(psgsdf_ctx* c, const std::vector<double>& part_info);   // comm.hip: (re)build the cross-rank mappings for the band just built (part_info: {need_lo, need_hi, own rows} of every rank)
void xr_release(psgsdf_ctx* c);
int xr_quiesce(psgsdf_ctx* c, double timeout_s);                      // comm.hip: close this rank's mappings and wait (bounded, no collective) until every rank that mapped this rank's memory has closed its own; 1 = a peer never reported (do not free rec_mem / xr / hx_mem)
int xr_probe(psgsdf_ctx* c);                                          // comm.hip: once per context (collective): which memory kind carries the in-kernel hand-offs between the real neighbours
int xr_alloc(psgsdf_ctx* c, void** p, size_t bytes, bool polled);     // comm.hip: memory another device writes while a kernel of this one reads it
int mg_commit(psgsdf_ctx* c);                                          // engine.hip: all-reduce + deliver the staged scalar read-backs
int set_local_grid(psgsdf_ctx* c, int z0, int z1);                     // engine.hip: this context owns global planes [z0, z1) (+ halo planes)

int host_stage(psgsdf_ctx* c, size_t bytes, void** p);   // api.hip: engine-owned pinned host memory of at least `bytes` (grows; valid until the next call)
int frontend_normals_dev(psgsdf_ctx* c, const float* d_depth, int width, int height, float* d_normals);   // api_frontend.hip: FALS normals, device to device
// ---- engine.hip
SweepArgs make_args(psgsdf_ctx* c, int laplacian_reg);
// slab mode: the context is attached to a communicator (also a one-rank one: the exchanges then run as one-rank collectives, which is how
// the RCCL path is exercised on a one-GPU box and how its overhead is measured, bench.py --force-slab)
inline bool slab_mode(const psgsdf_ctx* c) { return c->comm != nullptr || c->n_ranks > 1; }
// the scalar folds exchange their sums between the ranks themselves (device_common.h fold_exchange): no staging, no all-reduce
inline bool xs_active(const psgsdf_ctx* c) { return c->n_ranks > 1 && c->xs_enable && c->xr_ready && c->xf_table != nullptr; }
inline int band_blocks(const psgsdf_ctx* c) { return (c->row1 - c->row0 + kBlock - 1) / kBlock; }
inline double band_mean(const psgsdf_ctx* c, double sum) { return c->S_global ? sum / (double)c->S_global : 0.0; }   // (1/S) sum over the band of the WHOLE volume
inline float total_energy(const psgsdf_ctx* c, float E, float E_n, float E_l, float E_r = 0.f) { return E + c->reg_n * E_n + c->reg_l * E_l + c->reg_r * E_r; }   // OptimizerAux.cpp:261
int flush(psgsdf_ctx* c);
int mbox_reserve(psgsdf_ctx* c, int n, size_t* off, unsigned long long* key);   // n values + n check words; flushes first if the mailbox is full
int deliver(psgsdf_ctx* c);              // validate and consume every deferred read-back (the caller knows their producers have run)
int deliver_first(psgsdf_ctx* c, size_t count, bool told_landed);   // ... the first `count` of them (told_landed: a marker / status word said they had arrived -- a wait is then counted as a late read-back)
bool readback_landed(const psge::Deferred& d);     // non-blocking: values and check words agree
int read_parts(psgsdf_ctx* c, const int* slots, int n, double* out);
int read_frame_energy(psgsdf_ctx* c, int col_e, double* E, double* nobs);
int read_parts_deferred(psgsdf_ctx* c, const int* slots, int n, std::function<void(const double*)> consume);
int read_frame_energy_deferred(psgsdf_ctx* c, int col_e, std::function<void(double, double)> consume);
int reserve_frame_energy_deferred(psgsdf_ctx* c, std::function<void(double, double)> consume, double** dev_slot, unsigned long long* key);
void materialize_fold(psgsdf_ctx* c);
void fold_by_kernel(psgsdf_ctx* c, FoldReq& f);   // the fold as a kernel of its own, now (a launch that was to take it did not happen / something consumes it first)
void take_fold(psgsdf_ctx* c, SweepArgs& a, unsigned writes);
int ensure_host_buf(psgsdf_ctx* c, size_t n);
void free_dense(psgsdf_ctx* c);
int alloc_dense(psgsdf_ctx* c, DenseView& d, long long nvox, int KW, bool with_rowof);
int build_band(psgsdf_ctx* c);
int derive(psgsdf_ctx* c, int update_grad);
int ps_energy(psgsdf_ctx* c, double* E, int64_t* nobs);

// ---- loop.hip
struct LoopState { float E, E_n, E_l, E_prev; int laplacian_reg; float E_r; };
void cgf_shape(int nblk, int* G, int* rows);
bool cgf_solve_shape(psgsdf_ctx* c, int* G, int* rows, bool any_ranks = false);      // can the whole solve run as ONE persistent kernel on this context? (any_ranks: the shape test alone, before the cross-rank mappings exist)
int pcg_solve(psgsdf_ctx* c, SweepArgs& a, int* iters_out, int* success_out, double* err_out,
              const std::function<void(const double*)>& tail = nullptr, bool gate_on_converged = true, bool* tail_ran = nullptr);
int albedo_reg_energy(psgsdf_ctx* c, double* Er);
int albedo_reg_solve(psgsdf_ctx* c, const SweepArgs& a, int* iters_out, int* ok_out, double* err_out);
int step_begin(psgsdf_ctx* c, int block, int laplacian_reg, psgsdf_step_stats* st, std::function<void(double, double)> deferred_consumer = nullptr, bool may_apply = true);
int step_finish(psgsdf_ctx* c, int block, int laplacian_reg, psgsdf_step_stats* st, bool defer_reg_sums = false);
int do_step(psgsdf_ctx* c, int block, int laplacian_reg, psgsdf_step_stats* st);
int read_frame_solver_stats(psgsdf_ctx* c, int kind, psgsdf_step_stats* st);   // kind 0 light, 1 pose; synchronises the stream
int run_loop(psgsdf_ctx* c, int flags, LoopState& L, int max_iters, bool full, psgsdf_iter_stats* stats, int stats_cap, int* n_done, int* result,
             psgsdf_iter_cb on_iter, void* user);
int do_upsample(psgsdf_ctx* c);

}  // namespace psge
