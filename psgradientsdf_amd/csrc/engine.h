// engine.h — data layout shared by the host orchestration (engine.hip, loop.hip, api*.hip; engine_internal.h) and the gfx950 kernels
// (band.hip, sweeps.hip, dist.hip, pcg.hip, albedo_reg.hip, frontend.hip).  Internal: the public boundary is include/psgsdf.h.
//
// Layout in HBM:
//   dense grid  : SoA planes dist | gx | gy | gz | weight | r | g | b (float, x-fastest) + packed
//                 visibility words + row_of[lin] (int32, -1 = not in band).  Source of truth for
//                 band construction, 2x refinement and download.
//   band        : the surface band (|d| <= sqrt(3) vs, seen >= 1; OptimizerAux.cpp:237-257)
//                 compacted in ascending linear index, every per-voxel quantity as its own
//                 padded plane so that lane i of a wavefront touches element i of each plane.
//   frames      : float32 RGB images, interleaved, resident for the whole optimisation;
//                 per-frame pose / light in a 96-byte record staged through LDS by every sweep.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psg {

constexpr int kBlock = 256;          // threads per workgroup (4 wavefronts of 64)
constexpr int kMaxFramesLds = 640;   // frame records staged in dynamic LDS: 96 B each, 60 KiB at the limit
constexpr int kNQ = 19;              // columns of one assembled distance-system row (ELL)
constexpr int kNQCommon = 13;        // self + 6 axis neighbours + 6 mixed-sign pairs: all a forward-only stencil touches
constexpr int kMaxBasis = 9;

struct FrameP {       // 24 floats = 96 B
    float R[9];       // camera->world rotation, row-major (Optimizer.h:52-55)
    float t[3];       // camera centre                      (Optimizer.h:57-60)
    float l[9];       // SH light of this frame; LED: global RGB intensity in l[0..2]
    float pad[3];
};

struct Cam { float fx, fy, cx, cy; int W, H; };

struct GridP {             // the grid THIS context holds: the whole volume, or (multi-rank) its z-slab plus one halo plane on each inner side
    int dim[3];
    long long nvox;
    float vs, vs_inv;
    float origin[3];       // world position of voxel (0, 0, 0) of the WHOLE volume (VoxelGrid.h:130)
    float T;
    int koff;              // global z index of local plane 0: world z of local voxel k is origin[2] + vs * (k + koff)
};

struct Robust { int loss; float lambda, lambda_sq, inv_lambda; };

// Band view: plain pointers into one big allocation; Spad = S rounded up to kBlock.
struct Band {
    int S, Spad, KW;
    int* lin;                 // [Spad] linear voxel index
    float* dist;              // [Spad]
    float* g[3];              // stored gradient (un-normalised), SdfVoxel::grad
    float* rho[3];            // albedo r,g,b
    uint64_t* vis;            // [KW][Spad] keyframe visibility words
    int* nb;                  // [6][Spad] band row of +x,-x,+y,-y,+z,-z neighbour or -1
    int* dirb;                // [Spad] bit k set: the +k neighbour is in the band, i.e. the stencil of axis k is forward (static)
    float* nbd;               // [6][Spad] distance of that neighbour when it is NOT in the band (static)
    int* col;                 // [kNQ][Spad] band row of each ELL column offset (absent: the row itself)
    unsigned* colp;           // [9][Spad] the same as 16-bit deltas col - row, columns (2w+1, 2w+2) in word w: half the index bytes of a PCG pass
    int col16;                // 1 if every |col - row| fits 16 bits (then the PCG reads colp instead of col)
    int reach;                // max |col - row| over the band: how far (in rows) a row's ELL columns reach
    // derived per voxel, refreshed whenever dist / grad change (k_derive)
    float* gfd[3];            // finite-difference gradient (Optimizer.cpp:287-364), un-normalised
    float4* vp[3];            // derived state, packed because the frame-major sweeps GATHER it per observation and are bound by L1 line
                              // look-ups (3 16-byte gathers instead of 12 4-byte ones): {xs, rho.r}, {gn, rho.g}, {nfd, rho.b} with
                              // xs = x_v - d*normalized(grad) (OptimizerAux.cpp:215), gn = normalized(stored grad), nfd = normalized(gfd)
                              // (the normal every residual is rendered with); .w mirrors rho (set_rho)
    // albedo diagonal system
    float* aH; float* ab;     // [3][Spad]
    // distance system: per-voxel 4x4 block (10 sym) + 4 rhs, then assembled ELL rows
    float* blk;               // [14][Spad]
    float* H;                 // [kNQ][Spad]
    int* hx;                  // [Spad] 1 if any of the 6 rare columns of this row is non-zero
    float* rhs; float* x; float* t;
    float4* rec[2];           // [Spad] PCG records {r, t, p, inv} of the previous pass, double-buffered: ONE 16-byte gather per column
    // per-frame observation lists (static per band): rows visible in frame f, ascending
    int* obs_ptr;             // [F+1]
    int* obs_rows;            // [obs_ptr[F]]
    int obs_max;              // longest list
    long long obs_ptr_total;  // obs_ptr[F]: all observations of the owned rows
};

// accumulators for the per-frame normal equations and the scalar reductions (double)
struct Accum {
    double* frame;            // [F][kFrameRow] per-frame rows: H (upper triangle) | rhs | energy | n_obs
    double* part;             // [SC_COUNT][PB] per-workgroup partials of the scalar reductions
    int PB;                   // capacity (workgroups) of one partial slot
    // frame-major sweeps: every workgroup stores ITS partial row (no floating-point atomics) and the last workgroup of a frame to arrive
    // sums them in launch order into `frame` (sweeps.hip: frame_rows_publish): the rows are run-to-run reproducible
    double* fpart;            // [F][fcap][kFrameRow]
    int fcap;                 // capacity (workgroups per frame)
    int* fdone;               // [F] arrival counters, zero outside a sweep
};
constexpr int kFrameRow = 64;        // doubles per frame accumulator row (SH2: 45 + 9 + 2 = 56)
constexpr int kPcgMaxBlocks = 2048;  // workgroups of one PCG pass (grid-stride)
constexpr int kCgfMaxBlocks = 768;   // workgroups of one fused PCG pass: at most 3 partials per thread and sum (pcg.hip)

enum { SC_ENERGY = 0, SC_NOBS = 1, SC_EN = 2, SC_EL = 3, SC_ACCEPT = 4, SC_AUX0 = 5, SC_AUX1 = 6, SC_AUX2 = 7, SC_COUNT = 8 };

// "reg albedo" (||grad rho_c|| regulariser, Optimizer.cpp:221-245,396-460,593-647): allocated only when the weight is non-zero
struct AlbedoReg {
    int* anb;        // [3][Spad] band row of the stencil neighbour of each axis (forward iff in band), -1 = outside the band
    int* back;       // [Spad] bit a set: axis a uses the backward neighbour
    float* anrho;    // [9][Spad] (axis, channel) static albedo of a stencil neighbour outside the band
    float* J;        // [12][Spad] (slot, channel): d ||grad rho_c|| / d rho_c(slot), slot 0 = the voxel, 1..3 = x/y/z neighbour
    float* res;      // [3][Spad] ||grad rho_c||
    float* rhs; float* diag; float* diag0; float* x; float* r; float* p; float* q; float* t;   // CG over the 3S unknowns, [3][Spad] each (diag0 = undamped diagonal)
    double* cgs;     // device-driven CG (single rank, loop.hip albedo_reg_solve): [0] 1 = converged, [1] iterations done, [2] |r|^2, [3 + parity] r.z of the last update
    float weight;
};

// a pending fold of per-workgroup partials (previous kernel's scalar reductions) that the NEXT kernel performs in its
// first workgroup instead of a 1-workgroup kernel of its own (saves a launch + boundary per scalar read-back)
// `key` != 0: `out` is a slot of the host-mapped mailbox and every value is followed (n doubles further on) by its check word
// bits(value) ^ (key + index): the host accepts a read-back only when the pair matches (engine.hip: deliver), whatever order the two
// words reach host memory in and however they are ordered against the marker / status word it was told to wait for.
struct XfTable;
// xf != nullptr (multi-rank): the folded sums are this SLAB's; the folding thread exchanges them with the other ranks through the mailbox regions
// (device_common.h fold_exchange) and delivers the sums over all slabs -- no all-reduce, no staging: the read-back is global at its source.
struct FoldReq { int n; int id[4]; int nblk; double* out; unsigned long long key; const XfTable* xf; long long xf_epoch; };

// keyframe images: float RGB [F][H][W][3] (psgsdf_set_keyframes) or RGBA8 words [F][H][W] (psgsdf_set_keyframes_u8); exactly one is set
struct ImgSrc {
    const float* f32;
    const unsigned* u8; float scale;   // colour = (float)byte * scale
    bool idx32;                        // f32: the stack is < 4 GiB, tap offsets fit 32 bits (device_common.h: sample)
};

struct SweepArgs {
    Band b;
    const FrameP* frames;     // [F]
    ImgSrc im;
    int F;
    Cam cam;
    GridP grid;
    Robust rob;
    Accum acc;
    int model;                // 0 SH1, 1 SH2, 2 LED
    int quirks;
    float reg_n, reg_l;
    int normal_reg, laplacian_reg;
    float damping;
    int row0, row1;           // band rows this context owns: [row0, row1) (whole band on one GPU; a z-slab per rank otherwise)
    FoldReq fold;             // n = 0: nothing pending
    AlbedoReg ar;             // ar.anb == nullptr unless "reg albedo" != 0
    double* pcg_part; double* pcg_fs;   // fused PCG state (pcg.hip)
    int pcg_asm;              // persistent solve assembles the distance system itself (no k_assemble launch in front of it)
    unsigned pcg_epoch;       // serial number of this context's distance solve (k_cgp_solve<.., TM>: the tag of its exchanged values; the launch sites count it)
    int pcg_pipe;             // ... and runs the pipelined recurrences (pcg.hip k_cgp_solve: the sums of a pass travel while the next pass gathers)
    int pcg_xcd_local;        // ... and keeps the records of workgroups whose neighbours all run on their own XCD in that XCD's L2 (plain stores)
    int pcg_apply;            // ... and applies the distance update itself (no k_apply_dist behind it): 1 = when finished, 2 = only on Success
    double* pcg_gran; int pcg_gran_n;   // persistent solve: the tagged per-workgroup sums, zeroed by the assembly kernel when non-null
    int pcg_fuse_init;        // assembly kernel also initialises the PCG (x = 0, records of pass -1, |b|^2 partials): no k_cgf_init launch
    int pcg_init_blocks;      // workgroups that wrote the |b|^2 partials (0: the pass kernel's own grid)
    int fuse_apply;           // albedo sweep: solve the voxel's diagonal system and apply the update in the same thread (no normal equations stored); 2: and keep the old albedo for an undo
    const int* vm_order;      // voxel-major dispatch order of the distance sweep: physical workgroup -> logical block, heaviest first (nullptr: identity)
    float* fm_led_light;      // fm_solve, LED: the light vector [3] next to the frames' copies (updated by the sweep's last workgroup)
    const XfTable* xf;        // multi-rank: the frame rows travel through the ranks' mailbox regions (nullptr: single rank / RCCL all-reduce path)
    long long xf_epoch;       // number of this exchange (flag value; its parity selects the buffer)
    int xcd_map;              // workgroup -> work mapping that gives every XCD (physical workgroup id mod 8) one contiguous range of band rows / observation chunks: its L2 then holds an eighth of the band (device_common.h xcd_remap)
    int fm_solve;             // frame-major sweeps: the last workgroup of a frame solves the frame's light (SH) / pose block, the last frame sums the energy columns (sweeps.hip frame_rows_publish): no solve launch
    FrameP* fm_frames; float* fm_undo; double* fm_e_out; unsigned long long fm_e_key;
    const double* gate;       // speculative launch: the kernel does nothing unless *gate != 0 (nullptr = always run); see pcg_solve
    const double* ext;        // multi-rank PCG: the 7 globally reduced sums of the previous pass (|b|^2 in ext[0] for pass 0), else nullptr
};

// Cross-rank hand-offs of the persistent distance solve on a z-slab partition (pcg.hip k_cgf_solve<.., MR = true>, comm.hip xr_setup): every rank owns one
// small mailbox region that ALL ranks can write (IPC-mapped over xGMI), and the records of the rows next to a cut are written straight into the
// neighbour's halo rows.  Both live in memory another device may write and this device may poll WHILE A KERNEL RUNS: the region is allocated
// uncached, the two record planes fine-grained (comm.hip xr_alloc; plain hipMalloc memory is only coherent across devices at kernel boundaries),
// and the hand-off forms are verified between the real neighbours before they are relied on (comm.hip xr_probe).
// Nothing in a region is ever cleared between solves: every tag carries the solve's EPOCH next to the pass tag, so a word of an earlier solve -- or a
// word a fast rank wrote before this rank even started its solve -- can neither be mistaken for the current one nor be wiped.
// Layout of a region, in doubles:
constexpr int kXrMaxRanks = 32;      // rank-level sums: [2 buffers][8 planes][kXrMaxRanks] tagged granules (plane 7: |b|^2 with the sums of pass 0)
constexpr int kXrPeerTags = 64;      // 'records are out' tags of the neighbour's workgroups next to the cut: [2 sides][3 buffers (2 pass parities + prologue)][kXrPeerTags]
constexpr int kXrRankGran = 0, kXrPtag = 2 * 8 * kXrMaxRanks, kXrAbort = kXrPtag + 2 * 3 * kXrPeerTags, kXrDoubles = kXrAbort + 8;
// kXrAbort + 0: the persistent solve's abort flag (raised in every rank's region; cleared with the region at the next band).  kXrLate + 0 / 1 / 2: the
// number of the last frame-row exchange / scalar fold / halo pull of THIS rank whose bounded wait for a peer expired (0: none) -- the NaN such an
// exchange hands on says "something is wrong", these words say what (lateness signalled out of band, not through the payload alone)
constexpr int kXrLate = kXrAbort + 1;
constexpr int kXrNonce = kXrAbort + 7;   // stamped by the owner at every set-up, read back by every peer through its mapping (comm.hip xr_setup)
struct XrArgs {
    int rank, n_ranks;               // n_ranks == 0: single-rank solve (every field below unused)
    double* region[kXrMaxRanks];     // every rank's mailbox region (own included)
    float4* lo_rec[2]; float4* hi_rec[2];   // where this slab's boundary records go: first upper-halo row of the lower neighbour / first lower-halo row of the upper one (per record buffer)
    float4* lo_base[2];                     // row 0 of the lower neighbour's record planes (k_cgp_solve stores ONE double per row at the head of a plane: row index counted in doubles)
    int give_lo, give_hi;            // own rows the lower / upper neighbour holds as halo (the first give_lo / last give_hi own rows)
    int wait_lo, wait_hi;            // workgroups of the lower / upper neighbour whose tags this slab's cut-side workgroups wait for
    int need_lo, need_hi;            // halo rows this slab gathers from
    unsigned epoch;                  // solve counter of the context (the same on every rank), 14 bits: tags of the cross-rank words are (epoch << 2 | pass tag)
};
// Cross-rank exchange of the frame-major sweeps' per-frame rows (sweeps.hip frame_rows_publish, multi-rank): behind the fixed words of a rank's
// mailbox region sit, for two alternating sweeps, R x F payload rows of kFrameRow doubles and R x F flags.  The last workgroup of frame f on rank r
// stores ITS slab's final row into [buf][r][f] of EVERY rank's region (system-scope write-through stores), drains them, then sets the flags to the
// sweep's number; it then waits for the R flags of frame f in its OWN region and adds the R rows in rank order -- every rank holds the same global
// row without a collective, and solves the frame on the spot as a single-rank context does.
struct XfTable {
    int n_ranks, rank, F, spin_max;  // spin_max: polls a wait inside a kernel may take before the peer counts as lost (2^24; PSGSDF_XWAIT_LOG2 for the failure tests)
    long long pay, flg;              // offsets (doubles) of the payload rows / the flags inside a region
    long long spay, sflg;            // scalar folds (FoldReq::xf): [2 buffers][R ranks][8 values] and [2][R] flags
    double* region[32];              // (= kXrMaxRanks) every rank's mailbox region, own included
};
constexpr int kHxWords = 16;            // 4-byte words per row a halo push carries at most (planes x width of comm_halo: 14 voxel-block planes, 15 of the albedo regulariser, one float4 record)
constexpr unsigned kXrEpochMask = 0x3fffu;      // (the host clears every region behind an all-rank barrier whenever the epoch wraps: loop.hip pcg_solve)

// ---- launchers implemented in the kernel files (all asynchronous on `s`) ----------------------
constexpr int kTryPackFlags = 2048;   // workgroups of k_try_pack_f32 = words of its flag array (one per workgroup: no same-address stores)
void launch_normals_cache(int W, int H, int r, double fx_inv, double fy_inv, double cx, double cy, double* work /*12 planes of W*H doubles*/, float* out /*9 float planes*/, hipStream_t s);   // NormalEstimator::cache on the device
void launch_try_pack_f32(const float* rgb, unsigned* rgba, size_t npix, size_t stride, float scale, int* fail /*[kTryPackFlags], zeroed*/, hipStream_t s);   // float RGB that is exactly (float)byte * scale -> RGBA8 words; *fail != 0 otherwise
void launch_pack_rgb8(const uint8_t* rgb, unsigned* rgba, size_t npix, hipStream_t s);   // [npix][3] bytes -> one RGBA8 word per pixel
void launch_select_vis(const uint64_t* vis_seq, int wpv_seq, uint64_t* vis_key, int KW, const int* frame_idx, int F, long long nvox, hipStream_t s);
void launch_band_flags(const float* dist, const uint64_t* vis_key, int KW, float vs, long long nvox, int* flags, hipStream_t s);
// exclusive scan of flags -> row_of (-1 where flag==0); returns total through d_total (device int)
void launch_band_scan(int* flags_inout_rowof, long long nvox, int* block_sums, int* d_total, hipStream_t s);
struct DenseView { float* dist; float* g[3]; float* weight; float* rho[3]; uint64_t* vis; int KW; int* row_of; };
void launch_band_fill(const DenseView& d, const GridP& grid, Band b, int* d_reach, hipStream_t s);   // *d_reach = max |col - row| (atomicMax; zero it first)
void launch_band_scatter(const DenseView& d, Band b, hipStream_t s);
void launch_derive(const SweepArgs& a, int update_grad, hipStream_t s);
constexpr int kObsChunk = 2048;      // rows per workgroup of the observation-list builders
void launch_obs_count(const Band& b, int F, int row0, int row1, int* counts, hipStream_t s);       // counts[F][nch]
void launch_block_work(const Band& b, int row0, int row1, int* work, hipStream_t s);   // work[block of kBlock owned rows] = sum over its wavefronts of the busiest lane's visible frames
void launch_obs_fill(const Band& b, int F, int row0, int row1, const int* offsets, hipStream_t s); // offsets[F][nch] -> b.obs_rows
void launch_lower_bound(const int* lin, int S, int target, int* out, hipStream_t s);               // *out = number of band rows with lin < target (lin ascending)
struct SlotList { int n; int id[8]; };
void launch_sum_parts(const double* part, int PB, int nblk, const SlotList& slots, double* out, unsigned long long key, hipStream_t s, const XfTable* xf = nullptr, long long xf_epoch = 0);   // key: FoldReq
void launch_frame_cols(const double* frame, int F, int col, double* out, unsigned long long key, hipStream_t s);
void launch_zero_f64(double* p, int n, hipStream_t s);
struct CopySegs { int n; unsigned off[16]; unsigned len[16]; unsigned long long key[16]; };
void launch_copy_segs(const double* src, double* dst, const CopySegs& segs, hipStream_t s);   // dst[off + i] = src[off + i] for every segment, + the check words (FoldReq) behind each segment
void launch_marker(double* p, double v, hipStream_t s);
void launch_init_albedo(const SweepArgs& a, hipStream_t s);
void launch_led_light_init(const SweepArgs& a, hipStream_t s);
void launch_energy(const SweepArgs& a, hipStream_t s);
void launch_sweep_albedo(const SweepArgs& a, hipStream_t s);
void launch_apply_albedo(const SweepArgs& a, hipStream_t s);
void launch_restore_albedo(const SweepArgs& a, hipStream_t s);   // undo of a speculative fused albedo update (fuse_apply == 2)
int launch_sweep_light(const SweepArgs& a, hipStream_t s);    // both return the workgroups per frame they used (0: nothing was launched, the rows are untouched)
int launch_sweep_pose(const SweepArgs& a, hipStream_t s);
void launch_solve_light(const SweepArgs& a, FrameP* frames, float* led_light, double* e_out, unsigned long long e_key, float* undo, hipStream_t s);
void launch_restore_light(int F, FrameP* frames, float* led_light, const float* undo, hipStream_t s);   // undo of a speculative light update (undo: [F][9] + [3] floats)   // also sums the energy columns -> e_out (nullable; e_key: FoldReq)
void launch_solve_pose(const SweepArgs& a, FrameP* frames, double* e_out, unsigned long long e_key, hipStream_t s);
// the reference's own solver for the light / pose blocks (frame_solve.hip: ONE Eigen-style Jacobi-PCG over the block-diagonal system of all frames; PSGSDF_FRAME_SOLVE=eigen).
// stats (nullable, device): {iterations, ||r|| / ||b||, info() == Success, update applied}
bool frames_eigen_fits(int model, int F);
void launch_frames_eigen_light(const SweepArgs& a, FrameP* frames, float* led_light, double* e_out, unsigned long long e_key, float* undo, double* stats, hipStream_t s);
void launch_frames_eigen_pose(const SweepArgs& a, FrameP* frames, double* e_out, unsigned long long e_key, double* stats, hipStream_t s);
int launch_frames_eigen_raw(int nb, int n, const float* H, const float* b, float* x, double* stats, int max_it, hipStream_t s);   // nb blocks of n x n floats, n in {3, 4, 6, 9}; -1: unsupported shape
void launch_sweep_dist(const SweepArgs& a, hipStream_t s);
void launch_assemble(const SweepArgs& a, hipStream_t s);
void launch_cgf_init(const SweepArgs& a, double* fs, double* part, int G, hipStream_t s);
void launch_cgf_pass(const SweepArgs& a, double* fs, double* part, int G, int rows, int k, int kmax, double* mb, hipStream_t s, int ablate = 0);
void launch_cgf_sum(double* part, int G, int k, double* out, hipStream_t s);
// the whole solve as one persistent kernel (pcg.hip: k_cgf_solve); gran = [2][kSolveGranPlanes][kSolveMaxBlocksHost] tagged per-workgroup sums (plane 7: |b|^2 of
// the fused assembly in buffer 0, its 'records are out' flag in buffer 1), zeroed by the kernel in front (assembly kernel / distance sweep)
constexpr int kSolveGranPlanes = 8;
constexpr int kSolveThreadsHost = 512, kSolveMaxBlocksHost = 256, kSolveMaxRowsHost = 4, kSolveMbSlots = 24;      // {iters, |r|^2, |b|^2, status} + stage timestamps of the timing hook
int cgf_solve_max_blocks(int rows);      // resident workgroups per CU of the R-rows instance (occupancy query)
void launch_cgf_solve(const SweepArgs& a, double* fs, double* gran, int G, int rows_per_wg, int kmax, double* mb, unsigned long long mb_key, int force_passes, hipStream_t s, const XrArgs* xr = nullptr);   // xr: multi-rank (z-slab) solve   // mb[4] = check word over mb[0..3] (FoldReq)
// "reg albedo" path (albedo_reg.hip); every launch covers the whole band (single rank only)
void launch_areg_tables(const DenseView& d, const GridP& g, const SweepArgs& a, hipStream_t s);
void launch_areg_build(const SweepArgs& a, hipStream_t s);                       // J, res from the current albedo; sum of res -> SC_AUX0
void launch_areg_system(const SweepArgs& a, hipStream_t s);                      // rhs, diag (damped) from aH/ab + weight Jr^T(.)
void launch_areg_jx(const SweepArgs& a, const float* p, float* t, hipStream_t s);               // t = Jr p
void launch_areg_jt(const SweepArgs& a, const float* p, const float* t, float* q, hipStream_t s);   // q = (H_d + damping diag) p + weight Jr^T t; p.q -> SC_AUX0
void launch_areg_cg_init(const SweepArgs& a, hipStream_t s);                     // x = 0, r = rhs, p = r/diag; |b|^2 -> SC_AUX0, r.z -> SC_AUX1
// device-driven form (round 6): iteration i of Eigen's CG as four launches whose scalars (alpha, beta, the stop test) every workgroup derives from the
// previous kernel's per-workgroup partials and ar.cgs -- no read-back inside a chunk of iterations; a converged solve turns the rest of the chunk into no-ops
void launch_areg_cg_iteration(const SweepArgs& a, int i, int nblk, float thr, hipStream_t s);
void launch_areg_cg_update(const SweepArgs& a, float alpha, hipStream_t s);      // x += alpha p, r -= alpha q; |r|^2 -> SC_AUX0, r.z -> SC_AUX1
void launch_areg_cg_dir(const SweepArgs& a, float beta, hipStream_t s);          // p = r/diag + beta p
void launch_apply_albedo_delta(const SweepArgs& a, const float* delta, hipStream_t s);
void launch_matvec(const SweepArgs& a, const float* x, float* y, hipStream_t s);   // debug: y = H x (no damping)
void launch_apply_dist(const SweepArgs& a, hipStream_t s);
void launch_upsample(const DenseView& src, const DenseView& dst, const GridP& g_old, hipStream_t s);
void launch_fill_f32(float* p, float v, long long n, hipStream_t s);
void launch_normals(const float* depth, const float* cache, int W, int H, int r, double* tmp, float* out, hipStream_t s);
void launch_track(const DenseView& d, const GridP& g, const Cam& cam, const FrameP& fp, const float* depth, float z_min, float z_max, double* part, int nblk, int gdimz, int zown0, int zown1, hipStream_t s);
void launch_integrate(const DenseView& d, uint64_t* vis_seq, int wpv_seq, const GridP& g, const Cam& cam, const FrameP& fp,
                      const float* rgb, const float* depth, const float* normals, int counter, float z_min, float z_max, hipStream_t s);


}  // namespace psg
