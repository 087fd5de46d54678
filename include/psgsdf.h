/* psgsdf.h — C ABI of the MI355X-native Gradient-SDF photometric-stereo engine.
 *
 * This header is the drop-in boundary for the reference's photometric-stereo hot path.
 * The reference (Sangluisme/PSgradientSDF) has no FFI of its own: the seam is the C++
 * class interface that `main()` drives (cpp/voxel_ps/src/main_ps.cpp:193-202,323-330):
 *
 *     vOpt = new PsOptimizer|LedOptimizer(tSDF, voxel_size, K, output, opt_set_);
 *     vOpt->setImages(..); vOpt->setKeyframes(..); vOpt->setKeytimestamps(..); vOpt->setPoses(..);
 *     vOpt->init();  vOpt->alternatingOptimize(light, albedo, distance, pose);
 *
 * Every entry point below names the reference member it replaces (file:line under
 * /root/reference/cpp/include/).  All pointers are plain host pointers unless a name ends
 * in `_dev`; sizes are element counts; there are no C++/torch types in any signature.
 *
 * Conventions
 *   - every function returns 0 on success, a negative psgsdf_status otherwise and never throws;
 *     psgsdf_last_error(ctx) gives the message of the last failure on that context.
 *   - volume arrays are x-fastest (lin = i + j*Nx + k*Nx*Ny, VoxelGrid.h:79-82), SoA planes.
 *   - images are float32 RGB (not OpenCV BGR), row-major H x W x 3, values in [0,1].
 *   - poses are 4x4 row-major camera->world (Optimizer.h:52-60).
 *   - a context is owned by one host thread; it is not thread-safe (the reference is
 *     single-threaded and not re-entrant either).
 */
#ifndef PSGSDF_H_
#define PSGSDF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psgsdf_ctx psgsdf_ctx;

enum psgsdf_status {
    PSGSDF_OK = 0,
    PSGSDF_ERR_ARG = -1,        /* bad argument / call order                               */
    PSGSDF_ERR_DEVICE = -2,     /* HIP runtime error (no device, OOM, launch failure)       */
    PSGSDF_ERR_UNSUPPORTED = -3,/* a call the engine does not implement in this mode (e.g. frame fusion on a multi-rank context) */
    PSGSDF_ERR_STATE = -4,      /* called before the required earlier call                  */
    PSGSDF_ERR_COMM = -5        /* RCCL failure / no communicator on a multi-rank context    */
};

/* ModelType, OptimizerSettings.h:18-22 */
enum psgsdf_model { PSGSDF_SH1 = 0, PSGSDF_SH2 = 1, PSGSDF_LED = 2 };
/* LossFunction, OptimizerSettings.h:9-16 (same numeric values) */
enum psgsdf_loss { PSGSDF_L2 = 0, PSGSDF_CAUCHY = 1, PSGSDF_HUBER = 2, PSGSDF_TUKEY = 3, PSGSDF_TRUNC_L2 = 4 };
/* the four blocks of alternatingOptimize(light, albedo, distance, pose), Optimizer.h:178 */
enum psgsdf_block { PSGSDF_ALBEDO = 1, PSGSDF_LIGHT = 2, PSGSDF_DIST = 4, PSGSDF_POSE = 8, PSGSDF_ALL = 15 };

/* VoxelGrid(grid_dim, voxel_size, shift) + Sdf(T): VoxelGrid.h:127-133, Sdf.h:81-85.
 * origin = shift - 0.5*voxel_size*dim is derived inside (VoxelGrid.h:130). */
typedef struct psgsdf_grid_desc {
    int32_t dim[3];
    float voxel_size;
    float shift[3];      /* grid centre (the depth centroid in main_ps.cpp:178-183) */
    float truncation;    /* T = truncation_factor * voxel_size (main_ps.cpp:86)      */
} psgsdf_grid_desc;

/* OptimizerSettings (OptimizerSettings.h:24-51) + the engine-only knobs at the end. */
typedef struct psgsdf_settings {
    int32_t model;          /* psgsdf_model                                                   */
    int32_t loss;           /* psgsdf_loss                                                    */
    float lambda;           /* robust-loss scale                                              */
    float damping;          /* LM damping: H_ii *= (1+damping)                                */
    float reg_weight_rho;   /* "reg albedo" */
    float reg_weight_n;     /* "reg norm"     Eikonal weight                                  */
    float reg_weight_l;     /* "reg laplacian"                                                */
    int32_t max_it;         /* "max iter"                                                     */
    float conv_threshold;   /* "converge threshold"                                           */
    int32_t upsample;       /* "upsample": 2x refine at iteration 5                           */
    /* engine-only */
    int32_t ref_quirks;     /* 1 = replicate reference quirks (B6 LED sign,
                               B8 CG-gated updates); 0 = corrected variants                  */
    int32_t cg_max_it;      /* cap for the distance PCG; <=0 = Eigen's default 2*n            */
} psgsdf_settings;

/* per-sub-step statistics returned by psgsdf_step */
typedef struct psgsdf_step_stats {
    int32_t block;          /* psgsdf_block that ran                                          */
    int32_t cg_iters;       /* PCG iterations of the distance solve (0 for direct solves)     */
    int32_t cg_converged;   /* Eigen's info()==Success                                        */
    int32_t applied;        /* 1 if the update was applied (B8 gating)                        */
    double  e_in;           /* PS energy of the state the sweep started from                  */
    double  cg_error;       /* final ||r||/||b||                                              */
    int64_t n_accepted;     /* accepted per-voxel updates (OptimizerAux.cpp:149,183)          */
    int64_t n_obs;          /* visible, in-image observations seen by the sweep               */
} psgsdf_step_stats;

/* one record per Gauss-Newton iteration (the body of the while loop, PsOptimizer.cpp:303-425) */
typedef struct psgsdf_iter_stats {
    double e_after[4];      /* PS energy after albedo / light / dist / pose (NaN if block off) */
    double e_n, e_l;        /* un-weighted Eikonal / Laplacian energies after the dist step    */
    double e_total;         /* getTotalEnergy at the end of the iteration                      */
    double rel_diff;        /* |E_prev - E_total| / E_prev                                     */
    float  reg_weight_n;    /* effective (normalised) weights in force, PsOptimizer.cpp:277,283 */
    float  reg_weight_l;
    int32_t cg_iters;
    int32_t converged;      /* rel_diff < conv_threshold                                       */
    int32_t diverged;       /* E_total > E_prev                                                */
    int32_t upsampled;      /* this iteration ended with the 2x refine                         */
    double e_r;             /* un-weighted albedo-gradient energy (Optimizer.cpp:122-136) after the albedo step; 0 unless "reg albedo" */
    double e_n_in, e_l_in;  /* e_n / e_l as they stood BEFORE this iteration's dist step: what the reference's log lines of the blocks in front of it
                             * add to the PS energy (getTotalEnergy(E, E_n, E_l, E_r), PsOptimizer.cpp:311-331 -- E_n changes at :342 only) */
} psgsdf_iter_stats;

/* sizes the caller needs for downloads */
typedef struct psgsdf_info {
    int32_t dim[3];
    float voxel_size;
    float origin[3];
    int32_t n_frames;
    int32_t n_band;         /* |surface_points_|                                               */
    int32_t light_stride;   /* 4 (SH1), 9 (SH2), 3 (LED: one global RGB vector, n=1)           */
    int32_t vis_words;      /* 64-bit words of keyframe visibility per voxel                   */
    float reg_weight_n, reg_weight_l;   /* current effective weights                          */
} psgsdf_info;

/* ---- lifetime -------------------------------------------------------------------------- */

/* Replaces the PsOptimizer / LedOptimizer constructor (PsOptimizer.cpp:15-23,
 * LedOptimizer.cpp:15-23, Optimizer.cpp:16-27).  K is the 3x3 row-major intrinsic matrix
 * (only fx,fy,cx,cy are read, like OptimizerAux.cpp:209-212).  `device` is the HIP device
 * ordinal this context lives on. */
int psgsdf_create(const psgsdf_grid_desc* grid, const float K[9], const psgsdf_settings* settings,
                  int device, psgsdf_ctx** out);
void psgsdf_destroy(psgsdf_ctx* ctx);
const char* psgsdf_last_error(const psgsdf_ctx* ctx);
/* version string "psgsdf-hip gfx950 <build tag>" */
const char* psgsdf_version(void);

/* ---- inputs ---------------------------------------------------------------------------- */

/* Replaces the friend access to VolumetricGradSdf::tsdf_ / vis_ (VolumetricGradSdf.h:25-42):
 * hands the fused voxel state to the engine.  dist/weight: N^3; grad_xyz, rgb: 3 planes of N^3
 * (x|y|z, r|g|b); vis_words: N^3 * words_per_voxel 64-bit words, bit c of a voxel = "seen by
 * integrated frame c" (the per-voxel std::vector<bool>, VolumetricGradSdf.cpp:129-130). */
int psgsdf_upload_volume(psgsdf_ctx* ctx, const float* dist, const float* grad_xyz,
                         const float* weight, const float* rgb,
                         const uint64_t* vis_words, int words_per_voxel);

/* Replaces setImages + setKeyframes + setPoses (Optimizer.h:137-154).  frame_idx[f] is the
 * integration counter of keyframe f (selects the visibility bit, Optimizer.cpp:30-47). */
int psgsdf_set_keyframes(psgsdf_ctx* ctx, int n_frames, const int32_t* frame_idx,
                         const float* rgb_images, int width, int height, const float* poses);
/* The same with one pointer per image (n_frames pointers to width * height * 3 floats each): the reference keeps its keyframes as a std::vector<cv::Mat>,
 * one allocation per image (Optimizer.h:137-140) -- no gather into one array on the host. */
int psgsdf_set_keyframes_frames(psgsdf_ctx* ctx, int n_frames, const int32_t* frame_idx, const float* const* rgb_images, int width, int height, const float* poses);

/* The same with the keyframes as the reference's loader receives them: 8-bit interleaved RGB [F][H][W][3] plus the factor
 * of its conversion (ImageLoader.h:167-181: cv::imread, then convertTo(CV_32FC3, 1.0f / 255.0f)).  The engine samples
 * colour = (float)byte * scale, bit for bit what that conversion stores, from one RGBA8 word per pixel: the same results as
 * psgsdf_set_keyframes on the converted images with a quarter of the upload, a third of the resident image bytes and ~3 % more
 * iterations per second (profiles/r01_notes.md, step p). */
int psgsdf_set_keyframes_u8(psgsdf_ctx* ctx, int n_frames, const int32_t* frame_idx,
                            const uint8_t* rgb_images, float scale, int width, int height, const float* poses);

/* Replaces PsOptimizer::init / LedOptimizer::init (PsOptimizer.cpp:25-42,
 * LedOptimizer.cpp:25-36): select_vis, getSurfaceVoxel, light initialisation. */
int psgsdf_init(psgsdf_ctx* ctx);

/* ---- the hot path ---------------------------------------------------------------------- */

/* Optimizer::initAlbedo (Optimizer.cpp:50-81). */
int psgsdf_init_albedo(psgsdf_ctx* ctx);

/* getPSEnergy / getNormalEnergy / getLaplacianEnergy / getTotalEnergy
 * (PsOptimizer.cpp:47-78, Optimizer.cpp:86-119, OptimizerAux.cpp:259-269):
 * out = { E_ps, E_n, E_l, E_total } with the current effective weights. */
int psgsdf_energy(psgsdf_ctx* ctx, double out[4]);

/* The weight normalisation that opens alternatingOptimize (PsOptimizer.cpp:274-285):
 * reg_weight_n *= E/E_n, reg_weight_l *= E/E_l.  Returns E_total in *e_total. */
int psgsdf_normalize_weights(psgsdf_ctx* ctx, double* e_total);

/* One sub-step: optimize{Albedo,Light,Dist,Poses}All (PsOptimizer.cpp:85-234,
 * LedOptimizer.cpp:134-275).  `block` is one psgsdf_block value. */
int psgsdf_step(psgsdf_ctx* ctx, int block, psgsdf_step_stats* stats);

/* n_iters bodies of the alternation loop (PsOptimizer.cpp:303-366 / LedOptimizer.cpp:343-409)
 * with the blocks in `flags` enabled, energies and the convergence / divergence tests
 * evaluated for every iteration but never acted upon (no early exit, no file dumps, no
 * upsampling).  stats may be NULL, else n_iters records. */
int psgsdf_iterate(psgsdf_ctx* ctx, int flags, int n_iters, psgsdf_iter_stats* stats);

/* The full alternatingOptimize control flow (PsOptimizer.cpp:239-428,
 * LedOptimizer.cpp:279-478) minus file output: initAlbedo, weight normalisation, loop with
 * convergence / divergence exit, upsample at iteration 5, Laplacian schedule.
 * `on_iter` (may be NULL) is called after every iteration with the record and the number of
 * completed iterations so the host can write the reference's periodic dumps
 * (PsOptimizer.cpp:419-423); a non-zero return from it aborts the loop.
 * Returns in *n_done the iterations run, in *result 1 = converged (reference returns true),
 * 0 = diverged or max_it reached (reference returns false). */
typedef int (*psgsdf_iter_cb)(void* user, int iter_done, const psgsdf_iter_stats* rec);
int psgsdf_optimize(psgsdf_ctx* ctx, int flags, psgsdf_iter_stats* stats, int stats_cap,
                    int* n_done, int* result, psgsdf_iter_cb on_iter, void* user);

/* A PASSIVE per-iteration observer for psgsdf_optimize (progress narration, timing): called like on_iter as soon as the record of an
 * iteration is complete, but WITHOUT on_iter's guarantee that the context still holds the state of that iteration -- without an on_iter
 * callback psgsdf_optimize starts the next iteration (albedo / light blocks, undoable) before the stop decision on the previous one has
 * arrived.  The observer must not call into the context.  A non-zero return ends the loop on the iteration just reported (what the next
 * iteration applied speculatively is undone first).  NULL removes it.  No reference counterpart. */
int psgsdf_set_record_observer(psgsdf_ctx* ctx, psgsdf_iter_cb observer, void* user);

/* How often psgsdf_optimize's on_iter needs the EXACT state: with period p > 1 on_iter is only invoked for iterations with iter_done % p == 0
 * (and for the one that ended with the 2x refinement) -- voxelPS writes its meshes / point clouds every 3rd iteration (PsOptimizer.cpp:419-423) --
 * and the iterations in between are closed speculatively like a callback-free loop's; the record observer still sees every record.  Default 1. */
int psgsdf_set_on_iter_period(psgsdf_ctx* ctx, int period);

/* Optimizer::subsampling (OptimizerAux.cpp:622-684): 2x refine of grid + band rebuild. */
int psgsdf_upsample2x(psgsdf_ctx* ctx);

/* ---- state producer (SURVEY §8f "next" row 1) ------------------------------------------ */

/* VolumetricGradSdf::init (VolumetricGradSdf.cpp:14-38): an empty volume on the device — dist = T, grad = 0,
 * weight = 0, rgb = 0, no visibility — with room for `max_frames` integrated frames.  Alternative to
 * psgsdf_upload_volume. */
int psgsdf_volume_init(psgsdf_ctx* ctx, int max_frames);
/* VolumetricGradSdf::update (VolumetricGradSdf.cpp:51-138): fuse one RGB-D frame into the volume.
 * rgb: H*W*3 float RGB; depth: H*W metres (0 = invalid); normals_xyz: 3 planes of H*W, camera-frame, inward
 * pointing unit normals (what NormalEstimator::compute returns, NormalEstimator.h:150-176) -- or NULL: the engine
 * estimates them itself from `depth` on the device, as VolumetricGradSdf::update does (:59-61); pose: 4x4 row-major
 * camera->world; counter: index of this frame in the sequence (the visibility bit it sets, Sdf.h increase_counter). */
int psgsdf_integrate_frame(psgsdf_ctx* ctx, const float* rgb, const float* depth, const float* normals_xyz,
                           int width, int height, const float pose[16], int counter, float z_min, float z_max);

/* NormalEstimator::compute (normals/NormalEstimator.h:150-176): FALS normals of a depth map (11x11 window),
 * 3 planes of H*W, camera frame. */
int psgsdf_estimate_normals(psgsdf_ctx* ctx, const float* depth, int width, int height, float* normals_xyz);
/* RigidPointOptimizer::optimize (sdf_tracker/RigidPointOptimizer.cpp:12-79): frame-to-model depth tracking against the
 * volume on the device; `pose` (4x4 row-major camera->world) is the start value and receives the result.
 * Reference defaults: num_iterations 50, conv_threshold 1e-3, damping 1 (RigidOptimizer.h:41-47). */
int psgsdf_track(psgsdf_ctx* ctx, const float* depth, int width, int height, float pose[16], float z_min, float z_max,
                 int num_iterations, float conv_threshold, float damping, int* iters_out, int* converged);
/* the per-integrated-frame visibility words (N^3 * words); returns words per voxel (>0) or a negative status */
int psgsdf_download_vis_seq(psgsdf_ctx* ctx, uint64_t* out);

/* ---- outputs --------------------------------------------------------------------------- */

int psgsdf_get_info(psgsdf_ctx* ctx, psgsdf_info* info);
/* Dense state back to the host (what the reference's writers read through tSDF_->tsdf_,
 * OptimizerAux.cpp:278-577).  Any pointer may be NULL. vis_words: N^3 * info.vis_words. */
int psgsdf_download_volume(psgsdf_ctx* ctx, float* dist, float* grad_xyz, float* weight,
                           float* rgb, uint64_t* vis_words);
/* surface_points_ (ascending linear indices), n_band entries */
int psgsdf_download_band(psgsdf_ctx* ctx, int32_t* lin_idx);
/* poses_: n_frames * 16 row-major */
int psgsdf_download_poses(psgsdf_ctx* ctx, float* poses);
/* light_: n_frames * light_stride (SH) or 3 floats (LED) */
int psgsdf_download_light(psgsdf_ctx* ctx, float* light);
int psgsdf_upload_light(psgsdf_ctx* ctx, const float* light);

/* ---- the solver of the light and pose blocks ---------------------------------------------- */

/* optimizeLightAll / optimizePosesAll (PsOptimizer.cpp:175-203,207-234, LedOptimizer.cpp:134-160,245-275) hand the block-diagonal normal equations of
 * ALL frames to ONE Eigen::ConjugateGradient<SparseMatrix<float>> (Jacobi preconditioner, tolerance eps_f32, <= 2n passes); the LED pose update is
 * applied only when info() == Success.
 *   mode 0 (default): every frame's block solved directly, LDL^T in double, inside the sweep that summed it (no launch) -- the
 *           exact step of each block; it differs from the reference's by what a float CG leaves undetermined, cond(block) x eps_f32 of the step
 *           (SH1 / LED / pose: 1e-7 relative; SH2's 9 x 9 light blocks, cond ~2e4: 1e-3);
 *   mode 1: the reference's solver itself (csrc/frame_solve.hip: one workgroup, the same float recurrences, stop rule and info() semantics as Eigen's
 *           conjugate_gradient) in a kernel of its own behind the sweep: +5-25 % per iteration.
 * Also PSGSDF_FRAME_SOLVE=eigen|ldlt when the context is created.  Mode 1 holds at most 2048 unknowns (227 keyframes with SH2). */
int psgsdf_set_frame_solver(psgsdf_ctx* ctx, int mode);
/* What the last mode-1 solve of `block` (PSGSDF_LIGHT / PSGSDF_POSE) reported: Eigen's iterations(), error(), info() == Success, and whether the
 * update was applied (the LED pose gate).  Synchronises the context's stream.  Any pointer may be NULL.  psgsdf_step fills the same numbers into
 * its psgsdf_step_stats. */
int psgsdf_get_frame_solver_stats(psgsdf_ctx* ctx, int block, int32_t* iterations, double* error, int32_t* converged, int32_t* applied);

/* ---- the writers' geometry, extracted on the device ---------------------- */

/* What the reference writes every third iteration (PsOptimizer.cpp:419-423) is computed from the dense state: a mesh by marching cubes and a point
 * cloud of the band.  These calls do that on the device and hand back compact arrays in engine-owned pinned host memory, valid until the next
 * extraction call on the context; the host only formats text.
 * Multi-rank contexts (round 5): collective calls; every rank gets ITS share -- the cells whose lower z-plane it owns, its own band rows / voxels /
 * planes of the crop box (which is the whole volume's: lo, dim) -- in the single context's order: the shares concatenated in rank order are the single
 * context's arrays.  extract_sdf returns the planes k of the box with z0 <= lo[2] + k < z1 (psgsdf_mg_info); a share may be empty (NULL, 0).
 * psgsdf_comm_allreduce_host gives a host the counts before it (host/ps_optimizer.hpp: every rank writes its lines into its place in the one file).
 *
 * psgsdf_extract_mesh: Optimizer::extract_mesh (OptimizerAux.cpp:278-363) = crop box of |d| <= sqrt(3) vs, tsdf = -dist, 8-bit colours, then
 *   MarchingCubes::computeIsoSurface / computeTriangles (third/mesh/MarchingCubes.cpp:314-637): cells in (z, y, x) order, faces in the classic
 *   table's order, degenerate faces dropped, non-indexed vertices.  xyz: n_vertices * 3 floats in grid-local coordinates (three consecutive
 *   vertices = one face), rgb: n_vertices * 3 bytes.  The same floats and bytes as the host-side pass (host/marching_cubes.hpp), so *_mesh.ply
 *   comes out byte for byte.
 * psgsdf_extract_pointcloud: which = 0: Optimizer::save_pointcloud (OptimizerAux.cpp:456-511): the band voxels (ascending) with |d| < sqrt(3) vs;
 *   which = 1: VolumetricGradSdf::extract_pc (VolumetricGradSdf.cpp:320-376): every voxel with weight > 0 and |d| < sqrt(3) vs.
 *   xyz_nxyz: n_points * 6 floats (x - d g^ in grid-local coordinates, then g^), rgb: n_points * 3 ints = int(255 * colour), as the reference prints them.
 * psgsdf_extract_sdf: the block Optimizer::saveSDF / VolumetricGradSdf::saveSDF write (OptimizerAux.cpp:513-577): lo = first voxel of the crop box,
 *   dim = its extent, neg_dist = dim[0] * dim[1] * dim[2] values of -dist, x fastest.  dim = 0 if no voxel lies within sqrt(3) vs of the surface. */
int psgsdf_extract_mesh(psgsdf_ctx* ctx, const float** xyz, const uint8_t** rgb, int64_t* n_vertices);
int psgsdf_extract_pointcloud(psgsdf_ctx* ctx, int which, const float** xyz_nxyz, const int32_t** rgb, int64_t* n_points);
int psgsdf_extract_sdf(psgsdf_ctx* ctx, int32_t lo[3], int32_t dim[3], const float** neg_dist);

/* ---- multi-GPU (z-slab partition, one context per rank, one process per GPU) -------------- */

/* psgsdf_destroy on a multi-rank context is NOT a collective: a rank closes its mappings of the other ranks' exchange memory, tells their owners, and
 * frees what it exported itself once every rank that mapped it has reported the same -- or after PSGSDF_DESTROY_TIMEOUT_S (15 s), in which case that
 * memory is leaked rather than freed under a peer that may still be writing to it.  Ranks may fail, leave early or destroy their contexts in any order.
 *
 * The reference is a single process (main_ps.cpp:41-343) and has no counterpart of this section.  The volume is cut along z into n_ranks
 * slabs of (about) equal band count; a context attached to a rank keeps ITS slab (plus one halo plane on each inner side) on its device and
 * computes only there, and EVERY entry point above keeps its meaning: psgsdf_upload_volume (every rank passes the whole volume and keeps its
 * slab of it) / psgsdf_init / psgsdf_step / psgsdf_iterate / psgsdf_optimize / psgsdf_upsample2x become collective calls (all ranks make them,
 * in the same order) that return the same global energies, counts, poses and lights on every rank; psgsdf_get_info reports the whole volume.
 * psgsdf_download_volume writes the z-planes the rank OWNS into the caller's whole-volume arrays (the slabs tile the volume);
 * psgsdf_download_band returns the rank's own band voxels (global linear indices).  The exchanges -- all-reduce of the per-frame light /
 * pose rows, of the 7 sums of a PCG pass and of the folded scalars; halo rows of the per-voxel blocks, the PCG records and the distances
 * with the two z-neighbours -- are enqueued by the engine itself on its HIP stream.
 *
 *   rank 0:  psgsdf_comm_unique_id(id);  (hand the 128 bytes to the other ranks: file, socket, MPI_Bcast, torch.distributed ...)
 *   all   :  psgsdf_create(.., device, &ctx);  psgsdf_comm_init(ctx, id, rank, n_ranks);  then the usual call sequence.
 */
int psgsdf_comm_unique_id(uint8_t id[128]);                       /* ncclGetUniqueId; PSGSDF_ERR_COMM if librccl cannot be loaded */
/* RCCL communicator over xGMI for this context's device (ncclCommInitRank: blocks until all ranks have called).  Before psgsdf_upload_volume. */
int psgsdf_comm_init(psgsdf_ctx* ctx, const uint8_t id[128], int rank, int n_ranks);

/* Slab-local upload: a host that cannot (or should not) hold the whole volume on every rank.
 *   1. every rank counts the band candidates of SOME z-planes (any split, e.g. plane k on rank k mod n): psgsdf_slab_plane_count on one plane's
 *      nx*ny distances + visibility words -> its entry of a gdim[2]-long histogram, 0 for planes it did not look at;
 *   2. psgsdf_plan_slab (collective: one all-reduce of the histogram) returns the planes [z0, z1) this rank OWNS -- the cuts
 *      psgsdf_upload_volume would have chosen (equal band count);
 *   3. psgsdf_upload_volume_slab with arrays that hold only the planes [max(0, z0-1), min(nz, z1+1)) (own planes + one halo plane per inner side;
 *      grad / rgb: three consecutive blocks of that many voxels).
 * Everything after that (psgsdf_set_keyframes, psgsdf_init, ...) is as with psgsdf_upload_volume. */
int psgsdf_slab_plane_count(psgsdf_ctx* ctx, const float* dist_plane, const uint64_t* vis_plane, int words_per_voxel, double* count);
int psgsdf_plan_slab(psgsdf_ctx* ctx, const double* plane_counts /* [gdim z] */, int* z0, int* z1);
int psgsdf_upload_volume_slab(psgsdf_ctx* ctx, int z0, int z1, const float* dist, const float* grad_xyz, const float* weight,
                              const float* rgb, const uint64_t* vis_words, int words_per_voxel);

/* The same with a transport supplied by the caller (a host that already owns a communicator: MPI, a test harness).  All pointers are
 * device pointers; each primitive must be ordered after the work already enqueued on `hip_stream` and its result must be visible to
 * work enqueued on that stream afterwards (a blocking implementation may simply synchronise the stream).  Return 0 on success. */
typedef struct psgsdf_comm_xfer { void* ptr_dev; size_t bytes; int peer; } psgsdf_comm_xfer;
typedef struct psgsdf_comm_ops {
    void* user;
    int (*allreduce_f64)(void* user, double* buf_dev, int n, void* hip_stream);                 /* in-place sum over all ranks */
    int (*sendrecv)(void* user, const psgsdf_comm_xfer* sends, int n_sends,
                    const psgsdf_comm_xfer* recvs, int n_recvs, void* hip_stream);              /* peers: rank-1 / rank+1 (any rank in psgsdf_rebalance_slabs); matched in list order per peer */
} psgsdf_comm_ops;
int psgsdf_comm_init_ext(psgsdf_ctx* ctx, const psgsdf_comm_ops* ops, int rank, int n_ranks);
/* A built-in transport of that kind for the ranks of ONE node: peer_fd[r] = a connected stream socket (socketpair / TCP) to rank r, entry `rank`
 * ignored; the launcher owns the sockets.  Host-staged and blocking: for ranks that share a device (RCCL refuses that -- the one-GPU rehearsals of
 * `voxelPS --gpus N`) or a node without a working RCCL.  A peer that stops answering fails the call after PSGSDF_SOCKET_TIMEOUT_S (120) seconds. */
int psgsdf_comm_init_sockets(psgsdf_ctx* ctx, const int* peer_fd, int rank, int n_ranks);
/* In-place sum over the ranks of n doubles in HOST memory through the context's communicator (collective; nothing to do on one rank): what a multi-rank
 * host needs to place its share of an output. */
int psgsdf_comm_allreduce_host(psgsdf_ctx* ctx, double* buf, int n);

/* A volume fused slab-parallel (psgsdf_volume_init / psgsdf_integrate_frame on a multi-rank context: every rank fuses every frame into the z-planes it
 * holds, VolumetricGradSdf.cpp:78-134 touches each voxel independently; the slabs are cut by HEIGHT because the band does not exist yet) is re-cut
 * into slabs of equal band-candidate count -- the partition psgsdf_upload_volume chooses -- and the planes move to their new ranks.  Collective;
 * a no-op on one rank.  Uses the transport's sendrecv with ANY rank as peer, not only the z-neighbours.  No reference counterpart. */
int psgsdf_rebalance_slabs(psgsdf_ctx* ctx);

/* run every launch of this context on a caller-owned HIP stream */
int psgsdf_set_stream(psgsdf_ctx* ctx, void* hip_stream);
/* out = { S (band voxels of the whole volume), band rows this context holds, row0, row1, halo, F, rank, n_ranks, need_lo, need_hi, z0, z1 }:
 * the context owns the global z-planes [z0, z1) = its band rows [row0, row1); the rows before / behind them are the halo planes
 * (need_lo / need_hi rows, refreshed from the z-neighbours); halo = max(need_lo, need_hi). */
int psgsdf_mg_info(psgsdf_ctx* ctx, int32_t out[12]);
/* collectives + halo exchanges this context has enqueued since it was created */
int psgsdf_comm_stats(psgsdf_ctx* ctx, int64_t* n_collectives);

/* ---- measurement / test hooks (not part of the reference seam) -------------------------- */
/* timing probe (profiles/r05_notes.md section 5): out[4] = ms of { distance sweep alone, distance solve alone, the two back to back, the solve started on a
 * second stream together with the sweep } -- the upper bound of what starting the solve under the sweep's tail could hide */
int psgsdf_debug_overlap_probe(psgsdf_ctx* ctx, int reps, double* out);
/* the FALS estimator's per-resolution cache (NormalEstimator::cache, NormalEstimator.h:52-125) as the device computed it: 9 planes of width * height floats
 * (ray / (1 + x0^2 + y0^2): 3, the inverse of the box-filtered 3x3 matrix: 6) */
int psgsdf_debug_normals_cache(psgsdf_ctx* ctx, int width, int height, float* cache9);
/* the mode-1 frame solver alone (known-answer tests): Eigen's conjugate_gradient with the Jacobi preconditioner on a caller-supplied block-diagonal
 * system of n_blocks blocks of n x n floats (row-major, used as given; n in {3, 4, 6, 9}, n_blocks * n <= 2048), right-hand side b, at most max_it
 * passes (<= 0: Eigen's default 2 * n_blocks * n).  x: n_blocks * n floats out. */
int psgsdf_debug_frame_cg(psgsdf_ctx* ctx, int n_blocks, int n, const float* H, const float* b, float* x, int max_it,
                          int32_t* iterations, double* error, int32_t* converged);

/* last measured kernel durations in ms keyed by name; names[i] are static strings.
 * Returns the number of entries written (<= cap). */
int psgsdf_kernel_times(psgsdf_ctx* ctx, const char** names, double* ms, int64_t* launches, int cap);
int psgsdf_reset_kernel_times(psgsdf_ctx* ctx);
/* enable per-kernel hipEvent timing (adds a sync per launch: measurement mode only) */
int psgsdf_set_profiling(psgsdf_ctx* ctx, int enabled);
/* time ONE kernel (by its name in psgsdf_kernel_times) with hipEvent pairs recorded on the launch
 * stream and no host synchronisation; resolved by the next psgsdf_kernel_times call.  NULL/"" = off.
 * "name/N" samples every N-th launch (an event pair costs ~3 us of stream time). */
int psgsdf_watch_kernel(psgsdf_ctx* ctx, const char* name);
/* Builds the distance normal equations at the current state and returns, for the n_band rows:
 * diag (H_ii before damping), rhs b, and y = H*x for the supplied x (may be NULL). */
/* the persistent PCG solve run for exactly `passes` passes, stop rule off: average kernel time over `reps` launches;
 * shape = { workgroups, rows per thread }; stamps (may be NULL) = stage timestamps of pass 8 of two workgroups, 100 MHz ticks.
 * PSGSDF_ERR_UNSUPPORTED where the per-pass kernels are used instead. */
int psgsdf_debug_time_pcg_solve(psgsdf_ctx* ctx, int passes, int reps, double* ms_per_launch, int32_t shape[2], double stamps[16]);
int psgsdf_debug_dist_system(psgsdf_ctx* ctx, float* diag, float* rhs, const float* x, float* y);
/* timing only: average ms of `reps` back-to-back launches of the fused PCG pass on the current distance system with
 * `blocks` workgroups (0 = one row per thread) and ablation bits (see pcg.hip: k_cgf_pass); leaves the PCG state undefined */
int psgsdf_debug_time_pcg_pass(psgsdf_ctx* ctx, int blocks, int rows_in_flight, int ablate, int reps, double* avg_ms, long long* stamps /* nullable: [blocks][8] */);
/* rows / 64-row groups of the current distance system that use any of the 6 rare ELL columns (launch-shape diagnostics) */
int psgsdf_debug_rare_rows(psgsdf_ctx* ctx, int64_t* rows, int64_t* waves);
/* per-frame light / pose normal equations at the current state: H (n*n row-major) and b (n)
 * per frame, n = light_stride or 6.  `block` = PSGSDF_LIGHT or PSGSDF_POSE. */
int psgsdf_debug_frame_system(psgsdf_ctx* ctx, int block, double* H, double* b);
/* albedo diagonal system: H (3*n_band), b (3*n_band), channel-interleaved per row */
int psgsdf_debug_albedo_system(psgsdf_ctx* ctx, float* H, float* b);
/* Host <-> device hand-off statistics of this context (no reference counterpart: the reference is one host thread).
 *   out[0] scalar read-backs validated against their check words      out[1] of those: not complete yet when the marker / status word
 *   the host waited for had already arrived (waited for; PSGSDF_MBOX_CHECK=0 takes them as they are: the round-2 behaviour)
 *   out[2] distance steps re-run on the per-pass kernels because the persistent solve could not get its workgroups co-resident
 *          (+ 1e6 if the float keyframes of psgsdf_set_keyframes turned out to be 8-bit data and are held as RGBA8 words)
 *   out[3] 1e6 x iterations started speculatively (before the stop decision on the previous one) + those of them that were undone
 *   out[4] 1 if this (multi-rank) context holds the cross-rank mappings of the persistent solve (+ 10 x halo exchanges done by push / pull kernels
 *          through those mappings instead of the communicator), out[5] distance solves run through it,
 *   out[6] memory kind the hand-off probe chose for the record planes another device writes (-1 not probed: single rank; 0 none passed: cross-rank
 *   solve off; 1 fine-grained; 2 uncached; 3 coarse, pinned by PSGSDF_XR_MEM), out[7] 1e6 x stale records + timed-out waits the probe saw (all ranks) */
int psgsdf_debug_sync_stats(psgsdf_ctx* ctx, int64_t out[8]);

/* ---- tuning knobs ------------------------------------------------------------------------- */

/* The engine reads the following environment variables when a context is created (a few, marked *, when they are used).  All of them only choose
 * between equivalent execution strategies -- results agree to rounding, most bit for bit (tests/test_knobs_gpu.py) -- and none is needed in
 * normal operation.  psgsdf_get_tuning reports what was set and what it resolved to.
 *   PSGSDF_PCG_POLL, PSGSDF_SPECULATE, PSGSDF_SPECULATE_MR, PSGSDF_FOLD_IN_NEXT, PSGSDF_FUSE_ALBEDO, PSGSDF_FUSE_PCG_INIT        (0 / 1) host-side scheduling
 *   PSGSDF_PCG_PERSIST, PSGSDF_PCG_PIPELINE, PSGSDF_PCG_TAGM (0: the pipelined solve's exchanged values without their own tags -- round 4's hand-off; 1: self-validating on one rank only; default 2: between ranks too), PSGSDF_PCG_PREFETCH, PSGSDF_PCG_FUSE_ASM, PSGSDF_PCG_FUSE_APPLY, PSGSDF_PCG_XCD_LOCAL,
 *   PSGSDF_PCG_COL16*, PSGSDF_PCG_ROWS*, PSGSDF_PCG_BLOCKS*                                                                        distance solve
 *   PSGSDF_FRAME_SOLVE (ldlt | eigen: psgsdf_set_frame_solver; the ONE knob that changes results beyond rounding),
 *   PSGSDF_FM_SOLVE, PSGSDF_FM_ROWS*, PSGSDF_IMG_COMPACT, PSGSDF_XCD_MAP, PSGSDF_XCD_STRIPE                                        sweeps
 *   PSGSDF_AREG_DEVICE (0: the `reg albedo` CG driven by the host, two read-backs per iteration)                                   regularised albedo solve
 *   PSGSDF_XR, PSGSDF_XF, PSGSDF_XS, PSGSDF_XH (0: that exchange through the communicator instead of IPC-mapped memory), PSGSDF_XR_MEM* (fine | uncached |
 *   coarse), PSGSDF_XWAIT_LOG2 (log2 of the polls an in-kernel wait for another rank may take), PSGSDF_CU_MASK (lo:hi)              multi-rank
 *   PSGSDF_WAIT_TIMEOUT_S*, PSGSDF_DESTROY_TIMEOUT_S*, PSGSDF_SOLVE_DUMP*                                                           host waits / diagnostics
 * DESIGN.md section 4 has the table with defaults and measured effects.
 * NOT in libpsgsdf.so: PSGSDF_FAULT_SOLVE, PSGSDF_FAULT_HALO (fault injection), PSGSDF_PCG_ABLATE (timing ablations: results are WRONG),
 * PSGSDF_MBOX_CHECK=0 (re-opens a race fixed in round 3).  They exist only in the development build libpsgsdf_dev.so (-DPSGSDF_DEV) the tests and
 * tools load; the product library ignores them and says so under "ignored_dev_only".
 *
 * psgsdf_get_tuning writes one JSON object {"build": .., "env": {variables that were set, verbatim}, "ignored_dev_only": {..}, "effective": {the
 * switches as resolved, defaults included}} into `json` (at most cap - 1 characters + terminator; json may be NULL) and returns the length the whole
 * text needs, or a negative status. */
int psgsdf_get_tuning(psgsdf_ctx* ctx, char* json, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PSGSDF_H_ */
