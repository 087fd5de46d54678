/* psgsdf_oracle.c — CPU restatement of the reference's photometric-stereo hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under psgradientsdf_amd/ may include, link or call this
 * file; it is the checker used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  The shipped product is the HIP engine (psgradientsdf_amd/csrc).
 *
 * PARITY UNPINNED: the reference (Sangluisme/PSgradientSDF) has no tests, golden vectors or
 * fixtures for this path, and it cannot be built in this image (Eigen, Sophus, OpenCV,
 * CLI11, nlohmann/json are absent).  This restatement is pinned instead by
 * analytic known-answer tests and by the enabled form of the reference's own disabled
 * numeric-vs-analytic Jacobian diagnostic (PsOptimizerJa.cpp:293-318,514-517) in tests/
 * (test_oracle_kat.py, test_oracle_kat2.py: distance / pose / albedo Jacobians incl. the LED
 * pose block, the Eikonal row, the Laplacian diagonal, the assembled distance system against
 * an explicit J^T W J, Eigen-CG semantics, SO3::exp, band membership, forward-model KATs).
 *
 * Where the HIP engine's arithmetic differs from this file (all far inside
 * the parity tolerances): observation sums per voxel in float before the double reductions;
 * block-diagonal light / pose systems solved per block by LDL^T in double instead of one global
 * Jacobi-PCG; the distance PCG's fused recurrences (one reduction per pass); FMA contraction of
 * the per-observation algebra; v_rcp_f32 in the Cauchy weight, v_log_f32 x ln2 in the Cauchy loss; float
 * bilinear weights; the Jacobian chain image_grad * pi_grad * R^T * d(point) contracted from the right
 * (channel-independent rows first) instead of a 3x3 product per channel, p.(R^T dx) as (R p).dx in
 * the LED distance Jacobian, the SH2 light contracted with dSH/dn before the stencil slots;
 * quotients sharing a divisor via one correctly rounded reciprocal + Markstein correction (same
 * bits); out-of-bounds reads of the
 * reference clamped.  NOT a deviation any more:
 * the Jacobians' second projection fx*px/pz (PsOptimizerJa.cpp:70-76) is reproduced by both.
 *
 * Arithmetic follows the reference: float32 per observation, evaluated in the reference's
 * operation order (no FMA contraction: build with -ffp-contract=off), double only where the
 * reference's C++ promotes to double (bilinear weights Auxilary.h:47, 1./z OptimizerAux.cpp:219,
 * std::pow in Optimizer.cpp:278).  One deliberate deviation: sums over many observations
 * (energies, normal-equation entries, CG dot products) are accumulated in double and rounded
 * to float once, because the reference's float summation order (Eigen sparse products) cannot
 * be reproduced and control flow must not depend on it (SURVEY.md §7 hard part 2).
 *
 * Every function cites the reference lines it follows, relative to
 * /root/reference/cpp/include/.  The exported orc_* functions mirror include/psgsdf.h
 * one-to-one so the same test driver runs the engine and the oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/psgsdf.h"

#define MAXB 9 /* max SH basis */

typedef struct orc_ctx {
    /* grid (VoxelGrid.h:27-34) */
    int dim[3];
    size_t nvox;
    float vs, vs_inv;
    float shift[3], origin[3];
    float T;
    float fx, fy, cx, cy;
    psgsdf_settings set;
    float reg_n, reg_l; /* effective weights (settings_->reg_weight_n/l are mutated, B9) */
    float reg_r;        /* "reg albedo": never normalised (PsOptimizer.cpp:279) */
    /* dense voxel state (SdfVoxel, Sdfvoxel.h:6-13) in SoA */
    float *dist, *gx, *gy, *gz, *weight, *r, *g, *b;
    uint64_t* vis_seq; int wpv_seq;  /* per integrated frame */
    uint64_t* vis;     int wpv;      /* per keyframe, after select_vis */
    /* keyframes */
    int F, W, H;
    int* frame_idx;
    float* img;    /* F*H*W*3 RGB */
    float* poses;  /* F*16 row-major */
    float* light;  /* F*MAXB (SH) or 3 (LED) */
    int basis;
    /* band (surface_points_) */
    int S;
    int* band;
    int* row_of;
    int inited;
    /* multi-rank (slab) mode: owned band rows [row0,row1); exchange buffers mirror the engine's layout */
    int rank, n_ranks, row0, row1, halo, Spad;
    double* mg_frame; double mg_scal[64]; double mg_ext[8]; int mg_fold_base; int need[2];
    float *mg_dist, *mg_blk, *mg_rec[2], *mg_rho, *mg_grad;
    double cg_bb; int cg_stop; double* cg_hist;
    float *cg_x, *cg_r, *cg_t, *cg_p, *cg_inv; double* cg_sc;
    void* mg_sys; /* dist_sys* of the current solve */
    int solver_mode; /* 0 = direct per-block solves, 1 = Eigen-style global Jacobi-PCG */
    int fs_iters[2], fs_ok[2], fs_applied[2]; double fs_err[2]; /* the last light [0] / pose [1] solve: Eigen's iterations(), info() == Success, update applied, error() */
    int faithful;    /* 1 = band membership as the reference does it: std::find over surface_points_ (Optimizer.cpp:470), O(S) per look-up */
    long long n_find; /* look-ups made through band_find (both modes) */
    int threads;
    char err[256];
    /* last dist system (debug) */
} orc_ctx;

/* ------------------------------------------------------------------ small helpers */

static inline float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline float norm3(const float a[3]) { return sqrtf(dot3(a, a)); }
/* Eigen normalized(): z = squaredNorm; z>0 ? v/sqrt(z) : v  */
static inline void normalized3(const float v[3], float o[3]) {
    float z = dot3(v, v);
    if (z > 0.f) { float s = sqrtf(z); o[0] = v[0] / s; o[1] = v[1] / s; o[2] = v[2] / s; }
    else { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; }
}
/* o = M^T v for row-major 3x3 M (R.transpose() * v) */
static inline void mulT3(const float* M, const float v[3], float o[3]) {
    for (int i = 0; i < 3; ++i) o[i] = (M[0 * 3 + i] * v[0] + M[1 * 3 + i] * v[1]) + M[2 * 3 + i] * v[2];
}
static inline void mul3(const float* M, const float v[3], float o[3]) {
    for (int i = 0; i < 3; ++i) o[i] = (M[i * 3 + 0] * v[0] + M[i * 3 + 1] * v[1]) + M[i * 3 + 2] * v[2];
}
static inline void pose_Rt(const orc_ctx* c, int f, float R[9], float t[3]) {
    const float* P = c->poses + 16 * f; /* Optimizer.h:52-60 */
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = P[i * 4 + j]; t[i] = P[i * 4 + 3]; }
}
static inline void line2idx(const orc_ctx* c, int lin, int idx[3]) { /* VoxelGrid.h:88-97 */
    int nxy = c->dim[0] * c->dim[1];
    int k = lin / nxy; int rest = lin - k * nxy; int j = rest / c->dim[0]; int i = rest - j * c->dim[0];
    idx[0] = i; idx[1] = j; idx[2] = k;
}
static inline void voxel2world(const orc_ctx* c, const int idx[3], float x[3]) { /* VoxelGrid.h:38-40 */
    for (int a = 0; a < 3; ++a) x[a] = c->origin[a] + c->vs * (float)idx[a];
}
static inline int vis_bit(const orc_ctx* c, int lin, int f) {
    return (int)((c->vis[(size_t)lin * c->wpv + (f >> 6)] >> (f & 63)) & 1ull);
}
static inline const float* pix(const orc_ctx* c, int f, int row, int col) {
    /* clamp: only reachable through the reference's one-past reads at the image corner (B16) */
    if (row < 0) row = 0; if (row >= c->H) row = c->H - 1;
    if (col < 0) col = 0; if (col >= c->W) col = c->W - 1;
    return c->img + (((size_t)f * c->H + row) * c->W + col) * 3;
}

/* Auxilary.h:41-61 interpolateImage(m=row coordinate, n=column coordinate) */
static void interpolate_image(const orc_ctx* c, int f, float m, float n, float out[3]) {
    int x = (int)floorf(m), y = (int)floorf(n);
    if ((x + 1) < c->H && (y + 1) < c->W) {
        const float* p10 = pix(c, f, x + 1, y);
        const float* p00 = pix(c, f, x, y);
        const float* p11 = pix(c, f, x + 1, y + 1);
        const float* p01 = pix(c, f, x, y + 1);
        double w1 = ((double)y + 1.0 - (double)n) * (double)(m - (float)x);          /* double * float */
        double w2 = ((double)y + 1.0 - (double)n) * ((double)x + 1.0 - (double)m);  /* double * double */
        float  w3 = (n - (float)y) * (m - (float)x);                                  /* float * float */
        double w4 = (double)(n - (float)y) * ((double)x + 1.0 - (double)m);           /* float * double */
        for (int ch = 0; ch < 3; ++ch) {
            float t1 = (float)((double)p10[ch] * w1);
            float t2 = (float)((double)p00[ch] * w2);
            float t3 = p11[ch] * w3;
            float t4 = (float)((double)p01[ch] * w4);
            out[ch] = ((t1 + t2) + t3) + t4;
        }
    } else { /* the two middle branches are unreachable after the bounds test */
        const float* p = pix(c, f, x, y);
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
    }
}

/* Auxilary.h:64-123 computeImageGradient(m=row, n=col, direction) */
static void image_gradient(const orc_ctx* c, int f, float m, float n, int direction, float out[3]) {
    int x = (int)floorf(m), y = (int)floorf(n);
    float w01 = m - (float)x, w11 = n - (float)y;
    float w00 = (float)(1.0 - (double)w01), w10 = (float)(1.0 - (double)w11);
    int rows = c->H, cols = c->W;
    float v0[3] = {0, 0, 0}, v1[3] = {0, 0, 0};
    int two = 0; float wa = 0, wb = 0;
    if (direction == 0) {
        if ((x + 1) < rows && (y + 1) < cols) {
            for (int ch = 0; ch < 3; ++ch) { v0[ch] = pix(c, f, x, y + 1)[ch] - pix(c, f, x, y)[ch]; v1[ch] = pix(c, f, x + 1, y + 1)[ch] - pix(c, f, x + 1, y)[ch]; }
            two = 1; wa = w00; wb = w01;
        } else if ((x + 1) >= rows) {
            for (int ch = 0; ch < 3; ++ch) v0[ch] = pix(c, f, x, y + 1)[ch] - pix(c, f, x, y)[ch];
        } else { /* x+1<rows && y+1>=cols */
            for (int ch = 0; ch < 3; ++ch) { v0[ch] = -pix(c, f, x, y - 1)[ch] + pix(c, f, x, y)[ch]; v1[ch] = -pix(c, f, x + 1, y - 1)[ch] + pix(c, f, x + 1, y)[ch]; }
            two = 1; wa = w00; wb = w01;
        }
    } else {
        if ((x + 1) < rows && (y + 1) < cols) {
            for (int ch = 0; ch < 3; ++ch) { v0[ch] = pix(c, f, x + 1, y)[ch] - pix(c, f, x, y)[ch]; v1[ch] = pix(c, f, x + 1, y + 1)[ch] - pix(c, f, x, y + 1)[ch]; }
            two = 1; wa = w10; wb = w11;
        } else if ((x + 1) >= rows && (y + 1) < cols) {
            for (int ch = 0; ch < 3; ++ch) { v0[ch] = -pix(c, f, x - 1, y)[ch] + pix(c, f, x, y)[ch]; v1[ch] = -pix(c, f, x - 1, y + 1)[ch] + pix(c, f, x, y + 1)[ch]; }
            two = 1; wa = w10; wb = w11;
        } else { /* y+1>=cols */
            for (int ch = 0; ch < 3; ++ch) v0[ch] = pix(c, f, x + 1, y)[ch] - pix(c, f, x, y)[ch];
        }
    }
    for (int ch = 0; ch < 3; ++ch) out[ch] = two ? (wa * v0[ch] + wb * v1[ch]) : v0[ch];
}

/* per-observation geometry shared by getIntensity and the Jacobians */
typedef struct obs_geom {
    float point[3]; /* camera-frame surface point */
    float m, n;     /* column, row */
    float gn[3];    /* v.grad.normalized() */
} obs_geom;

/* OptimizerAux.cpp:207-231 getIntensity (use_div=0) and the projection inside the Jacobians
 * (PsOptimizerJa.cpp:70-76, use_div=1: fx*px/pz instead of fx*px*z_inv). */
static int project(const orc_ctx* c, int lin, const float R[9], const float t[3], int use_div, obs_geom* o) {
    int idx[3]; line2idx(c, lin, idx);
    float xv[3]; voxel2world(c, idx, xv);
    float gr[3] = {c->gx[lin], c->gy[lin], c->gz[lin]};
    normalized3(gr, o->gn);
    float d = c->dist[lin];
    float tmp[3];
    for (int a = 0; a < 3; ++a) tmp[a] = (xv[a] - d * o->gn[a]) - t[a];
    mulT3(R, tmp, o->point);
    if (use_div) {
        o->m = c->fx * o->point[0] / o->point[2] + c->cx;
        o->n = c->fy * o->point[1] / o->point[2] + c->cy;
    } else {
        float z_inv = (float)(1.0 / (double)o->point[2]);
        o->m = c->fx * o->point[0] * z_inv + c->cx;
        o->n = c->fy * o->point[1] * z_inv + c->cy;
    }
    /* reference: if (m<0 || m>=cols || n<0 || n>=rows) return false; NaN treated as outside */
    if (!(o->m >= 0.f && o->m < (float)c->W && o->n >= 0.f && o->n < (float)c->H)) return 0;
    return 1;
}
static int get_intensity(const orc_ctx* c, int lin, int f, const float R[9], const float t[3], float I[3], obs_geom* og) {
    obs_geom o;
    if (!project(c, lin, R, t, 0, &o)) return 0;
    interpolate_image(c, f, o.n, o.m, I); /* OptimizerAux.cpp:228 */
    if (og) *og = o;
    return 1;
}

/* Band row of linear index ln, or -1.  Indexed mode (default): the row_of table.  Faithful mode (orc_set_faithful): what the
 * reference does at every one of its membership tests -- std::find(surface_points_.begin(), surface_points_.end(), lin_idx), a linear
 * scan of the ascending band list (Optimizer.cpp:470,520,565,574,628; PsOptimizerJa.cpp:523,541; LedOptimizerJa.cpp:444,462).  Same
 * result, O(S) per look-up: the reference's dominant cost. */
static inline int band_find(const orc_ctx* c, size_t ln) {
    if (!c->faithful) return c->row_of[ln];
    const int* b = c->band; const int S = c->S, key = (int)ln;
    for (int i = 0; i < S; ++i) if (b[i] == key) return i;
    return -1;
}
/* Optimizer.cpp:462-474 ifValidDirection(idx, +1, pos): bound test is `>` (B2), membership by
 * linear index (std::find over surface_points_ == row_of lookup). */
static inline int valid_forward(const orc_ctx* c, int lin, const int idx[3], int pos) {
    if (idx[pos] + 1 > c->dim[pos]) return 0;
    size_t stride = pos == 0 ? 1 : (pos == 1 ? (size_t)c->dim[0] : (size_t)c->dim[0] * c->dim[1]);
    size_t ln = (size_t)lin + stride;
    if (ln >= c->nvox) return 0;
    return band_find(c, ln) >= 0;
}
static inline float dist_at(const orc_ctx* c, long lin, float fallback) {
    if (lin < 0 || (size_t)lin >= c->nvox) return fallback; /* reference reads out of bounds here (UB) */
    return c->dist[lin];
}
/* Optimizer.cpp:287-364 computeDistGrad: (n, dir) */
static void dist_grad(const orc_ctx* c, int lin, float n[3], float dir[3]) {
    int idx[3]; line2idx(c, lin, idx);
    long stride[3] = {1, c->dim[0], (long)c->dim[0] * c->dim[1]};
    float d = c->dist[lin];
    for (int a = 0; a < 3; ++a) {
        dir[a] = valid_forward(c, lin, idx, a) ? 1.0f : -1.0f;
        float dn = dist_at(c, (long)lin + (long)dir[a] * stride[a], d);
        n[a] = dir[a] * (dn - d);
    }
    for (int a = 0; a < 3; ++a) n[a] = n[a] * c->vs_inv;
}
/* Optimizer.cpp:368-393 computeDistLaplacian */
static float dist_laplacian(const orc_ctx* c, int lin) {
    long stride[3] = {1, c->dim[0], (long)c->dim[0] * c->dim[1]};
    float d = c->dist[lin];
    float dd[3];
    for (int a = 0; a < 3; ++a) {
        float p1 = dist_at(c, (long)lin + stride[a], d), p0 = dist_at(c, (long)lin - stride[a], d);
        dd[a] = p1 + p0 - 2 * d;
    }
    return (dd[0] + dd[1] + dd[2]) * c->vs_inv * c->vs_inv;
}
/* Optimizer.cpp:269-284 normalJacobian(grad, direction, lag=false) */
static void normal_jacobian(const orc_ctx* c, const float grad[3], const float direction[3], float J[3]) {
    float n_d[3] = {-c->vs_inv * direction[0], -c->vs_inv * direction[1], -c->vs_inv * direction[2]};
    float N_inv = (float)(1.0 / (double)fmaxf(norm3(grad), 0.001f));
    float dN = (float)(pow((double)N_inv, 3) * (double)dot3(n_d, grad));
    for (int a = 0; a < 3; ++a) J[a] = N_inv * n_d[a] - dN * grad[a];
}
/* PsOptimizerJa.cpp:17-28 */
static void SH(const float n[3], int order, float* sh) {
    sh[0] = 1.0f; sh[1] = n[0]; sh[2] = n[1]; sh[3] = n[2];
    if (order == 2) { sh[4] = n[0] * n[1]; sh[5] = n[0] * n[2]; sh[6] = n[1] * n[2]; sh[7] = n[0] * n[0] - n[1] * n[1]; sh[8] = n[0] * n[0] - n[2] * n[2]; }
}
static inline float dotn(const float* a, const float* b, int n) { float s = 0.f; for (int i = 0; i < n; ++i) s += a[i] * b[i]; return s; }
static inline int sh_order(const orc_ctx* c) { return c->set.model == PSGSDF_SH2 ? 2 : 1; }

/* PsOptimizerJa.cpp:30-40 / LedOptimizerJa.cpp:15-29 renderedIntensity; og is the geometry
 * of the same observation (LED needs `point`). */
static void rendered_intensity(const orc_ctx* c, int lin, int f, const float R[9], const obs_geom* og, float out[3]) {
    float n[3], dir[3]; dist_grad(c, lin, n, dir);
    float nn[3]; normalized3(n, nn);
    float irr;
    if (c->set.model == PSGSDF_LED) {
        float Rp[3]; mul3(R, og->point, Rp);
        irr = -dot3(nn, Rp);
        float ld = (float)pow((double)norm3(og->point), 3);
        irr /= ld;
        out[0] = c->r[lin] * c->light[0] * irr; out[1] = c->g[lin] * c->light[1] * irr; out[2] = c->b[lin] * c->light[2] * irr;
    } else {
        float sh[MAXB]; SH(nn, sh_order(c), sh);
        irr = dotn(c->light + (size_t)f * MAXB, sh, c->basis);
        out[0] = c->r[lin] * irr; out[1] = c->g[lin] * irr; out[2] = c->b[lin] * irr;
    }
}

/* Optimizer.cpp:140-161 computeWeight (per channel) */
static inline float robust_weight(const orc_ctx* c, float r) {
    float lam = c->set.lambda, lam_sq = lam * lam;
    switch (c->set.loss) {
        case PSGSDF_CAUCHY: { float x = r / lam; return 1.0f / (1.0f + x * x); }
        case PSGSDF_TUKEY: { float x = r / lam; float w = (1.0f - x * x); w = w * w; return (r * r < lam_sq) ? w : 0.0f; }
        case PSGSDF_HUBER: { float w = lam * fabsf(1.0f / r); return (r * r < lam_sq) ? 1.0f : w; }
        case PSGSDF_TRUNC_L2: return (r * r < lam_sq) ? 1.0f : 0.0f;
        default: return 1.0f;
    }
}
/* Optimizer.cpp:164-186 computeLoss (per channel; the reference sums the 3 channels) */
static inline float robust_loss(const orc_ctx* c, float r) {
    float lam = c->set.lambda, lam_sq = lam * lam;
    switch (c->set.loss) {
        case PSGSDF_CAUCHY: { float x = r / lam; return logf(1.0f + x * x); }
        case PSGSDF_TUKEY: { float x = r / lam; float u = 1.0f - x * x; float v = 1.0f - u * u * u; return (r * r < lam_sq) ? v : 1.0f; }
        case PSGSDF_HUBER: { return (r * r < lam_sq) ? 0.5f * (r * r) : lam * (fabsf(r) - 0.5f * lam * 1.0f); }
        case PSGSDF_TRUNC_L2: { float x = fminf(fmaxf(r, -lam), lam); return x * x; }
        default: return r * r;
    }
}

/* ------------------------------------------------------------------ band */

static void mg_setup(orc_ctx* c);
/* OptimizerAux.cpp:237-257 getSurfaceVoxel */
static void build_band(orc_ctx* c) {
    free(c->band); free(c->row_of);
    c->row_of = (int*)malloc(sizeof(int) * c->nvox);
    int cnt = 0;
    for (size_t lin = 0; lin < c->nvox; ++lin) {
        int seen = 0;
        for (int w = 0; w < c->wpv; ++w) seen |= (c->vis[lin * c->wpv + w] != 0);
        if ((double)fabsf(c->dist[lin]) <= sqrt(3.0) * (double)c->vs && seen) c->row_of[lin] = cnt++;
        else c->row_of[lin] = -1;
    }
    c->S = cnt;
    c->band = (int*)malloc(sizeof(int) * (cnt > 0 ? cnt : 1));
    for (size_t lin = 0; lin < c->nvox; ++lin) if (c->row_of[lin] >= 0) c->band[c->row_of[lin]] = (int)lin;
    mg_setup(c);
}

/* Optimizer.cpp:30-47 select_vis: keyframe bit f := sequence bit frame_idx[f] */
static void select_vis(orc_ctx* c) {
    free(c->vis);
    c->wpv = (c->F + 63) / 64; if (c->wpv < 1) c->wpv = 1;
    c->vis = (uint64_t*)calloc(c->nvox * c->wpv, sizeof(uint64_t));
    for (size_t lin = 0; lin < c->nvox; ++lin)
        for (int f = 0; f < c->F; ++f) {
            int s = c->frame_idx[f];
            if (s < 0 || s >= 64 * c->wpv_seq) continue;
            if ((c->vis_seq[lin * c->wpv_seq + (s >> 6)] >> (s & 63)) & 1ull) c->vis[lin * c->wpv + (f >> 6)] |= 1ull << (f & 63);
        }
}

/* ------------------------------------------------------------------ energies */

/* PsOptimizer.cpp:47-78 / LedOptimizer.cpp:40-71 getPSEnergy */
static double ps_energy(const orc_ctx* c, long long* n_obs_out) {
    double E = 0.0; long long nobs = 0;
#pragma omp parallel for reduction(+ : E, nobs) schedule(static) num_threads(c->threads)
    for (int j = c->row0; j < c->row1; ++j) {
        int lin = c->band[j];
        for (int f = 0; f < c->F; ++f) {
            if (!vis_bit(c, lin, f)) continue;
            float R[9], t[3]; pose_Rt(c, f, R, t);
            float I[3]; obs_geom og;
            if (!get_intensity(c, lin, f, R, t, I, &og)) continue;
            float ren[3]; rendered_intensity(c, lin, f, R, &og, ren);
            float l = 0.f;
            for (int ch = 0; ch < 3; ++ch) l += robust_loss(c, I[ch] - ren[ch]);
            E += (double)l; nobs++;
        }
    }
    if (n_obs_out) *n_obs_out = nobs;
    return c->S ? E / (double)c->S : 0.0;
}
/* Optimizer.cpp:86-103 getNormalEnergy */
static double normal_energy(const orc_ctx* c) {
    double E = 0.0;
    for (int j = c->row0; j < c->row1; ++j) { float n[3], d[3]; dist_grad(c, c->band[j], n, d); float e = norm3(n) - 1; E += (double)(e * e); }
    return c->S ? E / (double)c->S : 0.0;
}
/* Optimizer.cpp:106-119 getLaplacianEnergy */
static double laplacian_energy(const orc_ctx* c) {
    double E = 0.0;
    for (int j = c->row0; j < c->row1; ++j) { float e = dist_laplacian(c, c->band[j]); E += (double)(e * e); }
    return c->S ? E / (double)c->S : 0.0;
}

/* Optimizer.cpp:396-460 computeAlbedoGrad: G[ch][axis] = dir_axis * (rho_ch(neighbour) - rho_ch(v)) / vs with the stencil
 * direction of the distance gradient (forward iff the forward neighbour is a band voxel); the neighbour is read from the
 * grid whether or not it is in the band.  nb_lin[a] = its linear index (the voxel itself if that would leave the grid). */
static void albedo_grad(const orc_ctx* c, int lin, float G[9], float dir[3], long nb_lin[3]) {
    int idx[3]; line2idx(c, lin, idx);
    long stride[3] = {1, c->dim[0], (long)c->dim[0] * c->dim[1]};
    const float* rho[3] = {c->r, c->g, c->b};
    for (int a = 0; a < 3; ++a) {
        dir[a] = valid_forward(c, lin, idx, a) ? 1.0f : -1.0f;
        long ln = (long)lin + (long)dir[a] * stride[a];          /* idx2line arithmetic, as the neighbour tables of the distance stencil */
        if (ln < 0 || (size_t)ln >= c->nvox) ln = lin;           /* the reference reads out of bounds here (UB): zero difference */
        nb_lin[a] = ln;
        for (int ch = 0; ch < 3; ++ch) G[ch * 3 + a] = (dir[a] * (rho[ch][ln] - rho[ch][lin])) * c->vs_inv;
    }
}
/* Optimizer.cpp:122-136 getAlbedoRegEnergy: mean over the band of sum_ch ||grad rho_ch|| (the norm, not its square) */
static double albedo_reg_energy(const orc_ctx* c) {
    double E = 0.0;
    for (int j = c->row0; j < c->row1; ++j) { float G[9], dir[3]; long nb[3]; albedo_grad(c, c->band[j], G, dir, nb);
        float e = 0.f; for (int ch = 0; ch < 3; ++ch) e += norm3(G + 3 * ch); E += (double)e; }
    return c->S ? E / (double)c->S : 0.0;
}
/* Optimizer.cpp:221-245 albedoRegJacobian(v): J[slot][ch], slot 0 = the voxel, 1..3 = its x/y/z stencil neighbour; res[ch] = ||grad rho_ch|| */
static void albedo_reg_jacobian(const orc_ctx* c, int lin, float J[4][3], float res[3], long nb_lin[3]) {
    float G[9], dir[3]; albedo_grad(c, lin, G, dir, nb_lin);
    float r_d[3] = {-c->vs_inv * dir[0], -c->vs_inv * dir[1], -c->vs_inv * dir[2]};
    for (int ch = 0; ch < 3; ++ch) {
        const float* g = G + 3 * ch;
        float gn = norm3(g); res[ch] = gn;
        J[0][ch] = (g[0] * r_d[0] + g[1] * r_d[1]) + g[2] * r_d[2];
        for (int a = 0; a < 3; ++a) J[a + 1][ch] = g[a] * (c->vs_inv * dir[a]);
        if (gn != 0.0f) for (int q = 0; q < 4; ++q) J[q][ch] /= gn;
    }
}

/* ------------------------------------------------------------------ linear algebra */

/* Eigen::ConjugateGradient<SparseMatrix<float>> with the default DiagonalPreconditioner
 * (SURVEY B18): float vectors, x0 = 0, threshold = max(tol^2*|b|^2, FLT_MIN), tol = eps_f32,
 * maxIters = 2n unless capped; dots accumulated in double.  matvec(user, p, out). */
typedef void (*matvec_fn)(void* user, const float* p, float* out);
typedef struct cg_result { int iters; double error; int success; } cg_result;
static double ddot(const float* a, const float* b, int n) { double s = 0; for (int i = 0; i < n; ++i) s += (double)a[i] * (double)b[i]; return s; }
static cg_result eigen_cg(int n, matvec_fn mv, void* user, const float* diag, const float* rhs, float* x, int max_it) {
    cg_result res = {0, 0.0, 1};
    const float tol = FLT_EPSILON;
    int maxIters = max_it > 0 ? max_it : 2 * n;
    memset(x, 0, sizeof(float) * n);
    float rhsNorm2 = (float)ddot(rhs, rhs, n);
    if (rhsNorm2 == 0) { res.iters = 0; res.error = 0; return res; }
    float* r = (float*)malloc(sizeof(float) * n); float* p = (float*)malloc(sizeof(float) * n);
    float* z = (float*)malloc(sizeof(float) * n); float* tmp = (float*)malloc(sizeof(float) * n);
    float* inv = (float*)malloc(sizeof(float) * n);
    for (int i = 0; i < n; ++i) inv[i] = diag[i] != 0.f ? 1.0f / diag[i] : 1.0f;
    memcpy(r, rhs, sizeof(float) * n); /* residual = rhs - A*0 */
    float threshold = fmaxf(tol * tol * rhsNorm2, FLT_MIN);
    float residualNorm2 = (float)ddot(r, r, n);
    int i = 0;
    if (residualNorm2 >= threshold) {
        for (int k = 0; k < n; ++k) p[k] = inv[k] * r[k];
        float absNew = (float)ddot(r, p, n);
        while (i < maxIters) {
            mv(user, p, tmp);
            float alpha = absNew / (float)ddot(p, tmp, n);
            for (int k = 0; k < n; ++k) { x[k] += alpha * p[k]; r[k] -= alpha * tmp[k]; }
            residualNorm2 = (float)ddot(r, r, n);
            if (residualNorm2 < threshold) break;
            for (int k = 0; k < n; ++k) z[k] = inv[k] * r[k];
            float absOld = absNew;
            absNew = (float)ddot(r, z, n);
            float beta = absNew / absOld;
            for (int k = 0; k < n; ++k) p[k] = z[k] + beta * p[k];
            i++;
        }
    }
    res.iters = i;
    res.error = sqrt((double)residualNorm2 / (double)rhsNorm2);
    res.success = res.error <= (double)tol;
    free(r); free(p); free(z); free(tmp); free(inv);
    return res;
}

/* dense symmetric solve in double with LDL^T, zero for non-positive pivots */
static void solve_spd(int n, const double* Hin, const double* bin, double* x) {
    double L[MAXB * MAXB]; double D[MAXB]; double y[MAXB];
    double scale = 0; for (int i = 0; i < n; ++i) if (fabs(Hin[i * n + i]) > scale) scale = fabs(Hin[i * n + i]);
    double tiny = scale * 1e-12;
    memset(L, 0, sizeof(L));
    for (int j = 0; j < n; ++j) {
        double d = Hin[j * n + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k] * D[k];
        D[j] = d;
        L[j * n + j] = 1.0;
        for (int i = j + 1; i < n; ++i) {
            double s = Hin[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k] * D[k];
            L[i * n + j] = (d > tiny) ? s / d : 0.0;
        }
    }
    for (int i = 0; i < n; ++i) { double s = bin[i]; for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k]; y[i] = s; }
    for (int i = 0; i < n; ++i) y[i] = (D[i] > tiny) ? y[i] / D[i] : 0.0;
    for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k]; x[i] = s; }
}

/* block-diagonal system of nb blocks of size n: float H (damped), b; solved either per block
 * (solver_mode 0) or as one global Eigen CG (solver_mode 1, what the reference does). */
typedef struct blockdiag { int nb, n; const float* H; } blockdiag;
static void blockdiag_mv(void* user, const float* p, float* out) {
    blockdiag* B = (blockdiag*)user;
    for (int k = 0; k < B->nb; ++k) for (int i = 0; i < B->n; ++i) {
        double s = 0; for (int j = 0; j < B->n; ++j) s += (double)B->H[(size_t)k * B->n * B->n + i * B->n + j] * (double)p[k * B->n + j];
        out[k * B->n + i] = (float)s;
    }
}
static cg_result solve_blockdiag(const orc_ctx* c, int nb, int n, const float* H, const float* b, float* x) {
    cg_result res = {0, 0.0, 1};
    if (c->solver_mode == 1) {
        blockdiag B = {nb, n, H};
        float* diag = (float*)malloc(sizeof(float) * nb * n);
        for (int k = 0; k < nb; ++k) for (int i = 0; i < n; ++i) diag[k * n + i] = H[(size_t)k * n * n + i * n + i];
        res = eigen_cg(nb * n, blockdiag_mv, &B, diag, b, x, 0);
        free(diag);
        return res;
    }
    for (int k = 0; k < nb; ++k) {
        double Hd[MAXB * MAXB], bd[MAXB], xd[MAXB];
        for (int i = 0; i < n * n; ++i) Hd[i] = (double)H[(size_t)k * n * n + i];
        for (int i = 0; i < n; ++i) bd[i] = (double)b[k * n + i];
        solve_spd(n, Hd, bd, xd);
        for (int i = 0; i < n; ++i) x[k * n + i] = (float)xd[i];
    }
    return res;
}

/* Sophus SO3::exp(omega).matrix(): quaternion exponential then Eigen toRotationMatrix */
static void so3_exp(const float w[3], float R[9]) {
    float theta_sq = dot3(w, w);
    float imag, real;
    if (theta_sq < 1e-10f /* Sophus Constants<float>::epsilon()^2 ~ (1e-5)^2 */) {
        float theta_po4 = theta_sq * theta_sq;
        imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_po4;
        real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_po4;
    } else {
        float theta = sqrtf(theta_sq), half = 0.5f * theta;
        imag = sinf(half) / theta; real = cosf(half);
    }
    float qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
    /* Eigen QuaternionBase::toRotationMatrix */
    float tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

/* ------------------------------------------------------------------ sub-steps */

/* the residual/weight pair of one observation (computeResidual body, PsOptimizerJa.cpp:586-619) */
static int residual_obs(const orc_ctx* c, int lin, int f, const float R[9], const float t[3], float r[3], float w[3], obs_geom* og) {
    float I[3];
    if (!get_intensity(c, lin, f, R, t, I, og)) return 0;
    float ren[3]; rendered_intensity(c, lin, f, R, og, ren);
    for (int ch = 0; ch < 3; ++ch) { r[ch] = I[ch] - ren[ch]; w[ch] = robust_weight(c, r[ch]); }
    return 1;
}

/* rhoJacobian: PsOptimizerJa.cpp:118-122 (scalar) / LedOptimizerJa.cpp:85-99 (3-vector) */
static void rho_jacobian(const orc_ctx* c, int lin, int f, const float R[9], const float t[3], float J[3]) {
    float gr[3] = {c->gx[lin], c->gy[lin], c->gz[lin]}, n[3]; normalized3(gr, n);
    if (c->set.model == PSGSDF_LED) {
        int idx[3]; line2idx(c, lin, idx); float xv[3]; voxel2world(c, idx, xv);
        float tmp[3], point[3]; float d = c->dist[lin];
        for (int a = 0; a < 3; ++a) tmp[a] = (xv[a] - d * n[a]) - t[a];
        mulT3(R, tmp, point);
        float Rp[3]; mul3(R, point, Rp);
        float refl = dot3(n, Rp);
        refl /= (float)pow((double)norm3(point), 3);
        for (int ch = 0; ch < 3; ++ch) J[ch] = refl * c->light[ch];
    } else {
        float sh[MAXB]; SH(n, sh_order(c), sh);
        float j = -dotn(c->light + (size_t)f * MAXB, sh, c->basis);
        J[0] = J[1] = J[2] = j;
    }
}

/* albedo normal equations: H (3S), b (3S) float, from double sums.
 * optimizeAlbedoAll, PsOptimizer.cpp:85-121 / LedOptimizer.cpp:162-196 */
static void albedo_system(const orc_ctx* c, float* H, float* b, double* e_in, long long* nobs_out) {
    double E = 0; long long nobs = 0;
#pragma omp parallel for reduction(+ : E, nobs) schedule(static) num_threads(c->threads)
    for (int j = c->row0; j < c->row1; ++j) {
        int lin = c->band[j];
        double Hd[3] = {0, 0, 0}, bd[3] = {0, 0, 0};
        for (int f = 0; f < c->F; ++f) {
            if (!vis_bit(c, lin, f)) continue;
            float R[9], t[3]; pose_Rt(c, f, R, t);
            float r[3], w[3]; obs_geom og;
            if (!residual_obs(c, lin, f, R, t, r, w, &og)) continue;
            float J[3]; rho_jacobian(c, lin, f, R, t, J);
            float l = 0;
            for (int ch = 0; ch < 3; ++ch) {
                float jw = J[ch] * w[ch];
                Hd[ch] += (double)(jw * J[ch]); bd[ch] += (double)(jw * r[ch]);
                l += robust_loss(c, r[ch]);
            }
            E += (double)l; nobs++;
        }
        for (int ch = 0; ch < 3; ++ch) { H[3 * j + ch] = (float)Hd[ch]; b[3 * j + ch] = (float)bd[ch]; }
    }
    if (e_in) *e_in = c->S ? E / c->S : 0.0;
    if (nobs_out) *nobs_out = nobs;
}

typedef struct { long long key; double v; } coo_t;
static int coo_cmp(const void* a, const void* b) { long long x = ((const coo_t*)a)->key, y = ((const coo_t*)b)->key; return x < y ? -1 : (x > y ? 1 : 0); }
typedef struct { int n; const int* rowptr; const int* colidx; const float* val; float damping; } csr_mv_ctx;
static void csr_mv(void* user, const float* p, float* out) {
    const csr_mv_ctx* m = (const csr_mv_ctx*)user;
    for (int i = 0; i < m->n; ++i) { double acc = 0;
        for (int k = m->rowptr[i]; k < m->rowptr[i + 1]; ++k) { float v = m->val[k]; if (m->colidx[k] == i && m->damping != 0.f) v += m->damping * v; acc += (double)v * (double)p[m->colidx[k]]; }
        out[i] = (float)acc; }
}
/* optimizeAlbedoAll with the ||grad rho|| regulariser (PsOptimizer.cpp:85-121, Optimizer.cpp:593-647): H = J^T W J (diagonal)
 * + reg_rho Jr^T Jr over the 3S unknowns (3*row + channel), Eigen ConjugateGradient, update only on success for the SH
 * optimiser (PsOptimizer.cpp:117-119), always for the LED one (LedOptimizer.cpp:195).  Quirk (ref_quirks): the blue
 * self-entry of Jr sits in the GREEN column (Optimizer.cpp:617 `Tri2(3*row+2, 3*row+1, ...)`). */
static int step_albedo_reg(orc_ctx* c, const float* Hd, const float* bd, float* delta, cg_result* cr) {
    const int S = c->S, n = 3 * S;
    coo_t* coo = (coo_t*)malloc(sizeof(coo_t) * ((size_t)48 * S + (size_t)n + 1)); size_t nc = 0;
    double* rhs = (double*)calloc(n + 1, sizeof(double));
    for (int i = 0; i < n; ++i) { coo[nc].key = (long long)i * n + i; coo[nc].v = (double)Hd[i]; nc++; rhs[i] = (double)bd[i]; }
    for (int j = 0; j < S; ++j) {
        float J[4][3], res[3]; long nb[3]; albedo_reg_jacobian(c, c->band[j], J, res, nb);
        for (int ch = 0; ch < 3; ++ch) {
            int col[4]; float e[4]; int m = 0;
            col[m] = 3 * j + ((ch == 2 && c->set.ref_quirks) ? 1 : ch); e[m] = J[0][ch]; m++;
            for (int a = 0; a < 3; ++a) { int r = (nb[a] != (long)c->band[j]) ? band_find(c, (size_t)nb[a]) : -1; if (r >= 0) { col[m] = 3 * r + ch; e[m] = J[a + 1][ch]; m++; } }
            for (int p = 0; p < m; ++p) { rhs[col[p]] += (double)c->reg_r * (double)e[p] * (double)res[ch];
                for (int q = 0; q < m; ++q) { coo[nc].key = (long long)col[p] * n + col[q]; coo[nc].v = (double)c->reg_r * (double)e[p] * (double)e[q]; nc++; } }
        }
    }
    qsort(coo, nc, sizeof(coo_t), coo_cmp);
    int* rowptr = (int*)calloc(n + 2, sizeof(int)); int* colidx = (int*)malloc(sizeof(int) * (nc + 1)); float* val = (float*)malloc(sizeof(float) * (nc + 1));
    float* diag = (float*)calloc(n + 1, sizeof(float)); float* rf = (float*)malloc(sizeof(float) * (n + 1));
    size_t nnz = 0;
    for (size_t i = 0; i < nc;) { size_t k = i; double v = 0; while (k < nc && coo[k].key == coo[i].key) { v += coo[k].v; ++k; }
        int row = (int)(coo[i].key / n), cl = (int)(coo[i].key % n); colidx[nnz] = cl; val[nnz] = (float)v; rowptr[row + 1]++; if (row == cl) diag[row] = (float)v; nnz++; i = k; }
    for (int i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    for (int i = 0; i < n; ++i) { rf[i] = (float)rhs[i]; if (c->set.damping != 0.f) diag[i] += c->set.damping * diag[i]; }
    csr_mv_ctx m = {n, rowptr, colidx, val, c->set.damping};
    *cr = eigen_cg(n, csr_mv, &m, diag, rf, delta, c->set.cg_max_it);
    free(coo); free(rhs); free(rowptr); free(colidx); free(val); free(diag); free(rf);
    return 0;
}

static int step_albedo(orc_ctx* c, psgsdf_step_stats* st) {
    int n = 3 * c->S;
    float* H = (float*)malloc(sizeof(float) * (n > 0 ? n : 1)); float* b = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
    double e_in; long long nobs;
    albedo_system(c, H, b, &e_in, &nobs);
    float damping = c->set.damping;
    long long count = 0;
    cg_result cr = {1, 0.0, 1}; int apply = 1;
    float* delta = NULL;
    if (c->reg_r != 0.0f) {
        if (c->n_ranks > 1) { free(H); free(b); return PSGSDF_ERR_UNSUPPORTED; }
        delta = (float*)calloc(n + 1, sizeof(float));
        step_albedo_reg(c, H, b, delta, &cr);
        if (c->set.model != PSGSDF_LED && !cr.success) apply = 0;      /* PsOptimizer.cpp:117-119 */
    }
    /* without the regulariser the system is diagonal: Jacobi-PCG is exact after one step: delta = b / H */
    if (apply) for (int j = c->row0; j < c->row1; ++j) {
        int lin = c->band[j];
        float* rho[3] = {&c->r[lin], &c->g[lin], &c->b[lin]};
        for (int ch = 0; ch < 3; ++ch) {
            float dl;
            if (delta) dl = delta[3 * j + ch];
            else { float h = H[3 * j + ch];
                if (damping != 0.0f) h += damping * h; /* PsOptimizer.cpp:103-105 */
                dl = (h != 0.f) ? b[3 * j + ch] / h : 0.f; }
            float v = *rho[ch] - dl; /* updateAlbedo, OptimizerAux.cpp:120-150 */
            if (v > 0.0f && v < 1.0f) { *rho[ch] = v; count++; }
        }
    }
    if (st) { st->block = PSGSDF_ALBEDO; st->cg_iters = cr.iters; st->cg_converged = cr.success; st->applied = apply; st->e_in = e_in; st->cg_error = cr.error; st->n_accepted = count; st->n_obs = nobs; }
    free(H); free(b); free(delta);
    return 0;
}

/* light normal equations per frame (PS: basis x basis per frame, lightJacobian
 * PsOptimizerJa.cpp:132-143,323-371; LED: one 3-vector, LedOptimizerJa.cpp:101-115,299-346). */
static void light_system(const orc_ctx* c, double* H, double* b, double* e_in, long long* nobs_out) {
    int nb = c->set.model == PSGSDF_LED ? 1 : c->F, n = c->basis;
    memset(H, 0, sizeof(double) * nb * n * n); memset(b, 0, sizeof(double) * nb * n);
    double E = 0; long long nobs = 0;
    for (int f = 0; f < c->F; ++f) {
        float R[9], t[3]; pose_Rt(c, f, R, t);
        double* Hf = H + (c->set.model == PSGSDF_LED ? 0 : (size_t)f * n * n);
        double* bf = b + (c->set.model == PSGSDF_LED ? 0 : (size_t)f * n);
        for (int j = c->row0; j < c->row1; ++j) {
            int lin = c->band[j];
            if (!vis_bit(c, lin, f)) continue;
            float r[3], w[3]; obs_geom og;
            if (!residual_obs(c, lin, f, R, t, r, w, &og)) continue;
            float rho[3] = {c->r[lin], c->g[lin], c->b[lin]};
            float l = 0;
            if (c->set.model == PSGSDF_LED) {
                /* LightJacobian: reflectance * rho, column = channel */
                float gr[3] = {c->gx[lin], c->gy[lin], c->gz[lin]}, nn[3]; normalized3(gr, nn);
                float Rp[3]; mul3(R, og.point, Rp);
                float refl = dot3(nn, Rp); refl /= (float)pow((double)norm3(og.point), 3);
                for (int ch = 0; ch < 3; ++ch) {
                    float J = refl * rho[ch]; float jw = J * w[ch];
                    Hf[ch * 3 + ch] += (double)(jw * J); bf[ch] += (double)(jw * r[ch]);
                    l += robust_loss(c, r[ch]);
                }
            } else {
                float gr[3] = {c->gx[lin], c->gy[lin], c->gz[lin]}, nn[3]; normalized3(gr, nn);
                float sh[MAXB]; SH(nn, sh_order(c), sh);
                for (int ch = 0; ch < 3; ++ch) {
                    float J[MAXB]; for (int i = 0; i < n; ++i) J[i] = -rho[ch] * sh[i];
                    for (int i = 0; i < n; ++i) {
                        float jw = J[i] * w[ch];
                        for (int k = 0; k < n; ++k) Hf[i * n + k] += (double)(jw * J[k]);
                        bf[i] += (double)(jw * r[ch]);
                    }
                    l += robust_loss(c, r[ch]);
                }
            }
            E += (double)l; nobs++;
        }
    }
    if (e_in) *e_in = c->S ? E / c->S : 0.0;
    if (nobs_out) *nobs_out = nobs;
}

static int light_finish(orc_ctx* c, double* H, double* b, double e_in, long long nobs, psgsdf_step_stats* st);
static int step_light(orc_ctx* c, psgsdf_step_stats* st) {
    int led = c->set.model == PSGSDF_LED;
    int nb = led ? 1 : c->F, n = c->basis;
    double* H = (double*)malloc(sizeof(double) * nb * n * n); double* b = (double*)malloc(sizeof(double) * nb * n);
    double e_in; long long nobs;
    light_system(c, H, b, &e_in, &nobs);
    return light_finish(c, H, b, e_in, nobs, st);
}
static int light_finish(orc_ctx* c, double* H, double* b, double e_in, long long nobs, psgsdf_step_stats* st) {
    int led = c->set.model == PSGSDF_LED;
    int nb = led ? 1 : c->F, n = c->basis;
    float* Hf = (float*)malloc(sizeof(float) * nb * n * n); float* bf = (float*)malloc(sizeof(float) * nb * n); float* x = (float*)malloc(sizeof(float) * nb * n);
    for (int i = 0; i < nb * n * n; ++i) Hf[i] = (float)H[i];
    for (int i = 0; i < nb * n; ++i) bf[i] = (float)b[i];
    if (led && c->set.damping != 0.0f) /* LedOptimizer.cpp:144-146; PS light has no damping */
        for (int i = 0; i < n; ++i) Hf[i * n + i] += c->set.damping * Hf[i * n + i];
    cg_result cr = solve_blockdiag(c, nb, n, Hf, bf, x);
    if (led) for (int ch = 0; ch < 3; ++ch) c->light[ch] -= x[ch]; /* LedOptimizer.cpp:159 */
    else for (int f = 0; f < c->F; ++f) for (int i = 0; i < n; ++i) c->light[(size_t)f * MAXB + i] -= x[f * n + i]; /* PsOptimizer.cpp:199-201 */
    c->fs_iters[0] = cr.iters; c->fs_ok[0] = cr.success; c->fs_applied[0] = 1; c->fs_err[0] = cr.error;
    if (st) { st->block = PSGSDF_LIGHT; st->cg_iters = cr.iters; st->cg_converged = cr.success; st->applied = 1; st->e_in = e_in; st->cg_error = cr.error; st->n_accepted = nb; st->n_obs = nobs; }
    free(H); free(b); free(Hf); free(bf); free(x);
    return 0;
}

/* G = image_grad(3x2) * pi_grad(2x3), PsOptimizerJa.cpp:78-90 */
static void image_pi_grad(const orc_ctx* c, int f, const obs_geom* og, float G[9]) {
    float gu[3], gv[3];
    image_gradient(c, f, og->n, og->m, 0, gu);
    image_gradient(c, f, og->n, og->m, 1, gv);
    float z_inv = (float)(1.0 / (double)og->point[2]);
    float z_inv_sq = z_inv * z_inv;
    float p00 = c->fx * z_inv, p02 = -c->fx * og->point[0] * z_inv_sq, p11 = c->fy * z_inv, p12 = -c->fy * og->point[1] * z_inv_sq;
    for (int ch = 0; ch < 3; ++ch) {
        G[ch * 3 + 0] = gu[ch] * p00 + gv[ch] * 0.0f;
        G[ch * 3 + 1] = gu[ch] * 0.0f + gv[ch] * p11;
        G[ch * 3 + 2] = gu[ch] * p02 + gv[ch] * p12;
    }
}

/* poseJacobian: PsOptimizerJa.cpp:61-115 / LedOptimizerJa.cpp:32-81.  J is 3x6 row-major. */
static int pose_jacobian(const orc_ctx* c, int lin, int f, const float R[9], const float t[3], float J[18]) {
    obs_geom og;
    if (!project(c, lin, R, t, 1, &og)) return 0;
    float G[9]; image_pi_grad(c, f, &og, G);
    /* -G * R^T */
    for (int ch = 0; ch < 3; ++ch) for (int k = 0; k < 3; ++k) {
        float s = (G[ch * 3 + 0] * R[k * 3 + 0] + G[ch * 3 + 1] * R[k * 3 + 1]) + G[ch * 3 + 2] * R[k * 3 + 2];
        J[ch * 6 + k] = -s;
    }
    /* G * skew(point) */
    const float* p = og.point;
    float sk[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
    for (int ch = 0; ch < 3; ++ch) for (int k = 0; k < 3; ++k)
        J[ch * 6 + 3 + k] = (G[ch * 3 + 0] * sk[0 * 3 + k] + G[ch * 3 + 1] * sk[1 * 3 + k]) + G[ch * 3 + 2] * sk[2 * 3 + k];
    if (c->set.model == PSGSDF_LED) {
        float l = (float)pow((double)norm3(p), 3);
        float rho[3] = {c->r[lin], c->g[lin], c->b[lin]};
        for (int ch = 0; ch < 3; ++ch) {
            float s = -(rho[ch] * c->light[ch] / l);
            for (int k = 0; k < 3; ++k) J[ch * 6 + k] += s * og.gn[k]; /* LED_t_grad */
            /* LED_R_grad is identically zero: skew(p)*p = 0 (LedOptimizerJa.cpp:71) */
        }
    }
    return 1;
}

static void pose_system(const orc_ctx* c, double* H, double* b, double* e_in, long long* nobs_out) {
    memset(H, 0, sizeof(double) * c->F * 36); memset(b, 0, sizeof(double) * c->F * 6);
    double E = 0; long long nobs = 0;
#pragma omp parallel for reduction(+ : E, nobs) schedule(static) num_threads(c->threads)
    for (int f = 0; f < c->F; ++f) {
        float R[9], t[3]; pose_Rt(c, f, R, t);
        double* Hf = H + (size_t)f * 36; double* bf = b + (size_t)f * 6;
        for (int j = c->row0; j < c->row1; ++j) {
            int lin = c->band[j];
            if (!vis_bit(c, lin, f)) continue;
            float r[3], w[3]; obs_geom og;
            int ok_r = residual_obs(c, lin, f, R, t, r, w, &og);
            float J[18];
            int ok_j = pose_jacobian(c, lin, f, R, t, J);
            if (ok_r) { float l = 0; for (int ch = 0; ch < 3; ++ch) l += robust_loss(c, r[ch]); E += (double)l; nobs++; }
            if (!ok_r || !ok_j) continue; /* W = 0 or J row empty */
            for (int ch = 0; ch < 3; ++ch) for (int i = 0; i < 6; ++i) {
                float jw = J[ch * 6 + i] * w[ch];
                for (int k = 0; k < 6; ++k) Hf[i * 6 + k] += (double)(jw * J[ch * 6 + k]);
                bf[i] += (double)(jw * r[ch]);
            }
        }
    }
    if (e_in) *e_in = c->S ? E / c->S : 0.0;
    if (nobs_out) *nobs_out = nobs;
}

static int pose_finish(orc_ctx* c, double* H, double* b, double e_in, long long nobs, psgsdf_step_stats* st);
static int step_pose(orc_ctx* c, psgsdf_step_stats* st) {
    int nb = c->F;
    double* H = (double*)malloc(sizeof(double) * nb * 36); double* b = (double*)malloc(sizeof(double) * nb * 6);
    double e_in; long long nobs;
    pose_system(c, H, b, &e_in, &nobs);
    return pose_finish(c, H, b, e_in, nobs, st);
}
static int pose_finish(orc_ctx* c, double* H, double* b, double e_in, long long nobs, psgsdf_step_stats* st) {
    int nb = c->F, n = 6;
    float* Hf = (float*)malloc(sizeof(float) * nb * 36); float* bf = (float*)malloc(sizeof(float) * nb * 6); float* x = (float*)malloc(sizeof(float) * nb * 6);
    for (int i = 0; i < nb * 36; ++i) Hf[i] = (float)H[i];
    for (int i = 0; i < nb * 6; ++i) bf[i] = (float)b[i];
    if (c->set.damping != 0.0f) for (int k = 0; k < nb; ++k) for (int i = 0; i < 6; ++i) Hf[k * 36 + i * 6 + i] += c->set.damping * Hf[k * 36 + i * 6 + i];
    cg_result cr = solve_blockdiag(c, nb, n, Hf, bf, x);
    int apply = 1;
    if (c->set.model == PSGSDF_LED && c->set.ref_quirks && !cr.success) apply = 0; /* LedOptimizer.cpp:271-273 */
    if (apply) for (int f = 0; f < c->F; ++f) { /* updatePose, OptimizerAux.cpp:190-205 */
        float* P = c->poses + 16 * f; const float* xi = x + 6 * f;
        float R[9], t[3]; pose_Rt(c, f, R, t);
        float mw[3] = {-xi[3], -xi[4], -xi[5]}, E3[9]; so3_exp(mw, E3);
        for (int i = 0; i < 3; ++i) {
            P[i * 4 + 3] = t[i] - xi[i];
            for (int k = 0; k < 3; ++k) P[i * 4 + k] = (R[i * 3 + 0] * E3[0 * 3 + k] + R[i * 3 + 1] * E3[1 * 3 + k]) + R[i * 3 + 2] * E3[2 * 3 + k];
        }
    }
    c->fs_iters[1] = cr.iters; c->fs_ok[1] = cr.success; c->fs_applied[1] = apply; c->fs_err[1] = cr.error;
    if (st) { st->block = PSGSDF_POSE; st->cg_iters = cr.iters; st->cg_converged = cr.success; st->applied = apply; st->e_in = e_in; st->cg_error = cr.error; st->n_accepted = apply ? nb : 0; st->n_obs = nobs; }
    free(H); free(b); free(Hf); free(bf); free(x);
    return 0;
}

/* distJacobian per observation: PsOptimizerJa.cpp:160-289 / LedOptimizerJa.cpp:117-218.
 * J[k][ch], k = 0..3 (self, x-, y-, z-stencil neighbour); dir returned. */
static int dist_jacobian(const orc_ctx* c, int lin, int f, const float R[9], const float t[3], float J[4][3], float dir[3]) {
    obs_geom og;
    if (!project(c, lin, R, t, 1, &og)) return 0;
    float G[9]; image_pi_grad(c, f, &og, G);
    float grad[3]; dist_grad(c, lin, grad, dir);
    float dn[4][3];
    normal_jacobian(c, grad, dir, dn[0]);
    int led = c->set.model == PSGSDF_LED;
    for (int a = 0; a < 3; ++a) {
        float nd[3] = {0, 0, 0};
        /* SH: n_d1[a] -= dir[a] (PsOptimizerJa.cpp:200-210); LED: += (LedOptimizerJa.cpp:157-167, B6) */
        if (led && c->set.ref_quirks) nd[a] += dir[a]; else nd[a] -= dir[a];
        normal_jacobian(c, grad, nd, dn[a + 1]);
    }
    float d = c->dist[lin];
    float dx[4][3];
    for (int a = 0; a < 3; ++a) dx[0][a] = -og.gn[a] - d * dn[0][a];
    for (int k = 1; k < 4; ++k) for (int a = 0; a < 3; ++a) dx[k][a] = -d * dn[k][a];
    /* dI_k = G * R^T * dx_k  (Eigen evaluates (G*R^T) first, then * vector) */
    float GRt[9];
    for (int ch = 0; ch < 3; ++ch) for (int k = 0; k < 3; ++k)
        GRt[ch * 3 + k] = (G[ch * 3 + 0] * R[k * 3 + 0] + G[ch * 3 + 1] * R[k * 3 + 1]) + G[ch * 3 + 2] * R[k * 3 + 2];
    float dI[4][3];
    for (int k = 0; k < 4; ++k) mul3(GRt, dx[k], dI[k]);
    float rho[3] = {c->r[lin], c->g[lin], c->b[lin]};
    if (!led) {
        const float* l = c->light + (size_t)f * MAXB;
        if (sh_order(c) == 1) {
            for (int k = 0; k < 4; ++k) for (int ch = 0; ch < 3; ++ch) {
                float dr[3] = {rho[ch] * l[1], rho[ch] * l[2], rho[ch] * l[3]};
                J[k][ch] = dI[k][ch] - dot3(dr, dn[k]);
            }
        } else {
            float nh[3]; normalized3(grad, nh);
            float D[3][9] = {{0, 1, 0, 0, nh[1], nh[2], 0, 2 * nh[0], 2 * nh[0]},
                             {0, 0, 1, 0, nh[0], 0, nh[2], -2 * nh[1], 0},
                             {0, 0, 0, 1, 0, nh[0], nh[1], 0, -2 * nh[2]}};
            for (int k = 0; k < 4; ++k) {
                float dsh[9];
                for (int i = 0; i < 9; ++i) dsh[i] = (D[0][i] * dn[k][0] + D[1][i] * dn[k][1]) + D[2][i] * dn[k][2];
                for (int ch = 0; ch < 3; ++ch) {
                    float s = 0; for (int i = 0; i < 9; ++i) s += (rho[ch] * l[i]) * dsh[i];
                    J[k][ch] = dI[k][ch] - s;
                }
            }
        }
    } else {
        float Rp[3]; mul3(R, og.point, Rp);
        float nh[3]; normalized3(grad, nh);
        float pn = norm3(og.point);
        float radius = (float)pow((double)pn, 3);
        float p5 = (float)pow((double)pn, 5);
        float nRp = dot3(nh, Rp);
        for (int k = 0; k < 4; ++k) {
            float dm = dot3(dn[k], Rp) + dot3(nh, dx[k]);
            /* point^T * R^T * dx = (R*point) . dx */
            float tmp[3]; mulT3(R, dx[k], tmp); /* R^T dx */
            float dm2 = -3 * dot3(og.point, tmp) / p5;
            dm = dm / radius + dm2 * nRp;
            for (int ch = 0; ch < 3; ++ch) J[k][ch] = dI[k][ch] + (rho[ch] * c->light[ch]) * dm;
        }
    }
    return 1;
}

/* Optimizer.cpp:196-218 distRegJacobian(v, idx, Jr_d) + residual (477-537) */
static void eikonal_row(const orc_ctx* c, int lin, float Jr[4], float* res, float dir[3]) {
    float grad[3]; dist_grad(c, lin, grad, dir);
    float n_d[3] = {-c->vs_inv * dir[0], -c->vs_inv * dir[1], -c->vs_inv * dir[2]};
    Jr[0] = dot3(grad, n_d);
    for (int a = 0; a < 3; ++a) Jr[a + 1] = grad[a] * (c->vs_inv * dir[a]);
    float gn = norm3(grad);
    if (gn > 0.0f) for (int k = 0; k < 4; ++k) Jr[k] /= gn;
    *res = gn - 1;
}

/* The assembled distance system: per band voxel a 4x4 block over {self, 3 stencil neighbours};
 * H = sum_j P_j^T B_j P_j.  Stored explicitly as sorted COO -> CSR in float. */
typedef struct dist_sys {
    int S;
    int* cols;     /* S*4 row indices of the stencil slots (-1 = column dropped) */
    double* B;     /* S*16 */
    double* g;     /* S*4  */
    /* CSR */
    int* rowptr; int* colidx; float* val; float* diag; float* rhs;
} dist_sys;

static void dist_assemble(const orc_ctx* c, dist_sys* s);

static void dist_sys_free(dist_sys* s) { free(s->cols); free(s->B); free(s->g); free(s->rowptr); free(s->colidx); free(s->val); free(s->diag); free(s->rhs); memset(s, 0, sizeof(*s)); }

/* optimizeDistAll assembly, PsOptimizer.cpp:124-154 / LedOptimizer.cpp:198-228 */
static void dist_system(const orc_ctx* c, int normal_reg, int laplacian_reg, dist_sys* s, double* e_in, long long* nobs_out) {
    int S = c->S;
    memset(s, 0, sizeof(*s));
    s->S = S;
    s->cols = (int*)malloc(sizeof(int) * 4 * (S + 1)); s->B = (double*)calloc((size_t)16 * (S + 1), sizeof(double)); s->g = (double*)calloc((size_t)4 * (S + 1), sizeof(double));
    long stride[3] = {1, c->dim[0], (long)c->dim[0] * c->dim[1]};
    double E = 0; long long nobs = 0;
#pragma omp parallel for reduction(+ : E, nobs) schedule(static) num_threads(c->threads)
    for (int j = c->row0; j < c->row1; ++j) {
        int lin = c->band[j];
        float gtmp[3], dir[3]; dist_grad(c, lin, gtmp, dir);
        int* cols = s->cols + 4 * j; double* B = s->B + 16 * j; double* g = s->g + 4 * j;
        cols[0] = j;
        for (int a = 0; a < 3; ++a) {
            long ln = (long)lin + (long)dir[a] * stride[a];
            cols[a + 1] = (ln >= 0 && (size_t)ln < c->nvox) ? band_find(c, (size_t)ln) : -1;
        }
        for (int f = 0; f < c->F; ++f) {
            if (!vis_bit(c, lin, f)) continue;
            float R[9], t[3]; pose_Rt(c, f, R, t);
            float r[3], w[3]; obs_geom og;
            int ok_r = residual_obs(c, lin, f, R, t, r, w, &og);
            float J[4][3], d2[3];
            int ok_j = dist_jacobian(c, lin, f, R, t, J, d2);
            if (ok_r) { float l = 0; for (int ch = 0; ch < 3; ++ch) l += robust_loss(c, r[ch]); E += (double)l; nobs++; }
            if (!ok_r || !ok_j) continue;
            for (int ch = 0; ch < 3; ++ch) for (int a = 0; a < 4; ++a) {
                float jw = J[a][ch] * w[ch];
                for (int bq = 0; bq < 4; ++bq) B[a * 4 + bq] += (double)(jw * J[bq][ch]);
                g[a] += (double)(jw * r[ch]);
            }
        }
        if (normal_reg) {
            float Jr[4], res, d3[3]; eikonal_row(c, lin, Jr, &res, d3);
            for (int a = 0; a < 4; ++a) {
                for (int bq = 0; bq < 4; ++bq) B[a * 4 + bq] += (double)(c->reg_n * (Jr[a] * Jr[bq]));
                g[a] += (double)(c->reg_n * (Jr[a] * res));
            }
        }
        if (laplacian_reg) { /* diagonal only (B3), Optimizer.cpp:540-590 */
            float vs2 = c->vs_inv * c->vs_inv; float Jl = -6 * vs2; float res = dist_laplacian(c, lin);
            B[0] += (double)(c->reg_l * (Jl * Jl)); g[0] += (double)(c->reg_l * (Jl * res));
        }
    }
    if (e_in) *e_in = S ? E / S : 0.0;
    if (nobs_out) *nobs_out = nobs;
    /* the engine keeps the per-voxel blocks in float32 planes (and exchanges them between ranks in that form):
     * round here too, absent columns zeroed, so that 1-rank and n-rank runs assemble from identical numbers */
    for (int j = c->row0; j < c->row1; ++j) for (int a = 0; a < 4; ++a) {
        int ea = s->cols[4 * j + a] >= 0;
        s->g[4 * j + a] = ea ? (double)(float)s->g[4 * j + a] : 0.0;
        for (int bq = 0; bq < 4; ++bq) { int eb = s->cols[4 * j + bq] >= 0; s->B[16 * j + a * 4 + bq] = (ea && eb) ? (double)(float)s->B[16 * j + a * 4 + bq] : 0.0; }
    }
    if (c->n_ranks > 1) return;   /* slab mode: blocks are exchanged, then dist_assemble() */
    dist_assemble(c, s);
}
/* assemble H rows [row0,row1) from the blocks of every contributing voxel (own + halo): drop absent columns */
static void dist_assemble(const orc_ctx* c, dist_sys* s) {
    int S = c->S;
    long stride[3] = {1, c->dim[0], (long)c->dim[0] * c->dim[1]};
    int jlo = c->row0 - c->need[0] < 0 ? 0 : c->row0 - c->need[0], jhi = c->row1 + c->need[1] > S ? S : c->row1 + c->need[1];
    if (c->n_ranks > 1) for (int j = jlo; j < jhi; ++j) {   /* stencil columns of halo voxels are static: recompute */
        if (j >= c->row0 && j < c->row1) continue;
        int lin = c->band[j]; int idx[3]; line2idx(c, lin, idx);
        s->cols[4 * j] = j;
        for (int a = 0; a < 3; ++a) { float dir = valid_forward(c, lin, idx, a) ? 1.0f : -1.0f; long ln = (long)lin + (long)dir * stride[a]; s->cols[4 * j + a + 1] = (ln >= 0 && (size_t)ln < c->nvox) ? band_find(c, (size_t)ln) : -1; }
    }
    size_t ncoo = 0; coo_t* coo = (coo_t*)malloc(sizeof(coo_t) * 16 * (size_t)(S + 1));
    double* rhs = (double*)calloc(S + 1, sizeof(double));
    for (int j = jlo; j < jhi; ++j) for (int a = 0; a < 4; ++a) {
        int ra = s->cols[4 * j + a]; if (ra < c->row0 || ra >= c->row1) continue;
        rhs[ra] += s->g[4 * j + a];
        for (int bq = 0; bq < 4; ++bq) { int cb = s->cols[4 * j + bq]; if (cb < 0) continue; coo[ncoo].key = (long long)ra * S + cb; coo[ncoo].v = s->B[16 * j + a * 4 + bq]; ncoo++; }
    }
    qsort(coo, ncoo, sizeof(coo_t), coo_cmp);
    s->rowptr = (int*)calloc(S + 2, sizeof(int)); s->colidx = (int*)malloc(sizeof(int) * (ncoo + 1)); s->val = (float*)malloc(sizeof(float) * (ncoo + 1));
    s->diag = (float*)calloc(S + 1, sizeof(float)); s->rhs = (float*)malloc(sizeof(float) * (S + 1));
    size_t nnz = 0;
    for (size_t i = 0; i < ncoo;) {
        size_t k = i; double v = 0;
        while (k < ncoo && coo[k].key == coo[i].key) { v += coo[k].v; ++k; }
        int row = (int)(coo[i].key / S), col = (int)(coo[i].key % S);
        s->colidx[nnz] = col; s->val[nnz] = (float)v; s->rowptr[row + 1]++;
        if (row == col) s->diag[row] = (float)v;
        nnz++; i = k;
    }
    for (int i = 0; i < S; ++i) s->rowptr[i + 1] += s->rowptr[i];
    for (int i = 0; i < S; ++i) s->rhs[i] = (float)rhs[i];
    free(coo); free(rhs);
}
typedef struct { const dist_sys* s; float damping; } dist_mv_ctx;
static void dist_mv(void* user, const float* p, float* out) {
    const dist_mv_ctx* m = (const dist_mv_ctx*)user; const dist_sys* s = m->s;
    for (int i = 0; i < s->S; ++i) {
        double acc = 0;
        if (s->rowptr[i + 1] == s->rowptr[i]) { out[i] = 0.f; continue; }
        for (int k = s->rowptr[i]; k < s->rowptr[i + 1]; ++k) {
            float v = s->val[k];
            if (s->colidx[k] == i && m->damping != 0.f) v += m->damping * v; /* H.diagonal() += damping*H.diagonal() */
            acc += (double)v * (double)p[s->colidx[k]];
        }
        out[i] = (float)acc;
    }
}

/* updateGrad, OptimizerAux.cpp:152-160 */
static void update_grad(orc_ctx* c) {
    float* ng = (float*)malloc(sizeof(float) * 3 * (c->S + 1));
    for (int j = c->row0; j < c->row1; ++j) { float d[3]; dist_grad(c, c->band[j], ng + 3 * j, d); }
    for (int j = c->row0; j < c->row1; ++j) { int lin = c->band[j]; c->gx[lin] = ng[3 * j]; c->gy[lin] = ng[3 * j + 1]; c->gz[lin] = ng[3 * j + 2]; }
    free(ng);
}

static int step_dist(orc_ctx* c, int laplacian_reg, psgsdf_step_stats* st) {
    int normal_reg = c->reg_n != 0.0f;
    dist_sys s; double e_in; long long nobs;
    dist_system(c, normal_reg, laplacian_reg, &s, &e_in, &nobs);
    int S = c->S;
    float* x = (float*)calloc(S + 1, sizeof(float));
    float* dd = (float*)malloc(sizeof(float) * (S + 1));
    for (int i = 0; i < S; ++i) { dd[i] = s.diag[i]; if (c->set.damping != 0.f) dd[i] += c->set.damping * dd[i]; }
    dist_mv_ctx m = {&s, c->set.damping};
    cg_result cr = eigen_cg(S, dist_mv, &m, dd, s.rhs, x, c->set.cg_max_it);
    int apply = 1;
    if (c->set.model != PSGSDF_LED && c->set.ref_quirks && !cr.success) apply = 0; /* PsOptimizer.cpp:168-170 (B8) */
    long long count = 0;
    if (apply) { /* updateDist, OptimizerAux.cpp:162-188 */
        for (int j = c->row0; j < c->row1; ++j) {
            float d = x[j];
            if ((double)fabsf(d) < sqrt(3.0) * (double)c->vs) { c->dist[c->band[j]] -= d; count++; }
        }
        update_grad(c);
    }
    if (st) { st->block = PSGSDF_DIST; st->cg_iters = cr.iters; st->cg_converged = cr.success; st->applied = apply; st->e_in = e_in; st->cg_error = cr.error; st->n_accepted = count; st->n_obs = nobs; }
    free(x); free(dd); dist_sys_free(&s);
    return 0;
}

/* ------------------------------------------------------------------ init */

/* Optimizer.cpp:50-81 initAlbedo */
static void init_albedo(orc_ctx* c) {
    for (int j = c->row0; j < c->row1; ++j) {
        int lin = c->band[j]; int count = 0; float rho[3] = {0, 0, 0};
        for (int f = 0; f < c->F; ++f) {
            if (!vis_bit(c, lin, f)) continue;
            float R[9], t[3]; pose_Rt(c, f, R, t); float I[3];
            if (!get_intensity(c, lin, f, R, t, I, NULL)) continue;
            rho[0] += I[0]; rho[1] += I[1]; rho[2] += I[2]; count++;
        }
        if (count) { c->r[lin] = rho[0] / (float)count; c->g[lin] = rho[1] / (float)count; c->b[lin] = rho[2] / (float)count; }
    }
}
/* PsOptimizer.cpp:25-42 / LedOptimizer.cpp:25-36,76-112 */
static void init_light(orc_ctx* c) {
    free(c->light);
    if (c->set.model == PSGSDF_LED) {
        c->basis = 3; c->light = (float*)malloc(sizeof(float) * MAXB);
        c->light[0] = c->light[1] = c->light[2] = 1.0f;
        double I[3] = {0, 0, 0}, Rr[3] = {0, 0, 0};
        for (int j = c->row0; j < c->row1; ++j) { int lin = c->band[j];
            for (int f = 0; f < c->F; ++f) {
                if (!vis_bit(c, lin, f)) continue;
                float R[9], t[3]; pose_Rt(c, f, R, t); float in[3]; obs_geom og;
                if (!get_intensity(c, lin, f, R, t, in, &og)) continue;
                float ren[3]; rendered_intensity(c, lin, f, R, &og, ren);
                for (int ch = 0; ch < 3; ++ch) { I[ch] += in[ch]; Rr[ch] += ren[ch]; }
            } }
        for (int ch = 0; ch < 3; ++ch) { c->mg_scal[ch] = I[ch]; c->mg_scal[3 + ch] = Rr[ch]; }
        if (c->n_ranks == 1) for (int ch = 0; ch < 3; ++ch) c->light[ch] = (float)I[ch] / (float)Rr[ch];
    } else {
        c->basis = c->set.model == PSGSDF_SH2 ? 9 : 4;
        c->light = (float*)calloc((size_t)(c->F > 0 ? c->F : 1) * MAXB, sizeof(float));
        for (int f = 0; f < c->F; ++f) {
            float R[9], t[3]; pose_Rt(c, f, R, t);
            float s[3] = {0.0f, 0.0f, -1.0f}, Rs[3]; mul3(R, s, Rs);
            SH(Rs, sh_order(c), c->light + (size_t)f * MAXB);
            c->light[(size_t)f * MAXB] = 0.02f;
        }
    }
}

/* ------------------------------------------------------------------ upsample */

/* Optimizer::subsampling OptimizerAux.cpp:622-684 + VolumetricGradSdf::subsample
 * VolumetricGradSdf.cpp:469-494 + grid_subsample VoxelGrid.h:143-149 */
static void upsample2x(orc_ctx* c) {
    int nd[3] = {2 * c->dim[0], 2 * c->dim[1], 2 * c->dim[2]};
    size_t nn = 8 * c->nvox;
    float* dist = (float*)malloc(sizeof(float) * nn); float* gx = (float*)calloc(nn, sizeof(float)); float* gy = (float*)calloc(nn, sizeof(float)); float* gz = (float*)calloc(nn, sizeof(float));
    float* w = (float*)calloc(nn, sizeof(float)); float* r = (float*)malloc(sizeof(float) * nn); float* g = (float*)malloc(sizeof(float) * nn); float* b = (float*)malloc(sizeof(float) * nn);
    uint64_t* vis = (uint64_t*)calloc(nn * c->wpv, sizeof(uint64_t));
    for (size_t i = 0; i < nn; ++i) { dist[i] = c->T; r[i] = g[i] = b[i] = 0.5f; }
    float vs4 = (float)(0.25 * (double)c->vs);
    for (int k = 0; k < c->dim[2]; ++k) for (int j = 0; j < c->dim[1]; ++j) for (int i = 0; i < c->dim[0]; ++i) {
        size_t lin = (size_t)i + (size_t)j * c->dim[0] + (size_t)k * c->dim[0] * c->dim[1];
        if (c->dist[lin] == c->T) continue;
        float gr[3] = {c->gx[lin], c->gy[lin], c->gz[lin]}, gn[3]; normalized3(gr, gn);
        for (int sub = 0; sub < 8; ++sub) {
            int sx = sub & 1, sy = (sub >> 1) & 1, sz = (sub >> 2) & 1;
            float ax = sx ? gn[0] : -gn[0], ay = sy ? gn[1] : -gn[1], az = sz ? gn[2] : -gn[2];
            float d = c->dist[lin] + vs4 * (ax + ay + az);
            size_t ls = (size_t)(2 * i + sx) + (size_t)(2 * j + sy) * nd[0] + (size_t)(2 * k + sz) * nd[0] * nd[1];
            dist[ls] = d; gx[ls] = gr[0]; gy[ls] = gr[1]; gz[ls] = gr[2]; w[ls] = c->weight[lin]; r[ls] = c->r[lin]; g[ls] = c->g[lin]; b[ls] = c->b[lin];
            for (int q = 0; q < c->wpv; ++q) vis[ls * c->wpv + q] = c->vis[lin * c->wpv + q];
        }
    }
    free(c->dist); free(c->gx); free(c->gy); free(c->gz); free(c->weight); free(c->r); free(c->g); free(c->b); free(c->vis);
    c->dist = dist; c->gx = gx; c->gy = gy; c->gz = gz; c->weight = w; c->r = r; c->g = g; c->b = b; c->vis = vis;
    c->vs *= 0.5f; /* grid_subsample and Optimizer::voxel_size_ both halve */
    for (int a = 0; a < 3; ++a) c->dim[a] = nd[a];
    for (int a = 0; a < 3; ++a) c->origin[a] = c->shift[a] - (float)(0.5 * (double)c->vs) * (float)c->dim[a] - (float)(0.5 * (double)c->vs) * 1.0f;
    c->nvox = nn;
    c->vs_inv = (float)(1.0 / (double)c->vs);
    build_band(c);
}


/* ------------------------------------------------------------------ multi-rank (slab) phases
 * Mirror of psgsdf_mg_* (include/psgsdf.h): same buffers, same layouts, same phase ids, so the host program
 * psgradientsdf_amd/distributed.py can be exercised on CPU (gloo, world_size 2) against this oracle. */
#define FROW 64
static void mg_setup(orc_ctx* c) {
    int S = c->S;
    int C = (S + c->n_ranks - 1) / c->n_ranks;
    c->row0 = c->rank * C < S ? c->rank * C : S; c->row1 = c->row0 + C < S ? c->row0 + C : S;
    c->Spad = ((S + 255) / 256) * 256 + 256;
    /* rows the stencils of the owned rows reach outside [row0,row1): ELL columns = self, 6 axis, 12 pair offsets */
    c->halo = 0; c->need[0] = c->need[1] = 0;
    if (c->n_ranks > 1) {
        long st[3] = {1, c->dim[0], (long)c->dim[0] * c->dim[1]};
        for (int i = c->row0; i < c->row1; ++i) for (int ox = -1; ox <= 1; ++ox) for (int oy = -1; oy <= 1; ++oy) for (int oz = -1; oz <= 1; ++oz) {
            if ((ox != 0) + (oy != 0) + (oz != 0) > 2) continue;
            long ln = (long)c->band[i] + ox * st[0] + oy * st[1] + oz * st[2];
            if (ln < 0 || (size_t)ln >= c->nvox) continue;
            int r = band_find(c, (size_t)ln); if (r < 0) continue;
            if (r < c->row0 && c->row0 - r > c->need[0]) c->need[0] = c->row0 - r;
            if (r >= c->row1 && r - c->row1 + 1 > c->need[1]) c->need[1] = r - c->row1 + 1;
        }
        c->halo = c->need[0] > c->need[1] ? c->need[0] : c->need[1];
    }
    free(c->mg_frame); free(c->mg_dist); free(c->mg_blk); free(c->mg_rec[0]); free(c->mg_rec[1]); free(c->mg_rho); free(c->mg_grad); free(c->cg_hist);
    free(c->cg_x); free(c->cg_r); free(c->cg_t); free(c->cg_p); free(c->cg_inv); free(c->cg_sc);
    c->mg_frame = (double*)calloc((size_t)(c->F > 0 ? c->F : 1) * FROW, sizeof(double));
    c->mg_dist = (float*)calloc(c->Spad, sizeof(float)); c->mg_blk = (float*)calloc((size_t)14 * c->Spad, sizeof(float));
    c->mg_rec[0] = (float*)calloc((size_t)4 * c->Spad, sizeof(float)); c->mg_rec[1] = (float*)calloc((size_t)4 * c->Spad, sizeof(float));
    c->cg_hist = (double*)calloc(4096 + 2, sizeof(double)); c->mg_fold_base = 0;
    c->mg_rho = (float*)calloc((size_t)3 * c->Spad, sizeof(float)); c->mg_grad = (float*)calloc((size_t)3 * c->Spad, sizeof(float));
    c->cg_x = (float*)calloc(c->Spad, sizeof(float)); c->cg_r = (float*)calloc(c->Spad, sizeof(float)); c->cg_t = (float*)calloc(c->Spad, sizeof(float));
    c->cg_p = (float*)calloc(c->Spad, sizeof(float)); c->cg_inv = (float*)calloc(c->Spad, sizeof(float));
    c->cg_sc = (double*)calloc(4 + 3 * 4096, sizeof(double));
}
static int sym4(int a, int b) { if (a > b) { int t = a; a = b; b = t; } return a * 4 - (a * (a - 1)) / 2 + (b - a); }

/* Phase API of the oracle's multi-rank mirror (driven by psgradientsdf_amd/distributed.py in the gloo CPU tests).  The engine runs its
 * slab loop natively (loop.hip + comm.hip) and exports no phases; the ids below are private to the oracle and its host program. */
enum { PSGSDF_MG_BUF_FRAME_ACC = 0, PSGSDF_MG_BUF_SCAL = 1, PSGSDF_MG_BUF_PCG = 2, PSGSDF_MG_BUF_DIST = 3,
       PSGSDF_MG_BUF_BLK = 4, PSGSDF_MG_BUF_REC0 = 5, PSGSDF_MG_BUF_RHO = 6, PSGSDF_MG_BUF_GRAD = 7, PSGSDF_MG_BUF_REC1 = 8 };
enum { PSGSDF_MG_ENERGY = 0, PSGSDF_MG_INIT_ALBEDO = 1, PSGSDF_MG_LED_SUMS = 2, PSGSDF_MG_LED_SET = 3,
       PSGSDF_MG_SWEEP_ALBEDO = 4, PSGSDF_MG_APPLY_ALBEDO = 5, PSGSDF_MG_SWEEP_LIGHT = 6, PSGSDF_MG_SOLVE_LIGHT = 7,
       PSGSDF_MG_SWEEP_POSE = 8, PSGSDF_MG_SOLVE_POSE = 9, PSGSDF_MG_SWEEP_DIST = 10, PSGSDF_MG_ASSEMBLE = 11,
       PSGSDF_MG_PCG_INIT = 12, PSGSDF_MG_PCG_PASS = 13, PSGSDF_MG_APPLY_DIST = 14, PSGSDF_MG_DERIVE = 15 };
int orc_comm_init(orc_ctx* c, const uint8_t* id, int rank, int n_ranks) { (void)id; if (!c || rank < 0 || rank >= n_ranks) return PSGSDF_ERR_ARG; c->rank = rank; c->n_ranks = n_ranks; c->inited = 0; return 0; }
int orc_set_stream(orc_ctx* c, void* s) { (void)c; (void)s; return 0; }
int orc_mg_info(orc_ctx* c, int32_t out[10]) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    out[0] = c->S; out[1] = c->Spad; out[2] = c->row0; out[3] = c->row1; out[4] = c->halo; out[5] = c->F; out[6] = c->rank; out[7] = c->n_ranks; out[8] = c->need[0]; out[9] = c->need[1];
    return 0;
}
int orc_mg_buffer(orc_ctx* c, int which, void** ptr, int64_t* count) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    int64_t Sp = c->Spad;
    switch (which) {
        case PSGSDF_MG_BUF_FRAME_ACC: *ptr = c->mg_frame; *count = (int64_t)c->F * FROW; break;
        case PSGSDF_MG_BUF_SCAL: *ptr = c->mg_scal; *count = 64; break;
        case PSGSDF_MG_BUF_PCG: *ptr = c->mg_ext; *count = 8; break;
        case PSGSDF_MG_BUF_DIST: *ptr = c->mg_dist; *count = Sp; break;
        case PSGSDF_MG_BUF_BLK: *ptr = c->mg_blk; *count = 14 * Sp; break;
        case PSGSDF_MG_BUF_REC0: *ptr = c->mg_rec[0]; *count = 4 * Sp; break;
        case PSGSDF_MG_BUF_REC1: *ptr = c->mg_rec[1]; *count = 4 * Sp; break;
        case PSGSDF_MG_BUF_RHO: *ptr = c->mg_rho; *count = 3 * Sp; break;
        case PSGSDF_MG_BUF_GRAD: *ptr = c->mg_grad; *count = 3 * Sp; break;
        default: return PSGSDF_ERR_ARG;
    }
    return 0;
}
/* dense <-> band-compact exchange planes */
static void pack_state(orc_ctx* c) {
    int Sp = c->Spad;
    for (int j = c->row0; j < c->row1; ++j) { int lin = c->band[j];
        c->mg_dist[j] = c->dist[lin];
        c->mg_rho[j] = c->r[lin]; c->mg_rho[Sp + j] = c->g[lin]; c->mg_rho[2 * Sp + j] = c->b[lin];
        c->mg_grad[j] = c->gx[lin]; c->mg_grad[Sp + j] = c->gy[lin]; c->mg_grad[2 * Sp + j] = c->gz[lin]; }
}
static void unpack_state(orc_ctx* c, int lo, int hi, int with_rho_grad) {
    int Sp = c->Spad;
    for (int j = lo; j < hi; ++j) { if (j >= c->row0 && j < c->row1) continue; int lin = c->band[j];
        c->dist[lin] = c->mg_dist[j];
        if (with_rho_grad) { c->r[lin] = c->mg_rho[j]; c->g[lin] = c->mg_rho[Sp + j]; c->b[lin] = c->mg_rho[2 * Sp + j];
                             c->gx[lin] = c->mg_grad[j]; c->gy[lin] = c->mg_grad[Sp + j]; c->gz[lin] = c->mg_grad[2 * Sp + j]; } }
}
static float pcg_thr(float rhsN) { return fmaxf(FLT_EPSILON * FLT_EPSILON * rhsN, FLT_MIN); }

int orc_mg_phase(orc_ctx* c, int phase, int arg) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    int S = c->S, Sp = c->Spad, F = c->F;
    int led = c->set.model == PSGSDF_LED;
    double* fold = c->mg_scal + c->mg_fold_base;   /* where this phase's scalars land (orc_mg_fold_base) */
    switch (phase) {
        case PSGSDF_MG_ENERGY: { long long n; double e = ps_energy(c, &n); fold[0] = e * S; fold[1] = (double)n; return 0; }
        case PSGSDF_MG_INIT_ALBEDO: init_albedo(c); return 0;
        case PSGSDF_MG_LED_SUMS: { float keep[3] = {c->light[0], c->light[1], c->light[2]}; init_light(c); (void)keep; return 0; }   /* leaves sums in mg_scal[0..5] */
        case PSGSDF_MG_LED_SET: for (int ch = 0; ch < 3; ++ch) c->light[ch] = (float)c->mg_scal[ch] / (float)c->mg_scal[3 + ch]; return 0;
        case PSGSDF_MG_SWEEP_ALBEDO: {
            float* H = (float*)malloc(sizeof(float) * 3 * (S + 1)); float* b = (float*)malloc(sizeof(float) * 3 * (S + 1));
            double e; long long n; albedo_system(c, H, b, &e, &n);
            for (int j = c->row0; j < c->row1; ++j) for (int ch = 0; ch < 3; ++ch) { c->mg_blk[(size_t)ch * Sp + j] = H[3 * j + ch]; c->mg_blk[(size_t)(3 + ch) * Sp + j] = b[3 * j + ch]; }
            fold[0] = e * S; fold[1] = (double)n; free(H); free(b); return 0; }
        case PSGSDF_MG_APPLY_ALBEDO: {
            long long count = 0; float damping = c->set.damping;
            for (int j = c->row0; j < c->row1; ++j) { int lin = c->band[j]; float* rho[3] = {&c->r[lin], &c->g[lin], &c->b[lin]};
                for (int ch = 0; ch < 3; ++ch) { float h = c->mg_blk[(size_t)ch * Sp + j]; if (damping != 0.0f) h += damping * h;
                    float delta = (h != 0.f) ? c->mg_blk[(size_t)(3 + ch) * Sp + j] / h : 0.f; float v = *rho[ch] - delta;
                    if (v > 0.0f && v < 1.0f) { *rho[ch] = v; count++; } } }
            fold[0] = (double)count; return 0; }
        case PSGSDF_MG_SWEEP_LIGHT: {
            int nb = led ? 1 : F, n = c->basis, nh = led ? 3 : n * (n + 1) / 2;
            double* H = (double*)malloc(sizeof(double) * nb * n * n); double* b = (double*)malloc(sizeof(double) * nb * n); double e; long long no;
            light_system(c, H, b, &e, &no);
            memset(c->mg_frame, 0, sizeof(double) * (size_t)F * FROW);
            for (int k = 0; k < nb; ++k) { double* row = c->mg_frame + (size_t)k * FROW; int q = 0;
                if (led) { for (int i = 0; i < 3; ++i) { row[i] = H[i * 3 + i]; row[3 + i] = b[i]; } }
                else { for (int i = 0; i < n; ++i) for (int kk = i; kk < n; ++kk) row[q++] = H[(size_t)k * n * n + i * n + kk]; for (int i = 0; i < n; ++i) row[nh + i] = b[(size_t)k * n + i]; } }
            c->mg_frame[nh + n] = e * S; c->mg_frame[nh + n + 1] = (double)no;   /* energy / n_obs ride in row 0 */
            free(H); free(b); return 0; }
        case PSGSDF_MG_SOLVE_LIGHT: {
            int nb = led ? 1 : F, n = c->basis, nh = led ? 3 : n * (n + 1) / 2;
            double* H = (double*)calloc((size_t)nb * n * n, sizeof(double)); double* b = (double*)calloc((size_t)nb * n, sizeof(double));
            if (led) { for (int f = 0; f < F; ++f) for (int i = 0; i < 3; ++i) { H[i * 3 + i] += c->mg_frame[(size_t)f * FROW + i]; b[i] += c->mg_frame[(size_t)f * FROW + 3 + i]; } }
            else for (int k = 0; k < nb; ++k) { const double* row = c->mg_frame + (size_t)k * FROW; int q = 0;
                for (int i = 0; i < n; ++i) for (int kk = i; kk < n; ++kk) { H[(size_t)k * n * n + i * n + kk] = row[q]; H[(size_t)k * n * n + kk * n + i] = row[q]; ++q; }
                for (int i = 0; i < n; ++i) b[(size_t)k * n + i] = row[nh + i]; }
            return light_finish(c, H, b, 0, 0, NULL); }
        case PSGSDF_MG_SWEEP_POSE: {
            double* H = (double*)malloc(sizeof(double) * F * 36); double* b = (double*)malloc(sizeof(double) * F * 6); double e; long long no;
            pose_system(c, H, b, &e, &no);
            memset(c->mg_frame, 0, sizeof(double) * (size_t)F * FROW);
            for (int k = 0; k < F; ++k) { double* row = c->mg_frame + (size_t)k * FROW; int q = 0;
                for (int i = 0; i < 6; ++i) for (int kk = i; kk < 6; ++kk) row[q++] = H[(size_t)k * 36 + i * 6 + kk]; for (int i = 0; i < 6; ++i) row[21 + i] = b[(size_t)k * 6 + i]; }
            c->mg_frame[27] = e * S; c->mg_frame[28] = (double)no; free(H); free(b); return 0; }
        case PSGSDF_MG_SOLVE_POSE: {
            double* H = (double*)calloc((size_t)F * 36, sizeof(double)); double* b = (double*)calloc((size_t)F * 6, sizeof(double));
            for (int k = 0; k < F; ++k) { const double* row = c->mg_frame + (size_t)k * FROW; int q = 0;
                for (int i = 0; i < 6; ++i) for (int kk = i; kk < 6; ++kk) { H[(size_t)k * 36 + i * 6 + kk] = row[q]; H[(size_t)k * 36 + kk * 6 + i] = row[q]; ++q; }
                for (int i = 0; i < 6; ++i) b[(size_t)k * 6 + i] = row[21 + i]; }
            return pose_finish(c, H, b, 0, 0, NULL); }
        case PSGSDF_MG_SWEEP_DIST: {
            dist_sys* sy = (dist_sys*)c->mg_sys; if (sy) { dist_sys_free(sy); free(sy); }
            sy = (dist_sys*)calloc(1, sizeof(dist_sys)); c->mg_sys = sy;
            double e; long long no; dist_system(c, c->reg_n != 0.f, arg, sy, &e, &no);
            for (int j = c->row0; j < c->row1; ++j) { int q = 0;
                for (int a = 0; a < 4; ++a) for (int bq = a; bq < 4; ++bq) c->mg_blk[(size_t)(q++) * Sp + j] = (float)sy->B[16 * j + a * 4 + bq];
                for (int a = 0; a < 4; ++a) c->mg_blk[(size_t)(10 + a) * Sp + j] = (float)sy->g[4 * j + a]; }
            fold[0] = e * S; fold[1] = (double)no; return 0; }
        case PSGSDF_MG_ASSEMBLE: {
            dist_sys* sy = (dist_sys*)c->mg_sys; if (!sy) return PSGSDF_ERR_STATE;
            if (c->n_ranks > 1) {
                int jlo = c->row0 - c->need[0] < 0 ? 0 : c->row0 - c->need[0], jhi = c->row1 + c->need[1] > S ? S : c->row1 + c->need[1];
                for (int j = jlo; j < jhi; ++j) { if (j >= c->row0 && j < c->row1) continue;
                    for (int a = 0; a < 4; ++a) { for (int bq = 0; bq < 4; ++bq) sy->B[16 * j + a * 4 + bq] = (double)c->mg_blk[(size_t)sym4(a, bq) * Sp + j]; sy->g[4 * j + a] = (double)c->mg_blk[(size_t)(10 + a) * Sp + j]; } }
                dist_assemble(c, sy);
            }
            return 0; }
        /* Fused Jacobi-PCG in the formulation of the engine's k_cgf_pass (pcg.hip): kernel k finishes pass k-1 for the
         * owned rows (x += alpha p), re-derives r_k, z_k, p_k of every column from that column's record {r, t, p, inv} of
         * pass k-1 (halo records exchanged by the host program), runs t = A p_k and leaves the 7 local sums in mg_ext;
         * alpha, beta and the convergence test come from the all-reduced sums of pass k-1 found in mg_ext on entry.
         * Same recurrences as eigen_cg() above (Eigen ConjugateGradient.h), scalars in float, sums in double. */
        case PSGSDF_MG_PCG_INIT: {
            dist_sys* sy = (dist_sys*)c->mg_sys; if (!sy || !sy->rowptr) return PSGSDF_ERR_STATE;
            double bb = 0; float* rec = c->mg_rec[1];
            for (int i = c->row0; i < c->row1; ++i) { float dg = sy->diag[i]; if (c->set.damping != 0.f) dg += c->set.damping * dg;
                float inv = dg != 0.f ? 1.0f / dg : 1.0f; float r = sy->rhs[i];
                c->cg_x[i] = 0.f; rec[4 * i] = r; rec[4 * i + 1] = 0.f; rec[4 * i + 2] = 0.f; rec[4 * i + 3] = inv;
                bb += (double)r * r; }
            c->cg_stop = 0; c->cg_bb = 0; memset(c->mg_ext, 0, sizeof(c->mg_ext)); c->mg_ext[0] = bb; return 0; }
        case PSGSDF_MG_PCG_PASS: {
            dist_sys* sy = (dist_sys*)c->mg_sys; int k = arg; if (!sy || !sy->rowptr || k < 0 || k > 4096) return PSGSDF_ERR_ARG;
            int cap = c->set.cg_max_it > 0 ? c->set.cg_max_it : 2 * S; if (cap > 4096) cap = 4096;
            if (c->cg_stop && c->cg_stop <= k) return 0;
            float alpha_prev = 0.f, beta = 0.f, rr_cur, rhsN;
            if (k == 0) { c->cg_bb = c->mg_ext[0]; rhsN = (float)c->cg_bb; rr_cur = rhsN; c->cg_hist[0] = c->cg_bb; }
            else { const double* t = c->mg_ext; rhsN = (float)c->cg_bb;
                float rz_old = (float)t[5]; alpha_prev = rz_old / (float)t[0]; double al = (double)alpha_prev;
                float rz_cur = (float)(t[5] - 2.0 * al * t[1] + al * al * t[2]);
                rr_cur = (float)(t[6] - 2.0 * al * t[3] + al * al * t[4]);
                beta = rz_cur / rz_old; c->cg_hist[k] = (double)rr_cur; }
            int stop = rhsN == 0.f || k == cap || (k > 0 && rr_cur < pcg_thr(rhsN));
            if (stop) c->cg_stop = k + 1;
            const float* rin = c->mg_rec[(k + 1) & 1]; float* rout = c->mg_rec[k & 1];
            double sm[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int i = c->row0; i < c->row1; ++i) {
                const float* me = rin + 4 * i;
                if (k > 0) c->cg_x[i] += alpha_prev * me[2];
                if (stop) continue;
                float r_i = me[0] - alpha_prev * me[1], z_i = me[3] * r_i, p_i = z_i + beta * me[2];
                double acc = 0;
                for (int q = sy->rowptr[i]; q < sy->rowptr[i + 1]; ++q) { int cc = sy->colidx[q]; float v = sy->val[q];
                    if (cc == i) { if (c->set.damping != 0.f) v += c->set.damping * v; acc += (double)v * (double)p_i; }
                    else { const float* o = rin + 4 * cc; float rc = o[0] - alpha_prev * o[1]; acc += (double)v * (double)(o[3] * rc + beta * o[2]); } }
                float t = (float)acc;
                rout[4 * i] = r_i; rout[4 * i + 1] = t; rout[4 * i + 2] = p_i; rout[4 * i + 3] = me[3];
                double rd = r_i, td = t, iv = me[3];
                sm[0] += (double)p_i * td; sm[1] += iv * rd * td; sm[2] += iv * td * td; sm[3] += rd * td; sm[4] += td * td; sm[5] += rd * (double)z_i; sm[6] += rd * rd;
            }
            if (!stop) for (int q = 0; q < 7; ++q) c->mg_ext[q] = sm[q];
            return 0; }
        case PSGSDF_MG_APPLY_DIST: {
            long long count = 0;
            for (int j = c->row0; j < c->row1; ++j) { float d = c->cg_x[j]; if ((double)fabsf(d) < sqrt(3.0) * (double)c->vs) { c->dist[c->band[j]] -= d; count++; } }
            pack_state(c); fold[0] = (double)count; return 0; }
        case PSGSDF_MG_DERIVE: {
            int lo = c->row0 - c->need[0] < 0 ? 0 : c->row0 - c->need[0], hi = c->row1 + c->need[1] > S ? S : c->row1 + c->need[1];
            if (c->n_ranks > 1) unpack_state(c, lo, hi, 0);
            if (arg) update_grad(c);
            fold[0] = normal_energy(c) * S; fold[1] = laplacian_energy(c) * S; return 0; }
        default: return PSGSDF_ERR_ARG;
    }
}
int orc_mg_pcg_status(orc_ctx* c, int k0, int n, int32_t* iters, double* err) {
    if (!c || !c->inited || !iters || !err || k0 < 0 || n < 1 || n > 64) return PSGSDF_ERR_ARG;
    int cap = c->set.cg_max_it > 0 ? c->set.cg_max_it : 2 * c->S; if (cap > 4096) cap = 4096;
    float rhsN = (float)c->cg_hist[0];
    *iters = -1; *err = 0;
    if (rhsN == 0.f) { *iters = 0; return 0; }
    float thr = pcg_thr(rhsN), rn2 = rhsN;
    for (int q = 0; q < n && *iters < 0; ++q) { int kk = k0 + q; if (kk == 0) continue;
        rn2 = (float)c->cg_hist[kk]; if (rn2 < thr) *iters = kk - 1; else if (kk == cap) *iters = cap; }
    *err = sqrt((double)rn2 / (double)rhsN);
    return 0;
}
int orc_mg_fold_base(orc_ctx* c, int base) { if (!c || base < 0 || base >= 64) return PSGSDF_ERR_ARG; c->mg_fold_base = base; return 0; }
int orc_mg_set_reg_sums(orc_ctx* c, double en_sum, double el_sum) { (void)c; (void)en_sum; (void)el_sum; return 0; }   /* the oracle recomputes regulariser energies on demand */
int orc_mg_set_weights(orc_ctx* c, float reg_n, float reg_l) { c->reg_n = reg_n; c->reg_l = reg_l; return 0; }
/* final state gather support: owned rows -> exchange planes, all rows <- exchange planes */
int orc_mg_pack_state(orc_ctx* c) { pack_state(c); return 0; }
int orc_mg_unpack_state(orc_ctx* c) { unpack_state(c, 0, c->S, 1); return 0; }

/* ------------------------------------------------------------------ exported API (mirrors psgsdf.h) */

#define ORC_FAIL(c, code, msg) do { if (c) snprintf((c)->err, sizeof((c)->err), "%s", msg); return code; } while (0)

int orc_create(const psgsdf_grid_desc* grid, const float K[9], const psgsdf_settings* settings, int device, orc_ctx** out) {
    (void)device;
    if (!grid || !K || !settings || !out) return PSGSDF_ERR_ARG;
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    for (int a = 0; a < 3; ++a) { c->dim[a] = grid->dim[a]; c->shift[a] = grid->shift[a]; }
    c->nvox = (size_t)c->dim[0] * c->dim[1] * c->dim[2];
    c->vs = grid->voxel_size; c->vs_inv = 1.f / c->vs; c->T = grid->truncation;
    /* VoxelGrid.h:130: origin_ = shift_ - 0.5*voxel_size*grid_dim_.cast<float>() */
    for (int a = 0; a < 3; ++a) c->origin[a] = c->shift[a] - (float)(0.5 * (double)c->vs) * (float)c->dim[a];
    c->fx = K[0]; c->fy = K[4]; c->cx = K[2]; c->cy = K[5];
    c->set = *settings; c->reg_n = settings->reg_weight_n; c->reg_l = settings->reg_weight_l; c->reg_r = settings->reg_weight_rho;
    c->threads = 1; c->rank = 0; c->n_ranks = 1;
    *out = c;
    return 0;
}
void orc_destroy(orc_ctx* c) {
    if (!c) return;
    free(c->dist); free(c->gx); free(c->gy); free(c->gz); free(c->weight); free(c->r); free(c->g); free(c->b);
    free(c->vis_seq); free(c->vis); free(c->frame_idx); free(c->img); free(c->poses); free(c->light); free(c->band); free(c->row_of);
    free(c);
}
const char* orc_last_error(const orc_ctx* c) { return c ? c->err : "null context"; }
const char* orc_version(void) { return "psgsdf-oracle cpu (test infrastructure)"; }
int orc_set_solver_mode(orc_ctx* c, int mode) { c->solver_mode = mode; return 0; }
/* the engine's name for the same switch (include/psgsdf.h psgsdf_set_frame_solver): 1 = the reference's global Eigen CG over all frames' blocks */
int orc_set_frame_solver(orc_ctx* c, int mode) { if (mode != 0 && mode != 1) return PSGSDF_ERR_ARG; c->solver_mode = mode; return 0; }
int orc_get_frame_solver_stats(orc_ctx* c, int block, int32_t* iterations, double* error, int32_t* converged, int32_t* applied) {
    if (!c || (block != PSGSDF_LIGHT && block != PSGSDF_POSE)) return PSGSDF_ERR_ARG;
    const int k = block == PSGSDF_POSE;
    if (iterations) *iterations = c->fs_iters[k]; if (error) *error = c->fs_err[k]; if (converged) *converged = c->fs_ok[k]; if (applied) *applied = c->fs_applied[k];
    return 0;
}
/* eigen_cg on a block-diagonal system of nb blocks of n x n floats (the engine's psgsdf_debug_frame_cg) */
int orc_debug_frame_cg(orc_ctx* c, int nb, int n, const float* H, const float* b, float* x, int max_it, int* iters, double* err, int* ok) {
    (void)c;
    blockdiag B = {nb, n, H};
    float* diag = (float*)malloc(sizeof(float) * nb * n);
    for (int k = 0; k < nb; ++k) for (int i = 0; i < n; ++i) diag[k * n + i] = H[(size_t)k * n * n + i * n + i];
    cg_result r = eigen_cg(nb * n, blockdiag_mv, &B, diag, b, x, max_it);
    free(diag);
    if (iters) *iters = r.iters; if (err) *err = r.error; if (ok) *ok = r.success;
    return 0;
}
/* 1: every band-membership test is the reference's std::find over the band list (SURVEY 8d(i) "faithful mode"; same results, O(S) per test) */
int orc_set_faithful(orc_ctx* c, int on) { c->faithful = on; return 0; }
int orc_set_threads(orc_ctx* c, int n) { c->threads = n > 0 ? n : 1; return 0; }

static float* dupf(const float* p, size_t n) { float* q = (float*)malloc(sizeof(float) * n); memcpy(q, p, sizeof(float) * n); return q; }

int orc_upload_volume(orc_ctx* c, const float* dist, const float* grad_xyz, const float* weight, const float* rgb, const uint64_t* vis_words, int words_per_voxel) {
    if (!c || !dist || !grad_xyz || !weight || !rgb || !vis_words || words_per_voxel < 1) return PSGSDF_ERR_ARG;
    size_t n = c->nvox;
    free(c->dist); free(c->gx); free(c->gy); free(c->gz); free(c->weight); free(c->r); free(c->g); free(c->b); free(c->vis_seq);
    c->dist = dupf(dist, n); c->gx = dupf(grad_xyz, n); c->gy = dupf(grad_xyz + n, n); c->gz = dupf(grad_xyz + 2 * n, n);
    c->weight = dupf(weight, n); c->r = dupf(rgb, n); c->g = dupf(rgb + n, n); c->b = dupf(rgb + 2 * n, n);
    c->vis_seq = (uint64_t*)malloc(sizeof(uint64_t) * n * words_per_voxel); memcpy(c->vis_seq, vis_words, sizeof(uint64_t) * n * words_per_voxel);
    c->wpv_seq = words_per_voxel;
    c->inited = 0;
    return 0;
}
/* VolumetricGradSdf::init, VolumetricGradSdf.cpp:14-38 */
int orc_volume_init(orc_ctx* c, int max_frames) {
    if (!c || max_frames < 1) return PSGSDF_ERR_ARG;
    size_t n = c->nvox;
    free(c->dist); free(c->gx); free(c->gy); free(c->gz); free(c->weight); free(c->r); free(c->g); free(c->b); free(c->vis_seq);
    c->dist = (float*)malloc(sizeof(float) * n); for (size_t i = 0; i < n; ++i) c->dist[i] = c->T;
    c->gx = (float*)calloc(n, sizeof(float)); c->gy = (float*)calloc(n, sizeof(float)); c->gz = (float*)calloc(n, sizeof(float));
    c->weight = (float*)calloc(n, sizeof(float)); c->r = (float*)calloc(n, sizeof(float)); c->g = (float*)calloc(n, sizeof(float)); c->b = (float*)calloc(n, sizeof(float));
    c->wpv_seq = (max_frames + 63) / 64; c->vis_seq = (uint64_t*)calloc(n * c->wpv_seq, sizeof(uint64_t));
    c->inited = 0;
    return 0;
}
/* VolumetricGradSdf::update, VolumetricGradSdf.cpp:51-138 (+ truncate / weight, Sdf.h:44-66) */
int orc_integrate_frame(orc_ctx* c, const float* rgb, const float* depth, const float* normals_xyz, int W, int H, const float pose[16], int counter, float z_min, float z_max) {
    if (!c || !c->dist || !c->vis_seq || !rgb || !depth || !normals_xyz || counter < 0 || counter >= 64 * c->wpv_seq) return PSGSDF_ERR_ARG;
    const float fx = c->fx, fy = c->fy, cx = c->cx, cy = c->cy;
    float R[9], t[3];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = pose[i * 4 + j]; t[i] = pose[i * 4 + 3]; }
    const float T = c->T, inv_T = (float)(1.0 / (double)c->T);
    const float* nxp = normals_xyz; const float* nyp = normals_xyz + (size_t)W * H; const float* nzp = normals_xyz + 2 * (size_t)W * H;
    const double fx_inv = 1.0 / (double)fx, fy_inv = 1.0 / (double)fy;
    /* voxels are independent: the z-planes may be spread over host threads without changing a bit */
#pragma omp parallel for schedule(static) num_threads(c->threads)
    for (int k = 0; k < c->dim[2]; ++k) for (int j = 0; j < c->dim[1]; ++j) for (int i = 0; i < c->dim[0]; ++i) {
        size_t lin = (size_t)i + (size_t)c->dim[0] * j + (size_t)c->dim[0] * c->dim[1] * k;
        int idx[3] = {i, j, k}; float xv[3]; voxel2world(c, idx, xv);
        float tmp[3] = {xv[0] - t[0], xv[1] - t[1], xv[2] - t[2]}, point[3];
        mulT3(R, tmp, point);
        if (point[2] < 0.f) continue;
        const int n = (int)((double)(cx + fx * point[0] / point[2]) + 0.5);   /* +0.5 is a double literal in the reference */
        const int m = (int)((double)(cy + fy * point[1] / point[2]) + 0.5);
        if (n < 0 || n >= W || m < 0 || m >= H) continue;
        const float z = depth[(size_t)m * W + n];
        if (z <= z_min || z >= z_max) continue;
        const float sdf = z - point[2];
        float w = 0.f;
        if (sdf >= 0.) w = 1.f; else if (sdf >= -T) w = 1.f + sdf * inv_T;
        if (w == 0) continue;
        float normal[3] = {nxp[(size_t)m * W + n], nyp[(size_t)m * W + n], nzp[(size_t)m * W + n]};
        if (dot3(normal, normal) < .1) continue;
        float zi = (float)(1. / (double)point[2]);
        float xy_hom[3] = {zi * point[0], zi * point[1], zi * point[2]};
        /* n_sq_inv = 1/(1+x0^2+y0^2) evaluated in double and stored as float (NormalEstimator.h:66-77,100) */
        double x0 = fx_inv * ((double)n - (double)cx), y0 = fy_inv * ((double)m - (double)cy);
        float n_sq_inv = (float)(1.0 / (1.0 + x0 * x0 + y0 * y0));
        float dn = dot3(normal, xy_hom);
        if (dn * dn * n_sq_inv < .25 * .25) continue;
        c->weight[lin] += w;
        float ts = fmaxf(-T, fminf(T, sdf));
        c->dist[lin] += (ts - c->dist[lin]) * w / c->weight[lin];
        float Rn[3]; mul3(R, normal, Rn);
        c->gx[lin] -= w * Rn[0]; c->gy[lin] -= w * Rn[1]; c->gz[lin] -= w * Rn[2];
        const float* col = rgb + ((size_t)m * W + n) * 3;
        c->r[lin] += (col[0] - c->r[lin]) * w / c->weight[lin];
        c->g[lin] += (col[1] - c->g[lin]) * w / c->weight[lin];
        c->b[lin] += (col[2] - c->b[lin]) * w / c->weight[lin];
        c->vis_seq[lin * c->wpv_seq + (counter >> 6)] |= 1ull << (counter & 63);
    }
    return 0;
}
int orc_download_vis_seq(orc_ctx* c, uint64_t* out) { memcpy(out, c->vis_seq, sizeof(uint64_t) * c->nvox * c->wpv_seq); return c->wpv_seq; }


/* ------------------------------------------------------------------ front-end "next" rows
 * FALS normal estimation, normals/NormalEstimator.h:52-176 (cache in double, per-frame part in float with OpenCV's
 * double-accumulating, un-normalised 11x11 box filter and BORDER_REFLECT_101), and the depth tracker
 * RigidPointOptimizer::optimize_sampled, sdf_tracker/RigidPointOptimizer.cpp:12-79. */
static inline int reflect101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; } return i; }
static void box_filter_d(const double* src, double* dst, int W, int H, int r) {   /* (2r+1)^2 window, un-normalised */
    double* tmp = (double*)malloc(sizeof(double) * (size_t)W * H);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) { double s = 0; for (int k = -r; k <= r; ++k) s += src[(size_t)y * W + reflect101(x + k, W)]; tmp[(size_t)y * W + x] = s; }
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) { double s = 0; for (int k = -r; k <= r; ++k) s += tmp[(size_t)reflect101(y + k, H) * W + x]; dst[(size_t)y * W + x] = s; }
    free(tmp);
}
/* cache(): fills float planes {x0_n_sq_inv, y0_n_sq_inv, n_sq_inv, Q11, Q12, Q13, Q22, Q23, Q33} (9 * W*H) */
int orc_normals_cache(const float K[9], int W, int H, int radius, float* out9) {
    size_t n = (size_t)W * H;
    double fx_inv = 1. / (double)K[0], fy_inv = 1. / (double)K[4], cx = (double)K[2], cy = (double)K[5];
    double* a[6]; double* M[6];
    for (int i = 0; i < 6; ++i) { a[i] = (double*)malloc(sizeof(double) * n); M[i] = (double*)malloc(sizeof(double) * n); }
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        size_t p = (size_t)y * W + x;
        double x0 = fx_inv * ((double)x - cx), y0 = fy_inv * ((double)y - cy);
        double nsi = 1. / (1. + x0 * x0 + y0 * y0);
        a[0][p] = x0 * x0 * nsi; a[1][p] = x0 * y0 * nsi; a[2][p] = x0 * nsi; a[3][p] = y0 * y0 * nsi; a[4][p] = y0 * nsi; a[5][p] = nsi;
        out9[p] = (float)(x0 * nsi); out9[n + p] = (float)(y0 * nsi); out9[2 * n + p] = (float)nsi;
    }
    for (int i = 0; i < 6; ++i) box_filter_d(a[i], M[i], W, H, radius);
    for (size_t p = 0; p < n; ++p) {
        double M11 = M[0][p], M12 = M[1][p], M13 = M[2][p], M22 = M[3][p], M23 = M[4][p], M33 = M[5][p];
        double det = M11 * (M22 * M33) + 2 * M12 * (M23 * M13) - (M13 * (M13 * M22) + M12 * (M12 * M33) + M23 * (M23 * M11));
        double di = 1. / det;
        out9[3 * n + p] = (float)(di * (M22 * M33 - M23 * M23)); out9[4 * n + p] = (float)(di * (M13 * M23 - M12 * M33));
        out9[5 * n + p] = (float)(di * (M12 * M23 - M13 * M22)); out9[6 * n + p] = (float)(di * (M11 * M33 - M13 * M13));
        out9[7 * n + p] = (float)(di * (M12 * M13 - M11 * M23)); out9[8 * n + p] = (float)(di * (M11 * M22 - M12 * M12));
    }
    for (int i = 0; i < 6; ++i) { free(a[i]); free(M[i]); }
    return 0;
}
/* compute(): normals (3 planes) from a depth map, NormalEstimator.h:150-176 */
int orc_estimate_normals(orc_ctx* c, const float* depth, int W, int H, float* normals_xyz) {
    (void)0;
    size_t n = (size_t)W * H;
    float K[9] = {c->fx, 0, c->cx, 0, c->fy, c->cy, 0, 0, 1};
    float* cache = (float*)malloc(sizeof(float) * 9 * n);
    orc_normals_cache(K, W, H, 5, cache);
    double* src = (double*)malloc(sizeof(double) * n); double* b[3];
    for (int q = 0; q < 3; ++q) {
        b[q] = (double*)malloc(sizeof(double) * n);
        for (size_t p = 0; p < n; ++p) { float zi = depth[p] != 0.f ? 1.0f / depth[p] : 0.f; src[p] = (double)(cache[(size_t)q * n + p] * zi); }
        box_filter_d(src, b[q], W, H, 5);
    }
    for (size_t p = 0; p < n; ++p) {
        float b1 = (float)b[0][p], b2 = (float)b[1][p], b3 = (float)b[2][p];
        const float Q11 = cache[3 * n + p], Q12 = cache[4 * n + p], Q13 = cache[5 * n + p], Q22 = cache[6 * n + p], Q23 = cache[7 * n + p], Q33 = cache[8 * n + p];
        float nx = b1 * Q11 + b2 * Q12 + b3 * Q13, ny = b1 * Q12 + b2 * Q22 + b3 * Q23, nz = b1 * Q13 + b2 * Q23 + b3 * Q33;
        float nn = sqrtf(nx * nx + ny * ny + nz * nz);
        normals_xyz[p] = nx / nn; normals_xyz[n + p] = ny / nn; normals_xyz[2 * n + p] = nz / nn;
    }
    free(cache); free(src); for (int q = 0; q < 3; ++q) free(b[q]);
    return 0;
}

/* VoxelGrid::nearest_index (VoxelGrid.cpp:57-72) */
static long nearest_index(const orc_ctx* c, const float p[3]) {
    float fi[3];
    for (int a = 0; a < 3; ++a) fi[a] = (p[a] - c->origin[a]) / c->vs;
    if (fi[0] <= 0 || fi[1] <= 0 || fi[2] <= 0 || fi[0] >= (c->dim[0] - 1) || fi[1] >= (c->dim[1] - 1) || fi[2] >= (c->dim[2] - 1)) return -1;
    int im = (int)(fi[0] + 0.5), jm = (int)(fi[1] + 0.5), km = (int)(fi[2] + 0.5);
    return (long)im + (long)jm * c->dim[0] + (long)km * c->dim[0] * c->dim[1];
}
/* SE3::exp (Sophus): R = exp(omega), t = V upsilon */
static void se3_exp(const float xi[6], float T[16]) {
    float R[9]; so3_exp(xi + 3, R);
    const float* w = xi + 3; float th2 = dot3(w, w);
    float V[9];
    float Om[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, Om2[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Om2[i * 3 + j] = (Om[i * 3] * Om[j] + Om[i * 3 + 1] * Om[3 + j]) + Om[i * 3 + 2] * Om[6 + j];
    if (th2 < 1e-10f) { for (int i = 0; i < 9; ++i) V[i] = R[i]; }
    else { float th = sqrtf(th2); float a = (1.f - cosf(th)) / th2, b = (th - sinf(th)) / (th2 * th);
        for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.f : 0.f) + a * Om[i] + b * Om2[i]; }
    float t[3]; mul3(V, xi, t);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j]; T[i * 4 + 3] = t[i]; }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}
/* one Gauss-Newton pass of the tracker: fills H (21 upper), g (6), E, count for the given pose */
static void track_accumulate(const orc_ctx* c, const float* depth, int W, int H, const float pose[16], float z_min, float z_max, double acc[29]) {
    float R[9], t[3];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = pose[i * 4 + j]; t[i] = pose[i * 4 + 3]; }
    const float fx_inv = 1.f / c->fx, fy_inv = 1.f / c->fy;
    memset(acc, 0, sizeof(double) * 29);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        const float z = depth[(size_t)y * W + x];
        if (z <= z_min || z >= z_max) continue;
        const float x0 = ((float)x - c->cx) * fx_inv, y0 = ((float)y - c->cy) * fy_inv;
        float pc[3] = {x0 * z, y0 * z, z}, p[3]; mul3(R, pc, p);
        for (int a = 0; a < 3; ++a) p[a] += t[a];
        long I = nearest_index(c, p);
        if (I < 0 || !(c->weight[I] > 0)) continue;
        /* tsdf(point, &grad): VolumetricGradSdf.h:76-88 */
        float gr[3] = {c->gx[I], c->gy[I], c->gz[I]}, gn[3]; normalized3(gr, gn);
        int idx[3]; for (int a = 0; a < 3; ++a) idx[a] = (int)((p[a] - c->origin[a]) / c->vs + 0.5f);
        float xv[3]; voxel2world(c, idx, xv);
        float dv[3] = {xv[0] - p[0], xv[1] - p[1], xv[2] - p[2]};
        float phi = c->dist[I] + dot3(gn, dv);
        float gxi[6] = {gn[0], gn[1], gn[2], p[1] * gn[2] - p[2] * gn[1], p[2] * gn[0] - p[0] * gn[2], p[0] * gn[1] - p[1] * gn[0]};
        int q = 0;
        for (int i = 0; i < 6; ++i) { for (int k = i; k < 6; ++k) acc[q++] += (double)(gxi[i] * gxi[k]); acc[21 + i] += (double)(phi * gxi[i]); }
        acc[27] += (double)(phi * phi); acc[28] += 1.0;
    }
}
/* RigidPointOptimizer::optimize_sampled with sampling 1: pose (4x4 row-major) updated in place; returns 1 on convergence */
int orc_track(orc_ctx* c, const float* depth, int W, int H, float pose[16], float z_min, float z_max, int num_iterations, float conv_threshold, float damping, int* iters_out, int* converged) {
    if (converged) *converged = 0;
    for (int k = 0; k < num_iterations; ++k) {
        double acc[29]; track_accumulate(c, depth, W, H, pose, z_min, z_max, acc);
        if (acc[28] == 0) { if (iters_out) *iters_out = k; return 0; }
        double Hd[36], gd[6], xd[6]; int q = 0;
        for (int i = 0; i < 6; ++i) { for (int j = i; j < 6; ++j) { Hd[i * 6 + j] = (double)(float)acc[q]; Hd[j * 6 + i] = (double)(float)acc[q]; ++q; } gd[i] = (double)(float)acc[21 + i]; }
        solve_spd(6, Hd, gd, xd);
        float xi[6], n2 = 0; for (int i = 0; i < 6; ++i) { xi[i] = damping * (float)xd[i]; n2 += xi[i] * xi[i]; }
        if (n2 < conv_threshold * conv_threshold) { if (iters_out) *iters_out = k; if (converged) *converged = 1; return 0; }
        float mxi[6]; for (int i = 0; i < 6; ++i) mxi[i] = -xi[i];
        float E[16], P[16]; se3_exp(mxi, E);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float s_ = 0; for (int m = 0; m < 4; ++m) s_ += E[i * 4 + m] * pose[m * 4 + j]; P[i * 4 + j] = s_; }
        memcpy(pose, P, sizeof(P));
    }
    if (iters_out) *iters_out = num_iterations;
    return 0;
}
int orc_track_system(orc_ctx* c, const float* depth, int W, int H, const float pose[16], float z_min, float z_max, double acc[29]) { track_accumulate(c, depth, W, H, pose, z_min, z_max, acc); return 0; }

int orc_set_keyframes(orc_ctx* c, int n_frames, const int32_t* frame_idx, const float* rgb_images, int width, int height, const float* poses);
/* 8-bit keyframes: the reference's own conversion (ImageLoader.h:181, convertTo(CV_32FC3, scale): (float)byte * scale), then the float path */
int orc_set_keyframes_u8(orc_ctx* c, int n_frames, const int32_t* frame_idx, const uint8_t* rgb_images, float scale, int width, int height, const float* poses) {
    if (!c || n_frames <= 0 || !rgb_images || width <= 1 || height <= 1) ORC_FAIL(c, PSGSDF_ERR_ARG, "set_keyframes_u8: bad argument");
    const size_t n = (size_t)n_frames * width * height * 3;
    float* f = (float*)malloc(sizeof(float) * n);
    if (!f) ORC_FAIL(c, PSGSDF_ERR_DEVICE, "out of memory");
    for (size_t i = 0; i < n; ++i) f[i] = (float)rgb_images[i] * scale;
    const int rc = orc_set_keyframes(c, n_frames, frame_idx, f, width, height, poses);
    free(f);
    return rc;
}
int orc_set_keyframes(orc_ctx* c, int n_frames, const int32_t* frame_idx, const float* rgb_images, int width, int height, const float* poses) {
    if (!c || n_frames < 0 || (n_frames > 0 && (!frame_idx || !rgb_images || !poses))) return PSGSDF_ERR_ARG;
    free(c->frame_idx); free(c->img); free(c->poses);
    c->F = n_frames; c->W = width; c->H = height;
    c->frame_idx = (int*)malloc(sizeof(int) * (n_frames + 1)); memcpy(c->frame_idx, frame_idx, sizeof(int) * n_frames);
    c->img = dupf(rgb_images, (size_t)n_frames * width * height * 3);
    c->poses = dupf(poses, (size_t)n_frames * 16);
    c->inited = 0;
    return 0;
}
int orc_init(orc_ctx* c) {
    if (!c || !c->dist || !c->frame_idx) ORC_FAIL(c, PSGSDF_ERR_STATE, "upload_volume and set_keyframes first");
    select_vis(c);
    build_band(c);
    init_light(c);
    c->inited = 1;
    return 0;
}
int orc_init_albedo(orc_ctx* c) { if (!c || !c->inited) return PSGSDF_ERR_STATE; init_albedo(c); return 0; }

static float total_energy(const orc_ctx* c, float E, float E_n, float E_l, float E_r) { return E + c->reg_n * E_n + c->reg_l * E_l + c->reg_r * E_r; /* OptimizerAux.cpp:261 */ }

int orc_energy(orc_ctx* c, double out[4]) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    out[0] = ps_energy(c, NULL); out[1] = normal_energy(c); out[2] = laplacian_energy(c);
    out[3] = (double)total_energy(c, (float)out[0], c->reg_n != 0.f ? (float)out[1] : 0.f, c->reg_l != 0.f ? (float)out[2] : 0.f, c->reg_r != 0.f ? (float)albedo_reg_energy(c) : 0.f);
    return 0;
}
int orc_normalize_weights(orc_ctx* c, double* e_total) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    float E = (float)ps_energy(c, NULL), E_n = 0, E_l = 0;
    if (c->reg_n != 0.f) { E_n = (float)normal_energy(c); c->reg_n *= E / E_n; }
    if (c->reg_l != 0.f) { E_l = (float)laplacian_energy(c); c->reg_l *= E / E_l; }
    if (e_total) *e_total = (double)total_energy(c, E, E_n, E_l, c->reg_r != 0.f ? (float)albedo_reg_energy(c) : 0.f);   /* reg_rho is not normalised (PsOptimizer.cpp:279) */
    return 0;
}
int orc_step_ex(orc_ctx* c, int block, int laplacian_reg, psgsdf_step_stats* st) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    switch (block) {
        case PSGSDF_ALBEDO: return step_albedo(c, st);
        case PSGSDF_LIGHT: return step_light(c, st);
        case PSGSDF_DIST: return step_dist(c, laplacian_reg, st);
        case PSGSDF_POSE: return step_pose(c, st);
        default: return PSGSDF_ERR_ARG;
    }
}
int orc_step(orc_ctx* c, int block, psgsdf_step_stats* st) { return orc_step_ex(c, block, c && c->reg_l != 0.f, st); }

/* one body of the alternation loop; order SH: albedo,light,dist,pose (PsOptimizer.cpp:304-360),
 * LED: light,albedo,dist,pose (LedOptimizer.cpp:345-403) */
static void iterate_once(orc_ctx* c, int flags, int laplacian_reg, float* E, float* E_n, float* E_l, float* E_r, psgsdf_iter_stats* rec) {
    int led = c->set.model == PSGSDF_LED;
    int order[4] = {led ? PSGSDF_LIGHT : PSGSDF_ALBEDO, led ? PSGSDF_ALBEDO : PSGSDF_LIGHT, PSGSDF_DIST, PSGSDF_POSE};
    for (int q = 0; q < 4; ++q) rec->e_after[q] = NAN;
    rec->cg_iters = 0;
    rec->e_n_in = *E_n; rec->e_l_in = *E_l;   /* in force until this iteration's distance block (PsOptimizer.cpp:342-343) */
    for (int q = 0; q < 4; ++q) {
        int blk = order[q];
        if (!(flags & blk)) continue;
        psgsdf_step_stats st; memset(&st, 0, sizeof(st));
        orc_step_ex(c, blk, laplacian_reg, &st);
        *E = (float)ps_energy(c, NULL);
        if (blk == PSGSDF_DIST) {
            rec->cg_iters = st.cg_iters;
            if (c->reg_n != 0.f) *E_n = (float)normal_energy(c);
            if (laplacian_reg) *E_l = (float)laplacian_energy(c);
        }
        if (blk == PSGSDF_ALBEDO && c->reg_r != 0.f) *E_r = (float)albedo_reg_energy(c);   /* PsOptimizer.cpp:312 */
        int slot = blk == PSGSDF_ALBEDO ? 0 : blk == PSGSDF_LIGHT ? 1 : blk == PSGSDF_DIST ? 2 : 3;
        rec->e_after[slot] = (double)*E;
    }
    rec->e_n = *E_n; rec->e_l = *E_l; rec->e_r = *E_r;
    rec->e_total = (double)total_energy(c, *E, *E_n, *E_l, *E_r);
    rec->reg_weight_n = c->reg_n; rec->reg_weight_l = c->reg_l;
}

int orc_iterate(orc_ctx* c, int flags, int n_iters, psgsdf_iter_stats* stats) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    float E = (float)ps_energy(c, NULL), E_n = c->reg_n != 0.f ? (float)normal_energy(c) : 0.f, E_l = c->reg_l != 0.f ? (float)laplacian_energy(c) : 0.f;
    float E_r = c->reg_r != 0.f ? (float)albedo_reg_energy(c) : 0.f;
    float E_prev = total_energy(c, E, E_n, E_l, E_r);
    for (int it = 0; it < n_iters; ++it) {
        psgsdf_iter_stats rec; memset(&rec, 0, sizeof(rec));
        iterate_once(c, flags, c->reg_l != 0.f, &E, &E_n, &E_l, &E_r, &rec);
        float Et = (float)rec.e_total;
        rec.rel_diff = (double)(fabsf(E_prev - Et) / E_prev);
        rec.converged = rec.rel_diff < (double)c->set.conv_threshold; rec.diverged = E_prev < Et;
        E_prev = Et;
        if (stats) stats[it] = rec;
    }
    return 0;
}

/* alternatingOptimize, PsOptimizer.cpp:239-428 / LedOptimizer.cpp:279-478 (no file output) */
int orc_optimize(orc_ctx* c, int flags, psgsdf_iter_stats* stats, int stats_cap, int* n_done, int* result, psgsdf_iter_cb on_iter, void* user) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    int led = c->set.model == PSGSDF_LED;
    int laplacian_reg = c->reg_l != 0.f;
    init_albedo(c);
    float E = (float)ps_energy(c, NULL), E_n = 0, E_l = 0;
    if (c->reg_n != 0.f) { E_n = (float)normal_energy(c); c->reg_n *= E / E_n; }
    if (laplacian_reg) { E_l = (float)laplacian_energy(c); c->reg_l *= E / E_l; if (c->set.upsample) laplacian_reg = 0; }
    float E_r = c->reg_r != 0.f ? (float)albedo_reg_energy(c) : 0.f;   /* PsOptimizer.cpp:279 */
    float E_prev = total_energy(c, E, E_n, E_l, E_r);
    int iter = 0, done = 0; if (result) *result = 0;
    while (iter < c->set.max_it) {
        psgsdf_iter_stats rec; memset(&rec, 0, sizeof(rec));
        iterate_once(c, flags, laplacian_reg, &E, &E_n, &E_l, &E_r, &rec);
        float Et = (float)rec.e_total;
        rec.rel_diff = (double)(fabsf(E_prev - Et) / E_prev);
        rec.converged = rec.rel_diff < (double)c->set.conv_threshold;
        rec.diverged = !rec.converged && (E_prev < Et);
        int stop = rec.converged || rec.diverged;
        float E_last = Et;
        if (!stop && iter == 5 && c->set.upsample) {
            if (c->reg_l == 0.0f) c->reg_l = 1.0f;
            laplacian_reg = 1;
            upsample2x(c);
            E_l = (float)laplacian_energy(c);
            c->reg_l *= E / E_l;
            E_last = total_energy(c, E, E_n, E_l, E_r);
            rec.upsampled = 1;
        }
        if (!stop && c->set.upsample && (led ? iter == 15 : iter > 15)) c->reg_l = 0.0f;
        E_prev = E_last;
        if (stats && done < stats_cap) stats[done] = rec;
        done++;
        if (rec.converged) { if (result) *result = 1; break; }
        if (rec.diverged) break;
        ++iter;
        if (on_iter && on_iter(user, iter, &rec)) break;
    }
    if (n_done) *n_done = done;
    return 0;
}

int orc_upsample2x(orc_ctx* c) { if (!c || !c->inited) return PSGSDF_ERR_STATE; upsample2x(c); return 0; }

int orc_get_info(orc_ctx* c, psgsdf_info* info) {
    if (!c || !info) return PSGSDF_ERR_ARG;
    for (int a = 0; a < 3; ++a) { info->dim[a] = c->dim[a]; info->origin[a] = c->origin[a]; }
    info->voxel_size = c->vs; info->n_frames = c->F; info->n_band = c->S;
    info->light_stride = c->set.model == PSGSDF_LED ? 3 : (c->set.model == PSGSDF_SH2 ? 9 : 4);
    info->vis_words = c->wpv; info->reg_weight_n = c->reg_n; info->reg_weight_l = c->reg_l;
    return 0;
}
int orc_download_volume(orc_ctx* c, float* dist, float* grad_xyz, float* weight, float* rgb, uint64_t* vis_words) {
    if (!c || !c->dist) return PSGSDF_ERR_STATE;
    size_t n = c->nvox;
    if (dist) memcpy(dist, c->dist, sizeof(float) * n);
    if (grad_xyz) { memcpy(grad_xyz, c->gx, sizeof(float) * n); memcpy(grad_xyz + n, c->gy, sizeof(float) * n); memcpy(grad_xyz + 2 * n, c->gz, sizeof(float) * n); }
    if (weight) memcpy(weight, c->weight, sizeof(float) * n);
    if (rgb) { memcpy(rgb, c->r, sizeof(float) * n); memcpy(rgb + n, c->g, sizeof(float) * n); memcpy(rgb + 2 * n, c->b, sizeof(float) * n); }
    if (vis_words) { if (!c->vis) return PSGSDF_ERR_STATE; memcpy(vis_words, c->vis, sizeof(uint64_t) * n * c->wpv); }
    return 0;
}
int orc_download_band(orc_ctx* c, int32_t* lin_idx) { if (!c || !c->inited) return PSGSDF_ERR_STATE; memcpy(lin_idx, c->band, sizeof(int) * c->S); return 0; }
int orc_download_poses(orc_ctx* c, float* poses) { if (!c || !c->poses) return PSGSDF_ERR_STATE; memcpy(poses, c->poses, sizeof(float) * 16 * c->F); return 0; }
int orc_download_light(orc_ctx* c, float* light) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    if (c->set.model == PSGSDF_LED) { memcpy(light, c->light, sizeof(float) * 3); return 0; }
    for (int f = 0; f < c->F; ++f) memcpy(light + (size_t)f * c->basis, c->light + (size_t)f * MAXB, sizeof(float) * c->basis);
    return 0;
}
int orc_upload_light(orc_ctx* c, const float* light) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    if (c->set.model == PSGSDF_LED) { memcpy(c->light, light, sizeof(float) * 3); return 0; }
    for (int f = 0; f < c->F; ++f) memcpy(c->light + (size_t)f * MAXB, light + (size_t)f * c->basis, sizeof(float) * c->basis);
    return 0;
}

/* ---- test hooks mirroring psgsdf_debug_* */
int orc_debug_dist_system(orc_ctx* c, float* diag, float* rhs, const float* x, float* y) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    dist_sys s; dist_system(c, c->reg_n != 0.f, c->reg_l != 0.f, &s, NULL, NULL);
    if (diag) memcpy(diag, s.diag, sizeof(float) * c->S);
    if (rhs) memcpy(rhs, s.rhs, sizeof(float) * c->S);
    if (x && y) { dist_mv_ctx m = {&s, 0.0f}; dist_mv(&m, x, y); }
    dist_sys_free(&s);
    return 0;
}
int orc_debug_frame_system(orc_ctx* c, int block, double* H, double* b) {
    if (!c || !c->inited) return PSGSDF_ERR_STATE;
    if (block == PSGSDF_LIGHT) light_system(c, H, b, NULL, NULL);
    else if (block == PSGSDF_POSE) pose_system(c, H, b, NULL, NULL);
    else return PSGSDF_ERR_ARG;
    return 0;
}
int orc_debug_albedo_system(orc_ctx* c, float* H, float* b) { if (!c || !c->inited) return PSGSDF_ERR_STATE; albedo_system(c, H, b, NULL, NULL); return 0; }

/* ---- oracle-only probes used by the known-answer tests */

/* residual (3) of observation (band row j, frame f); returns 1 if visible & in image */
int orc_probe_residual(orc_ctx* c, int j, int f, float r[3], float w[3]) {
    int lin = c->band[j]; if (!vis_bit(c, lin, f)) return 0;
    float R[9], t[3]; pose_Rt(c, f, R, t); obs_geom og;
    return residual_obs(c, lin, f, R, t, r, w, &og);
}
/* analytic distance Jacobian block (4x3) + stencil rows of observation (j,f) */
int orc_probe_dist_jacobian(orc_ctx* c, int j, int f, float J[12], int rows[4]) {
    int lin = c->band[j]; if (!vis_bit(c, lin, f)) return 0;
    float R[9], t[3]; pose_Rt(c, f, R, t);
    float Jb[4][3], dir[3];
    if (!dist_jacobian(c, lin, f, R, t, Jb, dir)) return 0;
    long stride[3] = {1, c->dim[0], (long)c->dim[0] * c->dim[1]};
    rows[0] = j;
    for (int a = 0; a < 3; ++a) { long ln = (long)lin + (long)dir[a] * stride[a]; rows[a + 1] = (ln >= 0 && (size_t)ln < c->nvox) ? band_find(c, (size_t)ln) : -1; }
    for (int k = 0; k < 4; ++k) for (int ch = 0; ch < 3; ++ch) J[k * 3 + ch] = Jb[k][ch];
    return 1;
}
int orc_probe_pose_jacobian(orc_ctx* c, int j, int f, float J[18]) {
    int lin = c->band[j]; if (!vis_bit(c, lin, f)) return 0;
    float R[9], t[3]; pose_Rt(c, f, R, t);
    return pose_jacobian(c, lin, f, R, t, J);
}
int orc_probe_rho_jacobian(orc_ctx* c, int j, int f, float J[3]) {
    int lin = c->band[j]; float R[9], t[3]; pose_Rt(c, f, R, t); rho_jacobian(c, lin, f, R, t, J); return 1;
}
/* Eikonal row of band row j (Optimizer.cpp:196-218): Jr[4] over {self, 3 stencil neighbours}, residual |g| - 1, stencil rows (-1 = column dropped) */
int orc_probe_eikonal(orc_ctx* c, int j, float Jr[4], float* res, int rows[4]) {
    int lin = c->band[j]; float dir[3];
    eikonal_row(c, lin, Jr, res, dir);
    long stride[3] = {1, c->dim[0], (long)c->dim[0] * c->dim[1]};
    rows[0] = j;
    for (int a = 0; a < 3; ++a) { long ln = (long)lin + (long)dir[a] * stride[a]; rows[a + 1] = (ln >= 0 && (size_t)ln < c->nvox) ? band_find(c, (size_t)ln) : -1; }
    return 1;
}
/* Laplacian residual of band row j (Optimizer.cpp:368-393) and the only Jacobian entry the reference emits (B3): the diagonal -6 / vs^2 */
int orc_probe_laplacian(orc_ctx* c, int j, float* res, float* Jdiag) {
    *res = dist_laplacian(c, c->band[j]); *Jdiag = -6 * (c->vs_inv * c->vs_inv);
    return 1;
}
/* albedo regulariser of band row j: J[slot][ch] (slot 0 = the voxel, 1..3 = x/y/z stencil neighbour), res[ch] = ||grad rho_ch||, nb[a] = linear index of the neighbour */
int orc_probe_albedo_reg(orc_ctx* c, int j, float J[12], float res[3], int64_t nb[3]) {
    float Jm[4][3]; long nl[3]; albedo_reg_jacobian(c, c->band[j], Jm, res, nl);
    for (int q = 0; q < 4; ++q) for (int ch = 0; ch < 3; ++ch) J[q * 3 + ch] = Jm[q][ch];
    for (int a = 0; a < 3; ++a) nb[a] = (int64_t)nl[a];
    return 1;
}
/* direct state pokes for finite-difference tests */
int orc_poke_dist(orc_ctx* c, int lin, float v) { c->dist[lin] = v; return 0; }
float orc_peek_dist(orc_ctx* c, int lin) { return c->dist[lin]; }
int orc_poke_grad(orc_ctx* c, int lin, const float g[3]) { c->gx[lin] = g[0]; c->gy[lin] = g[1]; c->gz[lin] = g[2]; return 0; }
int orc_peek_grad(orc_ctx* c, int lin, float g[3]) { g[0] = c->gx[lin]; g[1] = c->gy[lin]; g[2] = c->gz[lin]; return 0; }
int orc_poke_rgb(orc_ctx* c, int lin, const float v[3]) { c->r[lin] = v[0]; c->g[lin] = v[1]; c->b[lin] = v[2]; return 0; }
int orc_peek_rgb(orc_ctx* c, int lin, float v[3]) { v[0] = c->r[lin]; v[1] = c->g[lin]; v[2] = c->b[lin]; return 0; }
int orc_poke_pose(orc_ctx* c, int f, const float P[16]) { memcpy(c->poses + 16 * f, P, sizeof(float) * 16); return 0; }
int orc_update_grad(orc_ctx* c) { update_grad(c); return 0; }
int orc_so3_exp(const float w[3], float R[9]) { so3_exp(w, R); return 0; }
/* Eigen-CG on a dense SPD matrix (n x n row-major) for the solver known-answer test */
typedef struct { int n; const float* A; } dense_mv_ctx;
static void dense_mv(void* u, const float* p, float* out) { dense_mv_ctx* d = (dense_mv_ctx*)u; for (int i = 0; i < d->n; ++i) { double s = 0; for (int j = 0; j < d->n; ++j) s += (double)d->A[i * d->n + j] * p[j]; out[i] = (float)s; } }
int orc_eigen_cg_dense(int n, const float* A, const float* b, float* x, int* iters, double* err) {
    float* diag = (float*)malloc(sizeof(float) * n); for (int i = 0; i < n; ++i) diag[i] = A[i * n + i];
    dense_mv_ctx d = {n, A}; cg_result r = eigen_cg(n, dense_mv, &d, diag, b, x, 0);
    if (iters) *iters = r.iters; if (err) *err = r.error; free(diag); return r.success;
}
