/*
 * fusion_oracle.c -- CPU restatement of the reference's dense-fusion stage.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call it.  The product path
 * (sfm_amd/) never links or imports anything from oracle/.
 *
 * PARITY STATUS: "parity unpinned".  The reference (CarterDiOrio/SfM) ships no tests,
 * fixtures or golden clouds for this path and cannot be built in
 * this image (needs Eigen, OpenCV, DBoW2, Ceres, Sophus, Boost).  This file is a line by
 * line restatement of the algorithm, pinned only by hand-derived known-answer tests
 * (tests/test_oracle_kat.py) and by an independent pure-Python restatement
 * (tests/py_restatement.py).
 *
 * What it follows (paths relative to /root/reference/reconstruction):
 *   src/dense_reconstruction.cpp:29-65    reconstruct()       -> orc_fuse()
 *   src/dense_reconstruction.cpp:67-124   process_frame()     -> fuse_frame()
 *   src/dense_reconstruction.cpp:126-175  process_pixel()     -> process_pixel()
 *   src/dense_reconstruction.cpp:177-234  geometric_check()   -> inlined in process_pixel()
 *   src/dense_reconstruction.cpp:263-290  voxel_filter()      -> voxel_filter()
 *   src/dense_reconstruction.cpp:292-313  jacobian/covariance -> orc_cov_trace()
 *   src/pinhole.cpp:8-30                  deproject           -> orc_deproject()
 *   src/pinhole.cpp:32-45                 project             -> orc_project()
 *   src/keyframe.cpp:275-294              point_in_frame      -> orc_point_in_frame()
 *   src/features.cpp:76-80                extract_color       -> colour read in process_pixel()
 *   include/reconstruction/dense_reconstruction.hpp:44-56     constants -> orc_default_params()
 *
 * Arithmetic model: IEEE-754 binary64, round-to-nearest, no FMA contraction (the
 * reference builds with clang -O3 and no -march, CMakeLists.txt:3-5 => SSE2).  Build this
 * file with -ffp-contract=off.
 *
 * Evaluation orders taken from Eigen 3.3/3.4 for fixed-size double expressions under SSE2
 * (Eigen is a header-only dependency that is absent here; the orders below are restated
 * from its published sources Core/Redux.h, Core/ProductEvaluators.h, Geometry/Homogeneous.h):
 *   - Matrix4d * Vector3d.homogeneous():  dst = T.leftCols<3>() * v (packet path, inner
 *     index ascending: ((m0*x + m1*y) + m2*z)), then dst += T.col(3).
 *   - Matrix3d * Matrix3d lazy coefficient: (a0*b0) + ((a1*b1) + (a2*b2))   (scalar path,
 *     redux_novec_unroller splits [0,1) / [1,3)).
 *   - Matrix3d::trace(): diagonal().sum() -> d0 + (d1 + d2)                 (same unroller).
 *   - Vector3d::norm()/dot(): linear vectorised redux, Packet2d: (e0 + e1) + e2.
 *   - J*S*J^T evaluates M = J*S into a temporary, then M * J^T.
 *
 * Deliberate canonicalisations: frames are processed in the order given
 * (the reference uses shared_ptr address order), neighbours in the order given (the
 * reference uses adjacency-vector order), and the output cloud is sorted by voxel key
 * (the reference emits unordered_map iteration order).
 *
 * One defined behaviour where the reference has UB: a NaN projection (0/0) passes the
 * reference's bounds test (keyframe.cpp:280) and indexes out of bounds; here it is
 * treated as "not visible".
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double base_line;       /* dense_reconstruction.hpp:44 */
    double T_cov;           /* :46 */
    double voxel_size;      /* :52 */
    double T_dist;          /* :54 */
    double min_dist_filter; /* :56 */
    double dir_dot;         /* dense_reconstruction.cpp:85 (0.30) */
    double S[3];            /* dense_reconstruction.cpp:23-25 diag(0.05,0.05,10) */
    double depth_div;       /* dense_reconstruction.cpp:309 (1000) */
    double close_frac;      /* dense_reconstruction.cpp:229 (0.50) */
} orc_params;

typedef struct {
    int32_t W, H;
    const uint16_t *depth;  /* row-major CV_16UC1 */
    uint64_t depth_stride;  /* bytes per row */
    const uint8_t *bgr;     /* row-major CV_8UC3, B,G,R */
    uint64_t bgr_stride;    /* bytes per row */
    double K[4];            /* fx, fy, cx, cy (pinhole.hpp:15-28) */
    double T_wk[16];        /* camera->world, column-major like Eigen::Matrix4d::data() */
    double T_kw[16];        /* world->camera, column-major */
    const uint32_t *nbrs;   /* Map::get_neighbors(kf,100) as frame indices, in order */
    uint32_t n_nbrs;
} orc_frame;

typedef struct {
    uint64_t n_points;
    double *xyz;            /* n_points x 3 */
    int32_t *bgr;           /* n_points x 3, B,G,R order as PointInfo::color */
    uint64_t *key;          /* n_points x 3 */
    uint64_t *tag;          /* n_points: (frame_index << 32) | raster_index */
    uint64_t *frame_before; /* per frame: "Points before voxel filter" (:118) */
    uint64_t *frame_after;  /* per frame: "Points after voxel filter"  (:120) */
    uint64_t global_before; /* "Points before voxel removal" (:56) */
    uint64_t global_after;  /* "Points after voxel removal"  (:58) */
    /* traffic counters for the roofline's algorithmic byte count */
    uint64_t n_px, n_gather, n_acc, n_hit, n_vox;
    /* where live pixels end: diagnostic split of the rejections */
    uint64_t n_live, n_rej_cov, n_rej_noobs, n_rej_far;
} orc_result;

void orc_default_params(orc_params *p)
{
    p->base_line = 0.50;
    p->T_cov = 0.5;
    p->voxel_size = 0.05;
    p->T_dist = 0.50;
    p->min_dist_filter = 0.0;
    p->dir_dot = 0.30;
    p->S[0] = 0.05; p->S[1] = 0.05; p->S[2] = 10.0;
    p->depth_div = 1000.0;
    p->close_frac = 0.50;
}

/* ---- small fixed-size algebra in Eigen's evaluation order -------------------------- */

/* Evaluation-order VARIANTS (tests/test_oracle_variants.py, tools/oracle_order_risk.py).  Eigen is absent from this
 * image, so the orders above are restated from its sources and cannot be checked against the real binary; these
 * switches rebuild the oracle with the plausible alternatives so that the risk is a measured number: how many voxel
 * keys, winners and coordinate bits change.  0 (the default) is the oracle; nothing else is ever used as a checker.
 *   1  trace() = (d0 + d1) + d2               instead of d0 + (d1 + d2)
 *   2  T*[p;1] row = (m0*x + (m1*y + m2*z)) + m3   instead of ((m0*x + m1*y) + m2*z) + m3
 *   4  3x3 product coefficient = (a0*b0 + a1*b1) + a2*b2   instead of a0*b0 + (a1*b1 + a2*b2)
 *   8  FMA contraction (what -march=native / -ffp-contract=fast would allow) in the products and sums above,
 *      in the projection (x/z)*f + c, in norm() and in the weighted sums */
static int g_variant = 0;
void orc_set_variant(int flags) { g_variant = flags; }
int orc_get_variant(void) { return g_variant; }

/* C = A*B, 3x3 row-major arrays, coefficient = a0*b0 + (a1*b1 + a2*b2). */
static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double t0 = A[i * 3 + 0] * B[0 * 3 + j];
            double t1 = A[i * 3 + 1] * B[1 * 3 + j];
            double t2 = A[i * 3 + 2] * B[2 * 3 + j];
            if (g_variant & 8) {
                if (g_variant & 4) C[i * 3 + j] = fma(A[i * 3 + 2], B[2 * 3 + j], fma(A[i * 3 + 1], B[1 * 3 + j], t0));
                else C[i * 3 + j] = fma(A[i * 3 + 0], B[0 * 3 + j], fma(A[i * 3 + 1], B[1 * 3 + j], t2));
            } else if (g_variant & 4) {
                C[i * 3 + j] = (t0 + t1) + t2;
            } else {
                C[i * 3 + j] = t0 + (t1 + t2);
            }
        }
}

/* rows 0..2 of T * [p;1]; T column-major 4x4. */
static void mat4_apply(const double T[16], const double p[3], double out[3])
{
    for (int r = 0; r < 3; r++) {
        if (g_variant & 8) {
            double acc = T[r + 0] * p[0];
            if (g_variant & 2) acc = fma(T[r + 0], p[0], fma(T[r + 4], p[1], T[r + 8] * p[2]));
            else acc = fma(T[r + 8], p[2], fma(T[r + 4], p[1], acc));
            out[r] = acc + T[r + 12];
        } else if (g_variant & 2) {
            out[r] = (T[r + 0] * p[0] + (T[r + 4] * p[1] + T[r + 8] * p[2])) + T[r + 12];
        } else {
            double acc = T[r + 0] * p[0];
            acc = acc + T[r + 4] * p[1];
            acc = acc + T[r + 8] * p[2];
            out[r] = acc + T[r + 12];
        }
    }
}

/* dense_reconstruction.cpp:292-313.  Returns trace(J*S*J^T). */
double orc_cov_trace(const orc_params *p, int u, int v, double d, double focal_length)
{
    double disparity = (focal_length * p->base_line) / (d / p->depth_div);
    double dd = disparity * disparity;
    double J[9] = {
        p->base_line / disparity, 0.0, (double)(-u) * p->base_line / dd,
        0.0, p->base_line / disparity, (double)(-v) * p->base_line / dd,
        0.0, 0.0, -focal_length * p->base_line / dd
    };
    double S[9] = { p->S[0], 0, 0, 0, p->S[1], 0, 0, 0, p->S[2] };
    double Jt[9], M[9], P[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            Jt[i * 3 + j] = J[j * 3 + i];
    mat3_mul(J, S, M);
    mat3_mul(M, Jt, P);
    if (g_variant & 1) return (P[0] + P[4]) + P[8];
    return P[0] + (P[4] + P[8]);
}

/* pinhole.cpp:8-30 */
void orc_deproject(const double K[4], int px, int py, double depth, double out[3])
{
    double x = ((double)px - K[2]) / K[0];
    double y = ((double)py - K[3]) / K[1];
    out[0] = x * depth;
    out[1] = y * depth;
    out[2] = depth;
}

/* pinhole.cpp:32-45; also returns camera-frame z for keyframe.cpp:285-288 */
void orc_project(const double K[4], const double T_kw[16], const double pw[3],
                 double *px, double *py, double *cz)
{
    double c[3];
    mat4_apply(T_kw, pw, c);
    if (g_variant & 8) {
        *px = fma(c[0] / c[2], K[0], K[2]);
        *py = fma(c[1] / c[2], K[1], K[3]);
    } else {
        *px = (c[0] / c[2]) * K[0] + K[2];
        *py = (c[1] / c[2]) * K[1] + K[3];
    }
    *cz = c[2];
}

/* keyframe.cpp:275-294.  Returns 1 and the floored pixel when visible. */
int orc_point_in_frame(const double K[4], const double T_kw[16], int W, int H,
                       const double pw[3], int *ox, int *oy)
{
    double px, py, cz;
    orc_project(K, T_kw, pw, &px, &py, &cz);
    if (px != px || py != py) return 0; /* defined here; UB in the reference */
    if (px < 0 || px >= (double)W || py < 0 || py >= (double)H) return 0;
    if (cz < 0) return 0;
    *ox = (int)floor(px);
    *oy = (int)floor(py);
    return 1;
}

/* dense_reconstruction.cpp:271-273: static_cast<size_t>(coord / voxel_size) as the
 * reference's toolchain (clang, x86-64, SSE2 cvttsd2si pair) evaluates it. */
uint64_t orc_voxel_key(double coord, double voxel_size)
{
    double q = coord / voxel_size;
    if (q != q) return 0x8000000000000000ull;
    double t = trunc(q);
    if (t >= -9223372036854775808.0 && t < 9223372036854775808.0)
        return (uint64_t)(int64_t)t;
    if (t >= 9223372036854775808.0 && t < 18446744073709551616.0)
        return (uint64_t)t;
    return 0x8000000000000000ull;
}

/* dense_reconstruction.cpp:72-89: keep neighbour lf of frame f? */
int orc_neighbor_kept(const orc_params *p, const double Twk_f[16], const double Twk_lf[16])
{
    double dx = Twk_lf[12] - Twk_f[12];
    double dy = Twk_lf[13] - Twk_f[13];
    double dz = Twk_lf[14] - Twk_f[14];
    double n = sqrt((dx * dx + dy * dy) + dz * dz);
    if (n > p->min_dist_filter) return 1;
    /* d1 = R_f * (0,0,1), d2 = R_lf * (0,0,1): third columns */
    double a0 = (Twk_f[0] * 0.0 + Twk_f[4] * 0.0) + Twk_f[8] * 1.0;
    double a1 = (Twk_f[1] * 0.0 + Twk_f[5] * 0.0) + Twk_f[9] * 1.0;
    double a2 = (Twk_f[2] * 0.0 + Twk_f[6] * 0.0) + Twk_f[10] * 1.0;
    double b0 = (Twk_lf[0] * 0.0 + Twk_lf[4] * 0.0) + Twk_lf[8] * 1.0;
    double b1 = (Twk_lf[1] * 0.0 + Twk_lf[5] * 0.0) + Twk_lf[9] * 1.0;
    double b2 = (Twk_lf[2] * 0.0 + Twk_lf[6] * 0.0) + Twk_lf[10] * 1.0;
    double dot = (a0 * b0 + a1 * b1) + a2 * b2;
    return dot < p->dir_dot;
}

/* ---- per-pixel work ---------------------------------------------------------------- */

typedef struct {
    double p[3];
    int32_t bgr[3];
    uint64_t tag;
} point_rec;

typedef struct {
    uint64_t n_gather, n_hit, n_live, n_rej_cov, n_rej_noobs, n_rej_far;
} px_counters;

static inline uint16_t depth_at(const orc_frame *f, int x, int y)
{
    return *(const uint16_t *)((const uint8_t *)f->depth + (uint64_t)y * f->depth_stride + 2u * (uint64_t)x);
}

/* process_pixel + geometric_check.  masks[k] is frame k's visibility mask (W*H bytes).
 * Returns 1 and fills *out when the pixel yields a point. */
static int process_pixel(const orc_params *P, const orc_frame *frames, uint8_t **masks,
                         uint32_t fi, const uint32_t *nb, uint32_t n_nb, int u, int v,
                         point_rec *out, px_counters *ctr,
                         double *obs_p, double *obs_w, uint32_t *obs_frame, uint32_t *obs_pix)
{
    const orc_frame *f = &frames[fi];
    uint16_t depth = depth_at(f, u, v);
    if (depth <= 0) return 0;                                        /* :132 */
    ctr->n_live++;
    double tr = orc_cov_trace(P, u, v, (double)depth, f->K[0]);      /* :136 */
    if (tr > P->T_cov) { ctr->n_rej_cov++; return 0; }               /* :138 */
    double pc[3], pw[3];
    orc_deproject(f->K, u, v, (double)depth, pc);                    /* :143 */
    mat4_apply(f->T_wk, pc, pw);                                     /* :144 */

    if (n_nb == 0) { ctr->n_rej_noobs++; return 0; }                 /* :180 */
    uint32_t n_obs = 0;
    for (uint32_t k = 0; k < n_nb; k++) {                            /* :187 */
        const orc_frame *lf = &frames[nb[k]];
        int x, y;
        if (!orc_point_in_frame(lf->K, lf->T_kw, lf->W, lf->H, pw, &x, &y)) continue;
        uint16_t d = depth_at(lf, x, y);                             /* :193, no d>0 test */
        ctr->n_gather++;
        double qc[3], qw[3];
        orc_deproject(lf->K, x, y, (double)d, qc);                   /* :196 */
        mat4_apply(lf->T_wk, qc, qw);                                /* :199 */
        double w = orc_cov_trace(P, x, y, (double)d, lf->K[0]);      /* :202 */
        if (w < P->T_cov) {                                          /* :204 */
            obs_p[3 * n_obs + 0] = qw[0];
            obs_p[3 * n_obs + 1] = qw[1];
            obs_p[3 * n_obs + 2] = qw[2];
            obs_w[n_obs] = w;
            obs_frame[n_obs] = nb[k];
            obs_pix[n_obs] = (uint32_t)y * (uint32_t)lf->W + (uint32_t)x;
            n_obs++;
        }
    }
    if (n_obs < 1) { ctr->n_rej_noobs++; return 0; }                 /* :218 */
    uint64_t close_count = 0;
    for (uint32_t i = 0; i < n_obs; i++) {                           /* :223 */
        double dx = obs_p[3 * i + 0] - pw[0];
        double dy = obs_p[3 * i + 1] - pw[1];
        double dz = obs_p[3 * i + 2] - pw[2];
        double n = (g_variant & 8) ? sqrt(fma(dz, dz, fma(dy, dy, dx * dx))) : sqrt((dx * dx + dy * dy) + dz * dz);
        if (n < P->T_dist) close_count++;
    }
    if ((double)close_count < (double)n_obs * P->close_frac) { ctr->n_rej_far++; return 0; } /* :229 */

    for (uint32_t i = 0; i < n_obs; i++) {                           /* :152-154 */
        masks[obs_frame[i]][obs_pix[i]] = 255;
        ctr->n_hit++;
    }
    double wp[3] = { 0, 0, 0 }, total = 0;                           /* :158-165 */
    for (uint32_t i = 0; i < n_obs; i++) {
        double inv = 1.0 / obs_w[i];
        if (g_variant & 8) {
            wp[0] = fma(obs_p[3 * i + 0], inv, wp[0]);
            wp[1] = fma(obs_p[3 * i + 1], inv, wp[1]);
            wp[2] = fma(obs_p[3 * i + 2], inv, wp[2]);
        } else {
        wp[0] = wp[0] + obs_p[3 * i + 0] * inv;
        wp[1] = wp[1] + obs_p[3 * i + 1] * inv;
        wp[2] = wp[2] + obs_p[3 * i + 2] * inv;
        }
        total = total + 1.0 / obs_w[i];
    }
    out->p[0] = wp[0] / total;
    out->p[1] = wp[1] / total;
    out->p[2] = wp[2] / total;
    const uint8_t *c = f->bgr + (uint64_t)v * f->bgr_stride + 3u * (uint64_t)u; /* :170 */
    out->bgr[0] = c[0]; out->bgr[1] = c[1]; out->bgr[2] = c[2];
    out->tag = ((uint64_t)fi << 32) | ((uint64_t)v * (uint64_t)f->W + (uint64_t)u);
    return 1;
}

/* ---- first-wins voxel filter (dense_reconstruction.cpp:263-290) --------------------- */

typedef struct { uint64_t k[3]; uint64_t first; } vslot; /* first = index+1, 0 = empty */

static uint64_t mix3(const uint64_t k[3])
{
    uint64_t h = k[0] * 0x9E3779B97F4A7C15ull;
    h ^= (k[1] + 0xC2B2AE3D27D4EB4Full) * 0xD6E8FEB86659FD93ull + (h << 6) + (h >> 2);
    h ^= (k[2] + 0x165667B19E3779F9ull) * 0xFF51AFD7ED558CCDull + (h << 6) + (h >> 2);
    h ^= h >> 29;
    return h;
}

/* Keeps, for each voxel, the first point in list order.  Returns the kept indices in
 * list order (the reference emits map order; order is canonicalised later). */
static uint64_t voxel_filter(const orc_params *P, const point_rec *pts, uint64_t n, uint64_t *kept)
{
    uint64_t cap = 16;
    while (cap < 2 * n + 1) cap <<= 1;
    vslot *tab = (vslot *)calloc(cap, sizeof(vslot));
    uint64_t n_kept = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t k[3] = { orc_voxel_key(pts[i].p[0], P->voxel_size),
                          orc_voxel_key(pts[i].p[1], P->voxel_size),
                          orc_voxel_key(pts[i].p[2], P->voxel_size) };
        uint64_t h = mix3(k) & (cap - 1);
        for (;;) {
            if (tab[h].first == 0) {
                tab[h].k[0] = k[0]; tab[h].k[1] = k[1]; tab[h].k[2] = k[2];
                tab[h].first = i + 1;
                kept[n_kept++] = i;
                break;
            }
            if (tab[h].k[0] == k[0] && tab[h].k[1] == k[1] && tab[h].k[2] == k[2]) break;
            h = (h + 1) & (cap - 1);
        }
    }
    free(tab);
    return n_kept;
}

/* ---- frame loop -------------------------------------------------------------------- */

typedef struct {
    const orc_params *P;
    const orc_frame *frames;
    uint8_t **masks;
    uint32_t fi;
    const uint32_t *nb;
    uint32_t n_nb;
    int row0, row1;
    uint8_t *got;
    point_rec *local;
    px_counters ctr;
} job;

static void *job_run(void *arg)
{
    job *j = (job *)arg;
    const orc_frame *f = &j->frames[j->fi];
    uint32_t cap = j->n_nb ? j->n_nb : 1;
    double *obs_p = (double *)malloc(sizeof(double) * 3 * cap);
    double *obs_w = (double *)malloc(sizeof(double) * cap);
    uint32_t *obs_f = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    uint32_t *obs_x = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    const uint8_t *mask = j->masks[j->fi];
    for (int v = j->row0; v < j->row1; v++)
        for (int u = 0; u < f->W; u++) {
            uint64_t i = (uint64_t)v * (uint64_t)f->W + (uint64_t)u;
            if (mask[i] <= 0) {                                      /* :100 */
                if (depth_at(f, u, v) > 0) {                         /* :101 */
                    if (process_pixel(j->P, j->frames, j->masks, j->fi, j->nb, j->n_nb, u, v,
                                      &j->local[i], &j->ctr, obs_p, obs_w, obs_f, obs_x))
                        j->got[i] = 1;
                }
            }
        }
    free(obs_p); free(obs_w); free(obs_f); free(obs_x);
    return NULL;
}

/* frames are processed in index order.  n_threads splits the pixel loop of one frame over
 * host threads by row blocks, like the reference's par_unseq loop (:95-110).
 *
 * orc_fuse_ex is the same loop cut at keyframe boundaries, for checking the multi-GPU split of ONE sequence
 * (no reference counterpart: the reference is one process): only frames [first, end) are processed (:48-52),
 * the others only lend their images and poses as neighbours and own a visibility mask.  mask_seed[i] (or NULL)
 * is frame i's mask at the start instead of zeros (:40-45) -- what the ranks before this one left --, and
 * mask_out[i] (or NULL) receives frame i's mask at the end.  orc_fuse == orc_fuse_ex(0, n_frames, NULL, NULL):
 * running the blocks one after the other, each seeded with its predecessors' masks, is by construction the
 * same sequence of process_frame calls on the same mask states. */
int orc_fuse_ex(const orc_params *P, const orc_frame *frames, uint32_t n_frames, int n_threads,
                uint32_t first, uint32_t end, const uint8_t *const *mask_seed, uint8_t *const *mask_out,
                orc_result *out);

int orc_fuse(const orc_params *P, const orc_frame *frames, uint32_t n_frames, int n_threads,
             orc_result *out)
{
    return orc_fuse_ex(P, frames, n_frames, n_threads, 0, n_frames, NULL, NULL, out);
}

int orc_fuse_ex(const orc_params *P, const orc_frame *frames, uint32_t n_frames, int n_threads,
                uint32_t first, uint32_t end, const uint8_t *const *mask_seed, uint8_t *const *mask_out,
                orc_result *out)
{
    memset(out, 0, sizeof(*out));
    if (n_threads < 1) n_threads = 1;
    if (end > n_frames) end = n_frames;
    if (first > end) first = end;
    uint8_t **masks = (uint8_t **)calloc(n_frames ? n_frames : 1, sizeof(uint8_t *));
    uint64_t max_px = 0, max_nb = 0;
    for (uint32_t i = 0; i < n_frames; i++) {                        /* :40-45 */
        uint64_t px = (uint64_t)frames[i].W * (uint64_t)frames[i].H;
        masks[i] = (uint8_t *)calloc(px ? px : 1, 1);
        if (mask_seed && mask_seed[i] && px) memcpy(masks[i], mask_seed[i], px);
        if (px > max_px) max_px = px;
        if (frames[i].n_nbrs > max_nb) max_nb = frames[i].n_nbrs;
        if (i >= first && i < end) out->n_px += px;
        for (uint32_t k = 0; k < frames[i].n_nbrs; k++)
            if (frames[i].nbrs[k] >= n_frames || frames[i].nbrs[k] == i) return -1;
    }
    out->frame_before = (uint64_t *)calloc(n_frames ? n_frames : 1, sizeof(uint64_t));
    out->frame_after = (uint64_t *)calloc(n_frames ? n_frames : 1, sizeof(uint64_t));

    uint64_t all_cap = 1024, all_n = 0;
    point_rec *all = (point_rec *)malloc(all_cap * sizeof(point_rec));
    uint8_t *got = (uint8_t *)malloc(max_px ? max_px : 1);
    point_rec *local = (point_rec *)malloc((max_px ? max_px : 1) * sizeof(point_rec));
    point_rec *fpts = (point_rec *)malloc((max_px ? max_px : 1) * sizeof(point_rec));
    uint64_t *kept = (uint64_t *)malloc((max_px ? max_px : 1) * sizeof(uint64_t));
    uint32_t *nb = (uint32_t *)malloc((max_nb ? max_nb : 1) * sizeof(uint32_t));
    job *jobs = (job *)calloc((size_t)n_threads, sizeof(job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));

    for (uint32_t fi = first; fi < end; fi++) {                      /* :48-52 */
        const orc_frame *f = &frames[fi];
        uint64_t px = (uint64_t)f->W * (uint64_t)f->H;
        uint32_t n_nb = 0;
        for (uint32_t k = 0; k < f->n_nbrs; k++)                     /* :72-89 */
            if (orc_neighbor_kept(P, f->T_wk, frames[f->nbrs[k]].T_wk)) nb[n_nb++] = f->nbrs[k];
        memset(got, 0, px);
        int nt = n_threads;
        if (nt > f->H) nt = f->H > 0 ? f->H : 1;
        for (int t = 0; t < nt; t++) {
            job *j = &jobs[t];
            j->P = P; j->frames = frames; j->masks = masks; j->fi = fi; j->nb = nb; j->n_nb = n_nb;
            j->row0 = (int)((int64_t)f->H * t / nt);
            j->row1 = (int)((int64_t)f->H * (t + 1) / nt);
            j->got = got; j->local = local; memset(&j->ctr, 0, sizeof j->ctr);
            if (nt == 1) job_run(j); else pthread_create(&th[t], NULL, job_run, j);
        }
        for (int t = 0; t < nt; t++) {
            if (nt > 1) pthread_join(th[t], NULL);
            out->n_gather += jobs[t].ctr.n_gather;
            out->n_hit += jobs[t].ctr.n_hit;
            out->n_live += jobs[t].ctr.n_live;
            out->n_rej_cov += jobs[t].ctr.n_rej_cov;
            out->n_rej_noobs += jobs[t].ctr.n_rej_noobs;
            out->n_rej_far += jobs[t].ctr.n_rej_far;
        }
        uint64_t n_f = 0;
        for (uint64_t i = 0; i < px; i++)                            /* :112-116 raster order */
            if (got[i]) fpts[n_f++] = local[i];
        out->frame_before[fi] = n_f;
        out->n_acc += n_f;
        uint64_t n_kept = voxel_filter(P, fpts, n_f, kept);          /* :119 */
        out->frame_after[fi] = n_kept;
        if (all_n + n_kept > all_cap) {
            while (all_n + n_kept > all_cap) all_cap <<= 1;
            all = (point_rec *)realloc(all, all_cap * sizeof(point_rec));
        }
        for (uint64_t i = 0; i < n_kept; i++) all[all_n++] = fpts[kept[i]]; /* :51 */
    }
    out->global_before = all_n;                                      /* :56 */
    uint64_t *gkept = (uint64_t *)malloc((all_n ? all_n : 1) * sizeof(uint64_t));
    uint64_t n_out = voxel_filter(P, all, all_n, gkept);             /* :57 */
    out->global_after = n_out;
    out->n_vox = n_out;
    out->n_points = n_out;
    out->xyz = (double *)malloc((n_out ? n_out : 1) * 3 * sizeof(double));
    out->bgr = (int32_t *)malloc((n_out ? n_out : 1) * 3 * sizeof(int32_t));
    out->key = (uint64_t *)malloc((n_out ? n_out : 1) * 3 * sizeof(uint64_t));
    out->tag = (uint64_t *)malloc((n_out ? n_out : 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < n_out; i++) {
        const point_rec *r = &all[gkept[i]];
        for (int c = 0; c < 3; c++) {
            out->xyz[3 * i + c] = r->p[c];
            out->bgr[3 * i + c] = r->bgr[c];
            out->key[3 * i + c] = orc_voxel_key(r->p[c], P->voxel_size);
        }
        out->tag[i] = r->tag;
    }
    for (uint32_t i = 0; i < n_frames; i++) {
        if (mask_out && mask_out[i]) memcpy(mask_out[i], masks[i], (uint64_t)frames[i].W * (uint64_t)frames[i].H);
        free(masks[i]);
    }
    free(masks); free(all); free(got); free(local); free(fpts); free(kept); free(gkept);
    free(nb); free(jobs); free(th);
    return 0;
}

void orc_free_result(orc_result *r)
{
    free(r->xyz); free(r->bgr); free(r->key); free(r->tag);
    free(r->frame_before); free(r->frame_after);
    memset(r, 0, sizeof(*r));
}
