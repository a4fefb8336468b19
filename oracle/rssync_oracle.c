/*
 * rssync_oracle.c -- CPU restatement (plain C, IEEE double) of the rs-sync
 * PreSync/Sync hot path.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see
 * rssync_oracle.h for what that means and why.
 *
 * Reference citations are into /root/reference/src/ (never read at run time).
 */
#define _GNU_SOURCE
#include "rssync_oracle.h"

#include <math.h>
#include <stdint.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* state: core/core_private.hpp:8-22 (FrameData, OptData), :24-42 (FrameState) */

typedef struct {
    int64_t id;
    size_t n;
    double *ts_a, *ts_b;     /* [n] seconds */
    double *rays_a, *rays_b; /* [3n] xyz interleaved */
    double M[3];             /* FrameState::motion_vec */
    double k;                /* FrameState::var_k */
} frame_t;

typedef struct {
    size_t n;
    double *y, *b, *c, *d; /* minispline.hpp:15 m_y, m_b, m_c, m_d */
} spline1_t;

struct ora_problem {
    double quats_start, sample_rate;
    spline1_t sp[4];
    size_t gyro_n;
    frame_t* frames; /* ascending id */
    size_t nframes, cap;
    uint64_t seed;
    int nthreads, max_outer, faithful, verbose, lbfgs_reeval;
    long lbfgs_best_not_last; /* line searches whose best step was not the last one tried */
    uint32_t sync_calls;
    /* test hooks: GuessMotion's winning hypothesis per selected frame of the last Sync, and winners to use
     * instead of the search in the next one (so that two implementations can be started from the same estimates) */
    int32_t* last_init;
    size_t n_last_init;
    int32_t* init_override;
    size_t n_init_override;
    /* frames selected by the last Sync (indices into frames) */
    size_t* sel;
    size_t nsel;
    char err[256];
};

static int fail(ora_problem* p, const char* msg) {
    snprintf(p->err, sizeof p->err, "%s", msg);
    return 1;
}

ora_problem* ora_create(void) {
    ora_problem* p = (ora_problem*)calloc(1, sizeof *p);
    p->seed = 0x5EED0000ULL;
    p->nthreads = 1;
    p->max_outer = 400; /* core_private.cpp:309 */
    p->faithful = 1;
    return p;
}

static void spline_free(spline1_t* s) {
    free(s->y); free(s->b); free(s->c); free(s->d);
    memset(s, 0, sizeof *s);
}

static void frame_free(frame_t* f) {
    free(f->ts_a); free(f->ts_b); free(f->rays_a); free(f->rays_b);
}

void ora_destroy(ora_problem* p) {
    if (!p) return;
    for (int i = 0; i < 4; ++i) spline_free(&p->sp[i]);
    for (size_t i = 0; i < p->nframes; ++i) frame_free(&p->frames[i]);
    free(p->frames);
    free(p->sel);
    free(p->last_init);
    free(p->init_override);
    free(p);
}

const char* ora_last_error(const ora_problem* p) { return p->err; }
void ora_set_seed(ora_problem* p, uint64_t seed) { p->seed = seed; }
void ora_set_threads(ora_problem* p, int n) { p->nthreads = n < 1 ? 1 : n; }
void ora_set_max_outer_iters(ora_problem* p, int it) { p->max_outer = it; }
void ora_set_faithful(ora_problem* p, int f) { p->faithful = f; }
void ora_set_verbose(ora_problem* p, int v) { p->verbose = v; }
void ora_set_lbfgs_reeval(ora_problem* p, int v) { p->lbfgs_reeval = v; }
long ora_lbfgs_best_not_last(const ora_problem* p) { return p->lbfgs_best_not_last; }
double ora_sample_rate(const ora_problem* p) { return p->sample_rate; }
double ora_quats_start(const ora_problem* p) { return p->quats_start; }
size_t ora_gyro_count(const ora_problem* p) { return p->gyro_n; }
size_t ora_frame_count(const ora_problem* p) { return p->nframes; }

void ora_gyro_knots(const ora_problem* p, double* out) {
    for (size_t i = 0; i < p->gyro_n; ++i)
        for (int c = 0; c < 4; ++c) out[4 * i + c] = p->sp[c].y[i];
}

/* ------------------------------------------------------------------ */
/* frames-parallel loop: stands in for std::for_each(std::execution::par, ...)
 * (core_private.cpp:73,231,245,263,348).  Results are written per frame and
 * summed afterwards in ascending frame order, so they do not depend on the
 * thread count (the reference's mutex-ordered sum does). */

typedef void (*pf_fn)(void* ctx, size_t i);
typedef struct { pf_fn fn; void* ctx; size_t n; size_t next; } pf_job;

static void* pf_worker(void* arg) {
    pf_job* j = (pf_job*)arg;
    for (;;) {
        size_t i = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (i >= j->n) break;
        j->fn(j->ctx, i);
    }
    return NULL;
}

static void parallel_for(int nthreads, size_t n, pf_fn fn, void* ctx) {
    if (nthreads <= 1 || n <= 1) {
        for (size_t i = 0; i < n; ++i) fn(ctx, i);
        return;
    }
    pf_job job = {fn, ctx, n, 0};
    int nt = nthreads;
    if ((size_t)nt > n) nt = (int)n;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nt);
    for (int t = 1; t < nt; ++t) pthread_create(&th[t], NULL, pf_worker, &job);
    pf_worker(&job);
    for (int t = 1; t < nt; ++t) pthread_join(th[t], NULL);
    free(th);
}

/* ------------------------------------------------------------------ */
/* core_support/minispline.cpp */

/* minispline.cpp:3-46.  The elimination is carried out in the same order as
 * the reference (forward on rows 0..n-3, backward on rows n-1..2, divide by
 * the diagonal) so the coefficients agree to the last bit. */
static int spline_set_points(spline1_t* s, const double* yv, size_t stride, size_t n) {
    if (n < 2) return 1;
    spline_free(s);
    s->n = n;
    s->y = (double*)malloc(n * sizeof(double));
    s->b = (double*)malloc(n * sizeof(double));
    s->c = (double*)malloc(n * sizeof(double));
    s->d = (double*)malloc(n * sizeof(double));
    double* A0 = (double*)calloc(n, sizeof(double)); /* sub-diagonal  A(i,0) */
    double* A1 = (double*)calloc(n, sizeof(double)); /* diagonal      A(i,1) */
    double* A2 = (double*)calloc(n, sizeof(double)); /* super-diagonal A(i,2) */
    double* y = s->y;
    double* c = s->c;
    for (size_t i = 0; i < n; ++i) y[i] = yv[i * stride];
    for (size_t i = 1; i + 1 < n; ++i) {
        A0[i] = 1.0 / 3.0;
        A1[i] = 2.0 / 3.0 * 2.0;
        A2[i] = 1.0 / 3.0;
        c[i] = y[i + 1] - 2 * y[i] + y[i - 1];
    }
    A1[0] = 2.0; A2[0] = 0.0; c[0] = 0.0;
    A1[n - 1] = 2.0; A0[n - 1] = 0.0; c[n - 1] = 0.0;
    for (size_t i = 0; i + 2 < n; ++i) {
        double k = 1. / A1[i] * A0[i + 1];
        A0[i + 1] -= A1[i] * k;
        A1[i + 1] -= A2[i] * k;
        c[i + 1] -= c[i] * k;
    }
    for (size_t i = n - 1; i > 1; --i) {
        double k = 1. / A1[i] * A2[i - 1];
        A1[i - 1] -= A0[i] * k;
        A2[i - 1] -= A1[i] * k;
        c[i - 1] -= c[i] * k;
    }
    for (size_t i = 0; i < n; ++i) c[i] /= A1[i];
    for (size_t i = 0; i + 1 < n; ++i) {
        s->d[i] = 1.0 / 3.0 * (c[i + 1] - c[i]);
        s->b[i] = (y[i + 1] - y[i]) - 1.0 / 3.0 * (2.0 * c[i] + c[i + 1]);
    }
    s->d[n - 1] = 0.0;
    s->b[n - 1] = 3.0 * s->d[n - 2] + 2.0 * c[n - 2] + s->b[n - 2];
    free(A0); free(A1); free(A2);
    return 0;
}

/* minispline.cpp:48-55, including the h = x - n extrapolation quirk for x >= n */
static inline double spline_eval1(const spline1_t* s, double x) {
    size_t n = s->n;
    double fi = floor(x);
    if (fi > (double)n) fi = (double)n;
    if (!(fi > 0.)) fi = 0.;
    size_t idx = (size_t)fi;
    double h = x - (double)idx;
    if (x < (double)idx) return (s->c[0] * h + s->b[0]) * h + s->y[0];
    if (x > (double)(n - 1)) return (s->c[n - 1] * h + s->b[n - 1]) * h + s->y[n - 1];
    return ((s->d[idx] * h + s->c[idx]) * h + s->b[idx]) * h + s->y[idx];
}

/* minispline.cpp:57-64 */
static inline double spline_deriv1(const spline1_t* s, double x) {
    size_t n = s->n;
    double fi = floor(x);
    if (fi > (double)n) fi = (double)n;
    if (!(fi > 0.)) fi = 0.;
    size_t idx = (size_t)fi;
    double h = x - (double)idx;
    if (x < 0) return 2.0 * s->c[0] * h + s->b[0];
    if (x > (double)(n - 1)) return 2.0 * s->c[n - 1] * h + s->b[n - 1];
    return (3.0 * s->d[idx] * h + 2.0 * s->c[idx]) * h + s->b[idx];
}

/* ndspline.cpp:21-27 / :29-35 */
void ora_spline_eval(const ora_problem* p, double x, double out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = spline_eval1(&p->sp[i], x);
}
void ora_spline_deriv(const ora_problem* p, double x, double out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = spline_deriv1(&p->sp[i], x);
}

/* ------------------------------------------------------------------ */
/* core_support/quat.cpp, [w,x,y,z] */

/* quat.cpp:33-38 */
static inline void quat_prod(const double p[4], const double q[4], double o[4]) {
    double r0 = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3];
    double r1 = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
    double r2 = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
    double r3 = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}
/* quat.cpp:40-43 */
static inline void quat_conj(const double q[4], double o[4]) {
    o[0] = q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = -q[3];
}
/* quat.cpp:45-47: q (0,p) conj(q), vector part */
static inline void quat_rotate_point(const double q[4], const double pt[3], double o[3]) {
    double pq[4] = {0, pt[0], pt[1], pt[2]}, qc[4], t[4], r[4];
    quat_conj(q, qc);
    quat_prod(pq, qc, t);
    quat_prod(q, t, r);
    o[0] = r[1]; o[1] = r[2]; o[2] = r[3];
}
/* quat.cpp:55-74.  acos is not clamped: a dot a hair above 1 gives NaN,
 * (NaN > 1e-9) is false, and the lerp branch is taken -- as in the reference. */
void ora_quat_slerp(const double p[4], const double qin[4], double t, double out[4]) {
    double q[4] = {qin[0], qin[1], qin[2], qin[3]};
    double dot = p[0] * q[0] + p[1] * q[1] + p[2] * q[2] + p[3] * q[3];
    if (dot < 0) {
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
        dot = p[0] * q[0] + p[1] * q[1] + p[2] * q[2] + p[3] * q[3];
    }
    double m1, m2;
    const double theta = acos(dot);
    if (theta > 1e-9) {
        const double st = sin(theta);
        m1 = sin((1 - t) * theta) / st;
        m2 = sin(t * theta) / st;
    } else {
        m1 = 1 - t;
        m2 = t;
    }
    for (int i = 0; i < 4; ++i) out[i] = m1 * p[i] + m2 * q[i];
}

/* arma::normalise on a 4-vector: divide by the 2-norm, by 1 if the norm is 0 */
static inline void normalise4(double q[4]) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n == 0) n = 1;
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static inline void cross3(const double a[3], const double b[3], double o[3]) {
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline double norm3(const double a[3]) {
    return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
}
/* inline_utils.hpp:5-11 */
static inline void safe_normalize3(double a[3]) {
    double n = norm3(a);
    if (n < 1e-12) return;
    a[0] /= n; a[1] /= n; a[2] /= n;
}
/* inline_utils.hpp:50 (std::clamp semantics, NaN passes through) */
static inline double clamp_k(double k) { return (k < 1e1) ? 1e1 : (1e3 < k) ? 1e3 : k; }

/* ------------------------------------------------------------------ */
/* setters */

static int finite_all(const double* v, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!isfinite(v[i])) return 0;
    return 1;
}

static int make_splines(ora_problem* p, const double* quats, size_t count) {
    /* ndspline.cpp:13-19: one spline per row of the 4 x count matrix */
    for (int c = 0; c < 4; ++c)
        if (spline_set_points(&p->sp[c], quats + c, 4, count)) return fail(p, "gyro: need >= 2 samples");
    p->gyro_n = count;
    return 0;
}

/* core_private.cpp:135-140 */
int ora_set_gyro_quaternions(ora_problem* p, const double* data, size_t count, double sample_rate,
                             double first_timestamp) {
    p->sample_rate = sample_rate;
    p->quats_start = first_timestamp;
    return make_splines(p, data, count);
}

/* core_private.cpp:142-190.  Integer arithmetic follows the reference's
 * types: the rate and the grid are computed in uint64, the first grid index
 * by a truncating division (the std::ceil at :152 acts on an integer). */
int ora_set_gyro_quaternions_ts(ora_problem* p, const int64_t* ts, const double* quats, size_t count) {
    static const uint64_t k_uhz_in_hz = 1000000ULL, k_us_in_sec = 1000000ULL;
    if (count < 2) return fail(p, "gyro: need >= 2 samples");
    uint64_t actual_sr_uhz = k_uhz_in_hz * k_us_in_sec * (uint64_t)count / (uint64_t)(ts[count - 1] - ts[0]);
    int rounded_sr_hz = (int)(round((double)actual_sr_uhz / 50. / (double)k_uhz_in_hz) * 50);
    if (rounded_sr_hz <= 0) return fail(p, "set-gyro-quaternions: non-positive sample rate");

    size_t cap = 1024, ng = 0;
    uint64_t* grid = (uint64_t*)malloc(cap * sizeof(uint64_t));
    for (int sample = (int)ceil((double)((uint64_t)(ts[0] * rounded_sr_hz) / k_us_in_sec));
         k_us_in_sec * (uint64_t)sample / (uint64_t)rounded_sr_hz < (uint64_t)ts[count - 1]; sample += 1) {
        if (ng == cap) { cap *= 2; grid = (uint64_t*)realloc(grid, cap * sizeof(uint64_t)); }
        grid[ng++] = k_us_in_sec * (uint64_t)sample / (uint64_t)rounded_sr_hz;
    }
    for (size_t i = 1; i < count; ++i) {
        if (ts[i - 1] > ts[i]) {
            free(grid);
            snprintf(p->err, sizeof p->err,
                     "set-gyro-quaternions:  timestamps out of order at pos %zu (%lld > %lld)", i,
                     (long long)ts[i - 1], (long long)ts[i]);
            return 1;
        }
    }
    if (ng < 2) { free(grid); return fail(p, "gyro: resampled grid has < 2 points"); }
    double* nq = (double*)malloc(ng * 4 * sizeof(double));
    for (size_t i = 0; i < ng; ++i) {
        uint64_t t = grid[i];
        /* std::lower_bound over int64 elements compared with a uint64 value */
        size_t lo = 0, hi = count;
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            if ((uint64_t)ts[mid] < t) lo = mid + 1; else hi = mid;
        }
        size_t idx = lo;
        if (idx > 0) {
            double tt = 1. * (double)(t - (uint64_t)ts[idx - 1]) / (double)(ts[idx] - ts[idx - 1]);
            ora_quat_slerp(quats + 4 * (idx - 1), quats + 4 * idx, tt, nq + 4 * i);
        } else {
            memcpy(nq + 4 * i, quats + 4 * idx, 4 * sizeof(double));
        }
        if (!finite_all(nq + 4 * i, 4)) {
            free(grid); free(nq);
            return fail(p, "set-gyro-quaternions: non-finite sample after interpolation");
        }
    }
    p->sample_rate = 1. * rounded_sr_hz;
    p->quats_start = 1. * (double)grid[0] / (double)k_us_in_sec;
    int rc = make_splines(p, nq, ng);
    free(grid); free(nq);
    return rc;
}

static frame_t* find_frame(const ora_problem* p, int64_t id) {
    size_t lo = 0, hi = p->nframes;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (p->frames[mid].id < id) lo = mid + 1; else hi = mid;
    }
    if (lo < p->nframes && p->frames[lo].id == id) return &p->frames[lo];
    return NULL;
}

size_t ora_frame_tracks(const ora_problem* p, int64_t frame) {
    frame_t* f = find_frame(p, frame);
    return f ? f->n : 0;
}

static double* dupd(const double* v, size_t n) {
    double* r = (double*)malloc((n ? n : 1) * sizeof(double));
    memcpy(r, v, n * sizeof(double));
    return r;
}

/* core_private.cpp:192-203; the map becomes a sorted array, data is copied */
int ora_set_track_result(ora_problem* p, int64_t frame, const double* ts_a, const double* ts_b,
                         const double* rays_a, const double* rays_b, size_t count) {
    if (!finite_all(rays_a, 3 * count)) return fail(p, "set-track-result: non-finite numbers in rays_a");
    if (!finite_all(rays_b, 3 * count)) return fail(p, "set-track-result: non-finite numbers in rays_b");
    if (!finite_all(ts_a, count)) return fail(p, "set-track-result: non-finite numbers in ts_a");
    if (!finite_all(ts_b, count)) return fail(p, "set-track-result: non-finite numbers in ts_b");
    frame_t* f = find_frame(p, frame);
    if (f) {
        frame_free(f);
    } else {
        if (p->nframes == p->cap) {
            p->cap = p->cap ? p->cap * 2 : 64;
            p->frames = (frame_t*)realloc(p->frames, p->cap * sizeof(frame_t));
        }
        size_t pos = p->nframes;
        while (pos > 0 && p->frames[pos - 1].id > frame) { p->frames[pos] = p->frames[pos - 1]; --pos; }
        f = &p->frames[pos];
        p->nframes++;
        memset(f, 0, sizeof *f);
        f->id = frame;
        f->k = 1e3; /* core_private.hpp:35 */
    }
    f->n = count;
    f->ts_a = dupd(ts_a, count);
    f->ts_b = dupd(ts_b, count);
    f->rays_a = dupd(rays_a, 3 * count);
    f->rays_b = dupd(rays_b, 3 * count);
    return 0;
}

/* ------------------------------------------------------------------ */
/* core_private.cpp:15-32: the residual matrix P (row-major N x 3) */

static void compute_problem(const ora_problem* p, const frame_t* f, double delay, double* P) {
    for (size_t i = 0; i < f->n; ++i) {
        double at = (f->ts_a[i] - p->quats_start + delay) * p->sample_rate; /* :19 */
        double bt = (f->ts_b[i] - p->quats_start + delay) * p->sample_rate; /* :20 */
        double a[4], b[4], ac[4], bc[4], ar[3], br[3];
        for (int c = 0; c < 4; ++c) { a[c] = spline_eval1(&p->sp[c], at); b[c] = spline_eval1(&p->sp[c], bt); }
        normalise4(a); normalise4(b); /* :24-25 */
        quat_conj(a, ac); quat_conj(b, bc);
        quat_rotate_point(ac, f->rays_a + 3 * i, ar); /* :26 */
        quat_rotate_point(bc, f->rays_b + 3 * i, br); /* :27 */
        cross3(ar, br, P + 3 * i);                    /* :28 */
    }
}

int ora_compute_problem(const ora_problem* p, int64_t frame, double delay, double* P) {
    frame_t* f = find_frame(p, frame);
    if (!f) return 1;
    compute_problem(p, f, delay, P);
    return 0;
}

/* analytic dP/d(delay): with ar = conj(qa) a qa and qa' = 0.5 qa W (W = body
 * rate), d(ar)/dx = ar x W where W = vec(2 conj(S) S' / |S|^2) is exactly
 * ndspline::rderiv (ndspline.cpp:45-49); x = (ts - start + delay) * fs. */
static void compute_problem_dd(const ora_problem* p, const frame_t* f, double delay, double* P, double* dP) {
    for (size_t i = 0; i < f->n; ++i) {
        double xt[2] = {(f->ts_a[i] - p->quats_start + delay) * p->sample_rate,
                        (f->ts_b[i] - p->quats_start + delay) * p->sample_rate};
        const double* ray[2] = {f->rays_a + 3 * i, f->rays_b + 3 * i};
        double r[2][3], dr[2][3];
        for (int s = 0; s < 2; ++s) {
            double q[4], dq[4], qc[4], w[4], qn[4];
            for (int c = 0; c < 4; ++c) { q[c] = spline_eval1(&p->sp[c], xt[s]); dq[c] = spline_deriv1(&p->sp[c], xt[s]); }
            double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
            quat_conj(q, qc);
            quat_prod(qc, dq, w);
            double W[3] = {2 * w[1] / n2, 2 * w[2] / n2, 2 * w[3] / n2};
            memcpy(qn, q, sizeof qn);
            normalise4(qn);
            quat_conj(qn, qc);
            quat_rotate_point(qc, ray[s], r[s]);
            cross3(r[s], W, dr[s]);
            for (int c = 0; c < 3; ++c) dr[s][c] *= p->sample_rate;
        }
        double t1[3], t2[3];
        cross3(r[0], r[1], P + 3 * i);
        cross3(dr[0], r[1], t1);
        cross3(r[0], dr[1], t2);
        for (int c = 0; c < 3; ++c) dP[3 * i + c] = t1[c] + t2[c];
    }
}

/* ------------------------------------------------------------------ */
/* sampler (replaces inline_utils.hpp:13-17, see header) */

static inline uint64_t sm64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

void ora_sample_pair(uint64_t seed, int64_t frame, uint32_t stream, uint32_t h, uint32_t n,
                     uint32_t* i0, uint32_t* i1) {
    uint64_t z = sm64(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)frame);
    z = sm64(z ^ (((uint64_t)stream << 32) | (uint64_t)h));
    uint32_t lo = (uint32_t)z, hi = (uint32_t)(z >> 32);
    uint32_t a = (uint32_t)(((uint64_t)lo * n) >> 32);
    uint32_t j = (uint32_t)(((uint64_t)hi * (n - 1)) >> 32); /* uniform over the other n-1 rows */
    *i0 = a;
    *i1 = j + (j >= a); /* core_private.cpp:42-43: second index drawn until it differs */
}

/* ------------------------------------------------------------------ */
/* sort: stands in for std::sort (core_private.cpp:51) */

static void insertion_sort(double* a, size_t n) {
    for (size_t i = 1; i < n; ++i) {
        double v = a[i];
        size_t j = i;
        while (j > 0 && v < a[j - 1]) { a[j] = a[j - 1]; --j; }
        a[j] = v;
    }
}
static void heap_sift(double* a, size_t start, size_t end) {
    size_t root = start;
    while (2 * root + 1 < end) {
        size_t ch = 2 * root + 1;
        if (ch + 1 < end && a[ch] < a[ch + 1]) ++ch;
        if (a[root] < a[ch]) { double t = a[root]; a[root] = a[ch]; a[ch] = t; root = ch; } else return;
    }
}
static void heap_sort(double* a, size_t n) {
    for (size_t s = n / 2; s-- > 0;) heap_sift(a, s, n);
    for (size_t e = n; e-- > 1;) { double t = a[0]; a[0] = a[e]; a[e] = t; heap_sift(a, 0, e); }
}
static void intro_sort(double* a, size_t n, int depth) {
    while (n > 24) {
        if (depth-- == 0) { heap_sort(a, n); return; }
        double x = a[0], y = a[n / 2], z = a[n - 1];
        double piv = (x < y) ? ((y < z) ? y : (x < z ? z : x)) : ((x < z) ? x : (y < z ? z : y));
        size_t i = 0, j = n - 1;
        for (;;) {
            while (a[i] < piv) ++i;
            while (piv < a[j]) --j;
            if (i >= j) break;
            double t = a[i]; a[i] = a[j]; a[j] = t;
            ++i; --j;
        }
        /* [0, j] and [j+1, n) */
        size_t nl = j + 1;
        if (nl < n - nl) { intro_sort(a, nl, depth); a += nl; n -= nl; }
        else { intro_sort(a + nl, n - nl, depth); n = nl; }
    }
    insertion_sort(a, n);
}
static void sort_doubles(double* a, size_t n) {
    int depth = 0;
    for (size_t m = n; m > 1; m >>= 1) depth += 2;
    intro_sort(a, n, depth);
}

/* ------------------------------------------------------------------ */
/* core_private.cpp:34-59: LMedS-style motion guess */

static void guess_motion(const double* P, size_t n, int max_iters, uint64_t seed, int64_t frame,
                         uint32_t stream, double* nP, double* r2, double M[3], int* best_h,
                         double* best_med) {
    for (size_t i = 0; i < n; ++i) { /* :35-36 */
        nP[3 * i] = P[3 * i]; nP[3 * i + 1] = P[3 * i + 1]; nP[3 * i + 2] = P[3 * i + 2];
        safe_normalize3(nP + 3 * i);
    }
    double best[3] = {0, 0, 0};
    double least = INFINITY;
    int bh = -1;
    for (int it = 0; it < max_iters; ++it) {
        uint32_t i0, i1;
        ora_sample_pair(seed, frame, stream, (uint32_t)it, (uint32_t)n, &i0, &i1);
        double v[3];
        cross3(P + 3 * i0, P + 3 * i1, v); /* :45-46, un-normalised rows */
        safe_normalize3(v);
        for (size_t i = 0; i < n; ++i) { /* :48-49 */
            double r = nP[3 * i] * v[0] + nP[3 * i + 1] * v[1] + nP[3 * i + 2] * v[2];
            r2[i] = r * r;
        }
        sort_doubles(r2, n);  /* :51 */
        double med = r2[n / 4]; /* :52 lower quartile */
        if (med < least) {      /* :53-56 strict */
            least = med;
            best[0] = v[0]; best[1] = v[1]; best[2] = v[2];
            bh = it;
        }
    }
    M[0] = best[0]; M[1] = best[1]; M[2] = best[2];
    if (best_h) *best_h = bh;
    if (best_med) *best_med = least;
}

int ora_guess_motion(const ora_problem* p, int64_t frame, double delay, int max_iters,
                     uint32_t stream, double M[3], int* best_h, double* best_med) {
    frame_t* f = find_frame(p, frame);
    if (!f || f->n < 2) return 1;
    double* buf = (double*)malloc(f->n * 7 * sizeof(double));
    compute_problem(p, f, delay, buf);
    guess_motion(buf, f->n, max_iters, p->seed, f->id, stream, buf + 3 * f->n, buf + 6 * f->n, M, best_h, best_med);
    free(buf);
    return 0;
}

/* core_private.cpp:75-85 (and :350-356): one frame's PreSync term.
 * returns 0 ok, else the index (1..4) of the reference's panic check that fired */
static int frame_presync_cost(const ora_problem* p, const frame_t* f, double delay, uint32_t stream,
                              double* buf, double* cost, int* best_h) {
    size_t n = f->n;
    double* P = buf;
    compute_problem(p, f, delay, P);
    int bad = 0;
    if (!finite_all(P, 3 * n)) bad = 1;
    double M[3];
    guess_motion(P, n, 20, p->seed, f->id, stream, buf + 3 * n, buf + 6 * n, M, best_h, NULL);
    if (!bad && !finite_all(M, 3)) bad = 2;
    double ss = 0;
    double* pm = buf + 6 * n;
    for (size_t i = 0; i < n; ++i) {
        pm[i] = P[3 * i] * M[0] + P[3 * i + 1] * M[1] + P[3 * i + 2] * M[2];
        ss += pm[i] * pm[i];
    }
    double k = clamp_k(1 / sqrt(ss) * 1e2); /* :79 */
    double scale = k / norm3(M);            /* :80 */
    double acc = 0;
    for (size_t i = 0; i < n; ++i) {
        double r = pm[i] * scale;
        if (!bad && !isfinite(r)) bad = 3;
        double rho = log1p(r * r); /* :82 */
        if (!bad && !isfinite(rho)) bad = 4;
        acc += sqrt(rho);
    }
    *cost = sqrt(acc); /* :85 */
    return bad;
}

int ora_frame_presync_cost(const ora_problem* p, int64_t frame, double delay, uint32_t stream,
                           double* cost, int* best_h) {
    frame_t* f = find_frame(p, frame);
    if (!f || f->n < 2) return -1;
    double* buf = (double*)malloc(f->n * 7 * sizeof(double));
    int bad = frame_presync_cost(p, f, delay, stream, buf, cost, best_h);
    free(buf);
    return bad;
}

static const char* k_presync_panics[5] = {
    "", "pre-sync: non-finite numbers in P", "pre-sync: non-finite numbers in M",
    "pre-sync: non-finite r", "pre-sync: non-finite rho"};

typedef struct {
    const ora_problem* p;
    const size_t* sel;
    double delay;
    uint32_t stream;
    double* costs; /* per selected frame */
    int* best_h;
    int* bad;
} presync_ctx;

static void presync_frame_fn(void* vctx, size_t i) {
    presync_ctx* c = (presync_ctx*)vctx;
    const frame_t* f = &c->p->frames[c->sel[i]];
    double* buf = (double*)malloc(f->n * 7 * sizeof(double));
    int bh = -1;
    c->bad[i] = frame_presync_cost(c->p, f, c->delay, c->stream, buf, &c->costs[i], &bh);
    if (c->best_h) c->best_h[i] = bh;
    free(buf);
}

static size_t select_frames(ora_problem* p, int64_t begin, int64_t end_excl) {
    free(p->sel);
    p->sel = (size_t*)malloc((p->nframes ? p->nframes : 1) * sizeof(size_t));
    p->nsel = 0;
    for (size_t i = 0; i < p->nframes; ++i)
        if (p->frames[i].id >= begin && p->frames[i].id < end_excl) p->sel[p->nsel++] = i;
    return p->nsel;
}

static int check_tracks(ora_problem* p) {
    if (p->gyro_n < 2) return fail(p, "no gyro data");
    for (size_t i = 0; i < p->nsel; ++i)
        if (p->frames[p->sel[i]].n < 2) return fail(p, "frame with < 2 tracks (reference would not terminate)");
    return 0;
}

/* core_private.cpp:61-90 with the per-frame matrix exposed */
int ora_presync_curve(ora_problem* p, double initial_delay, int64_t frame_begin,
                      int64_t frame_end, double search_step, double search_radius,
                      double* delays, double* costs, int cap, int* n_out, double* frame_costs,
                      int* best_h) {
    select_frames(p, frame_begin, frame_end); /* :65-68, end exclusive */
    if (check_tracks(p)) return 1;
    size_t nf = p->nsel;
    double* fc = (double*)malloc((nf ? nf : 1) * sizeof(double));
    int* bad = (int*)calloc(nf ? nf : 1, sizeof(int));
    int* bh = best_h ? (int*)malloc((nf ? nf : 1) * sizeof(int)) : NULL;
    int n = 0, rc = 0;
    /* :69-70: the candidate list is whatever this double loop produces */
    for (double delay = initial_delay - search_radius; delay < initial_delay + search_radius; delay += search_step) {
        if (n >= cap) { rc = fail(p, "presync: candidate capacity exceeded"); break; }
        presync_ctx ctx = {p, p->sel, delay, (uint32_t)n, fc, bh, bad};
        parallel_for(p->nthreads, nf, presync_frame_fn, &ctx);
        double cost = 0;
        for (size_t i = 0; i < nf; ++i) {
            if (bad[i] > 0 && !rc) rc = fail(p, k_presync_panics[bad[i]]);
            cost += fc[i]; /* :84-85, fixed ascending-frame order */
        }
        if (frame_costs) memcpy(frame_costs + (size_t)n * nf, fc, nf * sizeof(double));
        if (best_h) memcpy(best_h + (size_t)n * nf, bh, nf * sizeof(int));
        delays[n] = delay;
        costs[n] = cost;
        ++n;
        if (rc) break;
    }
    *n_out = n;
    free(fc); free(bad); free(bh);
    return rc;
}

/* core_private.cpp:205-209 -> :61-90; result = *std::min_element over pair(cost, delay) */
int ora_presync(ora_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end,
                double search_step, double search_radius, double* cost, double* delay) {
    double span = 2 * search_radius / search_step;
    if (!(span < 5e7)) return fail(p, "presync: too many candidates");
    int cap = (int)span + 8;
    double* d = (double*)malloc((size_t)cap * sizeof(double));
    double* c = (double*)malloc((size_t)cap * sizeof(double));
    int n = 0;
    int rc = ora_presync_curve(p, initial_delay, frame_begin, frame_end, search_step, search_radius, d, c, cap, &n, NULL, NULL);
    if (!rc) {
        if (n == 0) rc = fail(p, "presync: empty candidate list");
        else {
            int best = 0;
            for (int i = 1; i < n; ++i)
                if (c[i] < c[best] || (!(c[best] < c[i]) && d[i] < d[best])) best = i;
            *cost = c[best];
            *delay = d[best];
        }
    }
    free(d); free(c);
    return rc;
}

/* core_private.cpp:336-361: point_count delays including both ends, no panics */
int ora_debug_presync(ora_problem* p, double initial_delay, int64_t frame_begin,
                      int64_t frame_end, double search_radius, double* delays, double* costs,
                      int point_count) {
    select_frames(p, frame_begin, frame_end);
    if (check_tracks(p)) return 1;
    size_t nf = p->nsel;
    double* fc = (double*)malloc((nf ? nf : 1) * sizeof(double));
    int* bad = (int*)calloc(nf ? nf : 1, sizeof(int));
    for (int i = 0; i < point_count; ++i) {
        double delay = initial_delay - search_radius + 2 * search_radius * i / (point_count - 1); /* :345 */
        presync_ctx ctx = {p, p->sel, delay, ORA_STREAM_DEBUG + (uint32_t)i, fc, NULL, bad};
        parallel_for(p->nthreads, nf, presync_frame_fn, &ctx);
        double cost = 0;
        for (size_t j = 0; j < nf; ++j) cost += fc[j];
        delays[i] = delay;
        costs[i] = cost;
    }
    free(fc); free(bad);
    return 0;
}

/* ------------------------------------------------------------------ */
/* losses: core_private.cpp:92-123 */

/* :117-123 given P */
static double loss_from_P(const double* P, size_t n, const double M[3], double k) {
    double scale = k / norm3(M);
    double acc = 0;
    for (size_t i = 0; i < n; ++i) {
        double r = (P[3 * i] * M[0] + P[3 * i + 1] * M[1] + P[3 * i + 2] * M[2]) * scale;
        acc += log1p(r * r);
    }
    return acc;
}

/* The thesis' no-translation variant (thesis-text.pdf section 2.11 eq. (12), printed p.26; no code in the
 * reference snapshot): L = sum_j log(1 + (k |h_j|)^2) with h_j the j-th row of P (core_private.cpp:28). */
static double loss_simple_from_P(const double* P, size_t n, double k) {
    double acc = 0;
    for (size_t i = 0; i < n; ++i) {
        double r2 = (P[3 * i] * P[3 * i] + P[3 * i + 1] * P[3 * i + 1] + P[3 * i + 2] * P[3 * i + 2]) * k * k;
        acc += log1p(r2);
    }
    return acc;
}

/* :99-110,114 given P: loss through the v1..v8 chain and dL/dM in closed form.
 * With pm = P M, s = |M|^2 / k^2, u = pm^2 / s:
 *   dL/dM = sum_i 1/(1+u_i) * [ (2 pm_i / s) P_i - (pm_i^2 / s^2) (2 M / k^2) ]
 * which is j8 j7 (j6a j2 j1 + j6b j5 j4 j3) with the diagonal matrices kept implicit. */
static double loss_grad_from_P(const double* P, size_t n, const double M[3], double k, double g[3]) {
    double s = (M[0] * M[0] + M[1] * M[1] + M[2] * M[2]) / (k * k); /* v5 */
    double acc = 0, g0 = 0, g1 = 0, g2 = 0, gs = 0;
    for (size_t i = 0; i < n; ++i) {
        double pm = P[3 * i] * M[0] + P[3 * i + 1] * M[1] + P[3 * i + 2] * M[2]; /* v1 */
        double v2 = pm * pm;
        double u = v2 / s; /* v6 */
        acc += log1p(u);   /* v7, v8 */
        double w = 1. / (1. + u); /* j7 */
        double a = w * (2. * pm / s);
        g0 += a * P[3 * i]; g1 += a * P[3 * i + 1]; g2 += a * P[3 * i + 2];
        gs += w * (v2 / (s * s));
    }
    double t = gs * 2. / (k * k);
    g[0] = g0 - t * M[0]; g[1] = g1 - t * M[1]; g[2] = g2 - t * M[2];
    return acc;
}

static const double kNumericDiffStep = 1e-6; /* core_private.hpp:38 */

int ora_loss(const ora_problem* p, int64_t frame, double delay, const double M[3], double k,
             double* loss, double* dd_numeric, double* dd_analytic, double gM[3]) {
    frame_t* f = find_frame(p, frame);
    if (!f) return 1;
    size_t n = f->n;
    double* P = (double*)malloc(6 * n * sizeof(double));
    double* dP = P + 3 * n;
    double g[3];
    compute_problem_dd(p, f, delay, P, dP);
    double L = loss_grad_from_P(P, n, M, k, g);
    if (loss) *loss = L;
    if (gM) { gM[0] = g[0]; gM[1] = g[1]; gM[2] = g[2]; }
    if (dd_analytic) {
        double s = (M[0] * M[0] + M[1] * M[1] + M[2] * M[2]) / (k * k);
        double acc = 0;
        for (size_t i = 0; i < n; ++i) {
            double pm = P[3 * i] * M[0] + P[3 * i + 1] * M[1] + P[3 * i + 2] * M[2];
            double dpm = dP[3 * i] * M[0] + dP[3 * i + 1] * M[1] + dP[3 * i + 2] * M[2];
            double u = pm * pm / s;
            acc += 1. / (1. + u) * (2. * pm / s) * dpm;
        }
        *dd_analytic = acc;
    }
    if (dd_numeric) { /* :96-97,112 */
        compute_problem(p, f, delay - kNumericDiffStep, P);
        double ll = loss_from_P(P, n, M, k);
        compute_problem(p, f, delay + kNumericDiffStep, P);
        double lr = loss_from_P(P, n, M, k);
        *dd_numeric = (lr - ll) / 2 / kNumericDiffStep;
    }
    free(P);
    return 0;
}

/* ------------------------------------------------------------------ */
/* ens::L_BFGS restated.  THIRD-PARTY, NOT UNDER /root/reference: ensmallen is
 * an unpinned vcpkg dependency (vcpkg.json:14); call site core_private.cpp:264-294
 * sets MaxIterations = 200, MinGradientNorm = 1e-4 and leaves the defaults
 * numBasis 10, armijo 1e-4, wolfe 0.9, factr 1e-15, maxLineSearchTrials 50,
 * minStep 1e-20, maxStep 1e20.  This follows the published algorithm of
 * ensmallen 2.x (lbfgs_impl.hpp): two-loop recursion, scaling 1/|g| on the
 * first iteration and s.y / y.y afterwards, backtracking/expanding line
 * search (x0.5 on Armijo failure or strong-Wolfe overshoot, x2.1 on curvature
 * failure) that moves to the best-objective step.
 * When the best step is not the last one evaluated, the published LineSearch moves the ITERATE to
 * the best step but leaves functionValue and gradient as the LAST trial computed them; the
 * progress test, UpdateBasisSet (y = gradient - oldGradient) and the next search direction then
 * use that value and gradient.  That is the default here (lbfgs_reeval = 0).  ora_set_lbfgs_reeval(1)
 * selects the self-consistent variant instead (one more evaluation at the best step), which round
 * 1 used; tests/measure/ counts how often the two differ and what it does to Sync's delay. */

typedef struct {
    const ora_problem* p;
    const frame_t* f;
    double delay, k;
    double* P;    /* cached residual matrix (lean schedule) */
    double* Ptmp; /* scratch for the faithful schedule */
    int evals;
} motion_obj;

static double motion_eval(motion_obj* o, const double M[3], double g[3]) {
    o->evals++;
    if (o->p->faithful) {
        /* the reference recomputes P on every evaluation and also pays for the
         * discarded central difference in delay (core_private.cpp:94-97) */
        compute_problem(o->p, o->f, o->delay - kNumericDiffStep, o->Ptmp);
        volatile double sink = loss_from_P(o->Ptmp, o->f->n, M, o->k);
        compute_problem(o->p, o->f, o->delay + kNumericDiffStep, o->Ptmp);
        sink = loss_from_P(o->Ptmp, o->f->n, M, o->k);
        (void)sink;
        compute_problem(o->p, o->f, o->delay, o->P);
    }
    return loss_grad_from_P(o->P, o->f->n, M, o->k, g);
}

#define LB_NB 10
static inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static double lbfgs_minimise(motion_obj* o, double x[3], int* iters_out) {
    const int maxIterations = 200;       /* core_private.cpp:265 */
    const double minGradientNorm = 1e-4; /* core_private.cpp:266 */
    const double armijo = 1e-4, wolfe = 0.9, factr = 1e-15, minStep = 1e-20, maxStep = 1e20;
    const int maxLineSearchTrials = 50;
    double S[LB_NB][3], Y[LB_NB][3];
    double g[3], oldx[3], oldg[3], dir[3];
    double fval = motion_eval(o, x, g);
    int it = 0;
    for (; it != maxIterations; ++it) {
        double prev = fval;
        if (sqrt(dot3(g, g)) < minGradientNorm) break;
        if (isnan(fval)) break;
        /* ChooseScalingFactor */
        double scale;
        if (it > 0) {
            int pp = (it - 1) % LB_NB;
            double yy = dot3(Y[pp], Y[pp]);
            scale = dot3(S[pp], Y[pp]) / ((yy >= 1e-10) ? yy : 1.0);
        } else {
            double gn = sqrt(dot3(g, g));
            scale = (gn >= 1e-5) ? 1.0 / gn : 1.0;
        }
        if (scale == 0.0 || isnan(scale)) break;
        /* SearchDirection: two-loop recursion */
        double rho[LB_NB], alpha[LB_NB];
        dir[0] = g[0]; dir[1] = g[1]; dir[2] = g[2];
        int limit = (LB_NB > it) ? 0 : (it - LB_NB);
        for (int i = it; i != limit; --i) {
            int tp = (i + (LB_NB - 1)) % LB_NB;
            rho[it - i] = 1.0 / dot3(Y[tp], S[tp]);
            alpha[it - i] = rho[it - i] * dot3(S[tp], dir);
            for (int c = 0; c < 3; ++c) dir[c] -= alpha[it - i] * Y[tp][c];
        }
        for (int c = 0; c < 3; ++c) dir[c] *= scale;
        for (int i = limit; i < it; ++i) {
            int tp = i % LB_NB;
            double beta = rho[it - i - 1] * dot3(Y[tp], dir);
            for (int c = 0; c < 3; ++c) dir[c] += (alpha[it - i - 1] - beta) * S[tp][c];
        }
        for (int c = 0; c < 3; ++c) dir[c] = -dir[c];
        for (int c = 0; c < 3; ++c) { oldx[c] = x[c]; oldg[c] = g[c]; }
        /* LineSearch */
        double dg0 = dot3(g, dir);
        if (dg0 > 0.0) break; /* not a descent direction: failure */
        double f0 = fval, lin = armijo * dg0;
        double step = 1.0, bestStep = 1.0, bestObj = 1.79769313486231570e308, lastStep = 1.0;
        int trials = 0;
        for (;;) {
            double xn[3] = {x[0] + step * dir[0], x[1] + step * dir[1], x[2] + step * dir[2]};
            fval = motion_eval(o, xn, g);
            lastStep = step;
            if (fval < bestObj) { bestStep = step; bestObj = fval; }
            ++trials;
            double width;
            if (fval > f0 + step * lin) {
                width = 0.5;
            } else {
                double dg = dot3(g, dir);
                if (dg < wolfe * dg0) width = 2.1;
                else if (dg > -wolfe * dg0) width = 0.5;
                else break;
            }
            if (step < minStep || step > maxStep || trials >= maxLineSearchTrials) break;
            step *= width;
        }
        for (int c = 0; c < 3; ++c) x[c] += bestStep * dir[c];
        if (bestStep != lastStep) {
            __atomic_fetch_add(&((ora_problem*)o->p)->lbfgs_best_not_last, 1L, __ATOMIC_RELAXED);
            if (o->p->lbfgs_reeval) fval = motion_eval(o, x, g); /* variant: keep (x, f, g) consistent */
            /* default: f and g stay those of the last trial, as the published LineSearch leaves them */
        }
        if (bestStep == 0.0) break;
        double denom = fmax(fmax(fabs(prev), fabs(fval)), 1.0);
        if ((prev - fval) / denom <= factr) break;
        /* UpdateBasisSet */
        int op = it % LB_NB;
        for (int c = 0; c < 3; ++c) { S[op][c] = x[c] - oldx[c]; Y[op][c] = g[c] - oldg[c]; }
    }
    if (iters_out) *iters_out = it;
    return fval;
}

int ora_lbfgs_motion(const ora_problem* p, int64_t frame, double delay, double M[3], double k,
                     int* iters, int* evals, double* final_loss) {
    frame_t* f = find_frame(p, frame);
    if (!f) return 1;
    motion_obj o = {p, f, delay, k, NULL, NULL, 0};
    o.P = (double*)malloc(6 * f->n * sizeof(double));
    o.Ptmp = o.P + 3 * f->n;
    compute_problem(p, f, delay, o.P);
    double fv = lbfgs_minimise(&o, M, iters);
    if (evals) *evals = o.evals;
    if (final_loss) *final_loss = fv;
    free(o.P);
    return 0;
}

/* ------------------------------------------------------------------ */
/* Sync: core_private.cpp:211-334 */

typedef struct {
    ora_problem* p;
    double delay;
    uint32_t stream;
    double* out_loss;
    double* out_grad;
    int with_grad;
    int simplified; /* thesis section 2.11: no motion estimate, loss_simple_from_P */
} sync_ctx;

/* :218-223: GuessMotion (200 hypotheses) then GuessK at the initial delay */
static void sync_init_fn(void* vctx, size_t i) {
    sync_ctx* c = (sync_ctx*)vctx;
    frame_t* f = &c->p->frames[c->p->sel[i]];
    size_t n = f->n;
    double* buf = (double*)malloc(7 * n * sizeof(double));
    compute_problem(c->p, f, c->delay, buf);
    if (c->simplified) {
        /* GuessK's rule (:130-133, inline_utils.hpp:50) with |h_j| in place of h_j . v: k = gamma / |x|_2 */
        double ss = 0;
        for (size_t j = 0; j < n; ++j) ss += buf[3 * j] * buf[3 * j] + buf[3 * j + 1] * buf[3 * j + 1] + buf[3 * j + 2] * buf[3 * j + 2];
        f->k = clamp_k(1 / sqrt(ss) * 1e2);
        free(buf);
        return;
    }
    int bh = -1;
    if (c->p->init_override) { /* test hook: this hypothesis instead of the search's winner */
        bh = c->p->init_override[i];
        f->M[0] = f->M[1] = f->M[2] = 0;
        if (bh >= 0) {
            uint32_t i0, i1;
            ora_sample_pair(c->p->seed, f->id, c->stream, (uint32_t)bh, (uint32_t)n, &i0, &i1);
            cross3(buf + 3 * i0, buf + 3 * i1, f->M); /* :45-46 */
            safe_normalize3(f->M);
        }
    } else {
        guess_motion(buf, n, 200, c->p->seed, f->id, c->stream, buf + 3 * n, buf + 6 * n, f->M, &bh, NULL); /* :125-128 */
    }
    if (c->p->last_init) c->p->last_init[i] = bh;
    if (c->p->faithful) compute_problem(c->p, f, c->delay, buf); /* :131 recomputes P */
    double ss = 0;
    for (size_t j = 0; j < n; ++j) {
        double pm = buf[3 * j] * f->M[0] + buf[3 * j + 1] * f->M[1] + buf[3 * j + 2] * f->M[2];
        ss += pm * pm;
    }
    f->k = clamp_k(1 / sqrt(ss) * 1e2); /* :130-133 */
    free(buf);
}

/* :231-238 (with_grad) and :245-250: per-frame loss [+ central-difference d/d-delay] */
static void sync_loss_fn(void* vctx, size_t i) {
    sync_ctx* c = (sync_ctx*)vctx;
    frame_t* f = &c->p->frames[c->p->sel[i]];
    size_t n = f->n;
    double* P = (double*)malloc(3 * n * sizeof(double));
    if (c->simplified) { /* the same schedule (central difference of :96-97,112) on the simplified loss */
        if (c->with_grad) {
            compute_problem(c->p, f, c->delay - kNumericDiffStep, P);
            double ll = loss_simple_from_P(P, n, f->k);
            compute_problem(c->p, f, c->delay + kNumericDiffStep, P);
            double lr = loss_simple_from_P(P, n, f->k);
            c->out_grad[i] = (lr - ll) / 2 / kNumericDiffStep;
        }
        compute_problem(c->p, f, c->delay, P);
        c->out_loss[i] = loss_simple_from_P(P, n, f->k);
        free(P);
        return;
    }
    if (c->with_grad) {
        compute_problem(c->p, f, c->delay - kNumericDiffStep, P);
        double ll = loss_from_P(P, n, f->M, f->k);
        compute_problem(c->p, f, c->delay + kNumericDiffStep, P);
        double lr = loss_from_P(P, n, f->M, f->k);
        compute_problem(c->p, f, c->delay, P);
        double g[3];
        c->out_loss[i] = loss_grad_from_P(P, n, f->M, f->k, g);
        c->out_grad[i] = (lr - ll) / 2 / kNumericDiffStep;
    } else {
        compute_problem(c->p, f, c->delay, P);
        c->out_loss[i] = loss_from_P(P, n, f->M, f->k);
    }
    free(P);
}

/* :262-296: per-frame L-BFGS on the motion vector at fixed delay */
static void sync_motion_fn(void* vctx, size_t i) {
    sync_ctx* c = (sync_ctx*)vctx;
    frame_t* f = &c->p->frames[c->p->sel[i]];
    motion_obj o = {c->p, f, c->delay, f->k, NULL, NULL, 0};
    o.P = (double*)malloc(6 * f->n * sizeof(double));
    o.Ptmp = o.P + 3 * f->n;
    compute_problem(c->p, f, c->delay, o.P);
    lbfgs_minimise(&o, f->M, NULL);
    free(o.P);
}

static double sum_in_order(const double* v, size_t n) {
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += v[i];
    return s;
}

static int sync_trace_impl(ora_problem* p, int simplified, double initial_delay, int64_t frame_begin,
                           int64_t frame_end, double search_center, double search_radius, double* cost,
                           double* delay_out, double* trace, int cap, int* n_rows) {
    /* :218-219 end INCLUSIVE */
    select_frames(p, frame_begin, frame_end == INT64_MAX ? INT64_MAX : frame_end + 1);
    if (check_tracks(p)) return 1;
    size_t nf = p->nsel;
    double* fl = (double*)malloc((nf ? nf : 1) * sizeof(double));
    double* fg = (double*)malloc((nf ? nf : 1) * sizeof(double));
    double d = initial_delay;
    sync_ctx ctx = {p, d, ORA_STREAM_SYNC_INIT + p->sync_calls, fl, fg, 0, simplified};
    if (!simplified) p->sync_calls++;
    if (p->init_override && p->n_init_override != nf) {
        free(fl); free(fg);
        return fail(p, "init override: count differs from the selected frames");
    }
    free(p->last_init);
    p->last_init = (int32_t*)malloc((nf ? nf : 1) * sizeof(int32_t));
    p->n_last_init = nf;
    for (size_t i = 0; i < nf; ++i) p->last_init[i] = INT32_MIN;
    parallel_for(p->nthreads, nf, sync_init_fn, &ctx);
    free(p->init_override); /* for one call only */
    p->init_override = NULL;
    p->n_init_override = 0;

    const double c_armijo = 2e-4, decay = .1, t0 = 1e-3; /* :226 */
    const int max_bt = 10;
    const double delay_b = .3; /* :260 */
    double delay_v = 0;        /* :261 (zero-filled) */
    int converge_counter = 0, rows = 0;
    for (int it = 0; it < p->max_outer; ++it) { /* :309 */
        ctx.delay = d;
        if (!simplified) parallel_for(p->nthreads, nf, sync_motion_fn, &ctx); /* :311 */
        /* :298-305 do_opt_delay -> Backtrack::Step (backtrack.cpp:3-13) */
        double x0 = d - delay_b * delay_v;
        ctx.delay = x0; ctx.with_grad = 1;
        parallel_for(p->nthreads, nf, sync_loss_fn, &ctx);
        double v = sum_in_order(fl, nf), g = sum_in_order(fg, nf);
        double m = g * g, t = t0;
        int trials = 0;
        ctx.with_grad = 0;
        for (int i = 0; i < max_bt; ++i) {
            ctx.delay = x0 - t * g;
            parallel_for(p->nthreads, nf, sync_loss_fn, &ctx);
            double v1 = sum_in_order(fl, nf);
            ++trials;
            if (v - v1 >= t * c_armijo * m) break;
            t *= decay;
        }
        double step = -t * g;
        delay_v = delay_b * delay_v + step; /* :301 */
        d += delay_v;                       /* :302 */
        double step_size = fabs(step);      /* :304 */
        if (trace && rows < cap) {
            double* r = trace + 6 * rows;
            r[0] = d; r[1] = step; r[2] = v; r[3] = g; r[4] = t; r[5] = trials;
        }
        ++rows;
        if (step_size < 1e-4) converge_counter++; else converge_counter = 0; /* :316-320 */
        if (converge_counter > 5) break;                                     /* :322-324 */
        if (fabs(d - search_center) > search_radius) break;                  /* :326-328 */
        if (p->verbose) fprintf(stderr, "%g %g\n", d, step_size);            /* :330 */
    }
    ctx.delay = d; ctx.with_grad = 0;
    parallel_for(p->nthreads, nf, sync_loss_fn, &ctx); /* :333 */
    *cost = sum_in_order(fl, nf);
    *delay_out = d;
    if (n_rows) *n_rows = rows;
    free(fl); free(fg);
    return 0;
}

void ora_set_init_override(ora_problem* p, const int32_t* winners, size_t n) {
    free(p->init_override);
    p->init_override = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
    memcpy(p->init_override, winners, n * sizeof(int32_t));
    p->n_init_override = n;
}

size_t ora_last_init_winners(const ora_problem* p, int32_t* out, size_t cap) {
    for (size_t i = 0; i < p->n_last_init && i < cap; ++i) out[i] = p->last_init[i];
    return p->n_last_init;
}

int ora_sync_trace(ora_problem* p, double initial_delay, int64_t frame_begin,
                   int64_t frame_end, double search_center, double search_radius, double* cost,
                   double* delay_out, double* trace, int cap, int* n_rows) {
    return sync_trace_impl(p, 0, initial_delay, frame_begin, frame_end, search_center, search_radius, cost, delay_out,
                           trace, cap, n_rows);
}

/* Sync of the thesis' simplified mode (section 2.11): the outer loop of core_private.cpp:298-331 on the
 * no-translation loss; no GuessMotion, no per-frame motion optimisation, k from the row norms. */
int ora_sync_simplified_trace(ora_problem* p, double initial_delay, int64_t frame_begin,
                              int64_t frame_end, double search_center, double search_radius, double* cost,
                              double* delay_out, double* trace, int cap, int* n_rows) {
    return sync_trace_impl(p, 1, initial_delay, frame_begin, frame_end, search_center, search_radius, cost, delay_out,
                           trace, cap, n_rows);
}

/* one frame of the simplified mode: k at delay_k, then loss and its central-difference d/d-delay at delay */
int ora_loss_simplified(const ora_problem* p, int64_t frame, double delay_k, double delay, double* k_out, double* loss,
                        double* dd_numeric) {
    frame_t* f = find_frame(p, frame);
    if (!f) return 1;
    size_t n = f->n;
    double* P = (double*)malloc(3 * n * sizeof(double));
    compute_problem(p, f, delay_k, P);
    double ss = 0;
    for (size_t j = 0; j < 3 * n; ++j) ss += P[j] * P[j];
    const double k = clamp_k(1 / sqrt(ss) * 1e2);
    compute_problem(p, f, delay - kNumericDiffStep, P);
    double ll = loss_simple_from_P(P, n, k);
    compute_problem(p, f, delay + kNumericDiffStep, P);
    double lr = loss_simple_from_P(P, n, k);
    compute_problem(p, f, delay, P);
    *loss = loss_simple_from_P(P, n, k);
    *dd_numeric = (lr - ll) / 2 / kNumericDiffStep;
    *k_out = k;
    free(P);
    return 0;
}

int ora_sync(ora_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end,
             double search_center, double search_radius, double* cost, double* delay) {
    return ora_sync_trace(p, initial_delay, frame_begin, frame_end, search_center, search_radius, cost, delay, NULL, 0, NULL);
}

int ora_sync_state(const ora_problem* p, double* M, double* k, int cap, int* n_frames) {
    int n = 0;
    for (size_t i = 0; i < p->nsel && n < cap; ++i, ++n) {
        const frame_t* f = &p->frames[p->sel[i]];
        M[3 * n] = f->M[0]; M[3 * n + 1] = f->M[1]; M[3 * n + 2] = f->M[2];
        k[n] = f->k;
    }
    if (n_frames) *n_frames = n;
    return 0;
}
