/*
 * rssync_oracle.h -- CPU restatement of the rs-sync PreSync/Sync hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (rs-sync_amd/) may
 * include, link or call this.  It is used by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg, as the checker and as the timed CPU
 * baseline ("port").
 *
 * PARITY UNPINNED: the reference (VladimirP1/rs-sync) ships no tests, golden
 * vectors or fixtures for this path and cannot be
 * built in this image (every TU includes <armadillo>; ensmallen and TBB
 * headers are absent).  The oracle is therefore pinned only by independent
 * cross-checks (scipy natural cubic spline, scipy Rotation, finite
 * differences, ground-truth delay recovery on synthetic scenes) -- see
 * tests/test_oracle_*.py.
 *
 * Every function cites the reference file:line it restates.  All arithmetic
 * is IEEE double, as in the reference.
 *
 * Two deliberate, documented deviations from the shipped reference:
 *  (1) the LMedS hypothesis sampler is a counter-based RNG keyed on
 *      (seed, frame id, stream, hypothesis index) instead of a thread_local
 *      std::mt19937 seeded from std::random_device (inline_utils.hpp:13-17);
 *      the shipped reference is non-deterministic, so parity is only
 *      definable against a seeded sampler that the HIP path shares;
 *  (2) the motion Jacobian uses the O(N) closed form of the reference's
 *      dense N x N Jacobian chain (core_private.cpp:99-114); values are
 *      identical, cost is not (this favours the CPU baseline).
 */
#ifndef RSSYNC_ORACLE_H
#define RSSYNC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_problem ora_problem;

/* RNG stream tags shared with the HIP path (DESIGN.md "Sampler"). */
#define ORA_STREAM_SYNC_INIT 0x80000000u /* + sync call counter */
#define ORA_STREAM_DEBUG 0x40000000u     /* + point index (DebugPreSync) */

ora_problem* ora_create(void);
void ora_destroy(ora_problem* p);
const char* ora_last_error(const ora_problem* p);

/* knobs (the reference hard-codes all of these) */
void ora_set_seed(ora_problem* p, uint64_t seed);
void ora_set_threads(ora_problem* p, int nthreads);      /* frames-parallel pool */
void ora_set_max_outer_iters(ora_problem* p, int iters); /* reference: 400 */
void ora_set_faithful(ora_problem* p, int faithful);     /* 1 = reference evaluation schedule */
void ora_set_verbose(ora_problem* p, int verbose);       /* 1 = print "delay step" like the reference */
/* L-BFGS line search whose best step is not its last: 0 (default) = value and gradient stay those of the
 * last trial, as ensmallen's LineSearch leaves them; 1 = re-evaluate at the best step */
void ora_set_lbfgs_reeval(ora_problem* p, int reeval);
long ora_lbfgs_best_not_last(const ora_problem* p);      /* how many line searches ended that way so far */

/* core_private.cpp:135-140 */
int ora_set_gyro_quaternions(ora_problem* p, const double* data, size_t count,
                             double sample_rate, double first_timestamp);
/* core_private.cpp:142-190 */
int ora_set_gyro_quaternions_ts(ora_problem* p, const int64_t* timestamps_us,
                                const double* quats, size_t count);
/* core_private.cpp:192-203 (copies at call time) */
int ora_set_track_result(ora_problem* p, int64_t frame, const double* ts_a, const double* ts_b,
                         const double* rays_a, const double* rays_b, size_t count);
/* core_private.cpp:205-209 -> :61-90 */
int ora_presync(ora_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end,
                double search_step, double search_radius, double* cost, double* delay);
/* core_private.cpp:211-334 */
int ora_sync(ora_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end,
             double search_center, double search_radius, double* cost, double* delay);
/* core_private.cpp:336-361 */
int ora_debug_presync(ora_problem* p, double initial_delay, int64_t frame_begin,
                      int64_t frame_end, double search_radius, double* delays, double* costs,
                      int point_count);

/* ---- introspection used by the parity tests ---- */
double ora_sample_rate(const ora_problem* p);
double ora_quats_start(const ora_problem* p);
size_t ora_gyro_count(const ora_problem* p);
/* resampled grid quaternions [4*count] (after either setter) */
void ora_gyro_knots(const ora_problem* p, double* out);
size_t ora_frame_count(const ora_problem* p);
size_t ora_frame_tracks(const ora_problem* p, int64_t frame);

/* minispline.cpp:48-55 / :57-64 on the four components; x in knot units */
void ora_spline_eval(const ora_problem* p, double x, double out[4]);
void ora_spline_deriv(const ora_problem* p, double x, double out[4]);
/* quat.cpp:55-74 */
void ora_quat_slerp(const double p4[4], const double q4[4], double t, double out[4]);

/* core_private.cpp:15-32; P is row-major N x 3 */
int ora_compute_problem(const ora_problem* p, int64_t frame, double delay, double* P);
/* sampler: indices drawn for hypothesis h of (frame, stream) with n rows */
void ora_sample_pair(uint64_t seed, int64_t frame, uint32_t stream, uint32_t h, uint32_t n,
                     uint32_t* i0, uint32_t* i1);
/* core_private.cpp:34-59 on P(frame, delay); returns winning hypothesis + its quantile */
int ora_guess_motion(const ora_problem* p, int64_t frame, double delay, int max_iters,
                     uint32_t stream, double M[3], int* best_h, double* best_med);
/* one frame's PreSync term: core_private.cpp:75-85 */
int ora_frame_presync_cost(const ora_problem* p, int64_t frame, double delay, uint32_t stream,
                           double* cost, int* best_h);
/* the whole PreSync curve (candidate delays as the reference's double loop generates them,
 * core_private.cpp:69-70).  frame_costs (optional) is [n_cand][n_frames] row-major in
 * ascending frame-id order; best_h (optional) likewise. */
int ora_presync_curve(ora_problem* p, double initial_delay, int64_t frame_begin,
                      int64_t frame_end, double search_step, double search_radius,
                      double* delays, double* costs, int cap, int* n_out, double* frame_costs,
                      int* best_h);
/* core_private.cpp:92-123: loss, the reference's central-difference d/d-delay (:96-97,112),
 * the analytic d/d-delay (not in the reference) and dL/dM (closed form) */
int ora_loss(const ora_problem* p, int64_t frame, double delay, const double M[3], double k,
             double* loss, double* dd_numeric, double* dd_analytic, double gM[3]);
/* restated ens::L_BFGS on one frame at fixed delay (call site core_private.cpp:262-296) */
int ora_lbfgs_motion(const ora_problem* p, int64_t frame, double delay, double M[3], double k,
                     int* iters, int* evals, double* final_loss);
/* Sync with a per-outer-iteration trace: rows of {delay_after, step, loss_at_x0, t} */
/* test hooks: GuessMotion's winning hypothesis index per selected frame of the last Sync (INT32_MIN: simplified
 * mode), and a list of winners to use in place of the search in the NEXT Sync only */
void ora_set_init_override(ora_problem* p, const int32_t* winners, size_t n);
size_t ora_last_init_winners(const ora_problem* p, int32_t* out, size_t cap);
int ora_sync_trace(ora_problem* p, double initial_delay, int64_t frame_begin,
                   int64_t frame_end, double search_center, double search_radius, double* cost,
                   double* delay, double* trace, int cap, int* n_rows);
/* per-frame state after the last Sync: M[3*i..], k[i] in ascending frame-id order */
int ora_sync_state(const ora_problem* p, double* M, double* k, int cap, int* n_frames);
/* The thesis' simplified mode (thesis-text.pdf section 2.11 eq. (12), printed p.26; the reference snapshot has
 * no code for it): loss sum_j log1p((k |h_j|)^2) with h_j the rows of P, k per frame = clamp(100 / |x|_2, 10,
 * 1000) over x_j = |h_j| (GuessK's rule, core_private.cpp:130-133), Sync's outer loop (:298-331) unchanged. */
int ora_sync_simplified_trace(ora_problem* p, double initial_delay, int64_t frame_begin,
                              int64_t frame_end, double search_center, double search_radius, double* cost,
                              double* delay_out, double* trace, int cap, int* n_rows);
int ora_loss_simplified(const ora_problem* p, int64_t frame, double delay_k, double delay, double* k_out, double* loss,
                        double* dd_numeric);

/* ---- driver steps upstream of the ISyncProblem calls (rssync_oracle_driver.c) ---- */
typedef struct ora_lens { /* core_testcode.cpp:55-61 */
    double ro, fx, fy, cx, cy, k1, k2, k3, k4;
} ora_lens;
/* core_testcode.cpp:63-95 */
void ora_undistort_point(const ora_lens* lens, double px, double py, double out[2]);
/* core_testcode.cpp:135-152: px_* = n x {x, y}; outputs as SetTrackResult takes them */
void ora_pixels_to_tracks(const ora_lens* lens, double time_a, double time_b, double rows, const double* px_a,
                          const double* px_b, size_t n, double* ts_a, double* ts_b, double* rays_a, double* rays_b);
/* quat.cpp:5-17 */
void ora_quat_from_aa(const double aa[3], double out[4]);
/* core_testcode.cpp:36-51: rates n x 3 rad/s, timestamps seconds -> quats n x 4, int64 microseconds */
void ora_integrate_gyro(const double* timestamps_s, const double* rates, size_t n, double* quats, int64_t* ts_us);
/* telemetry-parser orientation string ("XYZ" = identity); 0 on success */
int ora_orient_rates(const double* rates, size_t n, const char* orientation, double* out);

#ifdef __cplusplus
}
#endif
#endif
