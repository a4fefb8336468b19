/*
 * rssync_c.h -- flat C-ABI over the ISyncProblem surface of librssync_core.so.
 *
 * The reference exposes only a C++ vtable (src/core/public/rssync.h:9-31).
 * A host that is not C++ (ctypes, cgo, JNI, N-API ...) binds these entry
 * points instead; each one forwards to exactly one ISyncProblem method, named
 * in its comment.  INTEGRATION.md shows both bindings.
 *
 * Unlike the C++ surface, these calls can report a failure instead of ending
 * the process: with rssync_set_panic_mode(1) a "panic" (invalid input, device
 * error) makes the call return non-zero and rssync_last_error() gives the
 * reason; with mode 0 (default) the reference behaviour applies (panic.txt,
 * exit(1)).
 *
 * The rssync_ext_* entry points are extensions the reference does not have:
 * every hyper-parameter in the reference is a literal and its sampler is
 * seeded from std::random_device, so a reproducible, benchmarkable and
 * multi-GPU build needs a few knobs.  None of them changes the meaning of the
 * six ISyncProblem calls at their defaults.
 */
#ifndef RSSYNC_C_H
#define RSSYNC_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rssync_problem rssync_problem;

/* CreateSyncProblem()  (rssync.h:31).  NULL if no HIP device is usable
 * (there is no CPU fallback in this library). */
rssync_problem* rssync_create(void);
/* delete through the virtual destructor (rssync.h:11) */
void rssync_destroy(rssync_problem* p);
/* A handle on an ISyncProblem* that CreateSyncProblem() of this library returned, for C++
 * clients that want the rssync_ext_* entry points on the object they already hold.  The handle
 * does not own the object: rssync_destroy() on it frees the handle only.  NULL if the pointer
 * is not this library's. */
rssync_problem* rssync_ext_borrow(void* isync_problem);
const char* rssync_last_error(void);
void rssync_set_panic_mode(int mode); /* 0 = panic.txt + exit(1); 1 = return status */

/* ISyncProblem::SetGyroQuaternions(const double*, size_t, double, double)  (rssync.h:13-14) */
int rssync_set_gyro_quaternions(rssync_problem* p, const double* data, size_t count,
                                double sample_rate, double first_timestamp);
/* ISyncProblem::SetGyroQuaternions(const int64_t*, const double*, size_t)  (rssync.h:15-16) */
int rssync_set_gyro_quaternions_ts(rssync_problem* p, const int64_t* timestamps_us,
                                   const double* quats, size_t count);
/* ISyncProblem::SetTrackResult  (rssync.h:17-18) */
int rssync_set_track_result(rssync_problem* p, int64_t frame, const double* ts_a,
                            const double* ts_b, const double* rays_a, const double* rays_b,
                            size_t count);
/* ISyncProblem::PreSync  (rssync.h:19-21); pair -> two out-parameters */
int rssync_pre_sync(rssync_problem* p, double initial_delay, int64_t frame_begin,
                    int64_t frame_end, double search_step, double search_radius, double* cost,
                    double* delay);
/* ISyncProblem::Sync  (rssync.h:22-24) */
int rssync_sync(rssync_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end,
                double search_center, double search_radius, double* cost, double* delay);
/* ISyncProblem::DebugPreSync  (rssync.h:26-28) */
int rssync_debug_pre_sync(rssync_problem* p, double initial_delay, int64_t frame_begin,
                          int64_t frame_end, double search_radius, double* delays, double* costs,
                          int point_count);

/* ---------------- extensions (not in the reference) ---------------- */

/* seed of the hypothesis sampler (default 0x5EED0000, env RSSYNC_SEED) */
int rssync_ext_set_seed(rssync_problem* p, uint64_t seed);
/* cap on Sync's outer iterations (reference literal 400, core_private.cpp:309; env RSSYNC_MAX_OUTER_ITERS) */
int rssync_ext_set_max_outer_iters(rssync_problem* p, int iters);
/* 0 silences the "delay step" progress lines Sync writes to stderr (core_private.cpp:330; env RSSYNC_QUIET=1) */
int rssync_ext_set_verbose(rssync_problem* p, int verbose);
/* The per-frame motion optimiser restates ens::L_BFGS (core_private.cpp:264-294; third party, unpinned).
 * When a line search's best step is not its last, 0 (default) keeps value and gradient as the last trial
 * left them while the iterate moves to the best step (the published LineSearch); 1 re-evaluates there. */
int rssync_ext_set_lbfgs_reeval(rssync_problem* p, int reeval);
/* Sync's outer loop normally runs on the device when one GPU holds all frames and no reduce hook is set (the
 * scalar decisions are taken between the launches, the host polls every few iterations); 1 keeps it on the
 * host, where it always runs otherwise.  Both give the same bits.  Environment: RSSYNC_HOST_LOOP=1. */
int rssync_ext_set_host_loop(rssync_problem* p, int host_loop);
/* With frames sharded over ranks: the library's own RCCL communicator keeps Sync's loop on the device (the window
 * sums are all-reduced on the stream between the kernels); a reduce hook normally means the host loop, one hook
 * call per launch.  1 = keep the loop on the device with a hook too: the hook is called between the kernels on the
 * window sums (the stream is drained at each call).  Same structure as the RCCL path, any transport. */
int rssync_ext_set_hook_device_loop(rssync_problem* p, int on);
/* line searches of the last rssync_ext_opt_motion call whose best step was not the last one tried */
int rssync_ext_lbfgs_best_not_last(rssync_problem* p, uint64_t* count);
/* ONE object, several GPUs (the reference parallelises over frames inside the object, core_private.cpp:73,231,
 * 245,263): the frames are spread over the listed devices of this process in contiguous blocks, every
 * PreSync/Sync launches on all of them and adds their partial sums on the host (a few kB) in an order that
 * does not depend on the device count -- results are bit-identical to the single-GPU run.  Also settable
 * without touching the client: environment RSSYNC_GPUS = "4" (devices 0..3) or "0,2,5".  Default: the
 * calling thread's current device.  May be called at any time; the data is uploaded again when next needed. */
int rssync_ext_set_devices(rssync_problem* p, const int* device_ids, int n_devices);
int rssync_ext_device_count(rssync_problem* p);
/* run the kernels on a caller-owned hipStream_t (NULL = internal stream); single-GPU objects only */
int rssync_ext_set_stream(rssync_problem* p, void* hip_stream);

/* Multi-GPU: frames are sharded over ranks (one process per GPU); the only
 * exchange on the path is a sum of a few doubles (PreSync: the candidate cost
 * vector; Sync: loss/gradient and the line-search losses).  The hook must sum
 * buf[0..n) in place over all ranks (e.g. an RCCL all-reduce) and return 0; any other return value
 * is a failed exchange and panics ("reduce hook failed"): a rank that carried on with its local
 * sums would take different decisions from the others.  n is not bounded by the library (batched
 * sync points exchange candidates x windows doubles).  NULL = single rank. */
typedef int (*rssync_reduce_fn)(double* buf, size_t n, void* user);
int rssync_ext_set_reduce_hook(rssync_problem* p, rssync_reduce_fn fn, void* user);

/* Native exchange instead of a reduce hook: the library keeps its own RCCL communicator (one rank
 * per process, librccl opened at run time).  Rank 0 calls rccl_unique_id (128 bytes), the host
 * hands the bytes to every rank by whatever means it has, every rank calls rccl_init; from then on
 * the sums of PreSync / Sync are all-reduced with ncclAllReduce on the problem's stream. */
/* preflight: 0 if this process can use RCCL at all (library and entry points resolve; no communication) -- the ranks
 * agree on the outcome of THIS before any of them enters the collective rccl_init (rs-sync_amd/dist.py);
 * library: which librccl is used ("... (already loaded in this process)" when the host's own copy was found) */
int rssync_ext_rccl_preflight(rssync_problem* p);
const char* rssync_ext_rccl_library(rssync_problem* p);
int rssync_ext_rccl_unique_id(rssync_problem* p, void* id128);
int rssync_ext_rccl_init(rssync_problem* p, const void* id128, int rank, int world_size);
/* leave that communicator (collective: every rank calls it); the problem is a single rank again */
int rssync_ext_rccl_shutdown(rssync_problem* p);
/* A NO-OP since round 5, kept so that callers written against rounds 2-4 still link.  (Then the kernel shapes followed
 * the largest per-frame track count of the whole problem and ranks had to agree on it.  Now a frame's kernels -- and the
 * order of its sums -- follow the frame's OWN track count, as the reference evaluates every frame on its own,
 * core_private.cpp:73-86, :231-238, :263-295: a frame's result is the same on any rank, in any window, without an
 * exchange.) */
int rssync_ext_set_tracks_hint(rssync_problem* p, uint32_t max_tracks_all_ranks);
/* exchanges with other ranks so far (reduce hook or native RCCL): calls and doubles summed */
int rssync_ext_exchange_stats(rssync_problem* p, uint64_t* calls, uint64_t* doubles);
/* The window executor (Sync of small frames as one device-scheduled launch).
 * check on: every call it has run is run again by the launch chain and must give the same bits, or the call panics
 * (a debug mode, also RSSYNC_EXECUTOR_CHECK=1).  stats: calls the executor completed, of those verified against the
 * chain, and queue[4] of the last run = {numbers claimed, numbers pushed, cells of the ring, waves launched}. */
int rssync_ext_set_executor_check(rssync_problem* p, int on);
/* With the check mode off, one executor call in `every` (counted over all objects of the process; default 256, also
 * RSSYNC_EXECUTOR_CHECK_EVERY; 0 = never) is verified the same way: the executor's cross-workgroup hand-offs are a
 * measured protocol, not an architectural guarantee, and this keeps a tripwire in production at < 1 % of its time. */
int rssync_ext_set_executor_check_every(rssync_problem* p, uint32_t every);
/* diagnostics: how the kernels' LDS spline windows were laid out for this problem's gyro rate (first device):
 * out = {widest frame in knots, knots per fp64 window, fp32 window of the last PreSync sweep (0 = the 80 knots compiled
 * into the kernel, else knots of dynamic LDS), its candidates per workgroup, the same window for the last GuessMotion
 * search, delays per pass of the line-search trials' kernel, widest frame counting only the two ends' ranges of each
 * pair (what the dynamic windows stage where that is fewer knots), 0} */
int rssync_ext_window_info(rssync_problem* p, uint32_t out[8]);
int rssync_ext_executor_stats(rssync_problem* p, uint64_t* runs, uint64_t* checked, uint32_t queue[4]);
/* How often a verified call of the production sample (one executor call in RSSYNC_EXECUTOR_CHECK_EVERY, default 256) did
 * NOT give the launch chain's bits.  Such a call reports the evidence on stderr, returns the CHAIN's results, and the object
 * uses the chain from then on (round 6; until then it panicked).  In the check mode (RSSYNC_EXECUTOR_CHECK=1 /
 * rssync_ext_set_executor_check) a difference is still a panic: that mode exists to find one.  Expected: 0, always. */
int rssync_ext_executor_mismatches(rssync_problem* p, uint64_t* count);
/* Near-static footage (a camera on a tripod, a slow pan: rows of the residual matrix below ~2e-4).  The reference computes
 * rows, norms and the safe_normalize decisions in double (core_private.cpp:19-28,45-46, inline_utils.hpp:5-11); PreSync's
 * fp32 sweep recomputes exactly those (frame, candidate) pairs from the fp64 streams.  *pairs = pairs recomputed so far on
 * this object, *sweeps = sweeps (PreSync / DebugPreSync / curve calls, per slice) that needed it, *searches = GuessMotion
 * searches (one per frame and Sync call) that took their rows from the fp64 streams, in place.  All stay 0 on ordinary
 * footage; RSSYNC_NO_FP64_ROWS=1 (read when a problem is created) switches the mechanism off. */
int rssync_ext_near_static_stats(rssync_problem* p, uint64_t* pairs, uint64_t* sweeps, uint64_t* searches);
/* TEST-VARIANTS build of the library only (the product returns an error): later PreSync-type sweeps also store the
 * |residual| bit patterns the LMedS selection worked on; _get returns the last sweep's, [candidate][frame of the selection]
 * [hypothesis][cap_rows] as uint32 (0xffffffff: no such row), dims = {candidates, frames, hypotheses, cap_rows} (out NULL:
 * only the dims).  tests/test_gpu_fuzz.py: the winner is the exact arg-min of the lower quartile, first wins
 * (core_private.cpp:48-56) -- asserted without a tolerance. */
int rssync_ext_debug_residuals(rssync_problem* p, int on, uint32_t cap_rows);
int rssync_ext_debug_residuals_get(rssync_problem* p, uint32_t* out, size_t n_words, uint32_t dims[4]);
/* Diagnostics of the bit-exactness tests (tests/test_gpu_bitexact.py).  GuessMotion's 200-hypothesis search runs
 * in fp32 and leaves one winning hypothesis index per slot (window-major, frames ascending); with recording on,
 * the winners of the last Sync / sync_windows / sync_simplified call can be read back, and set_init_override
 * installs a list in place of the search's result for the NEXT such call only -- so that two builds of the
 * library (the GPU one and the CPU stand-in of the tests) can be started from the same motion estimates. */
int rssync_ext_record_init_winners(rssync_problem* p, int on);
int rssync_ext_last_init_winners(rssync_problem* p, int32_t* out, size_t cap, size_t* n);
int rssync_ext_set_init_override(rssync_problem* p, const int32_t* winners, size_t n);
/* the Sync kernels' fp64 building blocks on caller data: op 0 a / b, 1 sqrt(a), 2 log1p and 1/(1+a) as the
 * kernels compute them (out[2i], out[2i+1]), 3 fma(a, b, a), 4 the kernels' 64-lane sum of each block of 64
 * values (out[block]) */
int rssync_ext_debug_math64(rssync_problem* p, int op, const double* a, const double* b, double* out, size_t n);
/* pack the tracks and the gyro spline and copy them to HBM now (otherwise done lazily by the
 * first PreSync/Sync/DebugPreSync after a setter) */
int rssync_ext_upload(rssync_problem* p);

/* diagnostics used by the parity tests and the benchmark */
int rssync_ext_sample_rate(rssync_problem* p, double* sample_rate, double* quats_start,
                           size_t* n_knots);
int rssync_ext_gyro_knots(rssync_problem* p, double* out, size_t cap); /* [4*n_knots] */
/* the spline table the kernels read: [n_knots][16] = y[4], b[4], c[4], d[4] over [w,x,y,z]; builds it if needed */
int rssync_ext_gyro_table(rssync_problem* p, double* out, size_t cap);
/* PreSync's whole curve: delays/costs [cap], optional per-frame matrices [n][n_frames] */
int rssync_ext_presync_curve(rssync_problem* p, double initial_delay, int64_t frame_begin,
                             int64_t frame_end, double search_step, double search_radius,
                             double* delays, double* costs, int cap, int* n_out,
                             double* frame_costs, int32_t* best_h, int* n_frames);
/* residual matrix of one frame (fp32, N x 3) and its d/d-delay at one delay */
int rssync_ext_problem_matrix(rssync_problem* p, int64_t frame, double delay, float* P, float* dP,
                              size_t cap_rows, size_t* n_rows);
/* Sync's GuessMotion/GuessK on frames [begin, end]; M[3n], k[n] in ascending frame order */
int rssync_ext_init_motion(rssync_problem* p, double delay, int64_t frame_begin, int64_t frame_end,
                           double* M, double* k, int cap, int* n_frames);
/* per-frame L-BFGS at a fixed delay on the current selection (after init_motion / Sync) */
int rssync_ext_opt_motion(rssync_problem* p, double delay, double* M, double* k, int cap,
                          int* n_frames, uint64_t* iters, uint64_t* evals);
int rssync_ext_set_motion(rssync_problem* p, const double* M, const double* k, int n_frames);
/* sum over the current selection of loss (and analytic d/d-delay) at n delays */
int rssync_ext_loss(rssync_problem* p, const double* delays, int n, double* loss, double* grad);
/* "Simplified" mode of the thesis (section 2.11 eq. (12), p.26): translation neglected, loss
 * sum over frames and tracks of log1p((k |h_j|)^2) with h_j = the j-th row of the residual matrix P
 * (core_private.cpp:28).  The optimisation is one-dimensional: Sync's outer loop (core_private.cpp:298-331:
 * backtracking step with momentum, the same stopping rules) without GuessMotion and without the per-frame
 * motion optimisation; k per frame is GuessK's rule (:130-133, inline_utils.hpp:50) applied to the row norms,
 * clamp(100 / sqrt(sum_j |h_j|^2), 10, 1000), at the initial delay.  The reference snapshot has no code for this
 * mode; the thesis describes it.  frame_end inclusive, as Sync. */
int rssync_ext_sync_simplified(rssync_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end,
                               double search_center, double search_radius, double* cost, double* delay);
/* its pieces, for the parity tests: k per frame at `delay` (selects [frame_begin, frame_end]), then the loss
 * and its analytic d/d-delay on that selection */
int rssync_ext_init_k_simplified(rssync_problem* p, double delay, int64_t frame_begin, int64_t frame_end, double* k,
                                 int cap, int* n_frames);
int rssync_ext_loss_simplified(rssync_problem* p, const double* delays, int n, double* loss, double* grad);
/* the residual matrix as the Sync kernels compute it (fp64; rssync_ext_problem_matrix: fp32, PreSync) */
int rssync_ext_problem_matrix64(rssync_problem* p, int64_t frame, double delay, double* P, double* dP, size_t cap_rows,
                                size_t* n_rows);

/* Upstream steps of the reference driver, moved behind the library.
 * set_track_pixels replaces the driver's undistort + normalise + row-time loop followed by
 * SetTrackResult (core_testcode.cpp:135-158): points_* are count x {x, y} pixel positions of the
 * tracked points in the current / next video frame, frame_time_* those frames' times in seconds,
 * image_rows the frame height; row time = frame_time + lens.ro * y / image_rows.  The fisheye
 * inverse (core_testcode.cpp:63-95) runs on the device in fp64 and writes the packed ray
 * streams directly.  A frame set this way replaces one set by rssync_set_track_result and vice
 * versa. */
typedef struct rssync_lens { /* core_testcode.cpp:55-61 */
    double ro;             /* rolling-shutter readout time, seconds */
    double fx, fy, cx, cy; /* pixels */
    double k1, k2, k3, k4;
} rssync_lens;
int rssync_ext_set_track_pixels(rssync_problem* p, int64_t frame, double frame_time_a, double frame_time_b,
                                const double* points_a, const double* points_b, size_t count,
                                const rssync_lens* lens, double image_rows);
/* optdata_fill_gyro (core_testcode.cpp:36-52): integrate angular rates (count x {x,y,z}, rad/s, at
 * timestamps in seconds) to orientations and hand them to the timestamped gyro setter.
 * orientation: telemetry-parser's three-letter axis string ("XYZ" = identity) or NULL. */
int rssync_ext_set_gyro_rates(rssync_problem* p, const double* timestamps_s, const double* rates, size_t count,
                              const char* orientation);
/* The driver's orientation-guessing sweep (core_testcode.cpp:186-224): for each orientation
 * string, set_gyro_rates(orientation) then PreSync(initial_delay, frame_begin, frame_end, step,
 * radius); costs[i] / delays[i] are that PreSync's result.  Host preparation of orientation i+1
 * overlaps the GPU sweep of orientation i.  The last orientation stays installed. */
int rssync_ext_orientation_sweep(rssync_problem* p, const double* timestamps_s, const double* rates, size_t count,
                                 const char* const* orientations, int n_orientations, double initial_delay,
                                 int64_t frame_begin, int64_t frame_end, double search_step, double search_radius,
                                 double* costs, double* delays);
/* the packed device ray streams of one frame ({ax,bx,ay,by} and {az,bz,ta,tb} per pair), for tests */
int rssync_ext_frame_rays(rssync_problem* p, int64_t frame, float* a4, float* b4, size_t cap, size_t* n);

/* Batched windows (the driver's loop, core_testcode.cpp:303-316, calls
 * PreSync + 4x Sync once per window position; these run all positions in one call).
 * pre_sync_windows: window w = PreSync(initial_delay, begins[w], ends[w], step, radius) -- the
 * candidate list is shared, the LMedS kernel runs once over the union of the windows' frames.
 * sync_windows: window w = the w-th of n consecutive Sync(initial_delays[w], begins[w], ends[w],
 * center, radius) calls; the windows advance in lock-step, one launch per stage for all of them.
 * Frame ranges follow the methods they batch: ends exclusive for pre_sync, inclusive for sync. */
int rssync_ext_pre_sync_windows(rssync_problem* p, double initial_delay, const int64_t* frame_begins,
                                const int64_t* frame_ends, int n_windows, double search_step,
                                double search_radius, double* costs, double* delays);
int rssync_ext_sync_windows(rssync_problem* p, const double* initial_delays, const int64_t* frame_begins,
                            const int64_t* frame_ends, int n_windows, double search_center,
                            double search_radius, double* costs, double* delays);
/* The reference driver's whole loop over sync points (core_testcode.cpp:303-316): for each
 * position pos: delay = initial_delay; if use_presync, delay = PreSync(delay, pos,
 * pos + sync_window, presync_step, presync_radius).second, else the search radius is infinite;
 * then sync_repeats (the driver: 4) times delay = Sync(delay, pos, pos + sync_window,
 * initial_delay, radius).second.  delays[w] / costs[w] (may be NULL) are the last Sync's result
 * for position w, identical to running that loop through the ISyncProblem methods. */
int rssync_ext_sync_points(rssync_problem* p, const int64_t* positions, int n_points, int64_t sync_window,
                           double initial_delay, int use_presync, double presync_step,
                           double presync_radius, int sync_repeats, double* costs, double* delays);
/* trace of one window of the last sync_windows / sync_points call (same rows as sync_trace) */
int rssync_ext_window_trace(rssync_problem* p, int window, double* trace, int cap_rows, int* n_rows);
/* trace of the last Sync: rows of {delay_after, step, loss_at_x0, grad_at_x0, t, trials} */
int rssync_ext_sync_trace(rssync_problem* p, double* trace, int cap_rows, int* n_rows);
/* the internal device context (rship_ctx*, include/rssync_hip.h) behind this problem, for
 * kernel-level tests and profiling tools; owned by the problem */
void* rssync_ext_device_context(rssync_problem* p);
/* HIP-event kernel timing: kind 0 LMedS tile (PreSync), 1 loss (line-search trials, final loss), 2 motion,
 * 3 reduce, 4 LMedS init (Sync), 5 packing, 6 gyro, 7 loss + analytic gradient */
int rssync_ext_profile(rssync_problem* p, int enable);
int rssync_ext_profile_get(rssync_problem* p, int kind, uint64_t* launches, double* total_ms);
int rssync_ext_profile_reset(rssync_problem* p);

#ifdef __cplusplus
}
#endif
#endif
