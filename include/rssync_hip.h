/*
 * rssync_hip.h -- thin C-ABI between the C++ host solver (sync_problem.cpp)
 * and the gfx950 kernels (rssync_kernels.hip).  Plain pointers and sizes, int
 * status codes (0 = ok), no C++ types and no exceptions across it.
 *
 * This is the INTERNAL boundary of librssync_core.so.  The drop-in surface a
 * client of the reference binds is the C++ vtable in rssync.h (and its flat C
 * mirror rssync_c.h).  Each entry below names the reference code it takes
 * over (paths relative to VladimirP1/rs-sync src/).
 *
 * Conventions
 *  - a "delay" reaches the device as  delay * sample_rate = kd + fd  with kd an
 *    int32 knot count and fd a fraction in [0,1) (fp32 for the PreSync sweep, fp64 for the Sync
 *    kernels): absolute times are never rounded to fp32;
 *  - rays live in HBM as two float4 streams per frame with the two ends of a
 *    ray pair interleaved, {ax,bx,ay,by} and {az,bz,ta,tb} (a 16-byte load
 *    yields (a,b) component pairs in adjacent registers, which the packed
 *    fp32 rotation consumes without moves); ta/tb are the ray's spline
 *    parameter minus the frame's integer base knot (32 B per ray pair);
 *  - spline coefficients are 4 float4 per knot: y, b, c, d over [w,x,y,z];
 *  - every call is synchronous on return (results are in the host buffers).
 */
#ifndef RSSYNC_HIP_H
#define RSSYNC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rship_ctx rship_ctx;

/* one record of the device frame table (56 bytes) */
#define RSHIP_NO_SPLIT 0xffffffffu
typedef struct rship_frame {
    uint32_t ray_offset; /* first ray of the frame in the packed streams */
    uint32_t n_rays;
    int32_t base_knot; /* floor(min over rays of (ts - start) * fs) */
    float tmin, tmax;  /* min / max of the per-ray offsets ta, tb as the fp32 streams hold them */
    /* The two ENDS of a frame pair cover far fewer knots than the pair: ts_a lies within one read-out time of the
     * current frame, ts_b of the next (core_testcode.cpp:144-145), a frame interval apart.  range_a / range_b: the
     * knots the a-end / b-end offsets touch at delay 0, relative to base_knot, as lo | hi << 16 (hi inclusive =
     * floor of the end's largest offset); RSHIP_NO_SPLIT = not known or not representable (the kernels then stage
     * the whole span tmin .. tmax, which is always valid). */
    uint32_t range_a;
    int64_t id;            /* caller's frame number (keys the hypothesis sampler) */
    double tmin64, tmax64; /* the same bounds as the fp64 streams hold them */
    uint32_t range_b;
    uint32_t reserved;
} rship_frame;

/* status bits reported by the LMedS kernel; the host turns them into the
 * reference's panic messages (core_private.cpp:76-83) */
#define RSHIP_BAD_P 1u
#define RSHIP_BAD_M 2u
#define RSHIP_BAD_R 4u
#define RSHIP_BAD_RHO 8u
/* not an error: the PreSync sweep met (frame, candidate) pairs whose rows are too small for its fp32 inputs (a near-static
 * camera: |P| below ~2e-4) and flagged them; rship_presync_collect recomputes those pairs from the fp64 streams before
 * it returns (kernels/lmeds.hpp, "fp64 rows"; reference: core_private.cpp:19-28,45-46 are double throughout) */
#define RSHIP_NEAR_STATIC 16u

/* kernel kinds for rship_profile_get */
#define RSHIP_K_LMEDS 0  /* PreSync tile kernel */
#define RSHIP_K_LOSS 1   /* residual + robust loss at a batch of delays (line-search trials, final loss) */
#define RSHIP_K_MOTION 2 /* per-frame motion L-BFGS */
#define RSHIP_K_REDUCE 3 /* over-frames sums */
#define RSHIP_K_INIT 4   /* the LMedS kernel in GuessMotion/GuessK mode (Sync start) */
#define RSHIP_K_PIXELS 5 /* packing kernels: raw records (rays or pixels) -> packed fp32 + fp64 streams */
#define RSHIP_K_GYRO 6   /* gyro pipeline: integration scan, resampling, spline solve */
#define RSHIP_K_LOSS_GRAD 7 /* residual + robust loss + analytic d/d-delay (one delay per window) */
#define RSHIP_K_COUNT 8

int rship_create(rship_ctx** out, int device /* -1 = current device */);
void rship_destroy(rship_ctx* c);
const char* rship_last_error(const rship_ctx* c);
/* launch on a caller-owned hipStream_t (NULL = the context's own stream) */
int rship_set_stream(rship_ctx* c, void* hip_stream);
int rship_max_tracks(void); /* largest per-frame track count accepted: 2^24, an indexing bound (the reference has none,
                               core_private.cpp:192-203; a problem holds at most 2^32 rays).  Up to 8192 tracks a frame's
                               rows live in registers / LDS; larger frames run the kernels' slow exact variants
                               (kernels/lmeds_big.hpp, loss64_kernel<0>, opt_motion64_kernel<0, 4>) */
/* Behaviour switches.  RSHIP_OPT_LBFGS_REEVAL: what the restated ens::L_BFGS does when a line search's
 * best step is not its last -- 0 (default): iterate at the best step, value and gradient as the last
 * trial left them (the published LineSearch); 1: evaluate once more at the best step. */
#define RSHIP_OPT_LBFGS_REEVAL 1
/* RSHIP_OPT_TRACKS_HINT: accepted and ignored since round 5.  (Rounds 2-4: the largest per-frame track count of the
 * whole problem, from which the kernels whose workgroup shape fixes the order of a frame's sums picked the shape.  The
 * shape now follows each frame's OWN track count -- size classes, rssync_kernels.hip -- as the reference evaluates each
 * frame on its own, core_private.cpp:73-86, :231-238, :263-295: a frame's sums are the same on any device, in any
 * selection and on any rank without anybody agreeing on anything.) */
#define RSHIP_OPT_TRACKS_HINT 2
int rship_set_option(rship_ctx* c, int option, int value);

/* OptData::quats (core_private.hpp:18) is built ON THE DEVICE from whatever the caller has: uniform
 * orientation samples, timestamped orientation samples, or timestamped angular rates.  The table holds 16
 * doubles per knot = y[4], b[4], c[4], d[4] over [w,x,y,z] (the natural spline of minispline.cpp:3-46); the
 * device keeps it in fp64 (Sync kernels) and a copy rounded once to fp32 (PreSync).
 *
 *   rship_gyro_uniform       core_private.cpp:135-140: the samples are the knots; spline solve.
 *   rship_gyro_timestamped   core_private.cpp:142-190: order check, the integer-microsecond grid at the rate
 *                            rounded to 50 Hz, slerp onto the grid, spline solve.
 *   rship_gyro_rates_upload  + rship_gyro_rates_integrate: the reference driver's optdata_fill_gyro
 *                            (core_testcode.cpp:36-52): q_i = normalise(dq_i q_{i-1}) as a scan of quaternion
 *                            products, timestamps truncated to microseconds, then as the timestamped route.
 *                            axis/sign permute the rate axes (output axis c = sign[c] * input axis axis[c]);
 *                            one upload serves any number of integrations (the orientation sweep).
 * The routes that can reject their input fill *out and return 0; `status` says what the reference would have
 * complained about (its order of checks).  A nonzero return is a device failure. */
#define RSHIP_GYRO_OK 0
#define RSHIP_GYRO_OUT_OF_ORDER 1 /* bad_pos, bad_a = ts[pos-1], bad_b = ts[pos] */
#define RSHIP_GYRO_BAD_KNOT 2     /* non-finite sample after interpolation */
#define RSHIP_GYRO_SHORT_GRID 3   /* fewer than 2 grid points */
#define RSHIP_GYRO_BAD_RATE 4     /* rate rounds to <= 0 Hz, or first == last timestamp */
#define RSHIP_GYRO_BAD_START 5
#define RSHIP_GYRO_BAD_INPUT 6    /* non-finite timestamp or rate (rates route) */
#define RSHIP_GYRO_TOO_LARGE 7    /* negative timestamps or more than 2^26 grid points */
typedef struct rship_gyro_result {
    double fs, start;      /* grid rate (Hz), time of the first knot (s) */
    uint64_t first_sample; /* grid index of the first knot */
    uint64_t bad_pos;
    int64_t bad_a, bad_b;
    uint32_t n_knots;
    int32_t status;
} rship_gyro_result;
int rship_gyro_uniform(rship_ctx* c, const double* quats, uint32_t n, double sample_rate);
int rship_gyro_timestamped(rship_ctx* c, const int64_t* ts_us, const double* quats, uint32_t n, rship_gyro_result* out);
int rship_gyro_rates_upload(rship_ctx* c, const double* ts_s, const double* rates, uint32_t n);
int rship_gyro_rates_integrate(rship_ctx* c, const int32_t axis[3], const double sign[3], rship_gyro_result* out);
/* rship_gyro_rates_integrate without waiting, for a BATCH of orientations (rship_presync_batch_begin): the kernels are
 * enqueued -- the table they build is the one the next enqueued sweep reads --, *out is the grid (a function of the end
 * timestamps alone; status = what the HOST can tell), and what the device has to complain about (non-finite input, a
 * non-finite knot) goes to status record `slot` (1 .. 255); rship_gyro_batch_status(n, status[n]) reads records 1 .. n
 * after the batch has been collected: status[i] as rship_gyro_rates_integrate would have returned it. */
int rship_gyro_rates_integrate_enqueue(rship_ctx* c, const int32_t axis[3], const double sign[3], uint32_t slot, rship_gyro_result* out);
int rship_gyro_batch_status(rship_ctx* c, uint32_t n, int32_t* status);
/* read back: the knots [n_knots][4] / the fp64 table [n_knots][16] (tests, rssync_ext_gyro_knots) */
int rship_gyro_knots(rship_ctx* c, double* out, uint32_t cap_knots);
int rship_gyro_table(rship_ctx* c, double* out16, uint32_t cap_knots);

/* OptData::frame_data (core_private.hpp:21).  Track data travels in three steps:
 *  1. SetTrackResult copies the caller's arrays into a host staging arena obtained from
 *     rship_host_alloc (pinned memory: the copy the reference's contract requires,
 *     core_private.cpp:192-203) as one record per frame,
 *         rays:   ts_a[n], ts_b[n], rays_a[3n], rays_b[3n]        (8n doubles)
 *         pixels: {xa, ya, xb, yb}[n]                             (4n doubles);
 *  2. rship_upload_raw copies a range of the arena to the same offsets of the context's raw
 *     buffer, asynchronously on the context's copy stream (it may be called while the caller is
 *     still feeding frames);
 *  3. rship_pack_frames runs the packing kernels: raw records -> the packed fp32 streams the
 *     PreSync kernel reads ({ax,bx,ay,by} / {az,bz,ta,tb}, 32 B per ray pair) and the fp64 streams
 *     the Sync kernels read ({ax,bx} {ay,by} {az,bz} {ta,tb} as double2, 64 B per ray pair), with
 *     ta/tb = (ts - start) * fs - base_knot evaluated in fp64 on the device.  Nothing is packed on
 *     the host and nothing but the raw records crosses PCIe. */
void* rship_host_alloc(size_t bytes); /* pinned where a HIP device exists; NULL on failure */
void rship_host_free(void* p);
int rship_upload_raw(rship_ctx* c, const double* host, uint64_t arena_offset, uint64_t n_doubles);

typedef struct rship_pack_frame {
    uint64_t raw_offset;         /* arena offset (doubles) of the frame's record */
    uint32_t ray_offset, n_rays; /* where its rays go in the packed streams */
    double base;                 /* base knot, as the frame table has it */
    uint32_t is_pixels, reserved;
    double time_a, time_b, rows; /* pixels only: frame times (s), image rows */
    double lens[9];              /* pixels only: ro, fx, fy, cx, cy, k1, k2, k3, k4 */
} rship_pack_frame;
/* (Re)build the packed streams of ALL frames: table[i] / pack[i] describe frame i, frames in
 * ascending id order.  *bad = number of rays with a non-finite packed value (pixel frames: the
 * reference checks the rays it is handed, core_private.cpp:199-200). */
int rship_pack_frames(rship_ctx* c, const rship_frame* table, const rship_pack_frame* pack, uint32_t n_frames,
                      uint64_t total_rays, double start, double fs, uint32_t* bad);

/* One object over several devices (optional): the frame-table records of ALL frames of the problem (only n_rays, tmin,
 * tmax, range_a, range_b are read), so that every shard plans its LDS spline windows from the same frames as a single
 * device holding everything would.  Call after rship_pack_frames (which forgets an earlier list). */
int rship_set_problem_frames(rship_ctx* c, const rship_frame* table_all, uint32_t n_all);

/* the frames a PreSync/Sync call works on (indices into the table; replaces the
 * frame filters at core_private.cpp:65-68, :218-219, :340-343) */
int rship_select_frames(rship_ctx* c, const uint32_t* idx, uint32_t n);
/* Batched windows: the selection is a list of SLOTS; slots
 * grp_off[w] .. grp_off[w+1] belong to window w, the same frame may occupy slots of several
 * windows, and the per-frame Sync state (M, k) is kept per slot.  Delays are then given per
 * window.  rship_select_frames = one window over all slots. */
int rship_select_slots(rship_ctx* c, const uint32_t* idx, uint32_t n, const uint32_t* grp_off,
                       uint32_t n_grp);

/* Sums over frames (the mutex-guarded accumulations of core_private.cpp:84-85, :235-237, :248-249) follow a
 * PLAN: window w = chunks win_chunk_off[w] .. [w+1]; chunk c = plan positions chunk_off[c] .. [c+1]; position
 * j = slot plan_idx[j] (NULL: j itself).  A chunk is summed sequentially, a window is the sequential sum of
 * its chunks.  The host puts into one chunk the slots of a window whose frame-table index lies in the same
 * block of 64 and splits frames over devices at multiples of 64: the chunk sums of several devices, added in
 * order, then equal the single-device window sums bit for bit. */
int rship_set_plan(rship_ctx* c, const uint32_t* plan_idx, uint32_t plan_len, const uint32_t* chunk_off,
                   uint32_t n_chunks, const uint32_t* win_chunk_off, uint32_t n_win);

/* pre_sync's per-frame body for every (selected slot, candidate delay):
 * opt_compute_problem + opt_guess_translational_motion(P, n_hyp) + cost
 * (core_private.cpp:75-85), then the sums of the plan.  enqueue returns at once (several devices work
 * concurrently); collect waits: win_costs[n_cand][n_win], chunk_costs[n_cand][n_chunks] (either may be
 * NULL), status bits, and the debug matrices frame_costs / best_h [n_cand][n_sel] if asked for.
 * kd64 / fd64 (or NULL: the fp32 split, widened): the same delays split in fp64, for the pairs whose rows the sweep
 * recomputes from the fp64 streams -- near-static frames, |P| below ~2e-4, where fp32 inputs are not enough to follow
 * the reference's double arithmetic (core_private.cpp:19-28,45-46); collect does that before it returns. */
int rship_presync_enqueue(rship_ctx* c, const int32_t* kd, const float* fd, const int32_t* kd64, const double* fd64,
                          uint32_t n_cand, uint32_t n_hyp, uint32_t stream_base, uint64_t seed, int want_frame_costs,
                          int want_best_h);
int rship_presync_collect(rship_ctx* c, uint32_t n_cand, double* win_costs, double* chunk_costs,
                          uint32_t* flags, double* frame_costs, int32_t* best_h);
/* A BATCH of n sweeps over the same candidates, selection and plan, collected together -- the orientation sweep of the
 * reference's driver (core_testcode.cpp:216-224; BASELINE config 5), where between two sweeps only the gyro table changes
 * (rship_gyro_rates_integrate_enqueue).  After _begin the next n rship_presync_enqueue calls keep their sums in slots
 * 0 .. n-1 and nothing is waited for; _collect waits ONCE: win_costs[n][n_cand][n_win], chunk_costs[n][n_cand][n_chunks]
 * (either may be NULL), flags[n].  A sweep whose flags carry RSHIP_NEAR_STATIC has not had its near-static pairs
 * recomputed from the fp64 streams (that needs the host between two launches): the caller repeats that sweep alone. */
int rship_presync_batch_begin(rship_ctx* c, uint32_t n, uint32_t n_cand);
int rship_presync_batch_collect(rship_ctx* c, uint32_t n_cand, double* win_costs, double* chunk_costs, uint32_t* flags);
/* TEST-VARIANTS build of the library only (tools/k2_build_variant.sh testvariants -DRSSYNC_TEST_VARIANTS=1; the product
 * refuses): later sweeps also store the |residual| bit patterns they worked on, [candidate][slot][hypothesis][cap_rows]
 * (0xffffffff: no such row); _get copies the last sweep's out (dims = {candidates, slots, hypotheses, cap_rows}; out may
 * be NULL to ask for the dims).  The anchor of tests/test_gpu_fuzz.py: the winner must be the exact arg-min of the
 * residuals' lower quartile with the reference's first-wins rule (core_private.cpp:48-56), no tolerance. */
int rship_debug_residuals(rship_ctx* c, int on, uint32_t cap_rows);
int rship_debug_residuals_get(rship_ctx* c, uint32_t* out, uint64_t n_words, uint32_t dims[4]);
/* out[0] = (frame, candidate) pairs recomputed with fp64 rows so far on this context, out[1] = sweeps that needed it,
 * out[2] = GuessMotion searches (rship_init_motion, the window executor) that took their rows from the fp64 streams */
int rship_near_static_stats(rship_ctx* c, uint64_t out[3]);

/* out[k] = rows / 256 of the LMedS tile the last PreSync sweep used for size class k (class 0, the one-wave kernel: rows per
 * lane; 0: class not in the selection, or class 5).  Classes 1 .. 4 sweep in their own shape (4, 8, 16, 32) or, where the
 * largest frame of the class in the selection needs no more rows, in a sub-shape (3, 5 .. 7, 9 .. 15; 18, 20 .. 30 in eight
 * waves), class 0 above 256 tracks with 5 .. 8 rows per lane: the same bits, fewer rows swept (RSSYNC_NO_SUBSHAPES=1: always
 * the class's own). */
int rship_lmeds_shapes(rship_ctx* c, uint32_t out[6]);

/* FrameState::GuessMotion (core_private.cpp:125-128): the 200-hypothesis LMedS search for every selected
 * slot, in the fp32 tile kernel; kd/fd hold one delay per window (fp32 split), window w samples with
 * stream + w * stream_stride.  Only the winning hypothesis index per slot is kept (on the device); the
 * next rship_opt_motion -- or rship_finish_init -- recomputes the winning pair of rows in fp64,
 * M = safe_normalize(P[i0] x P[i1]), and GuessK (:130-133) k = clamp(100 / |P M|, 10, 1000).  Asynchronous. */
int rship_init_motion(rship_ctx* c, const int32_t* kd, const float* fd, const int32_t* kd64, const double* fd64, uint32_t n_hyp,
                      uint32_t stream, uint32_t stream_stride, uint64_t seed);
/* finish a pending rship_init_motion at the same delays (fp64 split) without optimising */
int rship_finish_init(rship_ctx* c, const int32_t* kd, const double* fd);
/* no-translation variant (thesis section 2.11 eq. (12)): k = clamp(100 / sqrt(sum_j |P_j|^2), 10, 1000) per slot */
int rship_init_k_simple(rship_ctx* c, const int32_t* kd, const double* fd);

/* do_opt_motion (core_private.cpp:262-296): per-frame L-BFGS on the motion
 * vector at a fixed delay per window (fd = NaN skips a window), in fp64.
 * stats (optional) = {sum of iterations, sum of evaluations, line searches whose best step was not the last} */
int rship_opt_motion(rship_ctx* c, const int32_t* kd, const double* fd, uint64_t* stats);
/* the same with per-slot diagnostics: per_frame[2i] = L-BFGS iterations, [2i+1] = evaluations */
int rship_opt_motion_detail(rship_ctx* c, const int32_t* kd, const double* fd, uint32_t* per_frame,
                            uint32_t cap);

/* FrameState::Loss (core_private.cpp:117-123) of every selected slot at n_delays delays, in fp64, summed
 * under the plan (for Sync the plan's windows are the selection's groups); kd/fd are
 * [n_delays][n_groups] (fd = NaN skips a group); with want_grad also the analytic d/d-delay that replaces
 * the central difference at :96-97,112.  flags: RSHIP_LOSS_SIMPLIFIED = the no-translation loss
 * sum_j log1p((k |P_j|)^2) (thesis section 2.11 eq. (12)) instead.  collect: [n_delays][n_win] window sums
 * and [n_delays][n_chunks] chunk sums; any pointer may be NULL. */
#define RSHIP_LOSS_SIMPLIFIED 1u
int rship_loss_enqueue(rship_ctx* c, const int32_t* kd, const double* fd, uint32_t n_delays, int want_grad,
                       uint32_t flags);
int rship_loss_collect(rship_ctx* c, uint32_t n_delays, double* win_loss, double* win_grad,
                       double* chunk_loss, double* chunk_grad);

/* Sync's outer loop (core_private.cpp:298-331) for the W windows (= groups) of the selection with the scalar
 * decisions taken on the device between the launches (kernels/syncloop.hpp): the host only polls a counter
 * every few iterations.  Same arithmetic as the host loop in sync_problem.cpp; used when one device holds all
 * frames and no exchange with other ranks is needed.  d0[W] in; d_out[W], iters[W], trace[W][max_outer][6] out. */
int rship_has_device_loop(void); /* 1 where rship_sync_run exists (0 in the CPU test double) */
int rship_sync_run(rship_ctx* c, const double* d0, int max_outer, double search_center, double search_radius,
                   int simplified, double* d_out, int32_t* iters, double* trace);

/* The WINDOW EXECUTOR (kernels/executor.hpp): Sync for the W windows of the selection, `repeats` chained calls each
 * (the reference driver runs four per sync point, core_testcode.cpp:314), in ONE launch scheduled on the device --
 * tasks (window, phase, frame) pulled from a queue by persistent one-wave workgroups; windows advance
 * independently.  Frames of up to 512 tracks (rship_exec_supported), every window non-empty.  Call after
 * rship_select_slots + rship_set_plan (no rship_init_motion: the search is the executor's first phase).  Window w
 * samples its call r with stream_first + r + w * stream_stride.  Out: d_out[W], cost[W] = loss at the returned
 * delay, iters[W][repeats], trace[W][trace_rows][6] with the rows of a window's calls back to back
 * (trace_rows >= repeats * max_outer).  Same bits as the chain of launches. */
int rship_exec_supported(rship_ctx* c);
/* the last rship_sync_exec of this context: out[0] queue numbers claimed (head), out[1] queue numbers pushed (tail:
 * every task and end marker ever queued), out[2] cells of the queue ring (the numbers wrap around it), out[3] waves
 * launched; zeros before the first run (and in the CPU test double) */
int rship_exec_stats(rship_ctx* c, uint32_t out[4]);
/* the spline windows of the last launches: out[0] widest frame in knots, out[1] knots per fp64 window (dynamic LDS),
 * out[2] fp32 window of the last PreSync sweep (0 = the 80 knots compiled in, else knots of dynamic LDS), out[3] its
 * candidates per workgroup, out[4] as out[2] for the last GuessMotion search, out[5] delays per pass of the trials' kernel,
 * out[6] widest frame counting only the two ends' ranges of each pair (= out[0] where the table does not know them), out[7] 0 */
int rship_window_info(rship_ctx* c, uint32_t out[8]);
int rship_sync_exec(rship_ctx* c, const double* d0, int repeats, uint32_t stream_first, uint32_t stream_stride, uint64_t seed,
                    int max_outer, double search_center, double search_radius, double* d_out, double* cost, int32_t* iters,
                    double* trace, uint32_t trace_rows);

/* per-slot state in selection order: M[3n], k[n] */
int rship_get_motion(rship_ctx* c, double* M, double* k, uint32_t cap, uint32_t* n);
int rship_set_motion(rship_ctx* c, const double* M, const double* k, uint32_t n);

/* Frames given as tracked pixel positions instead of rays (the reference driver's step upstream of
 * SetTrackResult, core_testcode.cpp:63-95,135-158) are rship_pack_frame records with is_pixels = 1:
 * undistortion, normalisation, row time and knot offset run in fp64 inside the packing kernel. */

/* Native exchange for frame-sharded multi-GPU runs: an RCCL communicator owned by the context
 * (librccl is opened with dlopen on first use: no link-time dependency), one rank per process.
 * unique_id: rank 0 fills 128 bytes, the host distributes them (MPI, torch.distributed, a file...),
 * every rank then calls init.  allreduce sums n doubles in place over all ranks (host buffer). */
int rship_rccl_preflight(rship_ctx* c);        /* 0 = librccl and every entry point used are there (no communication) */
const char* rship_rccl_library(rship_ctx* c);   /* which librccl the entry points come from ("" before the first use) */
int rship_rccl_unique_id(rship_ctx* c, void* id128);
int rship_rccl_init(rship_ctx* c, const void* id128, int rank, int world);
int rship_rccl_allreduce(rship_ctx* c, double* buf, uint64_t n);
int rship_rccl_shutdown(rship_ctx* c); /* ncclCommDestroy; collective */
/* with a communicator, rship_sync_run all-reduces the window sums ON THE STREAM between its kernels (two
 * ncclAllReduce per enqueued iteration, no host round trip); this many the last run enqueued */
uint64_t rship_loop_exchanges(const rship_ctx* c);
/* The same loop with a HOST exchange instead of the communicator (the solver's reduce hook: any transport): the
 * window sums are copied to the host, fn(user, sums, n) adds the other ranks' in place (0 = ok), and they are copied
 * back -- the stream is drained at every exchange, the decisions still never leave the device.  NULL switches it off. */
typedef int (*rship_loop_exchange_fn)(void* user, double* sums, uint64_t n);
int rship_set_loop_exchange(rship_ctx* c, rship_loop_exchange_fn fn, void* user);

/* debug: the packed float4 streams of one frame of the table */
int rship_debug_rays(rship_ctx* c, uint32_t frame_index, float* a4, float* b4, uint32_t cap_rays);

/* residual matrix P (fp32, row-major N x 3) of one selected frame at one delay, as the PreSync
 * kernel computes it (tests); _64: as the Sync kernels compute it, in fp64 */
int rship_debug_problem(rship_ctx* c, uint32_t sel_index, int32_t kd, float fd, float* P,
                        float* dP, uint32_t cap_rows);
int rship_debug_problem64(rship_ctx* c, uint32_t sel_index, int32_t kd, double fd, double* P,
                          double* dP, uint32_t cap_rows);

/* the wave-level exact selection used by the LMedS kernel, on caller data (tests): for each of
 * n_problems rows of n (<= 2048) non-negative floats, out[2i] = bit pattern of the kq-th smallest
 * (0-based) if more than kq values lie below upper[i] (NULL = +inf), else 0xffffffff;
 * out[2i+1] = how many lie below the bound */
int rship_debug_select(rship_ctx* c, const float* vals, uint32_t n_problems, uint32_t n, uint32_t kq,
                       const float* upper, uint32_t* out);

/* the fp64 building blocks of the Sync kernels on caller data (tests: the CPU stand-in must give the same bits):
 * op 0 a / b, 1 sqrt(a), 2 log1p_rcp_f64(a) -> out[2i] = value, out[2i+1] = 1 / (1 + a), 3 fma(a, b, a),
 * 4 the kernels' wave sum of every block of 64 values of a -> out[block] */
int rship_debug_math64(rship_ctx* c, int op, const double* a, const double* b, double* out, uint32_t n);
/* GuessMotion's pending winners (hypothesis index per slot of the selection, local slot order) after
 * rship_init_motion: copy them out (get) and / or replace them (set); either may be NULL */
int rship_debug_init_h(rship_ctx* c, int32_t* get, const int32_t* set, uint32_t n);

/* per-wave trip counts of the LMedS tile kernel's stage C (builds with -DRSSYNC_K2_COUNTERS=1; zeros otherwise):
 * see kernels/lmeds.hpp for the meaning of the 16 slots */
int rship_debug_k2_counters(rship_ctx* c, uint64_t out[16], int reset);

/* HIP-event timing of every launch, accumulated per kernel kind */
int rship_profile_enable(rship_ctx* c, int on);
int rship_profile_get(rship_ctx* c, int kind, uint64_t* launches, double* total_ms);
int rship_profile_reset(rship_ctx* c);

#ifdef __cplusplus
}
#endif
#endif
