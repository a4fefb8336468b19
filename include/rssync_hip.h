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
 *    int32 knot count and fd an fp32 fraction in [0,1): absolute times are
 *    never rounded to fp32;
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

/* one record of the device frame table (32 bytes) */
typedef struct rship_frame {
    uint32_t ray_offset; /* first ray of the frame in the two float4 streams */
    uint32_t n_rays;
    int32_t base_knot; /* floor(min over rays of (ts - start) * fs) */
    float tmin, tmax;  /* min / max of the per-ray offsets ta, tb */
    uint32_t reserved;
    int64_t id; /* caller's frame number (keys the hypothesis sampler) */
} rship_frame;

/* status bits reported by the LMedS kernel; the host turns them into the
 * reference's panic messages (core_private.cpp:76-83) */
#define RSHIP_BAD_P 1u
#define RSHIP_BAD_M 2u
#define RSHIP_BAD_R 4u
#define RSHIP_BAD_RHO 8u

/* kernel kinds for rship_profile_get */
#define RSHIP_K_LMEDS 0  /* PreSync tile kernel */
#define RSHIP_K_LOSS 1   /* residual + robust loss (+ analytic d/d-delay) */
#define RSHIP_K_MOTION 2 /* per-frame motion L-BFGS */
#define RSHIP_K_REDUCE 3 /* over-frames sums */
#define RSHIP_K_INIT 4   /* the LMedS kernel in GuessMotion/GuessK mode (Sync start) */
#define RSHIP_K_PIXELS 5 /* pixel -> ray + row time (rship_rays_from_pixels) */
#define RSHIP_K_COUNT 6

int rship_create(rship_ctx** out, int device /* -1 = current device */);
void rship_destroy(rship_ctx* c);
const char* rship_last_error(const rship_ctx* c);
/* launch on a caller-owned hipStream_t (NULL = the context's own stream) */
int rship_set_stream(rship_ctx* c, void* hip_stream);
int rship_max_tracks(void); /* largest per-frame track count the kernels accept */
/* Behaviour switches.  RSHIP_OPT_LBFGS_REEVAL: what the restated ens::L_BFGS does when a line search's
 * best step is not its last -- 0 (default): iterate at the best step, value and gradient as the last
 * trial left them (the published LineSearch); 1: evaluate once more at the best step. */
#define RSHIP_OPT_LBFGS_REEVAL 1
int rship_set_option(rship_ctx* c, int option, int value);

/* OptData::quats (core_private.hpp:18): coefficient table built on the host by
 * the spline solver that replaces minispline.cpp:3-46 */
int rship_upload_spline(rship_ctx* c, const float* coef16, uint32_t n_knots, double sample_rate);

/* OptData::frame_data (core_private.hpp:21): all frames, packed.  Both ray pointers NULL =
 * allocate only (every frame is then filled by rship_rays_from_pixels). */
int rship_upload_frames(rship_ctx* c, const float* rays_xy4, const float* rays_zt4,
                        uint64_t total_rays, const rship_frame* table, uint32_t n_frames);

/* the frames a PreSync/Sync call works on (indices into the table; replaces the
 * frame filters at core_private.cpp:65-68, :218-219, :340-343) */
int rship_select_frames(rship_ctx* c, const uint32_t* idx, uint32_t n);
/* Batched windows: the selection is a list of SLOTS; slots
 * grp_off[w] .. grp_off[w+1] belong to window w, the same frame may occupy slots of several
 * windows, and the per-frame Sync state (M, k) is kept per slot.  Delays are then given per
 * window.  rship_select_frames = one window over all slots. */
int rship_select_slots(rship_ctx* c, const uint32_t* idx, uint32_t n, const uint32_t* grp_off,
                       uint32_t n_grp);

/* pre_sync's per-frame body for every (selected frame, candidate delay):
 * opt_compute_problem + opt_guess_translational_motion(P, n_hyp) + cost
 * (core_private.cpp:75-85).  costs[n_cand] = sum over selected frames.
 * Optional debug outputs (may be NULL): frame_costs / best_h [n_cand][n_sel]. */
int rship_presync_costs(rship_ctx* c, const int32_t* kd, const float* fd, uint32_t n_cand,
                        uint32_t n_hyp, uint32_t stream_base, uint64_t seed, double* costs,
                        uint32_t* flags, double* frame_costs, int32_t* best_h);
/* the same for several windows over one (ungrouped) selection: costs[n_cand][n_win], window w
 * summing the slots seg_idx[seg_off[w] .. seg_off[w+1]) (seg_idx NULL = the slots themselves) */
int rship_presync_window_costs(rship_ctx* c, const int32_t* kd, const float* fd, uint32_t n_cand,
                               uint32_t n_hyp, uint32_t stream_base, uint64_t seed,
                               const uint32_t* seg_idx, const uint32_t* seg_off, uint32_t n_win,
                               double* costs, uint32_t* flags, double* frame_costs, int32_t* best_h);

/* FrameState::GuessMotion + GuessK (core_private.cpp:125-133) for every
 * selected slot; kd/fd hold one delay per window, window w samples with
 * stream + w * stream_stride; results stay on the device */
int rship_init_motion(rship_ctx* c, const int32_t* kd, const float* fd, uint32_t n_hyp,
                      uint32_t stream, uint32_t stream_stride, uint64_t seed);

/* do_opt_motion (core_private.cpp:262-296): per-frame L-BFGS on the motion
 * vector at a fixed delay per window (fd = NaN skips a window).
 * stats (optional) = {sum of iterations, sum of evaluations, line searches whose best step was not the last} */
int rship_opt_motion(rship_ctx* c, const int32_t* kd, const float* fd, uint64_t* stats);
/* the same with per-slot diagnostics: per_frame[2i] = L-BFGS iterations, [2i+1] = evaluations */
int rship_opt_motion_detail(rship_ctx* c, const int32_t* kd, const float* fd, uint32_t* per_frame,
                            uint32_t cap);

/* per window, sum over its slots of FrameState::Loss at n_delays delays
 * (core_private.cpp:117-123); kd/fd are [n_delays][n_windows] (fd = NaN skips), loss/grad out
 * likewise; with grad != NULL also the analytic d/d-delay that replaces the central
 * difference at :96-97,112 */
int rship_loss(rship_ctx* c, const int32_t* kd, const float* fd, uint32_t n_delays, double* loss,
               double* grad);

/* per-slot state in selection order: M[3n], k[n] */
int rship_get_motion(rship_ctx* c, double* M, double* k, uint32_t cap, uint32_t* n);
int rship_set_motion(rship_ctx* c, const double* M, const double* k, uint32_t n);

/* A frame given as tracked pixel positions instead of rays (the reference driver's step
 * upstream of SetTrackResult, core_testcode.cpp:63-95,135-158). */
typedef struct rship_pixel_frame {
    double time_a, time_b, rows; /* times (s) of the current / next video frame; image rows */
    double lens[9];              /* ro, fx, fy, cx, cy, k1, k2, k3, k4 */
    double start, fs, base;      /* gyro grid (quats_start, sample_rate) and the frame's base knot */
    uint64_t px_offset;          /* first pair of the frame in px */
    uint32_t ray_offset, n_rays; /* where its rays live in the two float4 streams */
} rship_pixel_frame;

/* Undistort + normalise + row time + knot offset in fp64 on the device, written straight into
 * the packed ray streams uploaded by rship_upload_frames (whose slices for these frames may hold
 * anything).  px: 4 doubles per pair {xa, ya, xb, yb}.  *bad = number of non-finite outputs. */
int rship_rays_from_pixels(rship_ctx* c, const double* px, uint64_t n_pairs, const rship_pixel_frame* frames,
                           uint32_t n_frames, uint32_t* bad);

/* Native exchange for frame-sharded multi-GPU runs: an RCCL communicator owned by the context
 * (librccl is opened with dlopen on first use: no link-time dependency), one rank per process.
 * unique_id: rank 0 fills 128 bytes, the host distributes them (MPI, torch.distributed, a file...),
 * every rank then calls init.  allreduce sums n doubles in place over all ranks (host buffer). */
int rship_rccl_unique_id(rship_ctx* c, void* id128);
int rship_rccl_init(rship_ctx* c, const void* id128, int rank, int world);
int rship_rccl_allreduce(rship_ctx* c, double* buf, uint64_t n);

/* debug: the packed float4 streams of one frame of the table */
int rship_debug_rays(rship_ctx* c, uint32_t frame_index, float* a4, float* b4, uint32_t cap_rays);

/* residual matrix P (fp32, row-major N x 3) of one selected frame at one delay (tests) */
int rship_debug_problem(rship_ctx* c, uint32_t sel_index, int32_t kd, float fd, float* P,
                        float* dP, uint32_t cap_rows);

/* the wave-level exact selection used by the LMedS kernel, on caller data (tests): for each of
 * n_problems rows of n (<= 2048) non-negative floats, out[2i] = bit pattern of the kq-th smallest
 * (0-based) if more than kq values lie below upper[i] (NULL = +inf), else 0xffffffff;
 * out[2i+1] = how many lie below the bound */
int rship_debug_select(rship_ctx* c, const float* vals, uint32_t n_problems, uint32_t n, uint32_t kq,
                       const float* upper, uint32_t* out);

/* HIP-event timing of every launch, accumulated per kernel kind */
int rship_profile_enable(rship_ctx* c, int on);
int rship_profile_get(rship_ctx* c, int kind, uint64_t* launches, double* total_ms);
int rship_profile_reset(rship_ctx* c);

#ifdef __cplusplus
}
#endif
#endif
