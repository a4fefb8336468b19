// syncloop.hpp -- Sync's outer loop (core_private.cpp:298-331) kept on the device between kernel launches.
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
//
// One outer iteration is: per-frame motion optimisation at d; loss + gradient at x0 = d - 0.3 v; up to
// ten line-search losses; a handful of scalar decisions per window.  With the decisions on the host
// every iteration costs two or three stream synchronisations (~25 us each, more than the kernels of a
// 60-frame window).  Here the decisions are two small kernels between the launches (one workgroup per
// window, which first adds that window's per-slot sums in the plan's association, then lets thread 0
// run the scalar logic), and the host only looks at a counter of still-active windows every few
// iterations.  An iteration is five launches: motion, loss + gradient, decisions, trials, decisions.  The arithmetic of the decisions is the host loop's (sync_problem.cpp: sync_windows),
// operation for operation, with contraction off: both paths return the same bits.
#pragma once

namespace {

constexpr int kMaxBt = 10, kHalfBt = 5;

struct SyncWin { // per window
    double d, v;     // delay, momentum (delay_v, core_private.cpp:261)
    double x0;       // d - 0.3 v of this iteration
    double l1, g1;   // loss and d loss / d delay at x0
    int active, conv, hit, iters;
    int phase; // 1: the first nf trials of this iteration found nothing, the rest are evaluated in the next launch
    int nf;    // trials evaluated first (sync_step_kernel)
    int hit_prev; // index the previous search stopped at
};

struct SyncLoopParams {
    SyncWin* win;          // [W]
    uint32_t n_win;
    uint32_t win0, win1;   // the launch covers windows win0 .. win1 (a group of windows on its own stream)
    const double* part;    // per-slot sums of the launch just finished: [rows][n_sel]
    uint32_t n_sel, rows;
    // the plan (windows = groups): chunk_off over slots (identity positions), win_chunk_off
    const uint32_t* chunk_off;
    const uint32_t* win_chunk_off;
    double* chunk_tmp;     // [W][rows][max chunks per window]: scratch
    uint32_t chunk_stride; // max chunks per window
    // delays for the launches that follow
    int32_t* mo_kd; double* mo_fd;   // motion: [W]
    int32_t* lg_kd; double* lg_fd;   // loss + gradient: [W]
    int32_t* tr_kd; double* tr_fd;   // trials: [10][W]
    double fs;
    double ts[11];         // line-search step sizes t0 * decay^i (backtrack.cpp:7-12), computed by the host
    double c_armijo, delay_b, search_center, search_radius;
    int it, max_outer;
    int nf_fixed;          // 0: adaptive (below); k: always the first k trials first (tests: makes windows wait)
    // The smallest number of trials a line search's first launch evaluates.  Small windows: kHalfBt -- a trial costs
    // microseconds there and a window that has to wait for its later trials loses a whole iteration.  Large problems
    // (a trial launch of 4096 x 2048 ray pairs is 0.06 ms of fp64 work PER TRIAL): 1 -- exactly as many as the later
    // of the last two searches needed.  Only the batching depends on it, never a result.
    int nf_floor;
    // Frames sharded over ranks (one process per GPU, the library's RCCL communicator): the window sums of a launch are
    // written to ext_sums[row][window] by sync_sums_kernel, all-reduced over the ranks ON THE STREAM (ncclAllReduce
    // between the kernels, no host round trip), and the decision kernels read them there instead of adding the
    // per-slot values themselves.  Every rank sees the same sums and takes the same decisions.  Null: one rank.
    double* ext_sums;
    int* n_active;         // [max_outer]: windows still active after iteration i
    double* trace;         // [max_outer][W][6]: row k of window w is its k-th outer iteration
};


__device__ __forceinline__ void split64_dev(double delay, double fs, int32_t* kd, double* fd) {
#pragma clang fp contract(off)
    if (delay != delay) { *kd = 0; *fd = delay; return; } // NaN = window switched off
    const double D = delay * fs;
    const double kClamp = (double)(1 << 29);
    if (!(fabs(D) <= 1.79769313486231570e308)) { *kd = 0; *fd = 0.0; return; }
    const double fl = floor(D);
    if (fl > kClamp) { *kd = 1 << 29; *fd = 0.0; return; }
    if (fl < -kClamp) { *kd = -(1 << 29); *fd = 0.0; return; }
    *kd = (int32_t)fl;
    *fd = D - fl;
}

// window sums of p.part rows under the plan: chunks sequentially, then chunks in order (plan_sum_kernel's
// association); s_tot[r] for r < rows, valid for every thread after the call.  Small windows (the reference's
// 60-frame windows) are first copied to LDS with coalesced loads, so that the sequential additions do not
// wait for a global load each.
constexpr uint32_t kStageDoubles = 2048; // 16 KB
__device__ __forceinline__ void window_sums(const SyncLoopParams& p, uint32_t w, double* s_tot, double* s_stage) {
    const uint32_t c0 = p.win_chunk_off[w], c1 = p.win_chunk_off[w + 1], nc = c1 - c0;
    const uint32_t j_lo = p.chunk_off[c0], j_hi = p.chunk_off[c1], span = j_hi - j_lo; // the window's slots
    // a fixed region per window (not rows-dependent): groups of windows run these kernels side by side on their own
    // streams, one with 2 rows while another has 10
    double* tmp = p.chunk_tmp + (size_t)w * (2 * kMaxBt) * p.chunk_stride;
    const bool staged = p.rows * span <= kStageDoubles;
    if (staged) {
        for (uint32_t e = threadIdx.x; e < p.rows * span; e += blockDim.x)
            s_stage[e] = p.part[(size_t)(e / span) * p.n_sel + j_lo + e % span];
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < p.rows * nc; e += blockDim.x) {
        const uint32_t r = e / nc, c = c0 + e % nc;
        double acc = 0.0;
        uint32_t j = p.chunk_off[c];
        const uint32_t j1 = p.chunk_off[c + 1];
        if (staged) {
            const double* row = s_stage + (size_t)r * span - j_lo;
            for (; j + 8 <= j1; j += 8) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = row[j + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += v[q];
            }
            for (; j < j1; ++j) acc += row[j];
        } else {
            // a chunk has at most 64 slots (the host's kChunk): all of them in flight at once, then the
            // additions in slot order -- one memory round trip per chunk instead of eight
            const double* row = p.part + (size_t)r * p.n_sel;
            const uint32_t cnt = j1 - j;
            if (cnt <= 64) {
                double v[64];
#pragma unroll
                for (int q = 0; q < 64; ++q) v[q] = (uint32_t)q < cnt ? row[j + q] : 0.0;
#pragma unroll
                for (int q = 0; q < 64; ++q)
                    if ((uint32_t)q < cnt) acc += v[q];
            } else {
                for (; j < j1; ++j) acc += row[j];
            }
        }
        tmp[(size_t)r * p.chunk_stride + (c - c0)] = acc;
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < p.rows; r += blockDim.x) {
        // the chunk sums in order; sixteen loads in flight (one at a time, a 64-chunk window waited 64 L2 round trips)
        const double* t = tmp + (size_t)r * p.chunk_stride;
        double acc = 0.0;
        uint32_t c = 0;
        for (; c + 16 <= nc; c += 16) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = t[c + q];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += v[q];
        }
        for (; c < nc; ++c) acc += t[c];
        s_tot[r] = acc;
    }
    __syncthreads();
}

// stage 0: before the first iteration -- delays of the first motion and loss+gradient launches
__global__ __launch_bounds__(64) void sync_begin_kernel(SyncLoopParams p) {
#pragma clang fp contract(off)
    const uint32_t w = p.win0 + blockIdx.x * 64 + threadIdx.x;
    if (w >= p.win1) return;
    SyncWin& s = p.win[w];
    s.x0 = s.active ? s.d - p.delay_b * s.v : __builtin_nan("");
    split64_dev(s.active ? s.d : __builtin_nan(""), p.fs, &p.mo_kd[w], &p.mo_fd[w]);
    split64_dev(s.x0, p.fs, &p.lg_kd[w], &p.lg_fd[w]);
}

// Which trials a launch evaluates for a window.  The ten trials of a line search (backtrack.cpp:7-11) are
// evaluated in ONE launch per outer iteration: the first nf of them, nf = one more than the larger index at which the window's
// last two searches stopped, at least nf_floor (the step scale barely changes between iterations).  A window whose
// search finds nothing among them waits one iteration (phase 1: no motion, no gradient launch for it) in which
// the launch evaluates the remaining trials; then it steps.  The first trial that satisfies the Armijo test is
// taken in trial order either way -- what the sequential loop returns.
__device__ __forceinline__ bool trial_wanted(const SyncWin& s, int i) {
    if (!s.active) return false;
    return s.phase == 0 ? i < s.nf : i >= s.nf;
}

// after loss + gradient: the window's sums l1, g1 (backtrack.cpp:4) and how many trials go first
__device__ __forceinline__ void grad_decide(const SyncLoopParams& p, SyncWin& s, double l1, double g1) {
#pragma clang fp contract(off)
    if (s.active && s.phase == 0) { // (a waiting window keeps the loss and gradient of the iteration it waits in)
        s.l1 = l1;
        s.g1 = g1;
        s.hit = -1;
        if (s.iters == 0 && !p.nf_fixed) {
            // a call's first search has no history: steps of a millisecond and more have never been accepted, so the
            // trials up to the first shorter one (and one more) go into the first launch.  Only the batching
            // depends on this guess, never the result.
            int i = 0;
            while (i < kMaxBt - 1 && p.ts[i] * fabs(s.g1) >= 1e-3) ++i;
            const int want = i + 2;
            s.nf = want < kHalfBt ? kHalfBt : (want > kMaxBt ? kMaxBt : want);
        }
    }
}
// the delays of the trials the window wants evaluated next: tr_kd / tr_fd [10][stride], NaN for the others
__device__ __forceinline__ void trial_delays(const SyncLoopParams& p, const SyncWin& s, int32_t* tr_kd, double* tr_fd, size_t stride) {
#pragma clang fp contract(off)
    for (int i = 0; i < kMaxBt; ++i) {
        const double td = trial_wanted(s, i) ? s.x0 - p.ts[i] * s.g1 : __builtin_nan("");
        split64_dev(td, p.fs, &tr_kd[(size_t)i * stride], &tr_fd[(size_t)i * stride]);
    }
}

// rank mode: this rank's part of the window sums of the launch just finished -> ext_sums[row][window]
__global__ __launch_bounds__(kBlock) void sync_sums_kernel(SyncLoopParams p) {
#pragma clang fp contract(off)
    __shared__ double s_tot[2 * kMaxBt];
    __shared__ double s_stage[kStageDoubles];
    const uint32_t w = blockIdx.x + p.win0;
    window_sums(p, w, s_tot, s_stage);
    for (uint32_t r = threadIdx.x; r < p.rows; r += blockDim.x) p.ext_sums[(size_t)r * p.n_win + w] = s_tot[r];
}
// the window's sums for the decision kernels: added here, or (rank mode) read where the all-reduce left them
__device__ __forceinline__ void decision_sums(const SyncLoopParams& p, uint32_t w, double* s_tot, double* s_stage) {
    if (p.ext_sums) {
        for (uint32_t r = threadIdx.x; r < p.rows; r += blockDim.x) s_tot[r] = p.ext_sums[(size_t)r * p.n_win + w];
        __syncthreads();
    } else {
        window_sums(p, w, s_tot, s_stage);
    }
}

// stage G: after loss + gradient (rows: loss, gradient) -- the delays of the trial launch
__global__ __launch_bounds__(kBlock) void sync_grad_kernel(SyncLoopParams p) {
#pragma clang fp contract(off)
    __shared__ double s_tot[2 * kMaxBt];
    __shared__ double s_stage[kStageDoubles];
    const uint32_t w = blockIdx.x + p.win0;
    decision_sums(p, w, s_tot, s_stage);
    if (threadIdx.x != 0) return;
    SyncWin& s = p.win[w];
    grad_decide(p, s, s_tot[0], s_tot[1]);
    trial_delays(p, s, p.tr_kd + w, p.tr_fd + w, p.n_win);
}

// the first trial of rows [b0, b1) that satisfies the Armijo test (backtrack.cpp:9)
__device__ __forceinline__ void armijo(const SyncLoopParams& p, SyncWin& s, const double* lt, int b0, int b1) {
#pragma clang fp contract(off)
    if (!s.active || s.hit >= 0) return;
    const double m = s.g1 * s.g1;
    for (int i = b0; i < b1; ++i) {
        if (s.l1 - lt[i] >= p.ts[i] * p.c_armijo * m) {
            s.hit = i;
            break;
        }
    }
}

// after the trials: Armijo on the rows just evaluated (lt[10]); then either the window waits for its remaining
// trials (returns false) or it steps (core_private.cpp:298-305): momentum, delay, the stopping rules (:316-328) and
// the trace row, written at rows + iteration * row_stride.  Leaves x0 of the next iteration in s.x0.
__device__ __forceinline__ bool step_decide(const SyncLoopParams& p, SyncWin& s, const double* lt, double* rows, size_t row_stride) {
#pragma clang fp contract(off)
    bool step_now = false;
    if (s.active) {
        if (s.phase == 0) {
            armijo(p, s, lt, 0, s.nf);
            if (s.hit >= 0 || s.nf >= kMaxBt) step_now = true;
            else s.phase = 1; // nothing among the first nf: the others are evaluated next
        } else {
            armijo(p, s, lt, s.nf, kMaxBt);
            step_now = true;
        }
    }
    if (step_now) {
        const double v = s.l1, g = s.g1;
        // never satisfied: t0 * decay^max_bt, untested (backtrack.cpp:11-12)
        const double t = s.hit >= 0 ? p.ts[s.hit] : p.ts[kMaxBt];
        const int trials = s.hit >= 0 ? s.hit + 1 : kMaxBt;
        const double step = -t * g;
        s.v = p.delay_b * s.v + step; // :301
        s.d += s.v;                   // :302
        const double step_size = fabs(step);
        double* row = rows + (size_t)s.iters * row_stride;
        row[0] = s.d; row[1] = step; row[2] = v; row[3] = g; row[4] = t; row[5] = (double)trials;
        s.iters += 1;
        s.phase = 0;
        // as many as the later of the last two searches needed, at least nf_floor (five for small windows, one where
        // a trial is expensive: SyncLoopParams)
        const int last = s.hit >= 0 ? s.hit : kMaxBt - 1;
        const int want = (last > s.hit_prev ? last : s.hit_prev) + 1;
        s.hit_prev = last;
        s.nf = p.nf_fixed ? p.nf_fixed : (want < p.nf_floor ? p.nf_floor : (want > kMaxBt ? kMaxBt : want));
        if (step_size < 1e-4) s.conv++; else s.conv = 0;                            // :316-320
        bool stop = s.conv > 5;                                                     // :322-324
        if (!stop && fabs(s.d - p.search_center) > p.search_radius) stop = true;    // :326-328
        if (stop || s.iters == p.max_outer) s.active = 0;                           // :309
    }
    s.x0 = s.active ? s.d - p.delay_b * s.v : __builtin_nan("");
    return step_now;
}

// stage S: after the trials -- the decision above; the delays of the next iteration's launches
__global__ __launch_bounds__(kBlock) void sync_step_kernel(SyncLoopParams p) {
#pragma clang fp contract(off)
    __shared__ double s_tot[2 * kMaxBt];
    __shared__ double s_stage[kStageDoubles];
    const uint32_t w = blockIdx.x + p.win0;
    decision_sums(p, w, s_tot, s_stage);
    if (threadIdx.x != 0) return;
    SyncWin& s = p.win[w];
    step_decide(p, s, s_tot, p.trace + (size_t)w * 6, (size_t)p.n_win * 6); // [iteration of the window][window][6]
    if (s.active) atomicAdd(&p.n_active[p.it], 1);
    const bool go = s.active && s.phase == 0; // a waiting window has no motion and no gradient launch
    split64_dev(go ? s.d : __builtin_nan(""), p.fs, &p.mo_kd[w], &p.mo_fd[w]);
    split64_dev(go ? s.x0 : __builtin_nan(""), p.fs, &p.lg_kd[w], &p.lg_fd[w]);
}

} // namespace
