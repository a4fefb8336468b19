// executor.hpp -- the WINDOW EXECUTOR: Sync for many small windows in ONE launch, scheduled on the device.
// Part of the single HIP translation unit rssync_kernels.hip (included there, after syncloop.hpp).
//
// The reference driver's workload (core_testcode.cpp:270-316) is a hundred windows of ~60 frames x ~130 tracks,
// four chained Sync calls each.  As a chain of launches (rship_sync_run) all windows move in lock-step: every launch
// is as long as the slowest frame of the slowest window, most windows have long finished while the last ones still
// iterate, and a call cannot begin before the previous one has ended everywhere -- the device mostly waits.
//
// Here the unit of work is a TASK = (window, phase, frame slot), executed by one wave, and the phases of a window
//     INIT (GuessMotion's 200-hypothesis search)  ->  MOTION (finish init, per-frame L-BFGS at d)  ->
//     GRAD (loss + analytic derivative at x0)  ->  TRIALS (line-search losses)  ->  step  ->  MOTION ... ->
//     next call's INIT ...  ->  FINAL (loss at the returned delay)
// follow each other without the host: a grid of persistent one-wave workgroups pops tasks from a queue in global
// memory; the wave that finishes the LAST task of a window's phase (an atomic counter) adds that window's sums in the
// plan's order, runs the scalar decisions of syncloop.hpp and pushes the window's next tasks.  Windows never wait
// for each other, a slow frame only delays its own window, and the four calls of a sync point chain per window.
//
// Every number is the one the launch chain computes: the task bodies are the kernels' own (lmeds_small_body,
// opt_motion64_body in the one-wave-per-frame shape, and the loss in the four-wave association of loss64_kernel,
// evaluated by one wave), sums and decisions are the same functions.  tests/test_gpu_executor.py: identical bits.
//
// Termination: tasks never wait for other tasks, so every pushed task ends; a wave leaves when all windows are done
// (or a watchdog has tripped: a wave that polls an empty queue spin_limit times raises the abort flag, everybody
// leaves, the host reports the failure).  A grid larger than the chip holds is harmless: the workgroups that are
// resident can finish everything by themselves.
#pragma once

namespace {

constexpr int kExecMaxCalls = 8;
enum : int { kPhInit = 0, kPhMotion = 1, kPhGrad = 2, kPhTrials = 3, kPhFinal = 4, kPhDone = 5 };

struct ExecWin {
    SyncWin s;               // the loop state of the current call
    uint32_t slot0, n_slots; // the window's slots
    int phase, call;
    uint32_t remaining;      // tasks of the current phase still running
    int trace_base;          // trace rows written by the earlier calls
    double cost;             // loss at the returned delay (core_private.cpp:333), after the last call
    int iters_call[kExecMaxCalls];
};

struct ExecParams {
    SyncLoopParams lp;       // the constants of the loop (ts, c, momentum, window, iteration cap, fs)
    LmedsParams init;        // GuessMotion's search: kd / fd = in_kd / in_fd [W], win_stream, best_h = init_h
    Motion64Params mo;       // kd / fd = mo_kd / mo_fd [W]
    Loss64Params lo;         // rays, frames, selection, table, M, k
    ExecWin* win;
    uint32_t n_win, n_sel;
    const uint32_t* grp;     // slot -> window
    int32_t* in_kd; float* in_fd;    // [W] delay of the search (fp32 split)
    uint32_t* win_stream;            // [W] sampler stream of the window's current call
    int32_t* mo_kd; double* mo_fd;   // [W]
    int32_t* lg_kd; double* lg_fd;   // [W]
    int32_t* tr_kd; double* tr_fd;   // [10][W]
    double* part;                    // [10][n_sel]: GRAD rows 0 (loss), 1 (derivative); TRIALS rows 0..9; FINAL row 0
    const uint32_t* chunk_off;       // the plan (windows = groups, positions = slots)
    const uint32_t* win_chunk_off;
    double* trace;                   // [W][trace_rows][6]
    uint32_t trace_rows;
    uint32_t stream_first, stream_stride; // stream of window w in call r: stream_first + r + w * stream_stride
    int repeats;
    uint32_t* q;                     // task queue: slot + 1, 0 = empty
    uint32_t q_mask;
    uint32_t* q_head; uint32_t* q_tail;
    uint32_t* done;                  // windows finished
    uint32_t* abort_flag;
    uint32_t spin_limit;
};

// host split_delay (sync_problem.cpp) on the device: delay * fs = kd + fd, fd in [0, 1) as fp32
__device__ __forceinline__ void split32_dev(double delay, double fs, int32_t* kd, float* fd) {
#pragma clang fp contract(off)
    const double D = delay * fs;
    if (!(fabs(D) <= 1.79769313486231570e308)) { *kd = 0; *fd = 0.f; return; }
    double fl = floor(D);
    float f = (float)(D - fl);
    if (f >= 1.0f) { f = 0.f; fl += 1.0; }
    const double kClamp = (double)(1 << 29);
    if (fl > kClamp) fl = kClamp;
    if (fl < -kClamp) fl = -kClamp;
    *kd = (int32_t)fl;
    *fd = f;
}

__device__ __forceinline__ uint32_t exec_pop(const ExecParams& p) {
    uint32_t idx = 0;
    if (threadIdx.x == 0) idx = __hip_atomic_fetch_add(p.q_head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    idx = uniform_u32(idx);
    uint32_t* cell = p.q + (idx & p.q_mask);
    for (uint32_t spins = 0;; ++spins) {
        const uint32_t v = uniform_u32(__hip_atomic_load(cell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
        if (v) {
            if (threadIdx.x == 0) __hip_atomic_store(cell, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return v - 1u;
        }
        if (uniform_u32(__hip_atomic_load(p.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= p.n_win) return 0xffffffffu;
        if (uniform_u32(__hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return 0xffffffffu;
        if (spins > p.spin_limit) {
            if (threadIdx.x == 0) __hip_atomic_store(p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 0xffffffffu;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

// the window's slots as tasks of its (already published) next phase
__device__ __forceinline__ void exec_push(const ExecParams& p, ExecWin& w) {
    if (threadIdx.x == 0) __hip_atomic_store(&w.remaining, w.n_slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence(); // the window's new state and delays before its tasks
    uint32_t base = 0;
    if (threadIdx.x == 0) base = __hip_atomic_fetch_add(p.q_tail, w.n_slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = uniform_u32(base);
    for (uint32_t i = threadIdx.x; i < w.n_slots; i += 64)
        __hip_atomic_store(p.q + ((base + i) & p.q_mask), w.slot0 + i + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// FrameState::Loss (and its analytic d/d-delay) of one slot at one delay by ONE wave, in the association of
// loss64_kernel for frames of up to 256 tracks: thread t of that kernel's four waves holds row t, each wave is
// summed by wave_sum_f64, the four wave sums are added left to right.
template <bool GRAD>
__device__ __forceinline__ void loss64_wave(const Loss64Params& q, uint32_t sf, int kd, double fd, d4* s_win, double& L_out, double& G_out) {
    const int lane = threadIdx.x;
    const FrameRec fr = q.frames[q.sel[sf]];
    const uint32_t N = fr.n;
    const d3 Mv = d3{q.M[3 * sf], q.M[3 * sf + 1], q.M[3 * sf + 2]};
    const double inv_s = rs::loss_inv_s(false, q.k[sf], Mv);
    Spline64 sp;
    sp.g = q.coef;
    sp.n = q.n_knots;
    __syncthreads(); // the window's previous users are done
    frame_window64(sp, s_win, fr, kd);
    __syncthreads();
    const int base = fr.base_knot + kd;
    double Lw[4], Gw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        Lw[w] = 0.0;
        Gw[w] = 0.0;
        if ((uint32_t)w * 64u < N) { // (a wave without rows sums zeros to zero)
            const uint32_t row = (uint32_t)w * 64u + lane;
            double L = 0.0, G = 0.0;
            if (row < N) {
                const size_t idx = (size_t)fr.off + row;
                d3 P, dP;
                residual_row64<GRAD>(sp, q.rays.q0[idx], q.rays.q1[idx], q.rays.q2[idx], q.rays.q3[idx], base, fd, P, dP);
                rs::loss_row<GRAD, false>(P, dP, Mv, inv_s, L, G);
            }
            Lw[w] = wave_sum_f64(L);
            if (GRAD) Gw[w] = wave_sum_f64(G);
        }
    }
    L_out = Lw[0] + Lw[1] + Lw[2] + Lw[3];
    G_out = GRAD ? (Gw[0] + Gw[1] + Gw[2] + Gw[3]) * q.fs : 0.0;
}

// the window's sum of row r of part[] in the plan's order (window_sums / plan_sum_kernel: chunks sequentially, then
// the chunks in order); lane r < rows computes row r
__device__ __forceinline__ double exec_window_sum(const ExecParams& p, uint32_t w, uint32_t r) {
#pragma clang fp contract(off)
    const double* row = p.part + (size_t)r * p.n_sel;
    double tot = 0.0;
    for (uint32_t c = p.win_chunk_off[w]; c < p.win_chunk_off[w + 1]; ++c) {
        double acc = 0.0;
        for (uint32_t j = p.chunk_off[c]; j < p.chunk_off[c + 1]; ++j) acc += row[j];
        tot += acc;
    }
    return tot;
}

__device__ __forceinline__ double lane_bcast_d(double v, int lane) { return read_lane_d(v, lane); }

// the decisions of window w after the last task of its phase; executed by one whole wave
__device__ __forceinline__ void exec_decide(const ExecParams& p, uint32_t w) {
#pragma clang fp contract(off)
    ExecWin& W = p.win[w];
    SyncWin& s = W.s;
    const int lane = threadIdx.x;
    const int ph = W.phase;
    if (ph == kPhInit) {
        if (lane == 0) {
            split64_dev(s.d, p.lp.fs, &p.mo_kd[w], &p.mo_fd[w]);
            W.phase = kPhMotion;
        }
    } else if (ph == kPhMotion) {
        if (lane == 0) {
            split64_dev(s.x0, p.lp.fs, &p.lg_kd[w], &p.lg_fd[w]); // loss + derivative at x0 = d - 0.3 v (:298-299)
            W.phase = kPhGrad;
        }
    } else if (ph == kPhGrad) {
        const double sum = lane < 2 ? exec_window_sum(p, w, (uint32_t)lane) : 0.0;
        const double l1 = lane_bcast_d(sum, 0), g1 = lane_bcast_d(sum, 1);
        if (lane == 0) {
            grad_decide(p.lp, s, l1, g1);
            trial_delays(p.lp, s, p.tr_kd + w, p.tr_fd + w, p.n_win);
            W.phase = kPhTrials;
        }
    } else if (ph == kPhTrials) {
        double lt[kMaxBt];
        {
            const bool mine = lane < kMaxBt && trial_wanted(s, lane); // only the rows that were evaluated
            const double sum = mine ? exec_window_sum(p, w, (uint32_t)lane) : 0.0;
#pragma unroll
            for (int i = 0; i < kMaxBt; ++i) lt[i] = lane_bcast_d(sum, i);
        }
        if (lane == 0) {
            const bool stepped = step_decide(p.lp, s, lt, p.trace + ((size_t)w * p.trace_rows + W.trace_base) * 6, 6);
            if (!stepped) {
                trial_delays(p.lp, s, p.tr_kd + w, p.tr_fd + w, p.n_win); // the rest of the trials
            } else if (s.active) {
                split64_dev(s.d, p.lp.fs, &p.mo_kd[w], &p.mo_fd[w]); // :311 at the new delay
                W.phase = kPhMotion;
            } else {
                W.iters_call[W.call] = s.iters;
                W.trace_base += s.iters;
                if (W.call + 1 < p.repeats) { // the next Sync call of this sync point starts where this one ended (core_testcode.cpp:314)
                    W.call += 1;
                    const double d = s.d;
                    s = SyncWin{};
                    s.d = d;
                    s.active = 1;
                    s.hit = -1;
                    s.nf = p.lp.nf_fixed ? p.lp.nf_fixed : kHalfBt;
                    s.x0 = s.d - p.lp.delay_b * s.v;
                    p.win_stream[w] = p.stream_first + (uint32_t)W.call + w * p.stream_stride;
                    split32_dev(s.d, p.lp.fs, &p.in_kd[w], &p.in_fd[w]);
                    W.phase = kPhInit;
                } else {
                    split64_dev(s.d, p.lp.fs, &p.lg_kd[w], &p.lg_fd[w]); // :333
                    W.phase = kPhFinal;
                }
            }
        }
    } else if (ph == kPhFinal) {
        const double sum = lane < 1 ? exec_window_sum(p, w, 0u) : 0.0;
        if (lane == 0) {
            W.cost = sum;
            W.phase = kPhDone;
        }
        __threadfence();
        if (lane == 0) __hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    exec_push(p, W);
}

// LDS of one executor wave: the search's tile and fp32 window, the fp64 window and L-BFGS history
template <int RPT>
struct ExecLds {
    LmedsSmallLds<RPT> small;
    MotionLds<1> mo;
};

template <int RPT> // rows per lane of the one-wave kernels: frames of up to 64 * RPT tracks
__global__ __launch_bounds__(64) void sync_exec_kernel(ExecParams p) {
    __shared__ ExecLds<RPT> lds;
    const int lane = threadIdx.x;
    for (;;) {
        const uint32_t slot = exec_pop(p);
        if (slot == 0xffffffffu) break;
        const uint32_t w = p.grp[slot];
        ExecWin& W = p.win[w];
        const int ph = W.phase;
        if (ph == kPhInit) {
            lmeds_small_body<RPT, 1>(p.init, slot, 0u, lds.small);
        } else if (ph == kPhMotion) {
            opt_motion64_body<RPT, 1>(p.mo, slot, lds.mo);
        } else if (ph == kPhGrad) {
            double L, G;
            loss64_wave<true>(p.lo, slot, p.lg_kd[w], p.lg_fd[w], lds.mo.win, L, G);
            if (lane == 0) { p.part[slot] = L; p.part[(size_t)p.n_sel + slot] = G; }
        } else if (ph == kPhTrials) {
            for (int i = 0; i < kMaxBt; ++i) {
                const double fd = p.tr_fd[(size_t)i * p.n_win + w];
                if (fd != fd) continue; // not asked for
                double L, G;
                loss64_wave<false>(p.lo, slot, p.tr_kd[(size_t)i * p.n_win + w], fd, lds.mo.win, L, G);
                if (lane == 0) p.part[(size_t)i * p.n_sel + slot] = L;
            }
        } else if (ph == kPhFinal) {
            double L, G;
            loss64_wave<false>(p.lo, slot, p.lg_kd[w], p.lg_fd[w], lds.mo.win, L, G);
            if (lane == 0) p.part[slot] = L;
        }
        __threadfence(); // this task's results before the count
        uint32_t left = 0;
        if (lane == 0) left = __hip_atomic_fetch_sub(&W.remaining, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        left = uniform_u32(left);
        if (left == 1u) {
            __threadfence(); // everybody's results before the sums
            exec_decide(p, w);
        }
    }
}

} // namespace
