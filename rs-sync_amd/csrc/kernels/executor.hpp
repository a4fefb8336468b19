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
// (or a watchdog has tripped: a wave that waits for a task while nobody has pushed one for watchdog_ticks of the
// constant 100 MHz clock -- 5 s -- raises the abort flag, everybody leaves, the host reports the failure).  A grid larger than the chip holds is harmless: the workgroups that are
// resident can finish everything by themselves.
#pragma once

namespace {

// -DRSSYNC_EXEC_STATS=1: wall-clock ticks (s_memrealtime, 100 MHz) per activity, summed over the waves
#ifndef RSSYNC_EXEC_STATS
#define RSSYNC_EXEC_STATS 0
#endif
#if RSSYNC_EXEC_STATS
__device__ unsigned long long g_exec_stats[48]; // [16+ph] wall ticks of a window's phases (tasks pushed -> last task done), [32+ph] their number
__device__ unsigned long long g_exec_tlast[4096]; // 0-4 task phases, 5 decide, 6 pop (waiting included), 8+ph task counts, 13 decides, 14 pops
#define EXEC_T0() const unsigned long long t0__ = __builtin_amdgcn_s_memrealtime()
#define EXEC_ADD(i, n) do { if (threadIdx.x == 0) { atomicAdd(&g_exec_stats[i], __builtin_amdgcn_s_memrealtime() - t0__); atomicAdd(&g_exec_stats[n], 1ull); } } while (0)
#else
#define EXEC_T0() do { } while (0)
#define EXEC_ADD(i, n) do { } while (0)
#endif

constexpr int kExecMaxCalls = 8;
enum : int { kPhInit = 0, kPhMotion = 1, kPhGrad = 2, kPhTrials = 3, kPhFinal = 4, kPhDone = 5 };
// a queue cell's phase byte kPhOneTrial + i: ONE line-search trial (i) of a frame of more than 512 tracks -- such a
// frame's trials are a task each (for a one-wave frame all ten are one task of ~16 us; a 600-track frame's would be ~40 us
// in a row, the longest task of its window's phase)
constexpr int kPhOneTrial = 16;

struct ExecWin {
    SyncWin s;               // the loop state of the current call
    uint32_t slot0, n_slots; // the window's slots
    int phase, call;
    uint32_t remaining;      // tasks of the current phase still running (8-byte aligned, with its pad a word of its own)
    uint32_t remaining_pad;
    int trace_base;          // trace rows written by the earlier calls
    int pad2;
    double cost;             // loss at the returned delay (core_private.cpp:333), after the last call
    int iters_call[kExecMaxCalls];
};

struct ExecBig {
    const uint32_t* win_big_off; // [W + 1]: window w's frames of more than 512 tracks = big_list[win_big_off[w] .. [w + 1])
    const uint32_t* big_list;    // their slots
    const uint32_t* slot_info;
    float* big_tile;
    double* big_P;
    uint32_t big_rows;
    uint32_t init_whole; // bit k: class k stages whole pairs only
    uint32_t init_cap[6];
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
    // task queue: a ring of 64-bit cells {lap, slot}.  Entry i (a number that only grows) lives in cell i & q_mask and
    // carries lap (i >> q_shift) + 1: the consumer that claimed number i waits for exactly that lap, so a cell never
    // has to be cleared and a value left over from an earlier lap is never mistaken for a task.
    unsigned long long* q;
    uint32_t q_mask, q_shift;
    uint32_t* q_head; uint32_t* q_tail;
    uint32_t* done;                  // windows finished
    uint32_t* abort_flag;
    // (head, tail, done and abort each sit on a 128-byte line of their own: rship_sync_exec)
    // Frames of more than 512 tracks (exec_big.hpp): slot_info[slot] = 0 for a one-wave frame, else (entry + 1) << 3 | size
    // class; entry indexes the big frames' scratch -- big_tile (fp32 tile of the search: big_rows x 5 floats per entry)
    // and mo.scratch (the rows of P in fp64).  init_cap / init_whole: the fp32 window the launch chain's search kernel
    // gives each class (knots; whole pairs only), so that a frame takes the spline path it takes there.  Null: none.
    // (in a record of its own in global memory: the kernel is at its registers' limit, and a selection of one-wave frames
    // only -- the reference's workload -- pays nothing for it)
    const struct ExecBig* big;
    unsigned long long watchdog_ticks; // s_memrealtime ticks (100 MHz) without a push by anybody before a waiting wave gives up
};

// a pointer read from a record in memory (ExecBig) is GLOBAL memory; say so, or every access through it is a flat
// instruction (both wait counters, no clustering) -- the kernel's own arguments are known to be global
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
    return (T*)(__attribute__((address_space(1))) T*)p;
}

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

// ---- hand-off protocol (MI355X_MICROARCH.md, "inter-workgroup visibility"): every word another workgroup of this
// launch may have written is loaded with sc1 (ld_m<true>) and stored with sc1 (st_m<true>); a signal -- the add to a
// window's counter, a queue cell -- goes out only after the signalling wave's s_waitcnt vmcnt(0); the wave whose
// add came last (told by the value the add returned) reads the others' results.  Words that only ATOMICS change
// (the counters) are also read with atomics: an sc1 load of such a word was seen to return a stale value for
// seconds.  No agent-scope fence anywhere: a fence writes back / invalidates a whole L2 (~3.5 us, several times that
// with eight waves per CU), and per task that was 20x the task (this file's first version: 232 ms for the workload
// the launch chain does in 30).

// How an idle wave waits: it re-reads its cell every kExecSleep x 64 cycles and looks at the two counters that
// only atomics change (windows done, abort) every kExecPollMask + 1 reads.  With a thousand idle waves the
// atomic reads compete with the working waves' own atomics (the phase counters, the queue's head and tail): every 8th
// read, 27.6 ms for the 98 sync points; every 16th, 24.1; every 256th, 21.4 (profiles/r3_executor_stats.txt) -- so the
// "done" counter is not polled at all (end markers, exec_decide) and the abort flag every 1024th read.
constexpr uint32_t kExecPollMask = 1023;
constexpr int kExecSleep = 16;
__device__ __forceinline__ uint32_t exec_pop(const ExecParams& p) {
    uint32_t idx = 0;
    if (threadIdx.x == 0) idx = __hip_atomic_fetch_add(p.q_head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    idx = uniform_u32(idx);
    const unsigned long long* cell = p.q + (idx & p.q_mask);
    const uint32_t lap = (idx >> p.q_shift) + 1u;
    uint32_t seen_tail = 0xffffffffu;
    unsigned long long t_push = __builtin_amdgcn_s_memrealtime(); // when the tail was last seen to move (or this wait began)
    for (uint32_t spins = 0;; ++spins) {
        const unsigned long long v = ld_m<true>(cell);
        const uint32_t v_lap = uniform_u32((uint32_t)(v >> 32)), v_slot = uniform_u32((uint32_t)v);
        if (v_lap == lap) return v_slot;
        if ((spins & kExecPollMask) == kExecPollMask) {
            // rarely (every ~2 ms): the counters that only atomics change, lanes 0..2 one each in one round trip
            uint32_t v3 = 0;
            if (threadIdx.x < 3)
                v3 = __hip_atomic_fetch_add(threadIdx.x == 0 ? p.done : (threadIdx.x == 1 ? p.abort_flag : p.q_tail), 0u, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t dn = (uint32_t)__builtin_amdgcn_readlane((int)v3, 0), ab = (uint32_t)__builtin_amdgcn_readlane((int)v3, 1),
                           tl = (uint32_t)__builtin_amdgcn_readlane((int)v3, 2);
            if (dn >= p.n_win) return 0xffffffffu; // (normally the end marker arrives first)
            if (ab) return 0xffffffffu;
            // the watchdog measures the TIME since the last PUSH by anybody (the constant 100 MHz clock, not a number of
            // polls whose length depends on the load), not this wave's own wait: one long window at the end of a large
            // run keeps every other wave idle for as long as it takes
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (tl != seen_tail) { seen_tail = tl; t_push = now; }
            if (now - t_push > p.watchdog_ticks) {
                if (threadIdx.x == 0) __hip_atomic_fetch_add(p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return 0xffffffffu;
            }
        }
        __builtin_amdgcn_s_sleep(kExecSleep);
    }
}

// the window's slots as tasks of its (already stored) next phase
// (a cell's low word: the task's phase in the top byte -- the consumer need not fetch the window's record to learn
// it -- and the slot below; slots are < 2^24, checked by the launcher)
// BIG: in the trial phase a frame of more than 512 tracks gets one task PER TRIAL the window wants (kPhOneTrial + i);
// `s` is the window's loop state (which trials it wants).
template <bool BIG>
__device__ __forceinline__ void exec_push(const ExecParams& p, uint32_t w, uint32_t slot0, uint32_t n_slots, int phase, const SyncWin& s) {
    uint32_t extra = 0, per_big = 0, tmask = 0, b0 = 0;
    if constexpr (BIG) {
        if (phase == kPhTrials) {
            b0 = as_global(p.big->win_big_off)[w];
            const uint32_t nb = as_global(p.big->win_big_off)[w + 1] - b0;
            if (nb) {
                for (int i = 0; i < kMaxBt; ++i) tmask |= trial_wanted(s, i) ? (1u << i) : 0u;
                const uint32_t nwant = (uint32_t)__builtin_popcount(tmask);
                if (nwant > 1u) { per_big = nwant - 1u; extra = nb * per_big; }
                else tmask = 0; // (one trial or none: the ordinary task does)
            }
        }
    }
    if (threadIdx.x == 0) st_m<true>(&p.win[w].remaining, n_slots + extra);
    // the numbers of the new entries are reserved while the window's new state, its delays and its counter drain:
    // one wait for both, and only then the cells
    uint32_t base = 0;
    if (threadIdx.x == 0) base = __hip_atomic_fetch_add(p.q_tail, n_slots + extra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_stores();
    base = uniform_u32(base);
    for (uint32_t i = threadIdx.x; i < n_slots; i += 64) {
        const uint32_t e = base + i;
        uint32_t ph = (uint32_t)phase;
        if constexpr (BIG) {
            if (tmask && as_global(p.big->slot_info)[slot0 + i]) ph = (uint32_t)kPhOneTrial + (uint32_t)__builtin_ctz(tmask); // its first wanted trial
        }
        st_m<true>(p.q + (e & p.q_mask), ((unsigned long long)((e >> p.q_shift) + 1u) << 32) | (ph << 24) | (slot0 + i));
    }
    if constexpr (BIG) {
        for (uint32_t x = threadIdx.x; x < extra; x += 64) { // the big frames' further trials
            const uint32_t b = x / per_big, r = x % per_big + 1u; // the (r + 1)-th wanted trial
            uint32_t m = tmask;
            for (uint32_t q = 0; q < r; ++q) m &= m - 1u;
            const uint32_t e = base + n_slots + x;
            st_m<true>(p.q + (e & p.q_mask), ((unsigned long long)((e >> p.q_shift) + 1u) << 32) |
                                                 (((uint32_t)kPhOneTrial + (uint32_t)__builtin_ctz(m)) << 24) | as_global(p.big->big_list)[b0 + b]);
        }
    }
}

constexpr uint32_t kExecStage = rs::kPlanExecStage; // doubles of LDS the decisions may use for a window's per-slot values (10 KB; window_plan.hpp sizes the region)

// The window's sums of rows [0, rows) of part[] (only those in `mask`) in the plan's order (window_sums /
// plan_sum_kernel: a chunk sequentially, then the chunks in order); out[r] valid in every lane.  The values are
// fetched by all lanes at once (sc1) into LDS, the additions then run on LDS.
__device__ __forceinline__ void exec_window_sums(const ExecParams& p, uint32_t w, uint32_t slot0, uint32_t n_slots, uint32_t rows,
                                                 uint32_t mask, double* stage, double* out) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    const bool staged = rows * n_slots <= kExecStage;
    __syncthreads();
    if (staged) {
        for (uint32_t e = lane; e < rows * n_slots; e += 64) {
            const uint32_t r = e / n_slots, j = e % n_slots;
            if ((mask >> r) & 1u) stage[e] = ld_m<true>(&p.part[(size_t)r * p.n_sel + slot0 + j]);
        }
        __syncthreads();
    }
    double mine = 0.0;
    if ((uint32_t)lane < rows && ((mask >> lane) & 1u)) {
        const uint32_t r = (uint32_t)lane;
        double tot = 0.0;
        for (uint32_t c = p.win_chunk_off[w]; c < p.win_chunk_off[w + 1]; ++c) {
            double acc = 0.0;
            for (uint32_t j = p.chunk_off[c]; j < p.chunk_off[c + 1]; ++j)
                acc += staged ? stage[r * n_slots + (j - slot0)] : ld_m<true>(&p.part[(size_t)r * p.n_sel + j]);
            tot += acc;
        }
        mine = tot;
    }
    for (uint32_t r = 0; r < rows; ++r) out[r] = read_lane_d(mine, (int)r);
}

// a window's record moves between global memory and LDS as 8-byte words, all lanes at once
constexpr int kWinWords = (int)((sizeof(ExecWin) + 7) / 8);
static_assert(kWinWords <= 64, "ExecWin must fit one word per lane");
__device__ __forceinline__ void exec_load_win(const ExecWin* src, ExecWin* lds_copy) {
    __syncthreads();
    if ((int)threadIdx.x < kWinWords)
        ((unsigned long long*)lds_copy)[threadIdx.x] = ld_m<true>((const unsigned long long*)src + threadIdx.x);
    __syncthreads();
}
__device__ __forceinline__ void exec_store_win(ExecWin* dst, const ExecWin* lds_copy) {
    __syncthreads();
    // (the counter is not part of the copy: exec_push sets it, the tasks decrement it)
    constexpr int kSkip = (int)(offsetof(ExecWin, remaining) / 8);
    static_assert(offsetof(ExecWin, remaining) % 8 == 0 && offsetof(ExecWin, remaining_pad) == offsetof(ExecWin, remaining) + 4, "layout");
    if ((int)threadIdx.x < kWinWords && (int)threadIdx.x != kSkip)
        st_m<true>((unsigned long long*)dst + threadIdx.x, ((const unsigned long long*)lds_copy)[threadIdx.x]);
}

// the decisions of window w after the last task of its phase; executed by one whole wave
template <bool BIG>
__device__ __forceinline__ void exec_decide(const ExecParams& p, uint32_t w, ExecWin* L, double* stage) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
#if RSSYNC_EXEC_STATS
    const unsigned long long t_dec__ = __builtin_amdgcn_s_memrealtime();
#endif
    exec_load_win(&p.win[w], L);
    SyncWin& s = L->s;
    const int ph = L->phase;
#if RSSYNC_EXEC_STATS
    if (lane == 0 && w < 4096u) {
        const unsigned long long tl = ld_m<true>(&g_exec_tlast[w]); // (written by a wave of another XCD: past L1 / L2)
        if (tl) {
            atomicAdd(&g_exec_stats[16 + ph], t_dec__ - tl);
            atomicAdd(&g_exec_stats[32 + ph], 1ull);
        }
    }
#endif
    const uint32_t slot0 = L->slot0, n_slots = L->n_slots;
    bool finished = false;
    if (ph == kPhInit) {
        if (lane == 0) {
            int32_t kd; double fd;
            split64_dev(s.d, p.lp.fs, &kd, &fd);          // :311 at d ...
            st_m<true>(&p.mo_kd[w], kd); st_m<true>(&p.mo_fd[w], fd);
            split64_dev(s.x0, p.lp.fs, &kd, &fd);         // ... then loss + derivative at x0 = d - 0.3 v (:298-299)
            st_m<true>(&p.lg_kd[w], kd); st_m<true>(&p.lg_fd[w], fd);
            L->phase = kPhMotion;
        }
    } else if (ph == kPhMotion) { // (the frame's loss and derivative at x0 came with its motion task)
        double sums[2];
        exec_window_sums(p, w, slot0, n_slots, 2u, 3u, stage, sums);
        if (lane == 0) {
            grad_decide(p.lp, s, sums[0], sums[1]);
            int32_t kd[kMaxBt]; double fd[kMaxBt];
            trial_delays(p.lp, s, kd, fd, 1);
            for (int i = 0; i < kMaxBt; ++i) { st_m<true>(&p.tr_kd[(size_t)i * p.n_win + w], kd[i]); st_m<true>(&p.tr_fd[(size_t)i * p.n_win + w], fd[i]); }
            L->phase = kPhTrials;
        }
    } else if (ph == kPhTrials) {
        uint32_t mask = 0; // only the rows that were evaluated
        for (int i = 0; i < kMaxBt; ++i) mask |= trial_wanted(s, i) ? (1u << i) : 0u;
        double lt[kMaxBt];
        exec_window_sums(p, w, slot0, n_slots, (uint32_t)kMaxBt, mask, stage, lt);
        if (lane == 0) {
            const bool stepped = step_decide(p.lp, s, lt, p.trace + ((size_t)w * p.trace_rows + L->trace_base) * 6, 6);
            int32_t kd; double fd;
            if (!stepped) {
                int32_t kdv[kMaxBt]; double fdv[kMaxBt];
                trial_delays(p.lp, s, kdv, fdv, 1); // the rest of the trials
                for (int i = 0; i < kMaxBt; ++i) { st_m<true>(&p.tr_kd[(size_t)i * p.n_win + w], kdv[i]); st_m<true>(&p.tr_fd[(size_t)i * p.n_win + w], fdv[i]); }
            } else if (s.active) {
                split64_dev(s.d, p.lp.fs, &kd, &fd);
                st_m<true>(&p.mo_kd[w], kd); st_m<true>(&p.mo_fd[w], fd);
                split64_dev(s.x0, p.lp.fs, &kd, &fd);
                st_m<true>(&p.lg_kd[w], kd); st_m<true>(&p.lg_fd[w], fd);
                L->phase = kPhMotion;
            } else {
                L->iters_call[L->call] = s.iters;
                L->trace_base += s.iters;
                if (L->call + 1 < p.repeats) { // the next Sync call of this sync point starts where this one ended (core_testcode.cpp:314)
                    L->call += 1;
                    const double d = s.d;
                    s = SyncWin{};
                    s.d = d;
                    s.active = 1;
                    s.hit = -1;
                    s.nf = p.lp.nf_fixed ? p.lp.nf_fixed : p.lp.nf_floor;
                    s.x0 = s.d - p.lp.delay_b * s.v;
                    st_m<true>(&p.win_stream[w], p.stream_first + (uint32_t)L->call + w * p.stream_stride);
                    int32_t ikd; float ifd;
                    split32_dev(s.d, p.lp.fs, &ikd, &ifd);
                    st_m<true>(&p.in_kd[w], ikd); st_m<true>(&p.in_fd[w], ifd);
                    if (p.init.kd64) { // the same delay split in fp64, for a near-static frame's search (kernels/lmeds_small.hpp, MODE 1)
                        int32_t kd64; double fd64;
                        split64_dev(s.d, p.lp.fs, &kd64, &fd64);
                        st_m<true>(const_cast<int32_t*>(&p.init.kd64[w]), kd64);
                        st_m<true>(const_cast<double*>(&p.init.fd64[w]), fd64);
                    }
                    L->phase = kPhInit;
                } else {
                    split64_dev(s.d, p.lp.fs, &kd, &fd); // :333
                    st_m<true>(&p.lg_kd[w], kd); st_m<true>(&p.lg_fd[w], fd);
                    L->phase = kPhFinal;
                }
            }
        }
    } else if (ph == kPhFinal) {
        double sum[1];
        exec_window_sums(p, w, slot0, n_slots, 1u, 1u, stage, sum);
        if (lane == 0) {
            L->cost = sum[0];
            L->phase = kPhDone;
        }
        finished = true;
    }
    exec_store_win(&p.win[w], L);
    if (finished) {
        wait_stores();
        uint32_t before = 0;
        if (lane == 0) before = __hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (uniform_u32(before) + 1u == p.n_win) {
            // the last window: one end marker per wave of the launch (every wave takes exactly one and leaves), so
            // that idle waves need not poll the "windows done" counter
            const uint32_t n = gridDim.x;
            uint32_t base = 0;
            if (lane == 0) base = __hip_atomic_fetch_add(p.q_tail, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            base = uniform_u32(base);
            for (uint32_t i = lane; i < n; i += 64) {
                const uint32_t e = base + i;
                st_m<true>(p.q + (e & p.q_mask), ((unsigned long long)((e >> p.q_shift) + 1u) << 32) | 0xffffffffull);
            }
        }
        return;
    }
    exec_push<BIG>(p, w, slot0, n_slots, L->phase, L->s);
#if RSSYNC_EXEC_STATS
    if (lane == 0 && w < 4096u) st_m<true>(&g_exec_tlast[w], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

// LDS of one executor wave: the search's tile, the L-BFGS history, a window record -- and, as the launch's dynamic
// LDS, ONE region of win_cap x 128 bytes that is in turn the search's fp32 spline window, the fp64 spline window of
// the motion / loss tasks and the decisions' staging area for a window's per-slot values (a task is one of these at
// a time).  win_cap >= kWinMax is the problem's (rship_sync_exec), so high gyro rates stay on the LDS paths; fewer
// waves then share a CU.
template <int RPT>
struct ExecLds {
    LmedsSmallLds<RPT, 0> small;
    MotionLds<1> mo;
    ExecWin win;
};
// (the region is at least kExecStage doubles: rssync_kernels.hip, exec_region_bytes -- checked again by rship_sync_exec)

// RPT = rows per lane of the one-wave kernels: frames of up to 64 * RPT tracks.  BIG = the selection also holds frames of
// more than 512 tracks (exec_big.hpp: p.big) -- an instantiation of its own, so that the reference's workload (one-wave
// frames only) runs the very code it ran before there were size classes.
template <int RPT, bool BIG = false>
__global__ __launch_bounds__(64, (RPT == 8 && !BIG) ? 2 : 1) void sync_exec_kernel(ExecParams p) { // (RPT = 8: 256 VGPRs, two waves per SIMD; the others fit anyway;
                                                                                                     //  RPT = 8 with BIG would spill at 256: one wave per SIMD there)
    __shared__ ExecLds<RPT> lds;
    extern __shared__ d4 s_exec_region[]; // [4 * win_cap] d4 = win_cap x 128 bytes
#ifndef RSSYNC_EXEC_PRELOAD   // (-DRSSYNC_EXEC_PRELOAD=0: the loads where the rows need them, for the A/B of profiles/r5_exec_preload_ab.txt)
#define RSSYNC_EXEC_PRELOAD 1
#endif
    // rows per lane whose rays a loss task requests up-front (sync64.hpp: RowRays); three where a fourth would cost the
    // instantiation its second wave per SIMD (tests/test_kernel_resources.py)
    constexpr int kPre = RSSYNC_EXEC_PRELOAD ? (RPT < 4 ? RPT : ((BIG && RPT == 4) ? 3 : 4)) : 0;
    const int lane = threadIdx.x;
    for (;;) {
        uint32_t slot;
        {
            EXEC_T0();
            slot = exec_pop(p);
            EXEC_ADD(6, 14);
        }
        if (slot == 0xffffffffu) break;
        int ph = (int)(slot >> 24); // (the cell carries the phase: exec_push)
        slot &= 0x00ffffffu;
        int one_trial = -1; // BIG: this task is ONE trial of a frame of more than 512 tracks
        if (BIG && ph >= kPhOneTrial) { one_trial = ph - kPhOneTrial; ph = kPhTrials; }
        const uint32_t w = p.grp[slot];
        // a frame of more than 512 tracks: one wave in the four-wave kernels' association (exec_big.hpp); the table is
        // written before the launch and never changes
        uint32_t info = 0u;
        if constexpr (BIG) info = uniform_u32(as_global(p.big->slot_info)[slot]);
        const uint32_t big_k = info & 7u, big_entry = (info >> 3) - 1u;
        EXEC_T0();
        if (ph == kPhInit) {
            if (BIG && info)
                exec_big_init<true>(p.init, slot, as_global(p.big->big_tile) + (size_t)big_entry * p.big->big_rows * kExecBigFloats, p.big->big_rows,
                                    reinterpret_cast<f4*>(s_exec_region), p.mo.win_bytes, p.big->init_cap[big_k],
                                    ((p.big->init_whole >> big_k) & 1u) != 0, big_k == 5u);
            else
                lmeds_small_body<RPT, 1, true, 0>(p.init, slot, 0u, lds.small, reinterpret_cast<f4*>(s_exec_region));
        } else if (ph == kPhMotion) {
            // the frame's loss and derivative at x0 with the motion estimate just found (handed over in registers:
            // the values the body has stored)
            double mk[4];
            if (BIG && info) {
                Motion64Params mb = p.mo; // (the rows of P of the big entries: not part of the one-wave frames' parameters)
                mb.scratch = as_global(p.big->big_P);
                mb.scratch_rows = p.big->big_rows;
                opt_motion64_body<0, 1, true, true>(mb, slot, lds.mo, s_exec_region, mk, big_entry);
            } else opt_motion64_body<RPT, 1, true>(p.mo, slot, lds.mo, s_exec_region, mk);
            const d3 Mv = d3{mk[0], mk[1], mk[2]};
            double Lv, Gv;
            loss64_wave<true, false, kPre>(p.lo, slot, Mv, mk[3], ld_m<true>(&p.lg_kd[w]), ld_m<true>(&p.lg_fd[w]), s_exec_region, Lv, Gv);
            if (lane == 0) { st_m<true>(&p.part[slot], Lv); st_m<true>(&p.part[(size_t)p.n_sel + slot], Gv); }
        } else if (ph == kPhTrials || ph == kPhFinal) {
            const d3 Mv = d3{ld_m<true>(&p.lo.M[3 * slot]), ld_m<true>(&p.lo.M[3 * slot + 1]), ld_m<true>(&p.lo.M[3 * slot + 2])};
            const double kk = ld_m<true>(&p.lo.k[slot]);
            if (ph == kPhFinal) {
                double Lv, Gv;
                loss64_wave<false, false, kPre>(p.lo, slot, Mv, kk, ld_m<true>(&p.lg_kd[w]), ld_m<true>(&p.lg_fd[w]), s_exec_region, Lv, Gv);
                if (lane == 0) st_m<true>(&p.part[slot], Lv);
            } else if (BIG && one_trial >= 0) {
                double Lv, Gv;
                loss64_wave<false>(p.lo, slot, Mv, kk, ld_m<true>(&p.tr_kd[(size_t)one_trial * p.n_win + w]),
                                   ld_m<true>(&p.tr_fd[(size_t)one_trial * p.n_win + w]), s_exec_region, Lv, Gv);
                if (lane == 0) st_m<true>(&p.part[(size_t)one_trial * p.n_sel + slot], Lv);
            } else {
                // the window's ten trial delays in ONE round trip (lane i fetches trial i; a load past L1 takes ~1.5 us,
                // and ten of them one after the other were most of a trial task)
                double my_fd = __builtin_nan("");
                int my_kd = 0;
                if (lane < kMaxBt) {
                    my_fd = ld_m<true>(&p.tr_fd[(size_t)lane * p.n_win + w]);
                    my_kd = ld_m<true>(&p.tr_kd[(size_t)lane * p.n_win + w]);
                }
                for (int i = 0; i < kMaxBt; ++i) {
                    const double fd = read_lane_d(my_fd, i);
                    if (fd != fd) continue; // not asked for
                    double Lv, Gv;
                    loss64_wave<false, false, kPre>(p.lo, slot, Mv, kk, __builtin_amdgcn_readlane(my_kd, i), fd, s_exec_region, Lv, Gv);
                    if (lane == 0) st_m<true>(&p.part[(size_t)i * p.n_sel + slot], Lv);
                }
            }
        }
        EXEC_ADD(ph, 8 + ph);
        wait_stores(); // this task's results before the count
        uint32_t left = 0;
        if (lane == 0) left = __hip_atomic_fetch_sub(&p.win[w].remaining, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        left = uniform_u32(left);
        if (left == 1u) { // the last task of the window's phase: the add has returned, the others' results are there
            EXEC_T0();
            exec_decide<BIG>(p, w, &lds.win, (double*)s_exec_region);
            EXEC_ADD(5, 13);
        }
    }
}

} // namespace
