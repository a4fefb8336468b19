// rssync_kernels.hip -- gfx950 (CDNA4) kernels for the rs-sync PreSync/Sync hot
// path and the thin C-ABI the host solver calls (include/rssync_hip.h).
//
// Kernels (all wave64, no MFMA: the path has no dense contraction):
//   lmeds_kernel       one workgroup per (frame, chunk of candidate delays): residual
//                      matrix P into an LDS tile, LMedS hypothesis search with an exact
//                      lower-quartile selection, robust PreSync cost.  The same kernel
//                      in INIT mode is Sync's GuessMotion/GuessK.
//   lmeds_small_kernel the same for frames of up to 512 tracks: one wave per (frame, chunk), rows in registers.
//   loss64_kernel      one workgroup per frame: residual + robust loss (+ analytic
//                      d/d-delay) for a batch of delays, fp64 (the reference's arithmetic).
//   opt_motion64_kernel one workgroup per frame: P in registers (fp64), restated L-BFGS on the
//                      3-vector motion estimate; also finishes GuessMotion/GuessK in fp64.
//   pack_frames_kernel raw track records -> packed fp32 + fp64 streams.
//   plan_sum_kernel    sums over the frames of each window in an association that does not
//                      depend on the number of devices sharing the frames.
//   sync_*_kernel      Sync's outer loop kept on the device: window sums + the scalar decisions
//                      between the launches (kernels/syncloop.hpp).
//   gyro_*, spline_*   the gyro side: integration scan, microsecond-grid resampling, spline
//                      table (kernels/gyro.hpp).
// Data layout and the roofline that bounds each kernel: DESIGN.md.
// The kernels live in kernels/*.hpp (one header each, included below); this file holds the
// device context and the launchers.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#ifndef RSSYNC_TEST_VARIANTS
#define RSSYNC_TEST_VARIANTS 0 // 1: also the kernel variants that exist only so that a test can compare the product against them
#endif

#include "../../include/rssync_hip.h"
#include "device_math.hpp"
#include "sync_math.hpp"
#include "lens_math.hpp"
#include "gyro_math.hpp"
#include "window_plan.hpp"
#include "roctx_ranges.hpp"

using rs::f3;
using rs::f4;

// the kernels, one header per kernel, all in this translation unit
#include "kernels/common.hpp"
#include "kernels/rows64.hpp"
#include "kernels/lmeds.hpp"
#include "kernels/lmeds_small.hpp"
#include "kernels/lmeds_big.hpp"
#include "kernels/support.hpp"
#include "kernels/sync64.hpp"
#include "kernels/syncloop.hpp"
#include "kernels/exec_big.hpp"
#include "kernels/executor.hpp"
#include "kernels/gyro.hpp"

// ===========================================================================
// host side of the C-ABI

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};
// a device allocation that lives for one call (freed on every return path)
struct TempBuf : DevBuf {
    TempBuf() = default;
    TempBuf(const TempBuf&) = delete;
    TempBuf& operator=(const TempBuf&) = delete;
    ~TempBuf() {
        if (p) (void)hipFree(p);
    }
};

constexpr int kNumClasses = 6; // size classes of frames (rship_ctx::cls_slots)

struct rship_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr; // raw-record uploads, overlapping the caller's SetTrackResult loop
    hipEvent_t copy_done = nullptr;
    std::string err;
    // problem data
    DevBuf coef, coef64, raw, rays_a, rays_b, rays64, frames, sel, M, k, grp, grp_off, init_h;
    DevBuf mo_evals, mo_order; // per slot: evaluations of the last motion launch; launch order (longest first)
    // mo_order is a permutation of each of these slot ranges (the ranges the motion launches of the last call covered);
    // mo_identity: it is the identity, hence a permutation of any range
    std::vector<std::pair<uint32_t, uint32_t>> mo_ranges;
    bool mo_identity = false;
    // reduction plan (rship_set_plan): windows -> chunks -> slots
    DevBuf plan_idx, plan_chunk_off, plan_win_off, chunk_out, win_out;
    uint32_t plan_max_chunks = 0; // most chunks in one window
    // gyro pipeline (rship_gyro_*): inputs, intermediate orientations, grid knots, forward-sweep values, status
    DevBuf g_ts, g_rates, g_us, g_dq, g_q, g_knots, g_cf, g_status;
    uint32_t g_n = 0; // samples of the last rship_gyro_rates_upload
    int64_t g_first_us = 0, g_last_us = 0;
    std::vector<hipStream_t> loop_streams; // rship_sync_run: one per group of windows
    hipEvent_t loop_ready = nullptr;
    std::vector<uint32_t> h_grp_off;       // host copy of the selection's group offsets
    DevBuf loop_state;            // rship_sync_run: windows, delay arrays, counters, trace
    bool plan_has_idx = false;
    uint32_t plan_chunks = 0, plan_wins = 0, plan_len = 0;
    // what the last *_enqueue left for its *_collect
    struct Pend { uint32_t rows = 0; bool grad = false, frame_costs = false, best_h = false; size_t off_chunk = 0, off_flags = 0; } pend;
    // GuessMotion's hypothesis search has run and left winners in init_h: the next motion launch finishes it
    bool init_pending = false;
    uint64_t init_seed = 0;
    uint32_t init_stream = 0, init_stride = 0;
    uint32_t n_knots = 0, n_frames = 0, n_sel = 0, max_n = 0, n_grp = 1;
    uint64_t total_rays = 0;
    double fs = 0;
    int lbfgs_reeval = 0; // RSHIP_OPT_LBFGS_REEVAL
    // SIZE CLASSES (round 5).  Which kernel family a frame runs in -- and with it the association of its sums -- follows
    // from the frame's OWN track count: class 0 = up to one_wave_max tracks (the one-wave kernels), 1 .. 4 = the four-wave
    // kernels with 4 / 8 / 16-24 / 32 rows per thread (up to 1024 / 2048 / 6144 / 8192 tracks), 5 = more than 8192 (rows in
    // global memory).  A selection is cut into one slot list per class (cls_slots, ascending slots inside a class) and
    // every launch covers one class; the plan of the sums indexes by slot and does not notice.  The reference evaluates
    // each frame in its own lambda (core_private.cpp:73-86, :231-238, :263-295): a frame's result must not depend on what
    // else the problem, the window, the device or the rank holds.
    DevBuf cls_slots;                    // [n_sel] slots sorted by (class, slot)
    std::vector<uint32_t> h_cls_slots;
    uint32_t cls_off[kNumClasses + 1] = {};   // class k = h_cls_slots[cls_off[k] .. cls_off[k + 1])
    uint32_t cls_max_n[kNumClasses] = {};     // largest frame of the class IN THE SELECTION (rows per lane of the one-wave kernels, scratch rows)
    int cls_used = 0;                         // classes with slots
    // what the spline windows are planned from: the frames of this context's table, or -- rship_set_problem_frames: one
    // object over several devices -- of the whole problem, so that every shard plans like the single-device run
    std::vector<rs::FrameDims> own_dims, problem_dims;
    uint32_t cls_cap64[kNumClasses] = {80, 80, 80, 80, 80, 80}; // knots of the fp64 window per class (window_plan.hpp: cap64_frames)
    bool cls_compact = false;     // class 0's fp64 window is COMPACT: y and c only, 64 bytes per knot (kernels/sync64.hpp: Spline64::compact)
    bool no_compact = false;      // RSSYNC_NO_COMPACT_WINDOW=1 (A/B, the bits test): round 4's rule -- full records up to 144 knots, beyond that the table from L2
    bool force_general = false;   // RSSYNC_FORCE_GENERAL_SPLINE=1 (read at creation; tools/gpu_gyro_rate.py's "before" column): no dynamic
                                  // spline windows -- frames wider than 80 knots take the general path (table from L2), as in rounds 1-3
    uint32_t one_wave_max = 512;  // frames of up to this many tracks run the one-wave kernels (K2s, loss64_small, the executor); RSSYNC_ONE_WAVE_MAX (tests, A/B)
    uint32_t exec_big_max = 2048; // the window executor takes selections whose largest frame has up to this many tracks (RSSYNC_EXEC_BIG_MAX): a larger
                                  // frame's tasks, serial in ONE wave, would be the whole call (one 9000-track frame among 130-track ones: 170 ms
                                  // in the executor, 94 ms through the chain of launches, profiles/r5_syncpoints_mixed.json)
    uint32_t exec_big_share = 8;  // ... and in which at most one slot in this many holds a frame of more than 512 tracks (RSSYNC_EXEC_BIG_SHARE; 1 = any)
    bool force_big = false;       // RSSYNC_FORCE_BIG=1 (tests): every frame through the kernels for frames of more than 8192 tracks
    bool no_small_loss = false;   // RSSYNC_NO_SMALL_LOSS=1 (A/B): frames of up to 512 tracks in the four-wave loss kernel
    bool exact_select = false;   // RSSYNC_K2_EXACT_SELECT=1 (read once, at creation; only in the -DRSSYNC_TEST_VARIANTS=1 build): PreSync's
                                 // tile kernel with round 2's exact selection of every quartile instead of the lazy one (tests: identical results)
    bool no_subshapes = false;   // RSSYNC_NO_SUBSHAPES=1 (read once, at creation): every class sweeps in its own shape (A/B, tests)
    bool no_small_lmeds = false; // RSSYNC_NO_SMALL_LMEDS=1 (read once, at creation): the tile kernel for every frame size (A/B tests)
    float max_span = 0.f; // widest frame, in knots (frame table)
    float max_ends = 0.f; // the same counting only the two ends' ranges of each pair (rship_frame::range_a / range_b)
    int lds_per_cu = 160 * 1024;
    uint32_t last_lmeds_shape[kNumClasses] = {}; // rship_lmeds_shapes: rows / 256 of the tile the last PreSync sweep gave each class (0: none launched, or not the tile kernel)
    uint32_t last_lmeds_cap = 0, last_lmeds_chunk = 0, last_init_cap = 0; // rship_window_info: what the last launches used
    // native exchange (RCCL through dlopen)
    void* rccl_lib = nullptr;
    std::string rccl_path; // which librccl the symbols come from
    void* rccl_comm = nullptr;
    DevBuf rccl_buf;
    DevBuf big_scratch, mo_scratch; // frames of more than 8192 tracks: the LMedS tiles / the motion kernel's rows (per entry of class 5's slot list)
    rship_loop_exchange_fn loop_xchg = nullptr; // host exchange for the device-driven loop (rship_set_loop_exchange)
    void* loop_xchg_user = nullptr;
    uint64_t loop_exchanges = 0; // all-reduces the last rship_sync_run enqueued on the stream (rank mode)
    uint32_t exec_last[4] = {0, 0, 0, 0}; // rship_exec_stats: head, tail, ring cells, waves of the last rship_sync_exec
    std::vector<uint32_t> h_frame_n; // per table frame
    std::vector<uint32_t> h_sel;
    std::vector<uint32_t> h_delays, h_delays64; // staging of upload_delays / upload_delays64
    const float* d_fd = nullptr;    // device address of the fd half of the last upload
    DevBuf kd64;                    // kd[n] (int32) then fd[n] (double) for the fp64 kernels
    const double* d_fd64 = nullptr;
    // NEAR-STATIC frames: rows in fp64 (kernels/lmeds.hpp, "fp64 rows").  redo_mask: one bit per (slot, candidate), all zero
    // between calls (the R64 kernels clear what they serve); redo_count: pairs recomputed (device counter);
    // what the last presync enqueue left for a redo at collect time
    DevBuf redo_mask, redo_delays, redo_count, init_delays64; // (redo_count: two counters -- PreSync pairs, GuessMotion searches)
    bool no_fp64_rows = false;  // RSSYNC_NO_FP64_ROWS=1 (read at creation; the "before" column of profiles/r6_near_static.json): the watch is off
    bool redo_dirty = true;     // the mask may hold bits (fresh allocation, a call that failed half-way): cleared before the next sweep
    uint64_t near_launches = 0; // sweeps that went through the fp64 form
    struct PendRedo { bool armed = false; LmedsParams p{}; double step_knots = 0; uint32_t chunk = 1; bool uploaded = false; } pend_redo;
    std::vector<int32_t> h_kd64r;
    std::vector<double> h_fd64r;
    std::vector<uint32_t> h_init64; // staging of rship_init_motion's fp64 delays
    // A BATCH of sweeps collected together (rship_presync_batch_begin: the orientation sweep, BASELINE config 5): the results
    // of sweep b of n go to slot b of the (n times larger) sum / flag buffers, nothing is waited for in between
    uint32_t batch_n = 0, batch_next = 0, batch_rows = 0;
    size_t batch_chunk_stride = 0, batch_win_stride = 0; // doubles per slot in chunk_out / win_out
    // TEST-VARIANTS build: the sweep's own residuals (kernels/lmeds.hpp: LmedsParams::dump) of the last rship_presync_enqueue
    DevBuf dump;
    bool dump_on = false;
    uint32_t dump_rows = 0, dump_dims[4] = {0, 0, 0, 0};
    // scratch
    DevBuf kd, frame_cost, best_h, costs, part, flags, stats;
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    // profiling
    bool prof = false;
    struct Pending { int kind; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    uint64_t launches[RSHIP_K_COUNT] = {};
    double total_ms[RSHIP_K_COUNT] = {};
};

namespace {

std::mutex g_err_mutex; // rship_sync_run launches from several host threads
int set_err(rship_ctx* c, const char* what, hipError_t e) {
    std::lock_guard<std::mutex> lock(g_err_mutex);
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}
int set_err(rship_ctx* c, const std::string& what) {
    std::lock_guard<std::mutex> lock(g_err_mutex);
    c->err = what;
    return 1;
}

#define RS_HIP(call)                                         \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return set_err(c, #call, e__); \
    } while (0)

int ensure(rship_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) RS_HIP(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    RS_HIP(hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

int ensure_pinned(rship_ctx* c, size_t bytes) {
    if (bytes <= c->pinned_cap) return 0;
    if (c->pinned) RS_HIP(hipHostFree(c->pinned));
    c->pinned = nullptr;
    c->pinned_cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    RS_HIP(hipHostMalloc(&c->pinned, want, hipHostMallocDefault));
    c->pinned_cap = want;
    return 0;
}

// names of the RSHIP_K_* launch kinds: the roctx range around a kind's launches (RSSYNC_ROCTX=1, roctx_ranges.hpp)
const char* const kKindNames[RSHIP_K_COUNT] = {"rssync:K2 lmeds sweep", "rssync:K1 loss (trials)", "rssync:K3 motion L-BFGS", "rssync:window sums",
                                               "rssync:K2 GuessMotion search", "rssync:pack frames", "rssync:gyro pipeline", "rssync:K1 loss+gradient"};
struct ProfScope {
    rship_ctx* c;
    int kind;
    hipEvent_t a = nullptr, b = nullptr;
    rs::RoctxRange range;
    ProfScope(rship_ctx* c_, int kind_) : c(c_), kind(kind_), range(kKindNames[kind_]) {
        if (!c->prof) return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, c->stream);
    }
    ~ProfScope() {
        if (!c->prof) return;
        (void)hipEventRecord(b, c->stream);
        c->pending.push_back({kind, a, b});
    }
};

// called after a stream synchronisation: fold finished event pairs into the totals
void prof_collect(rship_ctx* c) {
    for (auto& pd : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pd.a, pd.b) == hipSuccess) {
            c->launches[pd.kind] += 1;
            c->total_ms[pd.kind] += (double)ms;
        }
        c->pool.push_back(pd.a);
        c->pool.push_back(pd.b);
    }
    c->pending.clear();
}

int sync_stream(rship_ctx* c) {
    RS_HIP(hipStreamSynchronize(c->stream));
    prof_collect(c);
    return 0;
}

constexpr int kMaxRpt = 32; // 8192 tracks per frame in registers / LDS; larger frames take the kernels' slow paths
// Rows per thread up to which PreSync's tile kernel runs as FOUR waves (class 3: 2049 .. 6144 tracks).  Round 6 (second half):
// until then class 3 ended at 4096 tracks and 4097 .. 8192 ran as eight waves, ONE workgroup per CU; a tile of up to 6144 rows
// (72 KB) still lets TWO four-wave workgroups share a CU, whose barrier phases cover each other: 17.1 .. 19.0 ms per 2^21 ray
// pairs at 4097 .. 6000 tracks against 22.9 .. 25.6 as eight waves (profiles/r6_k2_class3_6144_ab.txt).  Above 6144 rows only
// one workgroup fits whatever its shape, and eight waves are the better one (class 4).
constexpr int kFourWaveMaxRpt = 24;
constexpr int kWideBlock = 512; // the LMedS tile kernel's workgroup for 6145 .. 8192 tracks (class 4): eight waves x 16 rows per thread (kernels/lmeds.hpp: BLOCK)

// ---- size classes ---------------------------------------------------------------------------------------------
// class of a frame of n tracks (rship_ctx::cls_slots): 0 one wave per frame, 1 .. 4 four waves with 4 / 8 / 16-24 / 32 rows
// per thread (PreSync's tile kernel runs class 4 -- 6145 .. 8192 tracks -- as EIGHT waves of 16), 5 rows in global memory.  A thread adds its rows in order and rows beyond the frame add exact zeros, so a
// frame's sums do not depend on the rows-per-thread instantiation INSIDE a family (one wave / four waves): the classes
// 1 .. 4 differ in speed only (registers, workgroups per CU), 0 and 5 in the association (0) and the fp32 spline path (5).
static_assert(rs::kPlanBlock == (uint32_t)kBlock && rs::kPlanFourWaveMaxRpt == (uint32_t)kFourWaveMaxRpt && rs::kPlanMaxRpt == (uint32_t)kMaxRpt,
              "window_plan.hpp (the rules, checked on the CPU by tests/test_window_plan.py) and this file disagree on the size classes");
int class_of(const rship_ctx* c, uint32_t n) {
    if (c->force_big) return 5;
    return rs::plan_class_of(n, c->one_wave_max);
}
// track counts of class k (both ends inclusive)
void class_bounds(const rship_ctx* c, int k, uint32_t* lo, uint32_t* hi) {
    if (c->force_big) { *lo = k == 5 ? 0u : 1u; *hi = k == 5 ? 0xffffffffu : 0u; return; }
    const uint32_t top[6] = {c->one_wave_max, 4u * kBlock, 8u * kBlock, (uint32_t)kFourWaveMaxRpt * kBlock, (uint32_t)kMaxRpt * kBlock, 0xffffffffu};
    *hi = top[k];
    *lo = k == 0 ? 0u : top[k - 1] + 1u;
}
// rows per thread of the four-wave kernels of class k (0 = as many as the frame needs: class 5)
// (class 3, 2049 .. 6144 tracks: 16 while the selection's largest frame of the class has at most 4096 tracks, else 24 -- the
// same bits, a thread adds its rows in order and rows beyond the frame add exact zeros)
int class_rpt(const rship_ctx* c, int k) { return rs::plan_class_shape(k, c->cls_max_n[k]); }
// rows per lane of the one-wave kernels: 1 .. 4 up to 256 tracks, 8 for 257 .. 512 (rows beyond the frame contribute
// exact zeros, so one instantiation serves them all with the same bits)
int small_rpt(uint32_t n_all) { return rs::plan_small_rows(n_all, false); }
// rows per thread of a 256-thread workgroup that cover max_n tracks (a power of two, at least four): the four-wave
// kernels on frames of class 0 (RSSYNC_NO_SMALL_LMEDS / _LOSS: the tests' cross-checks of the two families)
int rpt_for(uint32_t max_n) {
    if (max_n > (uint32_t)kMaxRpt * kBlock) return 0;
    int rpt = 4;
    while ((uint32_t)rpt * kBlock < max_n) rpt *= 2;
    return rpt;
}
uint32_t big_rows(uint32_t max_n) { return (max_n + kBlock - 1) / kBlock * kBlock; }

// the frames the windows are planned from (rship_ctx::own_dims / problem_dims)
const std::vector<rs::FrameDims>& plan_dims(const rship_ctx* c) { return c->problem_dims.empty() ? c->own_dims : c->problem_dims; }
void update_class_caps(rship_ctx* c) {
    const std::vector<rs::FrameDims>& d = plan_dims(c);
    for (int k = 0; k < kNumClasses; ++k) {
        uint32_t lo, hi;
        class_bounds(c, k, &lo, &hi);
        bool comp = false;
        const bool may_compact = k == 0 && !c->no_compact && !c->no_small_loss && !c->force_general;
        c->cls_cap64[k] = c->force_general ? (uint32_t)rs::kPlanWinStatic : rs::cap64_frames(d.data(), d.size(), lo, hi, k == 0, may_compact ? &comp : nullptr);
        if (k == 0) c->cls_compact = comp;
    }
}

// One launch's share of the selection: the entries [pos0, pos0 + count) of the class-sorted slot list -- all of class k.
// list == nullptr: the selection is ONE class and the list is the identity (entry = slot): kernels then take
// slot0 + blockIdx.x, exactly as before there were classes (the benchmark's launches are unchanged).
struct ClassRange {
    int k;
    uint32_t pos0, count;
    const uint32_t* list; // device: c->cls_slots + pos0, or null
};
// the classes' shares of the slots [slot0, slot0 + count) (count = 0: the whole selection)
std::vector<ClassRange> class_ranges(const rship_ctx* c, uint32_t slot0 = 0, uint32_t count = 0) {
    std::vector<ClassRange> out;
    if (!count) { slot0 = 0; count = c->n_sel; }
    const uint32_t* L = c->h_cls_slots.data();
    for (int k = 0; k < kNumClasses; ++k) {
        const uint32_t* a = L + c->cls_off[k];
        const uint32_t* b = L + c->cls_off[k + 1];
        if (a == b) continue;
        const uint32_t* lo = std::lower_bound(a, b, slot0);
        const uint32_t* hi = std::lower_bound(a, b, slot0 + count);
        if (lo == hi) continue;
        const uint32_t pos0 = (uint32_t)(lo - L);
        out.push_back(ClassRange{k, pos0, (uint32_t)(hi - lo), c->cls_used > 1 ? (const uint32_t*)c->cls_slots.p + pos0 : nullptr});
    }
    return out;
}
// the class most slots of the selection belong to (what rship_window_info reports, what a one-class problem has)
int main_class(const rship_ctx* c) {
    int best = 0;
    uint32_t most = 0;
    for (int k = 0; k < kNumClasses; ++k) {
        const uint32_t n = c->cls_off[k + 1] - c->cls_off[k];
        if (n > most) { most = n; best = k; }
    }
    if (!most) { // no selection yet: the class of the table's largest frame
        uint32_t mx = 0;
        for (const rs::FrameDims& d : c->own_dims) mx = std::max(mx, d.n);
        best = class_of(c, mx);
    }
    return best;
}

// ---- spline windows in dynamic LDS (gyro rates above ~1.7 kHz: a frame spans 0.044 s x rate knots) ----
constexpr uint32_t kLossWinBytes = kLossBatch * kWinMax * 128u; // K1's LDS budget for its side-by-side windows (51 KB at 80 knots)
using rs::WinPlan;
static_assert(rs::kPlanWinStatic == (uint32_t)kWinMax, "window_plan.hpp and kernels/common.hpp disagree on the compiled-in window");
// The fp64 window a class's launches use (window_plan.hpp: cap64_frames).  The one-wave kernels (class 0: K1 / K3 in their
// one-wave shapes, the executor) stage a window PER EVALUATION for a frame's 260 coefficient fetches, and every knot of
// it is LDS that another wave of the CU cannot have: 96 knots leave the executor its eight waves per CU, 144 seven, 192
// five, 272 four.  Beyond kPlanCap64SmallMax knots the general path (260 x 128 bytes from L2, eight waves) is cheaper
// again.  Measured on 98 sync points of 61 x 130, two ends of a pair staged separately, outer iterations capped at 25 per
// call so that rates compare (profiles/r4_gyro_rate_small_frames.json, a build that allows 384 knots): 4 kHz 17.4 ms with
// 96 knots against 20.4 on the general path; 6 kHz 20.0 (144 knots) against 20.8; 8 kHz 25.4 (192) against 21.4; 12 kHz
// 33.6 (272) against 21.7.
uint32_t cap64_of(const rship_ctx* c, int k) { return c->cls_cap64[k]; }
bool compact_of(const rship_ctx* c, int k) { return k == 0 && c->cls_compact; }
size_t win64_bytes(const rship_ctx* c, int k) { return (size_t)c->cls_cap64[k] * (compact_of(c, k) ? 64u : 128u); } // one fp64 window of class k
// widest whole pair of class k among the frames the windows are planned from
float class_span(const rship_ctx* c, int k) {
    uint32_t lo, hi;
    class_bounds(c, k, &lo, &hi);
    float m = 0.f;
    for (const rs::FrameDims& d : plan_dims(c))
        if (d.n && d.n >= lo && d.n <= hi) m = std::max(m, d.span);
    return m;
}
template <class K>
uint32_t static_lds_of(K kernel) {
    hipFuncAttributes a{};
    if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kernel)) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return (uint32_t)a.sharedSizeBytes;
}
template <class K>
void allow_dynamic_lds(K kernel, size_t bytes) { // (more than 64 KB in all needs the opt-in on some runtimes; harmless otherwise)
    if (bytes > 32 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        (void)hipGetLastError();
    }
}

// ---- the LMedS kernels' spline window (fp32, 64 bytes per knot) ----
// A workgroup's window must hold its frame plus the knots its chunk of candidate delays spans.  Up to ~1.7 kHz
// of gyro rate that fits the kWinMax knots compiled into the kernels' LDS (the instantiations the benchmark runs).
// Beyond, the window moves to DYNAMIC LDS (WIN = 0 instantiations), as large as the class's frames need: the plan picks
// the largest number of workgroups per CU whose LDS share still holds the widest ELIGIBLE frame (window_plan.hpp:
// plan_window_frames) and a chunk of at least eight candidates, then the longest chunk (<= 32) that fits.  A frame that
// no window can hold takes the general path (table from L2), alone or in company.
// The tile kernel's SHAPES, by code = rows of the tile / 256.  Four waves: 4 / 8 / 16 or 24 rows per thread (classes 1 .. 3);
// eight waves of 16 rows (class 4, code 32).  Round 6, the SUB-SHAPES (PreSync's sweep only, MODE 0): every number of rows per
// thread from 3 to 23, and eight waves of 13 .. 15 (codes 26, 28, 30), for selections whose largest frame of the class needs no
// more rows (lmeds_shape below): a smaller tile, fewer residual registers, and no sweep over rows that no frame has.
template <int R, int B>
struct TileShape { static constexpr int rpt = R, block = B; };
template <int MODE, class F>
bool with_tile_shape(int code, F&& f) {
    switch (code) {
        case 4: f(TileShape<4, kBlock>{}); return true;
        case 8: f(TileShape<8, kBlock>{}); return true;
        case 16: f(TileShape<16, kBlock>{}); return true;
        case 24: f(TileShape<24, kBlock>{}); return true;
        case 32: f(TileShape<16, kWideBlock>{}); return true;
        default: break;
    }
    if constexpr (MODE == 0) {
        switch (code) {
#define RS_SUB4(G) case G: f(TileShape<G, kBlock>{}); return true;
#define RS_SUB8(G) case 2 * G: f(TileShape<G, kWideBlock>{}); return true;
            RS_SUB4(3) RS_SUB4(5) RS_SUB4(6) RS_SUB4(7) RS_SUB4(9) RS_SUB4(10) RS_SUB4(11) RS_SUB4(12) RS_SUB4(13) RS_SUB4(14) RS_SUB4(15)
            RS_SUB4(17) RS_SUB4(18) RS_SUB4(19) RS_SUB4(20) RS_SUB4(21) RS_SUB4(22) RS_SUB4(23)
            RS_SUB8(13) RS_SUB8(14) RS_SUB8(15)
#undef RS_SUB4
#undef RS_SUB8
            default: break;
        }
    }
    return false;
}
// workgroups per CU the window planner may aim at for a shape (the second __launch_bounds__ argument of its kernels)
int tile_shape_wgs(int code) { return code > kFourWaveMaxRpt ? 1 : (code >= 16 ? 2 : lmeds_waves(code)); }
template <int MODE>
uint32_t lmeds_dynamic_static_lds(int rpt, bool small) {
    if (small) {
        switch (rpt) {
            case 1: return static_lds_of(lmeds_small_kernel<1, MODE, 0>);
            case 2: return static_lds_of(lmeds_small_kernel<2, MODE, 0>);
            case 3: return static_lds_of(lmeds_small_kernel<3, MODE, 0>);
            case 4: return static_lds_of(lmeds_small_kernel<4, MODE, 0>);
            default: break;
        }
        if constexpr (MODE == 0) { // (the one-wave sub-shapes: lmeds_shape)
            switch (rpt) {
                case 5: return static_lds_of(lmeds_small_kernel<5, 0, 0>);
                case 6: return static_lds_of(lmeds_small_kernel<6, 0, 0>);
                case 7: return static_lds_of(lmeds_small_kernel<7, 0, 0>);
                default: break;
            }
        }
        return static_lds_of(lmeds_small_kernel<8, MODE, 0>);
    }
    uint32_t v = 0;
    with_tile_shape<MODE>(rpt, [&](auto sh) { v = static_lds_of(lmeds_kernel<decltype(sh)::rpt, MODE, 0, true, false, decltype(sh)::block>); });
    return v;
}
// static LDS of the tile kernel's compiled-in-window instantiation
template <int MODE>
uint32_t lmeds_static_lds(int rpt) {
    uint32_t v = 0;
    with_tile_shape<MODE>(rpt, [&](auto sh) { v = static_lds_of(lmeds_kernel<decltype(sh)::rpt, MODE, kWinMax, true, false, decltype(sh)::block>); });
    return v;
}
// which LMedS kernel the frames of class k get
enum class LmedsKind { Small, Tile, Big };
LmedsKind lmeds_kind(const rship_ctx* c, int k) {
    if (k == 5) return LmedsKind::Big;
    if (k == 0 && !c->no_small_lmeds) return LmedsKind::Small;
    return LmedsKind::Tile;
}
// rows per thread / per lane of class k's LMedS kernel in the current selection
int lmeds_rpt(const rship_ctx* c, int k) {
    const LmedsKind kind = lmeds_kind(c, k);
    if (kind == LmedsKind::Small) return small_rpt(c->cls_max_n[0]);
    if (kind == LmedsKind::Big) return 0;
    return k == 0 ? rpt_for(c->cls_max_n[0]) : class_rpt(c, k);
}
// ... and the SHAPE its launch takes (with_tile_shape's code): the class's own, or -- PreSync's sweep over classes 1 .. 4 -- the
// smallest sub-shape that holds the largest frame of the class IN THE SELECTION (the one-wave kernels have always followed
// cls_max_n[0] this way).  A clip of 1500-track frames sweeps 1536 rows per hypothesis instead of 2048, one of 4500-track
// frames 4608 instead of 6144 (the rule itself: window_plan.hpp, plan_sub_shape -- checked on the CPU); the bits are those of the class's own shape (a thread adds its rows in order, rows beyond the
// frame add exact zeros; the eight-wave shapes among themselves alike).  RSSYNC_NO_SUBSHAPES=1: the class's own shape always.
template <int MODE>
int lmeds_shape(const rship_ctx* c, int k) {
    const int full = lmeds_rpt(c, k);
    if (MODE != 0 || c->no_subshapes) return full;
    if (lmeds_kind(c, k) == LmedsKind::Small) return rs::plan_small_rows(c->cls_max_n[0], true); // (257 .. 512 tracks: 5 / 6 / 7 rows per lane where that is enough)
    if (k < 1 || k > 4 || lmeds_kind(c, k) != LmedsKind::Tile) return full;
#if RSSYNC_TEST_VARIANTS
    if (c->exact_select) return full; // (the exact-selection variant exists in the classes' own shapes only)
#endif
    return rs::plan_sub_shape(k, c->cls_max_n[k]);
}

template <int MODE>
WinPlan plan_lmeds_window(rship_ctx* c, int k, double step_knots, uint32_t chunk_want) {
    const LmedsKind kind = lmeds_kind(c, k);
    if (kind == LmedsKind::Big) { // (tiles in global memory, general path throughout)
        WinPlan w;
        w.chunk = chunk_want;
        return w;
    }
    const bool small = kind == LmedsKind::Small;
    const int rpt = lmeds_shape<MODE>(c, k);
    uint32_t lo, hi;
    class_bounds(c, k, &lo, &hi);
    const std::vector<rs::FrameDims>& d = plan_dims(c);
    // (the kernel's LDS footprint is only asked for when the compiled-in window does not do: plan_window's first test)
    const bool fits80 = rs::plan_fit((double)kWinMax, class_span(c, k), step_knots, chunk_want) >= std::min(8u, chunk_want);
    const uint32_t fixed = (fits80 || c->force_general) ? 0u : lmeds_dynamic_static_lds<MODE>(rpt, small);
    rs::WinPlan wp = rs::plan_window_frames(d.data(), d.size(), lo, hi, step_knots, chunk_want, small, small ? 20 : tile_shape_wgs(rpt), fixed,
                                            c->lds_per_cu, c->force_general);
    if (!small && !wp.cap && fits80 && !c->force_general && rpt == 16) {
        // the compiled-in window does, but a smaller one in dynamic LDS lets a third workgroup share the CU (window_plan.hpp)
        const uint32_t cap = rs::smaller_window_for_occupancy(class_span(c, k), step_knots, wp.chunk, lmeds_static_lds<MODE>(rpt),
                                                              lmeds_dynamic_static_lds<MODE>(rpt, false), lmeds_waves(rpt), c->lds_per_cu);
        if (cap) { wp.cap = cap; wp.whole_pair = true; wp.extra_wg = true; }
    }
    return wp;
}

// one class's part of an LMedS launch: p.slots / p.n_slots / p.chunk / p.n_chunks are set here
template <int MODE>
int launch_lmeds_class(rship_ctx* c, LmedsParams p, const ClassRange& r, const WinPlan& wp) {
    const int k = r.k;
    const LmedsKind kind = lmeds_kind(c, k);
    const int rpt = lmeds_shape<MODE>(c, k);
    if (MODE == 0) c->last_lmeds_shape[k] = kind == LmedsKind::Big ? 0u : (uint32_t)rpt; // (class 0: rows per lane of the one-wave kernel)
    // the tile (the lanes' rows) must hold the largest frame of the list: the kernels index LDS / registers by row without a bound of
    // their own, and since the sub-shapes the capacity follows the selection, not just the class
    if (kind != LmedsKind::Big && (uint32_t)rpt * (kind == LmedsKind::Small ? 64u : (uint32_t)kBlock) < c->cls_max_n[k])
        return set_err(c, "lmeds: the shape chosen for a size class does not hold its largest frame");
    p.slots = r.list;
    p.n_slots = r.count;
    p.chunk = wp.chunk;
    p.n_chunks = (p.n_cand + wp.chunk - 1) / wp.chunk;
    p.win_cap = wp.cap ? wp.cap : (uint32_t)kWinMax;
    if (wp.whole_pair) p.win_whole_pair = 1u;
    const size_t dyn = (size_t)wp.cap * 64u;
    if (kind == LmedsKind::Small) {
        // Frames of up to 512 tracks (the reference's own data: ~130): one wave per (frame, chunk) instead of a
        // four-wave workgroup (kernels/lmeds_small.hpp).
        const uint64_t g64 = (uint64_t)r.count * p.n_chunks;
        if (g64 > 0x7fffffffull) return set_err(c, "lmeds: grid too large");
        const uint32_t g1 = (uint32_t)g64;
        if (wp.cap) {
            switch (rpt) {
                case 1: hipLaunchKernelGGL((lmeds_small_kernel<1, MODE, 0>), dim3(g1), dim3(64), dyn, c->stream, p); break;
                case 2: hipLaunchKernelGGL((lmeds_small_kernel<2, MODE, 0>), dim3(g1), dim3(64), dyn, c->stream, p); break;
                case 3: hipLaunchKernelGGL((lmeds_small_kernel<3, MODE, 0>), dim3(g1), dim3(64), dyn, c->stream, p); break;
                case 4: hipLaunchKernelGGL((lmeds_small_kernel<4, MODE, 0>), dim3(g1), dim3(64), dyn, c->stream, p); break;
                // (5 .. 7: PreSync's sub-shapes, lmeds_shape; GuessMotion's search never asks for them)
                case 5: if constexpr (MODE == 0) { hipLaunchKernelGGL((lmeds_small_kernel<5, 0, 0>), dim3(g1), dim3(64), dyn, c->stream, p); break; }
                case 6: if constexpr (MODE == 0) { hipLaunchKernelGGL((lmeds_small_kernel<6, 0, 0>), dim3(g1), dim3(64), dyn, c->stream, p); break; }
                case 7: if constexpr (MODE == 0) { hipLaunchKernelGGL((lmeds_small_kernel<7, 0, 0>), dim3(g1), dim3(64), dyn, c->stream, p); break; }
                default: hipLaunchKernelGGL((lmeds_small_kernel<8, MODE, 0>), dim3(g1), dim3(64), dyn, c->stream, p); break;
            }
        } else {
            switch (rpt) {
                case 1: hipLaunchKernelGGL((lmeds_small_kernel<1, MODE>), dim3(g1), dim3(64), 0, c->stream, p); break;
                case 2: hipLaunchKernelGGL((lmeds_small_kernel<2, MODE>), dim3(g1), dim3(64), 0, c->stream, p); break;
                case 3: hipLaunchKernelGGL((lmeds_small_kernel<3, MODE>), dim3(g1), dim3(64), 0, c->stream, p); break;
                case 4: hipLaunchKernelGGL((lmeds_small_kernel<4, MODE>), dim3(g1), dim3(64), 0, c->stream, p); break;
                case 5: if constexpr (MODE == 0) { hipLaunchKernelGGL((lmeds_small_kernel<5, 0>), dim3(g1), dim3(64), 0, c->stream, p); break; }
                case 6: if constexpr (MODE == 0) { hipLaunchKernelGGL((lmeds_small_kernel<6, 0>), dim3(g1), dim3(64), 0, c->stream, p); break; }
                case 7: if constexpr (MODE == 0) { hipLaunchKernelGGL((lmeds_small_kernel<7, 0>), dim3(g1), dim3(64), 0, c->stream, p); break; }
                default: hipLaunchKernelGGL((lmeds_small_kernel<8, MODE>), dim3(g1), dim3(64), 0, c->stream, p); break;
            }
        }
        RS_HIP(hipGetLastError());
        return 0;
    }
    if (kind == LmedsKind::Big) {
        // more than 8192 tracks: the slow exact path (kernels/lmeds_big.hpp), tiles in a scratch that a fixed number of
        // workgroups share by walking over the (frame, chunk) items of the class
        const uint32_t rows = big_rows(c->cls_max_n[k]);
        const uint64_t total = (uint64_t)r.count * p.n_chunks;
        uint64_t cap_wgs = 2048;
        if (const char* e = std::getenv("RSSYNC_BIG_WGS")) { const long v = atol(e); if (v >= 64 && v <= 8192) cap_wgs = (uint64_t)v; } // (A/B: how many tiles are live at once)
        uint64_t g1 = total < cap_wgs ? total : cap_wgs;
        const uint64_t per_wg = (uint64_t)rows * kBigScratchFloats * 4;
        const uint64_t fit = ((uint64_t)4 << 30) / per_wg; // at most 4 GB of tiles
        if (g1 > fit) g1 = fit ? fit : 1;
        if (ensure(c, c->big_scratch, (size_t)(g1 * per_wg))) return 1;
        p.scratch = (float*)c->big_scratch.p;
        p.scratch_rows = rows;
        hipLaunchKernelGGL((lmeds_big_kernel<MODE>), dim3((uint32_t)g1), dim3(kBlock), 0, c->stream, p);
        RS_HIP(hipGetLastError());
        return 0;
    }
    // the tile kernel: blocks b and b + 8 share an XCD, the chunks of one frame stay on one XCD (kernels/lmeds.hpp)
    const uint64_t grid64 = (uint64_t)((r.count + 7) / 8) * 8 * p.n_chunks;
    if (grid64 > 0x7fffffffull) return set_err(c, "lmeds: grid too large");
    const uint32_t grid = (uint32_t)grid64;
    if (wp.cap) { // the window in dynamic LDS (gyro rates above ~1.7 kHz)
        if (rpt == 16 && wp.extra_wg) { // (a window small enough for a third workgroup per CU: the instantiation compiled for three)
            allow_dynamic_lds(lmeds_kernel<16, MODE, 1>, dyn);
            hipLaunchKernelGGL((lmeds_kernel<16, MODE, 1>), dim3(grid), dim3(kBlock), dyn, c->stream, p);
        } else if (!with_tile_shape<MODE>(rpt, [&](auto sh) {
                       constexpr int R = decltype(sh)::rpt, B = decltype(sh)::block;
                       allow_dynamic_lds(lmeds_kernel<R, MODE, 0, true, false, B>, dyn);
                       hipLaunchKernelGGL((lmeds_kernel<R, MODE, 0, true, false, B>), dim3(grid), dim3(B), dyn, c->stream, p);
                   })) {
            return set_err(c, "lmeds: unsupported rows-per-thread");
        }
        RS_HIP(hipGetLastError());
        return 0;
    }
#if RSSYNC_TEST_VARIANTS
    if (MODE == 0 && c->exact_select) {
        switch (rpt) {
            case 4: hipLaunchKernelGGL((lmeds_kernel<4, 0, kWinMax, false>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
            case 8: hipLaunchKernelGGL((lmeds_kernel<8, 0, kWinMax, false>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
            case 16: hipLaunchKernelGGL((lmeds_kernel<16, 0, kWinMax, false>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
            case 24: allow_dynamic_lds(lmeds_kernel<24, 0, kWinMax, false>, 0);
                     hipLaunchKernelGGL((lmeds_kernel<24, 0, kWinMax, false>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
            case 32: hipLaunchKernelGGL((lmeds_kernel<16, 0, kWinMax, false, false, kWideBlock>), dim3(grid), dim3(kWideBlock), 0, c->stream, p); break;
            default: return set_err(c, "lmeds: unsupported rows-per-thread");
        }
        RS_HIP(hipGetLastError());
        return 0;
    }
#endif
    if (!with_tile_shape<MODE>(rpt, [&](auto sh) {
            constexpr int R = decltype(sh)::rpt, B = decltype(sh)::block;
            if (B != kBlock || R > 16) allow_dynamic_lds(lmeds_kernel<R, MODE, kWinMax, true, false, B>, 0);
            hipLaunchKernelGGL((lmeds_kernel<R, MODE, kWinMax, true, false, B>), dim3(grid), dim3(B), 0, c->stream, p);
        }))
        return set_err(c, "lmeds: unsupported rows-per-thread");
    RS_HIP(hipGetLastError());
    return 0;
}

// The LMedS launches of one call: every size class of the selection with its own kernel, window plan and chunking (the
// results are indexed by candidate and slot, so the classes' chunkings do not meet).  main_cap / main_chunk: what the
// class with the most slots used (rship_window_info).
template <int MODE>
int launch_lmeds(rship_ctx* c, const LmedsParams& p, double step_knots, uint32_t chunk_want, uint32_t* main_cap, uint32_t* main_chunk) {
    ProfScope ps(c, MODE == 1 ? RSHIP_K_INIT : RSHIP_K_LMEDS);
    const int mk = main_class(c);
    if (MODE == 0) memset(c->last_lmeds_shape, 0, sizeof(c->last_lmeds_shape)); // (rship_lmeds_shapes: 0 for the classes this sweep does not launch)
    for (const ClassRange& r : class_ranges(c)) {
        const WinPlan wp = plan_lmeds_window<MODE>(c, r.k, step_knots, chunk_want);
        if (r.k == mk) {
            if (main_cap) *main_cap = wp.cap;
            if (main_chunk) *main_chunk = wp.chunk;
        }
        if (launch_lmeds_class<MODE>(c, p, r, wp)) return 1;
    }
    return 0;
}

// the two device counters of the fp64 rows (PreSync pairs; GuessMotion searches), zeroed when first allocated
int ensure_redo_count(rship_ctx* c) {
    if (c->redo_count.p) return 0;
    if (ensure(c, c->redo_count, 16)) return 1;
    RS_HIP(hipMemsetAsync(c->redo_count.p, 0, 16, c->stream));
    return 0;
}

// The fp64-rows form of a PreSync sweep (kernels/lmeds.hpp, "fp64 rows"): per size class the SAME grid and chunking as the
// fp32 launch (the plan is a function of the context's state, which has not changed since), R64 instantiations; a
// workgroup whose (frame, chunk) has no flagged candidate leaves at once.  Class 5 has taken its fp64 rows in the first
// launch already (kernels/lmeds_big.hpp).
int launch_lmeds_redo(rship_ctx* c, const LmedsParams& p_in, double step_knots, uint32_t chunk_want) {
    ProfScope ps(c, RSHIP_K_LMEDS);
    for (const ClassRange& r : class_ranges(c)) {
        const LmedsKind kind = lmeds_kind(c, r.k);
        if (kind == LmedsKind::Big) continue;
        const WinPlan wp = plan_lmeds_window<0>(c, r.k, step_knots, chunk_want);
        const int rpt = lmeds_rpt(c, r.k);
        LmedsParams p = p_in;
        p.slots = r.list;
        p.n_slots = r.count;
        p.chunk = wp.chunk;
        p.n_chunks = (p.n_cand + wp.chunk - 1) / wp.chunk;
        if (kind == LmedsKind::Small) {
            const uint32_t g1 = (uint32_t)((uint64_t)r.count * p.n_chunks); // (the fp32 launch has checked the size)
            switch (rpt) {
                case 1: hipLaunchKernelGGL((lmeds_small_kernel<1, 0, 0, true>), dim3(g1), dim3(64), 0, c->stream, p); break;
                case 2: hipLaunchKernelGGL((lmeds_small_kernel<2, 0, 0, true>), dim3(g1), dim3(64), 0, c->stream, p); break;
                case 3: hipLaunchKernelGGL((lmeds_small_kernel<3, 0, 0, true>), dim3(g1), dim3(64), 0, c->stream, p); break;
                case 4: hipLaunchKernelGGL((lmeds_small_kernel<4, 0, 0, true>), dim3(g1), dim3(64), 0, c->stream, p); break;
                default: hipLaunchKernelGGL((lmeds_small_kernel<8, 0, 0, true>), dim3(g1), dim3(64), 0, c->stream, p); break;
            }
        } else {
            const uint32_t grid = (uint32_t)((uint64_t)((r.count + 7) / 8) * 8 * p.n_chunks);
            switch (rpt) {
                case 4: hipLaunchKernelGGL((lmeds_kernel<4, 0, 0, true, true>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
                case 8: hipLaunchKernelGGL((lmeds_kernel<8, 0, 0, true, true>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
                case 16: hipLaunchKernelGGL((lmeds_kernel<16, 0, 0, true, true>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
                case 24: allow_dynamic_lds(lmeds_kernel<24, 0, 0, true, true>, 0);
                         hipLaunchKernelGGL((lmeds_kernel<24, 0, 0, true, true>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
                case 32: allow_dynamic_lds(lmeds_kernel<16, 0, 0, true, true, kWideBlock>, 0);
                         hipLaunchKernelGGL((lmeds_kernel<16, 0, 0, true, true, kWideBlock>), dim3(grid), dim3(kWideBlock), 0, c->stream, p); break;
                default: return set_err(c, "lmeds (fp64 rows): unsupported rows-per-thread");
            }
        }
        RS_HIP(hipGetLastError());
    }
    return 0;
}

// does class k's trial kernel use its compiled-in five windows of 80 knots? (whole pairs only: where the 80 knots are
// enough just because the two ends are staged separately, the dynamic instantiation runs)
bool loss_fixed80(const rship_ctx* c, int k) {
    return c->cls_cap64[k] == (uint32_t)kWinMax && (class_span(c, k) + 1.f <= (float)kWinMax || c->force_general);
}
uint32_t loss_nb_run(const rship_ctx* c, int k) {
    return loss_fixed80(c, k) ? (uint32_t)kLossBatch : std::max(1u, std::min((uint32_t)kLossBatchWide, kLossWinBytes / (c->cls_cap64[k] * 128u)));
}

// K1 over the slots [slot0, slot0 + count) of the selection (count = 0: all), one launch per size class
template <bool GRAD, bool SIMPLE>
int launch_loss64(rship_ctx* c, const Loss64Params& p_in, hipStream_t st = nullptr, uint32_t slot0 = 0, uint32_t count = 0) {
    if (!st) st = c->stream;
    ProfScope ps(c, GRAD ? RSHIP_K_LOSS_GRAD : RSHIP_K_LOSS);
    for (const ClassRange& r : class_ranges(c, slot0, count)) {
        Loss64Params p = p_in;
        const int k = r.k;
        p.slots = r.list;
        p.slot0 = r.list ? 0u : c->h_cls_slots[r.pos0]; // (one class: the list is the identity and entry = slot)
        // The spline windows.  Trials (no gradient): five delays per pass over the rows, their 80-knot windows compiled into
        // the kernel's LDS, while the class's frames fit 80 knots (gyro rates up to ~1.7 kHz); wider frames take the
        // dynamic-LDS instantiation with cap64 knots per window, as many per pass as 51 KB hold (at most three).  The gradient
        // launch has one window, always in dynamic LDS.
        const uint32_t cap64 = cap64_of(c, k);
        p.win_cap = cap64;
        p.win_compact = compact_of(c, k) ? 1u : 0u;
        const bool fixed80 = !GRAD && loss_fixed80(c, k);
        p.nb_run = GRAD ? 1u : loss_nb_run(c, k);
        const size_t dyn = fixed80 ? 0 : (size_t)p.nb_run * cap64 * 128u, dyn_small = win64_bytes(c, k);
        // frames of up to 512 tracks (the reference's own: ~130): one wave per slot instead of a four-wave workgroup that
        // half idles -- the same sums in the same order (loss64_wave), four times as many slots on the chip
        if (k == 0 && !c->no_small_loss) {
            hipLaunchKernelGGL((loss64_small_kernel<GRAD, SIMPLE>), dim3(r.count), dim3(64), dyn_small, st, p);
            RS_HIP(hipGetLastError());
            continue;
        }
        const int rpt = k == 0 ? rpt_for(c->cls_max_n[0]) : class_rpt(c, k);
#define RS_LOSS_CASE(R)                                                                                                        \
    case R:                                                                                                                    \
        if constexpr (!GRAD) {                                                                                                 \
            if (fixed80) { hipLaunchKernelGGL((loss64_kernel<R, GRAD, SIMPLE, kWinMax>), dim3(r.count), dim3(kBlock), 0, st, p); break; } \
        }                                                                                                                      \
        hipLaunchKernelGGL((loss64_kernel<R, GRAD, SIMPLE, 0>), dim3(r.count), dim3(kBlock), dyn, st, p);                      \
        break;
        switch (rpt) {
            RS_LOSS_CASE(0)
            RS_LOSS_CASE(4)
            RS_LOSS_CASE(8)
            RS_LOSS_CASE(16)
            RS_LOSS_CASE(24)
            RS_LOSS_CASE(32)
            default: return set_err(c, "loss: unsupported rows-per-thread");
        }
#undef RS_LOSS_CASE
        RS_HIP(hipGetLastError());
    }
    return 0;
}

// The workgroup shape of the motion kernel fixes the order in which a frame's row terms are added; it follows the frame's
// own size class (one wave up to 512 tracks, four above), so that a frame gets the same sums whichever selection, device or
// rank it is part of.
constexpr uint32_t kOrderMinSlots = 512; // fewer workgroups than the chip holds at once: nothing to order

// K3 over the slots [slot0, slot0 + count) of the selection (count = 0: all), one launch per size class; p.order is the
// class-sorted launch order (c->mo_order: entry e of the launch list is slot order[e])
int launch_motion64(rship_ctx* c, const Motion64Params& p_in, hipStream_t st = nullptr, uint32_t slot0 = 0, uint32_t count = 0) {
    if (!st) st = c->stream;
    ProfScope ps(c, RSHIP_K_MOTION);
    for (const ClassRange& r : class_ranges(c, slot0, count)) {
        Motion64Params p = p_in;
        const int k = r.k;
        p.slot0 = r.pos0; // (an index into order[]; with one class the list is the identity and this is the first slot)
        p.win_cap = cap64_of(c, k);
        p.win_compact = compact_of(c, k) ? 1u : 0u;
        const size_t dyn = win64_bytes(c, k); // the spline window (used once, for the rows of P)
        p.win_bytes = (uint32_t)dyn;
        const uint32_t cnt = r.count;
        // One wave per frame up to 512 tracks: the evaluations of a frame are dominated by their fixed part (five
        // wave reductions, the uniform L-BFGS bookkeeping) which every wave of a workgroup repeats, and a one-wave
        // frame needs no LDS exchange or barrier at all -- the reference's own workload (~130 tracks) ran 3x as
        // many frames per CU this way.  Above that, four waves: with 8 / 16 waves (4 / 2 rows per thread at 2048
        // tracks) the launch took 7.8 / 12.4 ms per bench step instead of 4.75, and with two waves 5.15: the kernel is
        // bound by the work per evaluation, not by its slowest frame.
        if (k == 0) {
            const uint32_t n = c->cls_max_n[0];
            if (n <= 64) hipLaunchKernelGGL((opt_motion64_kernel<1, 1>), dim3(cnt), dim3(64), dyn, st, p);
            else if (n <= 128) hipLaunchKernelGGL((opt_motion64_kernel<2, 1>), dim3(cnt), dim3(64), dyn, st, p);
            else if (n <= 192) hipLaunchKernelGGL((opt_motion64_kernel<3, 1>), dim3(cnt), dim3(64), dyn, st, p);
            else if (n <= 256) hipLaunchKernelGGL((opt_motion64_kernel<4, 1>), dim3(cnt), dim3(64), dyn, st, p);
            else hipLaunchKernelGGL((opt_motion64_kernel<8, 1>), dim3(cnt), dim3(64), dyn, st, p);
        } else if (k == 1) hipLaunchKernelGGL((opt_motion64_kernel<4, 4>), dim3(cnt), dim3(256), dyn, st, p);
        else if (k == 2) hipLaunchKernelGGL((opt_motion64_kernel<8, 4>), dim3(cnt), dim3(256), dyn, st, p);
        else if (k == 3 && class_rpt(c, 3) == 16) hipLaunchKernelGGL((opt_motion64_kernel<16, 4>), dim3(cnt), dim3(256), dyn, st, p);
        else if (k == 3) hipLaunchKernelGGL((opt_motion64_kernel<24, 4>), dim3(cnt), dim3(256), dyn, st, p);
        else if (k == 4) hipLaunchKernelGGL((opt_motion64_kernel<32, 4>), dim3(cnt), dim3(256), dyn, st, p);
        else { // rows of P in global memory (per entry of the class's list, set up by fill_motion), as many per thread as the frame needs
            if (!p.scratch || p.scratch_rows < c->cls_max_n[5]) return set_err(c, "motion: no scratch for frames of more than 8192 tracks");
            p.scratch0 = r.pos0 - c->cls_off[5];
            hipLaunchKernelGGL((opt_motion64_kernel<0, 4>), dim3(cnt), dim3(256), dyn, st, p);
        }
        RS_HIP(hipGetLastError());
        // the order of the NEXT launch over these slots, from this one's evaluation counts
        if (p.evals_out && p.order && c->mo_order.p && p.max_iters > 0 && !p.simple_k && cnt >= kOrderMinSlots) {
            hipLaunchKernelGGL(motion_order_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t*)p.evals_out, (const uint32_t*)c->cls_slots.p,
                               (uint32_t*)c->mo_order.p, r.pos0, cnt);
            RS_HIP(hipGetLastError());
        }
    }
    return 0;
}

// rows x (chunk sums, window sums) of in[rows][cols] under the current plan, into c->chunk_out / c->win_out
// (slot b of n_slots: a batch of sweeps keeps its sums side by side -- rship_presync_batch_begin)
int launch_plan_sum(rship_ctx* c, const double* in, uint32_t rows, uint32_t cols, uint32_t slot = 0, uint32_t n_slots = 1) {
    if (!c->plan_wins) return set_err(c, "no reduction plan set");
    const size_t cs = (size_t)rows * (c->plan_chunks + 1), ws = (size_t)rows * c->plan_wins;
    if (n_slots > 1 && (c->chunk_out.cap < cs * n_slots * 8 || c->win_out.cap < ws * n_slots * 8)) return set_err(c, "plan sum: the batch's buffers were not sized");
    if (n_slots == 1 && (ensure(c, c->chunk_out, cs * 8) || ensure(c, c->win_out, ws * 8))) return 1;
    ProfScope ps(c, RSHIP_K_REDUCE);
    hipLaunchKernelGGL(plan_sum_kernel, dim3(rows * c->plan_wins), dim3(64), 0, c->stream, in, cols,
                       c->plan_has_idx ? (const uint32_t*)c->plan_idx.p : nullptr, (const uint32_t*)c->plan_chunk_off.p,
                       c->plan_chunks, (const uint32_t*)c->plan_win_off.p, c->plan_wins, (double*)c->chunk_out.p + cs * slot,
                       (double*)c->win_out.p + ws * slot);
    RS_HIP(hipGetLastError());
    return 0;
}

// queue the copy of the last plan sum (rows x windows, then rows x chunks) into pinned memory at `at`
int queue_sums_to_host(rship_ctx* c, uint32_t rows, size_t at, size_t* off_chunk, size_t* end) {
    const size_t wb = (size_t)rows * c->plan_wins * 8, cb = (size_t)rows * c->plan_chunks * 8;
    if (ensure_pinned(c, at + wb + cb + 64)) return 1;
    RS_HIP(hipMemcpyAsync((char*)c->pinned + at, c->win_out.p, wb, hipMemcpyDeviceToHost, c->stream));
    if (cb) RS_HIP(hipMemcpyAsync((char*)c->pinned + at + wb, c->chunk_out.p, cb, hipMemcpyDeviceToHost, c->stream));
    *off_chunk = at + wb;
    *end = at + wb + cb;
    return 0;
}

// Every entry point works on the context's device, whatever the calling thread's current device
// is (a host with several GPUs in one process), and leaves the caller's choice as it found it.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const rship_ctx* c) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) (void)hipSetDevice(c->device);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// kd[n] then fd[n] in one device buffer, staged through the host so that it is ONE copy per call
// (the Sync loop uploads delays three times per outer iteration; API calls are what it waits for)
int upload_delays(rship_ctx* c, const int32_t* kd, const float* fd, size_t n) {
    if (ensure(c, c->kd, n * 8)) return 1;
    c->h_delays.resize(2 * n);
    memcpy(c->h_delays.data(), kd, n * 4);
    memcpy(c->h_delays.data() + n, fd, n * 4);
    RS_HIP(hipMemcpyAsync(c->kd.p, c->h_delays.data(), n * 8, hipMemcpyHostToDevice, c->stream));
    c->d_fd = (const float*)((const int32_t*)c->kd.p + n);
    return 0;
}
// the same for the fp64 kernels: kd[n] (int32), padded to 8 bytes, then fd[n] (double)
int upload_delays64(rship_ctx* c, const int32_t* kd, const double* fd, size_t n) {
    const size_t pad = (n + 1) / 2 * 2; // int32 slots before the doubles
    if (ensure(c, c->kd64, pad * 4 + n * 8)) return 1;
    c->h_delays64.resize(pad + 2 * n);
    memcpy(c->h_delays64.data(), kd, n * 4);
    memcpy(c->h_delays64.data() + pad, fd, n * 8);
    RS_HIP(hipMemcpyAsync(c->kd64.p, c->h_delays64.data(), pad * 4 + n * 8, hipMemcpyHostToDevice, c->stream));
    c->d_fd64 = (const double*)((const int32_t*)c->kd64.p + pad);
    return 0;
}
Rays64 rays64_of(const rship_ctx* c) {
    const double2* q0 = (const double2*)c->rays64.p;
    const size_t tr = (size_t)c->total_rays;
    return Rays64{q0, q0 + tr, q0 + 2 * tr, q0 + 3 * tr};
}
int check_ready(rship_ctx* c) {
    if (!c->n_knots) return set_err(c, "no gyro spline uploaded");
    if (!c->n_frames) return set_err(c, "no frames uploaded");
    if (!c->n_sel) return set_err(c, "no frames selected");
    return 0;
}

} // namespace

// ---- native exchange: RCCL, resolved at run time -----------------------------------------
// The four entry points used, with the signatures of <rccl/rccl.h> (ncclResult_t is an int enum,
// ncclUniqueId a 128-byte struct passed by value, ncclDouble = 8, ncclSum = 0).
namespace {
struct RcclId { char bytes[128]; };
using rccl_get_id_fn = int (*)(RcclId*);
using rccl_init_fn = int (*)(void**, int, RcclId, int);
using rccl_allreduce_fn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
using rccl_destroy_fn = int (*)(void*);

using rccl_abort_fn = int (*)(void*);

// WHICH librccl: the one that sits beside the HIP runtime THIS library is bound to.  A process may hold two HIP
// runtimes -- a PyTorch wheel ships its own libamdhip64 / libhsa-runtime64 / librccl next to the system's ROCm -- and a
// communicator only works with streams and device pointers of the runtime its librccl was built against: torch's
// librccl, found "already loaded" in the process, refused this library's stream (ncclCommInitRank -> 1, measured in
// round 4), the system's beside /opt/rocm/lib/libamdhip64 works; in a process where this library had bound to
// torch's runtime it would be the other way round.  So: the directory of the libamdhip64 that hipStreamSynchronize
// resolves to, then the usual names.
void* rccl_sym(rship_ctx* c, const char* name) {
    if (!c->rccl_lib) {
        Dl_info info{};
        std::string dir, hip_path;
        if (dladdr(reinterpret_cast<const void*>(&hipStreamSynchronize), &info) && info.dli_fname) {
            hip_path = info.dli_fname;
            const size_t cut = hip_path.rfind('/');
            if (cut != std::string::npos) dir = hip_path.substr(0, cut + 1);
        }
        std::vector<std::string> tries;
        if (!dir.empty()) { tries.push_back(dir + "librccl.so.1"); tries.push_back(dir + "librccl.so"); }
        for (const char* lib : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) tries.push_back(lib);
        for (size_t i = 0; i < tries.size() && !c->rccl_lib; ++i) {
            c->rccl_lib = dlopen(tries[i].c_str(), RTLD_NOW | RTLD_LOCAL);
            if (c->rccl_lib)
                c->rccl_path = tries[i] + (i < 2 && !dir.empty() ? " (beside the HIP runtime this library is bound to, " + hip_path + ")"
                                                                 : " (opened by name)");
        }
        if (!c->rccl_lib) {
            set_err(c, std::string("rccl: cannot open librccl: ") + dlerror());
            return nullptr;
        }
    }
    void* f = dlsym(c->rccl_lib, name);
    if (!f) set_err(c, std::string("rccl: missing symbol ") + name);
    return f;
}

// A local failure while peers may be waiting inside a collective: abort the communicator, so that their pending
// all-reduces end with an error instead of waiting for this rank for ever.
void rccl_abort_comm(rship_ctx* c) {
    if (!c->rccl_comm) return;
    std::string keep;
    { std::lock_guard<std::mutex> lock(g_err_mutex); keep = c->err; }
    if (auto ab = (rccl_abort_fn)rccl_sym(c, "ncclCommAbort")) (void)ab(c->rccl_comm);
    c->rccl_comm = nullptr;
    set_err(c, keep + " -- the RCCL communicator was aborted so that the other ranks fail instead of waiting");
}
} // namespace

extern "C" {

// An indexing bound, not a kernel one: 2^24 tracks per frame (the packing kernel's grid), 2^32 rays per problem
// (32-bit offsets).  Up to kMaxRpt * kBlock = 8192 tracks per frame the kernels keep a frame's rows in registers /
// LDS; beyond that they take their slow exact variants (rows in global memory).
int rship_max_tracks(void) { return 1 << 24; }
int rship_has_device_loop(void) { return 1; }

// Staging memory for the host solver: pinned, so that rship_upload_raw is a true asynchronous DMA
// at PCIe rate (a pageable source is staged through a bounce buffer at ~1 GB/s: round 1's 0.24 s
// for 268 MB).  Falls back to pageable memory if the runtime refuses to pin.
void* rship_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess && p) return p;
    (void)hipGetLastError();
    return nullptr;
}
void rship_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int rship_create(rship_ctx** out, int device) {
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return 2; // no GPU: the product path has no CPU fallback
    rship_ctx* c = new rship_ctx();
    if (const char* s = std::getenv("RSSYNC_NO_SMALL_LMEDS")) c->no_small_lmeds = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_NO_SUBSHAPES")) c->no_subshapes = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_K2_EXACT_SELECT")) c->exact_select = s[0] && s[0] != '0';
#if !RSSYNC_TEST_VARIANTS
    if (c->exact_select) { // (a test that asked for the variant must not silently compare the product with itself)
        fprintf(stderr, "rssync: RSSYNC_K2_EXACT_SELECT needs the test-variants build (tools/k2_build_variant.sh testvariants -DRSSYNC_TEST_VARIANTS=1)\n");
        delete c;
        return 5;
    }
#endif
    if (const char* s = std::getenv("RSSYNC_NO_SMALL_LOSS")) c->no_small_loss = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_FORCE_BIG")) c->force_big = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_FORCE_GENERAL_SPLINE")) c->force_general = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_NO_COMPACT_WINDOW")) c->no_compact = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_NO_FP64_ROWS")) c->no_fp64_rows = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_EXEC_BIG_MAX")) { const int v = atoi(s); if (v >= 0) c->exec_big_max = (uint32_t)v; }
    if (const char* s = std::getenv("RSSYNC_EXEC_BIG_SHARE")) { const int v = atoi(s); if (v >= 1) c->exec_big_share = (uint32_t)v; }
    if (const char* s = std::getenv("RSSYNC_ONE_WAVE_MAX")) { const int v = atoi(s); if (v >= 64 && v <= 64 * kSmallMaxRpt) c->one_wave_max = (uint32_t)v; }
    if (device >= 0) {
        e = hipSetDevice(device);
        if (e != hipSuccess) { delete c; return 3; }
        c->device = device;
    } else {
        (void)hipGetDevice(&c->device);
    }
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming);
    if (e != hipSuccess) { delete c; return 4; }
    c->stream = c->own_stream;
    {
        int v = 0; // LDS of a compute unit (gfx950: 160 KB): what the dynamic spline windows are planned against
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, c->device) == hipSuccess && v >= 64 * 1024) c->lds_per_cu = v;
        else (void)hipGetLastError();
    }
    *out = c;
    return 0;
}

void rship_destroy(rship_ctx* c) {
    if (!c) return;
    DeviceGuard dev_guard(c);
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    if (c->rccl_comm) {
        if (auto destroy = (rccl_destroy_fn)rccl_sym(c, "ncclCommDestroy")) (void)destroy(c->rccl_comm);
    }
    if (c->rccl_buf.p) (void)hipFree(c->rccl_buf.p);
    for (auto e : c->pool) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    DevBuf* bufs[] = {&c->coef, &c->coef64, &c->raw, &c->rays_a, &c->rays_b, &c->rays64, &c->frames, &c->sel, &c->M, &c->k, &c->grp, &c->grp_off,
                      &c->plan_idx, &c->plan_chunk_off, &c->plan_win_off, &c->chunk_out, &c->win_out, &c->loop_state, &c->kd, &c->kd64, &c->init_h,
                      &c->frame_cost, &c->best_h, &c->costs, &c->part, &c->flags, &c->stats, &c->redo_mask, &c->redo_delays, &c->redo_count, &c->init_delays64, &c->dump,
                      &c->big_scratch, &c->mo_scratch, &c->mo_evals, &c->mo_order,
                      &c->g_ts, &c->g_rates, &c->g_us, &c->g_dq, &c->g_q, &c->g_knots, &c->g_cf, &c->g_status};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (hipStream_t st : c->loop_streams) (void)hipStreamDestroy(st);
    if (c->loop_ready) (void)hipEventDestroy(c->loop_ready);
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* rship_last_error(const rship_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rship_set_option(rship_ctx* c, int option, int value) {
    switch (option) {
        case RSHIP_OPT_LBFGS_REEVAL: c->lbfgs_reeval = value != 0; return 0;
        case RSHIP_OPT_TRACKS_HINT: return 0; // (rounds 2-4: the kernel shapes followed the problem's largest frame; they follow each frame's own size now)
        default: return set_err(c, "set_option: unknown option");
    }
}

int rship_set_stream(rship_ctx* c, void* hip_stream) {
    DeviceGuard dev_guard(c);
    RS_HIP(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return 0;
}

// ---- gyro pipeline ----------------------------------------------------------------------------------------
namespace {

constexpr uint32_t kMaxKnots = rs::kMaxKnots;

const rs::SplinePivots* spline_pivots() {
    static const rs::SplinePivots t = [] {
        rs::SplinePivots p;
        p.cp[0] = 0.0;
        for (int i = 1; i < rs::kSplinePivots; ++i) p.cp[i] = rs::spline_next_pivot(p.cp[i - 1]);
        return p;
    }();
    return &t;
}

// knots (c->g_knots, n of them) -> both coefficient tables
int spline_from_knots(rship_ctx* c, uint32_t n, double sample_rate) {
    if (n < 2) return set_err(c, "spline: need >= 2 knots");
    if (n > kMaxKnots) return set_err(c, "spline: too many knots");
    const rs::SplinePivots& piv = *spline_pivots();
    if (rs::spline_next_pivot(piv.cp[rs::kSplinePivots - 1]) != piv.cp[rs::kSplinePivots - 1])
        return set_err(c, "spline: pivots not stationary"); // cannot happen in IEEE double (stationary after ~30 rows)
    const size_t n16 = (size_t)n * 16;
    if (ensure(c, c->coef64, n16 * 8) || ensure(c, c->coef, n16 * 4) || ensure(c, c->g_cf, (size_t)n * 32)) return 1;
    SplineParams p{};
    p.knots = (const double*)c->g_knots.p;
    p.cf = (double*)c->g_cf.p;
    p.coef64 = (double*)c->coef64.p;
    p.coef32 = (float*)c->coef.p;
    p.n = n;
    p.piv = piv;
    const uint32_t threads = ((n + kSplineRun - 1) / kSplineRun) * 4;
    {
        ProfScope ps(c, RSHIP_K_GYRO);
        hipLaunchKernelGGL(spline_forward_kernel, dim3((threads + 255) / 256), dim3(256), 0, c->stream, p);
        hipLaunchKernelGGL(spline_finish_kernel, dim3((threads + 255) / 256), dim3(256), 0, c->stream, p);
    }
    RS_HIP(hipGetLastError());
    c->n_knots = n;
    c->fs = sample_rate;
    return 0;
}

// shared tail of the two timestamped routes: order check done, ts (us) and quats on the device
// what the host makes of a status record (the reference's order of complaints, core_private.cpp:156-184)
int gyro_status_to_result(rship_ctx* c, const GyroStatus* h, int grid_status, const int64_t* d_ts, rship_gyro_result* out) {
    out->status = RSHIP_GYRO_OK;
    if (h->bad_input) out->status = RSHIP_GYRO_BAD_INPUT;
    else if (grid_status == RSHIP_GYRO_BAD_RATE || grid_status == RSHIP_GYRO_TOO_LARGE) out->status = grid_status;
    else if (h->out_of_order != kNoIndex) {
        int64_t pair[2];
        out->status = RSHIP_GYRO_OUT_OF_ORDER;
        out->bad_pos = h->out_of_order;
        RS_HIP(hipMemcpy(pair, d_ts + (h->out_of_order - 1), 16, hipMemcpyDeviceToHost));
        out->bad_a = pair[0];
        out->bad_b = pair[1];
    } else if (grid_status != RSHIP_GYRO_OK) out->status = grid_status;
    else if (h->bad_knot) out->status = RSHIP_GYRO_BAD_KNOT;
    else if (!std::isfinite(out->fs)) out->status = RSHIP_GYRO_BAD_RATE;
    else if (!std::isfinite(out->start)) out->status = RSHIP_GYRO_BAD_START;
    return 0;
}

// status_slot: which record of g_status the kernels report into; wait = false: nothing is read back (a batch of
// orientations: rship_gyro_rates_integrate_enqueue, the records are looked at by rship_gyro_batch_status)
int resample_and_solve(rship_ctx* c, const int64_t* d_ts, const double* d_quats, uint32_t n, int64_t first, int64_t last,
                       rship_gyro_result* out, uint32_t status_slot = 0, bool wait = true) {
    const int grid_status = rs::grid_of(first, last, n, kMaxKnots, out);
    GyroStatus* st = (GyroStatus*)c->g_status.p + status_slot;
    const bool have_grid = grid_status == RSHIP_GYRO_OK;
    if (have_grid) {
        if (ensure(c, c->g_knots, (size_t)out->n_knots * 32)) return 1;
        GyroResampleParams p{};
        p.ts = d_ts; p.quats = d_quats; p.knots = (double*)c->g_knots.p; p.st = st;
        p.n = n; p.m = out->n_knots; p.first_sample = out->first_sample; p.sr_hz = (uint64_t)out->fs;
        {
            ProfScope ps(c, RSHIP_K_GYRO);
            hipLaunchKernelGGL(gyro_resample_kernel, dim3((p.m + 255) / 256), dim3(256), 0, c->stream, p);
        }
        RS_HIP(hipGetLastError());
        if (spline_from_knots(c, out->n_knots, out->fs)) return 1;
    }
    if (!wait) {
        out->status = grid_status == RSHIP_GYRO_OK ? RSHIP_GYRO_OK : grid_status; // (the device's flags: rship_gyro_batch_status)
        return 0;
    }
    // one look at the status; the reference's order of complaints (core_private.cpp:156-184)
    if (ensure_pinned(c, 64)) return 1;
    GyroStatus* h = (GyroStatus*)c->pinned;
    RS_HIP(hipMemcpyAsync(h, st, sizeof(GyroStatus), hipMemcpyDeviceToHost, c->stream));
    if (sync_stream(c)) return 1;
    const GyroStatus hs = *h;
    if (gyro_status_to_result(c, &hs, grid_status, d_ts, out)) return 1;
    if (out->status != RSHIP_GYRO_OK) c->n_knots = 0; // whatever was built is not a table to sweep over
    return 0;
}

} // namespace

int rship_gyro_uniform(rship_ctx* c, const double* quats, uint32_t n, double sample_rate) {
    DeviceGuard dev_guard(c);
    if (n < 2) return set_err(c, "spline: need >= 2 knots");
    if (n > kMaxKnots) return set_err(c, "spline: too many knots");
    if (ensure(c, c->g_knots, (size_t)n * 32)) return 1;
    RS_HIP(hipMemcpyAsync(c->g_knots.p, quats, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    if (spline_from_knots(c, n, sample_rate)) return 1;
    return sync_stream(c); // the caller's array may go away
}

int rship_gyro_timestamped(rship_ctx* c, const int64_t* ts_us, const double* quats, uint32_t n, rship_gyro_result* out) {
    DeviceGuard dev_guard(c);
    if (n < 2) return set_err(c, "gyro: need >= 2 samples");
    if (ensure(c, c->g_us, (size_t)n * 8) || ensure(c, c->g_q, (size_t)n * 32) || ensure(c, c->g_status, sizeof(GyroStatus) * 256)) return 1;
    RS_HIP(hipMemcpyAsync(c->g_us.p, ts_us, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemcpyAsync(c->g_q.p, quats, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, RSHIP_K_GYRO);
        hipLaunchKernelGGL(gyro_status_reset_kernel, dim3(1), dim3(1), 0, c->stream, (GyroStatus*)c->g_status.p);
        hipLaunchKernelGGL(gyro_order_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const int64_t*)c->g_us.p,
                           (const double*)c->g_q.p, n, (GyroStatus*)c->g_status.p);
    }
    RS_HIP(hipGetLastError());
    return resample_and_solve(c, (const int64_t*)c->g_us.p, (const double*)c->g_q.p, n, ts_us[0], ts_us[n - 1], out);
}

int rship_gyro_rates_upload(rship_ctx* c, const double* ts_s, const double* rates, uint32_t n) {
    DeviceGuard dev_guard(c);
    if (n < 2) return set_err(c, "gyro: need >= 2 samples");
    if (ensure(c, c->g_ts, (size_t)n * 8) || ensure(c, c->g_rates, (size_t)n * 24)) return 1;
    RS_HIP(hipMemcpyAsync(c->g_ts.p, ts_s, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemcpyAsync(c->g_rates.p, rates, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipStreamSynchronize(c->stream)); // the caller's arrays may go away
    c->g_n = n;
    // the grid only needs the two end timestamps, truncated as the kernel truncates them (core_testcode.cpp:48-50)
    c->g_first_us = (int64_t)(ts_s[0] * 1000000);
    c->g_last_us = (int64_t)(ts_s[n - 1] * 1000000);
    return 0;
}

namespace {
constexpr uint32_t kGyroStatusSlots = 256; // records of g_status: slot 0 for the plain calls, 1 .. for a batch of orientations
int gyro_rates_integrate(rship_ctx* c, const int32_t axis[3], const double sign[3], rship_gyro_result* out, uint32_t status_slot, bool wait);
} // namespace
int rship_gyro_rates_integrate(rship_ctx* c, const int32_t axis[3], const double sign[3], rship_gyro_result* out) {
    DeviceGuard dev_guard(c);
    return gyro_rates_integrate(c, axis, sign, out, 0, true);
}
// The same without waiting: the kernels are enqueued, the table they build is the one the NEXT enqueued sweep reads, and
// the device's complaints (non-finite input, a non-finite knot) go to status record `slot` (1 .. 255), to be looked at by
// rship_gyro_batch_status after the batch has been collected.  out: the grid (a function of the end timestamps alone).
int rship_gyro_rates_integrate_enqueue(rship_ctx* c, const int32_t axis[3], const double sign[3], uint32_t slot, rship_gyro_result* out) {
    DeviceGuard dev_guard(c);
    if (!slot || slot >= kGyroStatusSlots) return set_err(c, "gyro: status slot out of range");
    return gyro_rates_integrate(c, axis, sign, out, slot, false);
}
// statuses of the slots 1 .. n of a batch (after the stream has been waited for): out[i].status as rship_gyro_rates_integrate
// would have reported it for the (i + 1)-th enqueued orientation
int rship_gyro_batch_status(rship_ctx* c, uint32_t n, int32_t* status) {
    DeviceGuard dev_guard(c);
    if (n >= kGyroStatusSlots) return set_err(c, "gyro: too many status slots");
    if (!n) return 0;
    std::vector<GyroStatus> h(n);
    RS_HIP(hipStreamSynchronize(c->stream));
    RS_HIP(hipMemcpy(h.data(), (const GyroStatus*)c->g_status.p + 1, (size_t)n * sizeof(GyroStatus), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) {
        rship_gyro_result r{};
        const int grid_status = rs::grid_of(c->g_first_us, c->g_last_us, c->g_n, kMaxKnots, &r);
        if (gyro_status_to_result(c, &h[i], grid_status, (const int64_t*)c->g_us.p, &r)) return 1;
        status[i] = r.status;
    }
    return 0;
}
namespace {
int gyro_rates_integrate(rship_ctx* c, const int32_t axis[3], const double sign[3], rship_gyro_result* out, uint32_t status_slot, bool wait) {
    const uint32_t n = c->g_n;
    if (n < 2) return set_err(c, "gyro: no rates uploaded");
    for (int k = 0; k < 3; ++k)
        if (axis[k] < 0 || axis[k] > 2) return set_err(c, "gyro: axis out of range");
    if (ensure(c, c->g_us, (size_t)n * 8) || ensure(c, c->g_dq, (size_t)n * 32) || ensure(c, c->g_q, (size_t)n * 32) ||
        ensure(c, c->g_status, sizeof(GyroStatus) * kGyroStatusSlots) || ensure(c, c->g_cf, (size_t)(n / kScanSegment + 2) * 64))
        return 1;
    GyroRatesParams p{};
    p.ts = (const double*)c->g_ts.p; p.rates = (const double*)c->g_rates.p;
    p.us = (int64_t*)c->g_us.p; p.dq = (double*)c->g_dq.p; p.st = (GyroStatus*)c->g_status.p + status_slot; p.n = n;
    for (int k = 0; k < 3; ++k) { p.axis[k] = axis[k]; p.sign[k] = sign[k]; }
    {
        ProfScope ps(c, RSHIP_K_GYRO);
        hipLaunchKernelGGL(gyro_status_reset_kernel, dim3(1), dim3(1), 0, c->stream, p.st);
        hipLaunchKernelGGL(gyro_rates_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, p);
        const uint32_t n_seg = (n + kScanSegment - 1) / kScanSegment;
        if (n_seg == 1) {
            hipLaunchKernelGGL(gyro_scan_kernel, dim3(1), dim3(kScanThreads), 0, c->stream, (const double*)c->g_dq.p, (double*)c->g_q.p, n,
                               kScanSegment, (double*)nullptr);
        } else {
            // segment totals -> their running products (the same kernel, one workgroup) -> every later segment times its prefix
            double* tot = (double*)c->g_cf.p; // scratch of the spline solve, not yet in use: [n_seg][4] twice
            double* pre = tot + 4 * (size_t)n_seg;
            hipLaunchKernelGGL(gyro_scan_kernel, dim3(n_seg), dim3(kScanThreads), 0, c->stream, (const double*)c->g_dq.p, (double*)c->g_q.p, n,
                               kScanSegment, tot);
            hipLaunchKernelGGL(gyro_scan_kernel, dim3(1), dim3(kScanThreads), 0, c->stream, (const double*)tot, pre, n_seg, n_seg, (double*)nullptr);
            hipLaunchKernelGGL(gyro_scan_fixup_kernel, dim3((n - kScanSegment + 255) / 256), dim3(256), 0, c->stream, (double*)c->g_q.p, n,
                               kScanSegment, (const double*)pre);
        }
    }
    RS_HIP(hipGetLastError());
    return resample_and_solve(c, (const int64_t*)c->g_us.p, (const double*)c->g_q.p, n, c->g_first_us, c->g_last_us, out, status_slot, wait);
}
} // namespace

int rship_gyro_knots(rship_ctx* c, double* out, uint32_t cap_knots) {
    DeviceGuard dev_guard(c);
    if (cap_knots < c->n_knots) return set_err(c, "gyro_knots: buffer too small");
    if (!c->n_knots) return 0;
    RS_HIP(hipStreamSynchronize(c->stream));
    RS_HIP(hipMemcpy(out, c->g_knots.p, (size_t)c->n_knots * 32, hipMemcpyDeviceToHost));
    return 0;
}

int rship_gyro_table(rship_ctx* c, double* out16, uint32_t cap_knots) {
    DeviceGuard dev_guard(c);
    if (cap_knots < c->n_knots) return set_err(c, "gyro_table: buffer too small");
    if (!c->n_knots) return 0;
    RS_HIP(hipStreamSynchronize(c->stream));
    RS_HIP(hipMemcpy(out16, c->coef64.p, (size_t)c->n_knots * 128, hipMemcpyDeviceToHost));
    return 0;
}

int rship_upload_raw(rship_ctx* c, const double* host, uint64_t arena_offset, uint64_t n_doubles) {
    DeviceGuard dev_guard(c);
    if (!n_doubles) return 0;
    const size_t need = (size_t)(arena_offset + n_doubles) * 8;
    if (need > c->raw.cap) {
        // grow geometrically, keeping what is there (earlier uploads may still be in flight)
        RS_HIP(hipStreamSynchronize(c->copy_stream));
        DevBuf bigger;
        size_t want = c->raw.cap * 2 > need ? c->raw.cap * 2 : need;
        want += want / 8 + 4096;
        RS_HIP(hipMalloc(&bigger.p, want));
        bigger.cap = want;
        if (c->raw.p) {
            hipError_t e = hipMemcpy(bigger.p, c->raw.p, c->raw.cap, hipMemcpyDeviceToDevice);
            if (e != hipSuccess) { (void)hipFree(bigger.p); return set_err(c, "upload_raw: grow", e); }
            // a packing kernel reading the old buffer may still be queued on the compute stream
            RS_HIP(hipStreamSynchronize(c->stream));
            RS_HIP(hipFree(c->raw.p));
        }
        c->raw = bigger;
    }
    RS_HIP(hipMemcpyAsync((double*)c->raw.p + arena_offset, host, (size_t)n_doubles * 8, hipMemcpyHostToDevice, c->copy_stream));
    return 0;
}

int rship_pack_frames(rship_ctx* c, const rship_frame* table, const rship_pack_frame* pack, uint32_t n_frames,
                      uint64_t total_rays, double start, double fs, uint32_t* bad) {
    DeviceGuard dev_guard(c);
    if (bad) *bad = 0;
    c->n_frames = 0;
    c->n_sel = 0;
    c->h_sel.clear();
    c->h_frame_n.assign(n_frames, 0);
    c->max_span = 0.f;
    c->max_ends = 0.f;
    c->own_dims.assign(n_frames, rs::FrameDims{0u, 0.f, 0.f});
    c->problem_dims.clear(); // (the host gives them again after packing: rship_set_problem_frames)
    memset(c->cls_off, 0, sizeof(c->cls_off));
    memset(c->cls_max_n, 0, sizeof(c->cls_max_n));
    c->cls_used = 0;
    c->h_cls_slots.clear();
    uint32_t max_n = 0;
    for (uint32_t i = 0; i < n_frames; ++i) {
        if ((uint64_t)table[i].ray_offset + table[i].n_rays > total_rays) return set_err(c, "frame table exceeds ray buffer");
        if (table[i].n_rays != pack[i].n_rays || table[i].ray_offset != pack[i].ray_offset)
            return set_err(c, "pack list does not match the frame table");
        const uint64_t rec = (uint64_t)pack[i].n_rays * (pack[i].is_pixels ? 4 : 8);
        if ((pack[i].raw_offset + rec) * 8 > c->raw.cap && rec) return set_err(c, "pack: record outside the uploaded raw data");
        if (table[i].n_rays > (uint32_t)rship_max_tracks())
            return set_err(c, "frame has more tracks than the kernels accept (" + std::to_string(rship_max_tracks()) + ")");
        c->h_frame_n[i] = table[i].n_rays;
        max_n = std::max(max_n, table[i].n_rays);
        const float span = rs::frame_span(table[i].tmin, table[i].tmax); // knots a frame touches at one delay
        if (table[i].n_rays && span > c->max_span) c->max_span = span;
        // (RSSYNC_FORCE_GENERAL_SPLINE, rounds 1-3: whole pairs only)
        const float ends = c->force_general ? span : rs::frame_ends(table[i].range_a, table[i].range_b, span);
        if (table[i].n_rays && ends > c->max_ends) c->max_ends = ends;
        c->own_dims[i] = rs::FrameDims{table[i].n_rays, span, ends};
    }
    update_class_caps(c);
    const size_t tr = (size_t)total_rays;
    if (ensure(c, c->rays_a, tr ? tr * 16 : 16) || ensure(c, c->rays_b, tr ? tr * 16 : 16) ||
        ensure(c, c->rays64, tr ? tr * 64 : 64))
        return 1;
    if (ensure(c, c->frames, (size_t)n_frames * sizeof(rship_frame) + 64)) return 1;
    c->total_rays = total_rays;
    if (n_frames && c->force_general) {
        // RSSYNC_FORCE_GENERAL_SPLINE (rounds 1-3, the sweep's "before" column): whole pairs only, 80-knot windows
        std::vector<rship_frame> legacy(table, table + n_frames);
        for (rship_frame& r : legacy) r.range_a = r.range_b = RSHIP_NO_SPLIT;
        c->max_ends = c->max_span;
        RS_HIP(hipMemcpy(c->frames.p, legacy.data(), (size_t)n_frames * sizeof(rship_frame), hipMemcpyHostToDevice));
    } else if (n_frames) {
        RS_HIP(hipMemcpyAsync(c->frames.p, table, (size_t)n_frames * sizeof(rship_frame), hipMemcpyHostToDevice, c->stream));
    }
    uint32_t nb = 0;
    if (n_frames && max_n) {
        TempBuf dpk, dbad;
        if (ensure(c, dpk, (size_t)n_frames * sizeof(rship_pack_frame)) || ensure(c, dbad, 16)) return 1;
        hipError_t e = hipMemcpyAsync(dpk.p, pack, (size_t)n_frames * sizeof(rship_pack_frame), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(dbad.p, 0, 16, c->stream);
        // every raw upload issued so far must have landed before the kernel reads the records
        if (e == hipSuccess) e = hipEventRecord(c->copy_done, c->copy_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->copy_done, 0);
        if (e == hipSuccess) {
            PackParams p{};
            p.raw = (const double*)c->raw.p;
            p.frames = (const rship_pack_frame*)dpk.p;
            p.rays_a = (f4*)c->rays_a.p;
            p.rays_b = (f4*)c->rays_b.p;
            p.q0 = (double2*)c->rays64.p;
            p.q1 = p.q0 + tr;
            p.q2 = p.q1 + tr;
            p.q3 = p.q2 + tr;
            p.start = start;
            p.fs = fs;
            p.bad = (uint32_t*)dbad.p;
            {
                ProfScope ps(c, RSHIP_K_PIXELS);
                hipLaunchKernelGGL(pack_frames_kernel, dim3(n_frames, (max_n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, p);
            }
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&nb, dbad.p, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        prof_collect(c);
        if (e != hipSuccess) return set_err(c, "pack_frames", e);
    } else {
        RS_HIP(hipStreamSynchronize(c->stream));
    }
    if (bad) *bad = nb;
    c->n_frames = n_frames;
    return 0;
}

// One object over several devices: the track count and the knot spans (tmin / tmax / range_a / range_b of the frame
// table) of ALL frames of the problem, so that this context plans its spline windows from the same frames as every other
// shard -- and as a single device holding everything would (window_plan.hpp: plan_window_frames).  Optional: without it
// the context's own table is what it plans from.
int rship_set_problem_frames(rship_ctx* c, const rship_frame* table_all, uint32_t n_all) {
    c->problem_dims.assign(n_all, rs::FrameDims{0u, 0.f, 0.f});
    for (uint32_t i = 0; i < n_all; ++i) {
        const float span = rs::frame_span(table_all[i].tmin, table_all[i].tmax);
        const float ends = c->force_general ? span : rs::frame_ends(table_all[i].range_a, table_all[i].range_b, span);
        c->problem_dims[i] = rs::FrameDims{table_all[i].n_rays, span, ends};
    }
    update_class_caps(c);
    return 0;
}

// Selection = list of slots.  A slot names a frame of the table; with groups (batched windows)
// the same frame may appear in several slots, slots of one group are contiguous, and the
// per-frame Sync state (M, k) is kept per slot.
int rship_select_slots(rship_ctx* c, const uint32_t* idx, uint32_t n, const uint32_t* grp_off, uint32_t n_grp) {
    DeviceGuard dev_guard(c);
    for (uint32_t i = 0; i < n; ++i)
        if (idx[i] >= c->n_frames) return set_err(c, "select: frame index out of range");
    if (n_grp < 1) n_grp = 1;
    if (grp_off && (grp_off[0] != 0 || grp_off[n_grp] != n)) return set_err(c, "select: bad group offsets");
    if (ensure(c, c->sel, (size_t)n * 4 + 4) || ensure(c, c->grp, (size_t)n * 4 + 4) ||
        ensure(c, c->grp_off, (size_t)(n_grp + 1) * 4))
        return 1;
    if (ensure(c, c->M, (size_t)n * 24 + 24) || ensure(c, c->k, (size_t)n * 8 + 8) || ensure(c, c->init_h, (size_t)n * 4 + 4)) return 1;
    if (ensure(c, c->mo_evals, (size_t)n * 4 + 4) || ensure(c, c->mo_order, (size_t)n * 4 + 4)) return 1;
    c->init_pending = false;
    std::vector<uint32_t> g(n, 0), off(n_grp + 1, 0);
    if (grp_off) {
        off.assign(grp_off, grp_off + n_grp + 1);
        for (uint32_t w = 0; w < n_grp; ++w)
            for (uint32_t j = off[w]; j < off[w + 1]; ++j) g[j] = w;
    } else {
        off[n_grp] = n; // single group (n_grp == 1)
    }
    if (n) {
        RS_HIP(hipMemcpyAsync(c->sel.p, idx, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        RS_HIP(hipMemcpyAsync(c->grp.p, g.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    }
    RS_HIP(hipMemcpyAsync(c->grp_off.p, off.data(), (size_t)(n_grp + 1) * 4, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemsetAsync(c->M.p, 0, (size_t)n * 24 + 24, c->stream));
    RS_HIP(hipMemsetAsync(c->k.p, 0, (size_t)n * 8 + 8, c->stream));
    RS_HIP(hipMemsetD32Async((hipDeviceptr_t)c->init_h.p, kInitNone, (size_t)n + 1, c->stream));
    // the slot list of every size class (ascending slots inside a class), and -- until a launch has left evaluation
    // counts -- the motion kernel's launch order = that list
    {
        std::vector<uint32_t> cnt(kNumClasses + 1, 0);
        uint32_t mx[kNumClasses] = {};
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t fn = c->h_frame_n[idx[i]];
            const int k = class_of(c, fn);
            cnt[k + 1] += 1;
            mx[k] = std::max(mx[k], fn);
        }
        c->cls_used = 0;
        for (int k = 0; k < kNumClasses; ++k) {
            if (cnt[k + 1]) c->cls_used += 1;
            c->cls_max_n[k] = mx[k];
            c->cls_off[k] = cnt[k];
            cnt[k + 1] += cnt[k];
        }
        c->cls_off[kNumClasses] = n;
        c->h_cls_slots.assign((size_t)n + 1, n);
        uint32_t at[kNumClasses];
        for (int k = 0; k < kNumClasses; ++k) at[k] = c->cls_off[k];
        for (uint32_t i = 0; i < n; ++i) c->h_cls_slots[at[class_of(c, c->h_frame_n[idx[i]])]++] = i;
        if (ensure(c, c->cls_slots, (size_t)n * 4 + 4)) return 1;
        RS_HIP(hipMemcpyAsync(c->cls_slots.p, c->h_cls_slots.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, c->stream));
        RS_HIP(hipMemcpyAsync(c->mo_order.p, c->h_cls_slots.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, c->stream));
        RS_HIP(hipMemsetAsync(c->mo_evals.p, 0, (size_t)n * 4 + 4, c->stream));
        c->mo_identity = true;
        c->mo_ranges.clear();
    }
    RS_HIP(hipStreamSynchronize(c->stream));
    c->h_sel.assign(idx, idx + n);
    c->h_grp_off = off;
    c->n_sel = n;
    c->n_grp = n_grp;
    c->max_n = 0;
    for (int k = 0; k < kNumClasses; ++k) c->max_n = std::max(c->max_n, c->cls_max_n[k]);
    return 0;
}

int rship_select_frames(rship_ctx* c, const uint32_t* idx, uint32_t n) { return rship_select_slots(c, idx, n, nullptr, 1); }

// The plan of the sums that follow: window w = chunks win_chunk_off[w] .. [w+1]; chunk c = plan positions
// chunk_off[c] .. [c+1]; position j = slot plan_idx[j] (NULL: j itself).  Chunks hold the slots of a window whose
// frame-table index falls into the same block of 64 (the host builds them that way).
int rship_set_plan(rship_ctx* c, const uint32_t* plan_idx, uint32_t plan_len, const uint32_t* chunk_off, uint32_t n_chunks,
                   const uint32_t* win_chunk_off, uint32_t n_win) {
    DeviceGuard dev_guard(c);
    if (n_win < 1) return set_err(c, "plan: no window");
    if (win_chunk_off[0] != 0 || win_chunk_off[n_win] != n_chunks) return set_err(c, "plan: bad window offsets");
    if (n_chunks && (chunk_off[0] != 0 || chunk_off[n_chunks] != plan_len)) return set_err(c, "plan: bad chunk offsets");
    for (uint32_t j = 0; plan_idx && j < plan_len; ++j)
        if (plan_idx[j] >= c->n_sel) return set_err(c, "plan: slot out of range");
    if (!plan_idx && plan_len > c->n_sel) return set_err(c, "plan: slot out of range");
    if (ensure(c, c->plan_idx, (size_t)plan_len * 4 + 4) || ensure(c, c->plan_chunk_off, (size_t)(n_chunks + 1) * 4) ||
        ensure(c, c->plan_win_off, (size_t)(n_win + 1) * 4))
        return 1;
    if (plan_idx && plan_len) RS_HIP(hipMemcpyAsync(c->plan_idx.p, plan_idx, (size_t)plan_len * 4, hipMemcpyHostToDevice, c->stream));
    const uint32_t zero = 0;
    RS_HIP(hipMemcpyAsync(c->plan_chunk_off.p, n_chunks ? chunk_off : &zero, (size_t)(n_chunks + 1) * 4, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemcpyAsync(c->plan_win_off.p, win_chunk_off, (size_t)(n_win + 1) * 4, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipStreamSynchronize(c->stream));
    c->plan_has_idx = plan_idx != nullptr;
    c->plan_max_chunks = 0;
    for (uint32_t w = 0; w < n_win; ++w) c->plan_max_chunks = std::max(c->plan_max_chunks, win_chunk_off[w + 1] - win_chunk_off[w]);
    c->plan_chunks = n_chunks;
    c->plan_wins = n_win;
    c->plan_len = plan_len;
    return 0;
}

// pre_sync's per-frame body (core_private.cpp:75-85) for every (selected slot, candidate delay), then the
// sums of the current plan.  Asynchronous: rship_presync_collect waits and hands the results over.
int rship_presync_enqueue(rship_ctx* c, const int32_t* kd, const float* fd, const int32_t* kd64, const double* fd64, uint32_t n_cand, uint32_t n_hyp,
                          uint32_t stream_base, uint64_t seed, int want_frame_costs, int want_best_h) {
    DeviceGuard dev_guard(c);
    c->pend = rship_ctx::Pend{};
    c->pend_redo = rship_ctx::PendRedo{};
    if (!c->n_knots) return set_err(c, "no gyro spline uploaded");
    if (c->n_grp != 1) return set_err(c, "presync: the selection must not be grouped");
    const uint32_t ns = c->n_sel;
    // a batch of sweeps (rship_presync_batch_begin): this one is number `bslot` of `bn`; its sums and flags go to that slot
    const uint32_t bn = c->batch_n ? c->batch_n : 1u, bslot = c->batch_n ? c->batch_next : 0u;
    if (c->batch_n) {
        if (bslot >= bn) return set_err(c, "presync: more sweeps than the batch was opened for");
        if (n_cand != c->batch_rows) return set_err(c, "presync: every sweep of a batch has the same candidates");
        if (want_frame_costs || want_best_h) return set_err(c, "presync: no per-frame outputs inside a batch");
        c->batch_next += 1;
    }
    if (!n_cand || !ns) return 0; // nothing on this device: collect reports zeros
    if (ensure(c, c->frame_cost, (size_t)n_cand * ns * 8)) return 1;
    if (want_best_h && ensure(c, c->best_h, (size_t)n_cand * ns * 4)) return 1;
    if (ensure(c, c->flags, 16 + 4 * (size_t)bn)) return 1;
    if (bslot == 0) { // (a batch shares its candidates: uploaded once)
        if (upload_delays(c, kd, fd, n_cand)) return 1;
        RS_HIP(hipMemsetAsync(c->flags.p, 0, 16 + 4 * (size_t)bn, c->stream));
        if (c->batch_n) { // the batch's sums side by side
            if (!c->plan_wins) return set_err(c, "no reduction plan set");
            if (ensure(c, c->chunk_out, (size_t)bn * n_cand * (c->plan_chunks + 1) * 8) || ensure(c, c->win_out, (size_t)bn * n_cand * c->plan_wins * 8)) return 1;
        }
    }

    LmedsParams p{};
    p.rays_a = (const f4*)c->rays_a.p;
    p.rays_b = (const f4*)c->rays_b.p;
    p.frames = (const FrameRec*)c->frames.p;
    p.sel = (const uint32_t*)c->sel.p;
    p.n_sel = ns;
    p.coef = (const f4*)c->coef.p;
    p.n_knots = (int)c->n_knots;
    p.kd = (const int32_t*)c->kd.p;
    p.fd = c->d_fd;
    p.n_cand = n_cand;
    p.grp = nullptr;
    p.n_grp = 1;
    // enough workgroups to fill 256 CUs several times over, chunks long enough to
    // amortise the spline window (LDS) and to carry the provisional bound along
    uint64_t work = (uint64_t)n_cand * ns;
    uint32_t chunk = (uint32_t)(work / 8192);
    if (chunk < 1) chunk = 1;
    if (chunk > (uint32_t)kMaxChunk) chunk = kMaxChunk;
    if (chunk > n_cand) chunk = n_cand;
    // keep the chunk's delays inside what the LDS spline window can hold next to the widest frame (higher gyro
    // rates: a frame pair spans more knots); the window itself grows into dynamic LDS where it has to (plan_lmeds_window,
    // per size class)
    double step_knots = 0.0;
    if (n_cand > 1) step_knots = std::fabs(((double)kd[n_cand - 1] + fd[n_cand - 1] - (double)kd[0] - fd[0]) / (double)(n_cand - 1));
    p.n_hyp = n_hyp;
    p.stream_base = stream_base;
    p.seed = seed;
    p.frame_cost = (double*)c->frame_cost.p;
    p.best_h = want_best_h ? (int32_t*)c->best_h.p : nullptr;
    p.flags = (uint32_t*)c->flags.p + bslot;
    // the near-static watch (kernels/lmeds.hpp, "fp64 rows"): a bit per (slot, candidate) for the pairs whose rows are too
    // small for fp32 inputs; rship_presync_collect launches the fp64 form for them.  The candidates' delays split in fp64
    // (kd64 / fd64; without them the fp32 split, widened) are kept on the host until then -- except where the selection
    // holds frames of more than 8192 tracks, whose kernel takes the fp64 rows in place.
    if (!c->no_fp64_rows && c->rays64.p && c->coef64.p) {
        const uint32_t words = (n_cand + 31u) / 32u;
        const size_t mask_bytes = (size_t)ns * words * 4;
        const void* before = c->redo_mask.p;
        if (ensure(c, c->redo_mask, mask_bytes) || ensure_redo_count(c)) return 1;
        if (c->redo_mask.p != before || c->redo_dirty) {
            RS_HIP(hipMemsetAsync(c->redo_mask.p, 0, c->redo_mask.cap, c->stream));
            c->redo_dirty = false;
        }
        c->h_kd64r.assign(n_cand, 0);
        c->h_fd64r.assign(n_cand, 0.0);
        for (uint32_t i = 0; i < n_cand; ++i) {
            c->h_kd64r[i] = kd64 ? kd64[i] : kd[i];
            c->h_fd64r[i] = fd64 ? fd64[i] : (double)fd[i];
        }
        p.redo_mask = (uint32_t*)c->redo_mask.p;
        p.mask_words = words;
        p.src64 = Rows64Src{rays64_of(c), (const d4*)c->coef64.p, (int)c->n_knots};
        p.redo_count = (unsigned long long*)c->redo_count.p;
        c->pend_redo.uploaded = false;
        if (c->cls_off[5] != c->cls_off[6]) { // (frames of more than 8192 tracks: their kernel needs the fp64 delays now)
            const size_t pad = (n_cand + 1) / 2 * 2;
            if (ensure(c, c->redo_delays, pad * 4 + (size_t)n_cand * 8)) return 1;
            RS_HIP(hipMemcpyAsync(c->redo_delays.p, c->h_kd64r.data(), (size_t)n_cand * 4, hipMemcpyHostToDevice, c->stream));
            RS_HIP(hipMemcpyAsync((char*)c->redo_delays.p + pad * 4, c->h_fd64r.data(), (size_t)n_cand * 8, hipMemcpyHostToDevice, c->stream));
            c->pend_redo.uploaded = true;
            p.kd64 = (const int32_t*)c->redo_delays.p;
            p.fd64 = (const double*)((const char*)c->redo_delays.p + pad * 4);
        }
        c->pend_redo.armed = true;
        c->pend_redo.step_knots = step_knots;
        c->pend_redo.chunk = chunk;
    }
    if (c->dump_on) { // (TEST-VARIANTS build: rship_debug_residuals) -- the fp32 sweep's own residuals; no fp64 form of any pair
        const uint64_t words = (uint64_t)n_cand * ns * n_hyp * c->dump_rows;
        if (words > ((uint64_t)3 << 30) / 4) return set_err(c, "debug_residuals: more than 3 GB of residuals: sweep fewer candidates per call");
        if (ensure(c, c->dump, (size_t)words * 4)) return 1;
        RS_HIP(hipMemsetAsync(c->dump.p, 0xff, (size_t)words * 4, c->stream));
        p.dump = (uint32_t*)c->dump.p;
        p.dump_rows = c->dump_rows;
        p.redo_mask = nullptr;
        p.src64.coef = nullptr;
        c->pend_redo.armed = false;
        c->dump_dims[0] = n_cand; c->dump_dims[1] = ns; c->dump_dims[2] = n_hyp; c->dump_dims[3] = c->dump_rows;
    }
    if (launch_lmeds<0>(c, p, step_knots, chunk, &c->last_lmeds_cap, &c->last_lmeds_chunk)) return 1;
    c->pend_redo.p = p;
    if (c->batch_n) {
        // (inside a batch nothing is waited for, so the fp64 form cannot follow: a sweep that flags near-static pairs is
        // reported -- RSHIP_NEAR_STATIC in its flags -- and the caller redoes THAT sweep on its own; the bitmap is cleared
        // before the next sweep that uses it)
        c->pend_redo.armed = false;
        c->redo_dirty = true;
        if (launch_plan_sum(c, p.frame_cost, n_cand, ns, bslot, bn)) return 1;
        c->pend.rows = n_cand;
        return 0;
    }
    if (launch_plan_sum(c, p.frame_cost, n_cand, ns)) return 1;
    size_t end = 0;
    if (queue_sums_to_host(c, n_cand, 0, &c->pend.off_chunk, &end)) return 1;
    c->pend.off_flags = end;
    RS_HIP(hipMemcpyAsync((char*)c->pinned + end, c->flags.p, 4, hipMemcpyDeviceToHost, c->stream));
    c->pend.rows = n_cand;
    c->pend.frame_costs = want_frame_costs != 0;
    c->pend.best_h = want_best_h != 0;
    return 0;
}

// A batch of n sweeps over the same candidates, selection and plan whose results are collected TOGETHER (the orientation
// sweep of BASELINE config 5, core_testcode.cpp:216-224: between two sweeps only the gyro table changes --
// rship_gyro_rates_integrate_enqueue): after _begin, the next n rship_presync_enqueue calls return at once and keep their
// sums in slot 0 .. n-1; rship_presync_batch_collect waits ONCE and hands everything over, win_costs[n][n_cand][n_win],
// chunk_costs[n][n_cand][n_chunks] (either may be NULL), flags[n].  A sweep whose flags carry RSHIP_NEAR_STATIC has NOT had
// its near-static pairs recomputed in fp64 (that needs the host between two launches): the caller repeats that sweep alone.
int rship_presync_batch_begin(rship_ctx* c, uint32_t n, uint32_t n_cand) {
    if (!n) { // (cancel: a caller that gives up between _begin and _collect leaves no batch open)
        c->batch_n = c->batch_next = 0;
        c->pend = rship_ctx::Pend{};
        return 0;
    }
    if (n > 4096) return set_err(c, "presync batch: 1 .. 4096 sweeps");
    c->batch_n = n;
    c->batch_next = 0;
    c->batch_rows = n_cand;
    c->pend = rship_ctx::Pend{};
    return 0;
}
int rship_presync_batch_collect(rship_ctx* c, uint32_t n_cand, double* win_costs, double* chunk_costs, uint32_t* flags) {
    DeviceGuard dev_guard(c);
    const uint32_t n = c->batch_n;
    if (!n) return set_err(c, "presync batch: none open");
    c->batch_n = 0;
    if (flags) memset(flags, 0, (size_t)n * 4);
    if (!c->pend.rows) { // nothing was launched (no candidates or no slots on this device)
        if (win_costs) memset(win_costs, 0, (size_t)n * n_cand * (c->plan_wins ? c->plan_wins : 1) * 8);
        if (chunk_costs) memset(chunk_costs, 0, (size_t)n * n_cand * c->plan_chunks * 8);
        return 0;
    }
    if (c->batch_next != n || n_cand != c->batch_rows) return set_err(c, "presync batch: fewer sweeps were enqueued than the batch was opened for");
    // the slots hold [rows][chunks + 1] / [rows][wins]; one copy each, then the host drops the padding column
    const size_t cs = (size_t)n_cand * (c->plan_chunks + 1), ws = (size_t)n_cand * c->plan_wins;
    const size_t b_win = n * ws * 8, b_chunk = n * cs * 8, b_flags = (size_t)n * 4;
    if (ensure_pinned(c, b_win + b_chunk + b_flags + 64)) return 1;
    char* h = (char*)c->pinned;
    RS_HIP(hipMemcpyAsync(h, c->win_out.p, b_win, hipMemcpyDeviceToHost, c->stream));
    RS_HIP(hipMemcpyAsync(h + b_win, c->chunk_out.p, b_chunk, hipMemcpyDeviceToHost, c->stream));
    RS_HIP(hipMemcpyAsync(h + b_win + b_chunk, c->flags.p, b_flags, hipMemcpyDeviceToHost, c->stream));
    if (sync_stream(c)) return 1;
    if (win_costs) memcpy(win_costs, h, b_win);
    if (chunk_costs) {
        const double* src = (const double*)(h + b_win);
        // (plan_sum_kernel writes chunk sums [rows][plan_chunks] densely at the head of each slot)
        for (uint32_t b = 0; b < n; ++b) memcpy(chunk_costs + (size_t)b * n_cand * c->plan_chunks, src + (size_t)b * cs, (size_t)n_cand * c->plan_chunks * 8);
    }
    if (flags) memcpy(flags, h + b_win + b_chunk, b_flags);
    c->pend.rows = 0;
    return 0;
}

// win_costs[n_cand][n_win], chunk_costs[n_cand][n_chunks] (either may be NULL); debug outputs
// frame_costs / best_h [n_cand][n_sel] if they were asked for at enqueue time
int rship_presync_collect(rship_ctx* c, uint32_t n_cand, double* win_costs, double* chunk_costs, uint32_t* flags,
                          double* frame_costs, int32_t* best_h) {
    DeviceGuard dev_guard(c);
    if (flags) *flags = 0;
    if (!c->pend.rows) { // nothing was launched (no candidates or no slots on this device)
        if (win_costs) memset(win_costs, 0, (size_t)n_cand * (c->plan_wins ? c->plan_wins : 1) * 8);
        if (chunk_costs) memset(chunk_costs, 0, (size_t)n_cand * c->plan_chunks * 8);
        return 0;
    }
    if (n_cand != c->pend.rows) return set_err(c, "presync_collect: candidate count differs from the enqueue");
    if (sync_stream(c)) return 1;
    const uint32_t ns = c->n_sel;
    {
        // NEAR-STATIC pairs (kernels/lmeds.hpp, "fp64 rows"): the sweep has flagged (frame, candidate) pairs whose rows are
        // too small for its fp32 inputs.  Their costs are recomputed from the fp64 streams by the R64 form of the same
        // kernels, the sums are taken again, and only then does the caller see anything.  An ordinary scene never gets here.
        uint32_t fl = 0;
        memcpy(&fl, (char*)c->pinned + c->pend.off_flags, 4);
        if ((fl & RSHIP_NEAR_STATIC) && c->pend_redo.armed) {
            c->redo_dirty = true; // (until the fp64 launches have been enqueued: an error below leaves bits behind)
            LmedsParams p = c->pend_redo.p;
            const size_t pad = ((size_t)n_cand + 1) / 2 * 2;
            if (!c->pend_redo.uploaded) {
                if (ensure(c, c->redo_delays, pad * 4 + (size_t)n_cand * 8)) return 1;
                RS_HIP(hipMemcpyAsync(c->redo_delays.p, c->h_kd64r.data(), (size_t)n_cand * 4, hipMemcpyHostToDevice, c->stream));
                RS_HIP(hipMemcpyAsync((char*)c->redo_delays.p + pad * 4, c->h_fd64r.data(), (size_t)n_cand * 8, hipMemcpyHostToDevice, c->stream));
            }
            p.kd64 = (const int32_t*)c->redo_delays.p;
            p.fd64 = (const double*)((const char*)c->redo_delays.p + pad * 4);
            if (launch_lmeds_redo(c, p, c->pend_redo.step_knots, c->pend_redo.chunk)) return 1;
            c->redo_dirty = false;
            c->near_launches += 1;
            if (launch_plan_sum(c, p.frame_cost, n_cand, ns)) return 1;
            size_t end = 0;
            if (queue_sums_to_host(c, n_cand, 0, &c->pend.off_chunk, &end)) return 1;
            c->pend.off_flags = end;
            RS_HIP(hipMemcpyAsync((char*)c->pinned + end, c->flags.p, 4, hipMemcpyDeviceToHost, c->stream));
            if (sync_stream(c)) return 1;
        }
        c->pend_redo.armed = false;
    }
    if (win_costs) memcpy(win_costs, c->pinned, (size_t)n_cand * c->plan_wins * 8);
    if (chunk_costs) memcpy(chunk_costs, (char*)c->pinned + c->pend.off_chunk, (size_t)n_cand * c->plan_chunks * 8);
    if (flags) memcpy(flags, (char*)c->pinned + c->pend.off_flags, 4);
    if (frame_costs) RS_HIP(hipMemcpy(frame_costs, c->frame_cost.p, (size_t)n_cand * ns * 8, hipMemcpyDeviceToHost));
    if (best_h && c->pend.best_h) RS_HIP(hipMemcpy(best_h, c->best_h.p, (size_t)n_cand * ns * 4, hipMemcpyDeviceToHost));
    c->pend.rows = 0;
    return 0;
}

namespace {
// (returns 1 only if the scratch of the large-frame path cannot be allocated: c->err is set, the launch then fails)
int fill_motion(rship_ctx* c, Motion64Params& p) {
    p.scratch = nullptr;
    p.scratch_rows = 0;
    p.scratch0 = 0;
    if (c->cls_off[5] != c->cls_off[6]) { // frames of more than 8192 tracks: their rows of P live in global memory, per entry of the class's list
        const uint32_t rows = big_rows(c->cls_max_n[5]);
        if (ensure(c, c->mo_scratch, (size_t)(c->cls_off[6] - c->cls_off[5]) * 3 * rows * 8)) return 1;
        p.scratch = (double*)c->mo_scratch.p;
        p.scratch_rows = rows;
    }
    p.rays = rays64_of(c);
    p.frames = (const FrameRec*)c->frames.p;
    p.sel = (const uint32_t*)c->sel.p;
    p.n_sel = c->n_sel;
    p.coef = (const d4*)c->coef64.p;
    p.n_knots = (int)c->n_knots;
    p.kd = (const int32_t*)c->kd64.p;
    p.fd = c->d_fd64;
    p.grp = c->n_grp > 1 ? (const uint32_t*)c->grp.p : nullptr;
    p.M = (double*)c->M.p;
    p.k = (double*)c->k.p;
    p.reeval = c->lbfgs_reeval;
    p.max_iters = 200; // core_private.cpp:265
    p.init_h = (int32_t*)c->init_h.p; // per slot: kInitNone unless GuessMotion's search has left a winner to finish
    p.seed = c->init_seed;
    p.stream_base = c->init_stream;
    p.stream_stride = c->init_stride;
    p.simple_k = 0;
    p.evals_out = (uint32_t*)c->mo_evals.p;
    p.order = (const uint32_t*)c->mo_order.p; // the caller has called prepare_order for its slot ranges
    return 0;
}

// Before a call launches the motion kernel over the slot ranges `ranges` (one per stream group): the launch order
// left by earlier launches is only usable if it was built for the same ranges (a class's share of a range is a segment
// of the class-sorted list, permuted in place); otherwise back to the list itself.
int prepare_order(rship_ctx* c, const std::vector<std::pair<uint32_t, uint32_t>>& ranges) {
    if (!c->mo_identity && ranges != c->mo_ranges) {
        RS_HIP(hipStreamSynchronize(c->stream));
        RS_HIP(hipMemcpy(c->mo_order.p, c->h_cls_slots.data(), (size_t)(c->n_sel + 1) * 4, hipMemcpyHostToDevice));
    }
    c->mo_identity = false;
    c->mo_ranges = ranges;
    return 0;
}
int prepare_order_all(rship_ctx* c) { return prepare_order(c, {{0u, c->n_sel}}); }
} // namespace

// FrameState::GuessMotion's hypothesis search (core_private.cpp:125-128 -> :34-59, 200 hypotheses) in the
// fp32 tile kernel at one delay per group (kd/fd); the winners stay on the device and the next
// rship_opt_motion / rship_finish_init turns them into M and k in fp64.  Asynchronous.
int rship_init_motion(rship_ctx* c, const int32_t* kd, const float* fd, const int32_t* kd64, const double* fd64, uint32_t n_hyp, uint32_t stream,
                      uint32_t stream_stride, uint64_t seed) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (upload_delays(c, kd, fd, c->n_grp) || ensure(c, c->flags, 16)) return 1;
    RS_HIP(hipMemsetAsync(c->flags.p, 0, 16, c->stream));
    LmedsParams p{};
    p.rays_a = (const f4*)c->rays_a.p;
    p.rays_b = (const f4*)c->rays_b.p;
    p.frames = (const FrameRec*)c->frames.p;
    p.sel = (const uint32_t*)c->sel.p;
    p.n_sel = c->n_sel;
    p.coef = (const f4*)c->coef.p;
    p.n_knots = (int)c->n_knots;
    p.kd = (const int32_t*)c->kd.p;
    p.fd = c->d_fd;
    p.n_cand = 1;
    p.chunk = 1;
    p.n_chunks = 1;
    p.n_hyp = n_hyp;
    p.stream_base = stream; // window w uses stream + w * stride
    p.stream_stride = stream_stride;
    p.seed = seed;
    p.grp = c->n_grp > 1 ? (const uint32_t*)c->grp.p : nullptr;
    p.n_grp = c->n_grp;
    p.best_h = (int32_t*)c->init_h.p;
    p.flags = (uint32_t*)c->flags.p;
    // near-static frames: the search takes its rows from the fp64 streams IN PLACE (kernels/lmeds.hpp, MODE 1): the fp64
    // split of the windows' delays (kd64 / fd64 [n_grp]; without them the fp32 split, widened)
    if (!c->no_fp64_rows && c->rays64.p && c->coef64.p) {
        const uint32_t ng = c->n_grp;
        const size_t pad = ((size_t)ng + 1) / 2 * 2;
        if (ensure(c, c->init_delays64, pad * 4 + (size_t)ng * 8) || ensure_redo_count(c)) return 1;
        // one staged copy (as upload_delays64: kd[ng] as int32, padded to 8 bytes, then fd[ng])
        c->h_init64.assign(pad + 2 * (size_t)ng, 0u);
        for (uint32_t i = 0; i < ng; ++i) {
            const int32_t k = kd64 ? kd64[i] : kd[i];
            const double f = fd64 ? fd64[i] : (double)fd[i];
            memcpy(&c->h_init64[i], &k, 4);
            memcpy(&c->h_init64[pad + 2 * (size_t)i], &f, 8);
        }
        RS_HIP(hipMemcpyAsync(c->init_delays64.p, c->h_init64.data(), pad * 4 + (size_t)ng * 8, hipMemcpyHostToDevice, c->stream));
        p.src64 = Rows64Src{rays64_of(c), (const d4*)c->coef64.p, (int)c->n_knots};
        p.kd64 = (const int32_t*)c->init_delays64.p;
        p.fd64 = (const double*)((const char*)c->init_delays64.p + pad * 4);
        p.redo_count = (unsigned long long*)c->redo_count.p;
    }
    if (launch_lmeds<1>(c, p, 0.0, 1u, &c->last_init_cap, nullptr)) return 1;
    c->init_pending = true;
    c->init_seed = seed;
    c->init_stream = stream;
    c->init_stride = stream_stride;
    return 0;
}

// GuessMotion's winners -> M, and GuessK (core_private.cpp:130-133), in fp64, without optimising
int rship_finish_init(rship_ctx* c, const int32_t* kd, const double* fd) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (!c->init_pending) return 0;
    if (upload_delays64(c, kd, fd, c->n_grp)) return 1;
    Motion64Params p{};
    if (fill_motion(c, p)) return 1;
    p.max_iters = 0;
    if (prepare_order_all(c) || launch_motion64(c, p)) return 1;
    c->init_pending = false;
    return sync_stream(c);
}

// the no-translation variant's hyper-parameter: k = clamp(100 / sqrt(sum_j |P_j|^2), 10, 1000) per slot
// (GuessK, core_private.cpp:130-133, with |h_j| in place of v.h_j: thesis section 2.11 eq. (12))
int rship_init_k_simple(rship_ctx* c, const int32_t* kd, const double* fd) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (upload_delays64(c, kd, fd, c->n_grp)) return 1;
    Motion64Params p{};
    if (fill_motion(c, p)) return 1;
    p.init_h = nullptr;
    p.simple_k = 1;
    if (prepare_order_all(c) || launch_motion64(c, p)) return 1;
    return sync_stream(c);
}

int rship_opt_motion_detail(rship_ctx* c, const int32_t* kd, const double* fd, uint32_t* per_frame, uint32_t cap) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (cap < c->n_sel) return set_err(c, "opt_motion_detail: output too small");
    TempBuf d;
    if (ensure(c, d, (size_t)c->n_sel * 8 + 8)) return 1;
    RS_HIP(hipMemsetAsync(d.p, 0, (size_t)c->n_sel * 8 + 8, c->stream));
    if (upload_delays64(c, kd, fd, c->n_grp)) return 1;
    Motion64Params p{};
    if (fill_motion(c, p)) return 1;
    p.per_frame = (uint32_t*)d.p;
    int rc = prepare_order_all(c) || launch_motion64(c, p);
    c->init_pending = false;
    hipError_t e = hipStreamSynchronize(c->stream);
    prof_collect(c);
    if (!rc && e == hipSuccess) e = hipMemcpy(per_frame, d.p, (size_t)c->n_sel * 8, hipMemcpyDeviceToHost);
    if (rc) return 1;
    if (e != hipSuccess) return set_err(c, "opt_motion_detail", e);
    return 0;
}

int rship_opt_motion(rship_ctx* c, const int32_t* kd, const double* fd, uint64_t* stats) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (stats) {
        if (ensure(c, c->stats, 32)) return 1;
        RS_HIP(hipMemsetAsync(c->stats.p, 0, 32, c->stream));
    }
    if (upload_delays64(c, kd, fd, c->n_grp)) return 1;
    Motion64Params p{};
    if (fill_motion(c, p)) return 1;
    p.stats = stats ? (unsigned long long*)c->stats.p : nullptr;
    if (prepare_order_all(c) || launch_motion64(c, p)) return 1;
    c->init_pending = false;
    if (stats) {
        if (ensure_pinned(c, 32)) return 1;
        RS_HIP(hipMemcpyAsync(c->pinned, c->stats.p, 24, hipMemcpyDeviceToHost, c->stream));
        if (sync_stream(c)) return 1;
        memcpy(stats, c->pinned, 24);
        return 0;
    }
    return 0; // stays queued: the next loss call is ordered behind it on the stream
}

// kd/fd: [n_delays][n_grp] (one delay per group of the selection, fd = NaN skips a group).  The per-slot
// losses are summed under the current plan (for Sync: plan windows = groups).  Asynchronous.
int rship_loss_enqueue(rship_ctx* c, const int32_t* kd, const double* fd, uint32_t n_delays, int want_grad, uint32_t flags) {
    DeviceGuard dev_guard(c);
    c->pend = rship_ctx::Pend{};
    if (!c->n_knots) return set_err(c, "no gyro spline uploaded");
    const uint32_t ns = c->n_sel, ng = c->n_grp;
    if (!n_delays || !ns) return 0;
    const bool simple = (flags & RSHIP_LOSS_SIMPLIFIED) != 0;
    if (c->init_pending && !simple) return set_err(c, "loss: the motion initialisation has not been finished");
    if (upload_delays64(c, kd, fd, (size_t)n_delays * ng)) return 1;
    if (ensure(c, c->part, (size_t)n_delays * ns * 16)) return 1;
    Loss64Params p{};
    p.rays = rays64_of(c);
    p.frames = (const FrameRec*)c->frames.p;
    p.sel = (const uint32_t*)c->sel.p;
    p.n_sel = ns;
    p.coef = (const d4*)c->coef64.p;
    p.n_knots = (int)c->n_knots;
    p.fs = c->fs;
    p.kd = (const int32_t*)c->kd64.p;
    p.fd = c->d_fd64;
    p.n_delays = n_delays;
    p.grp = ng > 1 ? (const uint32_t*)c->grp.p : nullptr;
    p.n_grp = ng;
    p.M = (const double*)c->M.p;
    p.k = (const double*)c->k.p;
    p.part_loss = (double*)c->part.p;
    p.part_grad = p.part_loss + (size_t)n_delays * ns;
    int rc;
    if (simple) rc = want_grad ? launch_loss64<true, true>(c, p) : launch_loss64<false, true>(c, p);
    else rc = want_grad ? launch_loss64<true, false>(c, p) : launch_loss64<false, false>(c, p);
    if (rc) return 1;
    // rows [0, n_delays) = loss, [n_delays, 2 n_delays) = d loss / d delay
    const uint32_t rows = want_grad ? 2 * n_delays : n_delays;
    if (launch_plan_sum(c, p.part_loss, rows, ns)) return 1;
    size_t end = 0;
    if (queue_sums_to_host(c, rows, 0, &c->pend.off_chunk, &end)) return 1;
    c->pend.rows = rows;
    c->pend.grad = want_grad != 0;
    return 0;
}

// win_loss / win_grad [n_delays][n_win], chunk_loss / chunk_grad [n_delays][n_chunks]; any may be NULL
int rship_loss_collect(rship_ctx* c, uint32_t n_delays, double* win_loss, double* win_grad, double* chunk_loss,
                       double* chunk_grad) {
    DeviceGuard dev_guard(c);
    const size_t wn = (size_t)n_delays * (c->plan_wins ? c->plan_wins : 1), cn = (size_t)n_delays * c->plan_chunks;
    if (!c->pend.rows) {
        if (win_loss) memset(win_loss, 0, wn * 8);
        if (win_grad) memset(win_grad, 0, wn * 8);
        if (chunk_loss) memset(chunk_loss, 0, cn * 8);
        if (chunk_grad) memset(chunk_grad, 0, cn * 8);
        return 0;
    }
    if (n_delays * (c->pend.grad ? 2u : 1u) != c->pend.rows) return set_err(c, "loss_collect: row count differs from the enqueue");
    if (sync_stream(c)) return 1;
    const char* base = (const char*)c->pinned;
    if (win_loss) memcpy(win_loss, base, wn * 8);
    if (win_grad && c->pend.grad) memcpy(win_grad, base + wn * 8, wn * 8);
    if (chunk_loss) memcpy(chunk_loss, base + c->pend.off_chunk, cn * 8);
    if (chunk_grad && c->pend.grad) memcpy(chunk_grad, base + c->pend.off_chunk + cn * 8, cn * 8);
    c->pend.rows = 0;
    return 0;
}

// Sync's whole outer loop (core_private.cpp:298-331) for the W windows (= groups) of the selection, driven from
// the device: see kernels/syncloop.hpp.  d0[W] initial delays (after rship_init_motion, or rship_init_k_simple
// for the simplified mode); on return d_out[W], iters[W] and trace[W][max_outer][6] (rows as
// rssync_ext_sync_trace).  The plan must be one window per group over the slots in order.
// How many groups of windows run their loops side by side (each on its own stream).  An iteration is a chain of
// five short launches, each as long as its slowest frame; with all windows in one chain the device idles most
// of that time.  Windows do not see each other, so contiguous groups of them can run as independent chains.
static uint32_t loop_groups(const rship_ctx* c, uint32_t n_win) {
    uint32_t g = 4; // HIP's default number of hardware queues
    if (const char* s = std::getenv("RSSYNC_LOOP_STREAMS")) { const int v = atoi(s); if (v >= 1 && v <= 16) g = (uint32_t)v; }
    if (c->prof) g = 1; // the event pairs of the profile live on the context's stream
    const uint32_t by_size = n_win / 8; // a chain per fewer than ~8 windows does not pay for its launches
    if (g > by_size) g = by_size;
    return g < 1 ? 1 : g;
}

static int sync_run_body(rship_ctx* c, const double* d0, int max_outer, double search_center, double search_radius,
                         int simplified, double* d_out, int32_t* iters, double* trace);
int rship_sync_run(rship_ctx* c, const double* d0, int max_outer, double search_center, double search_radius,
                   int simplified, double* d_out, int32_t* iters, double* trace) {
    const int rc = sync_run_body(c, d0, max_outer, search_center, search_radius, simplified, d_out, iters, trace);
    // a rank-local failure (an allocation, a launch) in rank mode: the other ranks are, or will be, inside the next
    // all-reduce of this loop -- abort the communicator so that they fail too instead of waiting for this rank
    if (rc && c->rccl_comm) rccl_abort_comm(c);
    return rc;
}
static int sync_run_body(rship_ctx* c, const double* d0, int max_outer, double search_center, double search_radius,
                         int simplified, double* d_out, int32_t* iters, double* trace) {
    DeviceGuard dev_guard(c);
    // Rank mode: the frames are sharded over processes and this context holds the library's RCCL communicator.  The
    // loop is the same; the window sums are all-reduced on the stream between the kernels.  A rank may hold no frame of
    // the selection at all: it still takes part in every all-reduce.
    const bool ranked = c->rccl_comm != nullptr || c->loop_xchg != nullptr;
    if (!c->n_knots) return set_err(c, "no gyro spline uploaded");
    if (!ranked && check_ready(c)) return 1;
    const uint32_t W = c->n_grp, ns = c->n_sel;
    if (c->plan_wins != W || c->plan_has_idx || c->plan_len != ns) return set_err(c, "sync_run: the plan must be the selection's groups");
    if (max_outer <= 0) return set_err(c, "sync_run: no iterations");
    if (c->h_grp_off.size() != (size_t)W + 1) return set_err(c, "sync_run: no selection");
    const uint32_t G = ranked ? 1u : loop_groups(c, W); // (one communicator: its collectives on one stream, in one order)
    rccl_allreduce_fn allreduce = nullptr;
    if (ranked && !c->loop_xchg && !(allreduce = (rccl_allreduce_fn)rccl_sym(c, "ncclAllReduce"))) return 1;
    std::vector<double> xbuf; // host exchange: the sums of one launch
    c->loop_exchanges = 0;
    int nf_fixed = 0; // test knob: always evaluate exactly this many trials first
    if (const char* e = std::getenv("RSSYNC_LOOP_FIRST_TRIALS")) { const int v = atoi(e); if (v >= 1 && v <= kMaxBt) nf_fixed = v; }
    // how many trials a search's first launch holds at least (SyncLoopParams::nf_floor): five where a trial is cheap
    // (the reference's ~130-track frames), one for frames of 1024 tracks and more (dense trackers: every unneeded
    // trial of 4096 x 2048 ray pairs is 0.06 ms).  Only the batching depends on it, never a result.  In rank mode every
    // rank must batch its trials alike -- the sums of a trial come from all of them -- and no rank knows the others'
    // frames: always five there.  (Round 6 tried six, to spare a first search that waits its two extra exchanges: no step of the
    // benchmark's shapes has such a search -- the first search's own guess, syncloop.hpp step_decide, already asks for six --
    // 18 exchanges at 8 iterations and 22 at 10 with either floor, profiles/r6_multi_rank_rehearsal.json; not adopted.)
    int nf_floor = (!ranked && c->max_n >= 1024u) ? 1 : kHalfBt;
    if (const char* e = std::getenv("RSSYNC_LOOP_TRIALS_FLOOR")) { const int v = atoi(e); if (v >= 1 && v <= kMaxBt) nf_floor = v; }
    const int max_launch = 2 * max_outer; // a window whose line search needs its later trials waits one iteration for them
    // one allocation: windows | motion delays | loss delays | trial delays | per-group counters | trace | chunk scratch
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_win = take(W * sizeof(SyncWin));
    const size_t o_mokd = take(W * 4), o_mofd = take(W * 8), o_lgkd = take(W * 4), o_lgfd = take(W * 8);
    const size_t o_trkd = take((size_t)kMaxBt * W * 4), o_trfd = take((size_t)kMaxBt * W * 8);
    const size_t nact_stride = ((size_t)max_launch * 4 + 255) / 256 * 256;
    const size_t o_nact = take((size_t)G * nact_stride);
    const size_t o_trace = take((size_t)W * max_outer * 48);
    const size_t o_tmp = take((size_t)W * 2 * kMaxBt * (c->plan_max_chunks + 1) * 8);
    const size_t o_ext = take((size_t)2 * kMaxBt * W * 8);
    if (ensure(c, c->loop_state, off)) return 1;
    char* base = (char*)c->loop_state.p;
    if (ensure(c, c->part, (size_t)2 * kMaxBt * (ns + 1) * 8)) return 1;
    while (c->loop_streams.size() < G) {
        hipStream_t st = nullptr;
        RS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        c->loop_streams.push_back(st);
    }
    if (!c->loop_ready) RS_HIP(hipEventCreateWithFlags(&c->loop_ready, hipEventDisableTiming));

    std::vector<SyncWin> hw(W);
    for (uint32_t w = 0; w < W; ++w) {
        hw[w] = SyncWin{};
        hw[w].d = d0[w];
        hw[w].active = 1;
        hw[w].hit = -1;
        hw[w].nf = nf_fixed ? nf_fixed : nf_floor;
    }
    RS_HIP(hipMemcpyAsync(base + o_win, hw.data(), W * sizeof(SyncWin), hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemsetAsync(base + o_nact, 0, (size_t)G * nact_stride, c->stream));

    SyncLoopParams lp{};
    lp.win = (SyncWin*)(base + o_win);
    lp.n_win = W;
    lp.part = (const double*)c->part.p;
    lp.n_sel = ns;
    lp.chunk_off = (const uint32_t*)c->plan_chunk_off.p;
    lp.win_chunk_off = (const uint32_t*)c->plan_win_off.p;
    lp.chunk_tmp = (double*)(base + o_tmp);
    lp.chunk_stride = c->plan_max_chunks + 1;
    lp.mo_kd = (int32_t*)(base + o_mokd); lp.mo_fd = (double*)(base + o_mofd);
    lp.lg_kd = (int32_t*)(base + o_lgkd); lp.lg_fd = (double*)(base + o_lgfd);
    lp.tr_kd = (int32_t*)(base + o_trkd); lp.tr_fd = (double*)(base + o_trfd);
    lp.fs = c->fs;
    {
        double t = 1e-3; // t0, decay = 0.1 (core_private.cpp:226): the host loop's sequence, bit for bit
        for (int i = 0; i <= kMaxBt; ++i) { lp.ts[i] = t; t *= .1; }
    }
    lp.c_armijo = 2e-4;
    lp.delay_b = .3;
    lp.search_center = search_center;
    lp.search_radius = search_radius;
    lp.max_outer = max_outer;
    lp.nf_fixed = nf_fixed;
    lp.nf_floor = nf_floor;
    lp.trace = (double*)(base + o_trace);
    lp.ext_sums = ranked ? (double*)(base + o_ext) : nullptr;

    Motion64Params mp{};
    if (fill_motion(c, mp)) return 1;
    mp.kd = lp.mo_kd;
    mp.fd = lp.mo_fd;
    Loss64Params qp{};
    qp.rays = rays64_of(c);
    qp.frames = (const FrameRec*)c->frames.p;
    qp.sel = (const uint32_t*)c->sel.p;
    qp.n_sel = ns;
    qp.coef = (const d4*)c->coef64.p;
    qp.n_knots = (int)c->n_knots;
    qp.fs = c->fs;
    qp.grp = W > 1 ? (const uint32_t*)c->grp.p : nullptr;
    qp.n_grp = W;
    qp.M = (const double*)c->M.p;
    qp.k = (const double*)c->k.p;
    qp.part_loss = (double*)c->part.p;

    // the groups: windows [w0, w1) = slots [s0, s1), balanced by slots
    struct Group {
        uint32_t w0, w1, s0, s1;
        hipStream_t st;
        int* n_active;
        int* h_nact; // pinned
        int it = 0;
        bool done = false;
    };
    std::vector<Group> groups(G);
    if (ensure_pinned(c, (size_t)G * nact_stride + 64)) return 1;
    {
        uint32_t w = 0;
        for (uint32_t g = 0; g < G; ++g) {
            Group& gr = groups[g];
            gr.w0 = w;
            const uint64_t target = (uint64_t)ns * (g + 1) / G;
            while (w < W && (g + 1 == G || c->h_grp_off[w + 1] <= target || w == gr.w0)) ++w;
            if (g + 1 == G) w = W;
            gr.w1 = w;
            gr.s0 = c->h_grp_off[gr.w0];
            gr.s1 = c->h_grp_off[gr.w1];
            gr.st = G == 1 ? c->stream : c->loop_streams[g];
            gr.n_active = (int*)(base + o_nact + (size_t)g * nact_stride);
            gr.h_nact = (int*)((char*)c->pinned + (size_t)g * nact_stride);
            // (windows without frames still iterate: their sums are zero, the steps are zero and the convergence
            // counter stops them after six iterations, exactly like the host loop -- only a group without
            // windows has nothing to do)
            gr.done = gr.w1 == gr.w0;
        }
    }
    {
        std::vector<std::pair<uint32_t, uint32_t>> ranges; // (segments of the class-sorted list: one per group and class)
        for (const Group& gr : groups)
            if (gr.s1 > gr.s0)
                for (const ClassRange& r : class_ranges(c, gr.s0, gr.s1 - gr.s0)) ranges.emplace_back(r.pos0, r.count);
        if (prepare_order(c, ranges)) return 1;
    }
    if (G > 1) { // what the context's stream has queued (selection, GuessMotion, the copies above) comes first
        RS_HIP(hipEventRecord(c->loop_ready, c->stream));
        for (Group& gr : groups) RS_HIP(hipStreamWaitEvent(gr.st, c->loop_ready, 0));
    }
    auto enqueue_iteration = [&](Group& gr) -> int {
        SyncLoopParams l = lp;
        l.win0 = gr.w0; l.win1 = gr.w1;
        l.n_active = gr.n_active;
        l.it = gr.it;
        const uint32_t nw = gr.w1 - gr.w0, cnt = gr.s1 - gr.s0;
        const dim3 ctl_block(nw > 64 ? 64 : kBlock);
        Loss64Params q = qp;
        auto loss_launch = [&](bool grad) -> int {
            if (simplified) return grad ? launch_loss64<true, true>(c, q, gr.st, gr.s0, cnt) : launch_loss64<false, true>(c, q, gr.st, gr.s0, cnt);
            return grad ? launch_loss64<true, false>(c, q, gr.st, gr.s0, cnt) : launch_loss64<false, false>(c, q, gr.st, gr.s0, cnt);
        };
        if (gr.it == 0) {
            hipLaunchKernelGGL(sync_begin_kernel, dim3((nw + 63) / 64), dim3(64), 0, gr.st, l);
            RS_HIP(hipGetLastError());
        }
        if (!simplified && cnt) {
            if (launch_motion64(c, mp, gr.st, gr.s0, cnt)) return 1; // :311 (finishes a pending GuessMotion on its first launch)
        }
        // loss + gradient at x0 (:298-299 -> backtrack.cpp:4)
        q.kd = l.lg_kd; q.fd = l.lg_fd; q.n_delays = 1;
        q.part_grad = q.part_loss + (size_t)ns;
        if (cnt && loss_launch(true)) return 1;
        l.rows = 2;
        auto exchange = [&](uint32_t rows) -> int { // rank mode: this rank's window sums, then the sum over the ranks, on the stream
            if (!ranked) return 0;
            hipLaunchKernelGGL(sync_sums_kernel, dim3(nw), ctl_block, 0, gr.st, l);
            RS_HIP(hipGetLastError());
            c->loop_exchanges += 1;
            if (c->loop_xchg) { // through the host: drain, add the other ranks' sums, send back
                xbuf.resize((size_t)rows * W);
                RS_HIP(hipMemcpyAsync(xbuf.data(), l.ext_sums, xbuf.size() * 8, hipMemcpyDeviceToHost, gr.st));
                RS_HIP(hipStreamSynchronize(gr.st));
                const int rc = c->loop_xchg(c->loop_xchg_user, xbuf.data(), xbuf.size());
                if (rc) return set_err(c, "sync_run: the loop's host exchange failed (" + std::to_string(rc) + ")");
                RS_HIP(hipMemcpyAsync(l.ext_sums, xbuf.data(), xbuf.size() * 8, hipMemcpyHostToDevice, gr.st));
                RS_HIP(hipStreamSynchronize(gr.st)); // (xbuf is pageable and reused)
                return 0;
            }
            const int rc = allreduce(l.ext_sums, l.ext_sums, (size_t)rows * W, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->rccl_comm, gr.st);
            if (rc) return set_err(c, "rccl: ncclAllReduce failed (" + std::to_string(rc) + ")");
            return 0;
        };
        if (exchange(2)) return 1;
        {
            ProfScope ps(c, RSHIP_K_REDUCE);
            hipLaunchKernelGGL(sync_grad_kernel, dim3(nw), ctl_block, 0, gr.st, l);
        }
        // the trials each window asked for, one launch (kernels/syncloop.hpp: trial_wanted)
        q.kd = l.tr_kd; q.fd = l.tr_fd; q.n_delays = kMaxBt;
        q.part_grad = nullptr;
        l.rows = kMaxBt;
        if (cnt && loss_launch(false)) return 1;
        if (exchange(kMaxBt)) return 1;
        {
            ProfScope ps(c, RSHIP_K_REDUCE);
            hipLaunchKernelGGL(sync_step_kernel, dim3(nw), ctl_block, 0, gr.st, l);
        }
        RS_HIP(hipGetLastError());
        gr.it += 1;
        return 0;
    };

    const int kLook = 8; // iterations enqueued between two looks at a group's counter of active windows
    // one host thread per group: a chain is five launches per iteration and the launches of four chains from one
    // thread would make the host the slowest part
    auto run_group = [&](Group& gr) -> int {
        DeviceGuard guard(c); // the current device is a per-thread setting
        while (!gr.done) {
            // (after the first block the windows still active are the stragglers: shorter blocks waste fewer empty launches)
            // With ranks an over-enqueued iteration is not just five empty launches but TWO EXCHANGES every rank has to
            // take part in (round 5 at the benchmark's size: 8 outer iterations = 9 launches of the loop -- the first
            // search waits once for its later trials -- enqueued as 8 + 4 = 12: 26 exchanges per step against a floor of
            // 1 + 2 x 9 + 1 = 20).  Through the host hook every exchange drains the stream anyway, so the counter of
            // active windows is looked at after EVERY iteration there (one more small wait per iteration, no empty
            // iteration at all); with the RCCL communicator on the stream the blocks after the first are two iterations.
            const int later = !ranked ? kLook / 2 : (c->loop_xchg ? 1 : 2);
            const int first = (ranked && c->loop_xchg) ? 1 : kLook;
            const int until = std::min(max_launch, gr.it + (gr.it == 0 ? first : later));
            while (gr.it < until)
                if (enqueue_iteration(gr)) return 1;
            RS_HIP(hipMemcpyAsync(gr.h_nact, gr.n_active, (size_t)gr.it * 4, hipMemcpyDeviceToHost, gr.st));
            RS_HIP(hipStreamSynchronize(gr.st));
            gr.done = gr.h_nact[gr.it - 1] == 0 || gr.it >= max_launch;
        }
        return 0;
    };
    {
        std::vector<std::thread> workers;
        std::vector<int> rc(G, 0);
        for (uint32_t g = 1; g < G; ++g) workers.emplace_back([&, g] { rc[g] = run_group(groups[g]); });
        rc[0] = run_group(groups[0]);
        for (std::thread& t : workers) t.join();
        for (uint32_t g = 0; g < G; ++g)
            if (rc[g]) return 1;
    }
    int it_max = 0;
    for (const Group& gr : groups) it_max = std::max(it_max, gr.it);
    it_max = std::min(it_max, max_outer); // rows of the trace are indexed by a window's own iteration count
    c->init_pending = false;
    if (G == 1) prof_collect(c);
    // the windows and the rows of the iterations that ran ([iteration][window][6] on the device), through pinned memory
    const size_t tr_bytes = (size_t)it_max * W * 48, win_bytes = W * sizeof(SyncWin);
    if (ensure_pinned(c, tr_bytes + win_bytes + 64)) return 1;
    RS_HIP(hipMemcpyAsync(c->pinned, base + o_win, win_bytes, hipMemcpyDeviceToHost, c->stream));
    RS_HIP(hipMemcpyAsync((char*)c->pinned + win_bytes, base + o_trace, tr_bytes, hipMemcpyDeviceToHost, c->stream));
    if (sync_stream(c)) return 1;
    memcpy(hw.data(), c->pinned, win_bytes);
    const double* tr = (const double*)((const char*)c->pinned + win_bytes);
    for (uint32_t w = 0; w < W; ++w) {
        d_out[w] = hw[w].d;
        iters[w] = hw[w].iters;
        for (int k = 0; k < hw[w].iters; ++k) memcpy(trace + ((size_t)w * max_outer + k) * 6, tr + ((size_t)k * W + w) * 6, 48);
    }
    return 0;
}

// Sync for the W windows of the selection, `repeats` chained calls each (the reference driver's four, core_testcode.cpp:
// 314), in ONE launch of the window executor (kernels/executor.hpp): frames of up to 512 tracks.  d0[W] in; d_out[W],
// cost[W] (loss at the returned delay), iters[W][repeats], and the trace rows of all calls of a window back to back,
// trace[W][trace_rows][6] (trace_rows >= repeats * max_outer).  Window w samples call r with stream_first + r + w * stride.
namespace {
// the LDS region of an executor wave: the one-wave class's fp64 window, and at least the fp32 window the launch chain's
// search kernel gives every other class of the selection (exec_big.hpp stages it there)
size_t exec_region_bytes(rship_ctx* c) {
    // window_plan.hpp, exec_region_for: the one-wave class's fp64 window, the fp32 window the launch chain's search kernel
    // gives that class, and never less than the decisions' staging area (compact fp64 windows had let the
    // region shrink to 7-9 KB at 4.3-6.5 kHz on small frames, below the 10 KB a window of ~90-128 one-wave frames stages
    // in its trial phase and below a 116-knot search window; eight waves per CU still fit ~3.5 KB of static LDS + 10 KB)
    uint32_t search_cap = 0;
    if (c->cls_off[1] != c->cls_off[0]) search_cap = plan_lmeds_window<1>(c, 0, 0.0, 1u).cap;
    size_t region = rs::exec_region_for(c->cls_cap64[0], compact_of(c, 0), search_cap);
    // with frames of more than 512 tracks: 16 KB at least -- the search of such a frame keeps its unit rows (12 bytes each)
    // and keys there, its L-BFGS the rows of P (24 bytes each); with ~3.6 KB of static LDS eight waves still share a CU
    if (c->n_sel != c->cls_off[1] - c->cls_off[0]) region = std::max(region, (size_t)16 * 1024);
    for (int k = 1; k < 5; ++k)
        if (c->cls_off[k + 1] != c->cls_off[k]) {
            const WinPlan wp = plan_lmeds_window<1>(c, k, 0.0, 1u);
            region = std::max(region, (size_t)(wp.cap ? wp.cap : (uint32_t)kWinMax) * 64u);
        }
    return region;
}
} // namespace
int rship_exec_supported(rship_ctx* c) {
    // any frame size: frames of up to 512 tracks are one-wave tasks, larger ones are evaluated by one wave in their own
    // class's (four-wave) association (kernels/exec_big.hpp).  Not with RSSYNC_FORCE_BIG (every frame through the
    // large-frame kernels: a test mode of the launch chain) and not where a class's search window outgrows a wave's LDS.
    if (c->force_big || !c->n_sel || c->n_sel >= (1u << 24)) return 0; // (a queue cell holds the slot in 24 bits)
    for (uint32_t i : c->h_sel)
        if (c->h_frame_n[i] < 2) return 0;
    if (c->max_n > c->one_wave_max && c->max_n > c->exec_big_max) return 0; // (a frame that large is better off with four waves: the chain of launches)
    // frames of 6145 .. 8192 tracks (class 4) run the search in the tile kernel's EIGHT-wave shape since round 6; the executor's
    // one-wave emulation (kernels/exec_big.hpp) reproduces the FOUR-wave association (classes 1 .. 3) and the large-frame
    // kernel's (class 5): a selection with a class-4 frame is the chain's whatever RSSYNC_EXEC_BIG_MAX says
    if (c->cls_off[5] != c->cls_off[4]) return 0;
    // ... and so is a selection in which the larger frames are not the exception: a one-wave task in the four-wave
    // association is ~4x a one-wave frame's (BASELINE config 3, every frame 2048 tracks: 30 ms in the executor against
    // 9.8 ms through the chain).  Stragglers only: at most one slot in eight.
    if ((uint64_t)(c->n_sel - (c->cls_off[1] - c->cls_off[0])) * c->exec_big_share > c->n_sel) return 0;
    return exec_region_bytes(c) <= 48u * 1024u ? 1 : 0;
}

int rship_sync_exec(rship_ctx* c, const double* d0, int repeats, uint32_t stream_first, uint32_t stream_stride, uint64_t seed,
                    int max_outer, double search_center, double search_radius, double* d_out, double* cost, int32_t* iters,
                    double* trace, uint32_t trace_rows) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    const uint32_t W = c->n_grp, ns = c->n_sel;
    if (!rship_exec_supported(c)) return set_err(c, "sync_exec: this selection is not for the executor (RSSYNC_FORCE_BIG, a frame of fewer than 2 tracks, a search window beyond a wave's LDS)");
    if (c->plan_wins != W || c->plan_has_idx || c->plan_len != ns) return set_err(c, "sync_exec: the plan must be the selection's groups");
    if (max_outer <= 0 || repeats < 1 || repeats > kExecMaxCalls) return set_err(c, "sync_exec: bad iteration or call count");
    if (trace_rows < (uint32_t)repeats * (uint32_t)max_outer) return set_err(c, "sync_exec: trace too small");
    if (c->h_grp_off.size() != (size_t)W + 1) return set_err(c, "sync_exec: no selection");
    for (uint32_t w = 0; w < W; ++w)
        if (c->h_grp_off[w + 1] == c->h_grp_off[w]) return set_err(c, "sync_exec: a window without frames");
    int nf_fixed = 0;
    if (const char* e = std::getenv("RSSYNC_LOOP_FIRST_TRIALS")) { const int v = atoi(e); if (v >= 1 && v <= kMaxBt) nf_fixed = v; }
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    // One dynamic LDS region per wave serves as fp32 window, fp64 window and staging area (executor.hpp): cap64 knots x
    // 128 bytes -- 10 KB up to ~1.7 kHz of gyro rate, more for wider frames, and then fewer waves share a CU.
    const uint32_t exec_rpt = (uint32_t)small_rpt(c->cls_max_n[0]);
    const size_t region = exec_region_bytes(c);
    if (region < (size_t)kExecStage * sizeof(double)) return set_err(c, "sync_exec: the wave's LDS region is smaller than the decisions' staging area");
    const bool compact = compact_of(c, 0);
    const uint32_t cap64 = (uint32_t)(region / (compact ? 64u : 128u)); // knots of fp64 window the region holds (>= the one-wave class's plan)
    const bool with_big = ns != c->cls_off[1] - c->cls_off[0]; // frames of more than 512 tracks in the selection
    // what the chip holds at once: eight waves per CU at most (more would only idle), fewer where the LDS (~17 KB per wave
    // at 80 knots) or the registers (RPT = 8 with big frames: one wave per SIMD) say so
    uint32_t per_cu = 8;
    {
        int occ = 0;
        uint32_t fixed_lds = 0;
        hipError_t e = hipSuccess;
#define RS_EXEC_OCC(R)                                                                                                                  \
    case R:                                                                                                                             \
        fixed_lds = with_big ? static_lds_of(sync_exec_kernel<R, true>) : static_lds_of(sync_exec_kernel<R, false>);                    \
        e = with_big ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sync_exec_kernel<R, true>, 64, region)                        \
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sync_exec_kernel<R, false>, 64, region);                      \
        break;
        switch (exec_rpt) {
            RS_EXEC_OCC(1)
            RS_EXEC_OCC(2)
            RS_EXEC_OCC(3)
            RS_EXEC_OCC(4)
            default: RS_EXEC_OCC(8)
        }
#undef RS_EXEC_OCC
        if (e != hipSuccess) { (void)hipGetLastError(); occ = 0; }
        const size_t per_wave = fixed_lds + region + 512; // (the LDS bound, as rounds 3-4 computed it)
        const uint32_t fit = (uint32_t)((size_t)c->lds_per_cu / per_wave);
        if (fit < per_cu) per_cu = fit < 1 ? 1 : fit;
        if (occ >= 1 && (uint32_t)occ < per_cu) per_cu = (uint32_t)occ; // (and what the registers allow)
    }
    uint32_t waves = (uint32_t)n_cu * per_cu;
    if (waves > ns) waves = ns;
    // ring of {lap, slot} cells, several times the entries that can be outstanding (<= ns) plus the numbers idle waves
    // have claimed ahead (<= waves)
    uint32_t q_cap = 256, q_shift = 8;
    while (q_cap < 4 * (ns + 9 * (ns - (c->cls_off[1] - c->cls_off[0])) + waves)) { q_cap *= 2; ++q_shift; } // (a big frame's trials are up to ten tasks)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_win = take(W * sizeof(ExecWin));
    const size_t o_inkd = take(W * 4), o_infd = take(W * 4), o_stream = take(W * 4);
    const size_t o_inkd64 = take(W * 4), o_infd64 = take(W * 8); // the search's delay split in fp64 (near-static frames: fp64 rows in place)
    const size_t o_mokd = take(W * 4), o_mofd = take(W * 8), o_lgkd = take(W * 4), o_lgfd = take(W * 8);
    const size_t o_trkd = take((size_t)kMaxBt * W * 4), o_trfd = take((size_t)kMaxBt * W * 8);
    // the four control words -- queue head, queue tail, windows done, abort flag -- each on a 128-byte line of its own:
    // every pop adds to the head, every push to the tail, and the idle waves' (rare) looks at done / abort / tail
    // would otherwise all land on the line the working waves' atomics are queued on (MI355X_MICROARCH.md, atomics:
    // one contended line is served serially)
    constexpr size_t kCtlStride = 128;
    const size_t o_q = take((size_t)q_cap * 8), o_ctl = take(4 * kCtlStride);
    const size_t o_trace = take((size_t)W * trace_rows * 48);
    // frames of more than 512 tracks: the table slot -> (entry, class) and the entries' scratch (kernels/exec_big.hpp)
    const uint32_t n_big = ns - (c->cls_off[1] - c->cls_off[0]);
    uint32_t big_rows_n = 0;
    for (int k = 1; k < kNumClasses; ++k) big_rows_n = std::max(big_rows_n, c->cls_max_n[k]);
    big_rows_n = big_rows(big_rows_n);
    const size_t o_info = take(n_big ? (size_t)ns * 4 : 0), o_big = take(n_big ? sizeof(ExecBig) : 0);
    const size_t o_wboff = take(n_big ? (size_t)(W + 1) * 4 : 0), o_blist = take(n_big ? (size_t)n_big * 4 : 0);
    if (ensure(c, c->loop_state, off) || ensure(c, c->part, (size_t)2 * kMaxBt * ns * 8) || ensure(c, c->flags, 16)) return 1;
    if (n_big && (ensure(c, c->big_scratch, (size_t)n_big * big_rows_n * kExecBigFloats * 4) ||
                  ensure(c, c->mo_scratch, (size_t)n_big * 3 * big_rows_n * 8)))
        return 1;
    char* base = (char*)c->loop_state.p;
    std::vector<uint32_t> slot_info, win_big_off, big_list;
    if (n_big) {
        slot_info.assign(ns, 0u);
        uint32_t entry = 0;
        for (int k = 1; k < kNumClasses; ++k)
            for (uint32_t e = c->cls_off[k]; e < c->cls_off[k + 1]; ++e) slot_info[c->h_cls_slots[e]] = ((entry++ + 1u) << 3) | (uint32_t)k;
        win_big_off.assign(W + 1, 0u); // the big frames of every window (their trials are a task each: executor.hpp)
        for (uint32_t w = 0; w < W; ++w) {
            for (uint32_t j = c->h_grp_off[w]; j < c->h_grp_off[w + 1]; ++j)
                if (slot_info[j]) big_list.push_back(j);
            win_big_off[w + 1] = (uint32_t)big_list.size();
        }
        RS_HIP(hipMemcpyAsync(base + o_info, slot_info.data(), (size_t)ns * 4, hipMemcpyHostToDevice, c->stream));
        RS_HIP(hipMemcpyAsync(base + o_wboff, win_big_off.data(), (size_t)(W + 1) * 4, hipMemcpyHostToDevice, c->stream));
        RS_HIP(hipMemcpyAsync(base + o_blist, big_list.data(), (size_t)n_big * 4, hipMemcpyHostToDevice, c->stream));
    }

    // host-side initial state: every window in its first call, its slots queued for the search
    std::vector<ExecWin> hw(W);
    std::vector<int32_t> in_kd(W), in_kd64(W);
    std::vector<float> in_fd(W);
    std::vector<double> in_fd64(W);
    std::vector<uint32_t> streams(W);
    std::vector<unsigned long long> queue(q_cap, 0ull);
    const double kClamp = (double)(1 << 29);
    for (uint32_t w = 0; w < W; ++w) {
        ExecWin& e = hw[w];
        e = ExecWin{};
        e.s.d = d0[w];
        e.s.active = 1;
        e.s.hit = -1;
        e.s.nf = nf_fixed ? nf_fixed : kHalfBt;
        e.s.x0 = d0[w] - .3 * 0.0;
        e.slot0 = c->h_grp_off[w];
        e.n_slots = c->h_grp_off[w + 1] - c->h_grp_off[w];
        e.phase = kPhInit;
        e.remaining = e.n_slots;
        streams[w] = stream_first + w * stream_stride;
        // the fp32 split of the host solver (sync_problem.cpp: split_delay)
        const double D = d0[w] * c->fs;
        in_kd64[w] = 0; in_fd64[w] = 0.0; // (sync_problem.cpp: split_delay64)
        if (std::isfinite(D)) {
            const double fl64 = std::floor(D);
            if (fl64 > kClamp) in_kd64[w] = (int32_t)kClamp;
            else if (fl64 < -kClamp) in_kd64[w] = (int32_t)-kClamp;
            else { in_kd64[w] = (int32_t)fl64; in_fd64[w] = D - fl64; }
        }
        if (!std::isfinite(D)) { in_kd[w] = 0; in_fd[w] = 0.f; }
        else {
            double fl = std::floor(D);
            float f = (float)(D - fl);
            if (f >= 1.0f) { f = 0.f; fl += 1.0; }
            fl = std::min(std::max(fl, -kClamp), kClamp);
            in_kd[w] = (int32_t)fl;
            in_fd[w] = f;
        }
    }
    for (uint32_t j = 0; j < ns; ++j) queue[j] = (1ull << 32) | ((unsigned long long)kPhInit << 24) | j; // lap 1, the search
    uint32_t ctl[4 * kCtlStride / 4] = {};
    ctl[kCtlStride / 4] = ns; // head 0, tail ns, done 0, abort 0
    RS_HIP(hipMemcpyAsync(base + o_win, hw.data(), W * sizeof(ExecWin), hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemcpyAsync(base + o_inkd, in_kd.data(), W * 4, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemcpyAsync(base + o_infd, in_fd.data(), W * 4, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemcpyAsync(base + o_inkd64, in_kd64.data(), W * 4, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemcpyAsync(base + o_infd64, in_fd64.data(), W * 8, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemcpyAsync(base + o_stream, streams.data(), W * 4, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemcpyAsync(base + o_q, queue.data(), (size_t)q_cap * 8, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemcpyAsync(base + o_ctl, ctl, sizeof(ctl), hipMemcpyHostToDevice, c->stream));

    ExecParams ep{};
    {
        double t = 1e-3; // t0, decay = 0.1 (core_private.cpp:226): the host loop's sequence, bit for bit
        for (int i = 0; i <= kMaxBt; ++i) { ep.lp.ts[i] = t; t *= .1; }
    }
    ep.lp.fs = c->fs;
    ep.lp.c_armijo = 2e-4;
    ep.lp.delay_b = .3;
    ep.lp.search_center = search_center;
    ep.lp.search_radius = search_radius;
    ep.lp.max_outer = max_outer;
    ep.lp.nf_fixed = nf_fixed;
    ep.lp.nf_floor = kHalfBt; // (the executor runs frames of up to 512 tracks: a trial is microseconds)
    ep.win = (ExecWin*)(base + o_win);
    ep.n_win = W;
    ep.n_sel = ns;
    ep.grp = (const uint32_t*)c->grp.p;
    ep.in_kd = (int32_t*)(base + o_inkd); ep.in_fd = (float*)(base + o_infd);
    ep.win_stream = (uint32_t*)(base + o_stream);
    ep.mo_kd = (int32_t*)(base + o_mokd); ep.mo_fd = (double*)(base + o_mofd);
    ep.lg_kd = (int32_t*)(base + o_lgkd); ep.lg_fd = (double*)(base + o_lgfd);
    ep.tr_kd = (int32_t*)(base + o_trkd); ep.tr_fd = (double*)(base + o_trfd);
    ep.part = (double*)c->part.p;
    ep.chunk_off = (const uint32_t*)c->plan_chunk_off.p;
    ep.win_chunk_off = (const uint32_t*)c->plan_win_off.p;
    ep.trace = (double*)(base + o_trace);
    ep.trace_rows = trace_rows;
    ep.stream_first = stream_first;
    ep.stream_stride = stream_stride;
    ep.repeats = repeats;
    ep.q = (unsigned long long*)(base + o_q);
    ep.q_mask = q_cap - 1;
    ep.q_shift = q_shift;
    ep.q_head = (uint32_t*)(base + o_ctl);
    ep.q_tail = (uint32_t*)(base + o_ctl + kCtlStride);
    ep.done = (uint32_t*)(base + o_ctl + 2 * kCtlStride);
    ep.abort_flag = (uint32_t*)(base + o_ctl + 3 * kCtlStride);
    // the watchdog: nobody has pushed anything for this long (s_memrealtime ticks, 100 MHz) while a wave waits for a
    // task -- something is wrong.  (The longest task is one frame's 200-iteration L-BFGS: milliseconds.)
    double wd_s = 5.0;
    if (const char* s = std::getenv("RSSYNC_EXEC_WATCHDOG_S")) { const double v = atof(s); if (v > 0.0 && v < 600.0) wd_s = v; }
    ep.watchdog_ticks = (unsigned long long)(wd_s * 1e8);
    // GuessMotion's search (fp32, one wave per frame)
    ep.init.rays_a = (const f4*)c->rays_a.p;
    ep.init.rays_b = (const f4*)c->rays_b.p;
    ep.init.frames = (const FrameRec*)c->frames.p;
    ep.init.sel = (const uint32_t*)c->sel.p;
    ep.init.n_sel = ns;
    ep.init.slots = nullptr;
    ep.init.n_slots = ns;
    ep.init.coef = (const f4*)c->coef.p;
    ep.init.n_knots = (int)c->n_knots;
    ep.init.kd = ep.in_kd;
    ep.init.fd = ep.in_fd;
    ep.init.n_cand = 1; ep.init.chunk = 1; ep.init.n_chunks = 1;
    ep.init.n_hyp = 200; // core_private.cpp:127
    ep.init.seed = seed;
    ep.init.win_stream = ep.win_stream;
    ep.init.grp = ep.grp;
    ep.init.n_grp = W;
    ep.init.best_h = (int32_t*)c->init_h.p;
    ep.init.flags = (uint32_t*)c->flags.p;
    if (!c->no_fp64_rows && c->rays64.p && c->coef64.p) { // near-static frames: the search's rows from the fp64 streams, in place
        if (ensure_redo_count(c)) return 1;
        ep.init.src64 = Rows64Src{rays64_of(c), (const d4*)c->coef64.p, (int)c->n_knots};
        ep.init.kd64 = (const int32_t*)(base + o_inkd64);
        ep.init.fd64 = (const double*)(base + o_infd64);
        ep.init.redo_count = (unsigned long long*)c->redo_count.p;
    }
    // The search's fp32 window must put every frame on the SAME spline path as the launch chain's search kernel does
    // (interior and general path round differently in fp32: a near-tie between hypotheses could fall the other way and
    // the executor would no longer return the chain's bits -- caught by RSSYNC_EXECUTOR_CHECK on a randomised case at
    // 3.2 kHz in round 4): the capacity the chain's planner chooses, 80 knots where it keeps the compiled-in window.
    {
        const WinPlan wp_init = plan_lmeds_window<1>(c, 0, 0.0, 1u);
        ep.init.win_cap = wp_init.cap ? wp_init.cap : (uint32_t)kWinMax;
        ep.init.win_whole_pair = (wp_init.cap && !wp_init.whole_pair) ? 0u : 1u; // (the compiled-in window stages whole pairs: so must this one, or tiny frames take another path)
        if ((size_t)ep.init.win_cap * 64u > region) return set_err(c, "sync_exec: the search's window does not fit the wave's LDS region");
    }
    ExecBig hb{};
    ep.big = nullptr;
    if (n_big) {
        hb.slot_info = (const uint32_t*)(base + o_info);
        hb.win_big_off = (const uint32_t*)(base + o_wboff);
        hb.big_list = (const uint32_t*)(base + o_blist);
        hb.big_tile = (float*)c->big_scratch.p;
        hb.big_P = (double*)c->mo_scratch.p; // (per entry of the table above, not of class 5's list)
        hb.big_rows = big_rows_n;
        hb.init_whole = 0;
        for (int k = 0; k < kNumClasses; ++k) {
            hb.init_cap[k] = (uint32_t)kWinMax;
            if (k >= 1 && k <= 4 && c->cls_off[k + 1] != c->cls_off[k]) {
                const WinPlan wp = plan_lmeds_window<1>(c, k, 0.0, 1u);
                hb.init_cap[k] = wp.cap ? wp.cap : (uint32_t)kWinMax;
                if (!wp.cap || wp.whole_pair) hb.init_whole |= 1u << k; // (the compiled-in window stages whole pairs, and so does the small one that stands in for it)
            }
        }
        RS_HIP(hipMemcpyAsync(base + o_big, &hb, sizeof(hb), hipMemcpyHostToDevice, c->stream));
        ep.big = (const ExecBig*)(base + o_big);
    }
    // motion
    if (fill_motion(c, ep.mo)) return 1;
    ep.mo.kd = ep.mo_kd;
    ep.mo.fd = ep.mo_fd;
    ep.mo.grp = ep.grp;
    ep.mo.seed = seed;
    ep.mo.win_stream = ep.win_stream;
    ep.mo.order = nullptr;
    ep.mo.evals_out = nullptr;
    ep.mo.win_cap = cap64;
    ep.mo.win_compact = compact ? 1u : 0u;
    ep.mo.win_bytes = (uint32_t)region;
    // loss
    ep.lo.rays = rays64_of(c);
    ep.lo.frames = (const FrameRec*)c->frames.p;
    ep.lo.sel = (const uint32_t*)c->sel.p;
    ep.lo.n_sel = ns;
    ep.lo.coef = (const d4*)c->coef64.p;
    ep.lo.n_knots = (int)c->n_knots;
    ep.lo.fs = c->fs;
    ep.lo.M = (const double*)c->M.p;
    ep.lo.k = (const double*)c->k.p;
    ep.lo.win_cap = cap64;
    ep.lo.win_compact = compact ? 1u : 0u;
    ep.lo.nb_run = 1;

    {
        ProfScope ps(c, RSHIP_K_MOTION);
#define RS_EXEC_LAUNCH(R)                                                                                            \
    case R:                                                                                                          \
        if (with_big) hipLaunchKernelGGL((sync_exec_kernel<R, true>), dim3(waves), dim3(64), region, c->stream, ep);  \
        else hipLaunchKernelGGL((sync_exec_kernel<R, false>), dim3(waves), dim3(64), region, c->stream, ep);         \
        break;
        switch (exec_rpt) {
            RS_EXEC_LAUNCH(1)
            RS_EXEC_LAUNCH(2)
            RS_EXEC_LAUNCH(3)
            RS_EXEC_LAUNCH(4)
            default: RS_EXEC_LAUNCH(8)
        }
#undef RS_EXEC_LAUNCH
    }
    RS_HIP(hipGetLastError());
    c->init_pending = false;
    // results: the windows first (how many rows each has written), then that many rows of every window
    if (ensure_pinned(c, W * sizeof(ExecWin) + 64)) return 1;
    uint32_t* h_ctl = (uint32_t*)((char*)c->pinned + W * sizeof(ExecWin));
    RS_HIP(hipMemcpyAsync(c->pinned, base + o_win, W * sizeof(ExecWin), hipMemcpyDeviceToHost, c->stream));
    for (int i = 0; i < 4; ++i) RS_HIP(hipMemcpyAsync(h_ctl + i, base + o_ctl + i * kCtlStride, 4, hipMemcpyDeviceToHost, c->stream));
    if (sync_stream(c)) return 1;
#if RSSYNC_EXEC_STATS
    {
        unsigned long long st[48];
        RS_HIP(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_exec_stats), sizeof(st)));
        const unsigned long long zero[48] = {};
        RS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_exec_stats), zero, sizeof(zero)));
        std::vector<unsigned long long> zl(4096, 0ull);
        RS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_exec_tlast), zl.data(), zl.size() * 8));
        const char* nm[7] = {"init", "motion", "grad", "trials", "final", "decide", "pop"};
        const int cnt[7] = {8, 9, 10, 11, 12, 13, 14};
        for (int i = 0; i < 7; ++i)
            fprintf(stderr, "exec %-7s %9llu x  %10.3f ms wave-time  (%.2f us each)\n", nm[i], st[cnt[i]], st[i] * 1e-5,
                    st[cnt[i]] ? st[i] * 1e-2 / st[cnt[i]] : 0.0);
        for (int ph = 0; ph < 5; ++ph) // a window's phases on the wall clock: its tasks pushed -> its last task done
            if (st[32 + ph])
                fprintf(stderr, "exec phase %-7s %9llu x  %.2f us each (wall)\n", nm[ph], st[32 + ph], st[16 + ph] * 1e-2 / st[32 + ph]);
    }
#endif
    c->exec_last[0] = h_ctl[0]; c->exec_last[1] = h_ctl[1]; c->exec_last[2] = q_cap; c->exec_last[3] = waves;
    if (h_ctl[3]) {
        const ExecWin* e = (const ExecWin*)c->pinned;
        std::string st = "sync_exec: watchdog (a wave polled an empty task queue for seconds); queue head " + std::to_string(h_ctl[0]) +
                         " tail " + std::to_string(h_ctl[1]) + ", windows done " + std::to_string(h_ctl[2]) + " of " + std::to_string(W);
        for (uint32_t w = 0; w < W && w < 4; ++w)
            st += "; window " + std::to_string(w) + ": phase " + std::to_string(e[w].phase) + " call " + std::to_string(e[w].call) +
                  " tasks left " + std::to_string(e[w].remaining) + " of " + std::to_string(e[w].n_slots) + " iterations " +
                  std::to_string(e[w].s.iters) + " active " + std::to_string(e[w].s.active) + " trial phase " + std::to_string(e[w].s.phase);
        return set_err(c, st);
    }
    if (h_ctl[2] != W) return set_err(c, "sync_exec: " + std::to_string(h_ctl[2]) + " of " + std::to_string(W) + " windows finished");
    memcpy(hw.data(), c->pinned, W * sizeof(ExecWin));
    uint32_t rows_max = 0;
    for (uint32_t w = 0; w < W; ++w) rows_max = std::max(rows_max, (uint32_t)hw[w].trace_base);
    if (rows_max) {
        if (ensure_pinned(c, (size_t)W * rows_max * 48 + 64)) return 1;
        RS_HIP(hipMemcpy2DAsync(c->pinned, (size_t)rows_max * 48, base + o_trace, (size_t)trace_rows * 48, (size_t)rows_max * 48, W,
                                hipMemcpyDeviceToHost, c->stream));
        if (sync_stream(c)) return 1;
    }
    for (uint32_t w = 0; w < W; ++w) {
        d_out[w] = hw[w].s.d;
        cost[w] = hw[w].cost;
        for (int r = 0; r < repeats; ++r) iters[(size_t)w * repeats + r] = hw[w].iters_call[r];
        if (hw[w].trace_base)
            memcpy(trace + (size_t)w * trace_rows * 6, (const char*)c->pinned + (size_t)w * rows_max * 48, (size_t)hw[w].trace_base * 48);
    }
    return 0;
}

// How the spline windows of the last launches were laid out ("gyro rate"): out[0] widest frame in
// knots, out[1] knots per fp64 window (K1, K3, executor; dynamic LDS), out[2] fp32 window of the last PreSync sweep
// (0 = the 80 knots compiled into the kernel, else knots of dynamic LDS), out[3] its candidates per workgroup,
// out[4] the same for the last GuessMotion search, out[5] delays per pass of the trials' loss kernel, out[6] the widest
// frame counting only the two ends' ranges of each pair
int rship_window_info(rship_ctx* c, uint32_t out[8]) {
    out[0] = (uint32_t)c->max_span;
    out[6] = (uint32_t)c->max_ends;
    out[7] = c->cls_compact ? 1u : 0u; // the one-wave kernels' fp64 window is compact (64 bytes per knot)
    const int mk = main_class(c); // (of the class most slots of the selection belong to)
    out[1] = cap64_of(c, mk);
    out[2] = c->last_lmeds_cap;
    out[3] = c->last_lmeds_chunk;
    out[4] = c->last_init_cap;
    out[5] = loss_nb_run(c, mk);
    return 0;
}

// out[k] = rows / 256 of the LMedS tile the last PreSync sweep used for size class k (class 0 with its one-wave kernel: rows
// per lane; 0: the class was not in the selection, or is class 5): which of the kernels' shapes ran (lmeds_shape)
int rship_lmeds_shapes(rship_ctx* c, uint32_t out[6]) {
    for (int k = 0; k < kNumClasses; ++k) out[k] = c->last_lmeds_shape[k];
    return 0;
}

// TEST-VARIANTS build only: every later rship_presync_enqueue also stores the |residual| bit patterns of its sweep --
// [candidate][slot][hypothesis][cap_rows], 0xffffffff where there is no row -- for rship_debug_residuals_get (after the
// collect).  The product build has no such code in its kernels and refuses.
int rship_debug_residuals(rship_ctx* c, int on, uint32_t cap_rows) {
#if RSSYNC_TEST_VARIANTS
    if (on && !cap_rows) return set_err(c, "debug_residuals: cap_rows must be positive");
    c->dump_on = on != 0;
    c->dump_rows = cap_rows;
    return 0;
#else
    (void)on; (void)cap_rows;
    return set_err(c, "debug_residuals: only in the test-variants build (-DRSSYNC_TEST_VARIANTS=1)");
#endif
}
int rship_debug_residuals_get(rship_ctx* c, uint32_t* out, uint64_t n_words, uint32_t dims[4]) {
    DeviceGuard dev_guard(c);
    for (int i = 0; i < 4; ++i) dims[i] = c->dump_dims[i];
    const uint64_t have = (uint64_t)c->dump_dims[0] * c->dump_dims[1] * c->dump_dims[2] * c->dump_dims[3];
    if (!out) return 0;
    if (!have || n_words < have || !c->dump.p) return set_err(c, "debug_residuals_get: nothing stored, or the buffer is too small");
    RS_HIP(hipStreamSynchronize(c->stream));
    RS_HIP(hipMemcpy(out, c->dump.p, (size_t)have * 4, hipMemcpyDeviceToHost));
    return 0;
}
// out[0]: (frame, candidate) pairs of PreSync sweeps recomputed with fp64 rows so far (near-static frames: kernels/lmeds.hpp,
// "fp64 rows"); out[1]: sweeps that went through the fp64 form; out[2]: GuessMotion searches (one per frame and Sync call) that
// took their rows from the fp64 streams.  Zero on ordinary scenes.
int rship_near_static_stats(rship_ctx* c, uint64_t out[3]) {
    DeviceGuard dev_guard(c);
    out[0] = out[1] = out[2] = 0;
    if (c->redo_count.p) {
        RS_HIP(hipStreamSynchronize(c->stream));
        unsigned long long v[2] = {0, 0};
        RS_HIP(hipMemcpy(v, c->redo_count.p, 16, hipMemcpyDeviceToHost));
        out[0] = v[0];
        out[2] = v[1];
    }
    out[1] = c->near_launches;
    return 0;
}
int rship_exec_stats(rship_ctx* c, uint32_t out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = c->exec_last[i];
    return 0;
}

int rship_get_motion(rship_ctx* c, double* M, double* k, uint32_t cap, uint32_t* n) {
    DeviceGuard dev_guard(c);
    if (sync_stream(c)) return 1;
    uint32_t cnt = c->n_sel < cap ? c->n_sel : cap;
    if (cnt) {
        RS_HIP(hipMemcpy(M, c->M.p, (size_t)cnt * 24, hipMemcpyDeviceToHost));
        RS_HIP(hipMemcpy(k, c->k.p, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    }
    if (n) *n = cnt;
    return 0;
}

int rship_set_motion(rship_ctx* c, const double* M, const double* k, uint32_t n) {
    DeviceGuard dev_guard(c);
    if (n != c->n_sel) return set_err(c, "set_motion: count differs from the selection");
    if (sync_stream(c)) return 1;
    if (n) {
        RS_HIP(hipMemcpy(c->M.p, M, (size_t)n * 24, hipMemcpyHostToDevice));
        RS_HIP(hipMemcpy(c->k.p, k, (size_t)n * 8, hipMemcpyHostToDevice));
    }
    return 0;
}

// Can this process use RCCL at all?  Resolves the library and every entry point the exchange uses; no communication.
// (rssync_amd/dist.py: the ranks agree on the outcome of this BEFORE any of them enters the collective
// ncclCommInitRank, where a rank that cannot follow would leave the others waiting.)
int rship_rccl_preflight(rship_ctx* c) {
    for (const char* nm : {"ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy", "ncclCommAbort"})
        if (!rccl_sym(c, nm)) return 1;
    return 0;
}
const char* rship_rccl_library(rship_ctx* c) { return c->rccl_path.c_str(); }

int rship_rccl_unique_id(rship_ctx* c, void* id128) {
    auto get = (rccl_get_id_fn)rccl_sym(c, "ncclGetUniqueId");
    if (!get) return 1;
    RcclId id;
    const int rc = get(&id);
    if (rc) return set_err(c, "rccl: ncclGetUniqueId failed (" + std::to_string(rc) + ")");
    memcpy(id128, id.bytes, sizeof(id.bytes));
    return 0;
}

int rship_rccl_init(rship_ctx* c, const void* id128, int rank, int world) {
    DeviceGuard dev_guard(c);
    if (c->rccl_comm) return set_err(c, "rccl: already initialised");
    if (world < 1 || rank < 0 || rank >= world) return set_err(c, "rccl: bad rank / world size");
    auto init = (rccl_init_fn)rccl_sym(c, "ncclCommInitRank");
    if (!init || !rccl_sym(c, "ncclAllReduce")) return 1;
    RcclId id;
    memcpy(id.bytes, id128, sizeof(id.bytes));
    const int rc = init(&c->rccl_comm, world, id, rank);
    if (rc) {
        c->rccl_comm = nullptr;
        return set_err(c, "rccl: ncclCommInitRank failed (" + std::to_string(rc) + ")");
    }
    return 0;
}

int rship_rccl_allreduce(rship_ctx* c, double* buf, uint64_t n) {
    DeviceGuard dev_guard(c);
    if (!c->rccl_comm) return set_err(c, "rccl: not initialised");
    if (!n) return 0;
    auto allreduce = (rccl_allreduce_fn)rccl_sym(c, "ncclAllReduce");
    if (!allreduce) return 1;
    if (ensure(c, c->rccl_buf, (size_t)n * 8)) return 1;
    RS_HIP(hipMemcpyAsync(c->rccl_buf.p, buf, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    const int rc = allreduce(c->rccl_buf.p, c->rccl_buf.p, (size_t)n, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->rccl_comm, c->stream);
    if (rc) {
        set_err(c, "rccl: ncclAllReduce failed (" + std::to_string(rc) + ")");
        rccl_abort_comm(c);
        return 1;
    }
    RS_HIP(hipMemcpyAsync(buf, c->rccl_buf.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    RS_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

uint64_t rship_loop_exchanges(const rship_ctx* c) { return c->loop_exchanges; }
int rship_set_loop_exchange(rship_ctx* c, rship_loop_exchange_fn fn, void* user) {
    c->loop_xchg = fn;
    c->loop_xchg_user = user;
    return 0;
}

int rship_rccl_shutdown(rship_ctx* c) {
    DeviceGuard dev_guard(c);
    if (!c->rccl_comm) return 0;
    RS_HIP(hipStreamSynchronize(c->stream));
    auto destroy = (rccl_destroy_fn)rccl_sym(c, "ncclCommDestroy");
    if (!destroy) return 1;
    const int rc = destroy(c->rccl_comm);
    c->rccl_comm = nullptr;
    if (rc) return set_err(c, "rccl: ncclCommDestroy failed (" + std::to_string(rc) + ")");
    return 0;
}

int rship_debug_rays(rship_ctx* c, uint32_t frame_index, float* a4, float* b4, uint32_t cap_rays) {
    DeviceGuard dev_guard(c);
    if (frame_index >= c->n_frames) return set_err(c, "debug_rays: index out of range");
    rship_frame rec;
    hipError_t e = hipMemcpy(&rec, (const rship_frame*)c->frames.p + frame_index, sizeof(rec), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(c, "debug_rays", e);
    if (rec.n_rays > cap_rays) return set_err(c, "debug_rays: output too small");
    e = hipMemcpy(a4, (const f4*)c->rays_a.p + rec.ray_offset, (size_t)rec.n_rays * 16, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(b4, (const f4*)c->rays_b.p + rec.ray_offset, (size_t)rec.n_rays * 16, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(c, "debug_rays", e);
    return 0;
}

int rship_debug_problem(rship_ctx* c, uint32_t sel_index, int32_t kd, float fd, float* P, float* dP, uint32_t cap_rows) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (sel_index >= c->n_sel) return set_err(c, "debug_problem: index out of range");
    uint32_t fi = c->h_sel[sel_index];
    uint32_t n = c->h_frame_n[fi];
    if (n > cap_rows) return set_err(c, "debug_problem: output too small");
    TempBuf out;
    if (ensure(c, out, (size_t)n * 24 + 64)) return 1;
    DebugParams p{};
    p.rays_a = (const f4*)c->rays_a.p;
    p.rays_b = (const f4*)c->rays_b.p;
    p.frames = (const FrameRec*)c->frames.p;
    p.fi = fi;
    p.coef = (const f4*)c->coef.p;
    p.n_knots = (int)c->n_knots;
    p.fs = (float)c->fs;
    p.kd = kd;
    p.fd = fd;
    p.P = (float*)out.p;
    p.dP = dP ? (float*)out.p + (size_t)n * 3 : nullptr;
    hipLaunchKernelGGL(debug_problem_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(P, out.p, (size_t)n * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess && dP) e = hipMemcpy(dP, (float*)out.p + (size_t)n * 3, (size_t)n * 12, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(c, "debug_problem", e);
    return 0;
}

int rship_debug_problem64(rship_ctx* c, uint32_t sel_index, int32_t kd, double fd, double* P, double* dP, uint32_t cap_rows) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (sel_index >= c->n_sel) return set_err(c, "debug_problem: index out of range");
    uint32_t fi = c->h_sel[sel_index];
    uint32_t n = c->h_frame_n[fi];
    if (n > cap_rows) return set_err(c, "debug_problem: output too small");
    TempBuf out;
    if (ensure(c, out, (size_t)n * 48 + 64)) return 1;
    Debug64Params p{};
    p.rays = rays64_of(c);
    p.frames = (const FrameRec*)c->frames.p;
    p.fi = fi;
    p.coef = (const d4*)c->coef64.p;
    p.n_knots = (int)c->n_knots;
    p.fs = c->fs;
    p.kd = kd;
    p.fd = fd;
    p.P = (double*)out.p;
    p.dP = dP ? (double*)out.p + (size_t)n * 3 : nullptr;
    hipLaunchKernelGGL(debug_problem64_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(P, out.p, (size_t)n * 24, hipMemcpyDeviceToHost);
    if (e == hipSuccess && dP) e = hipMemcpy(dP, (double*)out.p + (size_t)n * 3, (size_t)n * 24, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(c, "debug_problem64", e);
    return 0;
}

int rship_debug_math64(rship_ctx* c, int op, const double* a, const double* b, double* out, uint32_t n) {
    DeviceGuard dev_guard(c);
    if (op < 0 || op > 5 || !n) return set_err(c, "debug_math64: bad arguments");
    const uint32_t blocks = (n + 63) / 64;
    const size_t n_out = op == 2 ? 2 * (size_t)n : (op == 4 ? blocks : n);
    TempBuf da, db, dout;
    if (ensure(c, da, (size_t)n * 8) || ensure(c, dout, n_out * 8) || (b && ensure(c, db, (size_t)n * 8))) return 1;
    hipError_t e = hipMemcpy(da.p, a, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) e = hipMemcpy(db.p, b, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(debug_math64_kernel, dim3(blocks), dim3(64), 0, c->stream, op, (const double*)da.p,
                           b ? (const double*)db.p : nullptr, (double*)dout.p, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, dout.p, n_out * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(c, "debug_math64", e);
    return 0;
}

// the pending winners of GuessMotion's hypothesis search, per slot of the selection: read (get) and / or
// overwrite (set) them before the next motion launch turns them into M and k
int rship_debug_init_h(rship_ctx* c, int32_t* get, const int32_t* set, uint32_t n) {
    DeviceGuard dev_guard(c);
    if (n != c->n_sel) return set_err(c, "debug_init_h: count differs from the selection");
    if (!n) return 0;
    RS_HIP(hipStreamSynchronize(c->stream));
    if (get) RS_HIP(hipMemcpy(get, c->init_h.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (set) RS_HIP(hipMemcpy(c->init_h.p, set, (size_t)n * 4, hipMemcpyHostToDevice));
    return 0;
}

// dynamic trip counts of the LMedS tile kernel (variant builds with -DRSSYNC_K2_COUNTERS=1 only; zeros otherwise)
int rship_debug_k2_counters(rship_ctx* c, uint64_t out[16], int reset) {
    DeviceGuard dev_guard(c);
    for (int i = 0; i < 16; ++i) out[i] = 0;
#if RSSYNC_K2_COUNTERS
    RS_HIP(hipStreamSynchronize(c->stream));
    RS_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k2_counters), 16 * sizeof(uint64_t)));
    if (reset) {
        const uint64_t zero[16] = {};
        RS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_k2_counters), zero, sizeof(zero)));
    }
#else
    (void)reset;
#endif
    return 0;
}

int rship_debug_select(rship_ctx* c, const float* vals, uint32_t n_problems, uint32_t n, uint32_t kq,
                       const float* upper, uint32_t* out) {
    DeviceGuard dev_guard(c);
    if (n > 2048 || !n_problems) return set_err(c, "debug_select: bad sizes");
    TempBuf dv, du, dout;
    if (ensure(c, dv, (size_t)n_problems * n * 4) || ensure(c, dout, (size_t)n_problems * 8)) return 1;
    if (upper && ensure(c, du, (size_t)n_problems * 4)) return 1;
    hipError_t e = hipMemcpy(dv.p, vals, (size_t)n_problems * n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && upper) e = hipMemcpy(du.p, upper, (size_t)n_problems * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(debug_select_kernel, dim3(n_problems), dim3(64), 0, c->stream, (const float*)dv.p, n, kq,
                           upper ? (const float*)du.p : nullptr, (uint32_t*)dout.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, dout.p, (size_t)n_problems * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(c, "debug_select", e);
    return 0;
}

int rship_profile_enable(rship_ctx* c, int on) {
    if (sync_stream(c)) return 1;
    c->prof = on != 0;
    return 0;
}

int rship_profile_get(rship_ctx* c, int kind, uint64_t* launches, double* total_ms) {
    if (kind < 0 || kind >= RSHIP_K_COUNT) return set_err(c, "profile: bad kind");
    if (sync_stream(c)) return 1;
    if (launches) *launches = c->launches[kind];
    if (total_ms) *total_ms = c->total_ms[kind];
    return 0;
}

int rship_profile_reset(rship_ctx* c) {
    if (sync_stream(c)) return 1;
    for (int i = 0; i < RSHIP_K_COUNT; ++i) { c->launches[i] = 0; c->total_ms[i] = 0; }
    return 0;
}

} // extern "C"
