// sync_problem.cpp -- host side of librssync_core.so: the ISyncProblem
// implementation that drives the gfx950 kernels through rssync_hip.h, and the
// flat C-ABI of rssync_c.h.
//
// What stays on the host, and why (DESIGN.md "Host / device split"):
//   * the O(1) part of the gyro setters (argument checks, the panics' wording); integration,
//     resampling and the spline solve run on the device (rship_gyro_*);
//   * fp64 copies of the tracks, their packing into the device layout, and the split of
//     every time into integer knot + fp32 fraction;
//   * the optimiser control flow of Sync (backtracking, momentum, convergence counters),
//     which consumes a handful of reduced doubles per step.
// Everything that touches rays runs on the device; there is no CPU fallback.
//
// Reference (VladimirP1/rs-sync, src/) counterparts are cited per function.
#include "../../include/rssync.h"
#include "../../include/rssync_c.h"
#include "../../include/rssync_hip.h"
#include "roctx_ranges.hpp"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// error convention: core_support/panic.cpp:7-15

struct PanicError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

int g_panic_mode = 0; // 0 = reference behaviour, 1 = throw (caught by the C-ABI)
thread_local std::string g_last_error;

[[noreturn]] void panic(const std::string& reason) {
    if (g_panic_mode == 1) throw PanicError(reason);
    {
        std::ofstream out("panic.txt");
        out << reason << std::endl;
    }
    std::cerr << "rssync panic: " << reason << std::endl;
    std::exit(1);
}

bool all_finite(const double* v, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(v[i])) return false;
    return true;
}

constexpr uint32_t kStreamSyncInit = 0x80000000u; // sampler stream of Sync's GuessMotion (+ call counter)
constexpr uint32_t kStreamDebug = 0x40000000u;    // sampler streams of DebugPreSync (+ point index)
constexpr int32_t kKnotClamp = 1 << 29;

// delay (s) -> device representation: delay * fs = kd + fd, kd integer, fd in [0,1)
struct DelaySplit {
    int32_t kd;
    float fd;
};

DelaySplit split_delay(double delay, double fs) {
    double D = delay * fs;
    if (!std::isfinite(D)) return {0, 0.f};
    double fl = std::floor(D);
    float fd = (float)(D - fl);
    if (fd >= 1.0f) { fd = 0.f; fl += 1.0; }
    if (fl > (double)kKnotClamp) fl = (double)kKnotClamp;
    if (fl < -(double)kKnotClamp) fl = -(double)kKnotClamp;
    return {(int32_t)fl, fd};
}

// the same split for the fp64 kernels: the fraction keeps full precision (D - floor(D) is exact)
struct DelaySplit64 {
    int32_t kd;
    double fd;
};

DelaySplit64 split_delay64(double delay, double fs) {
    double D = delay * fs;
    if (!std::isfinite(D)) return {0, 0.0};
    double fl = std::floor(D);
    if (fl > (double)kKnotClamp) return {kKnotClamp, 0.0};
    if (fl < -(double)kKnotClamp) return {-kKnotClamp, 0.0};
    return {(int32_t)fl, D - fl};
}

// Host staging of the track data (core_private.hpp:8-13 FrameData; "copy at call time" is the
// contract of core_private.cpp:192-203): one record per frame in a chunked arena of pinned memory,
// addressed by a single running offset in doubles.  The device keeps a raw buffer with the same
// offsets (rship_upload_raw); records are never moved, a frame that is set again gets a new record
// and the old one is dead space until the problem is destroyed.
class Arena {
   public:
    ~Arena() {
        for (auto& s : slabs_) {
            if (s.pinned) rship_host_free(s.p);
            else std::free(s.p);
        }
    }
    // space for n doubles that does not straddle a slab; returns its arena offset
    uint64_t alloc(uint64_t n, double** out) {
        if (slabs_.empty() || slabs_.back().used + n > slabs_.back().cap) {
            Slab s;
            s.base = size(); // offsets stay dense: the unused tail of the previous slab has none
            s.cap = std::max<uint64_t>(n, kSlabDoubles);
            s.p = (double*)rship_host_alloc((size_t)s.cap * 8);
            s.pinned = s.p != nullptr;
            if (!s.p) s.p = (double*)std::malloc((size_t)s.cap * 8); // the runtime refused to pin
            if (!s.p) return UINT64_MAX;
            slabs_.push_back(s);
        }
        Slab& s = slabs_.back();
        *out = s.p + s.used;
        const uint64_t off = s.base + s.used;
        s.used += n;
        return off;
    }
    uint64_t size() const { return slabs_.empty() ? 0 : slabs_.back().base + slabs_.back().used; }
    // contiguous host ranges covering arena offsets [lo, hi): fn(host pointer, arena offset, count)
    template <typename F>
    void ranges(uint64_t lo, uint64_t hi, F&& fn) const {
        for (const Slab& s : slabs_) {
            const uint64_t a = std::max(lo, s.base), b = std::min(hi, s.base + s.used);
            if (a < b) fn(s.p + (a - s.base), a, b - a);
        }
    }

   private:
    static constexpr uint64_t kSlabDoubles = (64u << 20) / 8; // 64 MB slabs
    struct Slab {
        double* p = nullptr;
        uint64_t base = 0, cap = 0, used = 0;
        bool pinned = false;
    };
    std::vector<Slab> slabs_;
};

struct HostFrame {
    uint64_t raw_off = 0; // arena offset of the record: rays = ts_a[n] ts_b[n] rays_a[3n] rays_b[3n]; pixels = {xa,ya,xb,yb}[n]
    uint32_t n = 0;
    double ts_min = 0, ts_max = 0; // over ts_a and ts_b: all the frame table needs from the data ...
    double a_min = 0, a_max = 0, b_min = 0, b_max = 0; // ... and per end (the kernels' spline windows stage the two ends separately)
    // frames given as tracked pixels (rssync_ext_set_track_pixels): the packing kernel undistorts
    bool from_pixels = false;
    double time_a = 0, time_b = 0, rows = 0;
    double lens[9] = {};
};

class SyncProblemHip final : public ISyncProblem {
   public:
    SyncProblemHip();
    ~SyncProblemHip() override;

    void SetGyroQuaternions(const double* data, size_t count, double sample_rate,
                            double first_timestamp) override;
    void SetGyroQuaternions(const int64_t* timestamps_us, const double* quats, size_t count) override;
    void SetTrackResult(int64_t frame, const double* ts_a, const double* ts_b, const double* rays_a,
                        const double* rays_b, size_t count) override;
    std::pair<double, double> PreSync(double initial_delay, int64_t frame_begin, int64_t frame_end,
                                      double search_step, double search_radius) override;
    std::pair<double, double> Sync(double initial_delay, int64_t frame_begin, int64_t frame_end,
                                   double search_center, double search_radius) override;
    void DebugPreSync(double initial_delay, int64_t frame_begin, int64_t frame_end, double search_radius,
                      double* delays, double* costs, int point_count) override;

    // extension entry points (rssync_c.h)
    void SetTrackPixels(int64_t frame, double time_a, double time_b, const double* px_a, const double* px_b,
                        size_t count, const double lens[9], double image_rows);
    void SetGyroRates(const double* timestamps_s, const double* rates, size_t count, const char* orientation);
    void orientation_sweep(const double* timestamps_s, const double* rates, size_t count,
                           const std::vector<std::string>& orientations, double initial_delay, int64_t frame_begin,
                           int64_t frame_end, double search_step, double search_radius, double* costs, double* delays);
    uint64_t seed = 0x5EED0000ULL;
    int max_outer = 400; // core_private.cpp:309
    bool verbose = true;
    rssync_reduce_fn reduce_fn = nullptr;
    void* reduce_user = nullptr;
    std::vector<double> trace; // 6 doubles per outer iteration of the last Sync

    double sample_rate() const { return fs_; }
    double quats_start() const { return start_; }
    const std::vector<double>& knots(); // fetched from the device when it built them
    size_t n_knots() const { return n_knots_; }
    rship_ctx* dev() { return shards_[0].ctx; }
    size_t n_devices() const { return shards_.size(); }
    void set_devices(const std::vector<int>& ids);
    void set_option(int option, int value);
    void profile_enable(int on);
    void profile_get(int kind, uint64_t* launches, double* total_ms);
    void profile_reset();
    // per-slot state in the GLOBAL slot order (window-major, frames ascending), gathered over the devices
    uint32_t get_motion(double* M, double* k, uint32_t cap);
    void set_motion(const double* M, const double* k, uint32_t n);
    void debug_problem(int64_t frame, double delay, float* P, float* dP, double* P64, double* dP64, size_t cap_rows);
    void debug_rays(int64_t frame, float* a4, float* b4, size_t cap);

    // pieces shared by the public calls and the diagnostics
    // (Rounds 2-4: ranks first agreed on the problem's largest frame here, one exchange per public call, because the
    // kernel shapes followed it.  Since round 5 a frame's kernels follow its OWN track count -- size classes in the device
    // context -- and nobody has to agree on anything: no exchange, in collective calls or diagnostics.)
    void ensure_device();
    void ensure_spline() { if (spline_dirty_) build_spline(); }
    uint32_t select(int64_t begin, int64_t end_exclusive);
    std::vector<double> sweep(const std::vector<double>& delays, uint32_t stream_base, bool panics,
                              double* frame_costs, int32_t* best_h);
    void sweep_windows(const std::vector<double>& delays, uint32_t stream_base, double* out, uint32_t* flags_out,
                       double* frame_costs, int32_t* best_h);
    void init_motion(const std::vector<double>& delays, uint32_t call_stride = 1);
    void opt_motion(const std::vector<double>& delays, uint64_t* stats);
    void finish_init(const std::vector<double>& delays);
    void init_k_simple(const std::vector<double>& delays);
    void loss(const std::vector<double>& delays, std::vector<double>& out_loss, std::vector<double>* out_grad,
              bool simplified = false);
    void select_windows(const std::vector<int64_t>& begins, const std::vector<int64_t>& ends_incl);
    void sync_windows(const std::vector<int64_t>& begins, const std::vector<int64_t>& ends_incl,
                      const std::vector<double>& initial, double search_center, double search_radius,
                      std::vector<double>& costs, std::vector<double>& delays_out, uint32_t call_stride = 1,
                      bool simplified = false);
    void sync_points(const std::vector<int64_t>& positions, int64_t window, double initial_delay, bool use_presync,
                     double presync_step, double presync_radius, int repeats, std::vector<double>& costs,
                     std::vector<double>& delays_out);
    void presync_windows(double initial_delay, const std::vector<int64_t>& begins, const std::vector<int64_t>& ends_excl,
                         double search_step, double search_radius, std::vector<double>& costs,
                         std::vector<double>& delays_out);
    std::vector<std::vector<double>> traces; // per window of the last sync_windows call
    bool native_exchange = false; // RCCL communicator inside the device context (rssync_ext_rccl_init)
    // With a reduce hook Sync's loop runs on the host (an exchange per launch).  hook_device_loop = true keeps it on
    // the device and calls the hook between the kernels instead (rship_set_loop_exchange): the same structure as with
    // the RCCL communicator, any transport -- and the way two ranks sharing one GPU can exercise that structure.
    bool hook_device_loop = false;
    bool distributed() const { return native_exchange || reduce_fn; }
    uint64_t exchange_calls = 0, exchange_doubles = 0; // sums exchanged with other ranks so far
    void rccl_shutdown();
    // (rounds 2-4: the largest per-frame track count over all ranks; a no-op since the kernels follow each frame's own size)
    void set_tracks_hint(uint32_t) {}
    void reduce(double* buf, size_t n) {
        if (distributed()) { exchange_calls += 1; exchange_doubles += n; }
        if (native_exchange) hip_check(shards_[0], rship_rccl_allreduce(shards_[0].ctx, buf, n), "rccl all-reduce");
        else if (reduce_fn) {
            const int rc = reduce_fn(buf, n, reduce_user);
            if (rc) panic("reduce hook failed (status " + std::to_string(rc) + "): the exchange of " + std::to_string(n) +
                          " doubles did not complete");
        }
    }
    const std::vector<uint32_t>& selection() const { return sel_; }
    int64_t table_id(uint32_t i) const { return table_ids_[i]; }
    size_t table_size() const { return table_ids_.size(); }
    bool has_frame(int64_t id) const { return frames_.count(id) != 0; }
    size_t frame_tracks(int64_t id) const { return frames_.at(id).n; }
    // diagnostics of the bit-exactness tests: GuessMotion's winning hypothesis per slot (global slot order) of
    // the last Sync-type call, and winners to install instead of the search's in the next one
    bool record_init = false;
    std::vector<int32_t> last_init_winners, init_override;
    void exchange_init_winners();
    uint32_t sync_calls = 0;
    // Frames of up to 512 tracks run Sync's calls of all windows as ONE device-scheduled launch (kernels/executor.hpp)
    // instead of the chain of launches: the same bits, 21 ms against 30-35 on the reference's workload
    // (profiles/r3_syncpoints.json).  RSSYNC_EXECUTOR=0 keeps the chain.  If the executor ever gives up (its
    // watchdog), the call is redone by the chain -- same values -- and this object stays with the chain.
    bool use_executor = true;
    bool executor_warned_ = false;
    bool executor_test_fail_ = false; // RSSYNC_EXECUTOR_FAIL=1 (tests): pretend it gave up
    // RSSYNC_EXECUTOR_CHECK=1 / rssync_ext_set_executor_check: every call the executor has run is run AGAIN by the launch
    // chain from the same inputs and the two results -- delays, costs, every trace row -- must be the same bits, or the
    // call panics.  A debug mode (it doubles the work): the executor's cross-workgroup hand-offs are a measured
    // protocol, not an architectural guarantee, and a stale word would otherwise be silent.
    bool executor_check = false;
    // Without the check mode, ONE executor call in executor_check_every (process-wide count; RSSYNC_EXECUTOR_CHECK_EVERY,
    // default 256, 0 = never) is verified the same way: a tripwire for the hand-off protocol in production at < 1 % of the
    // executor's time (a verified call costs about 2.5 calls: profiles/r5_syncpoints.json).
    uint32_t executor_check_every = 256;
    bool check_this_call();
    uint64_t executor_runs = 0, executor_checked = 0; // calls the executor completed / of those, verified against the chain
    // the chain re-run of a checked call runs with the executor and the progress lines off: restored on every way out,
    // exceptions (a panic of the chain run in throwing mode) included
    struct ChainRerun {
        SyncProblemHip* s;
        bool exec_was, verbose_was;
        explicit ChainRerun(SyncProblemHip* s_) : s(s_), exec_was(s_->use_executor), verbose_was(s_->verbose) { s->use_executor = false; s->verbose = false; }
        ~ChainRerun() { s->use_executor = exec_was; s->verbose = verbose_was; }
        ChainRerun(const ChainRerun&) = delete;
        ChainRerun& operator=(const ChainRerun&) = delete;
    };
    // -> true: identical.  In the check MODE (RSSYNC_EXECUTOR_CHECK=1) a difference is a panic; for a call of the production
    // SAMPLE (one in RSSYNC_EXECUTOR_CHECK_EVERY) it is reported on stderr, counted, the caller hands out the CHAIN's results
    // and the object stays with the chain from then on -- a deterministic, usable outcome instead of a sporadic panic
    bool check_against_chain(const char* what, const std::vector<double>& c_exec, const std::vector<double>& d_exec,
                             const std::vector<std::vector<double>>& tr_exec, const std::vector<double>& c_chain,
                             const std::vector<double>& d_chain);
    uint64_t executor_mismatches = 0;
    bool executor_ok(bool simplified);
    void executor_queue_stats(uint32_t out[4]) {
        out[0] = out[1] = out[2] = out[3] = 0;
        if (shards_.size() == 1 && shards_[0].ctx) (void)rship_exec_stats(shards_[0].ctx, out);
    }
    // TEST-VARIANTS build of the library: the sweep's own residuals (rssync_hip.h: rship_debug_residuals); one device only
    void debug_residuals(int on, uint32_t cap_rows) {
        ensure_device();
        if (shards_.size() != 1) panic("debug_residuals: one device only");
        hip_check(shards_[0], rship_debug_residuals(shards_[0].ctx, on, cap_rows), "debug_residuals");
    }
    void debug_residuals_get(uint32_t* out, size_t n_words, uint32_t dims[4]) {
        if (shards_.size() != 1) panic("debug_residuals: one device only");
        hip_check(shards_[0], rship_debug_residuals_get(shards_[0].ctx, out, n_words, dims), "debug_residuals_get");
    }
    // (frame, candidate) pairs of PreSync sweeps recomputed with fp64 rows / sweeps that needed it, over this object's devices
    void near_static_stats(uint64_t out[3]) {
        out[0] = out[1] = out[2] = 0;
        for (Shard& sh : shards_)
            if (sh.ctx) {
                uint64_t v[3] = {0, 0, 0};
                (void)rship_near_static_stats(sh.ctx, v);
                out[0] += v[0];
                out[1] = std::max(out[1], v[1]);
                out[2] += v[2];
            }
    }
    bool sync_exec(const std::vector<int64_t>& begins, const std::vector<int64_t>& ends_incl, const std::vector<double>& initial,
                   double search_center, double search_radius, int repeats, uint32_t stream_first, uint32_t stream_stride,
                   std::vector<double>& costs, std::vector<double>& delays_out);
    bool host_loop = false; // keep Sync's outer loop on the host even where the device could run it (tests)
    uint64_t last_best_not_last = 0; // of the last rssync_ext_opt_motion call

   private:
    // One device context per GPU this object drives (core_private.cpp:73,231,245,263 parallelise over
    // frames inside ONE object; here the frames of a shard live on one GPU).  Shards own contiguous
    // ranges of the sorted frame table, split at multiples of kChunk frames.
    struct Shard {
        rship_ctx* ctx = nullptr;
        int device = -1;
        uint32_t t0 = 0, t1 = 0;             // frame-table range [t0, t1)
        uint64_t raw_lo = 0;                 // arena offset of this device's raw offset 0
        std::vector<uint32_t> sel;           // local table indices of the current selection, in slot order
        std::vector<uint32_t> slot_global;   // global slot of each local slot
        uint32_t n_chunks = 0;               // current plan
        std::vector<uint32_t> win_chunk_off; // [windows + 1]
    };
    static constexpr uint32_t kChunk = 64;
    void hip_check(const Shard& sh, int rc, const char* what) {
        if (rc) panic(std::string("hip: ") + what + ": " + (sh.ctx ? rship_last_error(sh.ctx) : "no context") +
                      (shards_.size() > 1 ? " (device " + std::to_string(sh.device) + ")" : std::string()));
    }
    void create_shards(const std::vector<int>& ids);
    void destroy_shards();
    // options that live in the device contexts, kept here so that set_devices (new contexts) does not lose them
    int opt_lbfgs_reeval_ = 0, opt_profile_ = 0;
    // slots (table indices, window-major; off = window offsets or empty for one ungrouped window) -> per-shard
    // selections; plan windows = the given lists of slot positions (plan_off/plan_pos) or, if empty, the groups
    void apply_selection(const std::vector<uint32_t>& slots, const std::vector<uint32_t>& grp_off,
                         const std::vector<uint32_t>* plan_pos, const std::vector<uint32_t>* plan_off);
    // rows x windows sums from every shard's last enqueue: collect(shard, win_out, chunk_out)
    template <typename Collect>
    void combine(size_t rows, size_t n_win, Collect&& collect, double* out);
    void build_spline();
    void pack_frames();
    void accept_gyro(const rship_gyro_result& r);
    void upload_rates(const double* ts, const double* rates, size_t count);
    void integrate_rates(const char* orientation);
    double* stage_record(int64_t frame, uint64_t n_doubles, HostFrame& f);
    void upload_new_records();
    Arena arena_;
    uint64_t uploaded_ = 0; // arena offsets below this are on the device

    double fs_ = 0, start_ = 0;
    std::vector<double> knots_; // 4 per sample, [w,x,y,z]: the uniform setter's copy, or a cache of the device's
    bool knots_cached_ = false;
    size_t n_knots_ = 0;
    std::map<int64_t, HostFrame> frames_;
    std::vector<Shard> shards_;
    size_t plan_windows_ = 1;
    bool spline_dirty_ = true, frames_dirty_ = true;
    std::vector<int64_t> table_ids_;
    std::vector<uint32_t> sel_;
    std::vector<int64_t> window_key_; // begins + ends of the windows select_windows has on the devices; empty = none
    size_t n_windows_ = 1;
};

SyncProblemHip::SyncProblemHip() {
    if (const char* s = std::getenv("RSSYNC_SEED")) seed = std::strtoull(s, nullptr, 0);
    if (const char* s = std::getenv("RSSYNC_MAX_OUTER_ITERS")) max_outer = std::atoi(s);
    if (const char* s = std::getenv("RSSYNC_QUIET")) verbose = !(s[0] && s[0] != '0');
    if (const char* s = std::getenv("RSSYNC_HOST_LOOP")) host_loop = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_EXECUTOR")) use_executor = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_EXECUTOR_FAIL")) executor_test_fail_ = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_EXECUTOR_CHECK")) executor_check = s[0] && s[0] != '0';
    if (const char* s = std::getenv("RSSYNC_EXECUTOR_CHECK_EVERY")) { const long v = std::atol(s); if (v >= 0) executor_check_every = (uint32_t)v; }
    // RSSYNC_GPUS: how many GPUs this object spreads its frames over ("4" = devices 0..3) or which
    // ("0,2,5"); default: the calling thread's current device only
    std::vector<int> ids;
    if (const char* s = std::getenv("RSSYNC_GPUS")) {
        std::string v(s);
        if (v.find(',') == std::string::npos) {
            const int n = std::atoi(v.c_str());
            for (int i = 0; i < n; ++i) ids.push_back(i);
        } else {
            size_t pos = 0;
            while (pos <= v.size()) {
                size_t q = v.find(',', pos);
                if (q == std::string::npos) q = v.size();
                if (q > pos) ids.push_back(std::atoi(v.substr(pos, q - pos).c_str()));
                pos = q + 1;
            }
        }
    }
    if (ids.empty()) ids.push_back(-1);
    create_shards(ids);
}

SyncProblemHip::~SyncProblemHip() { destroy_shards(); }

void SyncProblemHip::create_shards(const std::vector<int>& ids) {
    for (int id : ids) {
        Shard sh;
        sh.device = id;
        int rc = rship_create(&sh.ctx, id);
        if (rc) {
            destroy_shards();
            if (rc == 5) panic("rssync: RSSYNC_K2_EXACT_SELECT is set but this build has no exact-selection kernels (test-variants build only)");
            panic("rssync: no usable HIP device (rship_create(" + std::to_string(id) + ") failed with " +
                  std::to_string(rc) + "); this library has no CPU fallback");
        }
        shards_.push_back(sh);
    }
    // what the caller had switched on applies to the new contexts as well
    if (opt_lbfgs_reeval_) set_option(RSHIP_OPT_LBFGS_REEVAL, opt_lbfgs_reeval_);
    if (opt_profile_) profile_enable(opt_profile_);
}

void SyncProblemHip::destroy_shards() {
    for (Shard& sh : shards_) rship_destroy(sh.ctx);
    shards_.clear();
}

// the GPUs this object drives (before or after the data has been set: it is uploaded again)
void SyncProblemHip::set_devices(const std::vector<int>& ids) {
    if (ids.empty()) panic("set-devices: empty device list");
    if (native_exchange) panic("set-devices: not after rccl_init");
    knots(); // the new devices rebuild the table from the knots: fetch them if only the old device has them
    destroy_shards();
    create_shards(ids);
    uploaded_ = 0;
    spline_dirty_ = true;
    frames_dirty_ = true;
    sel_.clear();
    window_key_.clear();
}

void SyncProblemHip::set_option(int option, int value) {
    if (option == RSHIP_OPT_LBFGS_REEVAL) opt_lbfgs_reeval_ = value != 0;
    for (Shard& sh : shards_) hip_check(sh, rship_set_option(sh.ctx, option, value), "set option");
}
void SyncProblemHip::profile_enable(int on) {
    opt_profile_ = on != 0;
    for (Shard& sh : shards_) hip_check(sh, rship_profile_enable(sh.ctx, on), "profile");
}

void SyncProblemHip::rccl_shutdown() {
    if (!native_exchange) return;
    hip_check(shards_[0], rship_rccl_shutdown(shards_[0].ctx), "rccl shutdown");
    native_exchange = false;
}
void SyncProblemHip::profile_reset() {
    for (Shard& sh : shards_) hip_check(sh, rship_profile_reset(sh.ctx), "profile");
}
void SyncProblemHip::profile_get(int kind, uint64_t* launches, double* total_ms) {
    uint64_t n = 0;
    double ms = 0;
    for (Shard& sh : shards_) {
        uint64_t a = 0;
        double b = 0;
        hip_check(sh, rship_profile_get(sh.ctx, kind, &a, &b), "profile");
        n += a;
        ms += b;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
}

// core_private.cpp:135-140: the samples are the knots.  Copied at call time; the table is built on the
// device at the first use (rship_gyro_uniform).
void SyncProblemHip::SetGyroQuaternions(const double* data, size_t count, double sample_rate,
                                        double first_timestamp) {
    rs::RoctxRange roctx_range("rssync:SetGyroQuaternions");
    if (count < 2) panic("set-gyro-quaternions: need at least 2 samples");
    if (count > (size_t)UINT32_MAX) panic("set-gyro-quaternions: too many samples");
    if (fs_ != sample_rate || start_ != first_timestamp) frames_dirty_ = true; // ray offsets depend on both
    fs_ = sample_rate;
    start_ = first_timestamp;
    knots_.assign(data, data + 4 * count);
    knots_cached_ = true;
    n_knots_ = count;
    spline_dirty_ = true;
}

// What the reference's timestamped setter would have complained about (core_private.cpp:147-184), in its words.
void SyncProblemHip::accept_gyro(const rship_gyro_result& r) {
    switch (r.status) {
        case RSHIP_GYRO_OK: break;
        case RSHIP_GYRO_BAD_INPUT: n_knots_ = 0; panic("set-gyro-rates: non-finite numbers");
        case RSHIP_GYRO_OUT_OF_ORDER:
            n_knots_ = 0;
            panic("set-gyro-quaternions:  timestamps out of order at pos " + std::to_string(r.bad_pos) + " (" +
                  std::to_string(r.bad_a) + " > " + std::to_string(r.bad_b) + ")");
        case RSHIP_GYRO_SHORT_GRID: n_knots_ = 0; panic("set-gyro-quaternions: resampled grid has fewer than 2 points");
        case RSHIP_GYRO_BAD_KNOT: n_knots_ = 0; panic("set-gyro-quaternions: non-finite sample after interpolation");
        case RSHIP_GYRO_BAD_START: n_knots_ = 0; panic("set-gyro-quaternions: non-finite first timestamp. wtf?");
        case RSHIP_GYRO_TOO_LARGE: n_knots_ = 0; panic("set-gyro-quaternions: resampled grid too large");
        default: n_knots_ = 0; panic("set-gyro-quaternions: non-finite sample rate. wtf?");
    }
    if (fs_ != r.fs || start_ != r.start) frames_dirty_ = true;
    fs_ = r.fs;
    start_ = r.start;
    n_knots_ = r.n_knots;
    knots_.clear();
    knots_cached_ = false; // on the device; fetched when somebody asks (knots())
    spline_dirty_ = false; // the device built the table
}

// core_private.cpp:142-190, on the device: order check, integer-microsecond grid, slerp, spline solve.
// Every device of the object does the (small) work itself rather than wait for a copy.
void SyncProblemHip::SetGyroQuaternions(const int64_t* ts, const double* quats, size_t count) {
    rs::RoctxRange roctx_range("rssync:SetGyroQuaternions(timestamped)");
    if (count < 2) panic("set-gyro-quaternions: need at least 2 samples");
    if (count > (size_t)UINT32_MAX) panic("set-gyro-quaternions: too many samples");
    rship_gyro_result r{};
    for (Shard& sh : shards_) hip_check(sh, rship_gyro_timestamped(sh.ctx, ts, quats, (uint32_t)count, &r), "gyro");
    accept_gyro(r);
}

const std::vector<double>& SyncProblemHip::knots() {
    if (!knots_cached_ && n_knots_) {
        knots_.resize(4 * n_knots_);
        hip_check(shards_[0], rship_gyro_knots(shards_[0].ctx, knots_.data(), (uint32_t)n_knots_), "gyro knots");
        knots_cached_ = true;
    }
    return knots_;
}

// a new record for `frame` in the staging arena (replaces any earlier one)
double* SyncProblemHip::stage_record(int64_t frame, uint64_t n_doubles, HostFrame& f) {
    double* dst = nullptr;
    f.raw_off = arena_.alloc(n_doubles ? n_doubles : 1, &dst);
    if (f.raw_off == UINT64_MAX) panic("set-track-result: out of host memory");
    (void)frame;
    return dst;
}

// Hand everything staged since the last call to the device, asynchronously: the DMA overlaps the
// caller's loop over frames (the reference driver tracks one frame pair at a time,
// core_testcode.cpp:135-158) and has usually finished by the time PreSync/Sync is called.
void SyncProblemHip::upload_new_records() {
    if (shards_.size() != 1) return; // several devices: who owns a frame is only known once all frames are (pack_frames)
    const uint64_t end = arena_.size();
    if (end <= uploaded_) return;
    arena_.ranges(uploaded_, end, [&](const double* host, uint64_t off, uint64_t cnt) {
        hip_check(shards_[0], rship_upload_raw(shards_[0].ctx, host, off, cnt), "upload tracks");
    });
    uploaded_ = end;
}

// core_private.cpp:192-203; the data is copied before returning
void SyncProblemHip::SetTrackResult(int64_t frame, const double* ts_a, const double* ts_b, const double* rays_a,
                                    const double* rays_b, size_t count) {
    if (!all_finite(rays_a, 3 * count)) panic("set-track-result: non-finite numbers in rays_a");
    if (!all_finite(rays_b, 3 * count)) panic("set-track-result: non-finite numbers in rays_b");
    if (!all_finite(ts_a, count)) panic("set-track-result: non-finite numbers in ts_a");
    if (!all_finite(ts_b, count)) panic("set-track-result: non-finite numbers in ts_b");
    if (count > (size_t)rship_max_tracks())
        panic("set-track-result: " + std::to_string(count) + " tracks in one frame; this build accepts at most " +
              std::to_string(rship_max_tracks()));
    HostFrame f;
    f.n = (uint32_t)count;
    double* rec = stage_record(frame, 8 * (uint64_t)count, f);
    std::memcpy(rec, ts_a, count * 8);
    std::memcpy(rec + count, ts_b, count * 8);
    std::memcpy(rec + 2 * count, rays_a, 3 * count * 8);
    std::memcpy(rec + 5 * count, rays_b, 3 * count * 8);
    if (count) {
        f.a_min = f.a_max = ts_a[0];
        f.b_min = f.b_max = ts_b[0];
        for (size_t i = 0; i < count; ++i) {
            f.a_min = std::min(f.a_min, ts_a[i]); f.a_max = std::max(f.a_max, ts_a[i]);
            f.b_min = std::min(f.b_min, ts_b[i]); f.b_max = std::max(f.b_max, ts_b[i]);
        }
        f.ts_min = std::min(f.a_min, f.b_min);
        f.ts_max = std::max(f.a_max, f.b_max);
    }
    frames_[frame] = f;
    frames_dirty_ = true;
    upload_new_records();
}

// The reference driver's per-frame step before SetTrackResult (core_testcode.cpp:135-158) with the
// arithmetic moved to the device: the host stages the pixels and keeps the range of the row times
// (:144-145, needed for the frame table); the packing kernel undistorts and normalises.
void SyncProblemHip::SetTrackPixels(int64_t frame, double time_a, double time_b, const double* px_a,
                                    const double* px_b, size_t count, const double lens[9], double image_rows) {
    if (!all_finite(px_a, 2 * count)) panic("set-track-pixels: non-finite numbers in points_a");
    if (!all_finite(px_b, 2 * count)) panic("set-track-pixels: non-finite numbers in points_b");
    if (!all_finite(lens, 9) || !std::isfinite(time_a) || !std::isfinite(time_b) || !std::isfinite(image_rows) ||
        image_rows == 0)
        panic("set-track-pixels: non-finite lens or frame parameters");
    if (count > (size_t)rship_max_tracks())
        panic("set-track-result: " + std::to_string(count) + " tracks in one frame; this build accepts at most " +
              std::to_string(rship_max_tracks()));
    HostFrame f;
    f.n = (uint32_t)count;
    f.from_pixels = true;
    f.time_a = time_a;
    f.time_b = time_b;
    f.rows = image_rows;
    std::copy(lens, lens + 9, f.lens);
    double* rec = stage_record(frame, 4 * (uint64_t)count, f);
    for (size_t i = 0; i < count; ++i) {
        rec[4 * i] = px_a[2 * i]; rec[4 * i + 1] = px_a[2 * i + 1];
        rec[4 * i + 2] = px_b[2 * i]; rec[4 * i + 3] = px_b[2 * i + 1];
        const double tsa = time_a + lens[0] * (px_a[2 * i + 1] / image_rows); // :144
        const double tsb = time_b + lens[0] * (px_b[2 * i + 1] / image_rows); // :145
        if (i == 0) { f.a_min = f.a_max = tsa; f.b_min = f.b_max = tsb; }
        f.a_min = std::min(f.a_min, tsa); f.a_max = std::max(f.a_max, tsa);
        f.b_min = std::min(f.b_min, tsb); f.b_max = std::max(f.b_max, tsb);
    }
    if (count) {
        f.ts_min = std::min(f.a_min, f.b_min);
        f.ts_max = std::max(f.a_max, f.b_max);
    }
    frames_[frame] = f;
    frames_dirty_ = true;
    upload_new_records();
}

// optdata_fill_gyro (core_testcode.cpp:36-52): q_0 = identity, q_i = normalise(dq_i * q_{i-1}) with
// dq_i the rotation by rate_i over (t_i - t_{i-1}) (quat.cpp:5-17), timestamps truncated to whole
// microseconds, then the timestamped setter -- all of it on the device (rship_gyro_rates_*).
// `orientation` is telemetry-parser's three-letter string (position = output axis, letter = input axis,
// upper case +, lower case -) or NULL.
static void parse_orientation(const char* orientation, int32_t axis[3], double sign[3]) {
    for (int c = 0; c < 3; ++c) { axis[c] = c; sign[c] = 1.0; }
    if (!orientation) return;
    if (std::strlen(orientation) != 3) panic("set-gyro-rates: orientation must have 3 letters");
    for (int c = 0; c < 3; ++c) {
        const char lo = (char)std::tolower((unsigned char)orientation[c]);
        if (lo < 'x' || lo > 'z') panic("set-gyro-rates: orientation letters are x, y, z");
        axis[c] = lo - 'x';
        sign[c] = orientation[c] == lo ? -1.0 : 1.0;
    }
}

void SyncProblemHip::upload_rates(const double* ts, const double* rates, size_t count) {
    if (count < 2) panic("set-gyro-rates: need at least 2 samples");
    if (count > (size_t)UINT32_MAX) panic("set-gyro-rates: too many samples");
    if (!std::isfinite(ts[0]) || !std::isfinite(ts[count - 1])) panic("set-gyro-rates: non-finite numbers");
    for (Shard& sh : shards_) hip_check(sh, rship_gyro_rates_upload(sh.ctx, ts, rates, (uint32_t)count), "gyro rates");
}

void SyncProblemHip::integrate_rates(const char* orientation) {
    int32_t axis[3];
    double sign[3];
    parse_orientation(orientation, axis, sign);
    rship_gyro_result r{};
    for (Shard& sh : shards_) hip_check(sh, rship_gyro_rates_integrate(sh.ctx, axis, sign, &r), "gyro integrate");
    accept_gyro(r);
}

void SyncProblemHip::SetGyroRates(const double* ts, const double* rates, size_t count, const char* orientation) {
    int32_t axis[3];
    double sign[3];
    parse_orientation(orientation, axis, sign); // complain about the string before anything is uploaded
    upload_rates(ts, rates, count);
    integrate_rates(orientation);
}

static const char* presync_panic(uint32_t flags);

// The orientation-guessing block of the reference driver (core_testcode.cpp:186-224): for each
// candidate IMU orientation re-integrate the rates, replace the gyro (tracks stay) and PreSync;
// the caller ranks the costs.  The rates are uploaded once; an orientation is a permutation of the rate
// axes inside the integration kernel.  The last orientation stays installed.
//
// Round 6: ONE pipeline instead of a blocking PreSync per orientation.  The reference's loop is sequential by construction;
// here nothing about orientation i + 1 depends on the host having seen orientation i's costs: the grid (rate, first
// sample, count) follows from the two end timestamps alone, so frames, selection, plan and candidates are the same for
// every orientation.  After the first orientation (whose gyro status is waited for: is there a table at all?) everything
// is ENQUEUED -- integrate, resample, spline, sweep, sums, per orientation -- into side-by-side result slots
// (rship_presync_batch_begin), then ONE wait, ONE copy and, with ranks, ONE exchange of the whole [orientations][candidates]
// matrix; the arg-min per orientation (the same lexicographic rule) comes after it.  Bits unchanged.  With 512 frames per
// GPU (the 8-GPU shard of BASELINE config 5: ~5 ms of kernels per orientation) the blocking form cost ~10 % of the sweep.
// An orientation whose sweep met near-static pairs (RSHIP_NEAR_STATIC: their fp64 form needs the host between two
// launches) is repeated on its own afterwards, the plain way.
void SyncProblemHip::orientation_sweep(const double* ts, const double* rates, size_t count,
                                       const std::vector<std::string>& orientations, double initial_delay,
                                       int64_t frame_begin, int64_t frame_end, double search_step,
                                       double search_radius, double* costs, double* delays) {
    rs::RoctxRange roctx_range("rssync:orientation_sweep");
    if (orientations.empty()) return;
    for (const std::string& o : orientations) {
        int32_t axis[3];
        double sign[3];
        parse_orientation(o.c_str(), axis, sign);
    }
    upload_rates(ts, rates, count);
    const size_t n_or = orientations.size();
    auto one_by_one = [&](size_t i) {
        integrate_rates(orientations[i].c_str());
        const std::pair<double, double> r = PreSync(initial_delay, frame_begin, frame_end, search_step, search_radius);
        costs[i] = r.first;
        delays[i] = r.second;
    };
    const char* env_pipeline = std::getenv("RSSYNC_SWEEP_PIPELINE");
    const bool no_pipeline = env_pipeline && env_pipeline[0] == '0';
    integrate_rates(orientations[0].c_str());
    ensure_device();
    select(frame_begin, frame_end);
    std::vector<double> cand; // the candidates are whatever this double loop yields (core_private.cpp:69-70), as in PreSync
    for (double delay = initial_delay - search_radius; delay < initial_delay + search_radius; delay += search_step) {
        cand.push_back(delay);
        if (cand.size() > 50000000) panic("pre-sync: more than 5e7 candidate delays");
    }
    if (cand.empty()) panic("pre-sync: empty candidate list");
    const size_t n = cand.size(), ns = sel_.size(), W = plan_windows_;
    const size_t slice = std::max<size_t>(64, (size_t)(256u << 20) / (8 * std::max<size_t>(ns, 1)));
    if (no_pipeline || n_or < 2 || n_or > 255 || n > slice || !ns || (uint64_t)n_or * n * (W + 1) > (1u << 28)) {
        // (one orientation, a candidate list that the sweep would slice, more orientations than there are status records,
        // or RSSYNC_SWEEP_PIPELINE=0 -- rounds 1-5's loop, kept for the A/B and the tests' comparison)
        const std::pair<double, double> r0 = PreSync(initial_delay, frame_begin, frame_end, search_step, search_radius);
        costs[0] = r0.first;
        delays[0] = r0.second;
        for (size_t i = 1; i < n_or; ++i) one_by_one(i);
        return;
    }
    std::vector<int32_t> kd(n), kd64(n);
    std::vector<float> fd(n);
    std::vector<double> fd64(n);
    for (size_t i = 0; i < n; ++i) {
        const DelaySplit sp = split_delay(cand[i], fs_);
        kd[i] = sp.kd; fd[i] = sp.fd;
        const DelaySplit64 sp64 = split_delay64(cand[i], fs_);
        kd64[i] = sp64.kd; fd64[i] = sp64.fd;
    }
    for (Shard& sh : shards_) hip_check(sh, rship_presync_batch_begin(sh.ctx, (uint32_t)n_or, (uint32_t)n), "presync batch");
    struct CloseBatch { // a panic between here and the collect (C mirror: an exception) must not leave the contexts in a batch
        SyncProblemHip* s;
        bool armed = true;
        ~CloseBatch() {
            if (armed)
                for (Shard& sh : s->shards_)
                    if (sh.ctx) (void)rship_presync_batch_begin(sh.ctx, 0, 0);
        }
    } close_batch{this};
    for (size_t i = 0; i < n_or; ++i) {
        if (i > 0) {
            int32_t axis[3];
            double sign[3];
            parse_orientation(orientations[i].c_str(), axis, sign);
            rship_gyro_result r{};
            for (Shard& sh : shards_)
                hip_check(sh, rship_gyro_rates_integrate_enqueue(sh.ctx, axis, sign, (uint32_t)i, &r), "gyro integrate");
            // (what the HOST can tell -- the grid -- is the first orientation's, which was accepted; the device's complaints
            // are read after the batch)
            if (r.status != RSHIP_GYRO_OK || r.fs != fs_ || r.start != start_ || r.n_knots != n_knots_)
                panic("orientation sweep: the gyro grid changed between two orientations of one rate stream");
        }
        for (Shard& sh : shards_)
            hip_check(sh, rship_presync_enqueue(sh.ctx, kd.data(), fd.data(), kd64.data(), fd64.data(), (uint32_t)n, 20 /* core_private.cpp:77 */,
                                                0u, seed, 0, 0),
                      "presync");
    }
    // ONE wait per device: all orientations' sums, [orientation][candidate] rows
    std::vector<double> all(n_or * n * W + 1, 0.0);
    std::vector<uint32_t> flags(n_or, 0u), fl_sh(n_or);
    combine(n_or * n, W, [&](Shard& sh, double* win, double* chunk) {
        hip_check(sh, rship_presync_batch_collect(sh.ctx, (uint32_t)n, win, chunk, fl_sh.data()), "presync batch");
        for (size_t i = 0; i < n_or; ++i) flags[i] |= fl_sh[i];
    }, all.data());
    close_batch.armed = false;
    // the device's gyro complaints, in the order the reference's loop would have met them
    std::vector<int32_t> gst(n_or, RSHIP_GYRO_OK);
    for (Shard& sh : shards_) {
        std::vector<int32_t> st(n_or, RSHIP_GYRO_OK);
        hip_check(sh, rship_gyro_batch_status(sh.ctx, (uint32_t)(n_or - 1), st.data() + 1), "gyro status");
        for (size_t i = 1; i < n_or; ++i)
            if (gst[i] == RSHIP_GYRO_OK) gst[i] = st[i];
    }
    // one exchange for the whole matrix; the flag bits ride along as small integers (five per orientation)
    std::vector<double> buf(all.begin(), all.begin() + n_or * n * W);
    for (size_t i = 0; i < n_or; ++i)
        for (int b = 0; b < 5; ++b) buf.push_back((double)((flags[i] >> b) & 1u));
    if (distributed()) reduce(buf.data(), buf.size());
    bool redone = false;
    for (size_t i = 0; i < n_or; ++i) {
        if (gst[i] != RSHIP_GYRO_OK) { // (the same complaint on every rank: the rates are replicated)
            rship_gyro_result r{};
            r.status = gst[i];
            accept_gyro(r);
        }
        const double* fl = buf.data() + n_or * n * W + 5 * i;
        const uint32_t fa = (fl[0] > 0 ? 1u : 0u) | (fl[1] > 0 ? 2u : 0u) | (fl[2] > 0 ? 4u : 0u) | (fl[3] > 0 ? 8u : 0u);
        if (const char* msg = presync_panic(fa)) panic(msg);
        if (fl[4] > 0) { // near-static pairs on some rank: this orientation once more, with their fp64 form (every rank agrees: the flag was summed)
            one_by_one(i);
            redone = true;
            continue;
        }
        const double* c = buf.data() + i * n * W; // (W == 1: select() makes one window)
        size_t best = 0; // *std::min_element over pair(cost, delay) (core_private.cpp:89)
        for (size_t k = 1; k < n; ++k)
            if (std::make_pair(c[k * W], cand[k]) < std::make_pair(c[best * W], cand[best])) best = k;
        costs[i] = c[best * W];
        delays[i] = cand[best];
    }
    if (redone) integrate_rates(orientations[n_or - 1].c_str()); // the last orientation stays installed
}

// the uniform route's table (the other routes leave it built)
void SyncProblemHip::build_spline() {
    const std::vector<double>& k = knots();
    for (Shard& sh : shards_) hip_check(sh, rship_gyro_uniform(sh.ctx, k.data(), (uint32_t)n_knots_, fs_), "gyro table");
    spline_dirty_ = false;
}

// Device layout of OptData::frame_data (core_private.hpp:21): frames in ascending id order; per
// ray pair the fp32 streams {ax,bx,ay,by} / {az,bz,ta,tb} (PreSync) and the fp64 streams
// {ax,bx} {ay,by} {az,bz} {ta,tb} (Sync).  ta/tb carry the spline parameter (ts - start) * fs
// (core_private.cpp:19-20 without the delay) relative to the frame's integer base knot, so fp32
// only ever holds a span of a few tens of knots.  The host's part is O(frames): the table (base
// knot and parameter range of every frame, from the time range noted at SetTrackResult) and the
// list of raw records; the streams themselves are written by the packing kernel.
void SyncProblemHip::pack_frames() {
    rs::RoctxRange roctx_range("rssync:pack_frames (upload + packing kernel)");
    const size_t nf = frames_.size();
    std::vector<rship_frame> table(nf);
    std::vector<rship_pack_frame> pack(nf);
    std::vector<uint64_t> cum(nf + 1, 0); // rays before frame i
    table_ids_.resize(nf);
    size_t s = 0;
    for (auto& [id, f] : frames_) {
        const size_t n = f.n;
        // (ts - start) * fs is monotonic in ts: its range comes from the range of ts
        const double xmin = (f.ts_min - start_) * fs_, xmax = (f.ts_max - start_) * fs_;
        double base = n ? std::floor(xmin) : 0.0;
        if (!(base > -(double)kKnotClamp)) base = -(double)kKnotClamp;
        if (base > (double)kKnotClamp) base = (double)kKnotClamp;
        rship_frame rec{};
        rec.n_rays = (uint32_t)n;
        rec.base_knot = (int32_t)base;
        rec.id = id;
        rec.tmin64 = n ? xmin - base : 0.0;
        rec.tmax64 = n ? xmax - base : 0.0;
        rec.tmin = (float)rec.tmin64;
        rec.tmax = (float)rec.tmax64;
        if (f.from_pixels) {
            // the device recomputes the row times from the pixels with the same fp64 operations; one
            // ulp of slack on the bounds costs nothing and makes the window independent of that
            rec.tmin = std::nextafterf(rec.tmin, -std::numeric_limits<float>::infinity());
            rec.tmax = std::nextafterf(rec.tmax, std::numeric_limits<float>::infinity());
            rec.tmin64 = std::nextafter(rec.tmin64, -std::numeric_limits<double>::infinity());
            rec.tmax64 = std::nextafter(rec.tmax64, std::numeric_limits<double>::infinity());
        }
        // the knots each END of the pair touches at delay 0, relative to base_knot (rship_frame::range_a / range_b):
        // floors of the end's smallest / largest offset, taken both on the fp64 value and on its fp32 rounding (the
        // fp32 stream's offset may round up across a knot), with the same one-ulp slack for pixel frames
        rec.range_a = rec.range_b = RSHIP_NO_SPLIT;
        if (n) {
            auto range = [&](double tlo, double thi) -> uint32_t {
                double lo64 = (tlo - start_) * fs_ - base, hi64 = (thi - start_) * fs_ - base;
                float lo32 = (float)lo64, hi32 = (float)hi64;
                if (f.from_pixels) {
                    lo64 = std::nextafter(lo64, -std::numeric_limits<double>::infinity());
                    hi64 = std::nextafter(hi64, std::numeric_limits<double>::infinity());
                    lo32 = std::nextafterf(lo32, -std::numeric_limits<float>::infinity());
                    hi32 = std::nextafterf(hi32, std::numeric_limits<float>::infinity());
                }
                const double lo = std::min(std::floor(lo64), (double)std::floor(lo32));
                const double hi = std::max(std::floor(hi64), (double)std::floor(hi32));
                if (!(lo >= 0.0) || !(hi >= lo) || !(hi <= 65000.0)) return RSHIP_NO_SPLIT;
                return (uint32_t)lo | ((uint32_t)hi << 16);
            };
            const uint32_t ra = range(f.a_min, f.a_max), rb = range(f.b_min, f.b_max);
            if (ra != RSHIP_NO_SPLIT && rb != RSHIP_NO_SPLIT) { rec.range_a = ra; rec.range_b = rb; }
        }
        rship_pack_frame pf{};
        pf.raw_offset = f.raw_off;
        pf.n_rays = rec.n_rays;
        pf.base = base;
        pf.is_pixels = f.from_pixels ? 1u : 0u;
        pf.time_a = f.time_a; pf.time_b = f.time_b; pf.rows = f.rows;
        std::copy(f.lens, f.lens + 9, pf.lens);
        table[s] = rec;
        pack[s] = pf;
        table_ids_[s] = id;
        cum[s + 1] = cum[s] + n;
        ++s;
    }
    // Contiguous blocks of the sorted frame list per device, balanced by ray count, cut at multiples of
    // kChunk frames (the sums over frames are defined on those blocks: DESIGN.md "Sums").
    const size_t S = shards_.size();
    std::vector<uint32_t> cut(S + 1, 0);
    cut[S] = (uint32_t)nf;
    for (size_t d = 1; d < S; ++d) {
        const double target = (double)cum[nf] * (double)d / (double)S;
        uint32_t best = cut[d - 1];
        double best_err = std::numeric_limits<double>::infinity();
        for (uint32_t i = cut[d - 1]; i <= nf; i += kChunk - (i % kChunk)) {
            const double err = std::fabs((double)cum[i] - target);
            if (err < best_err) { best_err = err; best = i; }
            if ((double)cum[i] > target) break;
        }
        cut[d] = best;
    }
    if (S == 1) upload_new_records();
    uint32_t bad = 0;
    for (size_t d = 0; d < S; ++d) {
        Shard& sh = shards_[d];
        sh.t0 = cut[d];
        sh.t1 = cut[d + 1];
        sh.sel.clear();
        sh.slot_global.clear();
        const uint32_t n = sh.t1 - sh.t0;
        if ((cum[sh.t1] - cum[sh.t0]) > 0xffffffffull) panic("sync: more than 2^32 rays on one device");
        sh.raw_lo = 0;
        if (S > 1) {
            // this device's records, as few copies as the arena layout allows
            std::vector<std::pair<uint64_t, uint64_t>> rng;
            for (uint32_t i = sh.t0; i < sh.t1; ++i) {
                const uint64_t len = (uint64_t)pack[i].n_rays * (pack[i].is_pixels ? 4 : 8);
                if (len) rng.emplace_back(pack[i].raw_offset, pack[i].raw_offset + len);
            }
            std::sort(rng.begin(), rng.end());
            sh.raw_lo = rng.empty() ? 0 : rng.front().first;
            for (size_t i = 0; i < rng.size();) {
                uint64_t lo = rng[i].first, hi = rng[i].second;
                size_t j = i + 1;
                while (j < rng.size() && rng[j].first <= hi) { hi = std::max(hi, rng[j].second); ++j; }
                arena_.ranges(lo, hi, [&](const double* host, uint64_t off, uint64_t cnt) {
                    hip_check(sh, rship_upload_raw(sh.ctx, host, off - sh.raw_lo, cnt), "upload tracks");
                });
                i = j;
            }
        }
        for (uint32_t i = sh.t0; i < sh.t1; ++i) {
            table[i].ray_offset = pack[i].ray_offset = (uint32_t)(cum[i] - cum[sh.t0]);
            pack[i].raw_offset -= sh.raw_lo;
        }
        uint32_t b = 0;
        hip_check(sh, rship_pack_frames(sh.ctx, table.data() + sh.t0, pack.data() + sh.t0, n, cum[sh.t1] - cum[sh.t0], start_,
                                        fs_, &b),
                  "pack frames");
        bad += b;
    }
    if (bad) panic("set-track-result: non-finite numbers in rays (" + std::to_string(bad) + " tracks; lens parameters?)");
    // several devices: every shard plans its LDS spline windows from the frames of the WHOLE problem, as the single
    // device would (window_plan.hpp: plan_window_frames), so that a frame takes the same spline path wherever it lives
    if (S > 1)
        for (Shard& sh : shards_) hip_check(sh, rship_set_problem_frames(sh.ctx, table.data(), (uint32_t)nf), "problem frames");
    sel_.clear();
    window_key_.clear();
    frames_dirty_ = false;
}

// Hand a selection to the devices.  `slots` are frame-table indices in slot order (window-major,
// ascending within a window); grp_off has the window offsets of a grouped selection (batched Sync:
// per-window delays and per-slot state) or is empty.  The plan of the sums is, by default, one window
// per group; plan_pos / plan_off give other windows as lists of slot positions (batched PreSync:
// windows may overlap).  Every shard gets its part, in the same order, and its own plan; chunk
// boundaries are multiples of kChunk in the frame table, which shard boundaries are as well.
void SyncProblemHip::apply_selection(const std::vector<uint32_t>& slots, const std::vector<uint32_t>& grp_off,
                                     const std::vector<uint32_t>* plan_pos, const std::vector<uint32_t>* plan_off) {
    window_key_.clear(); // whatever select_windows left on the devices is replaced
    const bool grouped = !grp_off.empty();
    const size_t n_grp = grouped ? grp_off.size() - 1 : 1;
    std::vector<uint32_t> def_off;
    if (!grouped) def_off = {0u, (uint32_t)slots.size()};
    const std::vector<uint32_t>& goff = grouped ? grp_off : def_off;
    const size_t n_win = plan_off ? plan_off->size() - 1 : n_grp;
    plan_windows_ = n_win;
    for (Shard& sh : shards_) {
        sh.sel.clear();
        sh.slot_global.clear();
        std::vector<uint32_t> loff(n_grp + 1, 0), local_of(slots.size(), 0xffffffffu);
        for (size_t w = 0; w < n_grp; ++w) {
            for (uint32_t j = goff[w]; j < goff[w + 1]; ++j) {
                if (slots[j] < sh.t0 || slots[j] >= sh.t1) continue;
                local_of[j] = (uint32_t)sh.sel.size();
                sh.sel.push_back(slots[j] - sh.t0);
                sh.slot_global.push_back(j);
            }
            loff[w + 1] = (uint32_t)sh.sel.size();
        }
        hip_check(sh, rship_select_slots(sh.ctx, sh.sel.data(), (uint32_t)sh.sel.size(), grouped ? loff.data() : nullptr,
                                         (uint32_t)n_grp),
                  "select slots");
        // plan: windows -> chunks (same block of kChunk table indices) -> local slot positions
        std::vector<uint32_t> pidx, coff{0u};
        sh.win_chunk_off.assign(1, 0u);
        for (size_t w = 0; w < n_win; ++w) {
            const uint32_t j0 = plan_off ? (*plan_off)[w] : goff[w], j1 = plan_off ? (*plan_off)[w + 1] : goff[w + 1];
            uint32_t cur_block = 0xffffffffu;
            for (uint32_t j = j0; j < j1; ++j) {
                const uint32_t pos = plan_pos ? (*plan_pos)[j] : j; // position in `slots`
                if (local_of[pos] == 0xffffffffu) continue;
                const uint32_t block = slots[pos] / kChunk;
                if (block != cur_block) {
                    if (cur_block != 0xffffffffu) coff.push_back((uint32_t)pidx.size());
                    cur_block = block;
                }
                pidx.push_back(local_of[pos]);
            }
            if (cur_block != 0xffffffffu) coff.push_back((uint32_t)pidx.size());
            sh.win_chunk_off.push_back((uint32_t)coff.size() - 1);
        }
        sh.n_chunks = (uint32_t)coff.size() - 1;
        bool identity = pidx.size() == sh.sel.size();
        for (size_t j = 0; j < pidx.size() && identity; ++j) identity = pidx[j] == j;
        hip_check(sh, rship_set_plan(sh.ctx, identity ? nullptr : pidx.data(), (uint32_t)pidx.size(), coff.data(), sh.n_chunks,
                                     sh.win_chunk_off.data(), (uint32_t)n_win),
                  "set plan");
    }
}

// Sums of every shard's last enqueue, rows x windows.  One device: its window sums.  Several: every
// window is the sequential sum of its chunks, devices in frame order -- the association the single
// device uses (plan_sum_kernel), so the totals do not depend on the number of devices.
template <typename Collect>
void SyncProblemHip::combine(size_t rows, size_t n_win, Collect&& collect, double* out) {
    if (shards_.size() == 1) {
        collect(shards_[0], out, (double*)nullptr);
        return;
    }
    std::fill(out, out + rows * n_win, 0.0);
    std::vector<double> chunk;
    for (Shard& sh : shards_) {
        chunk.assign(rows * (size_t)sh.n_chunks + 1, 0.0);
        collect(sh, (double*)nullptr, chunk.data());
        for (size_t r = 0; r < rows; ++r)
            for (size_t w = 0; w < n_win; ++w) {
                double acc = out[r * n_win + w];
                for (uint32_t c = sh.win_chunk_off[w]; c < sh.win_chunk_off[w + 1]; ++c) acc += chunk[r * sh.n_chunks + c];
                out[r * n_win + w] = acc;
            }
    }
}

void SyncProblemHip::ensure_device() {
    if (n_knots_ < 2) panic("sync: gyro data was not set");
    if (spline_dirty_) build_spline();
    if (frames_dirty_) pack_frames();
}

uint32_t SyncProblemHip::get_motion(double* M, double* k, uint32_t cap) {
    std::vector<double> lm, lk;
    for (Shard& sh : shards_) {
        const size_t nl = sh.sel.size();
        lm.assign(nl * 3 + 3, 0.0);
        lk.assign(nl + 1, 0.0);
        uint32_t got = 0;
        if (nl) hip_check(sh, rship_get_motion(sh.ctx, lm.data(), lk.data(), (uint32_t)nl, &got), "get motion");
        for (size_t j = 0; j < got; ++j) {
            const uint32_t gslot = sh.slot_global[j];
            if (gslot >= cap) continue;
            M[3 * gslot] = lm[3 * j]; M[3 * gslot + 1] = lm[3 * j + 1]; M[3 * gslot + 2] = lm[3 * j + 2];
            k[gslot] = lk[j];
        }
    }
    return (uint32_t)std::min<size_t>(sel_.size(), cap);
}

void SyncProblemHip::set_motion(const double* M, const double* k, uint32_t n) {
    if (n != sel_.size()) panic("set_motion: count differs from the selection");
    std::vector<double> lm, lk;
    for (Shard& sh : shards_) {
        const size_t nl = sh.sel.size();
        if (!nl) continue;
        lm.resize(nl * 3);
        lk.resize(nl);
        for (size_t j = 0; j < nl; ++j) {
            const uint32_t gslot = sh.slot_global[j];
            lm[3 * j] = M[3 * gslot]; lm[3 * j + 1] = M[3 * gslot + 1]; lm[3 * j + 2] = M[3 * gslot + 2];
            lk[j] = k[gslot];
        }
        hip_check(sh, rship_set_motion(sh.ctx, lm.data(), lk.data(), (uint32_t)nl), "set motion");
    }
}

// one frame's residual matrix as the PreSync kernel (fp32) or the Sync kernels (fp64) compute it
void SyncProblemHip::debug_problem(int64_t frame, double delay, float* P, float* dP, double* P64, double* dP64, size_t cap_rows) {
    select(frame, frame + 1);
    for (Shard& sh : shards_) {
        if (sh.sel.empty()) continue;
        if (P64) {
            DelaySplit64 ds = split_delay64(delay, fs_);
            hip_check(sh, rship_debug_problem64(sh.ctx, 0, ds.kd, ds.fd, P64, dP64, (uint32_t)cap_rows), "debug problem");
        } else {
            DelaySplit ds = split_delay(delay, fs_);
            hip_check(sh, rship_debug_problem(sh.ctx, 0, ds.kd, ds.fd, P, dP, (uint32_t)cap_rows), "debug problem");
        }
    }
}

void SyncProblemHip::debug_rays(int64_t frame, float* a4, float* b4, size_t cap) {
    for (uint32_t i = 0; i < table_ids_.size(); ++i) {
        if (table_ids_[i] != frame) continue;
        for (Shard& sh : shards_)
            if (i >= sh.t0 && i < sh.t1) hip_check(sh, rship_debug_rays(sh.ctx, i - sh.t0, a4, b4, (uint32_t)cap), "debug rays");
        return;
    }
    panic("frame_rays: no such frame");
}

// frame filters of core_private.cpp:65-68 / :218-219 / :340-343 on the sorted table
uint32_t SyncProblemHip::select(int64_t begin, int64_t end_exclusive) {
    sel_.clear();
    window_key_.clear();
    for (uint32_t i = 0; i < table_ids_.size(); ++i)
        if (table_ids_[i] >= begin && table_ids_[i] < end_exclusive) sel_.push_back(i);
    for (uint32_t i : sel_)
        if (frames_.at(table_ids_[i]).n < 2)
            panic("sync: frame " + std::to_string(table_ids_[i]) + " has fewer than 2 tracks");
    n_windows_ = 1;
    apply_selection(sel_, {}, nullptr, nullptr);
    return (uint32_t)sel_.size();
}

static const char* presync_panic(uint32_t flags) { // core_private.cpp:76-83, in the reference's order
    if (flags & RSHIP_BAD_P) return "pre-sync: non-finite numbers in P";
    if (flags & RSHIP_BAD_M) return "pre-sync: non-finite numbers in M";
    if (flags & RSHIP_BAD_R) return "pre-sync: non-finite r";
    if (flags & RSHIP_BAD_RHO) return "pre-sync: non-finite rho";
    return nullptr;
}

// costs[candidate][window] of a list of candidate delays under the current selection and plan, summed over
// this object's devices (not yet over ranks); per-slot debug matrices in global slot order if asked for
void SyncProblemHip::sweep_windows(const std::vector<double>& delays, uint32_t stream_base, double* out, uint32_t* flags_out,
                                   double* frame_costs, int32_t* best_h) {
    const size_t n = delays.size(), W = plan_windows_, ns = sel_.size();
    std::fill(out, out + n * W, 0.0);
    uint32_t flags = 0;
    if (ns && n) {
        std::vector<int32_t> kd(n), kd64(n);
        std::vector<float> fd(n);
        std::vector<double> fd64(n);
        for (size_t i = 0; i < n; ++i) {
            DelaySplit sp = split_delay(delays[i], fs_);
            kd[i] = sp.kd;
            fd[i] = sp.fd;
            // ... and in full precision, for the pairs whose rows the sweep takes from the fp64 streams (near-static frames:
            // the reference's arithmetic is double throughout, core_private.cpp:19-28,45-46)
            DelaySplit64 sp64 = split_delay64(delays[i], fs_);
            kd64[i] = sp64.kd;
            fd64[i] = sp64.fd;
        }
        // the devices keep a [candidates][frames] fp64 matrix: sweep very long candidate lists in
        // slices (the sampler stream is the global candidate index, so slicing changes nothing)
        const size_t slice = std::max<size_t>(64, (size_t)(256u << 20) / (8 * ns));
        std::vector<double> fc;
        std::vector<int32_t> bh;
        for (size_t b = 0; b < n; b += slice) {
            const size_t m = std::min(slice, n - b);
            for (Shard& sh : shards_) // every device starts its part ...
                hip_check(sh, rship_presync_enqueue(sh.ctx, kd.data() + b, fd.data() + b, kd64.data() + b, fd64.data() + b, (uint32_t)m, 20 /* core_private.cpp:77 */,
                                                    stream_base + (uint32_t)b, seed, frame_costs != nullptr, best_h != nullptr),
                          "presync");
            combine(m, W, [&](Shard& sh, double* win, double* chunk) { // ... and is waited for in turn
                uint32_t fl = 0;
                const size_t nl = sh.sel.size();
                if (frame_costs) fc.assign(m * nl + 1, 0.0);
                if (best_h) bh.assign(m * nl + 1, 0);
                hip_check(sh, rship_presync_collect(sh.ctx, (uint32_t)m, win, chunk, &fl, frame_costs ? fc.data() : nullptr,
                                                    best_h ? bh.data() : nullptr),
                          "presync");
                flags |= fl;
                for (size_t r = 0; r < m && (frame_costs || best_h); ++r)
                    for (size_t j = 0; j < nl; ++j) {
                        if (frame_costs) frame_costs[(b + r) * ns + sh.slot_global[j]] = fc[r * nl + j];
                        if (best_h) best_h[(b + r) * ns + sh.slot_global[j]] = bh[r * nl + j];
                    }
            }, out + b * W);
        }
    }
    *flags_out = flags;
}

// costs of a list of candidate delays on the current (single-window) selection, summed over ranks
std::vector<double> SyncProblemHip::sweep(const std::vector<double>& delays, uint32_t stream_base, bool panics,
                                          double* frame_costs, int32_t* best_h) {
    const size_t n = delays.size();
    std::vector<double> costs(n + 1, 0.0);
    uint32_t flags = 0;
    sweep_windows(delays, stream_base, costs.data(), &flags, frame_costs, best_h);
    // one exchange for the whole sweep; the flag bits ride along as small integers
    double fl[4] = {(double)((flags >> 0) & 1), (double)((flags >> 1) & 1), (double)((flags >> 2) & 1),
                    (double)((flags >> 3) & 1)};
    if (distributed()) {
        std::vector<double> buf(costs.begin(), costs.begin() + n);
        buf.insert(buf.end(), fl, fl + 4);
        reduce(buf.data(), buf.size());
        std::copy(buf.begin(), buf.begin() + n, costs.begin());
        std::copy(buf.begin() + n, buf.end(), fl);
    }
    if (panics) {
        uint32_t all = (fl[0] > 0 ? 1u : 0u) | (fl[1] > 0 ? 2u : 0u) | (fl[2] > 0 ? 4u : 0u) | (fl[3] > 0 ? 8u : 0u);
        if (const char* msg = presync_panic(all)) panic(msg);
    }
    costs.resize(n);
    return costs;
}

// core_private.cpp:205-209 -> :61-90
std::pair<double, double> SyncProblemHip::PreSync(double initial_delay, int64_t frame_begin, int64_t frame_end,
                                                  double search_step, double search_radius) {
    rs::RoctxRange roctx_range("rssync:PreSync");
    ensure_device();
    select(frame_begin, frame_end);
    std::vector<double> delays; // the candidates are whatever this double loop yields (:69-70)
    for (double delay = initial_delay - search_radius; delay < initial_delay + search_radius; delay += search_step) {
        delays.push_back(delay);
        if (delays.size() > 50000000) panic("pre-sync: more than 5e7 candidate delays");
    }
    if (delays.empty()) panic("pre-sync: empty candidate list");
    std::vector<double> costs = sweep(delays, 0, true, nullptr, nullptr);
    size_t best = 0; // *std::min_element over pair(cost, delay) (:89)
    for (size_t i = 1; i < costs.size(); ++i)
        if (std::make_pair(costs[i], delays[i]) < std::make_pair(costs[best], delays[best])) best = i;
    return {costs[best], delays[best]};
}

// core_private.cpp:336-361
void SyncProblemHip::DebugPreSync(double initial_delay, int64_t frame_begin, int64_t frame_end, double search_radius,
                                  double* delays, double* costs, int point_count) {
    rs::RoctxRange roctx_range("rssync:DebugPreSync");
    ensure_device();
    select(frame_begin, frame_end);
    std::vector<double> d((size_t)std::max(point_count, 0));
    for (int i = 0; i < point_count; ++i)
        d[i] = initial_delay - search_radius + 2 * search_radius * i / (point_count - 1); // :345
    std::vector<double> c = sweep(d, kStreamDebug, false, nullptr, nullptr);
    for (int i = 0; i < point_count; ++i) {
        delays[i] = d[i];
        costs[i] = c[i];
    }
}

static void split_all64(const std::vector<double>& delays, double fs, std::vector<int32_t>& kd, std::vector<double>& fd) {
    kd.resize(delays.size());
    fd.resize(delays.size());
    for (size_t i = 0; i < delays.size(); ++i) {
        if (delays[i] != delays[i]) {
            kd[i] = 0;
            fd[i] = std::numeric_limits<double>::quiet_NaN();
        } else {
            DelaySplit64 s = split_delay64(delays[i], fs);
            kd[i] = s.kd;
            fd[i] = s.fd;
        }
    }
}

// per-window delays -> device representation; NaN delay = window switched off
static void split_all(const std::vector<double>& delays, double fs, std::vector<int32_t>& kd, std::vector<float>& fd) {
    kd.resize(delays.size());
    fd.resize(delays.size());
    for (size_t i = 0; i < delays.size(); ++i) {
        if (delays[i] != delays[i]) {
            kd[i] = 0;
            fd[i] = std::numeric_limits<float>::quiet_NaN();
        } else {
            DelaySplit s = split_delay(delays[i], fs);
            kd[i] = s.kd;
            fd[i] = s.fd;
        }
    }
}

void SyncProblemHip::init_motion(const std::vector<double>& delays, uint32_t call_stride) {
    std::vector<int32_t> kd, kd64;
    std::vector<float> fd;
    std::vector<double> fd64; // (the same delays in full precision: a near-static frame's search takes its rows from the fp64 streams)
    split_all(delays, fs_, kd, fd);
    split_all64(delays, fs_, kd64, fd64);
    for (double& f : fd64)
        if (f != f) f = 0.0; // (split_all64 marks a window that is switched off with NaN; the search has no such windows)
    for (Shard& sh : shards_)
        if (!sh.sel.empty())
            hip_check(sh, rship_init_motion(sh.ctx, kd.data(), fd.data(), kd64.data(), fd64.data(), 200 /* core_private.cpp:127 */,
                                            kStreamSyncInit + sync_calls, call_stride, seed),
                      "init motion");
    if (record_init || !init_override.empty()) exchange_init_winners();
}

void SyncProblemHip::exchange_init_winners() {
    const size_t ns = sel_.size();
    const bool set = !init_override.empty();
    if (set && init_override.size() != ns)
        panic("init-override: " + std::to_string(init_override.size()) + " winners for " + std::to_string(ns) + " slots");
    if (record_init) last_init_winners.assign(ns, std::numeric_limits<int32_t>::min());
    std::vector<int32_t> got, put;
    for (Shard& sh : shards_) {
        const size_t nl = sh.sel.size();
        if (!nl) continue;
        got.assign(nl, 0);
        put.assign(nl, 0);
        for (size_t j = 0; j < nl && set; ++j) put[j] = init_override[sh.slot_global[j]];
        hip_check(sh, rship_debug_init_h(sh.ctx, record_init ? got.data() : nullptr, set ? put.data() : nullptr, (uint32_t)nl),
                  "init winners");
        for (size_t j = 0; j < nl && record_init; ++j) last_init_winners[sh.slot_global[j]] = set ? put[j] : got[j];
    }
    init_override.clear(); // for one call only
}

void SyncProblemHip::opt_motion(const std::vector<double>& delays, uint64_t* stats) {
    std::vector<int32_t> kd;
    std::vector<double> fd;
    split_all64(delays, fs_, kd, fd);
    if (stats) stats[0] = stats[1] = stats[2] = 0;
    for (Shard& sh : shards_) {
        if (sh.sel.empty()) continue;
        uint64_t st[3] = {0, 0, 0};
        hip_check(sh, rship_opt_motion(sh.ctx, kd.data(), fd.data(), stats ? st : nullptr), "opt motion");
        if (stats) { stats[0] += st[0]; stats[1] += st[1]; stats[2] += st[2]; }
    }
}

// M and k of GuessMotion/GuessK in fp64 from the winners of init_motion (same delays), without optimising
void SyncProblemHip::finish_init(const std::vector<double>& delays) {
    std::vector<int32_t> kd;
    std::vector<double> fd;
    split_all64(delays, fs_, kd, fd);
    for (Shard& sh : shards_)
        if (!sh.sel.empty()) hip_check(sh, rship_finish_init(sh.ctx, kd.data(), fd.data()), "finish init");
}

void SyncProblemHip::init_k_simple(const std::vector<double>& delays) {
    std::vector<int32_t> kd;
    std::vector<double> fd;
    split_all64(delays, fs_, kd, fd);
    for (Shard& sh : shards_)
        if (!sh.sel.empty()) hip_check(sh, rship_init_k_simple(sh.ctx, kd.data(), fd.data()), "init k (simplified)");
}

// per window: sum over its slots (over this object's devices and over ranks) of FrameState::Loss; delays is
// [n_delays][n_windows] row-major (NaN = skip that window), outputs likewise
void SyncProblemHip::loss(const std::vector<double>& delays, std::vector<double>& out_loss,
                          std::vector<double>* out_grad, bool simplified) {
    const size_t n = delays.size(), W = n_windows_, nd = n / W;
    std::vector<double> buf(2 * n, 0.0);
    if (!sel_.empty() && n) {
        std::vector<int32_t> kd;
        std::vector<double> fd;
        split_all64(delays, fs_, kd, fd);
        for (Shard& sh : shards_)
            hip_check(sh, rship_loss_enqueue(sh.ctx, kd.data(), fd.data(), (uint32_t)nd, out_grad != nullptr,
                                             simplified ? RSHIP_LOSS_SIMPLIFIED : 0u),
                      "loss");
        // rows: the nd losses, then (with a gradient) the nd derivatives
        const size_t rows = out_grad ? 2 * nd : nd;
        std::vector<double> win(rows * W), cw, cg;
        combine(rows, W, [&](Shard& sh, double* w_out, double* c_out) {
            if (w_out) {
                hip_check(sh, rship_loss_collect(sh.ctx, (uint32_t)nd, w_out, out_grad ? w_out + nd * W : nullptr, nullptr, nullptr),
                          "loss");
            } else {
                hip_check(sh, rship_loss_collect(sh.ctx, (uint32_t)nd, nullptr, nullptr, c_out,
                                                 out_grad ? c_out + nd * (size_t)sh.n_chunks : nullptr),
                          "loss");
            }
        }, win.data());
        std::copy(win.begin(), win.begin() + n, buf.begin());
        if (out_grad) std::copy(win.begin() + n, win.end(), buf.begin() + n);
    }
    reduce(buf.data(), out_grad ? 2 * n : n);
    out_loss.assign(buf.begin(), buf.begin() + n);
    if (out_grad) out_grad->assign(buf.begin() + n, buf.end());
}

// selection for batched Sync: window w = frames with begin[w] <= id <= end[w] (end inclusive,
// core_private.cpp:219); a frame covered by several windows gets one slot in each
void SyncProblemHip::select_windows(const std::vector<int64_t>& begins, const std::vector<int64_t>& ends_incl) {
    // the four Sync calls of a sync point select the same windows: what the devices hold is still right
    // (GuessMotion rewrites every slot's state before anything reads it)
    std::vector<int64_t> key(begins);
    key.insert(key.end(), ends_incl.begin(), ends_incl.end());
    if (!window_key_.empty() && key == window_key_) return;
    sel_.clear();
    std::vector<uint32_t> off(begins.size() + 1, 0);
    for (size_t w = 0; w < begins.size(); ++w) {
        for (uint32_t i = 0; i < table_ids_.size(); ++i)
            if (table_ids_[i] >= begins[w] && table_ids_[i] <= ends_incl[w]) sel_.push_back(i);
        off[w + 1] = (uint32_t)sel_.size();
    }
    for (uint32_t i : sel_)
        if (frames_.at(table_ids_[i]).n < 2)
            panic("sync: frame " + std::to_string(table_ids_[i]) + " has fewer than 2 tracks");
    n_windows_ = std::max<size_t>(1, begins.size());
    if (begins.empty()) off.assign(2, 0u);
    apply_selection(sel_, off, nullptr, nullptr);
    window_key_ = key; // apply_selection cleared it
}

// core_private.cpp:211-334 for W independent windows advanced in lock-step (W = 1 is
// ISyncProblem::Sync).  Every window keeps its own delay, momentum and convergence counter and
// stops on its own; what is batched is the device work: one motion-optimisation launch, one
// loss+gradient launch and one 10-trial line-search launch per outer iteration for all windows.
// Window w is bit-for-bit what the w-th of W consecutive Sync calls would return.
// Differences from the reference's schedule, none of which changes a value the reference
// would compute differently:
//  * d(loss)/d(delay) is the analytic derivative, not the +-1e-6 s central difference
//    (:96-97,112); they agree to ~1e-9 relative (tests/test_oracle_math.py);
//  * the <= 10 backtracking trials (backtrack.cpp:7-11) are evaluated in batched launches of
//    five and the first that satisfies the Armijo test is taken, which is what the
//    sequential loop returns;
//  * P is computed once per motion optimisation, not three times per evaluation (:94-97).
// Can the window executor (kernels/executor.hpp) run the current selection?  One device holding every frame, nobody
// else in the sums, frames of up to 512 tracks, no empty window.
bool SyncProblemHip::executor_ok(bool simplified) {
    if (!use_executor || simplified || shards_.size() != 1 || distributed() || host_loop || max_outer <= 0 || sel_.empty()) return false;
    if (!rship_has_device_loop() || !rship_exec_supported(shards_[0].ctx)) return false;
    if (record_init || !init_override.empty()) return false; // (the diagnostics hook into the separate search launch)
    const Shard& sh = shards_[0];
    if (sh.win_chunk_off.size() != n_windows_ + 1) return false;
    for (size_t w = 0; w < n_windows_; ++w)
        if (sh.win_chunk_off[w + 1] == sh.win_chunk_off[w]) return false; // a window without frames
    return true;
}

// `repeats` chained Sync calls on the selected windows in one launch; fills traces (all calls' rows per window).
// false: the executor gave up (its watchdog, an allocation) -- nothing the caller hands out has been touched, the
// chain of launches takes over from the same inputs and returns the same values.
bool SyncProblemHip::sync_exec(const std::vector<int64_t>& begins, const std::vector<int64_t>& ends_incl,
                               const std::vector<double>& initial, double search_center, double search_radius, int repeats,
                               uint32_t stream_first, uint32_t stream_stride, std::vector<double>& costs,
                               std::vector<double>& delays_out) {
    (void)begins; (void)ends_incl;
    const size_t W = n_windows_;
    Shard& sh = shards_[0];
    const uint32_t rows = (uint32_t)repeats * (uint32_t)max_outer;
    std::vector<double> tr((size_t)W * rows * 6, 0.0);
    std::vector<int32_t> its(W * (size_t)repeats, 0);
    std::vector<double> c_out(W, 0.0), d_out(W, 0.0);
    if (executor_test_fail_ ||
        rship_sync_exec(sh.ctx, initial.data(), repeats, stream_first, stream_stride, seed, max_outer, search_center,
                        search_radius, d_out.data(), c_out.data(), its.data(), tr.data(), rows)) {
        if (!executor_warned_) {
            std::cerr << "rssync: window executor: " << (executor_test_fail_ ? "switched off by RSSYNC_EXECUTOR_FAIL" : rship_last_error(sh.ctx))
                      << " -- continuing with the launch chain" << std::endl;
            executor_warned_ = true;
        }
        use_executor = false;
        return false;
    }
    costs = c_out;
    delays_out = d_out;
    traces.assign(W, {});
    for (size_t w = 0; w < W; ++w) {
        size_t n = 0;
        for (int r = 0; r < repeats; ++r) n += (size_t)its[w * repeats + r];
        traces[w].assign(tr.begin() + w * (size_t)rows * 6, tr.begin() + (w * (size_t)rows + n) * 6);
    }
    executor_runs += 1;
    return true;
}

// is this executor call one that gets verified?  (the check mode: every call; otherwise one in executor_check_every,
// counted over all objects of the process -- a service that syncs one clip per object still samples)
bool SyncProblemHip::check_this_call() {
    if (executor_check) return true;
    if (!executor_check_every) return false;
    static std::atomic<uint64_t> calls{0};
    return (calls.fetch_add(1, std::memory_order_relaxed) + 1) % executor_check_every == 0;
}

// RSSYNC_EXECUTOR_CHECK: the executor's results against the launch chain's for the same call (`traces` holds the chain's)
bool SyncProblemHip::check_against_chain(const char* what, const std::vector<double>& c_exec, const std::vector<double>& d_exec,
                                         const std::vector<std::vector<double>>& tr_exec, const std::vector<double>& c_chain,
                                         const std::vector<double>& d_chain) {
    auto same = [](const std::vector<double>& a, const std::vector<double>& b) {
        return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(double)) == 0);
    };
    std::string bad;
    if (!same(d_exec, d_chain)) bad = "returned delays";
    else if (!same(c_exec, c_chain)) bad = "returned costs";
    else if (tr_exec.size() != traces.size()) bad = "number of windows";
    else
        for (size_t w = 0; w < traces.size() && bad.empty(); ++w)
            if (!same(tr_exec[w], traces[w])) bad = "trace of window " + std::to_string(w);
    executor_checked += 1;
    if (bad.empty()) return true;
    const std::string msg = std::string("window executor check (") + what + "): the " + bad + " differ from the launch chain's -- a hand-off between "
                            "workgroups delivered a stale value; set RSSYNC_EXECUTOR=0 and report this";
    if (executor_check) panic(msg); // the check MODE: a difference is what it exists to find
    // a call of the production sample: say so, hand out the chain's results (the caller), and keep to the chain
    executor_mismatches += 1;
    size_t w_bad = 0;
    for (; w_bad < d_exec.size() && w_bad < d_chain.size(); ++w_bad)
        if (std::memcmp(&d_exec[w_bad], &d_chain[w_bad], sizeof(double)) != 0) break;
    std::cerr << "rssync: " << msg << "\nrssync: this call returns the launch chain's results and this object uses the chain from now on";
    if (w_bad < d_exec.size() && w_bad < d_chain.size())
        std::cerr << " (first differing window " << w_bad << ": executor " << d_exec[w_bad] << " s, chain " << d_chain[w_bad] << " s)";
    std::cerr << std::endl;
    return false;
}

void SyncProblemHip::sync_windows(const std::vector<int64_t>& begins, const std::vector<int64_t>& ends_incl,
                                  const std::vector<double>& initial, double search_center, double search_radius,
                                  std::vector<double>& costs, std::vector<double>& delays_out, uint32_t call_stride,
                                  bool simplified) {
    rs::RoctxRange roctx_range("rssync:sync_windows");
    ensure_device();
    const size_t W = begins.size();
    select_windows(begins, ends_incl);
    if (executor_ok(simplified) &&
        // frames of up to 512 tracks: the search, the loop and the final loss of every window as one launch
        sync_exec(begins, ends_incl, initial, search_center, search_radius, 1, kStreamSyncInit + sync_calls, call_stride, costs,
                  delays_out)) {
        if (check_this_call()) { // the same call once more through the launch chain (which advances the call counter itself)
            const std::vector<double> c_exec = costs;
            std::vector<double> d_exec = delays_out;
            // (TESTS: RSSYNC_EXECUTOR_INJECT_MISMATCH=1 makes the executor's first delay wrong by one ulp, as a stale hand-off
            // could, so that the handling of a mismatch can be exercised: tests/test_gpu_executor.py)
            if (const char* inj = std::getenv("RSSYNC_EXECUTOR_INJECT_MISMATCH"))
                if (inj[0] == '1' && !d_exec.empty()) { d_exec[0] = std::nextafter(d_exec[0], 1.0); delays_out[0] = d_exec[0]; }
            const std::vector<std::vector<double>> tr_exec = traces;
            std::vector<double> c_chain, d_chain;
            {
                ChainRerun scope(this);
                sync_windows(begins, ends_incl, initial, search_center, search_radius, c_chain, d_chain, call_stride, simplified);
            }
            if (!check_against_chain("sync_windows", c_exec, d_exec, tr_exec, c_chain, d_chain)) {
                costs = c_chain;        // (`traces` are the chain's already)
                delays_out = d_chain;
                use_executor = false;
            }
        } else if (call_stride == 1) sync_calls += (uint32_t)W;
        if (verbose && W == 1) { // :330, the lines the host loop would have written
            int conv = 0;
            for (size_t it = 0; it * 6 < traces[0].size(); ++it) {
                const double* row = &traces[0][it * 6];
                const double step_size = std::fabs(row[1]);
                if (step_size < 1e-4) conv++; else conv = 0;
                const bool stop = conv > 5 || std::fabs(row[0] - search_center) > search_radius;
                if (!stop) std::cerr << row[0] << " " << step_size << std::endl;
            }
        }
        return;
    }
    std::vector<double> d(initial);
    // :218-223; window w samples with stream SYNC_INIT + sync_calls + w * call_stride.  With
    // stride 1 the call consumes W consecutive call numbers; sync_points() interleaves
    // several batched calls and advances the counter itself.
    if (simplified) {
        init_k_simple(d); // no v_i to guess or optimise (thesis section 2.11): only the hyper-parameter
    } else {
        init_motion(d, call_stride);
        if (call_stride == 1) sync_calls += (uint32_t)W;
    }
    traces.assign(W, {});

    // One device holds every frame and nobody else takes part in the sums: the loop below runs on the device
    // (rship_sync_run, kernels/syncloop.hpp -- the same decisions, taken between the launches without a host
    // round trip), and only its results are read back.
    // With ranks and the library's own RCCL communicator the loop stays on the device as well: the window sums are
    // all-reduced on the stream between the kernels.  (Whether a rank holds frames of the selection must not decide
    // the path then: every rank has to issue the same collectives.)
    const bool hook_loop = hook_device_loop && reduce_fn && !native_exchange;
    const bool device_loop = shards_.size() == 1 && !host_loop && rship_has_device_loop() && max_outer > 0 &&
                             (native_exchange || hook_loop || (!distributed() && !sel_.empty()));
    if (device_loop) {
        Shard& sh = shards_[0];
        struct Tramp {
            static int call(void* user, double* sums, uint64_t n) {
                SyncProblemHip* self = static_cast<SyncProblemHip*>(user);
                const int rc = self->reduce_fn(sums, (size_t)n, self->reduce_user);
                if (rc == 0) { self->exchange_calls += 1; self->exchange_doubles += n; }
                return rc;
            }
        };
        rship_set_loop_exchange(sh.ctx, hook_loop ? &Tramp::call : nullptr, hook_loop ? this : nullptr);
        std::vector<double> tr((size_t)W * max_outer * 6, 0.0);
        std::vector<int32_t> its(W, 0);
        hip_check(sh, rship_sync_run(sh.ctx, d.data(), max_outer, search_center, search_radius, simplified ? 1 : 0, d.data(),
                                     its.data(), tr.data()),
                  "sync loop");
        if (native_exchange) { // the all-reduces the loop enqueued between its kernels
            const uint64_t n = rship_loop_exchanges(sh.ctx);
            exchange_calls += n;
            exchange_doubles += n * 6 * W; // (two rows + ten rows per pair of exchanges)
        }
        for (size_t w = 0; w < W; ++w)
            traces[w].assign(tr.begin() + w * (size_t)max_outer * 6, tr.begin() + (w * (size_t)max_outer + its[w]) * 6);
        if (verbose && W == 1) { // :330, the lines the host loop would have written
            int conv = 0;
            for (int it = 0; it < its[0]; ++it) {
                const double* row = &traces[0][(size_t)it * 6];
                const double step_size = std::fabs(row[1]);
                if (step_size < 1e-4) conv++; else conv = 0;
                const bool stop = conv > 5 || std::fabs(row[0] - search_center) > search_radius;
                if (!stop) std::cerr << row[0] << " " << step_size << std::endl;
            }
        }
        std::vector<double> lfin;
        loss(d, lfin, nullptr, simplified); // :333
        costs = lfin;
        delays_out = d;
        return;
    }

    const double c_armijo = 2e-4, decay = .1, t0 = 1e-3; // :226
    // the fewest trials a search's first batch holds: five where a trial is cheap, one where the selection has frames
    // of 1024 tracks and more -- but always five with ranks, which must batch alike without knowing each other's frames
    // (the same rule as the device loop's SyncLoopParams::nf_floor).  Only the batching depends on it, never a result.
    const int max_bt = 10;
    uint32_t sel_max_tracks = 0;
    for (uint32_t i : sel_) sel_max_tracks = std::max(sel_max_tracks, frames_.at(table_ids_[i]).n);
    int half_bt = (!distributed() && sel_max_tracks >= 1024u) ? 1 : 5;
    if (const char* e = std::getenv("RSSYNC_LOOP_TRIALS_FLOOR")) { const int v = std::atoi(e); if (v >= 1 && v <= max_bt) half_bt = v; }
    const double delay_b = .3; // :260
    const double kOff = std::numeric_limits<double>::quiet_NaN();
    std::vector<double> delay_v(W, 0.0); // :261
    std::vector<int> converge_counter(W, 0);
    std::vector<char> active(W, 1);
    costs.assign(W, 0.0);
    std::vector<double> l1, g1, lt, cur(W), x0(W), trial;
    double ts[16];
    {
        double t = t0;
        for (int i = 0; i < max_bt; ++i) { ts[i] = t; t *= decay; }
        ts[max_bt] = t;
    }
    size_t n_active = W;
    int prev_hit = -1;
    for (int it = 0; it < max_outer && n_active; ++it) { // :309
        for (size_t w = 0; w < W; ++w) cur[w] = active[w] ? d[w] : kOff;
        if (!simplified) opt_motion(cur, nullptr); // :311
        // do_opt_delay (:298-305) -> Backtrack::Step (backtrack.cpp:3-13)
        for (size_t w = 0; w < W; ++w) x0[w] = active[w] ? d[w] - delay_b * delay_v[w] : kOff;
        loss(x0, l1, &g1, simplified);
        // The <= 10 backtracking trials (backtrack.cpp:7-11) are evaluated in one or two batched launches
        // and the first that satisfies the Armijo test is taken, which is what the sequential loop
        // returns.  The first batch holds as many trials as the previous outer iteration needed (at
        // least five): the step scale barely changes between iterations, so the second batch -- the
        // remaining trials, for the windows whose first batch failed throughout -- is rarely launched.
        std::vector<int> hit(W, -1); // index of the first successful trial
        const int n_first = std::min(max_bt, std::max(half_bt, prev_hit + 1));
        for (int b0 = 0; b0 < max_bt; b0 = (b0 == 0 ? n_first : max_bt)) {
            const int nb = (b0 == 0 ? n_first : max_bt) - b0;
            if (nb <= 0) break;
            bool need = false;
            trial.assign((size_t)nb * W, kOff);
            for (size_t w = 0; w < W; ++w) {
                const bool want = active[w] && hit[w] < 0;
                need = need || want;
                for (int i = 0; i < nb && want; ++i) trial[(size_t)i * W + w] = x0[w] - ts[b0 + i] * g1[w];
            }
            if (!need) break;
            loss(trial, lt, nullptr, simplified);
            for (size_t w = 0; w < W; ++w) {
                if (!active[w] || hit[w] >= 0) continue;
                const double m = g1[w] * g1[w];
                for (int i = 0; i < nb; ++i) {
                    if (l1[w] - lt[(size_t)i * W + w] >= ts[b0 + i] * c_armijo * m) {
                        hit[w] = b0 + i;
                        break;
                    }
                }
            }
        }
        prev_hit = half_bt - 1;
        for (size_t w = 0; w < W; ++w)
            if (active[w]) prev_hit = std::max(prev_hit, hit[w] >= 0 ? hit[w] : max_bt - 1);
        for (size_t w = 0; w < W; ++w) {
            if (!active[w]) continue;
            const double v = l1[w], p = g1[w];
            // never satisfied: t0 * decay^max_bt, untested (backtrack.cpp:11-12)
            const double t = hit[w] >= 0 ? ts[hit[w]] : ts[max_bt];
            const int trials = hit[w] >= 0 ? hit[w] + 1 : max_bt;
            const double step = -t * p;
            delay_v[w] = delay_b * delay_v[w] + step; // :301
            d[w] += delay_v[w];                       // :302
            const double step_size = std::fabs(step);
            const double row[6] = {d[w], step, v, p, t, (double)trials};
            traces[w].insert(traces[w].end(), row, row + 6);
            if (step_size < 1e-4) converge_counter[w]++; else converge_counter[w] = 0; // :316-320
            bool stop = converge_counter[w] > 5;                                       // :322-324
            if (!stop && std::fabs(d[w] - search_center) > search_radius) stop = true; // :326-328
            if (!stop && verbose && W == 1) std::cerr << d[w] << " " << step_size << std::endl; // :330
            if (stop || it + 1 == max_outer) {
                active[w] = 0;
                --n_active;
            }
        }
    }
    // :333, the loss at the returned delay.  A window's motion estimates are frozen once it has left
    // its loop (the motion launch skips it), so one evaluation after the last window has finished gives
    // every window the value it would have got at the moment it stopped.
    if (!simplified && max_outer <= 0) finish_init(d); // no iteration at all: GuessMotion/GuessK only
    loss(d, l1, nullptr, simplified);
    costs = l1;
    delays_out = d;
}

std::pair<double, double> SyncProblemHip::Sync(double initial_delay, int64_t frame_begin, int64_t frame_end,
                                               double search_center, double search_radius) {
    rs::RoctxRange roctx_range("rssync:Sync");
    std::vector<double> c, d;
    sync_windows({frame_begin}, {frame_end}, {initial_delay}, search_center, search_radius, c, d);
    trace = traces[0];
    return {c[0], d[0]};
}

// PreSync for W windows that share the candidate list (same initial delay, step and radius --
// the reference driver's pattern, core_testcode.cpp:303-311): the LMedS kernel runs once over the
// union of the windows' frames, then each window sums its own frames.  Window w equals a
// PreSync call on [begin[w], end[w]).
void SyncProblemHip::presync_windows(double initial_delay, const std::vector<int64_t>& begins,
                                     const std::vector<int64_t>& ends_excl, double search_step, double search_radius,
                                     std::vector<double>& costs, std::vector<double>& delays_out) {
    rs::RoctxRange roctx_range("rssync:presync_windows");
    ensure_device();
    const size_t W = begins.size();
    int64_t lo = std::numeric_limits<int64_t>::max(), hi = std::numeric_limits<int64_t>::min();
    for (size_t w = 0; w < W; ++w) { lo = std::min(lo, begins[w]); hi = std::max(hi, ends_excl[w]); }
    select(lo, hi);
    // keep only frames some window covers; build the per-window slot lists
    std::vector<uint32_t> keep, slot_of(table_ids_.size(), 0xffffffffu);
    for (uint32_t i : sel_) {
        bool used = false;
        for (size_t w = 0; w < W && !used; ++w) used = table_ids_[i] >= begins[w] && table_ids_[i] < ends_excl[w];
        if (used) { slot_of[i] = (uint32_t)keep.size(); keep.push_back(i); }
    }
    sel_ = keep;
    n_windows_ = 1;
    std::vector<uint32_t> seg_idx, seg_off(W + 1, 0);
    for (size_t w = 0; w < W; ++w) {
        for (uint32_t i : sel_)
            if (table_ids_[i] >= begins[w] && table_ids_[i] < ends_excl[w]) seg_idx.push_back(slot_of[i]);
        seg_off[w + 1] = (uint32_t)seg_idx.size();
    }
    apply_selection(sel_, {}, &seg_idx, &seg_off); // one ungrouped selection, W (possibly overlapping) windows to sum
    std::vector<double> delays; // :69-70
    for (double delay = initial_delay - search_radius; delay < initial_delay + search_radius; delay += search_step) {
        delays.push_back(delay);
        if (delays.size() > 50000000) panic("pre-sync: more than 5e7 candidate delays");
    }
    if (delays.empty()) panic("pre-sync: empty candidate list");
    const size_t n = delays.size();
    std::vector<double> cw(n * W + 4, 0.0);
    uint32_t flags = 0;
    sweep_windows(delays, 0, cw.data(), &flags, nullptr, nullptr);
    for (int b = 0; b < 4; ++b) cw[n * W + b] = (double)((flags >> b) & 1);
    reduce(cw.data(), cw.size());
    uint32_t all = 0;
    for (int b = 0; b < 4; ++b) all |= cw[n * W + b] > 0 ? (1u << b) : 0u;
    if (const char* msg = presync_panic(all)) panic(msg);
    costs.assign(W, 0.0);
    delays_out.assign(W, 0.0);
    for (size_t w = 0; w < W; ++w) {
        size_t best = 0; // *std::min_element over pair(cost, delay) (:89)
        for (size_t i = 1; i < n; ++i)
            if (std::make_pair(cw[i * W + w], delays[i]) < std::make_pair(cw[best * W + w], delays[best])) best = i;
        costs[w] = cw[best * W + w];
        delays_out[w] = delays[best];
    }
}

// The reference driver's loop over sync points (core_testcode.cpp:303-316) as batched calls:
// for every position `pos`: delay = initial; optionally delay = PreSync(delay, pos, pos + window,
// step, radius).second; then `repeats` (4 in the driver) times delay = Sync(delay, pos,
// pos + window, initial, radius).second.  Position w, repeat r uses the sampler stream the
// sequential loop would (call number w * repeats + r), so the results are those of the loop.
void SyncProblemHip::sync_points(const std::vector<int64_t>& positions, int64_t window, double initial_delay,
                                 bool use_presync, double presync_step, double presync_radius, int repeats,
                                 std::vector<double>& costs, std::vector<double>& delays_out) {
    rs::RoctxRange roctx_range("rssync:sync_points");
    const size_t W = positions.size();
    std::vector<int64_t> ends(W);
    for (size_t w = 0; w < W; ++w) ends[w] = positions[w] + window;
    std::vector<double> d(W, initial_delay);
    costs.assign(W, 0.0);
    double radius = std::numeric_limits<double>::infinity(); // :307
    if (use_presync) {                                          // :308-312
        radius = presync_radius;
        std::vector<double> c;
        presync_windows(initial_delay, positions, ends, presync_step, presync_radius, c, d);
    }
    const uint32_t first_call = sync_calls;
    std::vector<double> c_exec, d_exec;
    std::vector<std::vector<double>> tr_exec;
    bool checking = false;
    if (repeats >= 1 && repeats <= 8) {
        // all repeats of all positions in ONE launch where the executor can run them: a position starts its next
        // call the moment it has finished the previous one
        ensure_device();
        select_windows(positions, ends);
        if (executor_ok(false) && sync_exec(positions, ends, d, initial_delay, radius, repeats, kStreamSyncInit + first_call,
                                            (uint32_t)repeats, costs, delays_out)) {
            sync_calls = first_call + (uint32_t)(W * (size_t)repeats);
            if (!check_this_call()) return;
            // verified call (RSSYNC_EXECUTOR_CHECK, or this one's turn): the chain below runs the same calls; compare, then
            // hand out the (identical) results
            c_exec = costs; d_exec = delays_out; tr_exec = traces;
            checking = true;
        }
    }
    std::unique_ptr<ChainRerun> rerun;
    if (checking) rerun.reset(new ChainRerun(this));
    std::vector<std::vector<double>> all(W);
    for (int r = 0; r < repeats; ++r) { // :314 (Sync's range is end-inclusive: window + 1 frames)
        sync_calls = first_call + (uint32_t)r;
        std::vector<double> dn;
        sync_windows(positions, ends, d, initial_delay, radius, costs, dn, (uint32_t)std::max(1, repeats));
        d = dn;
        for (size_t w = 0; w < W; ++w) all[w].insert(all[w].end(), traces[w].begin(), traces[w].end());
    }
    sync_calls = first_call + (uint32_t)(W * (size_t)std::max(0, repeats));
    traces = all; // every repeat's rows, in order
    delays_out = d;
    if (checking) {
        rerun.reset();
        if (!check_against_chain("sync_points", c_exec, d_exec, tr_exec, costs, delays_out)) use_executor = false; // (costs / delays_out / traces are the chain's)
    }
}

} // namespace

// core_private.cpp:363, :365
ISyncProblem* CreateSyncProblem() { return new SyncProblemHip(); }
ISyncProblem::~ISyncProblem() {}

// ===========================================================================
// flat C-ABI (rssync_c.h)

struct rssync_problem {
    ISyncProblem* iface;
    SyncProblemHip* impl;
    bool owns = true;
};

namespace {
template <typename F>
int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const PanicError& e) {
        g_last_error = e.what();
        return 1;
    } catch (const std::exception& e) {
        g_last_error = std::string("exception: ") + e.what();
        return 2;
    }
}
} // namespace

extern "C" {

rssync_problem* rssync_create(void) {
    rssync_problem* h = nullptr;
    int saved = g_panic_mode;
    g_panic_mode = 1; // a missing device is reported, not fatal, on this entry point
    int rc = guarded([&] {
        ISyncProblem* i = CreateSyncProblem();
        h = new rssync_problem{i, static_cast<SyncProblemHip*>(i)};
    });
    g_panic_mode = saved;
    return rc ? nullptr : h;
}

void rssync_destroy(rssync_problem* p) {
    if (!p) return;
    if (p->owns) delete p->iface;
    delete p;
}

rssync_problem* rssync_ext_borrow(void* isync_problem) {
    SyncProblemHip* impl = dynamic_cast<SyncProblemHip*>(static_cast<ISyncProblem*>(isync_problem));
    if (!impl) {
        g_last_error = "borrow: not an ISyncProblem created by this library";
        return nullptr;
    }
    return new rssync_problem{impl, impl, false};
}

const char* rssync_last_error(void) { return g_last_error.c_str(); }
void rssync_set_panic_mode(int mode) { g_panic_mode = mode ? 1 : 0; }

int rssync_set_gyro_quaternions(rssync_problem* p, const double* data, size_t count, double sample_rate,
                                double first_timestamp) {
    return guarded([&] { p->iface->SetGyroQuaternions(data, count, sample_rate, first_timestamp); });
}
int rssync_set_gyro_quaternions_ts(rssync_problem* p, const int64_t* timestamps_us, const double* quats, size_t count) {
    return guarded([&] { p->iface->SetGyroQuaternions(timestamps_us, quats, count); });
}
int rssync_set_track_result(rssync_problem* p, int64_t frame, const double* ts_a, const double* ts_b,
                            const double* rays_a, const double* rays_b, size_t count) {
    return guarded([&] { p->iface->SetTrackResult(frame, ts_a, ts_b, rays_a, rays_b, count); });
}
int rssync_pre_sync(rssync_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end, double search_step,
                    double search_radius, double* cost, double* delay) {
    return guarded([&] {
        auto r = p->iface->PreSync(initial_delay, frame_begin, frame_end, search_step, search_radius);
        *cost = r.first;
        *delay = r.second;
    });
}
int rssync_sync(rssync_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end, double search_center,
                double search_radius, double* cost, double* delay) {
    return guarded([&] {
        auto r = p->iface->Sync(initial_delay, frame_begin, frame_end, search_center, search_radius);
        *cost = r.first;
        *delay = r.second;
    });
}
int rssync_debug_pre_sync(rssync_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end,
                          double search_radius, double* delays, double* costs, int point_count) {
    return guarded([&] {
        p->iface->DebugPreSync(initial_delay, frame_begin, frame_end, search_radius, delays, costs, point_count);
    });
}

int rssync_ext_set_seed(rssync_problem* p, uint64_t seed) { p->impl->seed = seed; return 0; }
int rssync_ext_set_max_outer_iters(rssync_problem* p, int iters) { p->impl->max_outer = iters; return 0; }
int rssync_ext_set_verbose(rssync_problem* p, int verbose) { p->impl->verbose = verbose != 0; return 0; }
int rssync_ext_set_lbfgs_reeval(rssync_problem* p, int reeval) {
    return guarded([&] {
        p->impl->set_option(RSHIP_OPT_LBFGS_REEVAL, reeval);
    });
}
int rssync_ext_set_host_loop(rssync_problem* p, int host_loop) {
    p->impl->host_loop = host_loop != 0;
    return 0;
}
int rssync_ext_set_hook_device_loop(rssync_problem* p, int on) {
    p->impl->hook_device_loop = on != 0;
    return 0;
}
int rssync_ext_lbfgs_best_not_last(rssync_problem* p, uint64_t* count) {
    *count = p->impl->last_best_not_last;
    return 0;
}
int rssync_ext_set_stream(rssync_problem* p, void* hip_stream) {
    return guarded([&] {
        if (p->impl->n_devices() != 1) panic("set-stream: this object drives several GPUs; each keeps its own stream");
        if (rship_set_stream(p->impl->dev(), hip_stream)) panic(std::string("hip: set stream: ") + rship_last_error(p->impl->dev()));
    });
}
int rssync_ext_set_reduce_hook(rssync_problem* p, rssync_reduce_fn fn, void* user) {
    p->impl->reduce_fn = fn;
    p->impl->reduce_user = user;
    return 0;
}

int rssync_ext_rccl_preflight(rssync_problem* p) {
    return guarded([&] {
        if (rship_rccl_preflight(p->impl->dev())) panic(std::string("hip: ") + rship_last_error(p->impl->dev()));
    });
}
const char* rssync_ext_rccl_library(rssync_problem* p) { return rship_rccl_library(p->impl->dev()); }

int rssync_ext_rccl_unique_id(rssync_problem* p, void* id128) {
    return guarded([&] {
        if (rship_rccl_unique_id(p->impl->dev(), id128)) panic(std::string("hip: ") + rship_last_error(p->impl->dev()));
    });
}

int rssync_ext_rccl_init(rssync_problem* p, const void* id128, int rank, int world_size) {
    return guarded([&] {
        if (rship_rccl_init(p->impl->dev(), id128, rank, world_size)) panic(std::string("hip: ") + rship_last_error(p->impl->dev()));
        p->impl->native_exchange = true;
    });
}

int rssync_ext_rccl_shutdown(rssync_problem* p) {
    return guarded([&] { p->impl->rccl_shutdown(); });
}

int rssync_ext_set_tracks_hint(rssync_problem* p, uint32_t max_tracks_all_ranks) {
    p->impl->set_tracks_hint(max_tracks_all_ranks);
    return 0;
}

int rssync_ext_exchange_stats(rssync_problem* p, uint64_t* calls, uint64_t* doubles) {
    if (calls) *calls = p->impl->exchange_calls;
    if (doubles) *doubles = p->impl->exchange_doubles;
    return 0;
}

int rssync_ext_window_info(rssync_problem* p, uint32_t out[8]) {
    return guarded([&] {
        for (int i = 0; i < 8; ++i) out[i] = 0;
        if (rship_window_info(p->impl->dev(), out)) panic(std::string("hip: ") + rship_last_error(p->impl->dev()));
    });
}

int rssync_ext_set_executor_check(rssync_problem* p, int on) {
    p->impl->executor_check = on != 0;
    return 0;
}
int rssync_ext_set_executor_check_every(rssync_problem* p, uint32_t every) {
    p->impl->executor_check_every = every;
    return 0;
}
int rssync_ext_executor_stats(rssync_problem* p, uint64_t* runs, uint64_t* checked, uint32_t queue[4]) {
    return guarded([&] {
        if (runs) *runs = p->impl->executor_runs;
        if (checked) *checked = p->impl->executor_checked;
        if (queue) p->impl->executor_queue_stats(queue);
    });
}

int rssync_ext_debug_residuals(rssync_problem* p, int on, uint32_t cap_rows) {
    return guarded([&] { p->impl->debug_residuals(on, cap_rows); });
}
int rssync_ext_debug_residuals_get(rssync_problem* p, uint32_t* out, size_t n_words, uint32_t dims[4]) {
    return guarded([&] { p->impl->debug_residuals_get(out, n_words, dims); });
}

int rssync_ext_executor_mismatches(rssync_problem* p, uint64_t* count) {
    if (count) *count = p->impl->executor_mismatches;
    return 0;
}

int rssync_ext_near_static_stats(rssync_problem* p, uint64_t* pairs, uint64_t* sweeps, uint64_t* searches) {
    return guarded([&] {
        uint64_t v[3];
        p->impl->near_static_stats(v);
        if (pairs) *pairs = v[0];
        if (sweeps) *sweeps = v[1];
        if (searches) *searches = v[2];
    });
}

int rssync_ext_record_init_winners(rssync_problem* p, int on) {
    p->impl->record_init = on != 0;
    return 0;
}
int rssync_ext_last_init_winners(rssync_problem* p, int32_t* out, size_t cap, size_t* n) {
    const std::vector<int32_t>& w = p->impl->last_init_winners;
    if (n) *n = w.size();
    if (out) std::copy(w.begin(), w.begin() + std::min(cap, w.size()), out);
    return 0;
}
int rssync_ext_set_init_override(rssync_problem* p, const int32_t* winners, size_t n) {
    p->impl->init_override.assign(winners, winners + n);
    return 0;
}
int rssync_ext_debug_math64(rssync_problem* p, int op, const double* a, const double* b, double* out, size_t n) {
    return guarded([&] {
        if (rship_debug_math64(p->impl->dev(), op, a, b, out, (uint32_t)n)) panic(std::string("hip: ") + rship_last_error(p->impl->dev()));
    });
}

int rssync_ext_upload(rssync_problem* p) {
    return guarded([&] { p->impl->ensure_device(); });
}

int rssync_ext_sample_rate(rssync_problem* p, double* sample_rate, double* quats_start, size_t* n_knots) {
    if (sample_rate) *sample_rate = p->impl->sample_rate();
    if (quats_start) *quats_start = p->impl->quats_start();
    if (n_knots) *n_knots = p->impl->n_knots();
    return 0;
}
int rssync_ext_gyro_knots(rssync_problem* p, double* out, size_t cap) {
    return guarded([&] {
        const auto& k = p->impl->knots();
        if (cap < k.size()) panic("gyro-knots: buffer too small");
        std::copy(k.begin(), k.end(), out);
    });
}

int rssync_ext_gyro_table(rssync_problem* p, double* out, size_t cap) {
    return guarded([&] {
        SyncProblemHip* s = p->impl;
        if (s->n_knots() < 2) panic("sync: gyro data was not set");
        if (cap < 16 * s->n_knots()) panic("gyro-table: buffer too small");
        s->ensure_spline();
        if (rship_gyro_table(s->dev(), out, (uint32_t)s->n_knots())) panic(std::string("gyro table: ") + rship_last_error(s->dev()));
    });
}

int rssync_ext_presync_curve(rssync_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end,
                             double search_step, double search_radius, double* delays, double* costs, int cap,
                             int* n_out, double* frame_costs, int32_t* best_h, int* n_frames) {
    return guarded([&] {
        SyncProblemHip* s = p->impl;
        s->ensure_device();
        uint32_t nf = s->select(frame_begin, frame_end);
        std::vector<double> d;
        for (double delay = initial_delay - search_radius; delay < initial_delay + search_radius; delay += search_step) {
            if ((int)d.size() >= cap) panic("presync_curve: candidate capacity exceeded");
            d.push_back(delay);
        }
        std::vector<double> c = s->sweep(d, 0, false, frame_costs, best_h);
        std::copy(d.begin(), d.end(), delays);
        std::copy(c.begin(), c.end(), costs);
        if (n_out) *n_out = (int)d.size();
        if (n_frames) *n_frames = (int)nf;
    });
}

int rssync_ext_problem_matrix(rssync_problem* p, int64_t frame, double delay, float* P, float* dP, size_t cap_rows,
                              size_t* n_rows) {
    return guarded([&] {
        SyncProblemHip* s = p->impl;
        s->ensure_device();
        if (!s->has_frame(frame)) panic("problem_matrix: unknown frame");
        s->select(frame, frame + 1);
        s->debug_problem(frame, delay, P, dP, nullptr, nullptr, cap_rows);
        if (n_rows) *n_rows = s->frame_tracks(frame);
    });
}

int rssync_ext_init_motion(rssync_problem* p, double delay, int64_t frame_begin, int64_t frame_end, double* M, double* k,
                           int cap, int* n_frames) {
    return guarded([&] {
        SyncProblemHip* s = p->impl;
        s->ensure_device();
        s->select(frame_begin, frame_end == std::numeric_limits<int64_t>::max() ? frame_end : frame_end + 1);
        s->init_motion({delay});
        s->finish_init({delay});
        s->sync_calls++;
        const uint32_t n = s->get_motion(M, k, (uint32_t)cap);
        if (n_frames) *n_frames = (int)n;
    });
}

int rssync_ext_opt_motion(rssync_problem* p, double delay, double* M, double* k, int cap, int* n_frames, uint64_t* iters,
                          uint64_t* evals) {
    return guarded([&] {
        SyncProblemHip* s = p->impl;
        uint64_t st[3] = {0, 0, 0};
        s->opt_motion({delay}, st);
        const uint32_t n = s->get_motion(M, k, (uint32_t)cap);
        if (n_frames) *n_frames = (int)n;
        if (iters) *iters = st[0];
        if (evals) *evals = st[1];
        s->last_best_not_last = st[2];
    });
}

int rssync_ext_set_motion(rssync_problem* p, const double* M, const double* k, int n_frames) {
    return guarded([&] {
        p->impl->set_motion(M, k, (uint32_t)n_frames);
    });
}

int rssync_ext_loss(rssync_problem* p, const double* delays, int n, double* loss, double* grad) {
    return guarded([&] {
        std::vector<double> d(delays, delays + n), l, g;
        p->impl->loss(d, l, grad ? &g : nullptr);
        std::copy(l.begin(), l.end(), loss);
        if (grad) std::copy(g.begin(), g.end(), grad);
    });
}

int rssync_ext_pre_sync_windows(rssync_problem* p, double initial_delay, const int64_t* frame_begins,
                                const int64_t* frame_ends, int n_windows, double search_step, double search_radius,
                                double* costs, double* delays) {
    return guarded([&] {
        if (n_windows <= 0) return;
        std::vector<double> c, d;
        p->impl->presync_windows(initial_delay, std::vector<int64_t>(frame_begins, frame_begins + n_windows),
                                 std::vector<int64_t>(frame_ends, frame_ends + n_windows), search_step, search_radius, c,
                                 d);
        std::copy(c.begin(), c.end(), costs);
        std::copy(d.begin(), d.end(), delays);
    });
}

int rssync_ext_sync_windows(rssync_problem* p, const double* initial_delays, const int64_t* frame_begins,
                            const int64_t* frame_ends, int n_windows, double search_center, double search_radius,
                            double* costs, double* delays) {
    return guarded([&] {
        if (n_windows <= 0) return;
        std::vector<double> c, d;
        p->impl->sync_windows(std::vector<int64_t>(frame_begins, frame_begins + n_windows),
                              std::vector<int64_t>(frame_ends, frame_ends + n_windows),
                              std::vector<double>(initial_delays, initial_delays + n_windows), search_center,
                              search_radius, c, d);
        p->impl->trace = p->impl->traces[0];
        std::copy(c.begin(), c.end(), costs);
        std::copy(d.begin(), d.end(), delays);
    });
}

// Sync without translation (thesis section 2.11 eq. (12)): loss sum_j log1p((k |P_j|)^2), k per frame from
// GuessK with |P_j| in place of P_j . v, no motion estimate, otherwise core_private.cpp:211-334.
int rssync_ext_sync_simplified(rssync_problem* p, double initial_delay, int64_t frame_begin, int64_t frame_end,
                               double search_center, double search_radius, double* cost, double* delay) {
    return guarded([&] {
        std::vector<double> c, d;
        p->impl->sync_windows({frame_begin}, {frame_end}, {initial_delay}, search_center, search_radius, c, d, 1, true);
        p->impl->trace = p->impl->traces[0];
        *cost = c[0];
        *delay = d[0];
    });
}

int rssync_ext_init_k_simplified(rssync_problem* p, double delay, int64_t frame_begin, int64_t frame_end, double* k,
                                 int cap, int* n_frames) {
    return guarded([&] {
        SyncProblemHip* s = p->impl;
        s->ensure_device();
        s->select(frame_begin, frame_end == std::numeric_limits<int64_t>::max() ? frame_end : frame_end + 1);
        s->init_k_simple({delay});
        std::vector<double> M((size_t)std::max(cap, 0) * 3 + 3);
        const uint32_t n = s->get_motion(M.data(), k, (uint32_t)cap);
        if (n_frames) *n_frames = (int)n;
    });
}

int rssync_ext_loss_simplified(rssync_problem* p, const double* delays, int n, double* loss, double* grad) {
    return guarded([&] {
        std::vector<double> d(delays, delays + n), l, g;
        p->impl->loss(d, l, grad ? &g : nullptr, true);
        std::copy(l.begin(), l.end(), loss);
        if (grad) std::copy(g.begin(), g.end(), grad);
    });
}

int rssync_ext_problem_matrix64(rssync_problem* p, int64_t frame, double delay, double* P, double* dP, size_t cap_rows,
                                size_t* n_rows) {
    return guarded([&] {
        SyncProblemHip* s = p->impl;
        s->ensure_device();
        if (!s->has_frame(frame)) panic("problem_matrix: unknown frame");
        s->select(frame, frame + 1);
        s->debug_problem(frame, delay, nullptr, nullptr, P, dP, cap_rows);
        if (n_rows) *n_rows = s->frame_tracks(frame);
    });
}

int rssync_ext_set_track_pixels(rssync_problem* p, int64_t frame, double frame_time_a, double frame_time_b,
                                const double* points_a, const double* points_b, size_t count, const rssync_lens* lens,
                                double image_rows) {
    return guarded([&] {
        if (!lens) panic("set-track-pixels: no lens");
        const double l[9] = {lens->ro, lens->fx, lens->fy, lens->cx, lens->cy, lens->k1, lens->k2, lens->k3, lens->k4};
        p->impl->SetTrackPixels(frame, frame_time_a, frame_time_b, points_a, points_b, count, l, image_rows);
    });
}

int rssync_ext_set_gyro_rates(rssync_problem* p, const double* timestamps_s, const double* rates, size_t count,
                              const char* orientation) {
    return guarded([&] { p->impl->SetGyroRates(timestamps_s, rates, count, orientation); });
}

int rssync_ext_orientation_sweep(rssync_problem* p, const double* timestamps_s, const double* rates, size_t count,
                                 const char* const* orientations, int n_orientations, double initial_delay,
                                 int64_t frame_begin, int64_t frame_end, double search_step, double search_radius,
                                 double* costs, double* delays) {
    return guarded([&] {
        std::vector<std::string> o;
        for (int i = 0; i < n_orientations; ++i) o.emplace_back(orientations[i] ? orientations[i] : "XYZ");
        p->impl->orientation_sweep(timestamps_s, rates, count, o, initial_delay, frame_begin, frame_end, search_step,
                                   search_radius, costs, delays);
    });
}

int rssync_ext_frame_rays(rssync_problem* p, int64_t frame, float* a4, float* b4, size_t cap, size_t* n) {
    return guarded([&] {
        SyncProblemHip* s = p->impl;
        s->ensure_device();
        for (uint32_t i = 0;; ++i) {
            if (i >= s->table_size()) panic("frame_rays: no such frame");
            if (s->table_id(i) != frame) continue;
            const size_t cnt = s->frame_tracks(frame);
            if (n) *n = cnt;
            if (cnt > cap) panic("frame_rays: output too small");
            s->debug_rays(frame, a4, b4, cap);
            return;
        }
    });
}

int rssync_ext_sync_points(rssync_problem* p, const int64_t* positions, int n_points, int64_t sync_window,
                           double initial_delay, int use_presync, double presync_step, double presync_radius,
                           int sync_repeats, double* costs, double* delays) {
    return guarded([&] {
        if (n_points <= 0) return;
        std::vector<double> c, d;
        p->impl->sync_points(std::vector<int64_t>(positions, positions + n_points), sync_window, initial_delay,
                             use_presync != 0, presync_step, presync_radius, sync_repeats, c, d);
        p->impl->trace = p->impl->traces[0];
        if (costs) std::copy(c.begin(), c.end(), costs);
        std::copy(d.begin(), d.end(), delays);
    });
}

int rssync_ext_window_trace(rssync_problem* p, int window, double* trace, int cap_rows, int* n_rows) {
    return guarded([&] {
        if (window < 0 || (size_t)window >= p->impl->traces.size()) throw PanicError("window_trace: no such window");
        const std::vector<double>& t = p->impl->traces[(size_t)window];
        int rows = (int)(t.size() / 6);
        if (n_rows) *n_rows = rows;
        rows = std::min(rows, cap_rows);
        if (trace) std::copy(t.begin(), t.begin() + (size_t)rows * 6, trace);
    });
}

int rssync_ext_sync_trace(rssync_problem* p, double* trace, int cap_rows, int* n_rows) {
    const auto& t = p->impl->trace;
    int rows = (int)(t.size() / 6);
    int n = std::min(rows, cap_rows);
    std::copy(t.begin(), t.begin() + 6 * (size_t)n, trace);
    if (n_rows) *n_rows = rows;
    return 0;
}

void* rssync_ext_device_context(rssync_problem* p) { return p->impl->dev(); }

int rssync_ext_profile(rssync_problem* p, int enable) { return guarded([&] { p->impl->profile_enable(enable); }); }
int rssync_ext_profile_get(rssync_problem* p, int kind, uint64_t* launches, double* total_ms) {
    return guarded([&] { p->impl->profile_get(kind, launches, total_ms); });
}
int rssync_ext_profile_reset(rssync_problem* p) { return guarded([&] { p->impl->profile_reset(); }); }

// Spread this object's frames over several GPUs of the process (device ordinals; the same ordinal may be
// listed more than once).  Frames are re-uploaded by the next call that needs them.
int rssync_ext_set_devices(rssync_problem* p, const int* device_ids, int n_devices) {
    return guarded([&] { p->impl->set_devices(std::vector<int>(device_ids, device_ids + std::max(n_devices, 0))); });
}
int rssync_ext_device_count(rssync_problem* p) { return (int)p->impl->n_devices(); }

} // extern "C"
