// window_plan.hpp -- how large the kernels' LDS spline windows are and how many candidate delays a workgroup takes
// (the host side of DESIGN.md section 3, "Gyro rate").  Pure arithmetic on plain numbers, no HIP: included by
// rssync_kernels.hip (which supplies the kernels' LDS footprints) and compiled on its own by tests/test_window_plan.py.
//
// Reference: a frame's rays are evaluated at x = (ts - quats_start + delay) * sample_rate (core_private.cpp:19-20), so
// a frame pair spans (ts range) * sample_rate knots of the spline and a chunk of candidate delays (its span) more;
// the reference takes any sample rate (:135-140, :146-149).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>

namespace rs {

constexpr uint32_t kPlanWinStatic = 80;   // knots compiled into the kernels' LDS (kernels/common.hpp: kWinMax)
constexpr uint32_t kPlanSmallWinMax = 128; // one wave per frame (K2s): no dynamic window beyond this many knots
constexpr uint32_t kPlanCap64Max = 384;   // fp64 windows: 48 KB
#ifndef RSSYNC_PLAN_CAP64_SMALL_MAX   // (a measurement build sets 384: tools/gpu_gyro_rate.py's small-frame sweep beyond the rule)
#define RSSYNC_PLAN_CAP64_SMALL_MAX 144
#endif
constexpr uint32_t kPlanCap64SmallMax = RSSYNC_PLAN_CAP64_SMALL_MAX; // problems of small frames: the 80-knot window beyond this (measured: below)

// ---- size classes and the shapes of PreSync's LMedS kernels ("size classes", "sub-shapes") ----
// Which kernel family a frame runs in follows from its OWN track count (core_private.cpp:73-86 evaluates every frame in its
// own lambda): class 0 up to one_wave_max tracks (one wave per frame), 1 .. 3 four waves (up to 1024 / 2048 / 6144 tracks),
// 4 up to 8192 (PreSync's tile kernel as eight waves), 5 beyond (rows in global memory).
constexpr uint32_t kPlanBlock = 256;           // threads of a four-wave workgroup = rows per "row of the tile per thread"
constexpr uint32_t kPlanFourWaveMaxRpt = 24;   // class 3 ends at 24 x 256 = 6144 tracks: two four-wave workgroups still share a CU
constexpr uint32_t kPlanMaxRpt = 32;           // 8192 tracks: the largest tile in LDS
inline int plan_class_of(uint32_t n, uint32_t one_wave_max) {
    if (n <= one_wave_max) return 0;
    if (n <= 4u * kPlanBlock) return 1;
    if (n <= 8u * kPlanBlock) return 2;
    if (n <= kPlanFourWaveMaxRpt * kPlanBlock) return 3;
    if (n <= kPlanMaxRpt * kPlanBlock) return 4;
    return 5;
}
// The tile kernel's shape is named by its CODE = rows of the tile / 256.  A class's OWN shape: 4, 8, 16 -- or 24 once the
// selection's largest frame of class 3 has more than 4096 tracks --, 32 (eight waves of 16 rows per thread).  0: class 5.
inline int plan_class_shape(int k, uint32_t cls_max_n) {
    if (k <= 0 || k >= 5) return 0;
    if (k == 3) return cls_max_n > 16u * kPlanBlock ? (int)kPlanFourWaveMaxRpt : 16;
    return 4 << (k - 1);
}
// ... and the SUB-SHAPE PreSync's sweep takes: the smallest shape that holds the largest frame of the class in the selection
// (3 .. 24 in four waves: any number of rows per thread; 26 .. 32 in eight: even codes) -- the same bits, fewer rows swept
inline int plan_sub_shape(int k, uint32_t cls_max_n) {
    const int own = plan_class_shape(k, cls_max_n);
    if (!own) return 0;
    int need = (int)((cls_max_n + kPlanBlock - 1u) / kPlanBlock);
    if (need < 3) need = 3;
    if (k == 4) need += need & 1;
    return need < own ? need : own;
}
// rows per lane of the one-wave kernels for a selection whose largest small frame has max_n tracks: 1 .. 4 up to 256 tracks,
// 8 beyond (the family's own), or -- PreSync's sweep -- 5 .. 7 where that is enough
inline int plan_small_rows(uint32_t max_n, bool sub_shapes) {
    const uint32_t r = std::max(1u, (max_n + 63u) / 64u);
    if (r <= 4u) return (int)r;
    return (sub_shapes && r <= 7u) ? (int)r : 8;
}

struct WinPlan {
    uint32_t cap = 0;   // 0: the compiled-in 80-knot window; otherwise knots of dynamic LDS (64 bytes each)
    uint32_t chunk = 1; // candidates per workgroup
    bool whole_pair = false; // a dynamic window that stages whole pairs only, as the compiled-in one does (same path decisions)
    bool extra_wg = false;   // ... chosen because it lets one more workgroup share the CU (smaller_window_for_occupancy)
};

// Frames whose knots fit the compiled-in window can still be better off with a SMALLER window in dynamic LDS: where the
// kernel's LDS, not its registers, decides how many workgroups share a CU and the compiled-in window is the difference
// (round 5: lmeds_kernel<16, ...>, 3841 .. 4096 tracks since round 6's sub-shapes -- 48 KB of tile + the 5 KB window = two workgroups per CU, with a
// window of the ~30 knots a 400 Hz frame and its chunk touch three).  static_lds / dyn_fixed_lds: the two instantiations'
// static LDS; wg_cap: workgroups per CU the kernel's registers allow.  -> knots of the dynamic window, or 0 = keep the
// compiled-in one.  The dynamic window then stages whole pairs, as the compiled-in one does: the same frames take the
// interior path, the same bits.
inline uint32_t smaller_window_for_occupancy(double span, double step_knots, uint32_t chunk, uint32_t static_lds, uint32_t dyn_fixed_lds,
                                             int wg_cap, int lds_per_cu) {
    if (!static_lds || !dyn_fixed_lds) return 0;
    const double cs = step_knots > 0 ? (chunk - 1) * step_knots : 0.0;
    const uint32_t cap = ((uint32_t)std::ceil(span + 1.0 + cs) + 3u) / 4u * 4u + 4u;
    if (cap >= kPlanWinStatic) return 0;
    auto wgs = [&](uint32_t lds) { return std::min(wg_cap, (int)(lds_per_cu / (int)(lds + 1024u))); }; // (allocation granularity)
    return wgs(dyn_fixed_lds + cap * 64u) > wgs(static_lds) ? cap : 0u;
}

// knots the fp64 kernels' window holds for a table whose widest frame touches `max_span` knots at one delay -- or
// `max_ends` knots where only the two ends of each pair are staged (a frame's a-end and b-end ranges one after the
// other: kernels/common.hpp stage_window_ends; max_ends = max_span for a table without that information)
inline uint32_t cap64_for(float max_span, float max_ends) {
    uint32_t need = (uint32_t)std::ceil(std::max(std::min(max_span, max_ends), 0.f)) + 1u;
    need = (need + 15u) / 16u * 16u;
    return std::min(std::max(need, kPlanWinStatic), kPlanCap64Max);
}
// ... and what the launches of a problem actually use: `one_wave` = its frames run in the one-wave kernels (K1 / K3 in
// their one-wave shapes, the window executor), which stage a window per evaluation of a few hundred fetches
inline uint32_t cap64_used(uint32_t cap64, bool one_wave) {
    if (one_wave && cap64 > kPlanCap64SmallMax) return kPlanWinStatic;
    return cap64;
}

// candidates whose delays fit a window of `cap` knots next to the widest frame (span knots at one delay; neighbouring
// candidates step_knots apart), at most chunk_want
inline uint32_t plan_fit(double cap, double span, double step_knots, uint32_t chunk_want) {
    if (cap < span + 1.0) return 0;
    if (!(step_knots > 0)) return chunk_want;
    const double n = std::floor((cap - span - 1.0) / step_knots) + 1.0;
    return n >= (double)chunk_want ? chunk_want : (uint32_t)n;
}
// the same where the two ends of a pair are staged one after the other (`ends` knots at one delay, both ends' carry
// knots included): every candidate of the chunk widens BOTH ranges
inline uint32_t plan_fit_ends(double cap, double ends, double step_knots, uint32_t chunk_want) {
    if (cap < ends + 2.0) return 0;
    if (!(step_knots > 0)) return chunk_want;
    const double n = std::floor((cap - ends - 2.0) / (2.0 * step_knots)) + 1.0;
    return n >= (double)chunk_want ? chunk_want : (uint32_t)n;
}

// The fp32 window of an LMedS launch (PreSync sweep or GuessMotion's search).
//   span        widest frame of the table, knots touched at one delay
//   ends        the same counting only the two ends' ranges of each pair (= span where the table does not know them):
//               the dynamic-window kernels stage the ends separately where that is fewer knots
//   step_knots  distance between neighbouring candidate delays (0: one candidate per workgroup)
//   chunk_want  candidates per workgroup the launch would like (<= 32)
//   small       the one-wave kernels (frames of up to 512 tracks); wg_max: most workgroups per CU the kernel runs at
//   fixed_lds   static LDS of the dynamic-window instantiation, lds_per_cu the CU's LDS
//   legacy      rounds 1-3: never a dynamic window (RSSYNC_FORCE_GENERAL_SPLINE)
// -> cap == 0: the compiled-in window (if the chunk's frames do not fit it, their workgroups take the general path)
inline WinPlan plan_window(double span, double ends, double step_knots, uint32_t chunk_want, bool small, int wg_max, uint32_t fixed_lds,
                           int lds_per_cu, bool legacy) {
    WinPlan w;
    w.chunk = chunk_want;
    const uint32_t min_chunk = std::min(8u, chunk_want);
    const uint32_t f80 = plan_fit((double)kPlanWinStatic, span, step_knots, chunk_want);
    if (f80 >= min_chunk) { w.chunk = f80; return w; }
    if (legacy) { // shorten the chunk down to four candidates, else let the window overflow
        if (f80 >= 4) w.chunk = f80;
        return w;
    }
    if (!fixed_lds) return w;
    for (int wg = wg_max; wg >= 1; --wg) {
        const int share = lds_per_cu / wg - 1024; // (allocation granularity, alignment)
        if (share <= (int)fixed_lds) continue;
        uint32_t cap_t = std::min(2048u, ((uint32_t)share - fixed_lds) / 64u / 4u * 4u);
        // one wave per frame (K2s): a window of more than kPlanSmallWinMax knots costs more in resident waves than the
        // frame's few hundred coefficient fetches cost from L2 (profiles/r4_gyro_rate_sweep.json: a 228-knot window 1.98 ms
        // against 1.54 on the general path) -- never more than that; if even eight candidates do not fit it, the general path
        if (small) cap_t = std::min(cap_t, kPlanSmallWinMax);
        const uint32_t f1 = plan_fit((double)cap_t, span, step_knots, chunk_want);
        const uint32_t f2 = ends < span ? plan_fit_ends((double)cap_t, ends, step_knots, chunk_want) : 0u;
        const uint32_t f = std::max(f1, f2);
        if (f < min_chunk) continue;
        const double cs = step_knots > 0 ? (f - 1) * step_knots : 0.0;
        const double need = std::min(f1 >= f ? span + 1.0 + cs : 1e30, f2 >= f ? ends + 2.0 + 2.0 * cs : 1e30);
        const uint32_t cap = std::min(cap_t, ((uint32_t)std::ceil(need) + 3u) / 4u * 4u + 4u);
        w.chunk = f;
        w.cap = cap;
        return w;
    }
    return w; // does not fit any LDS share: the general path
}

// ---- per-FRAME planning (round 5) -----------------------------------------------------------------------------
// The reference evaluates every frame on its own (core_private.cpp:73-86, :231-238, :263-295): nothing about frame i
// knows frame j.  Rounds 1-4 planned a launch's window from the WIDEST frame of the table, so that one frame too wide
// for any LDS window sent every other frame of its problem to the general spline path (other fp32 roundings), and a
// shard that did not hold that frame planned differently from the single-device run.  Now a launch is planned from the
// frames that CAN have a window at all -- a frame is ELIGIBLE if its own knots (whole pair, or its two ends) plus the
// shortest chunk of candidates fit the largest window the kernel can run with; the others read the table from L2, as
// they would alone -- so whether a frame takes the interior path follows from the frame, not from its neighbours.
struct FrameDims {
    uint32_t n;  // tracks
    float span;  // knots the pair touches at one delay
    float ends;  // the same counting only the two ends' ranges (= span where unknown or not fewer)
};
inline float frame_span(float tmin, float tmax) { return std::floor(tmax) - std::floor(tmin) + 2.f; }
// range_a / range_b as rship_frame holds them (lo | hi << 16, 0xffffffff = unknown)
inline float frame_ends(uint32_t range_a, uint32_t range_b, float span) {
    if (range_a == 0xffffffffu || range_b == 0xffffffffu) return span;
    const int a_lo = (int)(range_a & 0xffffu), a_hi = (int)(range_a >> 16) + 1;
    const int b_lo = (int)(range_b & 0xffffu), b_hi = (int)(range_b >> 16) + 1;
    if (b_lo > a_hi + 1 || a_lo > b_hi + 1) return std::min(span, (float)((a_hi - a_lo + 1) + (b_hi - b_lo + 1)));
    return span;
}
// the largest dynamic window (knots) a kernel with `fixed_lds` bytes of static LDS can be launched with
inline uint32_t plan_cap_limit(bool small, uint32_t fixed_lds, int lds_per_cu) {
    const int share = lds_per_cu - 1024;
    if (share <= (int)fixed_lds) return 0;
    uint32_t cap_t = std::min(2048u, ((uint32_t)share - fixed_lds) / 64u / 4u * 4u);
    if (small) cap_t = std::min(cap_t, kPlanSmallWinMax);
    return cap_t;
}
// The fp32 window of an LMedS launch over the frames f[0 .. nf) whose track count lies in [n_lo, n_hi] (one size class).
// Same result as plan_window(span, ends, ...) of the class's widest frame whenever every frame of the class is eligible.
inline WinPlan plan_window_frames(const FrameDims* f, size_t nf, uint32_t n_lo, uint32_t n_hi, double step_knots, uint32_t chunk_want,
                                  bool small, int wg_max, uint32_t fixed_lds, int lds_per_cu, bool legacy) {
    WinPlan w;
    w.chunk = chunk_want;
    const uint32_t min_chunk = std::min(8u, chunk_want);
    double span_all = 0.0, ends_all = 0.0;
    for (size_t i = 0; i < nf; ++i)
        if (f[i].n >= n_lo && f[i].n <= n_hi && f[i].n) { span_all = std::max(span_all, (double)f[i].span); ends_all = std::max(ends_all, (double)f[i].ends); }
    const uint32_t f80 = plan_fit((double)kPlanWinStatic, span_all, step_knots, chunk_want);
    if (f80 >= min_chunk || legacy || !fixed_lds) return plan_window(span_all, ends_all, step_knots, chunk_want, small, wg_max, fixed_lds, lds_per_cu, legacy);
    const double limit = (double)plan_cap_limit(small, fixed_lds, lds_per_cu);
    double span_e = 0.0, ends_e = 0.0;
    bool any = false;
    for (size_t i = 0; i < nf; ++i) {
        if (f[i].n < n_lo || f[i].n > n_hi || !f[i].n) continue;
        const bool fits = plan_fit(limit, f[i].span, step_knots, min_chunk) >= min_chunk ||
                          (f[i].ends < f[i].span && plan_fit_ends(limit, f[i].ends, step_knots, min_chunk) >= min_chunk);
        if (!fits) continue;
        any = true;
        span_e = std::max(span_e, (double)f[i].span);
        ends_e = std::max(ends_e, (double)f[i].ends);
    }
    if (!any) return w; // nobody can have a window: the compiled-in one (its frames take the general path)
    return plan_window(span_e, ends_e, step_knots, chunk_want, small, wg_max, fixed_lds, lds_per_cu, legacy);
}
// knots of the fp64 window for the frames of one size class: the widest frame that a window of at most `limit` knots
// (kPlanCap64Max; kPlanCap64SmallMax for the one-wave kernels, above which a knot of window costs more in resident waves
// than the general path's fetches from L2) can hold; wider frames read the table from L2 -- the same bits either way
// One-wave kernels (round 5): beyond kPlanCompactFrom knots the window is COMPACT -- y and c of a knot only, 64 bytes
// instead of 128, b and d rebuilt per fetch with the table's own expressions (kernels/sync64.hpp: Spline64::compact) -- so
// that up to kPlanCompactMax knots still leave a CU its eight one-wave workgroups (8 kHz: a 130-track frame's two
// ends are 182 knots = 12 KB compact, eight waves per CU; as full records 23 KB, five waves: slower than the table from
// L2).  Up to kPlanCompactFrom knots (4 kHz) full records already leave eight waves per CU and cost no arithmetic.
constexpr uint32_t kPlanCompactFrom = 96;
// measured on 98 sync points of 61 x 130, iterations capped at 25 per call (profiles/r5_gyro_rate_small_frames.json), compact
// against round 4's rule (full records to 144 knots, then the table from L2): 6 kHz 19.0 ms against 20.2; 8 kHz (192 knots,
// 12 KB: eight waves per CU) 19.4 against 21.6; 12 kHz (288 knots, 18 KB: seven waves) 22.4 against 22.1 -- the rebuilt
// coefficients (~90 more fp64 instructions per row) eat what the window saves there.  So: compact up to 208 knots (13 KB,
// the most that leaves the executor its eight waves per CU), the table from L2 beyond.
constexpr uint32_t kPlanCompactMax = 208;
// (a compact range also holds the knot after its last one, per end: + 2)
inline uint32_t cap64_frames(const FrameDims* f, size_t nf, uint32_t n_lo, uint32_t n_hi, bool one_wave, bool* compact = nullptr) {
    const uint32_t limit = one_wave ? (compact ? kPlanCompactMax : kPlanCap64SmallMax) : kPlanCap64Max;
    uint32_t need = kPlanWinStatic;
    bool comp = false;
    for (size_t i = 0; i < nf; ++i) {
        if (f[i].n < n_lo || f[i].n > n_hi || !f[i].n) continue;
        const uint32_t raw = (uint32_t)std::ceil(std::max(std::min(f[i].span, f[i].ends), 0.f)) + 1u;
        const bool c = one_wave && compact && raw > kPlanCompactFrom;
        uint32_t k = ((c ? raw + 2u : raw) + 15u) / 16u * 16u;
        if (k <= limit && k > need) { need = k; }
        if (c && k <= limit) comp = true;
    }
    if (compact) *compact = comp;
    return need;
}

// ---- the window executor's per-wave LDS region (kernels/executor.hpp) -------------------------------------------
// ONE region of dynamic LDS per wave is in turn the search's fp32 spline window, the fp64 window of the motion / loss tasks
// and the decisions' staging area.  It must therefore hold the largest of the three:
//   * the one-wave class's fp64 window (cap64 knots at 128 bytes, 64 if compact),
//   * the fp32 window the launch chain's search kernel would use for the same frames (`search_cap` knots at 64 bytes,
//     0 = the compiled-in 80: the executor's search must take the chain's spline path, or near-ties fall the other way),
//   * kPlanExecStage doubles of staging (exec_window_sums: a window's per-slot values, rows x slots <= kPlanExecStage).
// Round 5's compact windows had let the first term fall to 7-9 KB (4.3-6.5 kHz on small frames) -- below both others.
constexpr uint32_t kPlanExecStage = 1280;
inline size_t exec_region_for(uint32_t cap64, bool compact, uint32_t search_cap) {
    const size_t win64 = (size_t)cap64 * (compact ? 64u : 128u);
    const size_t win32 = (size_t)(search_cap ? search_cap : kPlanWinStatic) * 64u;
    return std::max(std::max(win64, win32), (size_t)kPlanExecStage * 8u);
}

} // namespace rs
