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
constexpr uint32_t kPlanCap64SmallMax = 176; // problems of small frames: the 80-knot window beyond this

struct WinPlan {
    uint32_t cap = 0;   // 0: the compiled-in 80-knot window; otherwise knots of dynamic LDS (64 bytes each)
    uint32_t chunk = 1; // candidates per workgroup
};

// knots the fp64 kernels' window holds for a table whose widest frame touches `max_span` knots at one delay
inline uint32_t cap64_for(float max_span) {
    uint32_t need = (uint32_t)std::ceil(std::max(max_span, 0.f)) + 1u;
    need = (need + 15u) / 16u * 16u;
    return std::min(std::max(need, kPlanWinStatic), kPlanCap64Max);
}
// ... and what the launches of a problem whose largest frame has `n_all` tracks actually use
inline uint32_t cap64_used(uint32_t cap64, uint32_t n_all, bool force_big) {
    if (n_all <= 256u && !force_big && cap64 > kPlanCap64SmallMax) return kPlanWinStatic;
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

// The fp32 window of an LMedS launch (PreSync sweep or GuessMotion's search).
//   span        widest frame of the table, knots touched at one delay
//   step_knots  distance between neighbouring candidate delays (0: one candidate per workgroup)
//   chunk_want  candidates per workgroup the launch would like (<= 32)
//   small       the one-wave kernels (frames of up to 256 tracks); wg_max: most workgroups per CU the kernel runs at
//   fixed_lds   static LDS of the dynamic-window instantiation, lds_per_cu the CU's LDS
//   legacy      rounds 1-3: never a dynamic window (RSSYNC_FORCE_GENERAL_SPLINE)
// -> cap == 0: the compiled-in window (if the chunk's frames do not fit it, their workgroups take the general path)
inline WinPlan plan_window(double span, double step_knots, uint32_t chunk_want, bool small, int wg_max, uint32_t fixed_lds,
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
        const uint32_t cap_t = std::min(2048u, ((uint32_t)share - fixed_lds) / 64u / 4u * 4u);
        const uint32_t f = plan_fit((double)cap_t, span, step_knots, chunk_want);
        if (f < min_chunk) continue;
        const double need = span + 1.0 + (step_knots > 0 ? (f - 1) * step_knots : 0.0);
        const uint32_t cap = std::min(cap_t, ((uint32_t)std::ceil(need) + 3u) / 4u * 4u + 4u);
        // one wave per frame (K2s): a window of more than kPlanSmallWinMax knots costs more in resident waves than the
        // frame's few hundred coefficient fetches cost from L2 (profiles/r4_gyro_rate_sweep.json: 98 x 61 x 130 x 200
        // candidates, 2 kHz: 1.39 ms with a 112-knot window against 1.56 on the general path; 4 kHz: 1.98 ms with 228
        // knots against 1.54) -- keep the general path there
        if (small && cap > kPlanSmallWinMax) break;
        w.chunk = f;
        w.cap = cap;
        return w;
    }
    return w; // does not fit any LDS share: the general path
}

} // namespace rs
