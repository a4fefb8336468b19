// rows64.hpp -- one row of the residual matrix in fp64: the fp64 spline window, the four fp64 ray streams, residual_row64.
// Part of the single HIP translation unit rssync_kernels.hip (included there before lmeds.hpp).
//
// Until round 6 this lived at the top of sync64.hpp (K1 / K3 are its main users).  It moved in front of the LMedS kernels
// because PreSync's sweep now recomputes the rows of NEAR-STATIC frames in fp64 (lmeds.hpp: "fp64 rows"): the reference
// computes rows, norms and the safe_normalize decision in double (core_private.cpp:19-28,45-46, inline_utils.hpp:5-11),
// and a row of |P| ~ 1e-6 built from fp32 rays carries per cents of error.  The code is unchanged, bit for bit.
#pragma once

// compiled with contraction off (restored at the end of this header): what is fused is written as fma(), so that
// tests/cpu_device/rship_cpu.cpp reproduces the fp64 kernels bit for bit (sync64.hpp)
#pragma clang fp contract(off)

namespace {

using rs::d3;
using rs::d4;

// ---------------------------------------------------------------------------
// fp64 spline window in LDS: [kind][knot] of d4 (32 B), 10 KB at 80 knots.  The window lives in DYNAMIC LDS and its
// capacity is a launch parameter (Spline64::cap, >= kWinMax): the host sizes it for the problem's widest frame, so
// that gyro rates above ~1.7 kHz (a frame spans 0.044 s x rate knots) stay on the LDS paths; the stride between the
// coefficient kinds is then a run-time value -- three integer additions per fetch beside ~150 fp64 instructions.

struct Spline64 {
    const d4* __restrict__ g; // global table, 4 d4 per knot
    const d4* lds;            // [4][cap]
    int n;
    int w0, wlen;
    int path; // kPathGlobal / kPathLds64 / kPathInterior, uniform over the workgroup
    int cap;  // knots the window holds (set by the caller before staging)
    int w0b;  // run-time capacity only: the knot that maps to slot 0 for the B END's fetches (two ranges staged one after
              // the other, common.hpp: stage_window_ends); = w0 where the whole pair is staged
    // COMPACT (run-time capacity only; round 5): the window holds y and c of a knot only -- [2][cap] d4, 64 bytes per knot
    // instead of 128 -- and b, d are rebuilt per fetch from the knot and its successor with the expressions the table was
    // built with (rs::spline_segment_fast: the same bits).  The one-wave kernels stage a window per evaluation for ~260
    // fetches and every knot of it is LDS another wave of the CU cannot have; at 8 kHz a 130-track frame's two ends are
    // 182 knots: 23 KB as full records (five waves per CU: slower than reading the table from L2), 12 KB compact.  A
    // staged range then includes the knot after its last one.  Interior frames only (others read the table from L2).
    int compact;
};

// the general parameter logic (extrapolation branches) with the coefficients from the LDS window: a delay that puts a
// frame partly or wholly beyond the gyro track -- a line search's first trials are seconds away -- touches only knots
// of the CLAMPED range, which is what gets staged (round 3: those delays read the table from L2, 15-35 % slower)
constexpr int kPathLds64 = 1;

// CAP = compile-time capacity (the stride between the coefficient kinds folds into the ds_read offsets: the five-window
// loss kernel is at its register limit) or 0 = s.cap
template <int CAP = 0>
__device__ __forceinline__ void stage_window64(Spline64& s, d4* s_win, int lo, int hi) {
    const int n = s.n;
    const bool interior = lo >= 0 && hi <= n - 2;
    lo = lo < 0 ? 0 : (lo > n - 1 ? n - 1 : lo);
    hi = hi < 0 ? 0 : (hi > n - 1 ? n - 1 : hi);
    int wlen = hi - lo + 1;
    const int cap = CAP ? CAP : s.cap;
    if constexpr (CAP == 0) {
        if (s.compact) { // y and c of the knots lo .. hi + 1 (interior: hi + 1 <= n - 1 exists)
            s.w0 = lo;
            s.w0b = lo;
            s.lds = s_win;
            if (!interior || wlen + 1 > cap) { s.path = kPathGlobal; s.wlen = 0; return; }
            s.path = kPathInterior;
            s.wlen = wlen + 1;
            for (int e = threadIdx.x; e < (wlen + 1) * 2; e += blockDim.x) {
                const int knot = e >> 1, kind = e & 1;
                s_win[kind * cap + knot] = s.g[(size_t)(lo + knot) * 4 + 2 * kind];
            }
            return;
        }
    }
    s.path = wlen <= cap ? (interior ? kPathInterior : kPathLds64) : kPathGlobal;
    if (wlen > cap) wlen = cap;
    s.w0 = lo;
    s.w0b = lo;
    s.wlen = wlen;
    s.lds = s_win;
    for (int e = threadIdx.x; e < wlen * 4; e += blockDim.x) {
        int knot = e >> 2, kind = e & 3;
        s_win[kind * cap + knot] = s.g[(size_t)(lo + knot) * 4 + kind];
    }
}

template <int PATH, int CAP = 0, bool END_B = false>
__device__ __forceinline__ void fetch_coef64(const Spline64& s, int ci, d4& y, d4& b, d4& c, d4& d) {
    if (PATH == kPathGlobal) {
        const d4* p = s.g + (size_t)ci * 4;
        y = p[0]; b = p[1]; c = p[2]; d = p[3];
    } else {
        const int rel = ci - ((CAP == 0 && END_B) ? s.w0b : s.w0), cap = CAP ? CAP : s.cap;
        if constexpr (CAP == 0) {
            if (s.compact) { // b and d from this knot and the next, as spline_finish_kernel computed them (the same bits)
                y = s.lds[rel];
                c = s.lds[cap + rel];
                const d4 y1 = s.lds[rel + 1], c1 = s.lds[cap + rel + 1];
                rs::spline_segment_fast(y.x, y1.x, c.x, c1.x, &b.x, &d.x);
                rs::spline_segment_fast(y.y, y1.y, c.y, c1.y, &b.y, &d.y);
                rs::spline_segment_fast(y.z, y1.z, c.z, c1.z, &b.z, &d.z);
                rs::spline_segment_fast(y.w, y1.w, c.w, c1.w, &b.w, &d.w);
                return;
            }
        }
        y = s.lds[rel];
        b = s.lds[cap + rel];
        c = s.lds[2 * cap + rel];
        d = s.lds[3 * cap + rel];
    }
}

// the four fp64 streams of the frames: {ax,bx} {ay,by} {az,bz} {ta,tb}
struct Rays64 {
    const double2* __restrict__ q0;
    const double2* __restrict__ q1;
    const double2* __restrict__ q2;
    const double2* __restrict__ q3;
};

// one row of P = ar x br (core_private.cpp:24-28) and, if DERIV, dP/dx (x in knots), in fp64
template <bool DERIV, int PATH, int CAP = 0>
__device__ __forceinline__ void residual_row64(const Spline64& s, double2 X, double2 Y, double2 Z, double2 T, int base, double fd,
                                               d3& P, d3& dP) {
    d4 ya, ba, ca, da, yb, bb, cb, db;
    const rs::KnotT<double> ka = (PATH == kPathInterior) ? rs::spline_locate_interior(T.x, base, fd)
                                                         : rs::spline_locate(T.x, base, fd, s.n);
    fetch_coef64<PATH, CAP>(s, ka.ci, ya, ba, ca, da);
    const rs::KnotT<double> kb = (PATH == kPathInterior) ? rs::spline_locate_interior(T.y, base, fd)
                                                         : rs::spline_locate(T.y, base, fd, s.n);
    fetch_coef64<PATH, CAP, true>(s, kb.ci, yb, bb, cb, db);
    d3 ar, br, dar, dbr;
    rs::rotate_ray<DERIV>(ya, ba, ca, da, ka, d3{X.x, Y.x, Z.x}, ar, dar);
    rs::rotate_ray<DERIV>(yb, bb, cb, db, kb, d3{X.y, Y.y, Z.y}, br, dbr);
    P = rs::cross(ar, br);
    if (DERIV) dP = rs::add(rs::cross(dar, br), rs::cross(ar, dbr));
}

// (the path is uniform over the workgroup: chosen when the window was staged)
template <bool DERIV, int CAP = 0>
__device__ __forceinline__ void residual_row64_auto(const Spline64& s, double2 X, double2 Y, double2 Z, double2 T, int base, double fd,
                                                    d3& P, d3& dP) {
    if (s.path == kPathInterior) residual_row64<DERIV, kPathInterior, CAP>(s, X, Y, Z, T, base, fd, P, dP);
    else if (s.path == kPathLds64) residual_row64<DERIV, kPathLds64, CAP>(s, X, Y, Z, T, base, fd, P, dP);
    else residual_row64<DERIV, kPathGlobal, CAP>(s, X, Y, Z, T, base, fd, P, dP);
}
template <bool DERIV>
__device__ __forceinline__ void residual_row64(const Spline64& s, const Rays64& r, size_t idx, int base, double fd, d3& P,
                                               d3& dP) {
    residual_row64_auto<DERIV>(s, r.q0[idx], r.q1[idx], r.q2[idx], r.q3[idx], base, fd, P, dP);
}

// the frame's window at one delay: the two ends' knot ranges one after the other where the table knows them and the
// capacity is a run-time value (the same rule as stage_window_ends of the fp32 kernels), else the whole pair
template <int CAP = 0>
__device__ __forceinline__ void frame_window64(Spline64& sp, d4* s_win, const FrameRec& fr, int kd) {
    const int lo = fr.base_knot + (int)floor(fr.tmin64) + kd, hi = fr.base_knot + (int)floor(fr.tmax64) + kd + 1;
    if constexpr (CAP == 0) {
        const FrameKnots k = frame_knots(fr, lo, hi, kd, kd);
        const int cap = sp.cap, n = sp.n;
        const int lenA = k.a_hi - k.a_lo + 1, lenB = k.b_hi - k.b_lo + 1;
        const bool disjoint = k.b_lo > k.a_hi + 1 || k.a_lo > k.b_hi + 1;
        const bool interior = k.a_lo >= 0 && k.b_lo >= 0 && k.a_hi <= n - 2 && k.b_hi <= n - 2;
        if (sp.compact) {
            // each end's range and the knot after it: slots [0, lenA + 1) and [lenA + 1, lenA + lenB + 2)
            if (k.split && disjoint && interior && lenA + lenB + 2 <= cap && lenA + lenB + 2 < hi - lo + 2) {
                sp.path = kPathInterior;
                sp.w0 = k.a_lo;
                sp.w0b = k.b_lo - (lenA + 1);
                sp.wlen = lenA + lenB + 2;
                sp.lds = s_win;
                for (int e = threadIdx.x; e < (lenA + lenB + 2) * 2; e += blockDim.x) {
                    const int slot = e >> 1, kind = e & 1;
                    const int knot = slot <= lenA ? k.a_lo + slot : k.b_lo + (slot - lenA - 1);
                    s_win[kind * cap + slot] = sp.g[(size_t)knot * 4 + 2 * kind];
                }
                return;
            }
        } else if (k.split && disjoint && interior && lenA + lenB <= cap && lenA + lenB < hi - lo + 1) {
            sp.path = kPathInterior;
            sp.w0 = k.a_lo;
            sp.w0b = k.b_lo - lenA;
            sp.wlen = lenA + lenB;
            sp.lds = s_win;
            for (int e = threadIdx.x; e < (lenA + lenB) * 4; e += blockDim.x) {
                const int slot = e >> 2, kind = e & 3;
                const int knot = slot < lenA ? k.a_lo + slot : k.b_lo + (slot - lenA);
                s_win[kind * cap + slot] = sp.g[(size_t)knot * 4 + kind];
            }
            return;
        }
    }
    stage_window64<CAP>(sp, s_win, lo, hi);
}

// ---------------------------------------------------------------------------
// A row of P for the LMedS kernels' fp64 form: the UNIT row and |P| computed in double from the fp64 streams and the fp64
// table (general path: any parameter, coefficients from L2 -- this is the rare path, no window is staged for it), then
// rounded ONCE to fp32 for the tile.  core_private.cpp:24-28 (the row), :35-36 with inline_utils.hpp:5-11 (safe_normalize:
// a row with |P| < 1e-12 stays as it is, its "norm" for the hypothesis rule is 1).
struct Rows64Src {
    Rays64 rays;       // the frames' fp64 streams
    const d4* coef;    // fp64 spline table, 4 d4 per knot
    int n_knots;
};
struct Row64 {
    f3 n;        // unit row (or the row itself where safe_normalize leaves it)
    float nrm;   // |P| (1 for a row left alone)
    float n2;    // |P|^2 rounded to fp32 (the row watch's statistic: hypothesis()'s bound)
    bool finite;
};
__device__ __forceinline__ Row64 row64_unit(const Rows64Src& s, size_t idx, int base, double fd) {
    Spline64 sp;
    sp.g = s.coef;
    sp.lds = nullptr;
    sp.n = s.n_knots;
    sp.w0 = sp.wlen = sp.w0b = 0;
    sp.path = kPathGlobal;
    sp.cap = 0;
    sp.compact = 0;
    d3 P, dP;
    residual_row64<false, kPathGlobal>(sp, s.rays.q0[idx], s.rays.q1[idx], s.rays.q2[idx], s.rays.q3[idx], base, fd, P, dP);
    const double n2 = fma(P.z, P.z, fma(P.y, P.y, P.x * P.x));
    const double nr = sqrt(n2);             // arma::norm
    const bool tiny = nr < 1e-12;           // inline_utils.hpp:7 (a NaN is "not tiny": it is normalised into NaNs, as there)
    const double inv = tiny ? 1.0 : 1.0 / nr;
    Row64 r;
    r.n = f3{(float)(P.x * inv), (float)(P.y * inv), (float)(P.z * inv)};
    r.nrm = tiny ? 1.f : (float)nr;
    r.n2 = (float)n2;
    r.finite = fabs(n2) <= 1.79769313486231570e308; // (false for NaN)
    return r;
}

} // namespace

// (back to the translation unit's default for the fp32 headers that follow)
#pragma clang fp contract(fast)
