// gyro_math.hpp -- the per-sample arithmetic of the gyro pipeline (rates -> orientations -> uniform grid ->
// spline table), shared by the device kernels (kernels/gyro.hpp) and by the tests' CPU stand-in for the
// device (tests/cpu_device/rship_cpu.cpp), which runs the same formulas one sample after the other.
//
// What each function restates (reference = VladimirP1/rs-sync, src/):
//   gyro_delta       core_support/quat.cpp:5-17 via core_testcode.cpp:41-46 (rotation by rate * dt)
//   quat_mul_norm    core_testcode.cpp:44-45 (q_i = normalise(dq_i * q_{i-1}))
//   quat_slerp       core_support/quat.cpp:55-74
//   grid_*           core/core_private.cpp:147-160 (integer micro-hertz / microsecond grid)
//   spline_*         core_support/minispline.cpp:3-46 as a Thomas recurrence on unit-spaced knots
#pragma once

#include <stdint.h>
#include <math.h>

#include "../../include/rssync_hip.h"
#include "device_math.hpp"

namespace rs {

// rotation by the angular rate w (already multiplied by dt): [w,x,y,z]
RS_HD void gyro_delta(const double w[3], double d[4]) {
#pragma clang fp contract(off)
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (th2 > 0.) {
        const double th = sqrt(th2), half = th * 0.5, kk = sin(half) / th;
        d[0] = cos(half); d[1] = w[0] * kk; d[2] = w[1] * kk; d[3] = w[2] * kk;
    } else {
        d[0] = 1.; d[1] = w[0] * 0.5; d[2] = w[1] * 0.5; d[3] = w[2] * 0.5;
    }
}

// o = d * p (Hamilton product, [w,x,y,z])
RS_HD void quat_mul(const double d[4], const double p[4], double o[4]) {
#pragma clang fp contract(off)
    const double o0 = d[0] * p[0] - d[1] * p[1] - d[2] * p[2] - d[3] * p[3];
    const double o1 = d[0] * p[1] + d[1] * p[0] + d[2] * p[3] - d[3] * p[2];
    const double o2 = d[0] * p[2] - d[1] * p[3] + d[2] * p[0] + d[3] * p[1];
    const double o3 = d[0] * p[3] + d[1] * p[2] - d[2] * p[1] + d[3] * p[0];
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}

RS_HD void quat_normalise(double o[4]) {
#pragma clang fp contract(off)
    double nn = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    if (nn == 0) nn = 1;
    for (int c = 0; c < 4; ++c) o[c] = o[c] / nn;
}

// one integration step: q <- normalise(d * q)
RS_HD void quat_mul_norm(const double d[4], double q[4]) {
    double o[4];
    quat_mul(d, q, o);
    quat_normalise(o);
    for (int c = 0; c < 4; ++c) q[c] = o[c];
}

RS_HD void quat_slerp(const double* p, const double* q_in, double t, double* out) {
#pragma clang fp contract(off)
    double q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]};
    double d = p[0] * q[0] + p[1] * q[1] + p[2] * q[2] + p[3] * q[3];
    if (d < 0) {
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
        d = p[0] * q[0] + p[1] * q[1] + p[2] * q[2] + p[3] * q[3];
    }
    const double theta = acos(d); // unclamped: NaN falls through to the lerp branch
    double m1 = 1 - t, m2 = t;
    if (theta > 1e-9) {
        const double st = sin(theta);
        m1 = sin((1 - t) * theta) / st;
        m2 = sin(t * theta) / st;
    }
    for (int i = 0; i < 4; ++i) out[i] = m1 * p[i] + m2 * q[i];
}

// time of grid sample `sample` in microseconds at `sr_hz` (core_private.cpp:153-154, uint64 arithmetic)
RS_HD uint64_t grid_time_us(uint64_t sample, uint64_t sr_hz) { return 1000000ULL * sample / sr_hz; }

// first index in ts[0..count) whose (unsigned) value is >= t (std::lower_bound, core_private.cpp:166)
RS_HD uint32_t lower_bound_us(const int64_t* ts, uint32_t count, uint64_t t) {
    uint32_t lo = 0, n = count;
    while (n > 0) {
        const uint32_t half = n >> 1;
        if ((uint64_t)ts[lo + half] < t) { lo += half + 1; n -= half + 1; }
        else n = half;
    }
    return lo;
}

// one grid knot (core_private.cpp:162-176); returns false when the result is not finite
RS_HD bool resample_knot(const int64_t* ts, const double* quats, uint32_t count, uint64_t t, double* out) {
#pragma clang fp contract(off)
    uint32_t idx = lower_bound_us(ts, count, t);
    if (idx >= count) idx = count - 1; // only with timestamps out of order, which the caller reports
    if (idx > 0) {
        const double u = 1. * (double)(t - (uint64_t)ts[idx - 1]) / (double)(ts[idx] - ts[idx - 1]);
        quat_slerp(quats + 4 * (size_t)(idx - 1), quats + 4 * (size_t)idx, u, out);
    } else {
        for (int c = 0; c < 4; ++c) out[c] = quats[c];
    }
    bool ok = true;
    for (int c = 0; c < 4; ++c) ok = ok && (fabs(out[c]) <= 1.79769313486231570e308);
    return ok;
}

// Natural cubic spline on unit-spaced knots, c[0] = c[n-1] = 0, interior rows
//   c[i-1]/3 + 4 c[i]/3 + c[i+1]/3 = y[i+1] - 2 y[i] + y[i-1].
// The pivots of the forward sweep do not depend on the data: cp[0] = 0, cp[i] = (1/3) / (4/3 - cp[i-1]/3),
// stationary in fp64 after a few dozen rows; kSplinePivots of them are tabulated and the last is used from there on.
constexpr int kSplinePivots = 64;
struct SplinePivots { double cp[kSplinePivots]; };

RS_HD double spline_next_pivot(double cp_prev) {
#pragma clang fp contract(off)
    return (1.0 / 3.0) / (4.0 / 3.0 - cp_prev / 3.0);
}
RS_HD double spline_pivot(const SplinePivots& t, uint32_t i) { return t.cp[i < (uint32_t)kSplinePivots ? i : kSplinePivots - 1]; }

// forward row i (1 <= i <= n-2): c'[i] from c'[i-1]
RS_HD double spline_forward(double ym, double y0, double yp, double cp_prev, double c_prev) {
#pragma clang fp contract(off)
    const double rhs = yp - 2.0 * y0 + ym;
    const double denom = 4.0 / 3.0 - cp_prev / 3.0;
    return (rhs - c_prev / 3.0) / denom;
}
// backward row i: c[i] = c'[i] - cp[i] c[i+1]
RS_HD double spline_backward(double cf, double cp_i, double c_next) {
#pragma clang fp contract(off)
    return cf - cp_i * c_next;
}
// b and d of the segment that starts at knot i (minispline.cpp:36-41)
RS_HD void spline_segment(double y0, double y1, double c0, double c1, double* b, double* d) {
#pragma clang fp contract(off)
    *d = (c1 - c0) / 3.0;
    *b = (y1 - y0) - (2.0 * c0 + c1) / 3.0;
}
// x / 3.0, correctly rounded, in three instructions instead of an IEEE division (~35 on gfx950): q = x * RN(1/3) is within
// one ulp, r = x - 3 q is exact in an fma, and q + r * RN(1/3) rounds to RN(x / 3) (Markstein's theorem for a correctly
// rounded reciprocal; 3's significand is not all ones).  Exact for every double whose quotient is a normal number; the
// spline's coefficients are O(1) and their differences far above 1e-290.  tests/test_gpu_gyro.py compares it with the
// division on the device and on the host.
RS_HD double div3_exact(double x) {
#pragma clang fp contract(off)
    const double third = 0.33333333333333331482961625624739;
    const double q = x * third;
    const double r = fma(-3.0, q, x);
    return fma(r, third, q);
}
// spline_segment with that division: THE SAME BITS (compact fp64 windows, kernels/sync64.hpp: only y and c of a knot are
// kept in LDS, b and d are rebuilt per fetch)
RS_HD void spline_segment_fast(double y0, double y1, double c0, double c1, double* b, double* d) {
#pragma clang fp contract(off)
    *d = div3_exact(c1 - c0);
    *b = (y1 - y0) - div3_exact(2.0 * c0 + c1);
}
// the last knot's coefficients, used by the extrapolation (minispline.cpp:43-44)
RS_HD void spline_tail(double b_prev, double d_prev, double c_prev, double* b, double* d) {
#pragma clang fp contract(off)
    *d = 0.0;
    *b = 3.0 * d_prev + 2.0 * c_prev + b_prev;
}

constexpr uint32_t kMaxKnots = 1u << 26; // 8.6 GB of fp64 table: two days of a 400 Hz gyro

// The uniform grid of the timestamped setter (core_private.cpp:147-160) from the first and last timestamp:
// rate rounded to 50 Hz, first grid sample by a truncating division, samples while their time is below the
// last timestamp.  Everything in the reference's integer types; the count in closed form instead of a loop.
// Returns a RSHIP_GYRO_* status.
inline int grid_of(int64_t first, int64_t last, uint32_t count, uint32_t max_knots, rship_gyro_result* g) {
    constexpr uint64_t kUhzInHz = 1000000ULL, kUsInSec = 1000000ULL;
    g->fs = 0; g->start = 0; g->n_knots = 0; g->first_sample = 0;
    if (last == first) return RSHIP_GYRO_BAD_RATE; // the reference divides by zero here
    const uint64_t actual_sr_uhz = kUhzInHz * kUsInSec * (uint64_t)count / (uint64_t)(last - first);
    const int rounded_sr_hz = (int)(round((double)actual_sr_uhz / 50. / (double)kUhzInHz) * 50);
    if (rounded_sr_hz <= 0) return RSHIP_GYRO_BAD_RATE;
    if (first < 0 || last < 0 || last > (int64_t)(1LL << 50)) return RSHIP_GYRO_TOO_LARGE; // unsigned wrap-around in the reference
    const uint64_t sr = (uint64_t)rounded_sr_hz;
    const int first_sample = (int)((uint64_t)(first * rounded_sr_hz) / kUsInSec);
    // samples S with floor(1e6 S / sr) < last  <=>  1e6 S < last sr  <=>  S < ceil(last sr / 1e6)
    const uint64_t end_sample = ((uint64_t)last * sr + kUsInSec - 1) / kUsInSec;
    const uint64_t m = end_sample > (uint64_t)first_sample ? end_sample - (uint64_t)first_sample : 0;
    g->fs = 1. * rounded_sr_hz;
    g->first_sample = (uint64_t)first_sample;
    if (m < 2) return RSHIP_GYRO_SHORT_GRID;
    if (m > max_knots) return RSHIP_GYRO_TOO_LARGE;
    g->n_knots = (uint32_t)m;
    g->start = 1. * (double)(kUsInSec * (uint64_t)first_sample / sr) / (double)kUsInSec;
    return RSHIP_GYRO_OK;
}

} // namespace rs
