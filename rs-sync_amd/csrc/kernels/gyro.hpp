// gyro.hpp -- the gyro side of the problem on the device: angular rates -> orientations (a scan of
// quaternion products), orientations -> the uniform integer-microsecond grid, grid -> natural-spline table.
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
//
// Reference: core_testcode.cpp:36-52 (optdata_fill_gyro), core_private.cpp:142-190 (the timestamped
// setter), minispline.cpp:3-46 (spline coefficients).  The arithmetic per sample is gyro_math.hpp's; what is
// specific to the device is how the two sequential recurrences are cut:
//   * integration q_i = normalise(dq_i q_{i-1}) is a scan under the quaternion product: a workgroup takes a
//     segment of kScanSegment samples, every thread a contiguous chunk of it (sequential inside the chunk, with the
//     reference's normalisation per step), a 1024-wide scan of the chunk products in LDS, then the chunk again
//     from its prefix; with more than one segment the segment totals are scanned the same way by one workgroup
//     and a last pass multiplies every later segment by the product of the segments before it;
//   * the spline's tridiagonal solve has data-independent pivots (tabulated, gyro_math.hpp) and a
//     recurrence factor of 2 - sqrt(3) = 0.268 per row in both sweeps: what a row sees of a row k places away
//     is below 0.268^k, so every thread solves a short run of rows after a warm-up of kSplineWarm rows started
//     from zero (0.268^64 = 2.5e-37: the warm-up's start cannot be seen in fp64), all runs in parallel.
#pragma once

namespace {

constexpr uint32_t kNoIndex = 0xFFFFFFFFu;

struct GyroStatus {
    uint32_t out_of_order; // smallest i with ts[i-1] > ts[i], kNoIndex if none
    uint32_t bad_input;    // a non-finite timestamp or rate
    uint32_t bad_knot;     // a non-finite knot after interpolation
    uint32_t pad;
};

__global__ void gyro_status_reset_kernel(GyroStatus* st) {
    st->out_of_order = kNoIndex;
    st->bad_input = 0;
    st->bad_knot = 0;
    st->pad = 0;
}

struct GyroRatesParams {
    const double* ts;    // [n] seconds
    const double* rates; // [n][3] rad/s
    int64_t* us;         // [n] out: whole microseconds (core_testcode.cpp:48-50)
    double* dq;          // [n][4] out: rotation over (t_i - t_{i-1}); identity at 0
    GyroStatus* st;
    uint32_t n;
    int32_t axis[3];     // output axis c reads input axis axis[c] ...
    double sign[3];      // ... times sign[c] (telemetry-parser's orientation string)
};

__device__ __forceinline__ bool finite64(double v) { return fabs(v) <= 1.79769313486231570e308; }

__global__ __launch_bounds__(256) void gyro_rates_kernel(GyroRatesParams p) {
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const double t = p.ts[i];
    const double r[3] = {p.rates[3 * (size_t)i], p.rates[3 * (size_t)i + 1], p.rates[3 * (size_t)i + 2]};
    if (!(finite64(t) && finite64(r[0]) && finite64(r[1]) && finite64(r[2]))) atomicOr(&p.st->bad_input, 1u);
    const int64_t us = (int64_t)(t * 1000000);
    p.us[i] = us;
    double d[4] = {1., 0., 0., 0.};
    if (i > 0) {
        const double tp = p.ts[i - 1];
        if ((int64_t)(tp * 1000000) > us) atomicMin(&p.st->out_of_order, i);
        const double dt = t - tp;
        const double w[3] = {r[p.axis[0]] * p.sign[0] * dt, r[p.axis[1]] * p.sign[1] * dt, r[p.axis[2]] * p.sign[2] * dt};
        rs::gyro_delta(w, d);
    }
    for (int c = 0; c < 4; ++c) p.dq[4 * (size_t)i + c] = d[c];
}

__global__ __launch_bounds__(256) void gyro_order_kernel(const int64_t* ts, const double* quats, uint32_t n, GyroStatus* st) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i > 0 && ts[i - 1] > ts[i]) atomicMin(&st->out_of_order, i);
}

// q_0 = dq_0 (identity), q_i = normalise(dq_i * q_{i-1}) within each segment of `seg` samples (one workgroup of
// kScanThreads per segment); total[b] = the product over segment b (may be null)
constexpr int kScanThreads = 1024;
constexpr uint32_t kScanSegment = 32 * kScanThreads;

__global__ __launch_bounds__(kScanThreads) void gyro_scan_kernel(const double* __restrict__ dq_all, double* __restrict__ q_all, uint32_t n_all,
                                                                 uint32_t seg, double* __restrict__ total) {
    __shared__ double s[kScanThreads][4];
    const uint32_t t = threadIdx.x;
    const size_t first = (size_t)blockIdx.x * seg;
    const uint32_t n = n_all - first < seg ? (uint32_t)(n_all - first) : seg;
    const double* dq = dq_all + 4 * first;
    double* q = q_all + 4 * first;
    const uint32_t chunk = (n + kScanThreads - 1) / kScanThreads;
    const uint32_t lo = t * chunk < n ? t * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    double acc[4] = {1., 0., 0., 0.};
    for (uint32_t i = lo; i < hi; ++i) {
        const double d[4] = {dq[4 * (size_t)i], dq[4 * (size_t)i + 1], dq[4 * (size_t)i + 2], dq[4 * (size_t)i + 3]};
        rs::quat_mul_norm(d, acc);
    }
    for (int c = 0; c < 4; ++c) s[t][c] = acc[c];
    __syncthreads();
    // inclusive scan of the chunk products; the later chunk multiplies from the left
    for (uint32_t off = 1; off < (uint32_t)kScanThreads; off <<= 1) {
        double a[4] = {1., 0., 0., 0.};
        if (t >= off)
            for (int c = 0; c < 4; ++c) a[c] = s[t - off][c];
        __syncthreads();
        if (t >= off) {
            rs::quat_mul_norm(acc, a); // a <- normalise(acc * a)
            for (int c = 0; c < 4; ++c) { acc[c] = a[c]; s[t][c] = a[c]; }
        }
        __syncthreads();
    }
    double cur[4] = {1., 0., 0., 0.};
    if (t > 0)
        for (int c = 0; c < 4; ++c) cur[c] = s[t - 1][c];
    for (uint32_t i = lo; i < hi; ++i) {
        const double d[4] = {dq[4 * (size_t)i], dq[4 * (size_t)i + 1], dq[4 * (size_t)i + 2], dq[4 * (size_t)i + 3]};
        rs::quat_mul_norm(d, cur);
        for (int c = 0; c < 4; ++c) q[4 * (size_t)i + c] = cur[c];
    }
    if (total && t == kScanThreads - 1)
        for (int c = 0; c < 4; ++c) total[4 * (size_t)blockIdx.x + c] = s[t][c];
}

// q_i <- normalise(q_i * P) for the samples of segment b >= 1, P = product of the segments before it
__global__ __launch_bounds__(256) void gyro_scan_fixup_kernel(double* __restrict__ q, uint32_t n, uint32_t seg, const double* __restrict__ prefix) {
    const uint32_t i = seg + blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* p = prefix + 4 * (size_t)(i / seg - 1);
    double local[4] = {q[4 * (size_t)i], q[4 * (size_t)i + 1], q[4 * (size_t)i + 2], q[4 * (size_t)i + 3]};
    double acc[4] = {p[0], p[1], p[2], p[3]};
    rs::quat_mul_norm(local, acc); // acc <- normalise(local * acc)
    for (int c = 0; c < 4; ++c) q[4 * (size_t)i + c] = acc[c];
}

struct GyroResampleParams {
    const int64_t* ts;   // [n] microseconds
    const double* quats; // [n][4]
    double* knots;       // [m][4] out
    GyroStatus* st;
    uint32_t n, m;
    uint64_t first_sample, sr_hz;
};

__global__ __launch_bounds__(256) void gyro_resample_kernel(GyroResampleParams p) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.m) return;
    double out[4];
    if (!rs::resample_knot(p.ts, p.quats, p.n, rs::grid_time_us(p.first_sample + i, p.sr_hz), out)) atomicOr(&p.st->bad_knot, 1u);
    for (int c = 0; c < 4; ++c) p.knots[4 * (size_t)i + c] = out[c];
}

// ---- spline table ---------------------------------------------------------------------------------------
constexpr uint32_t kSplineRun = 32, kSplineWarm = 64;

struct SplineParams {
    const double* knots; // [n][4]
    double* cf;          // [n][4]: forward-sweep values c'
    double* coef64;      // [n][16] out: y[4] b[4] c[4] d[4]
    float* coef32;       // [n][16] out: the same rounded once (PreSync's table)
    uint32_t n;
    rs::SplinePivots piv;
};

// thread = (run of kSplineRun rows, quaternion component)
__global__ __launch_bounds__(256) void spline_forward_kernel(SplineParams p) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t comp = tid & 3u, s = (tid >> 2) * kSplineRun;
    if (s >= p.n) return;
    const uint32_t e = s + kSplineRun < p.n ? s + kSplineRun : p.n;
    const double* y = p.knots + comp;
    uint32_t i = s > kSplineWarm + 1 ? s - kSplineWarm : 1; // rows 1 .. n-2 carry an equation
    double c_prev = 0.0;                                      // c'[0] = 0; elsewhere the warm-up start
    if (s == 0) p.cf[comp] = 0.0;
    if (e == p.n) p.cf[4 * (size_t)(p.n - 1) + comp] = 0.0;
    const uint32_t last = e < p.n - 1 ? e : p.n - 1; // exclusive
    if (i >= last) return;
    double ym = y[4 * (size_t)(i - 1)], y0 = y[4 * (size_t)i];
    for (; i < last; ++i) {
        const double yp = y[4 * (size_t)(i + 1)];
        c_prev = rs::spline_forward(ym, y0, yp, rs::spline_pivot(p.piv, i - 1), c_prev);
        if (i >= s) p.cf[4 * (size_t)i + comp] = c_prev;
        ym = y0;
        y0 = yp;
    }
}

__global__ __launch_bounds__(256) void spline_finish_kernel(SplineParams p) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t comp = tid & 3u, s = (tid >> 2) * kSplineRun;
    if (s >= p.n) return;
    const uint32_t e = s + kSplineRun < p.n ? s + kSplineRun : p.n;
    const double* y = p.knots + comp;
    const double* cf = p.cf + comp;
    // c at the row the backward sweep starts from: the last row (zero), or a forward value far enough away
    uint32_t j = e + kSplineWarm < p.n - 1 ? e + kSplineWarm : p.n - 1;
    double c_next = j == p.n - 1 ? 0.0 : cf[4 * (size_t)j];
    auto put = [&](uint32_t i, double yv, double b, double c, double d) {
        double* r64 = p.coef64 + 16 * (size_t)i + comp;
        float* r32 = p.coef32 + 16 * (size_t)i + comp;
        r64[0] = yv; r64[4] = b; r64[8] = c; r64[12] = d;
        r32[0] = (float)yv; r32[4] = (float)b; r32[8] = (float)c; r32[12] = (float)d;
    };
    if (e == p.n) { // this run owns the last knot: its coefficients continue the segment before it
        const uint32_t l = p.n - 1;
        const double c_lm = l >= 1 ? (l - 1 >= 1 ? rs::spline_backward(cf[4 * (size_t)(l - 1)], rs::spline_pivot(p.piv, l - 1), 0.0) : 0.0) : 0.0;
        double b_prev, d_prev, b, d;
        rs::spline_segment(y[4 * (size_t)(l - 1)], y[4 * (size_t)l], c_lm, 0.0, &b_prev, &d_prev);
        rs::spline_tail(b_prev, d_prev, c_lm, &b, &d);
        put(l, y[4 * (size_t)l], b, 0.0, d);
    }
    // rows j-1 .. s; row 0 is pinned to zero
    for (uint32_t i = j; i-- > s;) {
        const double c_i = i >= 1 ? rs::spline_backward(cf[4 * (size_t)i], rs::spline_pivot(p.piv, i), c_next) : 0.0;
        if (i < e && i + 1 < p.n) {
            double b, d;
            rs::spline_segment(y[4 * (size_t)i], y[4 * (size_t)(i + 1)], c_i, c_next, &b, &d);
            put(i, y[4 * (size_t)i], b, c_i, d);
        }
        c_next = c_i;
    }
}

} // namespace
