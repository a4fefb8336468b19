/*
 * rssync_oracle_driver.c -- CPU restatement of the steps the reference DRIVER performs just
 * upstream of the ISyncProblem calls, plain C, IEEE double.
 *
 * TEST INFRASTRUCTURE ONLY (see rssync_oracle.h).  PARITY UNPINNED: the reference has no tests
 * for these either; they are pinned by independent checks in tests/test_oracle_driver.py (forward
 * fisheye model round trip, scipy Rotation for the integration).
 *
 * Restated from src/core_testcode.cpp:
 *   ora_undistort_point   :63-95    Newton inverse of the fisheye polynomial
 *   ora_pixels_to_tracks  :135-152  undistort, normalise([x, y, 1]), row time
 *   ora_integrate_gyro    :36-51    q_0 = 1, q_i = normalise(quat_from_aa(w_i (t_i - t_{i-1})) q_{i-1}),
 *                                   timestamps -> int64 microseconds by truncation
 * and src/core_support/quat.cpp:5-17 (quat_from_aa), :33-38 (quat_prod).
 * ora_orient_rates is telemetry-parser's orientation string (third party, not under
 * /root/reference; convention as recorded in SURVEY.md 8(c): position = output axis, letter =
 * input axis, upper case = +, lower case = -).
 */
#define _GNU_SOURCE /* M_PI under -std=c11 */
#include "rssync_oracle.h"

#include <math.h>
#include <string.h>

void ora_undistort_point(const ora_lens* lens, double px, double py, double out[2]) {
    static const double eps = 1e-9;
    if (sqrt(px * px + py * py) < 1e-8) { /* :64 */
        out[0] = 0;
        out[1] = 0;
        return;
    }
    const double x_ = (px - lens->cx) / lens->fx;
    const double y_ = (py - lens->cy) / lens->fy;
    const double theta_ = sqrt(x_ * x_ + y_ * y_);
    double theta = M_PI / 4.;
    for (int i = 0; i < 9; ++i) { /* kNumIterations, :66 */
        double theta2 = theta * theta, theta3 = theta2 * theta, theta4 = theta2 * theta2,
               theta5 = theta2 * theta3, theta6 = theta3 * theta3, theta7 = theta3 * theta4,
               theta8 = theta4 * theta4, theta9 = theta4 * theta5;
        double cur_theta_ = theta + lens->k1 * theta3 + lens->k2 * theta5 + lens->k3 * theta7 + lens->k4 * theta9;
        double cur_dTheta_ = 1 + 3 * lens->k1 * theta2 + 5 * lens->k2 * theta4 + 7 * lens->k3 * theta6 +
                             8 * lens->k4 * theta8; /* 8, as written at :80 */
        double error = cur_theta_ - theta_;
        double dthetaDtheta_ = 1. / cur_dTheta_;
        double new_theta = theta - error * dthetaDtheta_;
        /* :85-87; an infinite new_theta would spin forever in the reference -- bounded here */
        for (int guard = 0; guard < 1200 && (new_theta >= M_PI / 2. || new_theta <= 0.); ++guard)
            new_theta = (new_theta + theta) / 2.;
        theta = new_theta;
    }
    double r = tan(theta);
    double inv_cos_theta = 1. / cos(theta);
    double s = (theta_ < eps) ? inv_cos_theta : r / theta_;
    out[0] = x_ * s;
    out[1] = y_ * s;
}

void ora_pixels_to_tracks(const ora_lens* lens, double time_a, double time_b, double rows, const double* px_a,
                          const double* px_b, size_t n, double* ts_a, double* ts_b, double* rays_a, double* rays_b) {
    for (size_t i = 0; i < n; ++i) {
        double a[2], b[2];
        ora_undistort_point(lens, px_a[2 * i], px_a[2 * i + 1], a); /* :141 */
        ora_undistort_point(lens, px_b[2 * i], px_b[2 * i + 1], b);
        ts_a[i] = time_a + lens->ro * (px_a[2 * i + 1] / rows); /* :144 */
        ts_b[i] = time_b + lens->ro * (px_b[2 * i + 1] / rows); /* :145 */
        double na = sqrt(a[0] * a[0] + a[1] * a[1] + 1.), nb = sqrt(b[0] * b[0] + b[1] * b[1] + 1.);
        rays_a[3 * i] = a[0] / na; rays_a[3 * i + 1] = a[1] / na; rays_a[3 * i + 2] = 1. / na; /* :147-151 */
        rays_b[3 * i] = b[0] / nb; rays_b[3 * i + 1] = b[1] / nb; rays_b[3 * i + 2] = 1. / nb;
    }
}

void ora_quat_from_aa(const double aa[3], double out[4]) { /* quat.cpp:5-17 */
    const double theta_squared = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (theta_squared > 0.) {
        const double theta = sqrt(theta_squared);
        const double half_theta = theta * 0.5;
        const double k = sin(half_theta) / theta;
        out[0] = cos(half_theta); out[1] = aa[0] * k; out[2] = aa[1] * k; out[3] = aa[2] * k;
    } else {
        out[0] = 1.; out[1] = aa[0] * 0.5; out[2] = aa[1] * 0.5; out[3] = aa[2] * 0.5;
    }
}

static void quat_prod4(const double p[4], const double q[4], double o[4]) { /* quat.cpp:33-38 */
    o[0] = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3];
    o[1] = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
    o[2] = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
    o[3] = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
}

void ora_integrate_gyro(const double* timestamps_s, const double* rates, size_t n, double* quats, int64_t* ts_us) {
    if (!n) return;
    quats[0] = 1; quats[1] = 0; quats[2] = 0; quats[3] = 0; /* :41 */
    for (size_t i = 1; i < n; ++i) {
        const double dt = timestamps_s[i] - timestamps_s[i - 1];
        const double aa[3] = {rates[3 * i] * dt, rates[3 * i + 1] * dt, rates[3 * i + 2] * dt};
        double q[4], o[4];
        ora_quat_from_aa(aa, q); /* :43 */
        quat_prod4(q, quats + 4 * (i - 1), o);
        double nn = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
        if (nn == 0) nn = 1;
        for (int c = 0; c < 4; ++c) quats[4 * i + c] = o[c] / nn; /* :44 */
    }
    for (size_t i = 0; i < n; ++i) ts_us[i] = (int64_t)(timestamps_s[i] * 1000000); /* :48-50 */
}

int ora_orient_rates(const double* rates, size_t n, const char* orientation, double* out) {
    int axis[3];
    double sign[3];
    if (!orientation || strlen(orientation) != 3) return 1;
    for (int c = 0; c < 3; ++c) {
        const char ch = orientation[c];
        const char lo = (char)(ch | 0x20);
        if (lo < 'x' || lo > 'z') return 1;
        axis[c] = lo - 'x';
        sign[c] = (ch == lo) ? -1.0 : 1.0;
    }
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) out[3 * i + c] = rates[3 * i + axis[c]] * sign[c];
    return 0;
}
