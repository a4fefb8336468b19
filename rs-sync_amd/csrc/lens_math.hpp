// lens_math.hpp -- pixel -> (unit ray, row time) for one tracked point, in fp64.
//
// The step upstream of SetTrackResult in the reference driver:
//   undistort_point   src/core_testcode.cpp:63-95  (Newton inverse of the 4-coefficient fisheye
//                     polynomial, 9 iterations from pi/4, halving back into (0, pi/2))
//   row_time          :144-145  ts = frame_time + readout * (pixel_y / image_rows)
//   unit_ray          :147-152  normalise([x_u, y_u, 1])
// RS_HD like device_math.hpp: the kernel (rays_from_pixels_kernel) inlines these, the CPU test
// double compiles the same text with g++.  Contraction is switched off so that device, test
// double and oracle perform the same IEEE operations in the same order; what remains different
// between them is the libm behind tan/cos (< 1 ulp of fp64, gone after rounding to fp32).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RS_LHD __host__ __device__ inline
#else
#define RS_LHD inline
#endif

namespace rs {

struct Lens { // core_testcode.cpp:55-61
    double ro;             // rolling-shutter readout time of a frame, seconds
    double fx, fy, cx, cy; // pinhole part, pixels
    double k1, k2, k3, k4; // fisheye polynomial
};

RS_LHD void undistort_point(const Lens& lens, double px, double py, double* ux, double* uy) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double kHalfPi = 3.14159265358979323846 / 2., kQuarterPi = 3.14159265358979323846 / 4.;
    if (sqrt(px * px + py * py) < 1e-8) { // :64 (the norm of the PIXEL position, as written)
        *ux = 0;
        *uy = 0;
        return;
    }
    const double x_ = (px - lens.cx) / lens.fx;
    const double y_ = (py - lens.cy) / lens.fy;
    const double theta_ = sqrt(x_ * x_ + y_ * y_);
    double theta = kQuarterPi;
    for (int i = 0; i < 9; ++i) { // :73
        const double theta2 = theta * theta, theta3 = theta2 * theta, theta4 = theta2 * theta2,
                     theta5 = theta2 * theta3, theta6 = theta3 * theta3, theta7 = theta3 * theta4,
                     theta8 = theta4 * theta4, theta9 = theta4 * theta5;
        const double cur_theta_ = theta + lens.k1 * theta3 + lens.k2 * theta5 + lens.k3 * theta7 + lens.k4 * theta9;
        // the factor on k4 is 8 in the reference (:80), not 9; kept: it only slows the iteration
        const double cur_dTheta_ = 1 + 3 * lens.k1 * theta2 + 5 * lens.k2 * theta4 + 7 * lens.k3 * theta6 +
                                   8 * lens.k4 * theta8;
        const double error = cur_theta_ - theta_;
        const double dthetaDtheta_ = 1. / cur_dTheta_;
        double new_theta = theta - error * dthetaDtheta_;
        // :85-87 `while`: the midpoint sequence reaches theta (inside the interval) exactly after
        // at most ~1100 halvings of a finite fp64 distance, so the bound never binds -- it is
        // there so that a device wave always leaves the loop.
        for (int guard = 0; guard < 1200 && (new_theta >= kHalfPi || new_theta <= 0.); ++guard)
            new_theta = (new_theta + theta) / 2.;
        theta = new_theta;
    }
    const double r = tan(theta);
    const double inv_cos_theta = 1. / cos(theta);
    const double s = (theta_ < 1e-9) ? inv_cos_theta : r / theta_;
    *ux = x_ * s;
    *uy = y_ * s;
}

// one end of a track: pixel -> unit ray (:147-152) and absolute row time (:144-145)
RS_LHD void pixel_to_ray(const Lens& lens, double px, double py, double frame_time_s, double image_rows,
                         double* ray, double* ts) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    double ux, uy;
    undistort_point(lens, px, py, &ux, &uy);
    const double n = sqrt(ux * ux + uy * uy + 1.);
    const double inv = n > 0 ? n : 1.; // arma::normalise leaves a zero vector alone; cannot happen with z = 1
    ray[0] = ux / inv;
    ray[1] = uy / inv;
    ray[2] = 1. / inv;
    *ts = frame_time_s + lens.ro * (py / image_rows);
}

// spline parameter of a row time relative to the frame's integer base knot
// (sync_problem.cpp pack_frames: (ts - start) * fs - base, core_private.cpp:19-20 without the delay)
RS_LHD double knot_offset(double ts, double start, double fs, double base) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return (ts - start) * fs - base;
}

} // namespace rs
