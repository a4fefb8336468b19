// lens_math.hpp -- pixel -> (unit ray, row time) for one tracked point, in fp64.
//
// The step upstream of SetTrackResult in the reference driver:
//   undistort_point   src/core_testcode.cpp:63-95  (Newton inverse of the 4-coefficient fisheye
//                     polynomial, 9 iterations from pi/4, halving back into (0, pi/2); written
//                     here in Horner form, see the function)
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

// Inverse of the fisheye model  rd = th * (1 + k1 th^2 + k2 th^4 + k3 th^6 + k4 th^8)  by the
// reference's fixed schedule (core_testcode.cpp:63-95): nine Newton steps from pi/4, each step
// pulled back towards the previous iterate by halving while it lies outside (0, pi/2).  Both
// polynomials are evaluated by Horner's rule in q = th^2 (the reference expands the powers
// th^2 .. th^9 one by one; the two forms agree to a few fp64 ulps, far below the fp32 result).
// The slope polynomial keeps the reference's coefficient 8 on k4 (:80; the exact derivative has
// 9): it changes the path of the iteration, not its fixed point.
RS_LHD void undistort_point(const Lens& lens, double px, double py, double* ux, double* uy) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double kPi = 3.14159265358979323846;
    *ux = 0;
    *uy = 0;
    if (sqrt(px * px + py * py) < 1e-8) return; // :64 tests the PIXEL position, not the centred one
    const double xn = (px - lens.cx) / lens.fx, yn = (py - lens.cy) / lens.fy;
    const double rd = sqrt(xn * xn + yn * yn); // distorted angle the model must reproduce
    double th = kPi / 4.;
    for (int step = 0; step < 9; ++step) {
        const double q = th * th;
        const double model = th * (1. + q * (lens.k1 + q * (lens.k2 + q * (lens.k3 + q * lens.k4))));
        const double slope = 1. + q * (3. * lens.k1 + q * (5. * lens.k2 + q * (7. * lens.k3 + q * (8. * lens.k4))));
        double next = th - (model - rd) / slope;
        // the midpoints approach th, which lies inside the interval, and reach it exactly after at
        // most ~1100 halvings of a finite fp64 distance: the bound only guarantees that a wave leaves
        for (int guard = 0; guard < 1200 && (next <= 0. || next >= kPi / 2.); ++guard) next = 0.5 * (next + th);
        th = next;
    }
    // (x, y) scaled from distorted radius rd to tan(th); at the image centre the ratio is taken as 1 / cos (:91-93)
    const double gain = (rd < 1e-9) ? 1. / cos(th) : tan(th) / rd;
    *ux = xn * gain;
    *uy = yn * gain;
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
