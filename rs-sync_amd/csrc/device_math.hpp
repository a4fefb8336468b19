// device_math.hpp -- per-ray arithmetic of the rs-sync hot path in fp32.
//
// These are the functions the gfx950 kernels inline (kernels/*.hpp).  They
// are marked RS_HD so that tests/ can also compile them with g++ and compare
// them with the fp64 oracle on the CPU before anything is launched on a GPU;
// the product only ever runs them on the device.
//
// What each function restates (reference = VladimirP1/rs-sync, src/):
//   sample_pair      inline_utils.hpp:13-17 + core_private.cpp:41-43 (seeded)
//   spline_locate    core_support/minispline.cpp:48-53 (index/branch selection)
//   quat_at          ndspline.cpp:21-35 -> minispline.cpp:48-64 (value + derivative)
//   residual_row     core/core_private.cpp:19-28 (one row of P) and its d/d-delay
//   loss terms       core_private.cpp:99-123
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define RS_HD __host__ __device__ __forceinline__
#else
#define RS_HD inline
#endif

namespace rs {

// fast reciprocal / reciprocal square root: v_rcp_f32 / v_rsq_f32 on the device (about 1 ulp)
RS_HD float rcp_fast(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
RS_HD float rsqrt_fast(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}

struct f3 { float x, y, z; };
struct alignas(16) f4 { float x, y, z, w; };

RS_HD f3 cross(f3 a, f3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RS_HD float dot(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
RS_HD float dot4(f4 a, f4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
RS_HD f3 scale(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
RS_HD f3 add(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

// ---- hypothesis sampler, identical integer arithmetic to oracle/ora_sample_pair ----
RS_HD uint64_t sm64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
RS_HD void sample_pair(uint64_t seed, int64_t frame, uint32_t stream, uint32_t h, uint32_t n,
                       uint32_t& i0, uint32_t& i1) {
    uint64_t z = sm64(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)frame);
    z = sm64(z ^ (((uint64_t)stream << 32) | (uint64_t)h));
    uint32_t lo = (uint32_t)z, hi = (uint32_t)(z >> 32);
    uint32_t a = (uint32_t)(((uint64_t)lo * n) >> 32);
    uint32_t j = (uint32_t)(((uint64_t)hi * (n - 1)) >> 32);
    i0 = a;
    i1 = j + (j >= a ? 1u : 0u);
}

// ---- spline ----
// A spline parameter is carried as  x = idx + f  with an integer knot idx and
// a fraction f in [0,1): the host splits (ts - start) * fs into a per-frame
// integer base knot plus a per-ray fp32 offset t, and delay * fs into an
// integer kd plus a fraction fd, so no absolute time is ever rounded to fp32.
struct Knot {
    int ci;    // coefficient row to use, always in [0, n-1]
    float h;   // local parameter
    bool quad; // true on the two extrapolation branches (cubic term dropped)
};

RS_HD Knot spline_locate(float t, int base, float fd, int n) {
    float fl = floorf(t);
    float f = (t - fl) + fd;
    int idx = base + (int)fl;
    if (f >= 1.f) { f -= 1.f; idx += 1; }
    Knot k;
    if (idx < 0) { // minispline.cpp:52  x < idx(=0): quadratic from knot 0, h = x
        k.ci = 0; k.h = (float)idx + f; k.quad = true;
    } else if (idx > n - 1 || (idx == n - 1 && f > 0.f)) {
        // minispline.cpp:53  x > n-1: quadratic from the last knot; idx is clamped to n,
        // so h = x - (n-1) inside the last interval and x - n beyond it (reference quirk)
        int idc = idx < n ? idx : n;
        k.ci = n - 1; k.h = (float)(idx - idc) + f; k.quad = true;
    } else { // minispline.cpp:54
        k.ci = idx; k.h = f; k.quad = false;
    }
    return k;
}

// Same as spline_locate when the caller knows (for a whole workgroup) that every parameter falls
// strictly inside the knots, 0 <= idx <= n-2: no extrapolation branch can be taken.
RS_HD Knot spline_locate_interior(float t, int base, float fd) {
    float fl = floorf(t);
    float f = (t - fl) + fd;
    int idx = base + (int)fl;
    if (f >= 1.f) { f -= 1.f; idx += 1; }
    return Knot{idx, f, false};
}

// coefficient row = {y, b, c, d}, each an f4 over the quaternion components [w,x,y,z]
RS_HD f4 horner(f4 y, f4 b, f4 c, f4 d, float h) {
    return {fmaf(fmaf(fmaf(d.x, h, c.x), h, b.x), h, y.x), fmaf(fmaf(fmaf(d.y, h, c.y), h, b.y), h, y.y),
            fmaf(fmaf(fmaf(d.z, h, c.z), h, b.z), h, y.z), fmaf(fmaf(fmaf(d.w, h, c.w), h, b.w), h, y.w)};
}
RS_HD f4 horner_deriv(f4 b, f4 c, f4 d, float h) { // minispline.cpp:57-64
    return {fmaf(fmaf(3.f * d.x, h, 2.f * c.x), h, b.x), fmaf(fmaf(3.f * d.y, h, 2.f * c.y), h, b.y),
            fmaf(fmaf(3.f * d.z, h, 2.f * c.z), h, b.z), fmaf(fmaf(3.f * d.w, h, 2.f * c.w), h, b.w)};
}

// R(q/|q|)^T v for a quaternion q = (w, u) of squared norm n2, without normalising q first:
//   v + (2 / n2) (u x (u x v) - w (u x v))
// == vec(conj(qn) (0,v) qn) with qn = q/|q|, i.e. quat_rotate_point(quat_conj(qn), v)
// (quat.cpp:45-47 after arma::normalise, core_private.cpp:24-27).  n2 == 0 leaves v unchanged.
RS_HD f3 rotate_inv(f4 q, float two_over_n2, f3 v) {
    f3 u = {q.y, q.z, q.w};
    f3 t = cross(u, v);
    f3 t2 = cross(u, t);
    return {fmaf(two_over_n2, t2.x - q.x * t.x, v.x), fmaf(two_over_n2, t2.y - q.x * t.y, v.y),
            fmaf(two_over_n2, t2.z - q.x * t.z, v.z)};
}

// One end of a ray pair: rotated ray r = R(S(x)/|S(x)|)^T ray and, if DERIV,
// dr/dx = r x W with W = vec(2 conj(S) S' / |S|^2) (ndspline.cpp:45-49).
template <bool DERIV>
RS_HD void rotate_ray(f4 y, f4 b, f4 c, f4 d, Knot kn, f3 ray, f3& r, f3& dr) {
    if (kn.quad) d = {0.f, 0.f, 0.f, 0.f};
    f4 q = horner(y, b, c, d, kn.h);
    float n2 = dot4(q, q);
    float s = (n2 > 0.f) ? 2.f * rcp_fast(n2) : 0.f;
    r = rotate_inv(q, s, ray);
    if (DERIV) {
        f4 dq = horner_deriv(b, c, d, kn.h);
        f3 u = {q.y, q.z, q.w}, du = {dq.y, dq.z, dq.w};
        f3 uxdu = cross(u, du);
        f3 W = {s * (q.x * du.x - dq.x * u.x - uxdu.x), s * (q.x * du.y - dq.x * u.y - uxdu.y),
                s * (q.x * du.z - dq.x * u.z - uxdu.z)};
        dr = cross(r, W);
    }
}

// ---- robust loss terms (core_private.cpp:99-110,117-123) ----
// u = (P.M)^2 / s with s = |M|^2 / k^2.  Returns log1p(u) and, through the
// out-parameters, the weights of the closed-form gradients.
// log1p for u >= 0 in ~15 instructions (libm's log1pf is ~100 on the device): with w = fl(1 + u)
// and the exact rounding error c = (w - 1) - u,  log1p(u) = log(w) - c / w + O(c^2).
// Relative error stays at the fp32 rounding level down to u = 0 (w == 1 gives back u itself).
RS_HD float log1p_pos(float u) {
    const float w = 1.0f + u;
    const float c = (w - 1.0f) - u;
    return logf(w) - c * rcp_fast(w);
}
// Same with the hardware logarithm (v_log_f32, about 1 ulp of log2) in place of libm's logf: 7
// instructions.  Used only for the PreSync cost, which is compared between candidates, never
// differentiated or line-searched.
RS_HD float log1p_pos_fast(float u) {
    const float w = 1.0f + u;
    const float c = (w - 1.0f) - u;
#if defined(__HIP_DEVICE_COMPILE__)
    return fmaf(__builtin_amdgcn_logf(w), 0.69314718055994531f, -c * rcp_fast(w));
#else
    return log2f(w) * 0.69314718055994531f - c * rcp_fast(w);
#endif
}
RS_HD float loss_term(float pm, float inv_s) { return log1p_pos(pm * pm * inv_s); }

// fp64 log1p(u) and 1/(1+u) for u >= 0 in one go, for the motion optimiser's objective and
// gradient (core_private.cpp:99-114 in closed form).  libm's log1p plus an IEEE division are
// ~150 instructions per row on the device; this is ~50 and accurate to a few ulp:
//   w = 1 + u (its rounding error wl is carried to first order),  w = 2^e m,  m in [sqrt(1/2), sqrt(2)),
//   log(m) = 2 atanh(s),  s = (m - 1)/(m + 1),  |s| <= 0.1716  (odd series in s up to s^19),
//   log1p(u) = e ln2 + log(m) + wl / w.
// Both 1/(m+1) and 1/m (hence 1/w) come from ONE reciprocal of m (m + 1), refined by two
// Newton steps from the hardware seed.
RS_HD double log1p_rcp_f64(double u, double* rc) {
    const double w = 1.0 + u;
    const double wl = u - (w - 1.0);
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_amdgcn_frexp_mant(w); // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(w);
#else
    int e;
    double m = frexp(w, &e);
#endif
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double m1 = m + 1.0;
    const double p = m * m1;
#if defined(__HIP_DEVICE_COMPILE__)
    double ip = __builtin_amdgcn_rcp(p);
#else
    double ip = (double)(1.0f / (float)p); // a seed of comparable quality to the hardware's
#endif
    ip = fma(fma(-p, ip, 1.0), ip, ip);
    ip = fma(fma(-p, ip, 1.0), ip, ip);
    const double inv_m = ip * m1, inv_m1 = ip * m;
    const double sft = (m - 1.0) * inv_m1;
    const double z = sft * sft;
    double q = 1.0 / 19.0;
    q = fma(q, z, 1.0 / 17.0);
    q = fma(q, z, 1.0 / 15.0);
    q = fma(q, z, 1.0 / 13.0);
    q = fma(q, z, 1.0 / 11.0);
    q = fma(q, z, 1.0 / 9.0);
    q = fma(q, z, 1.0 / 7.0);
    q = fma(q, z, 1.0 / 5.0);
    q = fma(q, z, 1.0 / 3.0);
    const double logm = fma(sft + sft, q * z, sft + sft);
    const double r = ldexp(inv_m, -e); // 1 / w
    *rc = r;
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, logm + fma(ed, 1.90821492927058770002e-10, wl * r));
}

} // namespace rs
