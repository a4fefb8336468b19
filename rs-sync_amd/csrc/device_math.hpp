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

// Contraction is OFF for everything in this header (restored at its end): every fused multiply-add is spelled
// fma_t / fmaf / fma, so the bits do not depend on a compiler's choice of what to fuse.
#if defined(__clang__)
#pragma clang fp contract(off)
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

// 3- and 4-vectors over fp32 (PreSync kernel) and fp64 (Sync kernels; the reference's arithmetic is
// IEEE double throughout, core_private.cpp).  Everything below is written once for both.
template <typename T> struct v3 { T x, y, z; };
template <typename T> struct alignas(16) v4 { T x, y, z, w; };
using f3 = v3<float>;
using f4 = v4<float>;
using d3 = v3<double>;
using d4 = v4<double>;

RS_HD float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
RS_HD double fma_t(double a, double b, double c) { return fma(a, b, c); }
RS_HD float floor_t(float x) { return floorf(x); }
RS_HD double floor_t(double x) { return floor(x); }
// 2 / n2 for the rotation: the hardware reciprocal in fp32 (about 1 ulp), a true division in fp64
RS_HD float two_over(float n2) { return 2.f * rcp_fast(n2); }
RS_HD double two_over(double n2) { return 2.0 / n2; }

// Every multiply-add below is written out (fma_t) and the functions are compiled with contraction off: the
// value of an expression then does not depend on what a compiler chooses to fuse, and the CPU stand-in of the
// tests (g++ -ffp-contract=off) computes the same bits as the device (tests/test_gpu_bitexact.py).
template <typename T> RS_HD v3<T> cross(v3<T> a, v3<T> b) {
    return {fma_t(a.y, b.z, -(a.z * b.y)), fma_t(a.z, b.x, -(a.x * b.z)), fma_t(a.x, b.y, -(a.y * b.x))};
}
template <typename T> RS_HD T dot(v3<T> a, v3<T> b) { return fma_t(a.x, b.x, fma_t(a.y, b.y, a.z * b.z)); }
template <typename T> RS_HD T dot4(v4<T> a, v4<T> b) { return fma_t(a.x, b.x, fma_t(a.y, b.y, fma_t(a.z, b.z, a.w * b.w))); }
template <typename T> RS_HD v3<T> scale(v3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> RS_HD v3<T> add(v3<T> a, v3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

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
// a fraction f in [0,1): the frame table holds an integer base knot, every ray its offset t from
// it ((ts - start) * fs - base), and delay * fs is split into an integer kd plus a fraction fd,
// so no absolute time is ever rounded to the working precision.
template <typename T>
struct KnotT {
    int ci;    // coefficient row to use, always in [0, n-1]
    T h;       // local parameter
    bool quad; // true on the two extrapolation branches (cubic term dropped)
};
using Knot = KnotT<float>;

template <typename T>
RS_HD KnotT<T> spline_locate(T t, int base, T fd, int n) {
    T fl = floor_t(t);
    T f = (t - fl) + fd;
    int idx = base + (int)fl;
    if (f >= (T)1) { f -= (T)1; idx += 1; }
    KnotT<T> k;
    if (idx < 0) { // minispline.cpp:52  x < idx(=0): quadratic from knot 0, h = x
        k.ci = 0; k.h = (T)idx + f; k.quad = true;
    } else if (idx > n - 1 || (idx == n - 1 && f > (T)0)) {
        // minispline.cpp:53  x > n-1: quadratic from the last knot; idx is clamped to n,
        // so h = x - (n-1) inside the last interval and x - n beyond it (reference quirk)
        int idc = idx < n ? idx : n;
        k.ci = n - 1; k.h = (T)(idx - idc) + f; k.quad = true;
    } else { // minispline.cpp:54
        k.ci = idx; k.h = f; k.quad = false;
    }
    return k;
}

// Same as spline_locate when the caller knows (for a whole workgroup) that every parameter falls
// strictly inside the knots, 0 <= idx <= n-2: no extrapolation branch can be taken.
template <typename T>
RS_HD KnotT<T> spline_locate_interior(T t, int base, T fd) {
    T fl = floor_t(t);
    T f = (t - fl) + fd;
    int idx = base + (int)fl;
    if (f >= (T)1) { f -= (T)1; idx += 1; }
    return KnotT<T>{idx, f, false};
}

// coefficient row = {y, b, c, d}, each a 4-vector over the quaternion components [w,x,y,z]
template <typename T>
RS_HD v4<T> horner(v4<T> y, v4<T> b, v4<T> c, v4<T> d, T h) {
    return {fma_t(fma_t(fma_t(d.x, h, c.x), h, b.x), h, y.x), fma_t(fma_t(fma_t(d.y, h, c.y), h, b.y), h, y.y),
            fma_t(fma_t(fma_t(d.z, h, c.z), h, b.z), h, y.z), fma_t(fma_t(fma_t(d.w, h, c.w), h, b.w), h, y.w)};
}
template <typename T>
RS_HD v4<T> horner_deriv(v4<T> b, v4<T> c, v4<T> d, T h) { // minispline.cpp:57-64
    return {fma_t(fma_t((T)3 * d.x, h, (T)2 * c.x), h, b.x), fma_t(fma_t((T)3 * d.y, h, (T)2 * c.y), h, b.y),
            fma_t(fma_t((T)3 * d.z, h, (T)2 * c.z), h, b.z), fma_t(fma_t((T)3 * d.w, h, (T)2 * c.w), h, b.w)};
}

// R(q/|q|)^T v for a quaternion q = (w, u) of squared norm n2, without normalising q first:
//   v + (2 / n2) (u x (u x v) - w (u x v))
// == vec(conj(qn) (0,v) qn) with qn = q/|q|, i.e. quat_rotate_point(quat_conj(qn), v)
// (quat.cpp:45-47 after arma::normalise, core_private.cpp:24-27).  n2 == 0 leaves v unchanged.
template <typename T>
RS_HD v3<T> rotate_inv(v4<T> q, T two_over_n2, v3<T> v) {
    v3<T> u = {q.y, q.z, q.w};
    v3<T> t = cross(u, v);
    v3<T> t2 = cross(u, t);
    return {fma_t(two_over_n2, fma_t(-q.x, t.x, t2.x), v.x), fma_t(two_over_n2, fma_t(-q.x, t.y, t2.y), v.y),
            fma_t(two_over_n2, fma_t(-q.x, t.z, t2.z), v.z)};
}

// One end of a ray pair: rotated ray r = R(S(x)/|S(x)|)^T ray and, if DERIV,
// dr/dx = r x W with W = vec(2 conj(S) S' / |S|^2) (ndspline.cpp:45-49).
template <bool DERIV, typename T>
RS_HD void rotate_ray(v4<T> y, v4<T> b, v4<T> c, v4<T> d, KnotT<T> kn, v3<T> ray, v3<T>& r, v3<T>& dr) {
    if (kn.quad) d = {(T)0, (T)0, (T)0, (T)0};
    v4<T> q = horner(y, b, c, d, kn.h);
    T n2 = dot4(q, q);
    T s = (n2 > (T)0) ? two_over(n2) : (T)0;
    r = rotate_inv(q, s, ray);
    if (DERIV) {
        v4<T> dq = horner_deriv(b, c, d, kn.h);
        v3<T> u = {q.y, q.z, q.w}, du = {dq.y, dq.z, dq.w};
        v3<T> uxdu = cross(u, du);
        v3<T> W = {s * (fma_t(q.x, du.x, -(dq.x * u.x)) - uxdu.x), s * (fma_t(q.x, du.y, -(dq.x * u.y)) - uxdu.y),
                   s * (fma_t(q.x, du.z, -(dq.x * u.z)) - uxdu.z)};
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
// The correction c / w only matters while w is close to 1 (c <= ulp(w) / 2, and log(w) grows): 1 / w is taken as
// max(2 - w, 0), exact to (w - 1)^2 where it matters -- the result stays within 9e-8 relative of log1p over
// 1e-12 .. 1e6 (3e-8 with a true reciprocal; v_log_f32 itself is good to ~1e-7) -- and the row loses one of its
// three quarter-rate instructions (round 3 A/B: -0.15 ms per PreSync launch).
RS_HD float log1p_pos_fast(float u) {
    const float w = 1.0f + u;
    const float c = (w - 1.0f) - u;
#if defined(__HIP_DEVICE_COMPILE__)
    return fmaf(__builtin_amdgcn_logf(w), 0.69314718055994531f, -c * fmaxf(2.0f - w, 0.f));
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
// Both 1/(m+1) and 1/m (hence 1/w) come from ONE reciprocal of m (m + 1) -- an IEEE division, correctly
// rounded on the device and on the host alike.  (Round 2 refined the hardware's v_rcp_f64 seed with two Newton
// steps: five instructions fewer, but the last bit then depended on a seed no CPU can reproduce, and the
// device could not be compared bit for bit with anything.)
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
    const double ip = 1.0 / p;
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

// (End of the contraction-off region.  NOT "restored" to a guessed state: only the HIP translation unit, whose
// default is -ffp-contract=fast-honor-pragmas and whose fp32 kernels are written for it, switches fusion back on.  Any
// other clang translation unit that includes this header -- a clang build of the CPU stand-in compiled with
// -ffp-contract=off -- keeps contraction OFF to its end: the bit-exactness this header exists for.)
#if defined(__clang__) && defined(__HIPCC__)
#pragma clang fp contract(fast)
#endif
