// sync_math.hpp -- the fp64 arithmetic of the Sync kernels that is NOT per-ray geometry (that is
// device_math.hpp): the row terms of the robust loss and of its gradients, and the restated ens::L_BFGS.
//
// Written once, for the gfx950 kernels (kernels/sync64.hpp) and for the CPU stand-in of the device ABI that
// the tests link (tests/cpu_device/rship_cpu.cpp).  Contraction is off and every fused multiply-add is
// spelled fma(): with the sums over rows taken in the kernels' association, the stand-in reproduces the
// device's Sync bit for bit (tests/test_gpu_bitexact.py) -- which is what pins the claim that the remaining
// device-vs-oracle differences on noisy data are reassociation and nothing else.
//
// Reference (VladimirP1/rs-sync, src/):
//   loss_row      core/core_private.cpp:117-123 (loss), :92-115 (its Jacobians, in closed form)
//   motion_row    core_private.cpp:99-114 (dL/dM in closed form)
//   lbfgs3        ens::L_BFGS as called at core_private.cpp:264-294 (third party, unpinned: restated from the
//                 published ensmallen 2.x algorithm, lbfgs_impl.hpp)
#pragma once

#include "device_math.hpp"

#if defined(__HIPCC__)
#define RS_DEV __device__ __forceinline__
#else
#define RS_DEV inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace rs {

RS_DEV double dot3(const double* a, const double* b) { return fma(a[0], b[0], fma(a[1], b[1], a[2] * b[2])); }

RS_DEV double clamp_k64(double k) { return (k < 10.0) ? 10.0 : ((1000.0 < k) ? 1000.0 : k); } // inline_utils.hpp:50

// r = (P.M) k / |M|  (core_private.cpp:120)  ->  u = (P.M)^2 inv_s;  no-translation variant: u = |P|^2 k^2
RS_DEV double loss_inv_s(bool simple, double kk, d3 Mv) { return simple ? kk * kk : kk * kk / dot(Mv, Mv); }

// K1, one row: L += log1p(u) (core_private.cpp:121-122) and, with GRAD, the analytic d/d-delay term
//   1/(1+u) (2 pm / s) (dP/dd . M)   (replaces the central difference of :96-97,112); dP is per knot here,
// the caller scales the sum by the sample rate.
template <bool GRAD, bool SIMPLE>
RS_DEV void loss_row(d3 P, d3 dP, d3 Mv, double inv_s, double& L, double& G) {
    double w;
    if (SIMPLE) {
        const double u = dot(P, P) * inv_s;
        L += log1p_rcp_f64(u, &w);
        if (GRAD) G = fma(w * 2.0 * inv_s, dot(P, dP), G);
    } else {
        const double pm = dot(P, Mv);
        const double u = pm * pm * inv_s;
        L += log1p_rcp_f64(u, &w);
        if (GRAD) G = fma(w * 2.0 * pm * inv_s, dot(dP, Mv), G);
    }
}

// K3, one row: loss and the row's part of t = sum_j w_j (2 P_j.x / s) P_j at the motion estimate x
RS_DEV void motion_row(d3 P, const double x[3], double inv_s, double& L, double& a0, double& a1, double& a2) {
    const double pm = fma(P.x, x[0], fma(P.y, x[1], P.z * x[2]));
    const double v2 = pm * pm;
    const double u = v2 * inv_s;
    double w; // 1 / (1 + u)
    L += log1p_rcp_f64(u, &w);
    const double a = w * 2.0 * pm * inv_s;
    a0 = fma(a, P.x, a0);
    a1 = fma(a, P.y, a1);
    a2 = fma(a, P.z, a2);
}
// s = |x|^2 / k^2 of core_private.cpp:100-104 as its reciprocal k^2 / |x|^2, and 1 / |x|^2 for motion_finish: two
// independent divisions at the head of an evaluation.  (Round 2 divided three times in a row -- |x|^2 / k^2, its
// reciprocal, and (x.t) / |x|^2 after the sums; a division is a dozen fp64 instructions of the ~300 of a small
// frame's evaluation.)
RS_DEV double motion_inv_s(const double x[3], double k2, double* inv_xx_out) {
    const double xx = dot3(x, x);
    *inv_xx_out = 1.0 / xx;
    return k2 / xx;
}
// after the sums t = {L, t_x, t_y, t_z}: the loss does not depend on |x|, so its gradient is t without its
// component along x  (x.t = 2 sum_j w_j u_j is exactly the sum the chain rule's second term needs)
RS_DEV double motion_finish(const double x[3], double inv_xx, const double t[4], double g[3]) {
    const double tt = dot3(x, t + 1) * inv_xx;
    g[0] = fma(-tt, x[0], t[1]);
    g[1] = fma(-tt, x[1], t[2]);
    g[2] = fma(-tt, x[2], t[3]);
    return t[0];
}

constexpr int kLbfgsBasis = 10; // numBasis (ens::L_BFGS default)

// ens::L_BFGS on a 3-vector: MaxIterations = max_iterations (200, core_private.cpp:265), MinGradientNorm 1e-4
// (:266), library defaults otherwise (Armijo 1e-4, Wolfe 0.9, factr 1e-15, 50 line-search trials, step in
// [1e-20, 1e20]).  When a line search's best step is not its last, the published LineSearch moves the iterate to
// the best step and leaves value and gradient as the last trial computed them (reeval = 0); reeval = 1 evaluates
// once more at the best step.
//   ev(x, g)  -> value, gradient in g (the caller's sums over rows)
//   h         history of the last kLbfgsBasis (s, y) pairs: h.S(i), h.Y(i) -> const double*, h.inv_ys(i) = 1 / (y.s),
//             h.store(i, s, y) (the device keeps it in LDS and brackets the store with barriers), and the
//             two-loop scratch h.rho(i), h.alpha(i) -> double&
// Returns the number of iterations; x is the end point.
template <class Ev, class Hist>
RS_DEV int lbfgs3(Ev& ev, Hist& h, double x[3], int max_iterations, int reeval, int* best_not_last_out) {
    const double minGradientNorm = 1e-4;
    const double armijo = 1e-4, wolfe = 0.9, factr = 1e-15, minStep = 1e-20, maxStep = 1e20;
    const int maxLineSearchTrials = 50;
    constexpr int NB = kLbfgsBasis;

    double g[3], oldx[3], oldg[3], dir[3];
    double fval = ev(x, g);
    int it = 0, best_not_last = 0;
    for (; it != max_iterations; ++it) {
        const double prev = fval;
        if (sqrt(dot3(g, g)) < minGradientNorm) break;
        if (fval != fval) break;
        double scale;
        if (it > 0) {
            const int pp = (it - 1) % NB;
            const double yy = dot3(h.Y(pp), h.Y(pp));
            scale = dot3(h.S(pp), h.Y(pp)) / ((yy >= 1e-10) ? yy : 1.0);
        } else {
            const double gn = sqrt(dot3(g, g));
            scale = (gn >= 1e-5) ? 1.0 / gn : 1.0;
        }
        if (scale == 0.0 || scale != scale) break;
        // two-loop recursion
        dir[0] = g[0]; dir[1] = g[1]; dir[2] = g[2];
        const int limit = (NB > it) ? 0 : (it - NB);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
        for (int i = it; i != limit; --i) {
            const int tp = (i + (NB - 1)) % NB;
            const double r = h.inv_ys(tp); // 1 / (y . s), computed when the pair was stored
            const double al = r * dot3(h.S(tp), dir);
            h.rho(it - i) = r; // it - i in [0, NB)
            h.alpha(it - i) = al;
            const double* y = h.Y(tp);
            dir[0] = fma(-al, y[0], dir[0]); dir[1] = fma(-al, y[1], dir[1]); dir[2] = fma(-al, y[2], dir[2]);
        }
        dir[0] *= scale; dir[1] *= scale; dir[2] *= scale;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
        for (int i = limit; i < it; ++i) {
            const int tp = i % NB;
            const double beta = h.rho(it - i - 1) * dot3(h.Y(tp), dir);
            const double cf = h.alpha(it - i - 1) - beta;
            const double* s = h.S(tp);
            dir[0] = fma(cf, s[0], dir[0]); dir[1] = fma(cf, s[1], dir[1]); dir[2] = fma(cf, s[2], dir[2]);
        }
        dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2];
        oldx[0] = x[0]; oldx[1] = x[1]; oldx[2] = x[2];
        oldg[0] = g[0]; oldg[1] = g[1]; oldg[2] = g[2];
        // line search
        const double dg0 = dot3(g, dir);
        if (dg0 > 0.0) break;
        const double f0 = fval, lin = armijo * dg0;
        double step = 1.0, bestStep = 1.0, bestObj = 1.79769313486231570e308, lastStep = 1.0;
        int trials = 0;
        for (;;) {
            const double xn[3] = {fma(step, dir[0], x[0]), fma(step, dir[1], x[1]), fma(step, dir[2], x[2])};
            fval = ev(xn, g);
            lastStep = step;
            if (fval < bestObj) { bestStep = step; bestObj = fval; }
            ++trials;
            double width;
            if (fval > fma(step, lin, f0)) {
                width = 0.5;
            } else {
                const double dg = dot3(g, dir);
                if (dg < wolfe * dg0) width = 2.1;
                else if (dg > -wolfe * dg0) width = 0.5;
                else break;
            }
            if (step < minStep || step > maxStep || trials >= maxLineSearchTrials) break;
            step *= width;
        }
        x[0] = fma(bestStep, dir[0], x[0]); x[1] = fma(bestStep, dir[1], x[1]); x[2] = fma(bestStep, dir[2], x[2]);
        if (bestStep != lastStep) {
            ++best_not_last;
            if (reeval) fval = ev(x, g);
        }
        if (bestStep == 0.0) break;
        const double denom = fmax(fmax(fabs(prev), fabs(fval)), 1.0);
        if ((prev - fval) / denom <= factr) break;
        double sv[3], yv[3];
        for (int c = 0; c < 3; ++c) { sv[c] = x[c] - oldx[c]; yv[c] = g[c] - oldg[c]; }
        h.store(it % NB, sv, yv);
    }
    if (best_not_last_out) *best_not_last_out = best_not_last;
    return it;
}

} // namespace rs

// (End of the contraction-off region.  NOT "restored" to a guessed state: only the HIP translation unit, whose
// default is -ffp-contract=fast-honor-pragmas and whose fp32 kernels are written for it, switches fusion back on.  Any
// other clang translation unit that includes this header -- a clang build of the CPU stand-in compiled with
// -ffp-contract=off -- keeps contraction OFF to its end: the bit-exactness this header exists for.)
#if defined(__clang__) && defined(__HIPCC__)
#pragma clang fp contract(fast)
#endif
