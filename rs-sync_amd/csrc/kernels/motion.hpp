// motion.hpp -- K3: per-frame motion L-BFGS
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
#pragma once

namespace {

// ---------------------------------------------------------------------------
// K3: per-frame L-BFGS on the motion vector, P resident in registers.
// Restates ens::L_BFGS as called at core_private.cpp:264-294 (MaxIterations 200,
// MinGradientNorm 1e-4, library defaults otherwise) from the published ensmallen 2.x
// algorithm (lbfgs_impl.hpp; the dependency is unpinned and not under the reference tree).
// When a line search's best step is not its last, the published LineSearch moves the iterate
// to the best step and leaves value and gradient as the last trial computed them; that is the
// default (MotionParams::reeval = 0).  reeval = 1 evaluates once more at the best step so that
// (x, f, g) stay consistent (round 1's choice; kept for comparison).  Control flow is uniform:
// every thread runs the same fp64 scalar logic on the same reduced sums.

struct MotionParams {
    const f4* rays_a;
    const f4* rays_b;
    const FrameRec* frames;
    const uint32_t* sel;
    uint32_t n_sel;
    const f4* coef;
    int n_knots;
    const int32_t* kd; // [n_grp]
    const float* fd;   // NaN = skip the group's slots
    const uint32_t* grp;
    double* M; // per selection slot
    const double* k;
    unsigned long long* stats; // [0] += iterations, [1] += evaluations, [2] += line searches with best != last step
    int reeval;                // see the header comment
    uint32_t* per_frame;       // optional [n_sel][2]: iterations, evaluations
};

constexpr int kNB = 10; // numBasis

template <int RPT>
struct MotionEval {
    f3 P[RPT];
    double (*part)[4][5]; // [2][4][5] LDS, double-buffered
    int buf;
    double k2;
    int evals;

    // loss and dL/dM at x (core_private.cpp:99-114 in closed form).  The rows of P are fp32 data,
    // but the objective is evaluated in fp64 (fp64 FMA issues at the fp32 rate on gfx950): with fp32
    // terms its noise floor sits above the optimiser's stopping thresholds and frames dither through
    // long line searches, and the slowest frame's serial chain is what the launch waits for.
    __device__ __forceinline__ double operator()(const double x[3], double g[3]) {
        const double s = (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) / k2;
        const double inv_s = 1.0 / s;
        double L = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0, gs = 0.0;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const double px = (double)P[j].x, py = (double)P[j].y, pz = (double)P[j].z;
            const double pm = fma(px, x[0], fma(py, x[1], pz * x[2]));
            const double v2 = pm * pm;
            const double u = v2 * inv_s;
            double w; // 1 / (1 + u)
            L += rs::log1p_rcp_f64(u, &w);
            const double a = w * 2.0 * pm * inv_s;
            a0 = fma(a, px, a0);
            a1 = fma(a, py, a1);
            a2 = fma(a, pz, a2);
            gs = fma(w * v2, inv_s * inv_s, gs);
        }
        double r0 = wave_sum_f64(L), r1 = wave_sum_f64(a0), r2 = wave_sum_f64(a1), r3 = wave_sum_f64(a2),
               r4 = wave_sum_f64(gs);
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            part[buf][wave][0] = r0; part[buf][wave][1] = r1; part[buf][wave][2] = r2;
            part[buf][wave][3] = r3; part[buf][wave][4] = r4;
        }
        __syncthreads();
        double t[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) t[q] = part[buf][0][q] + part[buf][1][q] + part[buf][2][q] + part[buf][3][q];
        buf ^= 1;
        ++evals;
        const double tt = t[4] * 2.0 / k2;
        g[0] = t[1] - tt * x[0];
        g[1] = t[2] - tt * x[1];
        g[2] = t[3] - tt * x[2];
        return t[0];
    }
};

__device__ __forceinline__ double dot3d(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

template <int RPT>
__global__ __launch_bounds__(kBlock, 4) void opt_motion_kernel(MotionParams p) {
    __shared__ f4 s_win[4 * kWinMax];
    __shared__ double s_part[2][4][5];
    __shared__ double s_S[kNB][3], s_Y[kNB][3];
    // two-loop scratch: every thread writes the same values and reads them back itself;
    // the barrier inside each evaluation separates one iteration's use from the next
    __shared__ double s_rho[kNB], s_alpha[kNB];
    const int tid = threadIdx.x;
    const uint32_t sf = blockIdx.x;
    const uint32_t fi = p.sel[sf];
    const FrameRec fr = p.frames[fi];
    const uint32_t N = fr.n;
    const uint32_t grp = p.grp ? p.grp[sf] : 0u;
    const int kd = p.kd[grp];
    const float fd = p.fd[grp];
    if (fd != fd) return; // this window is not being optimised in this call (workgroup-uniform)

    Spline sp;
    sp.g = p.coef;
    sp.n = p.n_knots;
    stage_window(sp, s_win, fr.base_knot + (int)floorf(fr.tmin) + kd, fr.base_knot + (int)floorf(fr.tmax) + kd + 1);
    __syncthreads();

    MotionEval<RPT> ev;
    ev.part = s_part;
    ev.buf = 0;
    ev.evals = 0;
    const double kk = p.k[sf];
    ev.k2 = kk * kk;
    const int base = fr.base_knot + kd;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        uint32_t row = j * kBlock + tid;
        f3 P = f3{0, 0, 0}, dP;
        if (row < N) {
            if (sp.path == kPathInterior) residual_row<false, kPathInterior>(sp, p.rays_a[fr.off + row], p.rays_b[fr.off + row], base, fd, P, dP);
            else residual_row<false, kPathGlobal>(sp, p.rays_a[fr.off + row], p.rays_b[fr.off + row], base, fd, P, dP);
        }
        ev.P[j] = P; // zero rows contribute log1p(0) = 0 and no gradient
    }

    const int maxIterations = 200;       // core_private.cpp:265
    const double minGradientNorm = 1e-4; // core_private.cpp:266
    const double armijo = 1e-4, wolfe = 0.9, factr = 1e-15, minStep = 1e-20, maxStep = 1e20;
    const int maxLineSearchTrials = 50;

    double x[3] = {p.M[3 * sf], p.M[3 * sf + 1], p.M[3 * sf + 2]};
    double g[3], oldx[3], oldg[3], dir[3];
    double fval = ev(x, g);
    int it = 0, best_not_last = 0;
    for (; it != maxIterations; ++it) {
        const double prev = fval;
        if (sqrt(dot3d(g, g)) < minGradientNorm) break;
        if (fval != fval) break;
        double scale;
        if (it > 0) {
            const int pp = (it - 1) % kNB;
            const double yy = dot3d(s_Y[pp], s_Y[pp]);
            scale = dot3d(s_S[pp], s_Y[pp]) / ((yy >= 1e-10) ? yy : 1.0);
        } else {
            const double gn = sqrt(dot3d(g, g));
            scale = (gn >= 1e-5) ? 1.0 / gn : 1.0;
        }
        if (scale == 0.0 || scale != scale) break;
        // two-loop recursion
        dir[0] = g[0]; dir[1] = g[1]; dir[2] = g[2];
        const int limit = (kNB > it) ? 0 : (it - kNB);
#pragma unroll 1
        for (int i = it; i != limit; --i) {
            const int tp = (i + (kNB - 1)) % kNB;
            const double r = 1.0 / dot3d(s_Y[tp], s_S[tp]);
            const double al = r * dot3d(s_S[tp], dir);
            s_rho[it - i] = r; // it - i in [0, kNB)
            s_alpha[it - i] = al;
            dir[0] -= al * s_Y[tp][0]; dir[1] -= al * s_Y[tp][1]; dir[2] -= al * s_Y[tp][2];
        }
        dir[0] *= scale; dir[1] *= scale; dir[2] *= scale;
#pragma unroll 1
        for (int i = limit; i < it; ++i) {
            const int tp = i % kNB;
            const double beta = s_rho[it - i - 1] * dot3d(s_Y[tp], dir);
            const double cf = s_alpha[it - i - 1] - beta;
            dir[0] += cf * s_S[tp][0]; dir[1] += cf * s_S[tp][1]; dir[2] += cf * s_S[tp][2];
        }
        dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2];
        oldx[0] = x[0]; oldx[1] = x[1]; oldx[2] = x[2];
        oldg[0] = g[0]; oldg[1] = g[1]; oldg[2] = g[2];
        // line search
        const double dg0 = dot3d(g, dir);
        if (dg0 > 0.0) break;
        const double f0 = fval, lin = armijo * dg0;
        double step = 1.0, bestStep = 1.0, bestObj = 1.79769313486231570e308, lastStep = 1.0;
        int trials = 0;
        for (;;) {
            double xn[3] = {x[0] + step * dir[0], x[1] + step * dir[1], x[2] + step * dir[2]};
            fval = ev(xn, g);
            lastStep = step;
            if (fval < bestObj) { bestStep = step; bestObj = fval; }
            ++trials;
            double width;
            if (fval > f0 + step * lin) {
                width = 0.5;
            } else {
                const double dg = dot3d(g, dir);
                if (dg < wolfe * dg0) width = 2.1;
                else if (dg > -wolfe * dg0) width = 0.5;
                else break;
            }
            if (step < minStep || step > maxStep || trials >= maxLineSearchTrials) break;
            step *= width;
        }
        x[0] += bestStep * dir[0]; x[1] += bestStep * dir[1]; x[2] += bestStep * dir[2];
        if (bestStep != lastStep) {
            ++best_not_last;
            if (p.reeval) fval = ev(x, g);
        }
        if (bestStep == 0.0) break;
        const double denom = fmax(fmax(fabs(prev), fabs(fval)), 1.0);
        if ((prev - fval) / denom <= factr) break;
        const int op = it % kNB;
        __syncthreads(); // every thread has finished reading the history for this iteration
        if (tid == 0) {
            for (int c = 0; c < 3; ++c) { s_S[op][c] = x[c] - oldx[c]; s_Y[op][c] = g[c] - oldg[c]; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        p.M[3 * sf] = x[0]; p.M[3 * sf + 1] = x[1]; p.M[3 * sf + 2] = x[2];
        if (p.stats) {
            atomicAdd(&p.stats[0], (unsigned long long)it);
            atomicAdd(&p.stats[1], (unsigned long long)ev.evals);
            atomicAdd(&p.stats[2], (unsigned long long)best_not_last);
        }
        if (p.per_frame) {
            p.per_frame[2 * sf] = (uint32_t)it;
            p.per_frame[2 * sf + 1] = (uint32_t)ev.evals;
        }
    }
}

} // namespace
