// loss.hpp -- K1: residual + robust loss (+ analytic d/d-delay) per frame for a batch of delays
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
#pragma once

namespace {

// ---------------------------------------------------------------------------
// K1: residual + robust loss (+ analytic d/d-delay) per frame for a batch of delays

struct LossParams {
    const f4* rays_a;
    const f4* rays_b;
    const FrameRec* frames;
    const uint32_t* sel;
    uint32_t n_sel;
    const f4* coef;
    int n_knots;
    float fs;
    const int32_t* kd; // [n_delays][n_grp]
    const float* fd;   // NaN = this group is skipped (its partial sums are written as 0)
    uint32_t n_delays;
    const uint32_t* grp;
    uint32_t n_grp;
    const double* M; // per selection slot
    const double* k;
    double* part_loss; // [n_delays][n_sel]
    double* part_grad; // [n_delays][n_sel] (GRAD)
};

// this thread's rows of one frame at one delay: sum of log1p(u) and of the d/d-delay terms
template <bool GRAD, int PATH>
__device__ __forceinline__ void loss_row(const Spline& sp, f4 A, f4 B, int base, float fd, f3 Mv, float inv_s, float& L,
                                         float& G) {
    f3 P, dP;
    residual_row<GRAD, PATH>(sp, A, B, base, fd, P, dP);
    const float pm = rs::dot(P, Mv);
    const float u = pm * pm * inv_s;
    L += rs::log1p_pos(u); // core_private.cpp:121-122
    if (GRAD) {
        // dL/dd = sum 1/(1+u) * (2 pm / s) * (dP/dd . M), dP/dd = fs * dP/dx
        const float w = rs::rcp_fast(1.f + u);
        G = fmaf(w * 2.f * pm * inv_s, rs::dot(dP, Mv), G);
    }
}

// this thread's rows of one frame at one delay, rays read from memory (single-delay launches)
template <int RPT, bool GRAD, int PATH>
__device__ __forceinline__ void loss_rows(const Spline& sp, const f4* __restrict__ rays_a,
                                          const f4* __restrict__ rays_b, uint32_t N, int base, float fd, f3 Mv,
                                          float inv_s, float& L, float& G) {
#pragma unroll 1
    for (int j = 0; j < RPT; ++j) {
        const uint32_t row = j * kBlock + threadIdx.x;
        if (row < N) loss_row<GRAD, PATH>(sp, rays_a[row], rays_b[row], base, fd, Mv, inv_s, L, G);
    }
}

// the same with the rays already in registers (batches of delays: the line search's ten trials)
template <int RPT, bool GRAD, int PATH>
__device__ __forceinline__ void loss_rows_cached(const Spline& sp, const f4 (&ra)[RPT], const f4 (&rb)[RPT], uint32_t N,
                                                 int base, float fd, f3 Mv, float inv_s, float& L, float& G) {
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const uint32_t row = j * kBlock + threadIdx.x;
        if (row < N) loss_row<GRAD, PATH>(sp, ra[j], rb[j], base, fd, Mv, inv_s, L, G);
    }
}

template <int RPT, bool GRAD>
__global__ __launch_bounds__(kBlock, loss_waves(RPT, GRAD)) void loss_kernel(LossParams p) {
    __shared__ f4 s_win[4 * kWinMax];
    __shared__ double s_red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t sf = blockIdx.x;
    const uint32_t fi = p.sel[sf];
    const FrameRec fr = p.frames[fi];
    const uint32_t N = fr.n;

    const f4* __restrict__ rays_a = p.rays_a + fr.off;
    const f4* __restrict__ rays_b = p.rays_b + fr.off;
    // A batch of delays (no gradient: the ten backtracking trials) keeps this thread's rays in
    // registers, 8 floats per row: re-reading the frame per delay made that launch bound by the
    // L2/Infinity-Cache side (2.4 GB for 268 MB of rays), not by its arithmetic.
    constexpr bool kCache = !GRAD;
    f4 ra[kCache ? RPT : 1], rb[kCache ? RPT : 1];
    if (kCache) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const uint32_t row = j * kBlock + tid;
            ra[j] = row < N ? rays_a[row] : f4{0.f, 0.f, 0.f, 0.f};
            rb[j] = row < N ? rays_b[row] : f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const uint32_t g = p.grp ? p.grp[sf] : 0u;
    const double Mx = p.M[3 * sf], My = p.M[3 * sf + 1], Mz = p.M[3 * sf + 2], kk = p.k[sf];
    const f3 Mv = f3{(float)Mx, (float)My, (float)Mz};
    // r = (P.M) k / |M|  (core_private.cpp:120)  ->  u = (P.M)^2 * inv_s
    const float inv_s = (float)(kk * kk / (Mx * Mx + My * My + Mz * Mz));

    Spline sp;
    sp.g = p.coef;
    sp.n = p.n_knots;
    for (uint32_t b = 0; b < p.n_delays; ++b) {
        const int kd = p.kd[b * p.n_grp + g];
        const float fd = p.fd[b * p.n_grp + g];
        if (fd != fd) { // group switched off for this evaluation (workgroup-uniform)
            if (tid == 0) {
                p.part_loss[(size_t)b * p.n_sel + sf] = 0.0;
                if (GRAD) p.part_grad[(size_t)b * p.n_sel + sf] = 0.0;
            }
            continue;
        }
        __syncthreads(); // window and s_red reuse
        stage_window(sp, s_win, fr.base_knot + (int)floorf(fr.tmin) + kd, fr.base_knot + (int)floorf(fr.tmax) + kd + 1);
        __syncthreads();
        const int base = fr.base_knot + kd;
        float L = 0.f, G = 0.f;
        if (kCache) {
            if (sp.path == kPathInterior) loss_rows_cached<kCache ? RPT : 1, GRAD, kPathInterior>(sp, ra, rb, N, base, fd, Mv, inv_s, L, G);
            else loss_rows_cached<kCache ? RPT : 1, GRAD, kPathGlobal>(sp, ra, rb, N, base, fd, Mv, inv_s, L, G);
        } else {
            if (sp.path == kPathInterior) loss_rows<RPT, GRAD, kPathInterior>(sp, rays_a, rays_b, N, base, fd, Mv, inv_s, L, G);
            else loss_rows<RPT, GRAD, kPathGlobal>(sp, rays_a, rays_b, N, base, fd, Mv, inv_s, L, G);
        }
        double Lw = wave_sum_f64((double)L);
        double Gw = GRAD ? wave_sum_f64((double)G) : 0.0;
        if (lane == 0) {
            s_red[0][wave] = Lw;
            s_red[1][wave] = Gw;
        }
        __syncthreads();
        if (tid == 0) {
            p.part_loss[(size_t)b * p.n_sel + sf] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
            if (GRAD)
                p.part_grad[(size_t)b * p.n_sel + sf] =
                    (s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]) * (double)p.fs;
        }
    }
}

} // namespace
