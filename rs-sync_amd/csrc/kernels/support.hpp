// support.hpp -- segment sums, pixels -> rays, debug kernels
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
#pragma once

namespace {

// ---------------------------------------------------------------------------
// out[r][w] = sum over j in [off[w], off[w+1]) of in[r][idx ? idx[j] : j]: per-window (segment)
// sums over frames with a fixed association, so results are bitwise reproducible and a window
// summed inside a batch equals the same window summed alone.  With one segment covering all
// columns this is the plain over-frames sum.

__global__ __launch_bounds__(kBlock) void segment_sum_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                            uint32_t n_cols, const uint32_t* __restrict__ idx,
                                                            const uint32_t* __restrict__ off, uint32_t n_seg) {
    __shared__ double s_red[4];
    const uint32_t r = blockIdx.x / n_seg, w = blockIdx.x % n_seg;
    const uint32_t j0 = off ? off[w] : 0u, j1 = off ? off[w + 1] : n_cols;
    const double* row = in + (size_t)r * n_cols;
    double acc = 0.0;
    for (uint32_t j = j0 + threadIdx.x; j < j1; j += kBlock) acc += row[idx ? idx[j] : j];
    double wsum = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// debug: P (and dP/dd) rows of one frame
struct DebugParams {
    const f4* rays_a;
    const f4* rays_b;
    const FrameRec* frames;
    uint32_t fi;
    const f4* coef;
    int n_knots;
    float fs;
    int32_t kd;
    float fd;
    float* P;
    float* dP;
};

__global__ __launch_bounds__(kBlock) void debug_problem_kernel(DebugParams p) {
    __shared__ f4 s_win[4 * kWinMax];
    const FrameRec fr = p.frames[p.fi];
    Spline sp;
    sp.g = p.coef;
    sp.n = p.n_knots;
    stage_window(sp, s_win, fr.base_knot + (int)floorf(fr.tmin) + p.kd, fr.base_knot + (int)floorf(fr.tmax) + p.kd + 1);
    __syncthreads();
    for (uint32_t row = blockIdx.x * kBlock + threadIdx.x; row < fr.n; row += gridDim.x * kBlock) {
        f3 P, dP;
        if (sp.path == kPathInterior) residual_row<true, kPathInterior>(sp, p.rays_a[fr.off + row], p.rays_b[fr.off + row], fr.base_knot + p.kd, p.fd, P, dP);
        else residual_row<true, kPathGlobal>(sp, p.rays_a[fr.off + row], p.rays_b[fr.off + row], fr.base_knot + p.kd, p.fd, P, dP);
        p.P[3 * row] = P.x; p.P[3 * row + 1] = P.y; p.P[3 * row + 2] = P.z;
        if (p.dP) { p.dP[3 * row] = dP.x * p.fs; p.dP[3 * row + 1] = dP.y * p.fs; p.dP[3 * row + 2] = dP.z * p.fs; }
    }
}

// ---------------------------------------------------------------------------
// pixel -> ray (core_testcode.cpp:63-95,135-158).  One thread per tracked
// pair, fp64 (the reference's arithmetic; gfx950 issues fp64 FMA at the fp32 rate), results
// rounded once to the packed fp32 layout.  HBM: 32 B read + 32 B written per pair.
struct PixelParams {
    const double* px;
    const rship_pixel_frame* frames;
    f4* rays_a;
    f4* rays_b;
    uint32_t* bad;
};

__global__ __launch_bounds__(kBlock) void rays_from_pixels_kernel(PixelParams p) {
    const rship_pixel_frame& fr = p.frames[blockIdx.x];
    const uint32_t row = blockIdx.y * kBlock + threadIdx.x;
    if (row >= fr.n_rays) return;
    const double2* src = (const double2*)(p.px + 4 * (fr.px_offset + row));
    const double2 a = src[0], b = src[1];
    rs::Lens lens{fr.lens[0], fr.lens[1], fr.lens[2], fr.lens[3], fr.lens[4], fr.lens[5], fr.lens[6], fr.lens[7], fr.lens[8]};
    double ra[3], rb[3], tsa, tsb;
    rs::pixel_to_ray(lens, a.x, a.y, fr.time_a, fr.rows, ra, &tsa);
    rs::pixel_to_ray(lens, b.x, b.y, fr.time_b, fr.rows, rb, &tsb);
    const float ta = (float)rs::knot_offset(tsa, fr.start, fr.fs, fr.base);
    const float tb = (float)rs::knot_offset(tsb, fr.start, fr.fs, fr.base);
    f4 o0, o1;
    o0.x = (float)ra[0]; o0.y = (float)rb[0]; o0.z = (float)ra[1]; o0.w = (float)rb[1];
    o1.x = (float)ra[2]; o1.y = (float)rb[2]; o1.z = ta; o1.w = tb;
    const bool ok = finite_f(o0.x) && finite_f(o0.y) && finite_f(o0.z) && finite_f(o0.w) && finite_f(o1.x) &&
                    finite_f(o1.y) && finite_f(o1.z) && finite_f(o1.w);
    if (!ok) atomicAdd(p.bad, 1u);
    p.rays_a[fr.ray_offset + row] = o0;
    p.rays_b[fr.ray_offset + row] = o1;
}

// debug: the wave-level exact selection on caller-provided residuals (one wave per problem,
// 2048 slots, NaN-padded), exactly as the LMedS kernel drives it
__global__ __launch_bounds__(64) void debug_select_kernel(const float* __restrict__ vals, uint32_t n, uint32_t kq,
                                                          const float* __restrict__ upper, uint32_t* out) {
    constexpr int NR = 32;
    const int lane = threadIdx.x;
    const float* v = vals + (size_t)blockIdx.x * n;
    uint32_t r2[NR];
#pragma unroll
    for (int m = 0; m < NR; ++m) {
        uint32_t i = m * 64 + lane;
        r2[m] = (i < n) ? __float_as_uint(v[i]) : 0x7fc00000u;
    }
    uint32_t hi2 = upper ? __float_as_uint(upper[blockIdx.x]) : kInfBits;
    const uint32_t tot = wave_count_lt(r2, hi2);
    uint32_t res = 0xffffffffu; // "not better than the bound"
    if (tot > kq) {
        if (hi2 == kInfBits) {
            float mx = 0.f;
#pragma unroll
            for (int m = 0; m < NR; ++m) mx = fmaxf(mx, fabsf(__uint_as_float(r2[m])));
            mx = wave_max_f32(mx);
            if (finite_f(mx)) hi2 = __float_as_uint(mx) + 1u;
        }
        res = select_kth(r2, kq, hi2, tot);
    }
    if (lane == 0) {
        out[2 * blockIdx.x] = res;
        out[2 * blockIdx.x + 1] = tot;
    }
}

} // namespace
