// support.hpp -- segment sums, packing kernels (raw records -> packed streams), debug kernels
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
#pragma once

namespace {

// ---------------------------------------------------------------------------
// Sums over the frames of a window, in ONE association that does not depend on how many devices
// share the frames: slots are grouped into CHUNKS (the slots of the window whose frame-table index
// falls into the same block of 64; the host builds the plan), a chunk is summed sequentially in
// slot order, and a window is the sequential sum of its chunks in order.  A device that holds only
// part of a window contributes its chunk sums, and whoever adds the chunks in order -- this kernel
// on a single device, the host across several -- gets bit-identical totals.
//   in[r][slot]                 per-slot values of row r (a candidate delay, a line-search trial)
//   idx[j]                      slot of plan position j (NULL = j itself)
//   chunk_off[c] .. [c+1]       plan positions of chunk c
//   win_chunk_off[w] .. [w+1]   chunks of window w
//   chunk_out[r][c], win_out[r][w]
__global__ __launch_bounds__(64) void plan_sum_kernel(const double* __restrict__ in, uint32_t n_cols,
                                                         const uint32_t* __restrict__ idx,
                                                         const uint32_t* __restrict__ chunk_off, uint32_t n_chunks,
                                                         const uint32_t* __restrict__ win_chunk_off, uint32_t n_win,
                                                         double* __restrict__ chunk_out, double* __restrict__ win_out) {
    const uint32_t r = blockIdx.x / n_win, w = blockIdx.x % n_win;
    const uint32_t c0 = win_chunk_off[w], c1 = win_chunk_off[w + 1];
    const double* row = in + (size_t)r * n_cols;
    double* cout = chunk_out + (size_t)r * n_chunks;
    for (uint32_t c = c0 + threadIdx.x; c < c1; c += blockDim.x) {
        double acc = 0.0;
        uint32_t j = chunk_off[c];
        const uint32_t j1 = chunk_off[c + 1];
        for (; j + 8 <= j1; j += 8) { // eight loads in flight, the additions still in slot order
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = row[idx ? idx[j + q] : j + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        for (; j < j1; ++j) acc += row[idx ? idx[j] : j];
        cout[c] = acc;
    }
    __syncthreads(); // the chunk sums of this block are visible to its thread 0
    if (threadIdx.x == 0) {
        double acc = 0.0;
        uint32_t c = c0;
        for (; c + 16 <= c1; c += 16) { // sixteen loads in flight, the additions still in chunk order
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = cout[c + q];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += v[q];
        }
        for (; c < c1; ++c) acc += cout[c];
        win_out[blockIdx.x] = acc;
    }
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
// Packing kernels: raw records (what SetTrackResult / set_track_pixels staged, fp64) -> the packed
// streams.  One thread per ray pair; the spline parameter offset (ts - start) * fs - base is
// evaluated in fp64 here (core_private.cpp:19-20 without the delay, relative to the frame's base
// knot), written as fp64 for the Sync kernels and rounded once to fp32 for the PreSync kernel.
// Pixel frames run the driver's undistortion first (core_testcode.cpp:63-95,
// 135-158; fp64 = the reference's arithmetic).  HBM: 64 B read + 96 B written per pair.
struct PackParams {
    const double* raw;
    const rship_pack_frame* frames;
    f4* rays_a;
    f4* rays_b;
    double2* q0; // {ax,bx}
    double2* q1; // {ay,by}
    double2* q2; // {az,bz}
    double2* q3; // {ta,tb}
    double start, fs;
    uint32_t* bad;
};

__global__ __launch_bounds__(kBlock) void pack_frames_kernel(PackParams p) {
    const rship_pack_frame& fr = p.frames[blockIdx.x];
    const uint32_t n = fr.n_rays;
    for (uint32_t row = blockIdx.y * kBlock + threadIdx.x; row < n; row += gridDim.y * kBlock) {
        double ra[3], rb[3], tsa, tsb;
        const double* rec = p.raw + fr.raw_offset;
        if (fr.is_pixels) {
            const double2* src = (const double2*)(rec + 4 * (size_t)row);
            const double2 a = src[0], b = src[1];
            const rs::Lens lens{fr.lens[0], fr.lens[1], fr.lens[2], fr.lens[3], fr.lens[4], fr.lens[5], fr.lens[6], fr.lens[7], fr.lens[8]};
            rs::pixel_to_ray(lens, a.x, a.y, fr.time_a, fr.rows, ra, &tsa);
            rs::pixel_to_ray(lens, b.x, b.y, fr.time_b, fr.rows, rb, &tsb);
        } else {
            tsa = rec[row];
            tsb = rec[(size_t)n + row];
            const double* pa = rec + 2 * (size_t)n + 3 * (size_t)row;
            const double* pb = rec + 5 * (size_t)n + 3 * (size_t)row;
            ra[0] = pa[0]; ra[1] = pa[1]; ra[2] = pa[2];
            rb[0] = pb[0]; rb[1] = pb[1]; rb[2] = pb[2];
        }
        const double ta = rs::knot_offset(tsa, p.start, p.fs, fr.base);
        const double tb = rs::knot_offset(tsb, p.start, p.fs, fr.base);
        f4 o0, o1;
        o0.x = (float)ra[0]; o0.y = (float)rb[0]; o0.z = (float)ra[1]; o0.w = (float)rb[1];
        o1.x = (float)ra[2]; o1.y = (float)rb[2]; o1.z = (float)ta; o1.w = (float)tb;
        const bool ok = finite_f(o0.x) && finite_f(o0.y) && finite_f(o0.z) && finite_f(o0.w) && finite_f(o1.x) &&
                        finite_f(o1.y) && finite_f(o1.z) && finite_f(o1.w);
        if (!ok) atomicAdd(p.bad, 1u);
        const size_t o = (size_t)fr.ray_offset + row;
        p.rays_a[o] = o0;
        p.rays_b[o] = o1;
        p.q0[o] = double2{ra[0], rb[0]};
        p.q1[o] = double2{ra[1], rb[1]};
        p.q2[o] = double2{ra[2], rb[2]};
        p.q3[o] = double2{ta, tb};
    }
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
