// rssync_kernels.hip -- gfx950 (CDNA4) kernels for the rs-sync PreSync/Sync hot
// path and the thin C-ABI the host solver calls (include/rssync_hip.h).
//
// Kernels (all wave64, 256-thread workgroups, no MFMA: the path has no dense
// contraction):
//   lmeds_kernel       one workgroup per (frame, chunk of candidate delays): residual
//                      matrix P into an LDS tile, LMedS hypothesis search with an exact
//                      lower-quartile selection, robust PreSync cost.  The same kernel
//                      in INIT mode is Sync's GuessMotion/GuessK.
//   loss_kernel        one workgroup per frame: residual + robust loss (+ analytic
//                      d/d-delay) for a batch of delays, rays held in registers.
//   opt_motion_kernel  one workgroup per frame: P in registers, restated L-BFGS on the
//                      3-vector motion estimate.
//   segment_sum_kernel fixed-order fp64 sums over the frames of each window.
// Data layout and the roofline that bounds each kernel: DESIGN.md.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rssync_hip.h"
#include "device_math.hpp"
#include "lens_math.hpp"

using rs::f3;
using rs::f4;

namespace {

constexpr int kBlock = 256;
constexpr int kWinMax = 64;  // knots of the spline staged in LDS per workgroup
constexpr uint32_t kInfBits = 0x7f800000u;

// ---------------------------------------------------------------------------
// wave64 / workgroup reductions.  DPP row shifts + row broadcasts (gfx9 forms):
// after the six steps lane 63 holds the wave total.

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
    int v = (int)x;
    v += dpp_i<0x111, 0xf>(v); // row_shr:1
    v += dpp_i<0x112, 0xf>(v); // row_shr:2
    v += dpp_i<0x114, 0xf>(v); // row_shr:4
    v += dpp_i<0x118, 0xf>(v); // row_shr:8
    v += dpp_i<0x142, 0xa>(v); // row_bcast:15 -> rows 1,3
    v += dpp_i<0x143, 0xc>(v); // row_bcast:31 -> rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_f<0x111, 0xf>(v);
    v += dpp_f<0x112, 0xf>(v);
    v += dpp_f<0x114, 0xf>(v);
    v += dpp_f<0x118, 0xf>(v);
    v += dpp_f<0x142, 0xa>(v);
    v += dpp_f<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_d<0x111, 0xf>(v);
    v += dpp_d<0x112, 0xf>(v);
    v += dpp_d<0x114, 0xf>(v);
    v += dpp_d<0x118, 0xf>(v);
    v += dpp_d<0x142, 0xa>(v);
    v += dpp_d<0x143, 0xc>(v);
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// workgroup sum of a per-thread fp32 partial: fp32 inside the wave, fp64 across
// the four waves.  `slot` is a 4-double LDS scratch that the caller must not reuse before another
// barrier has passed (one barrier here).
__device__ __forceinline__ double block_sum(float v, double* slot) {
    float w = wave_sum_f32(v);
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = (double)w;
    __syncthreads();
    return slot[0] + slot[1] + slot[2] + slot[3];
}

__device__ __forceinline__ bool finite_f(float x) { return (__float_as_uint(x) & kInfBits) != kInfBits; }

// ---------------------------------------------------------------------------
// spline window in LDS: SoA by coefficient kind so that neighbouring knots
// fall into different banks (ds_read_b128 of kind k, knot j at (k*kWinMax + j) * 16 B).

struct Spline {
    const f4* __restrict__ g; // global table, 4 f4 per knot
    const f4* lds;            // [4][kWinMax]
    int n;                    // knots
    int w0, wlen;             // staged range [w0, w0 + wlen)
    int path;                 // kPathGlobal / kPathLds / kPathInterior, uniform over the workgroup
};

// How a workgroup reads spline coefficients.  The choice is made once per workgroup from the knot
// range it can touch, so that the hot loops carry no per-lane LDS-or-global selection (which
// would turn ds_read_b128 into flat loads) and, in the common case, no extrapolation logic.
constexpr int kPathGlobal = 0;   // general: any parameter (extrapolation branches included), table read from L2
constexpr int kPathInterior = 2; // staged in LDS and strictly inside the knots (0 <= idx <= n-2)

__device__ __forceinline__ void stage_window(Spline& s, f4* s_win, int lo, int hi) {
    const int n = s.n;
    const bool interior = lo >= 0 && hi <= n - 2;
    lo = lo < 0 ? 0 : (lo > n - 1 ? n - 1 : lo);
    hi = hi < 0 ? 0 : (hi > n - 1 ? n - 1 : hi);
    int wlen = hi - lo + 1;
    s.path = (wlen <= kWinMax && interior) ? kPathInterior : kPathGlobal;
    if (wlen > kWinMax) wlen = kWinMax;
    s.w0 = lo;
    s.wlen = wlen;
    s.lds = s_win;
    for (int e = threadIdx.x; e < wlen * 4; e += kBlock) {
        int knot = e >> 2, kind = e & 3;
        s_win[kind * kWinMax + knot] = s.g[(size_t)(lo + knot) * 4 + kind];
    }
}

template <int PATH>
__device__ __forceinline__ void fetch_coef(const Spline& s, int ci, f4& y, f4& b, f4& c, f4& d) {
    if (PATH == kPathGlobal) {
        const f4* p = s.g + (size_t)ci * 4;
        y = p[0]; b = p[1]; c = p[2]; d = p[3];
    } else {
        const int rel = ci - s.w0;
        y = s.lds[rel];
        b = s.lds[kWinMax + rel];
        c = s.lds[2 * kWinMax + rel];
        d = s.lds[3 * kWinMax + rel];
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));

// Knot index and in-knot fraction for the PreSync sweep's interior path: x = t + fd directly.
// t >= 0 (offsets are relative to the frame's base knot) and 0 <= fd < 1, so truncation is floor
// and v_fract is exact.  Compared with spline_locate_interior (fraction of t first, then + fd, then
// wrap) this rounds the sum at the magnitude of t (< 64 knots): 4e-6 knots = 10 ns of delay at
// 400 Hz, far below the sweep's grid -- and it is 4 VALU instead of 9 per ray.  Sync keeps the
// precise form (its line search compares losses at delays a few ns apart).
__device__ __forceinline__ rs::Knot locate_sweep(float t, int base, float fd) {
    const float x = t + fd;
    return rs::Knot{base + (int)x, __builtin_amdgcn_fractf(x), false};
}

// one row of P = ar x br (core_private.cpp:24-28) and, if DERIV, dP/dx (x in knots).
// A = {ax,bx,ay,by}, B = {az,bz,ta,tb} as stored in HBM.
template <bool DERIV, int PATH, bool SWEEP = false>
__device__ __forceinline__ void residual_row(const Spline& s, f4 A, f4 B, int base, float fd, f3& P, f3& dP) {
    f4 ya, ba, ca, da, yb, bb, cb, db;
    rs::Knot ka = (PATH == kPathInterior) ? (SWEEP ? locate_sweep(B.z, base, fd) : rs::spline_locate_interior(B.z, base, fd))
                                          : rs::spline_locate(B.z, base, fd, s.n);
    fetch_coef<PATH>(s, ka.ci, ya, ba, ca, da);
    rs::Knot kb = (PATH == kPathInterior) ? (SWEEP ? locate_sweep(B.w, base, fd) : rs::spline_locate_interior(B.w, base, fd))
                                          : rs::spline_locate(B.w, base, fd, s.n);
    fetch_coef<PATH>(s, kb.ci, yb, bb, cb, db);
    if (!DERIV && PATH == kPathInterior) {
        // hot path, all in packed fp32.  Horner per end on the component pairs (w,x), (y,z) exactly as
        // ds_read_b128 delivers them (same fma chain per component as rs::horner, so the values are
        // bit-identical), eight moves to regroup by component across the two ends, then both
        // rotations at once (lane halves = the two ends of the pair; the interleaved ray layout puts
        // the ray components in adjacent registers already).
        const v2f ha = {ka.h, ka.h}, hb = {kb.h, kb.h};
        const v2f a01 = ((v2f{da.x, da.y} * ha + v2f{ca.x, ca.y}) * ha + v2f{ba.x, ba.y}) * ha + v2f{ya.x, ya.y};
        const v2f a23 = ((v2f{da.z, da.w} * ha + v2f{ca.z, ca.w}) * ha + v2f{ba.z, ba.w}) * ha + v2f{ya.z, ya.w};
        const v2f b01 = ((v2f{db.x, db.y} * hb + v2f{cb.x, cb.y}) * hb + v2f{bb.x, bb.y}) * hb + v2f{yb.x, yb.y};
        const v2f b23 = ((v2f{db.z, db.w} * hb + v2f{cb.z, cb.w}) * hb + v2f{bb.z, bb.w}) * hb + v2f{yb.z, yb.w};
        // (eight v_mov_b32; the four v_pk_mov_b32 that would do the same made the kernel 17 % SLOWER
        // on gfx950 -- measured, tools/ubench/pk_mov.hip documents the operand selection)
        const v2f qw = __builtin_shufflevector(a01, b01, 0, 2), qx = __builtin_shufflevector(a01, b01, 1, 3);
        const v2f qy = __builtin_shufflevector(a23, b23, 0, 2), qz = __builtin_shufflevector(a23, b23, 1, 3);
        const v2f vx = {A.x, A.y}, vy = {A.z, A.w}, vz = {B.x, B.y};
        const v2f n2 = qw * qw + qx * qx + qy * qy + qz * qz;
        // R(q/|q|)^T v = v + (2/|q|^2) (u x (u x v) - w (u x v)), u = (qx,qy,qz)  (rs::rotate_inv);
        // |q|^2 = 0 leaves v unchanged (u = 0 times a large finite factor)
        const v2f sc = {2.f * rs::rcp_fast(fmaxf(n2.x, 1e-30f)), 2.f * rs::rcp_fast(fmaxf(n2.y, 1e-30f))};
        const v2f tx = qy * vz - qz * vy, ty = qz * vx - qx * vz, tz = qx * vy - qy * vx;
        const v2f ux = qy * tz - qz * ty, uy = qz * tx - qx * tz, uz = qx * ty - qy * tx;
        const v2f rx = vx + sc * (ux - qw * tx), ry = vy + sc * (uy - qw * ty), rz = vz + sc * (uz - qw * tz);
        P = f3{ry.x * rz.y - rz.x * ry.y, rz.x * rx.y - rx.x * rz.y, rx.x * ry.y - ry.x * rx.y}; // ar x br
    } else {
        f3 ar, br, dar, dbr;
        rs::rotate_ray<DERIV>(ya, ba, ca, da, ka, f3{A.x, A.z, B.x}, ar, dar);
        rs::rotate_ray<DERIV>(yb, bb, cb, db, kb, f3{A.y, A.w, B.y}, br, dbr);
        P = rs::cross(ar, br);
        if (DERIV) dP = rs::add(rs::cross(dar, br), rs::cross(ar, dbr));
    }
}

struct FrameRec { // == rship_frame
    uint32_t off, n;
    int32_t base_knot;
    float tmin, tmax;
    uint32_t reserved;
    int64_t id;
};
static_assert(sizeof(FrameRec) == sizeof(rship_frame), "frame record layout");

// ---------------------------------------------------------------------------
// K2: LMedS tile kernel

struct LmedsParams {
    const f4* rays_a;
    const f4* rays_b;
    const FrameRec* frames;
    const uint32_t* sel;
    uint32_t n_sel;
    const f4* coef;
    int n_knots;
    const int32_t* kd;
    const float* fd;
    uint32_t n_cand, chunk, n_chunks;
    uint32_t n_hyp, stream_base, stream_stride; // sampler stream = base + candidate + group * stride
    uint64_t seed;
    const uint32_t* grp; // slot -> group (window) or null; delays are indexed [candidate][group]
    uint32_t n_grp;      // >= 1
    double* frame_cost; // [n_cand][n_sel]
    int32_t* best_h;    // [n_cand][n_sel] or null
    double* M;          // INIT mode: per selection slot [3]
    double* k;          // INIT mode
    uint32_t* flags;
};

// ---- LMedS tile in LDS, struct-of-arrays: unit rows n = safe_normalize(P).  The norms |P|
// stay in the registers of the thread that owns the row (only stage D needs them).
struct Tile {
    float* nx;
    float* ny;
    float* nz;
};

// hypothesis direction v = safe_normalize(P[i0] x P[i1]) (core_private.cpp:45-46).  The tile
// holds unit rows, and P[i0] x P[i1] is a positive multiple of n[i0] x n[i1], so the direction is
// the same; the "leave it un-normalised below 1e-12" rule of safe_normalize (inline_utils.hpp:5-11)
// is applied to |n[i0] x n[i1]| instead of |P[i0] x P[i1]| (it only fires for rows parallel to
// within 1e-12 rad, where the hypothesis is noise either way).
__device__ __forceinline__ f3 hypothesis(const Tile& t, uint64_t seed, int64_t frame, uint32_t stream, uint32_t h,
                                         uint32_t n) {
    uint32_t i0, i1;
    rs::sample_pair(seed, frame, stream, h, n, i0, i1);
    f3 v = rs::cross(f3{t.nx[i0], t.ny[i0], t.nz[i0]}, f3{t.nx[i1], t.ny[i1], t.nz[i1]});
    float nn = sqrtf(rs::dot(v, v));
    if (!(nn < 1e-12f)) {
        float inv = 1.0f / nn;
        v = rs::scale(v, inv);
    }
    return v;
}

// wave-wide count of |r[]| < pivot (pivot: bit pattern of a non-negative float, uniform).
// The kernel is bound by VALU issue (one wave64 instruction per 4 cycles per SIMD, PMC-measured),
// while the scalar unit is mostly idle: each register costs ONE v_cmp (the abs modifier is free,
// NaN never counts) whose 64-lane mask is counted with s_bcnt1_i32_b64 and added on the SALU.
// The total arrives in an SGPR, so no cross-lane reduction is needed either.
template <int NR>
__device__ __forceinline__ uint32_t wave_count_lt(const uint32_t (&r)[NR], uint32_t pivot) {
    const float pv = __uint_as_float(pivot);
    uint32_t cnt = 0;
#pragma unroll
    for (int m = 0; m < NR; ++m)
        cnt += (uint32_t)__builtin_popcountll(__builtin_amdgcn_fcmpf(pv, fabsf(__uint_as_float(r[m])), 2 /* FCMP_OGT */));
    return cnt;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umin(uint32_t v) {
    uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_umin<0x111, 0xf>(v);
    v = dpp_umin<0x112, 0xf>(v);
    v = dpp_umin<0x114, 0xf>(v);
    v = dpp_umin<0x118, 0xf>(v);
    v = dpp_umin<0x142, 0xa>(v);
    v = dpp_umin<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fmax(float v) { // NaN-ignoring max; lanes without a source keep v
    float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
    return fmaxf(o, v);
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = dpp_fmax<0x111, 0xf>(v);
    v = dpp_fmax<0x112, 0xf>(v);
    v = dpp_fmax<0x114, 0xf>(v);
    v = dpp_fmax<0x118, 0xf>(v);
    v = dpp_fmax<0x142, 0xa>(v);
    v = dpp_fmax<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Exact kq-th smallest (0-based) of the wave's |r[]| as a bit pattern, given an exclusive upper
// bound hi with count(|r| < hi) = c_hi > kq.  |r| orders exactly like the r^2 the reference sorts
// (core_private.cpp:49-52), so this is the element std::sort would leave at index kq, before
// squaring.  A bracket [lo, hi) with counts c_lo <= kq < c_hi is narrowed by counting passes;
// pivots come from a secant step on the empirical CDF of |r| (close to uniform around the lower
// quartile, so the CDF is nearly linear there: ~8 passes instead of 31 bit-bisection passes), with
// bracket interpolation and plain bisection of the bit pattern as fallbacks.  Ends when the
// bracket is one bit pattern wide or holds exactly one element, which a min pass extracts.
// All bookkeeping is wave-uniform and kept on the scalar unit (bit patterns of non-negative
// floats order like unsigned integers); only the secant formula itself runs on the VALU.
template <int NR>
__device__ __forceinline__ uint32_t select_kth(const uint32_t (&r)[NR], uint32_t kq, uint32_t hi, uint32_t c_hi) {
    uint32_t lo = 0, c_lo = 0;
    uint32_t a1 = 0, c1 = 0, a2 = hi, c2 = c_hi; // the two most recent (pivot, count) points
    for (int it = 0;; ++it) {
        if (hi - lo == 1u) return lo;
        if (c_hi - c_lo == 1u) {
            // the single element in [lo, hi): smallest |x| >= lo; |x| < lo wraps to a huge difference
            uint32_t mn = 0xffffffffu;
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                uint32_t d = (r[m] & 0x7fffffffu) - lo;
                mn = d < mn ? d : mn;
            }
            return lo + wave_min_u32(mn);
        }
        uint32_t piv = 0;
        if (it < 24) {
            if (c2 != c1) { // secant through the last two points, aimed at rank kq + 1/2
                const float num = 0.5f * (float)(int)(2 * kq + 1 - 2 * c2);
                const float a3 = fmaf(num * (__uint_as_float(a2) - __uint_as_float(a1)), rs::rcp_fast((float)(int)(c2 - c1)),
                                      __uint_as_float(a2));
                piv = uniform_u32(__float_as_uint(a3));
            }
            if (!(piv > lo && piv < hi)) { // interpolate inside the bracket instead
                const float num = 0.5f * (float)(int)(2 * kq + 1 - 2 * c_lo);
                const float a3 = fmaf(num * (__uint_as_float(hi) - __uint_as_float(lo)), rs::rcp_fast((float)(c_hi - c_lo)),
                                      __uint_as_float(lo));
                piv = uniform_u32(__float_as_uint(a3));
            }
        }
        if (!(piv > lo && piv < hi)) piv = lo + ((hi - lo) >> 1); // bit bisection: guaranteed finish
        const uint32_t c = wave_count_lt(r, piv);
        a1 = a2; c1 = c2;
        a2 = piv; c2 = c;
        if (c <= kq) { lo = piv; c_lo = c; }
        else { hi = piv; c_hi = c; }
    }
}

// stage A of the LMedS kernel: this thread's rows of P for one delay, written to the LDS tile as
// unit rows, norms kept in nrm[]; returns RSHIP_BAD_P if a row is not finite.  Rows >= N are not
// touched: the kernel fills them with NaN once (their residuals compare above every threshold).
template <int PATH, bool SWEEP>
__device__ __forceinline__ uint32_t lmeds_row(const Spline& sp, const f4* __restrict__ rays_a,
                                              const f4* __restrict__ rays_b, uint32_t N, uint32_t row, int base, float fd,
                                              const Tile& tile, float& nrm) {
    uint32_t bad = 0;
    nrm = 0.f;
    if (row < N) {
        f3 P, dP;
        residual_row<false, PATH, SWEEP>(sp, rays_a[row], rays_b[row], base, fd, P, dP);
        const float n2 = rs::dot(P, P);
        if (!finite_f(n2)) bad = RSHIP_BAD_P;
        // safe_normalize (core_private.cpp:35-36): rows with |P| < 1e-12 stay as they are
        const bool tiny = n2 < 1e-24f;
        const float inv = tiny ? 1.f : rs::rsqrt_fast(n2);
        tile.nx[row] = P.x * inv; tile.ny[row] = P.y * inv; tile.nz[row] = P.z * inv;
        nrm = tiny ? 1.f : n2 * inv;
    }
    return bad;
}

template <int RPT, bool SWEEP>
__device__ __forceinline__ uint32_t lmeds_rows(const Spline& sp, const f4* __restrict__ rays_a,
                                               const f4* __restrict__ rays_b, uint32_t N, int base, float fd,
                                               const Tile& tile, float (&nrm)[RPT]) {
    uint32_t bad = 0;
    if (sp.path == kPathInterior) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            bad |= lmeds_row<kPathInterior, SWEEP>(sp, rays_a, rays_b, N, j * kBlock + threadIdx.x, base, fd, tile, nrm[j]);
        }
    } else { // rare (ends of the gyro track, wild delays): keep the code small, not fast
        float tmp[RPT];
#pragma unroll 1
        for (int j = 0; j < RPT; ++j)
            bad |= lmeds_row<kPathGlobal, false>(sp, rays_a, rays_b, N, j * kBlock + threadIdx.x, base, fd, tile, tmp[j]);
#pragma unroll
        for (int j = 0; j < RPT; ++j) nrm[j] = tmp[j];
    }
    return bad;
}

// waves per SIMD each kernel is compiled for (second __launch_bounds__ argument): the
// LMedS tile is LDS-limited to 3 workgroups per CU at 8 rows per thread
__host__ __device__ constexpr int lmeds_waves(int rpt) { return 5; }
__host__ __device__ constexpr int loss_waves(int rpt, bool grad) { return (grad || rpt >= 8) ? 3 : 4; }

constexpr int kMaxChunk = 32; // candidates per workgroup (rship: chunk <= kMaxChunk)
constexpr int kHypBatch = 64; // hypothesis directions prepared per batch (one per lane of wave 0)

// Pop the next index of an LDS work queue for the whole wave: lane 0 alone performs the atomic,
// the result is broadcast.  Written as one asm statement because hipcc's structuriser turns the
// obvious `if (lane == 0) j = atomicAdd(..); j = readfirstlane(j);` inside a loop into a per-lane
// waterfall that re-reads the queue head for the other lanes and never terminates.
__device__ __forceinline__ uint32_t wave_pop(uint32_t* counter) {
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)counter;
    const uint32_t one = 1u;
    uint32_t old;
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "ds_add_rtn_u32 %0, %2, %3\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(old), "=&s"(save)
                 : "v"(addr), "v"(one)
                 : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
}

template <int RPT, int MODE> // MODE 0: PreSync cost per candidate; 1: GuessMotion + GuessK
__global__ __launch_bounds__(kBlock, lmeds_waves(RPT)) void lmeds_kernel(LmedsParams p) {
    constexpr int ROWS = kBlock * RPT;
    constexpr int NR = 4 * RPT; // residual registers per lane: a wave spans the whole tile
    __shared__ __attribute__((aligned(16))) float s_n[3][ROWS];
    __shared__ f4 s_win[4 * kWinMax];
    __shared__ f4 s_hyp[kHypBatch];
    __shared__ double s_red[2][4];
    // best (quantile, hypothesis) so far, packed (bits << 32 | h): a 64-bit min is exactly
    // "smaller quantile wins, ties go to the earlier hypothesis" (core_private.cpp:53 strict <)
    __shared__ unsigned long long s_key;
    __shared__ uint32_t s_next; // hypothesis queue of the current batch

    const int tid = threadIdx.x, lane = tid & 63;
    // blocks b and b+8 share an XCD (round-robin dispatch): keep the chunks of one
    // frame on one XCD so its rays are fetched into one L2 only
    const uint32_t per = 8u * p.n_chunks;
    const uint32_t grp = blockIdx.x / per, within = blockIdx.x % per;
    const uint32_t sf = grp * 8u + (within & 7u);
    const uint32_t chunk = within >> 3;
    if (sf >= p.n_sel) return;
    const uint32_t fi = p.sel[sf];
    const FrameRec fr = p.frames[fi];
    const uint32_t N = fr.n;
    const uint32_t kq = N / 4; // core_private.cpp:52
    const uint32_t g = p.grp ? p.grp[sf] : 0u; // window this slot belongs to (batched Sync)
    const Tile tile{s_n[0], s_n[1], s_n[2]};

    // rays are re-read per candidate: the chunks of a frame share an XCD, so after the
    // first touch they come from that XCD's L2 (keeping them in registers costs 64 VGPRs)
    const f4* __restrict__ rays_a = p.rays_a + fr.off;
    const f4* __restrict__ rays_b = p.rays_b + fr.off;

    const uint32_t c0 = chunk * p.chunk;
    const uint32_t c1 = (c0 + p.chunk < p.n_cand) ? c0 + p.chunk : p.n_cand;
    if (c0 >= c1) return;

    // the chunk's delays, staged once: a scalar load per candidate would put an L2 round trip at
    // the head of every stage A
    __shared__ int s_kd[kMaxChunk];
    __shared__ float s_fd[kMaxChunk];
    if ((uint32_t)tid < c1 - c0) {
        s_kd[tid] = p.kd[(c0 + tid) * p.n_grp + g];
        s_fd[tid] = p.fd[(c0 + tid) * p.n_grp + g];
    }
    Spline sp;
    sp.g = p.coef;
    sp.n = p.n_knots;
    {
        int kd_lo = p.kd[c0 * p.n_grp + g], kd_hi = kd_lo;
        for (uint32_t c = c0 + 1; c < c1; ++c) {
            int v = p.kd[c * p.n_grp + g];
            kd_lo = v < kd_lo ? v : kd_lo;
            kd_hi = v > kd_hi ? v : kd_hi;
        }
        stage_window(sp, s_win, fr.base_knot + (int)floorf(fr.tmin) + kd_lo,
                     fr.base_knot + (int)floorf(fr.tmax) + kd_hi + 1);
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j) { // rows beyond N: NaN once, never rewritten
        const uint32_t row = j * kBlock + tid;
        if (row >= N) s_n[0][row] = s_n[1][row] = s_n[2][row] = __uint_as_float(0x7fc00000u);
    }
    __syncthreads();

    const f4* p4x = reinterpret_cast<const f4*>(tile.nx);
    const f4* p4y = reinterpret_cast<const f4*>(tile.ny);
    const f4* p4z = reinterpret_cast<const f4*>(tile.nz);
    uint32_t prev_best = kInfBits; // winning quantile of the previous candidate of this chunk

    for (uint32_t c = c0; c < c1; ++c) {
        const int base = fr.base_knot + s_kd[c - c0];
        const float fd = s_fd[c - c0];
        const uint32_t stream = p.stream_base + c + g * p.stream_stride; // g != 0 only for batched GuessMotion
        uint32_t bad = 0;
        // ---- stage A: rows of P -> LDS tile as unit rows; norms stay in registers ----
        float nrm[RPT];
        bad |= lmeds_rows<RPT, MODE == 0>(sp, rays_a, rays_b, N, base, fd, tile, nrm);

        // ---- stage C: the hypotheses.  The best quantile of the previous candidate (x1.25: between
        // neighbouring candidates it moves by -20..+26 %, 1st..99th percentile) serves as a
        // provisional bound: a hypothesis that has <= kq residuals below it is dropped after one
        // counting pass.  If nothing beats the bound (~2 % of candidates) the candidate is redone
        // without it, so the result is the exact arg-min either way.
        uint32_t guess = kInfBits;
        if (prev_best < 0x7e000000u && prev_best > 0x00800000u)
            guess = uniform_u32(__float_as_uint(__uint_as_float(prev_best) * 1.25f));
        unsigned long long best;
        for (;;) {
            if (tid == 0) s_key = ((unsigned long long)guess << 32);
            for (uint32_t batch = 0; batch < p.n_hyp; batch += kHypBatch) {
                const uint32_t nb = (p.n_hyp - batch < (uint32_t)kHypBatch) ? p.n_hyp - batch : (uint32_t)kHypBatch;
                __syncthreads(); // tile written / previous batch consumed
                if ((uint32_t)tid < nb) {
                    const f3 v = hypothesis(tile, p.seed, fr.id, stream, batch + tid, N);
                    s_hyp[tid] = f4{v.x, v.y, v.z, 0.f};
                }
                if (tid == 0) s_next = 0;
                __syncthreads();
                for (;;) { // waves pull hypotheses from the queue: no wave idles at the barrier
                    const uint32_t j = wave_pop(&s_next);
                    if (j >= nb) break;
                    const uint32_t h = batch + j;
                    const f4 hv = s_hyp[j];
                    // residuals r = nP v (core_private.cpp:48); |r| orders like the r^2 of :49-52
                    uint32_t r2[NR]; // registers 4m..4m+3 <-> rows 4 (64 m + lane) .. +3
#pragma unroll
                    for (int m = 0; m < NR / 4; ++m) {
                        if ((m & 1) == 0) __builtin_amdgcn_sched_barrier(0); // bound the LDS reads in flight
                        const int idx = m * 64 + lane;
                        // ds_read_b128 per array: full LDS rate (ds_read2_b64 pairs run at half of it)
                        const f4 x = p4x[idx], y = p4y[idx], z = p4z[idx];
                        const v2f r01 = v2f{x.x, x.y} * hv.x + v2f{y.x, y.y} * hv.y + v2f{z.x, z.y} * hv.z;
                        const v2f r23 = v2f{x.z, x.w} * hv.x + v2f{y.z, y.w} * hv.y + v2f{z.z, z.w} * hv.z;
                        r2[4 * m] = __float_as_uint(r01.x);
                        r2[4 * m + 1] = __float_as_uint(r01.y);
                        r2[4 * m + 2] = __float_as_uint(r23.x);
                        r2[4 * m + 3] = __float_as_uint(r23.y);
                    }
                    // (quantile_h, h) < (T, g)  <=>  more than kq |residuals| lie below T (+1 ulp if g > h):
                    // med < least_med of core_private.cpp:51-53 with the reference's first-wins tie rule
                    const unsigned long long key = __hip_atomic_load(&s_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t T = (uint32_t)(key >> 32), g = (uint32_t)key;
                    uint32_t hi2 = T + ((T != kInfBits && g > h) ? 1u : 0u);
                    const uint32_t tot = wave_count_lt(r2, hi2);
                    if (tot > kq) {
                        if (hi2 == kInfBits) { // no bound yet: start the bracket at the largest residual
                            float mx = 0.f;
#pragma unroll
                            for (int m = 0; m < NR; ++m) mx = fmaxf(mx, fabsf(__uint_as_float(r2[m])));
                            mx = wave_max_f32(mx);
                            if (finite_f(mx)) hi2 = __float_as_uint(mx) + 1u; // count(|r| < hi2) is still tot
                        }
                        const uint32_t kth = select_kth(r2, kq, hi2, tot);
                        if (lane == 0) atomicMin(&s_key, ((unsigned long long)kth << 32) | h);
                    }
                }
            }
            __syncthreads();
            best = s_key;
            if (guess == kInfBits || best != ((unsigned long long)guess << 32)) break;
            guess = kInfBits; // nothing beat the provisional bound: redo this candidate without it
            __syncthreads();  // everyone has read s_key before it is reset
        }
        const uint32_t bT = (uint32_t)(best >> 32);
        const int bH = (bT == kInfBits) ? -1 : (int)(uint32_t)best;
        prev_best = bT;
        f3 Mv = f3{0, 0, 0};
        if (bH >= 0) {
            if (p.n_hyp <= (uint32_t)kHypBatch) { // the winner's direction is still in the batch buffer
                const f4 hv = s_hyp[bH];
                Mv = f3{hv.x, hv.y, hv.z};
            } else {
                Mv = hypothesis(tile, p.seed, fr.id, stream, (uint32_t)bH, N);
            }
        }
        if (!(finite_f(Mv.x) && finite_f(Mv.y) && finite_f(Mv.z))) bad |= RSHIP_BAD_M;

        // ---- stage D: k = clamp(100 / |P M|), cost = sqrt(sum sqrt(log1p(r^2))) ----
        // Branch-free over the rows: a row beyond N has nrm = 0 but a NaN tile entry, so its
        // product is replaced by 0 with one select; zeros then contribute nothing below.
        float pm[RPT];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const uint32_t row = j * kBlock + tid;
            const float v = nrm[j] * rs::dot(f3{tile.nx[row], tile.ny[row], tile.nz[row]}, Mv);
            pm[j] = row < N ? v : 0.f;
            ss = fmaf(pm[j], pm[j], ss);
        }
        double ss_tot = block_sum(ss, s_red[0]);
        // core_private.cpp:79, 100 / ||P M|| as 100 * rsq (v_rsq_f32, 1 ulp); ss = 0 gives +inf -> clamp
        float kf = 100.0f * rs::rsqrt_fast((float)ss_tot);
        kf = (kf < 10.f) ? 10.f : ((1000.f < kf) ? 1000.f : kf);
        if (MODE == 1) {
            if (tid == 0) {
                p.M[3 * sf + 0] = (double)Mv.x;
                p.M[3 * sf + 1] = (double)Mv.y;
                p.M[3 * sf + 2] = (double)Mv.z;
                p.k[sf] = (double)kf;
            }
        } else {
            float sc = kf * rs::rsqrt_fast(rs::dot(Mv, Mv)); // core_private.cpp:80
            // a non-finite r or rho (core_private.cpp:81,83) makes the sums non-finite: NaN propagates
            // and all terms are >= 0, so the checks are made once on the sums, not per row
            float acc = 0.f, rsum = 0.f;
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const float r = pm[j] * sc;
                rsum += fabsf(r);
                const float rho = rs::log1p_pos_fast(r * r); // core_private.cpp:82
                // v_sqrt_f32 directly (1 ulp): libm's sqrtf adds range scaling for denormal inputs,
                // whose square roots (< 1e-19) cannot change a sum of O(1) terms in fp32
                acc += __builtin_amdgcn_sqrtf(rho);
            }
            if (!finite_f(rsum)) bad |= RSHIP_BAD_R;
            else if (!finite_f(acc)) bad |= RSHIP_BAD_RHO;
            double acc_tot = block_sum(acc, s_red[1]);
            if (tid == 0) {
                p.frame_cost[(size_t)c * p.n_sel + sf] = sqrt(acc_tot); // core_private.cpp:85
                if (p.best_h) p.best_h[(size_t)c * p.n_sel + sf] = bH;
            }
        }
        if (bad) atomicOr(p.flags, bad);
        // No barrier here.  What the next candidate overwrites before its first barrier is (a) this
        // thread's own tile rows and (b) s_key, by thread 0: every reader of s_key reads it before
        // the workgroup sum barrier of stage D, which thread 0 has passed by then.  s_hyp, s_next
        // and the sum slots are rewritten only after further barriers of the next candidate.
    }
}

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

// ---------------------------------------------------------------------------
// K3: per-frame L-BFGS on the motion vector, P resident in registers.
// Restates ens::L_BFGS as called at core_private.cpp:264-294 (MaxIterations 200,
// MinGradientNorm 1e-4, library defaults otherwise); the algorithm and the one
// stated choice (re-evaluate at the best step when it is not the last one tried)
// are those of oracle/rssync_oracle.c:lbfgs_minimise.  Control flow is uniform:
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
    unsigned long long* stats; // [0] += iterations, [1] += evaluations
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
    int it = 0;
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
        if (bestStep != lastStep) fval = ev(x, g);
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
        }
        if (p.per_frame) {
            p.per_frame[2 * sf] = (uint32_t)it;
            p.per_frame[2 * sf + 1] = (uint32_t)ev.evals;
        }
    }
}

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

// ===========================================================================
// host side of the C-ABI

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};
// a device allocation that lives for one call (freed on every return path)
struct TempBuf : DevBuf {
    TempBuf() = default;
    TempBuf(const TempBuf&) = delete;
    TempBuf& operator=(const TempBuf&) = delete;
    ~TempBuf() {
        if (p) (void)hipFree(p);
    }
};

struct rship_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    // problem data
    DevBuf coef, rays_a, rays_b, frames, sel, M, k, grp, grp_off, seg_idx, seg_off;
    uint32_t n_knots = 0, n_frames = 0, n_sel = 0, max_n = 0, n_grp = 1;
    uint64_t total_rays = 0;
    double fs = 0;
    std::vector<uint32_t> h_frame_n; // per table frame
    std::vector<uint32_t> h_sel;
    std::vector<uint32_t> h_delays; // staging of upload_delays
    const float* d_fd = nullptr;    // device address of the fd half of the last upload
    // scratch
    DevBuf kd, frame_cost, best_h, costs, part, flags, stats;
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    // profiling
    bool prof = false;
    struct Pending { int kind; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    uint64_t launches[RSHIP_K_COUNT] = {};
    double total_ms[RSHIP_K_COUNT] = {};
};

namespace {

int set_err(rship_ctx* c, const char* what, hipError_t e) {
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}
int set_err(rship_ctx* c, const std::string& what) {
    c->err = what;
    return 1;
}

#define RS_HIP(call)                                         \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return set_err(c, #call, e__); \
    } while (0)

int ensure(rship_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) RS_HIP(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    RS_HIP(hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

int ensure_pinned(rship_ctx* c, size_t bytes) {
    if (bytes <= c->pinned_cap) return 0;
    if (c->pinned) RS_HIP(hipHostFree(c->pinned));
    c->pinned = nullptr;
    c->pinned_cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    RS_HIP(hipHostMalloc(&c->pinned, want, hipHostMallocDefault));
    c->pinned_cap = want;
    return 0;
}

struct ProfScope {
    rship_ctx* c;
    int kind;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(rship_ctx* c_, int kind_) : c(c_), kind(kind_) {
        if (!c->prof) return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, c->stream);
    }
    ~ProfScope() {
        if (!c->prof) return;
        (void)hipEventRecord(b, c->stream);
        c->pending.push_back({kind, a, b});
    }
};

// called after a stream synchronisation: fold finished event pairs into the totals
void prof_collect(rship_ctx* c) {
    for (auto& pd : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pd.a, pd.b) == hipSuccess) {
            c->launches[pd.kind] += 1;
            c->total_ms[pd.kind] += (double)ms;
        }
        c->pool.push_back(pd.a);
        c->pool.push_back(pd.b);
    }
    c->pending.clear();
}

int sync_stream(rship_ctx* c) {
    RS_HIP(hipStreamSynchronize(c->stream));
    prof_collect(c);
    return 0;
}

int rpt_for(uint32_t max_n) {
    int rpt = 1;
    while ((uint32_t)rpt * kBlock < max_n) rpt *= 2;
    return rpt;
}

constexpr int kMaxRpt = 8;

uint32_t sel_max_n(const rship_ctx* c) {
    uint32_t m = 0;
    for (uint32_t i : c->h_sel) m = c->h_frame_n[i] > m ? c->h_frame_n[i] : m;
    return m;
}

template <int MODE>
int launch_lmeds(rship_ctx* c, const LmedsParams& p, int rpt, uint32_t grid) {
    ProfScope ps(c, MODE == 1 ? RSHIP_K_INIT : RSHIP_K_LMEDS);
    switch (rpt) {
        case 1: hipLaunchKernelGGL((lmeds_kernel<1, MODE>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
        case 2: hipLaunchKernelGGL((lmeds_kernel<2, MODE>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
        case 4: hipLaunchKernelGGL((lmeds_kernel<4, MODE>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
        case 8: hipLaunchKernelGGL((lmeds_kernel<8, MODE>), dim3(grid), dim3(kBlock), 0, c->stream, p); break;
        default: return set_err(c, "lmeds: unsupported rows-per-thread");
    }
    RS_HIP(hipGetLastError());
    return 0;
}

template <bool GRAD>
int launch_loss(rship_ctx* c, const LossParams& p, int rpt) {
    ProfScope ps(c, RSHIP_K_LOSS);
    switch (rpt) {
        case 1: hipLaunchKernelGGL((loss_kernel<1, GRAD>), dim3(p.n_sel), dim3(kBlock), 0, c->stream, p); break;
        case 2: hipLaunchKernelGGL((loss_kernel<2, GRAD>), dim3(p.n_sel), dim3(kBlock), 0, c->stream, p); break;
        case 4: hipLaunchKernelGGL((loss_kernel<4, GRAD>), dim3(p.n_sel), dim3(kBlock), 0, c->stream, p); break;
        case 8: hipLaunchKernelGGL((loss_kernel<8, GRAD>), dim3(p.n_sel), dim3(kBlock), 0, c->stream, p); break;
        default: return set_err(c, "loss: unsupported rows-per-thread");
    }
    RS_HIP(hipGetLastError());
    return 0;
}

int launch_motion(rship_ctx* c, const MotionParams& p, int rpt) {
    ProfScope ps(c, RSHIP_K_MOTION);
    switch (rpt) {
        case 1: hipLaunchKernelGGL((opt_motion_kernel<1>), dim3(p.n_sel), dim3(kBlock), 0, c->stream, p); break;
        case 2: hipLaunchKernelGGL((opt_motion_kernel<2>), dim3(p.n_sel), dim3(kBlock), 0, c->stream, p); break;
        case 4: hipLaunchKernelGGL((opt_motion_kernel<4>), dim3(p.n_sel), dim3(kBlock), 0, c->stream, p); break;
        case 8: hipLaunchKernelGGL((opt_motion_kernel<8>), dim3(p.n_sel), dim3(kBlock), 0, c->stream, p); break;
        default: return set_err(c, "motion: unsupported rows-per-thread");
    }
    RS_HIP(hipGetLastError());
    return 0;
}

int launch_reduce(rship_ctx* c, const double* in, double* out, uint32_t rows, uint32_t cols, const uint32_t* idx,
                  const uint32_t* off, uint32_t n_seg) {
    ProfScope ps(c, RSHIP_K_REDUCE);
    hipLaunchKernelGGL(segment_sum_kernel, dim3(rows * n_seg), dim3(kBlock), 0, c->stream, in, out, cols, idx, off, n_seg);
    RS_HIP(hipGetLastError());
    return 0;
}

// Every entry point works on the context's device, whatever the calling thread's current device
// is (a host with several GPUs in one process), and leaves the caller's choice as it found it.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const rship_ctx* c) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) (void)hipSetDevice(c->device);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// kd[n] then fd[n] in one device buffer, staged through the host so that it is ONE copy per call
// (the Sync loop uploads delays three times per outer iteration; API calls are what it waits for)
int upload_delays(rship_ctx* c, const int32_t* kd, const float* fd, size_t n) {
    if (ensure(c, c->kd, n * 8)) return 1;
    c->h_delays.resize(2 * n);
    memcpy(c->h_delays.data(), kd, n * 4);
    memcpy(c->h_delays.data() + n, fd, n * 4);
    RS_HIP(hipMemcpyAsync(c->kd.p, c->h_delays.data(), n * 8, hipMemcpyHostToDevice, c->stream));
    c->d_fd = (const float*)((const int32_t*)c->kd.p + n);
    return 0;
}
int check_ready(rship_ctx* c) {
    if (!c->n_knots) return set_err(c, "no gyro spline uploaded");
    if (!c->n_frames) return set_err(c, "no frames uploaded");
    if (!c->n_sel) return set_err(c, "no frames selected");
    return 0;
}

} // namespace

extern "C" {

int rship_max_tracks(void) { return kMaxRpt * kBlock; }

int rship_create(rship_ctx** out, int device) {
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return 2; // no GPU: the product path has no CPU fallback
    rship_ctx* c = new rship_ctx();
    if (device >= 0) {
        e = hipSetDevice(device);
        if (e != hipSuccess) { delete c; return 3; }
        c->device = device;
    } else {
        (void)hipGetDevice(&c->device);
    }
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return 4; }
    c->stream = c->own_stream;
    *out = c;
    return 0;
}

void rship_destroy(rship_ctx* c) {
    if (!c) return;
    DeviceGuard dev_guard(c);
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (auto e : c->pool) (void)hipEventDestroy(e);
    DevBuf* bufs[] = {&c->coef, &c->rays_a, &c->rays_b, &c->frames, &c->sel, &c->M, &c->k, &c->grp, &c->grp_off,
                      &c->seg_idx, &c->seg_off, &c->kd,
                      &c->frame_cost, &c->best_h, &c->costs, &c->part, &c->flags, &c->stats};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* rship_last_error(const rship_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rship_set_stream(rship_ctx* c, void* hip_stream) {
    DeviceGuard dev_guard(c);
    RS_HIP(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return 0;
}

int rship_upload_spline(rship_ctx* c, const float* coef16, uint32_t n_knots, double sample_rate) {
    DeviceGuard dev_guard(c);
    if (n_knots < 2) return set_err(c, "spline: need >= 2 knots");
    size_t bytes = (size_t)n_knots * 64;
    if (ensure(c, c->coef, bytes)) return 1;
    RS_HIP(hipMemcpyAsync(c->coef.p, coef16, bytes, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipStreamSynchronize(c->stream));
    c->n_knots = n_knots;
    c->fs = sample_rate;
    return 0;
}

int rship_upload_frames(rship_ctx* c, const float* rays_a4, const float* rays_b4, uint64_t total_rays,
                        const rship_frame* table, uint32_t n_frames) {
    DeviceGuard dev_guard(c);
    c->n_frames = 0;
    c->n_sel = 0;
    c->h_sel.clear();
    c->h_frame_n.assign(n_frames, 0);
    for (uint32_t i = 0; i < n_frames; ++i) {
        if ((uint64_t)table[i].ray_offset + table[i].n_rays > total_rays) return set_err(c, "frame table exceeds ray buffer");
        if (table[i].n_rays > (uint32_t)rship_max_tracks())
            return set_err(c, "frame has more tracks than the kernels accept (" + std::to_string(rship_max_tracks()) + ")");
        c->h_frame_n[i] = table[i].n_rays;
    }
    size_t rb = (size_t)total_rays * 16;
    if (ensure(c, c->rays_a, rb ? rb : 16) || ensure(c, c->rays_b, rb ? rb : 16)) return 1;
    if (ensure(c, c->frames, (size_t)n_frames * sizeof(rship_frame) + 32)) return 1;
    if (rb && rays_a4 && rays_b4) { // null: the caller fills the streams on the device (rship_rays_from_pixels)
        RS_HIP(hipMemcpyAsync(c->rays_a.p, rays_a4, rb, hipMemcpyHostToDevice, c->stream));
        RS_HIP(hipMemcpyAsync(c->rays_b.p, rays_b4, rb, hipMemcpyHostToDevice, c->stream));
    }
    if (n_frames) RS_HIP(hipMemcpyAsync(c->frames.p, table, (size_t)n_frames * sizeof(rship_frame), hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipStreamSynchronize(c->stream));
    c->n_frames = n_frames;
    c->total_rays = total_rays;
    return 0;
}

// Selection = list of slots.  A slot names a frame of the table; with groups (batched windows)
// the same frame may appear in several slots, slots of one group are contiguous, and the
// per-frame Sync state (M, k) is kept per slot.
int rship_select_slots(rship_ctx* c, const uint32_t* idx, uint32_t n, const uint32_t* grp_off, uint32_t n_grp) {
    DeviceGuard dev_guard(c);
    for (uint32_t i = 0; i < n; ++i)
        if (idx[i] >= c->n_frames) return set_err(c, "select: frame index out of range");
    if (n_grp < 1) n_grp = 1;
    if (grp_off && (grp_off[0] != 0 || grp_off[n_grp] != n)) return set_err(c, "select: bad group offsets");
    if (ensure(c, c->sel, (size_t)n * 4 + 4) || ensure(c, c->grp, (size_t)n * 4 + 4) ||
        ensure(c, c->grp_off, (size_t)(n_grp + 1) * 4))
        return 1;
    if (ensure(c, c->M, (size_t)n * 24 + 24) || ensure(c, c->k, (size_t)n * 8 + 8)) return 1;
    std::vector<uint32_t> g(n, 0), off(n_grp + 1, 0);
    if (grp_off) {
        off.assign(grp_off, grp_off + n_grp + 1);
        for (uint32_t w = 0; w < n_grp; ++w)
            for (uint32_t j = off[w]; j < off[w + 1]; ++j) g[j] = w;
    } else {
        off[n_grp] = n; // single group (n_grp == 1)
    }
    if (n) {
        RS_HIP(hipMemcpyAsync(c->sel.p, idx, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        RS_HIP(hipMemcpyAsync(c->grp.p, g.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    }
    RS_HIP(hipMemcpyAsync(c->grp_off.p, off.data(), (size_t)(n_grp + 1) * 4, hipMemcpyHostToDevice, c->stream));
    RS_HIP(hipMemsetAsync(c->M.p, 0, (size_t)n * 24 + 24, c->stream));
    RS_HIP(hipMemsetAsync(c->k.p, 0, (size_t)n * 8 + 8, c->stream));
    RS_HIP(hipStreamSynchronize(c->stream));
    c->h_sel.assign(idx, idx + n);
    c->n_sel = n;
    c->n_grp = n_grp;
    c->max_n = sel_max_n(c);
    return 0;
}

int rship_select_frames(rship_ctx* c, const uint32_t* idx, uint32_t n) { return rship_select_slots(c, idx, n, nullptr, 1); }

int rship_presync_costs(rship_ctx* c, const int32_t* kd, const float* fd, uint32_t n_cand, uint32_t n_hyp,
                        uint32_t stream_base, uint64_t seed, double* costs, uint32_t* flags, double* frame_costs,
                        int32_t* best_h) {
    return rship_presync_window_costs(c, kd, fd, n_cand, n_hyp, stream_base, seed, nullptr, nullptr, 1, costs, flags,
                                      frame_costs, best_h);
}

// costs[n_cand][n_win]: window w sums the frame costs of slots seg_idx[seg_off[w] .. seg_off[w+1])
// (NULL, NULL, 1 = one window over every selected slot).  The selection must be a single group.
int rship_presync_window_costs(rship_ctx* c, const int32_t* kd, const float* fd, uint32_t n_cand, uint32_t n_hyp,
                               uint32_t stream_base, uint64_t seed, const uint32_t* seg_idx, const uint32_t* seg_off,
                               uint32_t n_win, double* costs, uint32_t* flags, double* frame_costs, int32_t* best_h) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (c->n_grp != 1) return set_err(c, "presync: the selection must not be grouped");
    if (!n_cand) return 0;
    if (n_win < 1) n_win = 1;
    const uint32_t ns = c->n_sel;
    if (ensure(c, c->frame_cost, (size_t)n_cand * ns * 8) || ensure(c, c->costs, (size_t)n_cand * n_win * 8)) return 1;
    if (best_h && ensure(c, c->best_h, (size_t)n_cand * ns * 4)) return 1;
    if (ensure(c, c->flags, 16)) return 1;
    const uint32_t* d_idx = nullptr;
    const uint32_t* d_off = nullptr;
    if (seg_off) {
        const uint32_t total = seg_off[n_win];
        for (uint32_t j = 0; seg_idx && j < total; ++j)
            if (seg_idx[j] >= ns) return set_err(c, "presync: window index out of range");
        if (ensure(c, c->seg_idx, (size_t)total * 4 + 4) || ensure(c, c->seg_off, (size_t)(n_win + 1) * 4)) return 1;
        if (seg_idx && total) RS_HIP(hipMemcpyAsync(c->seg_idx.p, seg_idx, (size_t)total * 4, hipMemcpyHostToDevice, c->stream));
        RS_HIP(hipMemcpyAsync(c->seg_off.p, seg_off, (size_t)(n_win + 1) * 4, hipMemcpyHostToDevice, c->stream));
        d_idx = seg_idx ? (const uint32_t*)c->seg_idx.p : nullptr;
        d_off = (const uint32_t*)c->seg_off.p;
    }
    if (upload_delays(c, kd, fd, n_cand)) return 1;
    RS_HIP(hipMemsetAsync(c->flags.p, 0, 16, c->stream));

    LmedsParams p{};
    p.rays_a = (const f4*)c->rays_a.p;
    p.rays_b = (const f4*)c->rays_b.p;
    p.frames = (const FrameRec*)c->frames.p;
    p.sel = (const uint32_t*)c->sel.p;
    p.n_sel = ns;
    p.coef = (const f4*)c->coef.p;
    p.n_knots = (int)c->n_knots;
    p.kd = (const int32_t*)c->kd.p;
    p.fd = c->d_fd;
    p.n_cand = n_cand;
    p.grp = nullptr;
    p.n_grp = 1;
    // enough workgroups to fill 256 CUs several times over, chunks long enough to
    // amortise the spline window (LDS) and to carry the provisional bound along
    uint64_t work = (uint64_t)n_cand * ns;
    uint32_t chunk = (uint32_t)(work / 8192);
    if (chunk < 1) chunk = 1;
    if (chunk > (uint32_t)kMaxChunk) chunk = kMaxChunk;
    if (chunk > n_cand) chunk = n_cand;
    p.chunk = chunk;
    p.n_chunks = (n_cand + chunk - 1) / chunk;
    p.n_hyp = n_hyp;
    p.stream_base = stream_base;
    p.seed = seed;
    p.frame_cost = (double*)c->frame_cost.p;
    p.best_h = best_h ? (int32_t*)c->best_h.p : nullptr;
    p.flags = (uint32_t*)c->flags.p;
    uint32_t groups = (ns + 7) / 8;
    uint64_t grid = (uint64_t)groups * 8 * p.n_chunks;
    if (grid > 0x7fffffffull) return set_err(c, "presync: grid too large");
    if (launch_lmeds<0>(c, p, rpt_for(c->max_n), (uint32_t)grid)) return 1;
    if (launch_reduce(c, p.frame_cost, (double*)c->costs.p, n_cand, ns, d_idx, d_off, n_win)) return 1;

    const size_t out_bytes = (size_t)n_cand * n_win * 8;
    if (ensure_pinned(c, out_bytes + 16)) return 1;
    RS_HIP(hipMemcpyAsync(c->pinned, c->costs.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    RS_HIP(hipMemcpyAsync((char*)c->pinned + out_bytes, c->flags.p, 4, hipMemcpyDeviceToHost, c->stream));
    if (sync_stream(c)) return 1;
    memcpy(costs, c->pinned, out_bytes);
    if (flags) memcpy(flags, (char*)c->pinned + out_bytes, 4);
    if (frame_costs) RS_HIP(hipMemcpy(frame_costs, c->frame_cost.p, (size_t)n_cand * ns * 8, hipMemcpyDeviceToHost));
    if (best_h) RS_HIP(hipMemcpy(best_h, c->best_h.p, (size_t)n_cand * ns * 4, hipMemcpyDeviceToHost));
    return 0;
}

namespace {
// per-group delays -> device (kd/fd arrays of n entries)
void fill_motion(rship_ctx* c, MotionParams& p) {
    p.rays_a = (const f4*)c->rays_a.p;
    p.rays_b = (const f4*)c->rays_b.p;
    p.frames = (const FrameRec*)c->frames.p;
    p.sel = (const uint32_t*)c->sel.p;
    p.n_sel = c->n_sel;
    p.coef = (const f4*)c->coef.p;
    p.n_knots = (int)c->n_knots;
    p.kd = (const int32_t*)c->kd.p;
    p.fd = c->d_fd;
    p.grp = c->n_grp > 1 ? (const uint32_t*)c->grp.p : nullptr;
    p.M = (double*)c->M.p;
    p.k = (const double*)c->k.p;
}
} // namespace

// kd/fd: one delay per group of the selection
int rship_init_motion(rship_ctx* c, const int32_t* kd, const float* fd, uint32_t n_hyp, uint32_t stream,
                      uint32_t stream_stride, uint64_t seed) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (upload_delays(c, kd, fd, c->n_grp) || ensure(c, c->flags, 16)) return 1;
    RS_HIP(hipMemsetAsync(c->flags.p, 0, 16, c->stream));
    LmedsParams p{};
    p.rays_a = (const f4*)c->rays_a.p;
    p.rays_b = (const f4*)c->rays_b.p;
    p.frames = (const FrameRec*)c->frames.p;
    p.sel = (const uint32_t*)c->sel.p;
    p.n_sel = c->n_sel;
    p.coef = (const f4*)c->coef.p;
    p.n_knots = (int)c->n_knots;
    p.kd = (const int32_t*)c->kd.p;
    p.fd = c->d_fd;
    p.n_cand = 1;
    p.chunk = 1;
    p.n_chunks = 1;
    p.n_hyp = n_hyp;
    p.stream_base = stream; // window w uses stream + w * stride
    p.stream_stride = stream_stride;
    p.seed = seed;
    p.grp = c->n_grp > 1 ? (const uint32_t*)c->grp.p : nullptr;
    p.n_grp = c->n_grp;
    p.M = (double*)c->M.p;
    p.k = (double*)c->k.p;
    p.flags = (uint32_t*)c->flags.p;
    uint32_t groups = (c->n_sel + 7) / 8;
    if (launch_lmeds<1>(c, p, rpt_for(c->max_n), groups * 8)) return 1;
    return sync_stream(c);
}

int rship_opt_motion_detail(rship_ctx* c, const int32_t* kd, const float* fd, uint32_t* per_frame, uint32_t cap) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (cap < c->n_sel) return set_err(c, "opt_motion_detail: output too small");
    TempBuf d;
    if (ensure(c, d, (size_t)c->n_sel * 8 + 8)) return 1;
    RS_HIP(hipMemsetAsync(d.p, 0, (size_t)c->n_sel * 8 + 8, c->stream));
    if (upload_delays(c, kd, fd, c->n_grp)) return 1;
    MotionParams p{};
    fill_motion(c, p);
    p.per_frame = (uint32_t*)d.p;
    int rc = launch_motion(c, p, rpt_for(c->max_n));
    hipError_t e = hipStreamSynchronize(c->stream);
    prof_collect(c);
    if (!rc && e == hipSuccess) e = hipMemcpy(per_frame, d.p, (size_t)c->n_sel * 8, hipMemcpyDeviceToHost);
    if (rc) return 1;
    if (e != hipSuccess) return set_err(c, "opt_motion_detail", e);
    return 0;
}

int rship_opt_motion(rship_ctx* c, const int32_t* kd, const float* fd, uint64_t* stats) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (stats) {
        if (ensure(c, c->stats, 16)) return 1;
        RS_HIP(hipMemsetAsync(c->stats.p, 0, 16, c->stream));
    }
    if (upload_delays(c, kd, fd, c->n_grp)) return 1;
    MotionParams p{};
    fill_motion(c, p);
    p.stats = stats ? (unsigned long long*)c->stats.p : nullptr;
    if (launch_motion(c, p, rpt_for(c->max_n))) return 1;
    if (stats) {
        if (ensure_pinned(c, 16)) return 1;
        RS_HIP(hipMemcpyAsync(c->pinned, c->stats.p, 16, hipMemcpyDeviceToHost, c->stream));
        if (sync_stream(c)) return 1;
        memcpy(stats, c->pinned, 16);
        return 0;
    }
    return 0; // stays queued: the next loss call is ordered behind it on the stream
}

// kd/fd: [n_delays][n_grp]; loss/grad out: [n_delays][n_grp]
int rship_loss(rship_ctx* c, const int32_t* kd, const float* fd, uint32_t n_delays, double* loss, double* grad) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (!n_delays) return 0;
    const uint32_t ns = c->n_sel, ng = c->n_grp;
    if (upload_delays(c, kd, fd, (size_t)n_delays * ng)) return 1;
    if (ensure(c, c->part, (size_t)n_delays * ns * 16)) return 1;
    LossParams p{};
    p.rays_a = (const f4*)c->rays_a.p;
    p.rays_b = (const f4*)c->rays_b.p;
    p.frames = (const FrameRec*)c->frames.p;
    p.sel = (const uint32_t*)c->sel.p;
    p.n_sel = ns;
    p.coef = (const f4*)c->coef.p;
    p.n_knots = (int)c->n_knots;
    p.fs = (float)c->fs;
    p.kd = (const int32_t*)c->kd.p;
    p.fd = c->d_fd;
    p.n_delays = n_delays;
    p.grp = ng > 1 ? (const uint32_t*)c->grp.p : nullptr;
    p.n_grp = ng;
    p.M = (const double*)c->M.p;
    p.k = (const double*)c->k.p;
    p.part_loss = (double*)c->part.p;
    p.part_grad = p.part_loss + (size_t)n_delays * ns;
    const int rpt = rpt_for(c->max_n);
    if (grad ? launch_loss<true>(c, p, rpt) : launch_loss<false>(c, p, rpt)) return 1;
    // rows [0, n_delays) = loss, [n_delays, 2 n_delays) = grad; one sum per (row, group)
    uint32_t rows = grad ? 2 * n_delays : n_delays;
    // the few sums go straight into pinned host memory (visible once the stream has drained):
    // one API call less per evaluation than a device buffer + copy
    const size_t half = (size_t)n_delays * ng * 8;
    if (ensure_pinned(c, (size_t)rows * ng * 8)) return 1;
    void* d_out = nullptr;
    RS_HIP(hipHostGetDevicePointer(&d_out, c->pinned, 0));
    if (launch_reduce(c, p.part_loss, (double*)d_out, rows, ns, nullptr, ng > 1 ? (const uint32_t*)c->grp_off.p : nullptr, ng))
        return 1;
    if (sync_stream(c)) return 1;
    memcpy(loss, c->pinned, half);
    if (grad) memcpy(grad, (char*)c->pinned + half, half);
    return 0;
}

int rship_get_motion(rship_ctx* c, double* M, double* k, uint32_t cap, uint32_t* n) {
    DeviceGuard dev_guard(c);
    if (sync_stream(c)) return 1;
    uint32_t cnt = c->n_sel < cap ? c->n_sel : cap;
    if (cnt) {
        RS_HIP(hipMemcpy(M, c->M.p, (size_t)cnt * 24, hipMemcpyDeviceToHost));
        RS_HIP(hipMemcpy(k, c->k.p, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    }
    if (n) *n = cnt;
    return 0;
}

int rship_set_motion(rship_ctx* c, const double* M, const double* k, uint32_t n) {
    DeviceGuard dev_guard(c);
    if (n != c->n_sel) return set_err(c, "set_motion: count differs from the selection");
    if (sync_stream(c)) return 1;
    if (n) {
        RS_HIP(hipMemcpy(c->M.p, M, (size_t)n * 24, hipMemcpyHostToDevice));
        RS_HIP(hipMemcpy(c->k.p, k, (size_t)n * 8, hipMemcpyHostToDevice));
    }
    return 0;
}

int rship_rays_from_pixels(rship_ctx* c, const double* px, uint64_t n_pairs, const rship_pixel_frame* frames,
                           uint32_t n_frames, uint32_t* bad) {
    DeviceGuard dev_guard(c);
    if (bad) *bad = 0;
    if (!n_frames) return 0;
    uint32_t max_n = 0;
    for (uint32_t i = 0; i < n_frames; ++i) {
        const rship_pixel_frame& f = frames[i];
        if (f.px_offset + f.n_rays > n_pairs) return set_err(c, "rays_from_pixels: frame exceeds the pixel buffer");
        if ((uint64_t)f.ray_offset + f.n_rays > c->total_rays) return set_err(c, "rays_from_pixels: frame exceeds the ray buffer");
        max_n = std::max(max_n, f.n_rays);
    }
    if (!max_n) return 0;
    TempBuf dpx, dfr, dbad;
    if (ensure(c, dpx, (size_t)n_pairs * 32 + 32) || ensure(c, dfr, (size_t)n_frames * sizeof(rship_pixel_frame)) ||
        ensure(c, dbad, 16))
        return 1;
    hipError_t e = hipMemcpyAsync(dpx.p, px, (size_t)n_pairs * 32, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dfr.p, frames, (size_t)n_frames * sizeof(rship_pixel_frame), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dbad.p, 0, 16, c->stream);
    if (e == hipSuccess) {
        PixelParams p{(const double*)dpx.p, (const rship_pixel_frame*)dfr.p, (f4*)c->rays_a.p, (f4*)c->rays_b.p, (uint32_t*)dbad.p};
        {
            ProfScope ps(c, RSHIP_K_PIXELS);
            hipLaunchKernelGGL(rays_from_pixels_kernel, dim3(n_frames, (max_n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, p);
        }
        e = hipGetLastError();
    }
    uint32_t nb = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&nb, dbad.p, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    prof_collect(c);
    if (e != hipSuccess) return set_err(c, "rays_from_pixels", e);
    if (bad) *bad = nb;
    return 0;
}

int rship_debug_rays(rship_ctx* c, uint32_t frame_index, float* a4, float* b4, uint32_t cap_rays) {
    DeviceGuard dev_guard(c);
    if (frame_index >= c->n_frames) return set_err(c, "debug_rays: index out of range");
    rship_frame rec;
    hipError_t e = hipMemcpy(&rec, (const rship_frame*)c->frames.p + frame_index, sizeof(rec), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(c, "debug_rays", e);
    if (rec.n_rays > cap_rays) return set_err(c, "debug_rays: output too small");
    e = hipMemcpy(a4, (const f4*)c->rays_a.p + rec.ray_offset, (size_t)rec.n_rays * 16, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(b4, (const f4*)c->rays_b.p + rec.ray_offset, (size_t)rec.n_rays * 16, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(c, "debug_rays", e);
    return 0;
}

int rship_debug_problem(rship_ctx* c, uint32_t sel_index, int32_t kd, float fd, float* P, float* dP, uint32_t cap_rows) {
    DeviceGuard dev_guard(c);
    if (check_ready(c)) return 1;
    if (sel_index >= c->n_sel) return set_err(c, "debug_problem: index out of range");
    uint32_t fi = c->h_sel[sel_index];
    uint32_t n = c->h_frame_n[fi];
    if (n > cap_rows) return set_err(c, "debug_problem: output too small");
    TempBuf out;
    if (ensure(c, out, (size_t)n * 24 + 64)) return 1;
    DebugParams p{};
    p.rays_a = (const f4*)c->rays_a.p;
    p.rays_b = (const f4*)c->rays_b.p;
    p.frames = (const FrameRec*)c->frames.p;
    p.fi = fi;
    p.coef = (const f4*)c->coef.p;
    p.n_knots = (int)c->n_knots;
    p.fs = (float)c->fs;
    p.kd = kd;
    p.fd = fd;
    p.P = (float*)out.p;
    p.dP = dP ? (float*)out.p + (size_t)n * 3 : nullptr;
    hipLaunchKernelGGL(debug_problem_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(P, out.p, (size_t)n * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess && dP) e = hipMemcpy(dP, (float*)out.p + (size_t)n * 3, (size_t)n * 12, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(c, "debug_problem", e);
    return 0;
}

int rship_debug_select(rship_ctx* c, const float* vals, uint32_t n_problems, uint32_t n, uint32_t kq,
                       const float* upper, uint32_t* out) {
    DeviceGuard dev_guard(c);
    if (n > 2048 || !n_problems) return set_err(c, "debug_select: bad sizes");
    TempBuf dv, du, dout;
    if (ensure(c, dv, (size_t)n_problems * n * 4) || ensure(c, dout, (size_t)n_problems * 8)) return 1;
    if (upper && ensure(c, du, (size_t)n_problems * 4)) return 1;
    hipError_t e = hipMemcpy(dv.p, vals, (size_t)n_problems * n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && upper) e = hipMemcpy(du.p, upper, (size_t)n_problems * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(debug_select_kernel, dim3(n_problems), dim3(64), 0, c->stream, (const float*)dv.p, n, kq,
                           upper ? (const float*)du.p : nullptr, (uint32_t*)dout.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, dout.p, (size_t)n_problems * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(c, "debug_select", e);
    return 0;
}

int rship_profile_enable(rship_ctx* c, int on) {
    if (sync_stream(c)) return 1;
    c->prof = on != 0;
    return 0;
}

int rship_profile_get(rship_ctx* c, int kind, uint64_t* launches, double* total_ms) {
    if (kind < 0 || kind >= RSHIP_K_COUNT) return set_err(c, "profile: bad kind");
    if (sync_stream(c)) return 1;
    if (launches) *launches = c->launches[kind];
    if (total_ms) *total_ms = c->total_ms[kind];
    return 0;
}

int rship_profile_reset(rship_ctx* c) {
    if (sync_stream(c)) return 1;
    for (int i = 0; i < RSHIP_K_COUNT; ++i) { c->launches[i] = 0; c->total_ms[i] = 0; }
    return 0;
}

} // extern "C"
