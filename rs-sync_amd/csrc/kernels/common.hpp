// common.hpp -- launch constants, wave/workgroup reductions, the LDS spline window, one row of the residual matrix
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
#pragma once

namespace {


constexpr int kBlock = 256;
constexpr int kWinMax = 80;  // knots of the spline window that is a compile-time part of a workgroup's LDS (5 KB fp32 / 10 KB fp64):
                             // a frame spans 0.044 s x gyro rate knots, so this covers rates up to ~1.7 kHz.  Higher rates run
                             // the same code with the window in DYNAMIC LDS, sized per problem (CAP = 0 below, Spline::cap).
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

// row_shr:N with bound_ctrl: every lane is written (0 where the source lane is outside the row), so no register
// has to be cleared first -- half the instructions of the old-value form for a 64-bit operand
template <int CTRL>
__device__ __forceinline__ double dpp_shr_d(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane_d(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// Sum over the 64 lanes, the same value (and the same bits) in every lane: four shift-and-add steps inside the
// rows of 16, then the four row sums R0..R3 as (R3 + R2) + (R1 + R0) -- the association of the classic
// row_bcast:15 / row_bcast:31 ending, without its masked (old-value) moves.
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_shr_d<0x111>(v);
    v += dpp_shr_d<0x112>(v);
    v += dpp_shr_d<0x114>(v);
    v += dpp_shr_d<0x118>(v);
    const double r0 = read_lane_d(v, 15), r1 = read_lane_d(v, 31), r2 = read_lane_d(v, 47), r3 = read_lane_d(v, 63);
    return (r3 + r2) + (r1 + r0);
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

// the same for a workgroup of NW waves (NW = 4: the very association of block_sum)
template <int NW>
__device__ __forceinline__ double block_sum_n(float v, double* slot) {
    float w = wave_sum_f32(v);
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = (double)w;
    __syncthreads();
    double t = slot[0] + slot[1];
#pragma unroll
    for (int i = 2; i < NW; ++i) t += slot[i];
    return t;
}

// Data that other workgroups write DURING the launch (the window executor, executor.hpp): per-XCD L2s are not
// coherent and a CU's L1 is never refreshed, so such words are loaded and stored with `sc1` (relaxed agent-scope
// atomics: the access goes past L1 and is served / written through coherently) on both sides, and a signal (a
// counter add, a queue cell) is sent only after `s_waitcnt vmcnt(0)`.  SC1 = false: the plain access of every other kernel.
template <bool SC1, class T>
__device__ __forceinline__ T ld_m(const T* p) {
    if (SC1) return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool SC1, class T>
__device__ __forceinline__ void st_m(T* p, T v) {
    if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ bool finite_f(float x) { return (__float_as_uint(x) & kInfBits) != kInfBits; }

// a * b where a zero factor wins over NaN and infinity (v_mul_legacy_f32, the DX9 rule; every other product is
// v_mul_f32's, bit for bit: tools/ubench/mul_legacy.hip).  Stage D of the LMedS kernels multiplies a row's norm -- 0 for
// the rows beyond the frame -- with a dot product that is NaN for exactly those rows (their tile entries are NaN so
// that they never count below a threshold): one instruction instead of a product, a compare and a select.
__device__ __forceinline__ float mul_zero_wins(float a, float b) {
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct FrameRec { // == rship_frame
    uint32_t off, n;
    int32_t base_knot;
    float tmin, tmax;
    uint32_t range_a; // knots the a-end touches at delay 0, relative to base_knot: lo | hi << 16 (RSHIP_NO_SPLIT: unknown)
    int64_t id;
    double tmin64, tmax64;
    uint32_t range_b;
    uint32_t reserved;
};
static_assert(sizeof(FrameRec) == sizeof(rship_frame), "frame record layout");

// The knots a frame's spline window must hold while the integer parts of the delays span [kd_lo, kd_hi]: the whole
// pair [lo, hi] (the caller's, from tmin / tmax) and -- where the table knows them -- the two ENDS' ranges.  ts_a lies
// within one read-out time of the current frame and ts_b of the next (core_testcode.cpp:144-145), a frame interval
// apart: at high gyro rates the two ends together cover half the knots of the pair or fewer.
struct FrameKnots {
    int lo, hi;
    int a_lo, a_hi, b_lo, b_hi;
    bool split;
};
__device__ __forceinline__ FrameKnots frame_knots(const FrameRec& fr, int lo, int hi, int kd_lo, int kd_hi) {
    FrameKnots k;
    k.lo = lo; k.hi = hi;
    k.split = fr.range_a != RSHIP_NO_SPLIT && fr.range_b != RSHIP_NO_SPLIT;
    k.a_lo = fr.base_knot + (int)(fr.range_a & 0xffffu) + kd_lo;
    k.a_hi = fr.base_knot + (int)(fr.range_a >> 16) + kd_hi + 1; // (+1: the delay's fraction can carry into the next knot)
    k.b_lo = fr.base_knot + (int)(fr.range_b & 0xffffu) + kd_lo;
    k.b_hi = fr.base_knot + (int)(fr.range_b >> 16) + kd_hi + 1;
    return k;
}

// ---------------------------------------------------------------------------
// spline window in LDS: SoA by coefficient kind so that neighbouring knots
// fall into different banks (ds_read_b128 of kind k, knot j at (k*kWinMax + j) * 16 B).

struct Spline {
    const f4* __restrict__ g; // global table, 4 f4 per knot
    const f4* lds;            // [4][capacity]
    int n;                    // knots
    int w0, wlen;             // staged range [w0, w0 + wlen)
    int path;                 // kPathGlobal / kPathLds / kPathInterior, uniform over the workgroup
    int cap;                  // capacity of the window when it is not a compile-time constant (CAP = 0)
    bool whole_pair;          // CAP = 0 only: never stage the two ends separately (set by the caller; false = allowed)
    int w0b;                  // CAP = 0 only: knot index that maps to window slot 0 for the B END's fetches (two ranges
                              // staged one after the other: stage_window_ends); = w0 where one range is staged
};

// How a workgroup reads spline coefficients.  The choice is made once per workgroup from the knot
// range it can touch, so that the hot loops carry no per-lane LDS-or-global selection (which
// would turn ds_read_b128 into flat loads) and, in the common case, no extrapolation logic.
constexpr int kPathGlobal = 0;   // general: any parameter (extrapolation branches included), table read from L2
constexpr int kPathInterior = 2; // staged in LDS and strictly inside the knots (0 <= idx <= n-2)

// CAP = knots the LDS window holds: a compile-time constant (it is the stride between the four coefficient
// kinds, folded into the ds_read offsets) or 0 = s.cap, set by the caller (a window in dynamic LDS sized for the
// problem's gyro rate: three more address additions per fetch)
template <int CAP = kWinMax>
__device__ __forceinline__ void stage_window(Spline& s, f4* s_win, int lo, int hi, int n_threads = kBlock) {
    const int cap = CAP ? CAP : s.cap;
    const int n = s.n;
    const bool interior = lo >= 0 && hi <= n - 2;
    lo = lo < 0 ? 0 : (lo > n - 1 ? n - 1 : lo);
    hi = hi < 0 ? 0 : (hi > n - 1 ? n - 1 : hi);
    int wlen = hi - lo + 1;
    s.path = (wlen <= cap && interior) ? kPathInterior : kPathGlobal;
    if (wlen > cap) wlen = cap;
    s.w0 = lo;
    s.w0b = lo;
    s.wlen = wlen;
    s.lds = s_win;
    for (int e = threadIdx.x; e < wlen * 4; e += n_threads) {
        int knot = e >> 2, kind = e & 3;
        s_win[kind * cap + knot] = s.g[(size_t)(lo + knot) * 4 + kind];
    }
}

// The window of the dynamic-LDS instantiations (CAP = 0): where the frame table knows the two ends' ranges, they are
// disjoint and interior and together fit the capacity, the a-end's knots go to slots [0, lenA) and the b-end's to
// [lenA, lenA + lenB) -- fetches of the b end subtract w0b instead of w0 -- and the knots between them are not staged at
// all; in every other case the whole pair, exactly as stage_window does.  Compile-time windows (the 80-knot kernels the
// benchmark runs) always take the whole pair: their code does not change.
template <int CAP>
__device__ __forceinline__ void stage_window_ends(Spline& s, f4* s_win, const FrameKnots& k, int n_threads = kBlock) {
    if constexpr (CAP == 0) {
        const int cap = s.cap, n = s.n;
        const int lenA = k.a_hi - k.a_lo + 1, lenB = k.b_hi - k.b_lo + 1;
        const bool disjoint = k.b_lo > k.a_hi + 1 || k.a_lo > k.b_hi + 1;
        const bool interior = k.a_lo >= 0 && k.b_lo >= 0 && k.a_hi <= n - 2 && k.b_hi <= n - 2;
        if (k.split && !s.whole_pair && disjoint && interior && lenA + lenB <= cap && lenA + lenB < k.hi - k.lo + 1) {
            s.path = kPathInterior;
            s.w0 = k.a_lo;
            s.w0b = k.b_lo - lenA;
            s.wlen = lenA + lenB;
            s.lds = s_win;
            for (int e = threadIdx.x; e < (lenA + lenB) * 4; e += n_threads) {
                const int slot = e >> 2, kind = e & 3;
                const int knot = slot < lenA ? k.a_lo + slot : k.b_lo + (slot - lenA);
                s_win[kind * cap + slot] = s.g[(size_t)knot * 4 + kind];
            }
            return;
        }
    }
    stage_window<CAP>(s, s_win, k.lo, k.hi, n_threads);
}

template <int PATH, int CAP = kWinMax, bool END_B = false>
__device__ __forceinline__ void fetch_coef(const Spline& s, int ci, f4& y, f4& b, f4& c, f4& d) {
    if (PATH == kPathGlobal) {
        const f4* p = s.g + (size_t)ci * 4;
        y = p[0]; b = p[1]; c = p[2]; d = p[3];
    } else {
        const int cap = CAP ? CAP : s.cap;
        const int rel = ci - ((CAP == 0 && END_B) ? s.w0b : s.w0);
        y = s.lds[rel];
        b = s.lds[cap + rel];
        c = s.lds[2 * cap + rel];
        d = s.lds[3 * cap + rel];
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
// NEWTON (hot path only): the interpolated quaternions are within ~1e-5 of unit length (unit knots a few
// milliradians apart), so 1/|q|^2 = 2 - |q|^2 up to (1 - |q|^2)^2 -- below fp32 rounding while |1 - |q|^2| < 2^-12
// -- and the two v_rcp_f32 (quarter rate) with their clamps become one packed fma.  *qerr collects
// max |1 - |q|^2| over the rows a thread computes; the caller redoes them without NEWTON if any lane of the
// wave exceeds kNewtonMaxErr (non-unit knots: the reference normalises whatever it is given, ndspline.cpp:21-27).
constexpr float kNewtonMaxErr = 2.44140625e-4f; // 2^-12: (2^-12)^2 = 2^-24 relative, half an fp32 ulp

template <bool DERIV, int PATH, bool SWEEP = false, int CAP = kWinMax, bool NEWTON = false>
__device__ __forceinline__ void residual_row(const Spline& s, f4 A, f4 B, int base, float fd, f3& P, f3& dP, float* qerr = nullptr) {
    f4 ya, ba, ca, da, yb, bb, cb, db;
    rs::Knot ka = (PATH == kPathInterior) ? (SWEEP ? locate_sweep(B.z, base, fd) : rs::spline_locate_interior(B.z, base, fd))
                                          : rs::spline_locate(B.z, base, fd, s.n);
    fetch_coef<PATH, CAP>(s, ka.ci, ya, ba, ca, da);
    rs::Knot kb = (PATH == kPathInterior) ? (SWEEP ? locate_sweep(B.w, base, fd) : rs::spline_locate_interior(B.w, base, fd))
                                          : rs::spline_locate(B.w, base, fd, s.n);
    fetch_coef<PATH, CAP, true>(s, kb.ci, yb, bb, cb, db);
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
        v2f sc;
        if (NEWTON) {
            const v2f e = v2f{1.f, 1.f} - n2;
            sc = e * 2.f + v2f{2.f, 2.f}; // 2 (2 - n2)
            *qerr = __builtin_fmaxf(*qerr, __builtin_fmaxf(__builtin_fabsf(e.x), __builtin_fabsf(e.y)));
        } else {
            sc = v2f{2.f * rs::rcp_fast(fmaxf(n2.x, 1e-30f)), 2.f * rs::rcp_fast(fmaxf(n2.y, 1e-30f))};
        }
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

} // namespace
