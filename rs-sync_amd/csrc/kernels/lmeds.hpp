// lmeds.hpp -- K2: the LMedS tile kernel (PreSync sweep; GuessMotion/GuessK in INIT mode)
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
#pragma once

namespace {

// Dynamic trip counts for the instruction budget of K2 (profiles/r3_k2_budget.md): a variant build
// (-DRSSYNC_K2_COUNTERS=1, tools/k2_build_variant.sh) counts, per wave, how often each part of stage C runs.
#ifndef RSSYNC_K2_COUNTERS
#define RSSYNC_K2_COUNTERS 0
#endif
#if RSSYNC_K2_COUNTERS
__device__ unsigned long long g_k2_counters[16];
#define K2_COUNT(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_k2_counters[i], 1ull); } while (0)
#else
#define K2_COUNT(i) do { } while (0)
#endif
// -DRSSYNC_K2_TIMING=1 (with the counters): per wave, core-clock ticks spent waiting at the candidate loop's
// barriers [10] and in the whole kernel [11]; by barrier: tile written [12], directions ready [13], sweeps done [14],
// the two workgroup sums of stage D [15]   (tools/gpu_k2_counters.py prints the raw array)
#ifndef RSSYNC_K2_TIMING
#define RSSYNC_K2_TIMING 0
#endif
#if RSSYNC_K2_TIMING && RSSYNC_K2_COUNTERS
#define K2_SYNC(slot) do { const long long t__ = clock64(); __syncthreads(); const long long d__ = clock64() - t__; k2_bar += d__; k2_by[slot] += d__; } while (0)
#define K2_TIMED(slot, expr) do { const long long t__ = clock64(); expr; const long long d__ = clock64() - t__; k2_bar += d__; k2_by[slot] += d__; } while (0)
#else
#define K2_SYNC(slot) __syncthreads()
#define K2_TIMED(slot, expr) expr
#endif
// 0 (frame, candidate) pairs   1 queue pops   2 hypotheses swept   3 sweeps that beat the bound (exact selections)
// 4 counting passes inside the exact selection   5 selections ended by the single-element min pass
// 6 candidates redone without the provisional bound   7 sweeps that started without any bound (wave max)
// 8 contenders swept again and closed exactly (lazy selection: overlapping brackets)
// 9 sweeps that needed their last group of rows (early rejection did not apply)

// ---------------------------------------------------------------------------
// K2: LMedS tile kernel

struct LmedsParams {
    const f4* rays_a;
    const f4* rays_b;
    const FrameRec* frames;
    const uint32_t* sel;
    uint32_t n_sel;
    // The slots THIS launch works on: entry i of the launch is slot slots[i] (null: i itself), i < n_slots.  A launch covers
    // the slots of one SIZE CLASS of the selection (rssync_kernels.hip: a frame's kernel family follows its own track
    // count, whatever else the problem holds); results are indexed by slot (stride n_sel) as before.
    const uint32_t* slots;
    uint32_t n_slots;
    const f4* coef;
    int n_knots;
    const int32_t* kd;
    const float* fd;
    uint32_t n_cand, chunk, n_chunks;
    uint32_t n_hyp, stream_base, stream_stride; // sampler stream = base + candidate + group * stride
    const uint32_t* win_stream; // or, if not null (window executor): win_stream[group] + candidate
    uint64_t seed;
    const uint32_t* grp; // slot -> group (window) or null; delays are indexed [candidate][group]
    uint32_t n_grp;      // >= 1
    double* frame_cost; // [n_cand][n_sel]
    int32_t* best_h;    // [n_cand][n_sel] or null; INIT mode: [n_sel], the winning hypothesis per slot (-1 = none),
                        // from which opt_motion64_kernel recomputes M and k in fp64
    uint32_t* flags;
    // frames of more than 8192 tracks (kernels/lmeds_big.hpp): the workgroups' tiles in global memory,
    // gridDim.x x scratch_rows x 5 floats
    float* scratch;
    uint32_t scratch_rows;
    // knots the spline window in DYNAMIC LDS holds (the WIN = 0 / CAP = 0 instantiations: gyro rates whose frames
    // span more than kWinMax knots); the launch passes win_cap * 64 bytes of dynamic LDS
    uint32_t win_cap;
    uint32_t win_whole_pair; // 1: a dynamic window behaves like the compiled-in one -- whole pairs only (the window executor's
                             // search where the launch chain's search kernel uses the compiled-in window)
    // ---- NEAR-STATIC frames: rows in fp64 (round 6; MODE 0 only, "fp64 rows" below) ----
    // redo_mask (or null: the watch is off): one bit per (slot, candidate), word slot * mask_words + (c >> 5), bit c & 31.
    // The fp32 sweep SETS the bit of a pair whose rows are too small for fp32 inputs; the R64 instantiations -- launched
    // over the same grid once the host has seen RSHIP_NEAR_STATIC -- recompute exactly those pairs from the fp64 streams
    // (src64, delays kd64 / fd64 [n_cand]), overwrite frame_cost / best_h, and clear the bits they served.
    uint32_t* redo_mask;
    uint32_t mask_words;
    Rows64Src src64;
    const int32_t* kd64;
    const double* fd64;
    unsigned long long* redo_count; // [0] PreSync pairs recomputed in fp64 so far, [1] GuessMotion searches that took the fp64 form (debug ABI: rship_near_static_stats)
    // ---- TEST-VARIANTS build only (-DRSSYNC_TEST_VARIANTS=1; the product's kernels have no such code) ----
    // dump (or null): the |residual| bit patterns of the sweep itself, [candidate][slot][hypothesis][dump_rows] -- every
    // hypothesis of every (frame, candidate) against the tile and the directions the selection worked on -- so that a test
    // can demand, WITHOUT A TOLERANCE, that the winner is the exact arg-min of the sorted residuals' lower quartile with
    // the reference's first-wins rule (core_private.cpp:48-56; tests/test_gpu_fuzz.py: the anchor under flip_interval)
    uint32_t* dump;
    uint32_t dump_rows;
};

// fp64 ROWS FOR NEAR-STATIC FRAMES (round 6).  The reference computes the rows of P, their norms and the safe_normalize
// decisions in double (core_private.cpp:19-28,45-46, inline_utils.hpp:5-11).  The sweep's inputs are fp32 -- a ray
// component carries 6e-8 absolute -- which is 3e-5 of an ordinary row (|P| ~ 2e-3: translation / depth) but 3 % of a row
// of 2e-6: for a camera on a tripod or in a slow pan the fp32 sweep picked another LMedS winner than the reference in up to
// 19 % of the (frame, candidate) pairs (profiles/r5_near_static.json).  The fp64 streams that Sync reads are resident
// anyway.  So the sweep WATCHES for such pairs -- of the frame's first 64 rows, a quarter or more with |P|^2 below
// kNearStatic2: one v_cmp and one s_bcnt1 per wave and candidate in the hot kernel, nothing per row -- and flags them; the
// host, when it collects the sweep and sees the flag, launches the R64 form of the same kernels, which recompute only the
// flagged pairs: rows from the fp64 streams and the fp64 table (rows64.hpp: row64_unit -- norm, unit row and the
// safe_normalize test in double), rounded ONCE to the fp32 tile and fp32 norms; hypotheses (with the rows' fp64 norms in
// the hypothesis rule), sweeps, selection and stage D are the fp32 code, untouched.  An ordinary scene never sets a bit
// and never sees the second launch (rship_near_static_stats counts the pairs; the tests assert 0 on ordinary scenes).
// The sample is the first min(N, 64) rows in every kernel family, so the decision is the frame's and the candidate's.
#ifndef RSSYNC_NEAR_WATCH   // (-DRSSYNC_NEAR_WATCH=0: the sweep kernels without the watch -- round 5's code -- for the A/B of profiles/r6_k2_rows64_ab.txt)
#define RSSYNC_NEAR_WATCH 1
#endif
constexpr float kNearStatic2 = 4e-8f; // |P| < 2e-4: where profiles/r5_near_static.json first falls below 99 % (median |P| 1.2e-4)
__device__ __forceinline__ bool near_static_fires(uint32_t near, uint32_t N) { return 4u * near >= (N < 64u ? N : 64u); }

// ---- LMedS tile in LDS, struct-of-arrays: unit rows n = safe_normalize(P).  The norms |P|
// stay in the registers of the thread that owns the row (only stage D needs them).
template <class P> // pointer to float: generic (the kernels' LDS or global tiles), or address-space qualified (exec_big.hpp)
struct TileP {
    P nx;
    P ny;
    P nz;
};
using Tile = TileP<float*>;

// hypothesis direction v = safe_normalize(P[i0] x P[i1]) (core_private.cpp:45-46, inline_utils.hpp:5-11) on the
// UN-NORMALISED rows, as the reference has it: |P[i0] x P[i1]| < 1e-12 leaves v as it is -- a tiny vector whose residuals
// are scaled down with it and which therefore WINS the LMedS outright (a near-static camera: |P| ~ 1e-6).  The tile
// holds unit rows n_i with P_i = s_i n_i (s_i = |P_i|, or 1 for a row safe_normalize left alone), so
// P[i0] x P[i1] = s0 s1 (n0 x n1) and the rule reads  s0 s1 |n0 x n1| < 1e-12.  (Rounds 1-4 applied the threshold to
// |n0 x n1|: deviation 5 of DESIGN.md, gone.)  The norms stay in the registers of the threads that own the rows, so:
//   * smin2 = a lower bound of s_i s_j over ALL rows of the frame (0.999 x the smallest |P|^2 of stage A, 0 if any row is
//     below safe_normalize's threshold); nn smin2 >= 1e-12 decides the common case -- every hypothesis of an ordinary
//     scene -- with one multiplication;
//   * otherwise the two rows' norms are recomputed from the rays (scale(row) = row_scale_general below: the general form of
//     the row with the coefficients from the table, the same in every kernel family and on every spline path, so that
//     they all take the same decision) and the reference's rule is applied exactly.
template <class P, class ScaleFn>
__device__ __forceinline__ f3 hypothesis(const TileP<P>& t, uint64_t seed, int64_t frame, uint32_t stream, uint32_t h,
                                         uint32_t n, float smin2, ScaleFn&& scale) {
    uint32_t i0, i1;
    // (the frame's part of the sampler's hash is the same for every candidate and hypothesis: left alone, the compiler
    // computes it once and keeps a register pair alive through the whole tile kernel, which has none to spare -- two
    // 64-bit multiplications per call are cheaper than that pair's spill)
    asm volatile("" : "+s"(seed));
    rs::sample_pair(seed, frame, stream, h, n, i0, i1);
    f3 v = rs::cross(f3{t.nx[i0], t.ny[i0], t.nz[i0]}, f3{t.nx[i1], t.ny[i1], t.nz[i1]});
    float nn = sqrtf(rs::dot(v, v));
    if (nn * smin2 >= 1e-12f) { // |P[i0] x P[i1]| >= 1e-12 for certain
        float inv = 1.0f / nn;
        v = rs::scale(v, inv);
    } else {
        // (one row after the other, not unrolled: the two recomputations side by side would cost the tile kernel -- at
        // its 96 registers -- spills around this rare branch)
        float ss = 1.f;
#pragma unroll 1
        for (int e = 0; e < 2; ++e) ss *= scale(e == 0 ? i0 : i1);
        if (!(ss * nn < 1e-12f)) { // (a NaN is "normalised", as the reference's strict < does)
            float inv = 1.0f / nn;
            v = rs::scale(v, inv);
        } else {
            v = rs::scale(v, ss); // P[i0] x P[i1] itself
        }
    }
    return v;
}
// |P_row| as safe_normalize sees it (1 for a row it leaves alone)
__device__ __forceinline__ float row_scale_of(f3 P) {
    const float n2 = rs::dot(P, P);
    return n2 < 1e-24f ? 1.f : n2 * rs::rsqrt_fast(n2);
}
// ... of the row with rays A = {ax,bx,ay,by}, B = {az,bz,ta,tb}: the general form of the row (any parameter, coefficients
// from the table in L2), one END after the other and not unrolled -- this runs inside the tile kernel, which sits at its
// 96 registers, on a branch an ordinary scene takes once in a million hypotheses
__device__ __forceinline__ float row_scale_general(const f4* __restrict__ coef, int n_knots, f4 A, f4 B, int base, float fd) {
    f3 ar = f3{0, 0, 0}, br = f3{0, 0, 0};
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const float t = e ? B.w : B.z;
        const f3 ray = e ? f3{A.y, A.w, B.y} : f3{A.x, A.z, B.x};
        const rs::Knot k = rs::spline_locate(t, base, fd, n_knots);
        const f4* q = coef + (size_t)k.ci * 4;
        f3 r, d;
        rs::rotate_ray<false>(q[0], q[1], q[2], q[3], k, ray, r, d);
        if (e) br = r; else ar = r;
    }
    return row_scale_of(rs::cross(ar, br));
}
// the bound smin2 of hypothesis() from the smallest |P|^2 (bit pattern) over the frame's rows
__device__ __forceinline__ float smin2_of(uint32_t n2min_bits) {
    return n2min_bits < __float_as_uint(1e-24f) ? 0.f : __uint_as_float(n2min_bits) * 0.999f;
}

// wave-wide count of |r[]| < pivot (pivot: bit pattern of a non-negative float, uniform).
// Each register costs ONE v_cmp (the abs modifier is free, NaN never counts) whose 64-lane mask is counted
// with s_bcnt1_i32_b64 and added on the scalar unit; the total arrives in an SGPR, so no cross-lane
// reduction is needed.  (Measured alternative, round 2: v_cmp into VCC + v_addc_co_u32 into a per-lane
// counter for 3 of every 8 registers, to take load off the scalar unit -- two scalar instructions per
// register run at ~6 cycles per register per SIMD, tools/ubench/valu_rate.hip -- changed the kernel by
// -1.5 % .. +0.5 % depending on the build: not kept.)
template <int NR, int R0 = 0, int R1 = NR> // registers [R0, R1)
__device__ __forceinline__ uint32_t wave_count_lt(const uint32_t (&r)[NR], uint32_t pivot) {
    const float pv = __uint_as_float(pivot);
    uint32_t cnt = 0;
#pragma unroll
    for (int m = R0; m < R1; ++m)
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

// Quartile selection on the wave's |r[]| (NR registers per lane = the whole tile).  |r| orders exactly like the
// r^2 the reference sorts (core_private.cpp:49-52), so the kq-th smallest |r| (0-based) is the element std::sort
// would leave at index kq, before squaring.  Bit patterns of non-negative floats order like unsigned integers.
//
// A BRACKET [lo, hi) with counts c_lo = count(|r| < lo) <= kq < c_hi = count(|r| < hi) contains that element.
// narrow_kth narrows it by counting passes (wave_count_lt): pivots come from a secant step on the empirical CDF of
// |r| (close to uniform around the lower quartile, so the CDF is nearly linear there), with bracket interpolation
// and plain bisection of the bit pattern as fallbacks.  It stops
//   * when the bracket is one bit pattern wide, hi = lo + 1: lo IS the element (a bracket that holds exactly one
//     element is closed by a min pass that extracts it) -- always, if stop_elems == 0;
//   * or, if stop_elems > 0, as soon as hi < hi_limit and the bracket holds at most stop_elems elements: the
//     caller only needs to know the element that closely (lazy selection, below).
// All bookkeeping is wave-uniform and kept on the scalar unit; only the secant formula itself runs on the VALU.
struct Bracket {
    uint32_t lo, c_lo, hi, c_hi;
};
template <int NR>
__device__ __forceinline__ void narrow_kth(const uint32_t (&r)[NR], uint32_t kq, Bracket& b, uint32_t stop_elems, uint32_t hi_limit) {
    uint32_t lo = b.lo, c_lo = b.c_lo, hi = b.hi, c_hi = b.c_hi;
    uint32_t a1 = lo, c1 = c_lo, a2 = hi, c2 = c_hi; // the two most recent (pivot, count) points
    for (int it = 0;; ++it) {
        if (hi - lo == 1u) break;
        if (stop_elems && hi < hi_limit && c_hi - c_lo <= stop_elems) break;
        if (c_hi - c_lo == 1u) {
            K2_COUNT(5);
            // the single element in [lo, hi): smallest |x| >= lo; |x| < lo wraps to a huge difference
            uint32_t mn = 0xffffffffu;
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                uint32_t d = (r[m] & 0x7fffffffu) - lo;
                mn = d < mn ? d : mn;
            }
            lo += wave_min_u32(mn);
            hi = lo + 1u;
            break;
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
        K2_COUNT(4);
        a1 = a2; c1 = c2;
        a2 = piv; c2 = c;
        if (c <= kq) { lo = piv; c_lo = c; }
        else { hi = piv; c_hi = c; }
    }
    b.lo = lo; b.c_lo = c_lo; b.hi = hi; b.c_hi = c_hi;
}
// the exact kq-th smallest, given an exclusive upper bound hi with count(|r| < hi) = c_hi > kq
template <int NR>
__device__ __forceinline__ uint32_t select_kth(const uint32_t (&r)[NR], uint32_t kq, uint32_t hi, uint32_t c_hi) {
    Bracket b{0u, 0u, hi, c_hi};
    narrow_kth(r, kq, b, 0u, 0u);
    return b.lo;
}

// The frame's two ray streams as buffer resources: a row is addressed as (descriptor in SGPRs) + (one per-thread
// byte offset, tid * 16) + (a per-row scalar offset, j * 4096), so the eight rows of a thread need ONE address
// register instead of eight 64-bit pointers -- the kernel sits at the 96-VGPR limit of five waves per SIMD,
// and hipcc had hoisted those pointers out of the candidate loop and spilled them.  Rows beyond the frame
// read as zero (hardware range check).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct RayRsrc {
    __amdgpu_buffer_rsrc_t a, b;
};
__device__ __forceinline__ RayRsrc make_ray_rsrc(const f4* rays_a, const f4* rays_b, uint32_t n) {
    // the inputs are uniform over the workgroup (frame record via blockIdx); say so explicitly, or every load
    // is wrapped in a waterfall loop
    auto uni = [](const f4* p) {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        return (void*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    const int bytes = __builtin_amdgcn_readfirstlane((int)(n * 16u));
    RayRsrc r;
    r.a = __builtin_amdgcn_make_buffer_rsrc(uni(rays_a), 0, bytes, 0x00020000);
    r.b = __builtin_amdgcn_make_buffer_rsrc(uni(rays_b), 0, bytes, 0x00020000);
    return r;
}
__device__ __forceinline__ f4 load_ray(__amdgpu_buffer_rsrc_t rs_, uint32_t voff, uint32_t soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)voff, (int)soff, 0);
    return f4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}

// stage A of the LMedS kernel: this thread's rows of P for one delay, written to the LDS tile as
// unit rows, norms kept in nrm[]; returns RSHIP_BAD_P if a row is not finite.  Rows >= N are not
// touched: the kernel fills them with NaN once (their residuals compare above every threshold).
// what a thread's hot-path rows leave behind instead of per-row checks: the largest |1 - |q|^2| (residual_row's NEWTON),
// the smallest |P|^2 and the sum of the norms |P| -- a row below safe_normalize's 1e-12 or a quaternion off unit length
// sends the wave through the careful form of the rows again, a non-finite sum is a non-finite P (inf and NaN both end
// as NaN norms: rsqrt(inf) = 0, inf x 0)
struct RowWatch {
    float qerr = 0.f, nsum = 0.f;
    uint32_t n2min = 0x7f000000u; // bit pattern of the smallest |P|^2: non-negative floats order like their patterns, a NaN's
                                  // lies above them all (and an integer minimum needs no canonicalising v_max before it)
    uint32_t near = 0; // wave-uniform: how many of the wave's FIRST rows (j = 0: rows 0 .. 63 in wave 0) have |P|^2 < kNearStatic2
    __device__ __forceinline__ bool below_safe_normalize() const { return n2min < __float_as_uint(1e-24f); }
};

template <int PATH, bool SWEEP, int CAP, bool FAST = false>
__device__ __forceinline__ uint32_t lmeds_row(const Spline& sp, f4 A, f4 B, uint32_t N, uint32_t row, int base, float fd,
                                              const Tile& tile, float& nrm, RowWatch* watch = nullptr, bool first = false) {
    uint32_t bad = 0;
    nrm = 0.f;
    if (row < N) {
        f3 P, dP;
        residual_row<false, PATH, SWEEP, CAP, FAST>(sp, A, B, base, fd, P, dP, FAST ? &watch->qerr : nullptr);
        const float n2 = rs::dot(P, P);
        // the near-static watch (above): of the wave's first rows -- the lanes active here -- how many are tiny
        if (RSSYNC_NEAR_WATCH && first && watch) watch->near = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(n2 < kNearStatic2));
        if (FAST) {
            // (round 4: five instructions per row -- a class test and an OR for non-finite rows, a compare and two selects
            // for safe_normalize's rule -- became a minimum and an addition; profiles/r4_k2_rowwatch_ab.txt)
            const float inv = rs::rsqrt_fast(n2);
            tile.nx[row] = P.x * inv; tile.ny[row] = P.y * inv; tile.nz[row] = P.z * inv;
            nrm = n2 * inv;
            watch->n2min = min(watch->n2min, __float_as_uint(n2));
            watch->nsum += nrm;
        } else {
            if (!finite_f(n2)) bad = RSHIP_BAD_P;
            // safe_normalize (core_private.cpp:35-36): rows with |P| < 1e-12 stay as they are
            const bool tiny = n2 < 1e-24f;
            const float inv = tiny ? 1.f : rs::rsqrt_fast(n2);
            tile.nx[row] = P.x * inv; tile.ny[row] = P.y * inv; tile.nz[row] = P.z * inv;
            nrm = tiny ? 1.f : n2 * inv;
            if (watch) watch->n2min = min(watch->n2min, __float_as_uint(n2)); // (hypothesis(): the smallest |P|^2 of the frame)
        }
    }
    return bad;
}

// n2min: the smallest |P|^2 (bit pattern) of this thread's rows, for hypothesis()'s bound
template <int RPT, bool SWEEP, int CAP, int BLOCK = kBlock> // BLOCK: threads of the workgroup (row j of thread t = j * BLOCK + t)
__device__ __forceinline__ uint32_t lmeds_rows(const Spline& sp, const RayRsrc& rays, uint32_t N, int base, float fd,
                                               const Tile& tile, float (&nrm)[RPT], uint32_t& n2min, uint32_t& near) {
    uint32_t bad = 0;
    const uint32_t voff = threadIdx.x * 16u;
    if (sp.path == kPathInterior) {
        RowWatch watch;
        // (round 4, measured and dropped: the groups of 256 rows that lie inside the frame altogether -- all of them at 2048
        // tracks -- without the per-lane `row < N`, by a scalar test on N; three instructions fewer per row, but the kernel
        // sits at its 96 registers: the build spilled two values and ran 0.6 ms per launch SLOWER, profiles/r4_k2_allrows_ab.txt)
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const f4 A = load_ray(rays.a, voff, (uint32_t)j * BLOCK * 16u), B = load_ray(rays.b, voff, (uint32_t)j * BLOCK * 16u);
            (void)lmeds_row<kPathInterior, SWEEP, CAP, true>(sp, A, B, N, j * BLOCK + threadIdx.x, base, fd, tile, nrm[j], &watch, j == 0);
        }
        if (!finite_f(watch.nsum)) bad = RSHIP_BAD_P;
        // never, for orientations and rays that move: redo the wave's rows with the reciprocal and safe_normalize's select
        // (a non-finite row keeps its flag either way: the careful form tests it per row)
        if (__builtin_amdgcn_ballot_w64(watch.qerr >= kNewtonMaxErr || watch.below_safe_normalize()) != 0) {
            bad = 0;
            // (not unrolled: rare code kept small)
#pragma unroll 1
            for (int j = 0; j < RPT; ++j) {
                const f4 A = load_ray(rays.a, voff, (uint32_t)j * BLOCK * 16u), B = load_ray(rays.b, voff, (uint32_t)j * BLOCK * 16u);
                float t;
                bad |= lmeds_row<kPathInterior, SWEEP, CAP, false>(sp, A, B, N, j * BLOCK + threadIdx.x, base, fd, tile, t);
                nrm[j] = t;
            }
        }
        n2min = watch.n2min; // (of the hot form: a row below the threshold shows there too)
        near = watch.near;
    } else { // rare (ends of the gyro track, wild delays): keep the code small, not fast
        float tmp[RPT];
        RowWatch watch;
#pragma unroll 1
        for (int j = 0; j < RPT; ++j) {
            const f4 A = load_ray(rays.a, voff, (uint32_t)j * BLOCK * 16u), B = load_ray(rays.b, voff, (uint32_t)j * BLOCK * 16u);
            bad |= lmeds_row<kPathGlobal, false, CAP>(sp, A, B, N, j * BLOCK + threadIdx.x, base, fd, tile, tmp[j], &watch, j == 0);
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) nrm[j] = tmp[j];
        n2min = watch.n2min;
        near = watch.near;
    }
    return bad;
}

// stage A in its fp64 form (R64 instantiations: "fp64 rows" above): this thread's rows from the fp64 streams, unit rows to
// the tile, norms to s_nrm (LDS, so that the loop need not be unrolled and hypothesis() can look any row's norm up)
template <int RPT, int BLOCK = kBlock>
__device__ __forceinline__ uint32_t lmeds_rows64(const Rows64Src& src, uint32_t off, uint32_t N, int base, double fd, const Tile& tile,
                                                 float* s_nrm, uint32_t& n2min) {
    uint32_t bad = 0;
    n2min = 0x7f000000u;
#pragma unroll 1
    for (int j = 0; j < RPT; ++j) {
        const uint32_t row = j * BLOCK + threadIdx.x;
        if (row < N) {
            const Row64 r = row64_unit(src, (size_t)off + row, base, fd);
            if (!r.finite) bad = RSHIP_BAD_P;
            tile.nx[row] = r.n.x; tile.ny[row] = r.n.y; tile.nz[row] = r.n.z;
            s_nrm[row] = r.nrm;
            n2min = min(n2min, __float_as_uint(r.n2));
        } else {
            s_nrm[row] = 0.f;
        }
    }
    return bad;
}

// waves per SIMD the LMedS kernel is compiled for (second __launch_bounds__ argument).  Up to 2048 rows
// (8 per thread) the 24 KB tile lets five workgroups share a CU; 4096 / 8192 rows (16 / 32 per thread, 48 /
// 96 KB of tile, 64 / 128 residual registers per lane) run at two / one -- slower per row, but a frame of a
// dense tracker is accepted instead of refused.
#ifndef RSSYNC_K2_WAVES4   // (waves per SIMD the 1024-row instantiation is compiled for: A/B of profiles/r5_k2_waves4_ab.txt)
#define RSSYNC_K2_WAVES4 6
#endif
#ifndef RSSYNC_K2_WAVES16   // (workgroups per CU the planner may aim at for 16 rows per thread: 3 = with a small window in dynamic LDS, lmeds_kernel<16, ., 1>; 2: never)
#define RSSYNC_K2_WAVES16 3
#endif
// (round 6, the SUB-SHAPES: any number of rows per thread from 3 to 15 -- 9 to 15 in the eight-wave shape -- for selections whose
// largest frame of a class needs no more; a thread adds its rows in order and rows beyond the frame add exact zeros, so the shape
// changes speed, never a bit: rssync_kernels.hip, lmeds_shape)
#ifndef RSSYNC_SUBSHAPE_TUNE   // (0: the sub-shapes of 5 / 6 rows per thread compiled for five waves per SIMD like 8, 9 / 10 for three like 16: the A/B of profiles/r6_k2_subshape_waves_ab.txt)
#define RSSYNC_SUBSHAPE_TUNE 1
#endif
__host__ __device__ constexpr int lmeds_waves(int rpt) {
#if RSSYNC_SUBSHAPE_TUNE
    // 1280 / 1536-row tiles: 79-80 VGPRs and <= 25.4 KB let SIX workgroups share a CU (-7 % per launch); 2304 / 2560-row tiles: 127-128
    // VGPRs without a spill and <= 37.7 KB let FOUR (-17 %)
    if (rpt == 5 || rpt == 6) return 6;
    if (rpt == 9 || rpt == 10) return 4;
#endif
    return rpt <= 4 ? RSSYNC_K2_WAVES4 : (rpt <= 8 ? 5 : (rpt <= 16 ? RSSYNC_K2_WAVES16 : 1));
}
__host__ __device__ constexpr int loss_waves(int rpt, bool grad) { return (grad || rpt >= 8) ? 3 : 4; }

constexpr int kMaxChunk = 32; // candidates per workgroup (rship: chunk <= kMaxChunk); 64 and 100 measured: no change
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

// residuals r = nP v of one hypothesis for the whole tile, in the wave's registers (core_private.cpp:48); |r| orders
// like the r^2 of :49-52.  Registers 4m..4m+3 <-> rows 4 (64 m + lane) .. +3.
template <int NR, int G0 = 0, int G1 = NR / 4> // groups [G0, G1) of four rows per lane
__device__ __forceinline__ void sweep_tile(const f4* p4x, const f4* p4y, const f4* p4z, int lane, f3 hv, uint32_t (&r2)[NR]) {
#pragma unroll
    for (int m = G0; m < G1; ++m) {
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
}

// LAZY SELECTION (round 3).  The arg-min over hypotheses of (lower quartile, index) needs the quartiles only to
// COMPARE them.  Round 2 selected the exact quartile of every hypothesis that beat the current bound -- 3.9 of 20
// per candidate, ~6 counting passes and a min pass each: a fifth of the kernel's instructions
// (profiles/r3_k2_counters.json).  Now a hypothesis that beats the bound narrows its bracket only until it holds at
// most kLazyElems elements and ends below the bound, publishes the bracket's upper end as the new (exclusive)
// bound -- still an upper bound of the best quartile, so nothing better is ever turned away -- and is recorded as a
// CONTENDER {h, lo, hi, counts}.  At the end of the candidate, contenders whose lo is not below the smallest hi
// have lost for certain; if one contender is left it has won without its quartile ever being known exactly;
// if several are left (their brackets overlap) each is swept again and its bracket closed exactly from where it
// stopped, and the packed (quartile, index) minimum decides as before.  The winner is the exact arg-min with the
// reference's first-wins tie rule either way (tests: identical best_h and costs against the exact selection,
// RSSYNC_K2_EXACT_SELECT=1, on every (frame, candidate) of full-size sweeps).
constexpr uint32_t kLazyElems = 16; // round 3 A/B (profiles/r3_k2_ab2.txt), ms per launch: 1 (= closed at once) 45.7, 4: 41.2, 16: 40.2, 48: 41.1
constexpr int kContCap = 24; // contender records per candidate; beyond that a hypothesis closes its bracket at once

// WIN = knots of the LDS spline window: kWinMax, a compile-time part of the workgroup's LDS, or 0 = p.win_cap knots in
// DYNAMIC LDS (gyro rates above ~1.7 kHz: a frame spans more than kWinMax knots; the host sizes the window for the
// problem and shortens the candidate chunk so that frame + chunk fit -- fewer workgroups per CU, three more address
// additions per coefficient fetch, but still the interior path).  Round 2 measured a 28-knot window with a
// 24-entry direction buffer: 27,088 B of LDS and 80 VGPRs, SIX workgroups per CU instead of five (the occupancy
// API confirmed it) -- and the same launch time within 0.1 %, while four workgroups per CU had been 12 % slower
// than five: beyond five waves per SIMD the kernel no longer gains from more resident waves.
// LAZY = false is round 2's exact selection of every quartile that beats the bound: instantiated only in the
// test-variants build (-DRSSYNC_TEST_VARIANTS=1, tests/test_gpu_lazy_select.py), which demands identical winners and costs.
// WIN = 1 (16 rows per thread only): the dynamic window again, compiled for THREE workgroups per CU instead of two -- for the
// launches whose window is small enough that the kernel's LDS lets a third one in (rssync_kernels.hip: plan_lmeds_window;
// 168 VGPRs, 44 of the 213 the kernel wants spilled to scratch, and still 25 % faster: profiles/r5_k2_class3_ab.txt).  WIN = 0
// stays compiled for two: a large window (high gyro rates) leaves no room for a third workgroup and the spills would only cost.
// R64 = the fp64-rows form ("fp64 rows" above; MODE 0, WIN 0): the same grid, but a workgroup leaves at once unless the
// fp32 launch has flagged candidates of its (frame, chunk), and evaluates only those -- stage A from the fp64 streams.
// Not a hot kernel: compiled without an occupancy target.
// BLOCK = threads of the workgroup: 256 (four waves) everywhere but the WIDE shape of round 6 -- frames of 6145 .. 8192 tracks
// (4097 .. 8192 at first; tiles of up to 6144 rows leave a CU TWO four-wave workgroups and stay four waves: RPT up to 24)
// as RPT = 16, BLOCK = 512: eight waves, two per SIMD.  The tile of such a frame (96 KB) allows one workgroup per CU either
// way; as four waves of 32 rows per thread (rounds 3-5: 480 VGPRs) that was ONE wave per SIMD, which issues an instruction
// every ~5 cycles instead of every ~2.3 (DESIGN.md section 3's table): 0.35 of the benchmark class's rate per ray.
template <int RPT, int MODE, int WIN, bool LAZY = true, bool R64 = false, int BLOCK = kBlock> // MODE 0: PreSync cost per candidate; 1: GuessMotion's hypothesis search (Sync start)
// (MODE 1, GuessMotion's search -- one candidate per workgroup, 0.5 % of a bench step -- holds the fp64 form of the rows as a
// branch: compiled for four waves per SIMD up to 2048 rows, so that the branch does not spill; measured no slower)
__global__ __launch_bounds__(BLOCK, BLOCK == 512 ? 2 : (R64 ? 1 : (RPT >= 16 ? (WIN == 1 ? 3 : 2) : (MODE == 1 && RPT <= 8 ? 4 : lmeds_waves(RPT))))) void lmeds_kernel(LmedsParams p) {
    static_assert(!R64 || (MODE == 0 && WIN == 0 && LAZY), "the fp64-rows form exists for the PreSync sweep only");
    static_assert(BLOCK == 256 || (BLOCK == 512 && RPT >= 9 && RPT <= 16), "workgroup shapes: four waves, or eight for tiles of 4608 .. 8192 rows");
    static_assert(BLOCK == 512 || RPT <= 24, "four waves: up to 24 rows per thread");
    constexpr int NWAVE = BLOCK / 64;
    constexpr int CAPW = WIN == 1 ? 0 : WIN; // the window's compile-time capacity (0 = dynamic)
    constexpr int kHyp = kHypBatch;
    constexpr int ROWS = BLOCK * RPT;
    constexpr int NR = ROWS / 64; // residual registers per lane: a wave spans the whole tile
    __shared__ __attribute__((aligned(16))) float s_n[3][ROWS];
    f4* s_win;
    if constexpr (CAPW != 0) {
        __shared__ f4 s_win_static[4 * CAPW];
        s_win = s_win_static;
    } else {
        extern __shared__ f4 s_win_dynamic[];
        s_win = s_win_dynamic;
    }
    __shared__ f4 s_hyp[kHyp];
    __shared__ double s_red[2][NWAVE];
    // best (quantile, hypothesis) so far, packed (bits << 32 | h): a 64-bit min is exactly
    // "smaller quantile wins, ties go to the earlier hypothesis" (core_private.cpp:53 strict <)
    __shared__ unsigned long long s_key;
    __shared__ uint32_t s_next; // hypothesis queue of the current batch
    // lazy selection: the contenders of the current candidate, and the exact (quartile, index) keys of those that
    // had to be closed
    __shared__ uint32_t s_ch[LAZY ? kContCap : 1], s_clo[LAZY ? kContCap : 1], s_chi[LAZY ? kContCap : 1], s_cclo[LAZY ? kContCap : 1],
        s_cchi[LAZY ? kContCap : 1];
    __shared__ uint32_t s_ncont;
    __shared__ unsigned long long s_exact;
    __shared__ uint32_t s_min2[NWAVE]; // per wave: the smallest |P|^2 (bit pattern) of the candidate's rows (hypothesis(): smin2)
    __shared__ float s_nrm[(R64 || MODE == 1) ? ROWS : 1]; // the fp64 form's norms (rounded once): R64, and MODE 1 when it takes that form in place
    __shared__ uint32_t s_near1;            // MODE 1: the frame is near-static (decided by wave 0, read by all)
    const int tid = threadIdx.x, lane = tid & 63;
#if RSSYNC_K2_TIMING && RSSYNC_K2_COUNTERS
    long long k2_bar = 0, k2_by[4] = {0, 0, 0, 0};
    const long long k2_t0 = clock64();
#endif
    // blocks b and b+8 share an XCD (round-robin dispatch): keep the chunks of one
    // frame on one XCD so its rays are fetched into one L2 only
    const uint32_t per = 8u * p.n_chunks;
    const uint32_t grp = blockIdx.x / per, within = blockIdx.x % per;
    const uint32_t entry = grp * 8u + (within & 7u);
    const uint32_t chunk = within >> 3;
    if (entry >= p.n_slots) return;
    const uint32_t sf = p.slots ? p.slots[entry] : entry;
    const uint32_t fi = p.sel[sf];
    const FrameRec fr = p.frames[fi];
    const uint32_t N = fr.n;
    const uint32_t kq = N / 4; // core_private.cpp:52
    const uint32_t g = p.grp ? p.grp[sf] : 0u; // window this slot belongs to (batched Sync)
    const Tile tile{s_n[0], s_n[1], s_n[2]};

    // rays are re-read per candidate: the chunks of a frame share an XCD, so after the
    // first touch they come from that XCD's L2 (keeping them in registers costs 64 VGPRs)
    const RayRsrc rays = make_ray_rsrc(p.rays_a + fr.off, p.rays_b + fr.off, N);

    const uint32_t c0 = chunk * p.chunk;
    const uint32_t c1 = (c0 + p.chunk < p.n_cand) ? c0 + p.chunk : p.n_cand;
    if (c0 >= c1) return;
    // R64: the candidates of this chunk the fp32 launch flagged (bit i <-> candidate c0 + i); none: nothing to do
    uint32_t redo = 0;
    if constexpr (R64) {
        const uint32_t* mw = p.redo_mask + (size_t)sf * p.mask_words;
        const uint32_t w0 = c0 >> 5, w1 = (c1 - 1u) >> 5;
        const unsigned long long both = (unsigned long long)mw[w0] | (w1 != w0 ? (unsigned long long)mw[w1] << 32 : 0ull);
        redo = (uint32_t)(both >> (c0 & 31u));
        if (c1 - c0 < 32u) redo &= (1u << (c1 - c0)) - 1u;
        redo = uniform_u32(redo);
        if (!redo) return;
    }

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
    sp.cap = (int)p.win_cap;
    sp.whole_pair = p.win_whole_pair != 0;
    if constexpr (!R64) { // (the fp64 rows read the fp64 table from L2: no window)
        int kd_lo = p.kd[c0 * p.n_grp + g], kd_hi = kd_lo;
        for (uint32_t c = c0 + 1; c < c1; ++c) {
            int v = p.kd[c * p.n_grp + g];
            kd_lo = v < kd_lo ? v : kd_lo;
            kd_hi = v > kd_hi ? v : kd_hi;
        }
        stage_window_ends<CAPW>(sp, s_win, frame_knots(fr, fr.base_knot + (int)floorf(fr.tmin) + kd_lo,
                                                       fr.base_knot + (int)floorf(fr.tmax) + kd_hi + 1, kd_lo, kd_hi), BLOCK);
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j) { // rows beyond N: NaN once, never rewritten
        const uint32_t row = j * BLOCK + tid;
        if (row >= N) s_n[0][row] = s_n[1][row] = s_n[2][row] = __uint_as_float(0x7fc00000u);
    }
    __syncthreads();
    if constexpr (R64) { // every thread has read the chunk's bits: they are served, clear them (other chunks share the words)
        if (tid == 0) {
            const uint32_t w0 = c0 >> 5, w1 = (c1 - 1u) >> 5;
            uint32_t* mw = p.redo_mask + (size_t)sf * p.mask_words;
            atomicAnd(&mw[w0], ~(redo << (c0 & 31u)));
            if (w1 != w0) atomicAnd(&mw[w1], ~(uint32_t)((unsigned long long)redo >> (32u - (c0 & 31u))));
            atomicAdd(p.redo_count, (unsigned long long)__builtin_popcount(redo));
        }
    }

    const f4* p4x = reinterpret_cast<const f4*>(tile.nx);
    const f4* p4y = reinterpret_cast<const f4*>(tile.ny);
    const f4* p4z = reinterpret_cast<const f4*>(tile.nz);
    uint32_t prev_best = kInfBits; // winning quantile of the previous candidate of this chunk

    for (uint32_t c = c0; c < c1; ++c) {
        if constexpr (R64) {
            if (!((redo >> (c - c0)) & 1u)) continue;
        }
        const int base = fr.base_knot + s_kd[c - c0];
        const float fd = s_fd[c - c0];
        const uint32_t stream = p.stream_base + c + g * p.stream_stride; // g != 0 only for batched GuessMotion
        uint32_t bad = 0;
        // ---- stage A: rows of P -> LDS tile as unit rows; norms stay in registers ----
        float nrm[RPT];
        uint32_t n2min;
        bool use64 = false;   // MODE 1: this frame's rows were taken from the fp64 streams (near-static)
        int base64 = 0;
        double fd64 = 0.0;
        if constexpr (R64) {
            bad |= lmeds_rows64<RPT, BLOCK>(p.src64, fr.off, N, fr.base_knot + p.kd64[c], p.fd64[c], tile, s_nrm, n2min);
#pragma unroll
            for (int j = 0; j < RPT; ++j) nrm[j] = s_nrm[j * BLOCK + tid]; // (this thread's own stores)
        } else {
            uint32_t near;
            bad |= lmeds_rows<RPT, MODE == 0, CAPW, BLOCK>(sp, rays, N, base, fd, tile, nrm, n2min, near);
            // the near-static watch ("fp64 rows" above): thread 0's wave has counted the frame's first 64 rows
            if (RSSYNC_NEAR_WATCH && MODE == 0 && p.redo_mask && tid == 0 && near_static_fires(near, N)) {
                atomicOr(&p.redo_mask[(size_t)sf * p.mask_words + (c >> 5)], 1u << (c & 31u));
                bad |= RSHIP_NEAR_STATIC;
            }
            if constexpr (MODE == 1 && RSSYNC_NEAR_WATCH) {
                // GuessMotion's search: one candidate per workgroup, so the fp64 form of the rows is taken right here (the
                // same decision, the same rows as the sweep's second launch; the window executor's search task does the same,
                // kernels/lmeds_small.hpp / exec_big.hpp, so that executor and launch chain pick the same winner bit for bit)
                if (tid == 0) s_near1 = (p.src64.coef && near_static_fires(near, N)) ? 1u : 0u;
                __syncthreads();
                use64 = s_near1 != 0u;
                if (use64) {
                    base64 = fr.base_knot + p.kd64[c * p.n_grp + g];
                    fd64 = p.fd64[c * p.n_grp + g];
                    bad = lmeds_rows64<RPT, BLOCK>(p.src64, fr.off, N, base64, fd64, tile, s_nrm, n2min);
#pragma unroll
                    for (int j = 0; j < RPT; ++j) nrm[j] = s_nrm[j * BLOCK + tid]; // (this thread's own stores)
                    if (tid == 0) atomicAdd(p.redo_count + 1, 1ull);
                }
            }
        }
        {   // (written before the "tile written" barrier below, read by the hypotheses' lanes after it; the next
            // candidate's write comes after this candidate's last barrier)
            const uint32_t wmin = wave_min_u32(n2min);
            if (lane == 0) s_min2[tid >> 6] = wmin;
        }
        // |P_row| for hypothesis(), from the rays (R64, and MODE 1 in its fp64 form: the norms of stage A, in LDS -- read after the
        // "tile written" barrier)
        auto row_scale = [&](uint32_t row) -> float {
            if constexpr (R64) return s_nrm[row];
            else {
                if (MODE == 1 && use64) return s_nrm[row];
                return row_scale_general(p.coef, p.n_knots, load_ray(rays.a, row * 16u, 0u), load_ray(rays.b, row * 16u, 0u), base, fd);
            }
        };
        auto frame_smin2 = [&]() -> float {
            uint32_t m = s_min2[0];
#pragma unroll
            for (int w = 1; w < NWAVE; ++w) m = s_min2[w] < m ? s_min2[w] : m;
            return smin2_of(m);
        };

        // ---- stage C: the hypotheses.  The best quantile of the previous candidate (x1.25: between
        // neighbouring candidates it moves by -20..+26 %, 1st..99th percentile) serves as a
        // provisional bound: a hypothesis that has <= kq residuals below it is dropped after one
        // counting pass.  If nothing beats the bound (~2 % of candidates) the candidate is redone
        // without it, so the result is the exact arg-min either way.
        uint32_t guess = kInfBits;
        if (prev_best < 0x7e000000u && prev_best > 0x00800000u)
            guess = uniform_u32(__float_as_uint(__uint_as_float(prev_best) * 1.25f));
        uint32_t bT; // the winner's quartile (exact selection) or an upper bound of it (lazy): the next candidate's bound
        int bH;
        if (!LAZY) {
        unsigned long long best;
        for (;;) {
            if (tid == 0) s_key = ((unsigned long long)guess << 32);
            for (uint32_t batch = 0; batch < p.n_hyp; batch += kHyp) {
                const uint32_t nb = (p.n_hyp - batch < (uint32_t)kHyp) ? p.n_hyp - batch : (uint32_t)kHyp;
                __syncthreads(); // tile written / previous batch consumed
                if ((uint32_t)tid < nb) {
                    const f3 v = hypothesis(tile, p.seed, fr.id, stream, batch + tid, N, frame_smin2(), row_scale);
                    s_hyp[tid] = f4{v.x, v.y, v.z, 0.f};
                }
                if (tid == 0) s_next = 0;
                __syncthreads();
                for (;;) { // waves pull hypotheses from the queue: no wave idles at the barrier
                    const uint32_t j = wave_pop(&s_next);
                    K2_COUNT(1);
                    if (j >= nb) break;
                    K2_COUNT(2);
                    const uint32_t h = batch + j;
                    const f4 hv = s_hyp[j];
                    uint32_t r2[NR];
                    sweep_tile(p4x, p4y, p4z, lane, f3{hv.x, hv.y, hv.z}, r2);
                    // (quantile_h, h) < (T, g)  <=>  more than kq |residuals| lie below T (+1 ulp if g > h):
                    // med < least_med of core_private.cpp:51-53 with the reference's first-wins tie rule
                    const unsigned long long key = __hip_atomic_load(&s_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t T = (uint32_t)(key >> 32), g = (uint32_t)key;
                    uint32_t hi2 = T + ((T != kInfBits && g > h) ? 1u : 0u);
                    const uint32_t tot = wave_count_lt(r2, hi2);
                    if (tot > kq) {
                        K2_COUNT(3);
                        if (hi2 == kInfBits) { // no bound yet: start the bracket at the largest residual
                            K2_COUNT(7);
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
            if (tid == 0) K2_COUNT(6);
            __syncthreads();  // everyone has read s_key before it is reset
        }
        bT = (uint32_t)(best >> 32);
        bH = (bT == kInfBits) ? -1 : (int)(uint32_t)best;
        } else {
        // ---- lazy selection: s_key's high word is an EXCLUSIVE upper bound of the best quartile found so far ----
        uint32_t n_cont;
        for (;;) {
            if (tid == 0) { s_key = ((unsigned long long)guess << 32); s_ncont = 0; s_exact = ~0ull; }
            for (uint32_t batch = 0; batch < p.n_hyp; batch += kHyp) {
                const uint32_t nb = (p.n_hyp - batch < (uint32_t)kHyp) ? p.n_hyp - batch : (uint32_t)kHyp;
                K2_SYNC(0); // tile written / previous batch consumed
                if ((uint32_t)tid < nb) {
                    const f3 v = hypothesis(tile, p.seed, fr.id, stream, batch + tid, N, frame_smin2(), row_scale);
                    s_hyp[tid] = f4{v.x, v.y, v.z, 0.f};
                }
                if (tid == 0) s_next = 0;
                K2_SYNC(1);
                for (;;) {
                    const uint32_t j = wave_pop(&s_next);
                    K2_COUNT(1);
                    if (j >= nb) break;
                    K2_COUNT(2);
                    const uint32_t h = batch + j;
                    const f4 hv = s_hyp[j];
                    uint32_t r2[NR];
                    // quartile_h < T  <=>  more than kq |residuals| lie below T.  A hypothesis whose quartile EQUALS the
                    // best one passes as well (T is exclusive and above it): ties are settled among the contenders.
                    // (T is read before the sweep: a bound that has tightened meanwhile only makes this test milder.)
                    const uint32_t T = (uint32_t)(__hip_atomic_load(&s_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32);
                    uint32_t tot;
                    if (NR >= 16) {
                        // Four of five hypotheses only have to be turned away, and most of those have a few per cent of
                        // their residuals below T: once all rows but the last 256 are counted and even 256 more could
                        // not lift the count above kq, the last group of rows is neither read nor multiplied.
                        constexpr int GL = NR / 4 - 1;
                        sweep_tile<NR, 0, GL>(p4x, p4y, p4z, lane, f3{hv.x, hv.y, hv.z}, r2);
                        tot = wave_count_lt<NR, 0, 4 * GL>(r2, T);
                        if (tot + 256u <= kq) continue;
                        K2_COUNT(9);
                        sweep_tile<NR, GL, GL + 1>(p4x, p4y, p4z, lane, f3{hv.x, hv.y, hv.z}, r2);
                        tot += wave_count_lt<NR, 4 * GL, NR>(r2, T);
                    } else {
                        sweep_tile(p4x, p4y, p4z, lane, f3{hv.x, hv.y, hv.z}, r2);
                        tot = wave_count_lt(r2, T);
                    }
                    if (tot > kq) {
                        K2_COUNT(3);
                        Bracket b{0u, 0u, T, tot};
                        if (T == kInfBits) { // no bound yet: start the bracket at the largest residual
                            K2_COUNT(7);
                            float mx = 0.f;
#pragma unroll
                            for (int m = 0; m < NR; ++m) mx = fmaxf(mx, fabsf(__uint_as_float(r2[m])));
                            mx = wave_max_f32(mx);
                            if (finite_f(mx)) b.hi = __float_as_uint(mx) + 1u; // count(|r| < hi) is still tot
                        }
                        narrow_kth(r2, kq, b, kLazyElems, T); // until it ends below T and holds <= kLazyElems elements (or is closed)
                        uint32_t slot = 0;
                        if (lane == 0) {
                            atomicMin(&s_key, (unsigned long long)b.hi << 32);
                            slot = atomicAdd(&s_ncont, 1u);
                        }
                        slot = uniform_u32(slot);
                        if (slot < (uint32_t)kContCap) {
                            if (lane == 0) { s_ch[slot] = h; s_clo[slot] = b.lo; s_chi[slot] = b.hi; s_cclo[slot] = b.c_lo; s_cchi[slot] = b.c_hi; }
                        } else { // no room for a record: close the bracket now, the exact key takes part in the decision
                            narrow_kth(r2, kq, b, 0u, 0u);
                            if (lane == 0) atomicMin(&s_exact, ((unsigned long long)b.lo << 32) | h);
                        }
                    }
                }
            }
            K2_SYNC(2);
            n_cont = s_ncont;
            if (guess == kInfBits || n_cont != 0u) break;
            guess = kInfBits; // nothing beat the provisional bound: redo this candidate without it
            if (tid == 0) K2_COUNT(6);
            __syncthreads();  // everyone has read the counters before they are reset
        }
        // ---- the decision among the contenders (every thread computes the same) ----
        const uint32_t n_rec = n_cont < (uint32_t)kContCap ? n_cont : (uint32_t)kContCap;
        const unsigned long long ovf = s_exact; // only set when records overflowed
        uint32_t m_hi = kInfBits;
        for (uint32_t i = 0; i < n_rec; ++i) m_hi = s_chi[i] < m_hi ? s_chi[i] : m_hi;
        if (ovf != ~0ull && (uint32_t)(ovf >> 32) + 1u < m_hi) m_hi = (uint32_t)(ovf >> 32) + 1u;
        uint32_t n_alive = (ovf != ~0ull && (uint32_t)(ovf >> 32) < m_hi) ? 1u : 0u;
        uint32_t w_slot = 0;
        for (uint32_t i = 0; i < n_rec; ++i)
            if (s_clo[i] < m_hi) { ++n_alive; w_slot = i; } // (lo >= the smallest hi: its quartile is above another's)
        if (n_cont == 0u) {
            bH = -1;
            bT = kInfBits;
        } else if (n_alive == 1u && ovf == ~0ull) {
            bH = (int)s_ch[w_slot]; // the only one left: the arg-min, its quartile known to lie in [lo, hi)
            bT = s_chi[w_slot];
        } else {
            // overlapping brackets: close them exactly, from where they stopped
            __syncthreads(); // (every thread has read the records' decision inputs)
            if (tid == 0) s_next = 0;
            __syncthreads();
            for (;;) {
                const uint32_t j = wave_pop(&s_next);
                if (j >= n_rec) break;
                if (!(s_clo[j] < m_hi)) continue;
                K2_COUNT(8);
                const uint32_t h = s_ch[j];
                const f3 v = hypothesis(tile, p.seed, fr.id, stream, h, N, frame_smin2(), row_scale);
                uint32_t r2[NR];
                sweep_tile(p4x, p4y, p4z, lane, v, r2);
                Bracket b{s_clo[j], s_cclo[j], s_chi[j], s_cchi[j]};
                narrow_kth(r2, kq, b, 0u, 0u);
                // (the index is read again rather than kept across the sweep: the kernel has no register to spare here)
                if (lane == 0) atomicMin(&s_exact, ((unsigned long long)b.lo << 32) | s_ch[j]);
            }
            __syncthreads();
            const unsigned long long best = s_exact; // smaller quartile wins, ties go to the earlier hypothesis (core_private.cpp:53)
            bH = (int)(uint32_t)best;
            bT = (uint32_t)(best >> 32) + 1u;
        }
        }
        prev_best = bT;
        if (tid == 0) K2_COUNT(0);
        f3 Mv = f3{0, 0, 0};
        if (bH >= 0) {
            if (p.n_hyp <= (uint32_t)kHyp) { // the winner's direction is still in the batch buffer
                const f4 hv = s_hyp[bH];
                Mv = f3{hv.x, hv.y, hv.z};
            } else {
                Mv = hypothesis(tile, p.seed, fr.id, stream, (uint32_t)bH, N, frame_smin2(), row_scale);
            }
        }
        if (!(finite_f(Mv.x) && finite_f(Mv.y) && finite_f(Mv.z))) bad |= RSHIP_BAD_M;
        if (MODE == 1) { // GuessMotion: only the winner's index leaves this kernel (one candidate per workgroup)
            if (tid == 0) p.best_h[sf] = bH;
            continue;
        }

        // ---- stage D: k = clamp(100 / |P M|), cost = sqrt(sum sqrt(log1p(r^2))) ----
        // Branch-free over the rows: a row beyond N has nrm = 0 but a NaN tile entry: the product in which a zero factor
        // wins (mul_zero_wins) makes it 0, and zeros contribute nothing below.
        float pm[RPT];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const uint32_t row = j * BLOCK + tid;
            pm[j] = mul_zero_wins(nrm[j], rs::dot(f3{tile.nx[row], tile.ny[row], tile.nz[row]}, Mv));
            ss = fmaf(pm[j], pm[j], ss);
        }
        double ss_tot;
        K2_TIMED(3, ss_tot = block_sum_n<NWAVE>(ss, s_red[0]));
        // core_private.cpp:79, 100 / ||P M|| as 100 * rsq (v_rsq_f32, 1 ulp); ss = 0 gives +inf -> clamp
        float kf = 100.0f * rs::rsqrt_fast((float)ss_tot);
        kf = (kf < 10.f) ? 10.f : ((1000.f < kf) ? 1000.f : kf);
        {
            float sc = kf * rs::rsqrt_fast(rs::dot(Mv, Mv)); // core_private.cpp:80
            // a non-finite r or rho (core_private.cpp:81,83) makes the sum non-finite (NaN and inf propagate through r^2,
            // log1p and sqrt, and all terms are >= 0): the check is made once on the sum, and only if it fires does the
            // thread look which of the reference's two checks would have been first (round 4: the per-row |r| sum that
            // answered this beforehand cost 0.15 ms per launch, profiles/r4_k2_norsum_ab.txt)
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const float r = pm[j] * sc;
                const float rho = rs::log1p_pos_fast(r * r); // core_private.cpp:82
                // v_sqrt_f32 directly (1 ulp): libm's sqrtf adds range scaling for denormal inputs,
                // whose square roots (< 1e-19) cannot change a sum of O(1) terms in fp32
                acc += __builtin_amdgcn_sqrtf(rho);
            }
            if (!finite_f(acc)) {
                float rsum = 0.f;
#pragma unroll
                for (int j = 0; j < RPT; ++j) rsum += fabsf(pm[j] * sc);
                bad |= finite_f(rsum) ? RSHIP_BAD_RHO : RSHIP_BAD_R;
            }
            double acc_tot;
            K2_TIMED(3, acc_tot = block_sum_n<NWAVE>(acc, s_red[1]));
            if (tid == 0) {
                p.frame_cost[(size_t)c * p.n_sel + sf] = sqrt(acc_tot); // core_private.cpp:85
                if (p.best_h) p.best_h[(size_t)c * p.n_sel + sf] = bH;
            }
        }
#if RSSYNC_TEST_VARIANTS
        if (MODE == 0 && p.dump) {
            // every hypothesis against this thread's own rows of the tile, in sweep_tile's very expression (the packed
            // form, so that the compiler contracts it into the same mul + fma + fma); s_hyp still holds the candidate's
            // directions (n_hyp <= kHyp), and both are only rewritten after the next candidate's barriers
            const uint32_t nh = p.n_hyp < (uint32_t)kHyp ? p.n_hyp : (uint32_t)kHyp;
            for (uint32_t h = 0; h < nh; ++h) {
                const f4 hv = s_hyp[h];
                uint32_t* out = p.dump + (((size_t)c * p.n_sel + sf) * p.n_hyp + h) * p.dump_rows;
#pragma unroll 1
                for (int j = 0; j < RPT; ++j) {
                    const uint32_t row = j * BLOCK + tid;
                    if (row < N && row < p.dump_rows) {
                        const v2f r01 = v2f{tile.nx[row], 0.f} * hv.x + v2f{tile.ny[row], 0.f} * hv.y + v2f{tile.nz[row], 0.f} * hv.z;
                        out[row] = __float_as_uint(r01.x) & 0x7fffffffu;
                    }
                }
            }
        }
#endif
        if (bad) atomicOr(p.flags, bad);
        // No barrier here.  What the next candidate overwrites before its first barrier is (a) this
        // thread's own tile rows and (b) s_key, by thread 0: every reader of s_key reads it before
        // the workgroup sum barrier of stage D, which thread 0 has passed by then.  s_hyp, s_next
        // and the sum slots are rewritten only after further barriers of the next candidate.
    }
#if RSSYNC_K2_TIMING && RSSYNC_K2_COUNTERS
    if (lane == 0 && MODE == 0) {
        atomicAdd(&g_k2_counters[10], (unsigned long long)k2_bar);
        atomicAdd(&g_k2_counters[11], (unsigned long long)(clock64() - k2_t0));
        for (int q = 0; q < 4; ++q) atomicAdd(&g_k2_counters[12 + q], (unsigned long long)k2_by[q]);
    }
#endif
}

} // namespace
