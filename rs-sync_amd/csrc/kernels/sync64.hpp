// sync64.hpp -- the Sync kernels in fp64: K1 (residual + robust loss + analytic d/d-delay, also the
// no-translation variant) and K3 (per-frame motion L-BFGS, with GuessMotion/GuessK finished in fp64).
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
//
// Why fp64: the reference's arithmetic is IEEE double throughout (core_private.cpp), and Sync is a
// chaotic iteration on data with outliers -- the per-frame L-BFGS takes unit-length first steps on a
// non-convex loss, so a 1e-7 perturbation of one residual row (fp32 rounding of a ray, of a spline
// coefficient, or of one product) sends some frames into another basin and moves the returned delay
// by tenths of a millisecond.  With fp64 inputs (the raw records are packed into fp64 streams, the
// spline table stays fp64) and fp64 arithmetic the device follows the CPU solver to ~1e-12 per
// evaluation.  gfx950 issues fp64 FMA at half the fp32 rate; Sync is ~10 % of a PreSync + Sync step.
#pragma once

// The fp64 kernels are compiled with contraction off (restored at the end of this header): what is fused is
// written as fma(), so that tests/cpu_device/rship_cpu.cpp -- the same expressions under g++
// -ffp-contract=off, summed in the kernels' association -- reproduces them bit for bit.
#pragma clang fp contract(off)

namespace {

// (Spline64, the fp64 window, Rays64 and residual_row64: kernels/rows64.hpp, included before the LMedS kernels since round 6)

// ---------------------------------------------------------------------------
// K1: per slot, FrameState::Loss at a batch of delays (core_private.cpp:117-123):
//   r = (P M) k / |M|,  loss = sum log1p(r^2)
// and with GRAD the analytic d/d-delay that replaces the central difference of :96-97,112
//   dL/dd = sum 1/(1+u) (2 pm / s) (dP/dd . M),  u = pm^2 / s,  s = |M|^2 / k^2.
// SIMPLE = the thesis' no-translation variant (section 2.11 eq. (12)): u = k^2 |P|^2, no M.
// One workgroup per slot; rows are re-read per delay (the frame's 128 KB stay in L2).

struct Loss64Params {
    Rays64 rays;
    const FrameRec* frames;
    const uint32_t* sel;
    uint32_t n_sel;
    const d4* coef;
    int n_knots;
    double fs;
    const int32_t* kd; // [n_delays][n_grp]
    const double* fd;  // NaN = this group is skipped (its partial sums are written as 0)
    uint32_t n_delays;
    const uint32_t* grp;
    uint32_t n_grp;
    const double* M; // per selection slot
    const double* k;
    double* part_loss; // [n_delays][n_sel]
    double* part_grad; // [n_delays][n_sel] (GRAD)
    uint32_t slot0;    // the launch covers slots slot0 .. slot0 + gridDim.x (a group of windows on its own stream) ...
    const uint32_t* slots; // ... or, if not null, the slots slots[0 .. gridDim.x): the frames of one size class (rssync_kernels.hip)
    uint32_t win_cap;  // knots per spline window (dynamic LDS: nb_run x win_cap x 128 bytes; one window in the one-wave kernel)
    uint32_t win_compact; // 1: the window holds y and c only, 64 bytes per knot (Spline64::compact; the one-wave kernels at high gyro rates)
    uint32_t nb_run;   // delays evaluated per pass over the rows: kLossBatch while their windows fit the LDS, fewer for wide frames
};

// delays evaluated per pass over the rows: every ray pair is read once for kLossBatch delays (their spline
// windows sit side by side in LDS, 10 KB each).  The line search's trials are far apart in time (steps of
// 1e-3 .. 1e-12 times the gradient), so the windows cannot be shared; the rays can.  A pass per delay made
// this kernel HBM/L2-bound (64 B per ray pair per delay: 6.5 TB/s effective at 4096 x 2048).
constexpr int kLossBatch = 5; // (6, at two workgroups per CU: 5 % slower)
constexpr int kLossBatchWide = 3; // the same with the windows in dynamic LDS (run-time stride: 167 VGPRs at three; as many of them as 51 KB hold)

// RPT = rows per thread the launch covers (the largest frame's); 0 = as many as this frame needs (frames of more
// than 8192 tracks).  A thread adds its rows in order either way, so a frame's sums do not depend on RPT.
// CAP = knots per spline window: kWinMax compiled in (the five-window kernel of the line search's trials, whose
// registers leave no room for a run-time stride) or 0 = p.win_cap knots in dynamic LDS (the gradient kernel always:
// one window; the trials' kernel for frames wider than kWinMax knots, with p.nb_run <= kLossBatchWide windows)
template <int RPT, bool GRAD, bool SIMPLE, int CAP = 0>
__global__ __launch_bounds__(kBlock, 3) void loss64_kernel(Loss64Params p) {
    constexpr int NB = GRAD ? 1 : (CAP ? kLossBatch : (RPT == 0 ? 2 : kLossBatchWide)); // (RPT = 0, frames of more than 8192 tracks: three windows spill)
    d4* s_loss_win;
    uint32_t win_cap;
    if constexpr (CAP != 0) {
        __shared__ d4 s_loss_win_static[NB * 4 * CAP];
        s_loss_win = s_loss_win_static;
        win_cap = CAP;
    } else {
        extern __shared__ d4 s_loss_win_dynamic[]; // [nb_run][4 * win_cap]
        s_loss_win = s_loss_win_dynamic;
        win_cap = p.win_cap;
    }
    __shared__ double s_red[NB][2][4];
    const uint32_t nb_run = (GRAD || CAP) ? (uint32_t)NB : (p.nb_run < (uint32_t)NB ? (p.nb_run ? p.nb_run : 1u) : (uint32_t)NB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t sf = p.slots ? p.slots[blockIdx.x] : blockIdx.x + p.slot0;
    const uint32_t g = p.grp ? p.grp[sf] : 0u;
    {
        // nothing to evaluate for this slot's window (a finished window, a line search that has already
        // succeeded): zeros and out, before anything is fetched
        bool any = false;
        for (uint32_t b = 0; b < p.n_delays; ++b) {
            const double f = p.fd[b * p.n_grp + g];
            any = any || f == f;
        }
        if (!any) {
            for (uint32_t b = tid; b < p.n_delays; b += kBlock) {
                p.part_loss[(size_t)b * p.n_sel + sf] = 0.0;
                if (GRAD) p.part_grad[(size_t)b * p.n_sel + sf] = 0.0;
            }
            return;
        }
    }
    const uint32_t fi = p.sel[sf];
    const FrameRec fr = p.frames[fi];
    const uint32_t N = fr.n;
    const double Mx = SIMPLE ? 0.0 : p.M[3 * sf], My = SIMPLE ? 0.0 : p.M[3 * sf + 1], Mz = SIMPLE ? 0.0 : p.M[3 * sf + 2];
    const double kk = p.k[sf];
    const d3 Mv = d3{Mx, My, Mz};
    // r = (P.M) k / |M|  (core_private.cpp:120)  ->  u = (P.M)^2 * inv_s;  SIMPLE: u = |P|^2 k^2
    const double inv_s = rs::loss_inv_s(SIMPLE, kk, Mv);

    for (uint32_t b0 = 0; b0 < p.n_delays; b0 += nb_run) {
        Spline64 sp[NB];
        int base[NB];
        double fdv[NB];
        bool on[NB], any_on = false;
        int kdv[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const uint32_t b = b0 + q;
            on[q] = (uint32_t)q < nb_run && b < p.n_delays;
            kdv[q] = on[q] ? p.kd[b * p.n_grp + g] : 0;
            fdv[q] = on[q] ? p.fd[b * p.n_grp + g] : 0.0;
            if (fdv[q] != fdv[q]) on[q] = false; // window switched off for this evaluation (workgroup-uniform)
            any_on = any_on || on[q];
        }
        if (!any_on) { // a whole batch of skipped delays
            if ((uint32_t)tid < nb_run && b0 + tid < p.n_delays) {
                p.part_loss[(size_t)(b0 + tid) * p.n_sel + sf] = 0.0;
                if (GRAD) p.part_grad[(size_t)(b0 + tid) * p.n_sel + sf] = 0.0;
            }
            continue;
        }
        __syncthreads(); // windows and s_red of the previous group are free
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            sp[q].g = p.coef;
            sp[q].n = p.n_knots;
            sp[q].cap = (int)win_cap;
            sp[q].compact = CAP ? 0 : (int)p.win_compact;
            base[q] = fr.base_knot + kdv[q];
            if (on[q]) frame_window64<CAP>(sp[q], s_loss_win + (size_t)q * 4 * win_cap, fr, kdv[q]);
        }
        __syncthreads();
        double L[NB], G[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) L[q] = G[q] = 0.0;
        const int rpt = RPT ? RPT : (int)((N + kBlock - 1) / kBlock);
#pragma unroll 1 // (unroll 2 measured in round 3: the gradient launch stays at 0.122 ms; round 4: the next row's 64 bytes requested
                 // before this row's arithmetic, 158 VGPRs: 1.03 against 1.00 ms per bench step, profiles/r4_k1_prefetch_ab.txt)
        for (int j = 0; j < rpt; ++j) {
            const uint32_t row = j * kBlock + tid;
            if (row < N) {
                const size_t idx = (size_t)fr.off + row;
                const double2 X = p.rays.q0[idx], Y = p.rays.q1[idx], Z = p.rays.q2[idx], T = p.rays.q3[idx];
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    if (!on[q]) continue;
                    d3 P, dP;
                    residual_row64_auto<GRAD, CAP>(sp[q], X, Y, Z, T, base[q], fdv[q], P, dP);
                    rs::loss_row<GRAD, SIMPLE>(P, dP, Mv, inv_s, L[q], G[q]); // core_private.cpp:121-122
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const double Lw = wave_sum_f64(L[q]);
            const double Gw = GRAD ? wave_sum_f64(G[q]) : 0.0;
            if (lane == 0) {
                s_red[q][0][wave] = Lw;
                s_red[q][1][wave] = Gw;
            }
        }
        __syncthreads();
        if ((uint32_t)tid < nb_run && b0 + tid < p.n_delays) {
            const uint32_t b = b0 + tid;
            const double fdb = p.fd[b * p.n_grp + g];
            const bool live = fdb == fdb;
            const double mine = live ? s_red[tid][0][0] + s_red[tid][0][1] + s_red[tid][0][2] + s_red[tid][0][3] : 0.0;
            p.part_loss[(size_t)b * p.n_sel + sf] = mine;
            if (GRAD)
                p.part_grad[(size_t)b * p.n_sel + sf] =
                    live ? (s_red[tid][1][0] + s_red[tid][1][1] + s_red[tid][1][2] + s_red[tid][1][3]) * p.fs : 0.0;
        }
    }
}

// FrameState::Loss (and its analytic d/d-delay) of one slot at one delay by ONE wave, in the association of
// loss64_kernel: thread t of that kernel's four waves holds rows t, t + 256, ... (one row up to 256 tracks, two up to
// 512) and adds them in that order, each wave is summed by wave_sum_f64, the four wave sums are added left to right.  Mv, kk: the slot's motion estimate.
// (A frame of 130 tracks leaves two of the four waves of loss64_kernel's workgroup idle, and the workgroup holds its
// registers all the same: three per CU.  One wave per slot puts four times as many slots on the chip; the window
// executor evaluates its loss tasks this way too.)
// The rays of a lane's FIRST row of each of the first NPRE "waves" (rows w * 64 + lane), requested in one go.  Round 5: the
// loop below fetched a row's 64 bytes where it needed them -- one round trip past L1 per wave of rows, one after the other,
// after the round trip in which the spline window is staged.  Requested before the window's loads they arrive in the same
// round trip (loads return in order).  Only the order of the loads changes: the same values enter the same arithmetic in the
// same order.  (Holding them across a line search's ten evaluations instead -- one request per TASK -- was tried: +64 live
// registers through the whole task took the executor from two waves per SIMD to one.)
template <int NPRE>
struct RowRays {
    double2 X[NPRE ? NPRE : 1], Y[NPRE ? NPRE : 1], Z[NPRE ? NPRE : 1], T[NPRE ? NPRE : 1];
};
template <int NPRE>
__device__ __forceinline__ void load_row_rays(const Rays64& r, const FrameRec& fr, RowRays<NPRE>& o) {
#pragma unroll
    for (int w = 0; w < NPRE; ++w) {
        const uint32_t row = (uint32_t)w * 64u + threadIdx.x;
        if (row < fr.n) {
            const size_t idx = (size_t)fr.off + row;
            o.X[w] = r.q0[idx]; o.Y[w] = r.q1[idx]; o.Z[w] = r.q2[idx]; o.T[w] = r.q3[idx];
        }
    }
}

template <bool GRAD, bool SIMPLE = false, int NPRE = 0>
__device__ __forceinline__ void loss64_wave(const Loss64Params& q, uint32_t sf, d3 Mv, double kk, int kd, double fd, d4* s_win,
                                            double& L_out, double& G_out, const RowRays<NPRE>* given = nullptr) {
    const int lane = threadIdx.x;
    const FrameRec fr = q.frames[q.sel[sf]];
    const uint32_t N = fr.n;
    const double inv_s = rs::loss_inv_s(SIMPLE, kk, Mv);
    Spline64 sp;
    sp.g = q.coef;
    sp.n = q.n_knots;
    sp.cap = (int)q.win_cap;
    sp.compact = (int)q.win_compact;
    __syncthreads(); // the window's previous users are done
    RowRays<NPRE> mine;
    if (NPRE && !given) load_row_rays<NPRE>(q.rays, fr, mine); // (before the window's loads: they return in order, one round trip for both)
    const RowRays<NPRE>& pre = given ? *given : mine;
    frame_window64(sp, s_win, fr, kd);
    __syncthreads();
    const int base = fr.base_knot + kd;
    double Lw[4], Gw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        Lw[w] = 0.0;
        Gw[w] = 0.0;
        if ((uint32_t)w * 64u < N) { // (a wave without rows sums zeros to zero)
            double L = 0.0, G = 0.0;
            uint32_t row = (uint32_t)w * 64u + lane; // (the four-wave kernel's rows j * 256 + tid of this thread)
            if (w < NPRE) {
                if (row < N) {
                    d3 P, dP;
                    residual_row64_auto<GRAD>(sp, pre.X[w], pre.Y[w], pre.Z[w], pre.T[w], base, fd, P, dP);
                    rs::loss_row<GRAD, SIMPLE>(P, dP, Mv, inv_s, L, G);
                }
                row += kBlock;
            }
            for (; row < N; row += kBlock) {
                const size_t idx = (size_t)fr.off + row;
                d3 P, dP;
                residual_row64_auto<GRAD>(sp, q.rays.q0[idx], q.rays.q1[idx], q.rays.q2[idx], q.rays.q3[idx], base, fd, P, dP);
                rs::loss_row<GRAD, SIMPLE>(P, dP, Mv, inv_s, L, G);
            }
            Lw[w] = wave_sum_f64(L);
            if (GRAD) Gw[w] = wave_sum_f64(G);
        }
    }
    L_out = Lw[0] + Lw[1] + Lw[2] + Lw[3];
    G_out = GRAD ? (Gw[0] + Gw[1] + Gw[2] + Gw[3]) * q.fs : 0.0;
}

// K1 for frames of up to 512 tracks: one WAVE per slot, the delays one after the other (same bits as loss64_kernel:
// tests/test_gpu_mid_sizes.py::test_one_wave_loss_kernel_equals_the_workgroup_kernel)
template <bool GRAD, bool SIMPLE>
__global__ __launch_bounds__(64, 3) void loss64_small_kernel(Loss64Params p) {
    extern __shared__ d4 s_win[]; // [4 * win_cap]
    const int lane = threadIdx.x;
    const uint32_t sf = p.slots ? p.slots[blockIdx.x] : blockIdx.x + p.slot0;
    const uint32_t g = p.grp ? p.grp[sf] : 0u;
    const d3 Mv = SIMPLE ? d3{0.0, 0.0, 0.0} : d3{p.M[3 * sf], p.M[3 * sf + 1], p.M[3 * sf + 2]};
    const double kk = p.k[sf];
    for (uint32_t b = 0; b < p.n_delays; ++b) {
        const double fd = p.fd[b * p.n_grp + g];
        double Lv = 0.0, Gv = 0.0;
        if (fd == fd) loss64_wave<GRAD, SIMPLE>(p, sf, Mv, kk, p.kd[b * p.n_grp + g], fd, s_win, Lv, Gv); // (NaN: switched off, zeros)
        if (lane == 0) {
            p.part_loss[(size_t)b * p.n_sel + sf] = Lv;
            if (GRAD) p.part_grad[(size_t)b * p.n_sel + sf] = Gv;
        }
    }
}

// ---------------------------------------------------------------------------
// K3: per-slot L-BFGS on the motion vector, P resident in registers in fp64.
// Restates ens::L_BFGS as called at core_private.cpp:264-294 (MaxIterations 200,
// MinGradientNorm 1e-4, library defaults otherwise) from the published ensmallen 2.x
// algorithm (lbfgs_impl.hpp; the dependency is unpinned and not under the reference tree).
// When a line search's best step is not its last, the published LineSearch moves the iterate
// to the best step and leaves value and gradient as the last trial computed them; that is the
// default (reeval = 0).  reeval = 1 evaluates once more at the best step so that
// (x, f, g) stay consistent (round 1's choice; kept for comparison).  Control flow is uniform:
// every thread runs the same fp64 scalar logic on the same reduced sums.
//
// The kernel also FINISHES FrameState::GuessMotion / GuessK (core_private.cpp:125-133) in fp64:
// the LMedS search over 200 hypotheses runs in the fp32 tile kernel and leaves only the index of
// the winning hypothesis per slot (init_h); here the winning pair of rows is recomputed in fp64,
// M = safe_normalize(P[i0] x P[i1]) (core_private.cpp:45-46) and k = clamp(100 / |P M|, 10, 1000).

struct Motion64Params {
    Rays64 rays;
    const FrameRec* frames;
    const uint32_t* sel;
    uint32_t n_sel;
    const d4* coef;
    int n_knots;
    const int32_t* kd; // [n_grp]
    const double* fd;  // NaN = skip the group's slots
    const uint32_t* grp;
    double* M; // per selection slot
    double* k;
    unsigned long long* stats; // [0] += iterations, [1] += evaluations, [2] += line searches with best != last step
    uint32_t* per_frame;       // optional [n_sel][2]: iterations, evaluations
    int reeval;
    int max_iters; // 200 (core_private.cpp:265); 0 = finish the initialisation only
    // pending initialisation (GuessMotion finish): winning hypothesis per slot, INT_MIN = none pending
    int32_t* init_h;
    uint64_t seed;
    uint32_t stream_base, stream_stride; // sampler stream = base + group * stride
    const uint32_t* win_stream;          // or, if not null (window executor): win_stream[group]
    int simple_k; // 1: k = clamp(100 / sqrt(sum |P_j|^2)) only (no-translation variant), no M, no optimisation
    uint32_t slot0; // the launch covers slots slot0 .. slot0 + gridDim.x, or (order != null) the entries order[slot0 + b]
    // Longest first: workgroup b of the launch takes slot order[slot0 + b] (null: slot0 + b), a permutation of the
    // launch's slots -- the slots of one size class inside one stream group's range -- sorted by how many evaluations each needed in the previous launch (motion_order_kernel); a
    // launch is as long as its slowest frame plus whatever is queued behind it, and the frames that need 60
    // evaluations instead of 20 are the same ones from one outer iteration to the next.  Which workgroup computes a
    // slot changes nothing in the slot's result.
    const uint32_t* order;
    uint32_t* evals_out; // [n_sel]: evaluations of this launch per slot (input of the next ordering), or null
    // frames of more than 8192 tracks (RPT = 0): the rows of P per workgroup in global memory instead of registers,
    // 3 x scratch_rows doubles (x, y, z planes; scratch_rows a multiple of the workgroup size) per ENTRY of the class's
    // slot list -- workgroup b of the launch uses entry scratch0 + b (launches of several stream groups run side by
    // side on disjoint ranges of the list)
    double* scratch;
    uint32_t scratch_rows;
    uint32_t scratch0;
    uint32_t win_cap; // knots of the spline window (dynamic LDS: win_cap x 128 bytes, or x 64 with win_compact)
    uint32_t win_compact; // 1: Spline64::compact
    uint32_t win_bytes;   // bytes of that LDS region (EMU4: once the window has done its work the rows of P live there)
};

// -DRSSYNC_K3_TIMING=1 (with -DRSSYNC_K2_COUNTERS=1, whose counter array it shares): core-clock ticks of wave 0 of
// every workgroup of the motion kernel -- [10] inside evaluations, [11] evaluations, [12] whole L-BFGS, [13] frames
// optimised, [14] prologue (rows of P, GuessMotion finish)   (tools/gpu_k3_timing.py)
#ifndef RSSYNC_K3_TIMING
#define RSSYNC_K3_TIMING 0
#endif
#if RSSYNC_K3_TIMING && RSSYNC_K2_COUNTERS
#define K3_T0() const long long k3t0__ = clock64()
#define K3_ACC(v) do { (v) += clock64() - k3t0__; } while (0) // into a register: one atomic per workgroup at the end
#define K3_ADD(i, n) do { if (threadIdx.x == 0) { atomicAdd(&g_k2_counters[i], (unsigned long long)(clock64() - k3t0__)); if (n) atomicAdd(&g_k2_counters[n], 1ull); } } while (0)
#define K3_FLUSH(ev) do { if (threadIdx.x == 0) { atomicAdd(&g_k2_counters[10], (unsigned long long)(ev).t_eval); atomicAdd(&g_k2_counters[11], (unsigned long long)(ev).evals); } } while (0)
#else
#define K3_T0() do { } while (0)
#define K3_ACC(v) do { } while (0)
#define K3_ADD(i, n) do { } while (0)
#define K3_FLUSH(ev) do { } while (0)
#endif

constexpr int kInitNone = (int)0x80000000;
constexpr int kNB = rs::kLbfgsBasis; // numBasis (ens::L_BFGS default)

// NW = waves per workgroup (4 in the product: measured fastest, see launch_motion64)
// EMU4 (with RPT = 0, NW = 1): ONE wave evaluates the FOUR-wave kernel's association -- thread t of that kernel's four
// waves holds rows t, t + 256, ...; lane l walks the rows of threads l, 64 + l, 128 + l, 192 + l in turn ("virtual waves"),
// sums each virtual wave with wave_sum_f64 and adds the four wave sums left to right, exactly as the workgroup does
// through LDS.  The window executor runs frames of more than 512 tracks this way (executor.hpp): the bits of
// opt_motion64_kernel<R, 4>, whatever R (a thread adds its rows in order, rows beyond the frame are zeros).
template <int RPT, int NW, bool EMU4 = false>
struct MotionEval64 {
    static_assert(!EMU4 || (RPT == 0 && NW == 1), "the emulation reads its rows from global memory, one wave");
    d3 P[RPT ? RPT : 1];
    const double* gP; // RPT == 0: the rows in global memory, plane stride g_rows, rpt rows per thread (zero beyond the frame)
    uint32_t g_rows;
    int rpt;          // rows per thread (EMU4: per thread of the emulated four-wave workgroup)
    // EMU4: the same rows in the wave's LDS region where they fit (planes of n_lds rows, 24 bytes per row; rows beyond the
    // frame are not stored: they are zeros) -- an L-BFGS reads them ~30 times
    const __attribute__((address_space(3))) double* lP = nullptr;
    uint32_t n_lds = 0;
    double (*part)[NW][4]; // [2][NW][4] LDS, double-buffered
    int buf;
    double k2;
    int evals;
    long long t_eval = 0; // (timing build)

    // loss and dL/dM at x (core_private.cpp:99-114 in closed form).  With u_j = (P_j.x)^2 / s, s = |x|^2 / k^2:
    //   dL/dx = t - (sum_j w_j u_j / s) 2x / k^2,   t = sum_j w_j (2 P_j.x / s) P_j,   w_j = 1 / (1 + u_j),
    // and x.t = 2 sum_j w_j u_j, so the second term is x (x.t) / |x|^2: the loss does not depend on |x|, its
    // gradient is t without its component along x -- four sums over the rows instead of five.
    __device__ __forceinline__ double operator()(const double x[3], double g[3]) {
        K3_T0();
        double inv_xx;
        const double inv_s = rs::motion_inv_s(x, k2, &inv_xx);
        double t[4];
        if constexpr (EMU4) {
#pragma unroll 1
            for (int vw = 0; vw < 4; ++vw) {
                double L = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
                if (n_lds) {
                    for (int j = 0; j < rpt; ++j) {
                        const uint32_t i = (uint32_t)j * 256u + (uint32_t)vw * 64u + threadIdx.x;
                        const bool in = i < n_lds;
                        const uint32_t q = in ? i : 0u;
                        const d3 Pi = d3{in ? lP[q] : 0.0, in ? lP[n_lds + q] : 0.0, in ? lP[2u * n_lds + q] : 0.0};
                        rs::motion_row(Pi, x, inv_s, L, a0, a1, a2);
                    }
                } else
                for (int j = 0; j < rpt; ++j) {
                    const size_t i = (size_t)j * 256 + (size_t)vw * 64 + threadIdx.x;
                    rs::motion_row(d3{gP[i], gP[g_rows + i], gP[2 * (size_t)g_rows + i]}, x, inv_s, L, a0, a1, a2);
                }
                const double r0 = wave_sum_f64(L), r1 = wave_sum_f64(a0), r2 = wave_sum_f64(a1), r3 = wave_sum_f64(a2);
                if (vw == 0) { t[0] = r0; t[1] = r1; t[2] = r2; t[3] = r3; }
                else { t[0] += r0; t[1] += r1; t[2] += r2; t[3] += r3; }
            }
        } else {
        double L = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
        if constexpr (RPT != 0) {
#pragma unroll
            for (int j = 0; j < RPT; ++j) rs::motion_row(P[j], x, inv_s, L, a0, a1, a2);
        } else {
            for (int j = 0; j < rpt; ++j) {
                const size_t i = (size_t)j * (64 * NW) + threadIdx.x;
                rs::motion_row(d3{gP[i], gP[g_rows + i], gP[2 * (size_t)g_rows + i]}, x, inv_s, L, a0, a1, a2);
            }
        }
        double r0 = wave_sum_f64(L), r1 = wave_sum_f64(a0), r2 = wave_sum_f64(a1), r3 = wave_sum_f64(a2);
        t[0] = r0; t[1] = r1; t[2] = r2; t[3] = r3;
        if (NW > 1) { // the waves' sums through LDS; a one-wave frame has them already
            const int wave = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) {
                part[buf][wave][0] = r0; part[buf][wave][1] = r1; part[buf][wave][2] = r2;
                part[buf][wave][3] = r3;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double acc = part[buf][0][q];
#pragma unroll
                for (int w = 1; w < NW; ++w) acc += part[buf][w][q];
                t[q] = acc;
            }
            buf ^= 1;
        }
        }
        ++evals;
        const double fv = rs::motion_finish(x, inv_xx, t, g);
        K3_ACC(t_eval);
        return fv;
    }
};

// the L-BFGS history of rs::lbfgs3 in LDS: control flow is uniform over the workgroup, every thread runs the
// same scalar logic on the same sums; thread 0 stores a pair between two barriers
struct LbfgsHistLds {
    double (*s_S)[3];
    double (*s_Y)[3];
    double* s_inv_ys;
    double* s_rho;
    double* s_alpha;
    __device__ __forceinline__ const double* S(int i) const { return s_S[i]; }
    __device__ __forceinline__ const double* Y(int i) const { return s_Y[i]; }
    __device__ __forceinline__ double inv_ys(int i) const { return s_inv_ys[i]; }
    // two-loop scratch: every thread writes the same values and reads them back itself; the barrier inside each
    // evaluation separates one iteration's use from the next
    __device__ __forceinline__ double& rho(int i) { return s_rho[i]; }
    __device__ __forceinline__ double& alpha(int i) { return s_alpha[i]; }
    __device__ __forceinline__ void store(int op, const double sv[3], const double yv[3]) {
        __syncthreads(); // every thread has finished reading the history for this iteration
        if (threadIdx.x == 0) {
            for (int c = 0; c < 3; ++c) { s_S[op][c] = sv[c]; s_Y[op][c] = yv[c]; }
            s_inv_ys[op] = 1.0 / rs::dot3(yv, sv);
        }
        __syncthreads();
    }
};

__device__ __forceinline__ double clamp_k(double k) { return rs::clamp_k64(k); } // inline_utils.hpp:50

// LDS of one workgroup's work on a slot: the kernel below owns one; the window executor (executor.hpp) lends its own
// (the spline window, win_cap x 128 bytes, is the kernel's dynamic LDS and handed to the body separately)
template <int NW>
struct MotionLds {
    double part[2][NW][4];
    double S[kNB][3], Y[kNB][3];
    // two-loop scratch: every thread writes the same values and reads them back itself;
    // the barrier inside each evaluation separates one iteration's use from the next
    double rho[kNB], alpha[kNB];
    double inv_ys[kNB]; // 1 / (y . s) of each stored pair: the value the two-loop recursion divides for
    double red[NW];
};

template <int RPT, int NW, bool SC1 = false, bool EMU4 = false> // SC1: M, k, the pending winners and the delays are written by other workgroups of this launch; EMU4: MotionEval64
__device__ __forceinline__ void opt_motion64_body(const Motion64Params& p, uint32_t sf, MotionLds<NW>& lds, d4* s_win, double* mk_out = nullptr,
                                                  uint32_t scratch_entry = 0) {
    constexpr int kThreads = 64 * NW;
    double (*s_part)[NW][4] = lds.part;
    double (*s_S)[3] = lds.S;
    double (*s_Y)[3] = lds.Y;
    double* s_rho = lds.rho;
    double* s_alpha = lds.alpha;
    double* s_inv_ys = lds.inv_ys;
    double* s_red = lds.red;
    const int tid = threadIdx.x;
    const uint32_t fi = p.sel[sf];
    const FrameRec fr = p.frames[fi];
    const uint32_t N = fr.n;
    const uint32_t grp = p.grp ? p.grp[sf] : 0u;
    const int kd = ld_m<SC1>(&p.kd[grp]);
    const double fd = ld_m<SC1>(&p.fd[grp]);
    // (everything the slot may need from memory is requested at once: with SC1 a load is a round trip of ~1.5 us, and
    // the branches below would put three of them one after the other)
    const int pend = p.init_h ? ld_m<SC1>(&p.init_h[sf]) : kInitNone;
    const double m_x = ld_m<SC1>(&p.M[3 * sf]), m_y = ld_m<SC1>(&p.M[3 * sf + 1]), m_z = ld_m<SC1>(&p.M[3 * sf + 2]);
    const double m_k = ld_m<SC1>(&p.k[sf]);
    if (fd != fd) { // this window is not being optimised in this call (workgroup-uniform)
        if (tid == 0 && p.evals_out) p.evals_out[sf] = 0;
        if (mk_out) { mk_out[0] = m_x; mk_out[1] = m_y; mk_out[2] = m_z; mk_out[3] = m_k; } // (what memory holds)
        return;
    }

    Spline64 sp;
    sp.g = p.coef;
    sp.n = p.n_knots;
    sp.cap = (int)p.win_cap;
    sp.compact = (int)p.win_compact;
    frame_window64(sp, s_win, fr, kd);
    __syncthreads();

    MotionEval64<RPT, NW, EMU4> ev;
    ev.part = s_part;
    ev.buf = 0;
    ev.evals = 0;
    const int base = fr.base_knot + kd;
    ev.gP = nullptr;
    ev.g_rows = 0;
    ev.rpt = RPT;
    if constexpr (RPT != 0) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const uint32_t row = j * kThreads + tid;
            d3 P = d3{0, 0, 0}, dP;
            if (row < N) residual_row64<false>(sp, p.rays, (size_t)fr.off + row, base, fd, P, dP);
            ev.P[j] = P; // zero rows contribute log1p(0) = 0 and no gradient
        }
    } else { // the same rows, kept in this workgroup's scratch (only this thread reads what it writes)
        double* gp = p.scratch + (size_t)scratch_entry * 3 * p.scratch_rows;
        ev.gP = gp;
        ev.g_rows = p.scratch_rows;
        constexpr int kGroup = EMU4 ? 256 : kThreads; // threads of the workgroup whose association is evaluated
        ev.rpt = (int)((N + kGroup - 1) / kGroup);
        for (uint32_t row = tid; row < (uint32_t)ev.rpt * kGroup; row += kThreads) {
            d3 P = d3{0, 0, 0}, dP;
            if (row < N) residual_row64<false>(sp, p.rays, (size_t)fr.off + row, base, fd, P, dP);
            gp[row] = P.x; gp[p.scratch_rows + row] = P.y; gp[2 * (size_t)p.scratch_rows + row] = P.z;
        }
        // (EMU4 as well: lane l writes the rows = l mod 64 and reads exactly those -- only this thread reads what it writes)
    }

    double x[3];
    double kk;
    if (p.simple_k || pend != kInitNone) {
        // GuessMotion's winner recomputed in fp64, then GuessK (core_private.cpp:125-133)
        d3 Mv = d3{0, 0, 0};
        if (!p.simple_k && pend >= 0) {
            uint32_t i0, i1;
            rs::sample_pair(p.seed, fr.id, p.win_stream ? ld_m<SC1>(&p.win_stream[grp]) : p.stream_base + grp * p.stream_stride, (uint32_t)pend, N, i0, i1);
            d3 P0, P1, dP;
            residual_row64<false>(sp, p.rays, (size_t)fr.off + i0, base, fd, P0, dP);
            residual_row64<false>(sp, p.rays, (size_t)fr.off + i1, base, fd, P1, dP);
            Mv = rs::cross(P0, P1);
            const double nn = sqrt(rs::dot(Mv, Mv));
            if (!(nn < 1e-12)) Mv = rs::scale(Mv, 1.0 / nn); // safe_normalize, inline_utils.hpp:5-11
        }
        double tot;
        if constexpr (EMU4) { // the four virtual waves' sums, added left to right
            tot = 0.0;
            for (int vw = 0; vw < 4; ++vw) {
                double ss = 0.0;
                for (int j = 0; j < ev.rpt; ++j) {
                    const size_t i = (size_t)j * 256 + (size_t)vw * 64 + tid;
                    const d3 Pj = d3{ev.gP[i], ev.gP[ev.g_rows + i], ev.gP[2 * (size_t)ev.g_rows + i]};
                    const double pm = p.simple_k ? sqrt(rs::dot(Pj, Pj)) : rs::dot(Pj, Mv);
                    ss = fma(pm, pm, ss);
                }
                const double sw = wave_sum_f64(ss);
                tot = vw == 0 ? sw : tot + sw;
            }
        } else {
        double ss = 0.0;
        if constexpr (RPT != 0) {
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const double pm = p.simple_k ? sqrt(rs::dot(ev.P[j], ev.P[j])) : rs::dot(ev.P[j], Mv);
                ss = fma(pm, pm, ss);
            }
        } else {
            for (int j = 0; j < ev.rpt; ++j) {
                const size_t i = (size_t)j * kThreads + tid;
                const d3 Pj = d3{ev.gP[i], ev.gP[ev.g_rows + i], ev.gP[2 * (size_t)ev.g_rows + i]};
                const double pm = p.simple_k ? sqrt(rs::dot(Pj, Pj)) : rs::dot(Pj, Mv);
                ss = fma(pm, pm, ss);
            }
        }
        const double sw = wave_sum_f64(ss);
        if ((tid & 63) == 0) s_red[tid >> 6] = sw;
        __syncthreads();
        tot = s_red[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) tot += s_red[w];
        }
        kk = clamp_k(100.0 / sqrt(tot)); // :132; tot = 0 gives +inf -> 1000
        x[0] = Mv.x; x[1] = Mv.y; x[2] = Mv.z;
        if (tid == 0) {
            if (!p.simple_k) { st_m<SC1>(&p.M[3 * sf], x[0]); st_m<SC1>(&p.M[3 * sf + 1], x[1]); st_m<SC1>(&p.M[3 * sf + 2], x[2]); }
            st_m<SC1>(&p.k[sf], kk);
            if (p.init_h) st_m<SC1>(&p.init_h[sf], (int32_t)kInitNone);
        }
    } else {
        x[0] = m_x; x[1] = m_y; x[2] = m_z;
        kk = m_k;
    }
    if (mk_out) { mk_out[0] = x[0]; mk_out[1] = x[1]; mk_out[2] = x[2]; mk_out[3] = kk; }
    if (p.max_iters <= 0 || p.simple_k) return;
    ev.k2 = kk * kk;
    if constexpr (EMU4) {
        // the spline window has done its work: the rows of P take its place where they fit (a lane copies the rows it wrote)
        if ((size_t)N * 24u <= (size_t)p.win_bytes) {
            __syncthreads();
            __attribute__((address_space(3))) double* l = (__attribute__((address_space(3))) double*)reinterpret_cast<double*>(s_win);
            for (uint32_t row = tid; row < N; row += kThreads) {
                l[row] = ev.gP[row];
                l[N + row] = ev.gP[ev.g_rows + row];
                l[2 * N + row] = ev.gP[2 * (size_t)ev.g_rows + row];
            }
            __syncthreads();
            ev.lP = l;
            ev.n_lds = N;
        }
    }

    LbfgsHistLds hist{s_S, s_Y, s_inv_ys, s_rho, s_alpha};
    int best_not_last = 0;
    K3_T0();
    const int it = rs::lbfgs3(ev, hist, x, p.max_iters /* core_private.cpp:265 */, p.reeval, &best_not_last);
    K3_ADD(12, 13);
    K3_FLUSH(ev);
    if (mk_out) { mk_out[0] = x[0]; mk_out[1] = x[1]; mk_out[2] = x[2]; }
    if (tid == 0) {
        st_m<SC1>(&p.M[3 * sf], x[0]); st_m<SC1>(&p.M[3 * sf + 1], x[1]); st_m<SC1>(&p.M[3 * sf + 2], x[2]);
        if (p.stats) {
            atomicAdd(&p.stats[0], (unsigned long long)it);
            atomicAdd(&p.stats[1], (unsigned long long)ev.evals);
            atomicAdd(&p.stats[2], (unsigned long long)best_not_last);
        }
        if (p.per_frame) {
            p.per_frame[2 * sf] = (uint32_t)it;
            p.per_frame[2 * sf + 1] = (uint32_t)ev.evals;
        }
        if (p.evals_out) p.evals_out[sf] = (uint32_t)ev.evals;
    }
}

template <int RPT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? (RPT == 0 ? 2 : (RPT <= 8 ? 3 : (RPT <= 24 ? 2 : 1))) : (NW == 1 ? (RPT >= 8 ? 3 : 4) : 2)) void opt_motion64_kernel(Motion64Params p) {
    __shared__ MotionLds<NW> lds;
    extern __shared__ d4 s_motion_win[]; // [4 * win_cap]
    opt_motion64_body<RPT, NW>(p, p.order ? p.order[blockIdx.x + p.slot0] : blockIdx.x + p.slot0, lds, s_motion_win, nullptr,
                               p.scratch0 + blockIdx.x);
}

// order[pos0 .. pos0 + count) = the slots list[pos0 .. pos0 + count) (the slots of one size class inside one stream
// group's range, ascending) sorted by evals[slot] descending (a counting sort over min(evals, 255) in one workgroup; the
// order among equal counts is whatever the atomics give -- it only decides which workgroup computes which slot)
__global__ __launch_bounds__(1024) void motion_order_kernel(const uint32_t* __restrict__ evals, const uint32_t* __restrict__ list,
                                                            uint32_t* __restrict__ order, uint32_t pos0, uint32_t count) {
    __shared__ uint32_t s_bin[256];
    for (uint32_t b = threadIdx.x; b < 256; b += blockDim.x) s_bin[b] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
        const uint32_t e = evals[list[pos0 + i]];
        atomicAdd(&s_bin[e < 255u ? e : 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) { // start of each bin, largest count first
        uint32_t at = 0;
        for (int b = 255; b >= 0; --b) {
            const uint32_t n = s_bin[b];
            s_bin[b] = at;
            at += n;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
        const uint32_t s = list[pos0 + i];
        const uint32_t e = evals[s];
        const uint32_t pos = atomicAdd(&s_bin[e < 255u ? e : 255u], 1u);
        order[pos0 + pos] = s;
    }
}

// debug: fp64 P (and dP/dd) rows of one frame, as the Sync kernels compute them
struct Debug64Params {
    Rays64 rays;
    const FrameRec* frames;
    uint32_t fi;
    const d4* coef;
    int n_knots;
    double fs;
    int32_t kd;
    double fd;
    double* P;
    double* dP;
};

__global__ __launch_bounds__(kBlock) void debug_problem64_kernel(Debug64Params p) {
    __shared__ d4 s_win[4 * kWinMax];
    const FrameRec fr = p.frames[p.fi];
    Spline64 sp;
    sp.g = p.coef;
    sp.n = p.n_knots;
    sp.cap = kWinMax;
    sp.compact = 0;
    frame_window64(sp, s_win, fr, p.kd);
    __syncthreads();
    for (uint32_t row = blockIdx.x * kBlock + threadIdx.x; row < fr.n; row += gridDim.x * kBlock) {
        d3 P, dP;
        residual_row64<true>(sp, p.rays, (size_t)fr.off + row, fr.base_knot + p.kd, p.fd, P, dP);
        p.P[3 * row] = P.x; p.P[3 * row + 1] = P.y; p.P[3 * row + 2] = P.z;
        if (p.dP) { p.dP[3 * row] = dP.x * p.fs; p.dP[3 * row + 1] = dP.y * p.fs; p.dP[3 * row + 2] = dP.z * p.fs; }
    }
}

// debug: the fp64 building blocks whose bits the CPU stand-in must reproduce (tests/test_gpu_bitexact.py names
// the operation if one ever differs): op 0 a / b, 1 sqrt(a), 2 log1p_rcp_f64(a) -> {value, 1/(1+a)},
// 3 fma(a, b, a), 4 the wave sum of each block of 64 values of a (out[block]), 5 a / 3 by rs::div3_exact (the compact spline windows)
__global__ __launch_bounds__(64) void debug_math64_kernel(int op, const double* __restrict__ a, const double* __restrict__ b,
                                                          double* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    const double x = i < n ? a[i] : 0.0, y = (i < n && b) ? b[i] : 0.0;
    if (op == 4) {
        const double t = wave_sum_f64(x);
        if (threadIdx.x == 0) out[blockIdx.x] = t;
        return;
    }
    if (i >= n) return;
    if (op == 0) out[i] = x / y;
    else if (op == 1) out[i] = sqrt(x);
    else if (op == 2) { double rc; out[2 * i] = rs::log1p_rcp_f64(x, &rc); out[2 * i + 1] = rc; }
    else if (op == 5) out[i] = rs::div3_exact(x);
    else out[i] = fma(x, y, x);
}

} // namespace

// (this header is only ever part of the HIP translation unit rssync_kernels.hip, whose default -- and whose fp32
// kernels' assumption -- is -ffp-contract=fast-honor-pragmas: back to it for the headers that follow)
#pragma clang fp contract(fast)
