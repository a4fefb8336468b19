// lmeds_big.hpp -- K2 for frames of MORE than 8192 tracks (round 3: the reference accepts any count,
// core_private.cpp:192-203; the tile kernel's tile must fit LDS and a wave's registers).
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
//
// The slow, exact path: same rows, same hypotheses, same (quartile, index) arg-min and the same cost formula as
// lmeds_kernel, but the tile of unit rows lives in global memory (a per-workgroup scratch of 20 B per row: three
// coordinates, the norm, the residual key), the hypotheses of a candidate are taken in order by the whole
// workgroup, and the quartile of a hypothesis that beats the bound is found by counting passes of the whole
// workgroup over the keys.  A launch is a fixed number of workgroups that walk over the (frame, chunk) items, so the
// scratch does not grow with the problem.
// Round 6 gave it three cheap devices -- the previous candidate's winning quartile (x 1.25) as a provisional bound, as in the tile
// kernels (a candidate nothing beats it in is redone without it: the arg-min stays exact); narrow_kth's secant pivots instead
// of plain bisection of the bit pattern (~6 counting passes per quartile instead of 31); and ONE pass over the rows that counts
// all twenty hypotheses of a batch against the bound, so that only the survivors' keys are ever written and read (the tiles of
// all workgroups together exceed the L2: twenty sweeps per candidate were HBM traffic) -- and a direction is computed by one
// thread instead of by all 256: 218 -> 85 ms per 2^21 ray pairs x 800 candidates at 9000 tracks (profiles/r6_k2_big_ab.txt).
// The eight-wave tile kernel is 3.5 times faster per ray still (a tracker that produces such frames spends its time
// elsewhere: the reference sorts 10^4 residuals per hypothesis on one core).
#pragma once

namespace {

constexpr uint32_t kBigScratchFloats = 5; // per row: nx, ny, nz, |P|, key
constexpr int kBigBatch = 20; // hypotheses whose residuals one pass over the rows counts against the bound (PreSync tries 20 per candidate, GuessMotion 200: core_private.cpp:77,127)
#ifndef RSSYNC_BIG_OLD_SELECT   // (1: rounds 3-5's selection -- no provisional bound, plain bisection of the bit pattern: the A/B of profiles/r6_k2_big_ab.txt)
#define RSSYNC_BIG_OLD_SELECT 0
#endif

template <int MODE> // as lmeds_kernel: 0 PreSync cost per candidate, 1 GuessMotion's search
__global__ __launch_bounds__(kBlock) void lmeds_big_kernel(LmedsParams p) {
    __shared__ double s_red[2][4];
    __shared__ uint32_t s_cnt[2][4];
    __shared__ uint32_t s_near; // this candidate's rows are redone in fp64 (lmeds.hpp, "fp64 rows")
    __shared__ f4 s_hv[kBigBatch];        // the directions of the current batch of hypotheses
    __shared__ uint32_t s_bc[kBigBatch];  // ... and how many of the frame's |residuals| lie below the bound the batch started with
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t rows = p.scratch_rows; // a multiple of kBlock, >= the largest frame
    float* const mine = p.scratch + (size_t)blockIdx.x * rows * kBigScratchFloats;
    const Tile tile{mine, mine + rows, mine + 2 * (size_t)rows};
    float* const g_nrm = mine + 3 * (size_t)rows;
    uint32_t* const g_key = reinterpret_cast<uint32_t*>(mine + 4 * (size_t)rows);
    int buf = 0;

    // number of keys below T, over the whole workgroup (T: bit pattern, uniform)
    auto count_lt = [&](uint32_t N, uint32_t T) -> uint32_t {
        uint32_t cnt = 0;
        for (uint32_t row = tid; row < N; row += kBlock) cnt += g_key[row] < T ? 1u : 0u;
        const uint32_t w = wave_sum_u32(cnt);
        if (lane == 0) s_cnt[buf][wave] = w;
        __syncthreads();
        const uint32_t tot = s_cnt[buf][0] + s_cnt[buf][1] + s_cnt[buf][2] + s_cnt[buf][3];
        buf ^= 1; // (the next call's writes go to the other half: nobody is still reading it, a barrier lies between)
        return tot;
    };

    // smallest (key - base) over the frame's keys, wrapping (a key below base wraps to a huge difference), over the whole workgroup
    auto min_above = [&](uint32_t N, uint32_t base) -> uint32_t {
        uint32_t mn = 0xffffffffu;
        for (uint32_t row = tid; row < N; row += kBlock) { const uint32_t d = g_key[row] - base; mn = d < mn ? d : mn; }
        const uint32_t w = wave_min_u32(mn);
        if (lane == 0) s_cnt[buf][wave] = w;
        __syncthreads();
        uint32_t m = s_cnt[buf][0];
        for (int q = 1; q < 4; ++q) m = s_cnt[buf][q] < m ? s_cnt[buf][q] : m;
        buf ^= 1;
        return m;
    };
    // largest finite key (0 if none), over the whole workgroup
    auto max_finite = [&](uint32_t N) -> uint32_t {
        uint32_t mx = 0u;
        for (uint32_t row = tid; row < N; row += kBlock) { const uint32_t k = g_key[row]; mx = (k < kInfBits && k > mx) ? k : mx; }
        const uint32_t w = ~wave_min_u32(~mx);
        if (lane == 0) s_cnt[buf][wave] = w;
        __syncthreads();
        uint32_t m = s_cnt[buf][0];
        for (int q = 1; q < 4; ++q) m = s_cnt[buf][q] > m ? s_cnt[buf][q] : m;
        buf ^= 1;
        return m;
    };

    Spline sp;
    sp.g = p.coef;
    sp.n = p.n_knots;
    sp.lds = nullptr;
    sp.w0 = sp.wlen = 0;
    sp.path = kPathGlobal; // any parameter, table from L2

    const uint32_t total = p.n_slots * p.n_chunks;
    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
        const uint32_t entry = item / p.n_chunks, chunk = item % p.n_chunks;
        const uint32_t sf = p.slots ? p.slots[entry] : entry;
        const uint32_t fi = p.sel[sf];
        const FrameRec fr = p.frames[fi];
        const uint32_t N = fr.n;
        const uint32_t kq = N / 4; // core_private.cpp:52
        const uint32_t g = p.grp ? p.grp[sf] : 0u;
        const f4* ra = p.rays_a + fr.off;
        const f4* rb = p.rays_b + fr.off;
        const uint32_t c0 = chunk * p.chunk;
        const uint32_t c1 = (c0 + p.chunk < p.n_cand) ? c0 + p.chunk : p.n_cand;
        uint32_t prev_best = kInfBits; // winning quartile of the previous candidate of this chunk

        for (uint32_t c = c0; c < c1; ++c) {
            const int base = fr.base_knot + p.kd[c * p.n_grp + g];
            const float fd = p.fd[c * p.n_grp + g];
            const uint32_t stream = p.stream_base + c + g * p.stream_stride;
            uint32_t bad = 0;
            __syncthreads(); // the previous candidate's readers of the scratch are done
            // ---- stage A: unit rows and norms -> scratch ----
            RowWatch watch;
            for (uint32_t row = tid; row < N; row += kBlock) {
                float nrm;
                bad |= lmeds_row<kPathGlobal, false, kWinMax>(sp, ra[row], rb[row], N, row, base, fd, tile, nrm, &watch, row == (uint32_t)tid);
                g_nrm[row] = nrm;
            }
            // the near-static watch (lmeds.hpp, "fp64 rows"): thread 0's wave has counted the frame's first 64 rows.  This
            // kernel is the slow exact path anyway: the fp64 form of the rows is taken right here, no second launch.
            if (tid == 0) s_near = (p.src64.coef && near_static_fires(watch.near, N)) ? 1u : 0u;
            __syncthreads();
            if (s_near) { // (MODE 1 as well: GuessMotion's search takes the fp64 form in place in every kernel family)
                bad = 0;
                const int base64 = fr.base_knot + p.kd64[c * p.n_grp + g];
                const double fd64 = p.fd64[c * p.n_grp + g];
                for (uint32_t row = tid; row < N; row += kBlock) {
                    const Row64 r = row64_unit(p.src64, (size_t)fr.off + row, base64, fd64);
                    if (!r.finite) bad = RSHIP_BAD_P;
                    tile.nx[row] = r.n.x; tile.ny[row] = r.n.y; tile.nz[row] = r.n.z;
                    g_nrm[row] = r.nrm;
                }
                if (tid == 0) atomicAdd(p.redo_count + (MODE == 1 ? 1 : 0), 1ull);
                __syncthreads();
            }

            // ---- stage C: hypotheses in order; (T, bH) = best (quartile, index) so far, strict < (core_private.cpp:53).  The
            // previous candidate's winning quartile x 1.25 is a provisional bound, as in the tile kernels (lmeds.hpp): a
            // hypothesis with at most kq residuals below it is turned away after one counting pass; if nothing beats it the
            // candidate is redone without it, so the result is the exact arg-min either way.  (Every value below is uniform
            // over the workgroup: the counts come from LDS sums, the pivots from the same arithmetic in every thread.) ----
            uint32_t guess = kInfBits;
            if (!RSSYNC_BIG_OLD_SELECT && prev_best < 0x7e000000u && prev_best > 0x00800000u) guess = __float_as_uint(__uint_as_float(prev_best) * 1.25f);
            uint32_t T;
            int bH;
            f3 Mv;
            for (;;) {
            T = guess;
            bH = -1;
            Mv = f3{0, 0, 0};
            for (uint32_t batch = 0; batch < p.n_hyp; batch += (uint32_t)kBigBatch) {
                const uint32_t nb = (p.n_hyp - batch < (uint32_t)kBigBatch) ? p.n_hyp - batch : (uint32_t)kBigBatch;
                // the batch's directions, one thread each (the rows' norms are at hand: no bound, the reference's rule directly)
                __syncthreads(); // (the previous batch's readers of s_hv / s_bc are done; stage A's rows are visible)
                if ((uint32_t)tid < (uint32_t)kBigBatch) {
                    f3 v = f3{0, 0, 0};
                    if ((uint32_t)tid < nb) v = hypothesis(tile, p.seed, fr.id, stream, batch + tid, N, 0.f, [&](uint32_t row) -> float { return g_nrm[row]; });
                    s_hv[tid] = f4{v.x, v.y, v.z, 0.f};
                    s_bc[tid] = 0u;
                }
                __syncthreads();
                // ONE pass over the rows for the whole batch (round 6): how many |residuals| of each hypothesis lie below the bound
                // the batch starts with.  T only falls while the batch is worked through, so a hypothesis with at most kq below
                // THIS bound is out for certain -- without its keys ever being written or read (the tiles of all workgroups
                // together exceed the L2: 20 passes over rows and keys per candidate were HBM traffic).  No bound yet: all stay.
                const uint32_t T0 = T;
                if (!RSSYNC_BIG_OLD_SELECT && T0 != kInfBits) {
                    uint32_t cnt[kBigBatch];
#pragma unroll
                    for (int q = 0; q < kBigBatch; ++q) cnt[q] = 0u;
                    for (uint32_t row = tid; row < N; row += kBlock) {
                        const float x = tile.nx[row], y = tile.ny[row], z = tile.nz[row];
#pragma unroll
                        for (int q = 0; q < kBigBatch; ++q) {
                            const f4 hv = s_hv[q];
                            const float r = fmaf(z, hv.z, fmaf(y, hv.y, x * hv.x)); // :48, as sweep_tile (the keys' very expression)
                            cnt[q] += ((__float_as_uint(r) & 0x7fffffffu) < T0) ? 1u : 0u; // (a NaN's pattern is above every bound)
                        }
                    }
#pragma unroll
                    for (int q = 0; q < kBigBatch; ++q) {
                        const uint32_t w = wave_sum_u32(cnt[q]);
                        if (lane == 0 && w) atomicAdd(&s_bc[q], w);
                    }
                    __syncthreads();
                }
                for (uint32_t j = 0; j < nb; ++j) {
                    if (!RSSYNC_BIG_OLD_SELECT && T0 != kInfBits && s_bc[j] <= kq) continue; // (uniform: an LDS word)
                    const uint32_t h = batch + j;
                    const f4 hv4 = s_hv[j];
                    const f3 hv = f3{hv4.x, hv4.y, hv4.z};
                    for (uint32_t row = tid; row < N; row += kBlock) {
                        const float r = fmaf(tile.nz[row], hv.z, fmaf(tile.ny[row], hv.y, tile.nx[row] * hv.x)); // :48, as sweep_tile
                        const uint32_t a = __float_as_uint(r) & 0x7fffffffu;
                        g_key[row] = a > kInfBits ? 0xffffffffu : a; // NaN never counts
#if RSSYNC_TEST_VARIANTS
                        if (MODE == 0 && p.dump && row < p.dump_rows) p.dump[(((size_t)c * p.n_sel + sf) * p.n_hyp + h) * p.dump_rows + row] = a;
#endif
                    }
                    __syncthreads();
                    const uint32_t tot = count_lt(N, T);
                    if (tot > kq) { // quartile_h < T: find it.  Bracket [lo, hi): count(< lo) = c_lo <= kq < c_hi = count(< hi)
                        uint32_t lo = 0u, c_lo = 0u, hi = T, c_hi = tot;
                        if (hi == kInfBits) { // no bound yet: start the bracket at the largest residual (count(< hi) is still tot)
                            const uint32_t mx = max_finite(N);
                            hi = mx + 1u;
                        }
                        uint32_t a1 = lo, c1n = c_lo, a2 = hi, c2n = c_hi; // the two most recent (pivot, count) points (lmeds.hpp: narrow_kth)
                        for (int it = 0;; ++it) {
                            if (hi - lo == 1u) break;
                            if (c_hi - c_lo == 1u) { // the single element in [lo, hi): the smallest key >= lo
                                lo += min_above(N, lo);
                                hi = lo + 1u;
                                break;
                            }
                            uint32_t piv = 0;
                            if (!RSSYNC_BIG_OLD_SELECT && it < 24) {
                                if (c2n != c1n) { // secant through the last two points, aimed at rank kq + 1/2
                                    const float num = 0.5f * (float)(int)(2 * kq + 1 - 2 * c2n);
                                    const float a3 = fmaf(num * (__uint_as_float(a2) - __uint_as_float(a1)), rs::rcp_fast((float)(int)(c2n - c1n)), __uint_as_float(a2));
                                    piv = __float_as_uint(a3);
                                }
                                if (!(piv > lo && piv < hi)) { // interpolate inside the bracket instead
                                    const float num = 0.5f * (float)(int)(2 * kq + 1 - 2 * c_lo);
                                    const float a3 = fmaf(num * (__uint_as_float(hi) - __uint_as_float(lo)), rs::rcp_fast((float)(c_hi - c_lo)), __uint_as_float(lo));
                                    piv = __float_as_uint(a3);
                                }
                            }
                            if (!(piv > lo && piv < hi)) piv = lo + ((hi - lo) >> 1); // bit bisection: guaranteed finish
                            piv = uniform_u32(piv); // (the same in every lane by construction; said to the compiler)
                            const uint32_t cnt2 = count_lt(N, piv);
                            a1 = a2; c1n = c2n;
                            a2 = piv; c2n = cnt2;
                            if (cnt2 <= kq) { lo = piv; c_lo = cnt2; }
                            else { hi = piv; c_hi = cnt2; }
                        }
                        T = lo;
                        bH = (int)h;
                        Mv = hv;
                    }
                    __syncthreads(); // keys are rewritten by the next hypothesis
                }
            }
            if (guess == kInfBits || bH >= 0) break;
            guess = kInfBits; // nothing beat the provisional bound: once more without it
            }
            prev_best = bH >= 0 ? T : kInfBits;
            if (!(finite_f(Mv.x) && finite_f(Mv.y) && finite_f(Mv.z))) bad |= RSHIP_BAD_M;
            if (MODE == 1) {
                if (tid == 0) p.best_h[sf] = bH;
                if (bad) atomicOr(p.flags, bad);
                continue;
            }
            // ---- stage D: k = clamp(100 / |P M|), cost = sqrt(sum sqrt(log1p(r^2))), as lmeds_kernel ----
            float ss = 0.f;
            for (uint32_t row = tid; row < N; row += kBlock) {
                const float pm = g_nrm[row] * rs::dot(f3{tile.nx[row], tile.ny[row], tile.nz[row]}, Mv);
                ss = fmaf(pm, pm, ss);
            }
            const double ss_tot = block_sum(ss, s_red[0]);
            float kf = 100.0f * rs::rsqrt_fast((float)ss_tot); // core_private.cpp:79
            kf = (kf < 10.f) ? 10.f : ((1000.f < kf) ? 1000.f : kf);
            const float sc = kf * rs::rsqrt_fast(rs::dot(Mv, Mv)); // :80
            float acc = 0.f, rsum = 0.f;
            for (uint32_t row = tid; row < N; row += kBlock) {
                const float pm = g_nrm[row] * rs::dot(f3{tile.nx[row], tile.ny[row], tile.nz[row]}, Mv);
                const float r = pm * sc;
                rsum += fabsf(r);
                acc += __builtin_amdgcn_sqrtf(rs::log1p_pos_fast(r * r)); // :82
            }
            if (!finite_f(rsum)) bad |= RSHIP_BAD_R;
            else if (!finite_f(acc)) bad |= RSHIP_BAD_RHO;
            const double acc_tot = block_sum(acc, s_red[1]);
            if (tid == 0) {
                p.frame_cost[(size_t)c * p.n_sel + sf] = sqrt(acc_tot); // :85
                if (p.best_h) p.best_h[(size_t)c * p.n_sel + sf] = bH;
            }
            if (bad) atomicOr(p.flags, bad);
        }
    }
}

} // namespace
