// lmeds_big.hpp -- K2 for frames of MORE than 8192 tracks (round 3: the reference accepts any count,
// core_private.cpp:192-203; the tile kernel's tile must fit LDS and a wave's registers).
// Part of the single HIP translation unit rssync_kernels.hip (included there, in order).
//
// The slow, exact path: same rows, same hypotheses, same (quartile, index) arg-min and the same cost formula as
// lmeds_kernel, but the tile of unit rows lives in global memory (a per-workgroup scratch of 20 B per row: three
// coordinates, the norm, the residual key), the hypotheses of a candidate are taken in order by the whole
// workgroup, and the quartile of a hypothesis that beats the bound is found by counting passes of the whole
// workgroup over the keys (bisection of the bit pattern: at most 31 passes).  A launch is a fixed number of
// workgroups that walk over the (frame, chunk) items, so the scratch does not grow with the problem.
// Nothing here is tuned: a tracker that produces such frames spends its time elsewhere (the reference sorts
// 10^4 residuals per hypothesis on one core).
#pragma once

namespace {

constexpr uint32_t kBigScratchFloats = 5; // per row: nx, ny, nz, |P|, key

template <int MODE> // as lmeds_kernel: 0 PreSync cost per candidate, 1 GuessMotion's search
__global__ __launch_bounds__(kBlock) void lmeds_big_kernel(LmedsParams p) {
    __shared__ double s_red[2][4];
    __shared__ uint32_t s_cnt[2][4];
    __shared__ uint32_t s_near; // this candidate's rows are redone in fp64 (lmeds.hpp, "fp64 rows")
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

            // ---- stage C: hypotheses in order; (T, bH) = best (quartile, index) so far, strict < (core_private.cpp:53) ----
            uint32_t T = kInfBits;
            int bH = -1;
            f3 Mv = f3{0, 0, 0};
            for (uint32_t h = 0; h < p.n_hyp; ++h) {
                // (uniform: every thread computes it; the rows' norms are at hand: no bound, the reference's rule directly)
                const f3 hv = hypothesis(tile, p.seed, fr.id, stream, h, N, 0.f, [&](uint32_t row) -> float { return g_nrm[row]; });
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
                if (tot > kq) { // quartile_h < T: find it.  Bracket [lo, hi): count(< lo) <= kq < count(< hi)
                    uint32_t lo = 0u, hi = T;
                    while (hi - lo > 1u) {
                        const uint32_t mid = lo + ((hi - lo) >> 1);
                        if (count_lt(N, mid) > kq) hi = mid; else lo = mid;
                    }
                    T = lo;
                    bH = (int)h;
                    Mv = hv;
                }
                __syncthreads(); // keys are rewritten by the next hypothesis
            }
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
