// lmeds_small.hpp -- K2 for frames of up to 512 tracks: ONE WAVE per (frame, chunk of candidate delays).
// Part of the single HIP translation unit rssync_kernels.hip (included there, after lmeds.hpp).
//
// The reference's own data has ~130 tracks per frame.  The tile kernel (lmeds.hpp) spends a four-wave
// workgroup on such a frame: half its lanes have no row, and per candidate it crosses five workgroup barriers
// (tile written, hypotheses prepared by one wave while three wait, queue drained, two sums) for a few hundred
// instructions of work -- at 130 tracks it ran at ~12 k cycles per candidate with five workgroups per CU.
// Here a frame is one wave: its <= 4 rows per lane stay in registers for the whole candidate, the twenty
// hypotheses are tried one after the other in hypothesis order (the reference's sequential loop,
// core_private.cpp:41-56, with its strict "<": the first of equal medians wins), the best (quantile, index) lives
// in scalar registers, and nothing waits for another wave.  Arithmetic per row and per hypothesis is the tile
// kernel's, operation for operation (same P rows, same directions, same residual fma chain, same exact
// selection); only the order in which a frame's cost terms are added differs (one wave sum instead of four).
// Which of the two kernels a frame gets follows from the FRAME's own track count (up to 512: this one), like the motion
// kernel's shape, so that a frame's cost does not depend on the selection, the device or the rank it is evaluated in.
#pragma once

namespace {

#ifndef RSSYNC_K2S_KEEP_RAYS   // (1: a chunk's rays stay in registers across its candidates, up to four rows per lane: A/B of profiles/r6_k2s_keep_rays_ab.txt)
#define RSSYNC_K2S_KEEP_RAYS 0
#endif
constexpr int kSmallMaxRpt = 8; // 512 tracks (instantiated: 1, 2, 3, 4 rows per lane, and 8 for 257 .. 512 tracks)

// LDS of one wave's work on a (frame, chunk): the kernel below owns one; the window executor (executor.hpp) lends its own
// CAP = knots of the spline window inside this struct (kWinMax), or 0 = the window lives elsewhere (dynamic LDS sized
// for the problem's gyro rate; the body is given the pointer and reads the capacity from LmedsParams::win_cap)
template <int RPT, int CAP = kWinMax>
struct LmedsSmallLds {
    float n[3][64 * RPT]; // unit rows, for the hypotheses' row pairs
    f4 win[CAP ? 4 * CAP : 1];
    int kd[kMaxChunk];
    float fd[kMaxChunk];
};

// the work of one wave (64 threads, the whole workgroup) on slot sf, chunk `chunk`
// R64 = the fp64-rows form of the sweep (lmeds.hpp, "fp64 rows"; MODE 0, CAP 0): only the candidates of this (frame, chunk)
// that the fp32 launch flagged, their rows from the fp64 streams; s_nrm = 64 * RPT floats of LDS for the rows' norms
template <int RPT, int MODE, bool SC1 = false, int CAP = kWinMax, bool R64 = false> // SC1: the delays, the stream and the winners are shared with other workgroups of this launch
__device__ __forceinline__ void lmeds_small_body(const LmedsParams& p, uint32_t sf, uint32_t chunk, LmedsSmallLds<RPT, CAP>& lds,
                                                 f4* dyn_win = nullptr, float* s_nrm = nullptr) {
    static_assert(!R64 || (MODE == 0 && !SC1 && CAP == 0), "the fp64-rows form exists for the PreSync sweep only");
    constexpr int kHyp = kHypBatch;
    float (&s_n)[3][64 * RPT] = lds.n;
    f4* s_win = CAP ? lds.win : dyn_win;
    int* s_kd = lds.kd;
    float* s_fd = lds.fd;
    const int lane = threadIdx.x;
    if (sf >= p.n_sel) return;
    const uint32_t fi = p.sel[sf];
    const FrameRec fr = p.frames[fi];
    const uint32_t N = fr.n;
    const uint32_t kq = N / 4; // core_private.cpp:52
    const uint32_t g = p.grp ? p.grp[sf] : 0u;
    const Tile tile{s_n[0], s_n[1], s_n[2]};
    const RayRsrc rays = make_ray_rsrc(p.rays_a + fr.off, p.rays_b + fr.off, N);

    const uint32_t c0 = chunk * p.chunk;
    const uint32_t c1 = (c0 + p.chunk < p.n_cand) ? c0 + p.chunk : p.n_cand;
    if (c0 >= c1) return;
    uint32_t redo = 0; // R64: the flagged candidates of this chunk (bit i <-> candidate c0 + i)
    if constexpr (R64) {
        uint32_t* mw = p.redo_mask + (size_t)sf * p.mask_words;
        const uint32_t w0 = c0 >> 5, w1 = (c1 - 1u) >> 5;
        const unsigned long long both = (unsigned long long)mw[w0] | (w1 != w0 ? (unsigned long long)mw[w1] << 32 : 0ull);
        redo = (uint32_t)(both >> (c0 & 31u));
        if (c1 - c0 < 32u) redo &= (1u << (c1 - c0)) - 1u;
        redo = uniform_u32(redo);
        if (!redo) return;
        __syncthreads(); // (one wave: every lane has its copy before the bits are cleared)
        if (lane == 0) {
            atomicAnd(&mw[w0], ~(redo << (c0 & 31u)));
            if (w1 != w0) atomicAnd(&mw[w1], ~(uint32_t)((unsigned long long)redo >> (32u - (c0 & 31u))));
            atomicAdd(p.redo_count, (unsigned long long)__builtin_popcount(redo));
        }
    }
    if ((uint32_t)lane < c1 - c0) {
        s_kd[lane] = ld_m<SC1>(&p.kd[(c0 + lane) * p.n_grp + g]);
        s_fd[lane] = ld_m<SC1>(&p.fd[(c0 + lane) * p.n_grp + g]);
    }
    Spline sp;
    sp.g = p.coef;
    sp.n = p.n_knots;
    sp.cap = (int)p.win_cap;
    sp.whole_pair = p.win_whole_pair != 0;
    sp.path = kPathGlobal;
    if constexpr (!R64) { // (the fp64 rows read the fp64 table from L2: no window)
        int kd_lo = ld_m<SC1>(&p.kd[c0 * p.n_grp + g]), kd_hi = kd_lo;
        for (uint32_t c = c0 + 1; c < c1; ++c) {
            const int v = ld_m<SC1>(&p.kd[c * p.n_grp + g]);
            kd_lo = v < kd_lo ? v : kd_lo;
            kd_hi = v > kd_hi ? v : kd_hi;
        }
        stage_window_ends<CAP>(sp, s_win, frame_knots(fr, fr.base_knot + (int)floorf(fr.tmin) + kd_lo,
                                                       fr.base_knot + (int)floorf(fr.tmax) + kd_hi + 1, kd_lo, kd_hi), 64);
    }
    __syncthreads();

    uint32_t prev_best = kInfBits;
    const uint32_t voff = (uint32_t)lane * 16u;
    // the frame's rays, fetched once per chunk instead of once per candidate where that is few registers (8 per row)
    constexpr bool KEEP = RSSYNC_K2S_KEEP_RAYS && RPT <= 4 && MODE == 0 && !R64;
    f4 keepA[KEEP ? RPT : 1], keepB[KEEP ? RPT : 1];
    if constexpr (KEEP) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            keepA[j] = load_ray(rays.a, voff, (uint32_t)j * 64u * 16u);
            keepB[j] = load_ray(rays.b, voff, (uint32_t)j * 64u * 16u);
        }
    }
    for (uint32_t c = c0; c < c1; ++c) {
        if constexpr (R64) {
            if (!((redo >> (c - c0)) & 1u)) continue;
        }
        const int base = fr.base_knot + s_kd[c - c0];
        const float fd = s_fd[c - c0];
        const uint32_t stream = p.win_stream ? ld_m<SC1>(&p.win_stream[g]) + c : p.stream_base + c + g * p.stream_stride;
        uint32_t bad = 0;
        // ---- rows of P, as unit rows in registers (and in LDS for the row pairs); norms in registers ----
        float nx[RPT], ny[RPT], nz[RPT], nrm[RPT];
        // (the tile kernel's stage A, operation for operation -- including its reciprocal-free normalisation of the
        // interpolated quaternions and the fallback when they are not near unit length: common.hpp, NEWTON)
        auto rows = [&](auto fast_tag, RowWatch* watch) {
            constexpr bool FAST = decltype(fast_tag)::value; // (lmeds.hpp, lmeds_row: the hot form and the careful form)
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const uint32_t row = j * 64 + lane;
                f4 A, B;
                if constexpr (KEEP) { A = keepA[j]; B = keepB[j]; }
                else { A = load_ray(rays.a, voff, (uint32_t)j * 64u * 16u); B = load_ray(rays.b, voff, (uint32_t)j * 64u * 16u); }
                const float nan = __uint_as_float(0x7fc00000u);
                nx[j] = ny[j] = nz[j] = nan; // rows beyond N: their residuals compare above every threshold
                nrm[j] = 0.f;
                if (row < N) {
                    f3 P, dP;
                    if (sp.path == kPathInterior) residual_row<false, kPathInterior, MODE == 0, CAP, FAST>(sp, A, B, base, fd, P, dP, FAST ? &watch->qerr : nullptr);
                    else residual_row<false, kPathGlobal, false, CAP>(sp, A, B, base, fd, P, dP);
                    const float n2 = rs::dot(P, P);
                    // the near-static watch (lmeds.hpp): of the frame's first 64 rows -- the lanes active here -- how many are tiny
                    if (RSSYNC_NEAR_WATCH && j == 0 && watch) watch->near = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(n2 < kNearStatic2));
                    if (FAST) {
                        const float inv = rs::rsqrt_fast(n2);
                        nx[j] = P.x * inv; ny[j] = P.y * inv; nz[j] = P.z * inv;
                        nrm[j] = n2 * inv;
                        watch->n2min = min(watch->n2min, __float_as_uint(n2));
                        watch->nsum += nrm[j];
                    } else {
                        if (!finite_f(n2)) bad = RSHIP_BAD_P;
                        const bool tiny = n2 < 1e-24f; // safe_normalize (core_private.cpp:35-36)
                        const float inv = tiny ? 1.f : rs::rsqrt_fast(n2);
                        nx[j] = P.x * inv; ny[j] = P.y * inv; nz[j] = P.z * inv;
                        nrm[j] = tiny ? 1.f : n2 * inv;
                        if (watch) watch->n2min = min(watch->n2min, __float_as_uint(n2));
                    }
                }
                s_n[0][row] = nx[j]; s_n[1][row] = ny[j]; s_n[2][row] = nz[j];
            }
        };
        RowWatch watch;
        // the fp64 form of the rows (rows64.hpp: row64_unit): unit rows and norms from the fp64 streams, rounded once.  R64: the
        // sweep's second launch; MODE 1 (GuessMotion's search, one candidate per wave): taken IN PLACE when the watch fires -- in
        // the launch chain's kernel and in the window executor's search task alike (the same body: the same winner)
        int base64 = 0;
        double fd64 = 0.0;
        auto rows64 = [&]() {
            bad = 0;
            watch.n2min = 0x7f000000u;
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const uint32_t row = j * 64 + lane;
                const float nan = __uint_as_float(0x7fc00000u);
                nx[j] = ny[j] = nz[j] = nan;
                nrm[j] = 0.f;
                if (row < N) {
                    const Row64 r = row64_unit(p.src64, (size_t)fr.off + row, base64, fd64);
                    if (!r.finite) bad = RSHIP_BAD_P;
                    nx[j] = r.n.x; ny[j] = r.n.y; nz[j] = r.n.z;
                    nrm[j] = r.nrm;
                    watch.n2min = min(watch.n2min, __float_as_uint(r.n2));
                }
                s_n[0][row] = nx[j]; s_n[1][row] = ny[j]; s_n[2][row] = nz[j];
                if (s_nrm) s_nrm[row] = nrm[j];
            }
        };
        bool use64 = R64;
        if constexpr (R64) {
            base64 = fr.base_knot + p.kd64[c];
            fd64 = p.fd64[c];
            rows64();
        } else if (sp.path == kPathInterior) {
            rows(std::true_type{}, &watch);
            if (!finite_f(watch.nsum)) bad = RSHIP_BAD_P;
            if (__builtin_amdgcn_ballot_w64(watch.qerr >= kNewtonMaxErr || watch.below_safe_normalize()) != 0) {
                bad = 0;
                rows(std::false_type{}, nullptr);
            }
        } else {
            rows(std::false_type{}, &watch);
        }
        if constexpr (!R64) { // the near-static watch ("fp64 rows", lmeds.hpp): flag the pair for the fp64 form
            if (RSSYNC_NEAR_WATCH && MODE == 0 && p.redo_mask && near_static_fires(watch.near, N)) {
                if (lane == 0) { // (one lane: the flag word is an atomic per lane that carries a bit)
                    atomicOr(&p.redo_mask[(size_t)sf * p.mask_words + (c >> 5)], 1u << (c & 31u));
                    bad |= RSHIP_NEAR_STATIC;
                }
            }
            if (RSSYNC_NEAR_WATCH && MODE == 1 && p.src64.coef && near_static_fires(watch.near, N)) { // (wave-uniform)
                use64 = true;
                base64 = fr.base_knot + ld_m<SC1>(&p.kd64[c * p.n_grp + g]);
                fd64 = ld_m<SC1>(&p.fd64[c * p.n_grp + g]);
                rows64();
                if (lane == 0) atomicAdd(p.redo_count + 1, 1ull);
            }
        }
        __syncthreads(); // the wave's rows are in LDS
        // hypothesis(): the bound from the frame's smallest |P|^2, and |P_row| from the rays (the tile kernel's values, lmeds.hpp)
        const float smin2 = smin2_of(wave_min_u32(watch.n2min));
        auto row_scale = [&](uint32_t row) -> float {
            if constexpr (R64) return s_nrm[row]; // (the norms of the fp64 rows)
            else {
                if (MODE == 1 && use64) return row64_unit(p.src64, (size_t)fr.off + row, base64, fd64).nrm; // (the value stage A had)
                return row_scale_general(p.coef, p.n_knots, load_ray(rays.a, row * 16u, 0u), load_ray(rays.b, row * 16u, 0u), base, fd);
            }
        };

        // ---- the hypotheses, in order.  The previous candidate's best quantile (x1.25) is a provisional bound
        // as in the tile kernel; if nothing beats it the candidate is redone without it. ----
        uint32_t guess = kInfBits;
        if (prev_best < 0x7e000000u && prev_best > 0x00800000u) guess = uniform_u32(__float_as_uint(__uint_as_float(prev_best) * 1.25f));
        uint32_t T;
        int bH;
        f3 Mv;
        for (;;) {
            T = guess;
            bH = -1;
            Mv = f3{0, 0, 0};
            for (uint32_t batch = 0; batch < p.n_hyp; batch += kHyp) {
                const uint32_t nb = (p.n_hyp - batch < (uint32_t)kHyp) ? p.n_hyp - batch : (uint32_t)kHyp;
                f3 v = f3{0, 0, 0};
                if ((uint32_t)lane < nb) v = hypothesis(tile, p.seed, fr.id, stream, batch + lane, N, smin2, row_scale);
                for (uint32_t j = 0; j < nb; ++j) {
                    const float hx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), j));
                    const float hy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), j));
                    const float hz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.z), j));
                    // residuals r = nP v (core_private.cpp:48), two rows per packed operation, in the very expression
                    // of the tile kernel (so that the compiler contracts it into the same mul + fma + fma)
                    uint32_t r2[RPT];
#pragma unroll
                    for (int q = 0; q < RPT; q += 2) {
                        const int q1 = q + 1 < RPT ? q + 1 : q;
                        const v2f r01 = v2f{nx[q], nx[q1]} * hx + v2f{ny[q], ny[q1]} * hy + v2f{nz[q], nz[q1]} * hz;
                        r2[q] = __float_as_uint(r01.x);
                        if (q + 1 < RPT) r2[q + 1] = __float_as_uint(r01.y);
                    }
#if RSSYNC_TEST_VARIANTS
                    if (MODE == 0 && p.dump) { // (lmeds.hpp: LmedsParams::dump -- the residuals the selection below works on)
                        uint32_t* out = p.dump + (((size_t)c * p.n_sel + sf) * p.n_hyp + (batch + j)) * p.dump_rows;
#pragma unroll
                        for (int q = 0; q < RPT; ++q) {
                            const uint32_t row = q * 64 + lane;
                            if (row < N && row < p.dump_rows) out[row] = r2[q] & 0x7fffffffu;
                        }
                    }
#endif
                    // med < least_med (core_private.cpp:51-53): more than kq |residuals| below the best so far
                    uint32_t hi2 = T;
                    const uint32_t tot = wave_count_lt(r2, hi2);
                    if (tot > kq) {
                        if (hi2 == kInfBits) {
                            float mx = 0.f;
#pragma unroll
                            for (int q = 0; q < RPT; ++q) mx = fmaxf(mx, fabsf(__uint_as_float(r2[q])));
                            mx = wave_max_f32(mx);
                            if (finite_f(mx)) hi2 = __float_as_uint(mx) + 1u;
                        }
                        const uint32_t kth = select_kth(r2, kq, hi2, tot);
                        if (kth < T) { // (always, by the count; the comparison keeps "first wins" explicit)
                            T = kth;
                            bH = (int)(batch + j);
                            Mv = f3{hx, hy, hz};
                        }
                    }
                }
            }
            if (guess == kInfBits || bH >= 0) break;
            guess = kInfBits; // nothing beat the provisional bound: once more without it
        }
        const uint32_t bT = bH >= 0 ? T : kInfBits;
        prev_best = bT;
        if (!(finite_f(Mv.x) && finite_f(Mv.y) && finite_f(Mv.z))) bad |= RSHIP_BAD_M;
        if (MODE == 1) { // GuessMotion: only the winner's index leaves this kernel
            if (lane == 0) st_m<SC1>(&p.best_h[sf], (int32_t)bH);
            __syncthreads(); // the rows in LDS are read (hypotheses) before the next candidate rewrites them
            continue;
        }

        // ---- k = clamp(100 / |P M|), cost = sqrt(sum sqrt(log1p(r^2))) (core_private.cpp:79-85) ----
        float pm[RPT];
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            pm[q] = mul_zero_wins(nrm[q], rs::dot(f3{nx[q], ny[q], nz[q]}, Mv)); // (rows beyond N: norm 0, NaN entries)
            ss = fmaf(pm[q], pm[q], ss);
        }
        const double ss_tot = (double)wave_sum_f32(ss);
        float kf = 100.0f * rs::rsqrt_fast((float)ss_tot);
        kf = (kf < 10.f) ? 10.f : ((1000.f < kf) ? 1000.f : kf);
        {
            const float sc = kf * rs::rsqrt_fast(rs::dot(Mv, Mv));
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const float r = pm[q] * sc;
                acc += __builtin_amdgcn_sqrtf(rs::log1p_pos_fast(r * r));
            }
            if (!finite_f(acc)) { // (as in the tile kernel: which of the reference's two checks fires first, looked up only then)
                float rsum = 0.f;
#pragma unroll
                for (int q = 0; q < RPT; ++q) rsum += fabsf(pm[q] * sc);
                bad |= finite_f(rsum) ? RSHIP_BAD_RHO : RSHIP_BAD_R;
            }
            const double acc_tot = (double)wave_sum_f32(acc);
            if (lane == 0) {
                p.frame_cost[(size_t)c * p.n_sel + sf] = sqrt(acc_tot);
                if (p.best_h) p.best_h[(size_t)c * p.n_sel + sf] = bH;
            }
        }
        if (bad) atomicOr(p.flags, bad);
        __syncthreads(); // the rows in LDS are read (hypotheses) before the next candidate rewrites them
    }
}

template <int RPT, int MODE, int CAP = kWinMax, bool R64 = false>
// (MODE 1 -- GuessMotion's search, with the fp64 form of the rows as a branch -- at four waves per SIMD: no spills)
__global__ __launch_bounds__(64, R64 ? 1 : (MODE == 1 ? (RPT <= 4 ? 4 : 3) : (RPT <= 3 ? 6 : (RPT == 4 ? 5 : 3)))) void lmeds_small_kernel(LmedsParams p) {
    __shared__ LmedsSmallLds<RPT, CAP> lds;
    __shared__ float s_nrm[R64 ? 64 * RPT : 1];
    f4* dyn = nullptr;
    if constexpr (CAP == 0) {
        extern __shared__ f4 s_small_win_dynamic[];
        dyn = s_small_win_dynamic;
    }
    const uint32_t entry = blockIdx.x / p.n_chunks; // (the launch has n_slots x n_chunks workgroups)
    lmeds_small_body<RPT, MODE, false, CAP, R64>(p, p.slots ? p.slots[entry] : entry, blockIdx.x % p.n_chunks, lds, dyn, s_nrm);
}

} // namespace
