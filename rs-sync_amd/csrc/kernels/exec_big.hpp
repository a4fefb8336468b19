// exec_big.hpp -- frames of MORE than 512 tracks inside the window executor: one wave does a four-wave workgroup's work.
// Part of the single HIP translation unit rssync_kernels.hip (included there, before executor.hpp).
//
// The executor's unit of work is a task executed by ONE wave (executor.hpp).  The reference's own frames (~130 tracks)
// are one-wave frames anyway; but one frame of a clip with 513 tracks used to send every window of the clip to the chain
// of launches (round 4: 21 -> 34 ms on the driver workload), because the kernel family followed the problem's largest
// frame.  With size classes (round 5) a frame's bits are those of ITS class's kernels -- so a frame of a four-wave class
// met inside the executor is evaluated by one wave IN THE FOUR-WAVE KERNELS' ASSOCIATION, and the executor's results stay
// the launch chain's, bit for bit (RSSYNC_EXECUTOR_CHECK compares them):
//   * GuessMotion's search (lmeds_kernel<R, 1, .> / lmeds_big_kernel<1>): the rows of P as the four waves of the tile
//     kernel compute them -- "virtual wave" v = threads 64 v .. 64 v + 63 holds rows 256 j + 64 v + lane; the hot form
//     with its per-WAVE fallback to the careful form (lmeds_rows) is taken per virtual wave -- into a tile in global
//     memory; the hypotheses in order with the exact lower quartile by bisection, as lmeds_big_kernel: the same winner as
//     the tile kernel's lazy selection (the exact arg-min of (quartile, index), tests/test_gpu_lazy_select.py).
//   * the motion L-BFGS: opt_motion64_body<0, 1, true, EMU4 = true> (sync64.hpp): rows of P in global memory, the four
//     sums of an evaluation as four virtual-wave sums added left to right.
//   * the loss / derivative / trials: loss64_wave is that emulation already (any track count).
// A 600-track frame costs its wave ~4x a 130-track frame's task -- about the motion phase's slowest 130-track frame.
#pragma once

namespace {

constexpr uint32_t kExecBigFloats = 5; // per row of a big entry's fp32 tile: nx, ny, nz, |P|, key (as lmeds_big.hpp)

// GuessMotion's 200-hypothesis search for slot sf (a frame of class k >= 1) by one wave; winner -> p.best_h[sf].
//   tile_base   this entry's fp32 tile in global memory (rows x 5 floats, rows a multiple of 256 >= the frame)
//   s_win       the wave's LDS region (region_bytes >= cap x 64 bytes)
//   cap, whole  the window the launch chain's search kernel would give this frame's class (plan_lmeds_window<1>): 80 and
//               whole pairs for the compiled-in window, else the dynamic window's knots; general = class 5 (no window:
//               lmeds_big_kernel reads the table from L2)
template <bool SC1>
__device__ __forceinline__ void exec_big_init(const LmedsParams& p, uint32_t sf, float* tile_base, uint32_t rows, f4* s_win, uint32_t region_bytes,
                                              uint32_t cap, bool whole, bool general) {
    const int lane = threadIdx.x;
    const uint32_t fi = p.sel[sf];
    const FrameRec fr = p.frames[fi];
    const uint32_t N = fr.n;
    const uint32_t kq = N / 4; // core_private.cpp:52
    const uint32_t g = p.grp ? p.grp[sf] : 0u;
    const Tile tile{tile_base, tile_base + rows, tile_base + 2 * (size_t)rows};
    float* const g_nrm = tile_base + 3 * (size_t)rows;
    uint32_t* const g_key = reinterpret_cast<uint32_t*>(tile_base + 4 * (size_t)rows);
    const f4* ra = p.rays_a + fr.off;
    const f4* rb = p.rays_b + fr.off;
    const int kd = ld_m<SC1>(&p.kd[g]);
    const float fd = ld_m<SC1>(&p.fd[g]);
    const int base = fr.base_knot + kd;
    const uint32_t stream = p.win_stream ? ld_m<SC1>(&p.win_stream[g]) : p.stream_base + g * p.stream_stride;
    uint32_t bad = 0;

    Spline sp;
    sp.g = p.coef;
    sp.n = p.n_knots;
    sp.cap = (int)cap;
    sp.whole_pair = whole;
    sp.lds = nullptr;
    sp.w0 = sp.w0b = sp.wlen = 0;
    sp.path = kPathGlobal;
    __syncthreads(); // the region's previous user is done
    if (!general)
        stage_window_ends<0>(sp, s_win, frame_knots(fr, fr.base_knot + (int)floorf(fr.tmin) + kd, fr.base_knot + (int)floorf(fr.tmax) + kd + 1, kd, kd), 64);
    __syncthreads();

    // ---- stage A: unit rows and norms -> the tile, as the tile kernel's four waves compute them ----
    uint32_t n2min = 0x7f000000u; // the smallest |P|^2 of the frame (hypothesis(): smin2)
    uint32_t near = 0;            // the near-static watch of virtual wave 0 (lmeds.hpp: the frame's first 64 rows)
    if (sp.path == kPathInterior) {
        for (uint32_t vw = 0; vw < 4u; ++vw) { // virtual wave vw of lmeds_kernel: rows 256 j + 64 vw + lane
            RowWatch watch;
            for (uint32_t row = vw * 64u + lane; row < N; row += kBlock) {
                float nrm;
                (void)lmeds_row<kPathInterior, false, 0, true>(sp, ra[row], rb[row], N, row, base, fd, tile, nrm, &watch, row < 64u);
                g_nrm[row] = nrm;
            }
            if (vw == 0u) near = watch.near;
            uint32_t vbad = finite_f(watch.nsum) ? 0u : (uint32_t)RSHIP_BAD_P;
            if (__builtin_amdgcn_ballot_w64(watch.qerr >= kNewtonMaxErr || watch.below_safe_normalize()) != 0) { // (lmeds_rows: per wave)
                vbad = 0;
                for (uint32_t row = vw * 64u + lane; row < N; row += kBlock) {
                    float nrm;
                    vbad |= lmeds_row<kPathInterior, false, 0, false>(sp, ra[row], rb[row], N, row, base, fd, tile, nrm);
                    g_nrm[row] = nrm;
                }
            }
            bad |= vbad;
            n2min = min(n2min, watch.n2min);
        }
    } else {
        RowWatch watch;
        for (uint32_t row = lane; row < N; row += 64u) {
            float nrm;
            bad |= lmeds_row<kPathGlobal, false, 0>(sp, ra[row], rb[row], N, row, base, fd, tile, nrm, &watch, row < 64u);
            g_nrm[row] = nrm;
        }
        n2min = watch.n2min;
        near = watch.near;
    }
    // a near-static frame: the rows once more from the fp64 streams, as the launch chain's search kernel of the frame's class
    // takes them in place (lmeds.hpp MODE 1, lmeds_big.hpp): the same decision from the same 64 rows, the same rows
    const bool use64 = RSSYNC_NEAR_WATCH && p.src64.coef && near_static_fires(near, N);
    int base64 = 0;
    double fd64 = 0.0;
    if (use64) {
        base64 = fr.base_knot + ld_m<SC1>(&p.kd64[g]);
        fd64 = ld_m<SC1>(&p.fd64[g]);
        bad = 0;
        n2min = 0x7f000000u;
        for (uint32_t row = lane; row < N; row += 64u) {
            const Row64 r = row64_unit(p.src64, (size_t)fr.off + row, base64, fd64);
            if (!r.finite) bad = RSHIP_BAD_P;
            tile.nx[row] = r.n.x; tile.ny[row] = r.n.y; tile.nz[row] = r.n.z;
            g_nrm[row] = r.nrm;
            n2min = min(n2min, __float_as_uint(r.n2));
        }
        if (lane == 0) atomicAdd(p.redo_count + 1, 1ull);
    }
    // hypothesis(): as the kernel of the frame's class has it -- the tile kernel's bound and recomputed norms (lmeds.hpp),
    // or, for a frame of more than 8192 tracks, the stored norms directly (lmeds_big.hpp)
    const float smin2 = general ? 0.f : smin2_of(wave_min_u32(n2min));
    auto row_scale = [&](uint32_t row) -> float {
        if (general || use64) return g_nrm[row]; // (the fp64 form: the norms of stage A; the tile kernel recomputes the same value)
        return row_scale_general(p.coef, p.n_knots, ra[row], rb[row], base, fd);
    };
    __syncthreads(); // (one wave: the tile's stores are ordered before the loads below)

    // ---- stage C: hypotheses in order; (T, bH) = best (quartile, index) so far, strict < (core_private.cpp:53) ----
    // The spline window is no longer needed: where the unit rows fit the wave's LDS region (12 bytes per row: ~1360 rows in
    // 16 KB) they are copied there and the 200 sweeps read LDS; larger frames sweep the tile in global memory (L1 / L2).
    // A hypothesis is first only COUNTED against the bound (no key is stored: nine of ten are turned away by that); one
    // that beats it writes its keys and finds its quartile by bisection with counting passes, as lmeds_big_kernel.
    uint32_t T = kInfBits;
    int bH = -1;
    f3 Mv = f3{0, 0, 0};
    auto search = [&](const auto& t, auto keys) { // (the tile's and the keys' address spaces are part of the types: ds_read / global_load, not flat)
        for (uint32_t batch = 0; batch < p.n_hyp; batch += 64u) {
            const uint32_t nb = p.n_hyp - batch < 64u ? p.n_hyp - batch : 64u;
            // the directions of up to 64 hypotheses at once, one per lane (the sampler's hash is ~100 instructions), then
            // taken in order with v_readlane -- as lmeds_small_body does
            f3 v = f3{0, 0, 0};
            if ((uint32_t)lane < nb) v = hypothesis(t, p.seed, fr.id, stream, batch + lane, N, smin2, row_scale);
            for (uint32_t jh = 0; jh < nb; ++jh) {
                const f3 hv = f3{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), jh)),
                                 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), jh)),
                                 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.z), jh))};
                uint32_t cnt = 0;
                for (uint32_t row = lane; row < N; row += 64u) {
                    const float r = fmaf(t.nz[row], hv.z, fmaf(t.ny[row], hv.y, t.nx[row] * hv.x)); // :48, as sweep_tile
                    cnt += (__float_as_uint(r) & 0x7fffffffu) < T ? 1u : 0u; // (a NaN's pattern lies above kInfBits >= T: never counts)
                }
                const uint32_t tot = wave_sum_u32(cnt);
                if (tot > kq) { // quartile_h < T: find it.  Bracket [lo, hi): count(< lo) <= kq < count(< hi)
                    for (uint32_t row = lane; row < N; row += 64u) {
                        const float r = fmaf(t.nz[row], hv.z, fmaf(t.ny[row], hv.y, t.nx[row] * hv.x));
                        const uint32_t a = __float_as_uint(r) & 0x7fffffffu;
                        keys[row] = a > kInfBits ? 0xffffffffu : a;
                    }
                    auto count_lt = [&](uint32_t B) -> uint32_t {
                        uint32_t c2 = 0;
                        for (uint32_t row = lane; row < N; row += 64u) c2 += keys[row] < B ? 1u : 0u;
                        return wave_sum_u32(c2);
                    };
                    uint32_t lo = 0u, hi = T;
                    while (hi - lo > 1u) {
                        const uint32_t mid = lo + ((hi - lo) >> 1);
                        if (count_lt(mid) > kq) hi = mid; else lo = mid;
                    }
                    T = lo;
                    bH = (int)(batch + jh);
                    Mv = hv;
                }
            }
        }
    };
    typedef __attribute__((address_space(3))) float* lds_f;
    typedef __attribute__((address_space(3))) uint32_t* lds_u;
    typedef __attribute__((address_space(1))) float* glb_f;
    typedef __attribute__((address_space(1))) uint32_t* glb_u;
    const TileP<glb_f> gtile{(glb_f)tile.nx, (glb_f)tile.ny, (glb_f)tile.nz};
    if ((size_t)N * 12u <= region_bytes) {
        lds_f l = (lds_f)reinterpret_cast<float*>(s_win);
        for (uint32_t row = lane; row < N; row += 64u) { // planes nx | ny | nz, N apart
            l[row] = gtile.nx[row];
            l[N + row] = gtile.ny[row];
            l[2 * N + row] = gtile.nz[row];
        }
        __syncthreads();
        const TileP<lds_f> ltile{l, l + N, l + 2 * N};
        // (the keys of a hypothesis that beats the bound beside them if they fit as well: up to 31 counting passes read them)
        if ((size_t)N * 16u <= region_bytes) search(ltile, (lds_u)(l + 3 * N));
        else search(ltile, (glb_u)g_key);
    } else {
        search(gtile, (glb_u)g_key);
    }
    if (!(finite_f(Mv.x) && finite_f(Mv.y) && finite_f(Mv.z))) bad |= RSHIP_BAD_M;
    if (lane == 0) st_m<SC1>(&p.best_h[sf], (int32_t)bH);
    if (bad) atomicOr(p.flags, bad);
}

} // namespace
