"""The host arithmetic that sizes the kernels' LDS spline windows for a problem's gyro rate (rs-sync_amd/csrc/
window_plan.hpp; DESIGN.md section 3 "Gyro rate").  It is plain C++ without HIP, so it is compiled here with g++ behind a
three-function C shim and checked on the CPU: the benchmark's shape keeps the compiled-in window, every plan holds the
widest frame plus its chunk, no plan exceeds its share of the CU's LDS, and the rules for small frames.
Reference for what the window must hold: core_private.cpp:19-20 (x = (ts - start + delay) * sample_rate)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = r'''
#include "window_plan.hpp"
extern "C" {
void plan(double span, double ends, double step, unsigned want, int small_, int wg_max, unsigned fixed, int lds, int legacy, unsigned* out) {
    const rs::WinPlan w = rs::plan_window(span, ends, step, want, small_ != 0, wg_max, fixed, lds, legacy != 0);
    out[0] = w.cap; out[1] = w.chunk;
}
unsigned smaller(double span, double step, unsigned chunk, unsigned static_lds, unsigned dyn_fixed, int wg_cap, int lds) {
    return rs::smaller_window_for_occupancy(span, step, chunk, static_lds, dyn_fixed, wg_cap, lds);
}
unsigned cap64_for(float span, float ends) { return rs::cap64_for(span, ends); }
unsigned cap64_used(unsigned cap64, int one_wave) { return rs::cap64_used(cap64, one_wave != 0); }
// the one-wave class's fp64 window for ONE frame of n tracks (out[0] knots, out[1] compact) and the executor's region for it
void one_wave_window(unsigned n, float span, float ends, int may_compact, unsigned* out) {
    const rs::FrameDims f{n, span, ends};
    bool comp = false;
    out[0] = rs::cap64_frames(&f, 1, 0u, 512u, true, may_compact ? &comp : nullptr);
    out[1] = comp ? 1u : 0u;
}
unsigned long exec_region(unsigned cap64, int compact, unsigned search_cap) { return (unsigned long)rs::exec_region_for(cap64, compact != 0, search_cap); }
unsigned exec_stage() { return rs::kPlanExecStage; }
int class_of(unsigned n, unsigned one_wave_max) { return rs::plan_class_of(n, one_wave_max); }
int class_shape(int k, unsigned max_n) { return rs::plan_class_shape(k, max_n); }
int sub_shape(int k, unsigned max_n) { return rs::plan_sub_shape(k, max_n); }
int small_rows(unsigned max_n, int sub) { return rs::plan_small_rows(max_n, sub != 0); }
}
'''
LDS = 160 * 1024
TILE8 = 26432      # static LDS of lmeds_kernel<8, 0, 0, true> (2048 tracks), of the one-wave kernel for 130 tracks
SMALL3 = 2600


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("plan")
    src = d / "shim.cpp"
    src.write_text(SHIM)
    out = d / "libplan.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "rs-sync_amd", "csrc"), "-o", str(out), str(src)])
    L = ctypes.CDLL(str(out))
    L.plan.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_int,
                       ctypes.POINTER(ctypes.c_uint)]
    L.smaller.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    L.smaller.restype = ctypes.c_uint
    L.cap64_for.argtypes = [ctypes.c_float, ctypes.c_float]
    L.cap64_for.restype = ctypes.c_uint
    L.cap64_used.argtypes = [ctypes.c_uint, ctypes.c_int]
    L.cap64_used.restype = ctypes.c_uint
    L.one_wave_window.argtypes = [ctypes.c_uint, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]
    L.exec_region.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_uint]
    L.exec_region.restype = ctypes.c_ulong
    L.exec_stage.restype = ctypes.c_uint
    return L


def plan(L, span, step, want=32, small=False, wg_max=5, fixed=TILE8, lds=LDS, legacy=False, ends=None):
    out = (ctypes.c_uint * 2)()
    L.plan(span, span if ends is None else ends, step, want, int(small), wg_max, fixed, lds, int(legacy), out)
    return int(out[0]), int(out[1])


def span_of(fs):      # a frame pair of the synthetic camera: 1/30 s + 11.1 ms of readout, + 2 knots
    return np.floor(0.0444 * fs) + 2


def test_the_benchmark_shape_keeps_the_compiled_in_window(lib):
    # 400 Hz, candidates 0.5 ms = 0.2 knots apart, chunks of 32: 19.8 + 6.2 + 1 knots of 80
    assert plan(lib, span_of(400), 0.2) == (0, 32)
    assert plan(lib, span_of(1000), 0.5) == (0, 32)
    # close to the edge the chunk shortens before the window grows (at least eight candidates)
    cap, chunk = plan(lib, 70.0, 0.8)
    assert cap == 0 and 8 <= chunk < 32 and 70 + (chunk - 1) * 0.8 + 1 <= 80


def fit(cap, span, step, want):
    if cap < span + 1:
        return 0
    return want if step <= 0 else int(min(want, np.floor((cap - span - 1) / step) + 1))


@pytest.mark.parametrize("fs", [1800, 2000, 3200, 4000, 6400, 8000, 12000, 20000])
@pytest.mark.parametrize("fixed,wg_max", [(TILE8, 5), (14144, 5), (50000, 2), (99000, 1)])
def test_dynamic_plans_hold_the_frame_and_fit_the_lds(lib, fs, fixed, wg_max):
    span, step = span_of(fs), 0.0005 * fs
    cap, chunk = plan(lib, span, step, fixed=fixed, wg_max=wg_max)
    # the rule, restated: the most workgroups per CU whose LDS share holds the widest frame and eight candidates, then
    # the longest chunk that share holds, then no more knots than that chunk needs
    want = None
    for wg in range(wg_max, 0, -1):
        share = LDS // wg - 1024
        if share <= fixed:
            continue
        cap_t = min(2048, (share - fixed) // 64 // 4 * 4)
        f = fit(cap_t, span, step, 32)
        if f >= 8:
            want = (wg, cap_t, f)
            break
    if want is None:
        assert (cap, chunk) == (0, 32)            # nothing holds it: the compiled-in window, general path
        return
    wg, cap_t, f = want
    assert chunk == f and 8 <= chunk <= 32
    assert span + 1 + (chunk - 1) * step <= cap <= cap_t          # the widest frame and the whole chunk, within the share
    assert cap <= span + 1 + (chunk - 1) * step + 8               # ... and no more LDS than that needs
    assert fixed + cap * 64 + 1024 <= LDS // wg


def test_small_frames_keep_the_general_path_beyond_128_knots(lib):
    assert plan(lib, span_of(2000), 1.0, want=32, small=True, wg_max=20, fixed=SMALL3)[0] in range(92, 129)
    cap, chunk = plan(lib, span_of(4000), 2.0, want=32, small=True, wg_max=20, fixed=SMALL3)
    assert (cap, chunk) == (0, 32)
    # GuessMotion's search: one candidate per workgroup
    cap, chunk = plan(lib, span_of(2000), 0.0, want=1, small=True, wg_max=20, fixed=SMALL3)
    assert chunk == 1 and span_of(2000) + 1 <= cap <= span_of(2000) + 9
    assert plan(lib, span_of(4000), 0.0, want=1, small=True, wg_max=20, fixed=SMALL3) == (0, 1)


def test_legacy_switch_never_grows_the_window(lib):
    for fs in (400, 2000, 8000):
        cap, chunk = plan(lib, span_of(fs), 0.0005 * fs, legacy=True)
        assert cap == 0
    assert plan(lib, 74.0, 0.8, legacy=True)[1] in range(4, 9)


def test_fp64_window_capacity(lib):
    c64 = lambda span, ends=None: lib.cap64_for(span, span if ends is None else ends)
    assert c64(20.0) == 80 and c64(79.0) == 80 and c64(80.0) == 96
    assert c64(180.0) == 192 and c64(358.0) == 368 and c64(5000.0) == 384
    # where the table knows the two ends' ranges the window holds those: a 4 kHz pair (180 knots) needs 2 x 46
    assert c64(180.0, 92.0) == 96 and c64(91.0, 48.0) == 80 and c64(358.0, 182.0) == 192
    # problems whose frames run in the one-wave kernels keep 80 knots beyond 144
    assert lib.cap64_used(96, 1) == 96 and lib.cap64_used(144, 1) == 144 and lib.cap64_used(160, 1) == 80
    assert lib.cap64_used(224, 1) == 80 and lib.cap64_used(224, 0) == 224 and lib.cap64_used(384, 0) == 384


def ends_of(fs):     # the two ends of such a pair: 11.1 ms of read-out each, + the carry knot and the partial knot per end
    return 2 * (np.floor(0.0111 * fs) + 3)


@pytest.mark.parametrize("fs", [2000, 3200, 4000, 8000])
def test_staging_the_two_ends_needs_fewer_knots(lib, fs):
    span, ends, step = span_of(fs), ends_of(fs), 0.0005 * fs
    cap1, chunk1 = plan(lib, span, step)
    cap2, chunk2 = plan(lib, span, step, ends=ends)
    assert cap2 and cap2 <= cap1 and chunk2 >= 8
    # whichever form the plan counted on, its capacity holds it: the pair and the chunk, or the two ends, each widened by the chunk
    cs = (chunk2 - 1) * step
    assert cap2 >= min(span + 1 + cs, ends + 2 + 2 * cs)
    wg1 = next(w for w in range(5, 0, -1) if TILE8 + cap1 * 64 + 1024 <= LDS // w)
    wg2 = next(w for w in range(5, 0, -1) if TILE8 + cap2 * 64 + 1024 <= LDS // w)
    assert wg2 >= wg1
    if fs == 2000:
        assert wg2 == 5 and wg1 == 4          # 2 kHz: five workgroups per CU again


def test_a_smaller_window_where_it_buys_a_workgroup_per_cu(lib):
    """round 5 (profiles/r5_k2_class3_ab.txt): frames of 2049 .. 4096 tracks -- 56 136 B of LDS with the compiled-in window,
    51 024 B + the window in dynamic LDS, registers for three workgroups per CU.  At 400 Hz the frames and their chunk touch
    28 knots: a 32-knot window (2 KB) lets the third workgroup in; from ~600 Hz on the window they need does not, and the
    compiled-in one stays; a kernel compiled for two workgroups, or one whose tile leaves no room either way, never switches."""
    S16, D16 = 56136, 51024          # lmeds_kernel<16, 0, 80> / <16, 0, 0>
    assert lib.smaller(span_of(400), 0.2, 32, S16, D16, 3, LDS) == 32
    assert lib.smaller(span_of(400), 0.2, 32, S16, D16, 2, LDS) == 0          # (never more workgroups than the registers allow)
    assert lib.smaller(span_of(800), 0.4, 32, S16, D16, 3, LDS) == 0          # 38 + 12.4 + 1 knots: 56 of dynamic window do not fit three times
    assert lib.smaller(span_of(2000), 1.0, 15, S16, D16, 3, LDS) == 0         # (wider than the compiled-in window anyway)
    # whatever it returns holds the frames and the chunk, and is smaller than the compiled-in window
    for fs in (100, 200, 400, 500, 600):
        span, step = span_of(fs), 0.0005 * fs
        cap = lib.smaller(span, step, 32, S16, D16, 3, LDS)
        if cap:
            assert span + 1 + 31 * step <= cap < 80 and (D16 + 64 * cap + 1024) * 3 <= LDS
    # the benchmark's kernel (31 560 B / 26 432 B, five workgroups by its registers) and the 8192-row one (105 288 / 100 176, one)
    assert lib.smaller(span_of(400), 0.2, 32, 31560, TILE8, 5, LDS) == 0
    assert lib.smaller(span_of(400), 0.2, 32, 105288, 100176, 1, LDS) == 0
    assert lib.smaller(span_of(400), 0.2, 32, 0, D16, 3, LDS) == 0            # (no footprint given: no opinion)


def test_the_executors_region_holds_the_staging_area_and_the_search_window(lib):
    """ADVICE r5 (high + medium): with round 5's COMPACT fp64 windows (64 bytes per knot) the window executor's per-wave LDS
    region -- which is in turn the search's fp32 window, the fp64 window and the staging area of a window's decisions
    (kernels/executor.hpp) -- had shrunk to 7168 / 9216 bytes for frames whose two ends span ~100-140 knots (4.3-6.5 kHz on
    130-track frames): below the 10 240 bytes exec_window_sums stages for a window of 90-128 frames in its trial phase, and
    below the 116-knot fp32 window the launch chain's search kernel plans for the same frames (107-109 knots of ends).
    The region is now the largest of the three, for every width (window_plan.hpp: exec_region_for)."""
    stage_bytes = lib.exec_stage() * 8
    assert stage_bytes == 10240
    seen_compact_below_stage = False
    for ends in range(90, 211):
        for span in (ends, 2 * ends - 6):                  # the two ends as the planner counts them; the whole pair
            out = (ctypes.c_uint * 2)()
            lib.one_wave_window(130, float(span), float(ends), 1, out)
            cap64, compact = int(out[0]), bool(out[1])
            # the search's fp32 window as plan_lmeds_window<1> asks for it: one candidate per workgroup, one-wave kernel
            cap32, chunk = plan(lib, float(span), 0.0, want=1, small=True, wg_max=20, fixed=SMALL3, ends=float(ends))
            region = lib.exec_region(cap64, int(compact), cap32)
            win64 = cap64 * (64 if compact else 128)
            seen_compact_below_stage |= compact and win64 < stage_bytes
            assert region >= win64 and region >= stage_bytes and region >= (cap32 or 80) * 64, (span, ends, cap64, compact, cap32, region)
            # eight one-wave workgroups per CU wherever the fp64 window alone allowed them (~3.6 KB of static LDS + 512 B)
            if win64 <= stage_bytes:
                assert LDS // (region + 3600 + 512) >= 8
    assert seen_compact_below_stage                          # (the sweep does cover the case the finding is about)
    # the cases the finding names: compact 112 / 144 knots; a 116-knot search window beside a 7168-byte fp64 window
    assert lib.exec_region(112, 1, 0) == 10240 and lib.exec_region(144, 1, 0) == 10240
    assert lib.exec_region(112, 1, 116) == 10240 and lib.exec_region(112, 1, 128) == 10240
    assert lib.exec_region(80, 0, 0) == 10240 and lib.exec_region(144, 0, 0) == 144 * 128 and lib.exec_region(208, 1, 0) == 208 * 64


def test_size_classes_and_the_shapes_of_presyncs_kernels(lib):
    """rssync_kernels.hip's class_of / class_rpt / lmeds_shape are these rules (window_plan.hpp): a frame's class follows from
    its own track count (core_private.cpp:73-86: every frame in its own lambda); the shape PreSync's launch takes for a class
    holds the largest frame of the class in the selection, is the smallest that does, never exceeds the class's own, and -- in
    the eight-wave family -- is a whole number of rows per thread of a 512-thread workgroup.  (The kernels index the tile by
    row without a bound of their own: a shape too small would be an out-of-bounds write in LDS.)"""
    for f in (lib.class_of, lib.class_shape, lib.sub_shape, lib.small_rows):
        f.restype = ctypes.c_int
    lib.class_of.argtypes = [ctypes.c_uint, ctypes.c_uint]
    lib.class_shape.argtypes = [ctypes.c_int, ctypes.c_uint]
    lib.sub_shape.argtypes = [ctypes.c_int, ctypes.c_uint]
    lib.small_rows.argtypes = [ctypes.c_uint, ctypes.c_int]
    tops = {0: 512, 1: 1024, 2: 2048, 3: 6144, 4: 8192}
    prev = 0
    for n in range(2, 9001):
        k = lib.class_of(n, 512)
        assert k >= prev and (k == 5) == (n > 8192) and (k == 5 or n <= tops[k]) and (k == 0 or n > tops[k - 1] if k < 5 else True), (n, k)
        prev = k
        if k == 0:
            r, r_sub = lib.small_rows(n, 0), lib.small_rows(n, 1)
            assert 64 * r >= n and 64 * r_sub >= n and r_sub <= r and r in (1, 2, 3, 4, 8)
            assert r_sub == max(1, -(-n // 64)) or (r_sub == 8 and n > 448), (n, r_sub)
            continue
        if k == 5:
            assert lib.class_shape(5, n) == 0 and lib.sub_shape(5, n) == 0
            continue
        own, sub = lib.class_shape(k, n), lib.sub_shape(k, n)
        assert own == {1: 4, 2: 8, 3: 16 if n <= 4096 else 24, 4: 32}[k]
        assert 256 * sub >= n and sub <= own and sub >= 3, (n, k, sub, own)
        step = 2 if k == 4 else 1
        assert sub == 3 or 256 * (sub - step) < n, (n, k, sub)            # the smallest that holds the frame
        assert k != 4 or sub % 2 == 0
        assert (sub <= 24) == (k <= 3)                                    # four waves up to 24 rows per thread, eight above
    # the one-wave family's boundary is a setting (RSSYNC_ONE_WAVE_MAX): class 0 follows it, the others do not move
    assert lib.class_of(300, 256) == 1 and lib.class_of(256, 256) == 0 and lib.class_of(1025, 256) == 2
