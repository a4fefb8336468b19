"""HIP path vs the oracle, through the C-ABI (include/rssync_c.h), on a real MI355X.

Tolerances (fp32 evaluation on the device, fp64 in the oracle) are written next to each
assert.  Where the reference algorithm itself is chaotic -- the LMedS arg-min at near-ties,
and the restated L-BFGS, whose long first steps jump between basins of the non-convex
per-frame loss -- parity is asserted on what is well-defined: identical inputs to each
stage, agreement fractions, and end results against ground truth (DESIGN.md "Parity").
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 123


@pytest.fixture(scope="module")
def hip_small(small_case):
    import rssync_amd
    from conftest import fill
    return fill(rssync_amd.SyncProblem(seed=SEED), small_case)


@pytest.fixture(scope="module")
def ora_small(small_case):
    from oracle.oracle import OracleProblem
    from conftest import fill
    return fill(OracleProblem(seed=SEED, threads=os.cpu_count() or 1, faithful=False), small_case)


def test_native_library_is_the_one_running(hip_small):
    import rssync_amd
    with open("/proc/self/maps") as f:
        assert any("librssync_core.so" in line for line in f)
    assert os.path.samefile(rssync_amd.library_path(),
                            os.path.join(os.path.dirname(rssync_amd.__file__), "librssync_core.so"))


@pytest.mark.parametrize("frame,delay", [(0, 0.0), (3, 0.0371), (63, -0.15), (20, 0.19999)])
def test_residual_matrix_and_its_delay_derivative(hip_small, ora_small, small_case, frame, delay):
    N = small_case["N"]
    Ph, dPh = hip_small.problem_matrix(frame, delay, N, deriv=True)
    Po = ora_small.problem_matrix(frame, delay)
    # P = ar x br of unit vectors: absolute error of a few fp32 ulps of 1
    assert np.abs(Ph - Po).max() < 5e-7
    eps = 1e-6
    dPo = (ora_small.problem_matrix(frame, delay + eps) - ora_small.problem_matrix(frame, delay - eps)) / (2 * eps)
    assert np.abs(dPh - dPo).max() < 2e-5 * max(1.0, np.abs(dPo).max())


def test_residual_matrix_outside_the_gyro_span(hip_small, ora_small, small_case):
    """the three extrapolation branches of minispline.cpp:49-54, including the x >= n quirk"""
    N = small_case["N"]
    fs, start, n = ora_small.gyro_info()
    span = n / fs
    for delay in (-5.0, -1.2, span - 2.1, span - 1.0, span + 3.0):
        Ph = hip_small.problem_matrix(5, delay, N)
        Po = ora_small.problem_matrix(5, delay)
        assert np.all(np.isfinite(Ph))
        assert np.abs(Ph - Po).max() < 2e-4, delay  # extrapolated quaternions are far from unit: looser


def _hip_with_kernel(small_case, kernel, monkeypatch):
    """a fresh problem on the GPU whose <= 512-track frames run PreSync / GuessMotion in the one-wave kernel (the
    default) or in the four-wave tile kernel's one-row-per-thread instantiation, lmeds_kernel<1, .> (VERDICT r2,
    next #7: compared with the oracle directly, not only with the one-wave kernel).  The switch is read when the
    device context is created."""
    import rssync_amd
    from conftest import fill
    if kernel == "tile":
        monkeypatch.setenv("RSSYNC_NO_SMALL_LMEDS", "1")
    else:
        monkeypatch.delenv("RSSYNC_NO_SMALL_LMEDS", raising=False)
    p = rssync_amd.SyncProblem(seed=SEED)
    monkeypatch.delenv("RSSYNC_NO_SMALL_LMEDS", raising=False)
    return fill(p, small_case)


KERNELS = ["one-wave", "tile"]


@pytest.mark.parametrize("kernel", KERNELS)
def test_presync_curve_per_frame(ora_small, small_case, kernel, monkeypatch):
    F = small_case["F"]
    hip_small = _hip_with_kernel(small_case, kernel, monkeypatch)
    do, co, fco, bho = ora_small.presync_curve(0.0, 0, F, 0.002, 0.2, per_frame=F)
    dh, ch, fch, bhh = hip_small.presync_curve(0.0, 0, F, 0.002, 0.2, per_frame=F)
    np.testing.assert_array_equal(do, dh)  # candidate delays: bit-exact (core_private.cpp:69-70)
    same = bho == bhh
    # the winning hypothesis is an arg-min over 20 quantiles; fp32/fp64 may flip it at near-ties
    assert same.mean() > 0.995
    rel = np.abs(fch - fco) / np.abs(fco)
    assert rel[same].max() < 1e-3 and np.median(rel[same]) < 2e-6
    assert np.abs(ch - co).max() / co.mean() < 2e-3
    assert np.argmin(ch) == np.argmin(co)
    c1, d1 = hip_small.PreSync(0.0, 0, F, 0.002, 0.2)
    c2, d2 = ora_small.PreSync(0.0, 0, F, 0.002, 0.2)
    assert d1 == d2 and abs(c1 - c2) < 1e-3 * c2


def test_lmeds_selection_is_exact(hip_small, small_case):
    """The device's winning hypothesis must be the arg-min of the EXACT lower-quartile of its own
    fp32 residuals: recompute them from the device's P in numpy and compare indices."""
    from oracle import oracle as ora
    F, N = small_case["F"], small_case["N"]
    dh, ch, fch, bhh = hip_small.presync_curve(0.0, 0, F, 0.02, 0.1, per_frame=F)
    mismatches = 0
    for ci in (0, 4, 9):
        for fr in (0, 13, 40):
            P = hip_small.problem_matrix(fr, dh[ci], N).astype(np.float32)
            nrm = np.linalg.norm(P.astype(np.float64), axis=1)
            meds = []
            for h in range(20):
                i0, i1 = ora.sample_pair(SEED, fr, ci, h, N)
                v = np.cross(P[i0].astype(np.float64), P[i1].astype(np.float64))
                v /= np.linalg.norm(v)
                r2 = np.sort(((P.astype(np.float64) @ v) / nrm) ** 2)
                meds.append(r2[N // 4])
            order = np.argsort(meds)
            gap = (meds[order[1]] - meds[order[0]]) / meds[order[0]]
            if int(order[0]) != int(bhh[ci, fr]):
                assert gap < 1e-4  # only a genuine fp32 near-tie may differ
                mismatches += 1
    assert mismatches <= 1


@pytest.mark.parametrize("kernel", KERNELS)
def test_init_motion_matches_oracle(ora_small, small_case, kernel, monkeypatch):
    from oracle import oracle as ora
    F = small_case["F"]
    d0 = 0.036
    hip = _hip_with_kernel(small_case, kernel, monkeypatch)  # fresh: sampler stream = SYNC_INIT + 0
    Mh, kh = hip.init_motion(d0, 0, F - 1)
    agree = 0
    for f in range(F):
        Mo, bh, med = ora_small.guess_motion(f, d0, 200, ora.STREAM_SYNC_INIT + 0)
        # same winning pair of rows -> same direction up to the fp32 error of the rows themselves
        # (P = ar x br cancels to ~1e-2, so ~1e-5 relative per row)
        # the search over the 200 hypotheses runs in fp32 (it may flip at a near-tie of two quantiles);
        # the winner's direction and k are then recomputed in fp64 from the fp64 rows
        if np.abs(Mh[f] - Mo).max() < 1e-12:
            agree += 1
            P = ora_small.problem_matrix(f, d0)
            ko = np.clip(100 / np.linalg.norm(P @ Mo), 10, 1000)
            assert kh[f] == pytest.approx(ko, rel=1e-12)
    assert agree >= F - 2


def test_loss_and_analytic_gradient(hip_small, ora_small, small_case):
    """Sync's objective runs in fp64 on the device (fp64 rays, fp64 spline): the loss agrees with the
    CPU solver to 1e-12, the analytic d/d-delay with the oracle's analytic one to 1e-10 and with the
    reference's +-1e-6 s central difference (core_private.cpp:96-97,112) to that difference's own error"""
    F = small_case["F"]
    d0 = 0.036
    Mh, kh = hip_small.init_motion(d0, 0, F - 1)
    delays = [d0, d0 + 1e-3, 0.0, -0.17]
    Lh, Gh = hip_small.loss(delays, grad=True)
    for j, dd in enumerate(delays):
        L = Gn = Ga = 0.0
        for f in range(F):
            l, dn, da, _ = ora_small.loss(f, dd, Mh[f], kh[f])
            L += l
            Gn += dn
            Ga += da
        assert Lh[j] == pytest.approx(L, rel=1e-12)
        assert Gh[j] == pytest.approx(Ga, rel=1e-10, abs=1e-10 * abs(Lh[j]))
        assert Gh[j] == pytest.approx(Gn, rel=2e-6, abs=2e-6 * abs(Lh[j]))
    # loss-only launches give the same numbers
    np.testing.assert_allclose(hip_small.loss(delays), Lh, rtol=1e-14)


def test_fp64_residual_matrix_matches_the_cpu_solver(hip_small, ora_small, small_case):
    """the rows the Sync kernels work on (fp64 streams packed on the device from the raw records)"""
    N = small_case["N"]
    fs, start, n = ora_small.gyro_info()
    for frame, delay in [(0, 0.0), (3, 0.0371), (63, -0.15), (5, -5.0), (5, n / fs + 3.0)]:
        P, dP = hip_small.problem_matrix64(frame, delay, N, deriv=True)
        Po = ora_small.problem_matrix(frame, delay)
        assert np.abs(P - Po).max() < 1e-13 * max(1.0, np.abs(Po).max())
        eps = 1e-6
        dPo = (ora_small.problem_matrix(frame, delay + eps) - ora_small.problem_matrix(frame, delay - eps)) / (2 * eps)
        assert np.abs(dP - dPo).max() < 1e-6 * max(1.0, np.abs(dPo).max())


def test_motion_optimiser_against_oracle_from_identical_starts(ora_small, small_case):
    import rssync_amd
    from conftest import fill
    F = small_case["F"]
    d0 = 0.036
    hip_small = fill(rssync_amd.SyncProblem(seed=SEED), small_case)  # fresh: deterministic sampler stream
    Mh, kh = hip_small.init_motion(d0, 0, F - 1)
    L0 = hip_small.loss([d0])[0]
    M2, k2, its, evs = hip_small.opt_motion(d0)
    L1 = hip_small.loss([d0])[0]
    assert L1 < L0 and 1 <= its / F <= 200 and evs >= its
    np.testing.assert_array_equal(k2, kh)  # var_k is not optimised (core_private.cpp:294)
    same = 0
    Lo_sum = 0.0
    for f in range(F):
        Mo, it, ev, fl = ora_small.lbfgs_motion(f, d0, Mh[f], kh[f])
        lo = ora_small.loss(f, d0, Mo, kh[f])[0]
        Lo_sum += lo
        lh = ora_small.loss(f, d0, M2[f], k2[f])[0]
        # the loss does not depend on |M| (core_private.cpp:120): compare directions and losses
        if np.abs(M2[f] / np.linalg.norm(M2[f]) - Mo / np.linalg.norm(Mo)).max() < 1e-6 and abs(lh - lo) <= 1e-9 * lo:
            same += 1
    # fp64 rows, fp64 objective: the per-frame trajectories follow the CPU solver's; what is left is
    # the order of the sums (wave reductions vs a sequential loop), which the optimiser can amplify on
    # a frame whose line search sits at a branch
    assert same >= 0.95 * F
    assert L1 == pytest.approx(Lo_sum, rel=1e-6)


def test_sync_on_clean_data_recovers_truth_and_oracle(clean_case):
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    from conftest import fill
    F = clean_case["F"]
    h = fill(rssync_amd.SyncProblem(seed=SEED), clean_case)
    o = fill(OracleProblem(seed=SEED, threads=os.cpu_count() or 1, faithful=False), clean_case)
    ch, dh = h.PreSync(0.0, 0, F, 0.002, 0.1)
    co, do = o.PreSync(0.0, 0, F, 0.002, 0.1)
    assert dh == do
    c1, d1 = h.Sync(dh, 0, F - 1, 0.0, 0.1)
    c2, d2 = o.Sync(do, 0, F - 1, 0.0, 0.1)
    assert abs(d1 - synth.D_TRUE) < 1e-4   # north star: 1e-4 s
    assert abs(d1 - d2) < 1e-4
    tr = h.sync_trace()
    assert tr.shape[1] == 6 and 6 <= len(tr) <= 400


def test_sync_on_noisy_data_differs_from_the_cpu_solver_by_reassociation_only():
    """Noise 1e-3 rad + 10 % outliers (BASELINE config 1).  The device's Sync is BIT-IDENTICAL to the CPU stand-in
    that sums in the device's order (tests/test_gpu_bitexact.py), and that stand-in differs from the reference-order
    oracle only by rounding (tests/test_reassociation.py: the per-frame L-BFGS works on a loss that does not depend on
    |M| (core_private.cpp:120), so rounding moves its iterates along that direction and some frames end in another
    basin).  The tolerance is THIS scene's (tests/noisy_scenes.py, profiles/r5_reassociation.json: measured 3.2e-5 s,
    so the north-star 1e-4 s is asserted), and the device must land where the stand-in landed."""
    import noisy_scenes as ns
    scene = ns.config1_noisy()
    (r,) = ns.run_scene(scene, scene.device(), scene.oracle())
    m = ns.measured(scene.name)
    assert ns.bound_s(scene.name) == ns.NORTH_STAR_S
    assert abs(r["d_dev"] - r["d_ora"]) < ns.bound_s(scene.name), (r["d_dev"], r["d_ora"])
    assert r["d_dev"] == pytest.approx(m["delays_s"]["device_order"][0], rel=0, abs=1e-9)   # = the CPU stand-in's result
    assert r["c_dev"] == pytest.approx(r["c_ora"], rel=2.5 * m["cost_rel"])                  # (the cost follows the frames that changed basin)
    n = min(len(r["trace_dev"]), len(r["trace_ora"]))
    assert np.abs(r["trace_dev"][:n, 0] - r["trace_ora"][:n, 0]).max() <= 2.5 * m["delay_after_each_outer_iteration_max_abs_s"]


def test_debug_presync_and_frame_ranges(hip_small, ora_small):
    dh, ch = hip_small.DebugPreSync(0.01, 0, 8, 0.05, 11)
    do, co = ora_small.DebugPreSync(0.01, 0, 8, 0.05, 11)
    np.testing.assert_array_equal(dh, do)
    np.testing.assert_allclose(ch, co, rtol=5e-3)
    # PreSync end-exclusive, Sync end-inclusive (core_private.cpp:66 vs :219)
    _, c2 = hip_small.presync_curve(0.03, 10, 12, 0.002, 0.004)
    _, c2o = ora_small.presync_curve(0.03, 10, 12, 0.002, 0.004)
    np.testing.assert_allclose(c2, c2o, rtol=1e-3)
    M, k = hip_small.init_motion(0.03, 10, 12)
    assert len(k) == 3
    # empty selection: costs are zero, first candidate wins (min_element over equal costs)
    c, d = hip_small.PreSync(0.0, 1000, 1010, 0.01, 0.05)
    assert c == 0.0 and d == pytest.approx(-0.05)


@pytest.mark.parametrize("kernel", KERNELS)
def test_ragged_and_tiny_frames(small_case, kernel, monkeypatch):
    """frames with different track counts, down to the minimum of 2, and re-setting a frame"""
    import rssync_amd
    from oracle.oracle import OracleProblem
    g = small_case["gyro"]
    counts = [2, 3, 5, 63, 64, 65, 127, 255, 256, 200, 17]
    if kernel == "tile":
        monkeypatch.setenv("RSSYNC_NO_SMALL_LMEDS", "1")
    h = rssync_amd.SyncProblem(seed=SEED)
    monkeypatch.delenv("RSSYNC_NO_SMALL_LMEDS", raising=False)
    o = OracleProblem(seed=SEED, faithful=False)
    for p in (h, o):
        p.SetGyroQuaternions(g.quats, g.fs, g.t0)
        for i, n in enumerate(counts):
            fr, ta, tb, ra, rb = small_case["frames"][i]
            p.SetTrackResult(100 + 3 * i, ta[:n], tb[:n], ra[:n], rb[:n])  # sparse frame ids
        fr, ta, tb, ra, rb = small_case["frames"][20]
        p.SetTrackResult(100, ta[:9], tb[:9], ra[:9], rb[:9])  # overwrite the first
    nf = len(counts)
    dh, ch, fch, bhh = h.presync_curve(0.03, 0, 1000, 0.004, 0.02, per_frame=nf)
    do, co, fco, bho = o.presync_curve(0.03, 0, 1000, 0.004, 0.02, per_frame=nf)
    assert np.all(np.isfinite(fch)) and np.all(fch > 0)
    # with small N the lower quartile is (N <= 8) or sits next to (N ~ 16) the two rows that define
    # the hypothesis, whose residuals are pure rounding noise (1e-16 in fp64, 1e-8 in fp32): the
    # arg-min is not comparable there
    big = np.array([n >= 32 for n in [9] + counts[1:]])
    same = (bhh == bho)[:, big]
    assert same.mean() > 0.95, (bhh[:, big], bho[:, big])
    np.testing.assert_allclose(fch[:, big][same], fco[:, big][same], rtol=2e-3)
    with pytest.raises(rssync_amd.RsSyncError, match="fewer than 2 tracks"):
        fr, ta, tb, ra, rb = small_case["frames"][0]
        h.SetTrackResult(7, ta[:1], tb[:1], ra[:1], rb[:1])
        h.PreSync(0.0, 0, 1000, 0.01, 0.02)


def test_panics_are_reported(small_case):
    import rssync_amd
    h = rssync_amd.SyncProblem()
    fr, ta, tb, ra, rb = small_case["frames"][0]
    bad = ra.copy()
    bad[0, 0] = np.inf
    with pytest.raises(rssync_amd.RsSyncError, match="set-track-result: non-finite numbers in rays_a"):
        h.SetTrackResult(0, ta, tb, bad, rb)
    h.SetTrackResult(0, ta, tb, ra, rb)
    with pytest.raises(rssync_amd.RsSyncError, match="gyro data was not set"):
        h.PreSync(0.0, 0, 10, 0.01, 0.05)


def test_gyro_can_be_replaced_while_tracks_stay(small_case, hip_small, ora_small):
    """the 48-orientation sweep re-calls SetGyroQuaternions on one object (core_testcode.cpp:216-224)"""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    from conftest import fill
    F = small_case["F"]
    g2 = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=99)
    h = fill(rssync_amd.SyncProblem(seed=SEED), small_case)
    o = fill(OracleProblem(seed=SEED, faithful=False, threads=os.cpu_count() or 1), small_case)
    c_true = h.PreSync(0.0, 0, F, 0.004, 0.1)
    for p in (h, o):
        p.SetGyroQuaternions(g2.quats, g2.fs, g2.t0)  # a different (wrong) gyro track
    c_wrong = h.PreSync(0.0, 0, F, 0.004, 0.1)
    c_wrong_o = o.PreSync(0.0, 0, F, 0.004, 0.1)
    assert c_wrong[0] > c_true[0]                      # the true track explains the data better
    assert c_wrong[0] == pytest.approx(c_wrong_o[0], rel=2e-3)


def test_timestamped_gyro_overload():
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    F, N = 24, 128
    g = synth.make_gyro(1.0, 1.0 + (F + 2) / synth.FPS, seed=21)  # starts at t = 0 (unsigned arithmetic)
    ts_us, q = synth.make_timestamped(g, jitter=0.2, seed=4)
    h, o = rssync_amd.SyncProblem(seed=SEED), OracleProblem(seed=SEED, faithful=False)
    for p in (h, o):
        p.SetGyroQuaternionsTimestamped(ts_us, q)
        for fr, ta, tb, ra, rb in synth.make_frames(g, 30, 30 + F, N, seed=5):
            p.SetTrackResult(fr, ta, tb, ra, rb)
    assert h.gyro_info() == o.gyro_info()
    np.testing.assert_allclose(h.gyro_knots(), o.gyro_knots(), rtol=0, atol=1e-15)  # slerp on the device: its acos and sin
    ch, dh = h.PreSync(0.0, 30, 30 + F, 0.002, 0.1)
    co, do = o.PreSync(0.0, 30, 30 + F, 0.002, 0.1)
    assert dh == do and ch == pytest.approx(co, rel=2e-3)
    assert abs(dh - synth.D_TRUE) <= 0.0015


def test_full_size_properties():
    """BASELINE size 4096 x 2048 is too big for the oracle in a test; check size-independent
    properties on a 512 x 2048 slice: linearity of the cost sum over disjoint frame ranges and
    determinism of repeated sweeps."""
    import rssync_amd
    from rssync_amd import synth
    F, N = 512, 2048
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=31)
    h = rssync_amd.SyncProblem(seed=SEED, max_outer_iters=5)
    synth.fill(h, g, 0, F, N, seed=31)
    d, c_all = h.presync_curve(0.0, 0, F, 0.01, 0.1)
    _, c_a = h.presync_curve(0.0, 0, 200, 0.01, 0.1)
    _, c_b = h.presync_curve(0.0, 200, F, 0.01, 0.1)
    np.testing.assert_allclose(c_a + c_b, c_all, rtol=1e-12)   # the only coupling is a sum over frames
    _, c_again = h.presync_curve(0.0, 0, F, 0.01, 0.1)
    np.testing.assert_array_equal(c_all, c_again)              # fixed-order reductions: bitwise repeatable
    assert abs(d[np.argmin(c_all)] - synth.D_TRUE) <= 0.005 + 1e-12
    c, dd = h.Sync(d[np.argmin(c_all)], 0, F - 1, 0.0, 0.2)
    assert abs(dd - synth.D_TRUE) < 2e-3 and np.isfinite(c)


def test_wave_selection_is_exact_on_adversarial_data():
    """the exact lower-quartile selection (secant search on the CDF + counting passes) against
    np.sort, bit for bit: ties, constant data, 40 decades of dynamic range, tiny n, with and
    without an upper bound (the 'is this hypothesis better than the best so far' test)"""
    import ctypes as C
    import rssync_amd
    lib = rssync_amd.load_library()
    lib.rship_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.rship_destroy.argtypes = [C.c_void_p]
    lib.rship_debug_select.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                       C.c_void_p]
    ctx = C.c_void_p()
    assert lib.rship_create(C.byref(ctx), -1) == 0
    rng = np.random.default_rng(0)

    def run(vals, kq, upper=None):
        out = np.zeros((vals.shape[0], 2), dtype=np.uint32)
        u = np.ascontiguousarray(upper, dtype=np.float32) if upper is not None else None
        assert lib.rship_debug_select(ctx, vals.ctypes.data, vals.shape[0], vals.shape[1], kq,
                                      u.ctypes.data if u is not None else None, out.ctypes.data) == 0
        return out

    inf32 = np.float32(np.inf)
    for n, kq in [(2048, 512), (256, 64), (100, 25), (3, 0), (5, 1), (2048, 0), (2048, 2047), (17, 4)]:
        for dist in ("sq", "exp", "ties", "const"):
            P = 200
            if dist == "sq":
                vals = (rng.normal(size=(P, n)) * 1e-3) ** 2
            elif dist == "exp":
                vals = np.exp(rng.uniform(-40, 5, size=(P, n)))
            elif dist == "ties":
                vals = rng.integers(0, 7, size=(P, n)) * 0.125
            else:
                vals = np.full((P, n), 0.25)
            vals = np.ascontiguousarray(vals, dtype=np.float32)
            srt = np.sort(vals, axis=1)
            want = srt[:, kq].view(np.uint32)
            np.testing.assert_array_equal(run(vals, kq)[:, 0], want)
            kth = srt[:, kq]
            for up in (np.nextafter(kth, inf32) * np.float32(1.5) + np.float32(1e-30), kth, np.nextafter(kth, inf32)):
                got = run(vals, kq, up)
                cnt = (vals < up[:, None]).sum(1)
                np.testing.assert_array_equal(got[:, 1], cnt)
                np.testing.assert_array_equal(got[:, 0], np.where(cnt > kq, want, 0xffffffff))
    lib.rship_destroy(ctx)


def test_full_size_sync_matches_oracle():
    """BASELINE config 3: 4096 frames x 2048 tracks, Sync capped at 20 outer iterations, noise and
    10 % outliers, identical inputs.  North-star tolerances: returned delay within 1e-4 s of the CPU
    solver; per-iteration loss within the stated fp32 tolerance, 2e-4 relative (measured 7e-5)."""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    F, N, seed = 4096, 2048, 0x5EED0003
    g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=seed)
    h = rssync_amd.SyncProblem(seed=seed, max_outer_iters=20)
    o = OracleProblem(seed=seed, max_outer_iters=20, threads=min(os.cpu_count() or 1, 16), faithful=False)
    for fr in synth.make_frames(g, 0, F, N, seed=seed):
        h.SetTrackResult(*fr)
        o.SetTrackResult(*fr)
    for p in (h, o):
        p.SetGyroQuaternions(g.quats, g.fs, g.t0)
    ch, dh = h.Sync(0.0365, 0, F - 1, 0.0, 0.2)
    trh = h.sync_trace()
    co, do, tro = o.sync_trace(0.0365, 0, F - 1, 0.0, 0.2)
    assert abs(dh - do) < 1e-4
    assert len(trh) == len(tro)
    np.testing.assert_allclose(trh[:, 0], tro[:, 0], atol=1e-4)   # delay after every outer iteration
    np.testing.assert_allclose(trh[:, 2], tro[:, 2], rtol=2e-4)   # loss at every outer iteration
    assert ch == pytest.approx(co, rel=2e-4)
    assert abs(dh - synth.D_TRUE) < 5e-4


def test_presync_sweep_at_full_track_count():
    """BASELINE config 2/3 sweep (radius 200 ms, step 0.5 ms = 800 candidates) at 2048 tracks per
    frame, on a 48-frame slice the oracle can finish in seconds: the 8-rows-per-thread kernel
    (chunks of 32 candidates, provisional bounds carried from candidate to candidate)."""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    F, N, seed = 48, 2048, 0x5EED0003
    g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=seed)
    h = rssync_amd.SyncProblem(seed=seed)
    o = OracleProblem(seed=seed, threads=min(os.cpu_count() or 1, 16), faithful=False)
    synth.fill(h, g, 0, F, N, seed=seed)
    synth.fill(o, g, 0, F, N, seed=seed)
    dh, ch, fch, bhh = h.presync_curve(0.0, 0, F, 0.0005, 0.2, per_frame=F)
    do, co, fco, bho = o.presync_curve(0.0, 0, F, 0.0005, 0.2, per_frame=F)
    assert len(dh) == 800
    np.testing.assert_array_equal(dh, do)
    same = bhh == bho
    assert same.mean() > 0.995
    rel = np.abs(fch - fco) / fco
    assert rel[same].max() < 1e-3 and np.median(rel[same]) < 2e-6
    assert np.argmin(ch) == np.argmin(co)
    np.testing.assert_allclose(ch, co, rtol=2e-3)


def test_orientation_sweep_ranks_the_true_orientation_first():
    """BASELINE config 5 in miniature: the reference's orientation-guessing loop
    (core_testcode.cpp:216-232) -- 48 signed axis permutations of the gyro rates, SetGyroQuaternions
    (timestamped overload) re-called on ONE problem object that keeps its tracks, PreSync each,
    sort by cost.  The true orientation must come out on top, and costs must match the oracle."""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    F, N = 24, 256
    g = synth.make_gyro(1.0, 1.0 + (F + 2) / synth.FPS, seed=77)   # t0 = 0: timestamps must be >= 0
    frames = list(synth.make_frames(g, 30, 30 + F, N, seed=77))
    ts_us = np.round(g.times * 1e6).astype(np.int64)
    h = rssync_amd.SyncProblem(seed=SEED)
    o = OracleProblem(seed=SEED, threads=min(os.cpu_count() or 1, 16), faithful=False)
    for p in (h, o):
        for fr in frames:
            p.SetTrackResult(*fr)
    results = []
    for k, name in enumerate(synth.ORIENTATIONS):
        q = synth.gyro_for_orientation(g, name)
        h.SetGyroQuaternionsTimestamped(ts_us, q)
        cost, delay = h.PreSync(0.0, 30, 30 + F, 0.004, 0.1)
        results.append((cost, delay, name))
        if k % 12 == 0:
            o.SetGyroQuaternionsTimestamped(ts_us, q)
            co, do = o.PreSync(0.0, 30, 30 + F, 0.004, 0.1)
            assert cost == pytest.approx(co, rel=5e-3)
    results.sort()
    assert results[0][2] == "XYZ"
    assert results[0][0] < 0.95 * results[1][0]
    assert abs(results[0][1] - synth.D_TRUE) <= 0.003


def test_cxx_driver_through_the_vtable(tmp_path, small_case):
    """examples/sync_driver.cpp: the reference driver's call pattern (PreSync, then four Sync calls
    per sync point, core_testcode.cpp:303-316) through the C++ ISyncProblem surface must give the
    same delays as the flat C-ABI does from Python."""
    import struct
    import subprocess
    import rssync_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "sync_driver"
    libdir = os.path.join(root, "rs-sync_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "sync_driver.cpp"), "-o", str(exe), "-L", libdir,
                           "-lrssync_core", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    g = small_case["gyro"]
    window, distance = 24, 16
    blob = struct.pack("<qdd", g.quats.shape[0], g.fs, g.t0) + np.ascontiguousarray(g.quats).tobytes()
    blob += struct.pack("<q", len(small_case["frames"]))
    for fr, ta, tb, ra, rb in small_case["frames"]:
        blob += struct.pack("<qq", fr, len(ta)) + ta.tobytes() + tb.tobytes() + np.ascontiguousarray(ra).tobytes() + \
            np.ascontiguousarray(rb).tobytes()
    blob += struct.pack("<qqddd", window, distance, 0.0, 2.0, 100.0)
    inp = tmp_path / "input.bin"
    inp.write_bytes(blob)
    env = dict(os.environ, RSSYNC_SEED=str(SEED), RSSYNC_QUIET="1")
    out = subprocess.run([str(exe), str(inp)], capture_output=True, text=True, env=env, cwd=tmp_path, check=True).stdout
    got = [tuple(map(float, line.split(","))) for line in out.strip().splitlines()]
    from conftest import fill
    h = fill(rssync_amd.SyncProblem(seed=SEED), small_case)
    want = []
    for pos in range(0, small_case["F"] - window, distance):
        d = h.PreSync(0.0, pos, pos + window, 0.002, 0.1)[1]
        for _ in range(4):
            d = h.Sync(d, pos, pos + window, 0.0, 0.1)[1]
        want.append((float(pos), 1000 * d))
    assert len(got) == len(want) >= 2
    np.testing.assert_allclose(np.array(got), np.array(want), rtol=0, atol=1e-6)
    # the same loop as one rssync_ext_sync_points call on the borrowed object: identical text
    out_b = subprocess.run([str(exe), str(inp), "batched"], capture_output=True, text=True, env=env, cwd=tmp_path,
                           check=True).stdout
    assert out_b == out
    # ONE object on several GPUs with no source change: RSSYNC_GPUS makes CreateSyncProblem() spread its frames
    # over the listed devices (here two contexts on the one GPU of the box); the text must not change
    out_m = subprocess.run([str(exe), str(inp)], capture_output=True, text=True, env=dict(env, RSSYNC_GPUS="0,0"),
                           cwd=tmp_path, check=True).stdout
    assert out_m == out


def test_batched_windows_equal_sequential_calls(small_case):
    """the driver's window loop (core_testcode.cpp:303-316) as one batched call.
    Window w of pre_sync_windows / sync_windows must be what PreSync / the w-th consecutive Sync
    on that window returns: the same kernels run on the same rows with the same sampler stream,
    so the comparison is exact (costs: same summation order)."""
    import rssync_amd
    from conftest import fill
    F = small_case["F"]
    wins = [(0, F // 2 - 1), (F // 4, 3 * F // 4 - 1), (F // 2, F - 1), (3, 9), (0, F - 1)]
    b = [w[0] for w in wins]
    e = [w[1] for w in wins]
    d0 = [0.036, 0.030, 0.040, 0.036, 0.025]
    seq = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=30), small_case)
    bat = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=30), small_case)
    pc, pd = bat.pre_sync_windows(0.03, b, [x + 1 for x in e], 0.002, 0.05)
    for w in range(len(wins)):
        c, d = seq.PreSync(0.03, b[w], e[w] + 1, 0.002, 0.05)
        assert pd[w] == d and pc[w] == pytest.approx(c, rel=1e-14, abs=0)
    ref, ref_tr = [], []
    for w in range(len(wins)):
        ref.append(seq.Sync(d0[w], b[w], e[w], 0.03, 0.05))
        ref_tr.append(seq.sync_trace())
    costs, delays = bat.sync_windows(d0, b, e, 0.03, 0.05)
    for w in range(len(wins)):
        assert delays[w] == ref[w][1] and costs[w] == ref[w][0]
        np.testing.assert_array_equal(bat.window_trace(w), ref_tr[w])


def test_sync_points_equal_the_driver_loop(small_case):
    """rssync_ext_sync_points == the loop at core_testcode.cpp:303-316 through PreSync/Sync."""
    import rssync_amd
    from conftest import fill
    pos, window, init = [0, 10, 20, 30, 40], 20, 0.0
    seq = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=40), small_case)
    want = []
    for p0 in pos:
        d = seq.PreSync(init, p0, p0 + window, 0.002, 0.1)[1]
        for _ in range(4):
            c, d = seq.Sync(d, p0, p0 + window, init, 0.1)
        want.append((c, d))
    bat = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=40), small_case)
    costs, delays = bat.sync_points(pos, window, init, 0.002, 0.1, repeats=4)
    for w in range(len(pos)):
        assert (costs[w], delays[w]) == want[w]


def test_rccl_reduce_hook_on_the_device(tmp_path):
    """bench.py's multi-GPU exchange (torch.distributed "nccl" = RCCL, device staging tensor) with
    one rank on this box: same results as without a hook; the exchanges are counted."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "rccl.json"
    subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_worker.py"), str(port), str(out)], check=True,
                   timeout=300)
    r = json.load(open(out))
    assert r["backend"] == "nccl"
    assert r["plain"] == r["rccl"] == r["native"] == r["native_host"]
    # With the library's communicator Sync's loop stays on the device (rship_sync_run: window sums -> ncclAllReduce on
    # the stream -> decisions); with one rank the all-reduce is the identity, so every trace row is the plain device
    # loop's and the host loop's, bit for bit -- also with several windows, one of them without frames.
    for k in ("_trace", "_windows"):
        assert r["plain" + k] == r["rccl" + k] == r["native" + k] == r["native_host" + k], k
    its = r["native"][4]
    # the final loss, and before it -- device loop: two all-reduces per ENQUEUED iteration (a block of eight, then blocks
    # of two); host loop: one per launch.  Nothing else: since round 5 the ranks agree on no kernel shape (a frame's
    # kernels follow its own track count), which rounds 2-4 paid one more exchange per call for.
    assert r["native_sync_exchanges"] == 2 * _enqueued(its) + 1
    assert 2 * its + 1 <= r["native_host_sync_exchanges"] <= 3 * its + 1
    # PreSync: the sweep; Sync: 2 per outer iteration + the final loss
    assert r["rccl_sync_exchanges"] == 2 * r["rccl"][4] + 1 and r["exchanges"] > r["rccl_sync_exchanges"]


@pytest.mark.parametrize("fs,N", [(50.0, 600), (100.0, 200), (400.0, 200), (2000.0, 200), (2000.0, 400), (4000.0, 200), (2000.0, 600),
                                  (4000.0, 2048), (8000.0, 600), (12000.0, 200), (12000.0, 600)])
def test_gyro_rates_against_the_oracle(fs, N):
    """The reference takes any sample rate (core_private.cpp:135-140; the timestamped overload rounds to 50 Hz without a
    ceiling, :146-149).  A frame pair spans 0.044 s x rate knots of the spline: up to ~1.7 kHz that fits the 80-knot
    window compiled into the kernels' LDS; above it the window moves to dynamic LDS sized for the problem (K2 / K2s: the
    interior path with a shorter candidate chunk; K1, K3, the executor: up to 384 knots = 8.6 kHz), and only beyond that
    do the kernels read the table from L2 (12 kHz here).  Same checks as at 400 Hz, for every kernel family: one wave per
    frame (N = 200, 400), the tile kernel (600, 2048).  50 and 100 Hz (phone IMUs; a pair spans 3-6 knots) for the other end."""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    F = 24 if N <= 600 else 10
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, fs=fs, seed=21)
    frames = list(synth.make_frames(g, 0, F, N, seed=21))                # noise 1e-3 rad, 10 % outliers
    h = rssync_amd.SyncProblem(seed=SEED)
    o = OracleProblem(seed=SEED, threads=min(os.cpu_count() or 1, 16), faithful=False)
    for p in (h, o):
        p.SetGyroQuaternions(g.quats, g.fs, g.t0)
        for fr in frames:
            p.SetTrackResult(*fr)
    Ph = h.problem_matrix(5, 0.0371, N)
    assert np.abs(Ph - o.problem_matrix(5, 0.0371)).max() < 5e-7
    assert np.abs(h.problem_matrix64(5, 0.0371, N) - o.problem_matrix(5, 0.0371)).max() < 1e-13
    dh_, ch_, fch, bhh = h.presync_curve(0.0, 0, F, 0.002, 0.1, per_frame=F)
    do_, co_, fco, bho = o.presync_curve(0.0, 0, F, 0.002, 0.1, per_frame=F)
    np.testing.assert_array_equal(dh_, do_)
    same = bhh == bho
    assert same.mean() > 0.97, same.mean()                   # the fp32 search flips a near-tie now and then
    rel = np.abs(fch - fco) / fco
    assert rel[same].max() < 1e-3 and np.median(rel[same]) < 2e-6
    assert np.argmin(ch_) == np.argmin(co_)
    np.testing.assert_allclose(ch_, co_, rtol=5e-3)
    ch, dh = h.PreSync(0.0, 0, F, 0.002, 0.1)
    co, do = o.PreSync(0.0, 0, F, 0.002, 0.1)
    assert dh == do and ch == pytest.approx(co, rel=5e-3)
    w = h.window_info()
    span, ends = w["frame_span_knots"], w["frame_ends_knots"]
    assert abs(span - (0.0444 * fs + 2)) <= 3
    # the two ends of a pair (11 ms of read-out each) cover about half the knots of the pair (44 ms): what the dynamic
    # windows stage
    assert ends <= span and (span <= 24 or ends <= 0.62 * span + 8)
    need = min(span, ends)
    small = N <= 512    # one wave per frame: wide windows only while enough waves still share a CU (window_plan.hpp)
    want64 = max(80, (need + 1 + 15) // 16 * 16)
    if span <= 70:
        assert not w["presync_window_dynamic"] and w["fp64_window_knots"] == 80 and w["trial_delays_per_pass"] == 5
    elif need + 1 <= 384 and not small:                       # the window grew instead of the kernels leaving the LDS path
        assert w["presync_window_dynamic"] and w["presync_window_knots"] >= need and w["fp64_window_knots"] == want64
    elif small:
        assert w["presync_window_dynamic"] == (need + 3 <= 128), w
        assert w["fp64_window_knots"] == (80 if want64 > 144 else want64), w
    else:
        assert w["fp64_window_knots"] == 384                  # wider than any window: the table from L2 (still correct)
    Mh, kh = h.init_motion(dh, 0, F - 1)
    Lh, Gh = h.loss([dh, 0.03, 0.035, 0.0371, 0.04, 0.02], grad=True)     # six delays: more than one pass of the trials' kernel
    for j, dd in enumerate((dh, 0.03, 0.035, 0.0371, 0.04, 0.02)):
        per = [o.loss(f, dd, Mh[f], kh[f]) for f in range(F)]
        assert Lh[j] == pytest.approx(sum(p[0] for p in per), rel=1e-11)
        assert Gh[j] == pytest.approx(sum(p[2] for p in per), rel=1e-9, abs=1e-9 * abs(Lh[j]))
    np.testing.assert_allclose(h.loss([dh, 0.03, 0.035, 0.0371, 0.04, 0.02]), Lh, rtol=1e-14)
    # Sync on the noise-free scene of the same gyro track (fresh problems: the Sync-side call counter starts at zero on
    # both sides): the oracle's delay and the truth within the north-star 1e-4 s
    clean = list(synth.make_frames(g, 0, F, N, seed=21, noise=0.0, outliers=0.0))
    h2 = rssync_amd.SyncProblem(seed=SEED)
    o2 = OracleProblem(seed=SEED, threads=min(os.cpu_count() or 1, 16), faithful=False)
    for p in (h2, o2):
        p.SetGyroQuaternions(g.quats, g.fs, g.t0)
        for fr in clean:
            p.SetTrackResult(*fr)
    c2h, d2h = h2.Sync(0.036, 0, F - 1, 0.0, 0.1)
    c2o, d2o = o2.Sync(0.036, 0, F - 1, 0.0, 0.1)
    assert abs(d2h - d2o) < 1e-4 and abs(d2h - synth.D_TRUE) < 1e-4


def test_create_use_destroy_does_not_leak_device_memory(small_case):
    """100 problems created, used (PreSync, Sync, batched windows, pixel frames) and destroyed:
    free HBM returns to where it was (a long-running host must not creep)."""
    import ctypes
    import gc
    import rssync_amd
    from rssync_amd import synth
    from conftest import fill
    F = 16
    case = dict(small_case, frames=small_case["frames"][:F])
    rssync_amd.load_library()                      # pulls in the HIP runtime the library uses
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        assert hip.hipDeviceSynchronize() == 0
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    def once():
        p = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=3), case)
        p.PreSync(0.0, 0, F, 0.01, 0.05)
        p.Sync(0.03, 0, F - 1, 0.0, 0.2)
        p.sync_points([0, 4], 8, 0.03, 0.01, 0.05, repeats=1)
        pa = np.array([[500.0, 400.0], [900.0, 700.0], [1500.0, 300.0]])
        p.set_track_pixels(99, 3.3, 3.3333, pa, pa + 1.0, synth.LENS, synth.IMAGE_ROWS)
        p.upload()
        del p
        gc.collect()
    once()
    free0 = free_bytes()
    for _ in range(100):
        once()
    free1 = free_bytes()
    assert free0 - free1 < 32 << 20, (free0, free1)


def test_native_rccl_exchange_single_rank(small_case):
    """rssync_ext_rccl_unique_id / _init: the library's own RCCL communicator (librccl opened at
    run time) with one rank: the all-reduce is the identity, so results equal a hook-free run, and
    the exchange really goes through ncclAllReduce (a second init is refused)."""
    import rssync_amd
    from conftest import fill
    F = 24
    case = dict(small_case, frames=small_case["frames"][:F])
    plain = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=10), case)
    nat = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=10), case)
    import torch  # noqa: F401  (torch ships its OWN libamdhip64 + librccl: two HIP runtimes may share this process)
    nat.rccl_preflight()                                  # library + entry points resolve, nothing is communicated
    lib = nat.rccl_library()
    # the librccl beside the HIP runtime librssync_core is bound to -- not "whichever is already mapped": torch's copy
    # refuses this library's stream (ncclCommInitRank -> 1, measured in round 4)
    assert "librccl" in lib and "beside the HIP runtime this library is bound to" in lib, lib
    hip_rt = lib[lib.index("bound to, ") + 10:].rstrip(")")
    assert os.path.dirname(lib.split(" ")[0]) == os.path.dirname(hip_rt)
    uid = nat.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    nat.rccl_init(uid, 0, 1)
    with pytest.raises(rssync_amd.RsSyncError, match="already initialised"):
        nat.rccl_init(uid, 0, 1)
    assert nat.PreSync(0.0, 0, F, 0.004, 0.1) == plain.PreSync(0.0, 0, F, 0.004, 0.1)
    assert nat.Sync(0.036, 0, F - 1, 0.0, 0.2) == plain.Sync(0.036, 0, F - 1, 0.0, 0.2)
    got = nat.sync_points([0, 8], 10, 0.03, 0.004, 0.05, repeats=2)[1].tolist()
    assert got == plain.sync_points([0, 8], 10, 0.03, 0.004, 0.05, repeats=2)[1].tolist()


def test_device_driven_sync_loop_equals_the_host_loop(small_case, clean_case):
    """Sync's outer loop with the decisions taken on the device between launches (kernels/syncloop.hpp) against
    the same loop on the host (sync_problem.cpp): identical bits -- delays, costs, every trace row -- for single
    calls, batched sync points (windows finishing at different iterations) and the simplified mode."""
    import rssync_amd
    from conftest import fill
    for case in (small_case, clean_case):
        F = case["F"]
        dev = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=40), case)
        host = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=40), case)
        host.set_host_loop(True)
        for start in (0.036, 0.02):
            assert dev.Sync(start, 0, F - 1, 0.0, 0.2) == host.Sync(start, 0, F - 1, 0.0, 0.2)
            np.testing.assert_array_equal(dev.sync_trace(), host.sync_trace())
        assert dev.Sync(0.036, 0, F - 1, 0.0, 0.0005) == host.Sync(0.036, 0, F - 1, 0.0, 0.0005)   # leaves the search window
        pos = [0, 10, 20, 30, 40]
        cd, dd = dev.sync_points(pos, 20, 0.0, 0.002, 0.1, repeats=3)
        ch, dh = host.sync_points(pos, 20, 0.0, 0.002, 0.1, repeats=3)
        np.testing.assert_array_equal(dd, dh)
        np.testing.assert_array_equal(cd, ch)
        for w in range(len(pos)):
            np.testing.assert_array_equal(dev.window_trace(w), host.window_trace(w))
        assert dev.SyncSimplified(0.03, 0, F - 1, 0.0, 0.2) == host.SyncSimplified(0.03, 0, F - 1, 0.0, 0.2)
        np.testing.assert_array_equal(dev.sync_trace(), host.sync_trace())
    capped = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=3), small_case)
    capped_h = fill(rssync_amd.SyncProblem(seed=SEED, max_outer_iters=3), small_case)
    capped_h.set_host_loop(True)
    assert capped.Sync(0.03, 0, 63, 0.0, 0.2) == capped_h.Sync(0.03, 0, 63, 0.0, 0.2)
    assert len(capped.sync_trace()) == 3


def test_window_groups_on_concurrent_streams_equal_the_host_loop():
    """With 16 or more windows the device-driven loop runs contiguous groups of windows as independent chains of
    launches, each on its own stream from its own host thread (rship_sync_run).  Windows do not see each other:
    every delay, cost and trace row must be the host loop's, bit for bit, whatever the number of groups."""
    import rssync_amd
    from rssync_amd import synth
    F, N, WINDOW = 400, 96, 24
    g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=41)
    pos = list(range(0, F - WINDOW - 1, 11))   # 35 overlapping windows -> 4 groups
    assert len(pos) >= 32
    old = os.environ.get("RSSYNC_LOOP_STREAMS")
    runs = {}
    try:
        for streams in ("host", "1", "3", "4", "4/first=1", "2/first=3"):
            p = rssync_amd.SyncProblem(seed=SEED, verbose=False, max_outer_iters=60)
            synth.fill(p, g, 0, F, N, seed=41)
            os.environ.pop("RSSYNC_LOOP_FIRST_TRIALS", None)
            if streams == "host":
                p.set_host_loop(True)
            else:
                # "first=k": only k trials of a line search in the first launch, so that the windows wait an
                # iteration for the others (the searches of this scene stop at the fourth or fifth trial)
                if "/first=" in streams:
                    os.environ["RSSYNC_LOOP_FIRST_TRIALS"] = streams.split("=")[1]
                os.environ["RSSYNC_LOOP_STREAMS"] = streams.split("/")[0]
            c, d = p.sync_points(pos, WINDOW, 0.0, 0.002, 0.1, repeats=2)
            runs[streams] = (np.array(c), np.array(d), [np.array(p.window_trace(w)) for w in range(len(pos))])
    finally:
        os.environ.pop("RSSYNC_LOOP_FIRST_TRIALS", None)
        if old is None:
            os.environ.pop("RSSYNC_LOOP_STREAMS", None)
        else:
            os.environ["RSSYNC_LOOP_STREAMS"] = old
    ref = runs["host"]
    assert len({len(t) for t in ref[2]}) > 1   # windows stop at different iterations
    for streams in ("1", "3", "4", "4/first=1", "2/first=3"):
        c, d, tr = runs[streams]
        np.testing.assert_array_equal(d, ref[1])
        np.testing.assert_array_equal(c, ref[0])
        for w in range(len(pos)):
            np.testing.assert_array_equal(tr[w], ref[2][w])


def test_two_ranks_with_different_frame_sizes_pick_the_same_kernels(tmp_path):
    """the kernel a frame runs in (one-wave or tile LMedS; the motion kernel's shape) must follow
    the largest frame of the WHOLE problem, not of the rank.  Two ranks on this box's GPU (gloo for the sums), one
    with 96-track frames, one with 600-track frames: every frame's PreSync cost, winning hypothesis, GuessMotion
    estimate and GuessK are the single-process run's bit for bit; the window sums agree to their association."""
    import json
    import socket
    import subprocess
    import sys
    import rssync_amd
    from rssync_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = [str(tmp_path / f"r{r}.json") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "gpu_dist_worker.py"), str(r), "2", str(port), outs[r]])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [json.load(open(o)) for o in outs]
    F = 16
    n_of = lambda fr: 96 if fr < 8 else 600
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=6)
    one = rssync_amd.SyncProblem(seed=321, max_outer_iters=6)
    one.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr in range(F):
        one.SetTrackResult(*next(iter(synth.make_frames(gyro, fr, fr + 1, n_of(fr), seed=6))))
    d, c, fc, bh = one.presync_curve(0.0, 0, F, 0.004, 0.06, per_frame=F)
    M, k = one.init_motion(0.03, 0, F - 1)
    cs, ds = one.Sync(0.036, 0, F - 1, 0.0, 0.2)
    for r in res:
        b, e = r["frames"]
        np.testing.assert_array_equal(np.asarray(r["best_h"]), bh[:, b:e])
        np.testing.assert_array_equal(np.asarray(r["frame_costs"]).view(np.uint64), np.ascontiguousarray(fc[:, b:e]).view(np.uint64))
        np.testing.assert_array_equal(np.asarray(r["M"]).view(np.uint64), np.ascontiguousarray(M[b:e]).view(np.uint64))
        np.testing.assert_array_equal(np.asarray(r["k"]).view(np.uint64), np.ascontiguousarray(k[b:e]).view(np.uint64))
        np.testing.assert_allclose(r["curve"], c, rtol=1e-13)
        assert r["sync"][1] == pytest.approx(ds, abs=1e-9) and r["iters"] == len(one.sync_trace())
    assert res[0]["sync"] == res[1]["sync"] and res[0]["curve"] == res[1]["curve"]


def _enqueued(its):
    """iterations rship_sync_run enqueues with the RCCL communicator on the stream for a loop that ends after `its`: a block of
    eight, then blocks of two (round 6; four until then -- with ranks an empty iteration is two exchanges for everybody)"""
    return 8 if its <= 8 else 8 + 2 * -(-(its - 8) // 2)


def test_ranked_device_loop_with_two_ranks(tmp_path):
    """with ranks, Sync's loop stays on the device -- window sums of this rank -> sum over the
    ranks -> the decision kernels, on every rank.  Two ranks share this box's GPU (RCCL cannot: the sum travels through
    the reduce hook over gloo, called between the kernels; with the library's communicator the same step is an
    ncclAllReduce on the stream, test_rccl_reduce_hook_on_the_device).  Uneven frame split, windows that lie on one
    rank only / on both / on none: every trace row equals the host loop's with the same hook, on both ranks, bit for
    bit; and the exchanges are two per launch of the loop instead of two to three blocking ones per iteration."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = [str(tmp_path / f"r{r}.json") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "gpu_rank_loop_worker.py"), str(r), "2", str(port), outs[r]])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [json.load(open(o)) for o in outs]
    for r in res:
        for key in ("sync", "trace", "cw", "dw", "wtr", "one", "one_trace", "simplified", "simplified_trace"):
            assert r["host"][key] == r["device"][key], key
        assert len(r["device"]["trace"]) >= 5
        its = len(r["device"]["trace"])
        # through the hook the counter of active windows is looked at after EVERY iteration (round 6: the stream is drained at
        # each exchange anyway), so no iteration is enqueued in vain: two exchanges per launch of the loop -- an outer
        # iteration, or the extra launch of a search that waits for its later trials -- and the final loss.  (Round 5: two per
        # ENQUEUED iteration, a block of eight and blocks of four: 26 per bench step where 20 do.)
        assert 2 * its + 1 <= r["device"]["exchanges"] <= 2 * (its + 2) + 1, (its, r["device"]["exchanges"])
    for key in ("sync", "trace", "cw", "dw", "wtr", "one", "one_trace", "simplified"):
        assert res[0]["device"][key] == res[1]["device"][key], key               # both ranks took the same decisions
    assert len(res[1]["device"]["one_trace"]) >= 3                               # (rank 1 held none of that window's frames)


@pytest.mark.parametrize("kernel", KERNELS)
def test_rows_below_safe_normalizes_threshold_and_the_non_finite_r_panic(kernel, monkeypatch):
    """Rows of P that are exactly zero (here: tracks whose two rays are the zero vector) are below the 1e-12 of
    safe_normalize (core_private.cpp:35-36, inline_utils.hpp:5-11) and stay as they are; a hypothesis drawn from such a row
    is the zero vector, with it M = 0, k = 100 / 0 -> clamp, r = P M k / |M| = NaN: the reference panics on r (:81), before
    rho.  The kernels' hot path carries neither the per-row threshold test nor a per-row |r| check (a minimum of |P|^2 per
    thread sends the wave through the careful form of the rows; a non-finite cost sum looks up which check fires first):
    the same panic as the oracle, with every row zero and with three zero rows per frame among ordinary ones.
    (Rays that merely coincide after the rotation do not make zero rows here: the fp32 cross product's fused
    multiply-adds leave 1e-9 of rounding, which is normalised like any other row.)"""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem, OracleError
    F, N = 6, 200
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=8)
    frames = list(synth.make_frames(g, 0, F, N, seed=8))
    for zero_rows in (slice(None), [3, 77, 150]):
        if kernel == "tile":
            monkeypatch.setenv("RSSYNC_NO_SMALL_LMEDS", "1")
        h = rssync_amd.SyncProblem(seed=SEED)
        monkeypatch.delenv("RSSYNC_NO_SMALL_LMEDS", raising=False)
        o = OracleProblem(seed=SEED, threads=4, faithful=False)
        for p in (h, o):
            p.SetGyroQuaternions(g.quats, g.fs, g.t0)
            for fr, ta, tb, ra, rb in frames:
                ra, rb = ra.copy(), rb.copy()
                ra[zero_rows] = 0.0
                rb[zero_rows] = 0.0
                p.SetTrackResult(fr, ta, tb, ra, rb)
        with pytest.raises(OracleError, match="pre-sync: non-finite r"):
            o.PreSync(0.0, 0, F, 0.002, 0.02)
        with pytest.raises(rssync_amd.RsSyncError, match="pre-sync: non-finite r"):
            h.PreSync(0.0, 0, F, 0.002, 0.02)


def test_near_static_camera_the_hypothesis_rule_on_unnormalised_rows():
    """core_private.cpp:45-46: v = safe_normalize(cross(P[i0], P[i1])) -- the 1e-12 threshold of inline_utils.hpp:5-11
    applies to |P[i0] x P[i1]| of the UN-normalised rows.  For a near-static camera (|P| ~ translation / depth ~ 1e-6 at the
    true delay) that product is below the threshold for most pairs: the reference leaves v tiny, its residuals shrink
    with it and such a hypothesis wins the LMedS outright.  Rounds 1-4 applied the threshold to the unit rows' cross
    product and normalised those directions (DESIGN.md deviation 5, gone in round 5): the device then picked another winner
    wherever the oracle's is an un-normalised direction -- about half of the (frame, candidate) pairs.  Round 5 had the rule
    but not the precision: rows of 2e-6 built from fp32 rays (6e-8 absolute) are 3 % off, 84 % identical winners, and this
    test asserted 0.70 / 0.75 / rtol 0.06.  Round 6: the reference computes rows, norms and the rule in double
    (core_private.cpp:19-28), and so does the sweep for exactly such (frame, candidate) pairs -- rows from the fp64 streams,
    rounded once (kernels/lmeds.hpp, "fp64 rows") -- so the bounds are those of every other scene: 0.99 and 2e-3
    (tests/measure/gpu_near_static.py -> profiles/r6_near_static.json: 100 % identical winners at every |P| from 2e-3 down
    to 5e-7).  The ordinary frames beside them never take the fp64 form (the counter of the debug ABI)."""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    F, N = 12, 600
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=9)
    # frames 0 .. 7 near-static (5e-5 m per frame at 2 .. 50 m: |P| ~ 2e-6, ray noise in proportion), 8 .. 11 ordinary
    frames = list(synth.make_frames(g, 0, 8, N, seed=9, noise=1e-6, outliers=0.1, translation=5e-5))
    frames += list(synth.make_frames(g, 8, F, N, seed=9, noise=1e-3, outliers=0.1))
    h = rssync_amd.SyncProblem(seed=SEED)
    o = OracleProblem(seed=SEED, threads=min(os.cpu_count() or 1, 16), faithful=False)
    for p in (h, o):
        p.SetGyroQuaternions(g.quats, g.fs, g.t0)
        for fr in frames:
            p.SetTrackResult(*fr)
    dh, ch, fch, bhh = h.presync_curve(synth.D_TRUE, 0, F, 2e-6, 2e-5, per_frame=F)   # 20 candidates within 20 us of the truth
    do, co, fco, bho = o.presync_curve(synth.D_TRUE, 0, F, 2e-6, 2e-5, per_frame=F)
    np.testing.assert_array_equal(dh, do)
    # which of the oracle's winners are un-normalised directions (|M| far below 1)
    unn = np.array([[np.linalg.norm(o.guess_motion(f, float(do[c]), 20, c)[0]) < 0.5 for f in range(F)] for c in range(len(do))])
    assert 0.25 < unn[:, :8].mean() < 0.9, unn[:, :8].mean()      # the scene is in the regime the rule is about ...
    assert not unn[:, 8:].any()                                   # ... and the ordinary frames are not
    same = bhh == bho
    assert same[:, 8:].mean() > 0.99                              # ordinary frames beside them: as everywhere else
    assert same[:, :8][unn[:, :8]].mean() >= 0.99, same[:, :8][unn[:, :8]].mean()     # (rounds 1-4: ~0 here; round 5: 0.7-0.84)
    assert same[:, :8].mean() >= 0.99, same[:, :8].mean()
    rel = np.abs(fch - fco) / fco
    assert np.median(rel[same]) < 2e-3 and np.median(rel[:, :8][same[:, :8]]) < 1e-5
    np.testing.assert_allclose(ch, co, rtol=2e-3)
    # exactly the near-static frames' pairs went through the fp64 form: 8 frames x 20 candidates, none of the ordinary frames'
    st = h.near_static_stats()
    assert st["pairs"] == 8 * len(do) and st["sweeps"] == 1 and st["searches"] == 0, st
    # ... and with the mechanism off (RSSYNC_NO_FP64_ROWS=1: round 5's sweep) the fp32 rows show what they cost here
    os.environ["RSSYNC_NO_FP64_ROWS"] = "1"
    try:
        h32 = rssync_amd.SyncProblem(seed=SEED)
    finally:
        del os.environ["RSSYNC_NO_FP64_ROWS"]
    h32.SetGyroQuaternions(g.quats, g.fs, g.t0)
    for fr in frames:
        h32.SetTrackResult(*fr)
    bh32 = h32.presync_curve(synth.D_TRUE, 0, F, 2e-6, 2e-5, per_frame=F)[3]
    assert h32.near_static_stats()["pairs"] == 0
    assert 0.6 < (bh32 == bho)[:, :8].mean() < 0.95
    np.testing.assert_array_equal(bh32[:, 8:], bhh[:, 8:])       # (the ordinary frames' results do not depend on the switch)
    # the one-wave kernels take the same decisions as the tile kernel on the same data (every family recomputes the two
    # rows' norms with the same routine, lmeds.hpp: row_scale_general)
    small = list(synth.make_frames(g, 0, 8, 130, seed=9, noise=1e-6, outliers=0.1, translation=5e-5))
    res = {}
    for tile in ("0", "1"):
        os.environ["RSSYNC_NO_SMALL_LMEDS"] = tile
        try:
            q = rssync_amd.SyncProblem(seed=SEED)
        finally:
            del os.environ["RSSYNC_NO_SMALL_LMEDS"]
        q.SetGyroQuaternions(g.quats, g.fs, g.t0)
        for fr in small:
            q.SetTrackResult(*fr)
        res[tile] = q.presync_curve(synth.D_TRUE, 0, 8, 2e-6, 2e-5, per_frame=8)
    np.testing.assert_array_equal(res["0"][3], res["1"][3])        # winners: one wave per frame == four-wave tile kernel
    np.testing.assert_allclose(res["0"][2], res["1"][2], rtol=2e-6)
    # ... and both agree with the oracle on such frames (both took the fp64 form of the rows)
    os_ = OracleProblem(seed=SEED, threads=min(os.cpu_count() or 1, 16), faithful=False)
    os_.SetGyroQuaternions(g.quats, g.fs, g.t0)
    for fr in small:
        os_.SetTrackResult(*fr)
    bhs = os_.presync_curve(synth.D_TRUE, 0, 8, 2e-6, 2e-5, per_frame=8)[3]
    assert (res["0"][3] == bhs).mean() >= 0.99
    # the kernel for frames of more than 8192 tracks takes its fp64 rows in place (kernels/lmeds_big.hpp): the same frames forced through it
    os.environ["RSSYNC_FORCE_BIG"] = "1"
    try:
        qb = rssync_amd.SyncProblem(seed=SEED)
    finally:
        del os.environ["RSSYNC_FORCE_BIG"]
    qb.SetGyroQuaternions(g.quats, g.fs, g.t0)
    for fr in small:
        qb.SetTrackResult(*fr)
    rb = qb.presync_curve(synth.D_TRUE, 0, 8, 2e-6, 2e-5, per_frame=8)
    assert (rb[3] == bhs).mean() >= 0.99 and qb.near_static_stats()["pairs"] == 8 * len(rb[0])
    np.testing.assert_allclose(rb[2], res["0"][2], rtol=1e-5)


@pytest.mark.parametrize("N,F,n_cand", [(130, 96, 300), (600, 96, 300), (2048, 40, 700)])
def test_near_static_pairs_in_chunks_that_straddle_the_mask_words(N, F, n_cand):
    """The sweep flags near-static (frame, candidate) pairs in a bitmap -- one bit per pair, 32 candidates per word -- and the
    fp64 form of the kernels serves and clears the bits chunk by chunk (kernels/lmeds.hpp: LmedsParams::redo_mask).  Chunks of
    three candidates straddle the words.  A clip whose first half is near-static: exactly those pairs are recomputed,
    the winners are the oracle's, a second sweep finds the bitmap clean (the same count again, the same bits), and a
    PreSync over the whole clip returns the oracle's delay."""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=33)
    H = F // 2
    frames = list(synth.make_frames(g, 0, H, N, seed=33, noise=1e-6, outliers=0.1, translation=5e-5))
    frames += list(synth.make_frames(g, H, F, N, seed=33, noise=1e-3, outliers=0.1))
    h = rssync_amd.SyncProblem(seed=SEED)
    o = OracleProblem(seed=SEED, threads=min(os.cpu_count() or 1, 16), faithful=False)
    for p in (h, o):
        p.SetGyroQuaternions(g.quats, g.fs, g.t0)
        for fr in frames:
            p.SetTrackResult(*fr)
    step = 4e-5 / n_cand
    dh, ch, fch, bhh = h.presync_curve(synth.D_TRUE, 0, F, step, 2e-5, per_frame=F)
    n_cand = len(dh)
    assert h.window_info()["presync_chunk"] == 3, h.window_info()           # (32 is not a multiple of it)
    st = h.near_static_stats()
    assert st == dict(pairs=H * n_cand, sweeps=1, searches=0), st
    sub = np.arange(0, n_cand, 17)                              # the oracle on a sample of the candidates (its own streams follow the index)
    same = []
    for ci in sub:
        for f in range(0, F, 5):
            same.append(int(bhh[ci, f]) == int(o.guess_motion(f, float(dh[ci]), 20, int(ci))[1]))
    assert np.mean(same) >= 0.99, np.mean(same)
    d2, c2, fc2, bh2 = h.presync_curve(synth.D_TRUE, 0, F, step, 2e-5, per_frame=F)
    np.testing.assert_array_equal(fc2.view(np.uint64), fch.view(np.uint64))
    np.testing.assert_array_equal(bh2, bhh)
    assert h.near_static_stats() == dict(pairs=2 * H * n_cand, sweeps=2, searches=0)
    assert h.PreSync(synth.D_TRUE, 0, F, 1e-6, 2e-5)[1] == o.PreSync(synth.D_TRUE, 0, F, 1e-6, 2e-5)[1]


@pytest.mark.parametrize("N", [130, 400, 600, 2000, 5000, 7000, 9000])
def test_guess_motion_on_near_static_frames_takes_its_rows_from_the_fp64_streams(N, monkeypatch):
    """GuessMotion's 200-hypothesis search (core_private.cpp:125-128 -> :34-59) at the start of every Sync call is the same
    LMedS on the same rows as the sweep's: on near-static frames it takes the fp64 form IN PLACE (one candidate per workgroup:
    no second launch; kernels/lmeds.hpp MODE 1, lmeds_small.hpp, lmeds_big.hpp), in every kernel family -- and the window
    executor's search task does the same (exec_big.hpp for frames of more than 512 tracks), so that executor and launch chain
    pick the same winners bit for bit.  Against the oracle's fp64 search: the winners agree on every frame but a few near-ties
    (round 5, fp32 rows: about one frame in five differs here); ordinary frames beside them never take the form."""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    F = 16 if N <= 2000 else 6
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=19)
    H = F // 2
    frames = list(synth.make_frames(g, 0, H, N, seed=19, noise=1e-6, outliers=0.1, translation=5e-5))
    frames += list(synth.make_frames(g, H, F, N, seed=19, noise=1e-3, outliers=0.1))
    monkeypatch.setenv("RSSYNC_EXEC_BIG_SHARE", "1")
    monkeypatch.setenv("RSSYNC_EXEC_BIG_MAX", "16384")
    res = {}
    for name, env in (("executor", {}), ("chain", {"RSSYNC_EXECUTOR": "0"}), ("fp32", {"RSSYNC_EXECUTOR": "0", "RSSYNC_NO_FP64_ROWS": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        p = rssync_amd.SyncProblem(seed=SEED, max_outer_iters=4)
        for k in env:
            monkeypatch.delenv(k)
        p.SetGyroQuaternions(g.quats, g.fs, g.t0)
        for fr in frames:
            p.SetTrackResult(*fr)
        p.record_init_winners(name != "executor")      # (recording keeps a problem out of the executor: the chain's winners are read this way)
        r = p.Sync(synth.D_TRUE, 0, F - 1, 0.0, 0.1)
        res[name] = (r, p.sync_trace().copy(), p.last_init_winners() if name != "executor" else None, p.near_static_stats(), p.executor_stats()["runs"])
    o = OracleProblem(seed=SEED, max_outer_iters=4, threads=min(os.cpu_count() or 1, 16), faithful=False)
    o.SetGyroQuaternions(g.quats, g.fs, g.t0)
    for fr in frames:
        o.SetTrackResult(*fr)
    o.Sync(synth.D_TRUE, 0, F - 1, 0.0, 0.1)
    wo = o.last_init_winners()
    # the chain: exactly the near-static frames took the fp64 form; its winners are the oracle's (a near-tie may differ)
    assert res["chain"][3]["searches"] == H and res["fp32"][3]["searches"] == 0, (res["chain"][3], res["fp32"][3])
    same = res["chain"][2] == wo
    assert same[:H].mean() >= 0.85 and same.mean() >= 0.85, (same, N)
    # the ordinary frames' winners do not depend on the mechanism
    np.testing.assert_array_equal(res["chain"][2][H:], res["fp32"][2][H:])
    # the window executor (where the selection is its to take: class 4 is the chain's) == the chain, bit for bit
    if res["executor"][4]:
        assert res["executor"][0] == res["chain"][0]
        np.testing.assert_array_equal(res["executor"][1].view(np.uint64), res["chain"][1].view(np.uint64))
        # (with RSSYNC_EXECUTOR_CHECK=1 every executor call is run again by the chain, whose searches count as well)
        assert res["executor"][3]["searches"] == (2 * H if os.environ.get("RSSYNC_EXECUTOR_CHECK", "0") not in ("", "0") else H)
    else:
        assert N in (7000,), N           # (frames of 6145 .. 8192 tracks: the eight-wave tile kernel; the executor leaves them to the chain)


@pytest.mark.parametrize("N", [130, 600, 1500, 3000])
def test_ordinary_scenes_never_take_the_fp64_rows(N):
    """The near-static watch of the PreSync sweep (kernels/lmeds.hpp, "fp64 rows": a quarter of a frame's first 64 rows with
    |P| below 2e-4) must not fire on ordinary footage -- noisy or noise-free, at the true delay or 100 ms away from it, in
    any kernel family: the debug ABI's counter stays at zero, and the sweep's results are bit for bit those of the library
    with the mechanism switched off."""
    import rssync_amd
    from rssync_amd import synth
    F = 6
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=21)
    for noise, outliers in ((1e-3, 0.1), (0.0, 0.0)):
        frames = list(synth.make_frames(g, 0, F, N, seed=21, noise=noise, outliers=outliers))
        res = []
        for off in ("0", "1"):
            os.environ["RSSYNC_NO_FP64_ROWS"] = off
            try:
                q = rssync_amd.SyncProblem(seed=SEED)
            finally:
                del os.environ["RSSYNC_NO_FP64_ROWS"]
            q.SetGyroQuaternions(g.quats, g.fs, g.t0)
            for fr in frames:
                q.SetTrackResult(*fr)
            r = q.presync_curve(synth.D_TRUE, 0, F, 5e-4, 0.1, per_frame=F)      # 400 candidates, the true delay among them
            r2 = q.PreSync(0.0, 0, F, 0.002, 0.05)
            r3 = q.Sync(r2[1], 0, F - 1, 0.0, 0.1)        # (GuessMotion's search watches too)
            assert q.near_static_stats() == dict(pairs=0, sweeps=0, searches=0)
            res.append((r, r2, r3))
        np.testing.assert_array_equal(res[0][0][2].view(np.uint64), res[1][0][2].view(np.uint64))
        np.testing.assert_array_equal(res[0][0][3], res[1][0][3])
        assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
