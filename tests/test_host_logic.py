"""The product's host solver (sync_problem.cpp: packing, delay splitting, frame selection, the
Sync control loop, panics, the reduce hook) on a machine without a GPU, by linking it against
the CPU test double of the device ABI (tests/cpu_device).  The arithmetic is the kernels' own
header (device_math.hpp, fp32), so these are also CPU-side parity checks of that math against
the fp64 oracle."""
import os

import numpy as np
import pytest

from conftest import fill

SEED = 123


@pytest.fixture()
def host(hosttest_lib):
    import rssync_amd

    def make(case=None, **kw):
        p = rssync_amd.SyncProblem(seed=SEED, _lib=hosttest_lib, **kw)
        return fill(p, case) if case is not None else p
    return make


@pytest.fixture(scope="module")
def tiny_case(built):
    from rssync_amd import synth
    F, N = 16, 96
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=5)
    return dict(F=F, N=N, gyro=gyro, frames=list(synth.make_frames(gyro, 0, F, N, seed=5)))


@pytest.fixture(scope="module")
def tiny_clean(built):
    from rssync_amd import synth
    F, N = 16, 96
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=6)
    return dict(F=F, N=N, gyro=gyro, frames=list(synth.make_frames(gyro, 0, F, N, seed=6, noise=0.0, outliers=0.0)))


def _oracle(case, **kw):
    from oracle.oracle import OracleProblem
    return fill(OracleProblem(seed=SEED, threads=os.cpu_count() or 1, faithful=False, **kw), case)


def test_residual_rows_fp32_vs_oracle(host, tiny_case):
    h, o = host(tiny_case), _oracle(tiny_case)
    fs, start, n = o.gyro_info()
    for fr, d in [(0, 0.0), (7, 0.0371), (15, -0.12), (3, -3.0), (3, n / fs + 1.0)]:  # incl. both extrapolations
        P, dP = h.problem_matrix(fr, d, tiny_case["N"], deriv=True)
        Po = o.problem_matrix(fr, d)
        assert np.abs(P - Po).max() < (5e-7 if abs(d) < 1 else 2e-4)
        if abs(d) < 1:
            eps = 1e-6
            dPo = (o.problem_matrix(fr, d + eps) - o.problem_matrix(fr, d - eps)) / (2 * eps)
            assert np.abs(dP - dPo).max() < 2e-5 * max(1.0, np.abs(dPo).max())


def test_presync_matches_oracle(host, tiny_case):
    F = tiny_case["F"]
    h, o = host(tiny_case), _oracle(tiny_case)
    dh, ch, fch, bhh = h.presync_curve(0.0, 0, F, 0.004, 0.1, per_frame=F)
    do, co, fco, bho = o.presync_curve(0.0, 0, F, 0.004, 0.1, per_frame=F)
    np.testing.assert_array_equal(dh, do)
    same = bhh == bho
    assert same.mean() > 0.99
    np.testing.assert_allclose(fch[same], fco[same], rtol=1e-3)
    assert np.argmin(ch) == np.argmin(co)
    assert h.PreSync(0.0, 0, F, 0.004, 0.1)[1] == o.PreSync(0.0, 0, F, 0.004, 0.1)[1]
    d2, c2 = h.DebugPreSync(0.01, 0, 8, 0.05, 11)
    d3, c3 = o.DebugPreSync(0.01, 0, 8, 0.05, 11)
    np.testing.assert_array_equal(d2, d3)
    np.testing.assert_allclose(c2, c3, rtol=5e-3)


def test_loss_gradient_and_motion(host, tiny_case):
    F = tiny_case["F"]
    h, o = host(tiny_case), _oracle(tiny_case)
    d0 = 0.036
    M, k = h.init_motion(d0, 0, F - 1)
    L, G = h.loss([d0, d0 + 2e-3], grad=True)
    for j, dd in enumerate([d0, d0 + 2e-3]):
        Lo = Gn = 0.0
        for f in range(F):
            l, dn, da, _ = o.loss(f, dd, M[f], k[f])
            Lo += l
            Gn += dn
        assert L[j] == pytest.approx(Lo, rel=1e-6)
        assert G[j] == pytest.approx(Gn, rel=5e-4, abs=5e-4 * abs(L[j]))
    M2, k2, its, evs = h.opt_motion(d0)
    assert h.loss([d0])[0] < L[0] and evs >= its >= F
    np.testing.assert_array_equal(k2, k)


def test_sync_control_loop_matches_oracle_on_clean_data(host, tiny_clean):
    from rssync_amd import synth
    F = tiny_clean["F"]
    h, o = host(tiny_clean), _oracle(tiny_clean)
    ch, dh = h.Sync(0.036, 0, F - 1, 0.0, 0.2)
    co, do, tro = o.sync_trace(0.036, 0, F - 1, 0.0, 0.2)
    trh = h.sync_trace()
    assert abs(dh - synth.D_TRUE) < 1e-4 and abs(dh - do) < 1e-4
    assert len(trh) == len(tro)
    # same outer-loop decisions: accepted backtracking step and number of trials per iteration
    np.testing.assert_array_equal(trh[:, 5], tro[:, 5])
    np.testing.assert_allclose(trh[:, 4], tro[:, 4])
    np.testing.assert_allclose(trh[:, 0], tro[:, 0], atol=2e-5)   # delay after each iteration
    np.testing.assert_allclose(trh[:, 2], tro[:, 2], rtol=2e-2, atol=1e-3)  # loss at the look-ahead point


def test_iteration_cap_window_exit_and_frame_ranges(host, tiny_case):
    F = tiny_case["F"]
    h = host(tiny_case, max_outer_iters=3)
    h.Sync(0.036, 0, F - 1, 0.0, 0.2)
    assert len(h.sync_trace()) == 3
    h.set_max_outer_iters(400)
    h.Sync(0.036, 0, F - 1, 0.5, 1e-3)  # centre far away: leaves the window after one step
    assert len(h.sync_trace()) == 1
    M, k = h.init_motion(0.03, 4, 6)     # Sync range is end-inclusive (core_private.cpp:219)
    assert len(k) == 3
    d, c, fc, bh = h.presync_curve(0.03, 4, 6, 0.01, 0.02, per_frame=2)  # PreSync end-exclusive (:66)
    assert fc.shape[1] == 2
    c0, d0 = h.PreSync(0.0, 100, 200, 0.01, 0.05)  # no frames: zero cost, first candidate
    assert c0 == 0.0 and d0 == pytest.approx(-0.05)


def test_reduce_hook_is_called_twice_per_outer_iteration(host, tiny_case):
    F = tiny_case["F"]
    h = host(tiny_case, max_outer_iters=4)
    calls = []
    h.set_reduce_hook(lambda a: calls.append(len(a)))
    h.PreSync(0.0, 0, F, 0.01, 0.05)
    # the sweep: candidate costs + 4 status flags in ONE exchange -- and nothing else: a frame's kernels follow its own
    # track count (size classes), the ranks agree on nothing (rounds 2-4 began every call with a 149-double exchange)
    assert calls == [10 + 4]
    calls.clear()
    h.set_tracks_hint(tiny_case["N"])                  # (accepted and ignored: kept for callers written against round 4)
    h.PreSync(0.0, 0, F, 0.01, 0.05)
    assert calls == [10 + 4]
    calls.clear()
    h.Sync(0.036, 0, F - 1, 0.0, 0.2)
    iters = len(h.sync_trace())
    # per outer iteration {loss, grad} and the first five line-search trials (the second five only when all
    # of the first five fail the Armijo test: none here); then the final loss
    assert len(calls) == 2 * iters + 1
    assert calls[0] == 2 and calls[1] == 5 and calls[-1] == 1


def test_a_failing_reduce_hook_is_a_panic_not_a_silent_skip(host, tiny_case):
    """an exception inside the Python hook (ctypes would swallow it) or a non-zero status from a C hook
    must stop the call: a rank that continued with rank-local sums would diverge from the others"""
    import rssync_amd
    F = tiny_case["F"]
    h = host(tiny_case, max_outer_iters=2)

    def broken(arr):
        raise ValueError("staging buffer too small")

    h.set_reduce_hook(broken)
    with pytest.raises(rssync_amd.RsSyncError, match="reduce hook failed") as ei:
        h.PreSync(0.0, 0, F, 0.01, 0.05)
    assert isinstance(ei.value.__cause__, ValueError)
    h.set_reduce_hook(None)
    c, d = h.PreSync(0.0, 0, F, 0.01, 0.05)
    assert np.isfinite(c)


def test_panics_follow_the_reference_messages(host, tiny_case):
    import rssync_amd
    h = host()
    fr, ta, tb, ra, rb = tiny_case["frames"][0]
    for name, args in (("rays_a", (ta, tb, np.where(np.arange(ra.size).reshape(ra.shape) == 4, np.nan, ra), rb)),
                       ("rays_b", (ta, tb, ra, np.where(np.arange(rb.size).reshape(rb.shape) == 4, np.inf, rb))),
                       ("ts_a", (np.where(np.arange(ta.size) == 1, np.nan, ta), tb, ra, rb)),
                       ("ts_b", (ta, np.where(np.arange(tb.size) == 1, -np.inf, tb), ra, rb))):
        with pytest.raises(rssync_amd.RsSyncError, match="set-track-result: non-finite numbers in " + name):
            h.SetTrackResult(0, *args)
    h.SetTrackResult(0, ta, tb, ra, rb)
    with pytest.raises(rssync_amd.RsSyncError, match="gyro data was not set"):
        h.Sync(0.0, 0, 0, 0.0, 1.0)
    g = tiny_case["gyro"]
    h.SetGyroQuaternions(g.quats, g.fs, g.t0)
    h.SetTrackResult(1, ta[:1], tb[:1], ra[:1], rb[:1])
    with pytest.raises(rssync_amd.RsSyncError, match="fewer than 2 tracks"):
        h.PreSync(0.0, 0, 2, 0.01, 0.05)
    with pytest.raises(rssync_amd.RsSyncError, match="empty candidate list"):
        h.PreSync(0.0, 0, 1, 0.01, -0.05)
    ts = (np.arange(50) * 2500).astype(np.int64)
    ts[10], ts[11] = ts[11], ts[10]
    with pytest.raises(rssync_amd.RsSyncError, match="timestamps out of order at pos 11"):
        h.SetGyroQuaternionsTimestamped(ts, np.tile([1.0, 0, 0, 0], (50, 1)))
    n = 8193   # no per-frame limit, as in the reference (core_private.cpp:192-203): accepted like any other frame
    h.SetTrackResult(5, np.zeros(n), np.zeros(n), np.tile([0, 0, 1.0], (n, 1)), np.tile([0, 0, 1.0], (n, 1)))


def test_timestamped_gyro_and_gyro_replacement(host, tiny_case):
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    g = synth.make_gyro(1.0, 1.0 + 20 / synth.FPS, seed=21)
    ts_us, q = synth.make_timestamped(g, jitter=0.2, seed=4)
    h, o = host(), OracleProblem(seed=SEED, faithful=False)
    frames = list(synth.make_frames(g, 30, 42, 64, seed=5))
    for p in (h, o):
        p.SetGyroQuaternionsTimestamped(ts_us, q)
        for fr, ta, tb, ra, rb in frames:
            p.SetTrackResult(fr, ta, tb, ra, rb)
    assert h.gyro_info() == o.gyro_info()
    np.testing.assert_array_equal(h.gyro_knots(), o.gyro_knots())  # integer grid + slerp on the host
    c1 = h.PreSync(0.0, 30, 42, 0.004, 0.06)
    c2 = o.PreSync(0.0, 30, 42, 0.004, 0.06)
    assert c1[1] == c2[1] and c1[0] == pytest.approx(c2[0], rel=2e-3)
    # replace the gyro while the tracks stay (the 48-orientation sweep, core_testcode.cpp:216-224)
    g2 = synth.make_gyro(1.0, 1.0 + 20 / synth.FPS, seed=99)
    for p in (h, o):
        p.SetGyroQuaternions(g2.quats, g2.fs, g2.t0)
    w1 = h.PreSync(0.0, 30, 42, 0.004, 0.06)
    w2 = o.PreSync(0.0, 30, 42, 0.004, 0.06)
    assert w1[0] > c1[0] and w1[0] == pytest.approx(w2[0], rel=2e-3)


def test_large_absolute_times_lose_no_precision(host, tiny_clean):
    """Times reach the device as integer knot + fp32 offset, never as fp32 seconds: moving the
    whole recording 1000 s later (fp32 resolution there: 6e-5 s) must not change the loss curve,
    sampled here every 2 us."""
    import rssync_amd
    F = tiny_clean["F"]
    g = tiny_clean["gyro"]
    h0 = host(tiny_clean)
    h1 = host()
    shift = 1000.0
    h1.SetGyroQuaternions(g.quats, g.fs, g.t0 + shift)
    for fr, ta, tb, ra, rb in tiny_clean["frames"]:
        h1.SetTrackResult(fr, ta + shift, tb + shift, ra, rb)
    M, k = h0.init_motion(0.036, 0, F - 1)
    h1.init_motion(0.036, 0, F - 1)
    h1.set_motion(M, k)
    d = 0.0365 + 2e-6 * np.arange(12)
    L0, G0 = h0.loss(d, grad=True)
    L1, G1 = h1.loss(d, grad=True)
    np.testing.assert_allclose(L1, L0, rtol=2e-6)
    np.testing.assert_allclose(G1, G0, rtol=1e-3, atol=1e-3 * np.abs(G0).max())
    dd = np.diff(L0)
    assert np.all(dd > 0) or np.all(dd < 0)  # 2 us steps are resolved monotonically


WINDOWS = [(0, 7), (4, 11), (8, 15), (2, 5), (0, 15)]  # overlapping, nested and whole-range


def test_sync_windows_equal_consecutive_sync_calls(host, tiny_case):
    """rssync_ext_sync_windows: window w is what the w-th of W consecutive Sync calls returns
    (same sampler stream, same per-window stopping), overlapping windows included."""
    b = [w[0] for w in WINDOWS]
    e = [w[1] for w in WINDOWS]
    d0 = [0.036, 0.030, 0.040, 0.036, 0.02]
    seq = host(tiny_case, max_outer_iters=25)
    ref, ref_tr = [], []
    for w in range(len(WINDOWS)):
        ref.append(seq.Sync(d0[w], b[w], e[w], 0.03, 0.05))
        ref_tr.append(seq.sync_trace())
    bat = host(tiny_case, max_outer_iters=25)
    costs, delays = bat.sync_windows(d0, b, e, 0.03, 0.05)
    assert len({len(t) for t in ref_tr}) > 1          # the windows stop at different iterations
    for w in range(len(WINDOWS)):
        assert delays[w] == ref[w][1] and costs[w] == ref[w][0]
        np.testing.assert_array_equal(bat.window_trace(w), ref_tr[w])
    # the sampler stream advanced by W: the next Sync equals the (W+1)-th sequential one
    assert bat.Sync(0.036, 0, 15, 0.03, 0.05) == seq.Sync(0.036, 0, 15, 0.03, 0.05)
    with pytest.raises(Exception):
        bat.window_trace(1)                           # a plain Sync holds one window


def test_pre_sync_windows_equal_separate_presync_calls(host, tiny_case):
    b = [w[0] for w in WINDOWS] + [40]
    e = [w[1] + 1 for w in WINDOWS] + [50]            # PreSync ranges are end-exclusive; last one is empty
    h = host(tiny_case)
    costs, delays = h.pre_sync_windows(0.03, b, e, 0.004, 0.04)
    for w in range(len(b)):
        c, d = h.PreSync(0.03, b[w], e[w], 0.004, 0.04)
        assert delays[w] == d
        assert costs[w] == pytest.approx(c, rel=1e-14, abs=0)
    calls = []
    h.set_reduce_hook(lambda a: calls.append(len(a)))
    h.pre_sync_windows(0.03, b, e, 0.004, 0.04)
    assert calls == [20 * len(b) + 4]                 # one exchange for all windows
    calls.clear()
    h.set_max_outer_iters(3)
    h.sync_windows(0.036, b[:3], [x - 1 for x in e[:3]], 0.0, 0.5)
    assert calls == [6, 15] * 3 + [3]                 # {loss, grad} and the first 5 trials per window; final losses


def test_sync_points_equal_the_reference_driver_loop(host, tiny_case):
    """rssync_ext_sync_points == core_testcode.cpp:303-316 run through the ISyncProblem methods."""
    pos, window, init = [0, 3, 6, 9], 6, 0.02
    for presync in (True, False):
        seq = host(tiny_case, max_outer_iters=12)
        want = []
        for p0 in pos:
            d, radius = init, np.inf
            if presync:
                radius = 0.04
                d = seq.PreSync(d, p0, p0 + window, 0.004, radius)[1]
            for _ in range(4):
                c, d = seq.Sync(d, p0, p0 + window, init, radius)
            want.append((c, d))
        bat = host(tiny_case, max_outer_iters=12)
        costs, delays = bat.sync_points(pos, window, init, *((0.004, 0.04) if presync else ()), repeats=4)
        for w in range(len(pos)):
            assert (costs[w], delays[w]) == want[w]
        # both consumed 16 Sync call numbers
        assert bat.Sync(0.03, 0, 15, 0.0, 0.2) == seq.Sync(0.03, 0, 15, 0.0, 0.2)


def test_random_window_sets_batched_equal_sequential(host, tiny_case):
    """property: for ANY set of windows (overlapping, nested, empty, out of range, repeated) the
    batched calls return what the sequential method calls return"""
    from hypothesis import given, settings, strategies as st, HealthCheck
    F = tiny_case["F"]
    window = st.tuples(st.integers(-2, F + 1), st.integers(0, 9)).map(lambda t: (t[0], t[0] + t[1]))

    @settings(max_examples=12, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(st.lists(window, min_size=1, max_size=5), st.floats(0.02, 0.05))
    def check(wins, d0):
        b = [w[0] for w in wins]
        e = [w[1] for w in wins]
        seq = host(tiny_case, max_outer_iters=4)
        bat = host(tiny_case, max_outer_iters=4)
        try:
            want = [seq.Sync(d0, b[w], e[w], 0.03, 0.05) for w in range(len(wins))]
        except Exception as ex:          # a window with a <2-track frame would panic in both
            with pytest.raises(type(ex)):
                bat.sync_windows(d0, b, e, 0.03, 0.05)
            return
        costs, delays = bat.sync_windows(d0, b, e, 0.03, 0.05)
        assert [(c, d) for c, d in zip(costs, delays)] == want
        pc, pd = bat.pre_sync_windows(0.03, b, [x + 1 for x in e], 0.01, 0.03)
        for w in range(len(wins)):
            c, d = seq.PreSync(0.03, b[w], e[w] + 1, 0.01, 0.03)
            assert pd[w] == d and pc[w] == pytest.approx(c, rel=1e-14, abs=0)
    check()


def test_host_solver_is_clean_under_asan_ubsan(built, tmp_path):
    """sync_problem.cpp + the CPU test double compiled with -fsanitize=address,undefined and driven
    through every entry point (tests/asan_driver.py): no report.  (Sanitizers run on the CPU build
    only; the GPU pool has no ASan.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = tmp_path / "librssync_hosttest_asan.so"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", str(lib),
                           os.path.join(root, "rs-sync_amd", "csrc", "sync_problem.cpp"),
                           os.path.join(root, "tests", "cpu_device", "rship_cpu.cpp")])
    asan = subprocess.check_output(["g++", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "asan_driver.py"), str(lib)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("done"), r.stderr[-2000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]


def test_simplified_mode_matches_the_oracle_and_recovers_the_delay_without_translation(hosttest_lib):
    """thesis section 2.11 eq. (12): loss sum log1p((k |h_j|)^2), one-dimensional problem.  Product host
    code (+ CPU test double) against the oracle's restatement; on a scene without translation and
    without noise the true delay is recovered within 1e-4 s."""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    F, N = 16, 96
    g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=3)
    for kw, check_truth in ((dict(noise=0.0, outliers=0.0, translation=0.0), True), (dict(), False)):
        h = rssync_amd.SyncProblem(seed=1, _lib=hosttest_lib)
        o = OracleProblem(seed=1, faithful=False)
        synth.fill(h, g, 0, F, N, seed=3, **kw)
        synth.fill(o, g, 0, F, N, seed=3, **kw)
        ch, dh = h.SyncSimplified(0.0355, 0, F - 1, 0.0, 0.2)
        co, do, tro = o.sync_simplified_trace(0.0355, 0, F - 1, 0.0, 0.2)
        # (the oracle differentiates numerically, +-1e-6 s as the reference does; the product analytically:
        # the iterates agree to ~1e-11 s, and at a residual-free minimum the cost itself is that sensitive)
        assert dh == pytest.approx(do, abs=1e-9) and ch == pytest.approx(co, rel=1e-5, abs=1e-12)
        assert len(h.sync_trace()) == len(tro)
        if check_truth:
            assert abs(dh - synth.D_TRUE) < 1e-4
        k = h.init_k_simplified(0.0355, 0, F - 1)
        L, G = h.loss_simplified([0.0355, 0.03], grad=True)
        for j, d in enumerate((0.0355, 0.03)):
            per = [o.loss_simplified(f, 0.0355, d) for f in range(F)]
            np.testing.assert_allclose(k, [p[0] for p in per], rtol=1e-12)
            assert L[j] == pytest.approx(sum(p[1] for p in per), rel=1e-12, abs=1e-12)
            assert G[j] == pytest.approx(sum(p[2] for p in per), rel=1e-6, abs=1e-6 * abs(L[j]))   # analytic vs +-1e-6 s


def test_timestamped_grid_in_closed_form_equals_the_reference_loop(host):
    """The timestamped gyro setter's grid (core_private.cpp:147-160) is a loop in the reference and in the oracle;
    the product derives rate, first sample and count from the two end timestamps in closed form
    (gyro_math.hpp: grid_of, shared by the device route and this CPU stand-in).  Random recordings: same rate,
    same first knot time, same count, same knots; and the rates route (optdata_fill_gyro) on top of it."""
    from oracle import oracle as ora
    rng = np.random.default_rng(77)
    for case in range(120):
        n = int(rng.integers(2, 300))
        rate = float(rng.choice([47.0, 50.0, 99.0, 200.0, 399.7, 400.0, 1000.0, 1601.0, 3200.0]))
        first = int(rng.integers(0, 3_000_000))
        dt = 1e6 / rate
        ts = first + np.cumsum(np.maximum(1, np.round(dt * rng.uniform(0.6, 1.4, n)))).astype(np.int64)
        q = rng.standard_normal((n, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        h, o = host(), ora.OracleProblem()
        try:
            o.SetGyroQuaternionsTimestamped(ts, q)
        except ora.OracleError as e:          # e.g. fewer than two grid points: the same complaint on both sides
            with pytest.raises(Exception, match=str(e).split(":")[-1].strip()[:24]):
                h.SetGyroQuaternionsTimestamped(ts, q)
            continue
        h.SetGyroQuaternionsTimestamped(ts, q)
        assert h.gyro_info() == o.gyro_info(), (case, n, rate, first)
        np.testing.assert_array_equal(h.gyro_knots(), o.gyro_knots())
    for case in range(20):
        n = int(rng.integers(3, 400))
        t = 0.5 + np.cumsum(rng.uniform(0.001, 0.004, n))
        r = rng.standard_normal((n, 3))
        name = ["XYZ", "yXz", "ZxY", None][case % 4]
        q, us = ora.integrate_gyro(t, r, name)
        h, o = host(), ora.OracleProblem()
        h.set_gyro_rates(t, r, name)
        o.SetGyroQuaternionsTimestamped(us, q)
        assert h.gyro_info() == o.gyro_info()
        np.testing.assert_allclose(h.gyro_knots(), o.gyro_knots(), rtol=0, atol=1e-15)
