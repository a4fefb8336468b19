"""The window executor (kernels/executor.hpp: Sync for frames of up to 512 tracks as ONE device-scheduled launch --
tasks (window, phase, frame) pulled from a queue by persistent waves, windows advancing independently, a sync
point's four calls chained per window; the default for such frames, RSSYNC_EXECUTOR=0 -- read when a problem is
created -- keeps the chain) against the chain of launches: the same task bodies, sums and decisions, so the SAME BITS -- delays, costs, every trace row."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _two(**kw):
    import rssync_amd
    a = rssync_amd.SyncProblem(**kw)          # the executor (default)
    os.environ["RSSYNC_EXECUTOR"] = "0"
    try:
        b = rssync_amd.SyncProblem(**kw)      # the chain of launches
    finally:
        del os.environ["RSSYNC_EXECUTOR"]
    return a, b


def _bits(x):
    return np.ascontiguousarray(x, np.float64).view(np.uint64)


def _fill(ps, gyro, frames):
    for p in ps:
        p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
        for fr in frames:
            p.SetTrackResult(*fr)


@pytest.mark.parametrize("N", [512, 300, 256, 130, 64, 40])
def test_single_sync_call(built, N):
    from rssync_amd import synth
    F = 48
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=11)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=11))
    ex, ch = _two(seed=31, max_outer_iters=60)
    _fill((ex, ch), gyro, frames)
    d0 = ex.PreSync(0.0, 0, F, 0.002, 0.1)[1]
    assert d0 == ch.PreSync(0.0, 0, F, 0.002, 0.1)[1]
    r1, r2 = ex.Sync(d0, 0, F - 1, 0.0, 0.1), ch.Sync(d0, 0, F - 1, 0.0, 0.1)
    t1, t2 = ex.sync_trace(), ch.sync_trace()
    assert t1.shape == t2.shape and len(t1) >= 3
    np.testing.assert_array_equal(_bits(t1), _bits(t2))
    assert r1 == r2
    # a second call continues the sampler's call counter on both sides
    assert ex.Sync(r1[1], 0, F - 1, 0.0, 0.1) == ch.Sync(r2[1], 0, F - 1, 0.0, 0.1)
    np.testing.assert_array_equal(_bits(ex.sync_trace()), _bits(ch.sync_trace()))


def test_overlapping_ragged_windows_and_the_iteration_cap(built):
    from rssync_amd import synth
    F = 60
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=12)
    rng = np.random.default_rng(8)
    frames = []
    for fr in range(F):
        frames += list(synth.make_frames(gyro, fr, fr + 1, int(rng.integers(2, 256)), seed=12))
    ex, ch = _two(seed=32, max_outer_iters=7)   # the cap stops most windows before they converge
    _fill((ex, ch), gyro, frames)
    b = [0, 5, 11, 20, 21, 40, 3]
    e = [14, 25, 30, 44, 59, 59, 3]            # one window is a single frame
    d0 = np.linspace(0.034, 0.039, len(b))
    (c1, d1), (c2, d2) = ex.sync_windows(d0, b, e, 0.0, 0.2), ch.sync_windows(d0, b, e, 0.0, 0.2)
    np.testing.assert_array_equal(_bits(d1), _bits(d2))
    np.testing.assert_array_equal(_bits(c1), _bits(c2))
    for w in range(len(b)):
        np.testing.assert_array_equal(_bits(ex.window_trace(w)), _bits(ch.window_trace(w)))


@pytest.mark.parametrize("first_trials", [None, "1", "10"])
def test_sync_points_four_chained_calls(built, monkeypatch, first_trials):
    """the reference driver's loop (core_testcode.cpp:303-316): PreSync then four Sync calls per position; with
    RSSYNC_LOOP_FIRST_TRIALS=1 every line search has to wait for its later trials"""
    from rssync_amd import synth
    if first_trials:
        monkeypatch.setenv("RSSYNC_LOOP_FIRST_TRIALS", first_trials)
    F, N, window = 150, 130, 30
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=13)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=13))
    ex, ch = _two(seed=33, max_outer_iters=400)
    _fill((ex, ch), gyro, frames)
    pos = list(range(0, F - window - 1, 9))
    (c1, d1), (c2, d2) = ex.sync_points(pos, window, 0.0, 0.002, 0.1), ch.sync_points(pos, window, 0.0, 0.002, 0.1)
    np.testing.assert_array_equal(_bits(d1), _bits(d2))
    np.testing.assert_array_equal(_bits(c1), _bits(c2))
    for w in range(len(pos)):
        t1, t2 = ex.window_trace(w), ch.window_trace(w)
        assert t1.shape == t2.shape and len(t1) >= 4
        np.testing.assert_array_equal(_bits(t1), _bits(t2))
    # and the sequential calls through the ISyncProblem methods give the same delays
    ex2, _ = _two(seed=33, max_outer_iters=400)
    _fill((ex2,), gyro, frames)
    _, dd = ex2.sync_points(pos[:3], window, 0.0, 0.002, 0.1)
    ch3 = _two(seed=33, max_outer_iters=400)[1]
    _fill((ch3,), gyro, frames)
    seq3 = []
    for p0 in pos[:3]:
        d = ch3.PreSync(0.0, p0, p0 + window, 0.002, 0.1)[1]
        for _ in range(4):
            d = ch3.Sync(d, p0, p0 + window, 0.0, 0.1)[1]
        seq3.append(d)
    np.testing.assert_array_equal(_bits(dd), _bits(seq3))


def test_the_chain_takes_over_if_the_executor_gives_up(monkeypatch, capfd):
    """RSSYNC_EXECUTOR_FAIL=1 makes the executor report failure (as its watchdog would): the call is redone by the
    chain of launches from the same inputs -- same values, one line on stderr -- and the object stays with the chain"""
    import rssync_amd
    from rssync_amd import synth
    F, N = 30, 100
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=4)
    frames = list(synth.make_frames(g, 0, F, N, seed=4))
    monkeypatch.setenv("RSSYNC_EXECUTOR_FAIL", "1")
    a = rssync_amd.SyncProblem(seed=7, max_outer_iters=12)
    monkeypatch.delenv("RSSYNC_EXECUTOR_FAIL")
    monkeypatch.setenv("RSSYNC_EXECUTOR", "0")
    b = rssync_amd.SyncProblem(seed=7, max_outer_iters=12)
    _fill((a, b), g, frames)
    pos = [0, 5, 10]
    ra = a.sync_points(pos, 15, 0.0, 0.002, 0.1)
    rb = b.sync_points(pos, 15, 0.0, 0.002, 0.1)
    np.testing.assert_array_equal(_bits(ra[1]), _bits(rb[1]))
    np.testing.assert_array_equal(_bits(ra[0]), _bits(rb[0]))
    for w in range(len(pos)):
        np.testing.assert_array_equal(_bits(a.window_trace(w)), _bits(b.window_trace(w)))
    assert a.Sync(0.036, 0, F - 1, 0.0, 0.2) == b.Sync(0.036, 0, F - 1, 0.0, 0.2)
    err = capfd.readouterr().err
    assert err.count("continuing with the launch chain") == 1


def test_check_mode_reruns_every_call_through_the_chain_and_the_queue_ring_wraps(built):
    """RSSYNC_EXECUTOR_CHECK / set_executor_check: every call the executor runs (at its default eight one-wave
    workgroups per CU) is run again by the launch chain INSIDE the same call and must give the same bits, or the call
    panics -- the runtime form of this file's comparisons, for a protocol that is measured, not guaranteed
    (DESIGN.md section 4).  The same run must have wrapped the queue ring several times: a cell is reused lap after
    lap (lap-tagged cells, executor.hpp), which a run shorter than the ring would not exercise."""
    import rssync_amd
    from rssync_amd import synth
    F, N, window = 150, 130, 30
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=14)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=14))
    p = rssync_amd.SyncProblem(seed=34, max_outer_iters=400)
    _fill((p,), gyro, frames)
    p.set_executor_check(True)
    pos = list(range(0, F - window - 1, 9))
    c, d = p.sync_points(pos, window, 0.0, 0.002, 0.1)
    st = p.executor_stats()
    assert st["runs"] == 1 and st["checked"] == 1, st
    assert st["waves"] >= 8 * 64 or st["waves"] == len(pos) * (window + 1), st     # (eight per CU, or one per slot)
    # the ring: every task and end marker ever pushed has a number; the numbers wrapped the ring more than twice
    assert st["ring_cells"] >= 256 and st["tail"] > 2 * st["ring_cells"], st
    assert st["head"] >= st["tail"]                                           # every number pushed was claimed
    # single calls and lock-step windows are checked too
    r = p.Sync(float(d[0]), 0, 40, 0.0, 0.1)
    p.sync_windows([0.036, 0.037], [0, 50], [40, 90], 0.0, 0.1)
    st = p.executor_stats()
    assert st["runs"] == 3 and st["checked"] == 3, st
    assert np.isfinite(r[0]) and np.all(np.isfinite(c))
    # with the check off nothing is re-run
    p.set_executor_check(False)
    p.Sync(float(d[0]), 0, 40, 0.0, 0.1)
    st = p.executor_stats()
    assert st["runs"] == 4 and st["checked"] == 3, st


@pytest.mark.parametrize("fs,n_top", [(2000.0, 130), (3200.0, 130), (4000.0, 130), (8000.0, 130), (2000.0, 300), (4000.0, 512),
                                      (8000.0, 400)])
def test_executor_equals_the_chain_at_high_gyro_rates(built, fs, n_top):
    """Above ~1.7 kHz the spline windows are sized per problem, and for small frames the planner keeps the search on
    the general path beyond 128 knots: the executor's search must make the same choice as the launch chain's kernel
    (interior and general path round differently in fp32), or a near-tie between hypotheses falls the other way.  Ragged
    tiny frames make near-ties likely (a randomised 3.2 kHz case found exactly this in round 4).  n_top: the largest
    frame, which picks the instantiation (eight rows per lane from 257 tracks on)."""
    from rssync_amd import synth
    F = 24
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, fs=fs, seed=113)
    rng = np.random.default_rng(113)
    frames = []
    for fr in range(F):
        n = int(rng.choice([2, 4, 5, 7, 13, 16, 20, 23, 40, n_top]))
        n = n_top if fr == 3 else n
        frames += list(synth.make_frames(gyro, fr, fr + 1, n, seed=113, noise=0.0, outliers=0.0))
    ex, ch = _two(seed=113, max_outer_iters=60)
    _fill((ex, ch), gyro, frames)
    for d0 in (0.0355, 0.0372):
        r1, r2 = ex.Sync(d0, 0, F - 1, 0.0, 0.1), ch.Sync(d0, 0, F - 1, 0.0, 0.1)
        np.testing.assert_array_equal(_bits(ex.sync_trace()), _bits(ch.sync_trace()))
        assert r1 == r2
    assert ex.executor_stats()["runs"] == 2


@pytest.mark.parametrize("fs,F", [(4400.0, 100), (5000.0, 128), (6000.0, 120), (6400.0, 128)])
def test_large_windows_with_compact_fp64_windows_equal_the_chain(built, fs, F):
    """at 4.3-6.5 kHz a 130-track frame's two ends span 100-140 knots, the one-wave class's fp64 window is
    COMPACT (64 bytes per knot: 7-9 KB), and the executor's per-wave LDS region had shrunk with it -- below the 10 KB
    exec_window_sums stages for a window of ~90-128 frames in its trial phase (10 rows x slots doubles): stores past the
    workgroup's LDS are dropped, loads return 0, the line search decides on wrong sums.  No test reached it (compact tests
    had 14 frames, measurements 61-slot windows).  One window of 100-128 frames of 130 tracks, executor against the chain of
    launches: every trace row, delay and cost the same bits, and the executor really ran with a compact window."""
    from rssync_amd import synth
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, fs=fs, seed=77)
    frames = list(synth.make_frames(gyro, 0, F, 130, seed=77))
    ex, ch = _two(seed=77, max_outer_iters=12)
    _fill((ex, ch), gyro, frames)
    for d0 in (0.0362, 0.0375):
        r1, r2 = ex.Sync(d0, 0, F - 1, 0.0, 0.1), ch.Sync(d0, 0, F - 1, 0.0, 0.1)
        np.testing.assert_array_equal(_bits(ex.sync_trace()), _bits(ch.sync_trace()))
        assert r1 == r2
    assert ex.executor_stats()["runs"] == 2
    info = ex.window_info()
    assert info["fp64_window_compact"] and 96 < info["fp64_window_knots"] <= 208, info     # (the regime the finding is about)


def test_one_call_in_n_is_verified_in_production():
    """RSSYNC_EXECUTOR_CHECK is a debug mode (every call twice).  Without it the product still re-runs ONE executor call
    in N (default 256, here 3) through the launch chain and panics on a difference -- a tripwire for the cross-workgroup
    hand-off protocol (measured, not guaranteed: DESIGN.md section 4) at < 1 % of the executor's time.  The count is
    process-wide, so of any 6 consecutive calls with N = 3 exactly 2 are verified, whichever object makes them."""
    import rssync_amd
    from rssync_amd import synth
    if os.environ.get("RSSYNC_EXECUTOR_CHECK", "0") not in ("", "0"):
        pytest.skip("the suite is being run in the check mode itself: every call is verified, there is no sample to count")
    F, N = 40, 130
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=15)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=15))
    checked = runs = 0
    for _ in range(2):                 # two objects: the sample is over the process, not per object
        p = rssync_amd.SyncProblem(seed=35, max_outer_iters=40)
        _fill((p,), gyro, frames)
        p.set_executor_check_every(3)
        for call in range(3):
            p.Sync(0.036, 0, 30, 0.0, 0.1)
        st = p.executor_stats()
        runs += st["runs"]
        checked += st["checked"]
    assert runs == 6 and checked == 2, (runs, checked)
    q = rssync_amd.SyncProblem(seed=35, max_outer_iters=40)
    _fill((q,), gyro, frames)
    q.set_executor_check_every(0)      # off
    for call in range(4):
        q.Sync(0.036, 0, 30, 0.0, 0.1)
    assert q.executor_stats()["checked"] == 0


def test_a_mismatch_in_the_production_sample_is_reported_not_fatal(monkeypatch, capfd):
    """the production tripwire (one executor call in N re-run through the launch chain) used to PANIC on a
    difference -- a sporadic, unreproducible failure for whoever hit it.  Now such a call says so on stderr, returns the
    CHAIN's results, counts the event (executor_stats()['mismatches']) and the object keeps to the chain; only the check MODE
    (RSSYNC_EXECUTOR_CHECK=1), which exists to find a difference, still panics.  The difference is injected
    (RSSYNC_EXECUTOR_INJECT_MISMATCH=1: the executor's first delay off by one ulp, as a stale hand-off could leave it)."""
    import rssync_amd
    from rssync_amd import synth
    if os.environ.get("RSSYNC_EXECUTOR_CHECK", "0") not in ("", "0"):
        pytest.skip("the suite is being run in the check mode itself")
    F, N = 40, 130
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=15)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=15))
    ex, ch = _two(seed=35, max_outer_iters=40)
    _fill((ex, ch), gyro, frames)
    want = ch.Sync(0.036, 0, 30, 0.0, 0.1)
    ex.set_executor_check_every(1)
    monkeypatch.setenv("RSSYNC_EXECUTOR_INJECT_MISMATCH", "1")
    capfd.readouterr()
    got = ex.Sync(0.036, 0, 30, 0.0, 0.1)
    monkeypatch.delenv("RSSYNC_EXECUTOR_INJECT_MISMATCH")
    err = capfd.readouterr().err
    assert got == want                                            # the chain's result, not the (wrong) executor's
    assert "differ from the launch chain's" in err and "uses the chain from now on" in err
    st = ex.executor_stats()
    assert st["mismatches"] == 1 and st["checked"] == 1 and st["runs"] == 1
    np.testing.assert_array_equal(_bits(ex.sync_trace()), _bits(ch.sync_trace()))
    assert ex.Sync(0.0362, 0, 30, 0.0, 0.1) == ch.Sync(0.0362, 0, 30, 0.0, 0.1)
    assert ex.executor_stats()["runs"] == 1                        # the object stayed with the chain
    # the check MODE still panics on the same evidence
    q = rssync_amd.SyncProblem(seed=35, max_outer_iters=40)
    _fill((q,), gyro, frames)
    q.set_executor_check(True)
    monkeypatch.setenv("RSSYNC_EXECUTOR_INJECT_MISMATCH", "1")
    with pytest.raises(rssync_amd.RsSyncError, match="differ from the launch chain"):
        q.Sync(0.036, 0, 30, 0.0, 0.1)
