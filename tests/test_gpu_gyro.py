"""The gyro side on the device: angular rates -> orientations (scan of quaternion
products), orientations -> integer-microsecond grid (slerp), grid -> natural-spline table (two short-memory
sweeps cut into parallel runs).  Reference: core_testcode.cpp:36-52, core_private.cpp:142-190,
minispline.cpp:3-46.  The oracle does each of them one sample after the other."""
import numpy as np
import pytest

import rssync_amd
from rssync_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


def thomas_table(knots):
    """the sequential solve the device's parallel runs must reproduce (gyro_math.hpp's formulas, in order)"""
    n = knots.shape[0]
    cp = np.zeros(n)
    for i in range(1, n):
        cp[i] = (1.0 / 3.0) / (4.0 / 3.0 - cp[i - 1] / 3.0)
    out = np.zeros((n, 4, 4))
    for comp in range(4):
        y = knots[:, comp]
        cf = np.zeros(n)
        for i in range(1, n - 1):
            rhs = y[i + 1] - 2.0 * y[i] + y[i - 1]
            cf[i] = (rhs - cf[i - 1] / 3.0) / (4.0 / 3.0 - cp[i - 1] / 3.0)
        c = np.zeros(n)
        for i in range(n - 2, 0, -1):
            c[i] = cf[i] - cp[i] * c[i + 1]
        b, d = np.zeros(n), np.zeros(n)
        d[:-1] = (c[1:] - c[:-1]) / 3.0
        b[:-1] = (y[1:] - y[:-1]) - (2.0 * c[:-1] + c[1:]) / 3.0
        b[-1] = 3.0 * d[-2] + 2.0 * c[-2] + b[-2]
        out[:, 0, comp], out[:, 1, comp], out[:, 2, comp], out[:, 3, comp] = y, b, c, d
    return out


@pytest.mark.parametrize("n", [2, 3, 4, 31, 32, 33, 64, 65, 97, 130, 1000, 4099])
def test_spline_table_of_parallel_runs_is_the_sequential_solve(n):
    """run length 32, warm-up 64: every boundary case, and the bits of the sequential recurrence"""
    rng = np.random.default_rng(n)
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    h = rssync_amd.SyncProblem(verbose=False)
    h.SetGyroQuaternions(q, 400.0, 0.0)
    got = h.gyro_table()
    want = thomas_table(q)
    np.testing.assert_array_equal(got[:, 0], q)
    np.testing.assert_allclose(got, want, rtol=0, atol=4e-15)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {got.size} entries differ in the last bits"


def test_spline_table_matches_the_oracle_spline():
    g = synth.make_gyro(0.0, 3.0, seed=5)
    h = rssync_amd.SyncProblem(verbose=False)
    o = oracle.OracleProblem()
    h.SetGyroQuaternions(g.quats, g.fs, g.t0)
    o.SetGyroQuaternions(g.quats, g.fs, g.t0)
    tab = h.gyro_table()
    n = tab.shape[0]
    # value and derivative of the oracle's spline at knots and mid-points, inside and outside the span
    for x in [0.0, 0.5, 1.0, 17.25, n - 2.5, n - 1.0, -0.75, n + 1.5]:
        i = int(min(max(np.floor(x), 0), n - 1))
        t = x - i
        y, b, c, d = tab[i]
        if x < 0:
            val, der = y + b * x, b
        elif x > n - 1:
            val, der = y + b * (x - n), b   # minispline.cpp:53: index clamped to n, so h = x - n beyond the last interval
        else:
            val, der = y + t * (b + t * (c + t * d)), b + t * (2 * c + 3 * t * d)
        np.testing.assert_allclose(val, o.spline_eval(x), rtol=0, atol=1e-14, err_msg=f"x={x}")
        np.testing.assert_allclose(der, o.spline_deriv(x), rtol=0, atol=1e-13)


def test_timestamped_route_grid_exact_and_knots_to_the_last_bits():
    g = synth.make_gyro(1.0, 4.0, seed=21)
    rng = np.random.default_rng(3)
    ts = (np.arange(len(g.quats)) * (1e6 / g.fs) + rng.integers(-300, 300, len(g.quats))).astype(np.int64)
    ts = np.sort(np.abs(ts))
    h = rssync_amd.SyncProblem(verbose=False)
    o = oracle.OracleProblem()
    h.SetGyroQuaternionsTimestamped(ts, g.quats)
    o.SetGyroQuaternionsTimestamped(ts, g.quats)
    assert h.gyro_info() == o.gyro_info()  # rate, first knot time, knot count: integer arithmetic
    hk, ok = h.gyro_knots(), o.gyro_knots()
    # slerp: acos and sin of the device's math library against the host's, <= a few ulp of 1
    np.testing.assert_allclose(hk, ok, rtol=0, atol=1e-15)
    np.testing.assert_array_equal(h.gyro_table(), thomas_table(hk))


def test_rates_route_scan_matches_sequential_integration():
    """55k samples (a 2-minute 400 Hz log + jitter): 1024 chunks scanned in parallel vs one product after the other"""
    rng = np.random.default_rng(11)
    n = 55000
    t = 5.0 + np.cumsum(rng.uniform(0.0023, 0.0027, n))
    r = 0.6 * rng.standard_normal((n, 3)) + np.array([0.3, -0.2, 0.1])
    for orientation in [None, "XYZ", "yXz", "ZxY"]:
        q, ts_us = oracle.integrate_gyro(t, r, orientation)
        h = rssync_amd.SyncProblem(verbose=False)
        o = oracle.OracleProblem()
        h.set_gyro_rates(t, r, orientation)
        o.SetGyroQuaternionsTimestamped(ts_us, q)
        assert h.gyro_info() == o.gyro_info()
        # rounding of 55k products accumulates as a random walk: ~1e-16 * sqrt(n)
        np.testing.assert_allclose(h.gyro_knots(), o.gyro_knots(), rtol=0, atol=2e-13)


def test_rates_route_short_and_ragged_inputs():
    rng = np.random.default_rng(2)
    for n in [3, 5, 1023, 1024, 1025, 2049]:
        t = 1.0 + np.arange(n) * 0.0025
        r = rng.standard_normal((n, 3))
        q, ts_us = oracle.integrate_gyro(t, r, None)
        h = rssync_amd.SyncProblem(verbose=False)
        o = oracle.OracleProblem()
        h.set_gyro_rates(t, r)
        o.SetGyroQuaternionsTimestamped(ts_us, q)
        assert h.gyro_info() == o.gyro_info()
        np.testing.assert_allclose(h.gyro_knots(), o.gyro_knots(), rtol=0, atol=1e-14)


def test_gyro_complaints_are_the_references():
    h = rssync_amd.SyncProblem(verbose=False)
    q = np.tile([1.0, 0, 0, 0], (6, 1))
    with pytest.raises(rssync_amd.RsSyncError, match=r"timestamps out of order at pos 3 \(7500 > 5000\)"):
        h.SetGyroQuaternionsTimestamped(np.array([0, 2500, 7500, 5000, 10000, 12500]), q)
    with pytest.raises(rssync_amd.RsSyncError, match="non-finite sample rate"):
        h.SetGyroQuaternionsTimestamped(np.array([0, 0, 0, 0, 0, 0]), q)
    bad = q.copy()
    bad[2, 1] = np.nan
    with pytest.raises(rssync_amd.RsSyncError, match="non-finite sample after interpolation"):
        h.SetGyroQuaternionsTimestamped(np.arange(6) * 2500, bad)
    with pytest.raises(rssync_amd.RsSyncError, match="non-finite numbers"):
        h.set_gyro_rates(np.arange(6) * 0.0025, np.full((6, 3), np.inf))
    with pytest.raises(rssync_amd.RsSyncError, match="orientation letters"):
        h.set_gyro_rates(np.arange(6) * 0.0025, np.zeros((6, 3)), "XYW")
    with pytest.raises(rssync_amd.RsSyncError, match="gyro data was not set"):
        h.gyro_table()
    h.SetGyroQuaternionsTimestamped(np.arange(6) * 2500, q)   # and the object is usable afterwards
    assert h.gyro_info()[2] >= 2


def test_device_gyro_feeds_the_same_sync_as_host_integrated_quaternions():
    """rates -> (device) -> PreSync/Sync against quaternions integrated by the oracle -> timestamped setter"""
    F, N = 32, 128
    g = synth.make_gyro(1.0, 1.0 + (F + 2) / synth.FPS, seed=77)   # t0 = 0: timestamps must be >= 0
    q, ts_us = oracle.integrate_gyro(g.times, g.rates, None)
    a, b = rssync_amd.SyncProblem(seed=5, verbose=False), rssync_amd.SyncProblem(seed=5, verbose=False)
    a.set_gyro_rates(g.times, g.rates)
    b.SetGyroQuaternionsTimestamped(ts_us, q)
    for fr in synth.make_frames(g, 30, 30 + F, N, seed=9, noise=0.0, outliers=0.0):  # clean: Sync is not chaotic
        a.SetTrackResult(*fr)
        b.SetTrackResult(*fr)
    ca, da = a.PreSync(0.0, 30, 30 + F, 0.001, 0.05)
    cb, db = b.PreSync(0.0, 30, 30 + F, 0.001, 0.05)
    assert da == db and abs(ca - cb) <= 1e-5 * cb
    sa, sb = a.Sync(da, 30, 30 + F, 0.0, 0.1), b.Sync(db, 30, 30 + F, 0.0, 0.1)
    assert abs(sa[1] - sb[1]) < 1e-8


def test_random_recordings_grid_exact_knots_close():
    """the closed-form grid of the device route against the oracle's loop on random recordings (rates 47 Hz .. 3.2 kHz,
    jittered sample times, random first timestamp): rate, first knot time and count exact, knots to the last bits"""
    rng = np.random.default_rng(78)
    for case in range(60):
        n = int(rng.integers(2, 3000))
        rate = float(rng.choice([47.0, 50.0, 99.0, 200.0, 399.7, 400.0, 1000.0, 1601.0, 3200.0]))
        first = int(rng.integers(0, 3_000_000))
        ts = first + np.cumsum(np.maximum(1, np.round(1e6 / rate * rng.uniform(0.6, 1.4, n)))).astype(np.int64)
        q = rng.standard_normal((n, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        h, o = rssync_amd.SyncProblem(verbose=False), oracle.OracleProblem()
        try:
            o.SetGyroQuaternionsTimestamped(ts, q)
        except oracle.OracleError:
            with pytest.raises(rssync_amd.RsSyncError):
                h.SetGyroQuaternionsTimestamped(ts, q)
            continue
        h.SetGyroQuaternionsTimestamped(ts, q)
        assert h.gyro_info() == o.gyro_info(), (case, n, rate, first)
        np.testing.assert_allclose(h.gyro_knots(), o.gyro_knots(), rtol=0, atol=2e-15)
        np.testing.assert_array_equal(h.gyro_table(), thomas_table(h.gyro_knots()))
