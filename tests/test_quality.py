"""Result path + accuracy metric of the reference harness (rs-sync_amd/quality.py; reference
core_testcode.cpp:270-272,297-300,315 and python/plot_sync.py:19-20,46)."""
import numpy as np
import pytest


def test_linear_fit_rmse_is_plot_sync_metric(built):
    import scipy.stats as st
    from rssync_amd import quality
    rng = np.random.default_rng(3)
    x = np.arange(0, 3000, 30.0)
    y = 37.0 + 0.004 * x + 0.2 * rng.normal(size=x.size)
    slope, intercept, rmse = quality.linear_fit_rmse(x, y)
    r = st.linregress(x, y)                       # plot_sync.py:19
    ndata = r.intercept + r.slope * x             # :20
    assert slope == pytest.approx(r.slope, rel=1e-12) and intercept == pytest.approx(r.intercept, rel=1e-12)
    assert rmse == pytest.approx(np.std(ndata - y), rel=1e-9)   # :46
    with pytest.raises(ValueError):
        quality.linear_fit_rmse([5], [1.0])


def test_csv_round_trip_and_auto_sync_points(built, tmp_path):
    from rssync_amd import quality
    pos = quality.sync_points_auto(90, 400, 60, 30)   # for (pos = 90; pos + 60 < 400; pos += 30)
    assert pos[0] == 90 and pos[-1] + 60 < 400 <= pos[-1] + 30 + 60
    delays = 0.037 + 1e-5 * np.arange(len(pos))
    quality.write_sync_csv(tmp_path / "sync.csv", pos, delays)
    p2, d2 = quality.read_sync_csv(tmp_path / "sync.csv")
    np.testing.assert_array_equal(p2, pos)
    np.testing.assert_allclose(d2, 1000 * delays, rtol=1e-8)
    quality.write_debug_csv(tmp_path / "debug.csv", [0.0, 0.1], [3.0, 2.0])
    assert (tmp_path / "debug.csv").read_text().splitlines() == ["0,3", "0.1,2"]


def test_drift_changes_only_the_true_delay(built):
    from rssync_amd import synth
    g = synth.make_gyro(0.0, 1.0, seed=2)
    a = list(synth.make_frames(g, 0, 4, 16, seed=2))
    b = list(synth.make_frames(g, 0, 4, 16, seed=2, drift=0.0))
    c = list(synth.make_frames(g, 0, 4, 16, seed=2, drift=1e-3))
    for x, y, z in zip(a, b, c):
        for u, v in zip(x[1:], y[1:]):
            np.testing.assert_array_equal(u, v)
        np.testing.assert_array_equal(x[1], z[1])          # same row times and current-frame rays
        np.testing.assert_array_equal(x[3], z[3])
        assert np.abs(x[4] - z[4]).max() > 0               # next-frame rays moved with the delay


@pytest.mark.gpu
def test_clock_drift_scenario_hip_vs_oracle(built):
    """Noise-free scene whose true delay drifts by 0.2 ms/s: per sync point the HIP delay is within
    1e-4 s (north star) of the oracle's and of the truth at the window centre; the fitted drift
    and the RMSE of the fit agree."""
    import os
    import rssync_amd
    from rssync_amd import synth, quality
    from oracle.oracle import OracleProblem
    F, N, window, dist, drift, seed = 300, 130, 60, 30, 2e-4, 0x5EED0007
    g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=seed)
    frames = list(synth.make_frames(g, 0, F, N, seed=seed, drift=drift, noise=0.0, outliers=0.0))
    pos = quality.sync_points_auto(0, F, window, dist)

    def fill(p):
        p.SetGyroQuaternions(g.quats, g.fs, g.t0)
        for fr in frames:
            p.SetTrackResult(*fr)
        return p
    h = fill(rssync_amd.SyncProblem(seed=seed))
    _, dh = h.sync_points(pos, window, 0.0, 0.001, 0.1)
    o = fill(OracleProblem(seed=seed, threads=os.cpu_count() or 1, faithful=False))
    do = []
    for p0 in pos:
        d = o.PreSync(0.0, p0, p0 + window, 0.001, 0.1)[1]
        for _ in range(4):
            d = o.Sync(d, p0, p0 + window, 0.0, 0.1)[1]
        do.append(d)
    do = np.array(do)
    truth = synth.D_TRUE + drift * (np.array(pos) + window / 2) / synth.FPS
    assert np.abs(dh - do).max() < 1e-4 and np.abs(dh - truth).max() < 1e-4
    sh, ih, rh = quality.linear_fit_rmse(pos, 1e3 * dh)
    so, io, ro = quality.linear_fit_rmse(pos, 1e3 * do)
    want_slope = 1e3 * drift / synth.FPS                    # ms of delay per frame
    assert sh == pytest.approx(want_slope, rel=0.02) and sh == pytest.approx(so, rel=2e-3)
    assert rh < 0.05 and abs(rh - ro) < 5e-3                # ms
