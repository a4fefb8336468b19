"""the driver steps upstream of the path behind the library --
rssync_ext_set_track_pixels (undistort + normalise + row time on the device) and
rssync_ext_set_gyro_rates (integration to orientations) -- against the oracle's restatement of
core_testcode.cpp:36-52,63-95,135-158.  The CPU half runs the host solver on the test double
(same lens_math.hpp text the kernel inlines); the GPU half runs the kernel."""
import os

import numpy as np
import pytest

SEED = 321


def _scene(F=12, N=96, noise_px=0.3, outliers=0.1, seed=9):
    from rssync_amd import synth
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=seed)
    frames = list(synth.make_pixel_frames(gyro, 0, F, N, seed=seed, noise_px=noise_px, outliers=outliers))
    return gyro, frames


def _feed_pixels(p, gyro, frames):
    from rssync_amd import synth
    p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr, ta, tb, pa, pb in frames:
        p.set_track_pixels(fr, ta, tb, pa, pb, synth.LENS, synth.IMAGE_ROWS)
    return p


def _feed_oracle_tracks(p, gyro, frames):
    """the reference driver's way: undistort on the host (here: the oracle), then SetTrackResult"""
    from oracle import oracle
    from rssync_amd import synth
    p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr, ta, tb, pa, pb in frames:
        p.SetTrackResult(fr, *oracle.pixels_to_tracks(synth.LENS, ta, tb, synth.IMAGE_ROWS, pa, pb))
    return p


def _check_pixel_path(make):
    gyro, frames = _scene()
    hp = _feed_pixels(make(), gyro, frames)          # pixels in, device makes the rays
    hr = _feed_oracle_tracks(make(), gyro, frames)   # oracle makes the rays, host packs them
    total = diff = 0
    for fr, *_ in frames:
        a1, b1 = hp.frame_rays(fr)
        a2, b2 = hr.frame_rays(fr)
        # unit-vector components: at most one fp32 ulp of 1 apart (different libm behind tan/cos,
        # division vs rounding order), and bit-identical in the vast majority
        assert np.abs(a1 - a2).max() <= 1.2e-7 and np.abs(b1[:, :2] - b2[:, :2]).max() <= 1.2e-7
        # knot offsets: a few tens of knots, same fp64 operations -> identical
        np.testing.assert_array_equal(b1[:, 2:], b2[:, 2:])
        total += a1.size + b1.size
        diff += int((a1 != a2).sum() + (b1 != b2).sum())
    assert diff <= 0.01 * total
    F = len(frames)
    c1, d1 = hp.PreSync(0.0, 0, F, 0.002, 0.1)
    c2, d2 = hr.PreSync(0.0, 0, F, 0.002, 0.1)
    assert d1 == d2 and c1 == pytest.approx(c2, rel=1e-5)
    return hp


def test_pixel_frames_on_the_host_solver(hosttest_lib, built):
    import rssync_amd
    hp = _check_pixel_path(lambda: rssync_amd.SyncProblem(seed=SEED, _lib=hosttest_lib))
    # a pixel frame replaces a ray frame and vice versa; bad input panics with a reason
    from rssync_amd import synth
    with pytest.raises(rssync_amd.RsSyncError, match="non-finite"):
        hp.set_track_pixels(3, 0.1, 0.13, [[np.nan, 1.0], [2.0, 3.0]], [[1.0, 1.0], [2.0, 3.0]], synth.LENS, 1520)
    with pytest.raises(rssync_amd.RsSyncError, match="lens or frame"):
        hp.set_track_pixels(3, 0.1, 0.13, [[5.0, 1.0], [2.0, 3.0]], [[1.0, 1.0], [2.0, 3.0]], synth.LENS, 0.0)
    with pytest.raises(ValueError):
        hp.set_track_pixels(3, 0.1, 0.13, [[5.0, 1.0]], [[1.0, 1.0], [2.0, 3.0]], synth.LENS, 1520)
    bad_lens = list(synth.LENS)
    bad_lens[1] = 0.0   # fx = 0: x_ = inf -> non-finite rays, reported when the frames are packed
    hp.set_track_pixels(3, 0.1, 0.13, [[5.0, 1.0], [2.0, 3.0]], [[1.0, 1.0], [2.0, 3.0]], bad_lens, 1520)
    with pytest.raises(rssync_amd.RsSyncError, match="non-finite numbers in rays"):
        hp.PreSync(0.0, 0, 12, 0.002, 0.1)


def _check_gyro_rates(make):
    from oracle import oracle
    from oracle.oracle import OracleProblem
    from rssync_amd import synth
    gyro = synth.make_gyro(0.0, 1.0, seed=4, margin=0.2)
    rng = np.random.default_rng(4)
    n = gyro.rates.shape[0]
    t = 5.0 + np.cumsum(rng.uniform(0.9, 1.1, size=n)) / gyro.fs
    for orient in (None, "XYZ", "zXy"):
        h = make()
        h.set_gyro_rates(t, gyro.rates, orient)
        o = OracleProblem(seed=SEED)
        q, us = oracle.integrate_gyro(t, gyro.rates, orient)
        o.SetGyroQuaternionsTimestamped(us, q)
        knots_h = h.gyro_knots()
        assert h.gyro_info() == o.gyro_info()
        # sequential fp64 products on both sides; libm sin/cos may differ in the last bit
        assert np.abs(knots_h - o.gyro_knots()).max() < 1e-13
    with pytest.raises(Exception, match="orientation"):
        make().set_gyro_rates(t, gyro.rates, "XYW")


def test_gyro_rates_on_the_host_solver(hosttest_lib, built):
    import rssync_amd
    _check_gyro_rates(lambda: rssync_amd.SyncProblem(seed=SEED, _lib=hosttest_lib))


@pytest.mark.gpu
def test_pixel_frames_on_the_device(built):
    import rssync_amd
    _check_pixel_path(lambda: rssync_amd.SyncProblem(seed=SEED))


@pytest.mark.gpu
def test_gyro_rates_on_the_device_build(built):
    import rssync_amd
    _check_gyro_rates(lambda: rssync_amd.SyncProblem(seed=SEED))


@pytest.mark.gpu
def test_pixels_end_to_end_recover_the_true_delay(built):
    """noise-free pixel scene: rates -> orientations and pixels -> rays inside the library, then
    PreSync + Sync: the true delay within 1e-4 s, and the oracle fed the driver's way agrees"""
    import rssync_amd
    from oracle.oracle import OracleProblem
    from rssync_amd import synth
    gyro, frames = _scene(F=40, N=160, noise_px=0.0, outliers=0.0, seed=13)
    h = _feed_pixels(rssync_amd.SyncProblem(seed=SEED), gyro, frames)
    o = _feed_oracle_tracks(OracleProblem(seed=SEED, threads=os.cpu_count() or 1, faithful=False), gyro, frames)
    F = len(frames)
    dh = h.PreSync(0.0, 0, F, 0.002, 0.1)[1]
    do = o.PreSync(0.0, 0, F, 0.002, 0.1)[1]
    assert dh == do
    ch, dh = h.Sync(dh, 0, F - 1, 0.0, 0.2)
    co, do = o.Sync(do, 0, F - 1, 0.0, 0.2)
    assert abs(dh - synth.D_TRUE) < 1e-4 and abs(dh - do) < 1e-4


def _check_orientation_sweep(make, F, N, margin, tol=0.004, **scene):
    """rssync_ext_orientation_sweep == set_gyro_rates + PreSync per orientation (exactly), and the
    orientation the rays were generated with has the lowest cost (core_testcode.cpp:186-232).  Round 6: the sweep is ONE
    pipeline (every orientation enqueued back to back, one wait); RSSYNC_SWEEP_PIPELINE=0 keeps rounds 1-5's loop -- the
    same numbers either way."""
    from rssync_amd import synth
    g = synth.make_gyro(1.0, 1.0 + (F + 2) / synth.FPS, seed=77)   # t0 = 0: timestamps must be >= 0
    frames = list(synth.make_frames(g, 30, 30 + F, N, seed=77, **scene))
    names = list(synth.ORIENTATIONS[:9]) + ["XYZ"]
    seq, bat = make(), make()
    for p in (seq, bat):
        for fr in frames:
            p.SetTrackResult(*fr)
    want = []
    for name in names:
        seq.set_gyro_rates(g.times, g.rates, name)
        want.append(seq.PreSync(0.0, 30, 30 + F, 0.004, 0.1))
    costs, delays = bat.orientation_sweep(g.times, g.rates, names, 0.0, 30, 30 + F, 0.004, 0.1)
    for i in range(len(names)):
        assert (costs[i], delays[i]) == want[i]
    order = np.argsort(costs)
    assert names[order[0]] == "XYZ" and costs[order[0]] < margin * costs[order[1]]
    assert abs(delays[order[0]] - synth.D_TRUE) <= tol                  # grid step 4 ms
    np.testing.assert_array_equal(bat.gyro_knots(), seq.gyro_knots())   # the last orientation stays installed
    os.environ["RSSYNC_SWEEP_PIPELINE"] = "0"
    try:
        costs0, delays0 = bat.orientation_sweep(g.times, g.rates, names, 0.0, 30, 30 + F, 0.004, 0.1)
    finally:
        del os.environ["RSSYNC_SWEEP_PIPELINE"]
    np.testing.assert_array_equal(costs0, costs)
    np.testing.assert_array_equal(delays0, delays)
    with pytest.raises(Exception, match="orientation"):
        bat.orientation_sweep(g.times, g.rates, ["XYZ", "abc"], 0.0, 30, 30 + F, 0.004, 0.1)


def test_orientation_sweep_on_the_host_solver(hosttest_lib, built):
    import rssync_amd
    _check_orientation_sweep(lambda: rssync_amd.SyncProblem(seed=SEED, _lib=hosttest_lib), 10, 64, 0.97, tol=0.0081)


@pytest.mark.gpu
def test_orientation_sweep_on_the_device(built):
    import rssync_amd
    _check_orientation_sweep(lambda: rssync_amd.SyncProblem(seed=SEED), 24, 256, 0.95)


@pytest.mark.gpu
def test_orientation_sweep_with_near_static_frames(built):
    """the pipeline cannot give an orientation's near-static pairs their fp64 rows (that needs the host between two launches,
    kernels/lmeds.hpp "fp64 rows"): such an orientation -- here the true one, around the true delay -- is flagged and repeated
    on its own, and the sweep still equals the per-orientation calls exactly; the last orientation stays installed"""
    import rssync_amd
    made = []

    def make():
        made.append(rssync_amd.SyncProblem(seed=SEED))
        return made[-1]
    _check_orientation_sweep(make, 16, 200, 1.0, tol=0.0081, translation=5e-5, noise=1e-6)
    assert made[1].near_static_stats()["pairs"] > 0


def test_a_frame_can_switch_between_rays_and_pixels(hosttest_lib, built):
    """SetTrackResult on a frame that was given as pixels (and the reverse) replaces it completely
    (core_private.cpp:194: re-setting a frame overwrites it)."""
    import rssync_amd
    from oracle import oracle
    from rssync_amd import synth
    gyro, frames = _scene(F=6, N=40)
    p = rssync_amd.SyncProblem(seed=SEED, _lib=hosttest_lib)
    p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr, ta, tb, pa, pb in frames:
        p.set_track_pixels(fr, ta, tb, pa, pb, synth.LENS, synth.IMAGE_ROWS)
    base = p.PreSync(0.0, 0, 6, 0.004, 0.05)
    fr, ta, tb, pa, pb = frames[2]
    tracks = oracle.pixels_to_tracks(synth.LENS, ta, tb, synth.IMAGE_ROWS, pa, pb)
    p.SetTrackResult(fr, *tracks)                      # same content, now as rays
    a_rays = p.frame_rays(fr)
    again = p.PreSync(0.0, 0, 6, 0.004, 0.05)
    assert again[1] == base[1] and again[0] == pytest.approx(base[0], rel=1e-6)
    p.SetTrackResult(fr, *[x[:10] for x in tracks])    # fewer tracks: the old ones are gone
    assert p.frame_rays(fr)[0].shape == (10, 4)
    p.set_track_pixels(fr, ta, tb, pa, pb, synth.LENS, synth.IMAGE_ROWS)   # and back to pixels
    a_pix = p.frame_rays(fr)
    assert a_pix[0].shape == (40, 4) and np.abs(a_pix[0] - a_rays[0]).max() <= 1.2e-7
    assert p.PreSync(0.0, 0, 6, 0.004, 0.05) == base
