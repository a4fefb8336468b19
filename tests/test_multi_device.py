"""One object, several GPUs (rssync_ext_set_devices / RSSYNC_GPUS): the frames are spread over the devices in
contiguous blocks cut at multiples of 64 frames, every device works on its block, and the host adds chunk
sums in frame order -- the association the single-device kernel uses.  Results must therefore be
BIT-IDENTICAL to the single-device run, whatever the device count (the reference parallelises over
frames inside one object, core_private.cpp:73,231,245,263).

CPU: the product's host solver on the test double with several fake devices.  GPU: several contexts on
the one device of the test box (the driver's multi-GPU node is not available to these tests)."""
import os

import numpy as np
import pytest

F, N = 200, 40


def _case(fs=400.0):
    from rssync_amd import synth
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, fs=fs, seed=11)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=11))
    # ragged: a few frames with other track counts, sparse ids at the end
    out = []
    for fr, ta, tb, ra, rb in frames:
        n = N if fr % 7 else max(2, N - fr % 13)
        out.append((fr if fr < F - 10 else fr + 1000, ta[:n], tb[:n], ra[:n], rb[:n]))
    return gyro, out


def _fill(p, case):
    g, frames = case
    p.SetGyroQuaternions(g.quats, g.fs, g.t0)
    for fr in frames:
        p.SetTrackResult(*fr)
    return p


def _run(p):
    res = {}
    res["presync"] = p.PreSync(0.0, 0, 5000, 0.004, 0.06)
    d, c, fc, bh = p.presync_curve(0.0, 10, 150, 0.01, 0.05, per_frame=140)
    res["curve"] = (c.tolist(), fc.tolist(), bh.tolist())
    res["sync"] = p.Sync(res["presync"][1], 0, 5000, 0.0, 0.2)
    res["trace"] = p.sync_trace().tolist()
    M, k = p.init_motion(0.036, 30, 170)
    res["init"] = (M.tolist(), k.tolist())
    M2, k2, its, evs = p.opt_motion(0.036)
    res["opt"] = (M2.tolist(), k2.tolist(), its, evs)
    res["loss"] = [x.tolist() for x in p.loss([0.036, 0.03, -0.1], grad=True)]
    res["windows"] = [x.tolist() for x in p.pre_sync_windows(0.03, [0, 50, 60, 120], [70, 130, 64, 1300], 0.004, 0.04)]
    res["points"] = [x.tolist() for x in p.sync_points([0, 40, 100, 130], 60, 0.03, 0.004, 0.04, repeats=2)]
    res["point_traces"] = [p.window_trace(w).tolist() for w in range(4)]
    res["simplified"] = p.SyncSimplified(0.036, 20, 180, 0.0, 0.2)
    res["debug"] = [x.tolist() for x in p.DebugPreSync(0.02, 60, 70, 0.03, 7)]
    res["P"] = p.problem_matrix64(130, 0.0371, N).tolist()
    return res


def _check(one, many):
    assert one.keys() == many.keys()
    for key in one:
        assert one[key] == many[key], key   # exact equality, floats included


def test_frames_sharded_over_fake_devices_equal_one_device(hosttest_lib):
    import rssync_amd
    case = _case()
    one = _run(_fill(rssync_amd.SyncProblem(seed=5, max_outer_iters=12, _lib=hosttest_lib), case))
    for ids in ([0, 1], [0, 1, 2], [3, 3, 3, 3, 3]):
        p = rssync_amd.SyncProblem(seed=5, max_outer_iters=12, _lib=hosttest_lib)
        p.set_devices(ids)
        assert p.device_count() == len(ids)
        _check(one, _run(_fill(p, case)))
    # devices changed after the data was set: everything is uploaded again
    p = _fill(rssync_amd.SyncProblem(seed=5, max_outer_iters=12, _lib=hosttest_lib), case)
    a = p.PreSync(0.0, 0, 5000, 0.004, 0.06)
    p.set_devices([0, 1])
    assert p.PreSync(0.0, 0, 5000, 0.004, 0.06) == a


def test_environment_variable_selects_the_devices(hosttest_lib, monkeypatch):
    import rssync_amd
    monkeypatch.setenv("RSSYNC_GPUS", "3")
    assert rssync_amd.SyncProblem(_lib=hosttest_lib).device_count() == 3
    monkeypatch.setenv("RSSYNC_GPUS", "0,0")
    assert rssync_amd.SyncProblem(_lib=hosttest_lib).device_count() == 2
    monkeypatch.delenv("RSSYNC_GPUS")
    assert rssync_amd.SyncProblem(_lib=hosttest_lib).device_count() == 1


@pytest.mark.gpu
def test_several_contexts_on_the_gpu_equal_one(tmp_path):
    """the same on the device: two and three contexts (streams, buffers, kernels of their own) on the one GPU
    of the test box, frames split between them; and the C++ client of examples/sync_driver.cpp with
    RSSYNC_GPUS set, no source change"""
    import rssync_amd
    case = _case()
    one = _run(_fill(rssync_amd.SyncProblem(seed=5, max_outer_iters=12), case))
    for ids in ([0, 0], [0, 0, 0]):
        p = rssync_amd.SyncProblem(seed=5, max_outer_iters=12)
        p.set_devices(ids)
        _check(one, _run(_fill(p, case)))


@pytest.mark.gpu
@pytest.mark.parametrize("fs", [2000.0, 4000.0])
def test_several_contexts_at_high_gyro_rates(fs):
    """Above ~1.7 kHz every context plans its own spline windows (capacity, candidates per workgroup) from the frames
    IT holds; which path a frame's rows take must not depend on that, or the sums of a sharded object would no longer be
    the single-device run's bits."""
    import rssync_amd
    case = _case(fs)
    one = _run(_fill(rssync_amd.SyncProblem(seed=5, max_outer_iters=12), case))
    p = rssync_amd.SyncProblem(seed=5, max_outer_iters=12)
    p.set_devices([0, 0, 0])
    _check(one, _run(_fill(p, case)))


def _case_mixed_spans(fs):
    """as _case, but the frames' knot spans differ: most frames keep only the tracks of the first 40 % of the read-out
    (narrow: their two ends fit a one-wave kernel's LDS window), the frames from 150 on keep all rows (at 6 kHz: two ends
    of 11 ms x 6 kHz = 66 knots each, more than the 128 knots such a window may have)"""
    from rssync_amd import synth
    gyro, frames = _case(fs)
    out = []
    for fr, ta, tb, ra, rb in frames:
        if fr < 150:
            f0 = fr if fr < 1000 else fr - 1000
            keep = (ta - f0 / synth.FPS <= 0.4 * synth.READOUT) & (tb - (f0 + 1) / synth.FPS <= 0.4 * synth.READOUT)
            if keep.sum() >= 4:
                ta, tb, ra, rb = ta[keep], tb[keep], ra[keep], rb[keep]
        out.append((fr, ta, tb, ra, rb))
    return gyro, out


@pytest.mark.gpu
def test_several_contexts_with_frames_wider_than_any_window():
    """in the cap-limited regime (one-wave kernels, a frame wider than the 128 knots their LDS window may have) a
    shard that did not hold the wide frames planned a window from its own widest frame and put an 80 .. 128-knot frame on the
    interior path, while the single-device run -- planning from the widest frame of all -- fell back to the general path
    for the same frame: other fp32 roundings, the sharded object no longer the single-device run's bits.  Now every context
    plans from the frames of the WHOLE problem (rship_set_problem_frames) and a launch's window is planned from the frames
    that can have one at all (window_plan.hpp: plan_window_frames), so that which path a frame takes follows from the
    frame.  The wide frames live in the last shard only."""
    import rssync_amd
    case = _case_mixed_spans(6000.0)
    one_p = _fill(rssync_amd.SyncProblem(seed=5, max_outer_iters=12), case)
    one = _run(one_p)
    info = one_p.window_info()
    assert info["frame_ends_knots"] > 128, info   # the widest frame's two ends: beyond a one-wave kernel's window
    p = rssync_amd.SyncProblem(seed=5, max_outer_iters=12)
    p.set_devices([0, 0, 0])
    _check(one, _run(_fill(p, case)))
    # and the narrow frames take the interior path although wide ones exist: alone in a problem they give the same bits
    g, frames = case
    narrow = [f for f in frames if f[0] < 64]
    a = _fill(rssync_amd.SyncProblem(seed=5, max_outer_iters=12), (g, narrow))
    d1, c1, fc1, bh1 = a.presync_curve(0.0, 0, 64, 0.01, 0.05, per_frame=len(narrow))
    d2, c2, fc2, bh2 = one_p.presync_curve(0.0, 0, 64, 0.01, 0.05, per_frame=len(narrow))
    np.testing.assert_array_equal(fc1, fc2)
    np.testing.assert_array_equal(bh1, bh2)


def _gyro_routes(make, device_lists):
    """the gyro routes that run on the device (timestamped samples, angular rates, orientation sweep): every
    device of the object builds the table itself, and a change of devices rebuilds it from the knots"""
    from rssync_amd import synth
    Fg = 40
    g = synth.make_gyro(1.0, 1.0 + (Fg + 2) / synth.FPS, seed=13)   # first sample at t = 0 (unsigned timestamps)
    frames = list(synth.make_frames(g, 30, 30 + Fg, 48, seed=13))
    names = ["XYZ", "yXz", "ZxY"]
    ref = None
    for ids in device_lists:
        p = make()
        if ids:
            p.set_devices(ids)
        for fr in frames:
            p.SetTrackResult(*fr)
        out = {}
        p.set_gyro_rates(g.times, g.rates, "XYZ")
        out["info"] = p.gyro_info()
        out["knots"] = p.gyro_knots().tolist()
        out["table"] = p.gyro_table().tolist()
        out["presync"] = p.PreSync(0.0, 30, 30 + Fg, 0.004, 0.06)
        out["sync"] = p.Sync(out["presync"][1], 30, 30 + Fg - 1, 0.0, 0.2)
        out["sweep"] = [x.tolist() for x in p.orientation_sweep(g.times, g.rates, names, 0.0, 30, 30 + Fg, 0.004, 0.06)]
        ts_us = np.round(g.times * 1e6).astype(np.int64)
        p.SetGyroQuaternionsTimestamped(ts_us, g.quats)
        out["ts_presync"] = p.PreSync(0.0, 30, 30 + Fg, 0.004, 0.06)
        p.set_devices([0, 0] if not ids else [0])   # the table follows the object to its new devices
        out["moved"] = p.PreSync(0.0, 30, 30 + Fg, 0.004, 0.06)
        assert out["moved"] == out["ts_presync"]
        if ref is None:
            ref = out
        else:
            _check(ref, out)


def test_device_gyro_routes_on_fake_devices(hosttest_lib):
    import rssync_amd
    _gyro_routes(lambda: rssync_amd.SyncProblem(seed=5, max_outer_iters=12, _lib=hosttest_lib), [None, [0, 1], [2, 2, 2]])


@pytest.mark.gpu
def test_device_gyro_routes_on_several_contexts():
    import rssync_amd
    _gyro_routes(lambda: rssync_amd.SyncProblem(seed=5, max_outer_iters=12), [None, [0, 0], [0, 0, 0]])
