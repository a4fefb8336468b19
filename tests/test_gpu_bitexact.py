"""Device vs DEVICE-ASSOCIATION ORACLE, bit for bit.

The CPU stand-in of the device ABI (tests/cpu_device/rship_cpu.cpp, linked with the product's own host solver)
evaluates Sync's fp64 arithmetic from the same source as the kernels (rs-sync_amd/csrc/device_math.hpp,
sync_math.hpp; contraction off on both sides, IEEE division and square root) and sums the rows in the kernels'
association: rows-per-thread partials, the row_shr / readlane wave tree, waves in order, the plan's chunks.
So every loss, gradient, motion estimate and the whole Sync trace must be IDENTICAL -- `assert_array_equal`,
no tolerance -- on noisy data with outliers, where device and reference-order oracle differ by tenths of a
millisecond (reassociation feeding a chaotic iteration: tests/measure/reassociation.py measures that part on the
CPU alone, profiles/r3_reassociation.json).

GuessMotion's hypothesis SEARCH runs in fp32 on the device with hardware reciprocals (the stand-in's fp32
search is sequential IEEE), so the two can pick different winners on near-ties (< 0.1 % of frames): the tests
start the stand-in from the device's winners (rssync_ext_set_init_override) and report how many differed.
If a building block ever differs, test_fp64_building_blocks names the operation.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_small.npz")


def _pair(hosttest_lib, **kw):
    import rssync_amd
    return rssync_amd.SyncProblem(**kw), rssync_amd.SyncProblem(_lib=hosttest_lib, **kw)


def _fill_both(gpu, cpu, gyro, frames):
    for p in (gpu, cpu):
        p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
        for fr in frames:
            p.SetTrackResult(*fr)


def _bits(x):
    return np.ascontiguousarray(x, np.float64).view(np.uint64)


def test_fp64_building_blocks(hosttest_lib):
    """division, square root, the kernels' log1p / reciprocal, fma and the 64-lane sum: same bits on the device and
    on the host, and (division, square root) the correctly rounded IEEE results numpy gives"""
    gpu, cpu = _pair(hosttest_lib)
    rng = np.random.default_rng(11)
    n = 1 << 16
    a = np.concatenate([10.0 ** rng.uniform(-300, 300, n // 2), 10.0 ** rng.uniform(-12, 4, n // 2)])
    b = np.concatenate([10.0 ** rng.uniform(-300, 300, n // 2), 1.0 + rng.uniform(0, 3, n // 2)])
    u = np.concatenate([10.0 ** rng.uniform(-30, 12, n - 8), [0.0, 1e-320, 1.0, 2.0 ** -53, 2.0 ** -52, 0.5, 1e300, 3.0]])
    s = rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 6, n)
    names = {0: "division", 1: "sqrt", 2: "log1p_rcp_f64", 3: "fma", 4: "wave_sum_f64"}
    for op, (x, y) in {0: (a, b), 1: (a, None), 2: (u, None), 3: (s, b), 4: (s, None)}.items():
        g, c = gpu.debug_math64(op, x, y), cpu.debug_math64(op, x, y)
        bad = np.flatnonzero(_bits(g).ravel() != _bits(c).ravel())
        assert bad.size == 0, "%s differs between device and host in %d of %d values (first at %d: %r vs %r)" % (
            names[op], bad.size, g.size, bad[0], g.ravel()[bad[0]], c.ravel()[bad[0]])
    with np.errstate(over="ignore", under="ignore"):
        np.testing.assert_array_equal(gpu.debug_math64(0, a, b), a / b)
    np.testing.assert_array_equal(gpu.debug_math64(1, a), np.sqrt(a))
    # x / 3 in three instructions (rs::div3_exact: the compact spline windows rebuild b and d of a knot with it) IS the
    # correctly rounded division, on the device and on the host, over the whole range of normal quotients
    t = np.concatenate([s, a[: n // 2], -a[n // 2:], [3.0, 1.0, 2.0 ** -1000, 2.0 ** 1000, 1.0 + 2.0 ** -52, 0.0]])
    np.testing.assert_array_equal(_bits(gpu.debug_math64(5, t)), _bits(t / 3.0))
    np.testing.assert_array_equal(_bits(cpu.debug_math64(5, t)), _bits(t / 3.0))
    # and the routine is log1p to a few ulp
    lg = gpu.debug_math64(2, u)
    np.testing.assert_allclose(lg[:, 0], np.log1p(u), rtol=1e-15, atol=0)
    np.testing.assert_allclose(lg[:, 1], 1.0 / (1.0 + u), rtol=5e-16, atol=0)


def _scene(F, N, seed, **kw):
    from rssync_amd import synth
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=seed)
    return gyro, list(synth.make_frames(gyro, 0, F, N, seed=seed, **kw))


@pytest.mark.parametrize("F,N", [(12, 130), (10, 600), (6, 2048), (7, 257), (3, 9000)])
def test_every_evaluation_is_bit_identical(hosttest_lib, F, N):
    """per evaluation: fp64 rows, GuessMotion's M / GuessK, the loss and its analytic derivative at several delays,
    one motion optimisation (end points, iteration and evaluation counts) -- noisy scene, 10 % outliers"""
    gyro, frames = _scene(F, N, seed=40 + N)
    gpu, cpu = _pair(hosttest_lib, seed=77, max_outer_iters=20)
    _fill_both(gpu, cpu, gyro, frames)
    d0 = 0.0352
    for fr in (0, F - 1):
        Pg, dPg = gpu.problem_matrix64(fr, d0, N, deriv=True)
        Pc, dPc = cpu.problem_matrix64(fr, d0, N, deriv=True)
        np.testing.assert_array_equal(_bits(Pg), _bits(Pc))
        np.testing.assert_array_equal(_bits(dPg), _bits(dPc))
    gpu.record_init_winners()
    Mg, kg = gpu.init_motion(d0, 0, F - 1)
    win = gpu.last_init_winners()
    # the stand-in's own fp32 search (a second instance: every call advances the sampler's call counter) ...
    import rssync_amd
    own = rssync_amd.SyncProblem(seed=77, max_outer_iters=20, _lib=hosttest_lib)
    _fill_both(own, own, gyro, frames)
    own.record_init_winners()
    own.init_motion(d0, 0, F - 1)
    n_diff = int(np.sum(own.last_init_winners() != win))
    cpu.set_init_override(win)                     # ... and the stand-in started from the device's winners
    Mc, kc = cpu.init_motion(d0, 0, F - 1)
    np.testing.assert_array_equal(_bits(Mg), _bits(Mc))
    np.testing.assert_array_equal(_bits(kg), _bits(kc))
    assert n_diff <= max(1, F // 10), "fp32 searches disagree on %d of %d frames" % (n_diff, F)
    delays = np.array([d0, d0 - 3e-4, 0.0371, 0.05, d0 + 1e-9])
    lg, gg = gpu.loss(delays, grad=True)
    lc, gc = cpu.loss(delays, grad=True)
    np.testing.assert_array_equal(_bits(lg), _bits(lc))
    np.testing.assert_array_equal(_bits(gg), _bits(gc))
    np.testing.assert_array_equal(_bits(gpu.loss(delays)), _bits(cpu.loss(delays)))  # the five-delay batch kernel
    M2g, k2g, itg, evg = gpu.opt_motion(d0)
    M2c, k2c, itc, evc = cpu.opt_motion(d0)
    assert (itg, evg) == (itc, evc)
    np.testing.assert_array_equal(_bits(M2g), _bits(M2c))
    np.testing.assert_array_equal(_bits(gpu.loss(delays, grad=True)[1]), _bits(cpu.loss(delays, grad=True)[1]))


def _sync_both(gpu, cpu, call):
    gpu.record_init_winners()
    rg = call(gpu)
    cpu.set_init_override(gpu.last_init_winners())
    rc = call(cpu)
    return rg, rc


def test_noisy_sync_trace_is_bit_identical(hosttest_lib, small_case):
    """BASELINE config 1 (64 x 256, noise 1e-3 rad, 10 % outliers): every row of the Sync trace -- delay after the
    step, step, loss and derivative at the look-ahead point, accepted step length, trials -- and the returned
    (cost, delay), device loop against the stand-in's host loop"""
    gpu, cpu = _pair(hosttest_lib, seed=123, max_outer_iters=30)
    _fill_both(gpu, cpu, small_case["gyro"], small_case["frames"])
    F = small_case["F"]
    d0 = gpu.PreSync(0.0, 0, F, 0.002, 0.1)[1]
    rg, rc = _sync_both(gpu, cpu, lambda p: p.Sync(d0, 0, F - 1, 0.0, 0.1))
    tg, tc = gpu.sync_trace(), cpu.sync_trace()
    assert tg.shape == tc.shape and len(tg) >= 6
    np.testing.assert_array_equal(_bits(tg), _bits(tc))
    assert rg == rc
    # chained calls, as the reference driver makes them (core_testcode.cpp:314): the second starts where the first ended
    rg2, rc2 = _sync_both(gpu, cpu, lambda p: p.Sync(rg[1], 0, F - 1, 0.0, 0.1))
    np.testing.assert_array_equal(_bits(gpu.sync_trace()), _bits(cpu.sync_trace()))
    assert rg2 == rc2


def test_noisy_golden_fixture_trace_is_bit_identical(hosttest_lib):
    """the committed noisy fixture (24 x 128): device == device-order stand-in exactly; both within the
    reassociation scatter of the reference-order oracle's stored trace (see test_golden.py for that comparison)"""
    g = np.load(GOLD)
    gpu, cpu = _pair(hosttest_lib, seed=int(g["seed"]), max_outer_iters=400)
    for p in (gpu, cpu):
        p.SetGyroQuaternions(g["gyro_quats"], float(g["gyro_fs"]), float(g["gyro_t0"]))
        for i, fr in enumerate(g["frame_ids"]):
            p.SetTrackResult(int(fr), g["ts_a"][i], g["ts_b"][i], g["rays_a"][i], g["rays_b"][i])
    f0, F = int(g["frame_ids"][0]), len(g["frame_ids"])
    d0 = float(g["presync_result"][1])
    rg, rc = _sync_both(gpu, cpu, lambda p: p.Sync(d0, f0, f0 + F - 1, 0.0, 0.1))
    np.testing.assert_array_equal(_bits(gpu.sync_trace()), _bits(cpu.sync_trace()))
    assert rg == rc


def test_batched_windows_and_simplified_mode_are_bit_identical(hosttest_lib):
    """several windows in lock-step on the device's own loop (overlapping windows: a frame has one slot per
    window), ragged track counts 40..200, and the thesis' no-translation mode"""
    from rssync_amd import synth
    F = 40
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=9)
    rng = np.random.default_rng(5)
    frames = []
    for fr in range(F):
        frames += list(synth.make_frames(gyro, fr, fr + 1, int(rng.integers(40, 200)), seed=9))
    gpu, cpu = _pair(hosttest_lib, seed=5, max_outer_iters=25)
    _fill_both(gpu, cpu, gyro, frames)
    b, e = [0, 6, 12, 20, 25], [11, 17, 23, 31, 39]
    d0 = [0.035, 0.036, 0.0365, 0.037, 0.038]
    (cg, dg), (cc, dc) = _sync_both(gpu, cpu, lambda p: p.sync_windows(d0, b, e, 0.0, 0.2))
    np.testing.assert_array_equal(_bits(dg), _bits(dc))
    np.testing.assert_array_equal(_bits(cg), _bits(cc))
    for w in range(len(b)):
        np.testing.assert_array_equal(_bits(gpu.window_trace(w)), _bits(cpu.window_trace(w)))
    sg = gpu.SyncSimplified(0.0355, 0, F - 1, 0.0, 0.1)
    sc = cpu.SyncSimplified(0.0355, 0, F - 1, 0.0, 0.1)
    np.testing.assert_array_equal(_bits(gpu.sync_trace()), _bits(cpu.sync_trace()))
    assert sg == sc
