"""The LMedS tile kernel's SUB-SHAPES (round 6): speed, never a bit.

A frame of 513 .. 8192 tracks is swept by the tile kernel of its size class: four waves of 4 / 8 / 24 rows per thread, eight
waves of 16 above 6144 tracks (rssync_kernels.hip: class_of).  A clip whose frames have 1500 tracks swept 2048 rows per
hypothesis that way, one of 4100-track frames 8192.  Now PreSync's launch for a class takes the smallest shape that holds
the LARGEST frame of the class in the selection -- any number of rows per thread from 3 to 24, eight waves of 13 .. 16
(lmeds_shape, with_tile_shape) -- as the one-wave kernels have always followed the selection's largest small frame.  A thread adds its
rows in order and a row beyond the frame adds an exact zero, and the winner is an exact arg-min whatever the tile's size:
the reference's per-frame results (core_private.cpp:73-86) must come out bit for bit as from the class's own shape
(RSSYNC_NO_SUBSHAPES=1), which these tests demand, and which shape ran is read back (rship_lmeds_shapes).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 77
# (largest frame of the selection, its class, the shape expected, the class's own shape): every shape once, and the edges
def _case(n):
    if n <= 512:                       # one wave per frame: rows per lane (1 .. 4, then 8 -- or 5 .. 7 as sub-shapes)
        need = (n + 63) // 64
        return (n, 0, need, need if need <= 4 else 8)
    cls = 1 if n <= 1024 else (2 if n <= 2048 else (3 if n <= 6144 else 4))
    need = (n + 255) // 256
    if cls == 4:
        need += need & 1               # eight waves: tiles of 512 x rows-per-thread
    own = {1: 4, 2: 8, 3: 16 if n <= 4096 else 24, 4: 32}[cls]      # (class 3's own shape follows the selection's largest frame too: 16 or 24)
    return (n, cls, max(need, 3), own)


CASES = [_case(n) for n in [200, 257, 300, 350, 400, 448, 449, 512] + [513, 700, 768, 769, 1000] + [256 * g - 10 for g in range(5, 25)]
         + [1536, 1537, 2049, 4096, 4097, 6144] + [6145] + [256 * g - 10 for g in range(26, 33, 2)] + [6656, 6657, 8192]]


def _problem(gyro, frames, env=None):
    import rssync_amd
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        p = rssync_amd.SyncProblem(seed=SEED, verbose=False)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for f in frames:
        p.SetTrackResult(*f)
    return p


def _clip(n_max, lo, seed, **kw):
    """four frames of one class: the largest has n_max tracks, the others fewer (down to the class's first count)"""
    from rssync_amd import synth
    counts = [n_max, max(lo, n_max - 97), max(lo, (n_max + lo) // 2), lo]
    g = synth.make_gyro(0.0, (len(counts) + 2) / synth.FPS, seed=seed)
    frames = [next(iter(synth.make_frames(g, fr, fr + 1, n, seed=seed, **kw))) for fr, n in enumerate(counts)]
    return g, frames, counts


@pytest.mark.parametrize("n_max,cls,shape,own", CASES)
def test_a_sub_shape_gives_the_bits_of_the_class_shape(n_max, cls, shape, own):
    lo = {0: 100, 1: 513, 2: 1025, 3: 2049, 4: 6145}[cls]
    g, frames, counts = _clip(n_max, lo, seed=n_max, noise=5e-4, outliers=0.08)
    F = len(frames)
    sub = _problem(g, frames)
    full = _problem(g, frames, env={"RSSYNC_NO_SUBSHAPES": "1"})
    ds, cs, fcs, bhs = sub.presync_curve(0.03, 0, F, 0.001, 0.02, per_frame=F)          # 40 candidates: two chunks
    df, cf, fcf, bhf = full.presync_curve(0.03, 0, F, 0.001, 0.02, per_frame=F)
    want = [0] * 6
    want[cls] = shape
    assert sub.lmeds_shapes() == want, (sub.lmeds_shapes(), counts)
    want[cls] = own
    assert full.lmeds_shapes() == want
    np.testing.assert_array_equal(ds, df)
    np.testing.assert_array_equal(bhs, bhf, err_msg="LMedS winners")
    np.testing.assert_array_equal(fcs, fcf, err_msg="per-frame costs")
    np.testing.assert_array_equal(cs, cf, err_msg="the curve")
    assert sub.PreSync(0.0, 0, F, 0.001, 0.06) == full.PreSync(0.0, 0, F, 0.001, 0.06)


def test_the_shape_follows_the_selection_not_the_problem():
    """frames 0-1 have 1300 tracks, frame 2 has 2000: a sweep over 0..1 takes the 1536-row shape, one over all three the class's"""
    from rssync_amd import synth
    counts = [1300, 1290, 2000]
    g = synth.make_gyro(0.0, 5 / synth.FPS, seed=3)
    frames = [next(iter(synth.make_frames(g, fr, fr + 1, n, seed=3, noise=5e-4, outliers=0.05))) for fr, n in enumerate(counts)]
    p = _problem(g, frames)
    _, _, fc2, bh2 = p.presync_curve(0.03, 0, 2, 0.001, 0.01, per_frame=2)
    assert p.lmeds_shapes()[2] == 6
    _, _, fc3, bh3 = p.presync_curve(0.03, 0, 3, 0.001, 0.01, per_frame=3)
    assert p.lmeds_shapes() == [0, 0, 8, 0, 0, 0]          # (only the classes the last sweep launched are reported)
    np.testing.assert_array_equal(fc2, fc3[:, :2])
    np.testing.assert_array_equal(bh2, bh3[:, :2])


def test_sub_shapes_with_the_window_in_dynamic_lds():
    """gyro at 4 kHz: the spline window moves to dynamic LDS (WIN = 0 instantiations of the sub-shapes)"""
    from rssync_amd import synth
    for n_max, cls, shape in ((1400, 2, 6), (2600, 3, 11), (4500, 3, 18), (7000, 4, 28)):
        lo = {2: 1025, 3: 2049, 4: 6145}[cls]
        counts = [n_max, lo]
        g = synth.make_gyro(0.0, 4 / synth.FPS, seed=9, fs=4000.0)
        frames = [next(iter(synth.make_frames(g, fr, fr + 1, n, seed=9, noise=5e-4, outliers=0.05))) for fr, n in enumerate(counts)]
        sub = _problem(g, frames)
        full = _problem(g, frames, env={"RSSYNC_NO_SUBSHAPES": "1"})
        a = sub.presync_curve(0.03, 0, 2, 0.001, 0.012, per_frame=2)
        b = full.presync_curve(0.03, 0, 2, 0.001, 0.012, per_frame=2)
        assert sub.window_info()["presync_window_dynamic"], n_max
        assert sub.lmeds_shapes()[cls] == shape
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


def test_near_static_frames_in_a_sub_shape():
    """the fp64-rows form runs in the class's own shape over the pairs the sub-shape's watch flagged: same results as without sub-shapes"""
    from rssync_amd import synth
    counts = [1400, 1200]
    g = synth.make_gyro(0.0, 4 / synth.FPS, seed=21)
    frames = [next(iter(synth.make_frames(g, fr, fr + 1, n, seed=21, noise=1e-6, outliers=0.1, translation=5e-5))) for fr, n in enumerate(counts)]
    sub = _problem(g, frames)
    full = _problem(g, frames, env={"RSSYNC_NO_SUBSHAPES": "1"})
    a = sub.presync_curve(synth.D_TRUE, 0, 2, 0.0001, 0.002, per_frame=2)
    b = full.presync_curve(synth.D_TRUE, 0, 2, 0.0001, 0.002, per_frame=2)
    assert sub.near_static_stats()["pairs"] > 0 and sub.near_static_stats() == full.near_static_stats()
    assert sub.lmeds_shapes()[2] == 6
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("seed", range(8 + int(os.environ.get("RSSYNC_FUZZ_CASES", "0"))))
def test_random_mixed_selections(seed):
    """random clips -- 3 to 10 frames of ragged track counts from 40 to 7800 in one problem (up to five size classes at once), random
    gyro rate, random sweep -- with and without sub-shapes: every per-frame cost, every winner, the curve, PreSync's result and the
    Sync that follows, bit for bit"""
    from rssync_amd import synth
    rng = np.random.default_rng(4242 + seed)
    F = int(rng.integers(3, 11))
    fs = float(rng.choice([200.0, 400.0, 1000.0, 3200.0]))
    tops = rng.choice([300, 450, 640, 900, 1300, 1800, 2300, 3100, 4500, 5700, 6500, 7800], size=F)
    counts = [int(rng.integers(max(40, t // 2), t + 1)) for t in tops]
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, fs=fs, seed=seed)
    frames = [next(iter(synth.make_frames(g, fr, fr + 1, n, seed=seed, noise=float(rng.choice([0.0, 5e-4, 2e-3])), outliers=0.08)))
              for fr, n in enumerate(counts)]
    sub = _problem(g, frames)
    full = _problem(g, frames, env={"RSSYNC_NO_SUBSHAPES": "1"})
    step = float(rng.choice([0.0005, 0.001, 0.002]))
    radius = float(rng.uniform(0.005, 0.03))
    lo = int(rng.integers(0, F - 2))
    a = sub.presync_curve(0.03, lo, F, step, radius, per_frame=F - lo)
    b = full.presync_curve(0.03, lo, F, step, radius, per_frame=F - lo)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y, err_msg=str(counts))
    ra, rb = sub.PreSync(0.0, 0, F, 0.002, 0.08), full.PreSync(0.0, 0, F, 0.002, 0.08)
    assert ra == rb, counts
    assert sub.Sync(ra[1], 0, F - 1, 0.0, 0.2) == full.Sync(rb[1], 0, F - 1, 0.0, 0.2), counts
    np.testing.assert_array_equal(sub.sync_trace(), full.sync_trace())


def test_sub_shapes_on_the_general_spline_path():
    """candidate delays seconds away, beyond either end of the gyro track (minispline.cpp:49-54: the extrapolation branches): the
    rows take the kernels' general path (table from L2, the careful form of the row) -- in a sub-shape as in the class's own"""
    from rssync_amd import synth
    for n_max, cls, shape in ((700, 1, 3), (1400, 2, 6), (2700, 3, 11), (4500, 3, 18), (7000, 4, 28)):
        lo = {1: 513, 2: 1025, 3: 2049, 4: 6145}[cls]
        counts = [n_max, lo + 7]
        g = synth.make_gyro(0.0, 4 / synth.FPS, seed=13)
        frames = [next(iter(synth.make_frames(g, fr, fr + 1, n, seed=13, noise=5e-4, outliers=0.05))) for fr, n in enumerate(counts)]
        sub = _problem(g, frames)
        full = _problem(g, frames, env={"RSSYNC_NO_SUBSHAPES": "1"})
        for centre in (-2.5, 1.02, 3.0):       # before the track, across its end, after it
            a = sub.presync_curve(centre, 0, 2, 0.004, 0.05, per_frame=2)
            b = full.presync_curve(centre, 0, 2, 0.004, 0.05, per_frame=2)
            assert sub.lmeds_shapes()[cls] == shape
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y, err_msg="%d tracks, delays around %g s" % (n_max, centre))
