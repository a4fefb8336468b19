"""PreSync's lazy quartile selection (kernels/lmeds.hpp, round 3) against round 2's exact selection of every
quartile: the arg-min over hypotheses is exact either way, so winners and costs must be IDENTICAL for every
(frame, candidate) -- including scenes built to produce equal quartiles (duplicated tracks, noise-free data), where the
reference's first-wins tie rule decides.

The exact-selection kernels are not part of the product build: they exist in the TEST-VARIANTS build of the same
sources (-DRSSYNC_TEST_VARIANTS=1 -> rs-sync_amd/_variants/lib_testvariants.so, built by the fixture below when it is
missing or older than the sources; RSSYNC_K2_EXACT_SELECT=1, read when a problem is created, selects them there and
is refused by the product build).  The lazy side is the PRODUCT library."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# (the `variants_lib` fixture: tests/conftest.py)


def test_the_product_build_refuses_the_variant_switch(built, monkeypatch):
    import rssync_amd
    monkeypatch.setenv("RSSYNC_K2_EXACT_SELECT", "1")
    with pytest.raises(rssync_amd.RsSyncError):
        rssync_amd.SyncProblem(seed=1)


@pytest.fixture()
def _two(variants_lib):
    def make(seed):
        import rssync_amd
        lazy = rssync_amd.SyncProblem(seed=seed)                       # the product library
        os.environ["RSSYNC_K2_EXACT_SELECT"] = "1"
        try:
            exact = rssync_amd.SyncProblem(seed=seed, _lib=variants_lib)   # the same sources + the exact-selection kernels
        finally:
            del os.environ["RSSYNC_K2_EXACT_SELECT"]
        return lazy, exact
    return make


def _curves(p, F, step, radius, d0=0.0):
    return p.presync_curve(d0, 0, F, step, radius, per_frame=F)


@pytest.mark.parametrize("F,N,step,radius,kw", [
    (24, 2048, 0.0005, 0.05, {}),                                  # bench shape, 200 candidates (chunks of 32)
    (16, 2047, 0.001, 0.06, {"noise": 0.0, "outliers": 0.0}),      # noise-free: clusters of (near-)equal residuals
    (40, 600, 0.001, 0.06, {}),                                    # 4 rows per thread
    (30, 300, 0.002, 0.1, {}),                                     # 2 rows per thread
    (6, 5000, 0.002, 0.05, {}),                                    # 24 rows per thread (four waves: 4097 .. 6144 tracks)
    (4, 7000, 0.002, 0.05, {}),                                    # eight waves x 16 rows per thread (6145 .. 8192 tracks)
    (8, 3000, 0.002, 0.05, {"noise": 3e-3, "outliers": 0.3}),      # 16 rows per thread, heavy outliers
])
def test_lazy_selection_equals_exact_selection(built, _two, F, N, step, radius, kw):
    from rssync_amd import synth
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=70 + F)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=70 + F, **kw))
    lazy, exact = _two(seed=4242)
    for p in (lazy, exact):
        p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
        for fr in frames:
            p.SetTrackResult(*fr)
    dl, cl, fl, bl = _curves(lazy, F, step, radius)
    de, ce, fe, be = _curves(exact, F, step, radius)
    np.testing.assert_array_equal(bl, be)                # same winning hypothesis, every (frame, candidate)
    np.testing.assert_array_equal(fl.view(np.uint64), fe.view(np.uint64))
    np.testing.assert_array_equal(cl.view(np.uint64), ce.view(np.uint64))
    assert lazy.PreSync(0.0, 0, F, step, radius) == exact.PreSync(0.0, 0, F, step, radius)


def test_ties_go_to_the_earlier_hypothesis_either_way(built, _two):
    """tracks duplicated many times over: different row pairs give the SAME direction and so the same quartile, bit
    for bit; the first of them must win (core_private.cpp:53 strict <) under both selections"""
    from rssync_amd import synth
    F, N0, copies = 10, 40, 16
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=3)
    lazy, exact = _two(seed=99)
    for p in (lazy, exact):
        p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
        for fr, ta, tb, ra, rb in synth.make_frames(gyro, 0, F, N0, seed=3):
            p.SetTrackResult(fr, np.tile(ta, copies), np.tile(tb, copies), np.tile(ra, (copies, 1)), np.tile(rb, (copies, 1)))
    dl, cl, fl, bl = _curves(lazy, F, 0.002, 0.06)
    de, ce, fe, be = _curves(exact, F, 0.002, 0.06)
    np.testing.assert_array_equal(bl, be)
    np.testing.assert_array_equal(fl.view(np.uint64), fe.view(np.uint64))
