"""Device order vs reference order on noisy data, on the CPU alone.

tests/test_gpu_bitexact.py shows on the GPU that the device and the CPU stand-in (device association) produce the
SAME BITS along whole Sync traces.  So the difference between the device and the reference-order oracle IS the
difference between the stand-in and the oracle, which needs no GPU: same algorithm, same inputs, same GuessMotion
winners (transplanted), different rounding (association of the row sums, fused products, last bits of log1p).
On noise-free data that is 1e-11 s; on noisy windows the per-frame L-BFGS amplifies it.

Every tolerance is the scene's own (tests/noisy_scenes.py: `bound_s(scene)` = the north-star 1e-4 s where that
scene's measured maximum is below it, 2.5 x its measured maximum otherwise; measured by tests/measure/reassociation.py
into profiles/r5_reassociation.json), the file is checked to be current, and the claim "no worse than the algorithm's
own scatter" is asserted on the distribution: median and 90th percentile against the control (the reference-order
oracle started 1e-9 s away from itself).
"""
import numpy as np
import pytest

import noisy_scenes as ns

# how much larger than the control's the device-order scatter may be (measured: median 0.52x, p90 0.58x, max 0.92x)
SCATTER_FACTOR = 1.25


def test_noisy_windows_scatter_like_the_algorithm_itself(hosttest_lib):
    """the reference driver's workload shape, 24 windows of 61 x 130 with noise and outliers"""
    scene = ns.reference_workload_noisy()
    recs = ns.run_scene(scene, scene.device(hosttest_lib), scene.oracle(), control=scene.oracle())
    dev = np.abs([r["d_dev"] - r["d_ora"] for r in recs])
    ctl = np.abs([r["d_ctl"] - r["d_ora"] for r in recs])
    # (1) every window within the scene's bound
    assert dev.max() < ns.bound_s(scene.name), dev
    # (2) the distribution is the algorithm's own: rounding in another order moves the result no further than a
    #     1e-9 s change of the starting delay does
    assert np.median(dev) <= SCATTER_FACTOR * np.median(ctl), (np.median(dev), np.median(ctl))
    assert np.percentile(dev, 90) <= SCATTER_FACTOR * np.percentile(ctl, 90), (np.percentile(dev, 90), np.percentile(ctl, 90))
    assert dev.max() <= SCATTER_FACTOR * ctl.max(), (dev.max(), ctl.max())
    # (3) and the committed measurement is this build's
    m = ns.measured(scene.name)
    np.testing.assert_allclose([r["d_dev"] for r in recs], m["delays_s"]["device_order"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([r["d_ora"] for r in recs], m["delays_s"]["reference_order"], rtol=0, atol=1e-9)
    assert m["device_order_minus_reference_order_s"]["max"] > ns.NORTH_STAR_S   # (stated in BASELINE.md: NOT within the north star)
    assert dev.max() > 1e-9        # (if this ever fails the two have become the same arithmetic: tighten everything)


@pytest.mark.parametrize("name", ["config1_noisy", "golden_noisy", "big_8193"])
def test_single_call_scenes_and_the_measurement_file_is_current(hosttest_lib, name):
    """The scenes the GPU tests assert with `bound_s`: recomputed here with the stand-in, they must give the numbers
    of profiles/r5_reassociation.json (a change of the arithmetic that was not re-measured fails HERE, on the CPU),
    and the north-star 1e-4 s holds on each of them."""
    scene = ns.SCENES[name]()
    (r,) = ns.run_scene(scene, scene.device(hosttest_lib), scene.oracle())
    m = ns.measured(name)
    assert abs(abs(r["d_dev"] - r["d_ora"]) - m["device_order_minus_reference_order_s"]["max"]) < 1e-9
    assert [len(r["trace_dev"])] == m["outer_iterations"]["device_order"]
    assert [len(r["trace_ora"])] == m["outer_iterations"]["reference_order"]
    assert ns.bound_s(name) == ns.NORTH_STAR_S
    assert abs(r["d_dev"] - r["d_ora"]) < ns.NORTH_STAR_S


def test_noise_free_windows_agree_to_1e_9_s(hosttest_lib):
    from rssync_amd import synth
    scene = ns.Scene("clean", *ns._synth(70, 130, 58, noise=0.0, outliers=0.0), calls=[(0.0355, 0, 60, 0.0, 0.1)], seed=5)
    (r,) = ns.run_scene(scene, scene.device(hosttest_lib), scene.oracle())
    trd, tro = r["trace_dev"], r["trace_ora"]
    assert len(trd) == len(tro)
    np.testing.assert_allclose(trd[:, 0], tro[:, 0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(trd[:, 2], tro[:, 2], rtol=1e-5, atol=1e-9)   # (the loss is ~0 here: it is all L-BFGS stopping slack)
    np.testing.assert_array_equal(trd[:, 5], tro[:, 5])
    assert abs(r["d_dev"] - synth.D_TRUE) < 1e-4


def test_the_stated_miss_on_205_windows_over_five_clips(hosttest_lib):
    """Round 4 stated the one miss -- the north star's 1e-4 s on the reference's own workload shape WITH noise -- from 24
    overlapping windows of one clip.  Round 5: 5 independent clips x 41 windows half a window apart (the reference
    driver's spacing) = 205 windows (tests/measure/reassociation.py -> "pooled_reference_workload_noisy").  Asserted:
    the committed numbers are this build's (one clip recomputed here), the device-order scatter is no larger than the
    reference-order algorithm's own (control: the oracle started 1e-9 s away) at the median, the 90th percentile and the
    maximum, on the pooled sample and on every clip's median; and what fraction of the windows IS within 1e-4 s."""
    import json
    pooled = json.load(open(ns.MEASURED_FILE))["pooled_reference_workload_noisy"]
    assert pooled["windows"] == len(ns.POOLED_SEEDS) * ns.POOLED_WINDOWS_PER_SEED >= 200 and len(ns.POOLED_SEEDS) >= 5
    # (1) current: clip 33 recomputed
    scene = ns.reference_workload_noisy_clip(33)
    recs = ns.run_scene(scene, scene.device(hosttest_lib), scene.oracle(), control=scene.oracle())
    m = pooled["per_seed"]["33"]
    np.testing.assert_allclose([r["d_dev"] for r in recs], m["delays_s"]["device_order"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([r["d_ora"] for r in recs], m["delays_s"]["reference_order"], rtol=0, atol=1e-9)
    # (2) the distribution: rounding in another order moves a window no further than the algorithm's own chaos does
    dev, ctl = pooled["device_order_minus_reference_order_s"], pooled["control_reference_order_started_1e-9_s_away_s"]
    for q in ("median", "p90", "max"):
        assert dev[q] <= SCATTER_FACTOR * ctl[q], (q, dev[q], ctl[q])
    for sd in ns.POOLED_SEEDS:
        a = pooled["per_seed"][str(sd)]
        assert a["device_order_minus_reference_order_s"]["median"] <= SCATTER_FACTOR * a["control_reference_order_started_1e-9_s_away_s"]["median"], sd
    # (3) the miss, in numbers: most windows are within the north star, the worst is not (BASELINE.md states both)
    assert 0.75 < pooled["fraction_within_north_star_1e-4_s"] < 1.0
    assert dev["max"] > ns.NORTH_STAR_S and dev["median"] < ns.NORTH_STAR_S
