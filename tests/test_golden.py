"""Committed fixtures (tests/golden/oracle_small.npz, made by tests/golden/make_golden.py): the
oracle's outputs on a stored input.  They pin oracle and HIP path against drift; the reference has
no vectors of its own for this path (PARITY UNPINNED)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_small.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _fill(p, g):
    p.SetGyroQuaternions(g["gyro_quats"], float(g["gyro_fs"]), float(g["gyro_t0"]))
    for i, fr in enumerate(g["frame_ids"]):
        p.SetTrackResult(int(fr), g["ts_a"][i], g["ts_b"][i], g["rays_a"][i], g["rays_b"][i])
    return p


def test_oracle_reproduces_the_fixture(built, gold):
    from oracle import oracle as ora
    from oracle.oracle import OracleProblem
    seed = int(gold["seed"])
    F = len(gold["frame_ids"])
    f0 = int(gold["frame_ids"][0])
    o = _fill(OracleProblem(seed=seed, threads=os.cpu_count() or 1, faithful=False), gold)  # schedule-independent
    d, c, fc, bh = o.presync_curve(0.0, f0, f0 + F, 0.004, 0.1, per_frame=F)
    np.testing.assert_array_equal(d, gold["presync_delays"])
    np.testing.assert_array_equal(bh, gold["presync_best_h"])
    np.testing.assert_allclose(fc, gold["presync_frame_costs"], rtol=1e-12)
    np.testing.assert_allclose(c, gold["presync_costs"], rtol=1e-12)
    assert o.PreSync(0.0, f0, f0 + F, 0.004, 0.1)[1] == gold["presync_result"][1]
    dd, dc = o.DebugPreSync(0.0, f0, f0 + F, 0.1, 9)
    np.testing.assert_allclose(dc, gold["debug_costs"], rtol=1e-12)
    c2, d2, tr = o.sync_trace(float(gold["presync_result"][1]), f0, f0 + F - 1, 0.0, 0.1)
    np.testing.assert_allclose(tr, gold["sync_trace"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose([c2, d2], gold["sync_result"], rtol=1e-9)
    np.testing.assert_allclose(o.problem_matrix(33, 0.0371), gold["P_frame33"], rtol=0, atol=1e-15)
    oc = OracleProblem(seed=seed, threads=1, faithful=False)
    oc.SetGyroQuaternions(gold["clean_gyro_quats"], float(gold["clean_gyro_fs"]), float(gold["clean_gyro_t0"]))
    for i in range(len(gold["clean_ts_a"])):
        oc.SetTrackResult(i, gold["clean_ts_a"][i], gold["clean_ts_b"][i], gold["clean_rays_a"][i], gold["clean_rays_b"][i])
    Fc = len(gold["clean_ts_a"])
    c3, d3, tr3 = oc.sync_trace(0.0355, 0, Fc - 1, 0.0, 0.1)
    np.testing.assert_allclose(tr3, gold["clean_sync_trace"], rtol=1e-9, atol=1e-12)
    c4, d4, tr4 = oc.sync_simplified_trace(0.0355, 0, Fc - 1, 0.0, 0.1)
    np.testing.assert_allclose(tr4, gold["clean_simplified_trace"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose([c4, d4], gold["clean_simplified_result"], rtol=1e-9)
    o2 = OracleProblem(seed=seed)
    o2.SetGyroQuaternionsTimestamped(gold["ts_us"], gold["ts_quats"])
    assert o2.gyro_info()[:2] == (float(gold["ts_fs"]), float(gold["ts_start"]))
    np.testing.assert_allclose(o2.gyro_knots(), gold["ts_knots"], rtol=0, atol=1e-15)
    cases = [(0, 0, 0, 2), (30, 5, 7, 128), (-3, ora.STREAM_SYNC_INIT, 199, 2048), (2 ** 40, ora.STREAM_DEBUG + 3, 19, 17)]
    got = np.array([ora.sample_pair(seed, *cs) for cs in cases])
    np.testing.assert_array_equal(got, gold["sample_pairs"])  # integer sampler: bit-exact


def test_host_solver_against_the_fixture(hosttest_lib, gold):
    """same fixture through the product's host code + CPU test double (fp32 arithmetic)"""
    import rssync_amd
    _check_product(rssync_amd.SyncProblem(seed=int(gold["seed"]), _lib=hosttest_lib), gold)


@pytest.mark.gpu
def test_hip_path_against_the_fixture(gold):
    import rssync_amd
    _check_product(rssync_amd.SyncProblem(seed=int(gold["seed"])), gold)


def _check_product(h, gold):
    F = len(gold["frame_ids"])
    f0 = int(gold["frame_ids"][0])
    _fill(h, gold)
    d, c, fc, bh = h.presync_curve(0.0, f0, f0 + F, 0.004, 0.1, per_frame=F)
    np.testing.assert_array_equal(d, gold["presync_delays"])      # candidate delays: bit-exact
    same = bh == gold["presync_best_h"]
    assert same.mean() > 0.99                                      # arg-min flips only at fp32 near-ties
    np.testing.assert_allclose(fc[same], gold["presync_frame_costs"][same], rtol=1e-3)
    np.testing.assert_allclose(c, gold["presync_costs"], rtol=5e-3)
    cost, delay = h.PreSync(0.0, f0, f0 + F, 0.004, 0.1)
    assert delay == gold["presync_result"][1]
    P = h.problem_matrix(33, 0.0371, gold["P_frame33"].shape[0])
    assert np.abs(P - gold["P_frame33"]).max() < 5e-7
    dd, dc = h.DebugPreSync(0.0, f0, f0 + F, 0.1, 9)
    np.testing.assert_array_equal(dd, gold["debug_delays"])
    np.testing.assert_allclose(dc, gold["debug_costs"], rtol=5e-3)
    # Sync from the stored PreSync result against the stored trace (24 x 128, noise + outliers).  A noisy scene: the
    # optimiser amplifies rounding differences.  What the device computes is bit-identical to the CPU stand-in in
    # device order (tests/test_gpu_bitexact.py); stand-in and reference-order oracle differ by the reassociation
    # scatter measured per scene in profiles/r5_reassociation.json -- on THIS scene 1.8e-5 s, so the north-star
    # 1e-4 s is what is asserted (tests/noisy_scenes.py).  Both start from the oracle's GuessMotion winners.
    import noisy_scenes as ns
    scene = ns.golden_noisy()
    m = ns.measured(scene.name)
    o = scene.oracle()
    c2o, d2o, tro = o.sync_trace(float(gold["presync_result"][1]), f0, f0 + F - 1, 0.0, 0.1)
    np.testing.assert_allclose(tro, gold["sync_trace"], rtol=1e-9, atol=1e-12)   # (the oracle still reproduces the fixture)
    h.set_init_override(o.last_init_winners())
    c2, d2 = h.Sync(float(gold["presync_result"][1]), f0, f0 + F - 1, 0.0, 0.1)
    tr = h.sync_trace()
    assert ns.bound_s(scene.name) == ns.NORTH_STAR_S
    assert abs(d2 - gold["sync_result"][1]) < ns.bound_s(scene.name)
    assert d2 == pytest.approx(m["delays_s"]["device_order"][0], rel=0, abs=1e-9)   # = the CPU stand-in's result
    assert c2 == pytest.approx(gold["sync_result"][0], rel=2.5 * m["cost_rel"])
    assert [len(tr)] == m["outer_iterations"]["device_order"]
    n = min(len(tr), len(gold["sync_trace"]))
    np.testing.assert_allclose(tr[:n, 0], gold["sync_trace"][:n, 0], atol=2.5 * m["delay_after_each_outer_iteration_max_abs_s"])
    # noise-free scene: every outer iteration of Sync (and of the simplified mode) follows the stored trace
    hc = type(h)(seed=int(gold["seed"]), _lib=h._lib)
    hc.SetGyroQuaternions(gold["clean_gyro_quats"], float(gold["clean_gyro_fs"]), float(gold["clean_gyro_t0"]))
    Fc = len(gold["clean_ts_a"])
    for i in range(Fc):
        hc.SetTrackResult(i, gold["clean_ts_a"][i], gold["clean_ts_b"][i], gold["clean_rays_a"][i], gold["clean_rays_b"][i])
    for call, key in ((hc.Sync, "clean_sync"), (hc.SyncSimplified, "clean_simplified")):
        c3, d3 = call(0.0355, 0, Fc - 1, 0.0, 0.1)
        tr3, want = hc.sync_trace(), gold[key + "_trace"]
        assert len(tr3) == len(want)
        np.testing.assert_allclose(tr3[:, 0], want[:, 0], rtol=0, atol=1e-7)     # delay after every outer iteration, seconds
        np.testing.assert_allclose(tr3[:, 2], want[:, 2], rtol=1e-5, atol=1e-9)  # loss at every outer iteration
        np.testing.assert_array_equal(tr3[:, 5], want[:, 5])                      # line-search trials taken
        assert abs(d3 - gold[key + "_result"][1]) < 1e-7
    h2 = type(h)(seed=int(gold["seed"]), _lib=h._lib)
    h2.SetGyroQuaternionsTimestamped(gold["ts_us"], gold["ts_quats"])
    assert h2.gyro_info()[:2] == (float(gold["ts_fs"]), float(gold["ts_start"]))
    np.testing.assert_array_equal(h2.gyro_knots(), gold["ts_knots"])  # host-side integer grid + slerp: bit-exact
