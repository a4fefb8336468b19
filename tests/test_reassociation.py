"""Device order vs reference order on noisy data, on the CPU alone.

tests/test_gpu_bitexact.py shows on the GPU that the device and the CPU stand-in (device association) produce the
SAME BITS along whole Sync traces.  So the difference between the device and the reference-order oracle IS the
difference between the stand-in and the oracle, which needs no GPU: same algorithm, same inputs, same GuessMotion
winners (transplanted), different rounding (association of the row sums, fused products, last bits of log1p).
On noise-free data that is 1e-11 s; on noisy windows the per-frame L-BFGS amplifies it.  The bounds asserted here are
those measured by tests/measure/reassociation.py (profiles/r3_reassociation.json), with a factor 2.5 of head room.
"""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEASURED = json.load(open(os.path.join(ROOT, "profiles", "r3_reassociation.json")))["scenes"]


def reassociation_bound_s():
    """largest device-order vs reference-order difference of one Sync call measured on noisy windows, x 2.5"""
    return 2.5 * max(MEASURED[k]["device_order_minus_reference_order_s"]["max"] for k in ("reference_workload_noisy", "config1_noisy"))


def _pair(hosttest_lib, gyro, frames, seed):
    import rssync_amd
    from oracle.oracle import OracleProblem
    dev = rssync_amd.SyncProblem(seed=seed, max_outer_iters=400, _lib=hosttest_lib)
    ora = OracleProblem(seed=seed, max_outer_iters=400, threads=os.cpu_count() or 1, faithful=False)
    for p in (dev, ora):
        p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
        for fr in frames:
            p.SetTrackResult(*fr)
    return dev, ora


def test_noisy_windows_differ_by_reassociation_only(hosttest_lib):
    from rssync_amd import synth
    F, N, window = 260, 130, 60
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=57)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=57))
    dev, ora = _pair(hosttest_lib, gyro, frames, seed=5)
    diffs = []
    for b in range(0, F - window - 1, 33):
        d0 = ora.PreSync(0.0, b, b + window, 0.002, 0.1)[1]
        co, do, tro = ora.sync_trace(d0, b, b + window, 0.0, 0.1)
        dev.set_init_override(ora.last_init_winners())   # same motion estimates to start from
        cd, dd = dev.Sync(d0, b, b + window, 0.0, 0.1)
        diffs.append(abs(dd - do))
        # the first evaluation (before anything is amplified) agrees to rounding: the same GuessMotion state, the
        # same loss up to the order of its sums
        Mk = ora.sync_state()
        assert len(Mk[1]) == window + 1
    diffs = np.array(diffs)
    assert diffs.max() < reassociation_bound_s(), diffs
    assert np.median(diffs) < 3e-4
    assert diffs.max() > 1e-9        # (if this ever fails the two have become the same arithmetic: tighten everything)


def test_noise_free_windows_agree_to_1e_9_s(hosttest_lib):
    from rssync_amd import synth
    F, N = 70, 130
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=58)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=58, noise=0.0, outliers=0.0))
    dev, ora = _pair(hosttest_lib, gyro, frames, seed=5)
    co, do, tro = ora.sync_trace(0.0355, 0, 60, 0.0, 0.1)
    dev.set_init_override(ora.last_init_winners())
    cd, dd = dev.Sync(0.0355, 0, 60, 0.0, 0.1)
    trd = dev.sync_trace()
    assert len(trd) == len(tro)
    np.testing.assert_allclose(trd[:, 0], tro[:, 0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(trd[:, 2], tro[:, 2], rtol=1e-5, atol=1e-9)   # (the loss is ~0 here: it is all L-BFGS stopping slack)
    np.testing.assert_array_equal(trd[:, 5], tro[:, 5])
    assert abs(dd - synth.D_TRUE) < 1e-4
