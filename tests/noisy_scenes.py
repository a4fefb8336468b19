"""The noisy scenes on which Sync's returned delay is compared with the reference-order oracle -- shared by the
measurement (tests/measure/reassociation.py -> profiles/r5_reassociation.json), the CPU tests
(tests/test_reassociation.py) and the GPU tests (test_gpu_parity.py, test_golden.py, test_gpu_mid_sizes.py), so that
a tolerance is always that scene's own measurement.

What is compared: Sync (core_private.cpp:211-334) on data with noise and outliers is a chaotic iteration -- the
per-frame objective does not depend on |M| (core_private.cpp:120), rounding moves the L-BFGS iterates along that
direction and some frames end in another basin.  The device's evaluations are bit-identical to the CPU stand-in of
tests/cpu_device (tests/test_gpu_bitexact.py), the stand-in differs from the oracle only in rounding (association of
the row sums, fused products, the last bits of log1p).  Both sides start from the SAME GuessMotion winners (the
oracle's are transplanted), so the fp32 / fp64 hypothesis search plays no part.

A scene = inputs + the list of Sync calls made on it.  `bound_s(name)`: the tolerance the tests assert,
    the north-star 1e-4 s                  where the scene's measured maximum is below it,
    2.5 x the scene's measured maximum     otherwise.
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEASURED_FILE = os.path.join(ROOT, "profiles", "r5_reassociation.json")
NORTH_STAR_S = 1e-4
SOLVER_SEED = 123        # tests/test_gpu_parity.py, tests/test_gpu_mid_sizes.py: SEED


class Scene:
    def __init__(self, name, gyro, frames, calls, seed, max_outer_iters=400, presync=None):
        self.name, self.gyro, self.frames, self.calls = name, gyro, frames, calls
        self.seed, self.max_outer_iters = seed, max_outer_iters
        self.presync = presync   # (step, radius): the initial delay of every call comes from the oracle's PreSync

    def fill(self, problem):
        g = self.gyro
        problem.SetGyroQuaternions(g.quats, g.fs, g.t0)
        for fr in self.frames:
            problem.SetTrackResult(*fr)
        return problem

    def oracle(self, threads=None):
        from oracle.oracle import OracleProblem
        return self.fill(OracleProblem(seed=self.seed, max_outer_iters=self.max_outer_iters,
                                       threads=threads or os.cpu_count() or 1, faithful=False))

    def device(self, lib=None):
        """the HIP problem, or with `lib` = the hosttest library the CPU stand-in in device order"""
        import rssync_amd
        return self.fill(rssync_amd.SyncProblem(seed=self.seed, max_outer_iters=self.max_outer_iters, _lib=lib))


def _synth(F, N, seed, **kw):
    from rssync_amd import synth
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=seed)
    return gyro, list(synth.make_frames(gyro, 0, F, N, seed=seed, **kw))


def reference_workload_noisy(n_win=24, data_seed=31, F=400, name="reference_workload_noisy"):
    """the reference driver's own shape: windows of 61 frames x 130 tracks (README.md:30-38, core_testcode.cpp:126-132),
    noise 1e-3 rad, 10 % outliers; PreSync then Sync per window (core_testcode.cpp:303-316)"""
    N, window = 130, 60
    gyro, frames = _synth(F, N, data_seed)
    step = (F - window - 1) // max(n_win - 1, 1)
    calls = [(None, w * step, w * step + window, 0.0, 0.1) for w in range(n_win)]
    return Scene(name, gyro, frames, calls, seed=99, presync=(0.002, 0.1))


# Round 5: the one stated miss (the north star's 1e-4 s on the reference's own workload shape WITH noise) on a sample
# that supports a distribution: 5 independent clips (data seeds), 41 windows each half a window apart (the reference
# driver's spacing, core_testcode.cpp:270-280) = 205 windows.  tests/measure/reassociation.py -> "pooled" in the file.
POOLED_SEEDS = (31, 32, 33, 34, 35)
POOLED_WINDOWS_PER_SEED = 41


def reference_workload_noisy_clip(data_seed):
    return reference_workload_noisy(n_win=POOLED_WINDOWS_PER_SEED, data_seed=data_seed, F=60 + 1 + 30 * (POOLED_WINDOWS_PER_SEED - 1) + 1,
                                    name="reference_workload_noisy_seed%d" % data_seed)


def reference_workload_clean(n_win=8):
    F, N, window = 200, 130, 60
    gyro, frames = _synth(F, N, 31, noise=0.0, outliers=0.0)
    step = (F - window - 1) // max(n_win - 1, 1)
    calls = [(None, w * step, w * step + window, 0.0, 0.1) for w in range(n_win)]
    return Scene("reference_workload_clean", gyro, frames, calls, seed=99, presync=(0.002, 0.1))


def config1_noisy():
    """BASELINE config 1 (64 frames x 256 tracks, 400 Hz), the `small_case` fixture of tests/conftest.py"""
    gyro, frames = _synth(64, 256, 1)
    return Scene("config1_noisy", gyro, frames, [(0.036, 0, 63, 0.0, 0.2)], seed=SOLVER_SEED)


def golden_noisy():
    """the committed fixture tests/golden/oracle_small.npz (24 frames x 128 tracks, noise + outliers)"""
    from types import SimpleNamespace
    g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_small.npz"))
    gyro = SimpleNamespace(quats=g["gyro_quats"], fs=float(g["gyro_fs"]), t0=float(g["gyro_t0"]))
    frames = [(int(fr), g["ts_a"][i], g["ts_b"][i], g["rays_a"][i], g["rays_b"][i]) for i, fr in enumerate(g["frame_ids"])]
    f0, F = int(g["frame_ids"][0]), len(g["frame_ids"])
    return Scene("golden_noisy", gyro, frames, [(float(g["presync_result"][1]), f0, f0 + F - 1, 0.0, 0.1)], seed=int(g["seed"]))


def big_frames(N):
    """4 frames of more than 8192 tracks (tests/test_gpu_mid_sizes.py::test_more_than_8192_tracks_per_frame), 12 outer iterations"""
    gyro, frames = _synth(4, N, 90 + N, noise=3e-4, outliers=0.05)
    return Scene("big_%d" % N, gyro, frames, [(0.036, 0, 3, 0.0, 0.2)], seed=SOLVER_SEED, max_outer_iters=12)


def mixed_frames():
    """one frame of 9000 tracks among four of 300 (tests/test_gpu_mid_sizes.py::test_large_and_small_frames_in_one_problem)"""
    from rssync_amd import synth
    F = 5
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=17)
    frames = [next(iter(synth.make_frames(gyro, fr, fr + 1, 9000 if fr == 2 else 300, seed=17, noise=3e-4, outliers=0.05)))
              for fr in range(F)]
    return Scene("mixed_9000_300", gyro, frames, [(0.036, 0, F - 1, 0.0, 0.2)], seed=SOLVER_SEED, max_outer_iters=12)


SCENES = {
    "reference_workload_noisy": reference_workload_noisy,
    "reference_workload_clean": reference_workload_clean,
    "config1_noisy": config1_noisy,
    "golden_noisy": golden_noisy,
    "big_8193": lambda: big_frames(8193),
    "big_10000": lambda: big_frames(10000),
    "mixed_9000_300": mixed_frames,
}


def run_scene(scene, dev, ora, control=None):
    """Every call of the scene on `dev` and on the oracle from the oracle's GuessMotion winners.
    -> list of dicts {d_dev, d_ora, c_dev, c_ora, trace_dev, trace_ora, d_ctl (control: the oracle started 1e-9 s away)}"""
    out = []
    for (d0, b, e, center, radius) in scene.calls:
        if d0 is None:
            d0 = ora.PreSync(0.0, b, e, scene.presync[0], scene.presync[1])[1]
        co, do, tro = ora.sync_trace(d0, b, e, center, radius)
        win = ora.last_init_winners()
        rec = {"d0": d0, "d_ora": do, "c_ora": co, "trace_ora": tro}
        if control is not None:
            control.set_init_override(win)
            rec["d_ctl"] = control.sync_trace(d0 + 1e-9, b, e, center, radius)[1]
        dev.set_init_override(win)
        cd, dd = dev.Sync(d0, b, e, center, radius)
        rec.update(d_dev=dd, c_dev=cd, trace_dev=dev.sync_trace())
        out.append(rec)
    return out


def stats(x):
    x = np.abs(np.asarray(x, float))
    return {"median": float(np.median(x)), "p90": float(np.percentile(x, 90)), "max": float(x.max())}


def measured(name):
    return json.load(open(MEASURED_FILE))["scenes"][name]


def bound_s(name):
    """the delay tolerance a test of scene `name` asserts, from that scene's own measurement"""
    m = measured(name)["device_order_minus_reference_order_s"]["max"]
    return NORTH_STAR_S if m < NORTH_STAR_S else 2.5 * m
