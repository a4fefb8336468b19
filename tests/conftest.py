import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The product verifies one window-executor call in 256 against the launch chain (a process-wide count:
    # sync_problem.cpp check_this_call).  The tests count executor runs and time calls, so the sampling is off here unless
    # a test asks for it (tests/test_gpu_executor.py::test_one_call_in_n_is_verified_in_production).
    os.environ.setdefault("RSSYNC_EXECUTOR_CHECK_EVERY", "0")


@pytest.fixture(scope="session")
def built():
    """Everything compiled (the driver's build() step); cheap when already up to date."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def hosttest_lib(built):
    """The product's host solver (sync_problem.cpp) linked against the CPU test double of the
    device ABI (tests/cpu_device/rship_cpu.cpp): host logic on a machine without a GPU."""
    import ctypes
    import subprocess
    from rssync_amd.problem import bind
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "librssync_hosttest.so")
    srcs = [os.path.join(ROOT, "rs-sync_amd", "csrc", "sync_problem.cpp"),
            os.path.join(ROOT, "tests", "cpu_device", "rship_cpu.cpp")]
    deps = srcs + [os.path.join(ROOT, "rs-sync_amd", "csrc", h) for h in ("device_math.hpp", "lens_math.hpp", "gyro_math.hpp", "roctx_ranges.hpp")] + \
        [os.path.join(ROOT, "include", h) for h in ("rssync.h", "rssync_c.h", "rssync_hip.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", out] + srcs)
    return bind(ctypes.CDLL(out))


@pytest.fixture(scope="session")
def variants_lib(built):
    """The TEST-VARIANTS build of the library (the product's sources with -DRSSYNC_TEST_VARIANTS=1: round 2's exact
    selection kernels, the sweep's residual dump), built when missing or older than the sources -- build() does the same,
    so a GPU box that receives the tree finds it ready."""
    import ctypes
    import subprocess
    from rssync_amd.problem import bind
    out = os.path.join(ROOT, "rs-sync_amd", "_variants", "lib_testvariants.so")
    src_dir = os.path.join(ROOT, "rs-sync_amd", "csrc")
    deps = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".hpp", ".cpp"))]
    deps += [os.path.join(src_dir, "kernels", f) for f in os.listdir(os.path.join(src_dir, "kernels"))]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "k2_build_variant.sh"), "testvariants", "-DRSSYNC_TEST_VARIANTS=1"])
    return bind(ctypes.CDLL(out))


@pytest.fixture(scope="session")
def small_case(built):
    """64 frames x 256 tracks, 400 Hz gyro (BASELINE config 1), noise + 10 % outliers."""
    from rssync_amd import synth
    F, N = 64, 256
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=1)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=1))
    return dict(F=F, N=N, gyro=gyro, frames=frames)


@pytest.fixture(scope="session")
def clean_case(built):
    """Same shape, no noise and no outliers: the residual is exactly zero at the true delay."""
    from rssync_amd import synth
    F, N = 64, 256
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=2)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=2, noise=0.0, outliers=0.0))
    return dict(F=F, N=N, gyro=gyro, frames=frames)


def fill(problem, case):
    g = case["gyro"]
    problem.SetGyroQuaternions(g.quats, g.fs, g.t0)
    for fr, ta, tb, ra, rb in case["frames"]:
        problem.SetTrackResult(fr, ta, tb, ra, rb)
    return problem


@pytest.fixture()
def oracle_small(small_case):
    from oracle.oracle import OracleProblem
    return fill(OracleProblem(seed=123, threads=os.cpu_count() or 1, faithful=False), small_case)


@pytest.fixture()
def oracle_clean(clean_case):
    from oracle.oracle import OracleProblem
    return fill(OracleProblem(seed=123, threads=os.cpu_count() or 1, faithful=False), clean_case)
