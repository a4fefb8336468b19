#!/usr/bin/env python3
"""What rounding-level differences do to Sync on noisy data -- a pure CPU measurement.

Two implementations of the same algorithm on the same inputs, started from the SAME motion estimates (GuessMotion's
winning hypotheses are transplanted, so the fp32/fp64 search plays no part):

  device order     tests/cpu_device/rship_cpu.cpp behind the product's host solver: the kernels' own fp64
                   arithmetic and summation order (bit-identical to the GPU: tests/test_gpu_bitexact.py)
  reference order  oracle/rssync_oracle.c: sequential sums, libm log1p, plain a*b+c expressions

They differ only in rounding (association of the sums over rows, which products are fused, the last bits of
log1p).  On noise-free scenes that is 1e-11 s; on the reference's own workload shape (60-frame windows, ~130
tracks, 1e-3 rad noise, 10 % outliers) the per-frame L-BFGS turns it into other basins for some frames.  Control:
the reference-order oracle against itself started 1e-9 s away.

    python tests/measure/reassociation.py > profiles/r3_reassociation.json
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rssync_amd  # noqa: E402
from rssync_amd import synth  # noqa: E402
from rssync_amd.problem import bind  # noqa: E402
from oracle.oracle import OracleProblem  # noqa: E402


def hosttest():
    out = os.path.join(ROOT, "tests", "_build", "librssync_hosttest.so")
    srcs = [os.path.join(ROOT, "rs-sync_amd", "csrc", "sync_problem.cpp"), os.path.join(ROOT, "tests", "cpu_device", "rship_cpu.cpp")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", out] + srcs)
    return bind(ctypes.CDLL(out))


def stats(x):
    x = np.abs(np.asarray(x, float))
    return {"median": float(np.median(x)), "p90": float(np.percentile(x, 90)), "max": float(x.max())}


def main():
    lib = hosttest()
    out = {"what": __doc__.split("\n")[0], "scenes": {}}
    for name, kw, F, N, window, n_win in (
            ("reference_workload_noisy", {}, 400, 130, 60, 24),
            ("config1_noisy", {}, 64, 256, 63, 1),
            ("reference_workload_clean", {"noise": 0.0, "outliers": 0.0}, 200, 130, 60, 8)):
        gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=31)
        frames = list(synth.make_frames(gyro, 0, F, N, seed=31, **kw))
        dev = rssync_amd.SyncProblem(seed=99, max_outer_iters=400, _lib=lib)
        ora = OracleProblem(seed=99, max_outer_iters=400, threads=os.cpu_count() or 1, faithful=False)
        ora2 = OracleProblem(seed=99, max_outer_iters=400, threads=os.cpu_count() or 1, faithful=False)
        for p in (dev, ora, ora2):
            p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
            for fr in frames:
                p.SetTrackResult(*fr)
        d_dev, d_ora, d_ctl, it_dev, it_ora, first_rows = [], [], [], [], [], []
        for w in range(n_win):
            b = w * ((F - window - 1) // max(n_win - 1, 1)) if n_win > 1 else 0
            e = b + window
            d0 = ora.PreSync(0.0, b, e, 0.002, 0.1)[1]
            co, do, tro = ora.sync_trace(d0, b, e, 0.0, 0.1)
            win = ora.last_init_winners()
            ora2.set_init_override(win)
            c2, d2, tr2 = ora2.sync_trace(d0 + 1e-9, b, e, 0.0, 0.1)
            dev.set_init_override(win)
            cd, dd = dev.Sync(d0, b, e, 0.0, 0.1)
            trd = dev.sync_trace()
            d_dev.append(dd - do)
            d_ctl.append(d2 - do)
            d_ora.append(do)
            it_dev.append(len(trd))
            it_ora.append(len(tro))
            # the first outer iteration, before anything has been amplified: loss and derivative at the look-ahead point
            first_rows.append([abs(trd[0, 2] - tro[0, 2]) / abs(tro[0, 2]), abs(trd[0, 3] - tro[0, 3]) / max(abs(tro[0, 3]), 1e-300)])
        fr_ = np.asarray(first_rows)
        out["scenes"][name] = {
            "frames_per_window": window + 1, "tracks": N, "windows": n_win,
            "device_order_minus_reference_order_s": stats(d_dev),
            "control_reference_order_started_1e-9_s_away_s": stats(d_ctl),
            "first_iteration_loss_rel": float(fr_[:, 0].max()), "first_iteration_derivative_rel": float(fr_[:, 1].max()),
            "outer_iterations": {"device_order": it_dev, "reference_order": it_ora},
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
