#!/usr/bin/env python3
"""What rounding-level differences do to Sync on noisy data -- a pure CPU measurement, per scene of tests/noisy_scenes.py.

Two implementations of the same algorithm on the same inputs, started from the SAME motion estimates (GuessMotion's
winning hypotheses are transplanted, so the fp32/fp64 search plays no part):

  device order     tests/cpu_device/rship_cpu.cpp behind the product's host solver: the kernels' own fp64
                   arithmetic and summation order (bit-identical to the GPU: tests/test_gpu_bitexact.py)
  reference order  oracle/rssync_oracle.c: sequential sums, libm log1p, plain a*b+c expressions

They differ only in rounding (association of the sums over rows, which products are fused, the last bits of
log1p).  On noise-free scenes that is 1e-11 s; on the reference's own workload shape (60-frame windows, ~130
tracks, 1e-3 rad noise, 10 % outliers) the per-frame L-BFGS turns it into other basins for some frames.  Control:
the reference-order oracle against itself started 1e-9 s away.

The file this writes is what the tests' tolerances are read from (noisy_scenes.bound_s), and
tests/test_reassociation.py::test_the_measurement_file_is_current recomputes two scenes and demands the same numbers:
re-run after ANY change to sync_math.hpp / device_math.hpp / the stand-in / the oracle.

    python tests/measure/reassociation.py > profiles/r5_reassociation.json
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import noisy_scenes as ns  # noqa: E402
from rssync_amd.problem import bind  # noqa: E402


def hosttest():
    out = os.path.join(ROOT, "tests", "_build", "librssync_hosttest.so")
    srcs = [os.path.join(ROOT, "rs-sync_amd", "csrc", "sync_problem.cpp"), os.path.join(ROOT, "tests", "cpu_device", "rship_cpu.cpp")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", out] + srcs)
    return bind(ctypes.CDLL(out))


def measure(scene, lib):
    dev, ora, ctl = scene.device(lib), scene.oracle(), scene.oracle()
    recs = ns.run_scene(scene, dev, ora, control=ctl)
    first = []
    trace_d = 0.0
    for r in recs:
        td, to = r["trace_dev"], r["trace_ora"]
        # the first outer iteration, before anything has been amplified: loss and derivative at the look-ahead point
        first.append([abs(td[0, 2] - to[0, 2]) / abs(to[0, 2]), abs(td[0, 3] - to[0, 3]) / max(abs(to[0, 3]), 1e-300)])
        n = min(len(td), len(to))
        trace_d = max(trace_d, float(np.abs(td[:n, 0] - to[:n, 0]).max()))
    first = np.asarray(first)
    return {
        "what": (scene.__doc__ or "").strip(),
        "calls": len(recs), "frames_per_call": scene.calls[0][2] - scene.calls[0][1] + 1, "tracks": int(len(scene.frames[0][1])),
        "max_outer_iters": scene.max_outer_iters,
        "device_order_minus_reference_order_s": ns.stats([r["d_dev"] - r["d_ora"] for r in recs]),
        "control_reference_order_started_1e-9_s_away_s": ns.stats([r["d_ctl"] - r["d_ora"] for r in recs]),
        "delay_after_each_outer_iteration_max_abs_s": trace_d,
        "cost_rel": float(max(abs(r["c_dev"] - r["c_ora"]) / abs(r["c_ora"]) for r in recs)),
        "first_iteration_loss_rel": float(first[:, 0].max()), "first_iteration_derivative_rel": float(first[:, 1].max()),
        "outer_iterations": {"device_order": [len(r["trace_dev"]) for r in recs], "reference_order": [len(r["trace_ora"]) for r in recs]},
        "delays_s": {"device_order": [r["d_dev"] for r in recs], "reference_order": [r["d_ora"] for r in recs]},
    }


def main():
    lib = hosttest()
    out = {"what": __doc__.split("\n")[0], "north_star_s": ns.NORTH_STAR_S, "scenes": {}}
    for name, make in ns.SCENES.items():
        scene = make()
        scene.__doc__ = make.__doc__ if make.__doc__ else (ns.big_frames.__doc__ if name.startswith("big_") else "")
        out["scenes"][name] = measure(scene, lib)
        m = out["scenes"][name]["device_order_minus_reference_order_s"]["max"]
        out["scenes"][name]["asserted_bound_s"] = ns.NORTH_STAR_S if m < ns.NORTH_STAR_S else 2.5 * m
        print(name, out["scenes"][name]["device_order_minus_reference_order_s"], file=sys.stderr, flush=True)
    # the reference's workload shape with noise on a sample that supports a distribution: 5 clips x 41 windows
    per_seed, dev_all, ctl_all = {}, [], []
    for sd in ns.POOLED_SEEDS:
        scene = ns.reference_workload_noisy_clip(sd)
        recs = ns.run_scene(scene, scene.device(lib), scene.oracle(), control=scene.oracle())
        dev = [r["d_dev"] - r["d_ora"] for r in recs]
        ctl = [r["d_ctl"] - r["d_ora"] for r in recs]
        per_seed[str(sd)] = {"windows": len(recs), "device_order_minus_reference_order_s": ns.stats(dev),
                             "control_reference_order_started_1e-9_s_away_s": ns.stats(ctl),
                             "delays_s": {"device_order": [r["d_dev"] for r in recs], "reference_order": [r["d_ora"] for r in recs]}}
        dev_all += dev
        ctl_all += ctl
        print("pooled seed", sd, per_seed[str(sd)]["device_order_minus_reference_order_s"], file=sys.stderr, flush=True)
    a = np.abs(np.asarray(dev_all))
    out["pooled_reference_workload_noisy"] = {
        "what": "5 independent clips (data seeds %s) x %d windows of 61 x 130 half a window apart, noise 1e-3 rad, 10 %% outliers: "
                "PreSync (oracle) then one Sync call per window on both sides from the oracle's GuessMotion winners" % (list(ns.POOLED_SEEDS), ns.POOLED_WINDOWS_PER_SEED),
        "windows": len(dev_all),
        "device_order_minus_reference_order_s": ns.stats(dev_all),
        "control_reference_order_started_1e-9_s_away_s": ns.stats(ctl_all),
        "fraction_within_north_star_1e-4_s": float((a <= ns.NORTH_STAR_S).mean()),
        "spread_over_seeds_of_the_median_s": [per_seed[str(sd)]["device_order_minus_reference_order_s"]["median"] for sd in ns.POOLED_SEEDS],
        "spread_over_seeds_of_the_max_s": [per_seed[str(sd)]["device_order_minus_reference_order_s"]["max"] for sd in ns.POOLED_SEEDS],
        "per_seed": per_seed}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
