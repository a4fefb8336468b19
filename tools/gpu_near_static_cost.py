#!/usr/bin/env python3
"""What the fp64 rows of near-static frames cost (kernels/lmeds.hpp, "fp64 rows"; reference: core_private.cpp:19-28 is
double).  One PreSync sweep -- 800 candidates, +-200 ms -- over F frames of N tracks for three kinds of footage:
  ordinary      translation 0.05 m per frame, hand-held rotation: no pair is flagged, the fp64 form never runs
  slow_pan      translation 5e-5 m per frame, hand-held rotation: |P| is tiny only where the candidate delay is within ~0.2 ms
                of the truth -- a handful of candidates per frame take the fp64 form
  tripod        translation 5e-5 m AND the rotation scaled down 1e4-fold: every candidate of every frame is near-static
                (such footage has nothing to synchronise on; the worst case for the mechanism)
GPU box:  python tools/gpu_near_static_cost.py > profiles/r6_near_static_cost.json"""
import json
import os
import sys
import time

import numpy as np
from scipy.interpolate import CubicSpline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rssync_amd
from rssync_amd import synth

F, N = int(os.environ.get("F", 512)), int(os.environ.get("N", 2048))


def still_gyro(g, scale):
    q = synth.integrate_gyro(g.rates * scale, np.full(len(g.rates), 1.0 / g.fs))
    return synth.Gyro(fs=g.fs, t0=g.t0, quats=q, times=g.times, rates=g.rates * scale,
                      spline=CubicSpline(np.arange(len(q), dtype=np.float64), q, axis=0, bc_type="natural"))


out = {"frames": F, "tracks": N, "candidates": 800, "cases": []}
g0 = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=5)
for name, gyro, translation, noise in (("ordinary", g0, 0.05, 1e-3), ("slow_pan", g0, 5e-5, 1e-6), ("tripod", still_gyro(g0, 1e-4), 5e-5, 1e-6)):
    res = {}
    for mode in ("fp64_rows", "fp32_rows_only"):
        if mode == "fp32_rows_only":
            os.environ["RSSYNC_NO_FP64_ROWS"] = "1"
        try:
            p = rssync_amd.SyncProblem(seed=11)
        finally:
            os.environ.pop("RSSYNC_NO_FP64_ROWS", None)
        p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
        for fr in synth.make_frames(gyro, 0, F, N, seed=5, noise=noise, outliers=0.1, translation=translation):
            p.SetTrackResult(*fr)
        p.upload()
        p.PreSync(0.0, 0, F, 0.0005, 0.2)          # warm-up
        p.profile(True)
        p.profile_reset()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            r = p.PreSync(0.0, 0, F, 0.0005, 0.2)
        wall = (time.perf_counter() - t0) / reps
        prof = p.profile_get()
        st = p.near_static_stats()
        res[mode] = {"presync_wall_ms": round(wall * 1e3, 3), "lmeds_kernel_ms_per_call": round(prof["lmeds"][1] / reps, 3),
                     "lmeds_launches_per_call": prof["lmeds"][0] / reps, "fp64_pairs_per_call": st["pairs"] / (reps + 1),
                     "share_of_pairs": st["pairs"] / (reps + 1) / (F * 800), "delay": r[1], "cost": r[0]}
    # the same sweep on the first FO frames against the oracle (fp64 throughout): which delay, how close the minimum's cost
    FO = min(F, 48)
    from oracle.oracle import OracleProblem
    o = OracleProblem(seed=11, threads=min(os.cpu_count() or 1, 16), faithful=False)
    o.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr in synth.make_frames(gyro, 0, FO, N, seed=5, noise=noise, outliers=0.1, translation=translation):
        o.SetTrackResult(*fr)
    ro = o.PreSync(0.0, 0, FO, 0.0005, 0.2)
    chk = {"frames": FO, "oracle": {"delay": ro[1], "cost": ro[0]}}
    for mode in ("fp64_rows", "fp32_rows_only"):
        if mode == "fp32_rows_only":
            os.environ["RSSYNC_NO_FP64_ROWS"] = "1"
        try:
            p = rssync_amd.SyncProblem(seed=11)
        finally:
            os.environ.pop("RSSYNC_NO_FP64_ROWS", None)
        p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
        for fr in synth.make_frames(gyro, 0, FO, N, seed=5, noise=noise, outliers=0.1, translation=translation):
            p.SetTrackResult(*fr)
        rh = p.PreSync(0.0, 0, FO, 0.0005, 0.2)
        chk[mode] = {"delay": rh[1], "cost": rh[0], "same_delay": bool(rh[1] == ro[1]), "cost_rel": abs(rh[0] - ro[0]) / abs(ro[0])}
    out["cases"].append({"footage": name, "translation_m": translation, **res, "against_the_oracle": chk})
print(json.dumps(out, indent=1))
