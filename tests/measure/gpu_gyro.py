"""Gyro pipeline on the device (rates -> orientations -> grid -> spline table): wall time of the setter, kernel
time, and the oracle's sequential route beside it.  Run on the GPU box; prints one JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rssync_amd
from oracle import oracle

out = {}
for n in (55_000, 1_000_000):
    rng = np.random.default_rng(n)
    t = 5.0 + np.cumsum(rng.uniform(0.0023, 0.0027, n))
    r = 0.6 * rng.standard_normal((n, 3))
    h = rssync_amd.SyncProblem(verbose=False)
    h.set_gyro_rates(t, r)                      # warm-up (allocations)
    walls = []
    for _ in range(9):
        t0 = time.perf_counter()
        h.set_gyro_rates(t, r, "yXz")
        walls.append(time.perf_counter() - t0)
    wall = float(np.median(walls))   # a fresh process sometimes spends ~2 ms in its first calls
    h.profile(True); h.profile_reset()
    h.set_gyro_rates(t, r, "yXz")
    prof = h.profile_get()["gyro"]
    t0 = time.perf_counter()
    q, us = oracle.integrate_gyro(t, r, "yXz")
    o = oracle.OracleProblem()
    o.SetGyroQuaternionsTimestamped(us, q)
    o.spline_eval(1.0)                          # forces the oracle's spline solve
    cpu = time.perf_counter() - t0
    err = float(np.abs(h.gyro_knots() - o.gyro_knots()).max())
    out[f"n={n}"] = {"knots": int(h.gyro_info()[2]), "setter_wall_ms": round(wall * 1e3, 3),
                     "kernels_ms": round(prof[1], 3), "profiled_regions": prof[0],
                     "oracle_sequential_ms": round(cpu * 1e3, 2), "max_knot_diff": err}
print(json.dumps(out))
