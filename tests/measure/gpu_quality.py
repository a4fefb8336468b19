"""The thesis' accuracy metric on a synthetic clock-drift scenario: delays at sync points, line
fit, RMSE (python/plot_sync.py) -- for the HIP library and, on the same inputs, the CPU oracle."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rssync_amd
from rssync_amd import synth, quality
from oracle.oracle import OracleProblem

F = int(os.environ.get("F", 600)); N = int(os.environ.get("N", 130))
WINDOW = int(os.environ.get("WINDOW", 60)); DIST = int(os.environ.get("DIST", 30))
DRIFT = float(os.environ.get("DRIFT", 2e-4)); SEED = 0x5EED0007
NOISE = float(os.environ.get("NOISE", 1e-3)); OUTL = float(os.environ.get("OUTLIERS", 0.1))
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=SEED)
frames = list(synth.make_frames(g, 0, F, N, seed=SEED, drift=DRIFT, noise=NOISE, outliers=OUTL))
pos = quality.sync_points_auto(0, F, WINDOW, DIST)


def fill(p):
    p.SetGyroQuaternions(g.quats, g.fs, g.t0)
    for fr in frames:
        p.SetTrackResult(*fr)
    return p


h = fill(rssync_amd.SyncProblem(seed=SEED))
t = time.perf_counter(); _, dh = h.sync_points(pos, WINDOW, 0.0, 0.001, 0.1); th = time.perf_counter() - t
o = fill(OracleProblem(seed=SEED, threads=os.cpu_count() or 1, faithful=False))
t = time.perf_counter()
do = []
for p0 in pos:
    d = o.PreSync(0.0, p0, p0 + WINDOW, 0.001, 0.1)[1]
    for _ in range(4):
        d = o.Sync(d, p0, p0 + WINDOW, 0.0, 0.1)[1]
    do.append(d)
to = time.perf_counter() - t
do = np.array(do)
mid = (np.array(pos) + WINDOW / 2) / synth.FPS
truth = synth.D_TRUE + DRIFT * mid
sh, ih, rh = quality.linear_fit_rmse(pos, 1e3 * dh)
so, io, ro = quality.linear_fit_rmse(pos, 1e3 * do)
print(json.dumps({"frames": F, "tracks": N, "window": WINDOW, "positions": len(pos), "drift_ms_per_frame": 1e3 * DRIFT / synth.FPS,
                  "hip": {"slope": sh, "intercept": ih, "rmse_ms": rh, "max_err_vs_truth_ms": float(1e3 * np.abs(dh - truth).max()), "s": round(th, 3)},
                  "oracle": {"slope": so, "intercept": io, "rmse_ms": ro, "max_err_vs_truth_ms": float(1e3 * np.abs(do - truth).max()), "s": round(to, 3)},
                  "max_abs_hip_minus_oracle_ms": float(1e3 * np.abs(dh - do).max())}))
