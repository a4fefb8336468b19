"""Where should the one-wave kernels (K2s, loss64_small, the window executor) stop and the four-wave kernels take over?
98 sync points of 61 frames x N tracks (PreSync + 4 x Sync, the reference driver's loop) for N around and above 256,
with RSSYNC_ONE_WAVE_MAX = 256 (rounds 1-4: tile kernels and the launch chain above 256 tracks) and = 512.  GPU box.

    python tools/gpu_one_wave_max.py > profiles/r4_one_wave_max.json
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd  # noqa: E402
from rssync_amd import synth  # noqa: E402

F, W, D = 3000, 60, 30
out = {"what": __doc__.split("\n")[0], "frames": F, "window": W, "positions": len(range(0, F - W - 1, D)), "by_tracks": {}}
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=6)
pos = list(range(0, F - W - 1, D))
for N in [int(x) for x in os.environ.get("TRACKS", "200,256,300,364,448,512").split(",")]:
    row = {}
    for limit in (256, 512):
        os.environ["RSSYNC_ONE_WAVE_MAX"] = str(limit)
        h = rssync_amd.SyncProblem(seed=6, verbose=False)
        synth.fill(h, g, 0, F, N, seed=6)
        h.upload()
        h.sync_points(pos, W, 0.0, 0.001, 0.1)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            c, d = h.sync_points(pos, W, 0.0, 0.001, 0.1)
            best = min(best, time.perf_counter() - t)
        iters = [len(h.window_trace(w)) for w in range(len(pos))]   # a position's four Sync calls together
        # the same with the outer iterations capped at 25 per call: the two kernel families round differently, so their
        # windows take different numbers of iterations, and an uncapped call is as long as its LONGEST window
        hb = rssync_amd.SyncProblem(seed=6, verbose=False, max_outer_iters=25)
        synth.fill(hb, g, 0, F, N, seed=6)
        hb.upload()
        hb.sync_points(pos, W, 0.0, 0.001, 0.1)
        best_b = 1e9
        for _ in range(3):
            t = time.perf_counter()
            hb.sync_points(pos, W, 0.0, 0.001, 0.1)
            best_b = min(best_b, time.perf_counter() - t)
        iters_b = [len(hb.window_trace(w)) for w in range(len(pos))]
        hb.close()
        h.profile(True)
        h.profile_reset()
        h.pre_sync_windows(0.0, pos, [p + W for p in pos], 0.001, 0.1)
        k2 = h.profile_get()["lmeds"]
        row[str(limit)] = {"sync_points_s": round(best, 4), "outer_iterations_per_position": {"mean": round(float(np.mean(iters)), 1), "max": int(np.max(iters))},
                           "sync_points_capped_s": round(best_b, 4), "outer_iterations_per_position_capped": {"mean": round(float(np.mean(iters_b)), 1), "max": int(np.max(iters_b))},
                           "presync_kernel_ms": round(k2[1] / max(k2[0], 1), 3),
                           "executor_runs": h.executor_stats()["runs"], "median_abs_err_ms": float(np.median(np.abs(d - synth.D_TRUE)) * 1e3),
                           "delays": [float(x) for x in d[:4]]}
        h.close()
    out["by_tracks"][N] = row
    print(N, {k: (v["sync_points_s"], v["outer_iterations_per_position"]["max"], v["sync_points_capped_s"], v["presync_kernel_ms"], v["executor_runs"]) for k, v in row.items()}, file=sys.stderr, flush=True)
os.environ.pop("RSSYNC_ONE_WAVE_MAX", None)
print(json.dumps(out, indent=1))
