"""Kernel throughput by SIZE CLASS: the same number of ray pairs (F x N constant) as frames of 130 ... 8192 tracks, PreSync
with 800 candidates + Sync -- what a ray costs in each kernel family (one wave per frame; four waves with 4 / 8 / 16 / 32
rows per thread).  The benchmark's class is 2048 tracks; this is the table for everybody else's tracker.  GPU box.

    python tools/gpu_by_class.py > profiles/r5_by_class.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd  # noqa: E402
from rssync_amd import synth  # noqa: E402

RAYS = int(os.environ.get("RAYS", 1 << 21))       # ray pairs per problem (a quarter of the benchmark's)
SIZES = [int(x) for x in os.environ.get("SIZES", "130,256,512,1024,2048,4096,8192").split(",")]
REPS = int(os.environ.get("REPS", 3))
CAND = 800


def one(N):
    F = max(8, RAYS // N)
    g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=3)
    h = rssync_amd.SyncProblem(seed=3, max_outer_iters=10, verbose=False)
    synth.fill(h, g, 0, F, N, seed=3)
    h.upload()
    c, d = h.PreSync(0.0, 0, F, 0.0005, 0.2)
    h.Sync(d, 0, F - 1, 0.0, 0.2)
    h.profile(True)
    best = None
    for _ in range(REPS):
        h.profile_reset()
        t = time.perf_counter()
        c, d = h.PreSync(0.0, 0, F, 0.0005, 0.2)
        t_pre = time.perf_counter() - t
        t = time.perf_counter()
        c2, d2 = h.Sync(d, 0, F - 1, 0.0, 0.2)
        t_sync = time.perf_counter() - t
        cur = {k: v[1] / v[0] for k, v in h.profile_get().items() if v[0]}
        cur["_presync_ms"], cur["_sync_ms"] = 1e3 * t_pre, 1e3 * t_sync
        best = cur if best is None else {k: min(best[k], cur[k]) for k in cur}
    its = len(h.sync_trace())
    out = {"frames": F, "tracks": N, "ray_pairs": F * N, "presync_ms": round(best["_presync_ms"], 3), "sync_ms": round(best["_sync_ms"], 3),
           "lmeds_kernel_ms": round(best["lmeds"], 4),
           "presync_ray_residuals_per_s": F * N * CAND / (best["lmeds"] * 1e-3),
           "sync_outer_iterations": its,
           "ms_per_launch": {k: round(v, 4) for k, v in best.items() if not k.startswith("_")},
           "presync_delay": d, "sync_delay": d2, "windows": h.window_info()}
    h.close()
    return out


if __name__ == "__main__":
    res = {}
    for N in SIZES:
        res[str(N)] = one(N)
        print("%d tracks done: lmeds %.3f ms" % (N, res[str(N)]["lmeds_kernel_ms"]), file=sys.stderr)
    ref = res.get("2048")
    if ref:
        for r in res.values():
            r["presync_rate_relative_to_2048_tracks"] = round(r["presync_ray_residuals_per_s"] / ref["presync_ray_residuals_per_s"], 3)
    print(json.dumps({"what": __doc__.split("\n\n")[0], "ray_pairs_per_problem": RAYS, "candidates": CAND, "by_tracks": res}, indent=1))
