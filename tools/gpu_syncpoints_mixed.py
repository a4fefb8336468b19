"""The driver workload (98 sync points of 61 frames x ~130 tracks, PreSync + 4 x Sync each: tools/gpu_syncpoints.py) with
ONE frame of 600 tracks inserted -- and with one of 9000.  Rounds 2-4 picked the kernel family from the problem's largest
frame: one 513-track frame sent every window of the clip out of the one-wave kernels and the window executor (21 -> 34 ms,
profiles/r4_syncpoints.json).  With size classes (round 5) the frame runs its own class's kernels -- inside the executor as
a one-wave task in the four-wave association (kernels/exec_big.hpp).  GPU box:  python tools/gpu_syncpoints_mixed.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth

F = int(os.environ.get("F", 3000)); N = int(os.environ.get("N", 130))
WINDOW = int(os.environ.get("WINDOW", 60)); DIST = int(os.environ.get("DIST", 30))
SEED = 0x5EED0006
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=SEED)
pos = list(range(0, F - WINDOW - 1, DIST))


def problem(big_at=None, big_n=600):
    h = rssync_amd.SyncProblem(seed=SEED, verbose=False)
    for fr in range(F):
        n = big_n if fr == big_at else N
        h.SetTrackResult(*next(iter(synth.make_frames(g, fr, fr + 1, n, seed=SEED))))
    h.SetGyroQuaternions(g.quats, g.fs, g.t0)
    h.upload()
    return h


def timed(h, reps=3):
    h.sync_points(pos, WINDOW, 0.0, 0.001, 0.1)  # warm-up
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        c, d = h.sync_points(pos, WINDOW, 0.0, 0.001, 0.1)
        best = min(best, time.perf_counter() - t)
    return best, d


out = {"frames": F, "tracks": N, "window": WINDOW, "positions": len(pos)}
os.environ.pop("RSSYNC_EXECUTOR", None)
t, d_pure = timed(problem())
out["all_130_tracks_executor_s"] = round(t, 4)
for big_n in (600, 9000):
    p = problem(big_at=1500, big_n=big_n)
    t, d = timed(p)
    out["one_frame_of_%d_executor_s" % big_n] = round(t, 4)
    out["one_frame_of_%d_executor_runs" % big_n] = p.executor_stats()["runs"]
    os.environ["RSSYNC_EXECUTOR"] = "0"
    q = problem(big_at=1500, big_n=big_n)
    t, dc = timed(q)
    os.environ.pop("RSSYNC_EXECUTOR", None)
    out["one_frame_of_%d_chain_s" % big_n] = round(t, 4)
    out["one_frame_of_%d_executor_equals_chain" % big_n] = bool(np.array_equal(d, dc))
    # windows that do not hold the large frame return the bits of the all-130 problem (the frame's neighbours do not care)
    far = [w for w, p0 in enumerate(pos) if not (p0 <= 1500 <= p0 + WINDOW)]
    out["one_frame_of_%d_other_windows_unchanged" % big_n] = bool(np.array_equal(d[far], d_pure[far]))
out["round4_same_workload"] = {"all_130_executor_s": 0.0205, "one_513_track_frame_s": 0.034, "source": "profiles/r4_syncpoints.json, r4_one_wave_max.json"}
print(json.dumps(out))
