"""K2 on frames of 4096 tracks (size class 3) by gyro rate: the window the planner picks and the launch time (GPU box)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth
out = {}
for fs in (400.0, 800.0, 2000.0, 4000.0):
    F, N = 512, 4096
    g = synth.make_gyro(0, (F + 2) / synth.FPS, fs=fs, seed=3)
    h = rssync_amd.SyncProblem(seed=3, verbose=False)
    synth.fill(h, g, 0, F, N, seed=3)
    h.upload(); h.PreSync(0.0, 0, F, 0.0005, 0.2)
    h.profile(True); best = None
    for _ in range(3):
        h.profile_reset(); c, d = h.PreSync(0.0, 0, F, 0.0005, 0.2)
        ms = h.profile_get()["lmeds"]; ms = ms[1] / ms[0]
        best = ms if best is None else min(best, ms)
    w = h.window_info()
    out[str(int(fs))] = {"lmeds_ms": round(best, 3), "delay": d, "window_knots": w.get("presync_window_knots"), "dynamic": w.get("presync_window_dynamic"), "chunk": w.get("presync_chunk")}
    h.close()
print(json.dumps(out))
