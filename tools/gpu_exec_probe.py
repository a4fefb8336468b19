"""The window executor on the reference workload shape with K windows (executor diagnostics: a stats build,
RSSYNC_LIB=rs-sync_amd/_variants/lib_execstats.so, prints wave-time per activity on stderr).
python tools/gpu_exec_probe.py [windows]      BIG_AT=frame BIG_N=tracks: one larger frame (kernels/exec_big.hpp); POS0=index of the
first position taken"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rssync_amd
from rssync_amd import synth
F, N, WINDOW, DIST = 3000, 130, 60, 30
K = int(sys.argv[1]) if len(sys.argv) > 1 else 98
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=0x5EED0006)
POS0 = int(os.environ.get("POS0", 0))
pos = list(range(0, F - WINDOW - 1, DIST))[POS0:POS0 + K]
BIG_AT, BIG_N = int(os.environ.get("BIG_AT", -1)), int(os.environ.get("BIG_N", 600))
h = rssync_amd.SyncProblem(seed=0x5EED0006, verbose=False)
h.SetGyroQuaternions(g.quats, g.fs, g.t0)
for fr in range(F):
    h.SetTrackResult(*next(iter(synth.make_frames(g, fr, fr + 1, BIG_N if fr == BIG_AT else N, seed=0x5EED0006))))
h.upload()
h.sync_points(pos, WINDOW, 0.0, 0.001, 0.1)
sys.stderr.write("---- timed call ----\n"); sys.stderr.flush()
t = time.perf_counter(); h.sync_points(pos, WINDOW, 0.0, 0.001, 0.1); dt = time.perf_counter() - t
iters = [len(h.window_trace(w)) for w in range(len(pos))]
print("windows", len(pos), "batched_s", round(dt, 5), "iterations mean/max", float(np.mean(iters)), int(np.max(iters)),
      "us per iteration of the longest window", round(dt * 1e6 / max(iters), 1), flush=True)
