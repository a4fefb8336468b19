"""one batched sync_points call on the reference workload shape, timed (executor diagnostics)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth
F, N, WINDOW, DIST = 3000, 130, 60, 30
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=0x5EED0006)
pos = list(range(0, F - WINDOW - 1, DIST))
h = rssync_amd.SyncProblem(seed=0x5EED0006, verbose=False)
synth.fill(h, g, 0, F, N, seed=0x5EED0006)
h.upload()
h.sync_points(pos, WINDOW, 0.0, 0.001, 0.1)
t = time.perf_counter(); h.sync_points(pos, WINDOW, 0.0, 0.001, 0.1); print("batched", time.perf_counter() - t, flush=True)
