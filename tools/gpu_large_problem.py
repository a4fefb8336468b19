"""Robustness run on the GPU box: 60 000 frames x 130 tracks in one problem (PreSync with 800 candidates, Sync, then a
sweep of 6000 candidates that the host cuts into slices of the [candidates][frames] cost matrix), checked for the true
delay.  Beside it: `python bench.py --frames 16384 --steps 2 --warmup 1 --cpu-frames 0` (BASELINE config 4's whole window on
one GPU: 193 ms per step, 1.40e11 ray-residuals/s).

    python tools/gpu_large_problem.py
"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import rssync_amd
from rssync_amd import synth
F, N = 60000, 130
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=9)
h = rssync_amd.SyncProblem(seed=9, verbose=False, max_outer_iters=30)
t = time.time(); synth.fill(h, g, 0, F, N, seed=9); print("fill", round(time.time() - t, 1), flush=True)
t = time.time(); c, d = h.PreSync(0.0, 0, F, 0.0005, 0.2); print("presync", round(time.time() - t, 3), c, d, flush=True)
t = time.time(); c2, d2 = h.Sync(d, 0, F - 1, 0.0, 0.2); print("sync", round(time.time() - t, 3), c2, d2, len(h.sync_trace()), flush=True)
assert abs(d2 - synth.D_TRUE) < 1e-3
# a sweep long enough for the candidate slices (the [candidates][frames] matrix is cut at 256 MB)
t = time.time(); c3, d3 = h.PreSync(0.0, 0, F, 0.0001, 0.3); print("presync 6000 candidates", round(time.time() - t, 3), c3, d3, flush=True)
assert abs(d3 - d) <= 0.0005
print("ok")
