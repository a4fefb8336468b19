import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth
F, N = 48, int(sys.argv[1]) if len(sys.argv) > 1 else 130
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=11)
h = rssync_amd.SyncProblem(seed=31, max_outer_iters=60, verbose=False)
synth.fill(h, g, 0, F, N, seed=11)
try:
    print(h.Sync(0.036, 0, F - 1, 0.0, 0.1), len(h.sync_trace()))
except Exception as e:
    print("FAILED:", e)
