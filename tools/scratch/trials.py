import sys, time, json
import numpy as np
sys.path.insert(0, "/root/repo")
import rssync_amd
from rssync_amd import synth
F, N = 1024, 2048
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=3)
h = rssync_amd.SyncProblem(seed=3, verbose=False, max_outer_iters=20)
synth.fill(h, g, 0, F, N, seed=3)
c0, d0 = h.PreSync(0.0, 0, F, 0.0005, 0.2)
c1, d1 = h.Sync(d0, 0, F - 1, 0.0, 0.2)
tr = np.array(h.sync_trace())
print("trials per iteration:", tr[:, 5].tolist(), "t:", tr[:, 4].tolist(), "grad:", tr[:, 3].tolist())
