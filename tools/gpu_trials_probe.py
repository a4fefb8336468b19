#!/usr/bin/env python3
"""Which trial of the line search Sync's outer iterations accept (trace column 5 = trials evaluated until the Armijo
test held), at the bench's size and on the reference workload's small windows.  python tools/gpu_trials_probe.py"""
import json
import os
import sys
from collections import Counter

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth

F, N = int(os.environ.get("F", 2048)), int(os.environ.get("N", 2048))
gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=0x5EED0001)
p = rssync_amd.SyncProblem(seed=0x5EED, max_outer_iters=20)
synth.fill(p, gyro, 0, F, N, seed=0x5EED0003)
c, d = p.PreSync(0.0, 0, F, 0.0005, 0.2)
rows = []
for call in range(4):
    c, d = p.Sync(d, 0, F - 1, 0.0, 0.2)
    tr = p.sync_trace()
    rows.append([int(x) for x in tr[:, 5]])
print(json.dumps({"bench_size": [F, N], "trials_per_iteration_by_call": rows}))
F2, N2, W, DIST = 3000, 130, 60, 30
g = synth.make_gyro(0, (F2 + 2) / synth.FPS, seed=0x5EED0006)
h = rssync_amd.SyncProblem(seed=0x5EED0006)
synth.fill(h, g, 0, F2, N2, seed=0x5EED0006)
pos = list(range(0, F2 - W - 1, DIST))
h.sync_points(pos, W, 0.0, 0.001, 0.1)
cnt = Counter()
for w in range(len(pos)):
    cnt.update(int(x) for x in h.window_trace(w)[:, 5])
print(json.dumps({"sync_points": len(pos), "trials_histogram(last call)": dict(sorted(cnt.items()))}))
