#!/usr/bin/env python3
"""A small client for a marker trace: with RSSYNC_ROCTX=1 the library brackets its public calls and its launch kinds with
roctx ranges (rs-sync_amd/csrc/roctx_ranges.hpp), so

    cd /tmp && RSSYNC_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats -d out -o tr --output-format csv -- \
        python3 /path/to/repo/examples/roctx_demo.py

shows which kernels belong to PreSync, to each Sync call and to a batched sync_points call (the reference driver's loop,
core_testcode.cpp:303-316).  Without the variable nothing is loaded and nothing is marked."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth

F, N = 200, 130
g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=5)
p = rssync_amd.SyncProblem(seed=5, verbose=False)
synth.fill(p, g, 0, F, N, seed=5)
d = p.PreSync(0.0, 0, 60, 0.001, 0.1)[1]
for _ in range(2):
    c, d = p.Sync(d, 0, 60, 0.0, 0.1)
pos = list(range(0, F - 61, 30))
costs, delays = p.sync_points(pos, 60, 0.0, 0.001, 0.1)
print("PreSync + 2 x Sync: delay %.6f; %d sync points: median delay %.6f" % (d, len(pos), sorted(delays)[len(pos) // 2]))
