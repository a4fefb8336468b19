#!/usr/bin/env python3
"""K3 (opt_motion64_kernel) at the bench's frame size: L-BFGS iterations / evaluations per frame and the launch time
of the calls an outer Sync loop makes (GuessMotion, then the motion optimisation at a sequence of delays that
converges like Sync's).  python tools/gpu_k3_probe.py [frames] [tracks]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=0x5EED0001)
p = rssync_amd.SyncProblem(seed=0x5EED, max_outer_iters=20)
synth.fill(p, gyro, 0, F, N, seed=0x5EED0003)
p.profile(True)
d = synth.D_TRUE + 4e-4
rows = []
p.init_motion(d, 0, F - 1)
for step in (0.0, -2e-4, -1e-4, -5e-5, -2e-5, -1e-5, -3e-6, -1e-6):
    d += step
    p.profile_reset()
    M, k, it, ev = p.opt_motion(d)
    n, ms = p.profile_get()["motion"]
    rows.append(dict(delay=d, iters_per_frame=it / F, evals_per_frame=ev / F, launches=n, ms=ms,
                     us_per_frame_eval=ms * 1e3 / max(ev, 1)))
    print(json.dumps(rows[-1]), flush=True)
