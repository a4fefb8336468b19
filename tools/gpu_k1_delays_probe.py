"""K1 (five delays per launch) at 2048 x 2048: launch time for delays that keep the frames inside the gyro track and
for delays that put them (far) outside it.  python tools/gpu_k1_delays_probe.py  (GPU box)"""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, rssync_amd
from rssync_amd import synth
F, N = 2048, 2048
gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=0x5EED0001)
p = rssync_amd.SyncProblem(seed=0x5EED, max_outer_iters=20)
synth.fill(p, gyro, 0, F, N, seed=0x5EED0003)
p.init_motion(0.037, 0, F - 1)
p.profile(True)
for name, delays in (("in-range x5", [0.037, 0.0371, 0.0369, 0.03705, 0.03695]),
                     ("far x5", [-5000.0, -500.0, -50.0, 100.0, 1000.0]),
                     ("bench-like", [0.037 - 4000, 0.037 - 400, 0.037 - 40, 0.037 - 4, 0.037 - 0.4]),
                     ("near-out x5", [-0.3, -1.0, 3.0, 80.0, -70.0])):
    p.loss(delays)
    p.profile_reset()
    for _ in range(5):
        p.loss(delays)
    n, ms = p.profile_get()["loss"]
    print(json.dumps({"case": name, "ms_per_launch": ms / n}), flush=True)
