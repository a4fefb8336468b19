"""PreSync + Sync time against the gyro sample rate (the LDS spline window holds 64 knots; above
~900 Hz a frame spans more and the kernels read the table from L2).  GPU box."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth
F, N = int(os.environ.get("F", 512)), int(os.environ.get("N", 2048))
out = {}
for fs in (400.0, 800.0, 1600.0, 3200.0):
    g = synth.make_gyro(0, (F + 2) / synth.FPS, fs=fs, seed=3)
    h = rssync_amd.SyncProblem(seed=3, max_outer_iters=10)
    synth.fill(h, g, 0, F, N, seed=3)
    h.upload()
    h.PreSync(0.0, 0, F, 0.0005, 0.2)
    t = time.perf_counter(); c, d = h.PreSync(0.0, 0, F, 0.0005, 0.2); t_pre = time.perf_counter() - t
    h.Sync(d, 0, F - 1, 0.0, 0.2)
    t = time.perf_counter(); h.Sync(d, 0, F - 1, 0.0, 0.2); t_sync = time.perf_counter() - t
    out[int(fs)] = {"presync_ms": round(1e3 * t_pre, 2), "sync_ms": round(1e3 * t_sync, 2), "delay": d,
                    "iters": len(h.sync_trace())}
print(json.dumps({"frames": F, "tracks": N, "by_gyro_hz": out}))
