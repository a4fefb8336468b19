"""Full-size (BASELINE config 3) Sync: HIP path vs oracle on identical inputs (4096 frames x 2048 tracks, noise 1e-3 rad,
10 % outliers, outer iterations capped at 20; core_private.cpp:211-334).  One JSON object on stdout.  GPU box.

    python tests/measure/gpu_fullsize_parity.py > profiles/r4_fullsize_sync_parity.json
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rssync_amd  # noqa: E402
from rssync_amd import synth  # noqa: E402
from oracle.oracle import OracleProblem  # noqa: E402

F, N = int(os.environ.get("F", 4096)), int(os.environ.get("N", 2048))
seed = 0x5EED0003
g = synth.make_gyro(0, (F + 2) / 30, seed=seed)
h = rssync_amd.SyncProblem(seed=seed, max_outer_iters=20)
o = OracleProblem(seed=seed, max_outer_iters=20, threads=16, faithful=False)
t = time.time()
for fr in synth.make_frames(g, 0, F, N, seed=seed):
    h.SetTrackResult(*fr)
    o.SetTrackResult(*fr)
for p in (h, o):
    p.SetGyroQuaternions(g.quats, g.fs, g.t0)
print("filled in %.1f s" % (time.time() - t), file=sys.stderr, flush=True)
d0 = 0.0365
h.upload()
h.Sync(d0, 0, F - 1, 0.0, 0.2)     # warm-up (advances the call counter: the timed call below samples stream + 1 ...)
h2 = rssync_amd.SyncProblem(seed=seed, max_outer_iters=20)   # ... so the compared call is a fresh problem's first, like the oracle's
for fr in synth.make_frames(g, 0, F, N, seed=seed):
    h2.SetTrackResult(*fr)
h2.SetGyroQuaternions(g.quats, g.fs, g.t0)
h2.upload()
t = time.time()
ch, dh = h2.Sync(d0, 0, F - 1, 0.0, 0.2)
th = time.time() - t
trh = h2.sync_trace()
t = time.time()
co, do, tro = o.sync_trace(d0, 0, F - 1, 0.0, 0.2)
to = time.time() - t
n = min(len(trh), len(tro))
print(json.dumps({
    "what": "Sync(0.0365, 0, %d, 0, 0.2), <= 20 outer iterations, %d frames x %d tracks, noise + 10 %% outliers: HIP (fp64 kernels) vs the CPU oracle" % (F - 1, F, N),
    "hip": {"delay": dh, "cost": ch, "outer_iterations": len(trh), "seconds_first_call_of_a_fresh_problem": round(th, 4)},
    "oracle": {"delay": do, "cost": co, "outer_iterations": len(tro), "seconds": round(to, 1), "threads": 16},
    "delay_abs_diff_s": abs(dh - do), "cost_rel_diff": abs(ch - co) / co,
    "delay_abs_diff_per_iteration_s": [float(x) for x in np.abs(trh[:n, 0] - tro[:n, 0])],
    "loss_rel_diff_per_iteration": [float(x) for x in np.abs(trh[:n, 2] - tro[:n, 2]) / tro[:n, 2]],
    "trials_per_iteration": [int(x) for x in trh[:, 5]],
    "truth_s": synth.D_TRUE}))
