"""The metric's own PreSync sweep at full size (BASELINE config 3's window: 4096 frames x 2048 tracks,
PreSync(0, 0, 4096, 0.0005, 0.2) = 800 candidates, core_private.cpp:61-90): the whole per-candidate cost curve of the
HIP path against the oracle's, the per-(frame, candidate) costs and winning hypotheses.  The oracle needs ~5 minutes
on 16 cores for its 6.7e9 ray-residuals (progress lines on stderr: the box kills a silent command).  GPU box.

    python tests/measure/gpu_config3_presync_parity.py > profiles/r4_config3_presync_parity.json
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
BLOCK = int(os.environ.get("BLOCK", 256))          # frames per oracle call (a progress line after each)
seed = 0x5EED0003
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=seed)
h = rssync_amd.SyncProblem(seed=seed)
o = OracleProblem(seed=seed, threads=os.cpu_count() or 1, faithful=False)
t = time.perf_counter()
for fr in synth.make_frames(g, 0, F, N, seed=seed):
    h.SetTrackResult(*fr)
    o.SetTrackResult(*fr)
for p in (h, o):
    p.SetGyroQuaternions(g.quats, g.fs, g.t0)
print("filled in %.1f s" % (time.perf_counter() - t), file=sys.stderr, flush=True)

t = time.perf_counter()
dh, ch, fch, bhh = h.presync_curve(0.0, 0, F, 0.0005, 0.2, per_frame=F)
th = time.perf_counter() - t
# the oracle block by block: the per-frame costs are what is compared; its curve is their sum over the frames in the
# reference's order (core_private.cpp:84-85 adds them under a mutex in whatever order the threads arrive)
fco = np.zeros_like(fch)
bho = np.zeros_like(bhh)
t = time.perf_counter()
for b in range(0, F, BLOCK):
    e = min(b + BLOCK, F)
    do, _, fc, bh = o.presync_curve(0.0, b, e, 0.0005, 0.2, per_frame=e - b)
    assert np.array_equal(dh, do)
    fco[:, b:e] = fc
    bho[:, b:e] = bh
    print("oracle frames %d..%d done, %.0f s" % (b, e, time.perf_counter() - t), file=sys.stderr, flush=True)
to = time.perf_counter() - t
co = fco.sum(axis=1)
rel = np.abs(ch - co) / co
frel = np.abs(fch - fco) / fco
same = (bhh == bho)
print(json.dumps({
    "what": "PreSync(0, 0, %d, 0.0005, 0.2) on %d frames x %d tracks (noise 1e-3 rad, 10 %% outliers): HIP path (fp32 search) vs the CPU oracle (fp64)" % (F, F, N),
    "frames": F, "tracks": N, "candidates": int(len(dh)), "ray_residuals": int(F) * int(N) * int(len(dh)),
    "hip_s": round(th, 3), "oracle_s": round(to, 1), "oracle_threads": os.cpu_count(),
    "argmin_delay": [float(dh[np.argmin(ch)]), float(dh[np.argmin(co)])],
    "argmin_index": [int(np.argmin(ch)), int(np.argmin(co))],
    "min_cost": [float(ch.min()), float(co.min())],
    "curve_rel_err": {"max": float(rel.max()), "median": float(np.median(rel))},
    "frame_cost_rel_err": {"max": float(frel.max()), "median": float(np.median(frel)), "p999": float(np.quantile(frel, 0.999))},
    "winning_hypothesis_identical": float(same.mean()),
    "frame_cost_rel_err_where_same_winner": {"max": float(frel[same].max()), "median": float(np.median(frel[same]))},
    "pairs": int(same.size)}))
