"""BASELINE config 2 at full size (1024 frames x 1024 tracks, PreSync(0, 0, 1024, 0.0005, 0.2) = 800
candidates): the whole per-candidate cost curve of the HIP path against the oracle's, plus the
per-(frame, candidate) winning hypotheses.  ~1 minute of oracle time on 16 cores.  GPU box."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rssync_amd
from rssync_amd import synth
from oracle.oracle import OracleProblem

F, N = int(os.environ.get("F", 1024)), int(os.environ.get("N", 1024))
seed = 0x5EED0002
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=seed)
h = rssync_amd.SyncProblem(seed=seed)
o = OracleProblem(seed=seed, threads=os.cpu_count() or 1, faithful=False)
for fr in synth.make_frames(g, 0, F, N, seed=seed):
    h.SetTrackResult(*fr); o.SetTrackResult(*fr)
for p in (h, o):
    p.SetGyroQuaternions(g.quats, g.fs, g.t0)
t = time.perf_counter(); dh, ch, fch, bhh = h.presync_curve(0.0, 0, F, 0.0005, 0.2, per_frame=F); th = time.perf_counter() - t
t = time.perf_counter(); do, co, fco, bho = o.presync_curve(0.0, 0, F, 0.0005, 0.2, per_frame=F); to = time.perf_counter() - t
assert np.array_equal(dh, do)
rel = np.abs(ch - co) / co
frel = np.abs(fch - fco) / fco
same = (bhh == bho)
print(json.dumps({"frames": F, "tracks": N, "candidates": int(len(dh)), "hip_s": round(th, 3), "oracle_s": round(to, 1),
                  "argmin_delay": [float(dh[np.argmin(ch)]), float(do[np.argmin(co)])],
                  "curve_rel_err": {"max": float(rel.max()), "median": float(np.median(rel))},
                  "frame_cost_rel_err": {"max": float(frel.max()), "median": float(np.median(frel)), "p999": float(np.quantile(frel, 0.999))},
                  "winning_hypothesis_identical": float(same.mean()),
                  "pairs": int(same.size)}))
