"""Full-size (BASELINE config 3) Sync: HIP path vs oracle on identical inputs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rssync_amd
from rssync_amd import synth
from oracle.oracle import OracleProblem
F, N = int(os.environ.get("F", 4096)), int(os.environ.get("N", 2048))
seed = 0x5EED0003
g = synth.make_gyro(0, (F + 2) / 30, seed=seed)
h = rssync_amd.SyncProblem(seed=seed, max_outer_iters=20)
o = OracleProblem(seed=seed, max_outer_iters=20, threads=16, faithful=False)
t = time.time()
for fr in synth.make_frames(g, 0, F, N, seed=seed):
    h.SetTrackResult(*fr); o.SetTrackResult(*fr)
for p in (h, o):
    p.SetGyroQuaternions(g.quats, g.fs, g.t0)
print(f"filled in {time.time()-t:.1f}s", flush=True)
d0 = 0.0365
t = time.time(); ch, dh = h.Sync(d0, 0, F - 1, 0.0, 0.2); th = time.time() - t
trh = h.sync_trace()
t = time.time(); co, do, tro = o.sync_trace(d0, 0, F - 1, 0.0, 0.2); to = time.time() - t
print(f"hip   : delay {dh:.7f} cost {ch:.4f} iters {len(trh)} ({th*1e3:.1f} ms)")
print(f"oracle: delay {do:.7f} cost {co:.4f} iters {len(tro)} ({to:.1f} s)")
n = min(len(trh), len(tro))
print("delay diff per iteration:", np.abs(trh[:n, 0] - tro[:n, 0]))
print("loss rel diff per iteration:", np.abs(trh[:n, 2] - tro[:n, 2]) / tro[:n, 2])
print("final |d_hip - d_oracle| =", abs(dh - do), " cost rel", abs(ch - co) / co)
