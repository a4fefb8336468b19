"""First-contact check of the HIP path against the oracle on a small problem (prints numbers)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rssync_amd  # noqa: E402
from rssync_amd import synth  # noqa: E402
from oracle.oracle import OracleProblem  # noqa: E402

F, N = int(os.environ.get("F", 64)), int(os.environ.get("N", 256))
SEED = 123
g = synth.make_gyro(0, (F + 2) / 30, seed=1)
o = OracleProblem(seed=SEED, threads=16, faithful=False)
h = rssync_amd.SyncProblem(seed=SEED)
synth.fill(o, g, 0, F, N, seed=1)
synth.fill(h, g, 0, F, N, seed=1)

# 1. residual matrix
for fr, d in [(0, 0.0), (3, 0.0371), (F - 1, -0.15)]:
    Po = o.problem_matrix(fr, d)
    Ph, dPh = h.problem_matrix(fr, d, N, deriv=True)
    eps = 1e-6
    dPo = (o.problem_matrix(fr, d + eps) - o.problem_matrix(fr, d - eps)) / (2 * eps)
    print(f"P frame {fr} d {d}: max abs err {np.abs(Ph - Po).max():.3e} (|P| max {np.abs(Po).max():.3e}); "
          f"dP max err {np.abs(dPh - dPo).max():.3e} (|dP| max {np.abs(dPo).max():.3e})")

# 2. presync curve with per-frame matrices
t = time.time()
do, co, fco, bho = o.presync_curve(0.0, 0, F, 0.002, 0.2, per_frame=F)
t_o = time.time() - t
t = time.time()
dh, ch, fch, bhh = h.presync_curve(0.0, 0, F, 0.002, 0.2, per_frame=F)
t_h = time.time() - t
print(f"presync: {len(do)} cands; oracle {t_o:.2f}s hip {t_h:.3f}s")
print("delays equal:", np.array_equal(do, dh))
same = (bho == bhh)
print(f"hypothesis index agreement {same.mean():.4f}")
rel = np.abs(fch - fco) / np.abs(fco)
print(f"frame cost rel err where same hyp: max {rel[same].max():.3e} median {np.median(rel[same]):.3e}")
print(f"total cost rel err: max {np.abs(ch - co).max() / co.mean():.3e}; argmin oracle {np.argmin(co)} hip {np.argmin(ch)}")
print("PreSync:", o.PreSync(0.0, 0, F, 0.002, 0.2), h.PreSync(0.0, 0, F, 0.002, 0.2))

# 3. init motion / loss / opt motion
d0 = 0.036
Mh, kh = h.init_motion(d0, 0, F - 1)
Mo = np.array([o.guess_motion(f, d0, 200, 0x80000000)[0] for f in range(F)])
cosang = np.abs((Mh * Mo).sum(1))
print(f"init motion: |cos| min {cosang.min():.6f}; exact-sign agree {(np.sign(Mh[:,0])==np.sign(Mo[:,0])).mean():.3f}; k range {kh.min():.1f}..{kh.max():.1f}")
Lh, Gh = h.loss([d0, d0 + 1e-3, 0.0], grad=True)
Lo = np.zeros(3); Gn = np.zeros(3); Ga = np.zeros(3)
for j, dd in enumerate([d0, d0 + 1e-3, 0.0]):
    for f in range(F):
        L, dn, da, _ = o.loss(f, dd, Mh[f], kh[f])
        Lo[j] += L; Gn[j] += dn; Ga[j] += da
print("loss hip", Lh, "oracle", Lo, "rel", np.abs(Lh - Lo) / Lo)
print("grad hip", Gh, "oracle analytic", Ga, "numeric", Gn)
t = time.time()
M2, k2, its, evs = h.opt_motion(d0)
print(f"opt_motion: {time.time()-t:.4f}s iters {its} evals {evs}")
Lo2 = 0.0; Lh2 = h.loss([d0])[0]; Lor = 0.0
for f in range(F):
    Mo2, it, ev, fl = o.lbfgs_motion(f, d0, Mh[f], kh[f])
    Lor += fl
    Lo2 += o.loss(f, d0, M2[f], k2[f])[0]
print(f"after motion opt: hip loss {Lh2:.6f} (oracle eval of hip M {Lo2:.6f}); oracle lbfgs loss {Lor:.6f}")

# 4. Sync
o2 = OracleProblem(seed=SEED, threads=16, faithful=False)
synth.fill(o2, g, 0, F, N, seed=1)
h2 = rssync_amd.SyncProblem(seed=SEED)
synth.fill(h2, g, 0, F, N, seed=1)
t = time.time(); co_, do_, tro = o2.sync_trace(0.036, 0, F - 1, 0.0, 0.2); t_o = time.time() - t
t = time.time(); ch_, dh_ = h2.Sync(0.036, 0, F - 1, 0.0, 0.2); t_h = time.time() - t
trh = h2.sync_trace()
print(f"Sync oracle: cost {co_:.4f} delay {do_:.7f} iters {len(tro)} ({t_o:.2f}s)")
print(f"Sync hip   : cost {ch_:.4f} delay {dh_:.7f} iters {len(trh)} ({t_h:.3f}s)")
n = min(len(tro), len(trh))
np.set_printoptions(linewidth=200, precision=6)
print("oracle trace\n", tro[:n][:6]); print("hip trace\n", trh[:n][:6])
