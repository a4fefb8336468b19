"""Driven by tests/test_host_logic.py::test_host_solver_is_clean_under_asan_ubsan: every entry point of
the host solver once, on the CPU test double, in a process with AddressSanitizer preloaded."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rssync_amd
from rssync_amd import synth
from rssync_amd.problem import bind
lib = bind(ctypes.CDLL(sys.argv[1]))
F, N = 16, 96
g = synth.make_gyro(1.0, 1.0 + (F + 2) / synth.FPS, seed=5)
p = rssync_amd.SyncProblem(seed=1, max_outer_iters=6, _lib=lib)
synth.fill(p, g, 30, 30 + F, N, seed=5)
print(p.PreSync(0.0, 30, 30 + F, 0.004, 0.05))
print(p.Sync(0.036, 30, 30 + F - 1, 0.0, 0.2))
print(p.sync_points([30, 34, 38], 6, 0.02, 0.004, 0.04, repeats=2))
print(p.pre_sync_windows(0.03, [30, 33, 90], [38, 40, 95], 0.004, 0.04))
d, c = p.DebugPreSync(0.03, 30, 40, 0.05, 9); print(c[:3])
p.set_gyro_rates(g.times, g.rates, "yXz")
print(p.orientation_sweep(g.times, g.rates, ["XYZ", "zyx", "Yxz"], 0.0, 30, 30 + F, 0.004, 0.05))
pa = np.random.default_rng(0).uniform([100, 100], [2000, 1400], size=(40, 2))
p.set_track_pixels(200, 7.0, 7.0333, pa, pa + 1.5, synth.LENS, synth.IMAGE_ROWS)
p.SetGyroQuaternions(g.quats, g.fs, g.t0)
print(p.PreSync(0.0, 30, 300, 0.004, 0.05))
a4, b4 = p.frame_rays(200); print(a4.shape)
ts_us, q = synth.make_timestamped(g, seed=1)
p.SetGyroQuaternionsTimestamped(ts_us, q)
print(p.PreSync(0.0, 30, 46, 0.004, 0.05))
try:
    p.Sync(0.03, 500, 600, 0.0, 0.2)
except Exception as e:
    print("ok:", e)
del p
print("done")
