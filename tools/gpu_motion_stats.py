import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth
F, N = int(os.environ.get("F", 1024)), int(os.environ.get("N", 2048))
g = synth.make_gyro(0, (F + 2) / 30, seed=0x5EED0003)
h = rssync_amd.SyncProblem(seed=0x5EED0003)
synth.fill(h, g, 0, F, N, seed=0x5EED0003)
lib = rssync_amd.load_library()
lib.rship_opt_motion_detail.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
d0 = 0.0365
M, k = h.init_motion(d0, 0, F - 1)
ctx = C.c_void_p(h.device_context())
D = d0 * g.fs
kd = np.array([int(np.floor(D))], np.int32); fd = np.array([D - np.floor(D)], np.float32)  # one window
for rep in range(3):
    st = np.zeros((F, 2), dtype=np.uint32)
    t = time.perf_counter()
    assert lib.rship_opt_motion_detail(ctx, kd.ctypes.data, fd.ctypes.data, st.ctypes.data, F) == 0
    dt = time.perf_counter() - t
    it, ev = st[:, 0], st[:, 1]
    print(f"call {rep}: {dt*1e3:.2f} ms  iters mean {it.mean():.1f} max {it.max()} | evals mean {ev.mean():.1f} p50 {np.median(ev):.0f} p99 {np.quantile(ev,0.99):.0f} max {ev.max()}  | frames at 200 iters: {(it>=200).sum()}")
