"""Timing of the LMedS tile kernel with a varying number of hypotheses (stage split)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth
F, N = int(os.environ.get("F", 1024)), int(os.environ.get("N", 2048))
g = synth.make_gyro(0, (F + 2) / 30, seed=3)
h = rssync_amd.SyncProblem(seed=3)
synth.fill(h, g, 0, F, N, seed=3)
h.upload()
h.presync_curve(0.0, 0, F, 0.01, 0.05)  # selects all frames
lib = rssync_amd.load_library()
ctx = C.c_void_p(h.device_context())
lib.rship_presync_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
fs = g.fs
delays = -0.2 + 0.0005 * np.arange(800)
D = delays * fs
kd = np.floor(D).astype(np.int32); fd = (D - np.floor(D)).astype(np.float32)
costs = np.zeros(800); flags = C.c_uint32()
for nh in [0, 4, 8, 20, 40, 20]:
    ts = []
    for rep in range(3):
        t = time.perf_counter()
        rc = lib.rship_presync_costs(ctx, kd.ctypes.data, fd.ctypes.data, 800, nh, 0, 3, costs.ctypes.data, C.byref(flags), None, None)
        ts.append(time.perf_counter() - t)
        assert rc == 0
    print(f"n_hyp {nh:3d}: {min(ts)*1e3:8.2f} ms  ({min(ts)/(F*800)*1e9:7.1f} ns per (frame,cand))", flush=True)
