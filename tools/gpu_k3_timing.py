#!/usr/bin/env python3
"""Where a motion optimisation's time goes (needs the timing variant of the library):

    bash tools/k2_build_variant.sh k3timing -DRSSYNC_K2_COUNTERS=1 -DRSSYNC_K3_TIMING=1
    RSSYNC_LIB=$PWD/rs-sync_amd/_variants/lib_k3timing.so python tools/gpu_k3_timing.py [frames] [tracks]

Core-clock ticks of wave 0 of every workgroup: inside evaluations / in the whole L-BFGS, per frame and per evaluation."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N = int(sys.argv[2]) if len(sys.argv) > 2 else 130
gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=0x5EED0001)
p = rssync_amd.SyncProblem(seed=0x5EED, max_outer_iters=20)
synth.fill(p, gyro, 0, F, N, seed=0x5EED0003)
lib = p._lib
lib.rship_debug_k2_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
buf = (C.c_uint64 * 16)()
d = synth.D_TRUE + 4e-4
p.init_motion(d, 0, F - 1)
for step in (0.0, -2e-4, -1e-4, -5e-5, -2e-5):
    d += step
    lib.rship_debug_k2_counters(p.device_context(), buf, 1)
    M, k, it, ev = p.opt_motion(d)
    lib.rship_debug_k2_counters(p.device_context(), buf, 1)
    r = [int(x) for x in buf]
    print(json.dumps({"frames": F, "tracks": N, "iters_per_frame": it / F, "evals_per_frame": ev / F,
                      "ticks_per_eval": r[10] / max(r[11], 1), "ticks_lbfgs_per_frame": r[12] / max(r[13], 1),
                      "share_of_lbfgs_inside_evals": r[10] / max(r[12], 1),
                      "ticks_outside_evals_per_iteration": (r[12] - r[10]) / max(it, 1)}), flush=True)
