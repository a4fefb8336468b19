#!/usr/bin/env python3
"""Dynamic trip counts of the LMedS tile kernel at bench size (needs the counters variant of the library):

    RSSYNC_LIB=$PWD/rs-sync_amd/_variants/lib_counters.so python tools/gpu_k2_counters.py [frames] > gpurun_out/k2_counters.json

One PreSync(radius 200 ms, step 0.5 ms) over `frames` x 2048 tracks; the 16 counters of kernels/lmeds.hpp divided by
the number of (frame, candidate) pairs.
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rssync_amd  # noqa: E402
from rssync_amd import synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = 2048
gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=0x5EED0003)
p = rssync_amd.SyncProblem(seed=0x5EED0003, max_outer_iters=20)
synth.fill(p, gyro, 0, F, N, seed=0x5EED0003)
p.upload()
lib = p._lib
lib.rship_debug_k2_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
buf = (C.c_uint64 * 16)()
lib.rship_debug_k2_counters(p.device_context(), buf, 1)
c, d = p.PreSync(0.0, 0, F, 0.0005, 0.2)
lib.rship_debug_k2_counters(p.device_context(), buf, 1)
raw = [int(x) for x in buf]
pairs = raw[0] or 1
names = ["frame_candidates", "queue_pops", "hypotheses_swept", "exact_selections", "selection_counting_passes",
         "selection_min_endings", "candidates_redone", "sweeps_without_bound", "contenders_closed_exactly", "sweeps_completed"]
out = {"frames": F, "tracks": N, "presync": [c, d], "raw": dict(zip(names, raw)),
       "per_frame_candidate": {n: raw[i] / pairs for i, n in enumerate(names)}}
print(json.dumps(out, indent=1))
if raw[11]:  # a -DRSSYNC_K2_TIMING=1 build: ticks of every wave waiting at the candidate loop's barriers
    import sys as _s
    _s.stderr.write(json.dumps({"barrier_share_of_wave_time": raw[10] / raw[11],
                                "by_barrier": {"tile_written": raw[12] / raw[11], "directions_ready": raw[13] / raw[11],
                                               "sweeps_done": raw[14] / raw[11], "stage_D_sums": raw[15] / raw[11]}}) + "\n")
