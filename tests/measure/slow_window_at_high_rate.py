"""Is the one slow window of the high-rate small-frame scene (tools/gpu_gyro_rate.py: 98 sync points of 61 x 130 at 6 kHz,
one of them 458 outer iterations over its four Sync calls against ~50 on average) the REFERENCE's behaviour, or this
build's?  CPU only: the device-association stand-in (bit-identical to the GPU, tests/test_gpu_bitexact.py) runs the batched
call; the oracle (the reference's loop, restated) runs the driver loop call by call on the same scene with the same
sampler streams, and the outer iterations of every position's calls are compared.

    python tests/measure/slow_window_at_high_rate.py > profiles/r4_slow_window_6khz.json
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rssync_amd  # noqa: E402
from rssync_amd import synth  # noqa: E402
from rssync_amd.problem import bind  # noqa: E402
from oracle.oracle import OracleProblem  # noqa: E402

FS = float(os.environ.get("FS", 6000))
Fs, Ns, W, D = 3000, 130, 60, 30
NPOS = int(os.environ.get("NPOS", 98))

out = os.path.join(ROOT, "tests", "_build", "librssync_hosttest.so")
if not os.path.exists(out):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", out,
                           os.path.join(ROOT, "rs-sync_amd", "csrc", "sync_problem.cpp"), os.path.join(ROOT, "tests", "cpu_device", "rship_cpu.cpp")])
lib = bind(ctypes.CDLL(out))

g = synth.make_gyro(0, (Fs + 2) / synth.FPS, fs=FS, seed=6)
pos = list(range(0, Fs - W - 1, D))[:NPOS]
last = pos[-1] + W + 2
dev = rssync_amd.SyncProblem(seed=6, verbose=False, _lib=lib)
ora = OracleProblem(seed=6, threads=min(os.cpu_count() or 1, 8), faithful=False)
for p in (dev, ora):
    p.SetGyroQuaternions(g.quats, g.fs, g.t0)
for fr in synth.make_frames(g, 0, last, Ns, seed=6):
    dev.SetTrackResult(*fr)
    ora.SetTrackResult(*fr)

cd, dd = dev.sync_points(pos, W, 0.0, 0.001, 0.1)
it_dev = [len(dev.window_trace(w)) for w in range(len(pos))]           # each position's four Sync calls together

it_ora, d_ora = [], []
for p0 in pos:   # the reference driver's loop (core_testcode.cpp:303-316)
    d = ora.PreSync(0.0, p0, p0 + W, 0.001, 0.1)[1]
    its = []
    for rep in range(4):
        tr = ora.sync_trace(d, p0, p0 + W, 0.0, 0.1, cap=2048)
        its.append(len(tr["trace"]) if isinstance(tr, dict) else len(tr[-1]))
        d = tr["delay"] if isinstance(tr, dict) else tr[1]
    it_ora.append(its)
    d_ora.append(d)
    print("position %d: oracle iterations %s, stand-in total %d" % (p0, its, it_dev[len(it_ora) - 1]), file=sys.stderr, flush=True)

it_ora = np.array(it_ora)
tot_ora = it_ora.sum(axis=1)
worst = int(np.argmax(it_dev))
print(json.dumps({
    "what": __doc__.split("\n")[0], "gyro_hz": FS, "positions": len(pos), "window": W, "tracks": Ns,
    "stand_in_iterations_of_the_four_calls": {"mean": float(np.mean(it_dev)), "max": int(np.max(it_dev)), "position_of_max": pos[worst]},
    "oracle_iterations_of_the_four_calls": {"mean": float(tot_ora.mean()), "max": int(tot_ora.max()), "position_of_max": pos[int(np.argmax(tot_ora))]},
    "oracle_iterations_per_call_at_the_stand_ins_slowest_position": it_ora[worst].tolist(),
    "positions_with_more_than_300_iterations": {"stand_in": [pos[i] for i in range(len(pos)) if it_dev[i] > 300],
                                                 "oracle": [pos[i] for i in range(len(pos)) if tot_ora[i] > 300]},
    "positions_where_the_totals_differ_by_more_than_20": [[pos[i], int(it_dev[i]), int(tot_ora[i])] for i in range(len(pos)) if abs(it_dev[i] - tot_ora[i]) > 20],
    "delay_difference_s": {"median": float(np.median(np.abs(dd - np.array(d_ora)))), "max": float(np.abs(dd - np.array(d_ora)).max())},
}, indent=1))
