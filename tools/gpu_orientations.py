"""BASELINE config 5: C3-sized problem (4096 frames x 2048 tracks), gyro given as rates at jittered
timestamps, 48 IMU orientations, PreSync each (core_testcode.cpp:186-232).  Times the batched
sweep (host preparation overlapped with the GPU) against the per-orientation calls.  GPU box."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth

F = int(os.environ.get("F", 4096)); N = int(os.environ.get("N", 2048)); NOR = int(os.environ.get("NOR", 48))
STEP = float(os.environ.get("STEP", 0.0005)); SEED = 0x5EED0005
t_first = 1.0
g = synth.make_gyro(t_first, t_first + (F + 2) / synth.FPS, seed=SEED)     # t0 = 0
f0 = int(round(t_first * synth.FPS))
h = rssync_amd.SyncProblem(seed=SEED)
t = time.perf_counter()
for fr in synth.make_frames(g, f0, f0 + F, N, seed=SEED):
    h.SetTrackResult(*fr)
gen = time.perf_counter() - t
names = list(synth.ORIENTATIONS[:NOR])
if "XYZ" not in names:
    names[-1] = "XYZ"
h.set_gyro_rates(g.times, g.rates, "XYZ"); h.upload()
h.PreSync(0.0, f0, f0 + F, STEP, 0.2)                                       # warm-up
t = time.perf_counter()
seq = []
for nm in names[:8]:
    h.set_gyro_rates(g.times, g.rates, nm)
    seq.append(h.PreSync(0.0, f0, f0 + F, STEP, 0.2))
t_seq = (time.perf_counter() - t) / 8
t = time.perf_counter(); h.set_gyro_rates(g.times, g.rates, names[0]); t_host = time.perf_counter() - t
h.profile(True); h.profile_reset()
t = time.perf_counter()
costs, delays = h.orientation_sweep(g.times, g.rates, names, 0.0, f0, f0 + F, STEP, 0.2)
t_bat = time.perf_counter() - t
k = h.profile_get()["lmeds"]
order = np.argsort(costs)
print(json.dumps({"frames": F, "tracks": N, "orientations": len(names), "gyro_samples": int(g.rates.shape[0]),
                  "candidates": int(round(0.4 / STEP)), "gen_s": round(gen, 1),
                  "per_orientation_ms": {"sequential_calls": round(1e3 * t_seq, 2), "batched_sweep": round(1e3 * t_bat / len(names), 2),
                                         "lmeds_kernel": round(k[1] / max(1, k[0]), 2), "host_gyro_prep": round(1e3 * t_host, 2)},
                  "sweep_s": round(t_bat, 3),
                  "ray_residuals_per_s": F * N * round(0.4 / STEP) * len(names) / t_bat,
                  "identical_to_sequential": bool(all((costs[i], delays[i]) == seq[i] for i in range(8))),
                  "best": names[order[0]], "best_delay": float(delays[order[0]]),
                  "cost_ratio_best_to_second": float(costs[order[0]] / costs[order[1]])}))
