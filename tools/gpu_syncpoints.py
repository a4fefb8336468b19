"""The reference driver's real workload (core_testcode.cpp:270-316): many small windows
(60 frames x ~130 tracks) along a video, PreSync + 4x Sync each.  Sequential ISyncProblem calls
vs the batched rssync_ext_sync_points, same results required.  Run on the GPU box."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
from rssync_amd import synth

F = int(os.environ.get("F", 3000)); N = int(os.environ.get("N", 130))
WINDOW = int(os.environ.get("WINDOW", 60)); DIST = int(os.environ.get("DIST", 30))
SEED = 0x5EED0006
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=SEED)
pos = list(range(0, F - WINDOW - 1, DIST))


def problem():
    h = rssync_amd.SyncProblem(seed=SEED, verbose=False)
    synth.fill(h, g, 0, F, N, seed=SEED)
    h.upload()
    return h


def loop(h):
    out = []
    for p0 in pos:
        d = h.PreSync(0.0, p0, p0 + WINDOW, 0.001, 0.1)[1]
        for _ in range(4):
            c, d = h.Sync(d, p0, p0 + WINDOW, 0.0, 0.1)
        out.append(d)
    return np.array(out)


os.environ["RSSYNC_EXECUTOR"] = "0"    # seq / bat / hostloop: the chain of launches
seq, bat = problem(), problem()
loop(seq); bat.sync_points(pos, WINDOW, 0.0, 0.001, 0.1)    # warm-up (also advances both streams equally)
t = time.perf_counter(); ds = loop(seq); t_seq = time.perf_counter() - t
t = time.perf_counter(); _, db = bat.sync_points(pos, WINDOW, 0.0, 0.001, 0.1); t_bat = time.perf_counter() - t
# kernel breakdown from a second, profiled run (two HIP events per launch make the host the bottleneck of a
# loop of ~10 us launches: not the run that is timed); same sampler streams as a third sequential pass would use
loop(seq)
bat.profile(True); bat.profile_reset()
t = time.perf_counter(); bat.sync_points(pos, WINDOW, 0.0, 0.001, 0.1); t_prof = time.perf_counter() - t
prof = bat.profile_get()
del os.environ["RSSYNC_EXECUTOR"]
execp = problem()                      # the window executor (default for frames of up to 512 tracks)
os.environ["RSSYNC_EXECUTOR"] = "0"
execp.sync_points(pos, WINDOW, 0.0, 0.001, 0.1)
t = time.perf_counter(); _, de = execp.sync_points(pos, WINDOW, 0.0, 0.001, 0.1); t_exec = time.perf_counter() - t
# the production tripwire (round 5): one executor call in RSSYNC_EXECUTOR_CHECK_EVERY (default 256) is re-run through the
# launch chain inside the call and compared bit for bit.  What a VERIFIED call costs: every call verified (every = 1).
del os.environ["RSSYNC_EXECUTOR"]
checked = problem()
checked.set_executor_check_every(1)
checked.sync_points(pos, WINDOW, 0.0, 0.001, 0.1)
t = time.perf_counter(); _, dc = checked.sync_points(pos, WINDOW, 0.0, 0.001, 0.1); t_checked = time.perf_counter() - t
stats_checked = checked.executor_stats()
os.environ["RSSYNC_EXECUTOR"] = "0"
hostloop = problem(); hostloop.set_host_loop(True)
hostloop.sync_points(pos, WINDOW, 0.0, 0.001, 0.1)
t = time.perf_counter(); _, dh = hostloop.sync_points(pos, WINDOW, 0.0, 0.001, 0.1); t_host = time.perf_counter() - t
iters = [len(bat.window_trace(w)) for w in range(len(pos))]
print(json.dumps({"frames": F, "tracks": N, "window": WINDOW, "positions": len(pos),
                  "sequential_s": round(t_seq, 4), "batched_s": round(t_bat, 4), "speedup": round(t_seq / t_bat, 2),
                  "batched_host_loop_s": round(t_host, 4), "batched_executor_s": round(t_exec, 4),
                  "executor_identical": bool(np.array_equal(de, db)), "batched_profiled_s": round(t_prof, 4),
                  "executor_check": {"every_call_verified_s": round(t_checked, 4), "identical": bool(np.array_equal(dc, de)),
                                     "a_verified_call_costs_x": round(t_checked / t_exec, 2),
                                     "average_overhead_at_one_in_256": round((t_checked - t_exec) / t_exec / 256, 5),
                                     "stats": stats_checked},
                  "identical": bool(np.array_equal(ds, db)), "max_abs_diff": float(np.abs(ds - db).max()),
                  "host_loop_identical": bool(np.array_equal(dh, db)),
                  "delay_err_vs_truth_ms": {"median": float(np.median(np.abs(db - synth.D_TRUE)) * 1e3),
                                           "max": float(np.abs(db - synth.D_TRUE).max() * 1e3)},
                  "outer_iters_per_position": {"mean": float(np.mean(iters)), "max": int(np.max(iters))},
                  "kernels": prof}))
