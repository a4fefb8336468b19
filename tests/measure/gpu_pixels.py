"""pack_frames_kernel on pixel frames at BASELINE size (4096 frames x 2048 tracks): kernel time, HBM rate,
and the host route it replaces (CPU undistort -> SetTrackResult -> host packing).  GPU box."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rssync_amd
from rssync_amd import synth
from oracle import oracle

F = int(os.environ.get("F", 4096)); N = int(os.environ.get("N", 2048))
g = synth.make_gyro(0, (F + 2) / synth.FPS, seed=1)
rng = np.random.default_rng(1)
pa = rng.uniform([50, 50], [synth.IMAGE_COLS - 50, synth.IMAGE_ROWS - 50], size=(F, N, 2))
pb = pa + rng.normal(scale=3.0, size=pa.shape)

h = rssync_amd.SyncProblem(seed=1)
h.SetGyroQuaternions(g.quats, g.fs, g.t0)
t = time.perf_counter()
for f in range(F):
    h.set_track_pixels(f, f / synth.FPS, (f + 1) / synth.FPS, pa[f], pb[f], synth.LENS, synth.IMAGE_ROWS)
t_set = time.perf_counter() - t
h.profile(True)
res = {}
for rep in range(3):
    h.profile_reset()
    # a change of the gyro time base re-packs every frame on the next use
    h.SetGyroQuaternions(g.quats, g.fs, g.t0 + 1e-3 * (rep + 1))
    t = time.perf_counter(); h.upload(); t_up = time.perf_counter() - t
    n, ms = h.profile_get()["pixels"]
    res = {"launches": n, "kernel_ms": ms, "upload_call_s": round(t_up, 4)}
pairs = F * N
res["GB_per_s"] = round(pairs * 128 / (res["kernel_ms"] * 1e-3) / 1e9, 1)  # 32 B of pixels read, 32 + 64 B of packed streams written
res["frac_of_8TBs"] = round(res["GB_per_s"] / 8000, 3)
# the host route on a sample of frames: oracle undistort (C, fp64, 1 thread) + SetTrackResult
S = min(F, 128)
t = time.perf_counter()
tracks = [oracle.pixels_to_tracks(synth.LENS, f / synth.FPS, (f + 1) / synth.FPS, synth.IMAGE_ROWS, pa[f], pb[f]) for f in range(S)]
t_cpu = (time.perf_counter() - t) * F / S
a1, b1 = h.frame_rays(7)
h2 = rssync_amd.SyncProblem(seed=1)
h2.SetGyroQuaternions(g.quats, g.fs, g.t0 + 3e-3)
for f in range(S):
    h2.SetTrackResult(f, *tracks[f])
a2, b2 = h2.frame_rays(7)
print(json.dumps({"frames": F, "tracks": N, "set_track_pixels_s": round(t_set, 3), "device": res,
                  "cpu_undistort_1thread_s_extrapolated": round(t_cpu, 2),
                  "frame7_max_abs_diff_vs_host_route": float(max(np.abs(a1 - a2).max(), np.abs(b1 - b2).max())),
                  "frame7_identical_fraction": float(((a1 == a2).mean() + (b1 == b2).mean()) / 2)}))
