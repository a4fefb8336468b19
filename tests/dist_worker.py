"""One rank of the world_size-2 gloo test: this rank's half of the frames in its own SyncProblem
(host solver + CPU test double), reduce hook = torch.distributed all_reduce."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import rssync_amd
    from rssync_amd import synth
    from rssync_amd.dist import make_reduce_hook, shard
    from rssync_amd.problem import bind
    lib = bind(ctypes.CDLL(os.path.join(ROOT, "tests", "_build", "librssync_hosttest.so")))
    F, N = 16, 96
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=6)
    b, e = shard(0, F, rank, world)
    p = rssync_amd.SyncProblem(seed=123, max_outer_iters=12, _lib=lib)
    synth.fill(p, gyro, b, e, N, seed=6, noise=0.0, outliers=0.0)
    hook = make_reduce_hook("cpu")
    p.set_reduce_hook(hook)
    c0, d0 = p.PreSync(0.0, 0, F, 0.004, 0.1)
    n_pre = hook.stats["calls"]
    c1, d1 = p.Sync(d0, 0, F - 1, 0.0, 0.2)
    n_sync = hook.stats["calls"] - n_pre
    iters = len(p.sync_trace())
    # the batched driver loop on sharded frames: windows straddle the two ranks' blocks
    pos = [0, 3, 6, 9]
    costs, delays = p.sync_points(pos, 6, 0.02, 0.004, 0.04, repeats=2)
    # an exchange larger than the hook's initial staging buffer (8192 doubles): 2100 candidates x 4 windows
    n_before = hook.stats["calls"]
    wc, wd = p.pre_sync_windows(0.02, [0, 3, 6, 9], [6, 9, 12, 15], 0.0001, 0.105)
    big = dict(costs=list(map(float, wc)), delays=list(map(float, wd)), calls=hook.stats["calls"] - n_before,
               doubles=hook.stats["doubles"])
    # ranks with DIFFERENT largest frames (rank 0: 96 tracks, rank 1: 600): a frame's kernels -- and with them the order
    # of its sums -- follow the frame's OWN track count, so a frame's sums are the same bits whichever rank holds it and
    # whatever the other ranks hold, without any exchange to agree on anything
    n_of = lambda fr: 96 if fr < 8 else 600
    q = rssync_amd.SyncProblem(seed=321, max_outer_iters=6, _lib=lib)
    q.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr in range(b, e):
        q.SetTrackResult(*next(iter(synth.make_frames(gyro, fr, fr + 1, n_of(fr), seed=6))))
    hook2 = make_reduce_hook("cpu")
    q.set_reduce_hook(hook2)
    M, k = q.init_motion(0.03, 0, F - 1)  # GuessMotion + GuessK of this rank's frames (rank-local: no exchange)
    init_calls = hook2.stats["calls"]
    mc, md = q.Sync(0.03, 0, F - 1, 0.0, 0.2)
    mixed = dict(sync=[mc, md], iters=len(q.sync_trace()), calls=hook2.stats["calls"] - init_calls, init_calls=init_calls,
                 M=M.tolist(), k=k.tolist())
    # BASELINE config 5 with ranks (core_testcode.cpp:184-233): the gyro arrives as rates at timestamps, replicated on
    # every rank; the whole sweep is ONE pipeline (round 6): every orientation's PreSync over the sharded frames enqueued back to
    # back, ONE exchange of the [orientations][candidates] cost matrix at the end
    F5, N5 = 12, 64
    g5 = synth.make_gyro(1.0, 1.0 + (F5 + 2) / synth.FPS, seed=77)   # t0 = 0: timestamps must be >= 0
    b5, e5 = shard(30, 30 + F5, rank, world)
    names = list(synth.ORIENTATIONS[:4]) + ["XYZ"]
    r = rssync_amd.SyncProblem(seed=55, _lib=lib)
    for fr in synth.make_frames(g5, b5, e5, N5, seed=77):
        r.SetTrackResult(*fr)
    hook3 = make_reduce_hook("cpu")
    r.set_reduce_hook(hook3)
    oc, od = r.orientation_sweep(g5.times, g5.rates, names, 0.0, 30, 30 + F5, 0.004, 0.1)
    sweep = dict(costs=list(map(float, oc)), delays=list(map(float, od)), calls=hook3.stats["calls"], frames=[b5, e5])
    res = dict(rank=rank, big=big, mixed=mixed, sweep=sweep, frames=[b, e], presync=[c0, d0], sync=[c1, d1], iters=iters,
               presync_exchanges=n_pre, sync_exchanges=n_sync,
               points=[list(map(float, costs)), list(map(float, delays))],
               points_iters=[len(p.window_trace(w)) for w in range(len(pos))])
    with open(out, "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
