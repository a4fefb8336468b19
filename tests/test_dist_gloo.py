"""The N>1 path on CPU: two gloo ranks, frames sharded, the only exchange a sum of a few
doubles.  Sharded and unsharded runs must agree (the sums differ only by fp64 association)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_equal_one(hosttest_lib, tmp_path):
    import rssync_amd
    from rssync_amd import synth
    from rssync_amd.dist import shard
    assert shard(0, 16, 0, 2) == (0, 8) and shard(0, 16, 1, 2) == (8, 16) and shard(0, 5, 3, 4) == (5, 5)
    port = _free_port()
    outs = [str(tmp_path / f"r{r}.json") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), "2", str(port),
                               outs[r]]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [json.load(open(o)) for o in outs]
    # single process, all frames, no hook
    F, N = 16, 96
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=6)
    one = rssync_amd.SyncProblem(seed=123, max_outer_iters=12, _lib=hosttest_lib)
    synth.fill(one, gyro, 0, F, N, seed=6, noise=0.0, outliers=0.0)
    c0, d0 = one.PreSync(0.0, 0, F, 0.004, 0.1)
    c1, d1 = one.Sync(d0, 0, F - 1, 0.0, 0.2)
    n_iters = len(one.sync_trace())
    assert res[0]["frames"] == [0, 8] and res[1]["frames"] == [8, 16]
    for r in res:
        assert r["presync"][1] == d0                       # same arg-min on every rank
        assert r["presync"][0] == pytest.approx(c0, rel=1e-12)
        assert r["sync"][1] == pytest.approx(d1, abs=1e-9)
        assert r["sync"][0] == pytest.approx(c1, rel=1e-9)
        assert r["iters"] == n_iters
        assert r["presync_exchanges"] == 1                 # one all-reduce for the whole sweep
        assert r["sync_exchanges"] == 2 * r["iters"] + 1   # <= 2 per outer iteration + final loss
    assert res[0]["sync"] == res[1]["sync"]                # replicated optimiser state
    assert abs(d1 - synth.D_TRUE) < 1e-4
    # sync points on sharded frames == unsharded (windows cross the shard boundary at frame 8)
    costs, delays = one.sync_points([0, 3, 6, 9], 6, 0.02, 0.004, 0.04, repeats=2)
    iters = [len(one.window_trace(w)) for w in range(4)]
    for r in res:
        assert r["points_iters"] == iters
        assert r["points"][1] == pytest.approx(list(delays), abs=1e-9)
        assert r["points"][0] == pytest.approx(list(costs), rel=1e-9)
    assert res[0]["points"] == res[1]["points"]
    # one exchange of 2100 candidates x 4 windows + 4 flags = 8404 doubles: more than the hook's initial
    # 8192-double staging buffer, which must grow (a refused or skipped all-reduce would leave the ranks
    # with different sums)
    wc, wd = one.pre_sync_windows(0.02, [0, 3, 6, 9], [6, 9, 12, 15], 0.0001, 0.105)
    for r in res:
        assert r["big"]["calls"] == 1
        assert r["big"]["delays"] == list(wd)
        assert r["big"]["costs"] == pytest.approx(list(wc), rel=1e-12)
    assert res[0]["big"]["costs"] == res[1]["big"]["costs"]
    # ranks whose largest frames differ (96 / 600 tracks): NO exchange to agree on anything, and every frame's GuessK is
    # the single-process run's BIT FOR BIT -- the kernels' shapes (in the stand-in the order of K3's sums: one wave per
    # frame up to 512 tracks, four above) follow each frame's OWN track count, as the reference evaluates each frame
    # in its own lambda (core_private.cpp:73-86, :231-238, :263-295)
    n_of = lambda fr: 96 if fr < 8 else 600
    mix = rssync_amd.SyncProblem(seed=321, max_outer_iters=6, _lib=hosttest_lib)
    mix.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr in range(F):
        mix.SetTrackResult(*next(iter(synth.make_frames(gyro, fr, fr + 1, n_of(fr), seed=6))))
    M1, k1 = mix.init_motion(0.03, 0, F - 1)
    mc, md = mix.Sync(0.03, 0, F - 1, 0.0, 0.2)
    its = len(mix.sync_trace())
    for r in res:
        m = r["mixed"]
        b, e = r["frames"]
        assert m["init_calls"] == 0                        # GuessMotion / GuessK are rank-local: nothing to agree on
        assert np.array_equal(np.asarray(m["k"]), k1[b:e]) and np.array_equal(np.asarray(m["M"]), M1[b:e])
        assert m["iters"] == its
        assert m["sync"][1] == pytest.approx(md, abs=1e-9) and m["sync"][0] == pytest.approx(mc, rel=1e-9)
        # two exchanges per iteration (three when a line search needs its later trials) + the final loss
        assert 2 * its + 1 <= m["calls"] <= 3 * its + 1
    assert res[0]["mixed"]["sync"] == res[1]["mixed"]["sync"]
    # the 96-track frames ALONE in a problem: the same bits as among the 600-track frames (rounds 2-4: other bits -- the
    # shape followed the problem's largest frame, and this line asserted the inequality)
    lone = rssync_amd.SyncProblem(seed=321, max_outer_iters=6, _lib=hosttest_lib)
    lone.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr in range(8):
        lone.SetTrackResult(*next(iter(synth.make_frames(gyro, fr, fr + 1, 96, seed=6))))
    M_lone, k_lone = lone.init_motion(0.03, 0, 7)
    assert np.array_equal(k_lone, k1[:8]) and np.array_equal(M_lone, M1[:8])
    # BASELINE config 5 with ranks: the orientation sweep (gyro as rates, replicated; frames sharded) == one process,
    # ONE exchange for the whole sweep (round 6: the orientations are pipelined, the [orientations][candidates] cost matrix
    # crosses the ranks once; rounds 1-5: one blocking exchange per orientation), the true orientation first on every rank
    F5, N5 = 12, 64
    g5 = synth.make_gyro(1.0, 1.0 + (F5 + 2) / synth.FPS, seed=77)
    names = list(synth.ORIENTATIONS[:4]) + ["XYZ"]
    o5 = rssync_amd.SyncProblem(seed=55, _lib=hosttest_lib)
    for fr in synth.make_frames(g5, 30, 30 + F5, N5, seed=77):
        o5.SetTrackResult(*fr)
    oc, od = o5.orientation_sweep(g5.times, g5.rates, names, 0.0, 30, 30 + F5, 0.004, 0.1)
    assert res[0]["sweep"]["frames"] == [30, 36] and res[1]["sweep"]["frames"] == [36, 42]
    for r in res:
        sw = r["sweep"]
        assert sw["calls"] == 1
        assert sw["delays"] == list(od)
        assert sw["costs"] == pytest.approx(list(oc), rel=1e-12)
        assert names[int(np.argmin(sw["costs"]))] == "XYZ"
    assert res[0]["sweep"]["costs"] == res[1]["sweep"]["costs"]
