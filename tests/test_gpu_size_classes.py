"""A frame's bits and speed depend on its OWN track count only (round 5: size classes).

The reference evaluates every frame in its own lambda (core_private.cpp:73-86 PreSync, :231-238 / :245-250 the loss
sums, :263-295 the per-frame L-BFGS): nothing about frame i knows frame j's track count.  Rounds 2-4 picked the kernel
family -- and with it the order of a frame's sums -- from the LARGEST frame of the whole problem.  Now a selection is cut
into one slot list per size class ({<= 512 tracks: one wave per frame}, {<= 1024, 2048, 6144, 8192: four waves with 4 / 8
/ 16-24 / 32 rows per thread}, {more: rows in global memory}) and every class runs its own kernels
(rssync_kernels.hip: class_of, class_ranges).  These tests demand, bit for bit, that a frame gives the same PreSync
costs and winners, the same GuessMotion / GuessK, the same loss and derivative and the same Sync trace whether it is
evaluated ALONE in a problem or among frames of every other class -- and that the whole still matches the oracle.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 123
THREADS = min(os.cpu_count() or 1, 16)
# one frame of every class: one wave (96, 300), four waves with 4 / 8 / 16 / 24 rows per thread (600, 1500, 3000, 5000), the
# class whose PreSync tile runs as eight waves (7000), rows in global memory (9000)
COUNTS = [96, 300, 600, 1500, 3000, 5000, 7000, 9000]


def _frames(gyro, counts, seed, **kw):
    from rssync_amd import synth
    return [next(iter(synth.make_frames(gyro, fr, fr + 1, n, seed=seed, **kw))) for fr, n in enumerate(counts)]


def _problem(gyro, frames, **kw):
    import rssync_amd
    p = rssync_amd.SyncProblem(seed=SEED, verbose=False, **kw)
    p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for f in frames:
        p.SetTrackResult(*f)
    return p


def test_a_frames_bits_do_not_depend_on_its_neighbours():
    from rssync_amd import synth
    F = len(COUNTS)
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=31)
    frames = _frames(g, COUNTS, seed=31, noise=5e-4, outliers=0.08)
    mixed = _problem(g, frames, max_outer_iters=8)
    d, c, fc, bh = mixed.presync_curve(0.03, 0, F, 0.001, 0.012, per_frame=F)       # 24 candidates x 8 frames
    M, k = mixed.init_motion(0.0362, 0, F - 1)
    L, G = mixed.loss([0.0362, 0.03, 0.041], grad=True)
    per_frame_loss = []
    for fr in range(F):
        lone = _problem(g, [frames[fr]], max_outer_iters=8)
        d1, c1, fc1, bh1 = lone.presync_curve(0.03, 0, F, 0.001, 0.012, per_frame=1)
        np.testing.assert_array_equal(d1, d)
        np.testing.assert_array_equal(bh1[:, 0], bh[:, fr], err_msg="PreSync winners of the %d-track frame" % COUNTS[fr])
        np.testing.assert_array_equal(fc1[:, 0], fc[:, fr], err_msg="PreSync costs of the %d-track frame" % COUNTS[fr])
        M1, k1 = lone.init_motion(0.0362, 0, F - 1)
        np.testing.assert_array_equal(M1[0], M[fr], err_msg="GuessMotion of the %d-track frame" % COUNTS[fr])
        np.testing.assert_array_equal(k1[0], k[fr], err_msg="GuessK of the %d-track frame" % COUNTS[fr])
        per_frame_loss.append(lone.loss([0.0362, 0.03, 0.041], grad=True))
        # Sync of the frame's own one-frame window: the mixed problem's table holds six other classes, the lone one none
        ca, da = lone.Sync(0.036, fr, fr, 0.0, 0.2)
        tra = lone.sync_trace()
        other = _problem(g, frames, max_outer_iters=8)
        other.init_motion(0.0362, 0, F - 1)        # (the lone problem has made one Sync-side call before: same sampler stream)
        cb, db = other.Sync(0.036, fr, fr, 0.0, 0.2)
        assert (ca, da) == (cb, db), COUNTS[fr]
        np.testing.assert_array_equal(tra, other.sync_trace())
    # the window sums of the mixed problem are the sequential sums of the frames' own values (one chunk of the plan)
    for j in range(3):
        acc_l = acc_g = 0.0
        for fr in range(F):
            acc_l += per_frame_loss[fr][0][j]
            acc_g += per_frame_loss[fr][1][j]
        assert L[j] == acc_l and G[j] == acc_g


def test_every_class_in_one_problem_against_the_oracle():
    """the same mixed problem against the CPU restatement: rows, PreSync winners and costs, GuessK, loss, Sync"""
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    from oracle import oracle as ora
    F = len(COUNTS)
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=31)
    frames = _frames(g, COUNTS, seed=31, noise=3e-4, outliers=0.05)
    h = _problem(g, frames, max_outer_iters=10)
    o = OracleProblem(seed=SEED, threads=THREADS, faithful=False, max_outer_iters=10)
    o.SetGyroQuaternions(g.quats, g.fs, g.t0)
    for f in frames:
        o.SetTrackResult(*f)
    dh, ch, fch, bhh = h.presync_curve(0.03, 0, F, 0.001, 0.012, per_frame=F)
    do, co, fco, bho = o.presync_curve(0.03, 0, F, 0.001, 0.012, per_frame=F)
    np.testing.assert_array_equal(dh, do)
    same = bhh == bho
    assert same.mean() > 0.97, same.mean()
    rel = np.abs(fch - fco) / fco
    assert rel[same].max() < 1e-3 and np.median(rel[same]) < 2e-6
    assert np.argmin(ch) == np.argmin(co)
    Mh, kh = h.init_motion(0.0362, 0, F - 1)
    agree = 0
    for f in range(F):
        Mo, _, _ = o.guess_motion(f, 0.0362, 200, ora.STREAM_SYNC_INIT + 0)
        if np.abs(Mh[f] - Mo).max() < 1e-12:
            agree += 1
            ko = np.clip(100 / np.linalg.norm(o.problem_matrix(f, 0.0362) @ Mo), 10, 1000)
            assert kh[f] == pytest.approx(ko, rel=1e-12)
    assert agree >= F - 1                          # the fp32 search may flip one near-tie
    Lh, Gh = h.loss([0.0362, 0.03], grad=True)
    for j, dd in enumerate((0.0362, 0.03)):
        per = [o.loss(f, dd, Mh[f], kh[f]) for f in range(F)]
        assert Lh[j] == pytest.approx(sum(p[0] for p in per), rel=1e-12)
        assert Gh[j] == pytest.approx(sum(p[2] for p in per), rel=1e-10, abs=1e-10 * abs(Lh[j]))
    # Sync from the oracle's GuessMotion winners (the fp32 search may flip a near-tie; the optimisation is what is compared)
    co2, do2, tro = o.sync_trace(0.036, 0, F - 1, 0.0, 0.2)
    h2 = _problem(g, frames, max_outer_iters=10)
    h2.set_init_override(o.last_init_winners())
    ch2, dh2 = h2.Sync(0.036, 0, F - 1, 0.0, 0.2)
    assert abs(dh2 - do2) < 1e-6 and ch2 == pytest.approx(co2, rel=1e-8)
    assert len(h2.sync_trace()) == len(tro)


@pytest.mark.parametrize("executor", ["1", "0"])
def test_batched_windows_of_mixed_classes_equal_the_sequential_calls(executor, monkeypatch):
    """sync_windows over windows that hold frames of several classes == the same windows one Sync call after the other,
    bit for bit -- through the window executor and through the launch chain (RSSYNC_EXECUTOR=0: one launch per class and
    step); and the batched PreSync of overlapping windows == PreSync per window"""
    from rssync_amd import synth
    monkeypatch.setenv("RSSYNC_EXECUTOR", executor)
    monkeypatch.setenv("RSSYNC_EXEC_BIG_SHARE", "1")     # (two larger frames among twelve: by default the chain's, not the executor's)
    counts = [130, 130, 600, 130, 96, 130, 1100, 130, 130, 300, 130, 130]
    F = len(counts)
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=44)
    frames = _frames(g, counts, seed=44, noise=5e-4, outliers=0.05)
    wins = [(0, 5), (3, 8), (6, 11), (7, 11)]                    # 2nd and 3rd hold two classes each, the 4th one class... and 1100
    seq = _problem(g, frames, max_outer_iters=10)
    want = [seq.Sync(0.036, b, e, 0.0, 0.2) for (b, e) in wins]
    want_tr = []
    seq2 = _problem(g, frames, max_outer_iters=10)
    for (b, e) in wins:
        seq2.Sync(0.036, b, e, 0.0, 0.2)
        want_tr.append(seq2.sync_trace().copy())
    bat = _problem(g, frames, max_outer_iters=10)
    costs, delays = bat.sync_windows([0.036] * len(wins), [w[0] for w in wins], [w[1] for w in wins], 0.0, 0.2)
    for w in range(len(wins)):
        assert (costs[w], delays[w]) == want[w], w
        np.testing.assert_array_equal(bat.window_trace(w), want_tr[w])
    pc, pd = bat.pre_sync_windows(0.03, [w[0] for w in wins], [w[1] + 1 for w in wins], 0.002, 0.02)
    for w, (b, e) in enumerate(wins):
        c1, d1 = seq.PreSync(0.03, b, e + 1, 0.002, 0.02)
        assert pd[w] == d1 and pc[w] == pytest.approx(c1, rel=1e-14)


@pytest.mark.parametrize("counts_at", [{7: 600, 19: 1100}, {7: 600, 19: 1100, 25: 300, 31: 9000}])
def test_the_executor_takes_windows_with_frames_of_any_class(counts_at, monkeypatch):
    """The window executor (one launch, tasks run by single waves) used to need every frame of the selection to be a
    one-wave frame: one 513-track frame sent every window of a clip to the chain of launches (21 -> 34 ms on the driver
    workload).  Now a larger frame's tasks are run by one wave in the FOUR-wave kernels' association
    (kernels/exec_big.hpp, MotionEval64<0, 1, EMU4>), so the executor's results stay the launch chain's bit for bit:
    the check mode re-runs every call through the chain and panics on any difference; and a problem that never uses
    the executor returns the same values.  Second case: 257 .. 512-track one-wave frames (eight rows per lane) together
    with larger ones -- the instantiation that runs one wave per SIMD -- and a frame of more than 8192 tracks (general
    spline path in the search, as lmeds_big_kernel)."""
    from rssync_amd import synth
    F = 40
    counts = [130 - 3 * (fr % 5) for fr in range(F)]
    for fr, n in counts_at.items():
        counts[fr] = n
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=52)
    frames = _frames(g, counts, seed=52, noise=5e-4, outliers=0.05)
    pos = [0, 6, 12, 18, 24]
    monkeypatch.delenv("RSSYNC_EXECUTOR", raising=False)
    monkeypatch.setenv("RSSYNC_EXEC_BIG_MAX", "16384")   # (by default the executor leaves selections with frames of more than 2048 tracks to the chain ...
    monkeypatch.setenv("RSSYNC_EXEC_BIG_SHARE", "1")     #  ... and those in which more than one slot in eight holds a larger frame)
    ex = _problem(g, frames, max_outer_iters=12)
    ex.set_executor_check(True)
    c_ex, d_ex = ex.sync_points(pos, 12, 0.0, 0.002, 0.04, repeats=3)
    st = ex.executor_stats()
    assert st["runs"] == 1 and st["checked"] == 1, st
    tr_ex = [ex.window_trace(w) for w in range(len(pos))]
    monkeypatch.setenv("RSSYNC_EXECUTOR", "0")
    ch = _problem(g, frames, max_outer_iters=12)
    c_ch, d_ch = ch.sync_points(pos, 12, 0.0, 0.002, 0.04, repeats=3)
    assert ch.executor_stats()["runs"] == 0
    np.testing.assert_array_equal(d_ex, d_ch)
    np.testing.assert_array_equal(c_ex, c_ch)
    for w in range(len(pos)):
        np.testing.assert_array_equal(tr_ex[w], ch.window_trace(w))
    # a single Sync call of a mixed window through the executor as well
    monkeypatch.delenv("RSSYNC_EXECUTOR", raising=False)
    one = _problem(g, frames, max_outer_iters=12)
    one.set_executor_check(True)
    c1, d1 = one.Sync(0.036, 4, 22, 0.0, 0.2)
    st = one.executor_stats()
    assert (st["runs"], st["checked"]) == (1, 1), st
    assert abs(d1 - synth.D_TRUE) < 2e-3
    if max(counts) > 2048:       # the default policy: such a selection goes through the chain of launches
        monkeypatch.delenv("RSSYNC_EXEC_BIG_MAX", raising=False)
        monkeypatch.delenv("RSSYNC_EXEC_BIG_SHARE", raising=False)
        dflt = _problem(g, frames, max_outer_iters=12)
        c2, d2 = dflt.sync_points(pos, 12, 0.0, 0.002, 0.04, repeats=3)
        assert dflt.executor_stats()["runs"] == 0
        np.testing.assert_array_equal(d2, d_ex)


def test_every_route_through_the_library_on_a_problem_of_mixed_classes():
    """Size classes cut EVERY launch of the library, not only PreSync / Sync of one object on one device: the host loop,
    two device contexts sharing the frames (contiguous blocks cut inside the table), DebugPreSync's candidate generator and
    the thesis' simplified mode all see the same per-class slot lists.  One problem with a frame of every class: each route
    gives the plain route's bits, and the simplified mode agrees with the oracle."""
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    counts = [96, 600, 300, 1500, 130, 3000, 5000, 7000, 9000, 200, 2048]       # classes interleaved along the table
    F = len(counts)
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=37)
    frames = _frames(g, counts, seed=37, noise=4e-4, outliers=0.06)

    def run(p):
        out = {}
        out["presync"] = p.PreSync(0.03, 0, F, 0.001, 0.012)
        out["debug"] = p.DebugPreSync(0.03, 0, F, 0.012, 9)
        out["sync"] = p.Sync(out["presync"][1], 0, F - 1, 0.0, 0.2)
        out["trace"] = np.array(p.sync_trace())
        out["simple"] = p.SyncSimplified(out["presync"][1], 0, F - 1, 0.0, 0.2)
        out["strace"] = np.array(p.sync_trace())
        out["windows"] = p.sync_windows([0.036, 0.0362, 0.0358], [0, 2, 5], [4, 7, F - 1], 0.0, 0.2)
        return out

    base = run(_problem(g, frames, max_outer_iters=8))
    host = _problem(g, frames, max_outer_iters=8)
    host.set_host_loop(True)
    two = _problem(g, frames, max_outer_iters=8)
    two.set_devices([0, 0])
    three = _problem(g, frames, max_outer_iters=8)
    three.set_devices([0, 0, 0])
    for name, p in (("host loop", host), ("two contexts", two), ("three contexts", three)):
        got = run(p)
        for key in base:
            a, b = base[key], got[key]
            if isinstance(a, tuple) and len(a) == 2 and np.ndim(a[0]) == 0:
                assert a == b, (name, key, a, b)
            else:
                for x, y in zip(a if isinstance(a, tuple) else (a,), b if isinstance(b, tuple) else (b,)):
                    np.testing.assert_array_equal(np.asarray(x), np.asarray(y), err_msg="%s: %s" % (name, key))
    # DebugPreSync's end points are PreSync's candidate values where the grids coincide (core_private.cpp:345: both ends)
    dd, dc = base["debug"]
    assert len(dd) == 9 and dd[0] == pytest.approx(0.03 - 0.012) and dd[-1] == pytest.approx(0.03 + 0.012)
    # the simplified mode against the oracle's restatement on the same mixed problem
    o = OracleProblem(seed=SEED, threads=THREADS, faithful=False, max_outer_iters=8)
    o.SetGyroQuaternions(g.quats, g.fs, g.t0)
    for f in frames:
        o.SetTrackResult(*f)
    cs, ds, tro = o.sync_simplified_trace(base["presync"][1], 0, F - 1, 0.0, 0.2)
    assert abs(base["simple"][1] - ds) < 1e-9 and base["simple"][0] == pytest.approx(cs, rel=1e-9)
    assert len(base["strace"]) == len(tro)


def test_resetting_a_frame_moves_it_to_its_new_class():
    """core_private.cpp:192-203: SetTrackResult on a frame that exists overwrites it (:194) -- with another track count the
    frame changes its size class between two calls of the same object.  After every overwrite the object must give what a
    fresh object holding the final frames gives, bit for bit (slot lists, window plans and staging follow the table, not
    the history), and the overwritten frame's old rows must be gone."""
    from rssync_amd import synth
    from oracle.oracle import OracleProblem
    counts = [130, 200, 130, 600, 130, 96]
    F = len(counts)
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=41)
    frames = _frames(g, counts, seed=41, noise=4e-4, outliers=0.06)
    p = _problem(g, frames, max_outer_iters=8)
    first = (p.PreSync(0.03, 0, F, 0.001, 0.012), p.Sync(0.036, 0, F - 1, 0.0, 0.2))
    # frame 2: 130 -> 700 tracks (one wave -> four waves); frame 3: 600 -> 40 (four waves -> one); frame 5: 96 -> 3000
    steps = [(2, 700), (3, 40), (5, 3000), (2, 130)]
    for n_done, (fr, n) in enumerate(steps, 1):
        new = next(iter(synth.make_frames(g, fr, fr + 1, n, seed=100 + n_done, noise=4e-4, outliers=0.06)))
        frames[fr] = new
        p.SetTrackResult(*new)
        got = (p.PreSync(0.03, 0, F, 0.001, 0.012), p.Sync(0.036, 0, F - 1, 0.0, 0.2), np.array(p.sync_trace()))
        fresh = _problem(g, frames, max_outer_iters=8)
        for _ in range(n_done):                          # (the sampler's stream of a Sync call advances with the object's calls)
            fresh.Sync(0.036, 0, F - 1, 0.0, 0.2)
        want = (fresh.PreSync(0.03, 0, F, 0.001, 0.012), fresh.Sync(0.036, 0, F - 1, 0.0, 0.2), np.array(fresh.sync_trace()))
        assert got[0] == want[0], (fr, n, got[0], want[0])
        assert got[1] == want[1], (fr, n, got[1], want[1])
        np.testing.assert_array_equal(got[2], want[2])
    assert first[0] != got[0]
    # and the final state against the oracle (PreSync arg-min and Sync from the same GuessMotion winners)
    o = OracleProblem(seed=SEED, threads=THREADS, faithful=False, max_outer_iters=8)
    o.SetGyroQuaternions(g.quats, g.fs, g.t0)
    for f in frames:
        o.SetTrackResult(*f)
    assert p.PreSync(0.03, 0, F, 0.001, 0.012)[1] == o.PreSync(0.03, 0, F, 0.001, 0.012)[1]
    co, do, tro = o.sync_trace(0.036, 0, F - 1, 0.0, 0.2)
    q = _problem(g, frames, max_outer_iters=8)
    q.set_init_override(o.last_init_winners())
    cq, dq = q.Sync(0.036, 0, F - 1, 0.0, 0.2)
    assert abs(dq - do) < 1e-6 and len(q.sync_trace()) == len(tro)


def test_orientation_sweep_over_frames_of_mixed_classes():
    """core_testcode.cpp:186-224 on a clip whose frames fall into four size classes: the batched sweep (gyro as rates,
    re-integrated and re-splined on the device per orientation, tracks kept) equals per-orientation set_gyro_rates + PreSync
    calls bit for bit, and the true orientation comes out first."""
    from rssync_amd import synth
    counts = [130, 600, 96, 1500, 300, 2500, 130, 700, 200, 130, 1100, 64]
    F = len(counts)
    g = synth.make_gyro(1.0, 1.0 + (F + 2) / synth.FPS, seed=43)
    frames = [next(iter(synth.make_frames(g, 30 + i, 31 + i, n, seed=43))) for i, n in enumerate(counts)]
    names = ["XYZ", "XZY", "yXZ", "Zxy", "xyz"]
    p = _problem(g, frames)
    costs, delays = p.orientation_sweep(g.times, g.rates, names, 0.0, 30, 30 + F, 0.004, 0.08)
    q = _problem(g, frames)
    for k, name in enumerate(names):
        q.set_gyro_rates(g.times, g.rates, name)
        c, d = q.PreSync(0.0, 30, 30 + F, 0.004, 0.08)
        assert (c, d) == (costs[k], delays[k]), (name, c, d, costs[k], delays[k])
    order = np.argsort(costs)
    assert names[order[0]] == "XYZ" and costs[order[0]] < 0.97 * costs[order[1]]
    assert abs(delays[order[0]] - synth.D_TRUE) <= 0.004
