"""Regenerates tests/golden/oracle_small.npz.

The reference ships no golden vectors for this path (PARITY UNPINNED), so these fixtures are the
ORACLE's outputs on a small committed input: they pin the oracle (and through it the HIP path)
against drift between rounds, not against the reference.  Inputs are stored in the file, so the
fixture does not depend on numpy's or scipy's generators staying the same.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import rssync_amd  # noqa: E402,F401
from rssync_amd import synth  # noqa: E402
from oracle import oracle as ora  # noqa: E402
from oracle.oracle import OracleProblem  # noqa: E402

SEED = 20261003


def main():
    F, N = 24, 128
    gyro = synth.make_gyro(1.0, 1.0 + (F + 2) / synth.FPS, seed=41)  # starts at t = 0
    frames = list(synth.make_frames(gyro, 30, 30 + F, N, seed=41))
    ts_us, q_ts = synth.make_timestamped(gyro, jitter=0.2, seed=42)
    o = OracleProblem(seed=SEED, threads=1, faithful=True)
    o.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr, ta, tb, ra, rb in frames:
        o.SetTrackResult(fr, ta, tb, ra, rb)
    delays, costs, fcost, bh = o.presync_curve(0.0, 30, 30 + F, 0.004, 0.1, per_frame=F)
    pre = o.PreSync(0.0, 30, 30 + F, 0.004, 0.1)
    dbg_d, dbg_c = o.DebugPreSync(0.0, 30, 30 + F, 0.1, 9)
    c, d, trace = o.sync_trace(pre[1], 30, 30 + F - 1, 0.0, 0.1)
    M, k = o.sync_state()
    P = o.problem_matrix(33, 0.0371)
    o2 = OracleProblem(seed=SEED)
    o2.SetGyroQuaternionsTimestamped(ts_us, q_ts)
    fs2, start2, n2 = o2.gyro_info()
    # a noise-free scene: there the comparison with the product is tight (1e-7 s per outer iteration),
    # while on the noisy one above the optimiser amplifies rounding differences (DESIGN.md "Parity")
    Fc, Nc = 16, 96
    gyro_c = synth.make_gyro(0.0, (Fc + 2) / synth.FPS, seed=43)
    frames_c = list(synth.make_frames(gyro_c, 0, Fc, Nc, seed=43, noise=0.0, outliers=0.0))
    oc = OracleProblem(seed=SEED, threads=1, faithful=True)
    oc.SetGyroQuaternions(gyro_c.quats, gyro_c.fs, gyro_c.t0)
    for fr, ta, tb, ra, rb in frames_c:
        oc.SetTrackResult(fr, ta, tb, ra, rb)
    cc, dc_, trace_c = oc.sync_trace(0.0355, 0, Fc - 1, 0.0, 0.1)
    cs, ds_, trace_s = oc.sync_simplified_trace(0.0355, 0, Fc - 1, 0.0, 0.1)
    pairs = np.array([ora.sample_pair(SEED, fr, st, h, n) for fr, st, h, n in
                      [(0, 0, 0, 2), (30, 5, 7, 128), (-3, ora.STREAM_SYNC_INIT, 199, 2048), (2 ** 40, ora.STREAM_DEBUG + 3, 19, 17)]])
    np.savez_compressed(
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small.npz"),
        seed=SEED, gyro_quats=gyro.quats, gyro_fs=gyro.fs, gyro_t0=gyro.t0,
        frame_ids=np.array([f[0] for f in frames]), ts_a=np.array([f[1] for f in frames]),
        ts_b=np.array([f[2] for f in frames]), rays_a=np.array([f[3] for f in frames]),
        rays_b=np.array([f[4] for f in frames]),
        presync_delays=delays, presync_costs=costs, presync_frame_costs=fcost, presync_best_h=bh,
        presync_result=np.array(pre), debug_delays=dbg_d, debug_costs=dbg_c,
        sync_result=np.array([c, d]), sync_trace=trace, sync_M=M, sync_k=k, P_frame33=P,
        clean_gyro_quats=gyro_c.quats, clean_gyro_fs=gyro_c.fs, clean_gyro_t0=gyro_c.t0,
        clean_ts_a=np.array([f[1] for f in frames_c]), clean_ts_b=np.array([f[2] for f in frames_c]),
        clean_rays_a=np.array([f[3] for f in frames_c]), clean_rays_b=np.array([f[4] for f in frames_c]),
        clean_sync_result=np.array([cc, dc_]), clean_sync_trace=trace_c,
        clean_simplified_result=np.array([cs, ds_]), clean_simplified_trace=trace_s,
        ts_us=ts_us, ts_quats=q_ts, ts_fs=fs2, ts_start=start2, ts_knots=o2.gyro_knots(), sample_pairs=pairs)
    print("written", F, N, "presync", pre, "sync", c, d, len(trace))


if __name__ == "__main__":
    main()
