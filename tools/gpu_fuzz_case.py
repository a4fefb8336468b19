"""One randomised case of tests/test_gpu_fuzz.py::test_random_noisy_case looked at closely (GPU box): where the device's
and the oracle's LMedS winners differ, are the two hypotheses' lower quartiles a near-tie in the reference's own fp64
arithmetic?  (A flip at a near-tie is the fp32 search's stated behaviour; anything else is a bug.)

    python tools/gpu_fuzz_case.py SEED [SEED ...]  > gpurun_out/fuzz_case.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rssync_amd  # noqa: E402
from rssync_amd import synth  # noqa: E402
from oracle import oracle as ora  # noqa: E402
import test_gpu_fuzz as fz  # noqa: E402


quartile_of = fz.lower_quartile_fp64


def look(seed):
    rng, g, frames, counts = fz.draw_case(seed, clean=False)
    h, o = fz.build(seed, g, frames)
    ids = [fr[0] for fr in frames]
    lo, hi = ids[0], ids[-1] + 1
    for _ in range(3):   # (the draws the test makes before the sweep's)
        int(rng.integers(len(frames)))
        float(rng.choice([rng.uniform(-0.05, 0.08), rng.uniform(-3.0, 3.0)]))
    step = float(rng.choice([0.0005, 0.001, 0.002, 0.004]))
    radius = float(rng.uniform(0.005, 0.06))
    centre = synth.D_TRUE + float(rng.uniform(-0.01, 0.01))
    nf = len(frames)
    dh, ch, fch, bhh = h.presync_curve(centre, lo, hi, step, radius, per_frame=nf)
    do, co, fco, bho = o.presync_curve(centre, lo, hi, step, radius, per_frame=nf)
    out = {"seed": seed, "gyro_hz": g.fs, "tracks": counts, "candidates": len(dh), "step": step, "flips": []}
    for c, j in zip(*np.nonzero(bhh != bho)):
        n = counts[j]
        if n < 48:
            continue
        P = o.problem_matrix(ids[j], float(do[c]))
        # the oracle's own winner, from which the sampler's stream of this candidate follows
        stream = None
        for s in (int(c), int(c) + 1):
            if o.frame_presync_cost(ids[j], float(do[c]), s)[1] == int(bho[c, j]):
                stream = s
                break
        rec = {"candidate": int(c), "frame": int(ids[j]), "tracks": n, "winner_hip": int(bhh[c, j]), "winner_oracle": int(bho[c, j]),
               "frame_cost_hip": float(fch[c, j]), "frame_cost_oracle": float(fco[c, j]), "stream": stream}
        if stream is not None:
            seed_o = seed
            q = {}
            for name, hh in (("hip", int(bhh[c, j])), ("oracle", int(bho[c, j]))):
                i0, i1 = ora.sample_pair(seed_o, ids[j], stream, hh, n)
                q[name] = quartile_of(P, i0, i1)
                rec["pair_" + name] = [i0, i1]
            nr = np.linalg.norm(P, axis=1)
            for name in ("hip", "oracle"):
                i0, i1 = rec["pair_" + name]
                cr = np.linalg.norm(np.cross(P[i0], P[i1]))
                rec["defining_rows_" + name] = {"norms": [float(nr[i0]), float(nr[i1])], "cross_norm": float(cr),
                                                 "sin_angle": float(cr / max(nr[i0] * nr[i1], 1e-300))}
            rec["quartile_interval_hip"] = fz.flip_interval(P, *rec["pair_hip"])
            rec["quartile_interval_oracle"] = fz.flip_interval(P, *rec["pair_oracle"])
            rec["row_norm_percentiles_1_10_50"] = [float(x) for x in np.percentile(nr, [1, 10, 50])]
            rec["quartile_r2_of_hips_winner_fp64"] = q["hip"]
            rec["quartile_r2_of_oracles_winner_fp64"] = q["oracle"]
            rec["relative_gap"] = abs(q["hip"] - q["oracle"]) / max(q["oracle"], 1e-300)
            rec["min_row_norm"] = float(np.linalg.norm(P, axis=1).min())
        out["flips"].append(rec)
    big = np.array([n >= 48 for n in counts])
    cbh, cbo = fch[:, big].sum(axis=1), fco[:, big].sum(axis=1)
    rel = np.abs(cbh - cbo) / cbo
    out["worst_candidate"] = int(np.argmax(rel))
    out["worst_rel"] = float(rel.max())
    out["frames_with_48_tracks_or_more"] = int(big.sum())
    return out


if __name__ == "__main__":
    print(json.dumps([look(int(s)) for s in sys.argv[1:]], indent=1))
