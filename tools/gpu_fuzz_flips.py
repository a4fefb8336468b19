#!/usr/bin/env python3
"""Which of flip_interval()'s three terms does each LMedS flip of a randomised soak need?  (the flip check of
tests/test_gpu_fuzz.py grew a term per red soak; round 6 put a tolerance-free anchor under it -- the device's winner is the
exact arg-min of its own residuals -- and FREEZES flip_interval.  This records, for every (frame, candidate) of N random
noisy cases where the device's and the oracle's winners differ, the smallest set of terms that explains the flip:
    none   the two hypotheses' fp64 lower quartiles are within 0.5 % of each other as they stand
    A      + the direction error of a hypothesis built from rows known to 5e-7: 0.2 x 5e-7 (1/|P[i0]| + 1/|P[i1]|) / sin(angle)
    A+B    + each residual's own row error, 5e-7 / |P_i|
    A+B+C  + rank uncertainty: rows within their error of the quartile value may change sides (m order statistics)
    unexplained   none of it: a failure of the test
Reference rule: core_private.cpp:48-56.  GPU box:  python tools/gpu_fuzz_flips.py 3000 > profiles/r6_fuzz_flips.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as T          # draw_case, build, flip_interval (frozen), sample_pair
from rssync_amd import synth


def interval(P, i0, i1, use_dir, use_row, use_rank):
    """flip_interval of tests/test_gpu_fuzz.py with its terms switchable (all three on: the same numbers, asserted below)"""
    nr = np.linalg.norm(P, axis=1)
    nP = P / np.where(nr < 1e-12, 1.0, nr)[:, None]
    v = np.cross(P[i0], P[i1])
    nv = np.linalg.norm(v)
    if nv >= 1e-12:
        v = v / nv
    r = np.abs(nP @ v)
    sin_a = nv / max(nr[i0] * nr[i1], 1e-300) if nv >= 1e-12 else 1.0
    err_v = 5e-7 * (1.0 / max(nr[i0], 1e-300) + 1.0 / max(nr[i1], 1e-300)) / max(sin_a, 1e-300)
    e = np.zeros_like(r)
    if use_dir:
        e = e + 0.2 * err_v * np.linalg.norm(v)
    if use_row:
        e = e + 5e-7 / np.maximum(nr, 1e-300)
    order = np.argsort(r)
    kq = len(P) // 4
    rq = r[order[kq]]
    m = max(int(np.sum(np.abs(r - rq) <= e)) - 1, 0) if use_rank else 0
    lo_k, hi_k = max(kq - m, 0), min(kq + m, len(P) - 1)
    return float(r[order[lo_k]] - e[order[lo_k]]), float(r[order[hi_k]] + e[order[hi_k]]), float(rq)


def explained(P, pair_h, pair_o, terms):
    iv = [interval(P, *pair_h, *terms), interval(P, *pair_o, *terms)]
    gap = max(iv[0][0], iv[1][0]) - min(iv[0][1], iv[1][1])
    return gap <= 5e-3 * iv[1][2], gap / max(iv[1][2], 1e-300)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    LEVELS = [("none", (False, False, False)), ("A", (True, False, False)), ("A+B", (True, True, False)), ("A+B+C", (True, True, True))]
    counts = {name: 0 for name, _ in LEVELS}
    counts["unexplained"] = 0
    pairs = flips_total = 0
    worst = []
    for seed in range(first, first + n_cases):
        rng, g, frames, cnt = T.draw_case(seed, clean=False)
        h, o = T.build(seed, g, frames)
        ids = [fr[0] for fr in frames]
        lo, hi = ids[0], ids[-1] + 1
        step = float(rng.choice([0.0005, 0.001, 0.002, 0.004]))
        radius = float(rng.uniform(0.005, 0.06))
        centre = synth.D_TRUE + float(rng.uniform(-0.01, 0.01))
        nf = len(frames)
        dh, ch, fch, bhh = h.presync_curve(centre, lo, hi, step, radius, per_frame=nf)
        do, co, fco, bho = o.presync_curve(centre, lo, hi, step, radius, per_frame=nf)
        big = np.array([n >= 48 for n in cnt])
        pairs += int(big.sum()) * len(dh)
        for c, j in zip(*np.nonzero(bhh != bho)):
            if not big[j] or bhh[c, j] < 0 or bho[c, j] < 0:
                continue
            flips_total += 1
            P = o.problem_matrix(ids[j], float(do[c]))
            ph = T.sample_pair(seed, ids[j], int(c), int(bhh[c, j]), cnt[j])
            po = T.sample_pair(seed, ids[j], int(c), int(bho[c, j]), cnt[j])
            assert interval(P, *ph, True, True, True) == T.flip_interval(P, *ph)      # (the frozen function, term for term)
            need = "unexplained"
            for name, terms in LEVELS:
                ok, rel = explained(P, ph, po, terms)
                if ok:
                    need = name
                    break
            counts[need] += 1
            if need in ("A+B+C", "unexplained") and len(worst) < 40:
                nr = np.linalg.norm(P, axis=1)
                worst.append({"seed": seed, "candidate": int(c), "frame": int(ids[j]), "tracks": int(cnt[j]), "needs": need,
                              "quartile_rel_diff_plain": explained(P, ph, po, (False, False, False))[1],
                              "smallest_defining_row": float(min(nr[list(ph) + list(po)]))})
        if (seed - first) % 100 == 99:
            print("... %d cases, %d flips %s" % (seed - first + 1, flips_total, counts), file=sys.stderr, flush=True)
    print(json.dumps({"cases": n_cases, "first_seed": first, "pairs_on_frames_of_48_tracks_or_more": pairs, "flips": flips_total,
                      "flip_rate": flips_total / max(pairs, 1), "terms_needed": counts,
                      "what": "smallest set of flip_interval's terms (tests/test_gpu_fuzz.py, frozen in round 6) under which the two "
                              "hypotheses' quartile intervals meet to within 0.5 % -- none: the plain fp64 quartiles already do",
                      "flips_that_needed_the_rank_term_or_more": worst}, indent=1))


if __name__ == "__main__":
    main()
