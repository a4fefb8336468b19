"""Near-static camera: |P| = |ar x br| ~ translation / depth is tiny at the true delay, so |P[i0] x P[i1]| falls below
safe_normalize's 1e-12 (core_private.cpp:45-46, inline_utils.hpp:5-11) and the reference leaves the hypothesis direction
UN-normalised: its residuals shrink with it and it wins the LMedS outright.  Device vs oracle per (frame, candidate) for a
range of translations: fraction of identical winners, how many hypotheses the oracle left un-normalised, cost agreement.
Round 6: the sweep recomputes such (frame, candidate) pairs from the fp64 streams (kernels/lmeds.hpp, "fp64 rows"); the
"fp32_rows" block of every case is the same library with RSSYNC_NO_FP64_ROWS=1 (round 5's behaviour), `fp64_pairs` how many
pairs took the fp64 form.
GPU box:  python tests/measure/gpu_near_static.py > profiles/r6_near_static.json   (N=130 / 600 / 2048 / 3000 via the environment)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rssync_amd
from rssync_amd import synth
from oracle.oracle import OracleProblem

SEED = 123
F, N = 12, int(os.environ.get("N", 600))
out = {"frames": F, "tracks": N, "cases": []}
# (ray noise in proportion to the rows' size, as in the ordinary scene: 1e-3 rad on |P| ~ 2e-3)
def run(translation, noise, fp64_rows):
    g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=9)
    frames = list(synth.make_frames(g, 0, F, N, seed=9, noise=noise, outliers=0.1, translation=translation))
    if fp64_rows:
        os.environ.pop("RSSYNC_NO_FP64_ROWS", None)
    else:
        os.environ["RSSYNC_NO_FP64_ROWS"] = "1"
    try:
        h = rssync_amd.SyncProblem(seed=SEED)
    finally:
        os.environ.pop("RSSYNC_NO_FP64_ROWS", None)
    o = OracleProblem(seed=SEED, threads=min(os.cpu_count() or 1, 16), faithful=False)
    for p in (h, o):
        p.SetGyroQuaternions(g.quats, g.fs, g.t0)
        for fr in frames:
            p.SetTrackResult(*fr)
    P = o.problem_matrix(3, synth.D_TRUE)
    norms = np.linalg.norm(P, axis=1)
    dh, ch, fch, bhh = h.presync_curve(synth.D_TRUE, 0, F, 2e-6, 2e-5, per_frame=F)
    do, co, fco, bho = o.presync_curve(synth.D_TRUE, 0, F, 2e-6, 2e-5, per_frame=F)
    same = bhh == bho
    rel = np.abs(fch - fco) / np.maximum(fco, 1e-300)
    # how many of the oracle's winners are UN-normalised directions (|M| far below 1), on a sample of (frame, candidate)
    unn = [float(np.linalg.norm(o.guess_motion(f, float(do[c]), 20, c)[0])) < 0.5 for f in range(0, F, 3) for c in range(0, len(do), 4)]
    st = h.near_static_stats()
    return {"translation_m": translation, "ray_noise_rad": noise, "oracle_winners_left_unnormalised": float(np.mean(unn)), "median_row_norm_at_true_delay": float(np.median(norms)),
            "candidates": int(len(dh)), "pairs": int(same.size), "fp64_pairs": int(st["pairs"]), "same_winner": float(same.mean()),
            "cost_rel_where_same": {"median": float(np.median(rel[same])) if same.any() else None,
                                    "max": float(rel[same].max()) if same.any() else None},
            "same_argmin": bool(np.argmin(ch) == np.argmin(co)),
            "curve_rel_max": float(np.abs(ch - co).max() / np.abs(co).max())}


# (ray noise in proportion to the rows' size, as in the ordinary scene: 1e-3 rad on |P| ~ 2e-3)
for translation, noise in ((0.05, 1e-3), (2.5e-3, 5e-5), (2.5e-4, 5e-6), (5e-5, 1e-6), (1e-5, 2e-7)):
    case = run(translation, noise, True)
    before = run(translation, noise, False)
    case["fp32_rows"] = {k: before[k] for k in ("same_winner", "cost_rel_where_same", "same_argmin", "curve_rel_max", "fp64_pairs")}
    out["cases"].append(case)
print(json.dumps(out, indent=1))
