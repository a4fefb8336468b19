"""How often does a line search of the restated ens::L_BFGS end with best step != last step, and what
does the choice made there (published: keep the last trial's value and gradient; variant: re-evaluate
at the best step) do to the delay Sync returns?  CPU only (the oracle), BASELINE configs 1 and a
256-frame sample of config 3; writes profiles/r2_lbfgs_fidelity.json.

    python tests/measure/lbfgs_fidelity.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from rssync_amd import synth  # noqa: E402
from oracle.oracle import OracleProblem  # noqa: E402


def run(F, N, seed, iters, noise, outliers, d0):
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=seed)
    frames = list(synth.make_frames(gyro, 0, F, N, seed=seed, noise=noise, outliers=outliers))
    out = {}
    for name, reeval in (("published", False), ("reeval", True)):
        o = OracleProblem(seed=seed, max_outer_iters=iters, threads=os.cpu_count() or 1, faithful=False,
                          lbfgs_reeval=reeval)
        o.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
        for fr in frames:
            o.SetTrackResult(*fr)
        t = time.time()
        c, d, tr = o.sync_trace(d0, 0, F - 1, 0.0, 0.2)
        out[name] = dict(delay=d, cost=c, outer_iters=len(tr), best_not_last=o.lbfgs_best_not_last(),
                         seconds=round(time.time() - t, 2))
    out["delay_difference_s"] = abs(out["published"]["delay"] - out["reeval"]["delay"])
    out["truth"] = synth.D_TRUE
    return out


def main():
    res = {
        "config1_noisy_64x256": run(64, 256, 1, 400, 1e-3, 0.10, 0.036),
        "config1_clean_64x256": run(64, 256, 2, 400, 0.0, 0.0, 0.036),
        "driver_shape_60x130_noisy": run(60, 130, 3, 400, 1e-3, 0.10, 0.036),
        "config3_sample_256x2048_20iters": run(256, 2048, 0x5EED0003, 20, 1e-3, 0.10, 0.0365),
    }
    path = os.path.join(ROOT, "profiles", "r2_lbfgs_fidelity.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
