"""The measured table of BASELINE.md is generated from the files under profiles/ (tools/make_baseline_tables.py): a
figure quoted there that the committed JSON / CSV do not give fails here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_baseline_table_is_what_the_profiles_give():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_baseline_tables.py"), "r6", "--check"],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_design_quotes_the_committed_kernel_statistics():
    """the figures DESIGN.md quotes for the two rooflines and the step are the ones profiles/r6_* give (three rounds in a
    row a csv was re-collected and the prose was not)"""
    import csv
    import json
    prof = os.path.join(ROOT, "profiles")
    stats = list(csv.DictReader(open(os.path.join(prof, "r6_bench_kernel_stats.csv"))))
    k2 = [r for r in stats if "lmeds_kernel<8, 0, 80, true, false, 256>" in r["Name"]][0]
    k1 = [r for r in stats if "loss64_kernel<8, true, false" in r["Name"]][0]
    bench = [json.loads(l) for l in open(os.path.join(prof, "r6_bench.json")) if l.startswith('{"metric"')][0]
    k2_ms, k1_us = float(k2["AverageNs"]) / 1e6, float(k1["AverageNs"]) / 1e3
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    want = ["**%.2f ms**" % k2_ms,                                                  # K2's mean launch time
            "**%.3f** of 8 TB/s" % (4096 * 2048 * 800 * 32 / k2_ms / 1e9 / 8),      # its contract roofline
            "mean **%.1f µs**" % k1_us,                                              # K1's gradient launch
            "%.3f of 8 TB/s" % (4096 * 2048 * 64 / k1_us / 1e6 / 8),
            "**%.2f ms = " % bench["ms_per_step"]]                                   # the step of the committed bench line
    missing = [w for w in want if w not in design]
    assert not missing, missing


def test_design_gyro_rate_table_is_the_committed_sweep():
    """DESIGN.md section 3's gyro-rate table and its small-frame paragraphs against profiles/r6_gyro_rate_sweep.json,
    r4_gyro_rate_small_frames.json (round 4's measurement build, which set the 144-knot rule) and r5_gyro_rate_small_frames.json
    (the compact windows) (the table was once left behind by a re-collection within the hour)"""
    import json
    prof = os.path.join(ROOT, "profiles")
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    sweep = json.load(open(os.path.join(prof, "r6_gyro_rate_sweep.json")))["by_gyro_hz"]
    want = []
    for hz, row in sweep.items():
        want.append("%.2f" % row["large"]["ms_per_launch"]["lmeds"])                      # K2 per launch
        if "large_general_path" in row:
            want.append("[%.2f = " % row["large_general_path"]["ms_per_launch"]["lmeds"])
        if int(hz) in (400, 4000):
            want.append("%.1f" % (1e3 * row["small_bounded"]["sync_points_s"]))           # 98 sync points, iterations capped
    small = json.load(open(os.path.join(prof, "r4_gyro_rate_small_frames.json")))["by_gyro_hz"]
    for hz, row in small.items():
        want.append("%.1f" % (1e3 * row["small_bounded"]["sync_points_s"]))
        want.append("%.1f" % (1e3 * row["small_general_path_bounded"]["sync_points_s"]))
    small5 = json.load(open(os.path.join(prof, "r5_gyro_rate_small_frames.json")))["by_gyro_hz"]
    for hz, row in small5.items():
        want.append("%.1f" % (1e3 * row["small_bounded"]["sync_points_s"]))
        if int(hz) in (6000, 8000):
            want.append("%.1f" % (1e3 * row["small_round4_window_rule_bounded"]["sync_points_s"]))
    missing = [w for w in want if w not in design]
    assert not missing, missing


def test_public_headers_do_not_claim_what_size_classes_retired():
    """Until round 5 the kernel family -- and with it a frame's bits -- followed the LARGEST frame of the whole problem: one
    frame above 8192 tracks sent every frame through the slow exact kernels, one above 512 sent a clip out of the one-wave
    kernels.  Since round 5 a frame's kernels follow its own track count (rssync_kernels.hip: class_of).  include/rssync.h
    went on telling clients the old behaviour for a round; the public headers and INTEGRATION.md must not say
    it again.  (`RSSYNC_EXEC_BIG_MAX` legitimately speaks of "windows whose largest frame": a policy of the window executor,
    not a statement about which kernels a frame runs.)"""
    import re
    retired = [r"follows the problem'?s largest frame", r"sends every frame of the problem", r"kernel\s+family follows the (problem|largest)",
               r"the largest frame of the (whole )?problem (decides|selects|chooses)"]
    for name in ("include/rssync.h", "include/rssync_c.h", "include/rssync_hip.h", "INTEGRATION.md"):
        text = re.sub(r"\s+", " ", open(os.path.join(ROOT, name)).read())
        text = re.sub(r"//|/\*|\*/| \* ", " ", text)
        text = re.sub(r"\s+", " ", text)
        for pat in retired:
            assert not re.search(pat, text, flags=re.I), (name, pat)
    h = re.sub(r"\s+", " ", open(os.path.join(ROOT, "include", "rssync.h")).read().replace("//", " "))
    h = re.sub(r"\s+", " ", h)
    assert "follow from ITS OWN track count" in h and "do not depend on its neighbours" in h
