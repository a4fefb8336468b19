"""The measured table of BASELINE.md is generated from the files under profiles/ (tools/make_baseline_tables.py): a
figure quoted there that the committed JSON / CSV do not give fails here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_baseline_table_is_what_the_profiles_give():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_baseline_tables.py"), "r4", "--check"],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
