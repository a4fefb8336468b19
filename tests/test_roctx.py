"""Optional roctx ranges (rs-sync_amd/csrc/roctx_ranges.hpp; SURVEY.md section 5 planned "roctx ranges around K1-K3"): with
RSSYNC_ROCTX=1 the public calls and the launch kinds are bracketed by roctxRangePushA / roctxRangePop of whatever roctx
library the process can load, so that `rocprofv3 --marker-trace --kernel-trace` of a client shows PreSync / Sync / sync-point
boundaries.  Checked on the CPU: the host solver linked to the CPU stand-in, a stub roctx library that writes what it is
called with to a file.  Without the switch nothing is loaded and nothing is called."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
#include <stdio.h>
#include <stdlib.h>
static int depth = 0;
static void say(const char* what, const char* name) {
    const char* path = getenv("ROCTX_STUB_LOG");
    if (!path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "%d %s %s\n", depth, what, name ? name : "");
    fclose(f);
}
int roctxRangePushA(const char* name) { say("push", name); return depth++; }
int roctxRangePop(void) { --depth; say("pop", ""); return depth; }
'''

DRIVER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import ctypes
import rssync_amd
from rssync_amd import synth
from rssync_amd.problem import bind
lib = bind(ctypes.CDLL(%(lib)r))
F, N = 8, 48
g = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=3)
p = rssync_amd.SyncProblem(seed=3, max_outer_iters=3, _lib=lib)
synth.fill(p, g, 0, F, N, seed=3)
d = p.PreSync(0.0, 0, F, 0.004, 0.05)[1]
p.Sync(d, 0, F - 1, 0.0, 0.1)
p.sync_points([0, 2], 4, 0.0, 0.004, 0.05, repeats=2)
'''


def _run(tmp_path, hosttest_lib, roctx_on):
    stub_dir = tmp_path / "stub"
    stub_dir.mkdir(exist_ok=True)
    src = stub_dir / "roctx_stub.c"
    src.write_text(STUB)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(stub_dir / "librocprofiler-sdk-roctx.so"), str(src)])
    log = tmp_path / ("roctx_%d.log" % int(roctx_on))
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = str(stub_dir) + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env["ROCTX_STUB_LOG"] = str(log)
    env.pop("RSSYNC_ROCTX", None)
    if roctx_on:
        env["RSSYNC_ROCTX"] = "1"
    code = DRIVER % {"root": ROOT, "lib": os.path.join(ROOT, "tests", "_build", "librssync_hosttest.so")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    return log.read_text().splitlines() if log.exists() else []


def test_ranges_around_the_public_calls_when_switched_on(hosttest_lib, tmp_path):
    lines = _run(tmp_path, hosttest_lib, True)
    pushes = [l.split(" ", 2) for l in lines if " push " in l]
    names = [p[2] for p in pushes]
    for want in ("rssync:SetGyroQuaternions", "rssync:PreSync", "rssync:Sync", "rssync:sync_points", "rssync:pack_frames (upload + packing kernel)"):
        assert want in names, (want, names[:20])
    # ranges nest and balance: every push has its pop, the public calls open at depth 0 and their work inside them
    assert sum(1 for l in lines if " push " in l) == sum(1 for l in lines if " pop " in l)
    depth_of = {p[2]: int(p[0]) for p in pushes}
    assert depth_of["rssync:PreSync"] == 0 and depth_of["rssync:Sync"] == 0 and depth_of["rssync:sync_points"] == 0
    assert depth_of["rssync:pack_frames (upload + packing kernel)"] == 1          # inside the first call that needed the device
    assert int(lines[-1].split(" ", 1)[0]) == 0


def test_nothing_is_loaded_or_called_without_the_switch(hosttest_lib, tmp_path):
    assert _run(tmp_path, hosttest_lib, False) == []
