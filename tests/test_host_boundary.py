"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol the headers
declare plus the reference's C++ symbols, and refuses loudly to run without a HIP device."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exports(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True)
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)) - {prefix + "reduce_fn"}


def test_library_exports_every_declared_symbol(built):
    import rssync_amd
    exp = _exports(rssync_amd.library_path())
    for header, prefix in (("rssync_c.h", "rssync_"), ("rssync_hip.h", "rship_")):
        decl = _declared(header, prefix)
        assert len(decl) > 10
        assert not (decl - exp), sorted(decl - exp)
    # the ctypes table binds exactly the C-ABI the header declares
    from rssync_amd.problem import SIGNATURES
    assert set(SIGNATURES) == _declared("rssync_c.h", "rssync_")


def test_reference_cxx_symbols_are_exported(built):
    """rssync.h:31 has C++ linkage; the vtable/typeinfo of ISyncProblem live where its
    out-of-line destructor is (core_private.cpp:363-365)."""
    import rssync_amd
    exp = _exports(rssync_amd.library_path())
    for sym in ("_Z17CreateSyncProblemv", "_ZN12ISyncProblemD0Ev", "_ZN12ISyncProblemD1Ev",
                "_ZN12ISyncProblemD2Ev", "_ZTV12ISyncProblem", "_ZTI12ISyncProblem", "_ZTS12ISyncProblem"):
        assert sym in exp, sym


def test_library_loads_and_binds(built):
    import rssync_amd
    lib = rssync_amd.load_library()
    assert lib.rssync_last_error() is not None
    lib.rship_max_tracks.restype = ctypes.c_int
    assert lib.rship_max_tracks() == 1 << 24


CLIENT = r"""
#include "rssync.h"
#include <cstdio>
#include <memory>
#include <vector>
int main() {
    std::unique_ptr<ISyncProblem> sp(CreateSyncProblem());   // core_testcode.cpp:248
    std::vector<double> q = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    sp->SetGyroQuaternions(q.data(), 3, 400.0, 0.0);
    auto r = sp->PreSync(0.0, 0, 1, 0.01, 0.05);
    std::printf("%g %g\n", r.first, r.second);
    return 0;
}
"""


REFERENCE_PUBLIC = "/root/reference/src/core/public"  # present in the build container only


@pytest.mark.parametrize("header_dir", ["repo", "reference"])
def test_cxx_client_built_against_the_header_links_and_fails_loudly_without_a_gpu(built, tmp_path, header_dir):
    """A client written against the reference's interface compiles and links unchanged -- against
    this repo's include/rssync.h and, where the reference checkout is present (the build container),
    against the reference's OWN src/core/public/rssync.h: that is the drop-in claim.  Without a
    HIP device the library must not compute anything: it follows the reference's panic convention
    (panic.txt + exit status 1, core_support/panic.cpp:7-15)."""
    import torch
    inc = os.path.join(ROOT, "include")
    if header_dir == "reference":
        if not os.path.exists(os.path.join(REFERENCE_PUBLIC, "rssync.h")):
            pytest.skip("reference checkout not present on this machine")
        inc = REFERENCE_PUBLIC
    src = tmp_path / "client.cpp"
    src.write_text(CLIENT)
    exe = tmp_path / "client"
    libdir = os.path.join(ROOT, "rs-sync_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I", inc, str(src), "-o", str(exe),
                           "-L", libdir, "-lrssync_core", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device path cannot be exercised here")
    res = subprocess.run([str(exe)], cwd=tmp_path, capture_output=True, text=True)
    assert res.returncode == 1
    assert "no usable HIP device" in (tmp_path / "panic.txt").read_text()
    assert "no CPU fallback" in res.stderr


def test_python_mirror_fails_loudly_without_a_gpu(built):
    import torch
    import rssync_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rssync_amd.RsSyncError, match="no usable HIP device"):
        rssync_amd.SyncProblem()


def test_product_does_not_reference_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(ROOT, "rs-sync_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_build" in dirpath:
            continue
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                for line in text.splitlines():
                    if re.search(r"^\s*(#\s*include|import|from)\b.*oracle", line):
                        raise AssertionError(f"{fn}: {line.strip()}")
    import rssync_amd
    out = subprocess.check_output(["ldd", rssync_amd.library_path()], text=True)
    assert "oracle" not in out


def test_fp64_log1p_rcp_of_the_motion_kernel(built, tmp_path):
    """rs::log1p_rcp_f64 (device_math.hpp; the motion optimiser's objective term) against libm
    over 45 decades: a few ulp for log1p(u) and 1/(1+u)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    src.write_text('''
#include "%s/rs-sync_amd/csrc/device_math.hpp"
#include <cmath>
#include <cstdio>
#include <random>
int main() {
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> ex(-30, 15), un(0, 4);
    double worst = 0, worst_rc = 0;
    const double fixed[] = {0.0, 5e-324, 1e-300, 1.0, 0.41421356237309515, 0.4142135623730949, 1e15, 3.0};
    for (int i = 0; i < 2000000; ++i) {
        const double u = i < 8 ? fixed[i] : (i %% 5 == 0 ? un(g) : std::pow(10.0, ex(g)));
        double rc;
        const double l = rs::log1p_rcp_f64(u, &rc), ref = std::log1p(u), rr = 1.0 / (1.0 + u);
        const double e1 = ref != 0 ? std::fabs(l - ref) / ref : std::fabs(l), e2 = std::fabs(rc - rr) / rr;
        worst = e1 > worst ? e1 : worst;
        worst_rc = e2 > worst_rc ? e2 : worst_rc;
    }
    std::printf("%%.3g %%.3g\\n", worst, worst_rc);
    return 0;
}
''' % root)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", str(exe), str(src)])
    worst, worst_rc = map(float, subprocess.check_output([str(exe)], text=True).split())
    assert worst < 1e-15 and worst_rc < 5e-16


def test_c_headers_are_valid_c99(tmp_path):
    """include/rssync_c.h and include/rssync_hip.h are consumed by C hosts (cgo, JNI glue): they
    must compile as plain C99, and a C client must link against the library."""
    import rssync_amd
    src = tmp_path / "client.c"
    src.write_text(r'''
#include "rssync_c.h"
#include "rssync_hip.h"
#include <stdio.h>
int main(void) {
    rssync_set_panic_mode(1);
    rssync_problem* p = rssync_create();          /* NULL here: no GPU, no CPU fallback */
    rssync_lens lens = {0.011, 1000, 1000, 500, 400, 0, 0, 0, 0};
    (void)lens;
    printf("%s|%d\n", p ? "created" : rssync_last_error(), rship_max_tracks());
    rssync_destroy(p);
    return 0;
}
''')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    libdir = os.path.dirname(rssync_amd.library_path())
    exe = tmp_path / "client"
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lrssync_core",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    import torch
    out = subprocess.run([str(exe)], capture_output=True, text=True, cwd=tmp_path)
    assert out.returncode == 0
    msg, tracks = out.stdout.strip().rsplit("|", 1)
    assert tracks == str(1 << 24)
    if not torch.cuda.is_available():
        assert "no usable HIP device" in msg and "no CPU fallback" in msg
