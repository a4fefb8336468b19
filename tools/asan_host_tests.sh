#!/bin/bash
# AddressSanitizer + UBSan over the host solver (sync_problem.cpp) linked to the CPU stand-in for the device
# (tests/cpu_device/rship_cpu.cpp): the host-logic, multi-device, golden and boundary tests, no GPU involved.
# (GPU sanitizers are not available on the pool; this covers everything that is not a kernel.)
#   bash tools/asan_host_tests.sh          -> prints the pytest summary and any sanitizer report lines
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p tests/_build
LIB=tests/_build/librssync_hosttest.so
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -o $LIB rs-sync_amd/csrc/sync_problem.cpp tests/cpu_device/rship_cpu.cpp
touch $LIB
ASAN=$(gcc -print-file-name=libasan.so)
STDCXX=$(gcc -print-file-name=libstdc++.so.6)   # before python's own: the runtime must find __cxa_throw
LD_PRELOAD="$ASAN $STDCXX" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests/test_host_logic.py tests/test_multi_device.py tests/test_golden.py tests/test_host_boundary.py \
    -q -m "not gpu" 2>&1 | tee /tmp/asan_host.log | tail -3
grep -n "runtime error\|ERROR: AddressSanitizer" /tmp/asan_host.log || echo "no sanitizer reports"
rm -f $LIB   # the next pytest run rebuilds the plain library
