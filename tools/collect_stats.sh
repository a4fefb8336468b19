#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (GPU box, from the repo root):
#   bash tools/collect_stats.sh gpurun_out/stats
# with --cpu-frames 0: the timed region is the default run's, but the cpu_baseline leg and (since round 5) the parity block's
# extra HIP pass over the 192-frame sample are left out -- that pass launches the SAME kernels on a 21 times smaller problem
# and would enter every kernel's mean duration (round 5's first collection: lmeds_kernel "mean 35.1 ms" over 6 + 1 calls).
# Leaves <out>/bench.json (the bench line of the profiled run) and <out>/kernel_stats.csv.
set -e -o pipefail
OUT=${1:-gpurun_out/stats}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf "$ROOT/$OUT/prof"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/prof" -o bench --output-format csv -- \
    python3 "$ROOT/bench.py" --cpu-frames 0 --no-driver-workload > "$ROOT/$OUT/bench.log" 2>&1)
grep "^{\"metric\"" "$ROOT/$OUT/bench.log" > "$ROOT/$OUT/bench.json"
f=$(find "$ROOT/$OUT/prof" -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/$OUT/kernel_stats.csv"
head -12 "$ROOT/$OUT/kernel_stats.csv"
