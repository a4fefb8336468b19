#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (GPU box, from the repo root):
#   bash tools/collect_stats.sh gpurun_out/stats
# Leaves <out>/bench.json (the bench line of the profiled run) and <out>/kernel_stats.csv.
set -e -o pipefail
OUT=${1:-gpurun_out/stats}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf "$ROOT/$OUT/prof"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/prof" -o bench --output-format csv -- \
    python3 "$ROOT/bench.py" > "$ROOT/$OUT/bench.log" 2>&1)
grep "^{\"metric\"" "$ROOT/$OUT/bench.log" > "$ROOT/$OUT/bench.json"
f=$(find "$ROOT/$OUT/prof" -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/$OUT/kernel_stats.csv"
head -12 "$ROOT/$OUT/kernel_stats.csv"
