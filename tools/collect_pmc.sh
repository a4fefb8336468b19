#!/bin/bash
# PMC passes for the bench workload (run on the GPU box from the repo root):
#   bash tools/collect_pmc.sh gpurun_out/pmc
# One rocprofv3 --pmc run per counter group (counters are collected in their own runs, never with
# --kernel-trace/--stats or trace domains), then tools/pmc_aggregate.py folds them into one JSON.
set -e -o pipefail
OUT=${1:-gpurun_out/pmc}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE")
i=0
for g in "${GROUPS_[@]}"; do
    d="$ROOT/$OUT/pass$i"
    rm -rf "$d"
    (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $g -d "$d" -o pmc --output-format csv -- \
        python3 "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-frames 0 --no-driver-workload > "$ROOT/$OUT/pass$i.log" 2>&1) || \
        { echo "pass $i ($g) failed"; tail -5 "$ROOT/$OUT/pass$i.log"; }
    echo "pass $i done: $g"
    i=$((i + 1))
done
python3 "$ROOT/tools/pmc_aggregate.py" "$ROOT/$OUT" > "$ROOT/$OUT/pmc_summary.json"
echo "wrote $OUT/pmc_summary.json"
