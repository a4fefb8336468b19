#!/bin/bash
# Runs a list of GPU steps one after the other on the GPU box; each under its own timeout, output to
# gpurun_out/<prefix>_<name>.log.  A step that fails an assertion does not stop the list; a step that is KILLED
# (timeout, signal) does -- nothing else is started on a GPU that may be hung.
#   bash tools/gpu_steps.sh PREFIX "name|seconds|command" ...
P=$1; shift
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/${P}_${name}.log" 2>&1
  rc=$?
  echo "   rc=$rc  $(( $(date +%s) - start ))s"
  tail -n 6 "gpurun_out/${P}_${name}.log" | cut -c1-400
  if [ $rc -ge 124 ]; then echo "   step killed: stopping here"; exit $rc; fi
done
exit 0
