#!/bin/bash
# A/B of lmeds_kernel builds on the GPU box: variants are prebuilt .so files under rs-sync_amd/_variants/
# (built here with tools/k2_build_variant.sh); rounds are interleaved in one call (same device, same session).
#   bash tools/k2_ab.sh ROUNDS name1 name2 ...
ROUNDS=$1; shift
cp rs-sync_amd/librssync_core.so /tmp/lib_orig.so
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    cp rs-sync_amd/_variants/lib_$v.so rs-sync_amd/librssync_core.so
    timeout -k 10 120 python bench.py --steps 6 --warmup 2 --cpu-frames 0 > gpurun_out/ab_$v.$r.log 2>&1
    python - <<PY
import json
for line in open('gpurun_out/ab_$v.$r.log'):
    if line.startswith('{"metric"'):
        d=json.loads(line); k=d['kernels']['lmeds']; print('round $r  %-12s lmeds %.3f ms/launch  presync %.2f ms  step %.2f ms  loss %.2f motion %.2f' % ('$v', k['total_ms']/k['launches'], d['presync_ms_per_step'], d['ms_per_step'], d['kernels']['loss']['total_ms']/d['steps'], d['kernels']['motion']['total_ms']/d['steps']))
PY
  done
done
cp /tmp/lib_orig.so rs-sync_amd/librssync_core.so
