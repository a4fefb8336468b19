#!/bin/bash
# A/B of K2 builds on frames of 2049 .. 4096 tracks (size class 3):  bash tools/k2c3_ab.sh ROUNDS name1 name2 ...
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do for v in "$@"; do
  if [ "$v" = head ]; then lib=$PWD/rs-sync_amd/librssync_core.so; else lib=$PWD/rs-sync_amd/_variants/lib_$v.so; fi
  RSSYNC_LIB=$lib SIZES=2048,3000,4096 REPS=3 python tools/gpu_by_class.py > gpurun_out/c3ab_$v.$r.json 2>/dev/null || { echo "$v failed"; exit 1; }
  python - <<PY
import json
d=json.load(open('gpurun_out/c3ab_$v.$r.json'))['by_tracks']
print('round $r  %-7s ' % '$v' + '  '.join('%s tracks: K2 %.3f ms (window %s knots, dynamic %s, chunk %s)' % (n, r['lmeds_kernel_ms'], r['windows'].get('presync_window_knots'), r['windows'].get('presync_window_dynamic'), r['windows'].get('presync_chunk')) for n,r in d.items()))
PY
done; done
