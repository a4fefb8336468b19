#!/bin/bash
# A/B of executor builds on the driver workload (98 sync points of 61 x 130) and on small frames at high gyro rates, rounds
# interleaved in one GPU call:   bash tools/exec_ab.sh ROUNDS name1 name2 ...   ("head" = the product build; others are
# rs-sync_amd/_variants/lib_NAME.so, tools/k2_build_variant.sh)
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    if [ "$v" = head ]; then lib=$PWD/rs-sync_amd/librssync_core.so; else lib=$PWD/rs-sync_amd/_variants/lib_$v.so; fi
    RSSYNC_LIB=$lib timeout -k 10 120 python tools/gpu_syncpoints.py > gpurun_out/xab_sp_$v.$r.json 2> gpurun_out/xab_sp_$v.$r.err || { echo "$v failed"; tail -3 gpurun_out/xab_sp_$v.$r.err; exit 1; }
    RSSYNC_LIB=$lib SMALL_ONLY=1 RATES=400,4000,12000 timeout -k 10 200 python tools/gpu_gyro_rate.py > gpurun_out/xab_rt_$v.$r.json 2> gpurun_out/xab_rt_$v.$r.err || { echo "$v failed (rates)"; tail -3 gpurun_out/xab_rt_$v.$r.err; exit 1; }
    python - <<PY
import json
s=json.load(open('gpurun_out/xab_sp_$v.$r.json')); g=json.load(open('gpurun_out/xab_rt_$v.$r.json'))['by_gyro_hz']
print('round $r  %-10s sync points: executor %.2f ms (identical to the chain: %s), chain %.2f ms;  capped, by gyro rate: %s' % ('$v', 1e3*s['batched_executor_s'], s['executor_identical'], 1e3*s['batched_s'],
      ', '.join('%s Hz %.2f ms' % (hz, 1e3*row['small_bounded']['sync_points_s']) for hz,row in g.items())))
PY
  done
done
