#!/bin/bash
# A/B of kernel builds on the GPU box: variants are prebuilt .so files under rs-sync_amd/_variants/
# (built here with tools/k2_build_variant.sh); rounds are interleaved in one call (same device, same session).
# A variant is selected through RSSYNC_LIB (rssync_amd.problem.library_path): the product .so is never touched.
#   [K2_AB_ARGS="--tracks 1024"] bash tools/k2_ab.sh ROUNDS name1 name2 ...      ("head" = the product build; "@VAR=VALUE" = the product build
#                                                     with that environment variable, e.g. @RSSYNC_K2_EXACT_SELECT=1)
ROUNDS=$1; shift
for v in "$@"; do
  [ "$v" = head ] || [ "${v:0:1}" = "@" ] || [ -f rs-sync_amd/_variants/lib_$v.so ] || { echo "missing variant $v" >&2; exit 2; }
done
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    envset=RSSYNC_AB_NONE=1
    if [ "$v" = head ]; then lib=$PWD/rs-sync_amd/librssync_core.so
    elif [ "${v:0:1}" = "@" ]; then lib=$PWD/rs-sync_amd/librssync_core.so; envset=${v:1}
    else lib=$PWD/rs-sync_amd/_variants/lib_$v.so; fi
    env "$envset" RSSYNC_LIB=$lib timeout -k 10 120 python bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-driver-workload $K2_AB_ARGS > gpurun_out/ab_$v.$r.log 2>&1 || { echo "variant $v failed (round $r)"; tail -3 gpurun_out/ab_$v.$r.log; exit 1; }
    python - <<PY
import json
for line in open('gpurun_out/ab_$v.$r.log'):
    if line.startswith('{"metric"'):
        d=json.loads(line); k=d['kernels']['lmeds']; s=d['steps']; K=d['kernels']
        print('round $r  %-12s lmeds %.3f ms/launch  presync %.2f ms  step %.2f ms  loss %.2f grad %.2f motion %.2f reduce %.2f' % ('$v', k['total_ms']/k['launches'], d['presync_ms_per_step'], d['ms_per_step'], K['loss']['total_ms']/s, K.get('loss_grad',{'total_ms':0})['total_ms']/s, K['motion']['total_ms']/s, K['reduce']['total_ms']/s))
PY
  done
done
