#!/bin/bash
# every measurement script of the round, one after the other, on the GPU box (run from the repo root):
#   bash tools/measure_all.sh r2        -> gpurun_out/r2_*.json / .txt
P=${1:-r2}
run() { name=$1; shift; echo "== $name"; timeout -k 10 600 "$@" > gpurun_out/${P}_$name 2> gpurun_out/${P}_$name.err || echo "   FAILED ($?)"; tail -c 600 gpurun_out/${P}_$name; echo; }
run syncpoints.json python tools/gpu_syncpoints.py
run orientation_sweep_c5.json python tools/gpu_orientations.py
run pixels_kernel.json python tests/measure/gpu_pixels.py
run gyro_device.json python tests/measure/gpu_gyro.py
run config2_parity.json python tests/measure/gpu_config2_parity.py
run fullsize_sync_parity.json python tests/measure/gpu_fullsize_parity.py
run gyro_rate_sweep.json python tools/gpu_gyro_rate.py
run quality_drift_noisy.json python tests/measure/gpu_quality.py
# (needs the measurement build: bash tools/k2_build_variant.sh cap384 -DRSSYNC_PLAN_CAP64_SMALL_MAX=384)
[ -f rs-sync_amd/_variants/lib_cap384.so ] && run gyro_rate_small_frames.json env RSSYNC_LIB=$PWD/rs-sync_amd/_variants/lib_cap384.so SMALL_ONLY=1 RATES=4000,6000,8000,12000 python tools/gpu_gyro_rate.py
[ -x tools/ubench/_build/handoff_probe ] && run handoff_probe.json tools/ubench/_build/handoff_probe 400
[ -x tools/ubench/_build/stagec_mfma ] && run k2_mfma_raw.txt tools/ubench/_build/stagec_mfma
