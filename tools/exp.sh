timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "device_driven or noisy or clean_data or vtable or sync_points or batched" > gpurun_out/r2_loop_tests.log 2>&1; tail -15 gpurun_out/r2_loop_tests.log
timeout -k 10 300 python tools/gpu_syncpoints.py > gpurun_out/r2_syncpoints_c.json 2>gpurun_out/r2_syncpoints_c.err; cat gpurun_out/r2_syncpoints_c.json
timeout -k 10 200 python bench.py --steps 5 --warmup 1 --cpu-frames 0 > gpurun_out/r2_bench4.log 2>&1; python - <<PY
import json
for line in open('gpurun_out/r2_bench4.log'):
    if line.startswith('{"metric"'):
        d=json.loads(line); print(round(d['value']/1e9,2), round(d['ms_per_step'],2), d['config']['sync_outer_iters'], {k:(v['launches'],round(v['total_ms']/d['steps'],2)) for k,v in d['kernels'].items()}, d['result']['sync_delay'])
PY
