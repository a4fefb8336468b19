#!/usr/bin/env python3
"""Benchmark of the rs-sync PreSync/Sync hot path on MI355X.

Metric (BASELINE.json): ray-residuals/sec (PreSync sweep + Sync iter), 4096
frames x 2048 tracks.  One "step" = one pass of the hot path over the window:
``PreSync(0, begin, end, 0.5 ms, 200 ms)`` (800 candidate delays, BASELINE
configs[1] sweep parameters) followed by ``Sync(d_presync, begin, end-1, 0, 0.2)``
capped at 20 outer iterations (configs[2]), on 4096 frames x 2048 tracks per GPU
(weak scaling: every rank owns 4096 frames, the window is world*4096 frames).

Nominal work per step: frames*tracks*candidates for PreSync +
frames*tracks per Sync outer iteration.  value = nominal ray-residuals of all
ranks / wall time (max over ranks); inputs are resident in HBM before timing.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PER_RR = 32      # 8 fp32 per ray pair read per (frame, delay) evaluation
FLOP_PER_RR_PRESYNC = 390       # ~230 flop per residual row + 20 hypotheses x ~8
FP32_VECTOR_PEAK_TF = 157.3     # MI355X_MICROARCH.md: peak FP32 (vector)
PMC_SUMMARY = "r2_pmc_summary.json"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU")
    ap.add_argument("--tracks", type=int, default=2048)
    ap.add_argument("--search-step", type=float, default=0.0005)
    ap.add_argument("--search-radius", type=float, default=0.2)
    ap.add_argument("--outer-iters", type=int, default=20)
    ap.add_argument("--cpu-frames", type=int, default=192, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--exchange", default="torch", choices=["torch", "native"],
                    help="multi-GPU sum: torch.distributed all_reduce through a Python hook (default), or the "
                         "library's own RCCL communicator (rssync_ext_rccl_init; nccl backend only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse "
                                                      "several ranks on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    dev = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    torch.zeros(1, device="cuda")  # make this process's HIP context current on its GPU before the library opens it
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)

    import rssync_amd
    from rssync_amd import synth

    F, N = args.frames, args.tracks
    f_begin, f_end = rank * F, (rank + 1) * F
    total_frames = world * F
    # one gyro track for the whole window, identical on every rank
    gyro = synth.make_gyro(0.0, (total_frames + 2) / synth.FPS, seed=0x5EED0003)
    prob = rssync_amd.SyncProblem(seed=0x5EED0003, max_outer_iters=args.outer_iters, verbose=False)
    # host side of the boundary, outside the timed region: generate the frames, then hand them over with the
    # reference's calls (SetTrackResult copies into pinned staging and starts the upload)
    t_gen = time.time()
    frames_in = list(synth.make_frames(gyro, f_begin, f_end, N, seed=0x5EED0003))
    t_gen = time.time() - t_gen
    t_set = time.time()
    prob.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr in frames_in:
        prob.SetTrackResult(*fr)
    t_set = time.time() - t_set
    del frames_in

    if world > 1:
        from rssync_amd.dist import make_reduce_hook, use_native_rccl
        # the only exchange of the path: a sum of a few doubles, as an RCCL all-reduce over xGMI
        if args.exchange == "native" and args.backend == "nccl":
            use_native_rccl(prob)
        else:
            prob.set_reduce_hook(make_reduce_hook("cuda" if args.backend == "nccl" else "cpu"))

    t_up = time.time()
    prob.upload()  # rays + spline into HBM before the timed region
    torch.cuda.synchronize()
    t_up = time.time() - t_up

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_cand = None
    iters_done = []
    result = {}

    def step():
        nonlocal n_cand
        c0, d0 = prob.PreSync(0.0, 0, total_frames, args.search_step, args.search_radius)
        c1, d1 = prob.Sync(d0, 0, total_frames - 1, 0.0, args.search_radius)
        iters_done.append(len(prob.sync_trace()))
        result.update(presync_delay=d0, presync_cost=c0, sync_delay=d1, sync_cost=c1)

    for _ in range(args.warmup):
        step()
    iters_done.clear()
    prob.profile(True)
    prob.profile_reset()
    barrier()
    t0 = time.perf_counter()
    t_pre = 0.0
    for _ in range(args.steps):
        ta = time.perf_counter()
        c0, d0 = prob.PreSync(0.0, 0, total_frames, args.search_step, args.search_radius)
        t_pre += time.perf_counter() - ta
        c1, d1 = prob.Sync(d0, 0, total_frames - 1, 0.0, args.search_radius)
        iters_done.append(len(prob.sync_trace()))
        result.update(presync_delay=d0, presync_cost=c0, sync_delay=d1, sync_cost=c1)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = prob.profile_get()
    prob.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # candidates exactly as the reference's double loop produces them (core_private.cpp:69-70)
    n_cand = 0
    d = 0.0 - args.search_radius
    while d < 0.0 + args.search_radius:
        n_cand += 1
        d += args.search_step
    rr_presync = total_frames * N * n_cand
    rr_sync = total_frames * N * (sum(iters_done) / max(len(iters_done), 1))
    rr_step = rr_presync + rr_sync
    value = rr_step * args.steps / elapsed

    out = None
    if rank == 0:
        # roofline of the dominant kernel (the PreSync LMedS tile kernel), from HIP events on the
        # stream it runs on: algorithmic bytes = 32 B x ray-residuals one launch processes (this
        # rank's frames x tracks x candidates) / average launch duration
        n_l, ms_l = prof["lmeds"]
        roof = None
        if n_l:
            avg_ms = ms_l / n_l
            alg_bytes = F * N * n_cand * BYTES_PER_RR
            ach = alg_bytes / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "lmeds_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "avg_launch_ms": round(avg_ms, 4), "launches": n_l,
                    "note": "equivalent bandwidth: 32 B per nominal ray-residual; the kernel reuses each ray "
                            "across the candidates of a chunk and is VALU/LDS-bound (DESIGN.md)"}
        # HBM traffic of that kernel: NOT measured by this run (counters need their own rocprofv3 --pmc passes);
        # read from the PMC summary committed under profiles/ (tools/collect_pmc.sh: separate passes for
        # FETCH_SIZE / WRITE_SIZE, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), which is
        # only valid for the default workload
        roof_flop = None
        if roof:
            # nominal arithmetic of the same launches: ~390 flop per PreSync ray-residual (230
            # for the residual row + 20 hypotheses x 8) against the fp32 vector peak, from this run's HIP events
            flops = F * N * n_cand * FLOP_PER_RR_PRESYNC
            ach_tf = flops / (roof["avg_launch_ms"] * 1e-3) / 1e12
            roof_flop = {"bound": "fp32-vector", "kernel": "lmeds_kernel", "achieved": round(ach_tf, 2),
                         "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach_tf / FP32_VECTOR_PEAK_TF, 4),
                         "note": "nominal 390 flop per ray-residual x ray-residuals of one launch / live launch time; "
                                 "what limits the kernel: DESIGN.md section 3 and profiles/r2_valu_rate.txt"}
        if roof and (F, N, n_cand) == (4096, 2048, 800):
            pmc_path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
            if os.path.exists(pmc_path):
                raw = json.load(open(pmc_path))
                key = [k for k in raw if k.startswith("lmeds_kernel<8, 0")]
                if key:
                    ctr = raw[key[0]]
                    roof["traffic"] = round((2 * ctr["FETCH_SIZE"]["mean_per_launch_KiB"] +
                                             ctr["WRITE_SIZE"]["mean_per_launch_KiB"]) * 1024 / 1e9, 4)
                    roof["traffic_unit"] = "GB per launch, from profiles/%s (a separate rocprofv3 --pmc run, not this one)" % PMC_SUMMARY
        kernels = {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in prof.items()}
        cpu = None
        if world == 1 and args.cpu_frames > 0:
            cpu = cpu_baseline(gyro, min(args.cpu_frames, F), N, args)
        out = {
            "metric": "ray-residuals/sec (PreSync sweep + Sync iter), 4096 frames x 2048 tracks",
            "value": value, "unit": "ray-residuals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "PreSync sweep in fp32 (92 % of the nominal work), Sync in fp64 (the reference's arithmetic)",
            "config": {"workload": "PreSync(radius 200 ms, step 0.5 ms) + Sync(<=20 outer iters)",
                       "frames_per_gpu": F, "tracks": N, "candidates": n_cand,
                       "sync_outer_iters": iters_done, "gyro_hz": gyro.fs, "parallelism": "frames sharded x%d" % world},
            "roofline": roof, "roofline_flop": roof_flop, "cpu_baseline": cpu, "kernels": kernels,
            "presync_ms_per_step": t_pre / args.steps * 1e3,
            "result": result, "host": {"gen_s": round(t_gen, 2), "set_track_result_s": round(t_set, 3), "pack_upload_s": round(t_up, 3),
                     "note": "set_track_result_s = the SetTrackResult loop over all frames (checks + copy into pinned "
                             "staging, upload started); pack_upload_s = waiting for that upload + packing kernel"},
        }
        if cpu:
            # reported for context only: the roofline fraction, not this ratio, says how good the kernels are
            out["gpu_over_cpu"] = round(value / world / cpu["value"], 1)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def host_cores():
    """CPU cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(gyro, frames, tracks, args):
    """The oracle (a port of the reference's CPU path, faithful evaluation schedule) timed on this
    host's cores on a bounded sample of the same workload: the first `frames` frames, all candidates."""
    from oracle.oracle import OracleProblem
    from rssync_amd import synth

    cores = host_cores()
    o = OracleProblem(seed=0x5EED0003, max_outer_iters=args.outer_iters, threads=cores, faithful=True)
    synth.fill(o, gyro, 0, frames, tracks, seed=0x5EED0003)
    t0 = time.perf_counter()
    delays, costs = o.presync_curve(0.0, 0, frames, args.search_step, args.search_radius)
    t_pre = time.perf_counter() - t0
    d0 = float(delays[int(np.argmin(costs))])
    t1 = time.perf_counter()
    c1, d1, tr = o.sync_trace(d0, 0, frames - 1, 0.0, args.search_radius)
    t_sync = time.perf_counter() - t1
    rr = frames * tracks * (len(delays) + len(tr))
    return {"value": rr / (t_pre + t_sync), "unit": "ray-residuals/s", "cores": cores, "kind": "port",
            "sample": "%d frames x %d tracks, %d candidates + %d Sync outer iterations (same inputs, "
                      "first frames of the window)" % (frames, tracks, len(delays), len(tr)),
            "presync_s": round(t_pre, 2), "sync_s": round(t_sync, 2), "presync_delay": d0, "sync_delay": d1}


if __name__ == "__main__":
    main()
