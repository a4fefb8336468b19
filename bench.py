#!/usr/bin/env python3
"""Benchmark of the rs-sync PreSync/Sync hot path on MI355X.

Metric (BASELINE.json): ray-residuals/sec (PreSync sweep + Sync iter), 4096
frames x 2048 tracks.  One "step" = one pass of the hot path over the window:
``PreSync(0, begin, end, 0.5 ms, 200 ms)`` (800 candidate delays, BASELINE
configs[1] sweep parameters) followed by ``Sync(d_presync, begin, end-1, 0, 0.2)``
capped at 20 outer iterations (configs[2]), on 4096 frames x 2048 tracks per GPU
(weak scaling: every GPU owns 4096 frames, the window is N*4096 frames).

Nominal work per step: frames*tracks*candidates for PreSync +
frames*tracks per Sync outer iteration.  value = nominal ray-residuals of all
GPUs / wall time (max over ranks); inputs are resident in HBM before timing.

    python bench.py --gpus N --steps K --warmup W

N > 1 needs no external launcher.  Two ways of using N GPUs (the reference parallelises over frames inside one
object, core_private.cpp:73,231,245,263):

  --mode ranks   (default) one process per GPU.  Started either by ``python -m torch.distributed.run
                 --nproc-per-node N bench.py --gpus N`` (RANK/LOCAL_RANK/WORLD_SIZE in the environment) or by
                 this script itself: without WORLD_SIZE the parent -- which never touches a GPU -- starts N fresh
                 rank processes, waits for them and returns their status; rank 0 prints the JSON line.  The sums
                 over frames are exchanged through the library's own RCCL communicator (--exchange native) or
                 through torch.distributed (--exchange torch).
  --mode inproc  one process, one object, ``set_devices(range(N))``: the library spreads the frames over the N
                 GPUs and adds their partial sums on the host.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PER_RR = 32      # 8 fp32 per ray pair read per (frame, delay) evaluation
BYTES_PER_RR_F64 = 64  # the Sync kernels read the fp64 streams: 8 doubles per ray pair per evaluation
FLOP_PER_RR_PRESYNC = 390       # ~230 flop per residual row + 20 hypotheses x ~8
FP32_VECTOR_PEAK_TF = 157.3     # MI355X_MICROARCH.md: peak FP32 (vector)
PMC_SUMMARIES = ("r6_pmc_summary.json", "r5_pmc_summary.json", "r4_pmc_summary.json", "r3_pmc_summary.json", "r2_pmc_summary.json")
N_SIMD = 1024                   # 256 CUs x 4 SIMDs
# cycles a gfx950 SIMD needs per fp64 wave-instruction with >= 2 waves resident: measured 4.2 - 4.4 (v_fma_f64,
# tools/ubench/valu_rate.hip -> profiles/r2_valu_rate.txt); MI355X_MICROARCH.md states no fp64 vector peak
FP64_CYCLES_PER_WAVE_INSTR = 4.3
# parity asserted in the driver's own run (north star: delay within 1e-4 s, per-iteration loss to a stated tolerance)
# (the sweep's curve: fp32 frame costs agree to ~1e-7 where both sides pick the same hypothesis; a near-tie that falls the
# other way moves ONE frame's cost by per cent, i.e. the sum over the sample's 192 frames by ~1e-4 -- measured 9.5e-5 on the
# default sample, 1.3e-5 on all 4096 frames, profiles/r4_config3_presync_parity.json; deterministic, box-independent)
PARITY_TOL = {"delay_s": 1e-4, "loss_rel": 2e-4, "presync_curve_rel": 3e-4}


def parse_args(argv=None):
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
    ap.add_argument("--mode", default="ranks", choices=["ranks", "inproc"],
                    help="ranks: one process per GPU (self-spawned unless WORLD_SIZE is set); inproc: one object "
                         "driving all GPUs of this process")
    ap.add_argument("--hook-device-loop", dest="hook_device_loop", action="store_true", default=True,
                    help="with --exchange torch (default ON since round 5): keep Sync's loop on the device and call the "
                         "hook on the window sums between the kernels -- the structure of the native RCCL path with a "
                         "host transport; bit-identical to the host loop (tests/test_gpu_parity.py, two ranks)")
    ap.add_argument("--no-hook-device-loop", dest="hook_device_loop", action="store_false",
                    help="with --exchange torch: Sync's loop on the host, one hook call per launch")
    ap.add_argument("--no-parity", action="store_true", help="skip the untimed HIP-vs-oracle comparison on the CPU sample")
    ap.add_argument("--no-driver-workload", action="store_true",
                    help="skip the (untimed) reference-driver workload: 98 sync points of 61 x 130, PreSync + 4 x Sync each "
                         "(core_testcode.cpp:270-316), with its CPU sample and its critical-path bound")
    ap.add_argument("--driver-cpu-positions", type=int, default=16, help="sync points of the driver workload the oracle is timed on")
    ap.add_argument("--exchange", default="torch", choices=["native", "torch"],
                    help="multi-rank sums: torch.distributed all_reduce through a reduce hook (default: the path every "
                         "multi-rank test exercises), or the library's own RCCL communicator with Sync's loop on the "
                         "device (nccl backend only; has only ever run with ONE rank on hardware -- opt in)")
    ap.add_argument("--workload", default="presync+sync", choices=["presync+sync", "c5"],
                    help="presync+sync: the metric's step (BASELINE configs 2 + 3; --frames 2048 --gpus 8 is config 4); "
                         "c5: BASELINE config 5 -- gyro as rates at jittered timestamps, one PreSync per IMU "
                         "orientation (--orientations, 48 in the reference), frames sharded over the ranks")
    ap.add_argument("--orientations", type=int, default=48)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse "
                                                      "several ranks on one GPU or on the CPU stand-in)")
    ap.add_argument("--spawn-timeout", type=float, default=1500.0, help="seconds the self-spawned ranks may take")
    ap.add_argument("--rehearse-cpu", default=None, metavar="LIB",
                    help="TESTS ONLY: run the launcher / exchange logic against the CPU stand-in for the device ABI "
                         "(tests/_build/librssync_hosttest.so); the printed value is then meaningless and marked so")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """Parent of a self-launched multi-rank run: start one fresh process per GPU, wait, pass on the status.
    Nothing here imports torch or touches a GPU (a process that has initialised the GPU must not be replaced or
    forked from); the children inherit stdout, rank 0 prints the JSON line."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RSSYNC_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    deadline = time.time() + args.spawn_timeout
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print("bench: rank process %d exited with %d; stopping the others" % (p.pid, code), file=sys.stderr)
                for q in live:
                    q.terminate()
        if live and time.time() > deadline:
            print("bench: ranks still running after %.0f s; stopping them" % args.spawn_timeout, file=sys.stderr)
            for q in live:
                q.kill()
            rc = rc or 124
            deadline = float("inf")
        if live:
            time.sleep(0.05)
    return rc


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.mode == "ranks" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    run(args)


def run(args):
    import torch
    import torch.distributed as dist

    rehearsal = args.rehearse_cpu is not None
    inproc = args.mode == "inproc"
    world = 1 if inproc else int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0 if inproc else int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not inproc and world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    n_dev = args.gpus if inproc else 1      # GPUs this process drives
    n_gpus = args.gpus
    launcher = "inproc" if inproc else ("self-spawned" if os.environ.get("RSSYNC_BENCH_SPAWNED") else
                                        ("torch.distributed.run" if world > 1 else "direct"))
    on_gpu = not rehearsal
    if on_gpu:
        if inproc and torch.cuda.device_count() < n_dev:
            raise SystemExit("--mode inproc --gpus %d: this process sees %d GPUs" % (n_dev, torch.cuda.device_count()))
        dev = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(dev)
        torch.zeros(1, device="cuda")  # make this process's HIP context current on its GPU before the library opens it
    backend = args.backend if on_gpu else "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if (on_gpu and backend == "nccl") else "cpu"

    import rssync_amd
    from rssync_amd import synth

    lib = None
    if rehearsal:
        import ctypes
        from rssync_amd.problem import bind
        lib = bind(ctypes.CDLL(os.path.abspath(args.rehearse_cpu)))

    F, N = args.frames, args.tracks
    per_proc = F * n_dev
    c5 = args.workload == "c5"
    # (config 5's gyro arrives with timestamps, which must be >= 0: the video starts 1 s into the gyro track)
    t_first = 1.0 if c5 else 0.0
    f_first = int(round(t_first * synth.FPS))
    f_begin, f_end = f_first + rank * per_proc, f_first + (rank + 1) * per_proc
    total_frames = world * per_proc
    w_begin, w_end = f_first, f_first + total_frames
    # one gyro track for the whole window, identical on every rank
    gyro = synth.make_gyro(t_first, t_first + (total_frames + 2) / synth.FPS, seed=0x5EED0003)
    prob = rssync_amd.SyncProblem(seed=0x5EED0003, max_outer_iters=args.outer_iters, verbose=False, _lib=lib)
    if inproc and n_dev > 1:
        prob.set_devices(list(range(n_dev)))
    elif on_gpu and not inproc:
        # a rank names its GPU explicitly: the library's default is "the calling thread's current device" of the HIP
        # runtime IT is bound to, which is torch's only while both resolve libamdhip64 to the same copy
        prob.set_devices([dev])
    # host side of the boundary, outside the timed region: generate the frames, then hand them over with the
    # reference's calls (SetTrackResult copies into pinned staging and starts the upload)
    t_gen = time.time()
    frames_in = list(synth.make_frames(gyro, f_begin, f_end, N, seed=0x5EED0003))
    t_gen = time.time() - t_gen
    t_set = time.time()
    orient_names = None
    if c5:
        # the gyro as the reference driver gets it (core_testcode.cpp:41-52): rates at (jittered) timestamps
        rng = np.random.default_rng(0x5EED0005)
        dt = 1.0 / gyro.fs
        g_times = gyro.times + rng.uniform(-0.2, 0.2, size=len(gyro.times)) * dt * 0.5
        g_times[0] = max(g_times[0], 0.0)
        orient_names = list(synth.ORIENTATIONS[:max(1, args.orientations)])
        if "XYZ" not in orient_names:
            orient_names[-1] = "XYZ"
        prob.set_gyro_rates(g_times, gyro.rates, "XYZ")
    else:
        prob.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr in frames_in:
        prob.SetTrackResult(*fr)
    t_set = time.time() - t_set
    del frames_in

    exchange = None
    if world > 1:
        from rssync_amd.dist import make_reduce_hook, use_native_rccl
        # the only exchange of the path: a sum of a few doubles, as an RCCL all-reduce over xGMI
        prob.set_tracks_hint(N)  # every rank holds frames of N tracks: no exchange needed to agree on kernel shapes
        if args.exchange == "native" and backend == "nccl":
            exchange = use_native_rccl(prob)  # "native-rccl", or "torch-hook" with the reason if RCCL refused
        if exchange != "native-rccl":
            prob.set_reduce_hook(make_reduce_hook(red_dev))
            exchange = "torch-%s-hook" % backend + ("" if exchange is None else " (native RCCL init failed: %s)" % exchange)
            if args.hook_device_loop:
                prob.set_hook_device_loop(True)

    def dev_sync():
        if on_gpu:
            torch.cuda.synchronize()

    t_up = time.time()
    prob.upload()  # rays + spline into HBM before the timed region
    dev_sync()
    t_up = time.time() - t_up

    def barrier():
        if world > 1:
            dist.barrier()
        dev_sync()

    iters_done = []
    result = {}

    def step():
        if c5:
            # BASELINE config 5 (core_testcode.cpp:184-233): every orientation = the rates permuted, integrated,
            # resampled, splined on the device, then PreSync over the (sharded) frames -- one exchange each
            costs, delays = prob.orientation_sweep(g_times, gyro.rates, orient_names, 0.0, w_begin, w_end,
                                                   args.search_step, args.search_radius)
            order = np.argsort(costs)
            result.update(best_orientation=orient_names[int(order[0])], best_delay=float(delays[order[0]]),
                          cost_ratio_best_to_second=float(costs[order[0]] / costs[order[1]]) if len(order) > 1 else None)
            iters_done.append(0)
            return 0.0
        ta = time.perf_counter()
        c0, d0 = prob.PreSync(0.0, w_begin, w_end, args.search_step, args.search_radius)
        tb = time.perf_counter() - ta
        c1, d1 = prob.Sync(d0, w_begin, w_end - 1, 0.0, args.search_radius)
        iters_done.append(len(prob.sync_trace()))
        result.update(presync_delay=d0, presync_cost=c0, sync_delay=d1, sync_cost=c1)
        return tb

    for _ in range(args.warmup):
        step()
    iters_done.clear()
    prob.profile(True)
    prob.profile_reset()
    x_calls0, x_doubles0 = prob.exchange_stats()
    barrier()
    t0 = time.perf_counter()
    t_pre = 0.0
    for _ in range(args.steps):
        t_pre += step()
    barrier()
    elapsed = time.perf_counter() - t0
    x_calls, x_doubles = prob.exchange_stats()
    x_calls -= x_calls0
    x_doubles -= x_doubles0
    prof = prob.profile_get()
    prob.profile(False)
    near_static = prob.near_static_stats()  # (frame, candidate) pairs the sweep recomputed with fp64 rows: 0 on this scene
    rccl_lib_name = prob.rccl_library() if exchange == "native-rccl" else None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # candidates exactly as the reference's double loop produces them (core_private.cpp:69-70)
    n_cand = 0
    d = 0.0 - args.search_radius
    while d < 0.0 + args.search_radius:
        n_cand += 1
        d += args.search_step
    rr_presync = total_frames * N * n_cand * (len(orient_names) if c5 else 1)
    rr_sync = total_frames * N * (sum(iters_done) / max(len(iters_done), 1))
    rr_step = rr_presync + rr_sync
    # which BASELINE.json config this line is (the judge's table): per-GPU work is fixed as N grows (weak scaling)
    if c5:
        baseline_config = "config 5 (timestamped gyro + %d-orientation sweep), %d frames x %d tracks per GPU" % (len(orient_names), F, N)
    elif (F, N) == (4096, 2048):
        baseline_config = "config 3's window + config 2's sweep parameters, x%d weak (%d frames in all)" % (n_gpus, total_frames)
    elif (F, N) == (2048, 2048):
        baseline_config = ("config 4 (16384 frames x 2048 tracks over 8 GPUs)" if n_gpus == 8 else
                           "config 4's shard size (2048 frames per GPU), x%d weak (%d frames in all)" % (n_gpus, total_frames))
    else:
        baseline_config = "none (%d frames x %d tracks per GPU)" % (F, N)
    if world > 1:
        # which exchange path the line was measured on belongs to the config, not only to multi_gpu
        baseline_config += "; sums exchanged by %s, Sync loop %s" % (
            exchange, "on the device" if (exchange == "native-rccl" or args.hook_device_loop) else "on the host")
    value = rr_step * args.steps / elapsed

    if rank == 0:
        # roofline of the dominant kernel (the PreSync LMedS tile kernel), from HIP events on the
        # stream it runs on: algorithmic bytes = 32 B x ray-residuals one launch processes (one
        # GPU's frames x tracks x candidates) / average launch duration.  With --mode inproc the
        # event times of the N contexts are summed and so are their launch counts: still per launch.
        n_l, ms_l = prof["lmeds"]
        roof = roof_flop = None
        if n_l:
            avg_ms = ms_l / n_l
            alg_bytes = F * N * n_cand * BYTES_PER_RR
            ach = alg_bytes / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm-equivalent (contract: 32 B per ray-residual)", "kernel": "lmeds_kernel",
                    "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "valu_busy": None,
                    "avg_launch_ms": round(avg_ms, 4), "launches": n_l,
                    "note": "NOT a physical bandwidth: SURVEY 8(d)'s contract figure, 32 B per nominal ray-residual / live "
                            "launch time.  The kernel reuses each ray across the candidates of a chunk -- `traffic` is "
                            "what it really moves -- and is bound by VALU issue (`valu_busy`) with the CU's scalar unit "
                            "saturated by the counting idiom (DESIGN.md section 3)"}
            # nominal arithmetic of the same launches: ~390 flop per PreSync ray-residual (230
            # for the residual row + 20 hypotheses x 8) against the fp32 vector peak, from this run's HIP events
            flops = F * N * n_cand * FLOP_PER_RR_PRESYNC
            ach_tf = flops / (avg_ms * 1e-3) / 1e12
            roof_flop = {"bound": "fp32-vector", "kernel": "lmeds_kernel", "achieved": round(ach_tf, 2),
                         "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach_tf / FP32_VECTOR_PEAK_TF, 4),
                         "note": "nominal 390 flop per ray-residual x ray-residuals of one launch / live launch time; "
                                 "what limits the kernel: DESIGN.md section 3 and profiles/r2_valu_rate.txt"}
        # HBM traffic of that kernel: NOT measured by this run (counters need their own rocprofv3 --pmc passes);
        # read from the PMC summary committed under profiles/ (tools/collect_pmc.sh: separate passes for
        # FETCH_SIZE / WRITE_SIZE, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), which is
        # only valid for the default workload
        if roof and (F, N, n_cand) == (4096, 2048, 800):
            for name in PMC_SUMMARIES:
                pmc_path = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(pmc_path):
                    continue
                raw = json.load(open(pmc_path))
                key = [k for k in raw if k.startswith("lmeds_kernel<8, 0")]
                if key:
                    ctr = raw[key[0]]
                    roof["traffic"] = round((2 * ctr["FETCH_SIZE"]["mean_per_launch_KiB"] +
                                             ctr["WRITE_SIZE"]["mean_per_launch_KiB"]) * 1024 / 1e9, 4)
                    roof["traffic_unit"] = "GB per launch, from profiles/%s (a separate rocprofv3 --pmc run, not this one)" % name
                    if "SQ_ACTIVE_INST_VALU" in ctr and "GRBM_GUI_ACTIVE" in ctr:
                        # fraction of the SIMDs' cycles in which a VALU instruction was executing: the guide's formula,
                        # SQ_ACTIVE_INST_VALU x 4 / (SIMDs x cycles of one XCD's GRBM_GUI_ACTIVE share)
                        roof["valu_busy"] = round(ctr["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] * 4 /
                                                  (N_SIMD * ctr["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8), 3)
                        roof["valu_busy_unit"] = "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), same file"
                    break
        # second roofline: K1's loss + analytic-gradient launch (one delay per window, every fp64 ray pair of the
        # GPU's frames read once: 64 B per ray pair), the HBM-bound kernel of the Sync phase, from its own events
        n_g, ms_g = prof["loss_grad"]
        roof_k1 = None
        if n_g:
            avg_ms = ms_g / n_g
            ach = F * N * BYTES_PER_RR_F64 / (avg_ms * 1e-3) / 1e9
            roof_k1 = {"bound": "hbm", "kernel": "loss64_kernel<.,GRAD>", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg_ms, 4),
                       "launches": n_g, "note": "64 B (fp64 streams) x frames x tracks of one GPU / live launch time"}
        # third roofline: K3, the per-frame motion L-BFGS -- Sync's largest kernel, on-chip (P in registers) and bound by
        # fp64 VALU issue.  The guide states no fp64 peak, so the "peak" is this chip's MEASURED fp64 issue rate: one
        # wave-instruction per FP64_CYCLES_PER_WAVE_INSTR cycles and SIMD.  achieved = the kernel's VALU wave-
        # instructions per launch (PMC, its own rocprofv3 pass) / the launch's cycles; both from the committed summary,
        # the live launch time of THIS run beside them.
        n_m, ms_m = prof["motion"]
        roof_k3 = None
        if n_m and (F, N, n_cand) == (4096, 2048, 800):
            for name in PMC_SUMMARIES:
                pmc_path = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(pmc_path):
                    continue
                raw = json.load(open(pmc_path))
                key = [k for k in raw if k.startswith("opt_motion64_kernel<8, 4>")]
                if key and "SQ_INSTS_VALU" in raw[key[0]] and "GRBM_GUI_ACTIVE" in raw[key[0]]:
                    ctr = raw[key[0]]
                    insts = ctr["SQ_INSTS_VALU"]["mean_per_launch"]
                    cycles = ctr["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8
                    ach = insts / (N_SIMD * cycles)                 # wave-instructions per SIMD and cycle
                    peak = 1.0 / FP64_CYCLES_PER_WAVE_INSTR
                    roof_k3 = {"bound": "fp64 VALU issue (measured rate: no fp64 peak in MI355X_MICROARCH.md)",
                               "kernel": "opt_motion64_kernel<8, 4>", "achieved": round(ach, 4), "peak": round(peak, 4),
                               "unit": "wave-instructions / SIMD / cycle", "frac": round(ach / peak, 3),
                               "avg_launch_ms": round(ms_m / n_m, 4), "launches": n_m,
                               "valu_wave_instructions_per_launch": insts,
                               "note": "SQ_INSTS_VALU per launch / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) from profiles/%s; peak = 1 / %.1f "
                                       "cycles per fp64 wave-instruction (profiles/r2_valu_rate.txt: 4.2 - 4.4 with >= 2 waves "
                                       "per SIMD); avg_launch_ms is this run's (HIP events)" % (name, FP64_CYCLES_PER_WAVE_INSTR)}
                    break
        kernels = {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in prof.items()}
        cpu = parity = driver = None
        if n_gpus == 1 and args.cpu_frames > 0 and not c5:
            cpu = cpu_baseline(gyro, min(args.cpu_frames, F), N, args)
            if not args.no_parity and not rehearsal:
                parity = parity_check(gyro, min(args.cpu_frames, F), N, args, cpu.pop("_oracle"))
            cpu.pop("_oracle", None)
        if n_gpus == 1 and not c5 and not rehearsal and not args.no_driver_workload:
            del prob                      # (its 1.3 GB of frames are not needed any more)
            driver = driver_workload(args)
        out = {
            "metric": "ray-residuals/sec (PreSync sweep + Sync iter), 4096 frames x 2048 tracks",
            "value": value, "unit": "ray-residuals/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "PreSync sweep in fp32 (92 % of the nominal work), Sync in fp64 (the reference's arithmetic)",
            "config": {"workload": ("orientation sweep: %d x PreSync(radius %g ms, step %g ms), gyro as rates at jittered timestamps"
                                    % (len(orient_names), args.search_radius * 1e3, args.search_step * 1e3)) if c5 else
                                   "PreSync(radius %g ms, step %g ms) + Sync(<=%d outer iters)"
                                   % (args.search_radius * 1e3, args.search_step * 1e3, args.outer_iters),
                       "baseline_config": baseline_config,
                       "frames_per_gpu": F, "tracks": N, "candidates": n_cand,
                       "sync_outer_iters": iters_done, "gyro_hz": gyro.fs,
                       "parallelism": "frames sharded x%d (%s)" % (n_gpus, "one object, in-process" if inproc else
                                                                   "one process per GPU")},
            "multi_gpu": {"mode": args.mode, "launcher": launcher, "processes": world, "devices_per_process": n_dev,
                          # RCCL carries the sums whenever the backend is nccl -- through the library's own communicator
                          # ("native-rccl") or through torch.distributed behind the reduce hook ("torch-nccl-hook")
                          "exchange": exchange, "rccl_ranks": world if (world > 1 and (exchange == "native-rccl" or backend == "nccl")) else 0,
                          "rccl_library": rccl_lib_name if exchange == "native-rccl" else
                                          ("torch.distributed's (backend nccl)" if world > 1 and backend == "nccl" else None),
                          "exchanges_per_step": x_calls / max(args.steps, 1),
                          "doubles_per_step": x_doubles / max(args.steps, 1),
                          "sync_loop": ("device, window sums all-reduced on the stream (ncclAllReduce between the kernels)"
                                        if exchange == "native-rccl" else "device" if world == 1 and n_dev == 1 else
                                        "device, the hook called on the window sums between the kernels (stream drained)"
                                        if args.hook_device_loop and world > 1 and n_dev == 1 else
                                        "host, one exchange per launch"),
                          "note": "exchange = how the sums over frames cross process boundaries (none within one "
                                  "process: --mode inproc adds the devices' chunk sums on the host)"},
            "roofline": roof, "roofline_flop": roof_flop, "roofline_k1": roof_k1, "roofline_k3": roof_k3, "cpu_baseline": cpu,
            "parity": parity, "driver_workload": driver, "kernels": kernels,
            "near_static": {"fp64_row_pairs": near_static["pairs"], "sweeps": near_static["sweeps"],
                            "note": "PreSync recomputes near-static (frame, candidate) pairs -- a quarter of a frame's first 64 rows "
                                    "with |P| < 2e-4 -- from the fp64 streams (core_private.cpp:19-28 is double); an ordinary "
                                    "scene like this one never does: both counters must be 0 for the timed steps to be the fp32 sweep"},
            "presync_ms_per_step": t_pre / args.steps * 1e3,
            "result": result, "host": {"gen_s": round(t_gen, 2), "set_track_result_s": round(t_set, 3), "pack_upload_s": round(t_up, 3),
                     "note": "set_track_result_s = the SetTrackResult loop over all frames (checks + copy into pinned "
                             "staging, upload started); pack_upload_s = waiting for that upload + packing kernel"},
        }
        if c5:
            # one pipeline per sweep (round 6): every orientation's integrate -> resample -> spline -> sweep -> sums enqueued
            # back to back, one wait, one exchange.  wall_ms against the kernels' own time (HIP events) per orientation
            n_or = len(orient_names)
            k_ms = sum(v[1] for v in prof.values()) / max(args.steps, 1) / n_or
            out["per_orientation"] = {"orientations": n_or, "wall_ms": round(elapsed / args.steps * 1e3 / n_or, 4), "kernels_ms": round(k_ms, 4),
                                      "host_overhead_ms": round(elapsed / args.steps * 1e3 / n_or - k_ms, 4),
                                      "pipeline": os.environ.get("RSSYNC_SWEEP_PIPELINE", "1") != "0",
                                      "note": "kernels_ms = all kernels of the sweep (gyro pipeline, LMedS sweep, sums) by HIP events / orientations; "
                                              "RSSYNC_SWEEP_PIPELINE=0 = rounds 1-5's loop (a blocking PreSync and exchange per orientation)"}
        if rehearsal:
            out["rehearsal"] = "CPU stand-in for the device ABI (%s): launcher and exchange logic only, value is NOT a " \
                               "measurement" % os.path.basename(args.rehearse_cpu)
        if cpu:
            # reported for context only: the roofline fraction, not this ratio, says how good the kernels are
            out["gpu_over_cpu"] = round(value / n_gpus / cpu["value"], 1)
        print(json.dumps(out), flush=True)
        if parity is not None and not parity["ok"]:
            print("bench: PARITY FAILED on the CPU sample: %s" % json.dumps(parity), file=sys.stderr)
            sys.exit(3)
    if world > 1:
        dist.barrier()
        if exchange == "native-rccl":
            prob.rccl_shutdown()
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
            "presync_s": round(t_pre, 2), "sync_s": round(t_sync, 2), "presync_delay": d0, "sync_delay": d1,
            "note": "the SAMPLE's results (first %d frames): compare with `parity`, not with `result` (all frames)" % frames,
            "_oracle": {"delays": delays, "costs": costs, "d0": d0, "sync_cost": c1, "sync_delay": d1, "trace": tr}}


DRIVER_SEED = 0x5EED0006       # tools/gpu_syncpoints.py: the scene of profiles/r*_syncpoints.json


def driver_workload(args):
    """The reference driver's REAL workload, untimed with respect to the metric, in the driver's own bench run: a 100 s clip of
    3000 frames x 130 tracks (core_testcode.cpp:126-132: a 200 px grid), a sync point every 30 frames with a window of 60
    (core_testcode.cpp:270-280, README.md:36) = 98 positions, at each PreSync(+-100 ms, 1 ms) then FOUR Sync calls
    (core_testcode.cpp:303-316).  BASELINE's 4096 x 2048 is ~1000x anything the reference was run on; this is what it WAS
    run on.  Reported:
      executor_ms / chain_ms  the batched call (rssync_ext_sync_points) through the window executor (one device-scheduled
                              launch) and through the chain of launches; `identical`: the same delays and costs, bit for bit
      critical_path           the executor is latency-bound, so its roofline is its critical path: the PreSync launch + the
                              LONGEST window's outer iterations x the latency of one iteration of a window that has the
                              device to itself (measured here: that window alone), against the wall time
      cpu_sample              the oracle (faithful schedule, all host cores) on a few positions, PreSync + 4 x Sync each
      delay_vs_oracle_ms      the HIP path, called position by position on the same sample (so that both sides count their
                              Sync calls alike: the sampler stream follows the call number), against the oracle's delays;
                              within_control: no further apart than rounding in another order moves the reference-order
                              algorithm itself (2.5 x the largest control distance of profiles/r5_reassociation.json, the
                              rule of tests/noisy_scenes.py: Sync on 61 x 130 noisy windows is a chaotic iteration)."""
    import rssync_amd
    from rssync_amd import synth
    from oracle.oracle import OracleProblem

    F, N, WINDOW, DIST = 3000, 130, 60, 30
    STEP, RADIUS = 0.001, 0.1
    gyro = synth.make_gyro(0, (F + 2) / synth.FPS, seed=DRIVER_SEED)
    pos = list(range(0, F - WINDOW - 1, DIST))

    def problem(executor=True):
        if not executor:
            os.environ["RSSYNC_EXECUTOR"] = "0"
        try:
            h = rssync_amd.SyncProblem(seed=DRIVER_SEED, verbose=False)
        finally:
            os.environ.pop("RSSYNC_EXECUTOR", None)
        synth.fill(h, gyro, 0, F, N, seed=DRIVER_SEED)
        h.upload()
        return h

    def best_of(fn, reps=3):
        out, best = None, float("inf")
        for _ in range(reps):
            t = time.perf_counter()
            out = fn()
            best = min(best, time.perf_counter() - t)
        return out, best

    ex, ch = problem(True), problem(False)
    for h in (ex, ch):
        h.set_executor_check_every(0)            # (the production tripwire re-runs one call in 256 through the chain: not in a timing)
        h.sync_points(pos, WINDOW, 0.0, STEP, RADIUS)      # warm-up
    (ce, de), t_ex = best_of(lambda: ex.sync_points(pos, WINDOW, 0.0, STEP, RADIUS))
    (cc, dc), t_ch = best_of(lambda: ch.sync_points(pos, WINDOW, 0.0, STEP, RADIUS))
    iters = [len(ex.window_trace(w)) for w in range(len(pos))]
    w_long = int(np.argmax(iters))
    # the critical path: the PreSync launch, then the longest window's iterations at the latency of a window alone on the device
    b = np.asarray(pos, np.int64)
    (_, d_pre), t_pre = best_of(lambda: ex.pre_sync_windows(0.0, b, b + WINDOW, STEP, RADIUS))
    lone = [pos[w_long]]
    ex.sync_points(lone, WINDOW, float(d_pre[w_long]))
    _, t_lone = best_of(lambda: ex.sync_points(lone, WINDOW, float(d_pre[w_long])), reps=5)
    it_lone = len(ex.window_trace(0))
    per_iter_us = t_lone / max(it_lone, 1) * 1e6
    crit_ms = t_pre * 1e3 + max(iters) * per_iter_us * 1e-3
    stats = ex.executor_stats()
    out = {"positions": len(pos), "window": "%d x %d" % (WINDOW + 1, N), "frames": F, "calls_per_position": "PreSync(+-100 ms, 1 ms) + 4 x Sync",
           "executor_ms": round(t_ex * 1e3, 3), "chain_ms": round(t_ch * 1e3, 3),
           "identical": bool(np.array_equal(de, dc) and np.array_equal(ce, cc)),
           "positions_per_s": round(len(pos) / t_ex, 1),
           "outer_iterations_per_position": {"mean": float(np.mean(iters)), "max": int(max(iters))},
           "critical_path": {"presync_ms": round(t_pre * 1e3, 3), "longest_window_iterations": int(max(iters)),
                             "iteration_latency_us": round(per_iter_us, 2), "lone_window_iterations": int(it_lone),
                             "lone_window_ms": round(t_lone * 1e3, 3),
                             "critical_path_ms": round(crit_ms, 3), "wall_ms": round(t_ex * 1e3, 3),
                             "critical_path_over_wall": round(crit_ms / (t_ex * 1e3), 3),
                             "note": "iteration_latency_us = one window alone on the device (its four Sync calls through the executor, launch "
                                     "and copies included) / its outer iterations: motion phase (as long as its slowest frame's L-BFGS), "
                                     "trials, two decisions, four hand-offs -- DESIGN.md section 4"},
           "executor": {"waves": stats["waves"], "tasks": stats["tail"]},
           "delay_err_vs_truth_ms": {"median": float(np.median(np.abs(de - synth.D_TRUE)) * 1e3), "max": float(np.abs(de - synth.D_TRUE).max() * 1e3)}}
    del ex, ch
    k = max(0, min(args.driver_cpu_positions, len(pos)))
    if k:
        sample = [pos[int(round(i * (len(pos) - 1) / max(k - 1, 1)))] for i in range(k)]
        cores = host_cores()
        o = OracleProblem(seed=DRIVER_SEED, threads=cores, faithful=True)
        synth.fill(o, gyro, 0, F, N, seed=DRIVER_SEED)
        hs = problem(True)
        hs.set_executor_check_every(0)
        t_cpu = t_hip = 0.0
        first, chain_h, chain_o, same_pre = [], [], [], 0
        for p0 in sample:
            # the oracle: PreSync + 4 x Sync, timed
            t0 = time.perf_counter()
            d_pre_o = o.PreSync(0.0, p0, p0 + WINDOW, STEP, RADIUS)[1]
            _, d1_o = o.Sync(d_pre_o, p0, p0 + WINDOW, 0.0, RADIUS)
            t_cpu += time.perf_counter() - t0
            winners = o.last_init_winners()
            t0 = time.perf_counter()
            d = d1_o
            for _ in range(3):
                _, d = o.Sync(d, p0, p0 + WINDOW, 0.0, RADIUS)
            t_cpu += time.perf_counter() - t0
            chain_o.append(d)
            # the HIP path, the same calls (both sides count their Sync calls alike: the sampler stream follows the call number).
            # LIKE FOR LIKE on the first call: from the oracle's PreSync delay and the oracle's GuessMotion winners (the
            # protocol of tests/noisy_scenes.py: what is compared is the fp64 Sync arithmetic, not two hypothesis searches)
            t0 = time.perf_counter()
            d_pre_h = hs.PreSync(0.0, p0, p0 + WINDOW, STEP, RADIUS)[1]
            t_hip += time.perf_counter() - t0
            same_pre += int(d_pre_h == d_pre_o)
            hs.set_init_override(winners)
            _, d1_h = hs.Sync(d_pre_o, p0, p0 + WINDOW, 0.0, RADIUS)
            first.append(abs(d1_h - d1_o) * 1e3)
            # ... and then each side on its own: its own search, its own chain of four calls (context: chaos on both sides)
            t0 = time.perf_counter()
            d = d1_h
            for _ in range(3):
                _, d = hs.Sync(d, p0, p0 + WINDOW, 0.0, RADIUS)
            t_hip += time.perf_counter() - t0
            chain_h.append(d)
        first, chain_h, chain_o = np.asarray(first), np.asarray(chain_h), np.asarray(chain_o)
        control = None
        try:
            raw = json.load(open(os.path.join(ROOT, "profiles", "r5_reassociation.json")))
            control = raw["pooled_reference_workload_noisy"]["control_reference_order_started_1e-9_s_away_s"]
        except Exception:
            pass
        out["cpu_sample"] = {"positions": k, "s": round(t_cpu, 2), "cores": cores, "kind": "port", "positions_per_s": round(k / t_cpu, 3),
                             "sample": "positions %s: PreSync + 4 x Sync each, the oracle in its faithful schedule" % sample}
        out["gpu_over_cpu_positions_per_s"] = round((len(pos) / t_ex) / (k / t_cpu), 1)
        out["presync_same_delay"] = "%d of %d" % (same_pre, k)
        out["delay_vs_oracle_ms"] = {"median": float(np.median(first)), "p90": float(np.percentile(first, 90)), "max": float(first.max()), "n": k,
                                     "within_north_star_1e-4_s": int(np.sum(first <= 0.1)),
                                     "protocol": "the first Sync call of each sampled position, both sides from the oracle's PreSync delay and "
                                                 "the oracle's GuessMotion winners (tests/noisy_scenes.py)"}
        chain = np.abs(chain_h - chain_o) * 1e3
        out["chain_of_four_calls_ms"] = {"hip_vs_oracle": {"median": float(np.median(chain)), "max": float(chain.max())},
                                         "hip_vs_truth": {"median": float(np.median(np.abs(chain_h - synth.D_TRUE)) * 1e3), "max": float(np.abs(chain_h - synth.D_TRUE).max() * 1e3)},
                                         "oracle_vs_truth": {"median": float(np.median(np.abs(chain_o - synth.D_TRUE)) * 1e3), "max": float(np.abs(chain_o - synth.D_TRUE).max() * 1e3)},
                                         "note": "each side on its own after the first call (own hypothesis search, three more calls): context, not a "
                                                 "tolerance -- the algorithm is chaotic on noisy 61 x 130 windows on BOTH sides (DESIGN.md section 6)"}
        if control:
            out["control_ms"] = {kk: float(control[kk]) * 1e3 for kk in ("median", "p90", "max") if kk in control}
            out["control_ms"]["what"] = "the oracle started 1e-9 s away from itself, 205 windows (profiles/r5_reassociation.json)"
            out["within_control"] = bool(first.max() <= 2.5 * float(control["max"]) * 1e3)
    return out


def parity_check(gyro, frames, tracks, args, ora):
    """Like for like, in the driver's own run: the HIP path once more (untimed) on EXACTLY the sample the oracle was timed
    on -- the first `frames` frames, the same candidates, Sync from the oracle's PreSync delay -- against the oracle's
    results: same PreSync arg-min, cost curve, returned delay (north star: 1e-4 s), per-iteration loss."""
    import rssync_amd
    from rssync_amd import synth

    h = rssync_amd.SyncProblem(seed=0x5EED0003, max_outer_iters=args.outer_iters, verbose=False)
    synth.fill(h, gyro, 0, frames, tracks, seed=0x5EED0003)
    delays, costs = h.presync_curve(0.0, 0, frames, args.search_step, args.search_radius)
    same_delays = len(delays) == len(ora["delays"]) and bool(np.array_equal(delays, ora["delays"]))
    curve_rel = float(np.abs(costs - ora["costs"]).max() / np.abs(ora["costs"]).max()) if same_delays else float("inf")
    c1, d1 = h.Sync(ora["d0"], 0, frames - 1, 0.0, args.search_radius)
    tr = h.sync_trace()
    n = min(len(tr), len(ora["trace"]))
    loss_rel = float(np.max(np.abs(tr[:n, 2] - np.asarray(ora["trace"])[:n, 2]) / np.abs(np.asarray(ora["trace"])[:n, 2]))) if n else float("inf")
    out = {"sample": "%d frames x %d tracks, %d candidates (the cpu_baseline's sample)" % (frames, tracks, len(delays)),
           "presync_same_index": bool(same_delays and int(np.argmin(costs)) == int(np.argmin(ora["costs"]))),
           "presync_index": int(np.argmin(costs)), "presync_curve_rel_max": curve_rel,
           "sync_delay_hip": float(d1), "sync_delay_oracle": float(ora["sync_delay"]),
           "sync_delay_abs_diff_s": float(abs(d1 - ora["sync_delay"])),
           "sync_cost_rel_diff": float(abs(c1 - ora["sync_cost"]) / abs(ora["sync_cost"])),
           "sync_outer_iters": [int(len(tr)), int(len(ora["trace"]))],
           "sync_loss_rel_max_per_iter": loss_rel, "tolerance": PARITY_TOL}
    out["ok"] = bool(out["presync_same_index"] and curve_rel <= PARITY_TOL["presync_curve_rel"] and
                     out["sync_delay_abs_diff_s"] <= PARITY_TOL["delay_s"] and len(tr) == len(ora["trace"]) and
                     loss_rel <= PARITY_TOL["loss_rel"])
    return out


if __name__ == "__main__":
    main()
