"""bench.py --gpus N without an external launcher: the parent starts N rank processes itself (never touching
a GPU), waits, and passes their status on.  Rehearsed here on the CPU stand-in for the device ABI with the
gloo backend (the same launcher and exchange code paths; the printed value is marked as a rehearsal).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--cpu-frames", "0", "--tracks", "96", "--search-step", "0.004",
         "--search-radius", "0.1", "--outer-iters", "6"]


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    return p.returncode, lines, p.stderr


def _lib(hosttest_lib):
    return os.path.join(ROOT, "tests", "_build", "librssync_hosttest.so")


def test_self_spawned_ranks_match_one_process(hosttest_lib):
    lib = _lib(hosttest_lib)
    rc, one, err = _run([sys.executable, "bench.py", "--gpus", "1", "--frames", "16", "--rehearse-cpu", lib] + SMALL)
    assert rc == 0 and len(one) == 1, err
    rc, two, err = _run([sys.executable, "bench.py", "--gpus", "2", "--frames", "8", "--rehearse-cpu", lib] + SMALL)
    assert rc == 0, err
    assert len(two) == 1, "exactly one JSON line (rank 0's)"
    a, b = json.loads(one[0]), json.loads(two[0])
    assert a["n_gpus"] == 1 and a["multi_gpu"]["launcher"] == "direct" and a["multi_gpu"]["exchange"] is None
    assert b["n_gpus"] == 2 and b["multi_gpu"]["launcher"] == "self-spawned" and b["multi_gpu"]["processes"] == 2
    assert b["multi_gpu"]["exchange"].startswith("torch-gloo-hook") and b["multi_gpu"]["exchanges_per_step"] >= 3
    assert "rehearsal" in a and "rehearsal" in b
    # the same 16-frame window, whole or sharded over two ranks: same arg-min, same refinement
    assert b["result"]["presync_delay"] == a["result"]["presync_delay"]
    assert b["result"]["sync_delay"] == pytest.approx(a["result"]["sync_delay"], abs=1e-9)
    assert b["config"]["sync_outer_iters"] == a["config"]["sync_outer_iters"]
    # weak scaling bookkeeping: per-GPU work fixed, whole-job value
    assert b["config"]["frames_per_gpu"] == 8 and b["scaling"] == "weak"


def test_external_launcher_still_works(hosttest_lib):
    lib = _lib(hosttest_lib)
    rc, out, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29611", "bench.py", "--gpus", "2", "--frames", "8",
                         "--rehearse-cpu", lib] + SMALL)
    assert rc == 0 and len(out) == 1, err
    d = json.loads(out[0])
    assert d["n_gpus"] == 2 and d["multi_gpu"]["launcher"] == "torch.distributed.run"


def test_inproc_mode_one_object_several_devices(hosttest_lib):
    lib = _lib(hosttest_lib)
    rc, out, err = _run([sys.executable, "bench.py", "--gpus", "2", "--mode", "inproc", "--frames", "8", "--rehearse-cpu", lib] + SMALL)
    assert rc == 0 and len(out) == 1, err
    d = json.loads(out[0])
    assert d["n_gpus"] == 2 and d["multi_gpu"]["mode"] == "inproc" and d["multi_gpu"]["devices_per_process"] == 2
    assert d["multi_gpu"]["exchanges_per_step"] == 0


def test_a_failing_rank_fails_the_run(hosttest_lib):
    rc, out, err = _run([sys.executable, "bench.py", "--gpus", "2", "--frames", "8", "--rehearse-cpu", "/nonexistent/lib.so"] + SMALL)
    assert rc != 0 and not out
    assert "rank process" in err


def test_baseline_config_is_named_and_config5_runs_with_ranks(hosttest_lib):
    """the JSON line says which BASELINE.json config it is; --workload c5 (timestamped gyro + orientation sweep,
    frames sharded over the ranks, ONE exchange for the whole pipelined sweep) gives the same ranking with 1 and 2 ranks"""
    lib = _lib(hosttest_lib)
    c5 = ["--workload", "c5", "--orientations", "4", "--steps", "1", "--warmup", "0", "--cpu-frames", "0", "--tracks", "64",
          "--search-step", "0.004", "--search-radius", "0.1"]
    rc, one, err = _run([sys.executable, "bench.py", "--gpus", "1", "--frames", "12", "--rehearse-cpu", lib] + c5)
    assert rc == 0 and len(one) == 1, err
    rc, two, err = _run([sys.executable, "bench.py", "--gpus", "2", "--frames", "6", "--rehearse-cpu", lib] + c5)
    assert rc == 0 and len(two) == 1, err
    a, b = json.loads(one[0]), json.loads(two[0])
    for d in (a, b):
        assert d["config"]["baseline_config"].startswith("config 5")
        assert d["result"]["best_orientation"] == "XYZ"
    assert b["result"]["best_delay"] == a["result"]["best_delay"]
    assert b["result"]["cost_ratio_best_to_second"] == pytest.approx(a["result"]["cost_ratio_best_to_second"], rel=1e-9)
    assert b["multi_gpu"]["exchanges_per_step"] == 1           # the [orientations][candidates] cost matrix, once (rounds 1-5: one per orientation)
    # the default workload names its config too; 2048 frames per GPU is config 4's shard size
    rc, out, err = _run([sys.executable, "bench.py", "--gpus", "1", "--frames", "16", "--rehearse-cpu", lib] + SMALL)
    assert rc == 0, err
    assert json.loads(out[0])["config"]["baseline_config"].startswith("none (16 frames")
    sys.path.insert(0, ROOT)
    import bench
    assert bench.parse_args([]).exchange == "torch" and bench.parse_args([]).frames == 4096


def test_eight_ranks_rehearsed_on_the_cpu_stand_in(hosttest_lib):
    """What the driver's first SCALE run will execute -- `bench.py --gpus 8`, one rank per GPU -- rehearsed with eight gloo
    ranks on the CPU stand-in (tiny frames): one JSON line, the config named, the exchange named, and the exchanges per
    step within the budget DESIGN.md section 4 writes down: 1 for PreSync + 2 per outer iteration (3 when a line search
    needs its later trials) + 1 for the final loss.  (The stand-in has no device loop: this is the host loop's count; the
    device loop's -- two per ENQUEUED iteration -- is asserted on the GPU, tests/test_gpu_parity.py.)"""
    lib = _lib(hosttest_lib)
    rc, out, err = _run([sys.executable, "bench.py", "--gpus", "8", "--frames", "4", "--rehearse-cpu", lib] + SMALL, timeout=900)
    assert rc == 0 and len(out) == 1, err[-2000:]
    d = json.loads(out[0])
    assert d["n_gpus"] == 8 and d["multi_gpu"]["processes"] == 8 and d["multi_gpu"]["launcher"] == "self-spawned"
    assert d["multi_gpu"]["exchange"].startswith("torch-gloo-hook")
    assert d["multi_gpu"]["rccl_ranks"] == 0                   # gloo here; with the nccl backend the line says 8 (below)
    assert d["config"]["baseline_config"].startswith("none (4 frames") and d["config"]["frames_per_gpu"] == 4
    its = max(d["config"]["sync_outer_iters"])
    assert 1 + 2 * min(d["config"]["sync_outer_iters"]) + 1 <= d["multi_gpu"]["exchanges_per_step"] <= 1 + 3 * its + 1
    # the same 32-frame window in one process: same arg-min, same refinement
    rc, one, err = _run([sys.executable, "bench.py", "--gpus", "1", "--frames", "32", "--rehearse-cpu", lib] + SMALL)
    assert rc == 0, err[-2000:]
    a = json.loads(one[0])
    assert d["result"]["presync_delay"] == a["result"]["presync_delay"]
    assert d["result"]["sync_delay"] == pytest.approx(a["result"]["sync_delay"], abs=1e-9)


def test_the_line_names_rccl_whenever_the_backend_is_nccl():
    """`rccl_ranks` = the world size whenever RCCL carries the sums -- through the library's own communicator or through
    torch.distributed behind the reduce hook (the default): round 4's line said 0 for an 8-rank RCCL run through the hook.
    (Source-level: a run with the nccl backend needs GPUs; the expression is what the line prints.)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"rccl_ranks": world if (world > 1 and (exchange == "native-rccl" or backend == "nccl")) else 0' in src
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args([])
    assert a.hook_device_loop is True and bench.parse_args(["--no-hook-device-loop"]).hook_device_loop is False
    assert a.exchange == "torch"          # (native stays opt-in until it has run with two real ranks: DESIGN.md section 4)
