"""Single-rank RCCL run on the GPU box: the product library on cuda:0 with the reduce hook of
bench.py's N>1 path (torch.distributed backend "nccl" = RCCL, staging tensor on the device).
With one rank the all-reduce is the identity, so the results must equal a run without a hook."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port, out = int(sys.argv[1]), sys.argv[2]
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    import rssync_amd
    from rssync_amd import synth
    from rssync_amd.dist import make_reduce_hook, use_native_rccl
    F, N = 32, 128
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=6)
    res = {}
    for name in ("plain", "rccl", "native", "native_host"):
        p = rssync_amd.SyncProblem(seed=123, max_outer_iters=12)
        synth.fill(p, gyro, 0, F, N, seed=6)
        if name == "rccl":
            hook = make_reduce_hook()          # picks the device from the backend
            p.set_reduce_hook(hook)
        if name.startswith("native"):          # the library's own communicator, id broadcast by torch
            use_native_rccl(p)
            if name == "native_host":          # ... with Sync's outer loop on the host (one exchange per launch)
                p.set_host_loop(True)
        c0, d0 = p.PreSync(0.0, 0, F, 0.004, 0.1)
        x0 = p.exchange_stats()[0]
        c1, d1 = p.Sync(d0, 0, F - 1, 0.0, 0.2)
        x1 = p.exchange_stats()[0]
        res[name] = [c0, d0, c1, d1, len(p.sync_trace())]
        res[name + "_trace"] = p.sync_trace().tolist()
        res[name + "_sync_exchanges"] = x1 - x0
        # several windows in lock-step, one of them without frames, one window per stream group otherwise
        cw, dw = p.sync_windows([0.036, 0.037, 0.0365, 0.03], [0, 8, 16, 100], [15, 23, 31, 120], 0.0, 0.2)
        res[name + "_windows"] = [cw.tolist(), dw.tolist(), [p.window_trace(w).tolist() for w in range(4)]]
        if name.startswith("native"):
            p.rccl_shutdown()
    res["exchanges"] = hook.stats["calls"]
    res["backend"] = dist.get_backend()
    with open(out, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
