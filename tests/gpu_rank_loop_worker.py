"""One rank of the 2-rank test of the RANKED device-driven Sync loop (both ranks share the box's one GPU, gloo
carries the sums; RCCL cannot put two ranks on one device).  The loop's structure with ranks -- this rank's window
sums -> sum over the ranks -> decisions on every rank -- is the RCCL path's; only the transport of the middle step
differs (rssync_ext_set_hook_device_loop: the reduce hook, called between the kernels)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import rssync_amd
    from rssync_amd import synth
    from rssync_amd.dist import make_reduce_hook
    F, N = 40, 130
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=8)
    b, e = (0, 26) if rank == 0 else (26, 40)       # uneven split
    res = {}
    for mode in ("host", "device"):
        p = rssync_amd.SyncProblem(seed=99, max_outer_iters=14)
        p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
        for fr in range(b, e):
            p.SetTrackResult(*next(iter(synth.make_frames(gyro, fr, fr + 1, N, seed=8))))
        p.set_reduce_hook(make_reduce_hook("cpu"))
        p.set_hook_device_loop(mode == "device")
        x0 = p.exchange_stats()[0]
        cs = p.Sync(0.036, 0, F - 1, 0.0, 0.2)
        x1 = p.exchange_stats()[0]
        tr = p.sync_trace().tolist()
        # several windows: [0, 20] lies on rank 0 alone, [28, 39] on rank 1 alone, [10, 35] on both, [100, 120] nowhere
        cw, dw = p.sync_windows([0.036, 0.037, 0.0365, 0.03], [0, 28, 10, 100], [20, 39, 35, 120], 0.0, 0.2)
        wtr = [p.window_trace(w).tolist() for w in range(4)]
        one = p.Sync(0.0362, 0, 20, 0.0, 0.2)        # a single window none of whose frames rank 1 holds
        one_tr = p.sync_trace().tolist()
        sc = p.SyncSimplified(0.0355, 0, F - 1, 0.0, 0.1)
        res[mode] = dict(sync=list(cs), trace=tr, exchanges=x1 - x0, cw=cw.tolist(), dw=dw.tolist(), wtr=wtr,
                         one=list(one), one_trace=one_tr,
                         simplified=list(sc), simplified_trace=p.sync_trace().tolist())
    with open(out, "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
