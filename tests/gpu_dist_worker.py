"""One rank of the 2-rank GPU test (both ranks share the box's one GPU; gloo carries the sums): rank 0 holds frames
of 96 tracks, rank 1 frames of 600 tracks.  A frame's kernels follow its OWN track count (size classes, round 5): rank 0
runs its frames through the one-wave kernels, and so does the single-process run that holds both kinds -- the same bits,
with no exchange to agree on anything (rounds 2-4: one exchange per call so that both took the tile kernels)."""
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
    F = 16
    n_of = lambda fr: 96 if fr < 8 else 600
    gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=6)
    b, e = (0, 8) if rank == 0 else (8, 16)
    p = rssync_amd.SyncProblem(seed=321, max_outer_iters=6)
    p.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr in range(b, e):
        p.SetTrackResult(*next(iter(synth.make_frames(gyro, fr, fr + 1, n_of(fr), seed=6))))
    p.set_reduce_hook(make_reduce_hook("cpu"))
    d, c, fc, bh = p.presync_curve(0.0, 0, F, 0.004, 0.06, per_frame=e - b)
    M, k = p.init_motion(0.03, 0, F - 1)
    cs, ds = p.Sync(0.036, 0, F - 1, 0.0, 0.2)
    with open(out, "w") as f:
        json.dump(dict(rank=rank, frames=[b, e], curve=c.tolist(), frame_costs=fc.tolist(), best_h=bh.tolist(),
                       M=M.tolist(), k=k.tolist(), sync=[cs, ds], iters=len(p.sync_trace())), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
