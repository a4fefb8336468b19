"""Cost of one exchange (a sum of a few doubles over the ranks) on the GPU box, one rank:
no exchange vs the torch.distributed hook vs the library's own RCCL communicator.
Measured through Sync on a small problem (2 exchanges per outer iteration + 1)."""
import json, os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import rssync_amd
from rssync_amd import synth
from rssync_amd.dist import make_reduce_hook, use_native_rccl
F, N = 32, 128
gyro = synth.make_gyro(0.0, (F + 2) / synth.FPS, seed=6)
out = {}
for name in ("plain", "torch_hook", "native_rccl"):
    p = rssync_amd.SyncProblem(seed=123, max_outer_iters=40)
    synth.fill(p, gyro, 0, F, N, seed=6)
    if name == "torch_hook":
        p.set_reduce_hook(make_reduce_hook())
    if name == "native_rccl":
        use_native_rccl(p)
    p.Sync(0.03, 0, F - 1, 0.0, 0.5)
    t = time.perf_counter()
    reps = 10
    for _ in range(reps):
        p.Sync(0.03, 0, F - 1, 0.0, 0.5)
    dt = (time.perf_counter() - t) / reps
    it = len(p.sync_trace())
    out[name] = {"sync_ms": round(1e3 * dt, 3), "outer_iters": it, "exchanges": 2 * it + 1}
for name in ("torch_hook", "native_rccl"):
    out[name]["us_per_exchange"] = round(1e6 * (out[name]["sync_ms"] - out["plain"]["sync_ms"]) / 1e3 / out[name]["exchanges"], 1)
print(json.dumps(out))
dist.destroy_process_group()
