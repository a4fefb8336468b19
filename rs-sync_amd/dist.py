"""Frame-sharded multi-GPU runs: one process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL on ROCm) for the only exchange the path has -- a sum of a few doubles per step.

Every rank holds its own frames in its own ``SyncProblem``; PreSync exchanges the candidate cost
vector once, Sync exchanges {loss, d loss / d delay} and the batched line-search losses (two
all-reduces per outer iteration).  The optimiser state is replicated: every rank sees the same
sums and takes the same decisions, so no other traffic is needed.
"""
import numpy as np


def make_reduce_hook(device=None, capacity=8192):
    """Return fn(np.ndarray float64) that sums the array in place over all ranks.

    device: torch device of the staging tensor ("cuda" for the nccl/RCCL backend, "cpu" for gloo).
    """
    import torch
    import torch.distributed as dist

    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    state = {"buf": torch.zeros(capacity, dtype=torch.float64, device=device)}
    stats = {"calls": 0, "doubles": 0}

    def hook(arr):
        n = arr.shape[0]
        if n > state["buf"].shape[0]:
            # batched sync points exchange candidates x windows doubles (98 x 800 = 78 k): grow, never refuse
            state["buf"] = torch.zeros(max(n, 2 * state["buf"].shape[0]), dtype=torch.float64, device=device)
        buf = state["buf"]
        buf[:n].copy_(torch.from_numpy(arr))
        dist.all_reduce(buf[:n])
        arr[:] = buf[:n].cpu().numpy()
        stats["calls"] += 1
        stats["doubles"] += n

    hook.stats = stats
    return hook


def use_native_rccl(problem, strict=False):
    """Give `problem` its own RCCL communicator (no Python in the exchange): rank 0's unique id is
    broadcast through the already initialised torch.distributed group, then every rank joins.

    Returns "native-rccl", or -- if any rank could not join (librccl missing, init refused) and `strict` is
    false -- the first failure's message after every rank has left the communicator again: the caller then
    installs a reduce hook instead.  The ranks agree on the outcome, so they never end up on different paths.
    """
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    err = None
    ids = [None]
    if rank == 0:
        try:
            ids = [problem.rccl_unique_id()]
        except Exception as exc:  # noqa: BLE001
            err = "rank 0: %s" % exc
    dist.broadcast_object_list(ids, src=0)
    joined = False
    if ids[0] is not None:
        try:
            problem.rccl_init(ids[0], rank, world)
            joined = True
        except Exception as exc:  # noqa: BLE001
            err = "rank %d: %s" % (rank, exc)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    ok = torch.tensor([1 if joined else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        return "native-rccl"
    errs = [None] * world
    dist.all_gather_object(errs, err)
    first = next((e for e in errs if e), "unknown")
    if joined:
        try:
            problem.rccl_shutdown()
        except Exception:  # noqa: BLE001
            pass
    if strict:
        raise RuntimeError("native RCCL exchange unavailable: " + first)
    return first


def shard(frame_begin, frame_end, rank, world):
    """Contiguous block of frames [begin, end) owned by `rank`."""
    n = frame_end - frame_begin
    per = (n + world - 1) // world
    b = frame_begin + rank * per
    return min(b, frame_end), min(b + per, frame_end)
