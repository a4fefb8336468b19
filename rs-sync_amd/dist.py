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

    ncclCommInitRank is COLLECTIVE: a rank that cannot follow would leave the others waiting inside it.  So first
    every rank checks locally that it can use RCCL at all (`rccl_preflight`: library and entry points resolve, rank 0
    also draws the unique id) and the ranks agree on that (all-reduce MIN) -- only then does anyone call `rccl_init`.
    Returns "native-rccl"; or, if some rank failed the PREFLIGHT and `strict` is false, that rank's message (no rank
    has touched RCCL's collective yet: the caller installs a reduce hook instead, the same on every rank).  A failure
    AFTER the preflight (inside the collective init) is fatal: it raises, the process should exit non-zero.
    """
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    err = None
    ids = [None]
    try:
        problem.rccl_preflight()
        if rank == 0:
            ids = [problem.rccl_unique_id()]
    except Exception as exc:  # noqa: BLE001
        err = "rank %d: %s" % (rank, exc)
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        errs = [None] * world
        dist.all_gather_object(errs, err)
        first = next((e for e in errs if e), "unknown")
        if strict:
            raise RuntimeError("native RCCL exchange unavailable: " + first)
        return first
    dist.broadcast_object_list(ids, src=0)
    problem.rccl_init(ids[0], rank, world)   # collective; a failure here is fatal by design (see above)
    return "native-rccl"


def shard(frame_begin, frame_end, rank, world):
    """Contiguous block of frames [begin, end) owned by `rank`."""
    n = frame_end - frame_begin
    per = (n + world - 1) // world
    b = frame_begin + rank * per
    return min(b, frame_end), min(b + per, frame_end)
