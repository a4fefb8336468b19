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

    The exchange is a few doubles, so its cost is latency: nothing is allocated per call (round 4's hook built a tensor from
    the array, copied it to the device, reduced, and came back through `.cpu().numpy()`, allocating twice per call).
      * "cuda": ONE pinned host buffer and ONE device tensor, made once and grown only if a call needs more (batched sync
        points exchange candidates x windows doubles): array -> pinned (memcpy) -> device (non-blocking DMA) -> all_reduce
        on the current stream -> pinned (non-blocking) -> one stream synchronisation -> array.
      * "cpu" (gloo): the all-reduce runs IN PLACE on the caller's array (torch.from_numpy shares its memory): no copy.
    """
    import torch
    import torch.distributed as dist

    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    on_gpu = str(device).startswith("cuda")
    state = {"dev": None, "pin": None, "pin_np": None, "cap": 0}
    stats = {"calls": 0, "doubles": 0, "grown": 0}

    def grow(n):
        cap = max(n, capacity, 2 * state["cap"])
        state["dev"] = torch.zeros(cap, dtype=torch.float64, device=device)
        state["pin"] = torch.zeros(cap, dtype=torch.float64).pin_memory()
        state["pin_np"] = state["pin"].numpy()
        state["cap"] = cap
        stats["grown"] += 1

    def hook(arr):
        n = arr.shape[0]
        if not on_gpu:
            t = torch.from_numpy(arr) if arr.flags["C_CONTIGUOUS"] and arr.flags["WRITEABLE"] else None
            if t is not None:
                dist.all_reduce(t)  # in place, on the caller's memory
            else:
                t = torch.tensor(arr, dtype=torch.float64)
                dist.all_reduce(t)
                arr[:] = t.numpy()
        else:
            if n > state["cap"]:
                grow(n)
            dev, pin, pin_np = state["dev"], state["pin"], state["pin_np"]
            pin_np[:n] = arr
            dev[:n].copy_(pin[:n], non_blocking=True)
            dist.all_reduce(dev[:n])
            pin[:n].copy_(dev[:n], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            arr[:] = pin_np[:n]
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
