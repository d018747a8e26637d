"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm,
"gloo" on CPU for tests).  Filters are independent, so a batch is sharded by contiguous index
ranges with no collective in the update path; the only exchange in scope is the Monte-Carlo
statistics reduction (montecarlo.go:18-59)."""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum(arr):
    """Sum a float64 numpy array over all ranks (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return arr
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
