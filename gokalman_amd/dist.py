"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm,
"gloo" on CPU for tests).  Filters are independent, so a batch is sharded by contiguous index
ranges with no collective in the update path; the only exchange in scope is the Monte-Carlo
statistics reduction (montecarlo.go:18-59)."""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: SURVEY section 8e's formula, GPU g owns [g N / G, (g + 1) N / G) (integer
    division; sizes differ by at most one) -- the same split kb_sharded_create makes (csrc/kb_sharded.hip), so that a job of one
    process per GPU and the one-process sharded batch put every filter (and every Monte-Carlo run) on the same device."""
    n_items, rank, world = int(n_items), int(rank), int(world)
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


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
