"""Worker of tests/test_distributed_gpu.py: one rank of a torch.distributed job (gloo; several ranks may share a GPU).
Each rank owns the contiguous shard [lo, hi) of the Monte-Carlo runs and of a Vanilla filter batch, drives them
through the C ABI, all-reduces the Monte-Carlo sums and gathers the filter shards on rank 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import dist as kd
from gokalman_amd import synth
from bench import STATOD

out, RUNS, STEPS, NF, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
dev = rank % torch.cuda.device_count()
torch.cuda.set_device(dev)

# Monte-Carlo shard (montecarlo.go:92-119): runs lo..hi-1 by GLOBAL index, sums all-reduced (montecarlo.go:18-59)
lo, hi = kd.shard_range(RUNS, rank, world)
s = {kk: np.array(v, dtype=np.float64) for kk, v in STATOD.items()}
kf = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, s["x0"], s["P0"], s["F"], s["G"], s["H"], s["Q"], s["R"], nfilters=hi - lo,
                             device=dev, noise=k.NOISE_AWGN, seed=99)
mc = ga.new_monte_carlo_runs(RUNS, STEPS, 2, np.zeros((1, 2)), kf, first_run=lo, reduce=kd.allreduce_sum)

# filter shard: the same seeded batch on every rank, each rank keeps rows lo..hi-1 (no collective in the update path)
d = synth.linear_batch(NF, 6, 3, T, seed=1234)
flo, fhi = kd.shard_range(NF, rank, world)
b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"][flo:fhi], d["P0"][flo:fhi], d["F"][flo:fhi], None, d["H"][flo:fhi],
                            d["Q"][flo:fhi], d["R"][flo:fhi], device=dev)
for t in range(T):
    b.update(d["y"][t, flo:fhi])
parts = [None] * world
dist.all_gather_object(parts, (b.get(k.STATE), b.get(k.COVAR)))   # result collection for the test only (uneven shards)
if rank == 0:
    np.savez(out, sums=mc.sums, x=np.concatenate([q[0] for q in parts]), P=np.concatenate([q[1] for q in parts]))
dist.barrier()
dist.destroy_process_group()
