"""Config D(i): Monte-Carlo ensemble of the statOD5044 pure predictor (montecarlo.go:92-119) sharded
over the ranks of a torch.distributed job (one process per GPU; `--dist-backend gloo` lets several
ranks share one GPU for testing).  Each rank runs `--runs` runs (weak scaling), the per-step
(sum, sum of squares) are all-reduced, rank 0 prints one JSON line.

    python scripts/bench_mc.py --runs 1048576
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/bench_mc.py --runs 1048576
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import dist as kd

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=1 << 20, help="runs per GPU")
ap.add_argument("--steps", type=int, default=1086)
ap.add_argument("--dist-backend", default="nccl")
args = ap.parse_args()
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count())
torch.cuda.set_device(local)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dist_backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(args.dist_backend)

F = np.array([[1, 0.1, 0, 7.726e-2], [4.015e-7, 1, 0, 1.545], [-2.319e-16, -1.732e-9, 1, 0.1], [-6.956e-15, -3.465e-8, 0, 1]])
G = np.array([[5e-3, 3.85e-7], [0.1, 1.157e-5], [-5.775e-11, 7.487e-7], [1.732e-9, 1.498e-5]])
H = np.array([[1.0, 0, 0, 0], [0, 0, 1, 0]])
Q = np.array([[6.669e-16, 1.001e-14, 3.823e-19, 5.150e-18], [1.001e-14, 2.002e-13, 1.030e-17, 1.545e-16],
              [3.862e-19, 1.030e-17, 6.667e-19, 1.000e-17], [5.150e-18, 1.545e-16, 1.000e-17, 2.000e-16]])
R = np.diag([2e-3, 2e-5]) / 0.1
x0, P0 = np.array([2, 0.5, 0, 0.0]), np.diag([5, 1, 0.01, 1e-5])

kf = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, x0, P0, F, G, H, Q, R, nfilters=args.runs, device=local,
                             noise=k.NOISE_AWGN, seed=2016)
first = rank * args.runs  # a run's noise depends only on its global index
ga.new_monte_carlo_runs(args.runs, 4, 2, np.zeros((1, 2)), kf, first_run=first)  # warm-up
if world > 1:
    dist.barrier()
torch.cuda.synchronize()
t = time.perf_counter()
mc = ga.new_monte_carlo_runs(args.runs * world, args.steps, 2, np.zeros((1, 2)), kf, first_run=first,
                             reduce=kd.allreduce_sum if world > 1 else None)
torch.cuda.synchronize()
dt = torch.tensor([time.perf_counter() - t], dtype=torch.float64)
if world > 1:
    if args.dist_backend == "nccl":
        dt = dt.cuda()
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"metric": "Monte-Carlo run-steps/s (whole job)", "value": world * args.runs * args.steps / float(dt.item()),
                      "n_gpus": world, "runs_total": world * args.runs, "steps": args.steps, "seconds": float(dt.item()),
                      "collective": "all_reduce(SUM) of %d doubles" % (args.steps * 2 * 4),
                      "mean_last": mc.mean(args.steps - 1).tolist(), "stddev_last": mc.stddev(args.steps - 1).tolist()}))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
