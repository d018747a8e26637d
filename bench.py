#!/usr/bin/env python3
"""bench.py -- headline benchmark: filter-update steps/s, 1M x 6-state/3-meas Vanilla, fp64.

One "step" = one LDKF.Update over this rank's batch of independent filters, i.e. ONE launch
of the Vanilla step kernel through the C ABI (kb_update_dev), measurements already in HBM.
Weak scaling: every GPU owns `--filters` filters (default 2^20); no data-path collective
(filters are independent).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (vanilla_reg_kernel<f64,6,3>) vs the HBM roof, measured live
                  with HIP events on the kernel's own stream; algorithmic bytes = 1488 B per
                  filter-step x filters per launch.
  cpu_baseline -- the CPU oracle (reference-order C restatement of vanilla.go:128-220, the
                  reference's Go toolchain is absent) timed on this host's cores on a bounded
                  sample of the same workload.  A reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_FILTER_STEP = 1488      # 8 B x (4n^2 + pn + p^2 + 2n + p), n=6, p=3
HBM_PEAK_GBPS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _cpu_baseline(d, budget_s=8.0):
    """Oracle timed on the host cores: bounded sample of the same workload."""
    from oracle import oracle as orc
    n_sample = min(d["x0"].shape[0], 65536)
    sub = {k: (v[:n_sample] if k != "y" else v[:, :n_sample]) for k, v in d.items()}
    T0 = 8
    best = None
    cands = sorted({orc.max_threads(), max(1, orc.max_threads() // 2)}, reverse=True)
    for th in cands:  # calibration: some hosts expose SMT siblings / quota-limited cores
        t = time.perf_counter()
        orc.ldkf_batch(orc.VANILLA, sub["x0"], sub["P0"], sub["F"], sub["H"], sub["Q"], sub["R"], sub["y"],
                       threads=th, steps=T0)
        rate = n_sample * T0 / (time.perf_counter() - t)
        if best is None or rate > best[1]:
            best = (th, rate)
    th, rate = best
    T = int(max(4, budget_s * rate / n_sample))
    t = time.perf_counter()
    _, _, nerr = orc.ldkf_batch(orc.VANILLA, sub["x0"], sub["P0"], sub["F"], sub["H"], sub["Q"], sub["R"], sub["y"],
                                threads=th, steps=T)
    dt = time.perf_counter() - t
    return {
        "value": n_sample * T / dt, "unit": "filter-update steps/s", "cores": th, "kind": "port",
        "sample": "%d filters x %d steps of the same synthetic batch (C oracle, OpenMP over filters, %.1f s)" % (n_sample, T, dt),
        "errors": int(nerr),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--filters", type=int, default=1 << 20, help="filters per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused-steps", type=int, default=16, help="T of the extra time-fused measurement (0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (testing the N>1 path on fewer GPUs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and world > ndev:
        raise SystemExit("%d ranks but %d GPUs: RCCL needs one GPU per rank" % (world, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)

    import gokalman_amd as ga
    from gokalman_amd import _capi as k
    from gokalman_amd import synth

    N = args.filters
    n, p = 6, 3
    POOL = 4  # distinct measurement sets cycled through the timed steps
    d = synth.linear_batch(N, n, p, POOL, seed=synth.SEED + rank)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], device=local_rank)
    y_dev = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda(local_rank)  # planar [POOL][p][N]
    ptrs = [y_dev[t].data_ptr() for t in range(POOL)]
    kstream = torch.cuda.ExternalStream(b.stream(), device=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        b.synchronize()

    for t in range(args.warmup):
        b.update_dev(ptrs[t % POOL], N)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(kstream)
    for t in range(args.steps):
        b.update_dev(ptrs[t % POOL], N)
    ev1.record(kstream)
    b.synchronize()
    torch.cuda.synchronize()
    local_s = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the kernel's stream
    if world > 1:
        tt = torch.tensor([local_s], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall_s = float(tt.item())
        dist.barrier()
    else:
        wall_s = local_s
    nbad = int(np.count_nonzero(b.status()))

    # extra: the caller loop fused into one launch (x, P, model resident in registers)
    fused = None
    if args.fused_steps > 0:
        T = args.fused_steps
        yy = y_dev.repeat((T + POOL - 1) // POOL, 1, 1)[:T].contiguous()
        b.update_steps_dev(yy.data_ptr(), N, T)
        b.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        f0.record(kstream)
        for _ in range(reps):
            b.update_steps_dev(yy.data_ptr(), N, T)
        f1.record(kstream)
        b.synchronize()
        fms = f0.elapsed_time(f1) / reps
        fused = {"steps_per_launch": T, "ms_per_launch": fms, "value": N * T / (fms * 1e-3),
                 "unit": "filter-update steps/s (1 GPU, kb_update_steps_dev)"}

    if rank == 0:
        value = world * N * args.steps / wall_s
        achieved = ALGO_BYTES_PER_FILTER_STEP * N / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))  # measured at tj["filters"] filters per launch; scales linearly with the batch
                traffic = tj["hbm_bytes_per_launch"] * (N / float(tj.get("filters", 1 << 20)))
            except Exception:
                traffic = None
        out = {
            "metric": "filter-update steps/s (whole node), 1M x 6-state Vanilla",
            "value": value, "unit": "filter-update steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall_s / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %d independent 6-state/3-meas Vanilla filters per GPU, fp64, "
                                   "per-filter F/H/Q/R, Noiseless, one kb_update_dev launch per step" % N,
                       "filters_per_gpu": N, "n": n, "p": p, "kernel": "vanilla_reg_kernel<double,6,3,0>",
                       "sharding": "independent filter shards, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FILTER_STEP * N},
            "filters_with_error_status": nbad,
        }
        if fused:
            out["fused"] = fused
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = _cpu_baseline(d)
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
