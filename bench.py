#!/usr/bin/env python3
"""bench.py -- headline benchmark: filter-update steps/s, 1M x 6-state/3-meas Vanilla, fp64.

One "step" = one LDKF.Update over this rank's batch of independent filters, i.e. ONE launch
of the Vanilla step kernel through the C ABI (kb_update_dev), measurements already in HBM.
Weak scaling: every GPU owns `--filters` filters (default 2^20); no collective in the update
path (filters are independent).  The job's epilogue holds the collectives
the path does have: an all-reduce of the per-rank step / error counts, and the Monte-Carlo
statistics reduction of montecarlo.go:18-59 (config D, `extra.mc`).

    python bench.py [--gpus N --steps K --warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment this process starts N ranks itself
(`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process, before
anything here touches the GPU) and forwards their exit code; under torchrun it is one rank.
It never reports a 1-rank number for an N-rank request: fewer visible GPUs than ranks is an error.

Rank 0 prints ONE JSON line (contract in the task statement): the compact headline object of gokalman_amd/benchline.py
(<= 8 KB: contract keys, `roofline`, `cpu_baseline`, `parity`, `ranks`, and {value, kernel_ms, frac, parity_ok} per leg under
`legs`).  The FULL document is written to --full-out (default gpurun_out/bench_full.json) and holds these objects:
  roofline     -- dominant kernel (vanilla_reg_kernel<f64,6,3>) against the HBM roof.  `frac` is
                  PHYSICAL: bytes the launch moves (the packed working set of 1104 B per filter, which
                  the rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE counters of the committed profile confirm)
                  / kernel time (HIP events on the kernel's stream, live) / 8 TB/s.  At 1M filters
                  that is a FABRIC-SIDE figure: the 226 MB state block lives in the Infinity Cache by
                  design, whose hits those counters include; `dram_frac_lower_bound` counts the model
                  + measurement stream alone (what certainly comes from DRAM) and `hbm_only` repeats
                  the out_of_cache measurement, where every byte does.  The contract's algorithmic
                  figure (1488 B per filter-step, full matrices) is `frac_algorithmic`; it can
                  exceed 1 because a SymDense carries only its upper triangle.
  repetitions  -- the timed block (K steps, barrier + synchronise on both sides) repeated; `value` is
                  the FIRST block (the contract's), the median is next to it.
  parity       -- 4096 filters x 20 steps of the same synthetic batch through the same entry point
                  against the CPU oracle (the checker; outside every timed region).
  strong_scaling -- `--filters` filters IN TOTAL split over the ranks (GPU g owns
                  [g N / G, (g + 1) N / G)); `value` / `scaling` stay the weak-scaling figures.
  host_path    -- kb_update with host measurements (H2D + pack + step + synchronise per call).
  out_of_cache -- the same kernel on a batch whose state block cannot stay in the Infinity Cache.
  fused        -- the caller loop inside one launch (VALU-issue-bound), with its issue-rate roofline.
  extra.mc / extra.hybrid_ekf -- config D sharded over the same ranks; extra.squareroot (config C) and extra.srif_fp32
  (config E) likewise, one shard per rank.
  cpu_baseline -- the CPU oracle (reference-order C restatement of vanilla.go:128-220, the
                  reference's Go toolchain is absent) timed on this host's cores on a bounded
                  sample of the same workload.  A reported baseline, not the target.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_STATE, N_MEAS = 6, 3
HEADLINE_KERNEL = "vanilla_reg_kernel<double, 6, 3, 0"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--filters", type=int, default=1 << 20, help="filters per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused-steps", type=int, default=100, help="T of the extra time-fused measurement (0 = skip); SURVEY 8d's T = 100 "
                    "steps per repetition: the per-launch load / store of state and model is amortised over T steps (T = 16: 12.5 G steps/s, T = 100: 20 G)")
    ap.add_argument("--ooc-filters", type=int, default=1 << 22,
                    help="filters of the out-of-Infinity-Cache measurement of the same kernel (0 = skip)")
    ap.add_argument("--mc-runs", type=int, default=1 << 20, help="Monte-Carlo runs per GPU for extra.mc (0 = skip)")
    ap.add_argument("--mc-steps", type=int, default=1086)
    ap.add_argument("--hybrid-total", type=int, default=8 << 20, help="filters of the WHOLE configs[3] Hybrid-EKF ensemble (8M), run once on one GPU when world == 1 (0 = skip)")
    ap.add_argument("--mc-total", type=int, default=8 << 20, help="runs of the WHOLE configs[3] ensemble (8M): every rank takes mc_total / world of them as "
                    "consecutive shards of --mc-runs runs (one rank: eight shards one after the other); 0 = skip extra.mc.ensemble")
    ap.add_argument("--hybrid-filters", type=int, default=1 << 20, help="Hybrid EKF filters per GPU for extra.hybrid_ekf (0 = skip)")
    ap.add_argument("--shared-filters", type=int, default=1 << 20, help="Vanilla 6/3 filters per GPU sharing one model for extra.shared_model (0 = skip)")
    ap.add_argument("--sqrt-filters", type=int, default=1 << 20, help="SquareRoot 6/3 filters per GPU for extra.squareroot, config C (0 = skip)")
    ap.add_argument("--srif-filters", type=int, default=1 << 18, help="SRIF 12/6 fp32 filters per GPU for extra.srif_fp32, config E (0 = skip)")
    ap.add_argument("--split-filters", type=int, default=1 << 18, help="Vanilla 12/6 fp64 filters per GPU for extra.vanilla_12x6: the split-lane kernels for 8 < n <= 16 (0 = skip)")
    ap.add_argument("--repeat", type=int, default=5, help="timed blocks of --steps steps (the first one is the contract's `value`)")
    ap.add_argument("--chisq-runs", type=int, default=1 << 20, help="chi-square runs per GPU for extra.chisq (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--init-dist", action="store_true",
                    help="initialise torch.distributed and run every collective even for ONE rank (exercises the RCCL code path on a single GPU)")
    ap.add_argument("--full-out", default=os.path.join("gpurun_out", "bench_full.json"),
                    help="where rank 0 writes the FULL result document (every leg with its roofline, provenance, parity details); the "
                    "final stdout line is the compact headline object of gokalman_amd/benchline.py (<= 8 KB), '' = do not write")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, one GPU per rank) or gloo (testing the N>1 path with several ranks on one GPU)")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """--gpus N > 1 without a torchrun environment: start the N ranks as a child job.  Nothing in this process has
    touched the GPU (torch.cuda.device_count() does not initialise it on this image); no exec."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.dist_backend == "nccl" and ndev < args.gpus:
        raise SystemExit("--gpus %d requested but only %d GPU(s) visible: RCCL needs one GPU per rank" % (args.gpus, ndev))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


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


def _parity(ga, k, synth):
    """The parity gate of SURVEY 8d inside the bench line: the first 4096 filters x 20 steps of the synthetic batch through the
    same entry point (kb_update_dev would need the device copy: kb_update is the same kernel) against the CPU oracle.  The
    oracle is the CHECKER here, outside every timed region."""
    import numpy as np
    from oracle import oracle as orc
    N, T = 4096, 20
    d = synth.linear_batch(N, N_STATE, N_MEAS, T)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    for t in range(T):
        b.update(d["y"][t], snapshot=False)
    xo, Po, nerr = orc.ldkf_batch(orc.VANILLA, d["x0"], d["P0"], d["F"], d["H"], d["Q"], d["R"], d["y"])
    ex, eP = synth.rel_frobenius(b.get(k.STATE), xo), synth.rel_frobenius(b.get(k.COVAR), Po)
    return {"filters": N, "steps": T, "max_rel_frobenius_state": ex, "max_rel_frobenius_covariance": eP, "tolerance": 1e-9,
            "ok": bool(ex <= 1e-9 and eP <= 1e-9 and nerr == 0 and not b.status().any()),
            "against": "oracle/gokalman_oracle.c (reference-order C restatement of vanilla.go:128-220)"}


def _leg_parity(ga, k, synth, leg):
    """The same gate for the secondary legs: 4096 filters x 20 steps of each leg's configuration through
    the entry points the leg times, against the CPU oracle -- outside every timed region, rank 0 only.  `srif_fp32` also reports
    the ACHIEVED error of the fp32 kernel (max relative Frobenius error of R and b against the fp64 oracle on the same inputs)."""
    import numpy as np
    import torch
    from oracle import oracle as orc
    N, T = 4096, 20
    out = {"filters": N, "steps": T, "against": "oracle/gokalman_oracle.c"}
    if leg == "information_12x6":
        # Information carries (i, I): those are compared (RAW_VEC / RAW_MAT), filter by filter against the oracle's
        # NewInformationFromState filters -- the first 512 of the 4096 (State() / Covariance() are inverses on top: cond x 1e-9)
        d = synth.linear_batch(N, 12, 6, T, seed=synth.SEED + 77)
        b = ga.FilterBatch.new_ldkf(k.INFORMATION, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_INFO_FROM_STATE)
        y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
        for t in range(T):
            b.update_dev(y[t].data_ptr(), N)
        b.synchronize()
        M = 512
        fs = [orc.Filter.information_from_state(d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i]) for i in range(M)]
        nerr = 0
        for f_, i in zip(fs, range(M)):
            for t in range(T):
                nerr += f_.update(d["y"][t, i]) != orc.OK
        ei = synth.rel_frobenius(b.get(k.RAW_VEC, 0, M), np.array([f_.raw_vec() for f_ in fs]))
        eI = synth.rel_frobenius(b.get(k.RAW_MAT, 0, M), np.array([f_.raw_mat() for f_ in fs]))
        out.update({"oracle_filters": M, "max_rel_frobenius_information_vector": ei, "max_rel_frobenius_information_matrix": eI, "tolerance": 1e-9,
                    "ok": bool(ei <= 1e-9 and eI <= 1e-9 and nerr == 0 and not b.status().any())})
        return out
    if leg == "squareroot_fused":
        # config C's time-fused kernel (Newton reciprocals, x / S / model resident): against the oracle, and against T one-step launches
        d = synth.linear_batch(N, N_STATE, N_MEAS, T, seed=synth.SEED + 79)
        b = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
        b1 = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
        y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
        torch.cuda.synchronize()
        b.update_steps_dev(y.data_ptr(), N, T)
        for t in range(T):
            b1.update_dev(y[t].data_ptr(), N)
        b.synchronize(); b1.synchronize()
        xo, Po, nerr = orc.ldkf_batch(orc.SQUAREROOT, d["x0"], d["P0"], d["F"], d["H"], d["Q"], d["R"], d["y"])
        ex, eP = synth.rel_frobenius(b.get(k.STATE), xo), synth.rel_frobenius(b.get(k.COVAR), Po)
        out.update({"max_rel_frobenius_state": ex, "max_rel_frobenius_covariance": eP, "tolerance": 1e-9,
                    "vs_per_step_kernel": {"state": synth.rel_frobenius(b.get(k.STATE), b1.get(k.STATE)),
                                           "covariance": synth.rel_frobenius(b.get(k.COVAR), b1.get(k.COVAR))},
                    "ok": bool(ex <= 1e-9 and eP <= 1e-9 and nerr == 0 and not b.status().any() and b.step() == T)})
        return out
    if leg == "fused":
        # kb_update_steps_dev (T steps inside one launch) evaluates the Joseph form in the distributed order with Newton reciprocals:
        # NOT bit-identical to T calls of kb_update_dev (include/gokalman_amd.h) -- held to the oracle like every other leg, and the
        # distance to the per-step kernel on the same input is reported next to it
        d = synth.linear_batch(N, N_STATE, N_MEAS, T, seed=synth.SEED + 78)
        b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
        b1 = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
        y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
        torch.cuda.synchronize()
        b.update_steps_dev(y.data_ptr(), N, T)
        for t in range(T):
            b1.update_dev(y[t].data_ptr(), N)
        b.synchronize(); b1.synchronize()
        xo, Po, nerr = orc.ldkf_batch(orc.VANILLA, d["x0"], d["P0"], d["F"], d["H"], d["Q"], d["R"], d["y"])
        ex, eP = synth.rel_frobenius(b.get(k.STATE), xo), synth.rel_frobenius(b.get(k.COVAR), Po)
        out.update({"max_rel_frobenius_state": ex, "max_rel_frobenius_covariance": eP, "tolerance": 1e-9,
                    "vs_per_step_kernel": {"state": synth.rel_frobenius(b.get(k.STATE), b1.get(k.STATE)),
                                           "covariance": synth.rel_frobenius(b.get(k.COVAR), b1.get(k.COVAR))},
                    "ok": bool(ex <= 1e-9 and eP <= 1e-9 and nerr == 0 and not b.status().any() and b.step() == T)})
        return out
    if leg in ("squareroot", "shared_model", "vanilla_12x6", "squareroot_12x6", "vanilla_10x4", "vanilla_16x8"):
        nn, pp = (12, 6) if leg in ("vanilla_12x6", "squareroot_12x6") else ((10, 4) if leg == "vanilla_10x4" else ((16, 8) if leg == "vanilla_16x8" else (N_STATE, N_MEAS)))
        d = synth.linear_batch(N, nn, pp, T, seed=synth.SEED + 77)
        if leg == "shared_model":
            for f in ("F", "H", "Q", "R"):
                d[f] = np.ascontiguousarray(np.broadcast_to(d[f][0], d[f].shape))
            b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"][0], None, d["H"][0], d["Q"][0], d["R"][0], nfilters=N)
            okind = orc.VANILLA
        else:
            kind, okind = (k.VANILLA, orc.VANILLA) if leg in ("vanilla_12x6", "vanilla_10x4", "vanilla_16x8") else (k.SQUAREROOT, orc.SQUAREROOT)
            b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
        y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
        for t in range(T):
            b.update_dev(y[t].data_ptr(), N)
        b.synchronize()
        xo, Po, nerr = orc.ldkf_batch(okind, d["x0"], d["P0"], d["F"], d["H"], d["Q"], d["R"], d["y"])
        ex, eP = synth.rel_frobenius(b.get(k.STATE), xo), synth.rel_frobenius(b.get(k.COVAR), Po)
        out.update({"max_rel_frobenius_state": ex, "max_rel_frobenius_covariance": eP, "tolerance": 1e-9,
                    "ok": bool(ex <= 1e-9 and eP <= 1e-9 and nerr == 0 and not b.status().any())})
        return out
    # the NLDKF legs: per-step Phi / Htilde handed over on the device (kb_prepare_dev), as the timed loops do
    fp32 = leg == "srif_fp32"
    n, p = (12, 6) if fp32 else (6, 2)
    tdt = torch.float32 if fp32 else torch.float64
    g = torch.Generator(device="cuda"); g.manual_seed(2016)
    rng = np.random.default_rng(99)
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10.0] * (n // 2) + [1.0] * (n - n // 2)
    if fp32:
        R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
        b = ga.FilterBatch(k.SRIF, n, p, 0, N, dtype=k.F32)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
        fs = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]
    else:
        R = np.broadcast_to(np.diag([1e-6, 1e-6]), (N, p, p)).copy()
        b = ga.FilterBatch(k.HYBRID, n, p, 0, N)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R[0], 2, p_rows=p); b.init(); b.enable_ekf()
        fs = [orc.Filter.hybrid(x0[i], P0[i], None, R[i], p) for i in range(N)]
        for f in fs:
            f.enable_ekf()
    # D(ii) as SURVEY 8d specifies it (R = 1e-6 against P0 = 10) is ILL-CONDITIONED: two correct fp64 evaluations of it drift ~1e-5 apart
    # on the worst filter.  The GATE of this leg is therefore the arbiter above fp64 (tests/golden/generated/hybrid_ekf_bench_6x2_hp.npz,
    # 60-digit restatement of hybrid.go:104-204 by tests/golden/make_highprec.py): engine error against the exact result <= 4 x the
    # oracle's error against it (`_hybrid_arbiter`); the 4096-filter distance to the oracle is reported next to it
    eye = torch.eye(n, dtype=tdt, device="cuda").reshape(n * n, 1)
    nerr = 0
    for t in range(T):
        Phi = (eye + 1e-2 * torch.randn(n * n, N, dtype=tdt, device="cuda", generator=g)).contiguous()
        Ht = torch.randn(p * n, N, dtype=tdt, device="cuda", generator=g)
        real = torch.randn(p, N, dtype=tdt, device="cuda", generator=g)
        comp = (real + (1e-2 if fp32 else 1e-3) * torch.randn(p, N, dtype=tdt, device="cuda", generator=g)).contiguous()
        torch.cuda.synchronize()   # the handle's stream does not wait for torch's: the arrays must be complete when they are handed over
        k.check(k.lib().kb_prepare_dev(b._h, Phi.data_ptr(), Ht.data_ptr(), N))
        k.check(k.lib().kb_update_nl_dev(b._h, real.data_ptr(), comp.data_ptr(), N))
        b.synchronize()
        Ph, Hh = Phi.double().cpu().numpy().T.reshape(N, n, n), Ht.double().cpu().numpy().T.reshape(N, p, n)
        rh, ch = real.double().cpu().numpy().T, comp.double().cpu().numpy().T
        for i, f in enumerate(fs):
            f.prepare(Ph[i], Hh[i])
            nerr += f.update_nl(rh[i], ch[i]) != orc.OK
    if fp32:
        eR = synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs]))
        eb = synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs]))
        out.update({"achieved_max_rel_frobenius_R": eR, "achieved_max_rel_frobenius_b": eb, "tolerance": SRIF_F32_TOL, "dtype": "f32 kernel against the fp64 oracle",
                    "ok": bool(eR <= SRIF_F32_TOL and eb <= SRIF_F32_TOL and nerr == 0 and not b.status().any())})
    else:
        xo, Po = np.array([f.state() for f in fs]), np.array([f.covariance() for f in fs])
        ex, eP = synth.rel_frobenius(b.get(k.STATE), xo), synth.rel_frobenius(b.get(k.COVAR), Po)
        arb = _hybrid_arbiter(ga, k)
        out.update({"max_rel_frobenius_state": ex, "max_rel_frobenius_covariance": eP,
                    "note": "4096 filters against the oracle: reported, not the gate (ill-conditioned: both are fp64 evaluations)",
                    "arbiter": arb, "tolerance": arb["rule"],
                    "ok": bool(arb["ok"] and ex <= 1e-3 and eP <= 1e-3 and nerr == 0 and not b.status().any())})
    return out


def _hybrid_arbiter(ga, k):
    """configs[3] D(ii) against the exact result: 64 filters x 20 steps of the bench generator's problem, engine (kb_prepare_dev +
    kb_update_nl_dev, the entry points the leg times) and oracle against tests/golden/generated/hybrid_ekf_bench_6x2_hp.npz."""
    import numpy as np
    import torch
    from oracle import oracle as orc
    from tests import highprec as hp
    z = hp.load("hybrid_ekf_bench_6x2")
    T, N = z["Phi"].shape[:2]
    n, p = 6, 2
    b = ga.FilterBatch(k.HYBRID, n, p, 0, N)
    b.set(k.X, z["x0"], 1); b.set(k.P, z["P0"], 2); b.set(k.R, z["R"], 2, p_rows=p); b.init(); b.enable_ekf()
    ex, eP = np.zeros((T, N)), np.zeros((T, N))
    for t in range(T):
        dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (z["Phi"][t].reshape(N, -1).T, z["Ht"][t].reshape(N, -1).T, z["real"][t].T, z["comp"][t].T)]
        torch.cuda.synchronize()
        k.check(k.lib().kb_prepare_dev(b._h, dev[0].data_ptr(), dev[1].data_ptr(), N))
        k.check(k.lib().kb_update_nl_dev(b._h, dev[2].data_ptr(), dev[3].data_ptr(), N))
        b.synchronize()
        ex[t], eP[t] = hp.rel_err(b.get(k.STATE), z["x"][t]), hp.rel_err(b.get(k.COVAR), z["P"][t])
    xo, Po = hp.oracle_hybrid(orc, z)
    ox = np.array([hp.rel_err(xo[t], z["x"][t]) for t in range(T)])
    oP = np.array([hp.rel_err(Po[t], z["P"][t]) for t in range(T)])
    ok = hp.passes(ex.max(axis=0), ox.max(axis=0)) and hp.passes(eP.max(axis=0), oP.max(axis=0)) and not b.status().any()
    return {"filters": N, "steps": T, "exact": "tests/golden/generated/hybrid_ekf_bench_6x2_hp.npz (mpmath, 60 digits)",
            "engine_vs_exact": {"state": hp.summary(ex.max(axis=0)), "covariance": hp.summary(eP.max(axis=0))},
            "oracle_vs_exact": {"state": hp.summary(ox.max(axis=0)), "covariance": hp.summary(oP.max(axis=0))},
            "rule": "engine error <= %g x oracle error + %g, worst and median filter (worst step of each)" % (hp.FACTOR, hp.FLOOR), "ok": bool(ok)}


def _hybrid_fused_parity(ga, k):
    """kb_update_nl_steps_dev on the Hybrid EKF 6/2 batch against T single Prepare + Update calls on the same arrays (4096 filters x 10
    steps): the same bits; and both against the oracle on a well-conditioned variant of the problem (R = 1e-3) at 1e-9."""
    import numpy as np
    import torch
    from oracle import oracle as orc
    from gokalman_amd import synth
    N, T, n, p = 4096, 10, 6, 2
    rng = np.random.default_rng(321)
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10, 10, 10, 1, 1, 1]
    Rm = np.diag([1e-3, 1e-3])
    g = torch.Generator(device="cuda"); g.manual_seed(19)
    Phi = (torch.eye(n, dtype=torch.float64, device="cuda").reshape(1, n * n, 1) + 1e-2 * torch.randn(T, n * n, N, dtype=torch.float64, device="cuda", generator=g)).contiguous()
    Ht = torch.randn(T, p * n, N, dtype=torch.float64, device="cuda", generator=g)
    real = torch.randn(T, p, N, dtype=torch.float64, device="cuda", generator=g)
    comp = (real + 1e-3 * torch.randn(T, p, N, dtype=torch.float64, device="cuda", generator=g)).contiguous()
    torch.cuda.synchronize()
    outs = []
    for fused in (True, False):
        b = ga.FilterBatch(k.HYBRID, n, p, 0, N)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, Rm, 2, p_rows=p); b.init(); b.enable_ekf()
        if fused:
            b.update_nl_steps_dev(Phi.data_ptr(), Ht.data_ptr(), N, n * n * N, p * n * N, real.data_ptr(), comp.data_ptr(), N, p * N, T)
            kernel = b.last_kernel()
        else:
            for t in range(T):
                k.check(k.lib().kb_prepare_dev(b._h, Phi[t].data_ptr(), Ht[t].data_ptr(), N))
                k.check(k.lib().kb_update_nl_dev(b._h, real[t].data_ptr(), comp[t].data_ptr(), N))
        b.synchronize()
        outs.append((b.get(k.STATE), b.get(k.COVAR), int(np.count_nonzero(b.status()))))
    same = bool(np.array_equal(outs[0][0].view(np.uint64), outs[1][0].view(np.uint64)) and np.array_equal(outs[0][1].view(np.uint64), outs[1][1].view(np.uint64)))
    M = 256
    Ph, Hh = Phi[:, :, :M].cpu().numpy(), Ht[:, :, :M].cpu().numpy()
    rh, ch = real[:, :, :M].cpu().numpy(), comp[:, :, :M].cpu().numpy()
    xo, Po, nerr = [], [], 0
    for i in range(M):
        f = orc.Filter.hybrid(x0[i], P0[i], None, Rm, p); f.enable_ekf()
        for t in range(T):
            f.prepare(Ph[t, :, i].reshape(n, n), Hh[t, :, i].reshape(p, n))
            nerr += f.update_nl(rh[t, :, i], ch[t, :, i]) != orc.OK
        xo.append(f.state()); Po.append(f.covariance())
    ex, eP = synth.rel_frobenius(outs[0][0][:M], np.array(xo)), synth.rel_frobenius(outs[0][1][:M], np.array(Po))
    return {"filters": N, "steps": T, "kernel": kernel, "bit_identical_to_single_steps": same, "oracle_filters": M, "max_rel_frobenius_state": ex,
            "max_rel_frobenius_covariance": eP, "tolerance": 1e-9,
            "ok": bool(same and ex <= 1e-9 and eP <= 1e-9 and nerr == 0 and outs[0][2] == 0 and "fused" in kernel)}


def _srif_fused_parity(ga, k):
    """kb_update_nl_steps_dev against T single Prepare + Update calls on the same arrays (4096 filters x 10 steps): the same bits; and the
    fp32 result against the fp64 oracle within SRIF_F32_TOL."""
    import numpy as np
    import torch
    from oracle import oracle as orc
    N, T, n, p = 4096, 10, 12, 6
    rng = np.random.default_rng(123)
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10.0] * 6 + [1.0] * 6
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    g = torch.Generator(device="cuda"); g.manual_seed(17)
    Phi = (torch.eye(n, dtype=torch.float32, device="cuda").reshape(1, n * n, 1) + 1e-2 * torch.randn(T, n * n, N, dtype=torch.float32, device="cuda", generator=g)).contiguous()
    Ht = torch.randn(T, p * n, N, dtype=torch.float32, device="cuda", generator=g)
    real = torch.randn(T, p, N, dtype=torch.float32, device="cuda", generator=g)
    comp = (real + 1e-2 * torch.randn(T, p, N, dtype=torch.float32, device="cuda", generator=g)).contiguous()
    torch.cuda.synchronize()
    outs = []
    for fused in (True, False):
        b = ga.FilterBatch(k.SRIF, n, p, 0, N, dtype=k.F32)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
        if fused:
            b.update_nl_steps_dev(Phi.data_ptr(), Ht.data_ptr(), N, n * n * N, p * n * N, real.data_ptr(), comp.data_ptr(), N, p * N, T)
            kernel = b.last_kernel()
        else:
            for t in range(T):
                k.check(k.lib().kb_prepare_dev(b._h, Phi[t].data_ptr(), Ht[t].data_ptr(), N))
                k.check(k.lib().kb_update_nl_dev(b._h, real[t].data_ptr(), comp[t].data_ptr(), N))
        b.synchronize()
        outs.append((b.get(k.RAW_MAT), b.get(k.RAW_VEC), int(np.count_nonzero(b.status()))))
    same = bool(np.array_equal(outs[0][0].view(np.uint64), outs[1][0].view(np.uint64)) and np.array_equal(outs[0][1].view(np.uint64), outs[1][1].view(np.uint64)))
    M = 256
    fs = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(M)]
    Ph, Hh = Phi[:, :, :M].double().cpu().numpy(), Ht[:, :, :M].double().cpu().numpy()
    rh, ch = real[:, :, :M].double().cpu().numpy(), comp[:, :, :M].double().cpu().numpy()
    nerr = 0
    for t in range(T):
        for i, f in enumerate(fs):
            f.prepare(Ph[t, :, i].reshape(n, n), Hh[t, :, i].reshape(p, n))
            nerr += f.update_nl(rh[t, :, i], ch[t, :, i]) != orc.OK
    from gokalman_amd import synth
    eR = synth.rel_frobenius(outs[0][0][:M], np.array([f.raw_mat() for f in fs]))
    eb = synth.rel_frobenius(outs[0][1][:M], np.array([f.raw_vec() for f in fs]))
    return {"filters": N, "steps": T, "kernel": kernel, "bit_identical_to_single_steps": same, "oracle_filters": M, "achieved_max_rel_frobenius_R": eR,
            "achieved_max_rel_frobenius_b": eb, "tolerance": SRIF_F32_TOL,
            "ok": bool(same and eR <= SRIF_F32_TOL and eb <= SRIF_F32_TOL and nerr == 0 and outs[0][2] == 0 and "fused" in kernel)}


SRIF_F32_TOL = 2e-5   # tests/test_srif_gpu.py uses the same figure (achieved: ~1.3e-6 on R, ~2.7e-6 on b)


STATOD = dict(  # examples/statOD5044/main.go:36-57
    F=[[1, 0.1, 0, 7.726e-2], [4.015e-7, 1, 0, 1.545], [-2.319e-16, -1.732e-9, 1, 0.1], [-6.956e-15, -3.465e-8, 0, 1]],
    G=[[5e-3, 3.85e-7], [0.1, 1.157e-5], [-5.775e-11, 7.487e-7], [1.732e-9, 1.498e-5]],
    H=[[1.0, 0, 0, 0], [0, 0, 1, 0]],
    Q=[[6.669e-16, 1.001e-14, 3.823e-19, 5.150e-18], [1.001e-14, 2.002e-13, 1.030e-17, 1.545e-16],
       [3.862e-19, 1.030e-17, 6.667e-19, 1.000e-17], [5.150e-18, 1.545e-16, 1.000e-17, 2.000e-16]],
    R=[[2e-2, 0], [0, 2e-4]], x0=[2, 0.5, 0, 0.0], P0=[[5, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0.01, 0], [0, 0, 0, 1e-5]])


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and world > ndev:
        raise SystemExit("%d ranks but %d GPUs: RCCL needs one GPU per rank" % (world, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    use_dist = world > 1 or args.init_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)

    import gokalman_amd as ga
    from gokalman_amd import _capi as k
    from gokalman_amd import dist as kd
    from gokalman_amd import roofline as rl
    from gokalman_amd import synth

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if not use_dist:
            return seconds, [seconds]
        t = torch.tensor([seconds], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank = [float(e.item()) for e in every]
        return max(per_rank), per_rank

    N = args.filters
    n, p = N_STATE, N_MEAS
    POOL = 4  # distinct measurement sets cycled through the timed steps
    d = synth.linear_batch(N, n, p, POOL, seed=synth.SEED + rank)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], device=local_rank)
    y_dev = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).to(dev)  # planar [POOL][p][N]
    ptrs = [y_dev[t].data_ptr() for t in range(POOL)]
    kstream = torch.cuda.ExternalStream(b.stream(), device=dev)

    def timed_steps(batch, stream, pointers, ld, steps, warmup):
        """warmup untimed steps, then EXACTLY `steps` launches bracketed by barrier + synchronize; returns
        (wall seconds of this rank, HIP-event ms per launch on the kernel's stream)."""
        for t in range(warmup):
            batch.update_dev(pointers[t % len(pointers)], ld)
        batch.synchronize()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for t in range(steps):
            batch.update_dev(pointers[t % len(pointers)], ld)
        ev1.record(stream)
        batch.synchronize()
        torch.cuda.synchronize()
        local = time.perf_counter() - t0
        barrier()
        return local, ev0.elapsed_time(ev1) / steps

    def warm_clocks(ms=60.0):
        """Untimed load before a timed region: after the host-side set-up the GPU sits at idle clocks and needs ~25 ms of
        sustained work to come back (a 256k-filter SRIF step measured 124 us in the first 25 steps of a cold process, 86 us
        from step 250 on).  Runs the resident headline batch; never inside a timed region."""
        for t in range(max(1, int(ms / 0.17 * (1 << 20) / max(N, 1 << 14)))):
            b.update_dev(ptrs[t % POOL], N)
        b.synchronize()

    warm_clocks()
    local_s, kernel_ms = timed_steps(b, kstream, ptrs, N, args.steps, args.warmup)
    wall_s, per_rank_s = max_over_ranks(local_s)
    blocks_wall, blocks_kernel = [wall_s], [kernel_ms]
    for _ in range(max(0, args.repeat - 1)):   # the same timed block again (value stays the first one's)
        ls, km = timed_steps(b, kstream, ptrs, N, args.steps, 0)
        ws, _ = max_over_ranks(ls)
        blocks_wall.append(ws); blocks_kernel.append(km)
    nbad = int(np.count_nonzero(b.status()))

    # ---- strong scaling: --filters filters in total, GPU g owns [g N / G, (g + 1) N / G) -----------------------
    strong = None
    if world > 1:
        lo, hi = kd.shard_range(N, rank, world)
        ds = synth.linear_batch(hi - lo, n, p, POOL, seed=synth.SEED + 7000 + rank)
        bs = ga.FilterBatch.new_ldkf(k.VANILLA, ds["x0"], ds["P0"], ds["F"], None, ds["H"], ds["Q"], ds["R"], device=local_rank)
        ys = torch.from_numpy(np.ascontiguousarray(ds["y"].transpose(0, 2, 1))).to(dev)
        sptrs = [ys[t].data_ptr() for t in range(POOL)]
        warm_clocks()
        s_local, s_kernel_ms = timed_steps(bs, torch.cuda.ExternalStream(bs.stream(), device=dev), sptrs, hi - lo, args.steps, args.warmup)
        s_wall, s_per_rank = max_over_ranks(s_local)
        strong = {"filters_total": N, "filters_per_gpu": [kd.shard_range(N, r, world)[1] - kd.shard_range(N, r, world)[0] for r in range(world)],
                  "ms_per_step": s_wall / args.steps * 1e3, "value": N * args.steps / s_wall, "unit": "filter-update steps/s (whole job)",
                  "kernel_ms_rank0": s_kernel_ms, "per_rank_ms_per_step": [v / args.steps * 1e3 for v in s_per_rank]}
        del bs, ys

    # ---- epilogue collective: what every rank did, summed over RCCL (the update path itself has no exchange) --------
    counts = torch.tensor([float(N) * args.steps, float(nbad), 1.0], dtype=torch.float64, device=coll_dev)
    if use_dist:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    total_filter_steps, total_bad, ranks_seen = float(counts[0].item()), int(counts[1].item()), int(counts[2].item())

    # ---- extra: the caller loop fused into one launch (x, P, model resident in registers) ---------------------------
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
        vk, vs = rl.load_valu(ROOT)
        if "vanilla_fused" in vk:
            fused["roofline"] = rl.valu_roofline(fms, (N + 63) // 64, vk["vanilla_fused"]["valu_insts_per_wave_per_step"] * T, vs)
        elif vs:
            fused["roofline"] = {"bound": "valu_issue", "frac": None, "source": vs}
        # the same loop with the Noise drawn inside the launch (AWGN, noise.go:109-164; round 5): bit-identical to T single steps while no filter fails
        if rank == 0:
            ba = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], device=local_rank,
                                         noise=k.NOISE_AWGN, seed=2016)
            ba.update_steps_dev(yy.data_ptr(), N, T)
            ba.synchronize()
            warm_clocks()
            sa = torch.cuda.ExternalStream(ba.stream())
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(sa)
            for _ in range(3):
                ba.update_steps_dev(yy.data_ptr(), N, T)
            a1.record(sa)
            ba.synchronize()
            ams = a0.elapsed_time(a1) / 3
            fused["awgn"] = {"steps_per_launch": T, "ms_per_launch": ams, "value": N * T / (ams * 1e-3), "errors": int(np.count_nonzero(ba.status())),
                             "unit": "filter-update steps/s (1 GPU, kb_update_steps_dev, AWGN drawn in the kernel)"}
            if "vanilla_fused_awgn" in vk:   # its own instantiation, its own VALU count (summarise_round.py selects by the full argument list)
                fused["awgn"]["roofline"] = rl.valu_roofline(ams, (N + 63) // 64, vk["vanilla_fused_awgn"]["valu_insts_per_wave_per_step"] * T, vs)
            del ba
        del yy
    torch.cuda.empty_cache()   # (the headline batch stays: warm_clocks() runs it before every later timed region)

    # ---- extra: the same kernel with the state block far outside the 256 MiB Infinity Cache -------------------------
    ooc = None
    if args.ooc_filters > N and rank == 0:
        M = args.ooc_filters
        reps_n = (M + N - 1) // N
        tile = lambda a: np.ascontiguousarray(np.concatenate([a] * reps_n, axis=0)[:M])
        b2 = ga.FilterBatch.new_ldkf(k.VANILLA, tile(d["x0"]), tile(d["P0"]), tile(d["F"]), None, tile(d["H"]),
                                     tile(d["Q"]), tile(d["R"]), device=local_rank)
        y2 = torch.from_numpy(np.ascontiguousarray(np.concatenate([d["y"][0]] * reps_n, axis=0)[:M].T)).to(dev)  # [p][M]
        s2 = torch.cuda.ExternalStream(b2.stream(), device=dev)
        warm_clocks()
        for _ in range(5):
            b2.update_dev(y2.data_ptr(), M)
        b2.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K2 = 20
        e0.record(s2)
        for _ in range(K2):
            b2.update_dev(y2.data_ptr(), M)
        e1.record(s2)
        b2.synchronize()
        ms2 = e0.elapsed_time(e1) / K2
        moved = rl.moved_bytes("vanilla", n, p)
        ooc = {"filters": M, "state_block_MB": M * (n + rl.tri(n)) * 8 / 1e6, "kernel_ms": ms2,
               "value": M / (ms2 * 1e-3), "unit": "filter-update steps/s (1 GPU)",
               "achieved": moved * M / (ms2 * 1e-3) / 1e9, "frac": moved * M / (ms2 * 1e-3) / 1e9 / rl.HBM_PEAK_GBPS,
               "frac_of_achievable": moved * M / (ms2 * 1e-3) / 1e9 / rl.HBM_ACHIEVABLE_GBPS,
               "note": "same kernel, state block >> 256 MiB Infinity Cache: every byte comes from / goes to HBM"}
        del b2, y2
        torch.cuda.empty_cache()
    barrier()

    # ---- extra: config D(i), montecarlo.go sharded by run index; the ONE collective of the path (montecarlo.go:18-59) --
    extra = {}
    if args.mc_runs > 0:
        s = {kk: np.array(v, dtype=np.float64) for kk, v in STATOD.items()}
        kf = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, s["x0"], s["P0"], s["F"], s["G"], s["H"], s["Q"], s["R"],
                                     nfilters=args.mc_runs, device=local_rank, noise=k.NOISE_AWGN, seed=2016)
        first = rank * args.mc_runs  # a run's noise depends only on its global index
        ga.new_monte_carlo_runs(args.mc_runs, 4, 2, np.zeros((1, 2)), kf, first_run=first)  # warm-up
        warm_clocks()
        barrier()
        t0 = time.perf_counter()
        mc = ga.new_monte_carlo_runs(args.mc_runs * world, args.mc_steps, 2, np.zeros((1, 2)), kf, first_run=first,
                                     reduce=kd.allreduce_sum if use_dist else None)
        torch.cuda.synchronize()
        mc_s, _ = max_over_ranks(time.perf_counter() - t0)
        vkernels, vsrc = rl.load_valu(ROOT)
        valu = {"kernels": vkernels}
        extra["mc"] = {"config": "configs[3] D(i): montecarlo.go pure-predictor statOD5044 (n=4, AWGN), runs sharded by global index",
                       "runs_total": world * args.mc_runs, "steps": args.mc_steps, "seconds": mc_s,
                       "value": world * args.mc_runs * args.mc_steps / mc_s, "unit": "run-steps/s (whole job)",
                       "collective": "all_reduce(SUM) of %d doubles over %s" % (args.mc_steps * 2 * 4,
                                                                                  "RCCL" if args.dist_backend == "nccl" and use_dist else
                                                                                  (args.dist_backend if use_dist else "one rank (identity)")),
                       "stddev_last": mc.stddev(args.mc_steps - 1).tolist()}
        e = valu.get("kernels", {}).get("mc")
        if e:   # VALU-issue roofline of mc_kernel (the reduction's host side and the all-reduce are in `seconds`, not in the floor)
            per_step = e["valu_insts_per_wave"] / float(e.get("steps_per_launch", 1086))
            extra["mc"]["roofline"] = rl.valu_roofline(mc_s * 1e3, (args.mc_runs + 63) // 64, per_step * args.mc_steps, vsrc)
            extra["mc"]["roofline"]["kernel_ms_is"] = "the whole NewMonteCarloRuns call (mc_kernel + fold + D2H of the sums + host object), not the kernel alone"
        # ---- configs[3] at its stated size: the 8 388 608-run ensemble, mc_total / world runs per rank as consecutive shards of
        # mc_runs runs (fresh batches: the noise of a run depends only on (seed, global run index, step)), partial sums added in
        # shard order, ONE all-reduce -- the reduction the 8-GPU job performs, rehearsed on however many GPUs there are
        if args.mc_total >= world * args.mc_runs and args.mc_total % (world * args.mc_runs) == 0:
            S = args.mc_total // (world * args.mc_runs)
            shards = [ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, s["x0"], s["P0"], s["F"], s["G"], s["H"], s["Q"], s["R"],
                                              nfilters=args.mc_runs, device=local_rank, noise=k.NOISE_AWGN, seed=2016) for _ in range(S)]
            warm_clocks()
            barrier()
            t0 = time.perf_counter()
            tot = None
            for si, kb_s in enumerate(shards):
                part = ga.new_monte_carlo_runs(args.mc_runs, args.mc_steps, 2, np.zeros((1, 2)), kb_s, first_run=(rank * S + si) * args.mc_runs, keep_runs=False).sums
                if tot is None:
                    tot = part.copy()
                else:
                    tot[:, :2, :] += part[:, :2, :]
            if use_dist:
                tot[:, :2, :] = kd.allreduce_sum(np.ascontiguousarray(tot[:, :2, :]))
            torch.cuda.synchronize()
            ens_s, _ = max_over_ranks(time.perf_counter() - t0)
            ens = ga.MonteCarloRuns(args.mc_total, args.mc_steps, 4, tot)
            # montecarlo.go:18-59 against what it estimates: Mean(k) = F^k x0, StdDev(k)^2 = diag(sum_j F^j Q F^jT), within 6 standard errors
            xk, Pk, ok_mc = s["x0"].copy(), np.zeros((4, 4)), True
            for t in range(args.mc_steps):
                xk, Pk = s["F"] @ xk, s["F"] @ Pk @ s["F"].T + s["Q"]
                if t in (0, 10, 100, args.mc_steps - 1):
                    sd = np.sqrt(np.diag(Pk))
                    ok_mc = ok_mc and bool(np.all(np.abs(ens.mean(t) - xk) <= 6 * sd / np.sqrt(args.mc_total) + 1e-12 * np.abs(xk)))
                    ok_mc = ok_mc and bool(np.all(np.abs(ens.stddev(t) / sd - 1.0) <= 6 / np.sqrt(2 * args.mc_total)))
            extra["mc"]["ensemble"] = {"config": "configs[3] at its stated size: %d runs x %d steps = %d shard(s) of %d runs per rank, sums added in shard order + one all-reduce"
                                                 % (args.mc_total, args.mc_steps, S, args.mc_runs),
                                       "runs_total": args.mc_total, "shards_per_rank": S, "seconds": ens_s,
                                       "value": args.mc_total * args.mc_steps / ens_s, "unit": "run-steps/s (whole job)",
                                       "matches_covariance_recursion": ok_mc, "stddev_last": ens.stddev(args.mc_steps - 1).tolist()}
            del shards
        elif rank == 0:
            print("bench: extra.mc.ensemble skipped: --mc-total %d is not a multiple of world x --mc-runs = %d x %d"
                  % (args.mc_total, world, args.mc_runs), file=sys.stderr, flush=True)
        # ---- chi-square on the same ensemble (chisquare.go:16-95): NIS / NEES sums all-reduced like the means ----
        if args.chisq_runs > 0:
            R = min(args.chisq_runs, args.mc_runs)
            truth = kf if R == args.mc_runs else ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, s["x0"], s["P0"], s["F"], s["G"], s["H"], s["Q"], s["R"],
                                                                         nfilters=R, device=local_rank, noise=k.NOISE_AWGN, seed=2016)
            ckf = ga.FilterBatch.new_ldkf(k.VANILLA, s["x0"], s["P0"], s["F"], s["G"], s["H"], s["Q"], s["R"], nfilters=R, device=local_rank)
            ga.new_chi_square(ckf, truth, np.zeros((1, 2)), steps=8, first_run=rank * R)   # warm-up
            warm_clocks()
            barrier()
            t0 = time.perf_counter()
            nis, nees = ga.new_chi_square(ckf, truth, np.zeros((1, 2)), steps=args.mc_steps, first_run=rank * R, total_runs=R * world,
                                          reduce=kd.allreduce_sum if use_dist else None)
            torch.cuda.synchronize()
            c_s, _ = max_over_ranks(time.perf_counter() - t0)
            extra["chisq"] = {"config": "NewChiSquare on the statOD5044 ensemble: truth + Vanilla filter + NEES / NIS fused in one launch, runs sharded",
                              "runs_total": world * R, "steps": args.mc_steps, "seconds": c_s, "value": world * R * args.mc_steps / c_s,
                              "unit": "run-steps/s (whole job)", "nis_mean": float(np.mean(nis)), "nees_mean": float(np.mean(nees))}
            e = valu.get("kernels", {}).get("chisq")
            if e:
                per_step = e["valu_insts_per_wave"] / float(e.get("steps_per_launch", 1086))
                extra["chisq"]["roofline"] = rl.valu_roofline(c_s * 1e3, (R + 63) // 64, per_step * args.mc_steps, vsrc)
                extra["chisq"]["roofline"]["kernel_ms_is"] = "the whole NewChiSquare call (replicate kf, chisq_kernel, fold, D2H), not the kernel alone"
            del ckf
        del kf
    # ---- extra: config D(ii), Hybrid EKF ensemble sharded the same way ---------------------------------------------
    if args.hybrid_filters > 0:
        M = args.hybrid_filters
        hn, hp = 6, 2
        g = torch.Generator(device=dev)
        g.manual_seed(7 + rank)
        x0 = np.random.default_rng(6 + rank).standard_normal((M, hn))
        P0 = np.zeros((M, hn, hn))
        P0[:, np.arange(hn), np.arange(hn)] = [10, 10, 10, 1, 1, 1]  # hybrid_test.go:174-180
        hb = ga.FilterBatch(k.HYBRID, hn, hp, 0, M, device=local_rank)
        hb.set(k.X, x0, 1); hb.set(k.P, P0, 2); hb.set(k.R, np.diag([1e-6, 1e-6]), 2, p_rows=hp); hb.init(); hb.enable_ekf()
        Phi = (torch.eye(hn, dtype=torch.float64, device=dev).reshape(hn * hn, 1)
               + 1e-2 * torch.randn(hn * hn, M, dtype=torch.float64, device=dev, generator=g)).contiguous()
        Ht = torch.randn(hp * hn, M, dtype=torch.float64, device=dev, generator=g)
        real = torch.randn(hp, M, dtype=torch.float64, device=dev, generator=g)
        comp = real + 1e-3 * torch.randn(hp, M, dtype=torch.float64, device=dev, generator=g)
        torch.cuda.synchronize()   # (the handle's stream does not wait for torch's)
        hs = torch.cuda.ExternalStream(hb.stream(), device=dev)

        def hstep():
            k.check(k.lib().kb_prepare_dev(hb._h, Phi.data_ptr(), Ht.data_ptr(), M))
            k.check(k.lib().kb_update_nl_dev(hb._h, real.data_ptr(), comp.data_ptr(), M))
        warm_clocks()
        for _ in range(10):
            hstep()
        hb.synchronize()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K3 = 50
        t0 = time.perf_counter()
        e0.record(hs)
        for _ in range(K3):
            hstep()
        e1.record(hs)
        hb.synchronize()
        h_s, _ = max_over_ranks(time.perf_counter() - t0)
        hms = e0.elapsed_time(e1) / K3
        hbad = torch.tensor([float(np.count_nonzero(hb.status()))], dtype=torch.float64, device=coll_dev)
        if use_dist:
            dist.all_reduce(hbad, op=dist.ReduceOp.SUM)
        extra["hybrid_ekf"] = {"config": "configs[3] D(ii): HybridKF EKF 6/2 fp64 ensemble, per-step Phi/Htilde read in place, filters sharded",
                               "filters_total": world * M, "steps": K3, "value": world * M * K3 / h_s,
                               "unit": "filter-update steps/s (whole job)", "kernel_ms": hms,
                               "roofline": rl.hbm_roofline(hms, M, rl.algorithmic_bytes("hybrid", hn, hp), rl.moved_bytes("hybrid", hn, hp),
                                                           *rl.load_traffic(ROOT, "hybrid_reg_kernel<double, 6, 2, true, false, true")),
                               "filters_with_error_status": int(hbad.item())}
        if args.fused_steps > 0 and rank == 0:
            # D(ii) with the caller loop `for k { Prepare(Phi_k, Htilde_k); Update(real_k, computed_k) }` inside ONE launch (round 6, kb_update_nl_steps_dev,
            # kb_hybrid_fused.hip): distinct Phi / Htilde / observations per step, x and P resident in registers between the steps; the same bits as T single calls
            TF = 10
            PhiT = (torch.eye(hn, dtype=torch.float64, device=dev).reshape(1, hn * hn, 1)
                    + 1e-2 * torch.randn(TF, hn * hn, M, dtype=torch.float64, device=dev, generator=g)).contiguous()
            HtT = torch.randn(TF, hp * hn, M, dtype=torch.float64, device=dev, generator=g)
            realT = torch.randn(TF, hp, M, dtype=torch.float64, device=dev, generator=g)
            compT = realT + 1e-3 * torch.randn(TF, hp, M, dtype=torch.float64, device=dev, generator=g)
            torch.cuda.synchronize()

            def hfstep():
                hb.update_nl_steps_dev(PhiT.data_ptr(), HtT.data_ptr(), M, hn * hn * M, hp * hn * M, realT.data_ptr(), compT.data_ptr(), M, hp * M, TF)
            warm_clocks()
            for _ in range(3):
                hfstep()
            hb.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record(hs)
            for _ in range(5):
                hfstep()
            g1.record(hs)
            hb.synchronize()
            hfms = g0.elapsed_time(g1) / 5
            moved_hf = rl.moved_bytes("hybrid", hn, hp) - 8 * (hn + rl.tri(hn)) - 8 * rl.tri(hp)   # x, P are not re-read, R is read once per launch
            extra["hybrid_ekf"]["fused"] = {"config": "the caller loop inside one launch (kb_update_nl_steps_dev): %d steps per launch, distinct Phi / Htilde / observations per step" % TF,
                                            "steps_per_launch": TF, "ms_per_launch": hfms, "kernel_ms": hfms / TF, "value": M * TF / (hfms * 1e-3),
                                            "unit": "filter-update steps/s (1 GPU, kb_update_nl_steps_dev)", "kernel": hb.last_kernel(),
                                            "filters_with_error_status": int(np.count_nonzero(hb.status())),
                                            "roofline": rl.hbm_roofline(hfms / TF, M, rl.algorithmic_bytes("hybrid", hn, hp), moved_hf, None,
                                                                        {"live": False, "analytic": "packed working set less the resident x, P and R"})}
            del PhiT, HtT, realT, compT
        del hb, Phi, Ht, real, comp
        # ---- configs[3] D(ii) at its stated size on ONE GPU: world x M filters = 8 388 608 by default, the
        # same generator; the first 4096 filters must come out bit-identical to the same filters run as a 4096-filter batch (a filter's
        # result may not depend on where in the grid it was computed), nobody may fail
        if world == 1 and args.hybrid_total > M:
            MT = args.hybrid_total
            g2 = torch.Generator(device=dev); g2.manual_seed(11)
            x0t = np.random.default_rng(12).standard_normal((MT, hn))
            P0t = np.zeros((MT, hn, hn)); P0t[:, np.arange(hn), np.arange(hn)] = [10, 10, 10, 1, 1, 1]
            big = ga.FilterBatch(k.HYBRID, hn, hp, 0, MT, device=local_rank)
            big.set(k.X, x0t, 1); big.set(k.P, P0t, 2); big.set(k.R, np.diag([1e-6, 1e-6]), 2, p_rows=hp); big.init(); big.enable_ekf()
            SM = 4096
            small = ga.FilterBatch(k.HYBRID, hn, hp, 0, SM, device=local_rank)
            small.set(k.X, x0t[:SM], 1); small.set(k.P, P0t[:SM], 2); small.set(k.R, np.diag([1e-6, 1e-6]), 2, p_rows=hp); small.init(); small.enable_ekf()
            del x0t, P0t
            PhiT = (torch.eye(hn, dtype=torch.float64, device=dev).reshape(hn * hn, 1)
                    + 1e-2 * torch.randn(hn * hn, MT, dtype=torch.float64, device=dev, generator=g2)).contiguous()
            HtT = torch.randn(hp * hn, MT, dtype=torch.float64, device=dev, generator=g2)
            realT = torch.randn(hp, MT, dtype=torch.float64, device=dev, generator=g2)
            compT = realT + 1e-3 * torch.randn(hp, MT, dtype=torch.float64, device=dev, generator=g2)
            torch.cuda.synchronize()
            bs = torch.cuda.ExternalStream(big.stream(), device=dev)

            def bstep(b_, ld):
                k.check(k.lib().kb_prepare_dev(b_._h, PhiT.data_ptr(), HtT.data_ptr(), ld))
                k.check(k.lib().kb_update_nl_dev(b_._h, realT.data_ptr(), compT.data_ptr(), ld))
            KT = 10
            for _ in range(KT):   # (the same steps on both batches: the timed repetitions below continue from here)
                bstep(big, MT); bstep(small, MT)
            big.synchronize(); small.synchronize()
            same = bool(np.array_equal(big.get(k.STATE, 0, SM), small.get(k.STATE)) and np.array_equal(big.get(k.COVAR, 0, SM), small.get(k.COVAR)))
            warm_clocks()
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record(bs)
            for _ in range(KT):
                bstep(big, MT)
            b1.record(bs)
            big.synchronize()
            bms = b0.elapsed_time(b1) / KT
            extra["hybrid_ekf"]["ensemble"] = {"config": "configs[3] D(ii) at its stated size on one GPU: %d Hybrid EKF filters, %.1f GB of Phi / Htilde / observations per step read in place" % (MT, MT * (hn * hn + hp * hn + 2 * hp) * 8 / 1e9),
                                               "filters_total": MT, "steps": KT, "kernel_ms": bms, "value": MT / (bms * 1e-3), "unit": "filter-update steps/s (1 GPU)",
                                               "roofline": rl.hbm_roofline(bms, MT, rl.algorithmic_bytes("hybrid", hn, hp), rl.moved_bytes("hybrid", hn, hp), None, {"analytic": "packed working set"}),
                                               "first_4096_filters_bit_identical_to_a_4096_filter_batch": same,
                                               "filters_with_error_status": int(np.count_nonzero(big.status()))}
            if args.fused_steps > 0:
                # ... and with the caller loop inside one launch (kb_update_nl_steps_dev; the 1.8 GB state block is streamed: kb_hybrid_fused.hip takes the
                # non-temporal arm of the state policy): 4 steps per launch, distinct operands per step
                TFE = 4
                PhiE = (torch.eye(hn, dtype=torch.float64, device=dev).reshape(1, hn * hn, 1)
                        + 1e-2 * torch.randn(TFE, hn * hn, MT, dtype=torch.float64, device=dev, generator=g2)).contiguous()
                HtE = torch.randn(TFE, hp * hn, MT, dtype=torch.float64, device=dev, generator=g2)
                realE = torch.randn(TFE, hp, MT, dtype=torch.float64, device=dev, generator=g2)
                compE = realE + 1e-3 * torch.randn(TFE, hp, MT, dtype=torch.float64, device=dev, generator=g2)
                torch.cuda.synchronize()

                def efstep():
                    big.update_nl_steps_dev(PhiE.data_ptr(), HtE.data_ptr(), MT, hn * hn * MT, hp * hn * MT, realE.data_ptr(), compE.data_ptr(), MT, hp * MT, TFE)
                warm_clocks()
                efstep(); efstep()
                big.synchronize()
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record(bs)
                for _ in range(3):
                    efstep()
                c1.record(bs)
                big.synchronize()
                efms = c0.elapsed_time(c1) / 3
                moved_ef = rl.moved_bytes("hybrid", hn, hp) - 8 * (hn + rl.tri(hn)) - 8 * rl.tri(hp)
                extra["hybrid_ekf"]["ensemble"]["fused"] = {"steps_per_launch": TFE, "ms_per_launch": efms, "kernel_ms": efms / TFE, "value": MT * TFE / (efms * 1e-3),
                                                            "unit": "filter-update steps/s (1 GPU, kb_update_nl_steps_dev)", "kernel": big.last_kernel(),
                                                            "filters_with_error_status": int(np.count_nonzero(big.status())),
                                                            "roofline": rl.hbm_roofline(efms / TFE, MT, rl.algorithmic_bytes("hybrid", hn, hp), moved_ef, None,
                                                                                        {"live": False, "analytic": "packed working set less the resident x, P and R"})}
                del PhiE, HtE, realE, compE
            del big, small, PhiT, HtT, realT, compT
            torch.cuda.empty_cache()

    # ---- extra: configs C (SquareRoot 6/3 fp64 on the headline's batch) and E (SRIF 12/6 fp32), one shard per rank --------
    def timed_leg(batch, fn, reps):
        stream = torch.cuda.ExternalStream(batch.stream(), device=dev)
        warm_clocks()
        for _ in range(10):
            fn()
        batch.synchronize()
        barrier()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a0.record(stream)
        for _ in range(reps):
            fn()
        a1.record(stream)
        batch.synchronize()
        leg_s, _ = max_over_ranks(time.perf_counter() - t0)
        bad = torch.tensor([float(np.count_nonzero(batch.status()))], dtype=torch.float64, device=coll_dev)
        if use_dist:
            dist.all_reduce(bad, op=dist.ReduceOp.SUM)
        return leg_s, a0.elapsed_time(a1) / reps, int(bad.item())

    if args.sqrt_filters > 0:
        M = args.sqrt_filters
        dq = synth.linear_batch(M, n, p, 1, seed=synth.SEED + 1000 + rank)
        yq = torch.from_numpy(np.ascontiguousarray(dq["y"][0].T)).to(dev)  # [p][M]
        sq = ga.FilterBatch.new_ldkf(k.SQUAREROOT, dq["x0"], dq["P0"], dq["F"], None, dq["H"], dq["Q"], dq["R"], device=local_rank)
        K4 = 50
        q_s, qms, qbad = timed_leg(sq, lambda: sq.update_dev(yq.data_ptr(), M), K4)
        extra["squareroot"] = {"config": "configs[2] C: %d SquareRoot 6/3 fp64 filters per GPU, same synthetic batch as the headline" % M,
                               "filters_total": world * M, "steps": K4, "value": world * M * K4 / q_s,
                               "unit": "filter-update steps/s (whole job)", "kernel_ms": qms,
                               "roofline": rl.hbm_roofline(qms, M, rl.algorithmic_bytes("squareroot", n, p), rl.moved_bytes("squareroot", n, p),
                                                           *rl.load_traffic(ROOT, "squareroot_reg_kernel<double, 6, 3, 0, false")),
                               "filters_with_error_status": qbad}
        if args.fused_steps > 0 and rank == 0:
            # config C with the caller loop inside one launch (round 5): x, S and the model resident over T steps; Newton reciprocals, so not
            # PROMISED to be T launches' bits (held to 1e-12 of them and to 1e-9 of the oracle: `parity` below; achieved since round 6: the same bits)
            T = args.fused_steps
            yT = yq.unsqueeze(0).repeat(T, 1, 1).contiguous()   # [T][p][M] (the same measurement every step: throughput only)
            sq.update_steps_dev(yT.data_ptr(), M, T)
            sq.synchronize()
            warm_clocks()
            sstream = torch.cuda.ExternalStream(sq.stream())
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record(sstream)
            for _ in range(3):
                sq.update_steps_dev(yT.data_ptr(), M, T)
            f1.record(sstream)
            sq.synchronize()
            fq = f0.elapsed_time(f1) / 3
            extra["squareroot"]["fused"] = {"steps_per_launch": T, "ms_per_launch": fq, "value": M * T / (fq * 1e-3),
                                            "unit": "filter-update steps/s (1 GPU, kb_update_steps_dev)", "filters_with_error_status": int(np.count_nonzero(sq.status())),
                                            "config": "the SAME measurement repeated every step on a converged batch: throughput only (steady-state best case)"}
            del yT
        del sq, yq
    if args.shared_filters > 0:
        # one model for the whole batch (the reference's own shape of use: ONE filter object; kb_replicate / BatchLDKF / the ensembles):
        # the kernels read tile 0's model block out of the L2 and only state + measurements move -- NOT the headline config, whose
        # models are per filter
        M = args.shared_filters
        dq = synth.linear_batch(M, n, p, 1, seed=synth.SEED + 2000 + rank)
        yq = torch.from_numpy(np.ascontiguousarray(dq["y"][0].T)).to(dev)  # [p][M]
        sh = ga.FilterBatch.new_ldkf(k.VANILLA, dq["x0"], dq["P0"], dq["F"][0], None, dq["H"][0], dq["Q"][0], dq["R"][0], nfilters=M, device=local_rank)
        K5 = 50
        h_s, hms, hbad = timed_leg(sh, lambda: sh.update_dev(yq.data_ptr(), M), K5)
        moved = 8 * (2 * (n + rl.tri(n)) + p)
        extra["shared_model"] = {"config": "%d Vanilla 6/3 fp64 filters per GPU sharing ONE model (F, H, Q, R uploaded with broadcast = 1)" % M,
                                 "filters_total": world * M, "steps": K5, "value": world * M * K5 / h_s,
                                 "unit": "filter-update steps/s (whole job)", "kernel_ms": hms,
                                 "roofline": rl.hbm_roofline(hms, M, rl.algorithmic_bytes("vanilla", n, p), moved,
                                                             *rl.load_traffic(ROOT, "vanilla_reg_kernel<double, 6, 3, 0, false, false, false, false, false, true>")),
                                 "filters_with_error_status": hbad,
                                 "note": "the model block (672 B) is read from the L2 by every wave; moved bytes = x, P read and written + y"}
        del sh, yq
    if args.split_filters > 0:
        # Vanilla beyond 8 states (the north star's envelope is n <= 16): 12 / 6, per-filter models, one filter split over four lanes
        M = args.split_filters
        dq = synth.linear_batch(M, 12, 6, 1, seed=synth.SEED + 3000 + rank)
        yq = torch.from_numpy(np.ascontiguousarray(dq["y"][0].T)).to(dev)  # [p][M]
        vb = ga.FilterBatch.new_ldkf(k.VANILLA, dq["x0"], dq["P0"], dq["F"], None, dq["H"], dq["Q"], dq["R"], device=local_rank)
        K6 = 50
        v_s, vms, vbad = timed_leg(vb, lambda: vb.update_dev(yq.data_ptr(), M), K6)
        extra["vanilla_12x6"] = {"config": "%d Vanilla 12/6 fp64 filters per GPU, per-filter models (kb_vanilla_split.h: one filter per four lanes)" % M,
                                 "filters_total": world * M, "steps": K6, "value": world * M * K6 / v_s,
                                 "unit": "filter-update steps/s (whole job)", "kernel_ms": vms,
                                 "roofline": rl.hbm_roofline(vms, M, rl.algorithmic_bytes("vanilla", 12, 6), rl.moved_bytes("vanilla", 12, 6),
                                                             *rl.load_traffic(ROOT, "vanilla_split_kernel<double, 12, 6, 0, 4, false, false, false, false, false, 0")),
                                 "filters_with_error_status": vbad}
        del vb, yq
        # ... and SquareRoot at the same size (kb_squareroot_split.h: the Householder panels distributed by columns over four lanes)
        dq = synth.linear_batch(M, 12, 6, 1, seed=synth.SEED + 4000 + rank)
        yq = torch.from_numpy(np.ascontiguousarray(dq["y"][0].T)).to(dev)
        qb = ga.FilterBatch.new_ldkf(k.SQUAREROOT, dq["x0"], dq["P0"], dq["F"], None, dq["H"], dq["Q"], dq["R"], device=local_rank)
        q_s, qms2, qbad2 = timed_leg(qb, lambda: qb.update_dev(yq.data_ptr(), M), K6)
        extra["squareroot_12x6"] = {"config": "%d SquareRoot 12/6 fp64 filters per GPU, per-filter models (kb_squareroot_split.h: one filter per four lanes)" % M,
                                    "filters_total": world * M, "steps": K6, "value": world * M * K6 / q_s,
                                    "unit": "filter-update steps/s (whole job)", "kernel_ms": qms2,
                                    "roofline": rl.hbm_roofline(qms2, M, rl.algorithmic_bytes("squareroot", 12, 6), rl.moved_bytes("squareroot", 12, 6),
                                                                *rl.load_traffic(ROOT, "squareroot_split_kernel<double, 12, 6, 0, 4, false, false")),
                                    "filters_with_error_status": qbad2}
        del qb, yq
        # ... Information at the same size (kb_information_split.h: the pivoted LU solve distributed over four lanes)
        dq = synth.linear_batch(M, 12, 6, 1, seed=synth.SEED + 5000 + rank)
        yq = torch.from_numpy(np.ascontiguousarray(dq["y"][0].T)).to(dev)
        ib = ga.FilterBatch.new_ldkf(k.INFORMATION, dq["x0"], dq["P0"], dq["F"], None, dq["H"], dq["Q"], dq["R"], device=local_rank, flags=k.FLAG_INFO_FROM_STATE)
        i_s, ims, ibad = timed_leg(ib, lambda: ib.update_dev(yq.data_ptr(), M), K6)
        extra["information_12x6"] = {"config": "%d Information 12/6 fp64 filters per GPU, per-filter models (kb_information_split.h: one filter per four lanes)" % M,
                                     "filters_total": world * M, "steps": K6, "value": world * M * K6 / i_s,
                                     "unit": "filter-update steps/s (whole job)", "kernel_ms": ims,
                                     "roofline": rl.hbm_roofline(ims, M, rl.algorithmic_bytes("information", 12, 6), rl.moved_bytes("information", 12, 6),
                                                                 *rl.load_traffic(ROOT, "information_split_kernel<double, 12, 6, 0, 4, false")),
                                     "filters_with_error_status": ibad}
        del ib, yq
        # ... and a member of the padded family: Vanilla 10 / 4 on the 12-state / 4-measurement instantiation (run-time dimensions)
        dq = synth.linear_batch(M, 10, 4, 1, seed=synth.SEED + 6000 + rank)
        yq = torch.from_numpy(np.ascontiguousarray(dq["y"][0].T)).to(dev)
        pb = ga.FilterBatch.new_ldkf(k.VANILLA, dq["x0"], dq["P0"], dq["F"], None, dq["H"], dq["Q"], dq["R"], device=local_rank)
        p_s, pms, pbad = timed_leg(pb, lambda: pb.update_dev(yq.data_ptr(), M), K6)
        extra["vanilla_10x4"] = {"config": "%d Vanilla 10/4 fp64 filters per GPU, per-filter models (padded: vanilla_split_kernel<double, 12, 4, 2, 4, GEN>)" % M,
                                 "filters_total": world * M, "steps": K6, "value": world * M * K6 / p_s,
                                 "unit": "filter-update steps/s (whole job)", "kernel_ms": pms,
                                 "roofline": rl.hbm_roofline(pms, M, rl.algorithmic_bytes("vanilla", 10, 4), rl.moved_bytes("vanilla", 10, 4),
                                                             *rl.load_traffic(ROOT, "vanilla_split_kernel<double, 12, 4, 2, 4, true, false, false")),
                                 "filters_with_error_status": pbad}
        del pb, yq
        # ... and the corner of the envelope: Vanilla 16 / 8, one filter over EIGHT lanes, the 8 x 8 inverse of S formed once per filter by its
        # lanes (round 5: kb_vanilla_split.h dist_inverse); its counter bytes include H's own columns read a second time
        dq = synth.linear_batch(M, 16, 8, 1, seed=synth.SEED + 7000 + rank)
        yq = torch.from_numpy(np.ascontiguousarray(dq["y"][0].T)).to(dev)
        cb = ga.FilterBatch.new_ldkf(k.VANILLA, dq["x0"], dq["P0"], dq["F"], None, dq["H"], dq["Q"], dq["R"], device=local_rank)
        c_s, cms, cbad = timed_leg(cb, lambda: cb.update_dev(yq.data_ptr(), M), K6)
        extra["vanilla_16x8"] = {"config": "%d Vanilla 16/8 fp64 filters per GPU, per-filter models (vanilla_split_kernel<double, 16, 8, 0, 8>: one filter per eight lanes)" % M,
                                 "filters_total": world * M, "steps": K6, "value": world * M * K6 / c_s,
                                 "unit": "filter-update steps/s (whole job)", "kernel_ms": cms,
                                 "roofline": rl.hbm_roofline(cms, M, rl.algorithmic_bytes("vanilla", 16, 8), rl.moved_bytes("vanilla", 16, 8),
                                                             *rl.load_traffic(ROOT, "vanilla_split_kernel<double, 16, 8, 0, 8, false, false, false")),
                                 "filters_with_error_status": cbad}
        del cb, yq
    if args.srif_filters > 0:
        M = args.srif_filters
        sn, sp = 12, 6
        rng = np.random.default_rng(5 + rank)
        x0 = rng.standard_normal((M, sn))
        P0 = np.zeros((M, sn, sn)); P0[:, np.arange(sn), np.arange(sn)] = [10.0] * 6 + [1.0] * 6
        R = np.zeros((M, sp, sp)); R[:, np.arange(sp), np.arange(sp)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(M, sp)))
        sb = ga.FilterBatch(k.SRIF, sn, sp, 0, M, dtype=k.F32, device=local_rank)
        sb.set(k.X, x0, 1); sb.set(k.P, P0, 2); sb.set(k.R, R, 2, p_rows=sp); sb.init()
        g = torch.Generator(device=dev)
        g.manual_seed(11 + rank)
        Phi = (torch.eye(sn, dtype=torch.float32, device=dev).reshape(sn * sn, 1)
               + 1e-2 * torch.randn(sn * sn, M, dtype=torch.float32, device=dev, generator=g)).contiguous()
        Ht = torch.randn(sp * sn, M, dtype=torch.float32, device=dev, generator=g)
        real = torch.randn(sp, M, dtype=torch.float32, device=dev, generator=g)
        comp = real + 1e-2 * torch.randn(sp, M, dtype=torch.float32, device=dev, generator=g)
        torch.cuda.synchronize()

        def sstep():
            k.check(k.lib().kb_prepare_dev(sb._h, Phi.data_ptr(), Ht.data_ptr(), M))
            k.check(k.lib().kb_update_nl_dev(sb._h, real.data_ptr(), comp.data_ptr(), M))
        K5 = 50
        r_s, rms, rbad = timed_leg(sb, sstep, K5)
        extra["srif_fp32"] = {"config": "configs[4] E: %d SRIF 12/6 fp32 filters per GPU, per-step Phi/Htilde read in place (kb_prepare_dev + kb_update_nl_dev)" % M,
                              "filters_total": world * M, "steps": K5, "value": world * M * K5 / r_s,
                              "unit": "filter-update steps/s (whole job)", "kernel_ms": rms, "dtype": "f32",
                              "roofline": rl.hbm_roofline(rms, M, 576 * 4, rl.moved_bytes("srif_pair", sn, sp, 4),
                                                          *rl.load_traffic(ROOT, "srif_pair_kernel<float, 12, 6, false, true")),
                              "filters_with_error_status": rbad}
        if args.fused_steps > 0 and rank == 0:
            # config E with the caller loop `for k { Prepare(Phi_k, Htilde_k); Update(real_k, computed_k) }` inside ONE launch (round 6,
            # kb_update_nl_steps_dev): distinct Phi / Htilde / observations per step, (b, R) resident in registers between the steps; the
            # same bits as T single calls (tests/test_srif_gpu.py, and `parity` below)
            TF = 20
            gF = torch.Generator(device=dev); gF.manual_seed(13)
            PhiT = (torch.eye(sn, dtype=torch.float32, device=dev).reshape(1, sn * sn, 1)
                    + 1e-2 * torch.randn(TF, sn * sn, M, dtype=torch.float32, device=dev, generator=gF)).contiguous()
            HtT = torch.randn(TF, sp * sn, M, dtype=torch.float32, device=dev, generator=gF)
            realT = torch.randn(TF, sp, M, dtype=torch.float32, device=dev, generator=gF)
            compT = realT + 1e-2 * torch.randn(TF, sp, M, dtype=torch.float32, device=dev, generator=gF)
            torch.cuda.synchronize()

            def fstep():
                sb.update_nl_steps_dev(PhiT.data_ptr(), HtT.data_ptr(), M, sn * sn * M, sp * sn * M, realT.data_ptr(), compT.data_ptr(), M, sp * M, TF)
            # (rank 0 only: no collective in here -- timed_leg() holds barriers and all-gathers every rank must enter)
            warm_clocks()
            for _ in range(5):
                fstep()
            sb.synchronize()
            sfs = torch.cuda.ExternalStream(sb.stream(), device=dev)
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            q0.record(sfs)
            for _ in range(10):
                fstep()
            q1.record(sfs)
            sb.synchronize()
            fms, fbad = q0.elapsed_time(q1) / 10, int(np.count_nonzero(sb.status()))
            moved_f = rl.moved_bytes("srif_pair", sn, sp, 4) - 4 * (sn + rl.tri(sn) + sn // 2)   # the own rows of (b, R) are not re-read
            extra["srif_fp32"]["fused"] = {"config": "the caller loop inside one launch (kb_update_nl_steps_dev): %d steps per launch, distinct Phi / Htilde / observations per step" % TF,
                                           "steps_per_launch": TF, "ms_per_launch": fms, "kernel_ms": fms / TF, "value": M * TF / (fms * 1e-3),
                                           "unit": "filter-update steps/s (1 GPU, kb_update_nl_steps_dev)", "filters_with_error_status": fbad,
                                           "roofline": rl.hbm_roofline(fms / TF, M, 576 * 4, moved_f, None, {"live": False, "analytic": "packed working set less the resident rows of (b, R)"})}
            del PhiT, HtT, realT, compT
        del sb, Phi, Ht, real, comp

    if rank == 0:
        value = total_filter_steps / wall_s
        counter_bpf, src = rl.load_traffic(ROOT, HEADLINE_KERNEL)
        roof = rl.hbm_roofline(kernel_ms, N, rl.algorithmic_bytes("vanilla", n, p), rl.moved_bytes("vanilla", n, p), counter_bpf, src)
        # what the fraction means at this batch size: the counters sit on the fabric side of the L2 and include
        # Infinity-Cache hits, and x, P (432 of the 1104 B) are cache-resident by design
        stream_bytes = 8 * (n * n + p * n + rl.tri(n) + rl.tri(p) + p)   # F, H, Q, R, y: read once per step, certainly from DRAM
        roof["side"] = "L2 <-> fabric (Infinity-Cache hits included): the state block (%d MB) is cache-resident at this batch size" % (N * (n + rl.tri(n)) * 8 // 1000000)
        roof["dram_frac_lower_bound"] = stream_bytes * N / (kernel_ms * 1e-3) / 1e9 / rl.HBM_PEAK_GBPS
        roof["dram_stream_bytes_per_filter_step"] = stream_bytes
        if ooc:
            roof["hbm_only"] = {"frac": ooc["frac"], "frac_of_achievable": ooc["frac_of_achievable"], "filters": ooc["filters"],
                                "note": "the same kernel with the state block far outside the Infinity Cache (out_of_cache): every byte is HBM traffic"}
        med = sorted(blocks_wall)[len(blocks_wall) // 2]
        out = {
            "metric": "filter-update steps/s (whole node), 1M x 6-state Vanilla",
            "value": value, "unit": "filter-update steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall_s / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %d independent 6-state/3-meas Vanilla filters per GPU, fp64, "
                                   "per-filter F/H/Q/R, Noiseless, one kb_update_dev launch per step" % N,
                       "filters_per_gpu": N, "n": n, "p": p, "kernel": "vanilla_reg_kernel<double,6,3,0>",
                       "sharding": "independent filter shards, no collective in the update path; epilogue all-reduce of counts"},
            "roofline": roof,
            "repetitions": {"blocks": len(blocks_wall), "steps_per_block": args.steps,
                            "ms_per_step_blocks": [w / args.steps * 1e3 for w in blocks_wall], "ms_per_step_median": med / args.steps * 1e3,
                            "value_median": total_filter_steps / med, "kernel_ms_blocks": blocks_kernel,
                            "note": "`value` / `ms_per_step` are the first block's (the contract's K steps)"},
            "filters_with_error_status": total_bad,
            "ranks": {"launched": world, "rccl_ranks_seen": ranks_seen, "backend": args.dist_backend if use_dist else "none",
                      "per_rank_ms_per_step": [s_ / args.steps * 1e3 for s_ in per_rank_s],
                      "filter_steps_counted": total_filter_steps},
        }
        out["strong_scaling"] = strong if strong is not None else {
            "filters_total": N, "filters_per_gpu": [N], "ms_per_step": out["ms_per_step"], "value": value,
            "unit": "filter-update steps/s (whole job)", "note": "one rank: identical to the weak-scaling figure"}
        if ooc:
            out["out_of_cache"] = ooc
        if fused:
            out["fused"] = fused
        if extra:
            out["extra"] = extra
        if not args.no_host_path:   # drop-in use at batch scale: host measurements, PCIe-inclusive (never `value`)
            yh = [np.ascontiguousarray(d["y"][t]) for t in range(POOL)]
            for t in range(3):
                b.update(yh[t % POOL], snapshot=False)
            t0 = time.perf_counter()
            KH = 10
            for t in range(KH):
                b.update(yh[t % POOL], snapshot=False)
            hdt = (time.perf_counter() - t0) / KH
            out["host_path"] = {"ms_per_step": hdt * 1e3, "value": N / hdt, "unit": "filter-update steps/s (1 GPU, kb_update)",
                                "note": "%d MB of host measurements per call: H2D copy + pack + step + synchronise" % (N * p * 8 // 1000000)}
        if not args.no_parity:
            out["parity"] = _parity(ga, k, synth)
            for leg in ("squareroot", "shared_model", "vanilla_12x6", "squareroot_12x6", "information_12x6", "vanilla_10x4", "vanilla_16x8", "hybrid_ekf", "srif_fp32"):   # every leg of `extra` proves itself (oracle = checker, untimed)
                if leg in extra:
                    extra[leg]["parity"] = _leg_parity(ga, k, synth, leg)
            if fused:   # the time-fused launch is a different kernel from the headline's: its own gate
                fused["parity"] = _leg_parity(ga, k, synth, "fused")
            if "squareroot" in extra and "fused" in extra["squareroot"]:
                extra["squareroot"]["fused"]["parity"] = _leg_parity(ga, k, synth, "squareroot_fused")
            if "srif_fp32" in extra and "fused" in extra["srif_fp32"]:
                extra["srif_fp32"]["fused"]["parity"] = _srif_fused_parity(ga, k)
            if "hybrid_ekf" in extra and "fused" in extra["hybrid_ekf"]:
                extra["hybrid_ekf"]["fused"]["parity"] = _hybrid_fused_parity(ga, k)
        if not args.no_cpu_baseline:   # rank 0 of any world size: the host cores are the same ones
            out["cpu_baseline"] = _cpu_baseline(d)
        # the full document goes to a side file; the ONE stdout line is its compact headline (round 5's 22.8 KB line was not parsed)
        from gokalman_amd import benchline
        full_path = None
        if args.full_out:
            full_path = args.full_out if os.path.isabs(args.full_out) else os.path.join(ROOT, args.full_out)
            try:
                os.makedirs(os.path.dirname(full_path), exist_ok=True)
                with open(full_path + ".tmp", "w") as fh:
                    json.dump(out, fh)
                os.replace(full_path + ".tmp", full_path)
            except OSError as exc:   # a read-only tree must not cost the bench line
                print("bench: could not write %s: %s" % (full_path, exc), file=sys.stderr, flush=True)
                full_path = None
        print(benchline.dumps(out, args.full_out if full_path else None), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
