"""Throughput of the other SURVEY section 8d configs (C: SquareRoot, D: MC + Hybrid, E: SRIF fp32) and
Information, through the C ABI with device-resident inputs.  Prints one JSON line per config.
usage: python scripts/bench_kinds.py [sqrt info srif hybrid mc] [--n N]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth
from gokalman_amd import roofline as rl

args = [a for a in sys.argv[1:] if not a.startswith("--")]
which = args or ["vsplit", "vpad", "sqsplit", "infsplit", "vfull", "vbase", "vshared", "vnoise", "vstrict", "sqrt", "info", "sshared", "srif", "srifpad", "hybrid", "hpad", "hstrict", "mc"]
Nopt = None
for a in sys.argv[1:]:
    if a.startswith("--srif-shapes="):
        continue
    if a.startswith("--n="):
        Nopt = int(a[4:])


_warmer = None


def warm_clocks(ms=60.0):
    """Untimed load before a timed region (see bench.py warm_clocks): after host-side set-up the GPU is at idle clocks and
    needs ~25 ms of sustained work to come back; without it the first configs of a cold process read 15-40 % slow."""
    global _warmer
    if _warmer is None:
        Nw = 1 << 18
        dw = synth.linear_batch(Nw, 6, 3, 1)
        yw = torch.from_numpy(np.ascontiguousarray(dw["y"].transpose(0, 2, 1))).cuda()
        _warmer = (ga.FilterBatch.new_ldkf(k.VANILLA, dw["x0"], dw["P0"], dw["F"], None, dw["H"], dw["Q"], dw["R"]), yw, Nw)
    wb, yw, Nw = _warmer
    for _ in range(int(ms / 0.045)):
        wb.update_dev(yw[0].data_ptr(), Nw)
    wb.synchronize()


def timed(b, fn, K=20, warm=5):
    torch.cuda.synchronize()   # arrays made by torch are complete before the handle's (non-blocking) stream reads them
    warm_clocks()
    s = torch.cuda.ExternalStream(b.stream())
    for _ in range(warm):
        fn()
    b.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(K):
        fn()
    e1.record(s)
    b.synchronize()
    return e0.elapsed_time(e1) / K


def report(name, N, ms, bytes_per, extra=None, dtype="f64", moved=None):
    """One JSON line per config, with the keys of bench.py's contract line (value = filter-update steps/s with the inputs
    resident in HBM).  roofline.frac is PHYSICAL: bytes the kernel moves (packed working set, gokalman_amd/roofline.py) /
    kernel time / 8 TB/s; the SURVEY 8d full-matrix figure is roofline.frac_algorithmic."""
    out = {"config": name, "filters": N, "ms_per_step": ms, "steps_per_s": N / (ms * 1e-3),
           "metric": "filter-update steps/s", "value": N / (ms * 1e-3), "unit": "filter-update steps/s", "n_gpus": 1,
           "higher_is_better": True, "dtype": dtype, "data": "synthetic",
           "roofline": rl.hbm_roofline(ms, N, bytes_per, moved if moved is not None else bytes_per)}
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)


if "vfull" in which:
    N = Nopt or (1 << 20)
    d = synth.linear_batch(N, 6, 3, 1)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE)
    ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
    report("B': Vanilla 6/3 f64, FULL_ESTIMATE (also writes P-, K, innovation, yhat: +480 B)", N, ms, 1488 + 480, {"errors": int(np.count_nonzero(b.status()))},
           moved=rl.moved_bytes("vanilla_full", 6, 3))
    del b

if "vbase" in which:   # the headline configuration (bench.py times it too): here as the same-box reference of the legs below
    N = Nopt or (1 << 20)
    d = synth.linear_batch(N, 6, 3, 1)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
    report("B: Vanilla 6/3 f64", N, ms, 1488, {"errors": int(np.count_nonzero(b.status()))}, moved=rl.moved_bytes("vanilla", 6, 3))
    del b

if "vsplit" in which:
    # Vanilla beyond 8 states: 12 / 6 (the orbit-determination size of config E) at config E's batch size, per-filter models:
    # one filter split over four lanes (kb_vanilla_split.h); the run-time-dimension kernel (KB_FLAG_STATEMENT_KERNELS) next to it
    N = Nopt or (1 << 18)
    d = synth.linear_batch(N, 12, 6, 1)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
    report("Vanilla 12/6 f64, one filter per four lanes", N, ms, rl.algorithmic_bytes("vanilla", 12, 6), {"errors": int(np.count_nonzero(b.status()))},
           moved=rl.moved_bytes("vanilla", 12, 6))
    del b
    if "--with-statement" in sys.argv:
        b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_STATEMENT_KERNELS)
        ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N), K=3, warm=1)
        report("Vanilla 12/6 f64, statement kernel (KB_FLAG_STATEMENT_KERNELS)", N, ms, rl.algorithmic_bytes("vanilla", 12, 6), {"errors": int(np.count_nonzero(b.status()))},
               moved=rl.moved_bytes("vanilla", 12, 6))
        del b

if "vpad" in which:
    # members of the padded family on the split kernels' run-time-dimension instantiations (Noiseless, state only): Vanilla 10 / 4 on
    # <12, 4>, SquareRoot 8 / 4 on <8, 4>, Information 8 / 4 on <8, 4>, Vanilla 16 / 4 on <16, 4> (eight lanes per filter); and the 12 / 6
    # Vanilla with KB_FLAG_FULL_ESTIMATE (the Estimate's extras leave where they are formed)
    N = Nopt or (1 << 18)
    for kind, kname, n, p, flags in ((k.VANILLA, "Vanilla", 10, 4, 0), (k.SQUAREROOT, "SquareRoot", 8, 4, 0), (k.INFORMATION, "Information", 8, 4, k.FLAG_INFO_FROM_STATE),
                                     (k.VANILLA, "Vanilla", 16, 4, 0), (k.VANILLA, "Vanilla FULL", 12, 6, k.FLAG_FULL_ESTIMATE),
                                     # (round 5: S^-1 once per filter at p = 7, 8 -- the exact 12 / 8 and 16 / 8 kernels, the padded <16, 8> one at 14 / 7)
                                     (k.VANILLA, "Vanilla", 12, 8, 0), (k.VANILLA, "Vanilla", 14, 7, 0), (k.VANILLA, "Vanilla", 16, 8, 0)):
        if p <= n // 2 and n % 2 == 0:
            d = synth.linear_batch(N, n, p, 1)
        else:   # (more measurements than position states: scripts/bench_split_shapes.py's dense-H problem)
            rng = np.random.default_rng(n)
            sc = (1.0 + 0.01 * rng.random(N))[:, None, None]
            d = dict(F=np.eye(n) + sc * (0.05 * rng.standard_normal((n, n))), H=sc * rng.standard_normal((p, n)), Q=sc * (1e-3 * np.eye(n)), R=sc * (1e-2 * np.eye(p)),
                     x0=np.zeros((N, n)), P0=np.broadcast_to(np.eye(n), (N, n, n)), y=rng.standard_normal((1, N, p)))
        y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
        b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=flags)
        ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
        fam = "vanilla_full" if flags & k.FLAG_FULL_ESTIMATE else kname.split()[0].lower()
        report("%s %d/%d f64, split-lane kernel" % (kname, n, p), N, ms, rl.algorithmic_bytes(fam if fam != "vanilla_full" else "vanilla", n, p),
               {"errors": int(np.count_nonzero(b.status()))}, moved=rl.moved_bytes(fam, n, p))
        del b, d, y

if "sqsplit" in which:
    # SquareRoot beyond 6 states: 12 / 6 at config E's batch size, per-filter models (kb_squareroot_split.h: one filter over four lanes,
    # the Householder panels distributed by columns)
    N = Nopt or (1 << 18)
    d = synth.linear_batch(N, 12, 6, 1)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    b = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
    report("SquareRoot 12/6 f64, one filter per four lanes", N, ms, rl.algorithmic_bytes("squareroot", 12, 6), {"errors": int(np.count_nonzero(b.status()))},
           moved=rl.moved_bytes("squareroot", 12, 6))
    del b
    if "--with-statement" in sys.argv:
        b = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_STATEMENT_KERNELS)
        ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N), K=3, warm=1)
        report("SquareRoot 12/6 f64, statement kernel (KB_FLAG_STATEMENT_KERNELS)", N, ms, rl.algorithmic_bytes("squareroot", 12, 6), {"errors": int(np.count_nonzero(b.status()))},
               moved=rl.moved_bytes("squareroot", 12, 6))
        del b

if "infsplit" in which:
    # Information beyond 6 states: 12 / 6 (kb_information_split.h: the pivoted LU solve distributed over four lanes)
    N = Nopt or (1 << 18)
    d = synth.linear_batch(N, 12, 6, 1)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    b = ga.FilterBatch.new_ldkf(k.INFORMATION, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_INFO_FROM_STATE)
    ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
    report("Information 12/6 f64, one filter per four lanes", N, ms, rl.algorithmic_bytes("information", 12, 6), {"errors": int(np.count_nonzero(b.status()))},
           moved=rl.moved_bytes("information", 12, 6))
    del b
    if "--with-statement" in sys.argv:
        b = ga.FilterBatch.new_ldkf(k.INFORMATION, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_INFO_FROM_STATE | k.FLAG_STATEMENT_KERNELS)
        ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N), K=3, warm=1)
        report("Information 12/6 f64, statement kernel (KB_FLAG_STATEMENT_KERNELS)", N, ms, rl.algorithmic_bytes("information", 12, 6), {"errors": int(np.count_nonzero(b.status()))},
               moved=rl.moved_bytes("information", 12, 6))
        del b

if "vstrict" in which:
    # KB_FLAG_STRICT_SYMCHECK: kb_vanilla_strict.hip (registers) against vanilla_gen_kernel (scratch arrays; KB_FLAG_STATEMENT_KERNELS)
    # -- bit-identical results (tests/test_symcheck_gpu.py), so the pair times the register kernel's gain
    N = Nopt or (1 << 20)
    d = synth.linear_batch(N, 6, 3, 1)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_STRICT_SYMCHECK)
    ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
    report("B strict: Vanilla 6/3 f64, STRICT_SYMCHECK, register kernel", N, ms, 1488, {"errors": int(np.count_nonzero(b.status()))}, moved=rl.moved_bytes("vanilla", 6, 3))
    del b
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_STRICT_SYMCHECK | k.FLAG_STATEMENT_KERNELS)
    ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N), K=5, warm=1)
    report("B strict: Vanilla 6/3 f64, STRICT_SYMCHECK, statement kernel (KB_FLAG_STATEMENT_KERNELS)", N, ms, 1488, {"errors": int(np.count_nonzero(b.status()))}, moved=rl.moved_bytes("vanilla", 6, 3))
    del b

if "vshared" in which:
    # ONE model for the whole batch (every model field uploaded with broadcast = 1: the reference's own use -- one filter object, many
    # runs / targets): the kernels read tile 0's model block, which stays in the L2, so only state and measurements move
    N = Nopt or (1 << 20)
    d = synth.linear_batch(N, 6, 3, 1)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"][0], None, d["H"][0], d["Q"][0], d["R"][0], nfilters=N)
    ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
    sb = 8 * (6 + 21 + 3 + 6 + 21)   # x, P read; y read; x, P written
    report("B shared model: Vanilla 6/3 f64, one F / H / Q / R for all filters", N, ms, 1488, {"errors": int(np.count_nonzero(b.status()))}, moved=sb)
    del b

if "sshared" in which:
    N = Nopt or (1 << 20)
    d = synth.linear_batch(N, 6, 3, 1)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    for name, kind, fl in (("C shared model: SquareRoot 6/3 f64", k.SQUAREROOT, 0), ("Information 6/3 f64 shared model (from state)", k.INFORMATION, k.FLAG_INFO_FROM_STATE)):
        b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"][0], None, d["H"][0], d["Q"][0], d["R"][0], nfilters=N, flags=fl)
        ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
        report(name, N, ms, 1488, {"errors": int(np.count_nonzero(b.status()))}, moved=8 * (6 + 21 + 3 + 6 + 21))
        del b

if "vnoise" in which:
    # config B with the reference's usual Noise object: AWGN (noise.go:109-164) -- three draws per filter-step (Process, Measurement,
    # Process: vanilla.go:146,157,195) from the device's Philox stream, on the register kernel (kb_vanilla_reg.h, NOISE = true)
    N = Nopt or (1 << 20)
    d = synth.linear_batch(N, 6, 3, 1)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], noise=k.NOISE_AWGN, seed=17)
    ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
    report("B + AWGN: Vanilla 6/3 f64, noise drawn on the device", N, ms, 1488, {"errors": int(np.count_nonzero(b.status()))},
           moved=rl.moved_bytes("vanilla_awgn", 6, 3))
    del b

if "sqrt" in which or "info" in which:
    N = Nopt or (1 << 20)
    d = synth.linear_batch(N, 6, 3, 1)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    for name, kind, fl, mv in (("C: SquareRoot 6/3 f64", k.SQUAREROOT, 0, "squareroot"), ("Information 6/3 f64 (from state)", k.INFORMATION, k.FLAG_INFO_FROM_STATE, "information")):
        if ("sqrt" in which and kind == k.SQUAREROOT) or ("info" in which and kind == k.INFORMATION):
            b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=fl)
            ms = timed(b, lambda: b.update_dev(y[0].data_ptr(), N))
            report(name, N, ms, 1488, {"errors": int(np.count_nonzero(b.status()))}, moved=rl.moved_bytes(mv, 6, 3))
            del b

if "srif" in which:
    N = Nopt or (1 << 18)
    n, p = 12, 6
    rng = np.random.default_rng(5)
    x0 = rng.standard_normal((N, n)); P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10.0] * 6 + [1.0] * 6
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    srif_flags = k.FLAG_FULL_ESTIMATE if "--srif-full" in sys.argv else 0   # (with the Estimate's extras: R-bar, yhat, the pre-fit residual, the innovation)
    for dt, nm, w in ((k.F32, "E: SRIF 12/6 f32" + (" FULL" if srif_flags else ""), 4), (k.F64, "SRIF 12/6 f64" + (" FULL" if srif_flags else ""), 8)):
        b = ga.FilterBatch(k.SRIF, n, p, 0, N, dtype=dt, flags=srif_flags)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
        tdt = torch.float32 if dt == k.F32 else torch.float64
        Phi = (torch.eye(n, dtype=tdt, device="cuda").reshape(n * n, 1) + 1e-2 * torch.randn(n * n, N, dtype=tdt, device="cuda")).contiguous()
        Ht = torch.randn(p * n, N, dtype=tdt, device="cuda")
        real = torch.randn(p, N, dtype=tdt, device="cuda"); comp = real + 1e-2 * torch.randn(p, N, dtype=tdt, device="cuda")
        b.prepare_dev = lambda: k.check(k.lib().kb_prepare_dev(b._h, Phi.data_ptr(), Ht.data_ptr(), N))
        def step():
            b.prepare_dev()
            k.check(k.lib().kb_update_nl_dev(b._h, real.data_ptr(), comp.data_ptr(), N))
        ms = timed(b, step, K=10)
        # update only (model already resident): re-arm the lock without re-uploading
        report(nm + " (prepare_dev + update_nl_dev)", N, ms, 576 * w, {"errors": int(np.count_nonzero(b.status()))}, dtype="f32" if dt == k.F32 else "f64",
               moved=rl.moved_bytes("srif_pair", n, p, w))
        del b

if "srifpad" in which:
    # further SRIF shapes of the two-lanes-per-filter kernel (kb_srif_pair*b.hip, *c.hip), fp64, against the statement kernel
    N = Nopt or (1 << 18)
    shapes = ((8, 2), (10, 4), (12, 2), (12, 5), (7, 3), (11, 4), (10, 6), (4, 2), (8, 8), (10, 8), (12, 8), (14, 4), (16, 4), (16, 6), (15, 3), (13, 8), (16, 8))
    for a_ in sys.argv[1:]:
        if a_.startswith("--srif-shapes="):   # e.g. --srif-shapes=16x6,14x4
            shapes = tuple(tuple(int(v) for v in t.split("x")) for t in a_[14:].split(","))
    for (n, p) in shapes:
        rng = np.random.default_rng(5)
        x0 = rng.standard_normal((N, n)); P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)])
        R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
        for flags, nm in ((0, "two lanes per filter"), (k.FLAG_STATEMENT_KERNELS, "statement kernel")):
            if flags and "--with-statement" not in sys.argv:
                continue
            b = ga.FilterBatch(k.SRIF, n, p, 0, N, dtype=(k.F32 if "--srif-f32" in sys.argv else k.F64), flags=flags)
            b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
            Phi = (torch.eye(n, dtype=(torch.float32 if "--srif-f32" in sys.argv else torch.float64), device="cuda").reshape(n * n, 1) + 1e-2 * torch.randn(n * n, N, dtype=(torch.float32 if "--srif-f32" in sys.argv else torch.float64), device="cuda")).contiguous()
            Ht = torch.randn(p * n, N, dtype=(torch.float32 if "--srif-f32" in sys.argv else torch.float64), device="cuda")
            real = torch.randn(p, N, dtype=(torch.float32 if "--srif-f32" in sys.argv else torch.float64), device="cuda"); comp = real + 1e-2 * torch.randn(p, N, dtype=(torch.float32 if "--srif-f32" in sys.argv else torch.float64), device="cuda")
            def step():
                k.check(k.lib().kb_prepare_dev(b._h, Phi.data_ptr(), Ht.data_ptr(), N))
                k.check(k.lib().kb_update_nl_dev(b._h, real.data_ptr(), comp.data_ptr(), N))
            ms = timed(b, step, K=10 if not flags else 3, warm=3 if not flags else 1)
            report("SRIF %d/%d %s, %s (prepare_dev + update_nl_dev)" % (n, p, "f32" if "--srif-f32" in sys.argv else "f64", "register kernel" if not flags else nm), N, ms, rl.algorithmic_bytes("srif", n, p, 4 if "--srif-f32" in sys.argv else 8), {"errors": int(np.count_nonzero(b.status()))},
                   moved=rl.moved_bytes("srif", n, p, 4 if "--srif-f32" in sys.argv else 8))
            del b

if "hybrid" in which:
    N = Nopt or (1 << 20)
    n, p = 6, 2
    rng = np.random.default_rng(6)
    x0 = rng.standard_normal((N, n)); P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10, 10, 10, 1, 1, 1]
    b = ga.FilterBatch(k.HYBRID, n, p, 0, N)
    b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, np.diag([1e-6, 1e-6]), 2, p_rows=p); b.init(); b.enable_ekf()
    Phi = (torch.eye(n, dtype=torch.float64, device="cuda").reshape(n * n, 1) + 1e-2 * torch.randn(n * n, N, dtype=torch.float64, device="cuda")).contiguous()
    Ht = torch.randn(p * n, N, dtype=torch.float64, device="cuda")
    real = torch.randn(p, N, dtype=torch.float64, device="cuda"); comp = real + 1e-3 * torch.randn(p, N, dtype=torch.float64, device="cuda")
    def step():
        k.check(k.lib().kb_prepare_dev(b._h, Phi.data_ptr(), Ht.data_ptr(), N))
        k.check(k.lib().kb_update_nl_dev(b._h, real.data_ptr(), comp.data_ptr(), N))
    ms = timed(b, step, K=10)
    report("D(ii): Hybrid EKF 6/2 f64 (prepare_dev + update_nl_dev)", N, ms, 1120, {"errors": int(np.count_nonzero(b.status()))}, moved=rl.moved_bytes("hybrid", 6, 2))
    del b

if "hpad" in which:
    # HybridKF shapes without an exact register kernel on the padded ones (kb_hybrid_reg.h PAD: any n <= 8, p <= 4), EKF, zero-copy Phi / Htilde
    N = Nopt or (1 << 20)
    for (n, p) in ((5, 2), (7, 3), (8, 4), (4, 2), (9, 2), (12, 4), (16, 6), (12, 8), (16, 8)):   # (beyond 8 states: kb_hybrid_split.hip)
        rng = np.random.default_rng(6)
        x0 = rng.standard_normal((N, n)); P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)])
        for flags, nm in ((0, "padded register kernel" if n <= 8 else "split-lane kernel"), (k.FLAG_STATEMENT_KERNELS, "statement kernel")):
            if flags and "--with-statement" not in sys.argv:
                continue
            b = ga.FilterBatch(k.HYBRID, n, p, 0, N, flags=flags)
            b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, np.diag(np.full(p, 1e-4)), 2, p_rows=p); b.init(); b.enable_ekf()
            Phi = (torch.eye(n, dtype=torch.float64, device="cuda").reshape(n * n, 1) + 1e-2 * torch.randn(n * n, N, dtype=torch.float64, device="cuda")).contiguous()
            Ht = torch.randn(p * n, N, dtype=torch.float64, device="cuda")
            real = torch.randn(p, N, dtype=torch.float64, device="cuda"); comp = real + 1e-3 * torch.randn(p, N, dtype=torch.float64, device="cuda")
            def step():
                k.check(k.lib().kb_prepare_dev(b._h, Phi.data_ptr(), Ht.data_ptr(), N))
                k.check(k.lib().kb_update_nl_dev(b._h, real.data_ptr(), comp.data_ptr(), N))
            ms = timed(b, step, K=10 if not flags else 3, warm=3 if not flags else 1)
            report("Hybrid EKF %d/%d f64, %s (prepare_dev + update_nl_dev)" % (n, p, nm), N, ms, rl.algorithmic_bytes("hybrid", n, p), {"errors": int(np.count_nonzero(b.status()))},
                   moved=rl.moved_bytes("hybrid", n, p))
            del b

if "hstrict" in which:
    # Hybrid CKF under KB_FLAG_STRICT_SYMCHECK: kb_hybrid_strict.hip (registers) against hybrid_gen_kernel (KB_FLAG_STATEMENT_KERNELS);
    # the model block is resident (kb_prepare_dev packs Phi / Htilde there for strict batches), bit-identical results
    N = Nopt or (1 << 20)
    n, p = 6, 2
    rng = np.random.default_rng(6)
    x0 = rng.standard_normal((N, n)); P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10, 10, 10, 1, 1, 1]
    Phi = (torch.eye(n, dtype=torch.float64, device="cuda").reshape(n * n, 1) + 1e-2 * torch.randn(n * n, N, dtype=torch.float64, device="cuda")).contiguous()
    Ht = torch.randn(p * n, N, dtype=torch.float64, device="cuda")
    real = torch.randn(p, N, dtype=torch.float64, device="cuda"); comp = real + 1e-2 * torch.randn(p, N, dtype=torch.float64, device="cuda")
    for nm, fl in (("register kernel", k.FLAG_STRICT_SYMCHECK), ("statement kernel", k.FLAG_STRICT_SYMCHECK | k.FLAG_STATEMENT_KERNELS)):
        b = ga.FilterBatch(k.HYBRID, n, p, 0, N, flags=fl)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, np.diag([1e-2, 1e-2]), 2, p_rows=p); b.init()
        def step():
            k.check(k.lib().kb_prepare_dev(b._h, Phi.data_ptr(), Ht.data_ptr(), N))
            k.check(k.lib().kb_update_nl_dev(b._h, real.data_ptr(), comp.data_ptr(), N))
        ms = timed(b, step, K=5, warm=2)
        report("D(ii) strict: Hybrid CKF 6/2 f64, STRICT_SYMCHECK, %s (prepare_dev packs + update_nl_dev)" % nm, N, ms, 1120,
               {"errors": int(np.count_nonzero(b.status()))}, moved=rl.moved_bytes("hybrid", 6, 2))
        del b

if "mc" in which:
    s = dict(  # examples/statOD5044/main.go:36-57
        F=np.array([[1, 0.1, 0, 7.726e-2], [4.015e-7, 1, 0, 1.545], [-2.319e-16, -1.732e-9, 1, 0.1], [-6.956e-15, -3.465e-8, 0, 1]]),
        G=np.array([[5e-3, 3.85e-7], [0.1, 1.157e-5], [-5.775e-11, 7.487e-7], [1.732e-9, 1.498e-5]]),
        H=np.array([[1.0, 0, 0, 0], [0, 0, 1, 0]]),
        Q=np.array([[6.669e-16, 1.001e-14, 3.823e-19, 5.150e-18], [1.001e-14, 2.002e-13, 1.030e-17, 1.545e-16],
                    [3.862e-19, 1.030e-17, 6.667e-19, 1.000e-17], [5.150e-18, 1.545e-16, 1.000e-17, 2.000e-16]]),
        R=np.diag([2e-3, 2e-5]) / 0.1, x0=np.array([2, 0.5, 0, 0.0]), P0=np.diag([5, 1, 0.01, 1e-5]))
    runs, steps = Nopt or (1 << 20), 1086
    kf = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, s["x0"], s["P0"], s["F"], s["G"], s["H"], s["Q"], s["R"],
                                 nfilters=runs, noise=k.NOISE_AWGN, seed=1)
    ga.new_monte_carlo_runs(runs, 8, 2, np.zeros((1, 2)), kf)
    warm_clocks()
    t = time.perf_counter()
    mc = ga.new_monte_carlo_runs(runs, steps, 2, np.zeros((1, 2)), kf)
    dt = time.perf_counter() - t
    print(json.dumps({"config": "D(i): MC pure-predictor statOD5044 n=4, AWGN", "runs": runs, "steps": steps, "seconds": dt,
                      "run_steps_per_s": runs * steps / dt, "stddev_last": mc.stddev(steps - 1).tolist()}), flush=True)
