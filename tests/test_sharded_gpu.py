"""The native multi-device driver (kb_sharded_*, csrc/kb_sharded.hip): one process, one handle + host thread
+ stream per shard.  The test box has ONE GPU, so the shards share it (the statistics reduction then adds the shards on the
host; the ncclAllReduce branch needs distinct devices and runs on the driver's multi-GPU node only):
  * shard boundaries are 8e's formula, ragged sizes included;
  * a sharded Vanilla / SquareRoot batch is BIT-equal to the unsharded one (filters share nothing: no collective in Update);
  * the Monte-Carlo / chi-square statistics of the sharded ensemble equal one batch running every run (a run's noise depends
    only on its global index), and the same numbers come out of the C++ host mirror (gokalman::ShardedBatch)."""
import os
import subprocess

import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind", [k.VANILLA, k.SQUAREROOT])
@pytest.mark.parametrize("N,shards", [(1000, 2), (4099, 3), (200, 8)])
def test_sharded_batch_is_bit_equal_to_the_unsharded_batch(kind, N, shards):
    steps = 6
    d = synth.linear_batch(N, 6, 3, steps)
    one = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    sh = ga.ShardedBatch(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], N, devices=[0] * shards)
    assert sh.shards() == shards
    assert [sh.first(g) for g in range(shards + 1)] == [(N * g) // shards for g in range(shards + 1)]   # GPU g owns [g N / G, (g + 1) N / G)
    from gokalman_amd import dist as kd   # the one-process-per-GPU job splits alike (ragged N included)
    assert [kd.shard_range(N, g, shards) for g in range(shards)] == [(sh.first(g), sh.first(g + 1)) for g in range(shards)]
    for t in range(steps):
        one.update(d["y"][t], snapshot=False)
        sh.update(d["y"][t])
    assert np.array_equal(sh.get(k.STATE, (6,)), one.get(k.STATE))
    assert np.array_equal(sh.get(k.COVAR, (6, 6)), one.get(k.COVAR))
    assert not sh.status().any()


def _statod():
    import bench
    return {kk: np.array(v, dtype=np.float64) for kk, v in bench.STATOD.items()}


def test_sharded_monte_carlo_and_chisquare_equal_the_single_batch():
    s = _statod()
    runs, steps, seed = 6000, 40, 77
    args = (s["x0"], s["P0"], s["F"], s["G"], s["H"], s["Q"], s["R"])
    one = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, *args, nfilters=runs, noise=k.NOISE_AWGN, seed=seed)
    mc1 = ga.new_monte_carlo_runs(runs, steps, 2, np.zeros((1, 2)), one, keep_runs=False)
    kf1 = ga.FilterBatch.new_ldkf(k.VANILLA, *args, nfilters=runs)
    nis1, nees1 = ga.new_chi_square(kf1, mc1, np.zeros((1, 2)))
    truth = ga.ShardedBatch(k.VANILLA_PREDICT, *args, runs, devices=[0, 0, 0], noise=k.NOISE_AWGN, seed=seed)
    kf = ga.ShardedBatch(k.VANILLA, *args, runs, devices=[0, 0, 0])
    mc = truth.monte_carlo(steps, np.zeros((1, 2)))
    assert not truth.used_rccl()          # three shards on one device: host sum (RCCL wants distinct devices)
    for t in (0, 1, steps // 2, steps - 1):
        assert np.allclose(mc.mean(t), mc1.mean(t), rtol=1e-12, atol=1e-15)
        assert np.allclose(mc.stddev(t), mc1.stddev(t), rtol=1e-9)
    nis, nees = truth.chi_square(kf, steps, np.zeros((1, 2)), replay_last_mc=True)
    assert np.allclose(nis, nis1, rtol=1e-10) and np.allclose(nees, nees1, rtol=1e-10)


def test_cpp_sharded_host():
    """gokalman::ShardedBatch (include/gokalman_amd.hpp) through a compiled host program: the same filters, sharded 2 ways on the
    one GPU, against gokalman::Batch driven filter by filter."""
    exe = "/tmp/gokalman_amd_sharded_host"
    lib = os.path.join(ROOT, "gokalman_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "sharded_host.cpp"),
                           "-o", exe, "-L" + lib, "-lgokalman_amd", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"])
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    out = dict(line.split(" ", 1) for line in res.stdout.strip().splitlines())
    assert out["shards"] == "2" and out["bit_equal_state"] == "1" and out["bit_equal_covariance"] == "1"
    assert out["mc_matches_single_batch"] == "1" and out["chi_matches_single_batch"] == "1" and out["used_rccl"] == "0"


def test_rccl_reduction_path_with_one_shard():
    """One GPU in the test loop: the ncclAllReduce branch (librccl loaded at run time, ncclCommInitAll from this one process, the
    grouped all-reduce on the shard's stream, the read-back) runs with ONE shard -- a one-rank all-reduce -- and must reproduce
    the plain batch."""
    s = _statod()
    runs, steps, seed = 3000, 30, 5
    args = (s["x0"], s["P0"], s["F"], s["G"], s["H"], s["Q"], s["R"])
    one = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, *args, nfilters=runs, noise=k.NOISE_AWGN, seed=seed)
    mc1 = ga.new_monte_carlo_runs(runs, steps, 2, np.zeros((1, 2)), one, keep_runs=False)
    truth = ga.ShardedBatch(k.VANILLA_PREDICT, *args, runs, devices=[0], noise=k.NOISE_AWGN, seed=seed)
    mc = truth.monte_carlo(steps, np.zeros((1, 2)))
    assert truth.used_rccl()
    for t in (0, steps - 1):
        assert np.allclose(mc.mean(t), mc1.mean(t), rtol=1e-12, atol=1e-15) and np.allclose(mc.stddev(t), mc1.stddev(t), rtol=1e-9)
    kf = ga.ShardedBatch(k.VANILLA, *args, runs, devices=[0])
    nis, nees = truth.chi_square(kf, steps, np.zeros((1, 2)))
    kf1 = ga.FilterBatch.new_ldkf(k.VANILLA, *args, nfilters=runs)
    nis1, nees1 = ga.new_chi_square(kf1, mc1, np.zeros((1, 2)))
    assert truth.used_rccl() and np.allclose(nis, nis1, rtol=1e-10) and np.allclose(nees, nees1, rtol=1e-10)


def test_config_d_at_its_stated_size_eight_shards_of_2_20_runs_equal_one_8m_run_batch():
    """BASELINE configs[3] at size: the 8 388 608-run ensemble of the statOD5044 pure predictor (1086 steps) as EIGHT shards of 2^20
    runs -- kb_sharded_* with eight handles on the one GPU of the test box: the host-sum branch of the reduction the 8-GPU run
    performs over RCCL -- against ONE batch of 8M runs.  A run's noise depends only on (seed, global run index, step), so both
    hold the same 8M trajectories; the sums are the same up to the order of the (atomic) floating-point additions, which is why
    the comparison is 1e-12 relative on the means, not bit for bit.  Both must also be what montecarlo.go:18-59 estimates:
    Mean(k) = F^k x0, StdDev(k)^2 = diag(sum_j F^j Q F^jT), within 6 standard errors of 8M runs."""
    s = _statod()
    runs, steps, seed = 8 << 20, 1086, 2016
    args = (s["x0"], s["P0"], s["F"], s["G"], s["H"], s["Q"], s["R"])
    truth = ga.ShardedBatch(k.VANILLA_PREDICT, *args, runs, devices=[0] * 8, noise=k.NOISE_AWGN, seed=seed)
    assert truth.shards() == 8 and [truth.first(g) for g in range(9)] == [g << 20 for g in range(9)]
    mc = truth.monte_carlo(steps, np.zeros((1, 2)))
    assert not truth.used_rccl()
    del truth
    one = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, *args, nfilters=runs, noise=k.NOISE_AWGN, seed=seed)
    mc1 = ga.new_monte_carlo_runs(runs, steps, 2, np.zeros((1, 2)), one, keep_runs=False)
    del one
    x, P = s["x0"].copy(), np.zeros((4, 4))
    for t in range(steps):
        x, P = s["F"] @ x, s["F"] @ P @ s["F"].T + s["Q"]
        if t in (0, 1, 10, 100, 500, steps - 1):
            sd = np.sqrt(np.diag(P))
            assert np.allclose(mc.mean(t), mc1.mean(t), rtol=1e-12, atol=1e-13 * np.max(sd)), t
            assert np.allclose(mc.stddev(t), mc1.stddev(t), rtol=1e-9), t
            for m in (mc, mc1):
                assert np.all(np.abs(m.mean(t) - x) <= 6 * sd / np.sqrt(runs) + 1e-12 * np.abs(x)), (t, m.mean(t), x)
                assert np.all(np.abs(m.stddev(t) / sd - 1.0) <= 6 / np.sqrt(2 * runs)), (t, m.stddev(t), sd)
