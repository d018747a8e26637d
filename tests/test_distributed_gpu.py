"""GPU twin of tests/test_distributed_cpu.py: a world_size-2 job (gloo, both ranks on the visible GPU) in which every
rank drives REAL FilterBatch shards through the C ABI.  The all-reduced Monte-Carlo sums must equal one process
running all runs (a run's noise depends only on its global index, montecarlo.go:92-119 semantics with first_run
offsets), and the sharded Vanilla batch must equal the unsharded one bit for bit (filters share nothing)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS, STEPS, NF, T = 5000, 40, 3001, 6   # odd sizes: uneven shards, partial tiles


@pytest.mark.gpu
def test_two_ranks_drive_filterbatch_shards(tmp_path):
    import gokalman_amd as ga
    from gokalman_amd import _capi as k
    from gokalman_amd import synth
    from bench import STATOD

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "mc_shard_worker.py"), out, str(RUNS), str(STEPS), str(NF), str(T)],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)

    st = {kk: np.array(v, dtype=np.float64) for kk, v in STATOD.items()}
    kf = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, st["x0"], st["P0"], st["F"], st["G"], st["H"], st["Q"], st["R"],
                                 nfilters=RUNS, noise=k.NOISE_AWGN, seed=99)
    one = ga.new_monte_carlo_runs(RUNS, STEPS, 2, np.zeros((1, 2)), kf)
    two = ga.MonteCarloRuns(RUNS, STEPS, 4, got["sums"])
    for t in range(STEPS):
        np.testing.assert_allclose(two.mean(t), one.mean(t), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(two.stddev(t), one.stddev(t), rtol=1e-9)

    d = synth.linear_batch(NF, 6, 3, T, seed=1234)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    for t in range(T):
        b.update(d["y"][t])
    assert np.array_equal(got["x"], b.get(k.STATE))
    assert np.array_equal(got["P"], b.get(k.COVAR))
