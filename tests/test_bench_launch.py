"""bench.py's N-rank contract: `--gpus N` yields N ranks or a non-zero exit, never a
silently smaller job.  The CPU half checks the refusal; the GPU half runs two ranks on ONE GPU over gloo
(every rank drives its own FilterBatch shard through the C ABI; the epilogue all-reduce and the Monte-Carlo
statistics reduction of montecarlo.go:18-59 really run)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def _json_line(stdout):
    """The ONE stdout line: the compact headline object (gokalman_amd/benchline.py), bounded in size."""
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    assert len(lines[0]) <= 8192, len(lines[0])
    return json.loads(lines[0])


def _run_full(args, tmp_path, tag):
    """Runs bench.py with --full-out into tmp_path; returns (compact line, full document, CompletedProcess)."""
    full = str(tmp_path / ("bench_full_%s.json" % tag))
    r = _run(args + ["--full-out", full])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["full"] == full
    with open(full) as fh:
        doc = json.load(fh)
    # the line is the document's headline: same figures, nothing re-measured
    assert line["value"] == doc["value"] and line["ms_per_step"] == doc["ms_per_step"] and line["n_gpus"] == doc["n_gpus"]
    assert abs(line["roofline"]["frac"] - doc["roofline"]["frac"]) < 1e-8
    return line, doc, r


def test_more_ranks_than_gpus_is_refused_not_downgraded():
    import torch
    ndev = torch.cuda.device_count()
    if ndev >= 8:
        pytest.skip("8 GPUs visible: the request is satisfiable")
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--filters", "4096"], timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no bench line may be printed for a refused job"
    assert "GPU" in (r.stderr + r.stdout) or "MI355X" in (r.stderr + r.stdout)


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


# (--fused-steps 8: the rank-0-only time-fused legs run under world = 2 as well -- a collective inside one of them hangs or kills the job:
# round 6 found the SRIF fused leg calling timed_leg(), which holds barriers, from rank 0 alone)
SMALL = ["--steps", "30", "--warmup", "5", "--filters", "65536", "--fused-steps", "8", "--ooc-filters", "0",
         "--mc-runs", "16384", "--mc-steps", "64", "--mc-total", "65536", "--split-filters", "8192", "--hybrid-filters", "16384", "--sqrt-filters", "16384", "--srif-filters", "8192", "--shared-filters", "16384",
         "--no-cpu-baseline"]


@pytest.mark.gpu
def test_two_ranks_self_launched_over_gloo_on_one_gpu(tmp_path):
    # (the one-rank job initialises torch.distributed over RCCL: communicator set-up with device_id, device tensors in all_gather /
    # all_reduce / barrier, the Monte-Carlo statistics reduction -- no multi-GPU box in the test loop, so at least the RCCL code path
    # itself runs, with one rank; until round 6 a third bench launch of its own)
    line1, one, _ = _run_full(["--gpus", "1", "--init-dist", "--dist-backend", "nccl"] + SMALL, tmp_path, "one")
    assert one["n_gpus"] == 1 and one["ranks"]["backend"] == "nccl" and one["ranks"]["rccl_ranks_seen"] == 1
    assert "RCCL" in one["extra"]["mc"]["collective"]
    assert one["ranks"]["filter_steps_counted"] == 65536 * 30
    line2, two, _ = _run_full(["--gpus", "2", "--dist-backend", "gloo"] + SMALL, tmp_path, "two")
    assert line2["ranks"]["launched"] == 2 and line2["legs"]["vanilla_12x6"]["parity_ok"] and line1["parity"]["ok"]
    for leg in ("fused", "squareroot.fused", "srif_fp32.fused", "hybrid_ekf.fused"):   # rank 0's own legs, present and proven in both jobs
        assert line1["legs"][leg]["parity_ok"] and line2["legs"][leg]["parity_ok"], leg
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["ranks"]["launched"] == 2 and two["ranks"]["rccl_ranks_seen"] == 2
    assert len(two["ranks"]["per_rank_ms_per_step"]) == 2
    # every rank's 65536 filters x 30 steps were counted by the epilogue all-reduce
    assert two["ranks"]["filter_steps_counted"] == 2 * 65536 * 30
    assert two["filters_with_error_status"] == 0
    # value is the whole job: both ranks share one GPU here, so between ~1x and ~2x of the one-rank figure
    assert 0.4 * one["value"] < two["value"] < 2.6 * one["value"]
    # the Monte-Carlo statistics are those of the union of the shards: 2 x 16384 runs, same per-step spread
    mc1, mc2 = one["extra"]["mc"], two["extra"]["mc"]
    assert mc2["runs_total"] == 2 * mc1["runs_total"]
    for a, b in zip(mc1["stddev_last"], mc2["stddev_last"]):
        assert abs(a - b) <= 0.05 * abs(a)
    # the whole ensemble (--mc-total runs IN TOTAL, as consecutive shards per rank + one all-reduce) is the same ensemble on one rank
    # (four shards) and on two (two shards each): same runs, same noise, sums equal to rounding; and it is what montecarlo.go estimates
    e1, e2 = mc1["ensemble"], mc2["ensemble"]
    assert e1["runs_total"] == e2["runs_total"] == 65536 and e1["shards_per_rank"] == 4 and e2["shards_per_rank"] == 2
    assert e1["matches_covariance_recursion"] and e2["matches_covariance_recursion"]
    for a, b in zip(e1["stddev_last"], e2["stddev_last"]):
        assert abs(a - b) <= 1e-9 * abs(a)
    # every leg proves itself against the oracle on rank 0 (4096 filters x 20 steps), the fp32 SRIF leg reports its achieved error
    for leg in ("squareroot", "shared_model", "vanilla_12x6", "squareroot_12x6", "information_12x6", "vanilla_10x4", "hybrid_ekf", "srif_fp32"):
        assert one["extra"][leg]["parity"]["ok"], (leg, one["extra"][leg]["parity"])
    assert one["extra"]["srif_fp32"]["parity"]["achieved_max_rel_frobenius_R"] <= one["extra"]["srif_fp32"]["parity"]["tolerance"]
    assert two["extra"]["vanilla_12x6"]["filters_total"] == 2 * 8192 and two["extra"]["vanilla_12x6"]["filters_with_error_status"] == 0
    assert two["extra"]["hybrid_ekf"]["filters_total"] == 2 * 16384
    assert two["extra"]["squareroot"]["filters_total"] == 2 * 16384 and two["extra"]["squareroot"]["filters_with_error_status"] == 0
    assert two["extra"]["srif_fp32"]["filters_total"] == 2 * 8192 and two["extra"]["srif_fp32"]["filters_with_error_status"] == 0
    assert two["extra"]["shared_model"]["filters_total"] == 2 * 16384 and two["extra"]["shared_model"]["filters_with_error_status"] == 0
    assert two["roofline"]["frac"] <= 1.0 and one["roofline"]["frac"] <= 1.0
    # strong scaling (GPU g owns [g N / G, (g + 1) N / G)): the same 65536 filters split over the two ranks
    assert two["strong_scaling"]["filters_per_gpu"] == [32768, 32768] and two["strong_scaling"]["filters_total"] == 65536
    assert len(two["strong_scaling"]["per_rank_ms_per_step"]) == 2 and two["strong_scaling"]["value"] > 0
    assert one["strong_scaling"]["filters_per_gpu"] == [65536]
    # the chi-square sums are all-reduced like the Monte-Carlo means
    assert two["extra"]["chisq"]["runs_total"] == 2 * one["extra"]["chisq"]["runs_total"] and 0.5 < two["extra"]["chisq"]["nis_mean"] < 5.0
    # parity against the oracle travels in the line (rank 0), and so does the repetition of the timed block
    assert one["parity"]["ok"] and two["parity"]["ok"] and one["parity"]["filters"] == 4096
    assert one["repetitions"]["blocks"] == 5 and len(one["repetitions"]["ms_per_step_blocks"]) == 5
    assert one["host_path"]["value"] < one["value"]


@pytest.mark.gpu
def test_bench_line_roofline_is_physical(tmp_path):
    """Default sizes (every fraction in the line is quoted at them); EVERY roofline object of the full document and every leg of
    the compact line carries 0 < frac <= 1 (task 2: the fused leg said 1.43)."""
    from gokalman_amd import benchline
    line, out, _ = _run_full(["--steps", "50", "--warmup", "5", "--ooc-filters", "0", "--mc-total", "0", "--no-cpu-baseline"], tmp_path, "phys")
    assert benchline.fraction_violations(out) == [] and benchline.fraction_violations(line) == []
    assert 0.0 < out["extra"]["squareroot"]["roofline"]["frac"] <= 1.0   # config C on the default 1M filters
    fr = out["fused"]["roofline"]
    assert fr["bound"] == "valu_issue"
    if fr["source"]["matches_sources"]:   # the VALU counts of profiles/valu_latest.json were measured on these very kernel sources
        assert 0.55 < fr["frac"] <= 1.0
        assert 0.55 < out["fused"]["awgn"]["roofline"]["frac"] <= 1.0
    else:                                 # ... or no figure at all: never a stale count against a new kernel
        assert fr["frac"] is None and "roofline" not in out["fused"]["awgn"]
    for leg in ("mc", "chisq", "hybrid_ekf", "squareroot", "shared_model", "vanilla_12x6", "squareroot_12x6", "information_12x6",
                "vanilla_10x4", "vanilla_16x8", "srif_fp32"):
        roof = out["extra"][leg].get("roofline")
        if roof is None:   # issue-bound legs (mc, chisq) quote a VALU count only while it was measured on these kernel sources
            assert leg in ("mc", "chisq") and not fr["source"]["matches_sources"], leg
            continue
        assert 0.0 < roof["frac"] <= 1.0, leg
        assert 0.0 < line["legs"][leg]["frac"] <= 1.0, leg
    roof = out["roofline"]
    assert 0.0 < roof["frac"] <= 1.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert roof["frac_algorithmic"] > roof["frac"]      # full-matrix convention overcounts the packed traffic
    assert roof["bytes_convention"]["algorithmic_bytes_per_filter_step"] == 1488
    assert roof["bytes_convention"]["moved_bytes_per_filter_step"] == 1104
    assert "traffic_source" in roof
    assert 0.0 < roof["dram_frac_lower_bound"] < roof["frac"] and "Infinity-Cache" in roof["side"]
