"""bench.py's final stdout line (gokalman_amd/benchline.py): bounded in size, a JSON round trip, the contract's keys present --
on a canned full document (round 5's 22.8 KB line, which the driver could not parse: BENCH_r05.parsed = null)."""
import json
import os

from gokalman_amd import benchline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANNED = os.path.join(ROOT, "profiles", "r05g", "bench_default_run.json")


def _doc():
    with open(CANNED) as fh:
        return json.load(fh)


def test_headline_line_is_small_and_round_trips():
    doc = _doc()
    assert len(json.dumps(doc)) > 20000          # the document that broke the driver's parser
    line = benchline.dumps(doc, "gpurun_out/bench_full.json")
    assert "\n" not in line and len(line) <= benchline.LIMIT == 8192
    assert len(line) <= 6144, "keep a margin: the driver's stdout tail is ~8 KB"
    back = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "ranks"):
        assert key in back, key
    assert back["value"] == doc["value"] and back["ms_per_step"] == doc["ms_per_step"]       # the headline figures are not rounded
    assert back["config"]["workload"].startswith("configs[1]")
    r = back["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert key in r, key
    assert abs(r["frac"] - doc["roofline"]["frac"]) < 1e-8 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert line.count("source_hash") == 1, "one copy of the counter provenance"
    assert back["cpu_baseline"] == doc["cpu_baseline"]
    assert back["parity"]["ok"] is True and back["parity"]["tolerance"] == 1e-9
    # per extra leg only {value, kernel_ms | ms_per_launch | seconds, frac, parity_ok, errors}
    for name in doc["extra"]:
        leg = back["legs"][name]
        assert set(leg) <= {"value", "kernel_ms", "ms_per_launch", "seconds", "frac", "bound", "parity_ok", "errors"}, (name, leg)
        assert abs(leg["value"] / doc["extra"][name]["value"] - 1) < 1e-5
    assert back["legs"]["srif_fp32"]["parity_ok"] is True
    assert back["full"] == "gpurun_out/bench_full.json"


def test_headline_refuses_to_grow_past_the_limit():
    doc = _doc()
    doc["extra"] = dict(doc["extra"])
    for i in range(200):
        doc["extra"]["leg_%03d" % i] = dict(doc["extra"]["squareroot"])
    try:
        benchline.dumps(doc)
    except ValueError as exc:
        assert "limit" in str(exc)
    else:
        raise AssertionError("an oversized line must raise, not print")


def test_every_fraction_is_gated_and_round_5s_fused_figure_is_caught():
    """fused.roofline.frac = 1.43 in that round's line (the AWGN kernel's VALU count priced the Noiseless
    leg).  The gate walks every `roofline` object; it is red on that document -- for that one entry only."""
    bad = benchline.fraction_violations(_doc())
    assert [p for p, _ in bad] == ["fused.roofline"], bad
    assert abs(bad[0][1] - 1.43) < 0.01
    compact = json.loads(benchline.dumps(_doc()))
    assert [p for p, _ in benchline.fraction_violations(compact)] == ["legs.fused"]
