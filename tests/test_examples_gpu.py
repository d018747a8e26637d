"""The reference's three example programs re-hosted on the engine (examples/*.py): jerkcar must reproduce the
reference's committed CSVs row for row in the exporter's %f format; robot and statOD5044 must produce
statistically consistent filters (E[NIS] = p, E[NEES] = n for a consistent filter)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))
pytestmark = pytest.mark.gpu


def test_jerkcar_example_reproduces_reference_csv_rows(tmp_path):
    import jerkcar as ex
    from tests import jerkcar as jc
    paths = ex.main(str(tmp_path))
    for name in ("vanilla", "information", "sqrt"):
        ours = [l.strip() for l in open(paths[name]) if l.strip() and not l.startswith("#")]
        ref = [l.strip() for l in open(os.path.join(jc.GOLDEN, name + ".csv")) if l.strip() and not l.startswith("#")]
        assert ours[0] == ref[0]                               # header line, exporter.go:60-91
        assert len(ours) == len(ref) == 2002
        a = np.array([[float(v) for v in l.split(",")] for l in ours[1:]])
        b = np.array([[float(v) for v in l.split(",")] for l in ref[1:]])
        assert np.max(np.abs(a - b)) <= 1.01e-6                 # both sides are rounded to 6 decimals
        same = sum(x == y for x, y in zip(ours[1:], ref[1:]))
        assert same >= 0.98 * (len(ref) - 1), (name, same)      # identical text but for last-digit rounding ties


def test_robot_example_chisquare_is_consistent(tmp_path):
    import robot as ex
    out = ex.main(str(tmp_path), runs=2048)
    assert 0.8 < out["nis_mean"] < 1.25          # p = 1
    assert os.path.exists(tmp_path / "chisquare.csv")
    # runs.AsCSV(headers) (examples/robot/main.go:43-47, montecarlo.go:62-89): every run's column, then mean and stddev
    lines = open(tmp_path / "montecarlo-xi.csv").read().split("\n")
    assert len(lines) == ex.STEPS + 1
    assert lines[0].startswith("xi-0,xi-1,") and lines[0].endswith("xi-2047,xi-mean,xi-stddev")
    row = np.array([float(v) for v in lines[ex.STEPS].split(",")])
    assert row.shape == (2048 + 2,)
    assert abs(row[:-2].mean() - row[-2]) <= 1.01e-6 and abs(row[:-2].std(ddof=1) - row[-1]) <= 1.01e-6   # all rounded to %f


def test_statod5044_example_runs_end_to_end(tmp_path):
    import statod5044 as ex
    out = ex.main(str(tmp_path), runs=64)
    hdr = open(tmp_path / "mc-ctrl-dr.csv").readline().strip().split(",")   # AsCSV: main.go:87-91
    assert hdr[0] == "dr-0" and hdr[-3:] == ["dr-63", "dr-mean", "dr-stddev"]
    assert np.all(np.isfinite(out["mc_stddev_last"])) and np.all(out["mc_stddev_last"] > 0)
    # every filter must follow the oracle (reference-order CPU restatement) run on the same measurements
    from oracle import oracle as orc
    kinds = {"vanilla": (orc.VANILLA, ex.x0, ex.P0), "information": (orc.INFORMATION, np.zeros(4), np.zeros((4, 4))),
             "sqrt": (orc.SQUAREROOT, ex.x0, ex.P0)}
    for name, (kind, xi, Pi) in kinds.items():
        assert np.all(np.isfinite(out["rms"][name]))
        f = orc.Filter.ldkf(kind, xi, Pi, ex.Fcl, ex.Gcl, ex.H, ex.Q, ex.R)
        worst = 0.0
        for s in range(ex.SAMPLES):
            assert f.update(out["measurements"][s], np.zeros(2)) == orc.OK
            worst = max(worst, np.linalg.norm(f.state() - out["history"][name][s]) / max(np.linalg.norm(f.state()), 1e-300))
        assert worst <= 1e-9, (name, worst)
    assert len(open(tmp_path / "chisquare.csv").readlines()) == ex.SAMPLES + 1


def test_robot_example_in_cpp_writes_the_same_files_as_the_python_one(tmp_path):
    """examples/robot.cpp is examples/robot/main.go transliterated onto include/gokalman_amd.hpp (the compiled-language twin of the Go
    shim: NewAWGN, NewPurePredictorVanilla, NewVanilla, NewMonteCarloRuns(sims, steps, 1, controls, mcKF), runs.AsCSV(headers),
    NewChiSquare(chiKF, runs, controls, true, true)).  Same seed and initial state as examples/robot.py: the three CSV files are
    identical byte for byte (the same C ABI calls, the same device noise streams, the same %f formatting)."""
    import subprocess
    import robot as ex
    exe = str(tmp_path / "robot_cpp")
    lib = os.path.join(ROOT, "gokalman_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "robot.cpp"),
                    "-L" + lib, "-lgokalman_amd", "-Wl,-rpath," + lib, "-o", exe], check=True)
    py_dir, cpp_dir = tmp_path / "py", tmp_path / "cpp"
    os.makedirs(cpp_dir)
    ex.main(str(py_dir), runs=50, seed=1)
    mc_x0 = np.linalg.cholesky(ex.P0) @ np.random.default_rng(1).standard_normal(2)      # robot.py's draw for seed 1
    r = subprocess.run([exe, str(cpp_dir), "1", "%.17g" % mc_x0[0], "%.17g" % mc_x0[1], "50"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for name in ("montecarlo-xi.csv", "montecarlo-xi_dot.csv", "chisquare.csv"):
        a, b = open(py_dir / name).read(), open(cpp_dir / name).read()
        assert a == b, name
    assert len(open(cpp_dir / "chisquare.csv").read().split("\n")) == ex.STEPS + 2
