"""The C++ host mirror (include/gokalman_amd.hpp) replays the reference's jerkcar scenario the way
examples/jerkcar/main.go drives the Go interface; rows are compared with the reference's CSVs."""
import os
import subprocess

import numpy as np
import pytest

from tests import jerkcar as jc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = "/tmp/gokalman_amd_jerkcar_host"


def _build():
    lib = os.path.join(ROOT, "gokalman_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "jerkcar_host.cpp"),
           "-o", EXE, "-L" + lib, "-lgokalman_amd", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def _run(kind):
    g = jc.GOLDEN
    return subprocess.run([EXE, kind, os.path.join(g, "uvec.csv"), os.path.join(g, "yacchist.csv"), os.path.join(g, "yposhist.csv")],
                          capture_output=True, text=True)


def test_cpp_host_builds_and_fails_loudly_without_gpu():
    import torch
    _build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    res = _run("vanilla")
    assert res.returncode == 3 and "no CPU fallback" in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("kind,fixture", [("vanilla", "vanilla"), ("sqrt", "sqrt"), ("information", "information")])
def test_cpp_host_replays_jerkcar_fixture(kind, fixture):
    _build()
    res = _run(kind)
    assert res.returncode == 0, res.stderr
    got = np.array([[float(v) for v in line.split(",")] for line in res.stdout.strip().splitlines()])
    exp = jc.load_expected(fixture)
    assert got.shape == exp.shape
    assert np.max(np.abs(got - exp)) <= 5.1e-7
