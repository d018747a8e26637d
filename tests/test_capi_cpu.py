"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol the header
declares, and refuses to run without a GPU (there is no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gokalman_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 30
    lib = ctypes.CDLL(k.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_every_declared_symbol():
    assert sorted(k.SIGNATURES) == _declared_functions()
    assert k.lib().kb_version().startswith(b"gokalman_amd")


def test_product_never_imports_the_oracle():
    """The product path must not import, include, link or call anything under oracle/."""
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#include\s+\"[^\"]*oracle)|libgokalman_oracle|\borc_[a-z_]+\s*\(", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gokalman_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not pat.search(src), (dirpath, f)


def test_no_cpu_fallback_without_gpu():
    if k.lib().kb_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(ga.KalmanError) as e:
        ga.FilterBatch(k.VANILLA, 6, 3, 0, 16)
    assert e.value.code == k.ERR_NO_DEVICE and "no CPU fallback" in e.value.message


def test_argument_validation_before_device_use():
    h = ctypes.c_void_p()
    assert k.lib().kb_create(ctypes.byref(h), 99, 6, 3, 0, 1, k.F64, 0, 0) == k.ERR_INVALID
    assert k.lib().kb_create(ctypes.byref(h), k.VANILLA, 17, 3, 0, 1, k.F64, 0, 0) == k.ERR_DIMS
    assert k.lib().kb_create(ctypes.byref(h), k.VANILLA, 6, 3, 0, 0, k.F64, 0, 0) == k.ERR_INVALID
    assert b"nfilters" in k.lib().kb_last_error()


def test_constructor_dimension_errors_are_the_reference_strings():
    # vanilla_test.go:9-27: checkMatDims messages (helper.go:99-130)
    with pytest.raises(ga.KalmanError, match=r"dimensions must agree: x0\(3x\.\.\.\) Covar0\(\.\.\.x2\)"):
        ga.FilterBatch.new_ldkf(k.VANILLA, np.zeros(3), np.eye(2), np.eye(3), None, np.ones((1, 3)), np.eye(3), np.eye(1))
    with pytest.raises(ga.KalmanError, match=r"dimensions must agree: H\(\.\.\.x2\) x0\(3x\.\.\.\)"):
        ga.FilterBatch.new_ldkf(k.VANILLA, np.zeros(3), np.eye(3), np.eye(3), None, np.ones((1, 2)), np.eye(3), np.eye(1))


def test_mc_stats_match_gonum_formulas():
    """kb_mc_stats is host arithmetic: mean = c + sum(d)/N, unbiased stddev (montecarlo.go:18-59)."""
    from oracle import oracle as orc
    rng = np.random.default_rng(3)
    runs, steps, n = 500, 7, 4
    c = rng.standard_normal((steps, n)) * 5
    x = c[:, None, :] + 1e-7 * rng.standard_normal((steps, runs, n))
    d = x - c[:, None, :]
    sums = np.stack([d.sum(axis=1), (d * d).sum(axis=1), c], axis=1)
    mc = ga.MonteCarloRuns(runs, steps, n, sums)
    for t in range(steps):
        mean, std = orc.mc_mean_stddev(x[t])
        assert np.allclose(mc.mean(t), mean, rtol=1e-13)
        assert np.allclose(mc.stddev(t), std, rtol=1e-6)
