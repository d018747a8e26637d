import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fence: member of the subset tests/test_fence_gpu.py runs once more with fenced device blocks (KB_DEBUG_FENCE=1)")


def pytest_sessionstart(session):
    """The suites bind the in-tree C-ABI library: build it (hipcc, gfx950 cross-compile works without a GPU)
    and the C oracle (gcc) when they are missing or stale, exactly as __graft_entry__.build() does."""
    from gokalman_amd import build as kb_build
    from oracle import oracle as orc
    kb_build.build()
    orc.build()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# The fenced re-run (tests/test_fence_gpu.py) covers the cases in which a kernel can touch memory behind a block: every padded shape,
# every partial tile / tail size, one sequence per kernel family -- NOT the whole suite again (task 6: the nested
# full run had grown to 40 % of the driver's GPU-test step).  Left out: the bench launches (1M-filter batches, no tails), the host /
# example programs and the sharded ensembles (the same kernels on round sizes), and the long embedded-jerkcar replays beyond one
# per kind (2000 steps of the kernels the shape sweep already runs at every shape).
_FENCE_SKIP_FILES = ("test_fence_gpu.py", "test_bench_launch.py", "test_cpp_host.py", "test_distributed_gpu.py", "test_sharded_gpu.py",
                     "test_examples_gpu.py", "test_highprec_gpu.py")   # (the last: whole 64-filter tiles of shapes the sweeps run at tail sizes)


def _in_fence_subset(item):
    if "gpu" not in item.keywords:
        return False
    fname = os.path.basename(str(item.fspath))
    if fname in _FENCE_SKIP_FILES:
        return False
    if fname == "test_jerkcar_embedded_gpu.py":
        cs = getattr(item, "callspec", None)
        if cs is None:
            return True
        return cs.params.get("name") in ("vanilla", "sqrt") or cs.params.get("kk") == 2
    return True


def pytest_collection_modifyitems(config, items):
    for item in items:
        if _in_fence_subset(item):
            item.add_marker(pytest.mark.fence)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Every bound looser than 1e-9 with the error achieved against it (tests/achieved.py)."""
    import json
    from tests import achieved
    rows = achieved.summary()
    if not rows:
        return
    tr = terminalreporter
    tr.section("bounds looser than 1e-9: achieved / bound (tests/achieved.py)")
    for r in rows:
        tr.write_line("%-110s %-22s achieved %.2e  bound %.0e  (slack x%.1f)" % (r["test"][:110], r["label"][:22], r["achieved"], r["bound"], min(r["slack"], 9.9e99)))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "achieved_errors.json"), "w") as fh:
            json.dump(rows, fh, indent=1)
    except OSError:
        pass
