"""The `fence` subset of the GPU suite (tests/conftest.py: every padded shape, every partial-tile / tail case, one sequence per kernel
family -- ~750 of the ~800 GPU tests, ~35 s) once more with every device block of a batch ending where its 2 MB mapping ends (KB_DEBUG_FENCE=1,
csrc/kb_api.hip dev_alloc): a kernel that reads or writes behind the last tile of a block -- the stand-in element of a padded shape,
a masked lane of a tail tile -- takes a memory fault there, where the parity tests on a roomy allocation see nothing (the SquareRoot
split kernel's stand-in for sqrt(R) at p = 2 was such a read: found by a 1M-filter bench, invisible to 80 green tests;
profiles/NOTES.md).  GPU AddressSanitizer is not available on the pool; this is the fence we have."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_gpu_suite_with_fenced_device_blocks():
    if os.environ.get("KB_DEBUG_FENCE"):
        pytest.skip("already inside the fenced run")
    env = dict(os.environ, KB_DEBUG_FENCE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu and fence", "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400)
    lines = r.stdout.splitlines()
    # what failed INSIDE goes first (the outer run shows one failing test and a tail): the nested test ids, the fault message if the
    # process died of one, the summary line -- and only then the last lines of the log
    inner = [ln for ln in lines if ln.startswith("FAILED ") or ln.startswith("ERROR ") or "Memory access fault" in ln or "HSA_STATUS" in ln or "Aborted" in ln]
    summary = [ln for ln in lines if " passed" in ln or " failed" in ln][-1:]
    tail = "\n".join(lines[-25:])
    head = "fenced run (KB_DEBUG_FENCE=1) exit code %d; inner failures: %s; %s" % (r.returncode, inner or "none reported (the process died: a fault?)", summary)
    assert r.returncode == 0, head + "\n" + tail
    assert " passed" in tail and "failed" not in tail, head + "\n" + tail
    npassed = int(summary[0].split(" passed")[0].split()[-1])
    assert npassed >= 600, "the fenced subset shrank to %d tests: " % npassed + head
