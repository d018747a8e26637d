"""Sanitizer runs of the host-side C / C++ (CPU build only: GPU AddressSanitizer is not available on this pool).
  * the oracle (oracle/*.c) rebuilt with -fsanitize=address,undefined and driven through its whole golden / cross-pin / VanLoan
    suite in a subprocess (LD_PRELOAD of the ASan runtime, as a Python host needs);
  * the C++ host programs of tests/cpp (include/gokalman_amd.hpp: matrix plumbing, error paths, estimate / Monte-Carlo value types)
    built with the same flags; without a GPU they must stop at the library's "no CPU fallback" error with a clean report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def _clean(report):
    return "AddressSanitizer" not in report and "runtime error:" not in report and "LeakSanitizer" not in report


def test_oracle_under_asan_and_ubsan():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no libasan in this toolchain")
    so = os.path.join(ROOT, "oracle", "libgokalman_oracle_san.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "libgokalman_oracle_san.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, GOKALMAN_ORACLE_SO=so, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", OMP_NUM_THREADS="2",
               UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_oracle_golden.py",
                        "tests/test_srif_crosspin_cpu.py", "tests/test_vanloan_cpu.py", "tests/test_chisquare_property_cpu.py"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert _clean(r.stdout + r.stderr), (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("src,args", [("estimate_semantics.cpp", []), ("sharded_host.cpp", []), ("jerkcar_host.cpp", ["vanilla"])])
def test_cpp_hosts_under_asan_and_ubsan(src, args):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: host ASan next to the HIP runtime is not what this test is about")
    exe = "/tmp/gokalman_amd_san_" + src.replace(".cpp", "")
    lib = os.path.join(ROOT, "gokalman_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", src), "-o", exe,
                           "-L" + lib, "-lgokalman_amd", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"])
    if src == "jerkcar_host.cpp":
        from tests import jerkcar as jc
        args = args + [os.path.join(jc.GOLDEN, f) for f in ("uvec.csv", "yacchist.csv", "yposhist.csv")]
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 3 and "no CPU fallback" in r.stderr, (r.returncode, r.stderr[-2000:])   # the library refuses, the host reports it
    assert _clean(r.stderr), r.stderr[-4000:]
