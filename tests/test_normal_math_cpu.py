"""csrc/kb_normal.h: the logarithm and the sine / cosine of the Box-Muller transform that the kernels and the host replay
(kb_noise_sample) share.  Compiled here with g++ (it is plain C++), checked against long double over 4.4M arguments of the form the
generator produces (k 2^-53), and a few results pinned bit for bit: the same bits must come out of the device
(tests/test_vanilla_gpu.py::test_device_normals_are_bit_identical_to_the_host_replay)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_normal_math_accuracy_and_pinned_bits(tmp_path):
    exe = str(tmp_path / "normal_math")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "normal_math.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()
    vals = {}
    for line in out:
        f = line.split()
        vals.setdefault(f[0], []).append(f[1:])
    assert float(vals["worst_neg2log_ulp"][0][0]) <= 1.0          # measured 0.835
    assert float(vals["worst_sincos_ulp"][0][0]) <= 2.0           # measured 1.59 (relative, each of sin and cos)
    assert vals["neg2log_of_1"][0][0] == "0"                      # +0, not -0: the radius of u = 1 is sqrt(+0)
    sc = {v[0]: (v[1], v[2]) for v in vals["sincos"]}
    assert sc["0"] == ("0000000000000000", "3ff0000000000000")
    assert sc["0.25"][0] == "3ff0000000000000" and sc["0.5"][1] == "bff0000000000000" and sc["0.75"][0] == "bff0000000000000"
    assert sc["0.125"] == ("3fe6a09e667f3bcd", "3fe6a09e667f3bcc")   # sqrt(1/2) and its neighbour: <= 1 ulp each
    nl = {v[0]: v[1] for v in vals["neg2log"]}
    assert nl["0.5"] == "3ff62e42fefa39ef" and nl["0.25"] == "40062e42fefa39ef"   # 2 ln 2, 4 ln 2 correctly rounded
    assert nl["1.1102230246251565e-16"] == "40525e4f7b2737fa"                     # u = 2^-53: the largest radius, -2 ln u = 73.47...
