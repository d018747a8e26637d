"""String() formatting helpers (gokalman_amd/strfmt.py): fmt's %v for float64 and the layout of mat64.Formatted, and the
reference's own format strings (vanilla.go:276-284, srif.go:283-289, noise.go:62-64)."""
import numpy as np

from gokalman_amd import strfmt


def test_go_v_matches_fmt_percent_v():
    # strconv 'g', shortest digits: exponent form from 1e6 on (fmt.Println(1e6) prints 1e+06) and below 1e-4
    cases = {1.0: "1", 0.45: "0.45", 1e-5: "1e-05", 123456789.0: "1.23456789e+08", 1e21: "1e+21", 1e20: "1e+20", 1e6: "1e+06",
             999999.0: "999999", 123456.789: "123456.789", 1234567.0: "1.234567e+06", -2.5e7: "-2.5e+07",
             0.0001: "0.0001", -2.5: "-2.5", 1 / 3: "0.3333333333333333", 5e-324: "5e-324", 100.0: "100", 0.0: "0",
             float("inf"): "+Inf", float("-inf"): "-Inf", 1.5e-7: "1.5e-07", 12345.678: "12345.678"}
    for x, want in cases.items():
        assert strfmt.go_v(x) == want, (x, strfmt.go_v(x), want)
    assert strfmt.go_v(float("nan")) == "NaN"
    # round trip of the positional / exponent forms
    rng = np.random.default_rng(1)
    for x in np.concatenate([rng.standard_normal(200) * 10.0 ** rng.integers(-12, 25, 200), [2.0 ** -1074, 2.0 ** 1023]]):
        assert float(strfmt.go_v(x).replace("+Inf", "inf")) == x


def test_formatted_layout():
    assert strfmt.formatted(np.array([[1, 2.5], [3, 4]]), "  ") == "⎡  1  2.5⎤\n  ⎣  3    4⎦"
    assert strfmt.formatted(np.array([1, 2, 3.25]), "  ") == "⎡   1⎤\n  ⎢   2⎥\n  ⎣3.25⎦"
    assert strfmt.formatted(np.array([[1, 2.5]]), "  ") == "[  1  2.5]"
    assert strfmt.formatted(np.array([7.0]), "") == "[7]"


def test_estimate_and_noise_strings_follow_the_reference_format_strings():
    x, y, P, K, i = np.array([1.0, 2.0]), np.array([0.5]), np.eye(2), np.array([[0.1], [0.2]]), np.array([0.25])
    s = strfmt.estimate_string("vanilla", x, y, P, K, P * 2, i)
    assert s.startswith("{\ns=⎡1⎤\n  ⎣2⎦\ny=[0.5]\nP=⎡1  0⎤\n  ⎣0  1⎦\nK=⎡0.1⎤\n  ⎣0.2⎦\nP-=⎡2  0⎤\n   ⎣0  2⎦\ni=[0.25]\n}")
    assert "K=" not in strfmt.estimate_string("information", x, y, P, None, P, x)
    assert strfmt.estimate_string("srif", x, y, P, None, P, None).endswith("P-=⎡1  0⎤\n   ⎣0  1⎦\n}")
    assert strfmt.noise_string("noiseless", np.eye(1), np.eye(1)) == "Noiseless{\nQ=[1]\nR=[1]}\n"
    assert strfmt.noise_string("awgn", np.eye(1), np.eye(1)).startswith("AWGN{")
    assert strfmt.noise_string("batch", None, None) == "BatchNoise"
    assert strfmt.filter_string("hybrid", None, None, None, "BatchNoise", 7) == "HybridKF [k=7]\nBatchNoise"
    assert strfmt.filter_string("information", np.eye(1), None, np.eye(1), "N").startswith("inv(F)=[1]\nG=<nil>\nH=[1]\nN")
