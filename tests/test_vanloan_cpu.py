"""Pins of the VanLoan oracle (oracle/vanloan_oracle.c, restating c2d.go:13-75): the reference's own test
(c2d_test.go:9-33), the closed form of the double integrator, scipy's expm, LAPACK's eigenvalues."""
import numpy as np
import scipy.linalg

from oracle import oracle as orc


def test_reference_vanloan_test_vectors():
    # c2d_test.go:10-27
    rc, F, Q = orc.van_loan([[0, 1], [0, 0]], [[0], [1]], [[1]], 0.1)
    assert rc == 0
    assert np.allclose(F, [[1, 0.1], [0, 1]], atol=1e-3, rtol=1e-3)
    assert np.allclose(Q, [[0.0003, 0.005], [0.005, 0.1]], atol=1e-3, rtol=1e-3)
    # c2d_test.go:29-32: Nyquist error
    rc, _, _ = orc.van_loan([[1, 1], [0, 1]], [[0], [1]], [[1]], 10)
    assert rc & 1


def test_double_integrator_closed_form():
    for dt in (0.01, 0.1, 1.0, 3.0):
        for w in (1.0, 2.5):
            rc, F, Q = orc.van_loan([[0, 1], [0, 0]], [[0], [1]], [[w]], dt)
            assert rc == 0
            assert np.allclose(F, [[1, dt], [0, 1]], rtol=1e-14, atol=1e-15)
            assert np.allclose(Q, w * np.array([[dt ** 3 / 3, dt ** 2 / 2], [dt ** 2 / 2, dt]]), rtol=1e-13, atol=1e-16)


def test_expm_against_scipy_all_pade_orders():
    rng = np.random.default_rng(5)
    for n in (2, 4, 8, 12):
        for scale in (1e-3, 0.05, 0.3, 1.0, 4.0, 30.0):     # covers Pade 3/5/7/9/13 and squaring
            A = rng.standard_normal((n, n)) * scale / n
            E, ref = orc.expm(A), scipy.linalg.expm(A)
            assert np.linalg.norm(E - ref) <= 1e-12 * np.linalg.norm(ref), (n, scale)


def test_eigvals_against_lapack():
    rng = np.random.default_rng(6)
    last_same = 0
    for trial in range(200):
        n = int(rng.integers(2, 9))
        A = rng.standard_normal((n, n))
        rc, w = orc.eigvals(A)
        assert rc == 0
        ref = np.linalg.eigvals(A)
        # same multiset
        key = lambda z: (round(z.real, 9), round(z.imag, 9))
        assert np.allclose(sorted(w, key=key), sorted(ref, key=key), atol=1e-9), trial
        last_same += abs(abs(w[-1]) - abs(ref[-1])) <= 1e-9 * max(1.0, abs(ref[-1]))
    # which eigenvalue comes last is an artefact of the QR iteration (PARITY UNPINNED, see vanloan_oracle.c);
    # on dense random matrices the restatement picks the same one as LAPACK's dgeev in the large majority of cases
    assert last_same >= 150, last_same


def test_vanloan_matches_block_exponential():
    rng = np.random.default_rng(7)
    for n, q in ((2, 1), (3, 2), (4, 2), (6, 3)):
        A = rng.standard_normal((n, n)) * 0.5
        G = rng.standard_normal((n, q))
        L = rng.standard_normal((q, q)); W = L @ L.T
        dt = 0.1
        rc, F, Q = orc.van_loan(A, G, W, dt)
        M = np.zeros((2 * n, 2 * n))
        M[:n, :n], M[:n, n:], M[n:, n:] = -A * dt, G @ W @ G.T * dt, A.T * dt
        E = scipy.linalg.expm(M)
        Fref = E[n:, n:].T
        assert np.allclose(F, Fref, rtol=1e-12, atol=1e-14)
        assert np.allclose(F, scipy.linalg.expm(A * dt), rtol=1e-12, atol=1e-14)
        Qref = Fref @ E[:n, n:]
        assert np.allclose(Q, np.triu(Qref) + np.triu(Qref, 1).T, rtol=1e-11, atol=1e-14)
