"""The LAPACK boundary of the reference — LAPACK.potrf!('U', ·) / LAPACK.potrs!('U', ·, ·) at
src/backward_pass.jl:69-73 — held against the REAL LAPACK (scipy.linalg.lapack.dpotrf / dpotrs, OpenBLAS).

Stated bars, on 1 000 random SPD matrices per size (condition number ≤ ~1e3):
    potrf  m ∈ {1, 2}  (acrobot: nu = 1, car: nu = 2 — the BASELINE headline configs)   bitwise
           m ∈ {5, 8}  OpenBLAS accumulates its dot products in SIMD/FMA order            ≤ 5e-12 · max|U|
    potrs  m = 1       (OpenBLAS's trsm multiplies by the inverted diagonal; so does the oracle)   bitwise
           m ∈ {2, 5, 8}  on LAPACK's own factor                                           ≤ 1e-13 · max|X|
and the reference's behaviour on a matrix that is NOT positive definite (return code ignored, src/backward_pass.jl:69).
"""
import numpy as np
import pytest
from scipy.linalg import lapack


def _spd(rng, m):
    A = rng.standard_normal((m, m + 3))
    return A @ A.T + 0.1 * np.eye(m)


def _orc(oracle, S, Bm):
    L = oracle.lib()
    m = S.shape[0]
    a = np.asfortranarray(S.copy())
    info = L.orc_potrf_U(a.ctypes.data_as(oracle.c_double_p), m)
    b = np.asfortranarray(Bm.copy())
    L.orc_potrs_U(a.ctypes.data_as(oracle.c_double_p), m, b.ctypes.data_as(oracle.c_double_p), Bm.shape[1])
    return a, b, info


@pytest.mark.parametrize("m", [1, 2, 5, 8])
def test_oracle_potrf_potrs_vs_scipy_lapack(oracle, m):
    rng = np.random.default_rng(100 + m)
    iu = np.triu_indices(m)
    worst_u, worst_x, worst_end = 0.0, 0.0, 0.0
    for _ in range(1000):
        S = _spd(rng, m)
        Bm = rng.standard_normal((m, 4))
        U, info = lapack.dpotrf(np.asfortranarray(S), lower=0, clean=0)
        X, _ = lapack.dpotrs(U, np.asfortranarray(Bm), lower=0)
        a, b, oinfo = _orc(oracle, S, Bm)
        assert info == 0 and oinfo == 0
        if m <= 2:
            assert np.array_equal(U[iu], a[iu])
        worst_u = max(worst_u, np.abs(U[iu] - a[iu]).max() / np.abs(U).max())
        # solve with LAPACK's own factor so that only the triangular solves are compared
        b2 = np.asfortranarray(Bm.copy())
        oracle.lib().orc_potrs_U(np.asfortranarray(U).ctypes.data_as(oracle.c_double_p), m,
                                 b2.ctypes.data_as(oracle.c_double_p), 4)
        if m == 1:
            assert np.array_equal(X, b2) and np.array_equal(X, b)
        worst_x = max(worst_x, np.abs(X - b2).max() / np.abs(X).max())
        worst_end = max(worst_end, np.abs(X - b).max() / np.abs(X).max())       # factor + solve, end to end
    assert worst_u <= 5e-12, worst_u
    assert worst_x <= 1e-13, worst_x
    assert worst_end <= 1e-10, worst_end


def test_failed_factorisation_matches_lapack_info(oracle):
    """Quu not positive definite: LAPACK returns info = j (first bad leading minor) and the reference carries on
    (src/backward_pass.jl:69). The oracle reports the same info and leaves the same leading rows."""
    S = np.array([[4.0, 2.0, 1.0], [2.0, 1.0, 3.0], [1.0, 3.0, 5.0]])        # 2x2 leading minor is singular
    U, info = lapack.dpotrf(np.asfortranarray(S), lower=0, clean=0)
    a, _, oinfo = _orc(oracle, S, np.ones((3, 1)))
    assert info == oinfo == 2
    assert np.array_equal(U[0], a[0])                                          # the completed first row
