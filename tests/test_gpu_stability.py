"""GPU leg of the stability tests ("next round" item 1): every solve that sits on the library's TRSM is
held to the backward-error bound that the reference's substitution-based solver delivers
(faer/src/linalg/triangular_solve.rs:98-198,452-484) on ill-conditioned inputs -- graded / Kahan / growth / factor
triangles, right-hand sides with small solutions (stability_cases.py).  The same cases and bounds are run against
the oracle on the CPU (test_stability_oracle.py)."""
import functools

import numpy as np
import pytest

import stability_cases as sc
from gpu_util import init_gpu, to_dev, to_host

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps
C_TRI = 4.0   # componentwise: |T X - B| <= C_TRI n eps (|T| |X| + |B|)
C_NORM = 8.0  # normwise: ||A X - B|| <= C_NORM n eps ||A|| ||X||


@functools.lru_cache(maxsize=None)
def _triangle(kind, n):
    return sc.triangle(kind, n, np.random.default_rng(n + len(kind)))


@functools.lru_cache(maxsize=None)
def _ill(n, cond, spd):
    return sc.ill_conditioned(n, cond, np.random.default_rng(n + int(spd)), spd=spd)


@pytest.mark.parametrize("kind", sc.TRI_KINDS)
@pytest.mark.parametrize("n", [100, 129, 700, 2000])
@pytest.mark.parametrize("small_solution", [False, True])
@pytest.mark.parametrize("upper,order", [(False, "F"), (True, "F"), (False, "C")])
def test_trsm_is_backward_stable(oracle, kind, n, small_solution, upper, order):
    if n == 2000 and (upper or order == "C"):
        pytest.skip("the large size runs once per kind (host-side reference arithmetic dominates the test time)")
    F = init_gpu()
    rng = np.random.default_rng(n + len(kind))
    t = _triangle(kind, n)
    k = 70 if n <= 700 else 9
    b = sc.tri_rhs(t, k, rng, small_solution)
    ref = b.copy(order="F")
    with np.errstate(all="ignore"):
        oracle.trsm(np.asfortranarray(t), ref, upper=False, unit=sc.is_unit(kind))
    if not np.isfinite(ref).all():
        pytest.skip("solution overflows")
    unit = sc.is_unit(kind)
    if upper:  # the same system written with an upper triangle: reverse rows and columns (triangular_solve.rs:578-604)
        tt, bb = np.ascontiguousarray(t[::-1, ::-1]), np.ascontiguousarray(b[::-1])
    else:
        tt, bb = t, b
    fn = {(False, False): F.solve_lower_triangular_in_place, (True, False): F.solve_upper_triangular_in_place,
          (False, True): F.solve_unit_lower_triangular_in_place, (True, True): F.solve_unit_upper_triangular_in_place}
    dx = to_dev(bb, order)
    fn[(upper, unit)](to_dev(tt, order), dx)
    got = to_host(dx)
    assert np.isfinite(got).all()
    assert sc.tri_backward_error(tt, got, bb) <= C_TRI * n * EPS
    # and parity with the oracle where the problem is well enough conditioned for a forward comparison to mean anything
    if kind in ("graded", "random_lu"):
        g = got[::-1] if upper else got
        assert np.abs(g - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("dtype", [np.float32])
@pytest.mark.parametrize("kind", ["mixed", "kahan", "growth"])
@pytest.mark.parametrize("n", [129, 700])
def test_trsm_is_backward_stable_f32(oracle, kind, n, dtype):
    F = init_gpu()
    rng = np.random.default_rng(n)
    t = sc.triangle(kind, n, rng)
    if kind == "mixed":  # keep the grading inside fp32's range of usefulness
        t = np.diag(np.logspace(0, -5, n)) + 1e-2 * np.tril(rng.standard_normal((n, n)), -1)
    t = t.astype(dtype)
    x0 = rng.standard_normal((n, 33)).astype(dtype)
    b = (t.astype(np.float64) @ x0.astype(np.float64)).astype(dtype)
    unit = sc.is_unit(kind)
    ref = b.copy(order="F")
    with np.errstate(all="ignore"):
        oracle.trsm(np.asfortranarray(t), ref, upper=False, unit=unit)
    if not np.isfinite(ref).all():
        pytest.skip("the substitution itself overflows in fp32 (error growth 2^k)")
    dx = to_dev(b)
    (F.solve_unit_lower_triangular_in_place if unit else F.solve_lower_triangular_in_place)(to_dev(t), dx)
    got = to_host(dx)
    assert np.isfinite(got).all()
    assert sc.tri_backward_error(t.astype(np.float64), got.astype(np.float64), b.astype(np.float64)) <= C_TRI * n * np.finfo(dtype).eps


@pytest.mark.parametrize("n", [100, 129, 700, 2000])
@pytest.mark.parametrize("cond", [1e8, 1e13])
def test_partial_piv_lu_solve_ill_conditioned(n, cond):
    """lu/partial_pivoting/solve.rs: P A = L U, x = U^-1 L^-1 P b; b = A x0 so that the solution is small"""
    F = init_gpu()
    rng = np.random.default_rng(n)
    a = _ill(n, cond, False)
    x0 = rng.standard_normal((n, 11))
    b = a @ x0
    lu = F.PartialPivLu(to_dev(a))
    x = to_dev(b)
    lu.solve_in_place(x)
    assert sc.norm_backward_error(a, to_host(x), b) <= C_NORM * n * EPS
    xt = to_dev(a.T @ x0)
    lu.solve_transpose_in_place(xt)
    assert sc.norm_backward_error(a.T, to_host(xt), a.T @ x0) <= C_NORM * n * EPS


@pytest.mark.parametrize("n", [100, 129, 700, 2000])
@pytest.mark.parametrize("cond", [1e4, 1e9])
def test_qr_solve_ill_conditioned(n, cond):
    """qr/no_pivoting/solve.rs: x = R^-1 Q^T b (cond stays below the reference's rank threshold 16 eps m, factor.rs:52-58)"""
    F = init_gpu()
    rng = np.random.default_rng(n + 1)
    a = _ill(n, cond, False)
    x0 = rng.standard_normal((n, 11))
    b = a @ x0
    qr = F.Qr(to_dev(a))
    x = to_dev(b)
    F.qr_solve_in_place(qr.qr, qr.q_coeff, x)
    assert sc.norm_backward_error(a, to_host(x), b) <= C_NORM * n * EPS


@pytest.mark.parametrize("n", [100, 129, 700, 2000, 2600])
@pytest.mark.parametrize("cond", [1e6, 1e12])
def test_llt_ill_conditioned(n, cond, monkeypatch):
    """cholesky/llt: the factorization's panel solves (A10 L00^-T, cholesky/ldlt/factor.rs:422-426) and llt::solve go
    through the same TRSM; an SPD matrix with cond up to 1e12 must factor with ||L L^T - A|| and the solve's
    residual at the n eps level.  n = 2600 runs the look-ahead driver (thresholds lowered)."""
    F = init_gpu()
    if n > 2048:
        monkeypatch.setenv("FAER_HIP_LLT_LA_MIN", "2048")
        monkeypatch.setenv("FAER_HIP_LLT_TAIL", "1024")
    rng = np.random.default_rng(n + 2)
    a = _ill(n, cond, True)
    d = to_dev(a)
    assert F.llt_factor_in_place(d) == 0
    L = np.tril(to_host(d))
    ll = L.astype(np.longdouble) @ L.T.astype(np.longdouble)
    assert float(np.linalg.norm(ll - a) / np.linalg.norm(a)) <= C_NORM * n * EPS
    x0 = rng.standard_normal((n, 7))
    b = a @ x0
    x = to_dev(b)
    F.llt_solve_in_place(d, x)
    assert sc.norm_backward_error(a, to_host(x), b) <= C_NORM * n * EPS


def test_singular_diagonal_entry_stays_local():
    """an exactly zero diagonal entry makes the rows that depend on it Inf / NaN and NOTHING else (substitution);
    an inverted diagonal block would spread it over the whole 128-block"""
    F = init_gpu()
    n, z = 300, 200
    rng = np.random.default_rng(5)
    t = np.tril(rng.standard_normal((n, n))) / n + np.eye(n)
    t[z, z] = 0.0
    b = rng.standard_normal((n, 4))
    x = to_dev(b)
    F.solve_lower_triangular_in_place(to_dev(t), x)
    got = to_host(x)
    assert np.isfinite(got[:z]).all()
    assert not np.isfinite(got[z]).any()
