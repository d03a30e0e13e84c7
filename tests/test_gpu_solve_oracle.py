"""GPU parity of the solve side against the ORACLE: the reference's solves are compositions of
routines the oracle restates -- triangular solves (triangular_solve.rs:420-604), the block Householder application
(householder.rs:438-604) and the permutations -- applied to the oracle's own factors:
  llt::solve (cholesky/llt/solve.rs:56)        = L^-1, then L^-T
  lu::solve (lu/partial_pivoting/solve.rs:20)   = P, unit-lower L^-1, U^-1;  transpose: U^-T, L^-T, P^T
  qr::solve_lstsq (qr/no_pivoting/solve.rs:38)  = Q^T through the block reflectors, R^-1 on the leading rows
Both sides run the same algorithm on factors that agree to c n eps kappa, so the solutions agree to c n eps kappa |x|."""
import numpy as np
import pytest

from gpu_util import EPS, init_gpu, rnd, spd, to_dev, to_host

pytestmark = pytest.mark.gpu


def well_conditioned(rng, n, dtype):
    return np.asarray(rng.standard_normal((n, n)) + 2 * np.sqrt(n) * np.eye(n), dtype=dtype, order="F")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,k", [(1, 1), (50, 7), (129, 130), (400, 33), (1000, 3)])
def test_llt_solve_vs_oracle(oracle, n, k, dtype):
    F = init_gpu()
    rng = np.random.default_rng(n + k)
    a, b = spd(rng, n, dtype), rnd(rng, n, k, dtype)
    l = a.copy(order="F")
    assert oracle.llt_in_place(l) == ("ok", 0)
    ref = b.copy(order="F")
    oracle.trsm(l, ref)
    oracle.trsm(l.T, ref, upper=True)
    dl = to_dev(a)
    F.llt_factor_in_place(dl)
    x = to_dev(b)
    F.llt_solve_in_place(dl, x)
    kappa = np.linalg.cond(a.astype(np.float64))
    assert np.abs(to_host(x).astype(np.float64) - ref).max() <= 64 * n * EPS[np.dtype(dtype)] * kappa * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,k", [(1, 1), (40, 5), (257, 64), (600, 9)])
def test_partial_piv_lu_solve_vs_oracle(oracle, n, k, dtype, transpose):
    F = init_gpu()
    rng = np.random.default_rng(3 * n + k)
    a, b = well_conditioned(rng, n, dtype), rnd(rng, n, k, dtype)
    lu = a.copy(order="F")
    perm, perm_inv, _ = oracle.lu_in_place(lu)
    if not transpose:  # A = P^T L U: x = U^-1 L^-1 (P b)
        ref = np.asfortranarray(b[perm])
        oracle.trsm(lu, ref, unit=True)
        oracle.trsm(lu, ref, upper=True)
    else:  # A^T = U^T L^T P: x = P^T L^-T U^-T b
        ref = b.copy(order="F")
        oracle.trsm(lu.T, ref)  # U^T is lower triangular
        oracle.trsm(lu.T, ref, upper=True, unit=True)
        ref = np.asfortranarray(ref[perm_inv])
    dlu = to_dev(a)
    pf, pb, _ = F.partial_piv_lu_factor_in_place(dlu)
    assert (pf.astype(np.int64) == perm).all()
    x = to_dev(b)
    F.partial_piv_lu_solve_in_place(dlu, pf, pb, x, transpose=transpose)
    kappa = np.linalg.cond(a.astype(np.float64))
    assert np.abs(to_host(x).astype(np.float64) - ref).max() <= 64 * n * EPS[np.dtype(dtype)] * kappa * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m,n,k", [(10, 2, 1), (100, 50, 3), (300, 300, 7), (2000, 130, 40)])
def test_qr_solve_lstsq_vs_oracle(oracle, m, n, k, dtype):
    F = init_gpu()
    rng = np.random.default_rng(m + n + k)
    a, b = rnd(rng, m, n, dtype), rnd(rng, m, k, dtype)
    bs = oracle.qr_recommended_block_size(m, n, dtype)
    qr, h = a.copy(order="F"), np.zeros((bs, n), dtype=dtype, order="F")
    assert oracle.qr_in_place(qr, h) == n
    ref = b.copy(order="F")
    oracle.apply_householder_sequence_left(qr, h, ref, True)
    top = np.asfortranarray(ref[:n])
    oracle.trsm(qr[:n, :n], top, upper=True)
    dqr, dh = to_dev(a), to_dev(np.zeros((bs, n), dtype=dtype))
    assert F.qr_factor_in_place(dqr, dh) == n
    x = to_dev(b)
    F.qr_solve_lstsq_in_place(dqr, dh, x)
    kappa = np.linalg.cond(a.astype(np.float64))
    assert np.abs(to_host(x)[:n].astype(np.float64) - top).max() <= 64 * max(m, n) * EPS[np.dtype(dtype)] * kappa * max(1.0, np.abs(top).max())
