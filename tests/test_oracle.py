"""Pins the CPU oracle (oracle/faer_oracle.c) before it is trusted as the
checker: (1) the reference's own known-answer vectors (tests/golden), (2) the
reference's property tests restated with its sizes and tolerances
(SURVEY.md section 4), (3) LAPACK (scipy) cross-checks.  CPU only."""
import json
import os

import numpy as np
import pytest
import scipy.linalg as sla

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPS = np.finfo(np.float64).eps


def rng(seed=0):
    return np.random.default_rng(seed)


def fmat(r, m, n, dtype=np.float64, order="F"):
    return np.asarray(r.standard_normal((m, n)), dtype=dtype, order=order)


# ------------------------------------------------------------------ golden
def test_golden_matmul_2x2(oracle):
    g = json.load(open(os.path.join(GOLD, "matmul_2x2.json")))
    lhs = np.array(g["lhs"], order="F")
    rhs = np.array(g["rhs"], order="F")
    acc = np.full((2, 2), np.nan, order="F")  # Replace must not read dst
    oracle.matmul(acc, lhs, rhs, alpha=g["alpha"])
    assert np.abs(acc - np.array(g["target"])).max() < g["tol"]


def test_golden_qr_lstsq(oracle):
    """faer/src/linalg/qr/mod.rs:116-191 restated against the oracle."""
    g = json.load(open(os.path.join(GOLD, "qr_lstsq_10x2.json")))
    a = np.array(g["a"], order="F")
    b = np.array(g["b"], order="F")
    x = np.array(g["expected_solution"])
    m, n = a.shape
    bs = oracle.qr_recommended_block_size(m, n)
    assert bs == 1  # 10*2 <= 16*16 (qr/no_pivoting/factor.rs:91-116)
    h = np.zeros((bs, min(m, n)), order="F")
    qr = a.copy(order="F")
    rank = oracle.qr_in_place(qr, h)
    assert rank == 2
    sol = b.copy(order="F")
    oracle.apply_householder_sequence_left(qr, h, sol, transpose=True)
    sol = np.asfortranarray(sol[:rank])
    oracle.trsm(np.asfortranarray(qr[:rank, :rank]), sol, upper=True)
    assert np.abs(sol - x).max() <= g["tol"]


# ------------------------------------------------------------------ matmul
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (15, 16, 17), (127, 129, 128), (64, 1, 33), (1, 70, 9), (5, 7, 0)])
@pytest.mark.parametrize("layout", ["FF", "CF", "FC", "CC"])
def test_matmul_vs_numpy(oracle, m, n, k, layout):
    r = rng(1)
    a = fmat(r, m, k, order=layout[0])
    b = fmat(r, k, n, order=layout[1])
    c0 = fmat(r, m, n)
    c = c0.copy(order="F")
    oracle.matmul(c, a, b, alpha=-0.5, accum_add=True)
    ref = c0 - 0.5 * (a @ b)
    tol = 4 * max(k, 1) * EPS * (np.abs(a) @ np.abs(b) + np.abs(c0) + 1e-300)
    assert (np.abs(c - ref) <= tol).all()
    c = np.full((m, n), np.nan, order="F")
    oracle.matmul(c, a, b, alpha=2.0)
    assert (np.abs(c - 2.0 * (a @ b)) <= tol).all()


def test_matmul_negative_strides(oracle):
    r = rng(2)
    a = fmat(r, 20, 30)[::-1, ::-1]
    b = fmat(r, 30, 10)[::-1, :]
    c = np.zeros((20, 10), order="F")[:, ::-1]
    oracle.matmul(c, a, b)
    assert np.abs(c - a @ b).max() < 1e-12


STRUCTS = ["rect", "lower", "upper", "strict_lower", "strict_upper", "unit_lower", "unit_upper"]


def dense_of(a, s):
    n = a.shape[0]
    if s == "rect":
        return a.copy()
    if s == "lower":
        return np.tril(a)
    if s == "upper":
        return np.triu(a)
    if s == "strict_lower":
        return np.tril(a, -1)
    if s == "strict_upper":
        return np.triu(a, 1)
    if s == "unit_lower":
        return np.tril(a, -1) + np.eye(n)
    if s == "unit_upper":
        return np.triu(a, 1) + np.eye(n)


def mask_of(n, s):
    i, j = np.indices((n, n))
    return {"rect": i >= -1, "lower": i >= j, "upper": i <= j, "strict_lower": i > j, "strict_upper": i < j,
            "unit_lower": i > j, "unit_upper": i < j}[s]


@pytest.mark.parametrize("cs", STRUCTS)
@pytest.mark.parametrize("as_", STRUCTS)
def test_triangular_matmul(oracle, cs, as_):
    """mirror of matmul/mod.rs:2216-2266 `test_triangular` (1e-10)."""
    r = rng(3)
    for bs_ in STRUCTS:
        n = int(r.integers(1, 40))
        a, b, c0 = fmat(r, n, n), fmat(r, n, n), fmat(r, n, n)
        c = c0.copy(order="F")
        oracle.matmul_triangular(c, cs, a, as_, b, bs_, alpha=2.5, accum_add=True)
        full = c0 + 2.5 * dense_of(a, as_) @ dense_of(b, bs_)
        mk = mask_of(n, cs)
        assert np.abs(c - full)[mk].max(initial=0) < 1e-10
        assert (c[~mk] == c0[~mk]).all()  # untouched part unchanged


# -------------------------------------------------------------------- trsm
@pytest.mark.parametrize("n,k", [(1, 3), (4, 5), (7, 2), (33, 70), (128, 65), (200, 9)])
@pytest.mark.parametrize("upper", [False, True])
@pytest.mark.parametrize("unit", [False, True])
def test_trsm(oracle, n, k, upper, unit):
    r = rng(4)
    t = fmat(r, n, n) / (n if unit else 1.0) + n * np.eye(n)  # unit: keep the triangle well conditioned
    b = fmat(r, n, k)
    x = b.copy(order="F")
    oracle.trsm(t, x, upper=upper, unit=unit)
    tt = np.triu(t) if upper else np.tril(t)
    if unit:
        np.fill_diagonal(tt, 1.0)
    assert np.abs(tt @ x - b).max() < 1e-10 * max(1, np.abs(b).max())


# ---------------------------------------------------------------- cholesky
def spd(r, n, dtype=np.float64):
    a = r.standard_normal((n, n))
    return np.asarray(a @ a.T + n * np.eye(n), dtype=dtype, order="F")


@pytest.mark.parametrize("n", [0, 1, 2, 4, 8, 31, 64, 65, 127, 240, 300])
def test_llt(oracle, n):
    """mirror of cholesky/ldlt/factor.rs:776-868 (tol 1e-12 relative)."""
    r = rng(5)
    a = spd(r, n)
    l = a.copy(order="F")
    upper_before = np.triu(l, 1).copy()
    st, cnt = oracle.llt_in_place(l)
    assert (st, cnt) == ("ok", 0)
    assert (np.triu(l, 1) == upper_before).all()  # strict upper untouched
    L = np.tril(l)
    if n:
        assert np.abs(L @ L.T - a).max() < 1e-12 * np.abs(a).max() * max(n, 1)
        assert np.abs(L - sla.cholesky(a, lower=True)).max() < 1e-11 * np.abs(L).max()


@pytest.mark.parametrize("params", [(32, 32), (2, 4), (64, 128)])
def test_llt_recursion_params(oracle, params):
    r = rng(6)
    a = spd(r, 200)
    l = a.copy(order="F")
    assert oracle.llt_in_place(l, recursion_threshold=params[0], block_size=params[1])[0] == "ok"
    L = np.tril(l)
    assert np.abs(L @ L.T - a).max() < 1e-12 * np.abs(a).max() * 200


@pytest.mark.parametrize("n,bad", [(10, 3), (100, 64), (300, 299), (200, 0)])
def test_llt_non_positive_pivot(oracle, n, bad):
    r = rng(7)
    a = spd(r, n)
    # make the leading (bad+1)x(bad+1) minor singular/indefinite at column `bad`
    a[bad, bad] = -1.0 if bad == 0 else (a[bad, :bad] @ np.linalg.solve(a[:bad, :bad], a[:bad, bad])) - 1.0
    l = a.copy(order="F")
    assert oracle.llt_in_place(l) == ("non_positive_pivot", bad)


def test_llt_regularization(oracle):
    a = np.diag([4.0, 1e-20, 9.0]).copy(order="F")
    st, cnt = oracle.llt_in_place(a, reg_delta=1.0, reg_eps=1e-10)
    assert (st, cnt) == ("ok", 1)
    # reference behaviour (cholesky/ldlt/factor.rs:122-174): the regularised value only enters D = sqrt(delta);
    # the stored column, diagonal included, is the ORIGINAL a_jj scaled by 1/sqrt(delta)
    assert np.allclose(np.diag(a), [2.0, 1e-20, 3.0])


# ---------------------------------------------------------------------- lu
@pytest.mark.parametrize("m,n", [(1, 1), (2, 2), (3, 3), (128, 128), (255, 255), (256, 256), (257, 257), (300, 8),
                                 (8, 300), (40, 17), (17, 40)])
@pytest.mark.parametrize("thr", [2, 16])
def test_plu(oracle, m, n, thr):
    """mirror of lu/partial_pivoting/factor.rs:304-404 `test_plu` (1e-13)."""
    r = rng(8)
    a = fmat(r, m, n)
    lu = a.copy(order="F")
    perm, perm_inv, nt = oracle.lu_in_place(lu, recursion_threshold=thr)
    size = min(m, n)
    L = np.tril(lu[:, :size], -1) + np.eye(m, size)
    U = np.triu(lu[:size, :])
    assert (perm_inv[perm] == np.arange(m)).all()
    assert np.abs(L @ U - a[perm]).max() < 1e-13 * max(m, n) * np.abs(a).max()
    assert np.abs(np.tril(lu, -1)).max(initial=0) <= 1.0 + 1e-15  # partial pivoting
    if m == n:
        # LAPACK picks the same pivots (first max) => same permutation & factors
        plu, piv = sla.lu_factor(a)
        p2 = np.arange(m)
        for i, p in enumerate(piv):
            p2[[i, p]] = p2[[p, i]]
        assert (p2 == perm).all()
        assert np.abs(plu - lu).max() < 1e-10 * np.abs(lu).max()
        sign = (-1) ** nt
        assert np.isclose(sign * np.prod(np.diag(lu)), np.linalg.det(a), rtol=1e-8)


def test_plu_strided(oracle):
    r = rng(9)
    a = fmat(r, 50, 50, order="C")
    lu = a.copy(order="C")
    perm, _, _ = oracle.lu_in_place(lu)
    L = np.tril(lu, -1) + np.eye(50)
    assert np.abs(L @ np.triu(lu) - a[perm]).max() < 1e-12


# ---------------------------------------------------------------------- qr
def q_from(oracle, qr, h):
    m = qr.shape[0]
    q = np.eye(m, order="F")
    oracle.apply_householder_sequence_left(qr, h, q, transpose=False)
    return q


@pytest.mark.parametrize("m,n", [(1, 1), (2, 1), (10, 2), (33, 33), (100, 40), (40, 100), (257, 64), (512, 200)])
@pytest.mark.parametrize("bs", [1, 4, 15, 32, None])
def test_qr_full_rank(oracle, m, n, bs):
    """mirror of qr/no_pivoting/factor.rs:327-538 `test_qr` (Q R ~ A, 1e-10)."""
    r = rng(10)
    a = fmat(r, m, n)
    size = min(m, n)
    if bs is None:
        bs = oracle.qr_recommended_block_size(m, n)
    bs = max(1, min(bs, size))
    h = np.zeros((bs, size), order="F")
    qr = a.copy(order="F")
    rank = oracle.qr_in_place(qr, h)
    assert rank == size
    R = np.triu(qr)
    q = q_from(oracle, qr, h)
    assert np.abs(q @ R - a).max() < 1e-10
    assert np.abs(q.T @ q - np.eye(m)).max() < 1e-10
    # householder factor convention (householder.rs:21-23): T upper, T_ii = tau_i = |v_i|^2/2,
    # T_ij = v_i^H v_j
    V = np.tril(qr[:, :size], -1) + np.eye(m, size)
    G = V.T @ V
    for j0 in range(0, size, bs):
        w = min(bs, size - j0)
        Tb = h[:w, j0:j0 + w]
        Gb = G[j0:j0 + w, j0:j0 + w]
        assert np.abs(np.triu(Tb, 1) - np.triu(Gb, 1)).max(initial=0) < 1e-10
        tails = np.array([m - 1 - j for j in range(j0, j0 + w)])
        d = np.diag(Tb)
        assert np.abs(d - 0.5 * np.diag(Gb))[tails > 0].max(initial=0) < 1e-10
        assert np.isinf(d[tails == 0]).all()  # empty tail => tau = inf (householder.rs:66-77)
    # same R as LAPACK up to row signs
    R2 = np.linalg.qr(a, mode="r")
    assert np.abs(np.abs(R[:size]) - np.abs(R2)).max() < 1e-9 * np.abs(R2).max()


@pytest.mark.parametrize("true_rank", [1, 2, 3, 5])
@pytest.mark.parametrize("bs", [1, 15])
def test_qr_rank_deficient(oracle, true_rank, bs):
    r = rng(11)
    m, n = 120, 60
    a = np.asfortranarray(fmat(r, m, true_rank) @ fmat(r, true_rank, n))
    size = min(m, n)
    h = np.zeros((bs, size), order="F")
    qr = a.copy(order="F")
    rank = oracle.qr_in_place(qr, h)
    assert true_rank <= rank < size
    R = np.triu(qr)
    q = q_from(oracle, qr, h)
    assert np.abs(q @ R - a).max() < 1e-10 * max(1.0, np.abs(a).max())


def test_norm_l2(oracle):
    """mirror of reductions/norm_l2.rs:198-219."""
    for f in [0.0, 1.0, 1e30, 1e250, 1e-30, 1e-250]:
        x = f * np.arange(1, 1024, dtype=np.float64)
        target = 0.0
        for v in x:
            target = np.hypot(v, target)
        got = oracle.norm_l2(x)
        if f == 0:
            assert got == 0
        else:
            assert abs(got - target) / target < 1e-13
    x = np.full(10_000_000, 0.3)
    assert abs(oracle.norm_l2(x) - np.sqrt(0.09 * 1e7)) / np.sqrt(0.09 * 1e7) < 1e-9


def test_f32_paths(oracle):
    r = rng(12)
    a = fmat(r, 300, 40, dtype=np.float32)
    h = np.zeros((8, 40), dtype=np.float32, order="F")
    qr = a.copy(order="F")
    assert oracle.qr_in_place(qr, h) == 40
    q = np.eye(300, dtype=np.float32, order="F")
    oracle.apply_householder_sequence_left(qr, h, q, transpose=False)
    assert np.abs(q @ np.triu(qr) - a).max() < 2e-4
    s = spd(r, 100, dtype=np.float32)
    l = s.copy(order="F")
    assert oracle.llt_in_place(l)[0] == "ok"
    L = np.tril(l)
    assert np.abs(L @ L.T - s).max() < 1e-3


def test_norm_l2_reference_accuracy_kat(oracle):
    """reductions/norm_l2.rs:216-218 (test_norm_l2): a 1e7-long column of 0.3 within 1e-14 of sqrt(0.09e7); this
    is what pins the pairwise summation tree (a sequential fp sum fails it)."""
    x = np.full(10_000_000, 0.3)
    target = np.sqrt(0.3 * 0.3 * 10_000_000.0)
    got = oracle.norm_l2(x)
    assert abs(got - target) / max(abs(got), target) < 1e-14
    for (m, n) in [(9, 10), (1023, 5), (42, 1)]:
        for factor in [0.0, 1.0, 1e30, 1e250, 1e-30, 1e-250]:
            mat = np.array([[factor * (i + j) for j in range(n)] for i in range(m)], dtype=np.float64)
            for j in range(n):
                col = np.ascontiguousarray(mat[:, j])
                tgt = 0.0
                for v in col:
                    tgt = np.hypot(v, tgt)
                r = oracle.norm_l2(col)
                if factor == 0.0:
                    assert r == tgt
                else:
                    assert abs(r - tgt) / max(abs(r), abs(tgt)) < 1e-14


# ------------------------------------------------------------------------------------------------ LDLT (section 8f, item 1)
def _quasi_definite(rng, n, dtype=np.float64):
    """[[H, B^T], [B, -G]] with H, G SPD: strongly factorizable without pivoting, D has n1 positive then negative entries"""
    n1 = n // 2
    h = rng.standard_normal((n, n))
    H = h[:n1, :n1] @ h[:n1, :n1].T + n * np.eye(n1)
    G = h[n1:, n1:] @ h[n1:, n1:].T + n * np.eye(n - n1)
    B = h[n1:, :n1]
    return np.asarray(np.block([[H, B.T], [B, -G]]), dtype=dtype, order="F"), n1


@pytest.mark.parametrize("n", [1, 2, 7, 64, 65, 129, 300])
def test_ldlt_reconstructs_and_keeps_the_upper_triangle(oracle, n):
    """cholesky/ldlt/factor.rs tests (:803-..., tolerance 1e-12 at n <= 64): L D L^H == A"""
    rng = np.random.default_rng(n)
    a, n1 = _quasi_definite(rng, n)
    w = a.copy(order="F")
    iu = np.triu_indices(n, 1)
    w[iu] = -3.25
    assert oracle.ldlt_in_place(w) == ("ok", 0)
    assert (w[iu] == -3.25).all()
    L, D = np.tril(w, -1) + np.eye(n), np.diag(w).copy()
    assert np.abs(L @ np.diag(D) @ L.T - a).max() <= 1e-12 * max(1.0, np.abs(a).max())
    assert (D[:n1] > 0).all() and (D[n1:] < 0).all()


def test_ldlt_zero_pivot_and_regularization(oracle):
    a = np.asfortranarray(np.array([[1.0, 2.0, 3.0], [2.0, 4.0, 1.0], [3.0, 1.0, 1.0]]))
    w = a.copy(order="F")
    assert oracle.ldlt_in_place(w) == ("zero_pivot", 1)
    assert w[0, 0] == 1.0 and w[1, 1] == 0.0  # D is written up to and including the failing index (:791-798)
    # the same matrix with dynamic regularization and expected signs (+, -, +): the zero pivot becomes -delta, not counted
    w = a.copy(order="F")
    assert oracle.ldlt_in_place(w, 1e-3, 1e-8, signs=[1, -1, 1]) == ("ok", 0)
    assert w[1, 1] == -1e-3
    # expected sign +1 on non positive pivots: corrected to +delta and counted (the last pivot turns negative once
    # the second one has been lifted to +delta, so two corrections)
    w = a.copy(order="F")
    assert oracle.ldlt_in_place(w, 1e-3, 1e-8, signs=[1, 1, 1]) == ("ok", 2)
    assert w[1, 1] == 1e-3 and w[2, 2] == 1e-3


# ------------------------------------------------------------------------------------ LU with full pivoting
@pytest.mark.parametrize("m,n,order", [(1, 1, "F"), (2, 3, "F"), (5, 5, "F"), (40, 30, "F"), (30, 40, "F"), (64, 64, "C"), (33, 50, "C"),
                                       (100, 100, "F")])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_full_piv_lu_oracle_properties(oracle, m, n, order, dtype):
    """lu/full_pivoting/factor.rs tests (:527-640): P A Q == L U within eps * n; plus what complete pivoting
    guarantees by construction: |l_ij| <= 1 and |u_kk| >= every entry of the trailing block at step k"""
    rng = np.random.default_rng(m * 31 + n)
    a = np.array(rng.standard_normal((m, n)), dtype=dtype, order=order)
    lu = a.copy(order=order)
    rp, rpi, cp, cpi, nt = oracle.full_piv_lu_in_place(lu)
    size = min(m, n)
    assert (rpi[rp] == np.arange(m)).all() and (cpi[cp] == np.arange(n)).all()
    L = (np.tril(lu[:, :size], -1) + np.eye(m, size)).astype(np.float64)
    U = np.triu(lu[:size, :]).astype(np.float64)
    e = np.finfo(dtype).eps
    assert np.abs(L @ U - a[rp][:, cp]).max() <= 16 * max(m, n) * e * np.abs(a).max()
    assert np.abs(np.tril(lu[:, :size], -1)).max(initial=0) <= 1.0 + 4 * e
    d = np.abs(np.diag(U))
    for k in range(size):
        assert d[k] + 1e-300 >= np.abs(U[k, k:]).max() * (1 - 4 * e)
    # transposition count == number of non-trivial row / column swaps that rebuild the permutations
    assert 0 <= nt <= 2 * size


def test_full_piv_lu_oracle_stops_on_an_exactly_singular_trailing_block(oracle):
    """factor.rs:324-332: a best score below the smallest positive normal ends the elimination with identity
    transpositions; the factors of the leading part are still exact"""
    a = np.zeros((6, 5), order="F")
    a[:2, :2] = [[4.0, 1.0], [2.0, 3.0]]
    lu = a.copy(order="F")
    rp, _, cp, _, nt = oracle.full_piv_lu_in_place(lu)
    L = np.tril(lu[:, :5], -1) + np.eye(6, 5)
    U = np.triu(lu[:5, :])
    assert np.abs(L @ U - a[rp][:, cp]).max() == 0.0
    assert (np.diag(U)[2:] == 0).all()


# ------------------------------------------------------------------------------------ QR with column pivoting
@pytest.mark.parametrize("m,n,order", [(1, 1, "F"), (5, 5, "F"), (40, 30, "F"), (30, 40, "F"), (64, 64, "C"), (200, 50, "F"), (1, 7, "F"), (7, 1, "F"),
                                       (300, 300, "F")])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_colpiv_qr_oracle_vs_lapack(oracle, m, n, order, dtype):
    """qr/col_pivoting/factor.rs tests (:397-526): A P == Q R within eps * n.  Pinned further against LAPACK's
    geqp3 (scipy.linalg.qr(pivoting=True)): same pivoting rule (largest down-dated column norm), so the same
    permutation and the same |diag R| on matrices whose column norms are well separated"""
    import scipy.linalg as sl

    rng = np.random.default_rng(m * n)
    a = np.array(rng.standard_normal((m, n)) * np.logspace(0, -3, n)[None, :], dtype=dtype, order=order)
    qr = a.copy(order=order)
    size = min(m, n)
    bs = oracle.qr_recommended_block_size(m, n, dtype)
    h = np.zeros((bs, size), dtype=dtype, order="F")
    cp, cpi, nt = oracle.colpiv_qr_in_place(qr, h)
    assert (cpi[cp] == np.arange(n)).all()
    out = np.zeros((m, n), dtype=dtype, order="F")
    out[:size, :] = np.triu(qr[:size, :])
    oracle.apply_householder_sequence_left(np.asfortranarray(qr[:, :size]), h, out, False)
    e = np.finfo(dtype).eps
    assert np.abs(out.astype(np.float64) - a[:, cp].astype(np.float64)).max() <= 32 * max(m, n) * e * np.abs(a).max()
    if dtype == np.float64:
        _, r_ref, p_ref = sl.qr(a, mode="economic", pivoting=True)
        assert np.array_equal(cp, p_ref)
        d, dr = np.abs(np.diag(qr[:size, :size])), np.abs(np.diag(r_ref))
        assert np.abs(d - dr).max() <= 64 * max(m, n) * e * dr.max()
