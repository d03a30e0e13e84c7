"""Ill-conditioned triangular / square systems shared by the CPU (oracle) and GPU (C-ABI) stability tests.

The reference solves triangular systems by substitution down to closed forms for n <= 4
(faer/src/linalg/triangular_solve.rs:98-198, recursion :452-484), which is backward stable for ANY
triangle: the computed X satisfies |T X - B| <= c n eps |T| |X| whatever cond(T).  A solver that multiplies
by explicitly inverted diagonal blocks only achieves that for well-conditioned blocks, and the difference
only shows when B = T X0 has a SMALL solution (|X| << |T^-1| |B|).  Every case here is therefore run with
both a random right-hand side and B = T X0.
"""
import numpy as np

TRI_KINDS = ["graded", "mixed", "kahan", "growth", "random_lu", "lu_u"]


def _coupled_blocks(make, n, rng, blk=256):
    """block lower triangular: `make(m)` triangles of at most `blk` rows on the diagonal (keeps the growth of the
    inverse inside one block), weak random coupling below them"""
    t = np.tril(rng.standard_normal((n, n)), -1) * 1e-3
    for j0 in range(0, n, blk):
        m = min(blk, n - j0)
        t[j0:j0 + m, j0:j0 + m] = make(m)
    return t


def triangle(kind, n, rng):
    """lower triangular n x n test matrix (float64); `unit`-style kinds have a unit diagonal."""
    if kind == "graded":  # diagonal 1 ... 1e-12, rows scaled with it (benign for every algorithm: a scaling)
        g = np.logspace(0, -12, n)
        t = np.tril(rng.standard_normal((n, n)), -1) / np.sqrt(n) + np.eye(n)
        return g[:, None] * t
    if kind == "mixed":  # graded diagonal, off-diagonal entries NOT scaled with it: cond(T_kk) up to 1e25
        g = np.logspace(0, -12, n)
        return np.diag(g) + 1e-3 * np.tril(rng.standard_normal((n, n)), -1)
    if kind == "kahan":  # transpose of Kahan's upper triangular matrix, theta = 1.2
        def make(m):
            c, s = np.cos(1.2), np.sin(1.2)
            r = np.eye(m) - c * np.triu(np.ones((m, m)), 1)
            return np.ascontiguousarray(((s ** np.arange(m))[:, None] * r).T)
        return _coupled_blocks(make, n, rng)
    if kind == "growth":  # unit lower, -1 below the diagonal on a band: inverse entries grow like 2^k
        def make(m):
            t = np.eye(m)
            for d in range(1, min(m, 40)):
                t -= np.diag(np.ones(m - d), -d)
            return t
        return _coupled_blocks(make, n, rng)
    if kind == "random_lu":  # the unit lower factor of a partially pivoted LU: |l_ij| <= 1
        import scipy.linalg as sla

        _, l, _ = sla.lu(rng.standard_normal((n, n)))
        return np.tril(l)
    if kind == "lu_u":  # transpose of the U factor of a matrix with cond 1e12
        import scipy.linalg as sla

        _, _, u = sla.lu(ill_conditioned(n, 1e12, rng))
        return np.ascontiguousarray(np.triu(u).T)
    raise ValueError(kind)


def is_unit(kind):
    return kind in ("growth", "random_lu")


def tri_rhs(t, k, rng, small_solution):
    n = t.shape[0]
    if small_solution:
        x0 = rng.standard_normal((n, k))
        return np.asarray(t.astype(np.longdouble) @ x0.astype(np.longdouble), dtype=np.float64)
    return rng.standard_normal((n, k))


def tri_backward_error(t, x, b):
    """max_ij |T X - B|_ij / (|T| |X| + |B|)_ij in extended precision (0/0 counts as 0)"""
    tl, xl, bl = t.astype(np.longdouble), x.astype(np.longdouble), b.astype(np.longdouble)
    r = np.abs(tl @ xl - bl)
    s = np.abs(tl) @ np.abs(xl) + np.abs(bl)
    with np.errstate(invalid="ignore", divide="ignore"):
        q = np.where(s > 0, r / s, 0.0)
    return float(q.max())


def ill_conditioned(n, cond, rng, spd=False):
    """dense n x n matrix with singular values logspace(0, -log10(cond))"""
    q1, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.logspace(0, -np.log10(cond), n)
    if spd:
        a = (q1 * s) @ q1.T
        return (a + a.T) / 2
    q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (q1 * s) @ q2.T


def norm_backward_error(a, x, b):
    """||A X - B||_F / (||A||_F ||X||_F) in extended precision"""
    al, xl, bl = a.astype(np.longdouble), x.astype(np.longdouble), b.astype(np.longdouble)
    return float(np.linalg.norm(al @ xl - bl) / (np.linalg.norm(al) * np.linalg.norm(xl)))
