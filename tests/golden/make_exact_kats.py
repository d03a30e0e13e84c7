"""Hand-derivable exact known-answer tests for LLT / partial-pivot LU / TRSM  ->  exact_kats.json

The reference holds no vector for these three (its only KATs are the 10 x 2 least-squares example and
the 2 x 2 matmul doctest), so the oracle is pinned for them by construction instead: every matrix
below has small dyadic-rational entries chosen so that EVERY intermediate quantity of the factorization -- whatever the
blocking, summation order or use of fused multiply-adds -- is exactly representable in binary floating point.  The expected
factors are computed here in exact rational arithmetic (fractions.Fraction), following nothing but the reference's rules:

  * LU: faer/src/linalg/lu/partial_pivoting/factor.rs:35-64 -- pivot = FIRST row of strictly largest |a_ij| (start from
    max = 0, so an all-zero column keeps the diagonal row), row interchange, multiply by the reciprocal pivot, rank-1 update;
    :274-277 -- perm = identity with the transpositions applied in order;
  * LLT: faer/src/linalg/cholesky/ldlt/factor.rs:147-174 -- l_jj = sqrt(d_j), column scaled by 1 / l_jj, `!(d > 0)` =>
    Err(j) with j the global column;
  * TRSM: faer/src/linalg/triangular_solve.rs (forward substitution; unit / generic diagonal).

Pivots are +-2^k and Cholesky diagonals 2^k, so reciprocals and square roots are exact; an implementation that agrees with
the reference must reproduce these factors BIT FOR BIT (tests/test_exact_kats.py: oracle on the CPU, library on the GPU).
The one case with an exactly singular middle column follows the reference into its NaNs (0 * (1/0)).

Run: python tests/golden/make_exact_kats.py   (pure Python, no reference needed; deterministic)."""
import json
import os
import random
from fractions import Fraction as Fr

HERE = os.path.dirname(os.path.abspath(__file__))


def exact_float(x):
    f = float(x)
    assert Fr(f) == x, f"{x} is not a binary floating-point number"
    return f


def matmul(a, b):
    n, k, m = len(a), len(b), len(b[0])
    return [[sum((a[i][t] * b[t][j] for t in range(k)), Fr(0)) for j in range(m)] for i in range(n)]


def transpose(a):
    return [list(r) for r in zip(*a)]


def budget_ok(mats, bits=44):
    """all entries are multiples of 2^-12 below 2^(bits-12): sums of a few hundred products of such numbers stay exact"""
    for mt in mats:
        for r in mt:
            for v in r:
                if v.denominator & (v.denominator - 1) or v.denominator > 4096 or abs(v) >= 2 ** (bits - 12):
                    return False
    return True


def lu_exact(a):
    """factor.rs:19-67 in exact arithmetic -> packed LU, pivot rows (absolute), perm (forward), transposition count"""
    a = [list(r) for r in a]
    m, n = len(a), len(a[0])
    size = min(m, n)
    piv, nt = [], 0
    perm = list(range(m))
    for j in range(size):
        imax, mx = j, Fr(0)
        for i in range(j, m):
            if abs(a[i][j]) > mx:
                mx, imax = abs(a[i][j]), i
        piv.append(imax)
        if imax != j:
            a[j], a[imax] = a[imax], a[j]
            perm[j], perm[imax] = perm[imax], perm[j]
            nt += 1
        assert a[j][j] != 0, "exactly singular column: use the float simulation"
        inv = 1 / a[j][j]
        for i in range(j + 1, m):
            a[i][j] *= inv
        for i in range(j + 1, m):
            l = a[i][j]
            if l:
                for c in range(j + 1, n):
                    a[i][c] -= l * a[j][c]
    return a, piv, perm, nt


def lu_float_sim(a):
    """the same rule literally in floating point (for the singular column: 1 / 0 = inf, 0 * inf = NaN)"""
    import numpy as np

    a = np.array([[float(v) for v in r] for r in a])
    m, n = a.shape
    perm = list(range(m))
    nt = 0
    with np.errstate(all="ignore"):
        for j in range(min(m, n)):
            imax, mx = j, 0.0
            for i in range(j, m):
                if abs(a[i, j]) > mx:
                    mx, imax = abs(a[i, j]), i
            if imax != j:
                a[[j, imax]] = a[[imax, j]]
                perm[j], perm[imax] = perm[imax], perm[j]
                nt += 1
            inv = np.float64(1.0) / a[j, j]
            a[j + 1:, j] *= inv
            a[j + 1:, j + 1:] -= np.outer(a[j + 1:, j], a[j, j + 1:])
    return a, perm, nt


def rand_lu(rng, m, n, ties=0):
    """A = P^T L U with |l| <= 1/2 (no ties unless asked for), pivots +-2^k, small integer U"""
    size = min(m, n)
    L = [[Fr(0)] * size for _ in range(m)]
    for i in range(m):
        for j in range(min(i, size)):
            L[i][j] = Fr(rng.choice([0, 0, 1, -1, 2, -2]), 4)
        if i < size:
            L[i][i] = Fr(1)
    U = [[Fr(0)] * n for _ in range(size)]
    for i in range(size):
        U[i][i] = Fr(rng.choice([1, -1, 2, -2, 4, -4]))
        for j in range(i + 1, n):
            U[i][j] = Fr(rng.randint(-3, 3))
    for _ in range(ties):  # |l| = 1: the first candidate in the CURRENT row order wins (strict comparison)
        j = rng.randrange(size - 1)
        i = rng.randrange(j + 1, m)
        L[i][j] = Fr(rng.choice([1, -1]))
    A = matmul(L, U)
    p = list(range(m))
    rng.shuffle(p)
    return [A[p[i]] for i in range(m)]


def rand_chol_factor(rng, n):
    L = [[Fr(0)] * n for _ in range(n)]
    for i in range(n):
        for j in range(i):
            L[i][j] = Fr(rng.choice([0, 0, 0, 1, -1, 2, -2, 1, -1]), rng.choice([1, 1, 2]))
        L[i][i] = Fr(rng.choice([1, 2, 4, 1, 2]))
    return L


def fl(mt):
    return [[exact_float(v) for v in r] for r in mt]


def main():
    rng = random.Random(20260923)
    out = {"about": "exact KATs, see tests/golden/make_exact_kats.py; matrices are lists of rows, lower-triangular ones (a_lower, l) "
                    "hold rows 0 .. i of row i only", "lu": [], "llt": [], "trsm": []}

    # ---------------- LU
    for name, m, n, ties in (("square8_ties", 8, 8, 5), ("tall10x6_ties", 10, 6, 3), ("wide5x9", 5, 9, 0), ("square136", 136, 136, 0)):
        while True:
            A = rand_lu(rng, m, n, ties)
            try:
                lu, piv, perm, nt = lu_exact([r[:min(m, n)] for r in A] if m < n else A)
            except AssertionError:
                continue
            if m < n:
                # factor.rs:278-285: the columns right of the square part get the interchanges and the unit-lower solve
                lu_sq = lu
                right = [A[perm[i]][m:] for i in range(m)]
                for j in range(m):
                    for i in range(j + 1, m):
                        l = lu_sq[i][j]
                        if l:
                            for c in range(n - m):
                                right[i][c] -= l * right[j][c]
                lu = [lu_sq[i] + right[i] for i in range(m)]
            if budget_ok([A, lu]):
                break
        if ties:
            assert any(abs(lu[i][j]) == 1 for j in range(min(m, n)) for i in range(j + 1, m)), "no tie survived"
        out["lu"].append({"name": name, "a": fl(A), "lu": fl(lu), "perm": perm, "transpositions": nt,
                          "rule": "lu/partial_pivoting/factor.rs:35-64,274-285"})
    # exactly singular LAST pivot (the zero is simply stored: no row below it)
    A = rand_lu(rng, 6, 6, 0)
    lu, piv, perm, nt = lu_exact(A)
    L = [[lu[i][j] if j < i else (Fr(1) if i == j else Fr(0)) for j in range(6)] for i in range(6)]
    U = [[lu[i][j] if j >= i else Fr(0) for j in range(6)] for i in range(6)]
    U[5][5] = Fr(0)
    A = matmul(L, U)
    inv = [0] * 6
    for i, p in enumerate(perm):
        inv[p] = i
    A = [A[inv[i]] for i in range(6)]
    lus, perms, nts = lu_float_sim(A)
    assert lus[5, 5] == 0.0 and not (lus != lus).any()
    out["lu"].append({"name": "singular_last_pivot6", "a": fl(A), "lu": lus.tolist(), "perm": perms, "transpositions": nts,
                      "rule": "factor.rs:35-43: an all-zero column keeps the diagonal"})
    # exactly singular MIDDLE column of a tall matrix: the reference divides by zero, 0 * inf = NaN from there on
    A = rand_lu(rng, 8, 5, 0)
    lu, piv, perm, nt = lu_exact(A)
    L = [[lu[i][j] if j < i else (Fr(1) if i == j else Fr(0)) for j in range(5)] for i in range(8)]
    U = [[lu[i][j] if j >= i else Fr(0) for j in range(5)] for i in range(5)]
    for i in range(8):
        L[i][2] = Fr(0) if i != 2 else Fr(1)
    U[2][2] = Fr(0)
    A = matmul(L, U)
    inv = [0] * 8
    for i, p in enumerate(perm):
        inv[p] = i
    A = [A[inv[i]] for i in range(8)]
    lus, perms, nts = lu_float_sim(A)
    assert (lus != lus).any()
    out["lu"].append({"name": "singular_middle_column8x5_nan", "a": fl(A), "lu": [[None if v != v else v for v in r] for r in lus.tolist()],
                      "perm": perms, "transpositions": nts, "rule": "factor.rs:45-64: recip(0) = inf, 0 * inf = NaN (None = NaN)"})

    # ---------------- LLT
    for name, n in (("n6", 6), ("n40", 40), ("n160", 160)):
        while True:
            L = rand_chol_factor(rng, n)
            A = matmul(L, transpose(L))
            if budget_ok([A, L]):
                break
        out["llt"].append({"name": name, "a_lower": fl([A[i][:i + 1] for i in range(n)]),
                           "l": fl([L[i][:i + 1] for i in range(n)]), "status": "ok", "rule": "cholesky/ldlt/factor.rs:147-174,367-498"})
    for name, n, j, pivot in (("n9_negative_pivot_at_5", 9, 5, Fr(-1)), ("n150_zero_pivot_at_131", 150, 131, Fr(0)), ("n7_zero_pivot_at_0", 7, 0, Fr(0))):
        L = rand_chol_factor(rng, n)
        A = matmul(L, transpose(L))
        A[j][j] = sum((L[j][k] * L[j][k] for k in range(j)), Fr(0)) + pivot  # d_j = pivot: `!(d > 0)` => Err(j)
        assert budget_ok([A])
        out["llt"].append({"name": name, "a_lower": fl([A[i][:i + 1] for i in range(n)]),
                           "status": "non_positive_pivot", "index": j, "rule": "factor.rs:163-168"})

    # ---------------- TRSM (lower; unit and generic diagonal)
    for name, n, k, unit in (("lower_n12_k5", 12, 5, False), ("unit_lower_n12_k5", 12, 5, True), ("lower_n150_k33", 150, 33, False),
                             ("unit_lower_n150_k33", 150, 33, True)):
        L = rand_chol_factor(rng, n)
        if unit:
            for i in range(n):
                L[i][i] = Fr(1)
        X = [[Fr(rng.randint(-4, 4)) for _ in range(k)] for _ in range(n)]
        B = matmul(L, X)
        assert budget_ok([B])
        Ls = [r[:] for r in L]
        if unit:
            for i in range(n):
                Ls[i][i] = Fr(rng.choice([3, 5, 7]))  # a unit solve must not read the stored diagonal
        out["trsm"].append({"name": name, "unit": unit, "l": fl([Ls[i][:i + 1] for i in range(n)]), "rhs": fl(B), "x": fl(X), "rule": "triangular_solve.rs"})

    path = os.path.join(HERE, "exact_kats.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
