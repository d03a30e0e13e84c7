"""CPU prototype (numpy) of the one-pass tall-skinny QR of csrc/tsqr.hip -- the small-matrix algebra only.

Checks, against the oracle's Householder QR (test infrastructure), that
  R~ = chol(P^T P)^T (fp64 Gram of fp32 data), Q1~ = A1 R~^-1, sign-choosing LU of I - Q1~ S = V1 U,
  R = S R~, M = -(U R)^-1, V = (P - [R; 0]) M, T = V1^T U^-1,
  trailing: D = R^-T (P^T X) (new top rows), Y = R^-1 (V1 U)^-1 (D - X_top), X' = X - P Y below the top block
reproduce faer's (V, T, R) (householder.rs:59-107, qr/no_pivoting/factor.rs:137-256) up to rounding.
Run: python tests/diag/proto_tsqr.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle  # noqa: E402

PW = 64


def panel_small(G, A1):
    """G: w x w Gram (fp64), A1: w x w top block.  Returns R, V1, U, M, T, S (all fp64)."""
    w = G.shape[0]
    Rt = np.linalg.cholesky(G).T
    W = np.linalg.solve(Rt.T, A1.T).T  # A1 Rt^-1
    s = np.zeros(w)
    L = np.eye(w)
    U = np.zeros((w, w))
    for j in range(w):
        alpha = W[j, j]
        s[j] = -1.0 if alpha >= 0 else 1.0
        piv = 1.0 - s[j] * alpha
        L[j + 1:, j] = -s[j] * W[j + 1:, j] / piv
        W[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], W[j, j + 1:])
    U = np.triu(np.eye(w) - W * s[None, :])
    R = s[:, None] * Rt
    Ui = np.linalg.inv(U)
    M = -np.linalg.solve(R, Ui)
    T = np.triu(L.T @ Ui)
    return R, L, U, M, T, s


def tsqr(a, f32=True):
    """a: m x n (fp32 data).  Returns packed qr (R upper, V below) and the full n x n T = striu(V^T V) + diag(tau)."""
    dt = np.float32 if f32 else np.float64
    A = a.astype(dt).copy()
    m, n = A.shape
    Tfull = np.zeros((n, n))
    for c0 in range(0, n, PW):
        w = min(PW, n - c0)
        P = A[c0:, c0:c0 + w]
        X = A[c0:, c0 + w:]
        G = P.astype(np.float64).T @ P.astype(np.float64)
        C = (P.T @ X).astype(np.float64)  # fp32 products in the kernel; partial sums in fp64
        R, V1, U, M, T, s = panel_small(G, P[:w].astype(np.float64))
        Tfull[c0:c0 + w, c0:c0 + w] = T
        if X.shape[1]:
            D = np.linalg.solve(R.T, C)
            E = D - X[:w].astype(np.float64)
            Y = np.linalg.solve(R, np.linalg.solve(U, np.linalg.solve(V1, E)))
            X[w:] = X[w:] - P[w:] @ Y.astype(dt)
            X[:w] = D.astype(dt)
        Vb = P[w:] @ M.astype(dt)
        P[w:] = Vb
        P[:w] = (np.triu(R) + np.tril(V1, -1)).astype(dt)
    V = np.tril(A.astype(np.float64), -1)[:, :n]
    V[np.arange(n), np.arange(n)] = 1.0
    VtV = V.T @ V
    Tfull = np.triu(VtV, 1) * (Tfull == 0) + Tfull  # cross-panel blocks: honest Gram of V
    return A, Tfull


def main():
    rng = np.random.default_rng(0)
    for (m, n, scale) in [(4000, 64, None), (20000, 256, None), (9000, 100, None), (6000, 128, "graded")]:
        a = rng.standard_normal((m, n)).astype(np.float32)
        if scale == "graded":
            a = (a.astype(np.float64) @ (np.eye(n) + 0.5 * rng.standard_normal((n, n)) / np.sqrt(n))).astype(np.float32)
        ref = a.copy(order="F")
        bs = n
        rh = np.zeros((bs, n), dtype=np.float32, order="F")
        assert oracle.qr_in_place(ref, rh) == n
        got, T = tsqr(a)
        ref64 = a.astype(np.float64).copy(order="F")
        rh64 = np.zeros((bs, n), order="F")
        oracle.qr_in_place(ref64, rh64)
        up = np.triu(np.ones((m, n), bool))
        e = np.finfo(np.float32).eps
        dR = np.abs(got - ref64)[up].max() / np.abs(ref64[up]).max()
        dV = np.abs(got - ref64)[~up].max()
        dR32 = np.abs(ref - ref64)[up].max() / np.abs(ref64[up]).max()
        dV32 = np.abs(ref - ref64)[~up].max()
        dT = np.abs(np.triu(T) - np.triu(rh64)).max() / np.abs(rh64).max()
        dT32 = np.abs(np.triu(rh.astype(np.float64)) - np.triu(rh64)).max() / np.abs(rh64).max()
        print(f"{m}x{n} {scale}: new vs fp64 oracle: R {dR / e:.2f} eps, V {dV / e:.2f} eps, T {dT / e:.2f} eps | "
              f"fp32 oracle vs fp64 oracle: R {dR32 / e:.2f}, V {dV32 / e:.2f}, T {dT32 / e:.2f}")


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def tsqr_algebraic_t(a):
    """same factorization; the cross-panel blocks of T from small matrices only (no pass over V):
    T_kl = -( T_k Z_k[:, l] + sum_{k<j<l} T_kj Z_j[:, l] + V_k[c_k : c_{l+1}]^T R[c_k : c_{l+1}, l] ) M_l,
    with Z_k = -V1^-1 (D - X_top) of step k (V_k^T X' = -T_k Z_k after the block reflector has been applied)."""
    dt = np.float32
    A = a.astype(dt).copy()
    m, n = A.shape
    npan = (n + PW - 1) // PW
    T = np.zeros((n, n))
    Z = [None] * npan
    Ms = [None] * npan
    for k in range(npan):
        c0 = k * PW
        w = min(PW, n - c0)
        P = A[c0:, c0:c0 + w]
        X = A[c0:, c0 + w:]
        G = P.astype(np.float64).T @ P.astype(np.float64)
        C = (P.T @ X).astype(np.float64)
        R, V1, U, M, Tk, s = panel_small(G, P[:w].astype(np.float64))
        T[c0:c0 + w, c0:c0 + w] = Tk
        Ms[k] = M
        if X.shape[1]:
            D = np.linalg.solve(R.T, C)
            E = D - X[:w].astype(np.float64)
            Zk = -np.linalg.solve(V1, E)
            Z[k] = np.zeros((w, n))
            Z[k][:, c0 + w:] = Zk
            Y = np.linalg.solve(R, np.linalg.solve(U, -Zk))
            X[w:] = X[w:] - P[w:] @ Y.astype(dt)
            X[:w] = D.astype(dt)
        P[w:] = P[w:] @ M.astype(dt)
        P[:w] = (np.triu(R) + np.tril(V1, -1)).astype(dt)
    A64 = A.astype(np.float64)
    for k in range(npan):
        ck = k * PW
        for l in range(k + 1, npan):
            cl = l * PW
            wl = min(PW, n - cl)
            cols = slice(cl, cl + wl)
            acc = T[ck:ck + PW, ck:ck + PW] @ Z[k][:, cols]
            for j in range(k + 1, l):
                cj = j * PW
                acc += T[ck:ck + PW, cj:cj + PW] @ Z[j][:, cols]
            Vk = np.tril(A64[ck:cl + wl, ck:ck + PW], -1)
            Vk[np.arange(PW), np.arange(PW)] = 1.0
            Rl = A64[ck:cl + wl, cols].copy()
            Rl[cl - ck:] = np.triu(Rl[cl - ck:])
            acc += Vk.T @ Rl
            T[ck:ck + PW, cols] = -acc @ Ms[l]
    return A, T


def check_algebraic():
    rng = np.random.default_rng(1)
    for (m, n) in [(20000, 256), (9000, 200)]:
        a = rng.standard_normal((m, n)).astype(np.float32)
        got, T = tsqr_algebraic_t(a)
        V = np.tril(got.astype(np.float64), -1)[:, :n]
        V[np.arange(n), np.arange(n)] = 1.0
        VtV = V.T @ V
        ref64 = a.astype(np.float64).copy(order="F")
        rh64 = np.zeros((n, n), order="F")
        oracle.qr_in_place(ref64, rh64)
        e = np.finfo(np.float32).eps
        print(f"{m}x{n}: algebraic T vs Gram of the stored V: {np.abs(np.triu(T, 1) - np.triu(VtV, 1)).max() / e:.2f} eps; "
              f"vs fp64 oracle T: {np.abs(np.triu(T) - np.triu(rh64)).max() / e:.2f} eps (max |T| {np.abs(rh64).max():.2f})")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "algebraic":
    check_algebraic()
