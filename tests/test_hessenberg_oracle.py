"""The CPU oracle's Hessenberg reduction (oracle_hessenberg_in_place, restating faer/src/linalg/evd/hessenberg.rs:230-408)
pinned the way the reference pins its own (hessenberg.rs:740-793, test_hessenberg_real): Q^H A Q through the block
Householder sequence of (V, H) equals the upper Hessenberg part of the output."""
import numpy as np
import pytest

from oracle import oracle as O
from test_tridiag_oracle import qh_a_q


def hess_of(v):
    return np.triu(v, -1)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,b", [(3, 3), (4, 3), (8, 3), (16, 3), (2, 1), (50, 8), (129, 32), (200, 1)])
def test_oracle_hessenberg_reference_property(n, b, dtype):
    rng = np.random.default_rng(n * 5 + b)
    a = np.asarray(rng.standard_normal((n, n)), dtype=dtype, order="F")
    v = a.copy(order="F")
    h = np.zeros((b, n - 1), dtype=dtype, order="F")
    O.hessenberg_in_place(v, h)
    hs = hess_of(v)
    got = qh_a_q(a, v, h)
    eps = np.finfo(dtype).eps
    scale = np.linalg.norm(a.astype(np.float64), 2) * n
    assert np.abs(got - hs).max() <= 64 * eps * scale
    # similarity: same spectrum (compared through the characteristic polynomial's symmetric functions is overkill:
    # sorted eigenvalues of a random matrix are well separated at these sizes)
    ev_a = np.sort_complex(np.linalg.eigvals(a.astype(np.float64)))
    ev_h = np.sort_complex(np.linalg.eigvals(hs.astype(np.float64)))
    assert np.abs(ev_a - ev_h).max() <= (1e-6 if dtype == np.float64 else 5e-2) * scale


def test_oracle_hessenberg_edge_cases():
    a = np.array([[2.5]], order="F")
    O.hessenberg_in_place(a, np.zeros((1, 0), order="F"))
    assert a[0, 0] == 2.5
    O.hessenberg_in_place(np.zeros((0, 0), order="F"), np.zeros((1, 0), order="F"))
    # already upper Hessenberg: all tails are zero, tau = +inf, the matrix is unchanged
    n = 6
    rng = np.random.default_rng(1)
    hmat = np.triu(rng.standard_normal((n, n)), -1)
    v, h = np.array(hmat, order="F"), np.zeros((2, n - 1), order="F")
    O.hessenberg_in_place(v, h)
    assert np.allclose(v, hmat)
    assert all(np.isinf(h[j % 2, j]) for j in range(n - 1))


@pytest.mark.parametrize("n", [256, 300, 515])
def test_unblocked_restatement_agrees_with_a_blocked_reduction(n):
    """From n = 256 (n^2 >= blocking_threshold) the reference runs hessenberg_gqvdg_blocked (evd/hessenberg.rs:563-565,
    568-736): the SAME reflectors of the SAME columns (beta = -sign(x_0) |x|, householder.rs:59-107) applied in blocked
    order.  The oracle restates only hessenberg_rearranged_unblocked (:230-408); this test documents that the variant does
    not matter beyond rounding by comparing the restatement with an independent BLOCKED reduction of the same sign
    convention, LAPACK's dgehrd (block reflectors, nb = 32): the Hessenberg matrices agree within c n eps ||A||, reflector j
    within that bound times ||A|| / |h(j + 1, j)| (a reflector x / (x_0 + sign |x|) moves by |dx| / |x|), and
    tau_faer = 1 / tau_lapack (H = I - v v^H / tau, :21-23)."""
    scipy_lapack = pytest.importorskip("scipy.linalg.lapack")
    rng = np.random.default_rng(n)
    a = np.asarray(rng.standard_normal((n, n)), order="F")
    v, h = a.copy(order="F"), np.zeros((1, n - 1), order="F")
    O.hessenberg_in_place(v, h)
    ht, tau, info = scipy_lapack.dgehrd(a, lo=0, hi=n - 1)
    assert info == 0
    eps = np.finfo(np.float64).eps
    scale = np.linalg.norm(a, 2)
    assert np.abs(hess_of(v) - hess_of(ht)).max() <= 64 * n * eps * scale
    sub = np.abs(np.diag(v, -1))
    cond = np.maximum(1.0, scale / np.where(sub != 0, sub, scale))
    for j in range(n - 2):
        assert np.abs(v[j + 2:, j] - ht[j + 2:, j]).max() <= 64 * n * eps * cond[j], j
        assert abs(h[0, j] - 1.0 / tau[j]) <= 64 * n * eps * cond[j] * max(1.0, abs(h[0, j])), j


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,b", [(3, 3), (4, 3), (8, 3), (16, 4), (2, 1), (50, 8), (129, 32), (65, 1), (256, 32), (300, 16)])
def test_oracle_blocked_hessenberg_reference_property(n, b, dtype):
    """hessenberg_gqvdg_blocked (evd/hessenberg.rs:568-736) restated: the reference's own property (:740-793) -- Q^H A Q
    through the block Householder sequence of (V, H) is the upper Hessenberg part of the output -- and the spectrum"""
    rng = np.random.default_rng(n * 7 + b)
    a = np.asarray(rng.standard_normal((n, n)), dtype=dtype, order="F")
    v = a.copy(order="F")
    h = np.zeros((b, n - 1), dtype=dtype, order="F")
    O.hessenberg_blocked_in_place(v, h)
    hs = hess_of(v)
    got = qh_a_q(a, v, h)
    eps = np.finfo(dtype).eps
    scale = np.linalg.norm(a.astype(np.float64), 2) * n
    assert np.abs(got - hs).max() <= 64 * eps * scale
    ev_a = np.sort_complex(np.linalg.eigvals(a.astype(np.float64)))
    ev_h = np.sort_complex(np.linalg.eigvals(hs.astype(np.float64)))
    assert np.abs(ev_a - ev_h).max() <= (1e-6 if dtype == np.float64 else 5e-2) * scale


@pytest.mark.parametrize("n,b", [(40, 8), (129, 32), (256, 32), (300, 16), (515, 32)])
def test_blocked_and_unblocked_restatements_agree(n, b):
    """the two variants of the reference make the same reflectors of the same columns in another order of operations:
    Hessenberg matrices within c n eps ||A||, reflector j and its tau within that times ||A|| / |h(j + 1, j)|, the block
    factors likewise (both are striu(V^H V) + diag(tau) of the same V)"""
    rng = np.random.default_rng(n + b)
    a = np.asarray(rng.standard_normal((n, n)), order="F")
    v1, h1 = a.copy(order="F"), np.zeros((b, n - 1), order="F")
    v2, h2 = a.copy(order="F"), np.zeros((b, n - 1), order="F")
    O.hessenberg_in_place(v1, h1)
    O.hessenberg_blocked_in_place(v2, h2)
    eps = np.finfo(np.float64).eps
    scale = np.linalg.norm(a, 2)
    assert np.abs(hess_of(v1) - hess_of(v2)).max() <= 64 * n * eps * scale
    sub = np.abs(np.diag(v1, -1))
    cond = np.maximum(1.0, scale / np.where(sub != 0, sub, scale))
    for j in range(n - 2):
        assert np.abs(v1[j + 2:, j] - v2[j + 2:, j]).max() <= 64 * n * eps * cond[j], j
    worst = cond.max()
    for j0 in range(0, n - 1, b):
        w = min(b, n - 1 - j0)
        t1, t2 = np.triu(h1[:w, j0:j0 + w]), np.triu(h2[:w, j0:j0 + w])
        fin = np.isfinite(t1)
        assert np.array_equal(fin, np.isfinite(t2))
        assert np.abs(t1[fin] - t2[fin]).max(initial=0) <= 64 * n * eps * worst * max(1.0, np.abs(t1[fin]).max(initial=0))


def test_reference_dispatch_matches_the_threshold():
    """evd/hessenberg.rs:562: n * n < 256 * 256 -> unblocked, else blocked"""
    rng = np.random.default_rng(3)
    for n, blocked in ((255, False), (256, True)):
        a = np.asarray(rng.standard_normal((n, n)), order="F")
        v, h = a.copy(order="F"), np.zeros((8, n - 1), order="F")
        O.hessenberg_reference_in_place(v, h)
        w, g = a.copy(order="F"), np.zeros((8, n - 1), order="F")
        (O.hessenberg_blocked_in_place if blocked else O.hessenberg_in_place)(w, g)
        assert np.array_equal(v, w) and np.array_equal(h, g)
