"""The CPU oracle's tridiagonalization (oracle_tridiag_in_place, restating faer/src/linalg/evd/tridiag.rs:274-535)
pinned the way the reference pins its own: evd/tridiag.rs:538-600 (test_tridiag_real) applies the block Householder
sequence stored in (V, H) to A from both sides and compares with the tridiagonal part of the output."""
import numpy as np
import pytest

from oracle import oracle as O


def tridiag_of(v):
    n = v.shape[0]
    t = np.zeros_like(v)
    for i in range(n):
        t[i, i] = v[i, i]
        if i + 1 < n:
            t[i + 1, i] = v[i + 1, i]
            t[i, i + 1] = v[i + 1, i]
    return t


def qh_a_q(a, v, h):
    """Q^H A Q with Q = the block Householder sequence of (v[1:, :n-1], h), as tridiag.rs:561-585 does it"""
    n = a.shape[0]
    out = np.array(a, order="F")
    for it in range(2):
        m = out if it == 0 else np.array(out.T, order="F")
        sub = np.array(m[1:, :], order="F")
        O.apply_householder_sequence_left(np.array(v[1:, : n - 1], order="F"), h, sub, transpose=True)
        m[1:, :] = sub
        out = m if it == 0 else np.array(m.T, order="F")
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,b", [(2, 3), (3, 3), (4, 3), (8, 3), (16, 3), (50, 8), (129, 32), (200, 1)])
def test_oracle_tridiag_reference_property(n, b, dtype):
    rng = np.random.default_rng(n * 7 + b)
    a = rng.standard_normal((n, n))
    a = np.asarray(a + a.T, dtype=dtype, order="F")
    v = a.copy(order="F")
    h = np.zeros((b, n - 1), dtype=dtype, order="F")
    O.tridiag_in_place(v, h)
    t = tridiag_of(v)
    got = qh_a_q(a, v, h)
    eps = np.finfo(dtype).eps
    scale = np.abs(a).max() * n
    assert np.abs(got - t).max() <= 64 * eps * scale
    # similarity: same spectrum
    ev_a = np.linalg.eigvalsh(a.astype(np.float64))
    ev_t = np.linalg.eigvalsh(t.astype(np.float64))
    assert np.abs(ev_a - ev_t).max() <= 64 * eps * scale
    # the strict upper triangle of the input is never written (tridiag.rs works on the lower triangle)
    iu = np.triu_indices(n, 1)
    assert np.array_equal(v[iu], a[iu])


def test_oracle_tridiag_edge_cases():
    # n = 1: nothing to do, H has no columns; n = 0: returns at once (tridiag.rs:288-290)
    a = np.array([[3.0]], order="F")
    h = np.zeros((2, 0), order="F")
    O.tridiag_in_place(a, h)
    assert a[0, 0] == 3.0
    a0 = np.zeros((0, 0), order="F")
    O.tridiag_in_place(a0, np.zeros((1, 0), order="F"))
    # already tridiagonal input: the tails are zero, tau = inf (householder.rs:70-77), T is the input
    n = 6
    t = np.diag(np.arange(1.0, n + 1)) + np.diag(np.full(n - 1, 0.5), -1) + np.diag(np.full(n - 1, 0.5), 1)
    v = np.array(t, order="F")
    h = np.zeros((2, n - 1), order="F")
    O.tridiag_in_place(v, h)
    assert np.allclose(tridiag_of(v), t)
    # block factors of width 2: tau on the diagonal of every block, v_i^H v_j = 0 above it
    assert all(np.isinf(h[j % 2, j]) for j in range(n - 1))
    assert all(h[0, j] == 0 for j in range(1, n - 1, 2))
