"""The CPU oracle's bidiagonalization (oracle_bidiag_in_place, restating faer/src/linalg/svd/bidiag.rs:47-255) pinned the
way the reference pins its own (bidiag.rs:380-440, test_bidiag_real): U^H A V, computed by applying the block Householder
sequences stored in (UV, H_left) and (UV, H_right), equals the upper bidiagonal part of the output."""
import numpy as np
import pytest

from oracle import oracle as O


def bidiag_of(uv):
    m, n = uv.shape
    b = np.zeros_like(uv)
    for j in range(n):
        b[j, j] = uv[j, j]
        if j + 1 < n:
            b[j, j + 1] = uv[j, j + 1]
    return b


def uh_a_v(a, uv, hl, hr):
    """bidiag.rs:404-428: Q_left^H A, then columns 1.. times Q_right (as Q_right^T applied to their transpose)"""
    m, n = a.shape
    size = min(m, n)
    out = np.array(a, order="F")
    O.apply_householder_sequence_left(np.array(uv[:, :size], order="F"), hl, out, transpose=True)
    if size > 1:
        v = np.array(uv[: size - 1, 1:size].T, order="F")  # (size - 1) x (size - 1), unit lower: the right reflectors
        a1t = np.array(out[:, 1:size].T, order="F")
        O.apply_householder_sequence_left(v, hr, a1t, transpose=True)
        out[:, 1:size] = a1t.T
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m,n,bl,br", [(8, 4, 4, 3), (8, 8, 4, 3), (2, 1, 1, 1), (3, 3, 1, 1), (40, 17, 8, 8), (130, 129, 32, 16), (300, 64, 5, 7)])
def test_oracle_bidiag_reference_property(m, n, bl, br, dtype):
    rng = np.random.default_rng(m * 13 + n)
    a = np.asarray(rng.standard_normal((m, n)), dtype=dtype, order="F")
    uv = a.copy(order="F")
    hl = np.zeros((bl, n), dtype=dtype, order="F")
    hr = np.zeros((br, n - 1), dtype=dtype, order="F")
    O.bidiag_in_place(uv, hl, hr)
    b = bidiag_of(uv)
    got = uh_a_v(a, uv, hl, hr)
    eps = np.finfo(dtype).eps
    scale = np.linalg.norm(a.astype(np.float64), 2) * max(m, n)
    assert np.abs(got - b).max() <= 64 * eps * scale
    # singular values are preserved
    sv_a = np.linalg.svd(a.astype(np.float64), compute_uv=False)
    sv_b = np.linalg.svd(b.astype(np.float64), compute_uv=False)
    assert np.abs(sv_a - sv_b).max() <= 64 * eps * scale


def test_oracle_bidiag_edge_cases():
    # a zero matrix: every reflector is the identity (tau = +inf, householder.rs:70-77), the output is zero
    uv, hl, hr = np.zeros((5, 3), order="F"), np.zeros((2, 3), order="F"), np.zeros((2, 2), order="F")
    O.bidiag_in_place(uv, hl, hr)
    assert np.all(uv == 0) and np.isinf(hl[0, 0]) and np.isinf(hr[0, 0])
    # an already upper bidiagonal matrix keeps its entries up to sign
    n = 6
    b = np.diag(np.arange(1.0, n + 1)) + np.diag(np.full(n - 1, 0.5), 1)
    uv, hl, hr = np.array(b, order="F"), np.zeros((2, n), order="F"), np.zeros((2, n - 1), order="F")
    O.bidiag_in_place(uv, hl, hr)
    assert np.allclose(np.abs(bidiag_of(uv)), np.abs(b))
    # a wide matrix is processed over min(m, n) columns; the last row is left normalised (bidiag.rs:173-175 breaks before
    # the right reflector of the last step): documented behaviour of the reference, reproduced literally
    rng = np.random.default_rng(2)
    a = np.asarray(rng.standard_normal((3, 7)), order="F")
    uv, hl, hr = a.copy(order="F"), np.zeros((1, 3), order="F"), np.zeros((1, 2), order="F")
    O.bidiag_in_place(uv, hl, hr)
    assert abs(np.linalg.norm(uv[2, 3:]) - 1.0) <= 1e-14
