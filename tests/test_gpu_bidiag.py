"""-m gpu parity tests of faer_hip_bidiag_in_place (csrc/qr.hip, "Bidiagonalization") against the CPU oracle's
restatement of faer/src/linalg/svd/bidiag.rs:47-255 and the reference's own property test (bidiag.rs:380-440)."""
import numpy as np
import pytest

from gpu_util import EPS, init_gpu, to_dev, to_host
from oracle import oracle as O
from test_bidiag_oracle import bidiag_of, uh_a_v

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m,n,bl,br", [(8, 4, 4, 3), (8, 8, 4, 3), (2, 1, 1, 1), (1, 1, 1, 1), (3, 3, 1, 1), (17, 16, 2, 3), (40, 17, 8, 8), (130, 129, 32, 16),
                                        (300, 64, 5, 7), (700, 300, 32, 32), (2000, 33, 16, 8)])
def test_bidiag_vs_oracle(m, n, bl, br, dtype):
    F = init_gpu()
    rng = np.random.default_rng(m * 13 + n)
    a = np.asarray(rng.standard_normal((m, n)), dtype=dtype, order="F")
    uo, hlo, hro = a.copy(order="F"), np.zeros((bl, n), dtype=dtype, order="F"), np.zeros((br, n - 1), dtype=dtype, order="F")
    O.bidiag_in_place(uo, hlo, hro)
    ud, hld, hrd = to_dev(a), to_dev(np.zeros((bl, n), dtype=dtype, order="F")), to_dev(np.zeros((br, n - 1), dtype=dtype, order="F"))
    F.bidiag_in_place(ud, hld, hrd)
    u, hl, hr = to_host(ud), to_host(hld), to_host(hrd)
    eps = EPS[np.dtype(dtype)]
    scale = np.linalg.norm(a.astype(np.float64), 2)
    mx = max(m, n)
    # B scales with A: same algorithm, another summation order.  A reflector v = x / (x_0 + sign |x|) moves by
    # |dx| / |x| when its column moves by dx ~ mx eps ||A||, and |x| is the magnitude of the bidiagonal entry it produces:
    # the O(1) quantities (reflectors, block factors) are compared with that conditioning
    assert np.abs(bidiag_of(u) - bidiag_of(uo)).max() <= 64 * mx * eps * scale
    # ... PER REFLECTOR: the left reflector of column j is conditioned by the diagonal entry it produces, the right reflector
    # of row j by the superdiagonal one -- not by the smallest entry of the whole bidiagonal
    bo = bidiag_of(uo).astype(np.float64)
    dg = np.abs(np.diag(bo))[:min(m, n)]
    sg = np.abs(np.diag(bo, 1))
    cl = np.maximum(1.0, scale / np.where(dg != 0, dg, scale))
    cr = np.maximum(1.0, scale / np.where(sg != 0, sg, scale))
    for j in range(min(m, n)):
        assert np.abs(u[j + 1:, j] - uo[j + 1:, j]).max(initial=0.0) <= 64 * mx * eps * cl[j], ("left", j)
        if j + 2 < n:
            assert np.abs(u[j, j + 2:] - uo[j, j + 2:]).max(initial=0.0) <= 64 * mx * eps * cr[j], ("right", j)
    for h, ho, cc, bb in ((hl, hlo, cl, bl), (hr, hro, cr, br)):
        fin = np.isfinite(ho)
        assert np.array_equal(np.isfinite(h), fin)
        for j in range(ho.shape[1]):  # column j of a block factor couples the reflectors of its block up to j
            cj = cc[(j // bb) * bb:j + 1].max(initial=1.0)
            fj = fin[:, j]
            assert np.abs(h[fj, j] - ho[fj, j]).max(initial=0.0) <= 64 * mx * eps * cj, j


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m,n,bl,br", [(8, 8, 4, 3), (200, 90, 8, 16), (515, 515, 32, 32)])
def test_bidiag_reference_property(m, n, bl, br, dtype):
    """bidiag.rs:380-440: U^H A V through the two block Householder sequences is the bidiagonal part of the output;
    the singular values are preserved."""
    F = init_gpu()
    rng = np.random.default_rng(m + n)
    a = np.asarray(rng.standard_normal((m, n)), dtype=dtype, order="F")
    ud, hld, hrd = to_dev(a), to_dev(np.zeros((bl, n), dtype=dtype, order="F")), to_dev(np.zeros((br, n - 1), dtype=dtype, order="F"))
    F.bidiag_in_place(ud, hld, hrd)
    u, hl, hr = (np.array(to_host(t), order="F") for t in (ud, hld, hrd))
    b = bidiag_of(u)
    eps = EPS[np.dtype(dtype)]
    scale = np.linalg.norm(a.astype(np.float64), 2) * max(m, n)
    assert np.abs(uh_a_v(a, u, hl, hr) - b).max() <= 64 * eps * scale
    sv_a = np.linalg.svd(a.astype(np.float64), compute_uv=False)
    sv_b = np.linalg.svd(b.astype(np.float64), compute_uv=False)
    assert np.abs(sv_a - sv_b).max() <= 64 * eps * scale


@pytest.mark.parametrize("m,n", [(4, 8), (17, 40), (1, 5), (64, 65)])
def test_bidiag_wide_matrix_like_the_reference(m, n):
    """m < n: the reference loops over min(m, n) columns and leaves the last row normalised (bidiag.rs:173-175 breaks
    before the right reflector); its SVD never passes such a matrix, the entry point reproduces the behaviour anyway"""
    F = init_gpu()
    rng = np.random.default_rng(m * 3 + n)
    a = np.asarray(rng.standard_normal((m, n)), order="F")
    uo, hlo, hro = a.copy(order="F"), np.zeros((2, m), order="F"), np.zeros((2, m - 1), order="F")
    O.bidiag_in_place(uo, hlo, hro)
    ud, hld, hrd = to_dev(a), to_dev(np.zeros((2, m), order="F")), to_dev(np.zeros((2, m - 1), order="F"))
    F.bidiag_in_place(ud, hld, hrd)
    tol = 1e-10 * max(1.0, np.abs(uo).max())
    assert np.abs(to_host(ud) - uo).max() <= tol
    for h, ho in ((to_host(hld), hlo), (to_host(hrd), hro)):
        fin = np.isfinite(ho)
        assert np.array_equal(np.isfinite(h), fin) and np.abs(h[fin] - ho[fin]).max(initial=0.0) <= tol


def test_bidiag_singular_values_n1500():
    F = init_gpu()
    m, n = 2500, 1500
    rng = np.random.default_rng(3)
    a = np.asarray(rng.standard_normal((m, n)), order="F")
    ud, hld, hrd = to_dev(a), to_dev(np.zeros((32, n), order="F")), to_dev(np.zeros((32, n - 1), order="F"))
    F.bidiag_in_place(ud, hld, hrd)
    b = bidiag_of(to_host(ud))[:n, :n]
    sv_a = np.linalg.svd(a, compute_uv=False)
    sv_b = np.linalg.svd(b, compute_uv=False)
    assert np.abs(sv_a - sv_b).max() <= 64 * m * EPS[np.dtype(np.float64)] * sv_a[0]


def test_bidiag_layouts_host_operands_and_edges():
    F = init_gpu()
    m, n, bl, br = 90, 60, 8, 4
    rng = np.random.default_rng(17)
    a = np.asarray(rng.standard_normal((m, n)), order="F")
    uo, hlo, hro = a.copy(order="F"), np.zeros((bl, n), order="F"), np.zeros((br, n - 1), order="F")
    O.bidiag_in_place(uo, hlo, hro)
    tol = 64 * m * EPS[np.dtype(np.float64)] * np.linalg.norm(a, 2)
    # row major on the device
    ud, hld, hrd = to_dev(a, order="C"), to_dev(np.zeros((bl, n)), order="C"), to_dev(np.zeros((br, n - 1)), order="C")
    F.bidiag_in_place(ud, hld, hrd)
    assert np.abs(to_host(ud) - uo).max() <= tol
    # host operands, a view inside a larger host matrix: the parent's other entries stay untouched
    big = np.full((m + 5, n + 3), -3.5, order="F")
    big[2:2 + m, 1:1 + n] = a
    hl, hr = np.zeros((bl, n), order="F"), np.zeros((br, n - 1), order="F")
    F.bidiag_in_place(big[2:2 + m, 1:1 + n], hl, hr)
    assert np.abs(big[2:2 + m, 1:1 + n] - uo).max() <= tol
    outside = np.ones_like(big, dtype=bool)
    outside[2:2 + m, 1:1 + n] = False
    assert np.all(big[outside] == -3.5)
    # an already upper bidiagonal matrix: every tail is zero, all taus are +inf, B is the input
    n2 = 6
    bmat = np.diag(np.arange(1.0, n2 + 1)) + np.diag(np.full(n2 - 1, 0.5), 1)
    ud, hld, hrd = to_dev(np.array(bmat, order="F")), to_dev(np.zeros((2, n2), order="F")), to_dev(np.zeros((2, n2 - 1), order="F"))
    uo2, hlo2, hro2 = np.array(bmat, order="F"), np.zeros((2, n2), order="F"), np.zeros((2, n2 - 1), order="F")
    O.bidiag_in_place(uo2, hlo2, hro2)
    F.bidiag_in_place(ud, hld, hrd)
    assert np.allclose(np.abs(bidiag_of(to_host(ud))), np.abs(bidiag_of(uo2)))
    assert np.array_equal(np.isinf(to_host(hld)), np.isinf(hlo2)) and np.array_equal(np.isinf(to_host(hrd)), np.isinf(hro2))
    # empty matrix
    F.bidiag_in_place(to_dev(np.zeros((3, 0), order="F")), to_dev(np.zeros((1, 0), order="F")), to_dev(np.zeros((1, 0), order="F")))
