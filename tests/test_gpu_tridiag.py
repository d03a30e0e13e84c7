"""-m gpu parity tests of faer_hip_tridiag_in_place (csrc/qr.hip, "Tridiagonalization") against the CPU oracle's
restatement of faer/src/linalg/evd/tridiag.rs:274-535 and the reference's own property test (tridiag.rs:538-600)."""
import numpy as np
import pytest

from gpu_util import EPS, init_gpu, to_dev, to_host
from oracle import oracle as O
from test_tridiag_oracle import qh_a_q, tridiag_of

pytestmark = pytest.mark.gpu


def sym(rng, n, dtype):
    a = rng.standard_normal((n, n))
    return np.asarray(a + a.T, dtype=dtype, order="F")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,b", [(2, 3), (3, 3), (4, 3), (8, 3), (16, 3), (17, 1), (33, 8), (50, 8), (129, 32), (300, 16), (700, 32)])
def test_tridiag_vs_oracle(n, b, dtype):
    F = init_gpu()
    rng = np.random.default_rng(n * 7 + b)
    a = sym(rng, n, dtype)
    vo, ho = a.copy(order="F"), np.zeros((b, n - 1), dtype=dtype, order="F")
    O.tridiag_in_place(vo, ho)
    vd, hd = to_dev(a), to_dev(np.zeros((b, n - 1), dtype=dtype, order="F"))
    F.tridiag_in_place(vd, hd)
    v, h = to_host(vd), to_host(hd)
    eps = EPS[np.dtype(dtype)]
    scale = np.linalg.norm(a.astype(np.float64), 2)  # the reduction is normwise backward stable: errors scale with ||A||_2
    # T (scales with A) and the reflectors / block factors (O(1) quantities): same algorithm, another summation order
    assert np.abs(tridiag_of(v) - tridiag_of(vo)).max() <= 64 * n * eps * scale
    il = np.tril_indices(n, -2)
    assert np.abs(v[il] - vo[il]).max(initial=0.0) <= 64 * n * eps
    fin = np.isfinite(ho)
    assert np.array_equal(np.isfinite(h), fin)
    assert np.abs(h[fin] - ho[fin]).max(initial=0.0) <= 64 * n * eps
    # the strict upper triangle is never written (tridiag.rs works on the lower triangle)
    iu = np.triu_indices(n, 1)
    assert np.array_equal(v[iu], a[iu])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,b", [(16, 3), (200, 8), (515, 32)])
def test_tridiag_reference_property(n, b, dtype):
    """tridiag.rs:538-600: Q^H A Q (block Householder sequence of (V, H) applied from both sides) is the tridiagonal
    part of the output; and a similarity keeps the spectrum."""
    F = init_gpu()
    rng = np.random.default_rng(n + b)
    a = sym(rng, n, dtype)
    vd, hd = to_dev(a), to_dev(np.zeros((b, n - 1), dtype=dtype, order="F"))
    F.tridiag_in_place(vd, hd)
    v, h = np.array(to_host(vd), order="F"), np.array(to_host(hd), order="F")
    t = tridiag_of(v)
    eps = EPS[np.dtype(dtype)]
    scale = np.abs(a).max() * n
    assert np.abs(qh_a_q(a, v, h) - t).max() <= 64 * eps * scale
    ev_a = np.linalg.eigvalsh(a.astype(np.float64))
    ev_t = np.linalg.eigvalsh(t.astype(np.float64))
    assert np.abs(ev_a - ev_t).max() <= 64 * eps * scale


def test_tridiag_spectrum_n2000():
    F = init_gpu()
    n = 2000
    rng = np.random.default_rng(5)
    a = sym(rng, n, np.float64)
    vd, hd = to_dev(a), to_dev(np.zeros((32, n - 1), order="F"))
    F.tridiag_in_place(vd, hd)
    t = tridiag_of(to_host(vd))
    import scipy.linalg as sla

    ev_t = sla.eigvalsh_tridiagonal(np.diag(t).copy(), np.diag(t, -1).copy())
    ev_a = np.linalg.eigvalsh(a)
    assert np.abs(ev_a - ev_t).max() <= 64 * n * EPS[np.dtype(np.float64)] * np.abs(a).max()


def test_tridiag_layouts_and_host_operands():
    """row-major device operand, host (numpy) operands staged by the library, a submatrix view of a larger host matrix"""
    F = init_gpu()
    n, b = 150, 8
    rng = np.random.default_rng(11)
    a = sym(rng, n, np.float64)
    vo, ho = a.copy(order="F"), np.zeros((b, n - 1), order="F")
    O.tridiag_in_place(vo, ho)
    il = np.tril_indices(n)
    tol = 64 * n * EPS[np.dtype(np.float64)] * np.abs(a).max()
    # row major on the device
    vd, hd = to_dev(a, order="C"), to_dev(np.zeros((b, n - 1)), order="C")
    F.tridiag_in_place(vd, hd)
    assert np.abs(to_host(vd)[il] - vo[il]).max() <= tol
    # host operands
    vh, hh = a.copy(order="F"), np.zeros((b, n - 1), order="F")
    F.tridiag_in_place(vh, hh)
    assert np.abs(vh[il] - vo[il]).max() <= tol
    # a view inside a larger host matrix: the parent's other entries stay untouched
    big = np.full((n + 7, n + 5), 7.25, order="F")
    big[3:3 + n, 2:2 + n] = a
    F.tridiag_in_place(big[3:3 + n, 2:2 + n], hh)
    assert np.abs(big[3:3 + n, 2:2 + n][il] - vo[il]).max() <= tol
    mask = np.ones_like(big, dtype=bool)
    mask[3:3 + n, 2:2 + n] = False
    assert np.all(big[mask] == 7.25)


def test_tridiag_edge_cases():
    F = init_gpu()
    # n = 0 / n = 1: nothing to do (tridiag.rs:288-290)
    F.tridiag_in_place(to_dev(np.zeros((0, 0), order="F")), to_dev(np.zeros((1, 0), order="F")))
    a1 = to_dev(np.array([[3.0]], order="F"))
    F.tridiag_in_place(a1, to_dev(np.zeros((2, 0), order="F")))
    assert to_host(a1)[0, 0] == 3.0
    # already tridiagonal: every tail is zero, tau = +inf (householder.rs:70-77), T is the input
    n = 6
    t = np.diag(np.arange(1.0, n + 1)) + np.diag(np.full(n - 1, 0.5), -1) + np.diag(np.full(n - 1, 0.5), 1)
    vd, hd = to_dev(np.array(t, order="F")), to_dev(np.zeros((2, n - 1), order="F"))
    F.tridiag_in_place(vd, hd)
    v, h = to_host(vd), to_host(hd)
    assert np.allclose(tridiag_of(v), t)
    assert all(np.isinf(h[j % 2, j]) for j in range(n - 1))
    # a zero matrix
    vd, hd = to_dev(np.zeros((5, 5), order="F")), to_dev(np.zeros((2, 4), order="F"))
    F.tridiag_in_place(vd, hd)
    assert np.all(to_host(vd) == 0)
