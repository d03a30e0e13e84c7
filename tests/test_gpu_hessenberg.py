"""-m gpu parity tests of faer_hip_hessenberg_in_place (csrc/qr.hip, "Hessenberg reduction") against the CPU oracle's
restatement of faer/src/linalg/evd/hessenberg.rs:230-408 and the reference's own property test (hessenberg.rs:740-793)."""
import numpy as np
import pytest

from gpu_util import EPS, init_gpu, to_dev, to_host
from oracle import oracle as O
from test_hessenberg_oracle import hess_of
from test_tridiag_oracle import qh_a_q

pytestmark = pytest.mark.gpu


def _vs_oracle(n, b, dtype, oracle_fn):
    F = init_gpu()
    rng = np.random.default_rng(n * 5 + b)
    a = np.asarray(rng.standard_normal((n, n)), dtype=dtype, order="F")
    vo, ho = a.copy(order="F"), np.zeros((b, n - 1), dtype=dtype, order="F")
    oracle_fn(vo, ho)
    vd, hd = to_dev(a), to_dev(np.zeros((b, n - 1), dtype=dtype, order="F"))
    F.hessenberg_in_place(vd, hd)
    v, h = to_host(vd), to_host(hd)
    eps = EPS[np.dtype(dtype)]
    scale = np.linalg.norm(a.astype(np.float64), 2)
    # H scales with A: same algorithm, another summation order.  A reflector moves by |dx| / |x| when its column moves by
    # dx ~ n eps ||A||, |x| being the subdiagonal entry it produces: the O(1) quantities are compared with that conditioning
    assert np.abs(hess_of(v) - hess_of(vo)).max() <= 64 * n * eps * scale
    # ... PER COLUMN: reflector j is conditioned by its own subdiagonal entry, not by the smallest one of the matrix
    sub = np.abs(np.diag(vo, -1)).astype(np.float64)
    cond = np.maximum(1.0, scale / np.where(sub != 0, sub, scale))  # one factor per reflector (column j -> entry (j + 1, j))
    for j in range(n - 2):
        assert np.abs(v[j + 2:, j] - vo[j + 2:, j]).max(initial=0.0) <= 64 * n * eps * cond[j], j
    fin = np.isfinite(ho)
    assert np.array_equal(np.isfinite(h), fin)
    for j in range(n - 1):  # column j of a block factor couples the reflectors of its block up to j
        cj = cond[(j // b) * b:j + 1].max()
        fj = fin[:, j]
        assert np.abs(h[fj, j] - ho[fj, j]).max(initial=0.0) <= 64 * n * eps * cj, j


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,b", [(2, 1), (3, 3), (4, 3), (8, 3), (16, 3), (17, 1), (33, 8), (50, 8), (129, 32), (300, 16), (700, 32)])
def test_hessenberg_vs_oracle(n, b, dtype):
    """against the restatement of hessenberg_rearranged_unblocked (evd/hessenberg.rs:230-408) at every size"""
    _vs_oracle(n, b, dtype, O.hessenberg_in_place)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,b", [(256, 32), (300, 16), (515, 32), (700, 8)])
def test_hessenberg_vs_the_variant_the_reference_runs(n, b, dtype):
    """n * n >= 256 * 256: the reference runs hessenberg_gqvdg_blocked (evd/hessenberg.rs:562-566, 568-736); the oracle
    restates that variant too (oracle_hessenberg_blocked_in_place) and the GPU path -- which runs the level-2 algorithm at
    every size -- is held to the same per-reflector bounds against it"""
    _vs_oracle(n, b, dtype, O.hessenberg_reference_in_place)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,b", [(16, 3), (200, 8), (515, 32)])
def test_hessenberg_reference_property(n, b, dtype):
    """hessenberg.rs:740-793: Q^H A Q through the block Householder sequence of (V, H) is the upper Hessenberg part of the
    output (n = 515 is on the reference's BLOCKED side of its size switch: same property, :859-900)."""
    F = init_gpu()
    rng = np.random.default_rng(n + b)
    a = np.asarray(rng.standard_normal((n, n)), dtype=dtype, order="F")
    vd, hd = to_dev(a), to_dev(np.zeros((b, n - 1), dtype=dtype, order="F"))
    F.hessenberg_in_place(vd, hd)
    v, h = np.array(to_host(vd), order="F"), np.array(to_host(hd), order="F")
    eps = EPS[np.dtype(dtype)]
    scale = np.linalg.norm(a.astype(np.float64), 2) * n
    assert np.abs(qh_a_q(a, v, h) - hess_of(v)).max() <= 64 * eps * scale


def test_hessenberg_spectrum_n1500():
    F = init_gpu()
    n = 1500
    rng = np.random.default_rng(9)
    a = np.asarray(rng.standard_normal((n, n)), order="F")
    vd, hd = to_dev(a), to_dev(np.zeros((32, n - 1), order="F"))
    F.hessenberg_in_place(vd, hd)
    hs = hess_of(to_host(vd))
    # similarity invariants that do not need an eigenvalue matching: trace and Frobenius norm (Q is orthogonal)
    assert abs(np.trace(hs) - np.trace(a)) <= 64 * n * EPS[np.dtype(np.float64)] * np.linalg.norm(a, 2)
    assert abs(np.linalg.norm(hs) - np.linalg.norm(a)) <= 64 * n * EPS[np.dtype(np.float64)] * np.linalg.norm(a)
    import scipy.linalg as sla

    ev_a = np.sort_complex(np.linalg.eigvals(a))
    ev_h = np.sort_complex(sla.eigvals(hs))
    assert np.abs(ev_a - ev_h).max() <= 1e-8 * np.linalg.norm(a, 2)


def test_hessenberg_layouts_host_operands_and_edges():
    F = init_gpu()
    n, b = 120, 8
    rng = np.random.default_rng(21)
    a = np.asarray(rng.standard_normal((n, n)), order="F")
    vo, ho = a.copy(order="F"), np.zeros((b, n - 1), order="F")
    O.hessenberg_in_place(vo, ho)
    tol = 64 * n * EPS[np.dtype(np.float64)] * np.linalg.norm(a, 2)
    vd, hd = to_dev(a, order="C"), to_dev(np.zeros((b, n - 1)), order="C")
    F.hessenberg_in_place(vd, hd)
    assert np.abs(hess_of(to_host(vd)) - hess_of(vo)).max() <= tol
    big = np.full((n + 4, n + 6), 1.75, order="F")
    big[1:1 + n, 3:3 + n] = a
    hh = np.zeros((b, n - 1), order="F")
    F.hessenberg_in_place(big[1:1 + n, 3:3 + n], hh)
    assert np.abs(hess_of(big[1:1 + n, 3:3 + n]) - hess_of(vo)).max() <= tol
    outside = np.ones_like(big, dtype=bool)
    outside[1:1 + n, 3:3 + n] = False
    assert np.all(big[outside] == 1.75)
    # n = 0, 1; an already upper Hessenberg matrix (all taus +inf, unchanged)
    F.hessenberg_in_place(to_dev(np.zeros((0, 0), order="F")), to_dev(np.zeros((1, 0), order="F")))
    a1 = to_dev(np.array([[2.5]], order="F"))
    F.hessenberg_in_place(a1, to_dev(np.zeros((1, 0), order="F")))
    assert to_host(a1)[0, 0] == 2.5
    hm = np.triu(rng.standard_normal((6, 6)), -1)
    vd, hd = to_dev(np.array(hm, order="F")), to_dev(np.zeros((2, 5), order="F"))
    F.hessenberg_in_place(vd, hd)
    assert np.allclose(to_host(vd), hm)
    assert all(np.isinf(to_host(hd)[j % 2, j]) for j in range(5))
