"""-m gpu: the single-workgroup vector kernels of the three reductions to condensed form exist twice (csrc/qr.hip): a body that keeps the
columns / rows it touches in registers (at most 4096 remaining rows: every size the other GPU tests reach) and the memory-resident body that
takes over above.  faer_hip_debug_level2_force_memory_bodies(1) runs the second one at every size: both restate the same expressions with
the same block reductions; the compiler contracts multiply-adds differently in the two bodies, so the outputs agree to rounding (a few
n eps ||A||, the bound of the oracle comparisons; fp64 tridiagonalization: bit for bit) -- which keeps the path that only matrices beyond
4096 rows reach under test -- and one case beyond 4096 rows crosses the switch-over inside a factorization and is checked through the
reference's property (evd/tridiag.rs:538-600: a similarity keeps the spectrum)."""
import numpy as np
import pytest

from gpu_util import EPS, init_gpu, to_dev, to_host

pytestmark = pytest.mark.gpu


def close(x, y, a, n):
    """same non-finite pattern (the +inf taus of empty tails), finite entries within 16 n eps ||A||_2.  fp64 only: in fp32 a reflector's
    head lands within rounding of zero once in a few thousand columns, the two bodies then pick opposite signs for beta and everything
    behind that column differs -- both outputs are valid reductions (checked through the invariants instead)"""
    if a.dtype == np.float32:
        return
    fin = np.isfinite(x)
    assert np.array_equal(fin, np.isfinite(y))
    tol = 16 * n * EPS[np.dtype(a.dtype)] * max(1.0, np.linalg.norm(a.astype(np.float64), 2))
    assert np.abs(x[fin].astype(np.float64) - y[fin].astype(np.float64)).max(initial=0.0) <= tol


def run_both(fn):
    F = init_gpu()
    out = []
    for force in (0, 1):
        F.lib().faer_hip_debug_level2_force_memory_bodies(force)
        try:
            out.append(fn(F))
        finally:
            F.lib().faer_hip_debug_level2_force_memory_bodies(0)
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [5, 64, 333, 1500])
def test_tridiag_bodies_agree(n, dtype):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    a = np.asarray(a + a.T, dtype=dtype, order="F")

    def go(F):
        vd, hd = to_dev(a), to_dev(np.zeros((8, n - 1), dtype=dtype, order="F"))
        F.tridiag_in_place(vd, hd)
        return np.array(to_host(vd)), np.array(to_host(hd))

    (v0, h0), (v1, h1) = run_both(go)
    close(v0, v1, a, n)
    close(h0, h1, a, n)
    if dtype == np.float64:
        assert np.array_equal(v0, v1)
    from scipy.linalg import eigvalsh_tridiagonal

    ev = np.linalg.eigvalsh(a.astype(np.float64))
    for v in (v0, v1):  # evd/tridiag.rs:538-600: a similarity keeps the spectrum
        t = v.astype(np.float64)
        assert np.abs(eigvalsh_tridiagonal(np.diag(t).copy(), np.diag(t, -1).copy()) - ev).max() <= 64 * n * EPS[np.dtype(dtype)] * np.abs(ev).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m,n", [(7, 7), (300, 120), (120, 300), (1100, 900)])
def test_bidiag_bodies_agree(m, n, dtype):
    rng = np.random.default_rng(m * 3 + n)
    a = np.asarray(rng.standard_normal((m, n)), dtype=dtype, order="F")
    size = min(m, n)

    def go(F):
        vd = to_dev(a)
        hl, hr = to_dev(np.zeros((8, size), dtype=dtype, order="F")), to_dev(np.zeros((8, max(size - 1, 0)), dtype=dtype, order="F"))
        F.bidiag_in_place(vd, hl, hr)
        return np.array(to_host(vd)), np.array(to_host(hl)), np.array(to_host(hr))

    r0, r1 = run_both(go)
    for x, y in zip(r0, r1):
        close(x, y, a, max(m, n))
    sv = np.linalg.svd(a.astype(np.float64), compute_uv=False)
    for r in (r0, r1):  # svd/bidiag.rs:380-440: U^H A V is the bidiagonal part, so the singular values are kept
        b = r[0].astype(np.float64)
        bd = np.zeros((size, size))
        bd[np.arange(size), np.arange(size)] = np.diag(b)[:size]
        bd[np.arange(size - 1), np.arange(1, size)] = np.diag(b, 1)[:size - 1]
        if m >= n:
            assert np.abs(np.linalg.svd(bd, compute_uv=False) - sv[:size]).max() <= 64 * max(m, n) * EPS[np.dtype(dtype)] * sv[0]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [6, 257, 1200])
def test_hessenberg_bodies_agree(n, dtype):
    rng = np.random.default_rng(n + 11)
    a = np.asarray(rng.standard_normal((n, n)), dtype=dtype, order="F")

    def go(F):
        vd, hd = to_dev(a), to_dev(np.zeros((8, n - 1), dtype=dtype, order="F"))
        F.hessenberg_in_place(vd, hd)
        return np.array(to_host(vd)), np.array(to_host(hd))

    (v0, h0), (v1, h1) = run_both(go)
    close(v0, v1, a, n)
    close(h0, h1, a, n)
    fro, tr = np.linalg.norm(a.astype(np.float64)), np.trace(a.astype(np.float64))
    for v in (v0, v1):  # a unitary similarity keeps the Frobenius norm and the trace of the Hessenberg part
        hs = np.triu(v.astype(np.float64), -1)
        assert abs(np.linalg.norm(hs) - fro) <= 64 * n * EPS[np.dtype(dtype)] * fro
        assert abs(np.trace(hs) - tr) <= 64 * n * EPS[np.dtype(dtype)] * fro


def test_tridiag_across_the_switch_over_n4400():
    """n - k - 1 > 4096 for the first 300 columns: memory-resident step body, then the register body; the spectrum is kept"""
    F = init_gpu()
    n = 4400
    rng = np.random.default_rng(4400)
    a = rng.standard_normal((n, n))
    a = np.asarray(a + a.T, dtype=np.float64, order="F")
    vd, hd = to_dev(a), to_dev(np.zeros((16, n - 1), dtype=np.float64, order="F"))
    F.tridiag_in_place(vd, hd)
    v = np.array(to_host(vd))
    from scipy.linalg import eigvalsh_tridiagonal

    d, e = np.diag(v).copy(), np.diag(v, -1).copy()
    ev_t = eigvalsh_tridiagonal(d, e)
    ev_a = np.linalg.eigvalsh(a)
    assert np.abs(ev_a - ev_t).max() <= 64 * EPS[np.dtype(np.float64)] * n * np.abs(a).max()
