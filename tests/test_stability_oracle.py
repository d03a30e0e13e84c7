"""CPU leg of the stability tests: the oracle (a restatement of the reference's substitution-based TRSM,
triangular_solve.rs:98-198,452-484) meets the componentwise backward-error bound on the ill-conditioned
triangles of stability_cases.py -- which pins the bound the GPU tests (test_gpu_stability.py) hold the
library to -- and an explicit-inverse block solver (round 1's algorithm) does NOT, i.e. the cases discriminate."""
import numpy as np
import pytest

import stability_cases as sc

EPS = np.finfo(np.float64).eps
C_TRI = 4.0  # |T X - B| <= C_TRI n eps (|T| |X| + |B|), componentwise


@pytest.mark.parametrize("kind", sc.TRI_KINDS)
@pytest.mark.parametrize("n", [100, 129, 700])
@pytest.mark.parametrize("small_solution", [False, True])
def test_oracle_trsm_is_backward_stable(oracle, kind, n, small_solution):
    rng = np.random.default_rng(n + len(kind))
    t = sc.triangle(kind, n, rng)
    b = sc.tri_rhs(t, 5, rng, small_solution)
    x = b.copy(order="F")
    with np.errstate(all="ignore"):
        oracle.trsm(np.asfortranarray(t), x, upper=False, unit=sc.is_unit(kind))
    if not np.isfinite(x).all():
        pytest.skip("solution overflows")
    assert sc.tri_backward_error(t, x, b) <= C_TRI * n * EPS


def test_cases_discriminate_explicit_block_inverses():
    """the round-1 algorithm (multiply by inverted 128 x 128 diagonal blocks) fails the bound by many orders of
    magnitude on the small-solution cases: the tests can tell the two algorithms apart"""
    rng = np.random.default_rng(1)
    n, nb = 300, 128
    worst = 0.0
    for kind in ("mixed", "kahan", "growth"):
        t = sc.triangle(kind, n, rng)
        b = sc.tri_rhs(t, 3, rng, True)
        x = b.copy()
        with np.errstate(all="ignore"):
            for j0 in range(0, n, nb):
                j1 = min(n, j0 + nb)
                x[j0:j1] = np.linalg.inv(t[j0:j1, j0:j1]) @ x[j0:j1]
                x[j1:] -= t[j1:, j0:j1] @ x[j0:j1]
        worst = max(worst, sc.tri_backward_error(t, x, b) / (n * EPS))
    assert worst > 1e6
