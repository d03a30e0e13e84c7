"""The schedule of the round-4 LU panel kernel (labels instead of interchanges, one exchange per column, lagging rank-1
updates with consumer-side correction of the published rows) reproduces the reference's unblocked elimination
(faer/src/linalg/lu/partial_pivoting/factor.rs:19-67) exactly: numpy model, no GPU (tests/diag/proto_lu_wpanel.py)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "diag"))
from proto_lu_wpanel import lu_small_leaf_model, lu_unblocked, lu_wpanel_model  # noqa: E402


@pytest.mark.parametrize("m,w,rows", [(64, 16, 8), (100, 32, 16), (33, 33, 7), (200, 64, 64), (50, 20, 50), (10, 10, 3), (129, 64, 64)])
@pytest.mark.parametrize("kind", ["gauss", "ties", "zero_column", "f32"])
def test_model_matches_unblocked_elimination(m, w, rows, kind):
    rng = np.random.default_rng(m * 1000 + w)
    a = rng.standard_normal((m, w))
    if kind == "ties":
        a = np.round(a * 2)
    if kind == "zero_column":
        a[:, w // 3] = 0
        a[: m // 2, w // 2] = a[0, w // 2]
    if kind == "f32":
        a = a.astype(np.float32)
    ref, piv = lu_unblocked(a)
    got, piv2 = lu_wpanel_model(a, rows)
    assert piv == piv2
    assert np.array_equal(ref, got, equal_nan=True)


@pytest.mark.parametrize("m,w,rows", [(64, 16, 8), (100, 32, 16), (33, 33, 7), (200, 64, 64), (50, 20, 50), (10, 10, 3), (129, 64, 64), (70, 5, 8),
                                      (64, 64, 16), (9, 9, 4)])
@pytest.mark.parametrize("kind", ["gauss", "ties", "zero_column", "f32"])
def test_small_leaf_model_matches_unblocked_elimination(m, w, rows, kind):
    """csrc/lu_small_leaf.h: labels, positions rotated by 8 per group of steps, per-wavefront candidate rows -- same pivots, same bits"""
    rng = np.random.default_rng(m * 1000 + w + 7)
    a = rng.standard_normal((m, w))
    if kind == "ties":
        a = np.round(a * 2)
    if kind == "zero_column":
        a[:, w // 3] = 0
        a[: m // 2, w // 2] = a[0, w // 2]
    if kind == "f32":
        a = a.astype(np.float32)
    ref, piv = lu_unblocked(a)
    got, piv2 = lu_small_leaf_model(a, rows)
    assert piv == piv2
    assert np.array_equal(ref, got, equal_nan=True)
