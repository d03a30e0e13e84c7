"""CPU check of the algebra of the one-pass tall-skinny QR (csrc/tsqr.hip) through its numpy prototype
(tests/diag/proto_tsqr.py): Gram matrix -> Cholesky -> sign-choosing LU -> Householder vectors / T blocks reproduce the
oracle's Householder QR (faer qr/no_pivoting/factor.rs:137-256, householder.rs:59-107), including the cross-panel blocks
of T computed from small matrices only.  Test infrastructure: nothing here is on the product path."""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

_spec = importlib.util.spec_from_file_location("proto_tsqr", os.path.join(ROOT, "tests", "diag", "proto_tsqr.py"))
proto = importlib.util.module_from_spec(_spec)
_argv = sys.argv
sys.argv = ["proto_tsqr", "--import-only"]  # (the module runs its demo only without arguments / with "algebraic")
try:
    _spec.loader.exec_module(proto)
finally:
    sys.argv = _argv

EPS = float(np.finfo(np.float32).eps)


def _oracle64(a):
    m, n = a.shape
    ref = a.astype(np.float64).copy(order="F")
    rh = np.zeros((n, n), order="F")
    assert oracle.qr_in_place(ref, rh) == n
    return ref, rh


@pytest.mark.parametrize("m,n", [(3000, 64), (6000, 128), (5000, 100), (8000, 256)])
def test_prototype_reproduces_the_oracle_factorization(m, n):
    rng = np.random.default_rng(m + n)
    a = rng.standard_normal((m, n)).astype(np.float32)
    got, T = proto.tsqr(a)
    ref, rh = _oracle64(a)
    up = np.triu(np.ones((m, n), bool))
    d = np.abs(got.astype(np.float64) - ref)
    assert d[up].max() <= 16 * EPS * np.abs(ref[up]).max()  # R (sign convention included)
    assert d[~up].max() <= 8 * EPS  # V
    assert np.abs(np.triu(T) - np.triu(rh)).max() <= 16 * EPS * np.abs(rh).max()


@pytest.mark.parametrize("m,n", [(6000, 192), (9000, 200)])
def test_cross_panel_t_blocks_from_small_matrices(m, n):
    rng = np.random.default_rng(7 * m + n)
    a = rng.standard_normal((m, n)).astype(np.float32)
    got, T = proto.tsqr_algebraic_t(a)
    V = np.tril(got.astype(np.float64), -1)[:, :n]
    V[np.arange(n), np.arange(n)] = 1.0
    gram = V.T @ V
    assert np.abs(np.triu(T, 1) - np.triu(gram, 1)).max() <= 16 * EPS  # T = striu(V^T V) + diag(tau)
    ref, rh = _oracle64(a)
    assert np.abs(np.triu(T) - np.triu(rh)).max() <= 16 * EPS * np.abs(rh).max()
