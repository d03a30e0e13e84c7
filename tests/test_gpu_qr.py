"""GPU parity: Householder QR through the C-ABI vs the CPU oracle and the reference's known-answer test
(test_qr, test_rank_deficient (shape), qr::tests::test_example)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from gpu_util import EPS, init_gpu, rnd, to_dev, to_host

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def q_from(F, dqr, dh, m, dtype):
    q = to_dev(np.eye(m, dtype=dtype))
    F.apply_block_householder_sequence_on_the_left_in_place(dqr, dh, q, transpose=False)
    return to_host(q)


def test_golden_qr_lstsq():
    """faer/src/linalg/qr/mod.rs:116-191: 10x2 least squares vs the numpy solution, 1e-6"""
    F = init_gpu()
    g = json.load(open(os.path.join(GOLD, "qr_lstsq_10x2.json")))
    a, b, x = np.array(g["a"]), np.array(g["b"]), np.array(g["expected_solution"])
    m, n = a.shape
    bs = F.qr_recommended_block_size(m, n)
    assert bs == 1
    dqr, dh = to_dev(a), to_dev(np.zeros((bs, n)))
    assert F.qr_factor_in_place(dqr, dh) == 2
    sol = to_dev(b)
    F.apply_block_householder_sequence_on_the_left_in_place(dqr, dh, sol, transpose=True)
    top = sol[:2, :]
    F.solve_upper_triangular_in_place(dqr[:2, :2], top)
    assert np.abs(to_host(top) - x).max() <= g["tol"]


@pytest.mark.parametrize("m,n", [(1, 1), (2, 1), (10, 2), (33, 33), (100, 40), (40, 100), (257, 64), (512, 200),
                                 (5000, 17), (3000, 256)])
@pytest.mark.parametrize("bs", [1, 4, 15, 32, None])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_qr_full_rank_vs_oracle(oracle, m, n, bs, dtype):
    """qr/no_pivoting/factor.rs:327-538 `test_qr`: Q R ~ A; plus factor-level parity with the oracle"""
    F = init_gpu()
    rng = np.random.default_rng(m * 131 + n)
    a = rnd(rng, m, n, dtype)
    size = min(m, n)
    if bs is None:
        bs = F.qr_recommended_block_size(m, n, dtype)
        assert bs == oracle.qr_recommended_block_size(m, n, dtype)
    bs = max(1, min(bs, size))
    dqr, dh = to_dev(a), to_dev(np.zeros((bs, size), dtype=dtype))
    rank = F.qr_factor_in_place(dqr, dh)
    assert rank == size
    qr, h = to_host(dqr), to_host(dh)
    ref, rh = a.copy(order="F"), np.zeros((bs, size), dtype=dtype, order="F")
    assert oracle.qr_in_place(ref, rh) == size
    e = EPS[np.dtype(dtype)]
    tol = 64 * max(m, n) * e * max(1.0, np.abs(a).max())
    R = np.triu(qr).astype(np.float64)
    q = q_from(F, dqr, dh, m, dtype).astype(np.float64)
    assert np.abs(q @ R - a).max() <= tol
    assert np.abs(q.T @ q - np.eye(m)).max() <= tol
    # same Householder vectors / R / T as the reference algorithm (unique up to rounding)
    assert np.abs(qr.astype(np.float64) - ref).max() <= 8 * tol
    fin = np.isfinite(rh)
    assert (np.isfinite(h) == fin).all()
    up = np.zeros_like(fin)
    for j0 in range(0, size, bs):
        w = min(bs, size - j0)
        up[:w, j0:j0 + w] = np.triu(np.ones((w, w), bool))
    assert np.abs(h.astype(np.float64) - rh)[fin & up].max(initial=0) <= 8 * tol * max(1.0, np.abs(rh[fin & up]).max(initial=0))


@pytest.mark.parametrize("true_rank", [1, 2, 3, 5])
@pytest.mark.parametrize("bs", [1, 15])
def test_qr_rank_deficient(oracle, true_rank, bs):
    """low rank products: rank >= true rank, Q R ~ A (factor.rs:327-538), same rank as the oracle"""
    F = init_gpu()
    rng = np.random.default_rng(11)
    m, n = 120, 60
    a = np.asfortranarray(rnd(rng, m, true_rank) @ rnd(rng, true_rank, n))
    size = min(m, n)
    dqr, dh = to_dev(a), to_dev(np.zeros((bs, size)))
    rank = F.qr_factor_in_place(dqr, dh)
    ref, rh = a.copy(order="F"), np.zeros((bs, size), order="F")
    assert rank == oracle.qr_in_place(ref, rh)
    assert true_rank <= rank < size
    h = to_host(dh)
    assert np.array_equal(np.isinf(h), np.isinf(rh)) and (h[:, rank:][np.isfinite(h[:, rank:])] == 0).all()
    q = q_from(F, dqr, dh, m, np.float64)
    assert np.abs(q @ np.triu(to_host(dqr)) - a).max() < 1e-10 * max(1.0, np.abs(a).max())


def test_qr_fp32_reference_rank_test_limit(oracle):
    """faer's rank test (factor.rs:52-58) uses threshold = eps * 16 * (m - row) * norm: for fp32 and
    m >= 524288 rows it exceeds the column norm itself, so the REFERENCE rejects every column (rank 0).
    Parity means reproducing that, not "fixing" it."""
    F = init_gpu()
    rng = np.random.default_rng(13)
    m, n = 600000, 3
    a = rnd(rng, m, n, np.float32)
    ref, rh = a.copy(order="F"), np.zeros((1, n), dtype=np.float32, order="F")
    assert oracle.qr_in_place(ref, rh) == 0
    dqr, dh = to_dev(a), to_dev(np.zeros((1, n), dtype=np.float32))
    assert F.qr_factor_in_place(dqr, dh) == 0
    h, got = to_host(dh), to_host(dqr)
    assert np.isinf(h).all() and np.isinf(rh).all()
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()


def test_qr_tall_skinny_property():
    """tall-skinny fp32 on the fast path (BASELINE config Q is 1e6 x 256; 5e5 rows is the largest power-of-ten
    style shape for which the reference itself performs a factorization, see the test above):
    R^T R ~ A^T A, T == striu(V^T V) + diag(|v|^2 / 2)"""
    import torch

    F = init_gpu()
    m, n = 500000, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn((n, m), dtype=torch.float32, device="cuda", generator=g).t()
    qr = a.clone()
    h = torch.zeros((n, n), dtype=torch.float32, device="cuda").t()
    assert F.qr_factor_in_place(qr, h) == n
    F.synchronize()
    R = torch.triu(qr[:n, :]).double()
    G = (a.double().t() @ a.double())
    assert ((R.t() @ R - G).abs().max() / G.abs().max()).item() < 5e-5
    V = torch.tril(qr, -1)
    V[:n, :n] += torch.eye(n, dtype=torch.float32, device="cuda")
    VtV = V.double().t() @ V.double()
    T = h.double()
    assert ((torch.diagonal(VtV) * 0.5 - torch.diagonal(T)).abs().max() / torch.diagonal(T).abs().max()).item() < 1e-4
    assert ((torch.triu(VtV, 1) - torch.triu(T, 1)).abs().max()).item() < 1e-2


def thin_q(F, dqr, dh, m, n):
    """the first n columns of Q (fp64 on the host)"""
    q = to_dev(np.eye(m, n, dtype=np.float32))
    F.apply_block_householder_sequence_on_the_left_in_place(dqr, dh, q, transpose=False)
    return to_host(q).astype(np.float64)


def _tall_vs_oracle(oracle, F, a, bs, lead=None, tol=(8.0, 2.0, 8.0)):
    """one-pass path (csrc/tsqr.hip) against the oracle's Householder QR: every column of R within tol[0] eps of that column's
    largest entry, V within tol[1] eps (its entries are O(1 / sqrt(m))), T within tol[2] eps max|T|.  Measured on well
    conditioned panels: 2-4 / 0.1-0.6 / 2-4 eps (tests/diag/gpu_qr_tall_fuzz.py), hence 8 / 2 / 8;
    conditioned panels pass cond-scaled tolerances.  The generic bound 64 max(m, n) eps would be ~1 here and say nothing"""
    import torch

    m, n = a.shape
    if lead is None:
        dqr = to_dev(a)
    else:  # a view with a leading dimension that is not a multiple of 4 (scalar loads in every kernel)
        buf = torch.zeros((n, lead), dtype=torch.float32, device="cuda")
        buf[:, :m] = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
        dqr = buf.t()[:m, :]
    dh = to_dev(np.zeros((bs, n), dtype=np.float32))
    assert F.qr_factor_in_place(dqr, dh) == n
    F.lib().faer_hip_debug_qr_one_pass_columns.restype = C.c_long
    assert F.lib().faer_hip_debug_qr_one_pass_columns() == n  # the whole factorization ran on the one-pass path
    qr, h = to_host(dqr), to_host(dh)
    ref, rh = a.copy(order="F"), np.zeros((bs, n), dtype=np.float32, order="F")
    assert oracle.qr_in_place(ref, rh) == n
    e = float(np.finfo(np.float32).eps)
    up = np.triu(np.ones((m, n), bool))
    d = np.abs(qr.astype(np.float64) - ref)
    dr = np.where(up, d, 0.0).max(axis=0) / np.where(up, np.abs(ref), 0.0).max(axis=0)  # per column of R
    assert dr.max() <= tol[0] * e, ("R", dr.max() / e)
    assert d[~up].max() <= tol[1] * e, ("V", d[~up].max() / e)
    tu = np.zeros((bs, n), bool)
    for j0 in range(0, n, bs):
        w = min(bs, n - j0)
        tu[:w, j0:j0 + w] = np.triu(np.ones((w, w), bool))
    assert np.isfinite(h).all()
    dt = np.abs(h.astype(np.float64) - rh)[tu].max()
    assert dt <= tol[2] * e * np.abs(rh[tu]).max(), ("T", dt / e / np.abs(rh[tu]).max())
    return dqr, dh, qr, h


def _q_properties(F, dqr, dh, a, c=16.0):
    """the condition-independent properties of a Householder QR: |Q^T Q - I| and |Q R - A| at c sqrt(m) eps"""
    m, n = a.shape
    e = float(np.finfo(np.float32).eps)
    q = thin_q(F, dqr, dh, m, n)
    R = np.triu(to_host(dqr)[:n]).astype(np.float64)
    assert np.abs(q.T @ q - np.eye(n)).max() <= c * np.sqrt(m) * e, ("QtQ", np.abs(q.T @ q - np.eye(n)).max() / (np.sqrt(m) * e))
    res = np.abs(q @ R - a).max(axis=0) / np.abs(a).max(axis=0)  # per column: scaled columns must not hide behind the largest one
    assert res.max() <= c * np.sqrt(m) * e, ("QR-A", res.max() / (np.sqrt(m) * e))


def _conditioned(rng, m, n, cond):
    """m x n fp32 matrix, entries O(1), whose 64-column panels each have singular values graded from 1 to 1 / cond (independent
    random subspaces: a panel keeps that conditioning after the earlier panels' reflectors have been applied)"""
    blocks = []
    for c0 in range(0, n, 64):
        w = min(64, n - c0)
        q1, _ = np.linalg.qr(rng.standard_normal((m, w)))
        q2, _ = np.linalg.qr(rng.standard_normal((w, w)))
        blocks.append((q1 * np.logspace(0.0, -np.log10(cond), w)) @ q2.T * np.sqrt(m))
    return np.asfortranarray(np.hstack(blocks).astype(np.float32))


@pytest.mark.parametrize("cond", [10.0, 30.0, 100.0, 400.0])
@pytest.mark.parametrize("m,n,bs", [(20000, 64, 64), (30000, 128, 128)])
def test_qr_tall_one_pass_conditioned_panels(oracle, m, n, bs, cond):
    """accepted panels between the Gaussian case (cond ~ 1.2) and the guard (cond_2 ~ 512, tsqr.hip TQ_COND_MAX): V = P M and
    the trailing rows of R amplify fp32 rounding by cond(panel), so the factors are compared at cond-scaled tolerances
    (householder.rs:59-107, factor.rs:52-64 semantics unchanged) AND through the properties that do not depend on it.
    What limits the conditioning of an accepted tall fp32 panel in practice is the REFERENCE's rank test, not the guard: a
    column is rejected once |R_jj| <= 16 eps (m - j) |column| (factor.rs:52-58), i.e. below 3.8 % of its norm at 20000 rows,
    so the cond = 400 cases come back rank deficient from the oracle -- there the GPU must return the oracle's rank (first
    panel refused, classic path) and the same pattern of skipped reflectors."""
    F = init_gpu()
    rng = np.random.default_rng(int(m + n + cond))
    a = _conditioned(rng, m, n, cond)
    first = np.linalg.cond(a[:, :64].astype(np.float64))
    assert cond * 0.99 <= first <= cond * 1.01  # the case is what it claims to be
    ref, rh = a.copy(order="F"), np.zeros((bs, n), dtype=np.float32, order="F")
    rk = oracle.qr_in_place(ref, rh)
    if rk == n:
        dqr, dh, _, _ = _tall_vs_oracle(oracle, F, a, bs, tol=(16.0 * cond, 4.0 * cond, 16.0 * cond))
        _q_properties(F, dqr, dh, a)
        return
    assert cond >= 100.0  # only the worst cases may be rank deficient by the reference's test
    dqr, dh = to_dev(a), to_dev(np.zeros((bs, n), dtype=np.float32))
    assert F.qr_factor_in_place(dqr, dh) == rk
    assert np.array_equal(np.isinf(to_host(dh)), np.isinf(rh))


@pytest.mark.parametrize("decades", [3, 6])
def test_qr_tall_one_pass_badly_scaled_columns(oracle, decades):
    """well-conditioned directions, column scales spread over 2 x `decades` decades: the guard looks at the EQUILIBRATED panel
    (a Cholesky factorization of D G D is as accurate as that of G), so these stay on the one-pass path at the Gaussian
    tolerances -- R compared column by column"""
    F = init_gpu()
    rng = np.random.default_rng(decades)
    m, n = 24000, 128
    a = rnd(rng, m, n, np.float32)
    a *= (10.0 ** rng.uniform(-decades, decades, n)).astype(np.float32)[None, :]
    dqr, dh, _, _ = _tall_vs_oracle(oracle, F, np.asfortranarray(a), 64)
    _q_properties(F, dqr, dh, a)


def test_qr_tall_range_guard_covers_every_column(oracle):
    """256 < n <= 512: a column beyond the first 256 whose scale is outside the fp32-safe range [1e-12, 1e12] must stop the
    one-pass path BEFORE anything is written (tq_range_rest_kernel); the classic path then factors the matrix"""
    F = init_gpu()
    rng = np.random.default_rng(9)
    m, n = 20000, 320
    a = rnd(rng, m, n, np.float32)
    a[:, 300] *= np.float32(1e-20)
    ref, rh = a.copy(order="F"), np.zeros((64, n), dtype=np.float32, order="F")
    rk = oracle.qr_in_place(ref, rh)
    dqr, dh = to_dev(a), to_dev(np.zeros((64, n), dtype=np.float32))
    assert F.qr_factor_in_place(dqr, dh) == rk
    F.lib().faer_hip_debug_qr_one_pass_columns.restype = C.c_long
    assert F.lib().faer_hip_debug_qr_one_pass_columns() == 0
    e = float(np.finfo(np.float32).eps)
    d = np.abs(to_host(dqr).astype(np.float64) - ref)
    scale = np.maximum(np.abs(ref).max(axis=0), 1e-30)
    assert (d.max(axis=0) / scale).max() <= 64 * np.sqrt(m) * e


@pytest.mark.parametrize("m,n,bs", [(20000, 64, 64), (20000, 64, 32), (16384, 130, 1), (40000, 100, None), (30000, 200, 16),
                                    (65536, 256, None), (50000, 256, 256), (24000, 192, 192), (200000, 64, None), (16500, 17, 1)])
def test_qr_tall_one_pass_vs_oracle(oracle, m, n, bs):
    """qr/no_pivoting/factor.rs:137-256 on the one-pass path: whole and ragged panels, blocks of Q_coeff narrower than,
    equal to and wider than a 64-column panel (the wider ones need the cross-panel blocks of T)"""
    F = init_gpu()
    rng = np.random.default_rng(m + n)
    a = rnd(rng, m, n, np.float32)
    if bs is None:
        bs = F.qr_recommended_block_size(m, n, np.float32)
        assert bs == oracle.qr_recommended_block_size(m, n, np.float32)
    _tall_vs_oracle(oracle, F, a, bs)


@pytest.mark.parametrize("m,n,bs", [(40000, 512, 512), (33000, 384, 128), (50000, 320, None), (36000, 448, 64), (20000, 512, 256),
                                    (30001, 300, 64)])
def test_qr_tall_one_pass_wide_panels_vs_oracle(oracle, m, n, bs):
    """256 < n <= 512 on the one-pass path: trailing matrices wider than one 192-column strip (several Gram / update launches
    per panel), the look-ahead of the panel kernel (on by itself from 192 more trailing columns), blocks of Q_coeff wider than
    256 columns (the general kernel of the cross-panel T blocks) and several blocks of Q_coeff of 2 - 4 panels each"""
    F = init_gpu()
    rng = np.random.default_rng(m + n)
    a = rnd(rng, m, n, np.float32)
    if bs is None:
        bs = F.qr_recommended_block_size(m, n, np.float32)
        assert bs == oracle.qr_recommended_block_size(m, n, np.float32)
    _tall_vs_oracle(oracle, F, a, bs)


def test_qr_tall_one_pass_unaligned_leading_dimension(oracle):
    F = init_gpu()
    rng = np.random.default_rng(77)
    a = rnd(rng, 20001, 70, np.float32)
    _tall_vs_oracle(oracle, F, a, 64, lead=20003)


@pytest.mark.parametrize("m", [500000])
def test_qr_tall_config_q_vs_oracle(oracle, m):
    """BASELINE config Q at the height the reference really factors (see test_qr_fp32_reference_rank_test_limit): the whole
    5e5 x 256 factorization against the oracle (about a minute of CPU)"""
    F = init_gpu()
    rng = np.random.default_rng(m)
    a = rnd(rng, m, 256, np.float32)
    _tall_vs_oracle(oracle, F, a, 256)


def test_qr_tall_falls_back_per_panel(oracle):
    """panels the one-pass path must refuse (tsqr.hip): an ill-conditioned panel in the middle (two nearly equal columns),
    a column that is zero below the diagonal (the reference's tau = +inf) and a rank-deficient matrix; the classic path
    takes over from that panel and the result is the oracle's"""
    F = init_gpu()
    rng = np.random.default_rng(5)
    m, n = 20000, 192
    e = float(np.finfo(np.float32).eps)
    # (1) the second panel is ill conditioned (not merely badly scaled) although every column keeps 75 % of its norm below the
    #     diagonal, far above the reference's rank threshold 16 eps m = 3.8 %: orthonormal columns times a Kahan-like upper
    #     triangle (unit columns, diagonal 0.75, equal entries above it) -- cond ~ 1e5 against the guard's ~512
    a = rnd(rng, m, n, np.float32)
    W = np.zeros((64, 64))
    W[0, 0] = 1.0
    for j in range(1, 64):
        W[:j, j] = -np.sqrt((1 - 0.75 ** 2) / j)
        W[j, j] = 0.75
    assert np.linalg.cond(W) > 2e4
    q2, _ = np.linalg.qr(rnd(rng, m, 64, np.float64))
    a[:, 64:128] = (q2 @ W * np.sqrt(m)).astype(np.float32)
    ref, rh = a.copy(order="F"), np.zeros((64, n), dtype=np.float32, order="F")
    assert oracle.qr_in_place(ref, rh) == n
    dqr, dh = to_dev(a), to_dev(np.zeros((64, n), dtype=np.float32))
    assert F.qr_factor_in_place(dqr, dh) == n
    F.lib().faer_hip_debug_qr_one_pass_columns.restype = C.c_long
    assert F.lib().faer_hip_debug_qr_one_pass_columns() == 64  # the first panel on the one-pass path, the ill-conditioned one refused
    q = thin_q(F, dqr, dh, m, n)
    R = np.triu(to_host(dqr)[:n]).astype(np.float64)
    assert np.abs(q @ R - a).max() <= 64 * np.sqrt(m) * e * np.abs(a).max()
    assert np.abs(q.T @ q - np.eye(n)).max() <= 64 * np.sqrt(m) * e
    # (1b) the same panel graded over four decades but well conditioned after equilibration: stays on the one-pass path
    a = rnd(rng, m, n, np.float32)
    a[:, 64:128] *= np.logspace(0, -4, 64, dtype=np.float32)[None, :]
    dq2, dh2, _, _ = _tall_vs_oracle(oracle, F, np.asfortranarray(a), 64)
    _q_properties(F, dq2, dh2, a)
    # (2) the first panel lives in the top 64 rows only (its reflectors leave the rows below alone), the second one is
    #     upper triangular from row 64 down: every tail of the second panel is exactly zero
    b = np.zeros((m, 128), dtype=np.float32, order="F")
    b[:64, :64] = 10 * np.eye(64, dtype=np.float32) + 0.1 * rnd(rng, 64, 64, np.float32)
    b[:64, 64:] = rnd(rng, 64, 64, np.float32)
    b[64:128, 64:] = np.triu(rnd(rng, 64, 64, np.float32)) + 3 * np.eye(64, dtype=np.float32)
    ref, rh = b.copy(order="F"), np.zeros((64, 128), dtype=np.float32, order="F")
    rk = oracle.qr_in_place(ref, rh)
    dqr, dh = to_dev(b), to_dev(np.zeros((64, 128), dtype=np.float32))
    assert F.qr_factor_in_place(dqr, dh) == rk
    assert 0 <= F.lib().faer_hip_debug_qr_one_pass_columns() < 128
    h = to_host(dh)
    assert np.array_equal(np.isinf(h), np.isinf(rh))
    assert np.abs(to_host(dqr) - ref).max() <= 64 * np.sqrt(m) * e * np.abs(ref).max()
    # (3) rank 100 of 192 columns
    c = (rnd(rng, m, 100, np.float64) @ rnd(rng, 100, n, np.float64)).astype(np.float32)
    ref, rh = c.copy(order="F"), np.zeros((64, n), dtype=np.float32, order="F")
    rk = oracle.qr_in_place(ref, rh)
    dqr, dh = to_dev(c), to_dev(np.zeros((64, n), dtype=np.float32))
    got = F.qr_factor_in_place(dqr, dh)
    assert got == rk  # the rank test sees the whole column above the diagonal (geqrf_classic `off`)
    q = thin_q(F, dqr, dh, m, n)
    assert np.abs(q @ np.triu(to_host(dqr)[:n]).astype(np.float64) - c).max() <= 256 * np.sqrt(m) * e * np.abs(c).max()


def test_qr_1e6_rows_fp32_matches_reference_semantics():
    """BASELINE config Q literally (1e6 x 256 fp32): the reference's rank test yields rank 0"""
    import torch

    F = init_gpu()
    m, n = 1000000, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn((n, m), dtype=torch.float32, device="cuda", generator=g).t()
    h = torch.zeros((n, n), dtype=torch.float32, device="cuda").t()
    assert F.qr_factor_in_place(a, h) == 0
    assert torch.isinf(torch.diagonal(h)).all().item()


@pytest.mark.parametrize("m,n,k", [(10, 2, 1), (100, 50, 3), (300, 300, 7), (2000, 130, 40)])
def test_qr_solve_lstsq_and_square(m, n, k):
    """qr/no_pivoting/solve.rs: least squares (reference test_lstsq: 100 x 50, k = 3) and, for square systems,
    A x = b and A^T x = b"""
    F = init_gpu()
    rng = np.random.default_rng(m + n)
    a = rnd(rng, m, n)
    b = rnd(rng, m, k)
    q = F.Qr(to_dev(a))
    x = to_dev(b)
    q.solve_lstsq_in_place(x)
    got = to_host(x)[:n]
    ref = np.linalg.lstsq(a, b, rcond=None)[0]
    assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    if m == n:
        y = to_dev(b)
        F.qr_solve_in_place(q.Q_basis(), q.Q_coeff(), y)
        assert np.abs(to_host(y) - np.linalg.solve(a, b)).max() <= 1e-8 * max(1.0, np.abs(ref).max())
        z = to_dev(b)
        F.qr_solve_in_place(q.Q_basis(), q.Q_coeff(), z, transpose=True)
        reft = np.linalg.solve(a.T, b)
        assert np.abs(to_host(z) - reft).max() <= 1e-8 * max(1.0, np.abs(reft).max())


def test_qr_is_bitwise_reproducible():
    """deep-K products inside QR are split over workgroups; their partial sums are combined in a fixed order
    (gemm.hip splitk_reduce_kernel), so two runs agree bit for bit -- like the reference with a fixed thread count"""
    import torch

    F = init_gpu()
    m, n = 200000, 64
    g = torch.Generator(device="cuda").manual_seed(23)
    a = torch.randn((n, m), dtype=torch.float32, device="cuda", generator=g).t()
    outs = []
    for _ in range(2):
        qr = a.clone()
        h = torch.zeros((n, n), dtype=torch.float32, device="cuda").t()
        assert F.qr_factor_in_place(qr, h) == n
        F.synchronize()
        outs.append((qr, h))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


# ------------------------------------------------------------------------------------ QR with column pivoting
@pytest.mark.parametrize("m,n,layout", [(1, 1, "F"), (5, 5, "F"), (40, 30, "F"), (30, 40, "F"), (64, 64, "C"), (200, 50, "F"), (1, 7, "F"), (7, 1, "F"),
                                        (300, 300, "F"), (1500, 260, "F"), (700, 900, "F"), (257, 257, "C"),
                                        (4100, 40, "F"), (5000, 33, "C")])  # (beyond 4096 rows: the memory-resident step body)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_colpiv_qr_vs_oracle(oracle, m, n, layout, dtype):
    """colpiv_qr_dev through the C-ABI: IDENTICAL column permutation (index work), R / reflectors / T blocks within
    tolerance of the oracle, least-squares and square solves"""
    F = init_gpu()
    rng = np.random.default_rng(m * n)
    a = np.array(rng.standard_normal((m, n)) * np.logspace(0, -3, n)[None, :], dtype=dtype, order=layout)
    size = min(m, n)
    bs = oracle.qr_recommended_block_size(m, n, dtype)
    ref = a.copy(order=layout)
    href = np.zeros((bs, size), dtype=dtype, order="F")
    cp, cpi, nt = oracle.colpiv_qr_in_place(ref, href)
    da = to_dev(a, layout)
    dh = to_dev(np.zeros((bs, size), dtype=dtype, order="F"))
    cf, cb, cnt = F.colpiv_qr_factor_in_place(da, dh)
    assert np.array_equal(cf.astype(np.int64), cp) and np.array_equal(cb.astype(np.int64), cpi) and cnt == nt
    e = EPS[np.dtype(dtype)]
    tol = 256 * max(m, n) * e * max(1.0, np.abs(a).max())
    assert np.abs(to_host(da) - ref).max() <= tol
    hg = to_host(dh)
    fin = np.isfinite(href)
    assert np.array_equal(np.isfinite(hg), fin) and np.array_equal(hg[~fin], href[~fin])  # tau = +inf for an empty tail
    assert np.abs(hg[fin] - href[fin]).max(initial=0) <= tol * 4
    if m >= n and n > 1:
        b = np.array(rng.standard_normal((m, 3)), dtype=dtype, order="F")
        x = to_dev(b)
        F.colpiv_qr_solve_in_place(da, dh, cf, cb, x, mode="lstsq")
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        sol = np.linalg.lstsq(a64, b64, rcond=None)[0]
        assert np.abs(to_host(x)[:n] - sol).max() <= 4096 * max(m, n) * e * np.linalg.cond(a64) * max(1.0, np.abs(sol).max())


def test_colpiv_qr_rank_revealing_property():
    """a numerically rank-deficient matrix (rank 37 of 120 columns, N = 900 rows): |r_kk| collapses after the rank,
    A P == Q R, independent of the oracle"""
    import torch

    F = init_gpu()
    m, n, rk = 900, 120, 37
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((m, rk)) @ rng.standard_normal((rk, n))).astype(np.float64)
    da = to_dev(a)
    bs = F.qr_recommended_block_size(m, n, np.float64)
    dh = to_dev(np.zeros((bs, n), order="F"))
    cf, cb, _ = F.colpiv_qr_factor_in_place(da, dh)
    d = np.abs(np.diag(to_host(da)[:n, :n]))
    assert d[rk - 1] > 1e-3 * d[0] and d[rk:].max() < 1e-10 * d[0]
    out = to_dev(np.full((m, n), np.nan, order="F"))
    F.qr_reconstruct(out, da, dh)
    assert np.abs(to_host(out) - a[:, cf.astype(int)]).max() <= 256 * m * 2.3e-16 * np.abs(a).max()


@pytest.mark.parametrize("m,n", [(9, 10), (1023, 5), (42, 1), (3000, 40)])
@pytest.mark.parametrize("factor", [1e30, 1e100, 1e160, 1e250, 1e-30, 1e-100, 1e-160, 1e-250])
def test_qr_norm_l2_scaling_cases(oracle, m, n, factor):
    """reductions/norm_l2.rs:174-197 (`test_norm_l2` factors) THROUGH qr_factor_in_place: the column norms of an
    fp64 matrix scaled by 1e+-250 must neither overflow nor underflow (the reference's norm_l2 keeps three scaled
    accumulators, norm_l2.rs:6-45,173-184).  Same rank, R, V and T as the oracle, relative to the scale."""
    F = init_gpu()
    rng = np.random.default_rng(m + n)
    a = np.asfortranarray(rnd(rng, m, n) * factor)
    size = min(m, n)
    bs = max(1, min(F.qr_recommended_block_size(m, n), size))
    dqr, dh = to_dev(a), to_dev(np.zeros((bs, size)))
    rank = F.qr_factor_in_place(dqr, dh)
    qr, h = to_host(dqr), to_host(dh)
    ref, rh = a.copy(order="F"), np.zeros((bs, size), order="F")
    assert oracle.qr_in_place(ref, rh) == size and rank == size
    assert np.isfinite(qr).all()
    e = EPS[np.dtype(np.float64)]
    up = np.triu(np.ones((m, n), bool))
    # R carries the scale, V (below the diagonal) and T do not
    assert np.abs(qr - ref)[up].max() <= 512 * max(m, n) * e * factor * np.abs(ref / factor)[up].max()
    assert np.abs(qr - ref)[~up].max(initial=0) <= 512 * max(m, n) * e
    tu = np.zeros((bs, size), bool)
    for j0 in range(0, size, bs):
        w = min(bs, size - j0)
        tu[:w, j0:j0 + w] = np.triu(np.ones((w, w), bool))
    fin = np.isfinite(rh)
    assert (np.isfinite(h) == fin).all()
    assert np.abs(h - rh)[fin & tu].max(initial=0) <= 512 * max(m, n) * e * max(1.0, np.abs(rh[fin & tu]).max(initial=0))


@pytest.mark.parametrize("mode", [0, 2, 3])
@pytest.mark.parametrize("m,n,bs", [(40000, 256, 256), (30001, 300, 64), (33000, 384, 128), (20000, 70, 64)])
def test_qr_tall_other_schedules_still_agree_with_the_oracle(oracle, m, n, bs, mode):
    """faer_hip_debug_qr_fused(0): the one-pass path with update and Gram products as separate launches (the schedule of rounds 3-5, kept
    for A/B measurements); (2): the fused schedule without the raw copy of the panel (what matrices of more than 4.19 M rows run); (3): the
    plain schedule on the streaming kernels written for fp64 data at the end of round 6, instantiated for fp32 (columns that are not 16-byte
    aligned -- 30001 rows -- stay on the default schedule).  All must keep producing the reference's R / V / T -- the default schedule is
    what every other test runs."""
    F = init_gpu()
    rng = np.random.default_rng(m + n)
    a = np.asfortranarray(rng.standard_normal((m, n)).astype(np.float32))
    F.lib().faer_hip_debug_qr_fused(mode)
    try:
        _tall_vs_oracle(oracle, F, a, bs)
    finally:
        F.lib().faer_hip_debug_qr_fused(1)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m,n,bs", [(2048, 2048, None), (3000, 1500, 64), (2500, 700, 128), (4096, 1024, 32), (2100, 320, 192)])
def test_qr_classic_path_one_pass_panels_vs_oracle(oracle, m, n, bs, dtype):
    """square / moderately tall matrices: the recursion of the classic path (qr.hip qr_rec) hands panels of up to 64 columns with at
    least 1024 rows to the one-pass panel of tsqr.hip (end of round 6); same factors as the oracle at the classic path's tolerance,
    with the switch off (recursion down to the 8-column leaves) as well, and the two agree"""
    F = init_gpu()
    rng = np.random.default_rng(m + 3 * n)
    a = rnd(rng, m, n, dtype)
    size = min(m, n)
    if bs is None:
        bs = F.qr_recommended_block_size(m, n, dtype)
    ref, rh = a.copy(order="F"), np.zeros((bs, size), dtype=dtype, order="F")
    assert oracle.qr_in_place(ref, rh) == size
    e = EPS[np.dtype(dtype)]
    tol = 64 * max(m, n) * e * max(1.0, np.abs(a).max())
    up = np.zeros((bs, size), bool)
    for j0 in range(0, size, bs):
        w = min(bs, size - j0)
        up[:w, j0:j0 + w] = np.triu(np.ones((w, w), bool))
    out = []
    for on in (1, 0):
        F.lib().faer_hip_debug_qr_panels_one_pass(on)
        try:
            dqr, dh = to_dev(a), to_dev(np.zeros((bs, size), dtype=dtype))
            assert F.qr_factor_in_place(dqr, dh) == size
        finally:
            F.lib().faer_hip_debug_qr_panels_one_pass(1)
        qr, h = to_host(dqr).astype(np.float64), to_host(dh).astype(np.float64)
        assert np.abs(qr - ref).max() <= 8 * tol
        fin = np.isfinite(rh)  # (the last reflector of a square matrix has tau = +inf)
        assert (np.isfinite(h) == fin).all()
        assert np.abs(h - np.where(fin, rh, 0.0))[fin & up].max() <= 8 * tol * max(1.0, np.abs(rh[fin & up]).max())
        out.append((qr, h))
    assert np.abs(out[0][0] - out[1][0]).max() <= 8 * tol


@pytest.mark.parametrize("bs", [64, 128])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_qr_classic_path_one_pass_panels_rank_deficient(oracle, dtype, bs):
    """a rank-deficient matrix whose dependent columns sit in a LATER panel: the one-pass panel must refuse it (rank test with the rows
    above the panel counted, factor.rs:52-58) and the result is the oracle's rank and pattern of skipped reflectors"""
    F = init_gpu()
    rng = np.random.default_rng(21)
    m, n, r = 2048, 512, 200
    a = np.asfortranarray((rnd(rng, m, r) @ rnd(rng, r, n)).astype(dtype))
    # (blocks of 128: the node of two panels whose SECOND panel holds the first dependent column -- the one-pass path completes the first
    # panel, applies it, stops; the recursion takes the second panel and T12)
    ref, rh = a.copy(order="F"), np.zeros((bs, n), dtype=dtype, order="F")
    rk = oracle.qr_in_place(ref, rh)
    dqr, dh = to_dev(a), to_dev(np.zeros((bs, n), dtype=dtype))
    assert F.qr_factor_in_place(dqr, dh) == rk
    assert r <= rk < n
    assert np.array_equal(np.isinf(to_host(dh)), np.isinf(rh))
    e = EPS[np.dtype(dtype)]
    q = q_from(F, dqr, dh, m, dtype).astype(np.float64)
    assert np.abs(q @ np.triu(to_host(dqr)).astype(np.float64) - a).max() <= 256 * np.sqrt(m) * e * np.abs(a).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m,n,bs,cols", [(1024, 256, 32, 256), (1100, 100, 32, 100), (2048, 256, 48, 256), (3000, 512, 64, 512), (1536, 512, 48, 512),
                                         (1024, 340, 40, 340), (4096, 256, 48, -1), (1023, 128, 64, -1), (1024, 342, 64, -1)])
def test_qr_moderately_tall_one_pass_shape_rule(oracle, m, n, bs, cols, dtype):
    """the whole-matrix one-pass path from 1024 rows and 3 rows per column (end of round 6; rounds 3-6: 16384 and 8), with blocks of Q_coeff
    that fit its 64-column panels directly and blocks that do not (T rebuilt from V and the taus, up to 3072 rows); shapes just outside the
    rule run the classic path.  Factors against the oracle at the classic path's tolerance."""
    F = init_gpu()
    rng = np.random.default_rng(m + 7 * n)
    a = rnd(rng, m, n, dtype)
    ref, rh = a.copy(order="F"), np.zeros((bs, n), dtype=dtype, order="F")
    assert oracle.qr_in_place(ref, rh) == n
    dqr, dh = to_dev(a), to_dev(np.zeros((bs, n), dtype=dtype))
    assert F.qr_factor_in_place(dqr, dh) == n
    F.lib().faer_hip_debug_qr_one_pass_columns.restype = C.c_long
    assert F.lib().faer_hip_debug_qr_one_pass_columns() == cols
    e = EPS[np.dtype(dtype)]
    tol = 64 * max(m, n) * e * max(1.0, np.abs(a).max())
    qr, h = to_host(dqr).astype(np.float64), to_host(dh).astype(np.float64)
    assert np.abs(qr - ref).max() <= 8 * tol
    up = np.zeros((bs, n), bool)
    for j0 in range(0, n, bs):
        w = min(bs, n - j0)
        up[:w, j0:j0 + w] = np.triu(np.ones((w, w), bool))
    assert np.isfinite(h).all()
    assert np.abs(h - rh)[up].max() <= 8 * tol * max(1.0, np.abs(rh[up]).max())
    q = q_from(F, dqr, dh, m, dtype).astype(np.float64)
    assert np.abs(q @ np.triu(qr) - a).max() <= tol
    assert np.abs(q.T @ q - np.eye(m)).max() <= tol


def test_qr_dispatch_fuzz_vs_oracle():
    """tools/gpu_qr_fuzz.py: random shapes, block sizes of Q_coeff, dtypes and views (row offsets, padded and odd column strides, a
    dependent column now and then) over the QR dispatch -- whole-matrix one-pass path, one-pass panels inside the classic recursion,
    rebuilt T blocks -- against the oracle; nothing outside a view may change"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_qr_fuzz.py"), "80", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "80 cases, 0 bad" in r.stdout
