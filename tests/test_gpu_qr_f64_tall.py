"""GPU parity: the one-pass tall-skinny QR path for fp64 data (csrc/tsqr.hip, tsqr_factor64; qr/no_pivoting/factor.rs:137-256,
householder.rs:59-107) through the C-ABI against the CPU oracle.  The path forms its Gram sums in fp64, so it keeps
well-conditioned panels only (condition guard: cond_2 below ~8) and hands everything else to the classic path panel by panel."""
import ctypes as C

import numpy as np
import pytest

from gpu_util import init_gpu, rnd, to_dev, to_host

pytestmark = pytest.mark.gpu
E = float(np.finfo(np.float64).eps)


def _one_pass_columns(F):
    F.lib().faer_hip_debug_qr_one_pass_columns.restype = C.c_long
    return F.lib().faer_hip_debug_qr_one_pass_columns()


def _factor(F, a, bs, lead=None):
    import torch

    m, n = a.shape
    if lead is None:
        dqr = to_dev(a)
    else:
        buf = torch.zeros((n, lead), dtype=torch.float64, device="cuda")
        buf[:, :m] = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
        dqr = buf.t()[:m, :]
    dh = to_dev(np.zeros((bs, n)))
    rank = F.qr_factor_in_place(dqr, dh)
    return dqr, dh, rank, _one_pass_columns(F)


def _errors(qr, h, ref, rh, bs):
    """R per column relative to that column's largest entry, V absolute (entries O(1 / sqrt(m))), T relative to max|T|: in eps"""
    m, n = ref.shape
    up = np.triu(np.ones((m, n), bool))
    d = np.abs(qr - ref)
    dr = (np.where(up, d, 0.0).max(axis=0) / np.where(up, np.abs(ref), 0.0).max(axis=0)).max()
    dv = d[~up].max()
    tu = np.zeros((bs, n), bool)
    for j0 in range(0, n, bs):
        w = min(bs, n - j0)
        tu[:w, j0:j0 + w] = np.triu(np.ones((w, w), bool))
    assert np.isfinite(h).all()
    dt = np.abs(h - rh)[tu].max() / np.abs(rh[tu]).max()
    return dr / E, dv / E, dt / E


def _vs_oracle(oracle, F, a, bs, lead=None, tol=(16.0, 2.0, 16.0), expect_cols=None):
    """measured on Gaussian panels (tools/gpu_qr_f64_onepass.py): R 2-3, V 0.1-0.5, T 2-4 eps on BOTH paths, hence 16 / 2 / 16"""
    m, n = a.shape
    dqr, dh, rank, cols = _factor(F, a, bs, lead)
    assert rank == n
    assert cols == (n if expect_cols is None else expect_cols), cols
    ref, rh = a.copy(order="F"), np.zeros((bs, n), order="F")
    assert oracle.qr_in_place(ref, rh) == n
    dr, dv, dt = _errors(to_host(dqr), to_host(dh), ref, rh, bs)
    assert dr <= tol[0], ("R", dr)
    assert dv <= tol[1], ("V", dv)
    assert dt <= tol[2], ("T", dt)
    return dqr, dh


def _q_properties(F, dqr, dh, a, c=16.0):
    """|Q^T Q - I| and |Q R - A| (per column) at c sqrt(m) eps: what a caller relies on whatever the panel's conditioning"""
    m, n = a.shape
    q = to_dev(np.eye(m, n))
    F.apply_block_householder_sequence_on_the_left_in_place(dqr, dh, q, transpose=False)
    q = to_host(q)
    R = np.triu(to_host(dqr)[:n])
    assert np.abs(q.T @ q - np.eye(n)).max() <= c * np.sqrt(m) * E, ("QtQ", np.abs(q.T @ q - np.eye(n)).max() / (np.sqrt(m) * E))
    res = np.abs(q @ R - a).max(axis=0) / np.abs(a).max(axis=0)
    assert res.max() <= c * np.sqrt(m) * E, ("QR-A", res.max() / (np.sqrt(m) * E))


def _conditioned(rng, m, n, cond):
    """m x n matrix, entries O(1), whose 64-column panels each have singular values graded from 1 to 1 / cond"""
    blocks = []
    for c0 in range(0, n, 64):
        w = min(64, n - c0)
        q1, _ = np.linalg.qr(rng.standard_normal((m, w)))
        q2, _ = np.linalg.qr(rng.standard_normal((w, w)))
        blocks.append((q1 * np.logspace(0.0, -np.log10(cond), w)) @ q2.T * np.sqrt(m))
    return np.asfortranarray(np.hstack(blocks))


@pytest.mark.parametrize("m,n,bs", [(20000, 64, 64), (20000, 64, 32), (16384, 130, 1), (40000, 100, None), (30000, 200, 16),
                                    (65536, 256, None), (50000, 256, 256), (24000, 192, 192), (200000, 64, None), (16500, 17, 1),
                                    (40000, 512, 512), (33000, 384, 128), (50000, 320, None), (36000, 448, 64), (20000, 512, 256)])
def test_qr_f64_tall_one_pass_vs_oracle(oracle, m, n, bs):
    """whole and ragged panels (odd first row below the last panel: the scalar variant of the update kernel), blocks of Q_coeff
    narrower than, equal to and wider than a panel (cross-panel blocks of T: both kernels), more than one strip of trailing columns"""
    F = init_gpu()
    rng = np.random.default_rng(m + n)
    a = rnd(rng, m, n)
    if bs is None:
        bs = F.qr_recommended_block_size(m, n, np.float64)
        assert bs == oracle.qr_recommended_block_size(m, n, np.float64)
    dqr, dh = _vs_oracle(oracle, F, a, bs)
    if n <= 200:
        _q_properties(F, dqr, dh, a)


def test_qr_f64_tall_padded_and_odd_leading_dimension(oracle):
    """faer's Mat layout (column stride padded to 64 bytes) and an odd column stride (8-byte aligned columns only: the scalar-access
    variants of the streaming kernels) both stay on the path -- same answer"""
    F = init_gpu()
    rng = np.random.default_rng(78)
    a = rnd(rng, 20001, 70)
    _vs_oracle(oracle, F, a, 64, lead=20008)
    _vs_oracle(oracle, F, a, 64, lead=20003)
    b = rnd(rng, 30000, 256)
    _vs_oracle(oracle, F, b, 256, lead=30001)


def test_qr_f64_tall_debug_switch_restores_classic_path(oracle):
    F = init_gpu()
    rng = np.random.default_rng(79)
    a = rnd(rng, 20000, 128)
    F.lib().faer_hip_debug_qr_one_pass_f64(0)
    try:
        _vs_oracle(oracle, F, a, 64, expect_cols=-1)
    finally:
        F.lib().faer_hip_debug_qr_one_pass_f64(1)
    _vs_oracle(oracle, F, a, 64)


@pytest.mark.parametrize("cond", [2.0, 6.0])
@pytest.mark.parametrize("m,n,bs", [(20000, 64, 64), (30000, 128, 128)])
def test_qr_f64_tall_one_pass_conditioned_panels_accepted(oracle, m, n, bs, cond):
    """panels between the Gaussian case and the guard: the fp64 Gram sums cost cond^2 eps in R and T (cond eps in V), so the
    factors are compared at cond^2-scaled tolerances and through the condition-independent properties"""
    F = init_gpu()
    rng = np.random.default_rng(int(m + n + 10 * cond))
    a = _conditioned(rng, m, n, cond)
    dqr, dh = _vs_oracle(oracle, F, a, bs, tol=(16.0 * cond ** 2, 4.0 * cond ** 2, 16.0 * cond ** 2))
    _q_properties(F, dqr, dh, a)


@pytest.mark.parametrize("cond", [30.0, 1e4, 1e9])
def test_qr_f64_tall_conditioned_panels_go_to_the_classic_path(oracle, cond):
    """beyond the guard the panel is refused before anything of it is written and the classic path takes over: at cond = 30 in the
    SECOND panel (the first one is Gaussian), else in the first; the result is the classic path's"""
    F = init_gpu()
    rng = np.random.default_rng(int(np.log10(cond)) + 5)
    m, n, bs = 24000, 128, 64
    a = _conditioned(rng, m, n, cond)
    if cond == 30.0:
        a[:, :64] = rnd(rng, m, 64)
    dqr, dh, rank, cols = _factor(F, a, bs)
    assert cols == (64 if cond == 30.0 else 0)
    ref, rh = a.copy(order="F"), np.zeros((bs, n), order="F")
    assert rank == oracle.qr_in_place(ref, rh) == n
    dr, dv, dt = _errors(to_host(dqr), to_host(dh), ref, rh, bs)
    assert dr <= 64.0 * cond and dv <= 64.0 * cond and dt <= 64.0 * cond, (dr, dv, dt)
    _q_properties(F, dqr, dh, a)


@pytest.mark.parametrize("decades", [3, 40])
def test_qr_f64_tall_badly_scaled_columns_stay(oracle, decades):
    """well-conditioned directions, column scales spread over 2 x `decades` decades: the guard looks at the equilibrated panel"""
    F = init_gpu()
    rng = np.random.default_rng(decades)
    m, n = 24000, 128
    a = rnd(rng, m, n) * (10.0 ** rng.uniform(-decades, decades, n))[None, :]
    dqr, dh = _vs_oracle(oracle, F, np.asfortranarray(a), 64)
    _q_properties(F, dqr, dh, a)


def test_qr_f64_tall_rank_deficient_matches_oracle(oracle):
    """a column that is (numerically) a combination of earlier ones and a column that is zero below the diagonal: the path stops in
    front of that panel, the classic path reproduces the oracle's rank and its pattern of skipped reflectors (tau = +inf)"""
    F = init_gpu()
    rng = np.random.default_rng(6)
    m, n, bs = 20000, 192, 64
    a = rnd(rng, m, n)
    a[:, 100] = a[:, :64] @ rnd(rng, 64, 1)[:, 0]
    ref, rh = a.copy(order="F"), np.zeros((bs, n), order="F")
    rk = oracle.qr_in_place(ref, rh)
    dqr, dh, rank, cols = _factor(F, a, bs)
    assert rank == rk and cols == 64
    assert np.array_equal(np.isinf(to_host(dh)), np.isinf(rh))
    # the first panel lives in the top 64 rows only (its reflectors leave the rows below alone), the second one is upper triangular
    # from row 64 down: every tail of the second panel is exactly zero (the reference's tau = +inf)
    b = np.zeros((m, 128), order="F")
    b[:64, :64] = 10 * np.eye(64) + 0.1 * rnd(rng, 64, 64)
    b[:64, 64:] = rnd(rng, 64, 64)
    b[64:128, 64:] = np.triu(rnd(rng, 64, 64)) + 3 * np.eye(64)
    ref, rh = b.copy(order="F"), np.zeros((bs, 128), order="F")
    rk = oracle.qr_in_place(ref, rh)
    dqr, dh, rank, cols = _factor(F, b, bs)
    assert rank == rk and 0 <= cols < 128
    assert np.array_equal(np.isinf(to_host(dh)), np.isinf(rh))
    assert np.abs(to_host(dqr) - ref).max() <= 64 * np.sqrt(m) * E * np.abs(ref).max()
    # rank 100 of 192 columns
    c = np.asfortranarray(rnd(rng, m, 100) @ rnd(rng, 100, n))
    ref, rh = c.copy(order="F"), np.zeros((bs, n), order="F")
    rk = oracle.qr_in_place(ref, rh)
    dqr, dh, rank, cols = _factor(F, c, bs)
    assert rank == rk and cols == 64


def test_qr_f64_tall_reproducible():
    F = init_gpu()
    rng = np.random.default_rng(3)
    a = rnd(rng, 40000, 256)
    out = []
    for _ in range(2):
        dqr, dh, rank, cols = _factor(F, a, 256)
        assert rank == 256 and cols == 256
        out.append((to_host(dqr).copy(), to_host(dh).copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("m", [500000])
def test_qr_f64_tall_config_q_shape_vs_oracle(oracle, m):
    """the BASELINE tall-skinny shape in fp64: the whole 5e5 x 256 factorization against the oracle (about a minute of CPU)"""
    F = init_gpu()
    rng = np.random.default_rng(m)
    a = rnd(rng, m, 256)
    _vs_oracle(oracle, F, a, 256)
