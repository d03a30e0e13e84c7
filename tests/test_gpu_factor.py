"""GPU parity: TRSM, LLT, partial-pivot LU through the C-ABI vs the CPU oracle, with the reference's own
test sizes and tolerances and full-size property tests."""
import ctypes as C

import numpy as np
import pytest

from gpu_util import EPS, init_gpu, rnd, spd, to_dev, to_host

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------ trsm
@pytest.mark.parametrize("n,k", [(1, 3), (4, 5), (7, 2), (33, 70), (64, 64), (65, 1), (128, 65), (200, 9), (300, 300)])
@pytest.mark.parametrize("upper", [False, True])
@pytest.mark.parametrize("unit", [False, True])
@pytest.mark.parametrize("order", ["F", "C"])
def test_trsm(oracle, n, k, upper, unit, order):
    F = init_gpu()
    rng = np.random.default_rng(n * 31 + k)
    t = rnd(rng, n, n) / (n if unit else 1.0) + n * np.eye(n)
    b = rnd(rng, n, k)
    dx = to_dev(b, order)
    fn = {(False, False): F.solve_lower_triangular_in_place, (True, False): F.solve_upper_triangular_in_place,
          (False, True): F.solve_unit_lower_triangular_in_place, (True, True): F.solve_unit_upper_triangular_in_place}
    fn[(upper, unit)](to_dev(t, order), dx)
    got = to_host(dx)
    ref = b.copy(order="F")
    oracle.trsm(t, ref, upper=upper, unit=unit)
    assert np.abs(got - ref).max() <= 64 * n * EPS[np.dtype(np.float64)] * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,k,order", [(128, 20000, "F"), (100, 17000, "C"), (300, 16500, "F")])
def test_trsm_many_right_hand_sides(oracle, n, k, order, dtype):
    """more right-hand sides than 16 per wavefront x one workgroup per CU: the substitution leaf runs with 32 right-hand
    sides per wavefront (trsm.hip, trsm_leaf128_launch); ragged last workgroup, both stride orders, multi-block triangle"""
    F = init_gpu()
    rng = np.random.default_rng(n + k)
    t = np.asarray(np.tril(rnd(rng, n, n)) / n + np.eye(n), dtype=dtype)
    b = rnd(rng, n, k, dtype)
    dx = to_dev(b, order)
    F.solve_lower_triangular_in_place(to_dev(t, order), dx)
    got = to_host(dx)
    ref = b.copy(order="F")
    oracle.trsm(t, ref, upper=False, unit=False)
    assert np.abs(got - ref).max() <= 64 * n * EPS[np.dtype(dtype)] * max(1.0, np.abs(ref).max())


# ------------------------------------------------------------------------------------------- llt
@pytest.mark.parametrize("n", [1, 2, 4, 8, 31, 64, 65, 127, 128, 129, 240, 300, 1024])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_llt_vs_oracle(oracle, n, dtype):
    """cholesky/ldlt/factor.rs:776-868 sizes; config R of BASELINE.json is n = 1024"""
    F = init_gpu()
    rng = np.random.default_rng(n)
    a = spd(rng, n, dtype)
    dl = to_dev(a)
    assert F.llt_factor_in_place(dl) == 0
    got = to_host(dl)
    ref = a.copy(order="F")
    assert oracle.llt_in_place(ref) == ("ok", 0)
    e = EPS[np.dtype(dtype)]
    assert (np.triu(got, 1) == np.triu(a, 1)).all()  # strict upper triangle untouched
    L = np.tril(got).astype(np.float64)
    assert np.abs(L @ L.T - a).max() <= 8 * n * e * np.abs(a).max()
    assert np.abs(np.tril(got) - np.tril(ref)).max() <= 64 * n * e * np.abs(ref).max()


@pytest.mark.parametrize("n,bad", [(10, 3), (100, 64), (300, 299), (200, 0), (700, 515)])
def test_llt_non_positive_pivot(oracle, n, bad):
    F = init_gpu()
    rng = np.random.default_rng(7)
    a = spd(rng, n)
    a[bad, bad] = -1.0 if bad == 0 else (a[bad, :bad] @ np.linalg.solve(a[:bad, :bad], a[:bad, bad])) - 1.0
    assert oracle.llt_in_place(a.copy(order="F")) == ("non_positive_pivot", bad)
    with pytest.raises(F.LltError) as ei:
        F.llt_factor_in_place(to_dev(a))
    assert ei.value.index == bad


def test_llt_regularization(oracle):
    F = init_gpu()
    a = np.diag([4.0, 1e-20, 9.0])
    d = to_dev(a)
    assert F.llt_factor_in_place(d, regularization=(1.0, 1e-10)) == 1
    ref = a.copy(order="F")
    assert oracle.llt_in_place(ref, reg_delta=1.0, reg_eps=1e-10) == ("ok", 1)
    assert np.allclose(np.diag(to_host(d)), np.diag(ref))


def test_llt_solve(oracle):
    """cholesky/llt/solve.rs:56: A X ~ B, eps*128*8n"""
    F = init_gpu()
    rng = np.random.default_rng(8)
    for n in (50, 200, 400):
        a, b = spd(rng, n), rnd(rng, n, 7)
        llt = F.Llt(to_dev(a))
        x = to_dev(b)
        llt.solve_in_place(x)
        assert np.abs(a @ to_host(x) - b).max() <= 2.2e-16 * 128 * 8 * n * np.abs(b).max() * 10


def test_llt_host_pointer():
    F = init_gpu()
    rng = np.random.default_rng(9)
    a = spd(rng, 333)
    l = a.copy(order="F")
    assert F.llt_factor_in_place(l) == 0
    L = np.tril(l)
    assert np.abs(L @ L.T - a).max() < 1e-10 * np.abs(a).max()


def test_llt_full_size_property(oracle):
    """BASELINE config C (N = 16384 fp64): ||L L^T x - A x|| via matvecs, no oracle"""
    import torch

    F = init_gpu()
    n = 16384
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
    a = (a @ a.t() + n * torch.eye(n, dtype=torch.float64, device="cuda")).t()
    l = a.clone()
    assert F.llt_factor_in_place(l) == 0
    F.synchronize()
    L = torch.tril(l)
    x = torch.randn((n, 4), dtype=torch.float64, device="cuda", generator=g)
    r = L @ (L.t() @ x) - a @ x
    assert r.abs().max().item() <= 64 * n * 2.3e-16 * (a.abs() @ x.abs()).max().item()
    # sampled parity at the BASELINE size: the leading k x k block of L is the Cholesky factor of the leading block of A
    # whatever follows it (cholesky/ldlt/factor.rs:367-498) -- compared entry by entry with the oracle's factorization
    k = 2048
    ref = np.asfortranarray(a[:k, :k].cpu().numpy())
    assert oracle.llt_in_place(ref) == ("ok", 0)
    d = np.abs(np.tril(l[:k, :k].cpu().numpy()) - np.tril(ref)).max()
    assert d <= 64 * k * 2.3e-16 * np.abs(np.tril(ref)).max(), d


@pytest.mark.parametrize("n,tail", [(4096, 0), (4363, 0), (5120 + 77, 2048), (5120 + 77, 1 << 20), (6144, 3000), (3072 + 5, 1024)])
def test_llt_lookahead_path(n, tail, monkeypatch):
    """the blocked driver of large matrices (potrf.hip): look-ahead steps on two CU-masked streams (1024 columns
    each) followed by the sequential tail of tall left-looking panels; thresholds lowered through the environment
    so that every mix (look-ahead only, both, tail only) and a ragged last step run at test sizes.  Entrywise
    L L^T == A on the lower triangle, untouched strict upper triangle, same answer twice"""
    import torch

    F = init_gpu()
    monkeypatch.setenv("FAER_HIP_LLT_LA_MIN", "2048")
    monkeypatch.setenv("FAER_HIP_LLT_TAIL", str(tail))
    g = torch.Generator(device="cuda").manual_seed(n)
    b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
    a = (b @ b.t() + n * torch.eye(n, dtype=torch.float64, device="cuda")).t()
    marked = a.clone()
    iu = torch.triu_indices(n, n, 1, device="cuda")
    marked[iu[0], iu[1]] = -7.5
    l = marked.clone()
    assert F.llt_factor_in_place(l) == 0
    F.synchronize()
    assert (l[iu[0], iu[1]] == -7.5).all().item()
    L = torch.tril(l)
    err = (torch.tril(L @ L.t() - a)).abs().max().item()
    assert err <= 32 * n * 2.3e-16 * a.abs().max().item()
    l2 = marked.clone()
    assert F.llt_factor_in_place(l2) == 0
    F.synchronize()
    assert torch.equal(l, l2)


@pytest.mark.parametrize("tail", [0, 2500, 1 << 20])
def test_llt_lookahead_failure_index(tail, monkeypatch):
    """first non-positive pivot deep inside a later step of the blocked driver (look-ahead step / tail panel):
    same index as the definition"""
    import torch

    F = init_gpu()
    monkeypatch.setenv("FAER_HIP_LLT_LA_MIN", "2048")
    monkeypatch.setenv("FAER_HIP_LLT_TAIL", str(tail))
    n, bad = 5000, 3333
    g = torch.Generator(device="cuda").manual_seed(5)
    b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
    a = (b @ b.t() + n * torch.eye(n, dtype=torch.float64, device="cuda")).t().clone()
    a[bad, bad] = -1.0
    with pytest.raises(F.LltError) as ei:
        F.llt_factor_in_place(a)
    assert ei.value.index == bad


# -------------------------------------------------------------------------------------------- lu
@pytest.mark.parametrize("m,n", [(1, 1), (2, 2), (3, 3), (31, 31), (32, 32), (33, 33), (128, 128), (255, 255), (256, 256),
                                 (257, 257), (300, 8), (8, 300), (40, 17), (17, 40), (1000, 1000), (2000, 64)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_plu_vs_oracle(oracle, m, n, dtype):
    """lu/partial_pivoting/factor.rs:304-404 `test_plu` sizes; P^-1 L U ~ A (1e-13 scaled)"""
    F = init_gpu()
    rng = np.random.default_rng(m * 7 + n)
    a = rnd(rng, m, n, dtype)
    dlu = to_dev(a)
    perm, perm_inv, nt = F.partial_piv_lu_factor_in_place(dlu)
    lu = to_host(dlu)
    perm = perm.astype(np.int64)
    ref = a.copy(order="F")
    rperm, rinv, rnt = oracle.lu_in_place(ref)
    size = min(m, n)
    e = EPS[np.dtype(dtype)]
    L = (np.tril(lu[:, :size], -1) + np.eye(m, size)).astype(np.float64)
    U = np.triu(lu[:size, :]).astype(np.float64)
    assert (perm_inv.astype(np.int64)[perm] == np.arange(m)).all()
    assert np.abs(L @ U - a[perm]).max() <= 16 * max(m, n) * e * np.abs(a).max()
    assert np.abs(np.tril(lu, -1)).max(initial=0) <= 1.0 + 4 * e
    # same pivot rule as the reference => same permutation (random data has no near ties)
    assert (perm == rperm).all() and nt == rnt
    # factor-level agreement with the oracle: two LU codes with the same pivots differ by the forward error of the
    # factorization, c n eps kappa, kappa = condition of the pivoted leading block
    kappa = np.linalg.cond(a[perm][:size, :size].astype(np.float64)) if size > 0 else 1.0
    assert np.abs(lu - ref).max() <= 4 * max(m, n) * e * kappa * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("m,n", [(3, 3), (33, 33), (300, 8), (8, 300), (257, 257), (1000, 700), (2000, 64), (700, 1000)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_plu_non_cooperative_leaves_vs_oracle(oracle, m, n, dtype):
    """the fallback of the cooperative panel kernel (getrf.hip getrf_leaf_general: taller panels than it can keep resident,
    rerun after an exchange timeout) forced on for ordinary shapes: identical pivots, factors within the forward error"""
    F = init_gpu()
    rng = np.random.default_rng(m * 5 + n)
    a = rnd(rng, m, n, dtype)
    ref = a.copy(order="F")
    rperm, rinv, rnt = oracle.lu_in_place(ref)
    dlu = to_dev(a)
    F.lib().faer_hip_debug_lu_force_general(1)
    try:
        perm, perm_inv, nt = F.partial_piv_lu_factor_in_place(dlu)
    finally:
        F.lib().faer_hip_debug_lu_force_general(0)
    lu = to_host(dlu)
    perm = perm.astype(np.int64)
    size = min(m, n)
    e = EPS[np.dtype(dtype)]
    assert (perm == rperm).all() and nt == rnt
    kappa = np.linalg.cond(a[perm][:size, :size].astype(np.float64))
    assert np.abs(lu - ref).max() <= 4 * max(m, n) * e * kappa * max(1.0, np.abs(ref).max())


def test_plu_taller_than_the_cooperative_limit(oracle):
    """1 100 000 fp64 rows: more than the cooperative kernel keeps resident on 256 CUs (1 048 576).  The reference has no
    such limit (lu/partial_pivoting/factor.rs:234-295); round 2 aborted here, now the non-cooperative leaf takes over"""
    F = init_gpu()
    m, n = 1100000, 6
    rng = np.random.default_rng(99)
    a = rnd(rng, m, n, np.float64)
    ref = a.copy(order="F")
    rperm, _, rnt = oracle.lu_in_place(ref)
    dlu = to_dev(a)
    perm, _, nt = F.partial_piv_lu_factor_in_place(dlu)
    assert (perm.astype(np.int64) == rperm).all() and nt == rnt
    assert np.abs(to_host(dlu) - ref).max() <= 1e-9 * np.abs(ref).max()


def test_plu_ties_and_zero_column(oracle):
    """first strictly largest |a_ij| wins; an all-zero column keeps the diagonal (factor.rs:35-43)"""
    F = init_gpu()
    a = np.array([[1.0, 2, 3, 4], [-1.0, 5, 6, 7], [1.0, 8, 9, 1], [0.5, 1, 1, 1]], order="F")
    ref = a.copy(order="F")
    rperm, _, _ = oracle.lu_in_place(ref)
    d = to_dev(a)
    perm, _, _ = F.partial_piv_lu_factor_in_place(d)
    assert (perm.astype(np.int64) == rperm).all()
    assert np.allclose(to_host(d), ref)
    # (600 rows: the cooperative leaf; 300 and 40 rows: the single-workgroup leaf, csrc/lu_small_leaf.h)
    for rows in (600, 300, 40):
        z = np.zeros((rows, 5), order="F")
        z[:, 1:] = np.random.default_rng(3).standard_normal((rows, 4))
        ref = z.copy(order="F")
        rperm, _, _ = oracle.lu_in_place(ref)
        d = to_dev(z)
        perm, _, _ = F.partial_piv_lu_factor_in_place(d)
        got = to_host(d)
        assert (perm.astype(np.int64) == rperm).all()
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.allclose(got[~np.isnan(got)], ref[~np.isnan(ref)])
    # ties inside a panel of several wavefronts: equal |a| in different wavefronts' rows, the smaller row index wins
    t = np.random.default_rng(5).standard_normal((200, 6))
    t[150, 0] = t[20, 0] = -(np.abs(t[:, 0]).max() + 1.0)
    t[199, 2] = 9.0
    t[70, 2] = -9.0
    ref = np.asfortranarray(t.copy())
    rperm, _, _ = oracle.lu_in_place(ref)
    d = to_dev(np.asfortranarray(t))
    perm, _, _ = F.partial_piv_lu_factor_in_place(d)
    assert (perm.astype(np.int64) == rperm).all()
    assert np.allclose(to_host(d), ref, rtol=1e-12, atol=1e-12)


def test_plu_full_size_property(oracle):
    """BASELINE config L on one GPU (N = 16384 fp64): ||L U x - P A x||, |L| <= 1"""
    import torch

    F = init_gpu()
    n = 16384
    g = torch.Generator(device="cuda").manual_seed(2)
    a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    lu = a.clone()
    perm, _, _ = F.partial_piv_lu_factor_in_place(lu)
    F.synchronize()
    p = torch.as_tensor(perm.astype(np.int64), device="cuda")
    assert sorted(perm.tolist()) == list(range(n))
    L = torch.tril(lu, -1) + torch.eye(n, dtype=torch.float64, device="cuda")
    assert torch.tril(lu, -1).abs().max().item() <= 1.0 + 1e-14
    x = torch.randn((n, 4), dtype=torch.float64, device="cuda", generator=g)
    r = L @ (torch.triu(lu) @ x) - a[p] @ x
    scale = (L.abs() @ (torch.triu(lu).abs() @ x.abs())).max().item()
    assert r.abs().max().item() <= 64 * n * 2.3e-16 * scale
    # sampled parity at the BASELINE size: the first k columns (pivots, L panel, leading block of U) depend on the first k
    # columns of A only (lu/partial_pivoting/factor.rs:68-187) -- compared with the oracle's factorization of that panel.  Rows
    # below k are moved again by later interchanges: row i of the final L is the oracle's row of the same ORIGINAL row.
    k = 512
    ref = np.asfortranarray(a[:, :k].cpu().numpy())
    rperm, rinv, _ = oracle.lu_in_place(ref)
    perm64 = perm.astype(np.int64)
    assert (perm64[:k] == rperm[:k]).all()  # identical pivots, column by column
    got = lu[:, :k].cpu().numpy()
    want = ref[rinv.astype(np.int64)[perm64], :]
    kappa = np.linalg.cond(a[p][:k, :k].cpu().numpy())
    assert np.abs(got - want).max() <= 4 * n * 2.3e-16 * kappa * max(1.0, np.abs(want).max()), np.abs(got - want).max()


@pytest.mark.parametrize("m,n", [(4096, 4096), (4500, 4321), (6000, 5000)])
def test_plu_lookahead_path(m, n):
    """min(m, n) >= 4096 runs the look-ahead LU driver (panel stream + bulk stream, 512-column steps, getrf.hip):
    P A == L U entrywise, |L| <= 1, a valid permutation, and the bitwise same answer twice"""
    import torch

    F = init_gpu()
    g = torch.Generator(device="cuda").manual_seed(m + n)
    a = torch.randn((n, m), dtype=torch.float64, device="cuda", generator=g).t()
    lu = a.clone()
    perm, perm_inv, _ = F.partial_piv_lu_factor_in_place(lu)
    F.synchronize()
    assert sorted(perm.tolist()) == list(range(m)) and all(perm[perm_inv[i]] == i for i in range(0, m, 97))
    p = torch.as_tensor(perm.astype(np.int64), device="cuda")
    size = min(m, n)
    L = torch.tril(lu, -1)[:, :size]
    L[:size, :size] += torch.eye(size, dtype=torch.float64, device="cuda")
    U = torch.triu(lu)[:size, :]
    assert torch.tril(lu, -1).abs().max().item() <= 1.0 + 1e-14
    err = (L @ U - a[p]).abs().max().item()
    assert err <= 16 * size * 2.3e-16 * (L.abs() @ U.abs()).max().item()
    lu2 = a.clone()
    perm2, _, _ = F.partial_piv_lu_factor_in_place(lu2)
    F.synchronize()
    assert np.array_equal(perm, perm2) and torch.equal(lu, lu2)


@pytest.mark.parametrize("plan", [(2048, 3584, 2048), (1024, 100000, 2048), (100000, 100000, 2048), (512, 1536, 2048)])
def test_plu_lookahead_phases_and_transitions_at_small_n(oracle, plan):
    """the look-ahead LU driver has three phases -- pipelined bulk-bound steps, plain bulk-bound steps, staged 256-column steps -- whose
    switch-over points sit at 9-13 k rows; `faer_hip_debug_lu_plan` moves them so that N = 5120 / 3072 runs every phase and every
    transition.  Pivots and factors must not depend on the plan: identical permutation to the default plan's and to the
    oracle's on the first panel, P A == L U, the bitwise same answer twice."""
    import torch

    F = init_gpu()
    nb2_from, pipe_from, la_min = plan
    for n in (5120, 3072):
        g = torch.Generator(device="cuda").manual_seed(n + 11)
        a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
        base = a.clone()
        perm0, _, _ = F.partial_piv_lu_factor_in_place(base)  # the tuned plan (n = 3072: the flat / recursive driver)
        F.synchronize()
        F.lib().faer_hip_debug_lu_plan(C.c_size_t(nb2_from), C.c_size_t(pipe_from), C.c_size_t(la_min))
        try:
            lu = a.clone()
            perm, _, _ = F.partial_piv_lu_factor_in_place(lu)
            lu2 = a.clone()
            perm2, _, _ = F.partial_piv_lu_factor_in_place(lu2)
            F.synchronize()
        finally:
            F.lib().faer_hip_debug_lu_plan(C.c_size_t(0), C.c_size_t(0), C.c_size_t(0))
        assert np.array_equal(perm, perm0) and np.array_equal(perm, perm2) and torch.equal(lu, lu2)
        p = torch.as_tensor(perm.astype(np.int64), device="cuda")
        L = torch.tril(lu, -1) + torch.eye(n, dtype=torch.float64, device="cuda")
        U = torch.triu(lu)
        assert (L @ U - a[p]).abs().max().item() <= 16 * n * 2.3e-16 * (L.abs() @ U.abs()).max().item()
        scale = max(1.0, lu.abs().max().item())
        assert (lu - base).abs().max().item() <= 4 * n * 2.3e-16 * 1e4 * scale  # same pivots: forward-error level agreement
        k = 256
        ref = np.asfortranarray(a[:, :k].cpu().numpy())
        rperm, _, _ = oracle.lu_in_place(ref)
        assert (perm.astype(np.int64)[:k] == rperm[:k]).all()


def test_lending_the_panel_cus_gives_the_bitwise_same_factors():
    """faer_hip_debug_lend_cus(1): the big trailing products of the look-ahead LU / Cholesky hand their tiles out through per-XCD
    counters and a helper launch on the panel stream takes some of them (gemm.hip GemmArgs::ticket).  Who computes a tile must not
    matter: bitwise the same factors and pivots as with lending off (the default, profiles/r06_exp_lend.txt)."""
    import torch

    F = init_gpu()
    n = 12288
    g = torch.Generator(device="cuda").manual_seed(21)
    a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    out = {}
    for lend in (0, 1):
        F.lib().faer_hip_debug_lend_cus(lend)
        try:
            lu = a.clone()
            perm, _, _ = F.partial_piv_lu_factor_in_place(lu)
            F.synchronize()
        finally:
            F.lib().faer_hip_debug_lend_cus(0)
        out[lend] = (perm, lu)
    assert np.array_equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    del out, lu
    b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
    spd_ = (b @ b.t() / n + 2 * torch.eye(n, dtype=torch.float64, device="cuda")).t()
    del b
    res = {}
    for lend in (0, 1):
        F.lib().faer_hip_debug_lend_cus(lend)
        try:
            l = spd_.clone()
            assert F.llt_factor_in_place(l) == 0
            F.synchronize()
        finally:
            F.lib().faer_hip_debug_lend_cus(0)
        res[lend] = l
    assert torch.equal(res[0], res[1])


def test_lookahead_paths_fp32():
    """fp32 through the two-stream drivers: LLT n = 8192 (L L^T == A) and LU n = 4608 (P A == L U), tolerances
    scaled with eps_f32"""
    import torch

    F = init_gpu()
    eps = 1.2e-7
    n = 8192
    g = torch.Generator(device="cuda").manual_seed(8)
    b = torch.randn((n, n), dtype=torch.float32, device="cuda", generator=g)
    a = (b @ b.t() + n * torch.eye(n, dtype=torch.float32, device="cuda")).t()
    l = a.clone()
    assert F.llt_factor_in_place(l) == 0
    F.synchronize()
    L = torch.tril(l).double()
    err = torch.tril(L @ L.t() - a.double()).abs().max().item()
    assert err <= 8 * n * eps * a.abs().max().item()
    n = 4608
    a = torch.randn((n, n), dtype=torch.float32, device="cuda", generator=g).t()
    lu = a.clone()
    perm, _, _ = F.partial_piv_lu_factor_in_place(lu)
    F.synchronize()
    assert sorted(perm.tolist()) == list(range(n))
    p = torch.as_tensor(perm.astype(np.int64), device="cuda")
    Lm = (torch.tril(lu, -1) + torch.eye(n, dtype=torch.float32, device="cuda")).double()
    U = torch.triu(lu).double()
    assert torch.tril(lu, -1).abs().max().item() <= 1.0 + 1e-6
    err = (Lm @ U - a.double()[p]).abs().max().item()
    assert err <= 8 * n * eps * (Lm.abs() @ U.abs()).max().item()


@pytest.mark.parametrize("m,n,dtype", [(200000, 64, np.float64), (600000, 16, np.float64), (300000, 40, np.float32), (1100000, 8, np.float32),
                                       (5000, 8, np.float64), (70000, 24, np.float64)])
def test_plu_tall_panels_pick_a_resident_leaf_shape(oracle, m, n, dtype):
    """tall panels: every workgroup of the cooperative leaf has to be resident, so the driver trades leaf width for
    rows per workgroup (64 x 512 ... 8 x 4096 rows; getrf.hip leaf_width_for).  Same pivots as the oracle."""
    F = init_gpu()
    rng = np.random.default_rng(m + n)
    a = rnd(rng, m, n, dtype)
    dlu = to_dev(a)
    perm, perm_inv, nt = F.partial_piv_lu_factor_in_place(dlu)
    lu = to_host(dlu)
    ref = a.copy(order="F")
    rperm, _, rnt = oracle.lu_in_place(ref)
    assert np.array_equal(perm.astype(np.int64), rperm) and nt == rnt
    e = EPS[np.dtype(dtype)]
    assert np.abs(lu - ref).max() <= 64 * n * e * max(1.0, np.abs(ref).max())


# -------------------------------------------------------------------------------------------- lu with full pivoting
@pytest.mark.parametrize("m,n,layout", [(1, 1, "F"), (2, 3, "F"), (5, 5, "F"), (40, 30, "F"), (30, 40, "F"), (64, 64, "C"), (33, 50, "C"),
                                        (300, 300, "F"), (1030, 700, "F"), (700, 1100, "C"), (2100, 9, "F")])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_full_piv_lu_vs_oracle(oracle, m, n, layout, dtype):
    """csrc/fplu.hip through the C-ABI: IDENTICAL permutations (the pivot search is index work: first maximum in
    (column, row) order of the view the reference works on), factors within tolerance, solve and transpose solve"""
    F = init_gpu()
    rng = np.random.default_rng(m * 31 + n)
    a = np.array(rng.standard_normal((m, n)), dtype=dtype, order=layout)
    ref = a.copy(order=layout)
    rp, rpi, cp, cpi, nt = oracle.full_piv_lu_in_place(ref)
    da = to_dev(a, layout)
    rf, rb, cf, cb, cnt = F.full_piv_lu_factor_in_place(da)
    assert np.array_equal(rf.astype(np.int64), rp) and np.array_equal(rb.astype(np.int64), rpi)
    assert np.array_equal(cf.astype(np.int64), cp) and np.array_equal(cb.astype(np.int64), cpi) and cnt == nt
    got = to_host(da)
    e = EPS[np.dtype(dtype)]
    assert np.abs(got - ref).max() <= 64 * max(m, n) * e * max(1.0, np.abs(ref).max())
    if m == n and m > 1:
        b = rnd(rng, n, 3, dtype)
        a64 = a.astype(np.float64)
        tol = 256 * n * e * np.linalg.cond(a64)
        x = to_dev(b)
        F.full_piv_lu_solve_in_place(da, rf, rb, cf, cb, x)
        assert np.abs(a64 @ to_host(x) - b).max() <= tol * np.abs(b).max()
        x = to_dev(b)
        F.full_piv_lu_solve_in_place(da, rf, rb, cf, cb, x, transpose=True)
        assert np.abs(a64.T @ to_host(x) - b).max() <= tol * np.abs(b).max()


@pytest.mark.parametrize("m,n,layout", [(1, 1, "F"), (1, 7, "F"), (7, 1, "F"), (5, 5, "F"), (300, 300, "F"), (1030, 700, "F"), (700, 1100, "C"),
                                        (2100, 9, "F"), (9, 2100, "F"), (1500, 1500, "C")])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_full_piv_lu_one_launch_path_equals_in_place_path(m, n, layout, dtype):
    """the default path (one launch per step between two scratch copies of the matrix) and the in-place path of rounds 1-6
    (faer_hip_debug_fplu_inplace(1): interchange launch + update launch per step) run the same arithmetic per entry and the same
    pivot rule: factors and permutations bit for bit -- Gaussian data, small integers (ties in the pivot search) and a matrix of
    rank 3 (the elimination ends early and the live block travels back from the scratch copy)"""
    F = init_gpu()
    rng = np.random.default_rng(m * 17 + n)
    cases = [np.array(rng.standard_normal((m, n)), dtype=dtype, order=layout),
             np.array(rng.integers(-3, 4, (m, n)), dtype=dtype, order=layout)]
    r = min(3, m, n)
    cases.append(np.array((rng.integers(-2, 3, (m, r)) @ rng.integers(-2, 3, (r, n))), dtype=dtype, order=layout))
    for a in cases:
        out = []
        for mode in (1, 0):
            F.lib().faer_hip_debug_fplu_inplace(mode)
            try:
                da = to_dev(a, layout)
                rf, rb, cf, cb, cnt = F.full_piv_lu_factor_in_place(da)
                out.append((to_host(da), rf, rb, cf, cb, cnt))
            finally:
                F.lib().faer_hip_debug_fplu_inplace(0)
        (g0, rf0, rb0, cf0, cb0, c0), (g1, rf1, rb1, cf1, cb1, c1) = out
        assert np.array_equal(rf0, rf1) and np.array_equal(rb0, rb1) and np.array_equal(cf0, cf1) and np.array_equal(cb0, cb1) and c0 == c1
        assert np.array_equal(g0, g1)


def test_full_piv_lu_singular_trailing_block_and_size_property():
    """exactly singular trailing block: the elimination stops like the reference's (identity transpositions from
    there on); N = 2048: P A Q x == L (U x) and |l_ij| <= 1, independent of the oracle"""
    import torch

    F = init_gpu()
    a = np.zeros((6, 5), order="F")
    a[:2, :2] = [[4.0, 1.0], [2.0, 3.0]]
    da = to_dev(a)
    rf, _, cf, _, _ = F.full_piv_lu_factor_in_place(da)
    lu = to_host(da)
    L = np.tril(lu[:, :5], -1) + np.eye(6, 5)
    assert np.abs(L @ np.triu(lu[:5, :]) - a[rf.astype(int)][:, cf.astype(int)]).max() == 0.0
    n = 2048
    g = torch.Generator(device="cuda").manual_seed(9)
    A = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    W = A.clone()
    rf, _, cf, _, _ = F.full_piv_lu_factor_in_place(W)
    F.synchronize()
    x = torch.randn((n, 2), dtype=torch.float64, device="cuda", generator=g)
    Lm = torch.tril(W, -1) + torch.eye(n, dtype=torch.float64, device="cuda")
    p = torch.as_tensor(rf.astype(np.int64), device="cuda")
    q = torch.as_tensor(cf.astype(np.int64), device="cuda")
    r = (Lm @ (torch.triu(W) @ x) - A[p][:, q] @ x).abs().max().item()
    assert r <= 64 * n * 2.3e-16 * (A.abs() @ x.abs()).max().item()
    assert torch.tril(W, -1).abs().max().item() <= 1.0 + 1e-12


# -------------------------------------------------------------------------------------------- distributed lu
@pytest.mark.gpu
@pytest.mark.parametrize("two_min", ["0", None])
@pytest.mark.parametrize("m,n,nb", [(512, 512, 64), (1000, 1000, 128), (700, 500, 96), (300, 420, 64)])
def test_dist_lu_device_backend_single_rank(oracle, m, n, nb, two_min, monkeypatch):
    """the device backend of the distributed LU (csrc/dist.hip) on ONE rank (the broadcast is the identity):
    same pivots as the oracle, factors within tolerance, exactly one broadcast per block column.  The multi-rank
    control flow of the same template is covered on CPU by tests/test_dist_lu.py (gloo, world_size 2 and 3).
    two_min = "0": every step on the bulk + panel streams (by default steps with little trailing work run on one stream)"""
    if two_min is not None:
        monkeypatch.setenv("FAER_HIP_DIST_TWO_MIN", two_min)
    F = init_gpu()
    rng = np.random.default_rng(21)
    a = rnd(rng, m, n)
    ref = a.copy(order="F")
    perm, perm_inv, nt = oracle.lu_in_place(ref)
    da = to_dev(a)
    calls = []
    fwd, bwd, cnt = F.dist_partial_piv_lu(da, n, nb, 0, 1, lambda t, root: calls.append((t.numel(), root)))
    got = to_host(da)
    assert np.array_equal(fwd.astype(np.int64), perm) and np.array_equal(bwd.astype(np.int64), perm_inv) and cnt == nt
    tol = 64 * max(m, n) * EPS[np.dtype(np.float64)]
    assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max())
    assert len(calls) == (min(m, n) + nb - 1) // nb and all(r == 0 for _, r in calls)


@pytest.mark.gpu
@pytest.mark.parametrize("use_async", [False, True])
def test_dist_lu_device_backend_async_transport(oracle, use_async):
    """the look-ahead protocol of the distributed LU with the optional asynchronous transport (ibcast / wait):
    every broadcast is begun once and awaited once before its panel is used; same result as the blocking form"""
    F = init_gpu()
    m = n = 640
    nb = 64
    rng = np.random.default_rng(22)
    a = rnd(rng, m, n)
    ref = a.copy(order="F")
    perm, _, nt = oracle.lu_in_place(ref)
    da = to_dev(a)
    log = []

    class Handle:
        def __init__(self, k):
            self.k = k

        def wait(self):
            log.append(("wait", self.k))

    def ibcast(t, root):
        log.append(("begin", len([e for e in log if e[0] == "begin"])))
        return Handle(log[-1][1])

    fwd, _, cnt = F.dist_partial_piv_lu(da, n, nb, 0, 1, lambda t, root: log.append(("blocking", 0)), ibcast=ibcast if use_async else None)
    assert np.array_equal(fwd.astype(np.int64), perm) and cnt == nt
    assert np.abs(to_host(da) - ref).max() <= 64 * n * EPS[np.dtype(np.float64)] * max(1.0, np.abs(ref).max())
    nblk = n // nb
    if use_async:
        begins = [e[1] for e in log if e[0] == "begin"]
        waits = [e[1] for e in log if e[0] == "wait"]
        assert begins == list(range(nblk)) and waits == list(range(nblk))
        # look-ahead: panel k+1 is on its way before panel k+2 .. and never more than two in flight
        for k in range(nblk):
            assert log.index(("begin", k)) < log.index(("wait", k))
            if k + 2 < nblk:
                assert log.index(("wait", k)) < log.index(("begin", k + 2))
    else:
        assert len(log) == nblk


# -------------------------------------------------------------------------------------------- distributed llt
@pytest.mark.gpu
@pytest.mark.parametrize("n,nb", [(512, 64), (1000, 128), (700, 96), (130, 256), (1536, 512)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dist_llt_device_backend_single_rank(oracle, n, nb, dtype):
    """the device backend of the distributed Cholesky (csrc/dist.hip) on ONE rank: factor within tolerance of the
    oracle, strict upper triangle untouched, up to four chunk broadcasts per block column.  Multi-rank control flow of the same
    template: tests/test_dist_llt.py (gloo, world_size 2 and 3)."""
    F = init_gpu()
    rng = np.random.default_rng(n + nb)
    a = spd(rng, n, dtype)
    ref = a.copy(order="F")
    assert oracle.llt_in_place(ref) == ("ok", 0)
    marked = a.copy(order="F")
    iu = np.triu_indices(n, 1)
    marked[iu] = -7.5
    da = to_dev(marked)
    calls = []
    assert F.dist_llt(da, n, min(nb, n), 0, 1, lambda t, root: calls.append(root)) == 0
    got = to_host(da)
    assert (got[iu] == -7.5).all()
    il = np.tril_indices(n)
    assert np.abs(got[il] - ref[il]).max() <= 64 * n * EPS[np.dtype(dtype)] * np.abs(ref[il]).max()
    nblk = (n + min(nb, n) - 1) // min(nb, n)
    assert len(calls) == sum(min(4, nblk - k - 1) for k in range(nblk))  # the row chunks of every panel (dist_llt.h, LLT_NCH)


@pytest.mark.gpu
def test_dist_llt_device_backend_failure_index():
    F = init_gpu()
    n, nb, bad = 900, 128, 517
    rng = np.random.default_rng(4)
    a = spd(rng, n, np.float64)
    a[bad, bad] = -1.0
    with pytest.raises(F.LltError) as ei:
        F.dist_llt(to_dev(a), n, nb, 0, 1, lambda t, root: None)
    assert ei.value.index == bad


def test_llt_lookahead_is_deterministic_under_interleaved_work(monkeypatch):
    """the two-stream look-ahead driver (n = 8192, look-ahead down to 2048 remaining rows) must give the bitwise
    same factor every time, also with unrelated work queued around it (a cross-stream race would show up as a
    mismatch or a spurious failure)"""
    import torch

    F = init_gpu()
    monkeypatch.setenv("FAER_HIP_LLT_TAIL", "2048")
    n = 8192
    g = torch.Generator(device="cuda").manual_seed(17)
    b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
    a = (b @ b.t() + n * torch.eye(n, dtype=torch.float64, device="cuda")).t()
    ref = None
    for it in range(6):
        work = a.clone()
        if it % 2 == 1:
            _ = b @ b
        assert F.llt_factor_in_place(work) == 0
        F.synchronize()
        if ref is None:
            ref = work
        else:
            assert torch.equal(torch.tril(work), torch.tril(ref)), f"iteration {it} differs"


# -------------------------------------------------------------------------------------------- solve side
@pytest.mark.parametrize("n,k", [(1, 1), (17, 3), (200, 5), (513, 64), (1500, 300)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_plu_solve_and_transpose_solve(n, k, dtype):
    """lu/partial_pivoting/solve.rs:20-80 through the C-ABI: A x = b and A^T x = b (reference test tolerances
    are 1e-10-ish at n ~ 100 in fp64; scaled with n and eps here)"""
    F = init_gpu()
    rng = np.random.default_rng(n + k)
    a = rnd(rng, n, n, dtype) + n * np.eye(n, dtype=dtype)  # well conditioned
    b = rnd(rng, n, k, dtype)
    lu = to_dev(a)
    perm, perm_inv, _ = F.partial_piv_lu_factor_in_place(lu)
    tol = 64 * n * EPS[np.dtype(dtype)]
    x = to_dev(b)
    F.partial_piv_lu_solve_in_place(lu, perm, perm_inv, x)
    xs = to_host(x).astype(np.float64)
    ref = np.linalg.solve(a.astype(np.float64), b.astype(np.float64))
    assert np.abs(xs - ref).max() <= tol * max(1.0, np.abs(ref).max())
    xt = to_dev(b)
    F.partial_piv_lu_solve_in_place(lu, perm, perm_inv, xt, transpose=True)
    reft = np.linalg.solve(a.astype(np.float64).T, b.astype(np.float64))
    assert np.abs(to_host(xt).astype(np.float64) - reft).max() <= tol * max(1.0, np.abs(reft).max())


# -------------------------------------------------------------------------------------------- ldlt
def _quasi_definite(rng, n, dtype=np.float64):
    n1 = n // 2
    h = rng.standard_normal((n, n))
    H = h[:n1, :n1] @ h[:n1, :n1].T + n * np.eye(n1)
    G = h[n1:, n1:] @ h[n1:, n1:].T + n * np.eye(n - n1)
    B = h[n1:, :n1]
    return np.asarray(np.block([[H, B.T], [B, -G]]), dtype=dtype, order="F"), n1


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 127, 128, 129, 257, 640, 1100])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ldlt_vs_oracle(oracle, n, dtype):
    F = init_gpu()
    rng = np.random.default_rng(n)
    a, n1 = _quasi_definite(rng, n, dtype)
    marked = a.copy(order="F")
    iu = np.triu_indices(n, 1)
    marked[iu] = -7.5
    ref = marked.copy(order="F")
    assert oracle.ldlt_in_place(ref) == ("ok", 0)
    d = to_dev(marked)
    assert F.ldlt_factor_in_place(d) == 0
    got = to_host(d)
    assert (got[iu] == -7.5).all()
    tol = 64 * n * EPS[np.dtype(dtype)]
    assert np.abs(np.tril(got) - np.tril(ref)).max() <= tol * max(1.0, np.abs(np.tril(ref)).max())
    D = np.diag(got).astype(np.float64)
    assert (D[:n1] > 0).all() and (D[n1:] < 0).all()


def test_ldlt_zero_pivot_regularization_and_solve(oracle):
    F = init_gpu()
    rng = np.random.default_rng(3)
    n, bad = 300, 211
    a, _ = _quasi_definite(rng, n)
    # make the leading (bad+1) x (bad+1) minor singular: row/column `bad` duplicates row/column 5
    s = a.copy(order="F")
    s[bad, :] = s[5, :]
    s[:, bad] = s[:, 5]
    s[bad, bad] = s[5, 5]
    ref = s.copy(order="F")
    kind, idx = oracle.ldlt_in_place(ref)
    d = to_dev(s)
    if kind == "zero_pivot":
        with pytest.raises(F.LdltError) as ei:
            F.ldlt_factor_in_place(d)
        assert ei.value.index == idx
    # regularization with expected signs: same count and factors as the oracle
    signs = np.where(np.arange(n) < n // 2, 1, -1).astype(np.int8)
    ref = s.copy(order="F")
    r = oracle.ldlt_in_place(ref, 1e-2, 1e-9, signs=signs)
    assert r[0] == "ok"
    d = to_dev(s)
    assert F.ldlt_factor_in_place(d, (1e-2, 1e-9), signs=signs) == r[1]
    got = to_host(d)
    assert np.abs(np.tril(got) - np.tril(ref)).max() <= 1e-6 * max(1.0, np.abs(np.tril(ref)).max())
    # solve with a well conditioned matrix
    ld = to_dev(a)
    assert F.ldlt_factor_in_place(ld) == 0
    b = rnd(rng, n, 5)
    x = to_dev(b)
    F.ldlt_solve_in_place(ld, x)
    assert np.abs(to_host(x) - np.linalg.solve(a, b)).max() <= 1e-9


def test_ldlt_large_property():
    """n = 4096 fp64: L D L^T == A entrywise on the lower triangle"""
    import torch

    F = init_gpu()
    n = 4096
    n1 = n // 2
    g = torch.Generator(device="cuda").manual_seed(12)
    h = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
    a = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    a[:n1, :n1] = h[:n1, :n1] @ h[:n1, :n1].t() + n * torch.eye(n1, dtype=torch.float64, device="cuda")
    a[n1:, n1:] = -(h[n1:, n1:] @ h[n1:, n1:].t() + n * torch.eye(n - n1, dtype=torch.float64, device="cuda"))
    a[n1:, :n1] = h[n1:, :n1]
    a[:n1, n1:] = h[n1:, :n1].t()
    w = a.t().clone().t()  # column major
    assert F.ldlt_factor_in_place(w) == 0
    F.synchronize()
    L = torch.tril(w, -1) + torch.eye(n, dtype=torch.float64, device="cuda")
    D = torch.diagonal(w)
    err = torch.tril(L @ (D[:, None] * L.t()) - a).abs().max().item()
    assert err <= 32 * n * 2.3e-16 * a.abs().max().item()


def test_llt_lookahead_on_a_view_with_reversed_rows_and_columns(oracle):
    """a Cholesky with n >= 2048 (the look-ahead driver, whose merged trailing update asks the GEMM for a tile skip) on a device
    view with NEGATIVE strides: the buffer-addressed loaders cannot express it, gemm_dev splits the merged update into the two
    plain products of the pointer-addressed kernel instead of refusing"""
    import ctypes as C
    F = init_gpu()
    n = 2304
    rng = np.random.default_rng(77)
    a = spd(rng, n)
    # store the matrix with rows AND columns reversed; the view (i, j) -> stored (n-1-i, n-1-j) is the matrix itself
    d = to_dev(np.asfortranarray(a[::-1, ::-1]))
    v = F.MatMut(d.data_ptr() + ((n - 1) * d.stride(0) + (n - 1) * d.stride(1)) * 8, n, n, -d.stride(0), -d.stride(1))
    reg = F.LltRegularization()
    st = F.lib().libfaer_v0_23_llt_factor_in_place_f64(v, reg, F.PAR_SEQ, F.MemAlloc(), F.lib().libfaer_v0_23_LltParams_f64())
    assert st.tag == 0
    got = to_host(d)[::-1, ::-1]
    ref = a.copy(order="F")
    assert oracle.llt_in_place(ref) == ("ok", 0)
    il = np.tril_indices(n)
    assert np.abs(got[il] - ref[il]).max() <= 64 * n * 2.3e-16 * np.abs(ref[il]).max()
