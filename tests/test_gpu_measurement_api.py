"""Round-5 entry points of the boundary that are not part of the reference's FFI (include/faer_hip.h): the kernel-class
profile, the hand-off probe and the lent copy of the LU.  GPU box only."""
import ctypes as C

import numpy as np
import pytest

from gpu_util import init_gpu, rnd, to_dev, to_host

pytestmark = pytest.mark.gpu


def test_prof_brackets_the_dominant_kernel_classes():
    import torch

    F = init_gpu()
    n = 4608  # the look-ahead LU: big-tile products and cooperative leaves
    a = torch.randn((n, n), dtype=torch.float64, device="cuda").t()
    F.prof_begin()
    F.partial_piv_lu_factor_in_place(a.clone())
    p = F.prof_end()
    assert p["lu_panel"]["launches"] == n // 64 and p["lu_panel"]["units"] == n and p["lu_panel"]["ms"] > 0
    assert p["mfma_products"]["launches"] > 0 and p["mfma_products"]["units"] > 0.5 * (2.0 * n ** 3 / 3.0)
    assert p["qr_update"]["launches"] == 0 and p["llt_leaf"]["launches"] == 0
    # outside a profile nothing is recorded; a second profile starts empty
    F.partial_piv_lu_factor_in_place(a.clone())
    F.prof_begin()
    spd = (a[:2304, :2304] @ a[:2304, :2304].t() + n * torch.eye(2304, dtype=torch.float64, device="cuda")).t().contiguous().t()
    F.llt_factor_in_place(spd)
    p = F.prof_end()
    assert p["lu_panel"]["launches"] == 0 and p["llt_leaf"]["units"] == 2304


def test_hop_probe_reports_a_plausible_latency():
    F = init_gpu()
    us = F.xwg_hop_us(500)
    assert 0.05 < us < 10.0, us


def test_lu_with_a_lent_copy_is_the_same_factorization(oracle):
    """faer_hip_partial_piv_lu_lend_copy: on an unshared GPU the copy is never used; the call consumes it and the next call
    runs without one"""
    F = init_gpu()
    rng = np.random.default_rng(3)
    m = n = 700
    a = rnd(rng, m, n)
    ref = a.copy(order="F")
    rperm, _, rnt = oracle.lu_in_place(ref)
    for lend in (True, False):
        d, cp = to_dev(a), to_dev(a)
        if lend:
            F.lib().faer_hip_partial_piv_lu_lend_copy(C.c_void_p(cp.data_ptr()), C.c_size_t(m), C.c_size_t(n), 8)
        perm, _, nt = F.partial_piv_lu_factor_in_place(d)
        assert (perm.astype(np.int64) == rperm).all() and nt == rnt
        assert np.abs(to_host(d) - ref).max() <= 4 * n * 2.3e-16 * np.linalg.cond(a[rperm]) * max(1.0, np.abs(ref).max())
        assert np.array_equal(to_host(cp), a)  # the lent copy is read-only
