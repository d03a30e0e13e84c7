"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Loads oracle/libfaer_oracle.so (built by oracle/Makefile from
faer_oracle.c) and exposes numpy-level helpers.  All matrices are numpy
arrays of dtype float64/float32 with arbitrary (element) strides; every
function works in place like its faer counterpart.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  Parity status: see faer_oracle.c (bit level UNPINNED,
tolerance level pinned).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BLOCK = {
    "rect": 0,
    "lower": 1,
    "upper": 2,
    "strict_lower": 3,
    "strict_upper": 4,
    "unit_lower": 5,
    "unit_upper": 6,
}


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libfaer_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("faer_oracle.c", "faer_oracle_impl.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfaer_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        for suf, ct in (("f64", C.c_double), ("f32", C.c_float)):
            getattr(_LIB, f"oracle_norm_l2_{suf}").restype = ct
            for name in ("oracle_llt_in_place", "oracle_ldlt_in_place", "oracle_lu_in_place", "oracle_qr_in_place",
                         "oracle_qr_recommended_block_size", "oracle_full_piv_lu_in_place", "oracle_colpiv_qr_in_place"):
                getattr(_LIB, f"{name}_{suf}").restype = C.c_long
    return _LIB


def _suf(a: np.ndarray):
    if a.dtype == np.float64:
        return "f64", C.c_double
    if a.dtype == np.float32:
        return "f32", C.c_float
    raise TypeError(a.dtype)


def _p(a: np.ndarray):
    return C.c_void_p(a.ctypes.data)


def _st(a: np.ndarray):
    assert a.ndim == 2
    it = a.itemsize
    return C.c_long(a.strides[0] // it), C.c_long(a.strides[1] // it)


def matmul(c, a, b, alpha=1.0, accum_add=False):
    suf, ct = _suf(c)
    m, n = c.shape
    k = a.shape[1]
    assert a.shape == (m, k) and b.shape == (k, n)
    getattr(lib(), f"oracle_matmul_{suf}")(_p(c), C.c_long(m), C.c_long(n), *_st(c), C.c_int(int(accum_add)),
                                          _p(a), C.c_long(k), *_st(a), _p(b), *_st(b), ct(alpha))
    return c


def matmul_triangular(c, c_s, a, a_s, b, b_s, alpha=1.0, accum_add=False):
    suf, ct = _suf(c)
    m, n = c.shape
    k = a.shape[1]
    getattr(lib(), f"oracle_matmul_triangular_{suf}")(
        _p(c), C.c_long(m), C.c_long(n), *_st(c), C.c_int(BLOCK[c_s]), C.c_int(int(accum_add)),
        _p(a), C.c_long(k), *_st(a), C.c_int(BLOCK[a_s]), _p(b), *_st(b), C.c_int(BLOCK[b_s]), ct(alpha))
    return c


def trsm(t, x, upper=False, unit=False):
    """x <- op(t)^-1 x (left side, in place)."""
    suf, _ = _suf(x)
    n = t.shape[0]
    assert t.shape == (n, n) and x.shape[0] == n
    getattr(lib(), f"oracle_trsm_{suf}")(_p(t), C.c_long(n), *_st(t), C.c_int(int(upper)), C.c_int(int(unit)),
                                        _p(x), C.c_long(x.shape[1]), *_st(x))
    return x


def llt_in_place(a, reg_delta=0.0, reg_eps=0.0, recursion_threshold=64, block_size=128):
    """returns ('ok', count) or ('non_positive_pivot', index)"""
    suf, ct = _suf(a)
    n = a.shape[0]
    r = getattr(lib(), f"oracle_llt_in_place_{suf}")(_p(a), C.c_long(n), *_st(a), ct(reg_delta), ct(reg_eps),
                                                    C.c_long(recursion_threshold), C.c_long(block_size))
    return ("ok", r) if r >= 0 else ("non_positive_pivot", -r - 1)


def ldlt_in_place(a, reg_delta=0.0, reg_eps=0.0, signs=None, recursion_threshold=64, block_size=128):
    """cholesky/ldlt/factor.rs:742-800: unit lower L below the diagonal, D on it.
    returns ('ok', count) or ('zero_pivot', index)"""
    suf, ct = _suf(a)
    n = a.shape[0]
    sp = None
    if signs is not None:
        signs = np.ascontiguousarray(signs, dtype=np.int8)
        sp = signs.ctypes.data_as(C.c_void_p)
    r = getattr(lib(), f"oracle_ldlt_in_place_{suf}")(_p(a), C.c_long(n), *_st(a), ct(reg_delta), ct(reg_eps), sp,
                                                     C.c_long(recursion_threshold), C.c_long(block_size))
    return ("ok", r) if r >= 0 else ("zero_pivot", -r - 1)


def lu_in_place(a, recursion_threshold=16):
    """returns (perm, perm_inv, transposition_count)"""
    suf, _ = _suf(a)
    m, n = a.shape
    perm = np.zeros(m, dtype=np.int64)
    perm_inv = np.zeros(m, dtype=np.int64)
    nt = getattr(lib(), f"oracle_lu_in_place_{suf}")(_p(a), C.c_long(m), C.c_long(n), *_st(a), _p(perm),
                                                    _p(perm_inv), C.c_long(recursion_threshold))
    return perm, perm_inv, nt


def full_piv_lu_in_place(a):
    """lu/full_pivoting/factor.rs:452-525.  returns (row_perm, row_perm_inv, col_perm, col_perm_inv, transposition_count);
    on exit A[row_perm][:, col_perm] == L U with L unit lower trapezoidal, U upper trapezoidal packed in `a`"""
    suf, _ = _suf(a)
    m, n = a.shape
    rp, rpi = np.zeros(m, dtype=np.int64), np.zeros(m, dtype=np.int64)
    cp, cpi = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    nt = getattr(lib(), f"oracle_full_piv_lu_in_place_{suf}")(_p(a), C.c_long(m), C.c_long(n), *_st(a), _p(rp), _p(rpi), _p(cp), _p(cpi))
    return rp, rpi, cp, cpi, nt


def colpiv_qr_in_place(a, h):
    """qr/col_pivoting/factor.rs:356-395.  a: m x n (QR of A[:, col_perm] packed like qr_in_place), h: block_size x
    min(m, n) Householder factors.  returns (col_perm, col_perm_inv, transposition_count)"""
    suf, _ = _suf(a)
    m, n = a.shape
    cp, cpi = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    nt = getattr(lib(), f"oracle_colpiv_qr_in_place_{suf}")(_p(a), C.c_long(m), C.c_long(n), *_st(a), _p(h), C.c_long(h.shape[0]), *_st(h), _p(cp),
                                                          _p(cpi))
    return cp, cpi, nt


def qr_recommended_block_size(m, n, dtype=np.float64):
    suf = "f64" if np.dtype(dtype) == np.float64 else "f32"
    return getattr(lib(), f"oracle_qr_recommended_block_size_{suf}")(C.c_long(m), C.c_long(n))


def qr_in_place(a, h, blocking_threshold=48 * 48):
    """h: block_size x min(m,n) householder factor (Q_coeff). returns rank"""
    suf, _ = _suf(a)
    m, n = a.shape
    bs = h.shape[0]
    assert h.shape[1] == min(m, n) and h.dtype == a.dtype
    return getattr(lib(), f"oracle_qr_in_place_{suf}")(_p(a), C.c_long(m), C.c_long(n), *_st(a), _p(h),
                                                      C.c_long(bs), *_st(h), C.c_long(blocking_threshold))


def tridiag_in_place(a, h):
    """evd/tridiag.rs:274: a (n x n, self-adjoint, lower triangle used) -> T on the diagonal / subdiagonal, reflectors
    below; h: block_size x (n - 1) block Householder factors"""
    suf, _ = _suf(a)
    n = a.shape[0]
    assert a.shape == (n, n) and h.shape[1] == max(n - 1, 0) and h.dtype == a.dtype
    getattr(lib(), f"oracle_tridiag_in_place_{suf}")(_p(a), C.c_long(n), *_st(a), _p(h), C.c_long(h.shape[0]), *_st(h))
    return a, h


def hessenberg_in_place(a, h):
    """evd/hessenberg.rs:549 (the unblocked variant :230): a (n x n) -> upper Hessenberg H with a = Q H Q^H, reflectors
    below the subdiagonal; h: block_size x (n - 1) block Householder factors"""
    suf, _ = _suf(a)
    n = a.shape[0]
    assert a.shape == (n, n) and h.shape[1] == max(n - 1, 0) and h.dtype == a.dtype
    getattr(lib(), f"oracle_hessenberg_in_place_{suf}")(_p(a), C.c_long(n), *_st(a), _p(h), C.c_long(h.shape[0]), *_st(h))
    return a, h


def hessenberg_blocked_in_place(a, h):
    """evd/hessenberg.rs:568-736 (hessenberg_gqvdg_blocked + hessenberg_gqvdg_unblocked :409-548): the variant the
    reference runs for n * n >= blocking_threshold; same arguments and output layout as hessenberg_in_place"""
    suf, _ = _suf(a)
    n = a.shape[0]
    assert a.shape == (n, n) and h.shape[1] == max(n - 1, 0) and h.dtype == a.dtype
    getattr(lib(), f"oracle_hessenberg_blocked_in_place_{suf}")(_p(a), C.c_long(n), *_st(a), _p(h), C.c_long(h.shape[0]), *_st(h))
    return a, h


def hessenberg_reference_in_place(a, h, blocking_threshold=256 * 256):
    """evd/hessenberg.rs:549-567: the dispatch of hessenberg_in_place (HessenbergParams::auto, :17-24)"""
    n = a.shape[0]
    return hessenberg_in_place(a, h) if n * n < blocking_threshold else hessenberg_blocked_in_place(a, h)


def bidiag_in_place(a, hl, hr):
    """svd/bidiag.rs:47: a -> upper bidiagonal B on the diagonal / superdiagonal (a = U B V^H), left reflectors
    below the diagonal (block factors hl: bl x n), right reflectors right of the superdiagonal (hr: br x (n - 1))"""
    suf, _ = _suf(a)
    m, n = a.shape
    size = min(m, n)
    assert hl.shape[1] == size and hr.shape[1] == max(size - 1, 0) and hl.dtype == a.dtype == hr.dtype
    getattr(lib(), f"oracle_bidiag_in_place_{suf}")(_p(a), C.c_long(m), C.c_long(n), *_st(a), _p(hl), C.c_long(hl.shape[0]), *_st(hl), _p(hr),
                                                   C.c_long(hr.shape[0]), *_st(hr))
    return a, hl, hr


def apply_householder_sequence_left(v, h, mat, transpose):
    suf, _ = _suf(mat)
    m, n = v.shape
    getattr(lib(), f"oracle_apply_householder_sequence_left_{suf}")(
        _p(v), C.c_long(m), C.c_long(n), *_st(v), _p(h), C.c_long(h.shape[0]), *_st(h), _p(mat),
        C.c_long(mat.shape[1]), *_st(mat), C.c_int(int(transpose)))
    return mat


def norm_l2(x):
    suf, _ = _suf(x)
    assert x.ndim == 1
    return getattr(lib(), f"oracle_norm_l2_{suf}")(_p(x), C.c_long(x.shape[0]), C.c_long(x.strides[0] // x.itemsize))
