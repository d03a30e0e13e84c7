"""faer_rs_amd -- thin host-side mirror of faer's low-level API over libfaer_hip.so.

Everything here is plumbing: it builds the repr(C) structs of include/faer_hip.h from numpy arrays
(host memory) or torch tensors (device memory) and calls the C-ABI through ctypes.  The compute lives in
the shared library (hand-written HIP for gfx950); there is no Python or CPU compute path, and importing
`lib()` raises if the library has not been built (run `python -c "import __graft_entry__ as g; g.build()"`).

Function names, argument order and error behaviour follow the reference:
  matmul                      faer/src/linalg/matmul/mod.rs:1617
  matmul_triangular           faer/src/linalg/matmul/triangular.rs:1193
  solve_*_triangular_in_place faer/src/linalg/triangular_solve.rs:220-419
  llt_factor_in_place         faer/src/linalg/cholesky/llt/factor.rs:67   (Result<LltInfo, LltError>)
  partial_piv_lu_factor_in_place  faer/src/linalg/lu/partial_pivoting/factor.rs:234
  qr_factor_in_place          faer/src/linalg/qr/no_pivoting/factor.rs:258
  Llt / PartialPivLu / Qr     faer/src/linalg/solvers.rs:770-1204 (high level owners)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfaer_hip.so")
if os.environ.get("FAER_HIP_LIB"):  # developer override (instrumented builds); never a different backend
    LIB_PATH = os.environ["FAER_HIP_LIB"]
_LIB = None

# ------------------------------------------------------------------ repr(C) structs (include/faer_hip.h)


class MatRef(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("nrows", C.c_size_t), ("ncols", C.c_size_t), ("row_stride", C.c_ssize_t),
                ("col_stride", C.c_ssize_t)]


class MatMut(MatRef):
    pass


class SliceMut(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("len", C.c_size_t)]


class SliceRef(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("len", C.c_size_t)]


class Par(C.Structure):
    _fields_ = [("tag", C.c_int), ("nthreads", C.c_size_t)]


class Layout(C.Structure):
    _fields_ = [("len_bytes", C.c_size_t), ("align_bytes", C.c_size_t)]


class MemAlloc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("len_bytes", C.c_size_t)]


class LltStatus(C.Structure):
    _fields_ = [("tag", C.c_int), ("value", C.c_size_t)]


class PartialPivLuStatus(C.Structure):
    _fields_ = [("tag", C.c_int), ("transposition_count", C.c_size_t)]


class QrStatus(C.Structure):
    _fields_ = [("tag", C.c_int), ("rank", C.c_size_t)]


class LltParams(C.Structure):
    _fields_ = [("recursion_threshold", C.c_size_t), ("block_size", C.c_size_t)]


class PartialPivLuParams(C.Structure):
    _fields_ = [("recursion_threshold", C.c_size_t), ("block_size", C.c_size_t), ("par_threshold", C.c_size_t)]


class QrParams(C.Structure):
    _fields_ = [("blocking_threshold", C.c_size_t), ("par_threshold", C.c_size_t)]


class LdltRegularization(C.Structure):
    """include/faer_hip.h FaerLdltRegularization {delta*, epsilon*, signs: SliceMut of i8}"""
    _fields_ = [("dynamic_regularization_delta", C.c_void_p), ("dynamic_regularization_epsilon", C.c_void_p),
                ("dynamic_regularization_signs", SliceMut)]


class VecRef(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("len", C.c_size_t), ("stride", C.c_ssize_t)]


class LdltError(Exception):
    """faer::linalg::cholesky::ldlt::factor::LdltError::ZeroPivot { index }"""

    def __init__(self, index):
        super().__init__(f"ZeroPivot {{ index: {index} }}")
        self.index = index


class LltRegularization(C.Structure):
    _fields_ = [("dynamic_regularization_delta", C.c_void_p), ("dynamic_regularization_epsilon", C.c_void_p)]


BCAST_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)


class Comm(C.Structure):
    _fields_ = [("rank", C.c_int), ("world_size", C.c_int), ("bcast", BCAST_FN), ("user", C.c_void_p)]


ACCUM_REPLACE, ACCUM_ADD = 0, 1
CONJ_NO = 0
BLOCK = {"rect": 0, "lower": 1, "upper": 2, "strict_lower": 3, "strict_upper": 4, "unit_lower": 5, "unit_upper": 6}
DST_FULL, DST_LOWER, DST_UPPER = 0, 1, 2
DTYPE_F32, DTYPE_F64 = 0, 1
PAR_SEQ = Par(0, 1)


class LltError(Exception):
    """faer::linalg::cholesky::llt::factor::LltError::NonPositivePivot { index }"""

    def __init__(self, index):
        super().__init__(f"NonPositivePivot {{ index: {index} }}")
        self.index = index


def lib():
    """Loads libfaer_hip.so.  Fails loudly when it is missing: there is no fallback path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (hipcc, gfx950). "
                           "faer_rs_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for suf in ("f64", "f32"):
        getattr(L, f"libfaer_v0_23_llt_factor_in_place_{suf}").restype = LltStatus
        getattr(L, f"libfaer_v0_23_qr_factor_in_place_{suf}").restype = QrStatus
        getattr(L, f"libfaer_v0_23_qr_recommended_block_size_{suf}").restype = C.c_size_t
        getattr(L, f"libfaer_v0_23_LltParams_{suf}").restype = LltParams
        getattr(L, f"libfaer_v0_23_LdltParams_{suf}").restype = LltParams
        getattr(L, f"libfaer_v0_23_ldlt_factor_in_place_{suf}").restype = LltStatus
        getattr(L, f"libfaer_v0_23_ldlt_factor_in_place_scratch_{suf}").restype = Layout
        getattr(L, f"libfaer_v0_23_ldlt_solve_in_place_scratch_{suf}").restype = Layout
        getattr(L, f"libfaer_v0_23_PartialPivLuParams_{suf}").restype = PartialPivLuParams
        getattr(L, f"libfaer_v0_23_QrParams_{suf}").restype = QrParams
        for it in ("u32", "u64"):
            getattr(L, f"libfaer_v0_23_partial_piv_lu_factor_in_place_{it}_{suf}").restype = PartialPivLuStatus
            getattr(L, f"libfaer_v0_23_partial_piv_lu_factor_in_place_scratch_{it}_{suf}").restype = Layout
        for name in ("llt_factor_in_place_scratch", "llt_solve_in_place_scratch", "qr_factor_in_place_scratch",
                     "apply_householder_on_the_left_scratch", "apply_householder_transpose_on_the_left_scratch"):
            getattr(L, f"libfaer_v0_23_{name}_{suf}").restype = Layout
        getattr(L, f"faer_hip_dist_partial_piv_lu_{suf}").restype = PartialPivLuStatus
    L.libfaer_v0_23_get_global_par.restype = Par
    L.faer_hip_version.restype = C.c_char_p
    L.faer_hip_device_count.restype = C.c_int
    L.faer_hip_get_stream.restype = C.c_void_p
    L.faer_hip_malloc.restype = C.c_void_p
    L.faer_hip_time_gemm_ms.restype = C.c_double
    L.faer_hip_mfma_peak_tflops.restype = C.c_double
    if hasattr(L, "faer_hip_xwg_hop_us"):  # (absent from older builds loaded through FAER_HIP_LIB for A/B runs)
        L.faer_hip_xwg_hop_us.restype = C.c_double
    L.faer_hip_dist_local_ncols.restype = C.c_size_t
    L.faer_hip_dist_panel_ws_scalars.restype = C.c_size_t
    _LIB = L
    import atexit

    atexit.register(L.faer_hip_shutdown)
    return L


# ------------------------------------------------------------------ operand marshalling
def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _dtype_suffix(x):
    if _is_torch(x):
        import torch

        if x.dtype == torch.float64:
            return "f64", C.c_double, 8
        if x.dtype == torch.float32:
            return "f32", C.c_float, 4
    else:
        if x.dtype == np.float64:
            return "f64", C.c_double, 8
        if x.dtype == np.float32:
            return "f32", C.c_float, 4
    raise TypeError(f"faer_rs_amd supports f32/f64 only, got {x.dtype}")


def _mat(x, cls=MatRef):
    """2-D numpy array / torch tensor -> MatRef/MatMut (strides in elements)."""
    if _is_torch(x):
        assert x.dim() == 2
        return cls(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), x.stride(1))
    assert x.ndim == 2
    it = x.itemsize
    return cls(x.ctypes.data, x.shape[0], x.shape[1], x.strides[0] // it, x.strides[1] // it)


def use_torch_stream():
    """Enqueue on torch's current HIP stream (torch is plumbing here: memory + streams)."""
    import torch

    lib().faer_hip_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))


def synchronize():
    lib().faer_hip_synchronize()


# ------------------------------------------------------------------ low level API (faer::linalg)
def matmul(c, accum, a, b, alpha=1.0, par=PAR_SEQ):
    suf, ct, _ = _dtype_suffix(c)
    al = ct(alpha)
    getattr(lib(), f"libfaer_v0_23_matmul_{suf}")(_mat(c, MatMut), C.c_int(accum), _mat(a), _mat(b), C.byref(al), par)
    return c


def matmul_triangular(c, c_block, accum, a, a_block, b, b_block, alpha=1.0, par=PAR_SEQ):
    suf, ct, _ = _dtype_suffix(c)
    al = ct(alpha)
    getattr(lib(), f"libfaer_v0_23_matmul_triangular_{suf}")(_mat(c, MatMut), C.c_int(BLOCK[c_block]), C.c_int(accum),
                                                           _mat(a), C.c_int(BLOCK[a_block]), _mat(b),
                                                           C.c_int(BLOCK[b_block]), C.byref(al), par)
    return c


def gemm(dst, dst_kind, accum, lhs, rhs, alpha=1.0, row_idx=None, col_idx=None, diag=None):
    """inner boundary: the private_gemm_x86::gemm call of faer (include/faer_hip.h section 1)."""
    suf, ct, _ = _dtype_suffix(dst)
    al = ct(alpha)
    m, k = lhs.shape
    n = rhs.shape[1]
    d = _mat(dst, MatMut)
    a, b = _mat(lhs), _mat(rhs)

    def idx(v):
        if v is None:
            return None, 1
        if _is_torch(v):  # device index array: int32 / int64 storage read as u32 / u64 (indices are non-negative)
            import torch

            assert v.dtype in (torch.int32, torch.int64) and v.is_contiguous()
            return C.c_void_p(v.data_ptr()), (1 if v.dtype == torch.int64 else 0)
        assert v.dtype in (np.uint32, np.uint64)
        return C.c_void_p(v.ctypes.data), (1 if v.dtype == np.uint64 else 0)

    ri, it1 = idx(row_idx)
    ci, it2 = idx(col_idx)
    itype = it1 if row_idx is not None else it2
    dg, dgs = (None, 0)
    if diag is not None:
        dg = C.c_void_p(diag.data_ptr() if _is_torch(diag) else diag.ctypes.data)
        dgs = diag.stride(0) if _is_torch(diag) else diag.strides[0] // diag.itemsize
    lib().faer_hip_gemm(C.c_int(DTYPE_F64 if suf == "f64" else DTYPE_F32), C.c_int(itype), C.c_size_t(m), C.c_size_t(n),
                        C.c_size_t(k), C.c_void_p(d.ptr), C.c_ssize_t(d.row_stride), C.c_ssize_t(d.col_stride), ri, ci,
                        C.c_int(dst_kind), C.c_int(accum), C.c_void_p(a.ptr), C.c_ssize_t(a.row_stride),
                        C.c_ssize_t(a.col_stride), C.c_bool(False), dg, C.c_ssize_t(dgs), C.c_void_p(b.ptr),
                        C.c_ssize_t(b.row_stride), C.c_ssize_t(b.col_stride), C.c_bool(False), C.byref(al), C.c_size_t(0))
    return dst


def _trsm(name, t, rhs, par):
    suf, _, _ = _dtype_suffix(rhs)
    getattr(lib(), f"libfaer_v0_23_{name}_{suf}")(_mat(t), C.c_int(CONJ_NO), _mat(rhs, MatMut), par)
    return rhs


def solve_lower_triangular_in_place(l, rhs, par=PAR_SEQ):
    return _trsm("solve_triangular_lower_in_place", l, rhs, par)


def solve_upper_triangular_in_place(u, rhs, par=PAR_SEQ):
    return _trsm("solve_triangular_upper_in_place", u, rhs, par)


def solve_unit_lower_triangular_in_place(l, rhs, par=PAR_SEQ):
    return _trsm("solve_unit_triangular_lower_in_place", l, rhs, par)


def solve_unit_upper_triangular_in_place(u, rhs, par=PAR_SEQ):
    return _trsm("solve_unit_triangular_upper_in_place", u, rhs, par)


def llt_factor_in_place(a, regularization=(0.0, 0.0), par=PAR_SEQ):
    """returns dynamic_regularization_count; raises LltError(index) like Err(NonPositivePivot{index})"""
    suf, ct, _ = _dtype_suffix(a)
    delta, eps = ct(regularization[0]), ct(regularization[1])
    reg = LltRegularization(C.cast(C.pointer(delta), C.c_void_p), C.cast(C.pointer(eps), C.c_void_p))
    L = lib()
    params = getattr(L, f"libfaer_v0_23_LltParams_{suf}")()
    st = getattr(L, f"libfaer_v0_23_llt_factor_in_place_{suf}")(_mat(a, MatMut), reg, par, MemAlloc(None, 0), params)
    if st.tag == 0:
        return st.value
    if st.tag == 1:
        raise LltError(st.value)
    raise RuntimeError("LltStatus::Unknown")


def ldlt_factor_in_place(a, regularization=(0.0, 0.0), signs=None, par=PAR_SEQ):
    """cholesky/ldlt/factor.rs:742-800: unit lower L strictly below the diagonal of `a`, D on it.
    returns dynamic_regularization_count; raises LdltError(index) like Err(ZeroPivot{index})"""
    suf, ct, _ = _dtype_suffix(a)
    delta, eps = ct(regularization[0]), ct(regularization[1])
    sl = SliceMut(None, 0)
    if signs is not None:
        signs = np.ascontiguousarray(signs, dtype=np.int8)
        sl = SliceMut(signs.ctypes.data, signs.shape[0])
    reg = LdltRegularization(C.cast(C.pointer(delta), C.c_void_p), C.cast(C.pointer(eps), C.c_void_p), sl)
    L = lib()
    params = getattr(L, f"libfaer_v0_23_LdltParams_{suf}")()
    st = getattr(L, f"libfaer_v0_23_ldlt_factor_in_place_{suf}")(_mat(a, MatMut), reg, par, MemAlloc(None, 0), params)
    if st.tag == 0:
        return st.value
    if st.tag == 1:
        raise LdltError(st.value)
    raise RuntimeError("LdltStatus::Unknown")


def ldlt_solve_in_place(ld, rhs, par=PAR_SEQ):
    """cholesky/ldlt/solve.rs:12-50 with L and D packed as ldlt_factor_in_place leaves them"""
    suf, _, _ = _dtype_suffix(ld)
    n = ld.shape[0]
    if _is_torch(ld):
        d = VecRef(ld.data_ptr(), n, ld.stride(0) + ld.stride(1))
    else:
        d = VecRef(ld.ctypes.data, n, (ld.strides[0] + ld.strides[1]) // ld.itemsize)
    getattr(lib(), f"libfaer_v0_23_ldlt_solve_in_place_{suf}")(_mat(ld), d, C.c_int(0), _mat(rhs, MatMut), par, MemAlloc(None, 0))
    return rhs


def llt_solve_in_place(l, rhs, par=PAR_SEQ):
    suf, _, _ = _dtype_suffix(rhs)
    getattr(lib(), f"libfaer_v0_23_llt_solve_in_place_{suf}")(_mat(l), C.c_int(CONJ_NO), _mat(rhs, MatMut), par,
                                                            MemAlloc(None, 0))
    return rhs


def partial_piv_lu_factor_in_place(a, index_dtype=np.uint64, par=PAR_SEQ):
    """returns (perm_fwd, perm_bwd, transposition_count); perm_fwd[i] = source row of row i of P*A"""
    suf, _, _ = _dtype_suffix(a)
    m = a.shape[0]
    it = "u64" if np.dtype(index_dtype) == np.uint64 else "u32"
    fwd = np.zeros(m, dtype=index_dtype)
    bwd = np.zeros(m, dtype=index_dtype)
    L = lib()
    params = getattr(L, f"libfaer_v0_23_PartialPivLuParams_{suf}")()
    st = getattr(L, f"libfaer_v0_23_partial_piv_lu_factor_in_place_{it}_{suf}")(
        _mat(a, MatMut), SliceMut(fwd.ctypes.data, m), SliceMut(bwd.ctypes.data, m), par, MemAlloc(None, 0), params)
    if st.tag != 0:
        raise RuntimeError("PartialPivLuStatus::Unknown")
    return fwd, bwd, st.transposition_count


class FullPivLuParams(C.Structure):
    _fields_ = [("par_threshold", C.c_size_t)]


def full_piv_lu_factor_in_place(a, index_dtype=np.uint64, par=PAR_SEQ):
    """lu/full_pivoting/factor.rs:452-525.  returns (row_fwd, row_bwd, col_fwd, col_bwd, transposition_count):
    A[row_fwd][:, col_fwd] == L U with L unit lower trapezoidal and U upper trapezoidal packed in `a`"""
    suf, _, _ = _dtype_suffix(a)
    m, n = a.shape
    it = "u64" if np.dtype(index_dtype) == np.uint64 else "u32"
    rf, rb = np.zeros(m, dtype=index_dtype), np.zeros(m, dtype=index_dtype)
    cf, cb = np.zeros(n, dtype=index_dtype), np.zeros(n, dtype=index_dtype)
    L = lib()
    pf = getattr(L, f"libfaer_v0_23_FullPivLuParams_{suf}")
    pf.restype = FullPivLuParams
    fn = getattr(L, f"libfaer_v0_23_full_piv_lu_factor_in_place_{it}_{suf}")
    fn.restype = PartialPivLuStatus  # same layout: {tag, transposition_count}
    st = fn(_mat(a, MatMut), SliceMut(rf.ctypes.data, m), SliceMut(rb.ctypes.data, m), SliceMut(cf.ctypes.data, n), SliceMut(cb.ctypes.data, n),
            par, MemAlloc(None, 0), pf())
    if st.tag != 0:
        raise RuntimeError("FullPivLuStatus::Unknown")
    return rf, rb, cf, cb, st.transposition_count


def full_piv_lu_solve_in_place(lu, row_fwd, row_bwd, col_fwd, col_bwd, rhs, transpose=False, par=PAR_SEQ):
    """lu/full_pivoting/solve.rs: rhs <- A^-1 rhs (or A^-T rhs)"""
    suf, _, _ = _dtype_suffix(lu)
    it = "u64" if np.dtype(row_fwd.dtype) == np.uint64 else "u32"
    name = "full_piv_lu_solve_transpose_in_place" if transpose else "full_piv_lu_solve_in_place"
    n = lu.shape[0]
    getattr(lib(), f"libfaer_v0_23_{name}_{it}_{suf}")(
        _mat(lu), _mat(lu), C.c_int(0), SliceRef(row_fwd.ctypes.data, n), SliceRef(row_bwd.ctypes.data, n), SliceRef(col_fwd.ctypes.data, n),
        SliceRef(col_bwd.ctypes.data, n), _mat(rhs, MatMut), par, MemAlloc(None, 0))
    return rhs


class ColPivQrParams(C.Structure):
    _fields_ = [("blocking_threshold", C.c_size_t), ("par_threshold", C.c_size_t)]


def colpiv_qr_factor_in_place(a, q_coeff, index_dtype=np.uint64, par=PAR_SEQ):
    """qr/col_pivoting/factor.rs:356-395.  returns (col_fwd, col_bwd, transposition_count): A[:, col_fwd] == Q R with the
    Householder basis below the diagonal of `a`, R on and above it, block factors in q_coeff (block_size x min(m, n))"""
    suf, _, _ = _dtype_suffix(a)
    n = a.shape[1]
    it = "u64" if np.dtype(index_dtype) == np.uint64 else "u32"
    cf, cb = np.zeros(n, dtype=index_dtype), np.zeros(n, dtype=index_dtype)
    L = lib()
    pf = getattr(L, f"libfaer_v0_23_ColPivQrParams_{suf}")
    pf.restype = ColPivQrParams
    fn = getattr(L, f"libfaer_v0_23_colpiv_qr_factor_in_place_{it}_{suf}")
    fn.restype = PartialPivLuStatus  # same layout: {tag, transposition_count}
    st = fn(_mat(a, MatMut), _mat(q_coeff, MatMut), SliceMut(cf.ctypes.data, n), SliceMut(cb.ctypes.data, n), par, MemAlloc(None, 0), pf())
    if st.tag != 0:
        raise RuntimeError("ColPivQrStatus::Unknown")
    return cf, cb, st.transposition_count


def tridiag_in_place(a, householder):
    """faer::linalg::evd::tridiag::tridiag_in_place (evd/tridiag.rs:274): `a` (n x n, self-adjoint, only the lower triangle
    is used) -> the tridiagonal T on its diagonal / subdiagonal (a = Q T Q^H), the reflectors of Q below the subdiagonal,
    their block factors in `householder` (block_size x (n - 1))"""
    suf, _, _ = _dtype_suffix(a)
    fn = getattr(lib(), f"faer_hip_tridiag_in_place_{suf}")
    fn.restype = None
    fn(_mat(a, MatMut), _mat(householder, MatMut))
    return a, householder


def hessenberg_in_place(a, householder):
    """faer::linalg::evd::hessenberg::hessenberg_in_place (evd/hessenberg.rs:549): `a` (n x n) -> the upper Hessenberg H on and
    above its subdiagonal (a = Q H Q^H), the reflectors of Q below it, their block factors in `householder`
    (block_size x (n - 1))"""
    suf, _, _ = _dtype_suffix(a)
    fn = getattr(lib(), f"faer_hip_hessenberg_in_place_{suf}")
    fn.restype = None
    fn(_mat(a, MatMut), _mat(householder, MatMut))
    return a, householder


def bidiag_in_place(a, h_left, h_right):
    """faer::linalg::svd::bidiag::bidiag_in_place (svd/bidiag.rs:47): `a` -> the upper bidiagonal B on its diagonal /
    superdiagonal (a = U B V^H for nrows >= ncols, the shape the reference's SVD uses), left reflectors below the diagonal
    (block factors h_left: bl x min(m, n)), right reflectors right of the superdiagonal (h_right: br x (min(m, n) - 1))"""
    suf, _, _ = _dtype_suffix(a)
    fn = getattr(lib(), f"faer_hip_bidiag_in_place_{suf}")
    fn.restype = None
    fn(_mat(a, MatMut), _mat(h_left, MatMut), _mat(h_right, MatMut))
    return a, h_left, h_right


def colpiv_qr_solve_in_place(qr, q_coeff, col_fwd, col_bwd, rhs, mode="lstsq", par=PAR_SEQ):
    """qr/col_pivoting/solve.rs; mode: 'lstsq' (m >= n, solution in the first n rows), 'solve' (square), 'transpose'"""
    suf, _, _ = _dtype_suffix(qr)
    it = "u64" if np.dtype(col_fwd.dtype) == np.uint64 else "u32"
    name = {"lstsq": "colpiv_qr_solve_lstsq_in_place", "solve": "colpiv_qr_solve_in_place", "transpose": "colpiv_qr_solve_transpose_in_place"}[mode]
    m, n = qr.shape
    size = min(m, n)
    getattr(lib(), f"libfaer_v0_23_{name}_{it}_{suf}")(_mat(qr[:, :size]), _mat(q_coeff), _mat(qr[:size, :]), C.c_int(0),
                                                     SliceRef(col_fwd.ctypes.data, n), SliceRef(col_bwd.ctypes.data, n), _mat(rhs, MatMut), par,
                                                     MemAlloc(None, 0))
    return rhs


def _it(p):
    return "u64" if np.dtype(p.dtype) == np.uint64 else "u32"


def full_piv_lu_reconstruct(out, lu, row_fwd, row_bwd, col_fwd, col_bwd, par=PAR_SEQ):
    """lu/full_pivoting/reconstruct.rs: out <- P^-1 L U Q^-1"""
    suf, _, _ = _dtype_suffix(lu)
    m, n = lu.shape
    getattr(lib(), f"libfaer_v0_23_full_piv_lu_reconstruct_{_it(row_fwd)}_{suf}")(
        _mat(out, MatMut), _mat(lu), _mat(lu), SliceRef(row_fwd.ctypes.data, m), SliceRef(row_bwd.ctypes.data, m), SliceRef(col_fwd.ctypes.data, n),
        SliceRef(col_bwd.ctypes.data, n), par, MemAlloc(None, 0))
    return out


def full_piv_lu_inverse(out, lu, row_fwd, row_bwd, col_fwd, col_bwd, par=PAR_SEQ):
    """lu/full_pivoting/inverse.rs: out <- A^-1"""
    suf, _, _ = _dtype_suffix(lu)
    n = lu.shape[0]
    getattr(lib(), f"libfaer_v0_23_full_piv_lu_inverse_{_it(row_fwd)}_{suf}")(
        _mat(out, MatMut), _mat(lu), _mat(lu), SliceRef(row_fwd.ctypes.data, n), SliceRef(row_bwd.ctypes.data, n), SliceRef(col_fwd.ctypes.data, n),
        SliceRef(col_bwd.ctypes.data, n), par, MemAlloc(None, 0))
    return out


def colpiv_qr_reconstruct(out, qr, q_coeff, col_fwd, col_bwd, par=PAR_SEQ):
    """qr/col_pivoting/reconstruct.rs: out <- Q R P^-1"""
    suf, _, _ = _dtype_suffix(qr)
    m, n = qr.shape
    size = min(m, n)
    getattr(lib(), f"libfaer_v0_23_colpiv_qr_reconstruct_{_it(col_fwd)}_{suf}")(
        _mat(out, MatMut), _mat(qr[:, :size]), _mat(q_coeff), _mat(qr[:size, :]), SliceRef(col_fwd.ctypes.data, n), SliceRef(col_bwd.ctypes.data, n), par,
        MemAlloc(None, 0))
    return out


def colpiv_qr_inverse(out, qr, q_coeff, col_fwd, col_bwd, par=PAR_SEQ):
    """qr/col_pivoting/inverse.rs: out <- A^-1 (square)"""
    suf, _, _ = _dtype_suffix(qr)
    n = qr.shape[0]
    getattr(lib(), f"libfaer_v0_23_colpiv_qr_inverse_{_it(col_fwd)}_{suf}")(
        _mat(out, MatMut), _mat(qr), _mat(q_coeff), _mat(qr), SliceRef(col_fwd.ctypes.data, n), SliceRef(col_bwd.ctypes.data, n), par, MemAlloc(None, 0))
    return out


BcastFn = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)


IbcastFn = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int)
WaitFn = C.CFUNCTYPE(None, C.c_void_p, C.c_int)


class HipComm(C.Structure):
    """include/faer_hip.h FaerHipComm {rank, world_size, bcast, user, ibcast, wait}"""
    _fields_ = [("rank", C.c_int), ("world_size", C.c_int), ("bcast", BcastFn), ("user", C.c_void_p), ("ibcast", IbcastFn),
                ("wait", WaitFn)]


def _make_comm(rank, world_size, panel_ws, bcast, ibcast=None):
    """FaerHipComm over a python transport.  bcast(tensor_view_of_the_device_buffer, root) is blocking (stream ordered);
    ibcast(view, root) may return a handle with .wait() (e.g. torch.distributed.broadcast(..., async_op=True)):
    the library then overlaps the transfer of the next panel with the trailing updates.  Returns (comm, keepalive)."""
    import torch

    ws_bytes = panel_ws.view(torch.uint8)
    base = panel_ws.data_ptr()
    pending = {}

    def cb(user, buf, nbytes, root):
        off = buf - base
        bcast(ws_bytes[off:off + nbytes], root)

    def icb(user, buf, nbytes, root, slot):
        off = buf - base
        pending[slot] = ibcast(ws_bytes[off:off + nbytes], root)

    def wcb(user, slot):
        h = pending.pop(slot, None)
        if h is not None:
            h.wait()

    fns = (BcastFn(cb), IbcastFn(icb) if ibcast else IbcastFn(), WaitFn(wcb) if ibcast else WaitFn())
    return HipComm(int(rank), int(world_size), fns[0], None, fns[1], fns[2]), fns


class RcclTransport:
    """Built-in transport of the distributed drivers (include/faer_hip.h, csrc/rccl_transport.hip): ncclBroadcast on a
    dedicated stream, event-ordered against the caller's stream; no Python callback in the factorization loop.

    Collective: every rank constructs it with the same 128-byte id (rank 0: RcclTransport.unique_id(), shipped to the
    others by the application, e.g. torch.distributed.broadcast of a uint8 tensor)."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        rc = lib().faer_hip_rccl_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"faer_hip_rccl_unique_id failed ({rc}): librccl.so not loadable?")
        return bytes(buf)

    def __init__(self, unique_id, rank, world_size):
        L = lib()
        L.faer_hip_rccl_create.restype = C.c_void_p
        L.faer_hip_rccl_comm.restype = HipComm
        L.faer_hip_rccl_comm.argtypes = [C.c_void_p]
        L.faer_hip_rccl_destroy.argtypes = [C.c_void_p]
        assert len(unique_id) == 128
        self.handle = L.faer_hip_rccl_create((C.c_ubyte * 128).from_buffer_copy(unique_id), int(rank), int(world_size))
        self.comm = L.faer_hip_rccl_comm(C.c_void_p(self.handle))

    def stats(self):
        """{ranks the communicator reports (ncclCommCount), broadcasts, bytes, device ms inside ncclBroadcast} since the last call"""
        out = (C.c_double * 4)()
        L = lib()
        L.faer_hip_rccl_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.faer_hip_rccl_stats(C.c_void_p(self.handle), out)
        return {"ncclCommCount": int(out[0]), "broadcasts": int(out[1]), "bytes": float(out[2]), "bcast_device_ms": float(out[3])}

    def close(self):
        if self.handle:
            lib().faer_hip_rccl_destroy(C.c_void_p(self.handle))
            self.handle = None


class LoopbackGroup:
    """Loop-back transport (csrc/loop_transport.hip, TEST infrastructure): the ranks of a distributed factorization as threads of this
    process on one GPU.  `group.rank(r)` must be called on the thread that runs rank r; the returned object is a `transport=` argument
    of dist_partial_piv_lu / dist_llt (it has the `.comm` of an RcclTransport)."""

    class Rank:
        def __init__(self, group, rank):
            L = lib()
            L.faer_hip_loopback_rank_create.restype = C.c_void_p
            L.faer_hip_loopback_rank_create.argtypes = [C.c_void_p, C.c_int]
            L.faer_hip_loopback_comm.restype = HipComm
            L.faer_hip_loopback_comm.argtypes = [C.c_void_p]
            self.handle = L.faer_hip_loopback_rank_create(C.c_void_p(group.handle), int(rank))
            self.comm = L.faer_hip_loopback_comm(C.c_void_p(self.handle))

        def stats(self):
            out = (C.c_double * 2)()
            L = lib()
            L.faer_hip_loopback_stats.argtypes = [C.c_void_p, C.c_void_p]
            L.faer_hip_loopback_stats(C.c_void_p(self.handle), out)
            return {"broadcasts": int(out[0]), "bytes": float(out[1])}

        def close(self):
            if self.handle:
                L = lib()
                L.faer_hip_loopback_rank_destroy.argtypes = [C.c_void_p]
                L.faer_hip_loopback_rank_destroy(C.c_void_p(self.handle))
                self.handle = None

    def __init__(self, world_size):
        L = lib()
        L.faer_hip_loopback_group_create.restype = C.c_void_p
        self.handle = L.faer_hip_loopback_group_create(C.c_int(int(world_size)))
        self.world_size = int(world_size)

    def rank(self, rank):
        return LoopbackGroup.Rank(self, rank)

    def close(self):
        if self.handle:
            L = lib()
            L.faer_hip_loopback_group_destroy.argtypes = [C.c_void_p]
            L.faer_hip_loopback_group_destroy(C.c_void_p(self.handle))
            self.handle = None


PROF_CLASSES = ("mfma_products", "lu_panel", "qr_update", "qr_gram", "qr_panel", "llt_leaf")
PROF_UNITS = ("flop", "columns", "bytes", "bytes", "launches", "columns")


def prof_begin():
    """Start bracketing the dominant kernel classes of this thread's library calls with timing events (faer_hip.h)."""
    lib().faer_hip_prof_begin()


def prof_end():
    """-> {class: {"ms", "launches", "units", "unit"}} of the launches since prof_begin (synchronises the device)."""
    out = (C.c_double * (3 * len(PROF_CLASSES)))()
    lib().faer_hip_prof_end(out)
    return {name: {"ms": out[3 * i], "launches": int(out[3 * i + 1]), "units": out[3 * i + 2], "unit": PROF_UNITS[i]}
            for i, name in enumerate(PROF_CLASSES)}


def prof_end_spans(cap=8192):
    """prof_end plus one record per profiled launch: (totals, [{"cls", "ms", "units", "d": (d0, d1, d2, d3), "stream", "t0_ms"}])"""
    out = (C.c_double * (3 * len(PROF_CLASSES)))()
    sp = (C.c_double * (8 * cap))()
    L = lib()
    L.faer_hip_prof_end_spans.restype = C.c_size_t
    n = L.faer_hip_prof_end_spans(out, sp, C.c_size_t(cap))
    tot = {name: {"ms": out[3 * i], "launches": int(out[3 * i + 1]), "units": out[3 * i + 2], "unit": PROF_UNITS[i]}
           for i, name in enumerate(PROF_CLASSES)}
    spans = []
    for i in range(n):
        r = sp[8 * i:8 * i + 8]
        d3 = int(r[6])
        spans.append({"cls": PROF_CLASSES[int(r[0])], "ms": r[1], "units": r[2], "d": (int(r[3]), int(r[4]), int(r[5]), d3 & 0xFFFF),
                      "stream": ("caller", "bulk", "panel", "side")[d3 >> 16], "t0_ms": r[7]})
    return tot, spans


def xwg_hop_us(iters=2000):
    """Idle-chip hand-off latency between two workgroups on different XCDs, microseconds per hop (< 0: timed out)."""
    return float(lib().faer_hip_xwg_hop_us(C.c_int(iters)))


def dist_last_stats():
    """the calling thread's last distributed factorization: device ms of the call, of the panels this rank owned, their count"""
    out = (C.c_double * 3)()
    lib().faer_hip_dist_last_stats(out)
    return {"total_device_ms": float(out[0]), "panel_device_ms": float(out[1]), "panels_owned": int(out[2])}


def dist_local_ncols(n, nb, rank, world_size):
    return lib().faer_hip_dist_local_ncols(C.c_size_t(n), C.c_size_t(nb), int(rank), int(world_size))


def dist_partial_piv_lu(a_local, n_global, nb, rank, world_size, bcast=None, panel_ws=None, ibcast=None, transport=None):
    """Distributed partial-pivot LU (1-D block-cyclic columns, one process per GPU; csrc/dist_lu.h).

    a_local  : this rank's block columns (nrows x dist_local_ncols, column major torch cuda tensor), factored in place
    bcast    : callable(torch uint8 tensor viewing the DEVICE broadcast buffer, root) -- the transport, e.g.
               lambda t, root: torch.distributed.broadcast(t, src=root)   (RCCL under the "nccl" backend)
    ibcast   : optional callable(view, root) -> handle with .wait() (async broadcast): enables the overlap of the next
               panel's transfer with the trailing updates
    transport: an RcclTransport instead of the two callables (direct RCCL, no Python in the loop)
    returns  : (perm_fwd, perm_bwd, transposition_count), identical on every rank"""
    import torch

    suf, _, _ = _dtype_suffix(a_local)
    m = a_local.shape[0]
    L = lib()
    dt = DTYPE_F64 if suf == "f64" else DTYPE_F32
    if panel_ws is None:
        nsc = L.faer_hip_dist_panel_ws_scalars(C.c_size_t(m), C.c_size_t(nb), C.c_int(dt))
        panel_ws = torch.empty(nsc, dtype=a_local.dtype, device=a_local.device)
    comm, _keep = (transport.comm, None) if transport is not None else _make_comm(rank, world_size, panel_ws, bcast, ibcast)
    base = panel_ws.data_ptr()
    fwd = np.zeros(m, dtype=np.uint64)
    bwd = np.zeros(m, dtype=np.uint64)
    st = getattr(L, f"faer_hip_dist_partial_piv_lu_{suf}")(_mat(a_local, MatMut), C.c_size_t(n_global), C.c_size_t(nb),
                                                            SliceMut(fwd.ctypes.data, m), SliceMut(bwd.ctypes.data, m), comm,
                                                            C.c_void_p(base))
    if st.tag != 0:
        raise RuntimeError("PartialPivLuStatus::Unknown")
    return fwd, bwd, st.transposition_count


def dist_llt(a_local, n_global, nb, rank, world_size, bcast=None, panel_ws=None, ibcast=None, regularization=(0.0, 0.0), transport=None):
    """Distributed Cholesky (lower; 1-D block-cyclic columns, one process per GPU; csrc/dist_llt.h).

    a_local : this rank's block columns at full height (n x dist_local_ncols, column major torch cuda tensor)
    returns : dynamic_regularization_count; raises LltError(index) (same index on every rank)"""
    import torch

    suf, ct, _ = _dtype_suffix(a_local)
    L = lib()
    dt = DTYPE_F64 if suf == "f64" else DTYPE_F32
    if panel_ws is None:
        L.faer_hip_dist_llt_ws_scalars.restype = C.c_size_t
        nsc = L.faer_hip_dist_llt_ws_scalars(C.c_size_t(n_global), C.c_size_t(nb), C.c_int(dt))
        panel_ws = torch.empty(nsc, dtype=a_local.dtype, device=a_local.device)
    comm, _keep = (transport.comm, None) if transport is not None else _make_comm(rank, world_size, panel_ws, bcast, ibcast)
    delta, eps = ct(regularization[0]), ct(regularization[1])
    reg = LltRegularization(C.cast(C.pointer(delta), C.c_void_p), C.cast(C.pointer(eps), C.c_void_p))
    fn = getattr(L, f"faer_hip_dist_llt_{suf}")
    fn.restype = LltStatus
    st = fn(_mat(a_local, MatMut), C.c_size_t(n_global), C.c_size_t(nb), reg, comm, C.c_void_p(panel_ws.data_ptr()))
    if st.tag == 0:
        return st.value
    if st.tag == 1:
        raise LltError(st.value)
    raise RuntimeError("LltStatus::Unknown")


def qr_recommended_block_size(nrows, ncols, dtype=np.float64):
    suf = "f64" if np.dtype(dtype) == np.float64 else "f32"
    return getattr(lib(), f"libfaer_v0_23_qr_recommended_block_size_{suf}")(C.c_size_t(nrows), C.c_size_t(ncols))


def qr_factor_in_place(a, q_coeff, par=PAR_SEQ):
    """q_coeff: block_size x min(m, n).  returns rank (QrInfo.rank)"""
    suf, _, _ = _dtype_suffix(a)
    L = lib()
    params = getattr(L, f"libfaer_v0_23_QrParams_{suf}")()
    st = getattr(L, f"libfaer_v0_23_qr_factor_in_place_{suf}")(_mat(a, MatMut), _mat(q_coeff, MatMut), par,
                                                             MemAlloc(None, 0), params)
    if st.tag != 0:
        raise RuntimeError("QrStatus::Unknown")
    return st.rank


def apply_block_householder_sequence_on_the_left_in_place(basis, factor, rhs, transpose=False, par=PAR_SEQ):
    suf, _, _ = _dtype_suffix(rhs)
    name = "apply_householder_transpose_on_the_left" if transpose else "apply_householder_on_the_left"
    getattr(lib(), f"libfaer_v0_23_{name}_{suf}")(_mat(basis), _mat(factor), C.c_int(CONJ_NO), _mat(rhs, MatMut), par,
                                                MemAlloc(None, 0))
    return rhs


def apply_block_householder_sequence_on_the_right_in_place(basis, factor, matrix, transpose=False, par=PAR_SEQ):
    """householder.rs:813-854: matrix <- matrix Q (or matrix Q^H)"""
    suf, _, _ = _dtype_suffix(matrix)
    name = "apply_householder_transpose_on_the_right" if transpose else "apply_householder_on_the_right"
    getattr(lib(), f"libfaer_v0_23_{name}_{suf}")(_mat(basis), _mat(factor), C.c_int(CONJ_NO), _mat(matrix, MatMut), par,
                                                MemAlloc(None, 0))
    return matrix


def _diag_vec(ld):
    n = ld.shape[0]
    if _is_torch(ld):
        return VecRef(ld.data_ptr(), n, ld.stride(0) + ld.stride(1))
    return VecRef(ld.ctypes.data, n, (ld.strides[0] + ld.strides[1]) // ld.itemsize)


def inverse_triangular_in_place(t_inv, t, upper=False, unit=False, par=PAR_SEQ):
    """triangular_inverse.rs: the (unit) lower / upper triangle of t_inv <- inverse of the triangle of t; the rest of
    t_inv (and, unit: its diagonal) is left untouched"""
    suf, _, _ = _dtype_suffix(t)
    name = f"inverse_{'unit_' if unit else ''}triangular_{'upper' if upper else 'lower'}_in_place"
    getattr(lib(), f"libfaer_v0_23_{name}_{suf}")(_mat(t_inv, MatMut), _mat(t), par)
    return t_inv


def llt_reconstruct(out, l, par=PAR_SEQ):
    """cholesky/llt/reconstruct.rs: lower(out) <- L L^H"""
    suf, _, _ = _dtype_suffix(l)
    getattr(lib(), f"libfaer_v0_23_llt_reconstruct_{suf}")(_mat(out, MatMut), _mat(l), par, MemAlloc(None, 0))
    return out


def llt_inverse(out, l, par=PAR_SEQ):
    """cholesky/llt/inverse.rs: lower(out) <- (L L^H)^-1"""
    suf, _, _ = _dtype_suffix(l)
    getattr(lib(), f"libfaer_v0_23_llt_inverse_{suf}")(_mat(out, MatMut), _mat(l), par, MemAlloc(None, 0))
    return out


def ldlt_reconstruct(out, ld, par=PAR_SEQ):
    """cholesky/ldlt/reconstruct.rs with L and D packed as ldlt_factor_in_place leaves them: lower(out) <- L D L^H"""
    suf, _, _ = _dtype_suffix(ld)
    getattr(lib(), f"libfaer_v0_23_ldlt_reconstruct_{suf}")(_mat(out, MatMut), _mat(ld), _diag_vec(ld), par, MemAlloc(None, 0))
    return out


def ldlt_inverse(out, ld, par=PAR_SEQ):
    """cholesky/ldlt/inverse.rs: lower(out) <- (L D L^H)^-1"""
    suf, _, _ = _dtype_suffix(ld)
    getattr(lib(), f"libfaer_v0_23_ldlt_inverse_{suf}")(_mat(out, MatMut), _mat(ld), _diag_vec(ld), par, MemAlloc(None, 0))
    return out


def partial_piv_lu_reconstruct(out, lu, perm_fwd, perm_bwd, par=PAR_SEQ):
    """lu/partial_pivoting/reconstruct.rs: out <- P^-1 L U (`lu` holds L and U packed, any m x n)"""
    suf, _, _ = _dtype_suffix(lu)
    it = "u64" if np.dtype(perm_fwd.dtype) == np.uint64 else "u32"
    m = lu.shape[0]
    getattr(lib(), f"libfaer_v0_23_partial_piv_lu_reconstruct_{it}_{suf}")(
        _mat(out, MatMut), _mat(lu), _mat(lu), SliceRef(perm_fwd.ctypes.data, m), SliceRef(perm_bwd.ctypes.data, m), par, MemAlloc(None, 0))
    return out


def partial_piv_lu_inverse(out, lu, perm_fwd, perm_bwd, par=PAR_SEQ):
    """lu/partial_pivoting/inverse.rs: out <- A^-1"""
    suf, _, _ = _dtype_suffix(lu)
    it = "u64" if np.dtype(perm_fwd.dtype) == np.uint64 else "u32"
    n = lu.shape[0]
    getattr(lib(), f"libfaer_v0_23_partial_piv_lu_inverse_{it}_{suf}")(
        _mat(out, MatMut), _mat(lu), _mat(lu), SliceRef(perm_fwd.ctypes.data, n), SliceRef(perm_bwd.ctypes.data, n), par, MemAlloc(None, 0))
    return out


def qr_reconstruct(out, qr, q_coeff, par=PAR_SEQ):
    """qr/no_pivoting/reconstruct.rs: out <- Q R (`qr` holds the Householder basis below the diagonal and R on / above it)"""
    suf, _, _ = _dtype_suffix(qr)
    m, n = qr.shape
    size = min(m, n)
    getattr(lib(), f"libfaer_v0_23_qr_reconstruct_{suf}")(_mat(out, MatMut), _mat(qr[:, :size]), _mat(q_coeff), _mat(qr[:size, :]), par,
                                                        MemAlloc(None, 0))
    return out


def qr_inverse(out, qr, q_coeff, par=PAR_SEQ):
    """qr/no_pivoting/inverse.rs: out <- A^-1 (square)"""
    suf, _, _ = _dtype_suffix(qr)
    getattr(lib(), f"libfaer_v0_23_qr_inverse_{suf}")(_mat(out, MatMut), _mat(qr), _mat(q_coeff), _mat(qr), par, MemAlloc(None, 0))
    return out


# ------------------------------------------------------------------ high level owners (faer/src/linalg/solvers.rs)
def _empty_like_f(a, shape):
    if _is_torch(a):
        import torch

        return torch.zeros(shape[::-1], dtype=a.dtype, device=a.device).t()  # column major
    return np.zeros(shape, dtype=a.dtype, order="F")


def _copy_f(a):
    if _is_torch(a):
        return a.t().contiguous().t()
    return np.array(a, order="F", copy=True)


def partial_piv_lu_solve_in_place(lu, perm_fwd, perm_bwd, rhs, transpose=False, par=PAR_SEQ):
    """lu/partial_pivoting/solve.rs:20-80: rhs <- A^-1 rhs (or A^-T rhs); `lu` holds L (unit lower) and U packed"""
    suf, _, _ = _dtype_suffix(lu)
    it = "u64" if np.dtype(perm_fwd.dtype) == np.uint64 else "u32"
    name = "partial_piv_lu_solve_transpose_in_place" if transpose else "partial_piv_lu_solve_in_place"
    n = lu.shape[0]
    getattr(lib(), f"libfaer_v0_23_{name}_{it}_{suf}")(
        _mat(lu), _mat(lu), C.c_int(0), SliceRef(perm_fwd.ctypes.data, n), SliceRef(perm_bwd.ctypes.data, n), _mat(rhs, MatMut), par,
        MemAlloc(None, 0))
    return rhs


def qr_solve_lstsq_in_place(qr, q_coeff, rhs, par=PAR_SEQ):
    """qr/no_pivoting/solve.rs:38-75: the least squares solution ends in the top ncols rows of rhs"""
    suf, _, _ = _dtype_suffix(qr)
    getattr(lib(), f"libfaer_v0_23_qr_solve_lstsq_in_place_{suf}")(_mat(qr), _mat(q_coeff), _mat(qr), C.c_int(0), _mat(rhs, MatMut), par,
                                                                   MemAlloc(None, 0))
    return rhs


def qr_solve_in_place(qr, q_coeff, rhs, transpose=False, par=PAR_SEQ):
    """qr/no_pivoting/solve.rs:98-175: square systems, rhs <- A^-1 rhs or A^-T rhs"""
    suf, _, _ = _dtype_suffix(qr)
    name = "qr_solve_transpose_in_place" if transpose else "qr_solve_in_place"
    getattr(lib(), f"libfaer_v0_23_{name}_{suf}")(_mat(qr), _mat(q_coeff), _mat(qr), C.c_int(0), _mat(rhs, MatMut), par, MemAlloc(None, 0))
    return rhs


class Llt:
    """solvers.rs:770-817: copies the lower triangle, factors, zeroes the strict upper triangle."""

    def __init__(self, a):
        self.l = _copy_f(a)
        llt_factor_in_place(self.l)
        if _is_torch(self.l):
            self.l.copy_(self.l.tril())
        else:
            self.l[:] = np.tril(self.l)

    def L(self):
        return self.l

    def solve_in_place(self, rhs):
        return llt_solve_in_place(self.l, rhs)


class PartialPivLu:
    """solvers.rs:955-1035"""

    def __init__(self, a):
        self.lu = _copy_f(a)
        self.perm, self.perm_inv, self.transposition_count = partial_piv_lu_factor_in_place(self.lu)

    def solve_in_place(self, rhs):
        return partial_piv_lu_solve_in_place(self.lu, self.perm, self.perm_inv, rhs)

    def solve_transpose_in_place(self, rhs):
        return partial_piv_lu_solve_in_place(self.lu, self.perm, self.perm_inv, rhs, transpose=True)


class Qr:
    """solvers.rs:1106-1204"""

    def __init__(self, a):
        m, n = a.shape
        self.qr = _copy_f(a)
        bs = qr_recommended_block_size(m, n, np.float64 if _dtype_suffix(a)[0] == "f64" else np.float32)
        self.q_coeff = _empty_like_f(a, (bs, min(m, n)))
        self.rank = qr_factor_in_place(self.qr, self.q_coeff)

    def Q_basis(self):
        return self.qr

    def Q_coeff(self):
        return self.q_coeff

    def solve_lstsq_in_place(self, rhs):
        return qr_solve_lstsq_in_place(self.qr, self.q_coeff, rhs)
