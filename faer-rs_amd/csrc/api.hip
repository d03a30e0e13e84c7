// extern "C" surface of libfaer_hip.so (see include/faer_hip.h for the contract and the reference
// citations of every entry point).  This file only validates shapes, stages host operands and forwards
// to the device drivers; there is deliberately no CPU compute path here.
#include <atomic>

#include "common.h"

using namespace fh;

namespace {

template <typename T> MatV<const T> view(FaerMatRef m)
{
	return MatV<const T>{static_cast<const T *>(m.ptr), (idx_t) m.nrows, (idx_t) m.ncols, (idx_t) m.row_stride,
			     (idx_t) m.col_stride};
}
template <typename T> MatV<T> view(FaerMatMut m)
{
	return MatV<T>{static_cast<T *>(m.ptr), (idx_t) m.nrows, (idx_t) m.ncols, (idx_t) m.row_stride,
		       (idx_t) m.col_stride};
}

std::atomic<int> g_par_tag{(int) FaerParTag_Rayon};
std::atomic<size_t> g_par_threads{0};

// ---- matmul ---------------------------------------------------------------------------------------
template <typename T> void matmul_api(FaerMatMut C, FaerAccum accum, FaerMatRef A, FaerMatRef B, const void *alpha)
{
	// faer/src/linalg/matmul/mod.rs:1562-1575 `precondition`
	FH_CHECK(C.nrows == A.nrows && C.ncols == B.ncols && A.ncols == B.nrows, "matmul: dimension mismatch");
	FH_CHECK(alpha != nullptr, "matmul: alpha is NULL");
	const bool add = accum == FaerAccum_Add;
	Staged<const T> a(view<T>(A), true, false), b(view<T>(B), true, false);
	Staged<T> c(view<T>(C), add, true); // Replace never reads dst
	gemm_dev<T>(c.dev, DST_FULL, add, a.dev, b.dev, *static_cast<const T *>(alpha));
}

template <typename T>
void matmul_triangular_api(FaerMatMut C, FaerBlock cb, FaerAccum accum, FaerMatRef A, FaerBlock ab, FaerMatRef B, FaerBlock bb,
			   const void *alpha)
{
	FH_CHECK(C.nrows == A.nrows && C.ncols == B.ncols && A.ncols == B.nrows, "matmul_triangular: dimension mismatch");
	// triangular operands must be square (faer/src/linalg/matmul/triangular.rs:1246-1262)
	FH_CHECK(cb == FaerBlock_Rectangular || C.nrows == C.ncols, "matmul_triangular: triangular dst must be square");
	FH_CHECK(ab == FaerBlock_Rectangular || A.nrows == A.ncols, "matmul_triangular: triangular lhs must be square");
	FH_CHECK(bb == FaerBlock_Rectangular || B.nrows == B.ncols, "matmul_triangular: triangular rhs must be square");
	const bool add = accum == FaerAccum_Add;
	Staged<const T> a(view<T>(A), true, false), b(view<T>(B), true, false);
	// a triangular dst keeps its other triangle => always copy in
	Staged<T> c(view<T>(C), add || cb != FaerBlock_Rectangular, true);
	matmul_triangular_dev<T>(c.dev, (int) cb, add, a.dev, (int) ab, b.dev, (int) bb, *static_cast<const T *>(alpha));
}

template <typename T> void trsm_api(FaerMatRef Tm, FaerMatMut rhs, bool upper, bool unit)
{
	FH_CHECK(Tm.nrows == Tm.ncols && rhs.nrows == Tm.ncols, "triangular solve: dimension mismatch");
	Staged<const T> t(view<T>(Tm), true, false);
	Staged<T> x(view<T>(rhs), true, true);
	if (upper)
		trsm_upper_dev<T>(t.dev, unit, x.dev);
	else
		trsm_lower_dev<T>(t.dev, unit, x.dev);
}

template <typename T> FaerLltStatus llt_api(FaerMatMut A, FaerLltRegularization reg)
{
	FH_CHECK(A.nrows == A.ncols, "llt: matrix must be square");
	T delta = reg.dynamic_regularization_delta ? *static_cast<const T *>(reg.dynamic_regularization_delta) : (T) 0;
	T eps = reg.dynamic_regularization_epsilon ? *static_cast<const T *>(reg.dynamic_regularization_epsilon) : (T) 0;
	long r;
	{
		Staged<T> a(view<T>(A), true, true);
		r = potrf_lower_dev<T>(a.dev, delta, eps);
	}
	FaerLltStatus st;
	memset(&st, 0, sizeof(st));
	if (r >= 0) {
		st.tag = FaerLltStatus_Ok;
		st.ok.dynamic_regularization_count = (size_t) r;
	} else {
		st.tag = FaerLltStatus_NonPositivePivot;
		st.non_positive_pivot.index = (size_t) (-r - 1);
	}
	return st;
}

template <typename T> FaerLdltStatus ldlt_api(FaerMatMut A, FaerLdltRegularization reg)
{
	FH_CHECK(A.nrows == A.ncols, "ldlt: matrix must be square");
	T delta = reg.dynamic_regularization_delta ? *static_cast<const T *>(reg.dynamic_regularization_delta) : (T) 0;
	T eps = reg.dynamic_regularization_epsilon ? *static_cast<const T *>(reg.dynamic_regularization_epsilon) : (T) 0;
	const signed char *signs = static_cast<const signed char *>(reg.dynamic_regularization_signs.ptr);
	if (signs) {
		FH_CHECK(reg.dynamic_regularization_signs.len >= A.nrows, "ldlt: the sign slice is shorter than the matrix");
		FH_CHECK(!is_device_ptr(signs), "ldlt: the sign slice must be host memory");
	}
	long r;
	{
		Staged<T> a(view<T>(A), true, true);
		r = sytrf_lower_dev<T>(a.dev, delta, eps, signs);
	}
	FaerLdltStatus st;
	memset(&st, 0, sizeof(st));
	if (r >= 0) {
		st.tag = FaerLdltStatus_Ok;
		st.ok.dynamic_regularization_count = (size_t) r;
	} else {
		st.tag = FaerLdltStatus_ZeroPivot;
		st.zero_pivot.index = (size_t) (-r - 1);
	}
	return st;
}

// cholesky/ldlt/solve.rs:12-50: L y = b (unit lower), y <- D^-1 y, L^H x = y
template <typename T> void ldlt_solve_api(FaerMatRef L, FaerVecRef D, FaerMatMut rhs)
{
	FH_CHECK(L.nrows == L.ncols && rhs.nrows == L.nrows && D.len == L.nrows, "ldlt solve: dimension mismatch");
	Staged<const T> l(view<T>(L), true, false);
	// D as an n x 1 view (stride in elements)
	FaerMatRef Dm{D.ptr, D.len, 1, D.stride, 0};
	Staged<const T> d(view<T>(Dm), true, false);
	Staged<T> x(view<T>(rhs), true, true);
	trsm_lower_dev<T>(l.dev, true, x.dev);
	scale_rows_recip_dev<T>(x.dev, d.dev.p, d.dev.rs);
	trsm_upper_dev<T>(l.dev.t(), true, x.dev);
}

template <typename T> void llt_solve_api(FaerMatRef L, FaerMatMut rhs)
{
	// cholesky/llt/solve.rs:12-35: L y = b ; L^H x = y
	FH_CHECK(L.nrows == L.ncols && rhs.nrows == L.nrows, "llt solve: dimension mismatch");
	Staged<const T> l(view<T>(L), true, false);
	Staged<T> x(view<T>(rhs), true, true);
	trsm_lower_dev<T>(l.dev, false, x.dev);
	trsm_upper_dev<T>(l.dev.t(), false, x.dev);
}

template <typename T, typename I> FaerPartialPivLuStatus lu_api(FaerMatMut A, FaerSliceMut pf, FaerSliceMut pb)
{
	const idx_t m = (idx_t) A.nrows;
	FH_CHECK((idx_t) pf.len == m && (idx_t) pb.len == m, "partial_piv_lu: perm slices must have nrows entries");
	FH_CHECK(!is_device_ptr(pf.ptr) && !is_device_ptr(pb.ptr), "partial_piv_lu: perm slices must be host memory");
	std::vector<idx_t> perm((size_t) m), perm_inv((size_t) m);
	long nt;
	{
		Staged<T> a(view<T>(A), true, true);
		nt = getrf_dev<T>(a.dev, perm.data(), perm_inv.data());
		if (nt < 0)
			a.writeback = false; // Unknown (exchange timeout without room for the rerun): a host operand keeps its input
	}
	I *f = static_cast<I *>(pf.ptr), *b = static_cast<I *>(pb.ptr);
	for (idx_t i = 0; i < m; ++i) {
		f[i] = (I) perm[(size_t) i];
		b[i] = (I) perm_inv[(size_t) i];
	}
	FaerPartialPivLuStatus st;
	memset(&st, 0, sizeof(st));
	st.tag = nt >= 0 ? FaerPartialPivLuStatus_Ok : FaerPartialPivLuStatus_Unknown; // Unknown: exchange timeout (getrf.hip)
	st.ok.transposition_count = nt >= 0 ? (size_t) nt : 0;
	return st;
}

size_t qr_block_size(size_t nrows, size_t ncols)
{
	// qr/no_pivoting/factor.rs:91-116
	const size_t prod = nrows * ncols;
	const size_t size = nrows < ncols ? nrows : ncols;
	size_t r;
	if (prod > 8192UL * 8192)
		r = 256;
	else if (prod > 2048UL * 2048)
		r = 128;
	else if (prod > 1024UL * 1024)
		r = 64;
	else if (prod > 512UL * 512)
		r = 48;
	else if (prod > 128UL * 128)
		r = 32;
	else if (prod > 32UL * 32)
		r = 8;
	else if (prod > 16UL * 16)
		r = 4;
	else
		r = 1;
	if (r > size)
		r = size;
	return r < 1 ? 1 : r;
}

template <typename T> FaerQrStatus qr_api(FaerMatMut A, FaerMatMut Q, FaerQrParams params)
{
	const size_t size = A.nrows < A.ncols ? A.nrows : A.ncols;
	FH_CHECK(Q.nrows > 0 && Q.ncols == size, "qr: Q_coeff must be block_size x min(nrows, ncols)");
	long rank;
	{
		Staged<T> a(view<T>(A), true, true);
		Staged<T> q(view<T>(Q), false, true);
		rank = geqrf_dev<T>(a.dev, q.dev, (idx_t) params.blocking_threshold);
	}
	FaerQrStatus st;
	memset(&st, 0, sizeof(st));
	st.tag = FaerQrStatus_Ok;
	st.ok.rank = (size_t) rank;
	return st;
}

template <typename T> void apply_hh_api(FaerMatRef V, FaerMatRef H, FaerMatMut rhs, bool transpose)
{
	const size_t size = V.nrows < V.ncols ? V.nrows : V.ncols;
	FH_CHECK(H.nrows > 0 && H.ncols == size && rhs.nrows == V.nrows, "apply_householder: dimension mismatch");
	Staged<const T> v(view<T>(V), true, false), h(view<T>(H), true, false);
	Staged<T> x(view<T>(rhs), true, true);
	apply_householder_sequence_left_dev<T>(v.dev, h.dev, x.dev, transpose);
}

// rhs[i, :] <- rhs[perm[i], :]   (perm/mod.rs:256-294 permute_rows with dst == a copy of src); perm is a HOST slice
template <typename T, typename I> void permute_rows_dev_api(MatV<T> X, const I *perm_host)
{
	const idx_t n = X.nrows, k = X.ncols;
	if (n == 0 || k == 0)
		return;
	std::vector<idx_t> p64((size_t) n);
	for (idx_t i = 0; i < n; ++i) {
		p64[(size_t) i] = (idx_t) perm_host[i];
		FH_CHECK(p64[(size_t) i] >= 0 && p64[(size_t) i] < n, "permutation index out of range");
	}
	Scratch pb((size_t) n * sizeof(idx_t)), tb((size_t) n * (size_t) k * sizeof(T));
	FH_HIP(hipMemcpyAsync(pb.p, p64.data(), (size_t) n * sizeof(idx_t), hipMemcpyHostToDevice, ctx().stream));
	MatV<T> tmp{tb.as<T>(), n, k, 1, n};
	gather_rows_dev<T>(tmp, X.c(), pb.as<idx_t>());
	copy_dev<T>(X, tmp.c());
	ctx().sync(); // p64 and the scratch buffers go out of scope
}

// lu/partial_pivoting/solve.rs:20-50 and :52-80
template <typename T, typename I>
void lu_solve_api(FaerMatRef L, FaerMatRef U, FaerSliceRef pf, FaerSliceRef pb, FaerMatMut rhs, bool transpose)
{
	const size_t n = L.nrows;
	FH_CHECK(L.ncols == n && U.nrows == n && U.ncols == n && rhs.nrows == n && pf.len >= n && pb.len >= n,
		 "partial_piv_lu solve: dimension mismatch");
	FH_CHECK(!is_device_ptr(pf.ptr) && !is_device_ptr(pb.ptr), "partial_piv_lu solve: perm slices must be host memory");
	Staged<const T> l(view<T>(L), true, false), u(view<T>(U), true, false);
	Staged<T> x(view<T>(rhs), true, true);
	if (!transpose) {
		permute_rows_dev_api<T, I>(x.dev, static_cast<const I *>(pf.ptr));
		trsm_lower_dev<T>(l.dev, true, x.dev);
		trsm_upper_dev<T>(u.dev, false, x.dev);
	} else {
		trsm_lower_dev<T>(u.dev.t(), false, x.dev);
		trsm_upper_dev<T>(l.dev.t(), true, x.dev);
		permute_rows_dev_api<T, I>(x.dev, static_cast<const I *>(pb.ptr)); // the inverse permutation
	}
}

// qr/no_pivoting/solve.rs:38-75 (lstsq / square) and :140-175 (transpose)
template <typename T> void qr_solve_api(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerMatMut rhs, bool transpose, bool square)
{
	const size_t m = Qb.nrows, n = Qb.ncols;
	const size_t size = m < n ? m : n;
	FH_CHECK(Qc.nrows > 0 && rhs.nrows == m && m >= n && Qc.ncols == size && R.nrows >= size && R.ncols == n,
		 "qr solve: dimension mismatch");
	if (square || transpose)
		FH_CHECK(m == n && R.nrows == n, "qr solve: the factorization must be square");
	Staged<const T> qb(view<T>(Qb), true, false), qc(view<T>(Qc), true, false), r(view<T>(R), true, false);
	Staged<T> x(view<T>(rhs), true, true);
	MatV<const T> Rtop = r.dev.sub(0, 0, (idx_t) size, (idx_t) n);
	if (!transpose) {
		apply_householder_sequence_left_dev<T>(qb.dev, qc.dev, x.dev, true); // Q^H rhs
		trsm_upper_dev<T>(Rtop, false, x.dev.sub(0, 0, (idx_t) size, x.dev.ncols));
	} else {
		trsm_lower_dev<T>(Rtop.t(), false, x.dev);
		apply_householder_sequence_left_dev<T>(qb.dev, qc.dev, x.dev, false); // Q rhs
	}
}

FaerLayout layout(size_t bytes, size_t align) { return FaerLayout{bytes, align}; }

// ---- triangular inverse (triangular_inverse.rs:43-230): dst's other triangle (and, unit: its diagonal) is untouched
template <typename T> void tri_inverse_api(FaerMatMut Out, FaerMatRef Tm, bool upper, bool unit)
{
	FH_CHECK(Tm.nrows == Tm.ncols && Out.nrows == Tm.nrows && Out.ncols == Tm.ncols, "triangular inverse: dimension mismatch");
	Staged<const T> t(view<T>(Tm), true, false);
	Staged<T> o(view<T>(Out), true, true); // keeps the triangle that is not written
	if (upper)
		tri_invert_lower_dev<T>(o.dev.t(), t.dev.t(), unit);
	else
		tri_invert_lower_dev<T>(o.dev, t.dev, unit);
}

// cholesky/llt/reconstruct.rs:14-40: lower(out) = L L^H
template <typename T> void llt_reconstruct_api(FaerMatMut Out, FaerMatRef L)
{
	FH_CHECK(L.nrows == L.ncols && Out.nrows == L.nrows && Out.ncols == L.nrows, "llt reconstruct: dimension mismatch");
	Staged<const T> l(view<T>(L), true, false);
	Staged<T> o(view<T>(Out), true, true);
	matmul_triangular_dev<T>(o.dev, 1, false, l.dev, 1, l.dev.t(), 2, (T) 1);
}

// cholesky/llt/inverse.rs:14-47: lower(out) = W^H W, W = inv(L)
template <typename T> void llt_inverse_api(FaerMatMut Out, FaerMatRef L)
{
	const idx_t n = (idx_t) L.nrows;
	FH_CHECK(L.nrows == L.ncols && Out.nrows == L.nrows && Out.ncols == L.nrows, "llt inverse: dimension mismatch");
	Staged<const T> l(view<T>(L), true, false);
	Staged<T> o(view<T>(Out), true, true);
	Scratch wb((size_t) n * (size_t) n * sizeof(T) + 256);
	MatV<T> W{wb.as<T>(), n, n, 1, n};
	tri_invert_lower_dev<T>(W, l.dev, false);
	matmul_triangular_dev<T>(o.dev, 1, false, W.t().c(), 2, W.c(), 1, (T) 1);
	ctx().sync();
}

// cholesky/ldlt/reconstruct.rs:14-62: lower(out) = (L D) L^H, L unit lower
template <typename T> void ldlt_reconstruct_api(FaerMatMut Out, FaerMatRef L, FaerVecRef D)
{
	const idx_t n = (idx_t) L.nrows;
	FH_CHECK(L.nrows == L.ncols && Out.nrows == L.nrows && Out.ncols == L.nrows && D.len == L.nrows, "ldlt reconstruct: dimension mismatch");
	Staged<const T> l(view<T>(L), true, false);
	FaerMatRef Dm{D.ptr, D.len, 1, D.stride, 0};
	Staged<const T> d(view<T>(Dm), true, false);
	Staged<T> o(view<T>(Out), true, true);
	Scratch wb((size_t) n * (size_t) n * sizeof(T) + 256);
	MatV<T> LxD{wb.as<T>(), n, n, 1, n};
	ldlt_scale_lower_dev<T>(LxD, l.dev, d.dev.p, d.dev.rs);
	matmul_triangular_dev<T>(o.dev, 1, false, LxD.c(), 1, l.dev.t(), 6, (T) 1);
	ctx().sync();
}

// cholesky/ldlt/inverse.rs:14-62: lower(out) = (W^H D^-1) W, W = inv(L) unit lower
template <typename T> void ldlt_inverse_api(FaerMatMut Out, FaerMatRef L, FaerVecRef D)
{
	const idx_t n = (idx_t) L.nrows;
	FH_CHECK(L.nrows == L.ncols && Out.nrows == L.nrows && Out.ncols == L.nrows && D.len == L.nrows, "ldlt inverse: dimension mismatch");
	Staged<const T> l(view<T>(L), true, false);
	FaerMatRef Dm{D.ptr, D.len, 1, D.stride, 0};
	Staged<const T> d(view<T>(Dm), true, false);
	Staged<T> o(view<T>(Out), true, true);
	Scratch wb((size_t) n * (size_t) n * sizeof(T) + 256);
	MatV<T> W{wb.as<T>(), n, n, 1, n};
	tri_invert_lower_dev<T>(W, l.dev, true);
	ldlt_inverse_prepare_dev<T>(W, d.dev.p, d.dev.rs);
	matmul_triangular_dev<T>(o.dev, 1, false, W.c(), 2, W.c(), 5, (T) 1);
	ctx().sync();
}

template <typename I> static void upload_perm(Scratch &buf, const void *perm_host, idx_t n)
{
	std::vector<idx_t> p64((size_t) n);
	for (idx_t i = 0; i < n; ++i) {
		p64[(size_t) i] = (idx_t) static_cast<const I *>(perm_host)[i];
		FH_CHECK(p64[(size_t) i] >= 0 && p64[(size_t) i] < n, "permutation index out of range");
	}
	FH_HIP(hipMemcpyAsync(buf.p, p64.data(), (size_t) n * sizeof(idx_t), hipMemcpyHostToDevice, ctx().stream));
	ctx().sync(); // p64 goes out of scope
}

// lu/partial_pivoting/reconstruct.rs:17-83: out = P^-1 (L U)
template <typename T, typename I> void lu_reconstruct_api(FaerMatMut Out, FaerMatRef L, FaerMatRef U, FaerSliceRef pf, FaerSliceRef pb)
{
	const idx_t m = (idx_t) L.nrows, n = (idx_t) U.ncols;
	const idx_t size = m < n ? m : n;
	FH_CHECK(Out.nrows == L.nrows && Out.ncols == U.ncols && (idx_t) L.ncols >= size && (idx_t) U.nrows >= size && pf.len >= L.nrows &&
			 pb.len >= L.nrows,
		 "partial_piv_lu reconstruct: dimension mismatch");
	FH_CHECK(!is_device_ptr(pf.ptr) && !is_device_ptr(pb.ptr), "partial_piv_lu reconstruct: perm slices must be host memory");
	if (m == 0 || n == 0)
		return;
	Staged<const T> l(view<T>(L), true, false), u(view<T>(U), true, false);
	Staged<T> o(view<T>(Out), false, true);
	Scratch tb((size_t) m * (size_t) n * sizeof(T)), pbuf((size_t) m * sizeof(idx_t));
	MatV<T> tmp{tb.as<T>(), m, n, 1, m};
	matmul_triangular_dev<T>(tmp.sub(0, 0, size, size), 0, false, l.dev.sub(0, 0, size, size), 5, u.dev.sub(0, 0, size, size), 2, (T) 1);
	if (m > n)
		matmul_triangular_dev<T>(tmp.sub(size, 0, m - size, size), 0, false, l.dev.sub(size, 0, m - size, size), 0, u.dev.sub(0, 0, size, size), 2,
					 (T) 1);
	if (m < n)
		matmul_triangular_dev<T>(tmp.sub(0, size, size, n - size), 0, false, l.dev.sub(0, 0, size, size), 5, u.dev.sub(0, size, size, n - size), 0,
					 (T) 1);
	upload_perm<I>(pbuf, pb.ptr, m); // permute_rows(out, tmp, perm.inverse()): out[i, :] = tmp[perm_bwd[i], :]
	gather_rows_dev<T>(o.dev, tmp.c(), pbuf.as<idx_t>());
	ctx().sync();
}

// lu/partial_pivoting/inverse.rs:15-62: out = U^-1 L^-1 P
template <typename T, typename I> void lu_inverse_api(FaerMatMut Out, FaerMatRef L, FaerMatRef U, FaerSliceRef pf, FaerSliceRef pb)
{
	const idx_t n = (idx_t) L.ncols;
	FH_CHECK((idx_t) L.nrows == n && (idx_t) U.nrows == n && (idx_t) U.ncols == n && (idx_t) Out.nrows == n && (idx_t) Out.ncols == n &&
			 (idx_t) pf.len >= n && (idx_t) pb.len >= n,
		 "partial_piv_lu inverse: dimension mismatch");
	FH_CHECK(!is_device_ptr(pf.ptr) && !is_device_ptr(pb.ptr), "partial_piv_lu inverse: perm slices must be host memory");
	if (n == 0)
		return;
	Staged<const T> l(view<T>(L), true, false), u(view<T>(U), true, false);
	Staged<T> o(view<T>(Out), false, true);
	Scratch tb((size_t) n * (size_t) n * sizeof(T)), pbuf((size_t) n * sizeof(idx_t));
	MatV<T> tmp{tb.as<T>(), n, n, 1, n};
	tri_invert_lower_dev<T>(o.dev, l.dev, true);	  // strict lower part of out
	tri_invert_lower_dev<T>(o.dev.t(), u.dev.t(), false); // upper part (with diagonal) of out
	matmul_triangular_dev<T>(tmp, 0, false, o.dev.c(), 2, o.dev.c(), 5, (T) 1);
	upload_perm<I>(pbuf, pb.ptr, n); // permute_cols(out, tmp, perm.inverse()): out[:, j] = tmp[:, perm_bwd[j]]
	gather_rows_dev<T>(o.dev.t(), tmp.t().c(), pbuf.as<idx_t>());
	ctx().sync();
}

// qr/no_pivoting/reconstruct.rs:18-52: out = Q [R; 0]
template <typename T> void qr_reconstruct_api(FaerMatMut Out, FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R)
{
	const idx_t m = (idx_t) Qb.nrows, n = (idx_t) R.ncols;
	const idx_t size = m < n ? m : n;
	FH_CHECK((idx_t) Out.nrows == m && (idx_t) Out.ncols == n && (idx_t) Qb.ncols == size && (idx_t) Qc.ncols == size && (idx_t) R.nrows == size &&
			 Qc.nrows > 0,
		 "qr reconstruct: dimension mismatch");
	Staged<const T> v(view<T>(Qb), true, false), h(view<T>(Qc), true, false), r(view<T>(R), true, false);
	Staged<T> o(view<T>(Out), false, true);
	MatV<const T> Rv = r.dev;
	zero_then_upper_dev<T>(o.dev, &Rv);
	apply_householder_sequence_left_dev<T>(v.dev, h.dev, o.dev, false);
}

// qr/no_pivoting/inverse.rs:17-58: out = R^-1 Q^H
template <typename T> void qr_inverse_api(FaerMatMut Out, FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R)
{
	const idx_t n = (idx_t) Qb.ncols;
	FH_CHECK(Qc.nrows > 0 && (idx_t) Qb.nrows == n && (idx_t) Qc.ncols == n && (idx_t) R.nrows == n && (idx_t) R.ncols == n &&
			 (idx_t) Out.nrows == n && (idx_t) Out.ncols == n,
		 "qr inverse: dimension mismatch");
	Staged<const T> v(view<T>(Qb), true, false), h(view<T>(Qc), true, false), r(view<T>(R), true, false);
	Staged<T> o(view<T>(Out), false, true);
	zero_then_upper_dev<T>(o.dev, nullptr);
	tri_invert_lower_dev<T>(o.dev.t(), r.dev.t(), false);
	// out <- out Q^H  ==  (Q out^T)^T   (householder.rs:836-854)
	apply_householder_sequence_left_dev<T>(v.dev, h.dev, o.dev.t(), false);
}

// householder.rs:813-854: M <- M Q (transpose == false) or M Q^H (transpose == true), as left applications on M^T
template <typename T> void apply_hh_right_api(FaerMatRef V, FaerMatRef H, FaerMatMut M, bool transpose)
{
	const size_t size = V.nrows < V.ncols ? V.nrows : V.ncols;
	FH_CHECK(H.nrows > 0 && H.ncols == size && M.ncols == V.nrows, "apply_householder (right): dimension mismatch");
	Staged<const T> v(view<T>(V), true, false), h(view<T>(H), true, false);
	Staged<T> x(view<T>(M), true, true);
	apply_householder_sequence_left_dev<T>(v.dev, h.dev, x.dev.t(), !transpose);
}

// lu/full_pivoting/factor.rs:452-525
template <typename T, typename I>
FaerFullPivLuStatus full_lu_api(FaerMatMut A, FaerSliceMut rpf, FaerSliceMut rpb, FaerSliceMut cpf, FaerSliceMut cpb)
{
	const idx_t m = (idx_t) A.nrows, n = (idx_t) A.ncols;
	FH_CHECK((idx_t) rpf.len == m && (idx_t) rpb.len == m && (idx_t) cpf.len == n && (idx_t) cpb.len == n,
		 "full_piv_lu: perm slices must have nrows / ncols entries");
	FH_CHECK(!is_device_ptr(rpf.ptr) && !is_device_ptr(rpb.ptr) && !is_device_ptr(cpf.ptr) && !is_device_ptr(cpb.ptr),
		 "full_piv_lu: perm slices must be host memory");
	std::vector<idx_t> rp((size_t) m), rpi((size_t) m), cp((size_t) n), cpi((size_t) n);
	long nt;
	{
		Staged<T> a(view<T>(A), true, true);
		nt = full_piv_lu_dev<T>(a.dev, rp.data(), rpi.data(), cp.data(), cpi.data());
	}
	for (idx_t i = 0; i < m; ++i) {
		static_cast<I *>(rpf.ptr)[i] = (I) rp[(size_t) i];
		static_cast<I *>(rpb.ptr)[i] = (I) rpi[(size_t) i];
	}
	for (idx_t j = 0; j < n; ++j) {
		static_cast<I *>(cpf.ptr)[j] = (I) cp[(size_t) j];
		static_cast<I *>(cpb.ptr)[j] = (I) cpi[(size_t) j];
	}
	FaerFullPivLuStatus st;
	memset(&st, 0, sizeof(st));
	st.tag = FaerFullPivLuStatus_Ok;
	st.ok.transposition_count = (size_t) nt;
	return st;
}

// lu/full_pivoting/solve.rs: A x = b (:22-60) and A^T x = b (:75-110)
template <typename T, typename I>
void full_lu_solve_api(FaerMatRef L, FaerMatRef U, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb, FaerMatMut rhs,
		       bool transpose)
{
	const size_t n = L.nrows;
	FH_CHECK(L.ncols == n && U.nrows == n && U.ncols == n && rhs.nrows == n && rpf.len >= n && rpb.len >= n && cpf.len >= n && cpb.len >= n,
		 "full_piv_lu solve: dimension mismatch");
	FH_CHECK(!is_device_ptr(rpf.ptr) && !is_device_ptr(rpb.ptr) && !is_device_ptr(cpf.ptr) && !is_device_ptr(cpb.ptr),
		 "full_piv_lu solve: perm slices must be host memory");
	Staged<const T> l(view<T>(L), true, false), u(view<T>(U), true, false);
	Staged<T> x(view<T>(rhs), true, true);
	if (!transpose) {
		permute_rows_dev_api<T, I>(x.dev, static_cast<const I *>(rpf.ptr));
		trsm_lower_dev<T>(l.dev, true, x.dev);
		trsm_upper_dev<T>(u.dev, false, x.dev);
		permute_rows_dev_api<T, I>(x.dev, static_cast<const I *>(cpb.ptr)); // col_perm.inverse()
	} else {
		permute_rows_dev_api<T, I>(x.dev, static_cast<const I *>(cpf.ptr));
		trsm_lower_dev<T>(u.dev.t(), false, x.dev);
		trsm_upper_dev<T>(l.dev.t(), true, x.dev);
		permute_rows_dev_api<T, I>(x.dev, static_cast<const I *>(rpb.ptr)); // row_perm.inverse()
	}
}

// evd/tridiag.rs:274-299
template <typename T> void tridiag_api(FaerMatMut A, FaerMatMut Hh)
{
	FH_CHECK(A.nrows == A.ncols, "tridiag: the matrix must be square");
	FH_CHECK(Hh.ncols == (A.nrows > 0 ? A.nrows - 1 : 0), "tridiag: householder must have n - 1 columns"); // tridiag.rs:287
	if (A.nrows == 0)
		return; // :288-290
	FH_CHECK(Hh.nrows > 0 || A.nrows == 1, "tridiag: householder must have at least one row");
	if (A.nrows == 1)
		return;
	Staged<T> a(view<T>(A), true, true);
	Staged<T> h(view<T>(Hh), true, true); // only the block upper triangles are written
	tridiag_dev<T>(a.dev, h.dev);
}

// evd/hessenberg.rs:549-566
template <typename T> void hessenberg_api(FaerMatMut A, FaerMatMut Hh)
{
	FH_CHECK(A.nrows == A.ncols, "hessenberg: the matrix must be square");
	FH_CHECK(Hh.ncols == (A.nrows > 0 ? A.nrows - 1 : 0), "hessenberg: householder must have n - 1 columns"); // :557-560
	if (A.nrows <= 1)
		return;
	Staged<T> a(view<T>(A), true, true);
	Staged<T> h(view<T>(Hh), true, true); // only the block upper triangles are written
	hessenberg_dev<T>(a.dev, h.dev);
}

// svd/bidiag.rs:47-66
template <typename T> void bidiag_api(FaerMatMut A, FaerMatMut Hl, FaerMatMut Hr)
{
	const size_t size = A.nrows < A.ncols ? A.nrows : A.ncols;
	FH_CHECK(Hl.ncols == size && Hr.ncols == (size > 0 ? size - 1 : 0), "bidiag: H_left / H_right must have min(m, n) and min(m, n) - 1 columns"); // :60-61
	if (size == 0)
		return;
	Staged<T> a(view<T>(A), true, true);
	Staged<T> hl(view<T>(Hl), true, true), hr(view<T>(Hr), true, true); // only the block upper triangles are written
	bidiag_dev<T>(a.dev, hl.dev, hr.dev);
}

// qr/col_pivoting/factor.rs:356-395
template <typename T, typename I> FaerColPivQrStatus colpiv_qr_api(FaerMatMut A, FaerMatMut Q, FaerSliceMut pf, FaerSliceMut pb)
{
	const idx_t n = (idx_t) A.ncols;
	const size_t size = A.nrows < A.ncols ? A.nrows : A.ncols;
	FH_CHECK(Q.nrows > 0 && Q.ncols == size, "colpiv_qr: Q_coeff must be block_size x min(nrows, ncols)");
	FH_CHECK((idx_t) pf.len == n && (idx_t) pb.len == n, "colpiv_qr: perm slices must have ncols entries");
	FH_CHECK(!is_device_ptr(pf.ptr) && !is_device_ptr(pb.ptr), "colpiv_qr: perm slices must be host memory");
	std::vector<idx_t> cp((size_t) n), cpi((size_t) n);
	long nt;
	{
		Staged<T> a(view<T>(A), true, true);
		Staged<T> q(view<T>(Q), false, true);
		nt = colpiv_qr_dev<T>(a.dev, q.dev, cp.data(), cpi.data());
	}
	for (idx_t j = 0; j < n; ++j) {
		static_cast<I *>(pf.ptr)[j] = (I) cp[(size_t) j];
		static_cast<I *>(pb.ptr)[j] = (I) cpi[(size_t) j];
	}
	FaerColPivQrStatus st;
	memset(&st, 0, sizeof(st));
	st.tag = FaerColPivQrStatus_Ok;
	st.ok.transposition_count = (size_t) nt;
	return st;
}

// qr/col_pivoting/solve.rs:52-80 (lstsq / square) and :158-184 (transpose)
template <typename T, typename I>
void colpiv_qr_solve_api(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerSliceRef pf, FaerSliceRef pb, FaerMatMut rhs, bool transpose, bool square)
{
	const size_t m = Qb.nrows, n = Qb.ncols;
	const size_t size = m < n ? m : n;
	FH_CHECK(pf.len >= n && pb.len >= n && !is_device_ptr(pf.ptr) && !is_device_ptr(pb.ptr), "colpiv_qr solve: perm slices (host memory, ncols entries)");
	if (!transpose) {
		qr_solve_api<T>(Qb, Qc, R, rhs, false, square);
		Staged<T> x(view<T>(rhs), true, true);
		permute_rows_dev_api<T, I>(x.dev.sub(0, 0, (idx_t) size, x.dev.ncols), static_cast<const I *>(pb.ptr)); // col_perm.inverse()
	} else {
		{
			Staged<T> x(view<T>(rhs), true, true);
			permute_rows_dev_api<T, I>(x.dev, static_cast<const I *>(pf.ptr));
		}
		qr_solve_api<T>(Qb, Qc, R, rhs, true, true);
	}
}

// out(i, j) = tmp(rmap[i], cmap[j]) with host index maps (two gathers through a second temporary)
template <typename T, typename I> static void gather_rows_cols(MatV<T> out, MatV<T> tmp, const void *rmap, const void *cmap)
{
	const idx_t m = out.nrows, n = out.ncols;
	Scratch rb((size_t) m * sizeof(idx_t) + 256), cb((size_t) n * sizeof(idx_t) + 256), t2((size_t) m * (size_t) n * sizeof(T) + 256);
	upload_perm<I>(rb, rmap, m);
	upload_perm<I>(cb, cmap, n);
	MatV<T> tmp2{t2.as<T>(), m, n, 1, m};
	gather_rows_dev<T>(tmp2, tmp.c(), rb.as<idx_t>());
	gather_rows_dev<T>(out.t(), tmp2.t().c(), cb.as<idx_t>());
	ctx().sync();
}

// lu/full_pivoting/reconstruct.rs: out(i, j) = (L U)(row_perm_inv[i], col_perm_inv[j])
template <typename T, typename I>
void full_lu_reconstruct_api(FaerMatMut Out, FaerMatRef L, FaerMatRef U, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb)
{
	(void) rpf;
	(void) cpf;
	const idx_t m = (idx_t) L.nrows, n = (idx_t) U.ncols;
	const idx_t size = m < n ? m : n;
	FH_CHECK((idx_t) Out.nrows == m && (idx_t) Out.ncols == n && (idx_t) L.ncols >= size && (idx_t) U.nrows >= size && (idx_t) rpb.len >= m &&
			 (idx_t) cpb.len >= n,
		 "full_piv_lu reconstruct: dimension mismatch");
	FH_CHECK(!is_device_ptr(rpb.ptr) && !is_device_ptr(cpb.ptr), "full_piv_lu reconstruct: perm slices must be host memory");
	if (m == 0 || n == 0)
		return;
	Staged<const T> l(view<T>(L), true, false), u(view<T>(U), true, false);
	Staged<T> o(view<T>(Out), false, true);
	Scratch tb((size_t) m * (size_t) n * sizeof(T));
	MatV<T> tmp{tb.as<T>(), m, n, 1, m};
	matmul_triangular_dev<T>(tmp.sub(0, 0, size, size), 0, false, l.dev.sub(0, 0, size, size), 5, u.dev.sub(0, 0, size, size), 2, (T) 1);
	if (m > n)
		matmul_triangular_dev<T>(tmp.sub(size, 0, m - size, size), 0, false, l.dev.sub(size, 0, m - size, size), 0, u.dev.sub(0, 0, size, size), 2,
					 (T) 1);
	if (m < n)
		matmul_triangular_dev<T>(tmp.sub(0, size, size, n - size), 0, false, l.dev.sub(0, 0, size, size), 5, u.dev.sub(0, size, size, n - size), 0,
					 (T) 1);
	gather_rows_cols<T, I>(o.dev, tmp, rpb.ptr, cpb.ptr);
}

// lu/full_pivoting/inverse.rs: out(i, j) = (U^-1 L^-1)(col_perm_inv[i], row_perm_inv[j])
template <typename T, typename I>
void full_lu_inverse_api(FaerMatMut Out, FaerMatRef L, FaerMatRef U, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb)
{
	(void) rpf;
	(void) cpf;
	const idx_t n = (idx_t) L.ncols;
	FH_CHECK((idx_t) L.nrows == n && (idx_t) U.nrows == n && (idx_t) U.ncols == n && (idx_t) Out.nrows == n && (idx_t) Out.ncols == n &&
			 (idx_t) rpb.len >= n && (idx_t) cpb.len >= n,
		 "full_piv_lu inverse: dimension mismatch");
	FH_CHECK(!is_device_ptr(rpb.ptr) && !is_device_ptr(cpb.ptr), "full_piv_lu inverse: perm slices must be host memory");
	if (n == 0)
		return;
	Staged<const T> l(view<T>(L), true, false), u(view<T>(U), true, false);
	Staged<T> o(view<T>(Out), false, true);
	Scratch tb((size_t) n * (size_t) n * sizeof(T));
	MatV<T> tmp{tb.as<T>(), n, n, 1, n};
	tri_invert_lower_dev<T>(o.dev, l.dev, true);
	tri_invert_lower_dev<T>(o.dev.t(), u.dev.t(), false);
	matmul_triangular_dev<T>(tmp, 0, false, o.dev.c(), 2, o.dev.c(), 5, (T) 1);
	gather_rows_cols<T, I>(o.dev, tmp, cpb.ptr, rpb.ptr);
}

// qr/col_pivoting/reconstruct.rs: (Q R) with its columns permuted back; inverse.rs: rows of R^-1 Q^H permuted back
template <typename T, typename I>
void colpiv_qr_reconstruct_api(FaerMatMut Out, FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerSliceRef pf, FaerSliceRef pb)
{
	(void) pf;
	const idx_t n = (idx_t) R.ncols;
	FH_CHECK((idx_t) pb.len >= n && !is_device_ptr(pb.ptr), "colpiv_qr reconstruct: perm slices (host memory, ncols entries)");
	qr_reconstruct_api<T>(Out, Qb, Qc, R);
	Staged<T> o(view<T>(Out), true, true);
	permute_rows_dev_api<T, I>(o.dev.t(), static_cast<const I *>(pb.ptr)); // permute_cols_in_place(out, col_perm.inverse())
}
template <typename T, typename I>
void colpiv_qr_inverse_api(FaerMatMut Out, FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerSliceRef pf, FaerSliceRef pb)
{
	(void) pf;
	const idx_t n = (idx_t) R.ncols;
	FH_CHECK((idx_t) pb.len >= n && !is_device_ptr(pb.ptr), "colpiv_qr inverse: perm slices (host memory, ncols entries)");
	qr_inverse_api<T>(Out, Qb, Qc, R);
	Staged<T> o(view<T>(Out), true, true);
	permute_rows_dev_api<T, I>(o.dev, static_cast<const I *>(pb.ptr)); // permute_rows_in_place(out, col_perm.inverse())
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------------------- inner boundary
void faer_hip_gemm(FaerHipDType dtype, FaerHipIType itype, size_t m, size_t n, size_t k, void *dst, ptrdiff_t dst_rs,
		   ptrdiff_t dst_cs, const void *row_idx, const void *col_idx, FaerHipDstKind dst_kind, FaerAccum accum,
		   const void *lhs, ptrdiff_t lhs_rs, ptrdiff_t lhs_cs, bool conj_lhs, const void *diag, ptrdiff_t diag_stride,
		   const void *rhs, ptrdiff_t rhs_rs, ptrdiff_t rhs_cs, bool conj_rhs, const void *alpha, size_t n_threads)
{
	(void) conj_lhs; // conjugation is the identity for real scalars
	(void) conj_rhs;
	(void) n_threads;
	FH_CHECK(dtype == FaerHipDType_F32 || dtype == FaerHipDType_F64, "faer_hip_gemm: only f32/f64 are supported");
	FH_CHECK(alpha != nullptr, "faer_hip_gemm: alpha is NULL");
	if (m == 0 || n == 0)
		return;
	const bool add = accum == FaerAccum_Add;
	auto run = [&](auto tag) {
		typedef decltype(tag) T;
		const size_t isz = itype == FaerHipIType_U64 ? 8 : 4;
		// extent of dst in rows / cols when scattered through the index arrays
		idx_t drows = (idx_t) m, dcols = (idx_t) n;
		std::vector<unsigned char> hri, hci;
		auto max_idx = [&](const void *p, size_t cnt) {
			idx_t mx = 0;
			for (size_t i = 0; i < cnt; ++i) {
				idx_t v = isz == 8 ? (idx_t) static_cast<const uint64_t *>(p)[i]
						   : (idx_t) static_cast<const uint32_t *>(p)[i];
				if (v > mx)
					mx = v;
			}
			return mx;
		};
		const bool ri_host = row_idx && !is_device_ptr(row_idx), ci_host = col_idx && !is_device_ptr(col_idx);
		FH_CHECK((!row_idx || ri_host || is_device_ptr(dst)) && (!col_idx || ci_host || is_device_ptr(dst)),
			 "faer_hip_gemm: device index arrays require a device dst");
		if (ri_host)
			drows = max_idx(row_idx, m) + 1;
		if (ci_host)
			dcols = max_idx(col_idx, n) + 1;
		MatV<T> D{static_cast<T *>(dst), drows, dcols, (idx_t) dst_rs, (idx_t) dst_cs};
		MatV<const T> A{static_cast<const T *>(lhs), (idx_t) m, (idx_t) k, (idx_t) lhs_rs, (idx_t) lhs_cs};
		MatV<const T> B{static_cast<const T *>(rhs), (idx_t) k, (idx_t) n, (idx_t) rhs_rs, (idx_t) rhs_cs};
		MatV<const T> Dg{static_cast<const T *>(diag), diag ? (idx_t) k : 0, 1, (idx_t) diag_stride, 0};
		Staged<const T> a(A, true, false), b(B, true, false), dg(Dg, true, false);
		// a non-Full or scattered dst keeps untouched entries => copy in
		Staged<T> c(D, add || dst_kind != FaerHipDstKind_Full || row_idx || col_idx, true);
		Staged<const unsigned char> ri(MatV<const unsigned char>{static_cast<const unsigned char *>(row_idx),
									  row_idx ? (idx_t) (m * isz) : 0, 1, 1, 0},
					       true, false);
		Staged<const unsigned char> ci(MatV<const unsigned char>{static_cast<const unsigned char *>(col_idx),
									  col_idx ? (idx_t) (n * isz) : 0, 1, 1, 0},
					       true, false);
		GemmExtra<T> ex;
		ex.row_idx = row_idx ? ri.dev.p : nullptr;
		ex.col_idx = col_idx ? ci.dev.p : nullptr;
		ex.idx64 = isz == 8;
		ex.diag = diag ? dg.dev.p : nullptr;
		ex.diag_stride = dg.dev.rs; // 1 for a staged host vector, the caller's stride for a device one
		MatV<T> Cv{c.dev.p, (idx_t) m, (idx_t) n, c.dev.rs, c.dev.cs};
		gemm_dev<T>(Cv, (DstKind) dst_kind, add, a.dev, b.dev, *static_cast<const T *>(alpha), &ex);
	};
	if (dtype == FaerHipDType_F64)
		run(double());
	else
		run(float());
}

// ---------------------------------------------------------------------------------------- outer boundary
#define FH_FOR_DTYPES(X) X(f64, double) X(f32, float)

#define X(suf, T)                                                                                                      \
	void libfaer_v0_23_matmul_##suf(FaerMatMut C, FaerAccum accum, FaerMatRef A, FaerMatRef B, const void *alpha,    \
					FaerPar par)                                                                        \
	{                                                                                                              \
		(void) par;                                                                                            \
		matmul_api<T>(C, accum, A, B, alpha);                                                                  \
	}                                                                                                              \
	void libfaer_v0_23_matmul_triangular_##suf(FaerMatMut C, FaerBlock C_block, FaerAccum accum, FaerMatRef A,      \
						   FaerBlock A_block, FaerMatRef B, FaerBlock B_block, const void *alpha,    \
						   FaerPar par)                                                             \
	{                                                                                                              \
		(void) par;                                                                                            \
		matmul_triangular_api<T>(C, C_block, accum, A, A_block, B, B_block, alpha);                             \
	}                                                                                                              \
	void libfaer_v0_23_solve_triangular_lower_in_place_##suf(FaerMatRef L, FaerConj cj, FaerMatMut rhs, FaerPar par) \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		trsm_api<T>(L, rhs, false, false);                                                                     \
	}                                                                                                              \
	void libfaer_v0_23_solve_triangular_upper_in_place_##suf(FaerMatRef U, FaerConj cj, FaerMatMut rhs, FaerPar par) \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		trsm_api<T>(U, rhs, true, false);                                                                      \
	}                                                                                                              \
	void libfaer_v0_23_solve_unit_triangular_lower_in_place_##suf(FaerMatRef L, FaerConj cj, FaerMatMut rhs,        \
								      FaerPar par)                                          \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		trsm_api<T>(L, rhs, false, true);                                                                      \
	}                                                                                                              \
	void libfaer_v0_23_solve_unit_triangular_upper_in_place_##suf(FaerMatRef U, FaerConj cj, FaerMatMut rhs,        \
								      FaerPar par)                                          \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		trsm_api<T>(U, rhs, true, true);                                                                       \
	}                                                                                                              \
	FaerLltParams libfaer_v0_23_LltParams_##suf(void) { return FaerLltParams{64, 128}; }                            \
	FaerLdltParams libfaer_v0_23_LdltParams_##suf(void) { return FaerLdltParams{64, 128}; }                         \
	FaerLayout libfaer_v0_23_ldlt_factor_in_place_scratch_##suf(size_t dim, FaerPar par, FaerLdltParams params)     \
	{                                                                                                              \
		(void) par;                                                                                            \
		(void) params;                                                                                         \
		return layout(dim * sizeof(T), 64);                                                                    \
	}                                                                                                              \
	FaerLdltStatus libfaer_v0_23_ldlt_factor_in_place_##suf(FaerMatMut A, FaerLdltRegularization reg, FaerPar par,  \
								FaerMemAlloc mem, FaerLdltParams params)                    \
	{                                                                                                              \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		(void) params;                                                                                         \
		return ldlt_api<T>(A, reg);                                                                            \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_ldlt_solve_in_place_scratch_##suf(size_t dim, size_t rhs_ncols, FaerPar par)           \
	{                                                                                                              \
		(void) dim;                                                                                            \
		(void) rhs_ncols;                                                                                      \
		(void) par;                                                                                            \
		return layout(0, 1);                                                                                   \
	}                                                                                                              \
	void libfaer_v0_23_ldlt_solve_in_place_##suf(FaerMatRef L, FaerVecRef D, FaerConj cj, FaerMatMut rhs, FaerPar par, \
						     FaerMemAlloc mem)                                                      \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		ldlt_solve_api<T>(L, D, rhs);                                                                          \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_llt_factor_in_place_scratch_##suf(size_t dim, FaerPar par, FaerLltParams params)       \
	{                                                                                                              \
		(void) par;                                                                                            \
		(void) params;                                                                                         \
		return layout(dim * sizeof(T), 64);                                                                    \
	}                                                                                                              \
	FaerLltStatus libfaer_v0_23_llt_factor_in_place_##suf(FaerMatMut A, FaerLltRegularization reg, FaerPar par,     \
							      FaerMemAlloc mem, FaerLltParams params)                        \
	{                                                                                                              \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		(void) params;                                                                                         \
		return llt_api<T>(A, reg);                                                                             \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_llt_solve_in_place_scratch_##suf(size_t dim, size_t rhs_ncols, FaerPar par)            \
	{                                                                                                              \
		(void) dim;                                                                                            \
		(void) rhs_ncols;                                                                                      \
		(void) par;                                                                                            \
		return layout(0, 1);                                                                                   \
	}                                                                                                              \
	void libfaer_v0_23_llt_solve_in_place_##suf(FaerMatRef L, FaerConj cj, FaerMatMut rhs, FaerPar par,             \
						    FaerMemAlloc mem)                                                       \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		llt_solve_api<T>(L, rhs);                                                                              \
	}                                                                                                              \
	FaerPartialPivLuParams libfaer_v0_23_PartialPivLuParams_##suf(void)                                             \
	{                                                                                                              \
		return FaerPartialPivLuParams{16, 64, 128 * 128};                                                      \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_partial_piv_lu_factor_in_place_scratch_u32_##suf(size_t dim, size_t bs, FaerPar par,   \
										  FaerPartialPivLuParams params)            \
	{                                                                                                              \
		(void) par;                                                                                            \
		(void) params;                                                                                         \
		return layout((dim < bs ? dim : bs) * 4, 4);                                                           \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_partial_piv_lu_factor_in_place_scratch_u64_##suf(size_t dim, size_t bs, FaerPar par,   \
										  FaerPartialPivLuParams params)            \
	{                                                                                                              \
		(void) par;                                                                                            \
		(void) params;                                                                                         \
		return layout((dim < bs ? dim : bs) * 8, 8);                                                           \
	}                                                                                                              \
	FaerPartialPivLuStatus libfaer_v0_23_partial_piv_lu_factor_in_place_u32_##suf(                                  \
		FaerMatMut A, FaerSliceMut pf, FaerSliceMut pb, FaerPar par, FaerMemAlloc mem, FaerPartialPivLuParams params) \
	{                                                                                                              \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		(void) params;                                                                                         \
		return lu_api<T, uint32_t>(A, pf, pb);                                                                 \
	}                                                                                                              \
	FaerPartialPivLuStatus libfaer_v0_23_partial_piv_lu_factor_in_place_u64_##suf(                                  \
		FaerMatMut A, FaerSliceMut pf, FaerSliceMut pb, FaerPar par, FaerMemAlloc mem, FaerPartialPivLuParams params) \
	{                                                                                                              \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		(void) params;                                                                                         \
		return lu_api<T, uint64_t>(A, pf, pb);                                                                 \
	}                                                                                                              \
	FaerFullPivLuParams libfaer_v0_23_FullPivLuParams_##suf(void) \
	{ \
		return FaerFullPivLuParams{256 * 512}; \
	} \
	FaerLayout libfaer_v0_23_full_piv_lu_factor_in_place_scratch_u32_##suf(size_t dim, size_t bs, FaerPar par, FaerFullPivLuParams params) \
	{ \
		(void) bs; (void) par; (void) params; return layout(dim * 2 * sizeof(size_t), sizeof(size_t)); \
	} \
	FaerFullPivLuStatus libfaer_v0_23_full_piv_lu_factor_in_place_u32_##suf(FaerMatMut A, FaerSliceMut rpf, FaerSliceMut rpb, FaerSliceMut cpf, FaerSliceMut cpb, FaerPar par, FaerMemAlloc mem, FaerFullPivLuParams params) \
	{ \
		(void) par; (void) mem; (void) params; return full_lu_api<T, uint32_t>(A, rpf, rpb, cpf, cpb); \
	} \
	FaerLayout libfaer_v0_23_full_piv_lu_solve_in_place_scratch_u32_##suf(size_t dim, size_t k, FaerPar par) \
	{ \
		(void) par; return layout(dim * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_full_piv_lu_solve_in_place_u32_##suf(FaerMatRef L, FaerMatRef U, FaerConj cj, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; full_lu_solve_api<T, uint32_t>(L, U, rpf, rpb, cpf, cpb, rhs, false); \
	} \
	FaerLayout libfaer_v0_23_full_piv_lu_solve_transpose_in_place_scratch_u32_##suf(size_t dim, size_t k, FaerPar par) \
	{ \
		(void) par; return layout(dim * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_full_piv_lu_solve_transpose_in_place_u32_##suf(FaerMatRef L, FaerMatRef U, FaerConj cj, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; full_lu_solve_api<T, uint32_t>(L, U, rpf, rpb, cpf, cpb, rhs, true); \
	} \
	FaerLayout libfaer_v0_23_full_piv_lu_factor_in_place_scratch_u64_##suf(size_t dim, size_t bs, FaerPar par, FaerFullPivLuParams params) \
	{ \
		(void) bs; (void) par; (void) params; return layout(dim * 2 * sizeof(size_t), sizeof(size_t)); \
	} \
	FaerFullPivLuStatus libfaer_v0_23_full_piv_lu_factor_in_place_u64_##suf(FaerMatMut A, FaerSliceMut rpf, FaerSliceMut rpb, FaerSliceMut cpf, FaerSliceMut cpb, FaerPar par, FaerMemAlloc mem, FaerFullPivLuParams params) \
	{ \
		(void) par; (void) mem; (void) params; return full_lu_api<T, uint64_t>(A, rpf, rpb, cpf, cpb); \
	} \
	FaerLayout libfaer_v0_23_full_piv_lu_solve_in_place_scratch_u64_##suf(size_t dim, size_t k, FaerPar par) \
	{ \
		(void) par; return layout(dim * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_full_piv_lu_solve_in_place_u64_##suf(FaerMatRef L, FaerMatRef U, FaerConj cj, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; full_lu_solve_api<T, uint64_t>(L, U, rpf, rpb, cpf, cpb, rhs, false); \
	} \
	FaerLayout libfaer_v0_23_full_piv_lu_solve_transpose_in_place_scratch_u64_##suf(size_t dim, size_t k, FaerPar par) \
	{ \
		(void) par; return layout(dim * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_full_piv_lu_solve_transpose_in_place_u64_##suf(FaerMatRef L, FaerMatRef U, FaerConj cj, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; full_lu_solve_api<T, uint64_t>(L, U, rpf, rpb, cpf, cpb, rhs, true); \
	} \
	FaerColPivQrParams libfaer_v0_23_ColPivQrParams_##suf(void) \
	{ \
		return FaerColPivQrParams{48 * 48, 192 * 256}; \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_factor_in_place_scratch_u32_##suf(size_t nrows, size_t ncols, size_t bs, FaerPar par, FaerColPivQrParams params) \
	{ \
		(void) nrows; (void) bs; (void) par; (void) params; return layout(ncols * 2 * sizeof(T), 64); \
	} \
	FaerColPivQrStatus libfaer_v0_23_colpiv_qr_factor_in_place_u32_##suf(FaerMatMut A, FaerMatMut Q_coeff, FaerSliceMut pf, FaerSliceMut pb, FaerPar par, FaerMemAlloc mem, FaerColPivQrParams params) \
	{ \
		(void) par; (void) mem; (void) params; return colpiv_qr_api<T, uint32_t>(A, Q_coeff, pf, pb); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_solve_in_place_scratch_u32_##suf(size_t dim, size_t bs, size_t k, FaerPar par) \
	{ \
		(void) par; return layout(bs * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_colpiv_qr_solve_in_place_u32_##suf(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerConj cj, FaerSliceRef pf, FaerSliceRef pb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; colpiv_qr_solve_api<T, uint32_t>(Qb, Qc, R, pf, pb, rhs, false, true); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_solve_transpose_in_place_scratch_u32_##suf(size_t dim, size_t bs, size_t k, FaerPar par) \
	{ \
		(void) par; return layout(bs * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_colpiv_qr_solve_transpose_in_place_u32_##suf(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerConj cj, FaerSliceRef pf, FaerSliceRef pb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; colpiv_qr_solve_api<T, uint32_t>(Qb, Qc, R, pf, pb, rhs, true, true); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_scratch_u32_##suf(size_t nrows, size_t ncols, size_t bs, size_t k, FaerPar par) \
	{ \
		(void) par; return layout(bs * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_u32_##suf(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerConj cj, FaerSliceRef pf, FaerSliceRef pb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; colpiv_qr_solve_api<T, uint32_t>(Qb, Qc, R, pf, pb, rhs, false, false); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_factor_in_place_scratch_u64_##suf(size_t nrows, size_t ncols, size_t bs, FaerPar par, FaerColPivQrParams params) \
	{ \
		(void) nrows; (void) bs; (void) par; (void) params; return layout(ncols * 2 * sizeof(T), 64); \
	} \
	FaerColPivQrStatus libfaer_v0_23_colpiv_qr_factor_in_place_u64_##suf(FaerMatMut A, FaerMatMut Q_coeff, FaerSliceMut pf, FaerSliceMut pb, FaerPar par, FaerMemAlloc mem, FaerColPivQrParams params) \
	{ \
		(void) par; (void) mem; (void) params; return colpiv_qr_api<T, uint64_t>(A, Q_coeff, pf, pb); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_solve_in_place_scratch_u64_##suf(size_t dim, size_t bs, size_t k, FaerPar par) \
	{ \
		(void) par; return layout(bs * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_colpiv_qr_solve_in_place_u64_##suf(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerConj cj, FaerSliceRef pf, FaerSliceRef pb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; colpiv_qr_solve_api<T, uint64_t>(Qb, Qc, R, pf, pb, rhs, false, true); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_solve_transpose_in_place_scratch_u64_##suf(size_t dim, size_t bs, size_t k, FaerPar par) \
	{ \
		(void) par; return layout(bs * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_colpiv_qr_solve_transpose_in_place_u64_##suf(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerConj cj, FaerSliceRef pf, FaerSliceRef pb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; colpiv_qr_solve_api<T, uint64_t>(Qb, Qc, R, pf, pb, rhs, true, true); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_scratch_u64_##suf(size_t nrows, size_t ncols, size_t bs, size_t k, FaerPar par) \
	{ \
		(void) par; return layout(bs * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_u64_##suf(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerConj cj, FaerSliceRef pf, FaerSliceRef pb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; colpiv_qr_solve_api<T, uint64_t>(Qb, Qc, R, pf, pb, rhs, false, false); \
	} \
	FaerLayout libfaer_v0_23_full_piv_lu_reconstruct_scratch_u32_##suf(size_t nrows, size_t ncols, FaerPar par) \
	{ \
		(void) par; return layout(nrows * ncols * sizeof(T), 64); \
	} \
	void libfaer_v0_23_full_piv_lu_reconstruct_u32_##suf(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; full_lu_reconstruct_api<T, uint32_t>(A, L, U, rpf, rpb, cpf, cpb); \
	} \
	FaerLayout libfaer_v0_23_full_piv_lu_inverse_scratch_u32_##suf(size_t dim, FaerPar par) \
	{ \
		(void) par; return layout(dim * dim * sizeof(T), 64); \
	} \
	void libfaer_v0_23_full_piv_lu_inverse_u32_##suf(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; full_lu_inverse_api<T, uint32_t>(A_inv, L, U, rpf, rpb, cpf, cpb); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_reconstruct_scratch_u32_##suf(size_t nrows, size_t ncols, size_t bs, FaerPar par) \
	{ \
		(void) nrows; (void) par; return layout(bs * ncols * sizeof(T), 64); \
	} \
	void libfaer_v0_23_colpiv_qr_reconstruct_u32_##suf(FaerMatMut A, FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerSliceRef pf, FaerSliceRef pb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; colpiv_qr_reconstruct_api<T, uint32_t>(A, Qb, Qc, R, pf, pb); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_inverse_scratch_u32_##suf(size_t dim, size_t bs, FaerPar par) \
	{ \
		(void) par; return layout(bs * dim * sizeof(T), 64); \
	} \
	void libfaer_v0_23_colpiv_qr_inverse_u32_##suf(FaerMatMut A_inv, FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerSliceRef pf, FaerSliceRef pb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; colpiv_qr_inverse_api<T, uint32_t>(A_inv, Qb, Qc, R, pf, pb); \
	} \
	FaerLayout libfaer_v0_23_full_piv_lu_reconstruct_scratch_u64_##suf(size_t nrows, size_t ncols, FaerPar par) \
	{ \
		(void) par; return layout(nrows * ncols * sizeof(T), 64); \
	} \
	void libfaer_v0_23_full_piv_lu_reconstruct_u64_##suf(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; full_lu_reconstruct_api<T, uint64_t>(A, L, U, rpf, rpb, cpf, cpb); \
	} \
	FaerLayout libfaer_v0_23_full_piv_lu_inverse_scratch_u64_##suf(size_t dim, FaerPar par) \
	{ \
		(void) par; return layout(dim * dim * sizeof(T), 64); \
	} \
	void libfaer_v0_23_full_piv_lu_inverse_u64_##suf(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef rpf, FaerSliceRef rpb, FaerSliceRef cpf, FaerSliceRef cpb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; full_lu_inverse_api<T, uint64_t>(A_inv, L, U, rpf, rpb, cpf, cpb); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_reconstruct_scratch_u64_##suf(size_t nrows, size_t ncols, size_t bs, FaerPar par) \
	{ \
		(void) nrows; (void) par; return layout(bs * ncols * sizeof(T), 64); \
	} \
	void libfaer_v0_23_colpiv_qr_reconstruct_u64_##suf(FaerMatMut A, FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerSliceRef pf, FaerSliceRef pb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; colpiv_qr_reconstruct_api<T, uint64_t>(A, Qb, Qc, R, pf, pb); \
	} \
	FaerLayout libfaer_v0_23_colpiv_qr_inverse_scratch_u64_##suf(size_t dim, size_t bs, FaerPar par) \
	{ \
		(void) par; return layout(bs * dim * sizeof(T), 64); \
	} \
	void libfaer_v0_23_colpiv_qr_inverse_u64_##suf(FaerMatMut A_inv, FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerSliceRef pf, FaerSliceRef pb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; colpiv_qr_inverse_api<T, uint64_t>(A_inv, Qb, Qc, R, pf, pb); \
	} \
	void libfaer_v0_23_inverse_triangular_lower_in_place_##suf(FaerMatMut T_inv, FaerMatRef Tm, FaerPar par) \
	{ \
		(void) par; tri_inverse_api<T>(T_inv, Tm, false, false); \
	} \
	void libfaer_v0_23_inverse_triangular_upper_in_place_##suf(FaerMatMut T_inv, FaerMatRef Tm, FaerPar par) \
	{ \
		(void) par; tri_inverse_api<T>(T_inv, Tm, true, false); \
	} \
	void libfaer_v0_23_inverse_unit_triangular_lower_in_place_##suf(FaerMatMut T_inv, FaerMatRef Tm, FaerPar par) \
	{ \
		(void) par; tri_inverse_api<T>(T_inv, Tm, false, true); \
	} \
	void libfaer_v0_23_inverse_unit_triangular_upper_in_place_##suf(FaerMatMut T_inv, FaerMatRef Tm, FaerPar par) \
	{ \
		(void) par; tri_inverse_api<T>(T_inv, Tm, true, true); \
	} \
	FaerLayout libfaer_v0_23_llt_reconstruct_scratch_##suf(size_t dim, FaerPar par) \
	{ \
		(void) dim; (void) par; return layout(0, 1); \
	} \
	void libfaer_v0_23_llt_reconstruct_##suf(FaerMatMut A, FaerMatRef L, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; llt_reconstruct_api<T>(A, L); \
	} \
	FaerLayout libfaer_v0_23_llt_inverse_scratch_##suf(size_t dim, FaerPar par) \
	{ \
		(void) par; return layout(dim * dim * sizeof(T), 64); \
	} \
	void libfaer_v0_23_llt_inverse_##suf(FaerMatMut A_inv, FaerMatRef L, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; llt_inverse_api<T>(A_inv, L); \
	} \
	FaerLayout libfaer_v0_23_ldlt_reconstruct_scratch_##suf(size_t dim, FaerPar par) \
	{ \
		(void) par; return layout(dim * dim * sizeof(T), 64); \
	} \
	void libfaer_v0_23_ldlt_reconstruct_##suf(FaerMatMut A, FaerMatRef L, FaerVecRef D, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; ldlt_reconstruct_api<T>(A, L, D); \
	} \
	FaerLayout libfaer_v0_23_ldlt_inverse_scratch_##suf(size_t dim, FaerPar par) \
	{ \
		(void) par; return layout(dim * dim * sizeof(T), 64); \
	} \
	void libfaer_v0_23_ldlt_inverse_##suf(FaerMatMut A_inv, FaerMatRef L, FaerVecRef D, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; ldlt_inverse_api<T>(A_inv, L, D); \
	} \
	FaerLayout libfaer_v0_23_partial_piv_lu_reconstruct_scratch_u32_##suf(size_t nrows, size_t ncols, FaerPar par) \
	{ \
		(void) par; return layout(nrows * ncols * sizeof(T), 64); \
	} \
	void libfaer_v0_23_partial_piv_lu_reconstruct_u32_##suf(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef pf, FaerSliceRef pb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; lu_reconstruct_api<T, uint32_t>(A, L, U, pf, pb); \
	} \
	FaerLayout libfaer_v0_23_partial_piv_lu_inverse_scratch_u32_##suf(size_t dim, FaerPar par) \
	{ \
		(void) par; return layout(dim * dim * sizeof(T), 64); \
	} \
	void libfaer_v0_23_partial_piv_lu_inverse_u32_##suf(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef pf, FaerSliceRef pb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; lu_inverse_api<T, uint32_t>(A_inv, L, U, pf, pb); \
	} \
	FaerLayout libfaer_v0_23_partial_piv_lu_reconstruct_scratch_u64_##suf(size_t nrows, size_t ncols, FaerPar par) \
	{ \
		(void) par; return layout(nrows * ncols * sizeof(T), 64); \
	} \
	void libfaer_v0_23_partial_piv_lu_reconstruct_u64_##suf(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef pf, FaerSliceRef pb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; lu_reconstruct_api<T, uint64_t>(A, L, U, pf, pb); \
	} \
	FaerLayout libfaer_v0_23_partial_piv_lu_inverse_scratch_u64_##suf(size_t dim, FaerPar par) \
	{ \
		(void) par; return layout(dim * dim * sizeof(T), 64); \
	} \
	void libfaer_v0_23_partial_piv_lu_inverse_u64_##suf(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef pf, FaerSliceRef pb, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; lu_inverse_api<T, uint64_t>(A_inv, L, U, pf, pb); \
	} \
	FaerLayout libfaer_v0_23_qr_reconstruct_scratch_##suf(size_t nrows, size_t ncols, size_t bs, FaerPar par) \
	{ \
		(void) nrows; (void) par; return layout(bs * ncols * sizeof(T), 64); \
	} \
	void libfaer_v0_23_qr_reconstruct_##suf(FaerMatMut A, FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; qr_reconstruct_api<T>(A, Qb, Qc, R); \
	} \
	FaerLayout libfaer_v0_23_qr_inverse_scratch_##suf(size_t dim, size_t bs, FaerPar par) \
	{ \
		(void) par; return layout(bs * dim * sizeof(T), 64); \
	} \
	void libfaer_v0_23_qr_inverse_##suf(FaerMatMut A_inv, FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) par; (void) mem; qr_inverse_api<T>(A_inv, Qb, Qc, R); \
	} \
	FaerLayout libfaer_v0_23_apply_householder_on_the_right_scratch_##suf(size_t dim, size_t bs, size_t k) \
	{ \
		(void) dim; return layout(bs * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_apply_householder_on_the_right_##suf(FaerMatRef V, FaerMatRef H, FaerConj cj, FaerMatMut M, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; apply_hh_right_api<T>(V, H, M, false); \
	} \
	FaerLayout libfaer_v0_23_apply_householder_transpose_on_the_right_scratch_##suf(size_t dim, size_t bs, size_t k) \
	{ \
		(void) dim; return layout(bs * k * sizeof(T), 64); \
	} \
	void libfaer_v0_23_apply_householder_transpose_on_the_right_##suf(FaerMatRef V, FaerMatRef H, FaerConj cj, FaerMatMut M, FaerPar par, FaerMemAlloc mem) \
	{ \
		(void) cj; (void) par; (void) mem; apply_hh_right_api<T>(V, H, M, true); \
	} \
	FaerQrParams libfaer_v0_23_QrParams_##suf(void) { return FaerQrParams{48 * 48, 192 * 256}; }                    \
	size_t libfaer_v0_23_qr_recommended_block_size_##suf(size_t nrows, size_t ncols)                                \
	{                                                                                                              \
		return qr_block_size(nrows, ncols);                                                                    \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_qr_factor_in_place_scratch_##suf(size_t nrows, size_t ncols, size_t bs, FaerPar par,   \
								  FaerQrParams params)                                      \
	{                                                                                                              \
		(void) nrows;                                                                                          \
		(void) par;                                                                                            \
		(void) params;                                                                                         \
		return layout(bs * ncols * sizeof(T), 64);                                                             \
	}                                                                                                              \
	FaerQrStatus libfaer_v0_23_qr_factor_in_place_##suf(FaerMatMut A, FaerMatMut Q, FaerPar par, FaerMemAlloc mem,  \
							    FaerQrParams params)                                            \
	{                                                                                                              \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		return qr_api<T>(A, Q, params);                                                                        \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_apply_householder_on_the_left_scratch_##suf(size_t dim, size_t bs, size_t k)           \
	{                                                                                                              \
		(void) dim;                                                                                            \
		return layout(bs * k * sizeof(T), 64);                                                                 \
	}                                                                                                              \
	void libfaer_v0_23_apply_householder_on_the_left_##suf(FaerMatRef V, FaerMatRef H, FaerConj cj, FaerMatMut rhs, \
							       FaerPar par, FaerMemAlloc mem)                               \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		apply_hh_api<T>(V, H, rhs, false);                                                                     \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_apply_householder_transpose_on_the_left_scratch_##suf(size_t dim, size_t bs, size_t k) \
	{                                                                                                              \
		(void) dim;                                                                                            \
		return layout(bs * k * sizeof(T), 64);                                                                 \
	}                                                                                                              \
	void libfaer_v0_23_apply_householder_transpose_on_the_left_##suf(FaerMatRef V, FaerMatRef H, FaerConj cj,       \
									 FaerMatMut rhs, FaerPar par, FaerMemAlloc mem)     \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		apply_hh_api<T>(V, H, rhs, true);                                                                      \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_partial_piv_lu_solve_in_place_scratch_u32_##suf(size_t dim, size_t k, FaerPar par)     \
	{                                                                                                              \
		(void) par;                                                                                            \
		return layout(dim * k * sizeof(T), 64);                                                                \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_partial_piv_lu_solve_in_place_scratch_u64_##suf(size_t dim, size_t k, FaerPar par)     \
	{                                                                                                              \
		(void) par;                                                                                            \
		return layout(dim * k * sizeof(T), 64);                                                                \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_scratch_u32_##suf(size_t dim, size_t k, FaerPar par) \
	{                                                                                                              \
		(void) par;                                                                                            \
		return layout(dim * k * sizeof(T), 64);                                                                \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_scratch_u64_##suf(size_t dim, size_t k, FaerPar par) \
	{                                                                                                              \
		(void) par;                                                                                            \
		return layout(dim * k * sizeof(T), 64);                                                                \
	}                                                                                                              \
	void libfaer_v0_23_partial_piv_lu_solve_in_place_u32_##suf(FaerMatRef L, FaerMatRef U, FaerConj cj, FaerSliceRef pf, \
								   FaerSliceRef pb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		lu_solve_api<T, uint32_t>(L, U, pf, pb, rhs, false);                                                   \
	}                                                                                                              \
	void libfaer_v0_23_partial_piv_lu_solve_in_place_u64_##suf(FaerMatRef L, FaerMatRef U, FaerConj cj, FaerSliceRef pf, \
								   FaerSliceRef pb, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem) \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		lu_solve_api<T, uint64_t>(L, U, pf, pb, rhs, false);                                                   \
	}                                                                                                              \
	void libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_u32_##suf(FaerMatRef L, FaerMatRef U, FaerConj cj,   \
									     FaerSliceRef pf, FaerSliceRef pb, FaerMatMut rhs, \
									     FaerPar par, FaerMemAlloc mem)                \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		lu_solve_api<T, uint32_t>(L, U, pf, pb, rhs, true);                                                    \
	}                                                                                                              \
	void libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_u64_##suf(FaerMatRef L, FaerMatRef U, FaerConj cj,   \
									     FaerSliceRef pf, FaerSliceRef pb, FaerMatMut rhs, \
									     FaerPar par, FaerMemAlloc mem)                \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		lu_solve_api<T, uint64_t>(L, U, pf, pb, rhs, true);                                                    \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_qr_solve_in_place_scratch_##suf(size_t dim, size_t bs, size_t k, FaerPar par)          \
	{                                                                                                              \
		(void) dim;                                                                                            \
		(void) par;                                                                                            \
		return layout(bs * k * sizeof(T), 64);                                                                 \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_qr_solve_transpose_in_place_scratch_##suf(size_t dim, size_t bs, size_t k, FaerPar par) \
	{                                                                                                              \
		(void) dim;                                                                                            \
		(void) par;                                                                                            \
		return layout(bs * k * sizeof(T), 64);                                                                 \
	}                                                                                                              \
	FaerLayout libfaer_v0_23_qr_solve_lstsq_in_place_scratch_##suf(size_t nrows, size_t ncols, size_t bs, size_t k, \
								       FaerPar par)                                         \
	{                                                                                                              \
		(void) nrows;                                                                                          \
		(void) ncols;                                                                                          \
		(void) par;                                                                                            \
		return layout(bs * k * sizeof(T), 64);                                                                 \
	}                                                                                                              \
	void libfaer_v0_23_qr_solve_in_place_##suf(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerConj cj, FaerMatMut rhs, \
						   FaerPar par, FaerMemAlloc mem)                                           \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		qr_solve_api<T>(Qb, Qc, R, rhs, false, true);                                                          \
	}                                                                                                              \
	void libfaer_v0_23_qr_solve_transpose_in_place_##suf(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerConj cj,   \
							     FaerMatMut rhs, FaerPar par, FaerMemAlloc mem)                 \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		qr_solve_api<T>(Qb, Qc, R, rhs, true, true);                                                           \
	}                                                                                                              \
	void libfaer_v0_23_qr_solve_lstsq_in_place_##suf(FaerMatRef Qb, FaerMatRef Qc, FaerMatRef R, FaerConj cj,       \
							 FaerMatMut rhs, FaerPar par, FaerMemAlloc mem)                     \
	{                                                                                                              \
		(void) cj;                                                                                             \
		(void) par;                                                                                            \
		(void) mem;                                                                                            \
		qr_solve_api<T>(Qb, Qc, R, rhs, false, false);                                                         \
	}
FH_FOR_DTYPES(X)
#undef X

FaerPar libfaer_v0_23_get_global_par(void)
{
	FaerPar p;
	p.tag = (FaerParTag) g_par_tag.load();
	p.nthreads = p.tag == FaerParTag_Seq ? 1 : g_par_threads.load();
	return p;
}
void libfaer_v0_23_set_global_par(FaerPar par)
{
	g_par_tag.store((int) par.tag);
	g_par_threads.store(par.nthreads);
}

// ---------------------------------------------------------------------------------------- runtime control
const char *faer_hip_version(void) { return "faer_hip 0.1 gfx950"; }

int faer_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		(void) hipGetLastError();
		return 0;
	}
	int ok = 0;
	for (int d = 0; d < n; ++d) {
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0)
			++ok;
	}
	return ok;
}
void faer_hip_set_device(int device)
{
	FH_HIP(hipSetDevice(device));
	Ctx &c = ctx();
	c.device = device;
}
void faer_hip_set_stream(void *s) { ctx().set_stream(static_cast<hipStream_t>(s)); }
void *faer_hip_get_stream(void) { return ctx().stream; }
void faer_hip_synchronize(void) { ctx().sync(); }
void faer_hip_shutdown(void) { fh::ctx_shutdown(); }
void *faer_hip_malloc(size_t bytes)
{
	ctx();
	void *p = nullptr;
	FH_HIP(hipMalloc(&p, bytes ? bytes : 1));
	return p;
}
void faer_hip_free(void *p)
{
	if (p)
		FH_HIP(hipFree(p));
}
void faer_hip_memcpy_h2d(void *d, const void *s, size_t bytes)
{
	FH_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, ctx().stream));
	ctx().sync();
}
void faer_hip_memcpy_d2h(void *d, const void *s, size_t bytes)
{
	FH_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, ctx().stream));
	ctx().sync();
}
void faer_hip_set_gemm_variant(int v) { ctx().gemm_variant = v; }
void faer_hip_debug_stream_xcc(int which, int nblocks, unsigned *out_host) { debug_stream_xcc(which, nblocks, out_host); }
size_t faer_hip_debug_llt_plan(size_t n, size_t tail_rows, size_t nb2, size_t *starts, size_t cap)
{
	const std::vector<idx_t> J = llt_plan((idx_t) n, (idx_t) tail_rows, (idx_t) nb2);
	for (size_t i = 0; i < J.size() && i < cap; ++i)
		starts[i] = (size_t) J[i];
	return J.size();
}
void faer_hip_debug_dump_timing(void)
{
	trsm_dump_timing();
	lu_dump_timing();
	gemm_dump_timing();
}

void faer_hip_tridiag_in_place_f64(FaerMatMut A, FaerMatMut householder) { tridiag_api<double>(A, householder); }
void faer_hip_tridiag_in_place_f32(FaerMatMut A, FaerMatMut householder) { tridiag_api<float>(A, householder); }
void faer_hip_bidiag_in_place_f64(FaerMatMut A, FaerMatMut Hl, FaerMatMut Hr) { bidiag_api<double>(A, Hl, Hr); }
void faer_hip_bidiag_in_place_f32(FaerMatMut A, FaerMatMut Hl, FaerMatMut Hr) { bidiag_api<float>(A, Hl, Hr); }
void faer_hip_hessenberg_in_place_f64(FaerMatMut A, FaerMatMut householder) { hessenberg_api<double>(A, householder); }
void faer_hip_hessenberg_in_place_f32(FaerMatMut A, FaerMatMut householder) { hessenberg_api<float>(A, householder); }

int faer_hip_debug_lu_leaf_width(size_t nrows, FaerHipDType dtype, int resident_workgroups)
{
	return lu_leaf_width((idx_t) nrows, dtype == FaerHipDType_F64 ? 8 : 4, resident_workgroups);
}
int faer_hip_debug_dist_two_streams_ok(size_t panel_rows, FaerHipDType dtype, int panel_cus, int all_cus)
{
	return dist_two_streams_ok((idx_t) panel_rows, dtype == FaerHipDType_F64 ? 8 : 4, panel_cus, all_cus) ? 1 : 0;
}
void faer_hip_debug_lu_force_general(int on) { lu_force_general(on); }
void faer_hip_debug_lend_cus(int on) { g_lend_cus.store(on); }
void faer_hip_debug_lu_plan(size_t nb2_from, size_t pipe_from, size_t la_min_cols) { lu_debug_plan((long) nb2_from, (long) pipe_from, (long) la_min_cols); }
long faer_hip_debug_qr_one_pass_columns(void) { return qr_last_one_pass_columns(); }
void faer_hip_debug_qr_fused(int on) { tsqr_debug_fused(on); }
void faer_hip_debug_qr_one_pass_f64(int on) { tsqr_debug_f64(on); }
void faer_hip_debug_qr_panels_one_pass(int on) { tsqr_debug_panels(on); }
void faer_hip_debug_qr_one_pass_shape_rule(long min_rows, long min_rows_per_column) { tsqr_debug_shape_rule(min_rows, min_rows_per_column); }
void faer_hip_debug_fplu_inplace(int on) { fplu_debug_inplace(on); }
void faer_hip_debug_level2_force_memory_bodies(int on) { level2_debug_force_memory_bodies(on); }
void *faer_hip_debug_internal_stream(int which)
{
	Ctx &c = ctx();
	FH_CHECK(c.lookahead_streams(), "debug_internal_stream: look-ahead streams unavailable");
	return which == 1 ? (void *) c.la_bulk : (void *) c.la_panel;
}

double faer_hip_time_gemm_ms(FaerHipDType dtype, size_t m, size_t n, size_t k, void *dst, ptrdiff_t dst_cs, const void *lhs,
			     ptrdiff_t lhs_cs, const void *rhs, ptrdiff_t rhs_cs, int iters)
{
	FH_CHECK(is_device_ptr(dst) && is_device_ptr(lhs) && is_device_ptr(rhs), "time_gemm: operands must be device memory");
	if (dtype == FaerHipDType_F64)
		return gemm_time_ms<double>(MatV<double>{(double *) dst, (idx_t) m, (idx_t) n, 1, (idx_t) dst_cs},
					    MatV<const double>{(const double *) lhs, (idx_t) m, (idx_t) k, 1, (idx_t) lhs_cs},
					    MatV<const double>{(const double *) rhs, (idx_t) k, (idx_t) n, 1, (idx_t) rhs_cs}, iters);
	FH_CHECK(dtype == FaerHipDType_F32, "time_gemm: f32/f64 only");
	return gemm_time_ms<float>(MatV<float>{(float *) dst, (idx_t) m, (idx_t) n, 1, (idx_t) dst_cs},
				   MatV<const float>{(const float *) lhs, (idx_t) m, (idx_t) k, 1, (idx_t) lhs_cs},
				   MatV<const float>{(const float *) rhs, (idx_t) k, (idx_t) n, 1, (idx_t) rhs_cs}, iters);
}
double faer_hip_mfma_peak_tflops(FaerHipDType dtype, int iters) { return mfma_peak_tflops(dtype == FaerHipDType_F64, iters); }

void faer_hip_prof_begin(void)
{
	Ctx &c = ctx();
	double scratch[Ctx::PROF_CLASSES * 3];
	prof_collect(scratch); // (drops spans of an unfinished profile)
	c.prof_on = true;
}
void faer_hip_prof_end(double *out18)
{
	FH_CHECK(out18 != nullptr, "prof_end: NULL output");
	Ctx &c = ctx();
	c.prof_on = false;
	FH_HIP(hipDeviceSynchronize());
	prof_collect(out18);
}
size_t faer_hip_prof_end_spans(double *out18, double *spans, size_t cap)
{
	FH_CHECK(out18 != nullptr && (spans != nullptr || cap == 0), "prof_end_spans: NULL output");
	Ctx &c = ctx();
	c.prof_on = false;
	FH_HIP(hipDeviceSynchronize());
	size_t n = 0;
	prof_collect(out18, spans, cap, &n);
	return n;
}
double faer_hip_xwg_hop_us(int iters) { return xwg_hop_us(iters); }
void faer_hip_partial_piv_lu_lend_copy(const void *device_copy, size_t nrows, size_t ncols, int elem_bytes)
{
	lu_lend_copy(device_copy, (idx_t) nrows, (idx_t) ncols, elem_bytes);
}

} // extern "C"
