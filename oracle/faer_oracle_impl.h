/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Type-generic body of the oracle; included twice by faer_oracle.c with
 *   T    = double / float
 *   FN() = name##_f64 / name##_f32
 *
 * Every routine restates, in plain C, the algorithm that faer 0.24.4 runs on
 * the CPU for the hot path.  Citations are
 * file:line under /root/reference/faer/src/linalg unless stated otherwise.
 *
 * Parity status: the GEMM arithmetic of the reference lives in unvendored
 * crates (private-gemm-x86 0.1.20 / gemm 0.19.0 / nano-gemm 0.2.2), so no
 * golden vector pins its summation order => bitwise parity is UNPINNED and
 * everything here is pinned at tolerance level only: against the reference's
 * known-answer test (qr/mod.rs:116-191), its 2x2 matmul doctests
 * (matmul/mod.rs:1595-1614) and LAPACK (scipy) residual checks.
 */

typedef struct {
	T *p;
	long nrows, ncols, rs, cs;
} FN(mat);

#define AT(M, i, j) ((M).p[(long)(i) * (M).rs + (long)(j) * (M).cs])

static inline FN(mat) FN(sub)(FN(mat) a, long r0, long c0, long nr, long nc)
{
	FN(mat) r;
	r.p = a.p + r0 * a.rs + c0 * a.cs;
	r.nrows = nr;
	r.ncols = nc;
	r.rs = a.rs;
	r.cs = a.cs;
	return r;
}
static inline FN(mat) FN(tr)(FN(mat) a)
{
	FN(mat) r = a;
	r.nrows = a.ncols;
	r.ncols = a.nrows;
	r.rs = a.cs;
	r.cs = a.rs;
	return r;
}
static inline FN(mat) FN(rev_rows)(FN(mat) a)
{
	FN(mat) r = a;
	if (a.nrows > 0)
		r.p = a.p + (a.nrows - 1) * a.rs;
	r.rs = -a.rs;
	return r;
}
static inline FN(mat) FN(rev_rows_cols)(FN(mat) a)
{
	FN(mat) r = FN(rev_rows)(a);
	if (a.ncols > 0)
		r.p = r.p + (a.ncols - 1) * a.cs;
	r.cs = -a.cs;
	return r;
}

/* ------------------------------------------------------------------ GEMM */
/* matmul/mod.rs:1176-1560 (contract) + :58-329 (in-tree micro-kernel):
 * dst <- [dst +] alpha * lhs * rhs.  Replace never reads dst (K==0 => zero
 * fill, :1194-1196).  Accumulation is a k-ordered FMA chain starting at 0,
 * combined once as alpha*acc [+ dst] (single K chunk). */
static void FN(gemm)(FN(mat) dst, int accum_add, FN(mat) lhs, FN(mat) rhs, T alpha)
{
	long m = dst.nrows, n = dst.ncols, k = lhs.ncols;
	if (m == 0 || n == 0)
		return;
#pragma omp parallel for schedule(static) if ((double)m * n * k > 1e6)
	for (long j = 0; j < n; j++) {
		T *tmp = (T *)malloc(sizeof(T) * (size_t)m);
		for (long i = 0; i < m; i++)
			tmp[i] = 0;
		for (long p = 0; p < k; p++) {
			T b = AT(rhs, p, j);
			if (lhs.rs == 1) {
				const T *a = lhs.p + p * lhs.cs;
				for (long i = 0; i < m; i++)
					tmp[i] = FMA(a[i], b, tmp[i]);
			} else {
				for (long i = 0; i < m; i++)
					tmp[i] = FMA(AT(lhs, i, p), b, tmp[i]);
			}
		}
		for (long i = 0; i < m; i++) {
			if (accum_add)
				AT(dst, i, j) = FMA(alpha, tmp[i], AT(dst, i, j));
			else
				AT(dst, i, j) = alpha * tmp[i];
		}
		free(tmp);
	}
}

/* -------------------------------------------------- triangular product */
/* triangular.rs:906-977 (BlockStructure), :1246-1495 (semantics).
 * structure codes follow faer-ffi/src/lib.rs:84-93:
 * 0 Rect, 1 TriLower, 2 TriUpper, 3 StrictLower, 4 StrictUpper, 5 UnitLower,
 * 6 UnitUpper.  Only the structured part of each operand is ACCESSED; only
 * the structured part of dst is WRITTEN (strict/unit => diagonal untouched). */
static inline int FN(in_struct)(int s, long i, long j)
{
	switch (s) {
	case 0: return 1;
	case 1: return i >= j;
	case 2: return i <= j;
	case 3: case 5: return i > j;
	case 4: case 6: return i < j;
	}
	return 0;
}
static inline T FN(sval)(FN(mat) a, int s, long i, long j)
{
	if ((s == 5 || s == 6) && i == j)
		return (T)1;
	if (!FN(in_struct)(s, i, j))
		return (T)0;
	return AT(a, i, j);
}
static void FN(matmul_triangular)(FN(mat) dst, int dst_s, int accum_add, FN(mat) lhs, int lhs_s,
				  FN(mat) rhs, int rhs_s, T alpha)
{
	long m = dst.nrows, n = dst.ncols, k = lhs.ncols;
	/* every (i, j) is an independent k-ordered chain: threading over columns does not change any result */
#pragma omp parallel for schedule(dynamic, 4) if ((double)m * n * k > 1e6)
	for (long j = 0; j < n; j++)
		for (long i = 0; i < m; i++) {
			if (!FN(in_struct)(dst_s, i, j))
				continue;
			T acc = 0;
			for (long p = 0; p < k; p++) {
				T a = FN(sval)(lhs, lhs_s, i, p);
				T b = FN(sval)(rhs, rhs_s, p, j);
				if (a != 0 && b != 0)
					acc = FMA(a, b, acc);
			}
			if (accum_add)
				AT(dst, i, j) = FMA(alpha, acc, AT(dst, i, j));
			else
				AT(dst, i, j) = alpha * acc;
		}
}

/* ------------------------------------------------------------------ TRSM */
/* triangular_solve.rs:200-215 block_size(), :16-198 base cases n<=4,
 * :420-604 recursion.  Solves tril * X = rhs in place (left side). */
static long FN(trsm_block_size)(long n)
{
	long base_rem = n / 2, r;
	if (n >= 32)
		r = (base_rem + 15) / 16 * 16;
	else if (n >= 16)
		r = (base_rem + 7) / 8 * 8;
	else if (n >= 8)
		r = (base_rem + 3) / 4 * 4;
	else
		r = base_rem;
	return n - r;
}
static void FN(trsm_lower)(FN(mat) tril, int unit, FN(mat) rhs)
{
	long n = tril.nrows, k = rhs.ncols;
	if (n == 0 || k == 0)
		return;
	if (k > 64 && n <= 128) { /* :430-449 */
		FN(trsm_lower)(tril, unit, FN(sub)(rhs, 0, 0, n, k / 2));
		FN(trsm_lower)(tril, unit, FN(sub)(rhs, 0, k / 2, n, k - k / 2));
		return;
	}
	if (n <= 4) {
		/* :98-198 : x_i <- x_i*(1/l_ii) + sum_j (-l_ij/l_ii) x_j  (non-unit)
		 * :16-96  : x_i <- x_i - sum_j l_ij x_j                   (unit)   */
		T inv[4], c[4][4];
		for (long i = 0; i < n; i++) {
			inv[i] = unit ? (T)1 : (T)1 / AT(tril, i, i);
			for (long j = 0; j < i; j++)
				c[i][j] = unit ? -AT(tril, i, j) : (-AT(tril, i, j)) * inv[i];
		}
		for (long col = 0; col < k; col++) {
			T y[4];
			for (long i = 0; i < n; i++) {
				T v = AT(rhs, i, col);
				if (!unit)
					v = v * inv[i];
				for (long j = 0; j < i; j++)
					v = v + c[i][j] * y[j];
				y[i] = v;
			}
			for (long i = 0; i < n; i++)
				AT(rhs, i, col) = y[i];
		}
		return;
	}
	long bs = FN(trsm_block_size)(n);
	FN(mat) tl = FN(sub)(tril, 0, 0, bs, bs);
	FN(mat) bl = FN(sub)(tril, bs, 0, n - bs, bs);
	FN(mat) br = FN(sub)(tril, bs, bs, n - bs, n - bs);
	FN(mat) top = FN(sub)(rhs, 0, 0, bs, k);
	FN(mat) bot = FN(sub)(rhs, bs, 0, n - bs, k);
	FN(trsm_lower)(tl, unit, top);
	FN(gemm)(bot, 1, bl, top, (T)-1);
	FN(trsm_lower)(br, unit, bot);
}
/* :578-604 upper = lower on the row/col reversed views */
static void FN(trsm_upper)(FN(mat) triu, int unit, FN(mat) rhs)
{
	FN(trsm_lower)(FN(rev_rows_cols)(triu), unit, FN(rev_rows)(rhs));
}

/* --------------------------------------------------------------- Cholesky */
/* cholesky/ldlt/factor.rs:299-366 (fallback == simd base semantics, :7-177):
 * left-looking; d = a_jj - sum; regularise; !(d>0) => Err(j); column
 * (including the diagonal) is multiplied by 1/sqrt(d). Returns count or
 * -(j+1) on failure. */
static long FN(chol_base)(FN(mat) A, T *D, int regularize, T eps, T delta)
{
	long n = A.nrows, count = 0;
	for (long j = 0; j < n; j++) {
		for (long i = j; i < n; i++) {
			T sum = 0;
			for (long k = 0; k < j; k++)
				sum = sum + AT(A, j, k) * AT(A, i, k);
			AT(A, i, j) = AT(A, i, j) - sum;
		}
		T diag = AT(A, j, j);
		if (regularize) {
			int small_or_negative = diag <= eps;
			if (small_or_negative) { /* sign == 1 for llt, :122-131 */
				diag = delta;
				count += 1;
			}
		}
		if (!(diag > 0)) {
			D[j] = diag;
			return -(j + 1);
		}
		diag = SQRT(diag);
		D[j] = diag;
		if (diag == 0 || !isfinite(diag))
			return -(j + 1);
		T inv = (T)1 / diag;
		for (long i = j; i < n; i++)
			AT(A, i, j) = AT(A, i, j) * inv;
	}
	return count;
}
static long FN(next_pow2)(long n)
{
	long p = 1;
	while (p < n)
		p <<= 1;
	return p;
}
/* cholesky/ldlt/factor.rs:367-498 (is_llt branch) */
static long FN(chol_rec)(FN(mat) A, T *D, long rec_threshold, long block_size, int regularize, T eps,
			 T delta)
{
	long n = A.ncols;
	if (n <= rec_threshold)
		return FN(chol_base)(A, D, regularize, eps, delta);
	long count = 0;
	long bs0 = FN(next_pow2)(n) / 2;
	if (bs0 > block_size)
		bs0 = block_size;
	long j = 0;
	while (j < n) {
		long bs = bs0 < n - j ? bs0 : n - j;
		FN(mat) A00 = FN(sub)(A, j, j, bs, bs);
		FN(mat) A10 = FN(sub)(A, j + bs, j, n - j - bs, bs);
		FN(mat) A11 = FN(sub)(A, j + bs, j + bs, n - j - bs, n - j - bs);
		long r = FN(chol_rec)(A00, D + j, rec_threshold, bs, regularize, eps, delta);
		if (r < 0)
			return -(j + (-r - 1) + 1);
		count += r;
		/* :422-426  A10^T <- L00^-1 A10^T  (i.e. A10 <- A10 L00^-T) */
		FN(trsm_lower)(A00, 0, FN(tr)(A10));
		/* :436-446  lower(A11) -= A10 A10^T */
		FN(matmul_triangular)(A11, 1, 1, A10, 0, FN(tr)(A10), 0, (T)-1);
		j += bs;
	}
	return count;
}
/* ------------------------------------------------------------------ LDLT */
/* cholesky/ldlt/factor.rs:299-366 with is_llt == false: a_ij -= sum_k a_jk a_ik D_k; the pivot is regularised
 * according to its expected sign (:318-340: only the sign == +1 correction is counted); a zero or non finite
 * pivot is an error (ZeroPivot); the column, diagonal included, is multiplied by 1/d. */
static long FN(ldlt_base)(FN(mat) A, T *D, int regularize, T eps, T delta, const signed char *signs)
{
	long n = A.nrows, count = 0;
	for (long j = 0; j < n; j++) {
		for (long i = j; i < n; i++) {
			T sum = 0;
			for (long k = 0; k < j; k++)
				sum = sum + AT(A, j, k) * (AT(A, i, k) * D[k]);
			AT(A, i, j) = AT(A, i, j) - sum;
		}
		T diag = AT(A, j, j);
		if (regularize) {
			int sign = signs ? (int)signs[j] : 0;
			int small_or_negative = diag <= eps;
			int minus_small_or_positive = diag >= -eps;
			if (sign == 1 && small_or_negative) {
				diag = delta;
				count += 1;
			} else if (sign == -1 && minus_small_or_positive) {
				diag = -delta;
			} else if (small_or_negative && minus_small_or_positive) {
				diag = diag < 0 ? -delta : delta;
			}
		}
		D[j] = diag;
		if (diag == 0 || !isfinite(diag))
			return -(j + 1);
		T inv = (T)1 / diag;
		for (long i = j; i < n; i++)
			AT(A, i, j) = AT(A, i, j) * inv;
	}
	return count;
}
/* cholesky/ldlt/factor.rs:367-498, !is_llt branch (:428-434 unit lower solve, :447-470 scaling by 1/D and the
 * diagonally weighted update).  The public entry point runs the left-looking sibling (:499-659), which
 * computes the same factors in a different summation order. */
static long FN(ldlt_rec)(FN(mat) A, T *D, long rec_threshold, long block_size, int regularize, T eps, T delta,
			 const signed char *signs)
{
	long n = A.ncols;
	if (n <= rec_threshold)
		return FN(ldlt_base)(A, D, regularize, eps, delta, signs);
	long count = 0;
	long bs0 = FN(next_pow2)(n) / 2;
	if (bs0 > block_size)
		bs0 = block_size;
	long j = 0;
	while (j < n) {
		long bs = bs0 < n - j ? bs0 : n - j;
		long r1 = n - j - bs;
		FN(mat) A00 = FN(sub)(A, j, j, bs, bs);
		FN(mat) A10 = FN(sub)(A, j + bs, j, r1, bs);
		FN(mat) A11 = FN(sub)(A, j + bs, j + bs, r1, r1);
		long r = FN(ldlt_rec)(A00, D + j, rec_threshold, bs, regularize, eps, delta, signs ? signs + j : 0);
		if (r < 0)
			return -(j + (-r - 1) + 1);
		count += r;
		FN(trsm_lower)(A00, 1, FN(tr)(A10)); /* A10 <- A10 L00^-T (unit lower) = L10 D0 */
		/* A11(lower) -= (L10 D0) L10^T, then A10 <- L10 */
		for (long jj = 0; jj < r1; jj++)
			for (long ii = jj; ii < r1; ii++) {
				T acc = 0;
				for (long k = 0; k < bs; k++)
					acc = FMA(AT(A10, ii, k), AT(A10, jj, k) * ((T)1 / D[j + k]), acc);
				AT(A11, ii, jj) = AT(A11, ii, jj) - acc;
			}
		for (long k = 0; k < bs; k++) {
			T d = (T)1 / D[j + k];
			for (long i = 0; i < r1; i++)
				AT(A10, i, k) = AT(A10, i, k) * d;
		}
		j += bs;
	}
	return count;
}
/* cholesky/ldlt/factor.rs:742-800 cholesky_in_place: L (unit lower, strictly below the diagonal) and D (on the
 * diagonal, written for the columns 0 .. index on failure).  Returns the regularization count or -(index+1). */
long FN(oracle_ldlt_in_place)(T *a, long n, long rs, long cs, T reg_delta, T reg_eps, const signed char *signs,
			      long rec_threshold, long block_size)
{
	FN(mat) A = {a, n, n, rs, cs};
	T *D = (T *)malloc(sizeof(T) * (size_t)(n > 0 ? n : 1));
	int regularize = (reg_delta > 0 && reg_eps > 0);
	long r = FN(ldlt_rec)(A, D, rec_threshold, block_size, regularize, reg_eps, reg_delta, signs);
	long init = r < 0 ? -r : n; /* index + 1 */
	for (long i = 0; i < init; i++)
		AT(A, i, i) = D[i];
	free(D);
	return r;
}

/* cholesky/llt/factor.rs:67-97.  Returns dynamic_regularization_count (>=0)
 * or -(index+1) for NonPositivePivot{index}. */
long FN(oracle_llt_in_place)(T *a, long n, long rs, long cs, T reg_delta, T reg_eps, long rec_threshold,
			     long block_size)
{
	FN(mat) A = {a, n, n, rs, cs};
	T *D = (T *)malloc(sizeof(T) * (size_t)(n > 0 ? n : 1));
	int regularize = (reg_delta > 0 && reg_eps > 0);
	long r = FN(chol_rec)(A, D, rec_threshold, block_size, regularize, reg_eps, reg_delta);
	free(D);
	return r;
}

/* --------------------------------------------------------------------- LU */
/* lu/partial_pivoting/factor.rs:19-67 */
static long FN(lu_unblocked)(FN(mat) M, long start, long end, long *trans)
{
	long m = M.nrows;
	long n_trans = 0;
	for (long j = start; j < end; j++) {
		long col = j, row = j - start;
		long imax = row;
		T max = 0;
		for (long i = imax; i < m; i++) {
			T a = FABS(AT(M, i, col));
			if (a > max) {
				max = a;
				imax = i;
			}
		}
		trans[row] = imax - row;
		if (imax != row) {
			for (long c = 0; c < M.ncols; c++) { /* perm/mod.rs:98-107 full rows */
				T t = AT(M, row, c);
				AT(M, row, c) = AT(M, imax, c);
				AT(M, imax, c) = t;
			}
			n_trans += 1;
		}
		FN(mat) S = FN(sub)(M, 0, start, m, end - start);
		T inv = (T)1 / AT(S, row, row);
		for (long i = row + 1; i < m; i++)
			AT(S, i, row) = AT(S, i, row) * inv;
		/* rank-1 update via matmul K=1 (:54-64) -> rank_update_imp
		 * (matmul/mod.rs:1038-1136): dst = fma(x_i, alpha*y_j, dst) */
		for (long c = row + 1; c < end - start; c++) {
			T y = (T)-1 * AT(S, row, c);
			for (long i = row + 1; i < m; i++)
				AT(S, i, c) = FMA(AT(S, i, row), y, AT(S, i, c));
		}
	}
	return n_trans;
}
/* lu/partial_pivoting/factor.rs:68-187 */
static long FN(lu_rec)(FN(mat) A, long start, long end, long *trans, long rec_threshold)
{
	long m = A.nrows, ncols = A.ncols, n = end - start;
	if (n <= rec_threshold)
		return FN(lu_unblocked)(A, start, end, trans);
	long half = n / 2;
	long pow = FN(next_pow2)(half);
	if (pow > 16)
		pow = 16;
	long bs = (half + pow - 1) / pow * pow;
	long n_trans = 0;
	FN(mat) P = FN(sub)(A, 0, start, m, n);
	n_trans += FN(lu_rec)(P, 0, bs, trans, rec_threshold);
	{
		FN(mat) A00 = FN(sub)(P, 0, 0, bs, bs);
		FN(mat) A01 = FN(sub)(P, 0, bs, bs, n - bs);
		FN(mat) A10 = FN(sub)(P, bs, 0, m - bs, bs);
		FN(mat) A11 = FN(sub)(P, bs, bs, m - bs, n - bs);
		FN(trsm_lower)(A00, 1, A01);
		FN(gemm)(A11, 1, A10, A01, (T)-1);
		n_trans += FN(lu_rec)(FN(sub)(P, bs, 0, m - bs, n), bs, n, trans + bs, rec_threshold);
	}
	/* :127-185 deferred swaps on the columns outside [start,end) */
	for (long c = 0; c < ncols; c++) {
		if (c >= start && c < end)
			continue;
		for (long j = 0; j < n; j++) {
			long t = trans[j] + j;
			T tmp = AT(A, j, c);
			AT(A, j, c) = AT(A, t, c);
			AT(A, t, c) = tmp;
		}
	}
	return n_trans;
}
/* lu/partial_pivoting/factor.rs:234-295.  perm[i] = source row of row i of
 * P*A; returns transposition_count. */
long FN(oracle_lu_in_place)(T *a, long m, long n, long rs, long cs, long *perm, long *perm_inv,
			    long rec_threshold)
{
	FN(mat) A = {a, m, n, rs, cs};
	long size = m < n ? m : n;
	for (long i = 0; i < m; i++)
		perm[i] = i;
	long *trans = (long *)calloc((size_t)(size > 0 ? size : 1), sizeof(long));
	long nt = FN(lu_rec)(A, 0, size, trans, rec_threshold);
	for (long idx = 0; idx < size; idx++) {
		long t = trans[idx];
		long tmp = perm[idx];
		perm[idx] = perm[idx + t];
		perm[idx + t] = tmp;
	}
	if (m < n)
		FN(trsm_lower)(FN(sub)(A, 0, 0, size, size), 1, FN(sub)(A, 0, size, size, n - size));
	for (long i = 0; i < m; i++)
		perm_inv[perm[i]] = i;
	free(trans);
	return nt;
}

/* ---------------------------------------------------------------- norm_l2 */
/* reductions/norm_l2.rs:6-45 (base: scaled accumulators sml/med/big), :46-62 (pairwise splitting at
 * next_power_of_two((n + 1) / 2) above LINEAR_IMPL_THRESHOLD = 128, reductions/mod.rs:1), :173-184 (selection).
 * The pairwise tree is what keeps a 1e7-long column within 1e-14 (test_norm_l2, norm_l2.rs:216-218); the
 * lane order inside a <= 128 element leaf is the SIMD width's and is not pinned by the reference. */
/* ------------------------------------------------------------------------------------------------
 * LU with full pivoting -- faer/src/linalg/lu/full_pivoting/factor.rs
 *   :255-273  best_in_matrix_fallback: column-major scan, strict '>' on |a| => first maximum in (col, row) order
 *   :316-430  lu_in_place_unblocked: per step swap row / col to the pivot, scale the pivot column by the reciprocal
 *             (the pivot ROW when the driver works on the transposed view), rank-1 update fused with the search of
 *             the next pivot; a best score below the smallest positive normal ends the elimination (identity
 *             transpositions from there on)
 *   :452-525  lu_in_place: works on the view whose row stride is the smaller one, builds the permutations from
 *             the transpositions
 * ------------------------------------------------------------------------------------------------ */
static void FN(fplu_best)(FN(mat) M, long *row, long *col, T *score)
{
	T max = 0;
	*row = 0;
	*col = 0;
	for (long j = 0; j < M.ncols; j++)
		for (long i = 0; i < M.nrows; i++) {
			T v = AT(M, i, j);
			v = v < 0 ? -v : v;
			if (v > max) {
				*row = i;
				*col = j;
				max = v;
			}
		}
	*score = max;
}

static long FN(fplu_unblocked)(FN(mat) A, long *row_trans, long *col_trans, int transpose)
{
	long m = A.nrows, n = A.ncols, n_trans = 0;
	if (m == 0 || n == 0)
		return 0;
	long size = m < n ? m : n;
	long max_row, max_col;
	T max_score;
	FN(fplu_best)(A, &max_row, &max_col, &max_score);
	for (long k = 0; k < size; k++) {
		if (max_score < TMIN) {
			for (long i = k; i < size; i++) {
				row_trans[i] = i;
				col_trans[i] = i;
			}
			break;
		}
		row_trans[k] = max_row;
		col_trans[k] = max_col;
		if (max_row != k) {
			for (long j = 0; j < n; j++) {
				T t = AT(A, k, j);
				AT(A, k, j) = AT(A, max_row, j);
				AT(A, max_row, j) = t;
			}
			n_trans++;
		}
		if (max_col != k) {
			for (long i = 0; i < m; i++) {
				T t = AT(A, i, k);
				AT(A, i, k) = AT(A, i, max_col);
				AT(A, i, max_col) = t;
			}
			n_trans++;
		}
		T inv = (T)1 / AT(A, k, k);
		if (transpose)
			for (long j = k + 1; j < n; j++)
				AT(A, k, j) *= inv;
		else
			for (long i = k + 1; i < m; i++)
				AT(A, i, k) *= inv;
		if (k + 1 == size)
			break;
		/* A11 -= A10[:, k] A01[k, :], then the next pivot */
		for (long j = k + 1; j < n; j++) {
			T r = AT(A, k, j);
			for (long i = k + 1; i < m; i++)
				AT(A, i, j) = FMA(-AT(A, i, k), r, AT(A, i, j));
		}
		FN(fplu_best)(FN(sub)(A, k + 1, k + 1, m - k - 1, n - k - 1), &max_row, &max_col, &max_score);
		max_row += k + 1;
		max_col += k + 1;
	}
	return n_trans;
}

long FN(oracle_full_piv_lu_in_place)(T *a, long m, long n, long rs, long cs, long *row_perm, long *row_perm_inv, long *col_perm,
				     long *col_perm_inv)
{
	FN(mat) A = {a, m, n, rs, cs};
	long size = m < n ? m : n;
	long *rt = (long *)calloc((size_t)(size > 0 ? size : 1), sizeof(long));
	long *ct = (long *)calloc((size_t)(size > 0 ? size : 1), sizeof(long));
	long ars = rs < 0 ? -rs : rs, acs = cs < 0 ? -cs : cs;
	long nt = ars < acs ? FN(fplu_unblocked)(A, rt, ct, 0) : FN(fplu_unblocked)(FN(tr)(A), ct, rt, 1);
	for (long i = 0; i < m; i++)
		row_perm[i] = i;
	for (long i = 0; i < size; i++) {
		long t = row_perm[i];
		row_perm[i] = row_perm[rt[i]];
		row_perm[rt[i]] = t;
	}
	for (long i = 0; i < m; i++)
		row_perm_inv[row_perm[i]] = i;
	for (long j = 0; j < n; j++)
		col_perm[j] = j;
	for (long i = 0; i < size; i++) {
		long t = col_perm[i];
		col_perm[i] = col_perm[ct[i]];
		col_perm[ct[i]] = t;
	}
	for (long j = 0; j < n; j++)
		col_perm_inv[col_perm[j]] = j;
	free(rt);
	free(ct);
	return nt;
}

static void FN(norm_l2_x3_rec)(const T *x, long n, long stride, T sml, T big, T acc[3])
{
	if (n <= 128) {
		T a_sml = 0, a_med = 0, a_big = 0;
		for (long i = 0; i < n; i++) {
			T v = x[i * stride];
			a_sml = FMA(v * sml, v * sml, a_sml);
			a_med = FMA(v, v, a_med);
			a_big = FMA(v * big, v * big, a_big);
		}
		acc[0] = a_sml;
		acc[1] = a_med;
		acc[2] = a_big;
		return;
	}
	long half = (n + 1) / 2, split = 1;
	while (split < half)
		split <<= 1;
	T a0[3], a1[3];
	FN(norm_l2_x3_rec)(x, split, stride, sml, big, a0);
	FN(norm_l2_x3_rec)(x + split * stride, n - split, stride, sml, big, a1);
	acc[0] = a0[0] + a1[0];
	acc[1] = a0[1] + a1[1];
	acc[2] = a0[2] + a1[2];
}

T FN(oracle_norm_l2)(const T *x, long n, long stride)
{
	T sml = SQRT(TMIN), big = SQRT(TMAX);
	T acc[3] = {0, 0, 0};
	if (n > 0)
		FN(norm_l2_x3_rec)(x, n, stride, sml, big, acc);
	if (acc[0] >= 1)
		return SQRT(acc[0]) * big;
	else if (acc[1] >= 1)
		return SQRT(acc[1]);
	else
		return SQRT(acc[2]) * sml;
}

/* ------------------------------------------------------------ Householder */
/* householder.rs:59-107.  tail is read from `in` (stride is) and v is written
 * to `out` (stride os); in==out for the in-place variant. */
typedef struct {
	T tau, norm;
	T hinv; /* head_with_beta_inv (householder.rs:46-47); +inf when the tail is negligible (:73-77) */
} FN(hinfo);
static FN(hinfo) FN(make_householder)(T *head, T *out, long os, const T *in, long is, long len)
{
	FN(hinfo) r;
	T tail_norm = FN(oracle_norm_l2)(in, len, is);
	T head_norm = FABS(*head);
	if (head_norm < TMIN) {
		*head = 0;
		head_norm = 0;
	}
	if (tail_norm < TMIN) {
		r.tau = (T)INFINITY;
		r.hinv = (T)INFINITY;
		r.norm = head_norm;
		return r;
	}
	T norm = HYPOT(head_norm, tail_norm);
	T sign = head_norm != 0 ? (*head) * ((T)1 / head_norm) : (T)1;
	T signed_norm = sign * norm;
	T head_with_beta = *head + signed_norm;
	T hinv = (T)1 / head_with_beta;
	for (long i = 0; i < len; i++)
		out[i * os] = in[i * is] * hinv;
	*head = -signed_norm;
	T t = tail_norm * FABS(hinv);
	r.tau = (T)0.5 * ((T)1 + t * t);
	r.norm = norm;
	r.hinv = hinv;
	return r;
}

/* 4-accumulator dot (matmul/mod.rs:692-726 semantics up to rounding) */
static T FN(dot)(const T *a, long as, const T *b, long bs, long n)
{
	T acc = 0;
	for (long i = 0; i < n; i++)
		acc = FMA(a[i * as], b[i * bs], acc);
	return acc;
}

/* qr/no_pivoting/factor.rs:11-86 */
static long FN(qr_unblocked)(FN(mat) A, T *H, long hs, long hsize, long row_start, long col_start)
{
	long m = A.nrows, n = A.ncols;
	long size = hsize;
	long col = col_start, row = row_start;
	long lim = size < m ? size : m;
	while (row < lim && col < n) {
		T norm_above = FN(oracle_norm_l2)(&AT(A, 0, col), row, A.rs);
		T *head = &AT(A, row, col);
		long tail_len = m - row - 1;
		T *tail_in = &AT(A, row + 1, col);
		FN(hinfo) info;
		const T *v;
		if (row == col) {
			info = FN(make_householder)(head, tail_in, A.rs, tail_in, A.rs, tail_len);
			v = tail_in;
		} else {
			T *out = &AT(A, row + 1, row);
			info = FN(make_householder)(head, out, A.rs, tail_in, A.rs, tail_len);
			long z = tail_len < col - row ? tail_len : col - row;
			for (long i = 0; i < z; i++)
				tail_in[i * A.rs] = 0;
			v = out;
		}
		/* NOTE: when row != col the head lives at (row,col) and stays there */
		T norm = HYPOT(info.norm, norm_above);
		T threshold = TEPS * (T)((double)(m - row) * 16.0) * norm;
		T tau_inv = (T)1 / info.tau;
		H[row * hs] = info.tau;
		if (tau_inv < TMIN) {
			if (info.norm > 0)
				row += 1;
		} else if (info.norm > threshold) {
			for (long c = col + 1; c < n; c++) {
				T *hd = &AT(A, row, c);
				T *tl = &AT(A, row + 1, c);
				T dot = *hd + FN(dot)(v, A.rs, tl, A.rs, tail_len);
				T k = -(dot * tau_inv);
				*hd += k;
				for (long i = 0; i < tail_len; i++)
					tl[i * A.rs] += k * v[i * A.rs];
			}
			row += 1;
		}
		col += 1;
	}
	return row;
}

/* householder.rs:132-272.  Fills the off-diagonal blocks of the upper
 * triangular factor T given the diagonal blocks of size prev_block_size:
 * T_ij = v_i^H v_j for i<j (V unit lower trapezoidal).  The recursion of the
 * reference only decides WHICH products are formed together; the values are
 * the strict upper triangle of V^H V outside the prev-size diagonal blocks
 * (or, when prev_block_size < 8, the whole strict upper triangle). */
static void FN(upgrade_householder_factor)(FN(mat) Tf, FN(mat) V, long block_size, long prev_block_size)
{
	long n = V.ncols, m = V.nrows;
	if (block_size == prev_block_size || Tf.nrows <= prev_block_size)
		return;
	long block_count = (Tf.nrows + block_size - 1) / block_size;
	if (block_count > 1) { /* :153-185 */
		long mid = block_count / 2; /* NB: reference splits at `mid` (a block count) */
		FN(upgrade_householder_factor)(FN(sub)(Tf, 0, 0, mid, mid), FN(sub)(V, 0, 0, m, mid), block_size,
					       prev_block_size);
		FN(upgrade_householder_factor)(FN(sub)(Tf, mid, mid, Tf.nrows - mid, Tf.ncols - mid),
					       FN(sub)(V, mid, mid, m - mid, n - mid), block_size,
					       prev_block_size);
		return;
	}
	for (long i = 0; i < n; i++)
		for (long j = i + 1; j < n; j++) {
			if (prev_block_size >= 8 && (i / prev_block_size) == (j / prev_block_size))
				continue; /* diagonal block already built by the child */
			/* v_i^H v_j : v_i has implicit 1 at row i, v_j implicit 1 at row j */
			T acc = AT(V, j, i); /* v_i[j] * 1 */
			for (long r = j + 1; r < m; r++)
				acc = FMA(AT(V, r, i), AT(V, r, j), acc);
			AT(Tf, i, j) = acc;
		}
}

/* householder.rs:370-620 (generic path).  forward != 0  => apply
 * (I - V T^-H V^H) (the "transpose" apply), else (I - V T^-1 V^H). */
static void FN(apply_block_householder)(FN(mat) V, FN(mat) Tf, FN(mat) M, int forward)
{
	long m = V.nrows, n = V.ncols, k = M.ncols;
	if (n == 0 || k == 0)
		return;
	T *tmpb = (T *)calloc((size_t)(n * k), sizeof(T));
	FN(mat) tmp = {tmpb, n, k, 1, n};
	/* tmp = V^H M with V unit lower trapezoidal */
	for (long c = 0; c < k; c++)
		for (long i = 0; i < n; i++) {
			T acc = AT(M, i, c);
			for (long r = i + 1; r < m; r++)
				acc = FMA(AT(V, r, i), AT(M, r, c), acc);
			AT(tmp, i, c) = acc;
		}
	if (forward)
		FN(trsm_lower)(FN(tr)(Tf), 0, tmp); /* T^-H tmp */
	else
		FN(trsm_upper)(Tf, 0, tmp); /* T^-1 tmp */
	/* M -= V tmp */
	for (long c = 0; c < k; c++)
		for (long r = 0; r < m; r++) {
			T acc = 0;
			long lim = r < n ? r : n; /* strictly-lower part of V */
			for (long i = 0; i < lim; i++)
				acc = FMA(AT(V, r, i), AT(tmp, i, c), acc);
			if (r < n)
				acc = acc + AT(tmp, r, c);
			AT(M, r, c) = AT(M, r, c) - acc;
		}
	free(tmpb);
}

/* qr/no_pivoting/factor.rs:137-256 */
static long FN(qr_blocked)(FN(mat) A, FN(mat) H, long row_start, long col_start, long blocking_threshold)
{
	long m = A.nrows, n = A.ncols;
	long size = m < n ? m : n;
	long block_size = H.nrows;
	if (block_size == 1)
		return FN(qr_unblocked)(A, H.p, H.cs, H.ncols, row_start, col_start);
	long sub_block_size0 = (m * n < blocking_threshold) ? 1 : block_size / 2;
	long col = col_start, row = row_start;
	while (row < size && col < n) {
		long bs = block_size;
		if (size - row < bs)
			bs = size - row;
		if (n - col < bs)
			bs = n - col;
		long sbs = bs < sub_block_size0 ? bs : sub_block_size0;
		long start = row;
		long offset = 0;
		while (offset < bs && col < n) {
			long bs2 = (n - col) < (bs - offset) ? (n - col) : (bs - offset);
			long sbs2 = bs2 < sbs ? bs2 : sbs;
			long new_row = FN(qr_blocked)(FN(sub)(A, 0, 0, m, col + bs2),
						      FN(sub)(H, offset, 0, sbs2, H.ncols), row, col,
						      blocking_threshold);
			long local = new_row - row;
			if (local > 0) {
				long k = 0;
				while (k < local) {
					long s = sbs2 < local - k ? sbs2 : local - k;
					if (k > 0) {
						/* :184-198 move the child's T block onto the diagonal */
						for (long jj = 0; jj < s; jj++)
							for (long ii = 0; ii <= jj; ii++)
								AT(H, offset + k + ii, row + k + jj) =
									AT(H, offset + ii, row + k + jj);
					}
					k += sbs2;
				}
				FN(upgrade_householder_factor)(FN(sub)(H, offset, row, local, local),
							       FN(sub)(A, row, row, m - row, local), local, sbs2);
				if (offset > 0) {
					/* :207-239 rebuild the whole strict upper triangle of the
					 * (offset+local) block as V^H V */
					long w = offset + local;
					FN(mat) Hb = FN(sub)(H, 0, start, w, w);
					FN(mat) Vb = FN(sub)(A, start, start, m - start, w);
					for (long i = 0; i < w; i++)
						for (long j = i + 1; j < w; j++) {
							T acc = AT(Vb, j, i);
							for (long r = j + 1; r < m - start; r++)
								acc = FMA(AT(Vb, r, i), AT(Vb, r, j), acc);
							AT(Hb, i, j) = acc;
						}
				}
			}
			long nright = n - (col + bs2);
			if (nright > 0 && local > 0)
				FN(apply_block_householder)(FN(sub)(A, row, row, m - row, local),
							    FN(sub)(H, offset, row, local, local),
							    FN(sub)(A, row, col + bs2, m - row, nright), 1);
			offset += local;
			row += local;
			col += bs2;
		}
	}
	return row;
}

/* qr/no_pivoting/factor.rs:91-116 */
long FN(oracle_qr_recommended_block_size)(long nrows, long ncols)
{
	long prod = nrows * ncols;
	long size = nrows < ncols ? nrows : ncols;
	long r;
	if (prod > 8192L * 8192)
		r = 256;
	else if (prod > 2048L * 2048)
		r = 128;
	else if (prod > 1024L * 1024)
		r = 64;
	else if (prod > 512L * 512)
		r = 48;
	else if (prod > 128L * 128)
		r = 32;
	else if (prod > 32L * 32)
		r = 8;
	else if (prod > 16L * 16)
		r = 4;
	else
		r = 1;
	if (r > size)
		r = size;
	if (r < 1)
		r = 1;
	return r;
}

/* qr/no_pivoting/factor.rs:258-301.  H is block_size x min(m,n). Returns rank. */
long FN(oracle_qr_in_place)(T *a, long m, long n, long rs, long cs, T *h, long block_size, long hrs, long hcs,
			    long blocking_threshold)
{
	FN(mat) A = {a, m, n, rs, cs};
	long size = m < n ? m : n;
	FN(mat) H = {h, block_size, size, hrs, hcs};
	long rank = FN(qr_blocked)(A, H, 0, 0, blocking_threshold);
	for (long j = rank; j < size; j++)
		for (long i = 0; i < block_size; i++)
			AT(H, i, j) = 0;
	long col = rank / block_size * block_size;
	while (col < size) {
		long bs = block_size < size - col ? block_size : size - col;
		long start = rank > col ? rank : col;
		for (long d = start; d < col + bs; d++)
			AT(H, d - col, d) = (T)INFINITY;
		col += bs;
	}
	return rank;
}

/* ------------------------------------------------------------------------------------------------
 * QR with column pivoting -- faer/src/linalg/qr/col_pivoting/factor.rs
 *   :107-330  qr_in_place_unblocked: columns scaled by 1 / (largest column norm); per step the remaining column
 *             of largest (down-dated) norm is swapped in, its reflector is made, and the rank-1 update of the
 *             trailing matrix is DELAYED by one step: it is applied while the next step computes its dot products
 *             (update_mat_and_dot_simd, :7-105) -- the path the reference takes for f32 / f64 column-major input --
 *             unless the best down-dated norm fell below sqrt(eps) times the best norm at the last recomputation:
 *             then the update is applied at once and all norms are recomputed (:178-203).  Norms are down-dated
 *             as sqrt(norm^2 - a_kj^2) (:96, :300).  The upper triangle is scaled back at the end (:305-309).
 *   :356-395  qr_in_place: T blocks of the reflectors (householder::upgrade_householder_factor with prev size 1)
 * `delayed_ok` == the reference's `T::SIMD_CAPABILITIES.is_simd() && A.row_stride() == 1`.
 * ------------------------------------------------------------------------------------------------ */
static long FN(colpiv_qr_unblocked)(FN(mat) A, T *H, long hs, long *col_perm, int delayed_ok)
{
	long m = A.nrows, n = A.ncols;
	long size = m < n ? m : n;
	long n_trans = 0;
	for (long j = 0; j < n; j++)
		col_perm[j] = j;
	if (size == 0)
		return 0;
	T *dot = (T *)calloc((size_t)(n > 0 ? n : 1), sizeof(T));
	T *norm = (T *)calloc((size_t)(n > 0 ? n : 1), sizeof(T));
	T best = 0;
	const T threshold = SQRT(TEPS);
	for (long j = 0; j < n; j++) {
		T val = FN(oracle_norm_l2)(&AT(A, 0, j), m, A.rs);
		norm[j] = val;
		if (val > best)
			best = val;
	}
	const T scale_fwd = best, scale_bwd = (T)1 / best;
	for (long j = 0; j < n; j++)
		for (long i = 0; i < m; i++)
			AT(A, i, j) *= scale_bwd;
	for (long j = 0; j < n; j++)
		norm[j] = norm[j] * scale_bwd;
	best = best * scale_bwd;
	T best_threshold = best * threshold;
	for (long k = 0; k < size; k++) {
		T new_best = 0;
		long best_col = k;
		for (long j = k; j < n; j++)
			if (norm[j] > new_best) {
				new_best = norm[j];
				best_col = j;
			}
		const int delayed = delayed_ok && k > 0 && new_best >= best_threshold;
		if (k > 0 && !delayed) {
			/* A11 += A10[:, k-1] dot[k:], then fresh norms */
			for (long j = k; j < n; j++) {
				T d = dot[j];
				for (long i = k; i < m; i++)
					AT(A, i, j) = FMA(AT(A, i, k - 1), d, AT(A, i, j));
			}
			best = 0;
			for (long j = k; j < n; j++) {
				T val = FN(oracle_norm_l2)(&AT(A, k, j), m - k, A.rs);
				norm[j] = val;
				if (val > best) {
					best = val;
					best_col = j;
				}
			}
			best_threshold = best * threshold;
		}
		if (best_col != k) {
			n_trans++;
			long tp = col_perm[best_col];
			col_perm[best_col] = col_perm[k];
			col_perm[k] = tp;
			for (long i = 0; i < m; i++) {
				T t = AT(A, i, k);
				AT(A, i, k) = AT(A, i, best_col);
				AT(A, i, best_col) = t;
			}
			T t = dot[k];
			dot[k] = dot[best_col];
			dot[best_col] = t;
			t = norm[k];
			norm[k] = norm[best_col];
			norm[best_col] = t;
		}
		const T l = delayed ? AT(A, k, k - 1) : (T)0;
		const T r = dot[k];
		if (delayed) {
			AT(A, k, k) += l * r;
			for (long i = k + 1; i < m; i++)
				AT(A, i, k) += r * AT(A, i, k - 1);
		}
		FN(hinfo) info = FN(make_householder)(&AT(A, k, k), m > k + 1 ? &AT(A, k + 1, k) : &AT(A, k, k), A.rs,
						      m > k + 1 ? &AT(A, k + 1, k) : &AT(A, k, k), A.rs, m - k - 1);
		const T tau_inv = (T)1 / info.tau;
		H[k * hs] = info.tau;
		if (k + 1 == size) {
			if (delayed)
				for (long j = k + 1; j < n; j++)
					AT(A, k, j) += l * dot[j];
			break;
		}
		for (long j = k + 1; j < n; j++) {
			if (delayed) {
				/* update_mat_and_dot_simd :60-98 (the SIMD lane split of the sum is not restated) */
				const T b0 = dot[j];
				T acc = 0;
				for (long i = k + 1; i < m; i++) {
					T dst = FMA(AT(A, i, k - 1), b0, AT(A, i, j));
					acc = FMA(AT(A, i, k), dst, acc);
					AT(A, i, j) = dst;
				}
				const T tmp = AT(A, k, j) + l * b0;
				const T d0 = (tmp + acc) * (-tau_inv);
				AT(A, k, j) = tmp + d0;
				dot[j] = d0;
			} else {
				T acc = AT(A, k, j);
				for (long i = k + 1; i < m; i++)
					acc = FMA(AT(A, i, k), AT(A, i, j), acc);
				const T d = -(acc * tau_inv);
				AT(A, k, j) += d;
				dot[j] = d;
			}
			norm[j] = SQRT(norm[j] * norm[j] - AT(A, k, j) * AT(A, k, j));
		}
	}
	for (long j = 0; j < n; j++)
		for (long i = 0; i <= j && i < m; i++)
			AT(A, i, j) *= scale_fwd;
	free(dot);
	free(norm);
	return n_trans;
}

long FN(oracle_colpiv_qr_in_place)(T *a, long m, long n, long rs, long cs, T *h, long block_size, long hrs, long hcs, long *col_perm,
				   long *col_perm_inv)
{
	FN(mat) A = {a, m, n, rs, cs};
	long size = m < n ? m : n;
	FN(mat) H = {h, block_size, size, hrs, hcs};
	long nt = FN(colpiv_qr_unblocked)(A, h, hcs, col_perm, rs == 1);
	for (long j = 0; j < n; j++)
		col_perm_inv[col_perm[j]] = j;
	/* factor.rs:372-393 */
	long j = 0;
	while (j < size) {
		long bs = block_size < size - j ? block_size : size - j;
		FN(mat) Hb = FN(sub)(H, 0, j, bs, bs);
		for (long c = 0; c < bs; c++)
			AT(Hb, c, c) = AT(Hb, 0, c);
		FN(upgrade_householder_factor)(Hb, FN(sub)(A, j, j, m - j, bs), bs, 1);
		j += bs;
	}
	return nt;
}

/* householder.rs:724-808 : sequence apply on the left.
 * transpose != 0 => Q^H * M (blocks first to last), else Q * M (last to first). */
void FN(oracle_apply_householder_sequence_left)(const T *v, long m, long n, long vrs, long vcs, const T *h,
						long block_size, long hrs, long hcs, T *mat, long k, long mrs,
						long mcs, int transpose)
{
	FN(mat) V = {(T *)v, m, n, vrs, vcs};
	long size = m < n ? m : n;
	FN(mat) H = {(T *)h, block_size, size, hrs, hcs};
	FN(mat) M = {mat, m, k, mrs, mcs};
	if (transpose) {
		long j = 0;
		while (j < size) {
			long bs = block_size < size - j ? block_size : size - j;
			FN(apply_block_householder)(FN(sub)(V, j, j, m - j, bs), FN(sub)(H, 0, j, bs, bs),
						    FN(sub)(M, j, 0, m - j, k), 1);
			j += bs;
		}
	} else {
		long j = size;
		long bs = size % block_size;
		if (bs == 0)
			bs = block_size;
		while (j > 0) {
			long jp = j - bs;
			bs = block_size;
			FN(apply_block_householder)(FN(sub)(V, jp, jp, m - jp, j - jp),
						    FN(sub)(H, 0, jp, j - jp, j - jp),
						    FN(sub)(M, jp, 0, m - jp, k), 0);
			j = jp;
		}
	}
}

/* ------------------------------------------------ tridiagonalization (evd) */
/* evd/tridiag.rs:174-272 (tridiag_fused_op_fallback; the SIMD variant :36-139 computes the same quantities in a
 * different summation order).  A: r x r, only the lower triangle is read / written.
 *   A(lower) -= u w^H + w u^H ;  z = f * tril(A) x ;  y = f * striu(A^H) x   (the caller adds z to y) */
static void FN(tridiag_fused_op)(FN(mat) A, T *y2, T *z2, const T *w2, const T *u2, const T *x2, T f)
{
	long r = A.ncols;
	FN(mat) U = {(T *)u2, r, 1, 1, r}, W = {(T *)w2, r, 1, 1, r}, X = {(T *)x2, r, 1, 1, r};
	FN(mat) Z = {z2, r, 1, 1, r}, Y = {y2, r, 1, 1, r};
	FN(matmul_triangular)(A, 1, 1, U, 0, FN(tr)(W), 0, (T)-1); /* :189-198 */
	FN(matmul_triangular)(A, 1, 1, W, 0, FN(tr)(U), 0, (T)-1); /* :199-209 */
	FN(matmul_triangular)(Z, 0, 0, A, 1, X, 0, f);            /* :226-236 */
	FN(matmul_triangular)(Y, 0, 0, FN(tr)(A), 4, X, 0, f);    /* :237-247 */
}

/* evd/tridiag.rs:274-535 (Par::Seq), real scalars.  a: n x n self-adjoint (lower triangle used), h: bs x (n - 1).
 * On return the diagonal and subdiagonal of a hold T, the essential parts of the reflectors sit below the
 * subdiagonal, h holds the block Householder factors of A.submatrix(1, 0, n - 1, n - 1). */
long FN(oracle_tridiag_in_place)(T *a, long n, long rs, long cs, T *h, long bs, long hrs, long hcs)
{
	if (n == 0)
		return 0;
	FN(mat) A = {a, n, n, rs, cs};
	FN(mat) H = {h, bs, n - 1, hrs, hcs};
	T *y = (T *)calloc((size_t)n, sizeof(T)), *w = (T *)calloc((size_t)n, sizeof(T)), *z = (T *)calloc((size_t)n, sizeof(T));
	T *xc = (T *)calloc((size_t)n, sizeof(T)), *uc = (T *)calloc((size_t)n, sizeof(T));
	for (long k = 0; k < n; k++) {
		/* :300-318 column k receives the rest of the rank-2 update of step k - 1 */
		if (k > 0) {
			T y1 = y[k];
			AT(A, k, k) -= y1 + y1;
			for (long i = k + 1; i < n; i++)
				AT(A, i, k) -= y1 * AT(A, i, k - 1) + y[i];
		}
		if (k + 1 == n)
			break;
		long k1 = k + 1, r = n - k - 2; /* r: rows below the reflector's head */
		/* :330-336 reflector of column k below the diagonal */
		FN(hinfo) hi = FN(make_householder)(&AT(A, k1, k), &AT(A, k + 2, k), rs, &AT(A, k + 2, k), rs, r);
		T tau_inv = (T)1 / hi.tau;
		AT(H, 0, k) = hi.tau;
		T *y1p = &y[k1], *y2 = y + k + 2, *w2 = w + k + 2, *z2 = z + k + 2;
		for (long i = 0; i < r; i++)
			xc[i] = AT(A, k + 2 + i, k);
		FN(mat) A22 = FN(sub)(A, k + 2, k + 2, r, r);
		if (k > 0) {
			/* :348-359 */
			T u1 = AT(A, k1, k - 1);
			AT(A, k1, k1) -= u1 * (*y1p) + (*y1p) * u1;
			for (long i = 0; i < r; i++) {
				uc[i] = AT(A, k + 2 + i, k - 1);
				AT(A, k + 2 + i, k1) -= uc[i] * (*y1p) + y2[i] * u1;
			}
			for (long i = 0; i < r; i++)
				w2[i] = y2[i];
			FN(tridiag_fused_op)(A22, y2, z2, w2, uc, xc, tau_inv); /* :362-376 */
			for (long i = 0; i < r; i++)
				y2[i] += z2[i]; /* :377-379 */
		} else {
			/* :461-483 */
			FN(mat) X = {xc, r, 1, 1, r}, Y = {y2, r, 1, 1, r};
			FN(matmul_triangular)(Y, 0, 0, A22, 1, X, 0, tau_inv);
			FN(matmul_triangular)(Y, 0, 1, FN(tr)(A22), 4, X, 0, tau_inv);
		}
		/* :484-511 */
		for (long i = 0; i < r; i++)
			y2[i] += AT(A, k + 2 + i, k1) * tau_inv;
		*y1p = (AT(A, k1, k1) + FN(dot)(&AT(A, k + 2, k1), rs, xc, 1, r)) * tau_inv;
		T b = ((*y1p + FN(dot)(xc, 1, y2, 1, r)) * (T)0.5) * tau_inv;
		*y1p -= b;
		for (long i = 0; i < r; i++)
			y2[i] -= b * xc[i];
	}
	/* :516-533 block Householder factors */
	if (n > 1) {
		long m = n - 1;
		FN(mat) V = FN(sub)(A, 1, 0, m, m);
		long j = 0;
		while (j < m) {
			long b = bs < m - j ? bs : m - j;
			FN(mat) Hb = FN(sub)(H, 0, j, b, b);
			for (long q = 0; q < b; q++)
				AT(Hb, q, q) = AT(Hb, 0, q);
			FN(upgrade_householder_factor)(Hb, FN(sub)(V, j, j, m - j, b), b, 1);
			j += b;
		}
	}
	free(y);
	free(w);
	free(z);
	free(xc);
	free(uc);
	return 0;
}

/* ----------------------------------------------- bidiagonalization (svd) */
/* svd/bidiag.rs:47-255 (Par::Seq; bidiag_fused_op_fallback :282-301), real scalars, m >= n (the reference's SVD
 * only calls it on tall matrices, and so do its tests :380-500).  a: m x n -> upper bidiagonal B on the diagonal and
 * superdiagonal with A = U B V^H; the left reflectors below the diagonal (block factors hl: bl x n), the right
 * reflectors right of the superdiagonal (block factors hr: br x (n - 1)). */
long FN(oracle_bidiag_in_place)(T *a, long m, long n, long rs, long cs, T *hl, long bl, long hlrs, long hlcs, T *hr, long br, long hrrs,
				long hrcs)
{
	long size = m < n ? m : n;
	if (size == 0)
		return 0;
	FN(mat) A = {a, m, n, rs, cs};
	FN(mat) Hl = {hl, bl, size, hlrs, hlcs};
	FN(mat) Hr = {hr, br, size - 1, hrrs, hrcs};
	T *y = (T *)calloc((size_t)n + 1, sizeof(T)), *z = (T *)calloc((size_t)m + 1, sizeof(T));
	for (long k = 0; k < size; k++) {
		long k1 = k - 1, rr = m - k - 1, cc = n - k - 1; /* A22 is rr x cc */
		T *y2 = y + k + 1, *z2 = z + k + 1;
		if (k > 0) { /* :80-98 */
			T y1 = y[k], z1 = z[k], up0 = AT(A, k, k1);
			AT(A, k, k) -= up0 * y1 + z1;
			for (long i = 0; i < rr; i++)
				AT(A, k + 1 + i, k) -= AT(A, k + 1 + i, k1) * y1 + z2[i];
			for (long j = 0; j < cc; j++)
				AT(A, k, k + 1 + j) -= up0 * y2[j] + z1 * AT(A, k1, k + 1 + j);
		}
		/* :99-102 left reflector */
		FN(hinfo) hi = FN(make_householder)(&AT(A, k, k), &AT(A, k + 1, k), rs, &AT(A, k + 1, k), rs, rr);
		T tl_inv = (T)1 / hi.tau;
		AT(Hl, 0, k) = hi.tau;
		if (k > 0) { /* bidiag_fused_op_fallback :292-300 */
			for (long j = 0; j < cc; j++)
				for (long i = 0; i < rr; i++) {
					T acc = AT(A, k + 1 + i, k1) * y2[j];
					AT(A, k + 1 + i, k + 1 + j) = FMA((T)-1, acc, AT(A, k + 1 + i, k + 1 + j));
				}
			for (long j = 0; j < cc; j++)
				for (long i = 0; i < rr; i++) {
					T acc = z2[i] * AT(A, k1, k + 1 + j);
					AT(A, k + 1 + i, k + 1 + j) = FMA((T)-1, acc, AT(A, k + 1 + i, k + 1 + j));
				}
		}
		for (long j = 0; j < cc; j++) { /* y2 = u^H A22 (:293-300 / :147-154) */
			T acc = 0;
			for (long i = 0; i < rr; i++)
				acc = FMA(AT(A, k + 1 + i, k), AT(A, k + 1 + i, k + 1 + j), acc);
			y2[j] = acc;
		}
		for (long j = 0; j < cc; j++) { /* :156-159 */
			y2[j] = (y2[j] + AT(A, k, k + 1 + j)) * tl_inv;
			AT(A, k, k + 1 + j) -= y2[j];
		}
		T norm = FN(oracle_norm_l2)(&AT(A, k, k + 1), cc, cs); /* :160-164 */
		T norm_inv = (T)1 / norm;
		if (norm != 0)
			for (long j = 0; j < cc; j++)
				AT(A, k, k + 1 + j) *= norm_inv;
		for (long i = 0; i < rr; i++) { /* z2 = A22 A12^H :165-172 */
			T acc = 0;
			for (long j = 0; j < cc; j++)
				acc = FMA(AT(A, k + 1 + i, k + 1 + j), AT(A, k, k + 1 + j), acc);
			z2[i] = acc;
		}
		if (k + 1 == size)
			break;
		/* :176-213 right reflector of the (normalised) row */
		FN(hinfo) hr_ = FN(make_householder)(&AT(A, k, k + 1), &AT(A, k, k + 2), cs, &AT(A, k, k + 2), cs, cc - 1);
		T tr_inv = (T)1 / hr_.tau;
		AT(Hr, 0, k) = hr_.tau;
		T beta = AT(A, k, k + 1);
		AT(A, k, k + 1) = beta * norm;
		T b = y2[0] + FN(dot)(y2 + 1, 1, &AT(A, k, k + 2), cs, cc - 1);
		if (hr_.hinv != (T)INFINITY) {
			for (long i = 0; i < rr; i++) {
				T w = z2[i] - AT(A, k + 1 + i, k + 1) * beta;
				w = w * hr_.hinv;
				w = w - AT(A, k + 1 + i, k) * b;
				z2[i] = w * tr_inv;
			}
		} else {
			for (long i = 0; i < rr; i++) {
				T w = AT(A, k + 1 + i, k + 1) - AT(A, k + 1 + i, k) * b;
				z2[i] = w * tr_inv;
			}
		}
	}
	/* :216-254 block Householder factors */
	for (long j = 0; j < size;) {
		long b = bl < size - j ? bl : size - j;
		FN(mat) Hb = FN(sub)(Hl, 0, j, b, b);
		for (long q = 0; q < b; q++)
			AT(Hb, q, q) = AT(Hb, 0, q);
		FN(upgrade_householder_factor)(Hb, FN(sub)(A, j, j, m - j, b), b, 1);
		j += b;
	}
	if (size > 1) {
		long sz = size - 1;
		FN(mat) At = FN(tr)(FN(sub)(A, 0, 1, sz, n - 1)); /* (n - 1) x sz */
		for (long j = 0; j < sz;) {
			long b = br < sz - j ? br : sz - j;
			FN(mat) Hb = FN(sub)(Hr, 0, j, b, b);
			for (long q = 0; q < b; q++)
				AT(Hb, q, q) = AT(Hb, 0, q);
			FN(upgrade_householder_factor)(Hb, FN(sub)(At, j, j, n - 1 - j, b), b, 1);
			j += b;
		}
	}
	free(y);
	free(z);
	return 0;
}

/* ------------------------------------------------- Hessenberg reduction (evd) */
/* evd/hessenberg.rs:230-408 (hessenberg_rearranged_unblocked, Par::Seq; fused op :149-193), real scalars.  The reference
 * runs this variant for n * n < 65536 and hessenberg_gqvdg_blocked (:568-736, restated further down) above: the same
 * Householder reflectors of the same columns, applied in another order of operations -- equal up to rounding, and the
 * reference's own tests pin both variants by the same property (:740-900).
 * a: n x n -> upper Hessenberg H (a = Q H Q^H) on and above the subdiagonal, the essential parts of the reflectors below
 * it; h: bs x (n - 1) block Householder factors of A.submatrix(1, 0, n - 1, n - 1). */
long FN(oracle_hessenberg_in_place)(T *a, long n, long rs, long cs, T *h, long bs, long hrs, long hcs)
{
	if (n == 0)
		return 0;
	FN(mat) A = {a, n, n, rs, cs};
	FN(mat) H = {h, bs, n - 1, hrs, hcs};
	T *y = (T *)calloc((size_t)n, sizeof(T)), *z = (T *)calloc((size_t)n, sizeof(T));
	T *v = (T *)calloc((size_t)n, sizeof(T)), *w = (T *)calloc((size_t)n, sizeof(T));
	for (long k = 0; k < n; k++) {
		long r = n - k - 1; /* A22 is r x r */
		T *y2 = y + k + 1, *z2 = z + k + 1, *v2 = v + k + 1, *w2 = w + k + 1;
		if (k > 0) { /* :266-281 */
			T y1 = y[k], z1 = z[k];
			long p = k - 1;
			AT(A, k, k) -= y1 + z1;
			for (long j = 0; j < r; j++)
				AT(A, k, k + 1 + j) -= y2[j] + z1 * AT(A, k + 1 + j, p);
			for (long i = 0; i < r; i++)
				AT(A, k + 1 + i, k) -= AT(A, k + 1 + i, p) * y1 + z2[i];
		}
		if (k + 1 == n)
			break;
		/* :294-305 reflector of column k below the subdiagonal; its head is 1 while the step runs */
		FN(hinfo) hi = FN(make_householder)(&AT(A, k + 1, k), &AT(A, k + 2, k), rs, &AT(A, k + 2, k), rs, r - 1);
		T tau_inv = (T)1 / hi.tau;
		T beta = AT(A, k + 1, k);
		AT(A, k + 1, k) = (T)1;
		AT(H, 0, k) = hi.tau;
		/* x2 = A[k+1.., k] */
		if (k > 0) { /* hessenberg_fused_op_fallback :160-192 */
			long p = k - 1;
			for (long j = 0; j < r; j++)
				for (long i = 0; i < r; i++) {
					T acc = AT(A, k + 1 + i, p) * y2[j];
					AT(A, k + 1 + i, k + 1 + j) = FMA((T)-1, acc, AT(A, k + 1 + i, k + 1 + j));
				}
			for (long j = 0; j < r; j++)
				for (long i = 0; i < r; i++) {
					T acc = z2[i] * AT(A, k + 1 + j, p);
					AT(A, k + 1 + i, k + 1 + j) = FMA((T)-1, acc, AT(A, k + 1 + i, k + 1 + j));
				}
		}
		for (long i = 0; i < r; i++) { /* w2 = A22 x2 */
			T acc = 0;
			for (long j = 0; j < r; j++)
				acc = FMA(AT(A, k + 1 + i, k + 1 + j), AT(A, k + 1 + j, k), acc);
			w2[i] = acc;
		}
		for (long j = 0; j < r; j++) { /* v2 = x2^H A22 */
			T acc = 0;
			for (long i = 0; i < r; i++)
				acc = FMA(AT(A, k + 1 + i, k), AT(A, k + 1 + i, k + 1 + j), acc);
			v2[j] = acc;
		}
		for (long i = 0; i < r; i++) { /* :322-323 / :325-340 */
			y2[i] = v2[i];
			z2[i] = w2[i];
		}
		/* :342-357 */
		T b = (FN(dot)(&AT(A, k + 1, k), rs, z2, 1, r) * (T)0.5) * tau_inv;
		for (long i = 0; i < r; i++) {
			T u = AT(A, k + 1 + i, k);
			y2[i] = (y2[i] - b * u) * tau_inv;
			z2[i] = (z2[i] - b * u) * tau_inv;
		}
		/* :358-362 row k right of the diagonal, :363-378 the rows above it: reflector applied from the right */
		{
			T d = FN(dot)(&AT(A, k, k + 1), cs, &AT(A, k + 1, k), rs, r) * tau_inv;
			for (long j = 0; j < r; j++)
				AT(A, k, k + 1 + j) -= d * AT(A, k + 1 + j, k);
		}
		for (long i = 0; i < k; i++) {
			T acc = 0;
			for (long j = 0; j < r; j++)
				acc = FMA(AT(A, i, k + 1 + j), AT(A, k + 1 + j, k), acc);
			w[i] = acc;
		}
		for (long j = 0; j < r; j++)
			for (long i = 0; i < k; i++) {
				T acc = w[i] * AT(A, k + 1 + j, k);
				AT(A, i, k + 1 + j) = FMA(-tau_inv, acc, AT(A, i, k + 1 + j));
			}
		AT(A, k + 1, k) = beta; /* :379 */
	}
	if (n > 1) { /* :382-406 */
		long m = n - 1;
		FN(mat) V = FN(sub)(A, 1, 0, m, m);
		long j = 0;
		while (j < m) {
			long b = bs < m - j ? bs : m - j;
			FN(mat) Hb = FN(sub)(H, 0, j, b, b);
			for (long q = 0; q < b; q++)
				AT(Hb, q, q) = AT(Hb, 0, q);
			FN(upgrade_householder_factor)(Hb, FN(sub)(V, j, j, m - j, b), b, 1);
			j += b;
		}
	}
	free(y);
	free(z);
	free(v);
	free(w);
	return 0;
}

/* --------------------------------------------------------- exported shims */
/* evd/hessenberg.rs:409-548 (hessenberg_gqvdg_unblocked): one block of b reflectors of the n x n matrix A (the caller's
 * trailing submatrix), left-looking -- column k is brought up to date with the k reflectors before it from both sides, then
 * its reflector is made; Z[:, k] = A[:, k+1:] u_k and the column k of the block's T factor come out of the same step.
 * While the block runs every reflector's head is 1 in memory; the subdiagonal entries are kept in beta. */
static void FN(hess_gqvdg_unblocked)(FN(mat) A, FN(mat) Z, FN(mat) H, T *beta)
{
	long n = A.nrows, b = H.nrows;
	T *xbuf = (T *)calloc((size_t)(n > 0 ? n : 1), sizeof(T));
	for (long k = 0; k < b; k++) {
		FN(mat) x0 = {xbuf, k, 1, 1, k > 0 ? k : 1};
		FN(mat) T00 = FN(sub)(H, 0, 0, k, k);
		FN(mat) U0 = FN(sub)(A, 0, 0, n, k);
		FN(mat) A1 = FN(sub)(A, 0, k, n, 1), A2 = FN(sub)(A, 0, k + 1, n, n - k - 1);
		FN(mat) Z0 = FN(sub)(Z, 0, 0, n, k), Z1 = FN(sub)(Z, 0, k, n, 1);
		FN(mat) U00 = FN(sub)(U0, 0, 0, k, k), U10 = FN(sub)(U0, k, 0, 1, k), U20 = FN(sub)(U0, k + 1, 0, n - k - 1, k);
		for (long i = 0; i < k; i++) /* :438 x0 = U10^H */
			xbuf[i] = AT(U10, 0, i);
		FN(trsm_upper)(T00, 0, x0);		 /* :439-443 */
		FN(gemm)(A1, 1, Z0, x0, (T)-1);		 /* :444-451 A1 -= Z0 x0 */
		FN(mat) A01 = FN(sub)(A1, 0, 0, k, 1), A21 = FN(sub)(A1, k + 1, 0, n - k - 1, 1);
		T *A11 = &AT(A1, k, 0);
		/* :455-476 x0 = U00^H (strict upper) A01 + A11 conj(U10^T) + U20^H A21 */
		FN(matmul_triangular)(x0, 0, 0, FN(tr)(U00), 4, A01, 0, (T)1);
		for (long i = 0; i < k; i++)
			xbuf[i] += (*A11) * AT(U10, 0, i);
		FN(gemm)(x0, 1, FN(tr)(U20), A21, (T)1);
		FN(trsm_lower)(FN(tr)(T00), 0, x0);	 /* :478-484 x0 <- T00^-H x0 */
		/* :485-505 */
		FN(matmul_triangular)(A01, 0, 1, U00, 3, x0, 0, (T)-1);
		*A11 -= FN(dot)(&AT(U10, 0, 0), U10.cs, xbuf, 1, k);
		FN(gemm)(A21, 1, U20, x0, (T)-1);
		T *t11 = &AT(H, k, k);
		if (k + 1 < n) { /* :506-513 */
			T *head = &AT(A21, 0, 0);
			FN(hinfo) hi = FN(make_householder)(head, &AT(A21, 1, 0), A21.rs, &AT(A21, 1, 0), A21.rs, n - k - 2);
			beta[k] = *head;
			*head = (T)1;
			*t11 = hi.tau;
		} else {
			*t11 = (T)INFINITY;
		}
		FN(gemm)(Z1, 0, A2, A21, (T)1);				      /* :517-524 Z1 = A2 A21 */
		FN(gemm)(FN(sub)(H, 0, k, k, 1), 0, FN(tr)(U20), A21, (T)1); /* :525-532 T01 = U20^H A21 */
	}
	free(xbuf);
}

/* evd/hessenberg.rs:568-736 (hessenberg_gqvdg_blocked): the variant the reference runs for n * n >= blocking_threshold
 * (256 * 256 by default, :21).  Per block of b = h.nrows columns: the block's reflectors and T factor (above), then the
 * block reflector applied from the right to the rows above the block (X0) and -- through Z = A U -- to the rows of the
 * block, and from the left to the columns right of the block.  a, h as in oracle_hessenberg_in_place. */
long FN(oracle_hessenberg_blocked_in_place)(T *a, long n, long rs, long cs, T *h, long b, long hrs, long hcs)
{
	if (n == 0)
		return 0;
	FN(mat) A = {a, n, n, rs, cs};
	FN(mat) H = {h, b, n - 1, hrs, hcs};
	T *zbuf = (T *)calloc((size_t)n * (size_t)b, sizeof(T)), *xb = (T *)calloc((size_t)n * (size_t)b, sizeof(T));
	T *beta = (T *)calloc((size_t)b, sizeof(T));
	FN(mat) Z = {zbuf, n, b, 1, n};
	long j = 0;
	while (j < n) {
		long bs = b < n - j ? b : n - j;
		long bu = bs < n - j - 1 ? bs : n - j - 1; /* bs_u */
		FN(mat) T1 = FN(sub)(H, 0, j, bu, bu);
		FN(hess_gqvdg_unblocked)(FN(sub)(A, j, j, n - j, n - j), FN(sub)(Z, j, 0, n - j, bs), T1, beta);
		{
			long r2 = n - j - bu; /* rows / columns after the block */
			FN(mat) X = {xb, n, bu, 1, n};
			FN(mat) X0 = FN(sub)(X, 0, 0, j, bu), X2 = FN(sub)(X, j + bu, 0, r2, bu);
			FN(mat) Z1 = FN(sub)(Z, j, 0, bu, bu), Z2 = FN(sub)(Z, j + bu, 0, r2, bu);
			FN(mat) A01 = FN(sub)(A, 0, j, j, bu), A02 = FN(sub)(A, 0, j + bu, j, r2);
			FN(mat) U1 = FN(sub)(A, j, j, bu, bu), A12 = FN(sub)(A, j, j + bu, bu, r2);
			FN(mat) U2 = FN(sub)(A, j + bu, j, r2, bu), A22 = FN(sub)(A, j + bu, j + bu, r2, r2);
			/* :610-649 rows above the block: X0 = [A01 A02] [U1; U2] T1^-1, [A01 A02] -= X0 [U1; U2]^H */
			FN(matmul_triangular)(X0, 0, 0, A01, 0, U1, 3, (T)1);
			FN(gemm)(X0, 1, A02, U2, (T)1);
			FN(trsm_lower)(FN(tr)(T1), 0, FN(tr)(X0));
			FN(matmul_triangular)(A01, 0, 1, X0, 0, FN(tr)(U1), 4, (T)-1);
			FN(gemm)(A02, 1, X0, FN(tr)(U2), (T)-1);
			/* :650-675 rows of the block and below: Z <- Z T1^-1, [A12; A22] -= Z U2^H */
			FN(trsm_lower)(FN(tr)(T1), 0, FN(tr)(Z1));
			FN(trsm_lower)(FN(tr)(T1), 0, FN(tr)(Z2));
			FN(gemm)(A12, 1, Z1, FN(tr)(U2), (T)-1);
			FN(gemm)(A22, 1, Z2, FN(tr)(U2), (T)-1);
			/* :676-723 from the left: X = T1^-H [U1; U2]^H [A12; A22], [A12; A22] -= [U1; U2] X */
			FN(mat) Xt = FN(tr)(X2);
			FN(matmul_triangular)(Xt, 0, 0, FN(tr)(U1), 4, A12, 0, (T)1);
			FN(gemm)(Xt, 1, FN(tr)(U2), A22, (T)1);
			FN(trsm_lower)(FN(tr)(T1), 0, Xt);
			FN(matmul_triangular)(A12, 0, 1, U1, 3, Xt, 0, (T)-1);
			FN(gemm)(A22, 1, U2, Xt, (T)-1);
		}
		for (long k = 0; k < bs; k++) /* :725-733 the subdiagonal back in place of the reflector heads */
			if (k + 1 < n - j)
				AT(A, j + k + 1, j + k) = beta[k];
		j += bs;
	}
	free(zbuf);
	free(xb);
	free(beta);
	return 0;
}

void FN(oracle_matmul)(T *c, long m, long n, long crs, long ccs, int accum_add, const T *a, long k, long ars,
		       long acs, const T *b, long brs, long bcs, T alpha)
{
	FN(mat) C = {c, m, n, crs, ccs};
	FN(mat) A = {(T *)a, m, k, ars, acs};
	FN(mat) B = {(T *)b, k, n, brs, bcs};
	if (k == 0) { /* matmul/mod.rs:1194-1196 */
		if (!accum_add)
			for (long j = 0; j < n; j++)
				for (long i = 0; i < m; i++)
					AT(C, i, j) = 0;
		return;
	}
	FN(gemm)(C, accum_add, A, B, alpha);
}
void FN(oracle_matmul_triangular)(T *c, long m, long n, long crs, long ccs, int c_s, int accum_add, const T *a,
				  long k, long ars, long acs, int a_s, const T *b, long brs, long bcs, int b_s,
				  T alpha)
{
	FN(mat) C = {c, m, n, crs, ccs};
	FN(mat) A = {(T *)a, m, k, ars, acs};
	FN(mat) B = {(T *)b, k, n, brs, bcs};
	FN(matmul_triangular)(C, c_s, accum_add, A, a_s, B, b_s, alpha);
}
/* side: 0 lower, 1 upper */
void FN(oracle_trsm)(const T *t, long n, long trs, long tcs, int upper, int unit, T *x, long k, long xrs,
		     long xcs)
{
	FN(mat) Tm = {(T *)t, n, n, trs, tcs};
	FN(mat) X = {x, n, k, xrs, xcs};
	if (upper)
		FN(trsm_upper)(Tm, unit, X);
	else
		FN(trsm_lower)(Tm, unit, X);
}
