// In-place triangular solve  X <- op(T)^-1 X  (left side) for gfx950.
//
// Replaces faer/src/linalg/triangular_solve.rs:16-604.  Same shape of
// algorithm as the reference -- recurse on the triangle, off-diagonal block by GEMM
// (triangular_solve.rs:420-604) -- but the leaf is GPU sized: instead of the reference's n <= 4 closed
// forms, one wavefront solves a 64 x 64 triangle against 64 right-hand sides:
//   * the triangle and its reciprocal diagonal sit in LDS and are read as wave-uniform broadcasts;
//   * every lane owns ONE right-hand side, held in 64 registers, and runs the column-oriented (axpy)
//     substitution: 2016 independent FMAs, no cross-lane traffic at all;
//   * X is staged through a padded LDS tile so that global accesses run along whichever of its two
//     strides is the small one (rows of a transposed Cholesky panel or columns of an LU block row);
//   * like the reference's base case (triangular_solve.rs:98-198) the diagonal enters as a reciprocal
//     computed once: x_j <- x_j * (1 / t_jj).
// Upper triangles are lower triangles on the row/column reversed views (triangular_solve.rs:578-604);
// every kernel here takes signed strides so the reversal is free.
#include "common.h"
#include "lds_blocks.h"

namespace fh {

template <typename T>
__global__ __launch_bounds__(64) void trsm_leaf_kernel(const T *__restrict__ Lp, idx_t lrs, idx_t lcs, int n, int unit,
							T *Xp, idx_t xrs, idx_t xcs, int nrhs, int lanes_along_rhs)
{
	constexpr int NB = 64, XP = NB + 1;
	__shared__ T Ls[NB * NB]; // column major: Ls[j * NB + i] = L[i][j]
	__shared__ T dinv[NB];
	__shared__ T Xs[NB * XP]; // Xs[i * XP + c]
	const int lane = threadIdx.x;
	const int c0 = blockIdx.x * NB;
	const int nc = min(NB, nrhs - c0);

	// ---- stage the triangle (identity padded); 8 independent loads in flight per lane ----
	for (int e0 = lane; e0 < NB * NB; e0 += 64 * 8) {
		T v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int e = e0 + u * 64;
			const int i = e % NB, j = e / NB;
			const bool in = i < n && j < i;
			const T x = Lp[in ? (idx_t) i * lrs + (idx_t) j * lcs : (idx_t) 0];
			v[u] = in ? x : (T) 0;
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int e = e0 + u * 64;
			Ls[(e / NB) * NB + e % NB] = v[u];
		}
	}
	{
		T d = (T) 1;
		if (lane < n && !unit)
			d = (T) 1 / Lp[(idx_t) lane * lrs + (idx_t) lane * lcs];
		dinv[lane] = d;
	}
	// ---- stage X: lanes run along the dimension with the smaller stride; batches of 8 loads ----
	if (lanes_along_rhs) {
		for (int i0 = 0; i0 < n; i0 += 8) {
			T v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const bool in = i0 + u < n && lane < nc;
				const T x = Xp[in ? (idx_t) (i0 + u) * xrs + (idx_t) (c0 + lane) * xcs : (idx_t) 0];
				v[u] = in ? x : (T) 0;
			}
#pragma unroll
			for (int u = 0; u < 8; ++u)
				if (i0 + u < n)
					Xs[(i0 + u) * XP + lane] = v[u];
		}
	} else {
		for (int cc = 0; cc < nc; cc += 8) {
			T v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const bool in = cc + u < nc && lane < n;
				const T x = Xp[in ? (idx_t) lane * xrs + (idx_t) (c0 + cc + u) * xcs : (idx_t) 0];
				v[u] = in ? x : (T) 0;
			}
#pragma unroll
			for (int u = 0; u < 8; ++u)
				if (cc + u < nc && lane < n)
					Xs[lane * XP + cc + u] = v[u];
		}
	}
	__syncthreads();

	T x[NB];
#pragma unroll
	for (int i = 0; i < NB; ++i)
		x[i] = (i < n && lane < nc) ? Xs[i * XP + lane] : (T) 0;

#pragma unroll
	for (int j = 0; j < NB; ++j) {
		if (j < n) { // wave-uniform
			const T xj = x[j] * dinv[j];
			x[j] = xj;
#pragma unroll
			for (int i = j + 1; i < NB; ++i)
				x[i] = __builtin_fma(-Ls[j * NB + i], xj, x[i]);
		}
	}

#pragma unroll
	for (int i = 0; i < NB; ++i)
		if (i < n)
			Xs[i * XP + lane] = x[i];
	__syncthreads();
	if (lanes_along_rhs) {
		for (int i = 0; i < n; ++i)
			if (lane < nc)
				Xp[(idx_t) i * xrs + (idx_t) (c0 + lane) * xcs] = Xs[i * XP + lane];
	} else {
		for (int c = 0; c < nc; ++c)
			if (lane < n)
				Xp[(idx_t) lane * xrs + (idx_t) (c0 + c) * xcs] = Xs[lane * XP + c];
	}
}

// ------------------------------------------------------------------------------------------------
// Many right-hand sides: invert the 128 x 128 diagonal blocks once (one workgroup per block, all blocks in
// ONE launch, MFMA recursive doubling in LDS -- lds_blocks.h) and turn every leaf of the recursion into an
// MFMA GEMM  X_k <- inv(T_kk) X_k.  The product is done IN PLACE with a tile that covers the whole aliased
// dimension (gemm.hip, GemmExtra::inplace).
// ------------------------------------------------------------------------------------------------
constexpr int TRSM_IB = LDS_NB;

template <typename T>
__global__ __launch_bounds__(LDS_NT) void trtri_diag_kernel(const T *__restrict__ Lp, idx_t lrs, idx_t lcs, int n, int unit,
							   T *__restrict__ W)
{
	// W block b: column major TRSM_IB x TRSM_IB, inverse of L[b*IB .., b*IB ..] (identity padded, zeros above)
	__shared__ T S[LDS_NB * LDS_LDP];
	const int b = blockIdx.x;
	const int r0 = b * TRSM_IB;
	const int nb = min(TRSM_IB, n - r0);
	lds_load_lower<T>(S, Lp + (idx_t) r0 * lrs + (idx_t) r0 * lcs, lrs, lcs, nb);
	__syncthreads();
	lds_tri_inv_inplace<T>(S, unit);
	lds_store_block<T>(S, W + (size_t) b * TRSM_IB * TRSM_IB, 1, TRSM_IB, TRSM_IB, false);
}

// W block b <- inverse of the b-th 128 x 128 diagonal block of the lower triangular L (identity padded)
template <typename T> void trtri_diag_dev(MatV<const T> L, bool unit, T *W)
{
	const idx_t n = L.nrows;
	if (n == 0)
		return;
	const idx_t nblk = (n + TRSM_IB - 1) / TRSM_IB;
	hipLaunchKernelGGL(trtri_diag_kernel<T>, dim3((unsigned) nblk), dim3(LDS_NT), 0, ctx().stream, L.p, L.rs, L.cs, (int) n, unit ? 1 : 0, W);
	FH_HIP(hipGetLastError());
}
template void trtri_diag_dev<double>(MatV<const double>, bool, double *);
template void trtri_diag_dev<float>(MatV<const float>, bool, float *);

template <typename T> static void trsm_inv_rec(MatV<const T> L, MatV<T> X, const T *W, idx_t b0)
{
	const idx_t n = L.nrows, k = X.ncols;
	if (n <= TRSM_IB) {
		MatV<const T> Winv{W + (size_t) b0 * TRSM_IB * TRSM_IB, n, n, 1, TRSM_IB};
		GemmExtra<T> ex;
		ex.inplace = 1; // X (the rhs operand) aliases dst; n <= 128 rows => one tile along the aliased dimension
		gemm_dev<T>(X, DST_FULL, false, Winv, X.c(), (T) 1, &ex);
		return;
	}
	const idx_t nblk = (n + TRSM_IB - 1) / TRSM_IB;
	const idx_t top = (nblk / 2) * TRSM_IB;
	MatV<T> Xt = X.sub(0, 0, top, k), Xb = X.sub(top, 0, n - top, k);
	trsm_inv_rec<T>(L.sub(0, 0, top, top), Xt, W, b0);
	gemm_dev<T>(Xb, DST_FULL, true, L.sub(top, 0, n - top, top), Xt.c(), (T) -1);
	trsm_inv_rec<T>(L.sub(top, top, n - top, n - top), Xb, W, b0 + top / TRSM_IB);
}

// X <- L^-1 X with the inverses of L's 128 x 128 diagonal blocks already in W (block i at W + i * 128 * 128,
// column major, identity padded): the Cholesky leaves produce them as a by-product (potrf.hip).
template <typename T> void trsm_lower_pre_dev(MatV<const T> L, MatV<T> X, const T *W)
{
	FH_CHECK(L.nrows == L.ncols && X.nrows == L.nrows, "trsm: shape mismatch");
	if (L.nrows == 0 || X.ncols == 0)
		return;
	trsm_inv_rec<T>(L, X, W, 0);
}

// triangular_solve.rs:200-215
static idx_t trsm_block_size(idx_t n)
{
	const idx_t base_rem = n / 2;
	idx_t r;
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

template <typename T> void trsm_lower_dev(MatV<const T> L, bool unit, MatV<T> X)
{
	FH_CHECK(L.nrows == L.ncols && X.nrows == L.nrows, "trsm: shape mismatch");
	const idx_t n = L.nrows, k = X.ncols;
	if (n == 0 || k == 0)
		return;
	if (n > 64 || k >= 64) { // explicit 128-block inverses + MFMA products; tiny solves keep the register leaf
		const idx_t nblk = (n + TRSM_IB - 1) / TRSM_IB;
		Scratch wb((size_t) nblk * TRSM_IB * TRSM_IB * sizeof(T));
		hipLaunchKernelGGL(trtri_diag_kernel<T>, dim3((unsigned) nblk), dim3(LDS_NT), 0, ctx().stream, L.p, L.rs, L.cs,
				   (int) n, unit ? 1 : 0, wb.as<T>());
		FH_HIP(hipGetLastError());
		trsm_inv_rec<T>(L, X, wb.as<T>(), 0);
		return;
	}
	if (n <= 64) {
		auto ab = [](idx_t v) { return v < 0 ? -v : v; };
		const int along_rhs = ab(X.cs) <= ab(X.rs) ? 1 : 0;
		FH_CHECK(k < (1L << 31), "trsm: too many right-hand sides");
		hipLaunchKernelGGL(trsm_leaf_kernel<T>, dim3((unsigned) ((k + 63) / 64)), dim3(64), 0, ctx().stream, L.p, L.rs,
				   L.cs, (int) n, unit ? 1 : 0, X.p, X.rs, X.cs, (int) k, along_rhs);
		FH_HIP(hipGetLastError());
		return;
	}
	// triangular_solve.rs:452-484: solve the top block, eliminate it from the bottom rows by GEMM, recurse
	const idx_t bs = trsm_block_size(n);
	MatV<T> top = X.sub(0, 0, bs, k), bot = X.sub(bs, 0, n - bs, k);
	trsm_lower_dev<T>(L.sub(0, 0, bs, bs), unit, top);
	gemm_dev<T>(bot, DST_FULL, true, L.sub(bs, 0, n - bs, bs), top.c(), (T) -1);
	trsm_lower_dev<T>(L.sub(bs, bs, n - bs, n - bs), unit, bot);
}

template <typename T> void trsm_upper_dev(MatV<const T> U, bool unit, MatV<T> X)
{
	trsm_lower_dev<T>(U.rev_rows().rev_cols(), unit, X.rev_rows());
}

template void trsm_lower_pre_dev<double>(MatV<const double>, MatV<double>, const double *);
template void trsm_lower_pre_dev<float>(MatV<const float>, MatV<float>, const float *);
template void trsm_lower_dev<double>(MatV<const double>, bool, MatV<double>);
template void trsm_lower_dev<float>(MatV<const float>, bool, MatV<float>);
template void trsm_upper_dev<double>(MatV<const double>, bool, MatV<double>);
template void trsm_upper_dev<float>(MatV<const float>, bool, MatV<float>);

} // namespace fh
