// Triangular inverse and the small elementwise helpers behind the reconstruct / inverse entry points of the
// factorizations: everything here is composition of the library's own
// structured product, TRSM machinery and gathers.
//
// triangular inverse -- faer/src/linalg/triangular_inverse.rs:43-123 (lower; the unit and upper forms are the same
// recursion on the same / the transposed views): inv([L00 0; L10 L11]) = [W00 0; -W11 L10 W00, W11].  The
// reference recurses by halves down to 1 x 1 / 2 x 2 closed forms; here the leaves are the LDS-resident 128 x 128
// MFMA inversions of trsm.hip (all diagonal blocks in ONE launch), and every level above is two structured
// products per block pair (only the lower triangles of dst are read, its strict upper triangle is never touched).
#include "common.h"
#include "lds_blocks.h"

namespace fh {

// The only consumer of explicit 128 x 128 block inverses left in the library: triangular_inverse.rs IS an explicit
// inverse.  One 512-thread workgroup per diagonal block, all blocks in one launch (lds_blocks.h: 16 x 16 blocks by
// substitution, then recursive doubling on the MFMA pipe in LDS).
template <typename T>
__global__ __launch_bounds__(LDS_NT) void trtri_diag_kernel(const T *__restrict__ Lp, idx_t lrs, idx_t lcs, int n, int unit,
							   T *__restrict__ W)
{
	// W block b: column major 128 x 128, inverse of L[b*128 .., b*128 ..] (identity padded, zeros above)
	__shared__ T S[LDS_NB * LDS_LDP];
	const int b = blockIdx.x;
	const int r0 = b * LDS_NB;
	const int nb = min(LDS_NB, n - r0);
	lds_load_lower<T>(S, Lp + (idx_t) r0 * lrs + (idx_t) r0 * lcs, lrs, lcs, nb);
	__syncthreads();
	lds_tri_inv_inplace<T>(S, unit);
	lds_store_block<T>(S, W + (size_t) b * LDS_NB * LDS_NB, 1, LDS_NB, LDS_NB, false);
}

template <typename T> static void trtri_diag_dev(MatV<const T> L, bool unit, T *W)
{
	const idx_t n = L.nrows;
	if (n == 0)
		return;
	const idx_t nblk = (n + LDS_NB - 1) / LDS_NB;
	hipLaunchKernelGGL(trtri_diag_kernel<T>, dim3((unsigned) nblk), dim3(LDS_NT), 0, ctx().stream, L.p, L.rs, L.cs, (int) n, unit ? 1 : 0, W);
	FH_HIP(hipGetLastError());
}

// dst(lower [strict]) <- W blocks: block b of the diagonal from W + b * 128 * 128 (column major 128 x 128)
template <typename T>
__global__ void place_tri_blocks_kernel(T *dst, idx_t rs, idx_t cs, int n, const T *__restrict__ W, int strict)
{
	const int b = blockIdx.x, r0 = b * 128;
	const int nb = min(128, n - r0);
	const T *src = W + (size_t) b * 128 * 128;
	for (int e = threadIdx.x; e < 128 * 128; e += blockDim.x) {
		const int i = e % 128, j = e / 128;
		if (i < nb && j < nb && (strict ? i > j : i >= j))
			dst[(idx_t) (r0 + i) * rs + (idx_t) (r0 + j) * cs] = src[e];
	}
}

template <typename T> void tri_invert_lower_dev(MatV<T> dst, MatV<const T> src, bool unit)
{
	const idx_t n = src.nrows;
	FH_CHECK(src.ncols == n && dst.nrows == n && dst.ncols == n, "triangular inverse: dimension mismatch");
	if (n == 0)
		return;
	FH_CHECK(n < (1L << 31), "triangular inverse: matrix too large");
	const idx_t nblk = (n + 127) / 128;
	Scratch wb((size_t) nblk * 128 * 128 * sizeof(T));
	trtri_diag_dev<T>(src, unit, wb.as<T>());
	hipLaunchKernelGGL(place_tri_blocks_kernel<T>, dim3((unsigned) nblk), dim3(256), 0, ctx().stream, dst.p, dst.rs, dst.cs, (int) n,
			   wb.as<const T>(), unit ? 1 : 0);
	FH_HIP(hipGetLastError());
	if (nblk == 1)
		return;
	// levels: pairs of h-wide diagonal blocks (h = 128, 256, ...): dst_bl = -dst_br * (src_bl * dst_tl)
	const int tri = unit ? 5 /* unit lower */ : 1 /* lower */;
	idx_t hmax = 128;
	while (hmax * 2 < n)
		hmax *= 2;
	Scratch tb((size_t) hmax * (size_t) hmax * sizeof(T));
	for (idx_t h = 128; h < n; h *= 2)
		for (idx_t base = 0; base + h < n; base += 2 * h) {
			const idx_t h2 = (n - base - h) < h ? (n - base - h) : h; // ragged last pair
			MatV<T> Tm{tb.as<T>(), h2, h, 1, h2};
			// T = src_bl * dst_tl        (triangular_inverse.rs:72-83; rhs structured: only its lower triangle is read)
			matmul_triangular_dev<T>(Tm, 0, false, src.sub(base + h, base, h2, h), 0, dst.sub(base, base, h, h).c(), tri, (T) 1);
			// dst_bl = -dst_br * T       (== the reference's solve with src_br, :84-86)
			matmul_triangular_dev<T>(dst.sub(base + h, base, h2, h), 0, false, dst.sub(base + h, base + h, h2, h2).c(), tri, Tm.c(), 0, (T) -1);
		}
}

// out(i, j) = L(i, j) * d(j) for i > j, out(j, j) = d(j)      (cholesky/ldlt/reconstruct.rs:33-43)
template <typename T>
__global__ void ldlt_scale_lower_kernel(T *out, idx_t ors, idx_t ocs, const T *__restrict__ L, idx_t lrs, idx_t lcs, const T *__restrict__ d,
					idx_t ds, idx_t n)
{
	const idx_t total = n * n;
	for (idx_t e = (idx_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (idx_t) gridDim.x * blockDim.x) {
		const idx_t i = e % n, j = e / n;
		if (i < j)
			continue;
		const T dj = d[j * ds];
		out[i * ors + j * ocs] = i == j ? dj : L[i * lrs + j * lcs] * dj;
	}
}

// cholesky/ldlt/inverse.rs:34-46 on W = inv(L) (strict lower part valid): W(j, j) = 1 / d(j); W(j, i) = W(i, j) / d(i), j < i
template <typename T> __global__ void ldlt_inverse_prepare_kernel(T *W, idx_t rs, idx_t cs, const T *__restrict__ d, idx_t ds, idx_t n)
{
	const idx_t total = n * n;
	for (idx_t e = (idx_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (idx_t) gridDim.x * blockDim.x) {
		const idx_t i = e % n, j = e / n; // i >= j: source element of the strict lower part / the diagonal
		if (i < j)
			continue;
		const T rinv = (T) 1 / d[i * ds];
		if (i == j)
			W[i * rs + i * cs] = rinv;
		else
			W[j * rs + i * cs] = W[i * rs + j * cs] * rinv;
	}
}

// out <- 0 everywhere, then the upper triangle (trapezoid) of R into its leading rows (qr/no_pivoting/reconstruct.rs:43-47)
template <typename T>
__global__ void zero_then_upper_kernel(T *out, idx_t ors, idx_t ocs, idx_t m, idx_t n, const T *__restrict__ R, idx_t rrs, idx_t rcs, idx_t rrows)
{
	const idx_t total = m * n;
	for (idx_t e = (idx_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (idx_t) gridDim.x * blockDim.x) {
		const idx_t i = e % m, j = e / m;
		out[i * ors + j * ocs] = (R && i < rrows && i <= j) ? R[i * rrs + j * rcs] : (T) 0;
	}
}

template <typename T> void ldlt_scale_lower_dev(MatV<T> out, MatV<const T> L, const T *d, idx_t ds)
{
	const idx_t n = L.nrows;
	if (n == 0)
		return;
	hipLaunchKernelGGL(ldlt_scale_lower_kernel<T>, dim3(1024), dim3(256), 0, ctx().stream, out.p, out.rs, out.cs, L.p, L.rs, L.cs, d, ds, n);
	FH_HIP(hipGetLastError());
}
template <typename T> void ldlt_inverse_prepare_dev(MatV<T> W, const T *d, idx_t ds)
{
	if (W.nrows == 0)
		return;
	hipLaunchKernelGGL(ldlt_inverse_prepare_kernel<T>, dim3(1024), dim3(256), 0, ctx().stream, W.p, W.rs, W.cs, d, ds, W.nrows);
	FH_HIP(hipGetLastError());
}
template <typename T> void zero_then_upper_dev(MatV<T> out, const MatV<const T> *R)
{
	if (out.nrows == 0 || out.ncols == 0)
		return;
	hipLaunchKernelGGL(zero_then_upper_kernel<T>, dim3(1024), dim3(256), 0, ctx().stream, out.p, out.rs, out.cs, out.nrows, out.ncols,
			   R ? R->p : (const T *) nullptr, R ? R->rs : 0, R ? R->cs : 0, R ? R->nrows : 0);
	FH_HIP(hipGetLastError());
}

#define FH_INST(T)                                                                                                     \
	template void tri_invert_lower_dev<T>(MatV<T>, MatV<const T>, bool);                                           \
	template void ldlt_scale_lower_dev<T>(MatV<T>, MatV<const T>, const T *, idx_t);                               \
	template void ldlt_inverse_prepare_dev<T>(MatV<T>, const T *, idx_t);                                          \
	template void zero_then_upper_dev<T>(MatV<T>, const MatV<const T> *);
FH_INST(double)
FH_INST(float)
#undef FH_INST

} // namespace fh
