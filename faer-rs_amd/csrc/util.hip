// Strided copy kernels (device <-> device views).
#include "common.h"

namespace fh {

template <typename T>
__global__ void copy_kernel(T *d, idx_t drs, idx_t dcs, const T *s, idx_t srs, idx_t scs, idx_t M, idx_t N)
{
	const idx_t total = M * N;
	for (idx_t e = (idx_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (idx_t) gridDim.x * blockDim.x) {
		const idx_t i = e % M, j = e / M;
		d[i * drs + j * dcs] = s[i * srs + j * scs];
	}
}

template <typename T> void copy_dev(MatV<T> dst, MatV<const T> src)
{
	FH_CHECK(dst.nrows == src.nrows && dst.ncols == src.ncols, "copy: shape mismatch");
	if (dst.nrows == 0 || dst.ncols == 0)
		return;
	auto ab = [](idx_t x) { return x < 0 ? -x : x; };
	if (ab(dst.cs) < ab(dst.rs)) { // fast index along the smaller dst stride
		dst = dst.t();
		src = src.t();
	}
	const idx_t total = dst.nrows * dst.ncols;
	idx_t blocks = (total + 255) / 256;
	if (blocks > 65536)
		blocks = 65536;
	hipLaunchKernelGGL(copy_kernel<T>, dim3((unsigned) blocks), dim3(256), 0, ctx().stream, dst.p, dst.rs, dst.cs, src.p,
			   src.rs, src.cs, dst.nrows, dst.ncols);
	FH_HIP(hipGetLastError());
}

// dst[i, j] = src[perm[i], j]   (perm in device memory)
template <typename T>
__global__ void gather_rows_perm_kernel(T *d, idx_t drs, idx_t dcs, const T *s, idx_t srs, idx_t scs, idx_t M, idx_t N,
					const idx_t *__restrict__ perm)
{
	const idx_t total = M * N;
	for (idx_t e = (idx_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (idx_t) gridDim.x * blockDim.x) {
		const idx_t i = e % M, j = e / M;
		d[i * drs + j * dcs] = s[perm[i] * srs + j * scs];
	}
}

template <typename T> void gather_rows_dev(MatV<T> dst, MatV<const T> src, const idx_t *perm_dev)
{
	FH_CHECK(dst.nrows == src.nrows && dst.ncols == src.ncols, "gather_rows: shape mismatch");
	if (dst.nrows == 0 || dst.ncols == 0)
		return;
	const idx_t total = dst.nrows * dst.ncols;
	idx_t blocks = (total + 255) / 256;
	if (blocks > 65536)
		blocks = 65536;
	hipLaunchKernelGGL(gather_rows_perm_kernel<T>, dim3((unsigned) blocks), dim3(256), 0, ctx().stream, dst.p, dst.rs, dst.cs, src.p,
			   src.rs, src.cs, dst.nrows, dst.ncols, perm_dev);
	FH_HIP(hipGetLastError());
}

// X[i, :] <- X[i, :] / d[i]   (cholesky/ldlt/solve.rs:31-41: multiplication by the reciprocal)
template <typename T> __global__ void scale_rows_recip_kernel(T *X, idx_t rs, idx_t cs, idx_t M, idx_t N, const T *__restrict__ d, idx_t ds)
{
	const idx_t total = M * N;
	for (idx_t e = (idx_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (idx_t) gridDim.x * blockDim.x) {
		const idx_t i = e % M, j = e / M;
		X[i * rs + j * cs] *= (T) 1 / d[i * ds];
	}
}

template <typename T> void scale_rows_recip_dev(MatV<T> X, const T *d, idx_t ds)
{
	if (X.nrows == 0 || X.ncols == 0)
		return;
	const idx_t total = X.nrows * X.ncols;
	idx_t blocks = (total + 255) / 256;
	if (blocks > 65536)
		blocks = 65536;
	hipLaunchKernelGGL(scale_rows_recip_kernel<T>, dim3((unsigned) blocks), dim3(256), 0, ctx().stream, X.p, X.rs, X.cs, X.nrows, X.ncols, d,
			   ds);
	FH_HIP(hipGetLastError());
}

template void scale_rows_recip_dev<double>(MatV<double>, const double *, idx_t);
template void scale_rows_recip_dev<float>(MatV<float>, const float *, idx_t);
template void gather_rows_dev<double>(MatV<double>, MatV<const double>, const idx_t *);
template void gather_rows_dev<float>(MatV<float>, MatV<const float>, const idx_t *);
template void copy_dev<double>(MatV<double>, MatV<const double>);
template void copy_dev<float>(MatV<float>, MatV<const float>);
template void copy_dev<long>(MatV<long>, MatV<const long>);

} // namespace fh
