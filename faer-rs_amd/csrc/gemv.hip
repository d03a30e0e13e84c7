// Level-2 shapes of the product for gfx950: matrix-vector (N == 1 or M == 1) and rank-1 update (K == 1).
//
// Replaces faer/src/linalg/matmul/mod.rs:757-873 (matvec_rowmajor), :874-1034 (matvec_colmajor) and :1035-1175
// (rank_update) -- SURVEY.md section 8a rows a7-a9 -- which matmul_imp (:1176-1310) selects for these shapes
// before it reaches the GEMM.  They are pure HBM streams (A is read once, 2 flop per element), so they do not
// go through the MFMA tile kernel: lanes run along whichever index of A has the unit stride, every lane keeps
// several independent loads in flight, partial sums meet in LDS, and a long reduction dimension is split over
// workgroups whose partial sums are added in a fixed order by a second small kernel (deterministic).
//   y <- [y +] alpha * A x,  A m x k:   unit row stride  -> gemv_mn_kernel (lanes along m, workgroups over m x k slices)
//                                       unit col stride  -> gemv_k_kernel  (one wave per row, lanes along k)
//   C <- [C +] alpha * a b^T         -> rank1_kernel (lanes along the unit stride of C)
// Algorithmic bytes: m k sizeof(T) (A) for the matrix-vector shapes, 2 m n sizeof(T) for the accumulating rank-1
// update; roofline = HBM.
#include "common.h"

namespace fh {

static inline idx_t iabs2(idx_t x) { return x < 0 ? -x : x; }

// lanes along m (A[i, p] at a[i * ars + p * acs], |ars| == 1): block = 256 threads = 256 rows, grid.y slices of k
template <typename T>
__global__ __launch_bounds__(256) void gemv_mn_kernel(int m, int k, const T *__restrict__ a, idx_t ars, idx_t acs,
						      const T *__restrict__ x, idx_t xs, T *y, idx_t ys, T alpha, int add, int k_per_slice,
						      int atomic, T *part)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	const int k0 = blockIdx.y * k_per_slice, k1 = min(k, k0 + k_per_slice);
	if (i >= m)
		return;
	const T *ap = a + (idx_t) i * ars;
	T acc[4] = {0, 0, 0, 0};
	int p = k0;
	for (; p + 4 <= k1; p += 4) {
#pragma unroll
		for (int u = 0; u < 4; ++u)
			acc[u] = fh_fma(ap[(idx_t) (p + u) * acs], x[(idx_t) (p + u) * xs], acc[u]);
	}
	for (; p < k1; ++p)
		acc[0] = fh_fma(ap[(idx_t) p * acs], x[(idx_t) p * xs], acc[0]);
	const T s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
	T *yp = y + (idx_t) i * ys;
	if (atomic)
		part[(size_t) blockIdx.y * m + i] = s; // slices are added in index order by gemv_reduce_kernel
	else if (add)
		*yp = fh_fma(alpha, s, *yp);
	else
		*yp = alpha * s;
}

// lanes along k (|acs| == 1): one wave per row, 4 rows per workgroup, grid.y slices of k
template <typename T>
__global__ __launch_bounds__(256) void gemv_k_kernel(int m, int k, const T *__restrict__ a, idx_t ars, idx_t acs,
						     const T *__restrict__ x, idx_t xs, T *y, idx_t ys, T alpha, int add, int k_per_slice,
						     int atomic, T *part)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int i = blockIdx.x * 4 + wave;
	const int k0 = blockIdx.y * k_per_slice, k1 = min(k, k0 + k_per_slice);
	if (i >= m)
		return;
	const T *ap = a + (idx_t) i * ars;
	T acc[4] = {0, 0, 0, 0};
	int p = k0 + lane;
	for (; p + 192 < k1; p += 256) {
#pragma unroll
		for (int u = 0; u < 4; ++u)
			acc[u] = fh_fma(ap[(idx_t) (p + 64 * u) * acs], x[(idx_t) (p + 64 * u) * xs], acc[u]);
	}
	for (; p < k1; p += 64)
		acc[0] = fh_fma(ap[(idx_t) p * acs], x[(idx_t) p * xs], acc[0]);
	T s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1)
		s += __shfl_xor(s, off, 64);
	if (lane == 0) {
		T *yp = y + (idx_t) i * ys;
		if (atomic)
			part[(size_t) blockIdx.y * m + i] = s;
		else if (add)
			*yp = fh_fma(alpha, s, *yp);
		else
			*yp = alpha * s;
	}
}

// y_i <- [y_i +] alpha * sum_z part[z][i]: 16 fixed slice segments per element, combined in segment order
// (deterministic split reduction, same scheme as splitk_reduce_kernel)
template <typename T>
__global__ __launch_bounds__(256) void gemv_reduce_kernel(int m, int slices, const T *__restrict__ part, T *y, idx_t ys, T alpha, int add)
{
	__shared__ T sp[16][17];
	const int le = threadIdx.x & 15, seg = threadIdx.x >> 4;
	const int i = blockIdx.x * 16 + le;
	const int zs = (slices + 15) / 16;
	const int z0 = seg * zs, z1 = min(slices, z0 + zs);
	T s = (T) 0;
	if (i < m)
		for (int z = z0; z < z1; ++z)
			s += part[(size_t) z * m + i];
	sp[seg][le] = s;
	__syncthreads();
	if (seg == 0 && i < m) {
		T sum = (T) 0;
#pragma unroll
		for (int k = 0; k < 16; ++k)
			sum += sp[k][le];
		T *yp = y + (idx_t) i * ys;
		*yp = add ? fh_fma(alpha, sum, *yp) : alpha * sum;
	}
}

// C[i, j] <- [C +] alpha * a_i * b_j, lanes along i (the caller passes the view whose row stride is the small one)
template <typename T>
__global__ __launch_bounds__(256) void rank1_kernel(int m, int n, T *c, idx_t crs, idx_t ccs, const T *__restrict__ a, idx_t as,
						    const T *__restrict__ b, idx_t bs, T alpha, int add)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= m)
		return;
	const T ai = alpha * a[(idx_t) i * as];
	for (int j = blockIdx.y; j < n; j += gridDim.y) {
		T *p = c + (idx_t) i * crs + (idx_t) j * ccs;
		const T bj = b[(idx_t) j * bs];
		*p = add ? fh_fma(ai, bj, *p) : ai * bj;
	}
}

// y <- [y +] alpha * A x; returns false when neither stride of A is the unit one (the caller falls back to the GEMM)
template <typename T> bool gemv_dev(idx_t m, idx_t k, MatV<const T> A, const T *x, idx_t xs, T *y, idx_t ys, T alpha, bool add)
{
	if (m >= (1L << 31) || k >= (1L << 31))
		return false;
	const bool mn = iabs2(A.rs) == 1, km = iabs2(A.cs) == 1;
	if (!mn && !km)
		return false;
	hipStream_t s = ctx().stream;
	// split the reduction dimension when the rows alone cannot fill the chip
	const idx_t row_blocks = mn ? (m + 255) / 256 : (m + 3) / 4;
	idx_t slices = 1;
	if (row_blocks < 1024 && k >= 4096) {
		slices = (2048 + row_blocks - 1) / row_blocks;
		if (slices > k / 1024)
			slices = k / 1024;
		if (slices < 1)
			slices = 1;
		if (slices > 65535)
			slices = 65535;
	}
	idx_t kps = (k + slices - 1) / slices;
	kps = (kps + 255) / 256 * 256;
	slices = (k + kps - 1) / kps;
	const int atomic = slices > 1 ? 1 : 0;
	Scratch partb(atomic ? (size_t) slices * (size_t) m * sizeof(T) : 256);
	dim3 grid((unsigned) row_blocks, (unsigned) slices);
	if (mn)
		hipLaunchKernelGGL(gemv_mn_kernel<T>, grid, dim3(256), 0, s, (int) m, (int) k, A.p, A.rs, A.cs, x, xs, y, ys, alpha, add ? 1 : 0,
				   (int) kps, atomic, partb.as<T>());
	else
		hipLaunchKernelGGL(gemv_k_kernel<T>, grid, dim3(256), 0, s, (int) m, (int) k, A.p, A.rs, A.cs, x, xs, y, ys, alpha, add ? 1 : 0,
				   (int) kps, atomic, partb.as<T>());
	if (atomic)
		hipLaunchKernelGGL(gemv_reduce_kernel<T>, dim3((unsigned) ((m + 15) / 16)), dim3(256), 0, s, (int) m, (int) slices, partb.as<T>(),
				   y, ys, alpha, add ? 1 : 0);
	FH_HIP(hipGetLastError());
	return true;
}

// C <- [C +] alpha * a b^T  (K == 1)
template <typename T> void rank1_dev(MatV<T> C, bool add, const T *a, idx_t as, const T *b, idx_t bs, T alpha)
{
	if (iabs2(C.cs) < iabs2(C.rs)) { // lanes along the smaller stride of C
		C = C.t();
		std::swap(a, b);
		std::swap(as, bs);
	}
	const idx_t m = C.nrows, n = C.ncols;
	dim3 grid((unsigned) ((m + 255) / 256), (unsigned) (n < 4096 ? n : 4096));
	hipLaunchKernelGGL(rank1_kernel<T>, grid, dim3(256), 0, ctx().stream, (int) m, (int) n, C.p, C.rs, C.cs, a, as, b, bs, alpha,
			   add ? 1 : 0);
	FH_HIP(hipGetLastError());
}

template bool gemv_dev<double>(idx_t, idx_t, MatV<const double>, const double *, idx_t, double *, idx_t, double, bool);
template bool gemv_dev<float>(idx_t, idx_t, MatV<const float>, const float *, idx_t, float *, idx_t, float, bool);
template void rank1_dev<double>(MatV<double>, bool, const double *, idx_t, const double *, idx_t, double);
template void rank1_dev<float>(MatV<float>, bool, const float *, idx_t, const float *, idx_t, float);

} // namespace fh
