// In-place lower Cholesky (LLT) for gfx950.
//
// Replaces faer/src/linalg/cholesky/llt/factor.rs:67-97 and the engine behind it,
// cholesky/ldlt/factor.rs:7-498.
//
// The reference is right-looking with 128-column steps (factor.rs:392).  A rank-128 fp64 update is only
// ~16 flop/byte -- at the MI355X machine balance -- so the GPU driver recurses by HALVES instead:
//     A00 = L00 L00^T (recurse) ; A10 <- A10 L00^-T (TRSM) ; A11 -= A10 A10^T (MFMA SYRK, K = n/2) ; recurse A11
// which performs the same arithmetic per entry but moves O(n^2 log n) bytes instead of O(n^3 / 128).
// Leaves (n <= 128) are factored by ONE 1024-thread workgroup with the block resident in LDS; it follows
// the reference's base kernel semantics exactly (factor.rs:122-174): d = a_jj (optionally regularised),
// fail if !(d > 0), column scaled by the reciprocal 1/sqrt(d) (so l_jj = a_jj * (1/sqrt(d))).
// Failure is reported through a device status word (first failing GLOBAL column index + 1); later
// kernels see it and become no-ops, so the host synchronises exactly once per factorization.
#include "common.h"

namespace fh {

constexpr int POTRF_NB = 128;

template <typename T>
__global__ __launch_bounds__(1024) void potrf_leaf_kernel(T *A, idx_t rs, idx_t cs, int n, int regularize, T eps, T delta,
							   int *status, int offset)
{
	constexpr int LD = POTRF_NB;
	__shared__ T S[LD * LD]; // column major working copy of the lower triangle
	const int tid = threadIdx.x;
	if (status[0] != 0)
		return; // an earlier block already failed
	for (int e = tid; e < n * n; e += 1024) {
		const int i = e % n, k = e / n;
		if (i >= k)
			S[k * LD + i] = A[(idx_t) i * rs + (idx_t) k * cs];
	}
	__syncthreads();
	const int lane = tid & 63, wave = tid >> 6;
	int count = 0;
	for (int j = 0; j < n; ++j) {
		T d = S[j * LD + j];
		if (regularize && d <= eps) { // cholesky/ldlt/factor.rs:122-131 (llt: sign == +1)
			d = delta;
			++count;
		}
		bool bad = !(d > (T) 0);
		const T sq = sqrt(d);
		bad = bad || sq == (T) 0 || !isfinite(sq);
		if (bad) { // uniform: every thread read the same d
			if (tid == 0)
				atomicCAS(status, 0, offset + j + 1);
			break;
		}
		const T inv = (T) 1 / sq;
		// final column j of L (diagonal included: l_jj = a_jj * inv, factor.rs:160-174)
		if (tid < n - j) {
			const int i = j + tid;
			A[(idx_t) i * rs + (idx_t) j * cs] = S[j * LD + i] * inv;
		}
		// trailing update: column k (one wave per column, lanes along rows)
		for (int k = j + 1 + wave; k < n; k += 16) {
			const T lkj = S[j * LD + k] * inv;
			for (int i = k + lane; i < n; i += 64)
				S[k * LD + i] = S[k * LD + i] - (S[j * LD + i] * inv) * lkj;
		}
		__syncthreads();
	}
	if (tid == 0 && count > 0)
		atomicAdd(status + 1, count);
}

template <typename T> static void potrf_rec(MatV<T> A, int regularize, T eps, T delta, int *status, idx_t offset)
{
	const idx_t n = A.nrows;
	if (n == 0)
		return;
	if (n <= POTRF_NB) {
		hipLaunchKernelGGL(potrf_leaf_kernel<T>, dim3(1), dim3(1024), 0, ctx().stream, A.p, A.rs, A.cs, (int) n, regularize,
				   eps, delta, status, (int) offset);
		FH_HIP(hipGetLastError());
		return;
	}
	const idx_t h = ((n / 2 + POTRF_NB - 1) / POTRF_NB) * POTRF_NB;
	MatV<T> A00 = A.sub(0, 0, h, h), A10 = A.sub(h, 0, n - h, h), A11 = A.sub(h, h, n - h, n - h);
	potrf_rec<T>(A00, regularize, eps, delta, status, offset);
	// A10 <- A10 L00^-T, expressed like the reference (cholesky/ldlt/factor.rs:422-426) as L00 \ A10^T
	trsm_lower_dev<T>(A00.c(), false, A10.t());
	// lower(A11) -= A10 A10^T  (cholesky/ldlt/factor.rs:436-446 -> triangular.rs:602 DstKind::Lower)
	gemm_dev<T>(A11, DST_LOWER, true, A10.c(), A10.t().c(), (T) -1);
	potrf_rec<T>(A11, regularize, eps, delta, status, offset + h);
}

template <typename T> long potrf_lower_dev(MatV<T> A, T reg_delta, T reg_eps)
{
	FH_CHECK(A.nrows == A.ncols, "potrf: matrix must be square");
	FH_CHECK(A.nrows < (1L << 30), "potrf: matrix too large");
	if (A.nrows == 0)
		return 0;
	Scratch st(64);
	int *status = st.as<int>();
	FH_HIP(hipMemsetAsync(status, 0, 64, ctx().stream));
	const int regularize = (reg_delta > (T) 0 && reg_eps > (T) 0) ? 1 : 0; // cholesky/llt/factor.rs:85-86
	potrf_rec<T>(A, regularize, reg_eps, reg_delta, status, 0);
	int h[2] = {0, 0};
	FH_HIP(hipMemcpyAsync(h, status, sizeof(h), hipMemcpyDeviceToHost, ctx().stream));
	ctx().sync();
	if (h[0] != 0)
		return -(long) h[0]; // -(index + 1)
	return (long) h[1];
}

template long potrf_lower_dev<double>(MatV<double>, double, double);
template long potrf_lower_dev<float>(MatV<float>, float, float);

} // namespace fh
